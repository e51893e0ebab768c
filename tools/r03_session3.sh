#!/bin/bash
# round 3, GPU session 3: full -m gpu suite, config 5 with the fused last pass, stream-mode (1998,1512) kernel stats, layered timing
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r03s3; rm -rf $O; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/summary.txt
tail -12 $O/pytest.log
timeout -k 10 200 python bench.py --workload wrcq_dvbs2 --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_wrcq.json 2> $O/bench_wrcq.err; echo "bench wrcq rc=$?" | tee -a $O/summary.txt
cut -c1-400 $O/bench_wrcq.json
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_wrcq -- python3 bench.py --workload wrcq_dvbs2 --steps 5 --warmup 2 --no-cpu-baseline --no-legs > $O/stats_wrcq.log 2>&1; echo "stats wrcq rc=$?" | tee -a $O/summary.txt
export LDPC_ENGINE_MODE=stream
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_basic_stream -- python3 bench.py --workload basic --steps 3 --warmup 1 --no-cpu-baseline --no-legs --no-stream-leg > $O/stats_basic_stream.log 2>&1; echo "stats basic stream rc=$?" | tee -a $O/summary.txt
unset LDPC_ENGINE_MODE
timeout -k 10 120 python tools/time_layered.py > $O/layered.jsonl 2> $O/layered.err; echo "layered rc=$?" | tee -a $O/summary.txt
cat $O/layered.jsonl
for d in stats_wrcq stats_basic_stream; do f=$(find $O/$d -name '*kernel_stats.csv' | head -1); echo "== $d"; head -14 "$f" | cut -c1-200; done
