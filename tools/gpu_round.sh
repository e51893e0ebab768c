#!/bin/bash
# One GPU-box session: full-size tests, bench lines for every workload, rocprofv3 stats and
# PMC passes (separate runs for FETCH_SIZE / WRITE_SIZE).  Outputs under gpurun_out/.
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/round6; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -q > $O/pytest_fullsize.log 2>&1; echo "pytest rc=$?" >> $O/pytest_fullsize.log
for w in basic neural2d rcq wrcq_dvbs2; do
  timeout -k 10 300 python bench.py --workload $w --steps 10 --warmup 3 > $O/bench_$w.json 2> $O/bench_$w.err || echo "bench $w failed" >> $O/errors.log
done
# 2-rank rehearsal of the N > 1 path on this one GPU (gloo stands in for RCCL, both ranks on cuda:0)
LDPC_BENCH_BACKEND=gloo timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 3 --warmup 1 --batch 8192 --no-cpu-baseline --no-stream-leg > $O/bench_2rank_gloo.json 2> $O/bench_2rank_gloo.err || echo "2-rank rehearsal failed" >> $O/errors.log
for w in basic rcq; do
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_$w -- python bench.py --workload $w --steps 5 --warmup 2 --no-cpu-baseline --no-stream-leg > $O/stats_$w.log 2>&1 || echo "stats $w failed" >> $O/errors.log
  timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/fetch_$w -- python bench.py --workload $w --steps 2 --warmup 1 --no-cpu-baseline --no-stream-leg --sweep-reps 3 > $O/fetch_$w.log 2>&1 || echo "fetch $w failed" >> $O/errors.log
  timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/write_$w -- python bench.py --workload $w --steps 2 --warmup 1 --no-cpu-baseline --no-stream-leg --sweep-reps 3 > $O/write_$w.log 2>&1 || echo "write $w failed" >> $O/errors.log
done
tail -3 $O/pytest_fullsize.log; cat $O/bench_*.json | cut -c1-400; cat $O/errors.log 2>/dev/null
