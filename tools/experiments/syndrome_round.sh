#!/bin/bash
# syndrome_latch with grouped loads: kernel time in config 5 (rocprofv3 stats), early-stop batched decode on the streaming engine
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/syndrome; rm -rf $O; mkdir -p $O
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 bench.py --workload wrcq_dvbs2 --steps 5 --warmup 2 --no-cpu-baseline --no-legs > $O/stats.log 2>&1
f=$(find $O/stats -name '*kernel_stats.csv' | head -1); grep -E "syndrome" "$f" | cut -c1-140
grep -o '"ms_per_step": [0-9.]*' $O/stats.log | head -1
timeout -k 10 300 python bench.py --workload wrcq_dvbs2 --early-stop --steps 5 --warmup 2 --no-cpu-baseline --no-legs > $O/es.json 2> $O/es.err; python3 -c "import json;d=json.load(open('$O/es.json'));print('early-stop decode', round(d['ms_per_step'],3))"
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -x -q -m gpu -k "not bench and (stream or sweeps or gather or pair)" > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -2 $O/pytest.log
