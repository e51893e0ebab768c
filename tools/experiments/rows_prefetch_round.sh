#!/bin/bash
# vn_last_rows with the lane-parallel index prefetch vs the previous build (config 5 bench line), + streaming-form parity tests
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r03rows; mkdir -p $O
for lib in pre_plan default pre_plan default; do
  if [ $lib = default ]; then unset LDPC_HIP_LIB; else export LDPC_HIP_LIB=$PWD/build_variants/$lib.so; fi
  timeout -k 10 200 python bench.py --workload wrcq_dvbs2 --steps 10 --warmup 3 --no-cpu-baseline 2>> $O/bench.err | python -c "
import sys, json
d = json.loads(sys.stdin.readline()); print('$lib', 'ms_per_step', round(d['ms_per_step'], 3), 'Mcw/s', round(d['value'] / 1e6, 4))" | tee -a $O/time.txt
done
unset LDPC_HIP_LIB
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_wrcq -- python3 bench.py --workload wrcq_dvbs2 --steps 8 --warmup 3 --no-cpu-baseline --no-legs > $O/stats_wrcq.log 2>&1
f=$(find $O/stats_wrcq -name '*kernel_stats.csv' | head -1); grep -E "vn_last_rows|transpose_in_q4" "$f" | cut -c1-60,200-330
timeout -k 10 800 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -m gpu -x -q -k "not bench and (stream or sweeps or gather)" > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
