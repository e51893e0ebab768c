#!/bin/bash
# boundary kernels (vn_last_rows, transpose_in_q4): XCD-contiguous chunk mapping vs plain order, non-temporal vs temporal row stores
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r03xcd; rm -rf $O; mkdir -p $O
for lib in rows_xcd0 default rows_nts0; do
  if [ $lib = default ]; then unset LDPC_HIP_LIB; else export LDPC_HIP_LIB=$PWD/build_variants/$lib.so; fi
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_$lib -- python3 bench.py --workload wrcq_dvbs2 --steps 8 --warmup 3 --no-cpu-baseline --no-legs > $O/stats_$lib.log 2>&1
  f=$(find $O/stats_$lib -name '*kernel_stats.csv' | head -1); echo "== $lib"; grep -E "vn_last_rows|transpose_in_q4" "$f" | sed 's/(ldpc::GraphDev[^"]*"//' | cut -c1-140
  grep -o '"ms_per_step": [0-9.]*' $O/stats_$lib.log | head -1
done
unset LDPC_HIP_LIB
timeout -k 10 800 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -m gpu -x -q -k "not bench and (stream or sweeps or gather)" > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
