#!/bin/bash
# round 3, GPU session 11: layered kernel A/B (in-order vs early-read form, both with the lean tail, packed regions) + full layered tests
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r03s11; rm -rf $O; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "layered or workspace_cache" > $O/pytest_layered.log 2>&1; echo "pytest layered rc=$?" | tee -a $O/summary.txt
tail -3 $O/pytest_layered.log
for rep in 1 2; do
  for lib in default lay_early; do
    if [ $lib = default ]; then unset LDPC_HIP_LIB; else export LDPC_HIP_LIB=$PWD/build_variants/$lib.so; fi
    echo "== $lib" | tee -a $O/layered_ab.txt
    timeout -k 10 120 python tools/time_layered.py 2>> $O/layered.err | tee -a $O/layered_ab.txt
  done
done
unset LDPC_HIP_LIB
