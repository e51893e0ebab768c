#!/bin/bash
# cn_sweep_f4 (register-held fp32 CN sweep): parity on the streaming forms + same-box A/B against the previous build
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r03cnf4; rm -rf $O; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_training.py -m gpu -x -q -k "not bench" > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/summary.txt
tail -4 $O/pytest.log
for rep in 1 2; do
  for lib in pre_cnf4 default; do
    if [ $lib = default ]; then unset LDPC_HIP_LIB; else export LDPC_HIP_LIB=$PWD/build_variants/$lib.so; fi
    for w in basic neural2d; do
      timeout -k 10 200 python tools/time_sweeps.py --workload $w --mode stream --tag $lib 2>> $O/time.err | cut -c1-400 | tee -a $O/time.jsonl
    done
  done
done
unset LDPC_HIP_LIB
