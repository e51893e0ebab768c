#!/bin/bash
# resident kernel: variable-phase plan loads unconditional with clamped indices (the prefetch stays in flight) vs the previous build
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r03plan; mkdir -p $O
for lib in pre_plan default pre_plan default; do
  if [ $lib = default ]; then unset LDPC_HIP_LIB; else export LDPC_HIP_LIB=$PWD/build_variants/$lib.so; fi
  for w in basic neural2d rcq basic_f64; do
    timeout -k 10 200 python tools/time_sweeps.py --workload $w --tag $lib 2>> $O/time.err | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print(d['tag'], d['workload'], 'decode_ms', round(d['decode_ms'], 4), 'Mcw/s', round(d['Mcw_s'], 2))" | tee -a $O/time.txt
  done
done
unset LDPC_HIP_LIB
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "auto" > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
