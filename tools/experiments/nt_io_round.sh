#!/bin/bash
# resident kernel: non-temporal LLR loads / decision stores (default) vs plain (nont.so) vs the round-start binary
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/ntio1; mkdir -p $O
for lib in default nont old default nont old; do
  if [ $lib = default ]; then unset LDPC_HIP_LIB; else export LDPC_HIP_LIB=$PWD/build_variants/$lib.so; fi
  for w in basic rcq neural2d; do
    timeout -k 10 200 python tools/time_sweeps.py --workload $w --tag $lib >> $O/time.jsonl 2>> $O/time.err
  done
done
unset LDPC_HIP_LIB
python - <<'PY'
import json
for l in open("gpurun_out/ntio1/time.jsonl"):
    d = json.loads(l); print(d["tag"], d["workload"], round(d["decode_ms"], 3))
PY
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -x -q > $O/pytest.log 2>&1; echo "rc=$?" >> $O/pytest.log; tail -3 $O/pytest.log
