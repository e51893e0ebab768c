#!/bin/bash
# dispatch forms of the resident kernel again, on top of the L1-resident plan
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/modes2; mkdir -p $O
for lib in default c1v0 c2v0 c0v1 default c1v0; do
  if [ $lib = default ]; then unset LDPC_HIP_LIB; else export LDPC_HIP_LIB=$PWD/build_variants/$lib.so; fi
  for w in basic rcq; do
    timeout -k 10 200 python tools/time_sweeps.py --workload $w --tag $lib >> $O/time.jsonl 2>> $O/time.err
  done
done
unset LDPC_HIP_LIB
python - <<'PY'
import json
for l in open("gpurun_out/modes2/time.jsonl"):
    d = json.loads(l); print(d["tag"], d["workload"], round(d["decode_ms"], 3))
PY
