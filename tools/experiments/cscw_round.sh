#!/bin/bash
# vn_sweep_q4 writing its V2C codes in CSC (variable-major) order: contiguous writes instead of scattered rows (timing only)
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/cscw; rm -rf $O; mkdir -p $O
for lib in default cscw default cscw; do
  if [ $lib = default ]; then unset LDPC_HIP_LIB; else export LDPC_HIP_LIB=$PWD/build_variants/$lib.so; fi
  timeout -k 10 200 python tools/time_sweeps.py --workload wrcq_dvbs2 --mode pair --tag $lib >> $O/time.jsonl 2>> $O/time.err
done
unset LDPC_HIP_LIB
python - <<'PY'
import json
for l in open("gpurun_out/cscw/time.jsonl"):
    d = json.loads(l); print(d["tag"], d["workload"], round(d["decode_ms"], 3), round(d.get("cn_ms", 0), 4), round(d.get("vn_ms", 0), 4))
PY
