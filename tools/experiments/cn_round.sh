#!/bin/bash
# streaming CN sweep with grouped loads: default (groups of 4) vs groups of 8 / 14 vs the round-start binary
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/cn1; mkdir -p $O
for lib in default cnu8 cnu14 old default; do
  if [ $lib = default ]; then unset LDPC_HIP_LIB; else export LDPC_HIP_LIB=$PWD/build_variants/$lib.so; fi
  for w in basic neural2d; do
    timeout -k 10 200 python tools/time_sweeps.py --workload $w --mode stream --tag $lib >> $O/time.jsonl 2>> $O/time.err
  done
done
unset LDPC_HIP_LIB
python - <<'PY'
import json
for l in open("gpurun_out/cn1/time.jsonl"):
    d = json.loads(l); print(d["tag"], d["workload"], "decode", round(d["decode_ms"], 3), "cn", round(d["cn_ms"], 4), round(d["cn_GBs"]), "vn", round(d["vn_ms"], 4), round(d["vn_GBs"]))
PY
