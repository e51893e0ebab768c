#!/bin/bash
# resident kernel at a 64-VGPR budget (8 waves per SIMD): 1024-thread workgroups, two per CU (build w8.so, probes build so
# LDPC_RESIDENT_NT applies)
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/nt2; mkdir -p $O
for w in basic rcq neural2d; do
  timeout -k 10 200 python tools/time_sweeps.py --workload $w --tag default >> $O/time.jsonl 2>> $O/time.err
  for nt in 512 768 1024; do
    LDPC_HIP_LIB=$PWD/build_variants/w8.so LDPC_RESIDENT_NT=$nt timeout -k 10 200 python tools/time_sweeps.py --workload $w --tag w8_nt$nt >> $O/time.jsonl 2>> $O/time.err
  done
done
python - <<'PY'
import json
for l in open("gpurun_out/nt2/time.jsonl"):
    d = json.loads(l); print(d["tag"], d["workload"], round(d["decode_ms"], 3), d["engine"]["threads_per_workgroup"])
PY
