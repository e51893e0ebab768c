#!/bin/bash
# cn_gather: one 48-byte record per edge + next-group scalar prefetch + alpha row in LDS (default) vs the
# two-array metadata of the previous build (gat_old.so)
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/gatrec; mkdir -p $O
for lib in default gat_old default gat_old; do
  if [ $lib = default ]; then unset LDPC_HIP_LIB; else export LDPC_HIP_LIB=$PWD/build_variants/$lib.so; fi
  timeout -k 10 200 python tools/time_sweeps.py --workload wrcq_dvbs2 --mode stream --tag $lib >> $O/time.jsonl 2>> $O/time.err
done
unset LDPC_HIP_LIB
python - <<'PY'
import json
for l in open("gpurun_out/gatrec/time.jsonl"):
    d = json.loads(l); print(d["tag"], d["workload"], round(d["decode_ms"], 3))
PY
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q > $O/pytest.log 2>&1; echo "rc=$?" >> $O/pytest.log; tail -3 $O/pytest.log
