#!/bin/bash
# vn_sweep_q4: consecutive variables per wave (default build = 4; vpw1 / vpw2 / vpw8 variants)
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/vpw1; mkdir -p $O
for lib in default vpw1 vpw2 vpw8 default vpw1; do
  if [ $lib = default ]; then unset LDPC_HIP_LIB; else export LDPC_HIP_LIB=$PWD/build_variants/$lib.so; fi
  for w in wrcq_dvbs2 rcq; do
    timeout -k 10 200 python tools/time_sweeps.py --workload $w --mode pair --tag $lib >> $O/time.jsonl 2>> $O/time.err
  done
done
unset LDPC_HIP_LIB
python - <<'PY'
import json
for l in open("gpurun_out/vpw1/time.jsonl"):
    d = json.loads(l); print(d["tag"], d["workload"], round(d["decode_ms"], 3), round(d.get("cn_ms", 0), 4), round(d.get("vn_ms", 0), 4))
PY
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -k "rcq or fuzz or random" > $O/pytest.log 2>&1; echo "rc=$?" >> $O/pytest.log; tail -3 $O/pytest.log
