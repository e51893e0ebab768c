#!/bin/bash
# code-pair form with the LLR -> V2C-code pass in front of iteration 0 (instead of the fp32 check sweep on gathered LLR rows)
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/pair5; rm -rf $O; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q > $O/pytest.log 2>&1; rc=$?; echo "rc=$rc" >> $O/pytest.log; tail -5 $O/pytest.log
[ $rc = 0 ] || exit $rc
for mode in pair pair; do
  for w in wrcq_dvbs2 rcq; do
    timeout -k 10 200 python tools/time_sweeps.py --workload $w --mode $mode --tag init >> $O/time.jsonl 2>> $O/time.err
  done
done
python - <<'PY'
import json
for l in open("gpurun_out/pair5/time.jsonl"):
    d = json.loads(l); print(d["tag"], d["workload"], round(d["decode_ms"], 3), round(d.get("cn_ms", 0), 4), round(d.get("vn_ms", 0), 4))
PY
