#!/bin/bash
# RCQ code-pair form (vn_sweep_q / cn_sweep_q: 1-byte codes both ways) vs the fused gather form, streaming engine
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/pair2; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q > $O/pytest.log 2>&1; rc=$?; echo "rc=$rc" >> $O/pytest.log; tail -5 $O/pytest.log
[ $rc = 0 ] || exit $rc
for mode in pair gather sweeps pair gather; do
  for w in wrcq_dvbs2 rcq; do
    timeout -k 10 200 python tools/time_sweeps.py --workload $w --mode $mode --tag $mode >> $O/time.jsonl 2>> $O/time.err
  done
done
python - <<'PY'
import json
for l in open("gpurun_out/pair2/time.jsonl"):
    d = json.loads(l); print(d["tag"], d["workload"], round(d["decode_ms"], 3), round(d.get("cn_ms", 0), 4), round(d.get("vn_ms", 0), 4), d["engine"]["stream_form"])
PY
