#!/bin/bash
# resident kernel with the packed plan (8-byte entries that stay in L1) vs the round-start binary: fixed T and early stop
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/plan1; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "rc=$?" >> $O/pytest.log; tail -3 $O/pytest.log
for lib in default nofs old default nofs old; do
  if [ $lib = default ]; then unset LDPC_HIP_LIB; else export LDPC_HIP_LIB=$PWD/build_variants/$lib.so; fi
  for w in basic rcq neural2d; do
    timeout -k 10 200 python tools/time_sweeps.py --workload $w --tag $lib >> $O/time.jsonl 2>> $O/time.err
  done
done
unset LDPC_HIP_LIB
python - <<'PY'
import json
for l in open("gpurun_out/plan1/time.jsonl"):
    d = json.loads(l); print(d["tag"], d["workload"], round(d["decode_ms"], 3))
PY
