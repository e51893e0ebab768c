#!/bin/bash
# resident-kernel round: full GPU suite, timings of the resident workloads, phase split (probes build)
set -o pipefail
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/res2; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "rc=$?" >> $O/pytest.log
tail -6 $O/pytest.log
for w in basic neural2d rcq basic_small wrcq_dvbs2; do
  timeout -k 10 200 python tools/time_sweeps.py --workload $w >> $O/time.jsonl 2>> $O/time.err
done
timeout -k 10 100 python tools/time_f64.py >> $O/time_f64.txt 2>> $O/time.err
cut -c1-200 $O/time.jsonl; cat $O/time_f64.txt
bash tools/resident_probe.sh > $O/probe.txt 2>&1; cat $O/probe.txt
