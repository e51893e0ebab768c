#!/bin/bash
set -o pipefail
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/gather1; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q > $O/pytest_parity.log 2>&1; echo "rc=$?" >> $O/pytest_parity.log
tail -5 $O/pytest_parity.log
for m in stream sweeps; do
  timeout -k 10 200 python tools/time_sweeps.py --workload wrcq_dvbs2 --mode $m >> $O/time.jsonl 2>> $O/time.err
  timeout -k 10 200 python tools/time_sweeps.py --workload rcq --mode $m >> $O/time.jsonl 2>> $O/time.err
done
cut -c1-600 $O/time.jsonl
