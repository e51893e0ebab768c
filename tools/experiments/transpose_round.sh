#!/bin/bash
# boundary transposes with 64-codeword x 512-byte blocks (256 B runs on both sides) vs 256 x 128 B (build of the commit before)
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/tr1; rm -rf $O; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_training.py -x -q > $O/pytest.log 2>&1; rc=$?; echo "rc=$rc" >> $O/pytest.log; tail -5 $O/pytest.log
[ $rc = 0 ] || exit $rc
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 bench.py --workload wrcq_dvbs2 --steps 5 --warmup 2 --no-cpu-baseline --no-legs > $O/stats.log 2>&1
python - <<'PY'
import csv, glob
f = sorted(glob.glob("gpurun_out/tr1/stats/**/*kernel_stats.csv", recursive=True))[-1]
for r in list(csv.DictReader(open(f)))[:9]:
    print(r["Name"][:60], r["Calls"], round(float(r["AverageNs"]) / 1e6, 4))
PY
tail -1 $O/stats.log | cut -c1-200
