#!/bin/bash
# does the config-5 decode scale linearly in the batch (strong-scaling leg at N = 1, 2, 4: 262144 / 131072 / 65536 per GPU)?
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r03; mkdir -p $O
for b in 32768 65536 131072 262144; do
  timeout -k 10 200 python tools/time_sweeps.py --workload wrcq_dvbs2 --batch $b --reps 3 2>> $O/scale_check.err | cut -c1-220 | tee -a $O/scale_check.jsonl
done
