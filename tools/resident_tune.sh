#!/bin/bash
# time the LDS-resident engine for several (G, NT) geometries; JSON lines to stdout
cd "$GRAFT_REPO_ROOT"
for w in basic rcq; do
for cfg in "2 512" "2 256" "2 1024" "4 1024" "4 512" "1 256" "1 512"; do
  set -- $cfg
  LDPC_RESIDENT_G=$1 LDPC_RESIDENT_NT=$2 timeout -k 10 120 python tools/time_sweeps.py --workload $w --tag "G$1_NT$2" 2>/dev/null | grep "^{"
done
done
