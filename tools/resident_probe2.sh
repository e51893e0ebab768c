#!/bin/bash
cd "$GRAFT_REPO_ROOT"
for d in 3 7 11 15; do echo -n "skip=$d "; LDPC_RES_DEBUG=$d python tools/time_sweeps.py --workload basic --tag skip$d 2>/dev/null | grep "^{" | cut -c1-95; done
for cfg in "2 512" "2 256" "1 512" "1 256" "1 384" "2 384" "2 448"; do set -- $cfg; echo -n "G$1 NT$2 "; LDPC_RESIDENT_G=$1 LDPC_RESIDENT_NT=$2 python tools/time_sweeps.py --workload basic 2>/dev/null | grep "^{" | cut -c1-95; done
