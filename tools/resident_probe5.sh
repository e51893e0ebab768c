#!/bin/bash
cd "$GRAFT_REPO_ROOT"
for cfg in "2 512" "2 640" "2 768" "2 1024" "1 1024" "1 768"; do set -- $cfg; echo -n "G$1 NT$2 "; LDPC_RESIDENT_G=$1 LDPC_RESIDENT_NT=$2 python tools/time_sweeps.py --workload basic 2>/dev/null | grep "^{" | cut -c1-95; done
