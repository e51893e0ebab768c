#!/bin/bash
cd "$GRAFT_REPO_ROOT"
for d in 15 31 47 63 79; do echo -n "skip=$d "; LDPC_RES_DEBUG=$d python tools/time_sweeps.py --workload basic 2>/dev/null | grep "^{" | cut -c1-95; done
