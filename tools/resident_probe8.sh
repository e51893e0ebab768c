#!/bin/bash
cd "$GRAFT_REPO_ROOT"
export LDPC_HIP_LIB=$GRAFT_REPO_ROOT/build_variants/probes.so
for d in 0 1 2 3 7 11 15 31 47 63; do echo -n "skip=$d "; LDPC_RES_DEBUG=$d python tools/time_sweeps.py --workload basic 2>/dev/null | grep "^{" | cut -c50-95; done
