#!/bin/bash
# phase split of the resident kernel: LDPC_RES_DEBUG bits skip phases (1 check, 2 variable, 4 output, 8 final posterior+syndrome,
# 16 LLR load, 32 init, 64 return at once) in a -DLDPC_RESIDENT_PROBES build
cd "$GRAFT_REPO_ROOT"
export LDPC_HIP_LIB=$GRAFT_REPO_ROOT/build_variants/probes.so
for d in 0 1 2 3 4 8 12 16 32 48 63 127 0; do echo -n "skip=$d "; LDPC_RES_DEBUG=$d python tools/time_sweeps.py --workload basic 2>/dev/null | grep "^{" | cut -c50-95; done
