#!/bin/bash
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/bar1; mkdir -p $O
export LDPC_HIP_LIB=$GRAFT_REPO_ROOT/build_variants/probes.so
for d in 0 128 129 130 0 128; do echo -n "skip=$d "; LDPC_RES_DEBUG=$d python tools/time_sweeps.py --workload basic 2>/dev/null | grep "^{" | cut -c50-95; done | tee $O/probe.txt
echo -n "no lane opt: "; LDPC_RESIDENT_NO_LANE_OPT=1 python tools/time_sweeps.py --workload basic 2>/dev/null | grep "^{" | cut -c50-95 | tee -a $O/probe.txt
