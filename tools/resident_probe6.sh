#!/bin/bash
cd "$GRAFT_REPO_ROOT"
echo -n "small natural "; python tools/time_sweeps.py --workload basic_small 2>/dev/null | grep "^{" | cut -c60-260
echo -n "small pad->2blocks "; LDPC_RES_LDS_PAD=20000 python tools/time_sweeps.py --workload basic_small 2>/dev/null | grep "^{" | cut -c60-260
echo -n "small pad->1block "; LDPC_RES_LDS_PAD=60000 python tools/time_sweeps.py --workload basic_small 2>/dev/null | grep "^{" | cut -c60-260
echo -n "full natural "; python tools/time_sweeps.py --workload basic 2>/dev/null | grep "^{" | cut -c60-260
