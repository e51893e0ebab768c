#!/bin/bash
cd "$GRAFT_REPO_ROOT"
for b in 1 2 3 4; do echo -n "bpc=$b "; LDPC_RES_BPC=$b python tools/time_sweeps.py --workload basic 2>/dev/null | grep "^{" | cut -c1-95; done
for st in 1 2 4 8 16; do echo -n "stagger=$st "; LDPC_RES_STAGGER=$st python tools/time_sweeps.py --workload basic 2>/dev/null | grep "^{" | cut -c1-95; done
echo -n "G1 NT512 "; LDPC_RESIDENT_G=1 LDPC_RESIDENT_NT=512 python tools/time_sweeps.py --workload basic 2>/dev/null | grep "^{" | cut -c1-95
echo -n "G1 NT512 bpc3 "; LDPC_RES_BPC=3 LDPC_RESIDENT_G=1 LDPC_RESIDENT_NT=512 python tools/time_sweeps.py --workload basic 2>/dev/null | grep "^{" | cut -c1-95
echo -n "G1 NT512 bpc2 "; LDPC_RES_BPC=2 LDPC_RESIDENT_G=1 LDPC_RESIDENT_NT=512 python tools/time_sweeps.py --workload basic 2>/dev/null | grep "^{" | cut -c1-95
