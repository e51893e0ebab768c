#!/bin/bash
# SQ / LDS counters of the resident kernel, one rocprofv3 --pmc pass per group (counters only with --kernel-trace),
# outputs under gpurun_out/counters/; tools/summarize_counters.py condenses them into profiles/.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/counters; mkdir -p $O
pass() { timeout -k 10 200 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $O/p$N -- python tools/time_sweeps.py --workload basic > $O/p$N.log 2>&1 || echo "pass $N failed" >> $O/errors.log; N=$((N+1)); }
N=1
pass SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM SQ_WAVES SQ_WAVE_CYCLES
pass SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_INST_LEVEL_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS
pass GRBM_COUNT GRBM_GUI_ACTIVE
ls $O; cat $O/errors.log 2>/dev/null
