#!/bin/bash
# round 3, GPU session 10: layered kernel with early reads + DPP forwarding: layered tests, timing, counters; bench layered leg
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r03s10; rm -rf $O; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "layered or workspace_cache" > $O/pytest_layered.log 2>&1; echo "pytest layered rc=$?" | tee -a $O/summary.txt
tail -5 $O/pytest_layered.log
timeout -k 10 120 python tools/time_layered.py > $O/layered.jsonl 2> $O/layered.err; echo "layered rc=$?" | tee -a $O/summary.txt
cat $O/layered.jsonl
timeout -k 10 300 python bench.py --workload rcq_layered --steps 5 --warmup 2 > $O/bench_layered.json 2> $O/bench_layered.err; echo "bench layered rc=$?" | tee -a $O/summary.txt
cut -c1-600 $O/bench_layered.json
timeout -k 10 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM SQ_WAVES SQ_WAVE_CYCLES --kernel-trace --output-format csv -d $O/ctr_rcq_layered_p1 -- python3 tools/time_sweeps.py --workload rcq_layered > $O/ctr1.log 2>&1; echo "ctr1 rc=$?" | tee -a $O/summary.txt
timeout -k 10 200 rocprofv3 --pmc SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_INST_LEVEL_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d $O/ctr_rcq_layered_p2 -- python3 tools/time_sweeps.py --workload rcq_layered > $O/ctr2.log 2>&1; echo "ctr2 rc=$?" | tee -a $O/summary.txt
tail -2 $O/ctr1.log
