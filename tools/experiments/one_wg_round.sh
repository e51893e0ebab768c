#!/bin/bash
# resident kernel, occupancy experiment (probes build): ONE workgroup per CU (LDS padded), 512 and 1024 threads, against the shipped two
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/onewg; rm -rf $O; mkdir -p $O
export LDPC_HIP_LIB=$PWD/build_variants/probes.so
run() { timeout -k 10 200 python bench.py --workload basic --steps 20 --warmup 5 --no-cpu-baseline --no-legs --no-stream-leg > $O/b.json 2> $O/b.err || echo "failed"; python3 -c "import json;d=json.load(open('$O/b.json'));print('$1', round(d['ms_per_step'],4), d['config']['engine'])" | tee -a $O/timings.txt; }
run "two WGs/CU, 512 threads"
LDPC_RES_LDS_PAD=20000 run "one WG/CU, 512 threads"
LDPC_RESIDENT_NT=1024 LDPC_RES_LDS_PAD=20000 run "one WG/CU, 1024 threads"
LDPC_RESIDENT_NT=1024 run "1024 threads, LDS allows two"
LDPC_RESIDENT_NT=768 LDPC_RES_LDS_PAD=20000 run "one WG/CU, 768 threads"
