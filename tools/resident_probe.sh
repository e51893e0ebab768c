#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/probe; mkdir -p $O
# needs a probe build: python -c "import _native; _native.build_native(force=True, defines=[\"LDPC_RESIDENT_PROBES\"])"
for d in 0 1 2 3; do
  echo -n "debug_skip=$d "; LDPC_RES_DEBUG=$d python tools/time_sweeps.py --workload basic --tag skip$d 2>/dev/null | grep "^{" | cut -c1-120
done
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INSTS_SMEM --kernel-trace --output-format csv -d $O/pmc1 -- python tools/time_sweeps.py --workload basic > $O/pmc1.log 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $O/pmc2 -- python tools/time_sweeps.py --workload basic > $O/pmc2.log 2>&1
rocprofv3 --pmc SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_BRANCH SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_ACTIVE_INST_MISC --kernel-trace --output-format csv -d $O/pmc3 -- python tools/time_sweeps.py --workload basic > $O/pmc3.log 2>&1
python - <<'PY'
import csv,glob,collections
# needs a probe build: python -c "import _native; _native.build_native(force=True, defines=[\"LDPC_RESIDENT_PROBES\"])"
for d in ("pmc1","pmc2","pmc3"):
    fs=glob.glob(f"gpurun_out/probe/{d}/**/*_counter_collection.csv",recursive=True)
    if not fs: print(d,"no output"); continue
    agg=collections.defaultdict(list)
    for r in csv.DictReader(open(fs[0])):
        if "resident" in r["Kernel_Name"]: agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k,v in agg.items(): print(d,k,len(v),sum(v)/len(v))
PY
