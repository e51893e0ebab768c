#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/probe7; mkdir -p $O
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM SQ_ACTIVE_INST_VALU --kernel-trace --output-format csv -d $O/pmc1 -- python tools/time_sweeps.py --workload basic > $O/pmc1.log 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_ANY SQ_INST_LEVEL_LDS --kernel-trace --output-format csv -d $O/pmc2 -- python tools/time_sweeps.py --workload basic > $O/pmc2.log 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE GRBM_COUNT --kernel-trace --output-format csv -d $O/pmc3 -- python tools/time_sweeps.py --workload basic > $O/pmc3.log 2>&1
rocprofv3 --pmc SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_LDS_ADDR_CONFLICT SQ_LDS_MEM_VIOLATIONS SQ_INSTS_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_BRANCH SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace --output-format csv -d $O/pmc4 -- python tools/time_sweeps.py --workload basic > $O/pmc4.log 2>&1
python - <<'PY'
import csv,glob,collections
for d in ("pmc1","pmc2","pmc3","pmc4"):
    fs=glob.glob(f"gpurun_out/probe7/{d}/**/*_counter_collection.csv",recursive=True)
    if not fs: print(d,"no output"); continue
    agg=collections.defaultdict(list); dur=[]
    for r in csv.DictReader(open(fs[0])):
        if "resident" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"])); dur.append(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))
    for k,v in agg.items(): print(d,k,len(v),sum(v)/len(v))
    if dur: print(d,"avg kernel ns",sum(dur)/len(dur))
PY
