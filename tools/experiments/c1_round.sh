#!/bin/bash
# does the register-held check form (LDPC_RES_CHECK_MODE=1) really execute fewer VALU instructions, and what does it do to time?
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/c1; mkdir -p $O
for lib in default c1v0; do
  if [ $lib = default ]; then unset LDPC_HIP_LIB; else export LDPC_HIP_LIB=$GRAFT_REPO_ROOT/build_variants/$lib.so; fi
  timeout -k 10 200 python tools/time_sweeps.py --workload basic --tag $lib >> $O/time.jsonl 2>> $O/time.err
  timeout -k 10 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU --kernel-trace --output-format csv -d $O/ctr_$lib -- python3 tools/time_sweeps.py --workload basic > $O/ctr_$lib.log 2>&1
done
unset LDPC_HIP_LIB
cut -c1-140 $O/time.jsonl
python - <<'PY'
import csv, glob, collections
for lib in ("default", "c1v0"):
    agg = collections.defaultdict(list)
    for f in glob.glob(f"gpurun_out/c1/ctr_{lib}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "resident_decode" in r["Kernel_Name"]:
                agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    print(lib, {k: round(sum(v) / len(v) / 262144, 1) for k, v in agg.items()})
PY
