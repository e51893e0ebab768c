#!/usr/bin/env python3
"""gpurun_out/<tag>/ctr_<workload>_p*/ (tools/gpu_round.sh) -> profiles/<tag>_<name>_counters.csv: per-launch averages of the
SQ / LDS counters of one kernel plus a few derived ratios.

    python tools/summarize_counters.py <tag> <workload> <kernel substring> <name>
    e.g. r02 basic resident_decode resident   |   r02 wrcq_dvbs2 vn_sweep_q4 pair_vn"""
import collections, csv, glob, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag, workload, kernel, name = sys.argv[1:5]
agg = collections.defaultdict(list)
for f in glob.glob(os.path.join(ROOT, "gpurun_out", tag, f"ctr_{workload}_p*", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if kernel in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
avg = {k: sum(v) / len(v) for k, v in agg.items()}
d = dict(avg)
if "SQ_WAVE_CYCLES" in avg and "SQ_ACTIVE_INST_VALU" in avg:
    d["derived_valu_active_per_wave_cycle"] = avg["SQ_ACTIVE_INST_VALU"] / avg["SQ_WAVE_CYCLES"]
if "SQ_BUSY_CYCLES" in avg and "SQ_ACTIVE_INST_VALU" in avg:
    d["derived_valu_active_per_busy_cycle"] = avg["SQ_ACTIVE_INST_VALU"] / avg["SQ_BUSY_CYCLES"]
if avg.get("SQ_LDS_IDX_ACTIVE") and "SQ_LDS_BANK_CONFLICT" in avg:
    d["derived_lds_conflict_share_of_lds_cycles"] = avg["SQ_LDS_BANK_CONFLICT"] / avg["SQ_LDS_IDX_ACTIVE"]
if "SQ_WAIT_ANY" in avg and "SQ_WAVE_CYCLES" in avg:
    d["derived_wait_share_of_wave_cycles"] = avg["SQ_WAIT_ANY"] / avg["SQ_WAVE_CYCLES"]
out = os.path.join(ROOT, "profiles", f"{tag}_{name}_counters.csv")
with open(out, "w", newline="") as fh:
    w = csv.writer(fh)
    w.writerow(["counter", "average_per_launch"])
    for k in sorted(d):
        w.writerow([k, d[k]])
print(open(out).read())
# per-launch counter values bench.py prices the kernel's instruction-issue and LDS-pipe limits with (roofline.valu_issue)
import json
cj = os.path.join(ROOT, "profiles", "counters.json")
db = json.load(open(cj)) if os.path.exists(cj) else {}
db.setdefault(workload, {})[kernel] = {"source": os.path.basename(out), **{k: avg[k] for k in sorted(avg)}}
json.dump(db, open(cj, "w"), indent=1)
