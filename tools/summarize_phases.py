#!/usr/bin/env python3
"""gpurun_out/<tag>/phase_skip*/ (tools/gpu_round.sh, probes build) -> profiles/<tag>_resident_phase_histogram.csv:
instructions per wave of ldpc::resident_decode with one phase skipped at a time, and each phase's share by difference.

    python tools/summarize_phases.py <tag>"""
import collections, csv, glob, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
NAMES = {0: "full kernel", 1: "check phase", 2: "variable phase", 4: "output", 8: "final posterior + syndrome", 16: "LLR load", 32: "init (v2c = llr)"}
rows = {}
for d in NAMES:
    agg = collections.defaultdict(list)
    for f in glob.glob(os.path.join(ROOT, "gpurun_out", tag, f"phase_skip{d}", "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if "resident_decode" in r["Kernel_Name"]:
                agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    if agg:
        rows[d] = {k: sum(v) / len(v) for k, v in agg.items()}
out = os.path.join(ROOT, "profiles", f"{tag}_resident_phase_histogram.csv")
cols = ["SQ_INSTS_VALU", "SQ_INSTS_LDS", "SQ_INSTS_SALU", "SQ_INSTS_VMEM"]
with open(out, "w", newline="") as fh:
    w = csv.writer(fh)
    w.writerow(["phase", "how", "waves"] + [c + "_per_wave" for c in cols] + ["busy_cycles"])
    full = rows.get(0)
    for d, name in NAMES.items():
        if d not in rows or not full:
            continue
        wv = full["SQ_WAVES"]
        if d == 0:
            w.writerow([name, "measured", int(wv)] + [f"{full[c] / wv:.1f}" for c in cols] + [f"{full['SQ_BUSY_CYCLES']:.0f}"])
        else:
            w.writerow([name, "full - (phase skipped)", int(wv)] + [f"{(full[c] - rows[d][c]) / wv:.1f}" for c in cols] +
                       [f"{full['SQ_BUSY_CYCLES'] - rows[d]['SQ_BUSY_CYCLES']:.0f}"])
print(open(out).read())
