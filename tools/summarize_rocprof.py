#!/usr/bin/env python3
"""Condense rocprofv3 output directories (gpurun_out/, scratch) into the small tracked
summaries under profiles/.

    python tools/summarize_rocprof.py <round-tag> <workload> <stats_dir> [<fetch_dir> <write_dir>]

Writes profiles/<tag>_<workload>_kernel_stats.csv (rocprofv3 --kernel-trace --stats, every
kernel of the timed command) and, when the two PMC passes are given, profiles/<tag>_<workload>_pmc.csv
plus the per-launch HBM bytes of the sweep kernels into profiles/traffic.json.
gfx950 correction (MI355X_MICROARCH.md "HBM"): FETCH_SIZE and WRITE_SIZE are in KiB;
FETCH_SIZE reports exactly half of the bytes of a wide (16 B/lane) coalesced streaming read,
so hbm_bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 for the message sweeps.
"""
import collections, csv, glob, json, os, re, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PROF = os.path.join(ROOT, "profiles")


def short(name):
    name = name.replace("void ", "")
    return re.sub(r"\(.*", "", name)


def main():
    tag, workload, stats_dir = sys.argv[1:4]
    os.makedirs(PROF, exist_ok=True)
    f = max(glob.glob(os.path.join(stats_dir, "**", "*_kernel_stats.csv"), recursive=True), key=os.path.getmtime)   # newest run
    rows = list(csv.DictReader(open(f)))
    out = os.path.join(PROF, f"{tag}_{workload}_kernel_stats.csv")
    with open(out, "w", newline="") as fh:
        w = csv.writer(fh)
        w.writerow(["Kernel", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
        for r in rows:
            w.writerow([short(r["Name"]), r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"], r["MinNs"], r["MaxNs"]])
    print("wrote", out)
    if len(sys.argv) >= 6:
        agg = collections.defaultdict(lambda: collections.defaultdict(list))
        for d in sys.argv[4:6]:
            f = max(glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True), key=os.path.getmtime)
            for r in csv.DictReader(open(f)):
                if "ldpc::" in r["Kernel_Name"]:
                    agg[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
        out = os.path.join(PROF, f"{tag}_{workload}_pmc.csv")
        traffic = {}
        with open(out, "w", newline="") as fh:
            w = csv.writer(fh)
            w.writerow(["Kernel", "Launches", "FETCH_SIZE_KiB_mean", "WRITE_SIZE_KiB_mean", "hbm_bytes_per_launch_corrected"])
            for k, c in sorted(agg.items()):
                fe = sum(c["FETCH_SIZE"]) / max(len(c["FETCH_SIZE"]), 1)
                wr = sum(c["WRITE_SIZE"]) / max(len(c["WRITE_SIZE"]), 1)
                hbm = (2 * fe + wr) * 1024
                w.writerow([k, len(c["FETCH_SIZE"]), f"{fe:.1f}", f"{wr:.1f}", f"{hbm:.0f}"])
                traffic[k] = hbm
        print("wrote", out)
        tj = os.path.join(PROF, "traffic.json")
        data = json.load(open(tj)) if os.path.exists(tj) else {}
        # Every number records the file it came from; a kernel that this pass did not measure is DROPPED from the
        # workload's entry (bench.py then reports traffic null) instead of inheriting an older build's figure.
        entry = {"correction": "(2*FETCH_SIZE + WRITE_SIZE) * 1024"}
        src = f"{tag}_{workload}_pmc.csv"

        def pick(prefix, steady=True):
            # the steady-state instantiation: FIRST = false sweeps (template argument list ends in "false" for vn LAST=false;
            # cn_sweep<T, VEC, FORM, FIRST, ...> has FIRST as its 4th argument)
            best = None
            for k, v in traffic.items():
                if k.split("<")[0] != "ldpc::" + prefix:
                    continue
                args = k[k.index("<") + 1:k.rindex(">")].split(", ") if "<" in k else []
                if prefix == "cn_sweep" and len(args) >= 4 and args[3] != "false":
                    continue
                if prefix == "cn_sweep_f4" and args and args[0] != "false":       # cn_sweep_f4<FIRST, BPC, ES>
                    continue
                if prefix == "vn_sweep" and len(args) >= 4 and args[3] != "false":
                    continue
                if best is None or v > best:
                    best = v
            return best

        for key in ("cn_sweep", "cn_sweep_f4", "vn_sweep", "resident_decode", "cn_gather", "cn_sweep_q4", "vn_sweep_q4", "vn_last_rows",
                    "transpose_in_q4", "transpose_in_v", "layered_lds"):
            v = pick(key)
            if v is not None:
                entry[key] = {"bytes_per_launch": v, "source": src}
        data[workload] = entry
        json.dump(data, open(tj, "w"), indent=1)
        print("wrote", tj)


if __name__ == "__main__":
    main()
