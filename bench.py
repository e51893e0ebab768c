#!/usr/bin/env python3
"""
bench.py -- decoded codewords/s at fixed iterations + roofline of the dominant kernel.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload basic|neural2d|rcq|wrcq_dvbs2|basic_f64]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

A step = ONE decode of one batch of synthetic LLRs (already resident in HBM) through the
hot path: T iterations of check + variable updates, syndrome, hard decisions out.
Default workload = BASELINE.json configs[1]: (1998,1512) code, BasicMinSumDecoder factor 0.7,
fp32, 10 iterations, batch 65536 per GPU, SNR 2.0 dB, fixed iterations (early_stop=False).

N > 1: one process per GPU.  Started bare (`python bench.py --gpus N`, no WORLD_SIZE in the
environment) the script spawns its N ranks itself as child processes BEFORE anything touches the
GPU, relays rank 0's JSON line and exits non-zero if a rank fails; started under
torch.distributed.run it is one of the ranks.  Weak scaling by default (every rank decodes its own
batch; `--strong` splits --batch over the ranks); every step ends with the RCCL all-gather of
the bit-packed hard decisions; value = all ranks' codewords / max-rank time.  A multi-rank line also
carries `distributed` (world size / backend as the process group reports them, every rank's device,
the all-gather timed alone, a content check of the gathered array on every rank) and
`sharded_workloads`: BASELINE config 5 -- (16200,7200) W-RCQ, T=20 -- weak (32768 per GPU) and strong
(262144 in total) with the same all-gather per step; rank 0 adds roofline and cpu_baseline as at N=1.

Prints ONE JSON line (rank 0) with the driver's fields plus
  roofline     : the dominant kernel of the timed step, timed live with HIP events on its stream.
                 LDS-resident engine: bound "lds" (conflict-free LDS time of its iteration phases / measured
                 time), its real HBM rate under "hbm" and the SURVEY 8d byte model under "hbm_formulation_equiv";
                 streaming engine: bound "hbm", ALGORITHMIC bytes of the CN sweep (8E per codeword fp32, 5E RCQ)
  stream_engine: the HBM-streaming engine on the same workload with the CN->VN sweep's HBM roofline (north_star)
  workloads    : bounded legs for the other BASELINE configs (Neural-2D, RCQ, (16200,7200) W-RCQ, fp64 Basic)
  cpu_baseline : the CPU oracle (C port of the reference loops) on a bounded sample
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
PKG_DIR = os.path.join(ROOT, "implementation-of-neural-ldpc-decoders-with-degree-specific-weight-sharing-and-rcq-quantization_amd")
for p in (PKG_DIR, ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)

HBM_PEAK_GBS = 8000.0      # MI355X HBM3E spec (MI355X_MICROARCH.md; ~6300 GB/s achievable)
QP = [(3.0, 1.3), (5.0, 1.3), (7.0, 1.3)]
METRIC = "decoded codewords/sec at fixed iters; achieved HBM GB/s vs peak"

WORKLOADS = {
    # name: (graph, iterations, default batch per GPU, arithmetic, description)
    "basic": ("ira_1998_1512", 10, 65536, "f32", "(1998,1512) IRA code, BasicMinSumDecoder factor=0.7, fp32"),
    "neural2d": ("ira_1998_1512", 10, 65536, "f32", "(1998,1512) IRA code, Neural2DMinSumDecoder type 2, fp32"),
    "rcq": ("ira_1998_1512", 10, 65536, "f32", "(1998,1512) IRA code, RCQMinSumDecoder bc=3 bv=8, 3 quantisers"),
    "wrcq_dvbs2": ("dvbs2_like_16200_7200", 20, 32768, "f32", "(16200,7200) DVB-S2-like code, WeightedRCQDecoder type 2 bc=3"),
    "basic_f64": ("ira_1998_1512", 10, 65536, "f64", "(1998,1512) IRA code, BasicMinSumDecoder factor=0.7, float64 as in the reference"),
    "rcq_layered": ("ira_1998_1512", 10, 65536, "f32", "(1998,1512) IRA code, RCQMinSumDecoder bc=3 layered=True (the reference's "
                                                       "layered schedule as it executes, rcq_decoder.py:281-350)"),
}
LEGS = ("neural2d", "rcq", "wrcq_dvbs2", "basic_f64", "rcq_layered")   # secondary legs of the default single-GPU run


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", choices=sorted(WORKLOADS), default="basic")
    ap.add_argument("--batch", type=int, default=0, help="codewords per GPU (default: the workload's)")
    ap.add_argument("--snr-db", type=float, default=2.0)
    ap.add_argument("--early-stop", action="store_true", help="reference early-exit semantics instead of fixed T")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--sweep-reps", type=int, default=20)
    ap.add_argument("--no-overlap", action="store_true", help="join every step's all-gather before the next decode")
    ap.add_argument("--no-stream-leg", action="store_true", help="skip the secondary streaming-engine measurement")
    ap.add_argument("--no-legs", action="store_true", help="skip the legs for the other BASELINE configs")
    ap.add_argument("--leg-steps", type=int, default=5)
    ap.add_argument("--force-dist", action="store_true",
                    help="initialise the process group and run the all-gather path even with one rank (checks the RCCL plumbing on a single GPU)")
    ap.add_argument("--strong", action="store_true",
                    help="strong scaling: --batch (default: the workload's) is the TOTAL over all GPUs, split evenly")
    ap.add_argument("--config5-total", type=int, default=262144,
                    help="total codewords of the strong-scaling config-5 leg of a multi-rank run (BASELINE.json: 262144)")
    ap.add_argument("--config5-max-per-gpu", type=int, default=262144,
                    help="skip that leg when total / N exceeds this many codewords per GPU (memory: ~0.42 MB per codeword)")
    return ap.parse_args(argv)


# ----------------------------------------------------------------------------------- self-launch
def free_port():
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def spawn_ranks(n, argv):
    """`python bench.py --gpus N` started bare: run the N ranks as CHILD processes (one per GPU) and relay rank 0's
    JSON line.  This parent never touches the GPU (no HIP call, no torch.cuda call) -- nothing is re-exec'ed.
    All children are polled: the first rank that exits non-zero ends the run (the others are killed, not left waiting
    in a collective until the process-group timeout), and every rank's stderr is kept and shown on failure."""
    import tempfile
    port = os.environ.get("MASTER_PORT") or str(free_port())
    procs, errs = [], []
    out0 = tempfile.TemporaryFile()
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=port)
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        errs.append(tempfile.TemporaryFile())
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env,
                                      stdout=out0 if r == 0 else subprocess.DEVNULL, stderr=errs[-1]))
    failed = None
    while failed is None and any(p.poll() is None for p in procs):
        for r, p in enumerate(procs):
            if p.poll() not in (None, 0):
                failed = r
                break
        else:
            time.sleep(0.2)
    if failed is not None:
        for p in procs:
            if p.poll() is None:
                p.kill()
    codes = [p.wait() for p in procs]
    out0.seek(0)
    lines = [l for l in out0.read().decode(errors="replace").splitlines() if l.startswith("{")]
    if any(codes) or len(lines) != 1:
        sys.stderr.write(f"bench.py: rank exit codes {codes}, {len(lines)} result line(s)\n")
        for r, f in enumerate(errs):
            f.seek(0)
            tail = f.read().decode(errors="replace")[-3000:]
            if tail.strip():
                sys.stderr.write(f"---- rank {r} stderr (tail) ----\n{tail}\n")
        rc = codes[failed] if failed is not None else next((c for c in codes if c), 1)
        raise SystemExit(rc if 0 < rc < 256 else 1)
    for r, f in enumerate(errs):                      # relay the ranks' diagnostics (warnings, RCCL banner) to our stderr
        f.seek(0)
        sys.stderr.write(f.read().decode(errors="replace"))
    print(lines[0], flush=True)


# ----------------------------------------------------------------------------------- workload construction
def synthetic_tables(dec, seed=4321):
    """'pretrained' weights stand-in: beta ~ U(0.5,1), alpha ~ U(0.8,1.2)"""
    import numpy as np
    import torch
    rng = np.random.default_rng(seed)
    with torch.no_grad():
        for k in sorted(dec.beta_weights.keys()):
            dec.beta_weights[k].fill_(float(np.float32(rng.uniform(0.5, 1.0))))
        for k in sorted(dec.alpha_weights.keys()):
            dec.alpha_weights[k].fill_(float(np.float32(rng.uniform(0.8, 1.2))))


def build_decoder(workload, device):
    """-> (engine, host decoder, code)"""
    import torch
    import codes
    from ldpc_decoder import BasicMinSumDecoder
    from neural_2d_decoder import Neural2DMinSumDecoder
    from rcq_decoder import RCQMinSumDecoder, WeightedRCQDecoder
    gname, T = WORKLOADS[workload][:2]
    code = codes.load_code(gname, max_iterations=T)
    if workload in ("basic", "basic_f64"):
        dec = BasicMinSumDecoder(code, factor=0.7)
        eng = dec._engine(torch.float64 if workload == "basic_f64" else torch.float32, device)
    elif workload == "neural2d":
        dec = Neural2DMinSumDecoder(code, weight_sharing_type=2, max_iterations=T)
        synthetic_tables(dec)
        eng = dec._get_engine(device)
    elif workload in ("rcq", "rcq_layered"):
        dec = RCQMinSumDecoder(code, bc=3, bv=8, quantizer_params=QP, max_iterations=T, layered=workload == "rcq_layered")
        eng = dec._get_engine(device)
    else:
        dec = WeightedRCQDecoder(code, bc=3, bv=8, quantizer_params=QP, weight_sharing_type=2, max_iterations=T)
        synthetic_tables(dec)
        eng = dec._get_engine(device)
    return eng, dec, code


def make_llr(batch, n, snr_db, seed, device, dtype=None):
    """all-zero codeword, decoder convention (+LLR = bit 0): llr = 2(1 + sigma z)/sigma^2"""
    import torch
    s2 = 10.0 ** (-snr_db / 10.0)
    gen = torch.Generator(device=device)
    gen.manual_seed(seed)
    z = torch.randn((batch, n), generator=gen, device=device, dtype=torch.float32)
    x = (2.0 * (1.0 + (s2 ** 0.5) * z) / s2).contiguous()
    return x if dtype in (None, torch.float32) else x.to(dtype)


def wants_posterior(workload):
    return workload in ("neural2d", "wrcq_dvbs2")      # those decoders return the posterior


def byte_model(workload, g, T, B):
    """SURVEY 8d ALGORITHMIC bytes (reference formulation: messages through HBM once per sweep)"""
    es = 8 if workload == "basic_f64" else 4
    rcq_like = workload in ("rcq", "wrcq_dvbs2", "rcq_layered")
    c2v = 1 if rcq_like else es
    per_iter = 2 * (es + c2v) * g.E + es * g.n                    # fp32: 16E + 4n, RCQ: 10E + 4n
    return {"decode": (T * per_iter + (es + 4) * g.n) * B,        # + posterior and int32 decisions out
            "cn_sweep": (es + c2v) * g.E * B,                      # read v2c, write c2v
            "vn_sweep": ((es + c2v) * g.E + es * g.n) * B,         # read c2v + llr, write v2c
            "iteration": per_iter * B,
            "compulsory_io": (es + 4) * g.n * B}                   # LLRs in, int32 decisions out (fused engine)


def resident_lds_model(g, T, B, G, ms, es=4):
    """What bounds the fused kernel itself: LDS traffic of its phases against the LDS rates of MI355X_MICROARCH.md
    ("LDS": ds_read_b64 256 B/clk/CU, ds_write_b64 ~85 B/clk/CU, 256 CUs at 2.4 GHz), conflict-free.  Slots are
    es*G bytes; per codeword group and iteration the check phase reads every slot twice and writes it once, the
    variable phase reads every slot and its LLR once and writes every slot once."""
    slot = es * G
    groups = (B + G - 1) // G
    reads = groups * slot * (T * 2 * g.E + T * (g.E + g.n))          # bytes
    writes = groups * slot * (T * g.E + T * g.E)
    rd_peak, wr_peak = 256 * 256 * 2.4e9, 85 * 256 * 2.4e9             # B/s, all CUs
    t_min = reads / rd_peak + writes / wr_peak
    return reads, writes, t_min * 1e3


def load_traffic(workload):
    """HBM bytes per launch from the PMC passes of the latest profiled build (profiles/traffic.json, written by
    tools/summarize_rocprof.py with the gfx950 correction); every number carries the file it came from"""
    tf = os.path.join(ROOT, "profiles", "traffic.json")
    try:
        db = json.load(open(tf))
        return {**db.get(workload + "_gather", {}), **db.get(workload, {})}     # `_gather`: the pass with that RCQ form forced
    except Exception:
        return {}


SIMDS, CUS, CLOCK_HZ = 1024, 256, 2.4e9            # MI355X: 256 CUs x 4 SIMDs, 2.4 GHz peak engine clock


def issue_limits(workload, kernel, B, ms):
    """Instruction-issue and LDS-pipe time of a kernel from the SQ counters of the latest profiled build
    (profiles/counters.json, written by tools/summarize_counters.py; per-launch averages at the workload's default batch):
      valu_issue: SQ_INSTS_VALU wave-instructions x 4 cycles / 1024 SIMDs / 2.4 GHz -- the time the VALUs alone need when every
                  instruction costs 4 cycles.  Measured on gfx950 (tools/probes/valu_issue_probe.hip, profiles/r03_valu_issue_probe.txt):
                  4.1-4.2 cycles for compares, selects, min/max/med3, shifts, v_perm, conversions, packed, DPP and anything with an
                  SGPR operand; 2.2-2.4 for v_add/sub/mul/fma_f32, and/or/xor, v_add/sub_u32, v_mov, v_bitop3 on VGPR / constant
                  operands once two waves share the SIMD -- the kernels here are mostly of the first kind, so this is an upper
                  estimate by the share of the second kind (`cycles_per_instruction` says which figure was used);
      lds_pipe  : SQ_LDS_IDX_ACTIVE cycles (LDS array busy, bank-conflict replays included) / 256 CUs / 2.4 GHz.
    `frac` = that time / the measured launch time; the larger one is the binding limit.  None when no counters are on file
    for this batch."""
    try:
        ent = json.load(open(os.path.join(ROOT, "profiles", "counters.json")))[workload][kernel]
    except Exception:
        return None
    if B != WORKLOADS[workload][2] or "SQ_INSTS_VALU" not in ent:
        return None
    out = {"source": ent.get("source"), "clock_GHz": CLOCK_HZ / 1e9}
    valu_ms = ent["SQ_INSTS_VALU"] * 4 / SIMDS / CLOCK_HZ * 1e3
    out["valu_issue"] = {"wave_instructions_per_launch": ent["SQ_INSTS_VALU"], "cycles_per_instruction": 4, "ms": valu_ms,
                         "frac": valu_ms / ms}
    if "SQ_LDS_IDX_ACTIVE" in ent:
        lds_ms = ent["SQ_LDS_IDX_ACTIVE"] / CUS / CLOCK_HZ * 1e3
        out["lds_pipe"] = {"busy_cycles_per_launch": ent["SQ_LDS_IDX_ACTIVE"], "ms": lds_ms, "frac": lds_ms / ms,
                           "bank_conflict_share": (ent.get("SQ_LDS_BANK_CONFLICT", 0.0) / ent["SQ_LDS_IDX_ACTIVE"])
                           if ent["SQ_LDS_IDX_ACTIVE"] else None}
    best = max((k for k in ("valu_issue", "lds_pipe") if k in out), key=lambda k: out[k]["frac"])
    out["binding"] = best
    return out


def traffic_of(db, kernel, workload=None, B=None):
    """(bytes per launch, source file) -- only for the batch the PMC passes ran at (the workload's default)"""
    ent = db.get(kernel)
    if isinstance(ent, dict) and (workload is None or B == WORKLOADS[workload][2]):
        return ent.get("bytes_per_launch"), ent.get("source")
    return None, None


def event_ms(fn, reps, torch):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) / reps


def stream_roofline(engine, workload, g, T, B, llr, reps, copy_gbs, torch):
    """CN / VN kernels of the streaming engine, timed live with HIP events on the stream they are launched on
    (torch's current stream).  RCQ decoders run the fused iteration kernel (cn_gather) instead of the two sweeps."""
    engine.decode(llr, early_stop=False, want_posterior=False)          # leaves valid state in the workspace
    bm = byte_model(workload, g, T, B)
    db = load_traffic(workload)
    info = engine.info()
    it = 1 if T > 1 else 0
    times = {}
    for which, name in ((0, "cn"), (1, "vn")):
        for _ in range(3):
            engine.debug_sweep(B, which, it)
        times[name] = event_ms(lambda: engine.debug_sweep(B, which, it), reps, torch)
    if info.get("stream_form") == "rcq-code-pair" and 1 <= it < T - 1:
        # Both directions are 1-byte codes: the check sweep reads and writes E bytes per codeword (integer-only), the
        # variable sweep reads E code bytes + the 4n LLR bytes and writes E bytes, quantising with the next iteration's
        # beta and thresholds.  The variable sweep is the longer of the two and the roofline entry; the pair is one
        # iteration (4E + 4n bytes per codeword against 10E + 4n of the fp32-V2C formulation).
        cn_b, vn_b = 2 * g.E * B, (2 * g.E + 4 * g.n) * B
        ach = vn_b / (times["vn"] * 1e-3) / 1e9
        tr, src = traffic_of(db, "vn_sweep_q4", workload, B)
        trc, srcc = traffic_of(db, "cn_sweep_q4", workload, B)
        t_it = times["cn"] + times["vn"]
        return {"bound": "hbm", "kernel": "ldpc::vn_sweep_q4 (variable sweep of the RCQ code-pair form: C2V codes + LLRs in, V2C codes "
                                          "out; streaming engine)",
                "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
                "algorithmic_bytes_per_launch": vn_b, "ms_per_launch": times["vn"],
                "traffic": tr, "traffic_source": src,
                "measured_copy_GBps": copy_gbs, "frac_of_measured_copy": ach / copy_gbs if copy_gbs else None,
                "cn_sweep_q4": {"ms_per_launch": times["cn"], "algorithmic_bytes_per_launch": cn_b,
                                "achieved": cn_b / (times["cn"] * 1e-3) / 1e9,
                                "frac": cn_b / (times["cn"] * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                "traffic": trc, "traffic_source": srcc},
                "iteration": {"ms": t_it, "algorithmic_bytes": cn_b + vn_b,
                              "achieved": (cn_b + vn_b) / (t_it * 1e-3) / 1e9,
                              "frac": (cn_b + vn_b) / (t_it * 1e-3) / 1e9 / HBM_PEAK_GBS},
                "hbm_formulation_equiv": {"bytes_per_launch": bm["iteration"],
                                          "achieved": bm["iteration"] / (t_it * 1e-3) / 1e9,
                                          "frac": bm["iteration"] / (t_it * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                          "note": "SURVEY 8d bytes of one iteration in the fp32-V2C two-sweep formulation (10E+4n per "
                                                  "codeword) over this form's cn + vn time"}}
    if info.get("stream_form") == "fused-rcq-iteration" and it >= 1:
        # One kernel = one whole iteration.  ALGORITHMIC bytes of this formulation: per edge the LLR of its variable (4 B)
        # and the codes of the variable's OTHER edges (dv-1 B) are read, one code byte is written -- 4E + sum_j dv(dv-1) + E
        # per codeword.  Re-reads of a row by the other checks of its variable miss the XCD-local L2 and are served through
        # the fabric (Infinity Cache / HBM), which is what FETCH_SIZE counts: the PMC traffic equals these bytes.
        import numpy as np
        reread = int((g.dv.astype(np.int64) * (g.dv.astype(np.int64) - 1)).sum())
        issued = (4 * g.E + reread + g.E) * B
        distinct = (4 * g.n + 2 * g.E) * B
        ach = issued / (times["cn"] * 1e-3) / 1e9
        tr, src = traffic_of(db, "cn_gather", workload, B)
        return {"bound": "hbm", "kernel": "ldpc::cn_gather (fused RCQ iteration: V2C recomputed from the 1-byte codes + LLRs, "
                                          "CN->VN codes written; streaming engine)",
                "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
                "algorithmic_bytes_per_launch": issued, "ms_per_launch": times["cn"],
                "traffic": tr, "traffic_source": src,
                "measured_copy_GBps": copy_gbs, "frac_of_measured_copy": ach / copy_gbs if copy_gbs else None,
                "distinct_bytes": {"bytes_per_launch": distinct, "GBps": distinct / (times["cn"] * 1e-3) / 1e9,
                                   "note": "LLRs + code bytes in and out once (4n + 2E per codeword): what would remain if every "
                                           "re-read hit in L2; the rest of the traffic is served by the Infinity Cache"},
                "hbm_formulation_equiv": {"bytes_per_launch": bm["iteration"],
                                          "achieved": bm["iteration"] / (times["cn"] * 1e-3) / 1e9,
                                          "frac": bm["iteration"] / (times["cn"] * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                          "note": "SURVEY 8d bytes of one iteration in the two-sweep formulation (10E+4n per codeword) "
                                                  "over this kernel's time: the same work as one cn_sweep + one vn_sweep launch"},
                "posterior_pass": {"ms_per_launch": times["vn"]}}
    ach = bm["cn_sweep"] / (times["cn"] * 1e-3) / 1e9
    tr, src = traffic_of(db, "cn_sweep_f4", workload, B)            # fp32 min-sum, check degrees <= 16: the register-held form
    if tr is None:
        tr, src = traffic_of(db, "cn_sweep", workload, B)
    return {"bound": "hbm", "kernel": "ldpc::cn_sweep_f4 / ldpc::cn_sweep (check-node / CN->VN message sweep, streaming engine)",
            "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
            "measured_copy_GBps": copy_gbs, "frac_of_measured_copy": ach / copy_gbs if copy_gbs else None,
            "traffic": tr, "traffic_source": src,
            "algorithmic_bytes_per_launch": bm["cn_sweep"], "ms_per_launch": times["cn"],
            "vn_sweep": {"ms_per_launch": times["vn"],
                         "achieved": bm["vn_sweep"] / (times["vn"] * 1e-3) / 1e9,
                         "algorithmic_bytes_per_launch": bm["vn_sweep"]}}


def resident_roofline(eng, workload, g, T, B, ms, copy_gbs):
    """The fused kernel keeps every message in LDS: its limiter is LDS / instruction issue, not HBM.  `frac` is the
    conflict-free LDS time of its iteration phases over the measured time; the real HBM rate and the SURVEY 8d byte
    model (which this kernel does not move) are reported beside it under their own names."""
    es = 8 if workload == "basic_f64" else 4
    info = eng.info()
    G = max(info["codewords_per_workgroup"], 1)
    reads, writes, t_min = resident_lds_model(g, T, B, G, ms, es=es)
    bm = byte_model(workload, g, T, B)
    db = load_traffic(workload)
    tr, src = traffic_of(db, "resident_decode", workload, B)
    hbm_bytes = tr if tr else bm["compulsory_io"]
    lds_ach = (reads + writes) / (ms * 1e-3) / 1e9
    return {"bound": "lds", "kernel": "ldpc::resident_decode (fused T-iteration decode, messages resident in LDS)",
            "achieved": lds_ach, "peak": lds_ach * ms / t_min, "unit": "GB/s", "frac": t_min / ms,
            # what actually binds the kernel: VALU issue and the LDS pipe's busy time (conflict replays included), from counters
            "issue_limits": issue_limits(workload, "resident_decode", B, ms),
            "ms_per_launch": ms, "lds_read_bytes": reads, "lds_write_bytes": writes, "lds_min_ms": t_min,
            "traffic": tr, "traffic_source": src,
            "hbm": {"bytes_per_launch": hbm_bytes, "bytes_kind": "PMC-measured" if tr else "compulsory (LLRs in + decisions out)",
                    "achieved": hbm_bytes / (ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS,
                    "frac": hbm_bytes / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS},
            "hbm_formulation_equiv": {"bytes_per_launch": bm["decode"],
                                      "achieved": bm["decode"] / (ms * 1e-3) / 1e9,
                                      "frac": bm["decode"] / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                      "note": "SURVEY 8d algorithmic bytes T(16E+4n)+8n per codeword (10E for RCQ) of the "
                                              "HBM-streaming formulation over this kernel's time -- an equivalence figure, "
                                              "not traffic: the fused kernel never moves these bytes"},
            "measured_copy_GBps": copy_gbs,
            "note": "peak = the blended LDS rate of this read/write mix (ds_read_b64 256 B/clk/CU, ds_write_b64 ~85 B/clk/CU, "
                    "256 CUs, 2.4 GHz); the variable phase's gathers/scatters take ~2 LDS passes per instruction, the rest "
                    "is VALU issue and barriers"}


def layered_roofline(eng, g, T, B, ms):
    """The layered walk is ONE dependent chain of m*T check updates per codeword; the LDS-resident kernel (ldpc_layered.hip)
    keeps `cw` codewords per one-wave workgroup and as many workgroups per CU as their posteriors fit 160 KiB of LDS.  What
    bounds it is the latency of one step (LDS read -> cross-lane min/sign butterfly -> quantise -> LDS write) times the chain
    length, times the rounds the batch needs at that residency -- reported as the achieved step time."""
    info = eng.info()
    if info["engine"] != "resident":
        return {"bound": "latency", "kernel": "ldpc::layered_rcq (streaming form: posteriors in HBM, one wave per 64-codeword tile)",
                "ms_per_launch": ms, "traffic": None}
    cw, lds = max(info["codewords_per_workgroup"], 1), max(info["lds_bytes"], 1)
    per_cu = max((160 * 1024) // lds, 1)
    waves = (B + cw - 1) // cw
    rounds = waves / (per_cu * CUS)
    steps = g.m * T
    step_ns = ms * 1e6 / max(rounds, 1.0) / max(steps, 1)
    lds_bytes = 2.0 * 4 * g.E * T * B                                # one 4-byte read and write per edge and iteration
    lim = issue_limits("rcq_layered", "layered_lds", B, ms)          # VALU issue / LDS pipe shares from the SQ counters on file
    tr, src = traffic_of(load_traffic("rcq_layered"), "layered_lds", "rcq_layered", B)
    return {"bound": "latency", "kernel": "ldpc::layered_lds (layered RCQ walk, posteriors resident in LDS, lanes over the edges of a check)",
            "ms_per_launch": ms, "codewords_per_wave": cw, "waves_per_cu": per_cu, "rounds": rounds, "dependent_steps": steps,
            "achieved": step_ns, "unit": "ns per dependent check step", "cycles_per_step_at_2.4GHz": step_ns * 2.4,
            "peak": None, "frac": lim["valu_issue"]["frac"] if lim else None, "frac_kind": "VALU issue time / launch time",
            "issue_limits": lim, "lds_traffic_GBps": lds_bytes / (ms * 1e-3) / 1e9, "traffic": tr, "traffic_source": src,
            "note": "latency-bound by construction: codewords in flight per CU = LDS capacity / 4n bytes; no HBM or LDS "
                    "bandwidth limit is near (HBM sees the LLRs in and the decisions out only)"}


def cpu_baseline(workload, dec, code, snr_db, budget_s=12.0):
    """The CPU oracle (oracle/ldpc_oracle.c, a C port of the reference's loops; the Python
    reference itself cannot travel to the GPU box) on a bounded sample of the same workload."""
    import numpy as np
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle
    oracle.build()
    g = code.tanner_graph()
    og = oracle.OracleGraph(n=g.n, check_ptr=g.check_ptr, var_idx=g.var_idx)
    T = WORKLOADS[workload][1]
    rng = np.random.default_rng(1234)
    s2 = 10.0 ** (-snr_db / 10.0)
    threads = oracle.num_threads()

    def run(cnt):
        x = (2.0 * (1.0 + np.sqrt(s2) * rng.standard_normal((cnt, g.n))) / s2)
        x = x.astype(np.float64 if workload == "basic_f64" else np.float32)
        t0 = time.perf_counter()
        if workload in ("basic", "basic_f64"):
            oracle.basic_minsum(og, x, 0.7, T, early_stop=False, dtype=x.dtype.type)
        elif workload == "neural2d":
            oracle.neural2d(og, x, 2, T, {k: float(v.item()) for k, v in dec.beta_weights.items()},
                            {k: float(v.item()) for k, v in dec.alpha_weights.items()}, early_stop=False)
        elif workload == "rcq":
            oracle.rcq(og, x, 3, QP, T, early_stop=False)
        elif workload == "rcq_layered":
            oracle.rcq_layered(og, x, 3, QP, T)             # early-stop semantics only; at 2 dB nothing converges: T iterations
        else:
            oracle.weighted_rcq(og, x, 3, QP, 2, T, {k: float(v.item()) for k, v in dec.beta_weights.items()},
                                {k: float(v.item()) for k, v in dec.alpha_weights.items()}, early_stop=False)
        return time.perf_counter() - t0

    probe = max(threads * 2, 8)
    t_probe = run(probe)
    cnt = int(max(probe, min(65536, probe * budget_s / max(t_probe, 1e-6))))
    t = run(cnt)
    return {"value": cnt / t, "unit": "codewords/s", "cores": threads, "kind": "port",
            "sample": f"{cnt} codewords of the same workload, fixed {T} iterations, OpenMP over codewords, {t:.1f} s"}


def copy_ceiling(device, torch):
    """Measured HBM ceiling of this box (SURVEY 8d asks for it beside the 8 TB/s spec): a 1 GiB -> 1 GiB
    device copy (read + write, far beyond the 256 MB Infinity Cache), best of 6, GB/s"""
    src = torch.empty(1 << 28, dtype=torch.float32, device=device).normal_()
    dst = torch.empty_like(src)
    best = 0.0
    for _ in range(6):
        ms = event_ms(lambda: dst.copy_(src), 1, torch)
        best = max(best, 2 * src.numel() * 4 / (ms * 1e-3) / 1e9)
    del src, dst
    return best


def fill_ceiling(device, torch):
    """Measured WRITE rate of this box: a 1 GiB device fill (write-only, no read), best of 6, GB/s -- reported beside the
    copy rate (read + write bytes of a library copy) so that a sweep's mixed read/write rate can be placed between them."""
    dst = torch.empty(1 << 28, dtype=torch.float32, device=device)
    best = 0.0
    for _ in range(6):
        ms = event_ms(lambda: dst.fill_(1.0), 1, torch)
        best = max(best, dst.numel() * 4 / (ms * 1e-3) / 1e9)
    del dst
    return best


def measure_leg(workload, device, snr_db, steps, reps, copy_gbs, with_cpu, torch):
    """one bounded leg of another BASELINE config on this GPU: decode rate + the roofline of its dominant kernel"""
    gname, T, B, dtype, desc = WORKLOADS[workload]
    eng, dec, code = build_decoder(workload, device)
    g = code.tanner_graph()
    llr = make_llr(B, g.n, snr_db, 1234, device, torch.float64 if dtype == "f64" else torch.float32)
    wp = wants_posterior(workload)
    run = lambda: eng.decode(llr, early_stop=False, want_bits=True, want_posterior=wp)
    run()
    torch.cuda.synchronize(device)
    ms = event_ms(run, steps, torch)
    res = run()
    assert int(res.iterations.min().item()) == T
    leg = {"workload": f"{desc}, {T} iterations, batch {B}, SNR {snr_db} dB, fixed-iteration flooding decode",
           "dtype": dtype, "batch": B, "steps": steps, "ms_per_step": ms, "value": B / (ms * 1e-3), "unit": "codewords/s",
           "engine": eng.info(), "edges": g.E, "n": g.n,
           "decode_algorithmic_GBps": byte_model(workload, g, T, B)["decode"] / (ms * 1e-3) / 1e9}
    if workload == "rcq_layered":
        leg["roofline"] = layered_roofline(eng, g, T, B, ms)
    elif eng.info()["engine"] == "resident":
        leg["roofline"] = resident_roofline(eng, workload, g, T, B, ms, copy_gbs)
    else:
        leg["roofline"] = stream_roofline(eng, workload, g, T, B, llr, max(reps // 2, 2), copy_gbs, torch)
    if with_cpu:
        leg["cpu_baseline"] = cpu_baseline(workload, dec, code, snr_db, budget_s=4.0)
    del eng, dec, llr, res
    torch.cuda.empty_cache()
    return leg


class ShardedRun:
    """One workload on this rank's GPU, every step ending with the all-gather of the bit-packed hard decisions
    (RCCL over xGMI with backend "nccl"; gloo = host-side rehearsal on a one-GPU box).  All ranks construct and drive
    their ShardedRun objects in the same order -- every method that talks to the process group is collective."""

    def __init__(self, workload, B, ctx, early, overlap, snr_db):
        import torch
        self.torch, self.ctx = torch, ctx
        self.workload, self.B, self.early, self.overlap = workload, int(B), bool(early), bool(overlap)
        gname, self.T, _, self.dtype, self.desc = WORKLOADS[workload]
        self.gname = gname
        self.eng, self.dec, self.code = build_decoder(workload, ctx["device"])
        self.g = self.code.tanner_graph()
        self.llr = make_llr(self.B, self.g.n, snr_db, 1234 + ctx["rank"], ctx["device"],
                            torch.float64 if self.dtype == "f64" else torch.float32)
        self.want_post = wants_posterior(workload)
        self.nbytes = (self.g.n + 7) // 8
        self.pending = []          # [(work, gathered, packed_src)] -- the previous step's all-gather, still in flight
        self.res = self.gathered = None

    def step(self):
        """decode this rank's shard, then all-gather the bit-packed hard decisions.  Overlapped form: the gather of step k
        runs on RCCL's stream while step k+1 decodes (separate buffers), and is joined one step later."""
        import torch.distributed as dist
        from sharding import all_gather_hard_decisions
        ctx, torch = self.ctx, self.torch
        res = self.eng.decode(self.llr, early_stop=self.early, want_bits=True, want_posterior=self.want_post,
                              want_packed=ctx["use_dist"])
        gathered = None
        if ctx["use_dist"]:
            total = self.B * ctx["world"]
            if ctx["backend"] != "nccl":                           # rehearsal path: host-side gather
                gathered = all_gather_hard_decisions(res.packed_bits.cpu(), total)
            elif not self.overlap:
                gathered = all_gather_hard_decisions(res.packed_bits, total)
            else:
                if self.pending:
                    self.pending.pop()[0].wait()
                gathered = torch.empty((total, self.nbytes), dtype=torch.uint8, device=ctx["device"])
                work = dist.all_gather_into_tensor(gathered, res.packed_bits, async_op=True)
                self.pending.append((work, gathered, res.packed_bits))
        self.res, self.gathered = res, gathered
        return res, gathered

    def fence(self):
        import torch.distributed as dist
        while self.pending:
            self.pending.pop()[0].wait()
        if self.ctx["use_dist"]:
            dist.barrier()
        self.torch.cuda.synchronize(self.ctx["device"])

    def timed(self, steps, warmup):
        """-> seconds for `steps` steps, MAX over ranks (barrier + device synchronise on both sides)"""
        for _ in range(warmup):
            self.step()
        self.fence()
        t0 = time.perf_counter()
        for _ in range(steps):
            self.step()
        self.fence()
        return max_over_ranks(time.perf_counter() - t0, self.ctx)

    def gather_alone(self, reps=3):
        """the all-gather of one step's hard decisions by itself (nothing overlapped): barrier, synchronise, gather,
        synchronise -- MAX over ranks per repetition; -> {"ms": best, "ms_all": [...], bytes, rates} or None"""
        ctx = self.ctx
        if not ctx["use_dist"] or self.res is None or self.res.packed_bits is None:
            return None
        from sharding import all_gather_hard_decisions
        src = self.res.packed_bits if ctx["backend"] == "nccl" else self.res.packed_bits.cpu()
        total = self.B * ctx["world"]
        times = []
        for _ in range(reps + 1):                                   # first repetition: communicator warm-up, dropped
            self.fence()
            t0 = time.perf_counter()
            if ctx["backend"] == "nccl":                           # the collective itself, also with one rank (RCCL's local copy)
                import torch.distributed as dist
                out = self.torch.empty((total, self.nbytes), dtype=self.torch.uint8, device=ctx["device"])
                dist.all_gather_into_tensor(out, src)
            else:
                out = all_gather_hard_decisions(src, total)
            self.torch.cuda.synchronize(ctx["device"])
            times.append(1e3 * max_over_ranks(time.perf_counter() - t0, ctx))
        del out
        times = times[1:]
        per_rank = self.B * self.nbytes
        best = min(times)
        w = ctx["world"]
        return {"ms": best, "ms_all": times, "bytes_per_rank": per_rank, "bytes_gathered": per_rank * w,
                # algorithm bandwidth: gathered bytes / time; bus bandwidth (what one rank's links carry): x (w-1)/w
                "algbw_GBps": per_rank * w / (best * 1e-3) / 1e9,
                "busbw_GBps": per_rank * (w - 1) / (best * 1e-3) / 1e9,
                "method": "barrier + synchronise, ONE all_gather_into_tensor, synchronise; max over ranks, best of %d" % reps}

    def verify(self):
        """after the last step, on EVERY rank: this rank's shard inside `gathered` equals its local packed_bits, and the
        byte sums of all shards (all-reduced) equal the byte sum of the gathered array -> dict (raises on mismatch)"""
        import torch.distributed as dist
        ctx, torch = self.ctx, self.torch
        res, gathered = self.res, self.gathered
        T = self.T
        its = res.iterations
        assert int(its.min().item()) >= 1 and int(its.max().item()) <= T
        if not self.early:
            assert int(its.min().item()) == T                         # nothing skipped: every iteration ran
        out = {"converged_fraction": float(res.success.float().mean().item())}
        if ctx["use_dist"] and gathered is not None:
            world, rank = ctx["world"], ctx["rank"]
            assert tuple(gathered.shape) == (world * self.B, self.nbytes)
            mine = gathered[rank * self.B:(rank + 1) * self.B]
            local = res.packed_bits.to(mine.device)
            own_ok = bool(torch.equal(mine, local))
            dev = ctx["device"] if ctx["backend"] == "nccl" else "cpu"
            sums = torch.tensor([int(local.to(torch.int64).sum().item()), int(own_ok)], dtype=torch.int64, device=dev)
            dist.all_reduce(sums, op=dist.ReduceOp.SUM)
            total_sum = int(gathered.to(torch.int64).sum().item())
            out["gather_check"] = {"own_shard_equal_on_ranks": int(sums[1].item()), "ranks": world,
                                   "sum_of_shard_checksums": int(sums[0].item()), "checksum_of_gathered": total_sum,
                                   "ok": bool(int(sums[1].item()) == world and int(sums[0].item()) == total_sum)}
            if not out["gather_check"]["ok"]:
                raise SystemExit(f"all-gather content check failed on rank {rank}: {out['gather_check']}")
        return out

    def release(self):
        self.fence()
        self.eng = self.dec = self.llr = self.res = self.gathered = None
        self.torch.cuda.empty_cache()


def max_over_ranks(x, ctx):
    if not ctx["use_dist"]:
        return float(x)
    import torch
    import torch.distributed as dist
    t = torch.tensor([x], dtype=torch.float64, device=ctx["device"] if ctx["backend"] == "nccl" else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def rank_inventory(ctx):
    """what the process group really is (not what the environment said): world size and backend as torch.distributed
    reports them, and every rank's device (index, name, PCI bus id / uuid where this torch exposes them)"""
    import torch
    import torch.distributed as dist
    p = torch.cuda.get_device_properties(ctx["device"])
    me = {"rank": ctx["rank"], "device_index": ctx["device"].index, "name": p.name, "pid": os.getpid(),
          "hip_visible_devices": os.environ.get("HIP_VISIBLE_DEVICES") or os.environ.get("ROCR_VISIBLE_DEVICES")}
    for k in ("pci_bus_id", "pci_device_id", "pci_domain_id", "uuid", "gcnArchName"):
        v = getattr(p, k, None)
        if v is not None:
            me[k] = v if isinstance(v, (int, float)) else str(v)
    if not ctx["use_dist"]:
        return {"world_size": 1, "backend": None, "ranks": [me]}
    everyone = [None] * dist.get_world_size()
    dist.all_gather_object(everyone, me)
    return {"world_size": dist.get_world_size(), "backend": dist.get_backend(), "ranks": everyone}


def sharded_leg(workload, B, strong, ctx, args):
    """BASELINE config 5 inside the multi-rank line: a short sharded measurement (weak: B per GPU; strong: B total), every
    step ending in the all-gather; collective on every rank, the dict matters on rank 0"""
    world = ctx["world"]
    per = B // world if strong else B
    run = ShardedRun(workload, per, ctx, early=False, overlap=not args.no_overlap, snr_db=args.snr_db)
    steps = max(args.leg_steps, 1)
    elapsed = run.timed(steps, 2)          # two warm-up steps: outputs alternate between two sets of buffers (the previous step's
                                           # are still referenced while the next decode allocates), both must exist before timing
    ver = run.verify()
    ga = run.gather_alone()
    info = run.eng.info()
    leg = {"workload": f"{run.desc}, {run.T} iterations, batch {per}/GPU x {world} GPUs, SNR {args.snr_db} dB, fixed-iteration "
                       f"flooding decode + all-gather of the bit-packed hard decisions",
           "scaling": "strong" if strong else "weak", "dtype": run.dtype, "batch_per_gpu": per, "global_batch": per * world,
           "steps": steps, "ms_per_step": 1e3 * elapsed / steps, "value": per * world * steps / elapsed, "unit": "codewords/s",
           "engine": info, "allgather": ga, **ver}
    run.release()
    return leg


def main():
    args = parse_args()
    env_world = os.environ.get("WORLD_SIZE")
    if args.gpus > 1 and env_world is None:
        return spawn_ranks(args.gpus, sys.argv[1:])          # before any GPU call in this process

    import numpy as np  # noqa: F401
    import torch
    import torch.distributed as dist

    # stdout carries exactly ONE line, the JSON result: libraries that print there (RCCL writes a five-line version
    # banner to stdout when its first communicator comes up) are sent to stderr for the duration of the run
    sys.stdout.flush()
    result_fd = os.dup(1)
    os.dup2(2, 1)

    world = int(env_world or "1")
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a ROCm GPU: the decode path has no CPU fallback")
    device = torch.device("cuda", local_rank % torch.cuda.device_count())
    torch.cuda.set_device(device)
    backend = os.environ.get("LDPC_BENCH_BACKEND", "nccl")           # "nccl" is RCCL on ROCm; gloo = 1-GPU rehearsal
    use_dist = world > 1 or args.force_dist
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29577")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=device)
        else:
            dist.init_process_group(backend=backend)
        world, rank = dist.get_world_size(), dist.get_rank()        # from here on: what the process group says
        if world != args.gpus:
            raise SystemExit(f"--gpus {args.gpus} but the process group has {world} ranks")
    ctx = {"device": device, "rank": rank, "world": world, "backend": backend, "use_dist": use_dist}

    gname, T, default_batch, dtype, desc = WORKLOADS[args.workload]
    B = args.batch or default_batch
    if args.strong:                                              # fixed total work, B / world per GPU
        if B % world:
            raise SystemExit(f"--strong: total batch {B} is not divisible by {world} GPUs")
        B //= world
    early = bool(args.early_stop)
    main_run = ShardedRun(args.workload, B, ctx, early=early, overlap=not args.no_overlap, snr_db=args.snr_db)
    eng, dec, code, g, llr = main_run.eng, main_run.dec, main_run.code, main_run.g, main_run.llr
    want_post = main_run.want_post

    elapsed = main_run.timed(args.steps, args.warmup)
    ms_per_step = 1e3 * elapsed / args.steps
    value = B * world * args.steps / elapsed
    ver = main_run.verify()                                       # nothing skipped; gathered == the ranks' shards
    inventory = rank_inventory(ctx)
    allgather = main_run.gather_alone()

    out = {
        "metric": METRIC,
        "value": value, "unit": "codewords/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "strong" if args.strong else "weak", "vs_baseline": None,
        "dtype": dtype, "data": "synthetic",
        "config": {"workload": f"{desc}, {T} iterations, batch {B}/GPU, SNR {args.snr_db} dB, "
                               f"{'early-stop' if early else 'fixed-iteration'} flooding decode",
                   "graph": gname, "n": g.n, "m": g.m, "edges": g.E, "iterations": T,
                   "batch_per_gpu": B, "global_batch": B * world, "parallelism": f"dp{world}",
                   "collective": ("all_gather(bit-packed hard decisions), " + backend) if use_dist else "none",
                   "overlap": (not args.no_overlap) if use_dist else None,
                   "converged_fraction": ver["converged_fraction"],
                   "dtype_note": "the reference's BasicMinSumDecoder computes in float64 (numpy default); the benchmark "
                                 "config is defined on fp32 (BASELINE.json north_star, SURVEY 8d) -- the float64 kernels "
                                 "run as the `basic_f64` leg; fp32-vs-fp64 decision mismatch rates are in BASELINE.md"},
    }
    if use_dist:
        out["distributed"] = {**inventory, "allgather": allgather, "gather_check": ver.get("gather_check"),
                              "step": "decode of this rank's shard, then all_gather_into_tensor of uint8[B, ceil(n/8)]; "
                                      + ("the gather of step k overlaps the decode of step k+1 (joined one step later)"
                                         if not args.no_overlap else "joined before the next decode (--no-overlap)")}

    # BASELINE config 5 as written, inside the multi-rank line (every rank takes part): weak 32768/GPU and strong 262144 total
    sharded = {}
    default_line = args.workload == "basic" and not args.batch and not early and not args.strong
    if use_dist and not args.no_legs and default_line:
        main_run.fence()
        llr = None
        main_run.res = main_run.gathered = None
        main_run.llr = None                                       # make room: the strong leg holds 262144 / N codewords per GPU
        torch.cuda.empty_cache()
        sharded["wrcq_dvbs2_weak"] = sharded_leg("wrcq_dvbs2", WORKLOADS["wrcq_dvbs2"][2], False, ctx, args)
        total = args.config5_total
        if total % world == 0 and total // world <= args.config5_max_per_gpu:
            sharded["wrcq_dvbs2_strong"] = sharded_leg("wrcq_dvbs2", total, True, ctx, args)
        else:
            sharded["wrcq_dvbs2_strong"] = {"skipped": f"{total} codewords over {world} GPUs = {total / world:.0f} per GPU "
                                                       f"(limit {args.config5_max_per_gpu})"}
        llr = make_llr(B, g.n, args.snr_db, 1234 + rank, device, torch.float64 if dtype == "f64" else torch.float32)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()                              # rank 0's single-GPU extras below need no group

    if rank == 0:
        if sharded:
            out["sharded_workloads"] = sharded
        out["config"]["engine"] = eng.info()
        reps = max(args.sweep_reps, 1)
        bm = byte_model(args.workload, g, T, B)
        copy_gbs = copy_ceiling(device, torch)
        out["measured_ceilings"] = {"copy_GBps": copy_gbs, "fill_GBps": fill_ceiling(device, torch),
                                    "note": "1 GiB device copy (read + write bytes) and 1 GiB device fill (write only), best of 6"}
        run = lambda: eng.decode(llr, early_stop=early, want_bits=True, want_posterior=want_post)
        if args.workload == "rcq_layered":
            out["roofline"] = layered_roofline(eng, g, T, B, event_ms(run, max(reps // 4, 2), torch))
        elif eng.info()["engine"] == "resident":
            # dominant kernel = ldpc::resident_decode (the whole decode is this one launch), timed with HIP events
            ms = event_ms(run, reps, torch)
            out["roofline"] = resident_roofline(eng, args.workload, g, T, B, ms, copy_gbs)
            if not args.no_stream_leg:
                # the HBM-bound engine (used for codes that do not fit LDS), same workload: the north_star's CN->VN sweep
                os.environ["LDPC_ENGINE_MODE"] = "stream"
                try:
                    eng_s, dec_s, _ = build_decoder(args.workload, device)
                finally:
                    os.environ.pop("LDPC_ENGINE_MODE", None)
                run_s = lambda: eng_s.decode(llr, early_stop=early, want_bits=True, want_posterior=want_post)
                run_s()
                ms_s = event_ms(run_s, 3, torch)
                out["stream_engine"] = {"ms_per_step": ms_s, "value": B / (ms_s * 1e-3), "engine": eng_s.info(),
                                        "roofline": stream_roofline(eng_s, args.workload, g, T, B, llr, reps, copy_gbs, torch)}
                del eng_s, dec_s
        else:
            out["roofline"] = stream_roofline(eng, args.workload, g, T, B, llr, reps, copy_gbs, torch)
        out["decode_algorithmic_GBps"] = bm["decode"] / (ms_per_step * 1e-3) / 1e9
        if not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args.workload, dec, code, args.snr_db)     # rank 0's host cores, any N
        if not use_dist and not args.no_legs and default_line:
            # the other BASELINE configs, bounded (a few steps each), so that one driver run carries every config
            del llr
            main_run.release()
            out["workloads"] = {}
            for w in LEGS:
                out["workloads"][w] = measure_leg(w, device, args.snr_db, max(args.leg_steps, 1), reps, copy_gbs,
                                                  not args.no_cpu_baseline, torch)
        sys.stdout.flush()
        os.dup2(result_fd, 1)
        print(json.dumps(out), flush=True)
        os.dup2(2, 1)


if __name__ == "__main__":
    main()
