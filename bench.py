#!/usr/bin/env python3
"""
bench.py -- decoded codewords/s at fixed iterations + HBM roofline of the dominant kernel.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload basic|neural2d|rcq|wrcq_dvbs2]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

A step = ONE decode of one batch of synthetic LLRs (already resident in HBM) through the
hot path: layout change, T x (check sweep + variable sweep), syndrome, hard decisions out.
Default workload = BASELINE.json configs[1]: (1998,1512) code, BasicMinSumDecoder factor 0.7,
fp32, 10 iterations, batch 65536 per GPU, SNR 2.0 dB, fixed iterations (early_stop=False).
N > 1: weak scaling, every rank decodes its own 65536 codewords and the step ends with the
RCCL all-gather of the bit-packed hard decisions; value = all ranks' codewords / max-rank time.

Two engines implement the path (identical results): the LDS-resident fused kernel (codes whose
state fits LDS, e.g. the (1998,1512) code) and the HBM-streaming sweep kernels (any code).  The
step uses whichever the library picks (config.engine); with the resident engine the streaming
engine is measured too and reported under "stream_engine".

Prints ONE JSON line (rank 0) with the driver's fields plus
  roofline     : the dominant kernel timed live with HIP events on its stream; ALGORITHMIC bytes
                 of SURVEY 8d (CN sweep: 8E per codeword fp32, 5E RCQ; whole decode:
                 T(16E+4n)+8n) / time vs 8 TB/s
  cpu_baseline : the CPU oracle (C port of the reference loops) on a bounded sample
"""
import argparse
import importlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
PKG_DIR = os.path.join(ROOT, "implementation-of-neural-ldpc-decoders-with-degree-specific-weight-sharing-and-rcq-quantization_amd")
for p in (PKG_DIR, ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

HBM_PEAK_GBS = 8000.0      # MI355X HBM3E spec (MI355X_MICROARCH.md; ~6300 GB/s achievable)
QP = [(3.0, 1.3), (5.0, 1.3), (7.0, 1.3)]

WORKLOADS = {
    # name: (graph, iterations, default batch per GPU, description)
    "basic": ("ira_1998_1512", 10, 65536, "(1998,1512) IRA code, BasicMinSumDecoder factor=0.7, fp32"),
    "neural2d": ("ira_1998_1512", 10, 65536, "(1998,1512) IRA code, Neural2DMinSumDecoder type 2, fp32"),
    "rcq": ("ira_1998_1512", 10, 65536, "(1998,1512) IRA code, RCQMinSumDecoder bc=3 bv=8, 3 quantisers"),
    "wrcq_dvbs2": ("dvbs2_like_16200_7200", 20, 32768, "(16200,7200) DVB-S2-like code, WeightedRCQDecoder type 2 bc=3"),
}


def synthetic_tables(dec, seed=4321):
    """'pretrained' weights stand-in: beta ~ U(0.5,1), alpha ~ U(0.8,1.2)"""
    rng = np.random.default_rng(seed)
    with torch.no_grad():
        for k in sorted(dec.beta_weights.keys()):
            dec.beta_weights[k].fill_(float(np.float32(rng.uniform(0.5, 1.0))))
        for k in sorted(dec.alpha_weights.keys()):
            dec.alpha_weights[k].fill_(float(np.float32(rng.uniform(0.8, 1.2))))


def build_decoder(workload, device):
    """-> (engine, host decoder, graph, oracle call for the CPU baseline)"""
    import codes
    from ldpc_decoder import BasicMinSumDecoder
    from neural_2d_decoder import Neural2DMinSumDecoder
    from rcq_decoder import RCQMinSumDecoder, WeightedRCQDecoder
    gname, T, _, _ = WORKLOADS[workload]
    code = codes.load_code(gname, max_iterations=T)
    if workload == "basic":
        dec = BasicMinSumDecoder(code, factor=0.7)
        eng = dec._engine(torch.float32, device)
    elif workload == "neural2d":
        dec = Neural2DMinSumDecoder(code, weight_sharing_type=2, max_iterations=T)
        synthetic_tables(dec)
        eng = dec._get_engine(device)
    elif workload == "rcq":
        dec = RCQMinSumDecoder(code, bc=3, bv=8, quantizer_params=QP, max_iterations=T)
        eng = dec._get_engine(device)
    else:
        dec = WeightedRCQDecoder(code, bc=3, bv=8, quantizer_params=QP, weight_sharing_type=2, max_iterations=T)
        synthetic_tables(dec)
        eng = dec._get_engine(device)
    return eng, dec, code


def make_llr(batch, n, snr_db, seed, device):
    """all-zero codeword, decoder convention (+LLR = bit 0): llr = 2(1 + sigma z)/sigma^2"""
    s2 = 10.0 ** (-snr_db / 10.0)
    gen = torch.Generator(device=device)
    gen.manual_seed(seed)
    z = torch.randn((batch, n), generator=gen, device=device, dtype=torch.float32)
    return (2.0 * (1.0 + (s2 ** 0.5) * z) / s2).contiguous()


def resident_lds_model(g, T, B, G, ms):
    """What bounds the fused kernel itself: LDS traffic of its phases against the LDS rates of MI355X_MICROARCH.md
    ("LDS": ds_read_b64 256 B/clk/CU, ds_write_b64 ~85 B/clk/CU, 256 CUs at 2.4 GHz), conflict-free.  Slots are
    4*G bytes; per codeword group and iteration the check phase reads every slot twice and writes it once, the
    variable phase reads every slot and its LLR once and writes every slot once."""
    slot = 4 * G
    groups = (B + G - 1) // G
    reads = groups * slot * (T * 2 * g.E + T * (g.E + g.n))          # bytes
    writes = groups * slot * (T * g.E + T * g.E)
    rd_peak, wr_peak = 256 * 256 * 2.4e9, 85 * 256 * 2.4e9             # B/s, all CUs
    t_min = reads / rd_peak + writes / wr_peak
    return {"bound": "lds", "lds_read_bytes": reads, "lds_write_bytes": writes, "lds_min_ms": t_min * 1e3,
            "frac": t_min * 1e3 / ms,
            "note": "conflict-free LDS time of the iteration phases / measured kernel time; the variable phase's gathers "
                    "and scatters take ~2 LDS passes per instruction, the rest is VALU issue and barriers"}


def cpu_baseline(workload, dec, code, snr_db, budget_s=12.0):
    """The CPU oracle (oracle/ldpc_oracle.c, a C port of the reference's loops; the Python
    reference itself cannot travel to the GPU box) on a bounded sample of the same workload."""
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
        x = (2.0 * (1.0 + np.sqrt(s2) * rng.standard_normal((cnt, g.n))) / s2).astype(np.float32)
        t0 = time.perf_counter()
        if workload == "basic":
            oracle.basic_minsum(og, x, 0.7, T, early_stop=False, dtype=np.float32)
        elif workload == "neural2d":
            oracle.neural2d(og, x, 2, T, {k: float(v.item()) for k, v in dec.beta_weights.items()},
                            {k: float(v.item()) for k, v in dec.alpha_weights.items()}, early_stop=False)
        elif workload == "rcq":
            oracle.rcq(og, x, 3, QP, T, early_stop=False)
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


def main():
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
    ap.add_argument("--force-dist", action="store_true",
                    help="initialise the process group and run the all-gather path even with one rank (checks the RCCL plumbing on a single GPU)")
    ap.add_argument("--strong", action="store_true",
                    help="strong scaling: --batch (default: the workload's) is the TOTAL over all GPUs, split evenly")
    args = ap.parse_args()

    # stdout carries exactly ONE line, the JSON result: libraries that print there (RCCL writes a five-line version
    # banner to stdout when its first communicator comes up) are sent to stderr for the duration of the run
    sys.stdout.flush()
    result_fd = os.dup(1)
    os.dup2(2, 1)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch multi-GPU runs with torch.distributed.run (one process per GPU)")
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

    gname, T, default_batch, desc = WORKLOADS[args.workload]
    B = args.batch or default_batch
    if args.strong:                                              # fixed total work, B / world per GPU
        if B % world:
            raise SystemExit(f"--strong: total batch {B} is not divisible by {world} GPUs")
        B //= world
    eng, dec, code = build_decoder(args.workload, device)
    g = code.tanner_graph()
    llr = make_llr(B, g.n, args.snr_db, 1234 + rank, device)
    want_post = args.workload in ("neural2d", "wrcq_dvbs2")      # those decoders return the posterior
    early = bool(args.early_stop)

    from sharding import all_gather_hard_decisions
    nbytes = (g.n + 7) // 8
    pending = []          # [(work, gathered, packed_src)] -- the previous step's all-gather, still in flight

    def step():
        """decode this rank's shard, then all-gather the bit-packed hard decisions.  The gather of step k
        runs on RCCL's stream while step k+1 decodes (separate buffers), and is joined one step later."""
        res = eng.decode(llr, early_stop=early, want_bits=True, want_posterior=want_post, want_packed=use_dist)
        gathered = None
        if use_dist:
            if backend != "nccl":                                  # rehearsal path: host-side gather
                gathered = all_gather_hard_decisions(res.packed_bits.cpu(), B * world)
            elif args.no_overlap:
                gathered = all_gather_hard_decisions(res.packed_bits, B * world)
            else:
                if pending:
                    pending.pop()[0].wait()
                gathered = torch.empty((world * B, nbytes), dtype=torch.uint8, device=device)
                work = dist.all_gather_into_tensor(gathered, res.packed_bits, async_op=True)
                pending.append((work, gathered, res.packed_bits))
        return res, gathered

    def fence():
        while pending:
            pending.pop()[0].wait()
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize(device)

    for _ in range(args.warmup):
        step()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        res, gathered = step()
    fence()
    elapsed = time.perf_counter() - t0
    if use_dist:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    ms_per_step = 1e3 * elapsed / args.steps
    value = B * world * args.steps / elapsed

    # sanity on the last step's outputs (nothing skipped): all iterations ran, outputs are binary
    its = res.iterations
    assert int(its.min().item()) >= 1 and int(its.max().item()) <= T
    if not early:
        assert int(its.min().item()) == T
    frac_ok = float(res.success.float().mean().item())

    out = {
        "metric": "decoded codewords/sec at fixed iters; achieved HBM GB/s vs peak",
        "value": value, "unit": "codewords/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "strong" if args.strong else "weak", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"{desc}, {T} iterations, batch {B}/GPU, SNR {args.snr_db} dB, "
                               f"{'early-stop' if early else 'fixed-iteration'} flooding decode",
                   "graph": gname, "n": g.n, "m": g.m, "edges": g.E, "iterations": T,
                   "batch_per_gpu": B, "global_batch": B * world, "parallelism": f"dp{world}",
                   "collective": "all_gather(bit-packed hard decisions)" if world > 1 else "none",
                   "converged_fraction": frac_ok},
    }

    if rank == 0:
        out["config"]["engine"] = eng.info()
        rcq_like = args.workload in ("rcq", "wrcq_dvbs2")
        per_cw_decode = T * ((10 if rcq_like else 16) * g.E + 4 * g.n) + 8 * g.n     # SURVEY 8d, whole decode
        bytes_cn = (5 if rcq_like else 8) * g.E * B            # CN sweep: read v2c 4E + write c2v 4E (1E as codes)
        bytes_vn = ((5 if rcq_like else 8) * g.E + 4 * g.n) * B
        reps = max(args.sweep_reps, 1)
        tf = os.path.join(ROOT, "profiles", "traffic.json")
        traffic_db = {}
        if os.path.exists(tf):
            try:
                traffic_db = json.load(open(tf)).get(args.workload, {})
            except Exception:
                traffic_db = {}

        def time_sweeps(engine):
            """CN / VN sweep kernels of the streaming engine, timed live with HIP events on the stream
            they are launched on (torch's current stream)"""
            engine.decode(llr, early_stop=False, want_posterior=False)          # leaves valid state in the workspace
            times = {}
            for which, name in ((0, "cn_sweep"), (1, "vn_sweep")):
                for _ in range(3):
                    engine.debug_sweep(B, which, 1)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(reps):
                    engine.debug_sweep(B, which, 1)
                e1.record()
                e1.synchronize()
                times[name] = e0.elapsed_time(e1) / reps                        # ms per launch
            return times

        def copy_ceiling():
            """Measured HBM ceiling of this box (SURVEY 8d asks for it beside the 8 TB/s spec): a 1 GiB -> 1 GiB
            device copy (read + write, far beyond the 256 MB Infinity Cache), best of 5, GB/s"""
            src = torch.empty(1 << 28, dtype=torch.float32, device=device).normal_()
            dst = torch.empty_like(src)
            best = 0.0
            for _ in range(6):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                dst.copy_(src)
                e1.record()
                e1.synchronize()
                best = max(best, 2 * src.numel() * 4 / (e0.elapsed_time(e1) * 1e-3) / 1e9)
            del src, dst
            return best

        copy_gbs = copy_ceiling()

        def stream_roofline(engine):
            times = time_sweeps(engine)
            ach = bytes_cn / (times["cn_sweep"] * 1e-3) / 1e9
            return {"bound": "hbm", "kernel": "ldpc::cn_sweep (check-node / CN->VN message sweep, streaming engine)",
                    "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
                    "measured_copy_GBps": copy_gbs, "frac_of_measured_copy": ach / copy_gbs,
                    "traffic": traffic_db.get("cn_sweep_bytes_per_launch"),
                    "algorithmic_bytes_per_launch": bytes_cn, "ms_per_launch": times["cn_sweep"],
                    "vn_sweep": {"ms_per_launch": times["vn_sweep"],
                                 "achieved": bytes_vn / (times["vn_sweep"] * 1e-3) / 1e9,
                                 "algorithmic_bytes_per_launch": bytes_vn}}

        if eng.info()["engine"] == "resident":
            # dominant kernel = ldpc::resident_decode (the whole decode is this one launch): time it with
            # HIP events; ALGORITHMIC bytes are those of the reference formulation, which this
            # kernel does not move -- messages stay in LDS -- hence frac can exceed 1 and `traffic` << them.
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                eng.decode(llr, early_stop=early, want_bits=True, want_posterior=want_post)
            e1.record()
            e1.synchronize()
            ms = e0.elapsed_time(e1) / reps
            ach = per_cw_decode * B / (ms * 1e-3) / 1e9
            out["roofline"] = {"bound": "hbm", "kernel": "ldpc::resident_decode (fused T-iteration decode, messages resident in LDS)",
                               "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
                               "traffic": traffic_db.get("resident_decode_bytes_per_launch"),
                               "measured_copy_GBps": copy_gbs,
                               "algorithmic_bytes_per_launch": per_cw_decode * B, "ms_per_launch": ms,
                               "own_limiter": resident_lds_model(g, T, B, eng.info()["codewords_per_workgroup"], ms),
                               "note": "algorithmic bytes = T(16E+4n)+8n per codeword (10E for RCQ) of the HBM-streaming "
                                       "formulation; the fused kernel keeps messages in LDS, so frac > 1 means it beats "
                                       "the HBM roofline of that formulation; its own limiter is LDS/instruction issue"}
            if not args.no_stream_leg:
                # the HBM-bound engine (used for codes that do not fit LDS), same workload, for the record
                os.environ["LDPC_ENGINE_MODE"] = "stream"
                try:
                    eng_s, dec_s, _ = build_decoder(args.workload, device)
                finally:
                    os.environ.pop("LDPC_ENGINE_MODE", None)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                eng_s.decode(llr, early_stop=early, want_bits=True, want_posterior=want_post)
                e0.record()
                for _ in range(3):
                    eng_s.decode(llr, early_stop=early, want_bits=True, want_posterior=want_post)
                e1.record()
                e1.synchronize()
                out["stream_engine"] = {"ms_per_step": e0.elapsed_time(e1) / 3,
                                        "value": B / (e0.elapsed_time(e1) / 3 * 1e-3), "roofline": stream_roofline(eng_s)}
                del eng_s, dec_s
        else:
            out["roofline"] = stream_roofline(eng)
        out["decode_algorithmic_GBps"] = per_cw_decode * B / (ms_per_step * 1e-3) / 1e9
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args.workload, dec, code, args.snr_db)
        sys.stdout.flush()
        os.dup2(result_fd, 1)
        print(json.dumps(out), flush=True)
        os.dup2(2, 1)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
