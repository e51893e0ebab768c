#!/usr/bin/env python3
"""Time one decode and the individual CN / VN sweeps of a workload with HIP events.
Used for A/B timing of kernel variants: LDPC_HIP_LIB=<variant .so> python tools/time_sweeps.py"""
import argparse, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="basic")
    ap.add_argument("--batch", type=int, default=0)
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--packed", action="store_true", help="also ask for the bit-packed hard decisions (multi-GPU wire format)")
    ap.add_argument("--tag", default=os.environ.get("LDPC_HIP_LIB", "default"))
    ap.add_argument("--mode", default="", help="engine mode: auto | stream | sweeps | resident")
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    if a.workload == "basic_small":           # 2/3-scale copy of the (1998,1512) code: fits 3 workgroups per CU
        import codes
        from ldpc_decoder import BasicMinSumDecoder, LDPCCode
        g0 = codes.generate_ira_code(n=1332, m=324, info_degrees={8: 144, 3: 864}, check_degrees=None, seed=1332)
        code = LDPCCode.from_graph(g0, k=1008, max_iterations=10)
        dec = BasicMinSumDecoder(code, 0.7)
        eng = dec._engine(torch.float32, dev)
        T, B = 10, a.batch or 65536
    else:
        gname, T, B0 = bench.WORKLOADS[a.workload][:3]
        B = a.batch or B0
        eng, dec, code = bench.build_decoder(a.workload, dev)
    if a.mode:
        eng.set_mode(a.mode)
    llr = bench.make_llr(B, code.n, 2.0, 1234, dev, torch.float64 if a.workload == "basic_f64" else torch.float32)
    for _ in range(2):
        eng.decode(llr, early_stop=False, want_posterior=False, want_packed=a.packed)
    torch.cuda.synchronize()
    ev = lambda: torch.cuda.Event(enable_timing=True)
    e0, e1 = ev(), ev()
    e0.record()
    for _ in range(5):
        eng.decode(llr, early_stop=False, want_posterior=False, want_packed=a.packed)
    e1.record(); e1.synchronize()
    out = {"tag": os.path.basename(a.tag), "mode": a.mode, "workload": a.workload, "B": B, "decode_ms": e0.elapsed_time(e1) / 5,
           "engine": eng.info()}
    out["Mcw_s"] = B / out["decode_ms"] / 1e3
    out["ns_per_edge_iter"] = out["decode_ms"] * 1e6 / (B * T * code.tanner_graph().E)
    if eng.info()["engine"] != "stream":
        print(json.dumps(out))
        return
    for which, name in ((0, "cn_ms"), (1, "vn_ms")):
        for _ in range(3):
            eng.debug_sweep(B, which, 1)
        e0, e1 = ev(), ev()
        e0.record()
        for _ in range(a.reps):
            eng.debug_sweep(B, which, 1)
        e1.record(); e1.synchronize()
        out[name] = e0.elapsed_time(e1) / a.reps
    g = code.tanner_graph()
    out["E"] = g.E
    rcq = a.workload in ("rcq", "wrcq_dvbs2")
    out["cn_GBs"] = (5 if rcq else 8) * g.E * B / out["cn_ms"] / 1e6
    out["vn_GBs"] = ((5 if rcq else 8) * g.E + 4 * g.n) * B / out["vn_ms"] / 1e6
    out["Mcw_s"] = B / out["decode_ms"] / 1e3
    print(json.dumps(out))


if __name__ == "__main__":
    main()
