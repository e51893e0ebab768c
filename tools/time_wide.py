#!/usr/bin/env python3
"""Wide checks (degree 64): the streaming engine's cn_sweep_wide (a check split over the four waves of a block, read once)
against the one-wave kernel's re-read path (LDPC_HIP_LIB=build_variants/nowide.so, built with -DLDPC_NO_WIDE_KERNEL), and the
LDS-resident engine with the check split over lane groups.  One JSON line per engine mode."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import numpy as np  # noqa: E402
import torch  # noqa: E402


def regular_wide_code(n=2048, m=128, dv=4, seed=7):
    """(dv, dc = n*dv/m) regular code from dv random layers, each giving every check n/m variables; a variable that a layer
    would connect to one of its checks a second time swaps places with a random other variable until none is left"""
    from ldpc_decoder import LDPCCode
    rng = np.random.default_rng(seed)
    per = n // m
    H = np.zeros((m, n), dtype=np.int8)
    for _ in range(dv):
        perm = rng.permutation(n)                       # perm[pos] = variable; check of position pos = pos // per
        for _ in range(100000):
            chk = np.arange(n) // per
            bad = np.flatnonzero(H[chk, perm] != 0)
            if bad.size == 0:
                break
            for pos in bad:
                other = int(rng.integers(0, n))
                perm[pos], perm[other] = perm[other], perm[pos]
        H[np.arange(n) // per, perm] = 1
    assert H.sum(0).min() == dv and H.sum(0).max() == dv and H.sum(1).min() == per * dv
    return LDPCCode(n=n, k=n - m, H=H.astype(np.int64), max_iterations=10)


def main():
    from ldpc_decoder import BasicMinSumDecoder
    dev = torch.device("cuda", 0)
    code = regular_wide_code()
    g = code.tanner_graph()
    B, T = 32768, 10
    llr = bench.make_llr(B, code.n, 2.0, 1234, dev)
    dec = BasicMinSumDecoder(code, 0.7)
    eng = dec._engine(torch.float32, dev)
    for mode in ("auto", "stream"):
        eng.set_mode(mode)
        for _ in range(2):
            eng.decode(llr, early_stop=False, want_posterior=False)
        ms = bench.event_ms(lambda: eng.decode(llr, early_stop=False, want_posterior=False), 5, torch)
        out = {"lib": os.path.basename(os.environ.get("LDPC_HIP_LIB", "default")), "mode": mode, "engine": eng.info()["engine"],
               "code": f"regular dv=4 dc={int(code.H.sum(1).max())} n={code.n} m={g.m} E={g.E}", "B": B, "T": T,
               "decode_ms": ms, "Mcw_s": B / ms / 1e3}
        if eng.info()["engine"] == "stream":
            eng.decode(llr, early_stop=False, want_posterior=False)
            for _ in range(3):
                eng.debug_sweep(B, 0, 1)
            cn = bench.event_ms(lambda: eng.debug_sweep(B, 0, 1), 20, torch)
            out["cn_sweep_ms"] = cn                      # with the wide kernel: skip-all cn_sweep launch + cn_sweep_wide
            out["cn_GBs_algorithmic"] = 8 * g.E * B / cn / 1e6
        print(json.dumps(out))


if __name__ == "__main__":
    main()
