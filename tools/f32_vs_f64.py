#!/usr/bin/env python3
"""How often does the fp32 BasicMinSumDecoder path (the benchmark's arithmetic) decide differently from the float64
path (the reference's arithmetic, ldpc_decoder.py:80-81,116-120)?  Both run on the GPU engine; the float64 kernels
are bit-exact against the reference's goldens.  Prints one JSON line per case: fraction of codewords whose iteration
count / hard decisions differ, and the fraction of differing bits."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import numpy as np  # noqa: E402
import torch  # noqa: E402


def compare(dec, x64, early_stop):
    b64, s64, i64 = dec.decode(x64, early_stop=early_stop)
    b32, s32, i32 = dec.decode(x64.float(), early_stop=early_stop)
    cw_bits = (b64 != b32).any(dim=1)
    return {"codewords": int(x64.shape[0]), "early_stop": early_stop,
            "frac_codewords_iters_differ": float((i64 != i32).float().mean()),
            "frac_codewords_bits_differ": float(cw_bits.float().mean()),
            "frac_codewords_success_differ": float((s64 != s32).float().mean()),
            "frac_bits_differ": float((b64 != b32).float().mean()),
            "mean_iters_f64": float(i64.float().mean())}


def main():
    import codes
    from ldpc_decoder import BasicMinSumDecoder
    dev = torch.device("cuda", 0)
    code = codes.load_code("ira_1998_1512", max_iterations=10)
    dec = BasicMinSumDecoder(code, 0.7)
    g = np.load(os.path.join(ROOT, "tests", "golden", "ira_basic.npz"))
    x = torch.from_numpy(g["llr"]).to(dev)
    print(json.dumps({"case": "ira_basic golden (reference inputs)", **compare(dec, x, True)}))
    for snr in (2.0, 5.0):
        x = bench.make_llr(65536, code.n, snr, 1234, dev).double()
        for es in (True, False):
            print(json.dumps({"case": f"fresh 65536 codewords, {snr} dB", **compare(dec, x, es)}))


if __name__ == "__main__":
    main()
