#!/usr/bin/env python3
"""One codeword per call on the (16200,7200) code (streaming engine: ~60 small launches per decode): the reference's own call
shape on the code that does not fit LDS.  Early stop (the reference's default) and fixed T."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import numpy as np, torch  # noqa: E402
import codes  # noqa: E402
from rcq_decoder import WeightedRCQDecoder, RCQMinSumDecoder  # noqa: E402
from ldpc_decoder import BasicMinSumDecoder  # noqa: E402

QP = [(3.0, 1.3), (5.0, 1.3), (7.0, 1.3)]
code = codes.load_code("dvbs2_like_16200_7200", max_iterations=20)
dev = torch.device("cuda", 0)
for snr in (2.0, 6.0):
    llr = bench.make_llr(1, code.n, snr, 1, dev)[0].cpu()
    decs = {"WeightedRCQDecoder type 2": WeightedRCQDecoder(code, 3, 8, QP, weight_sharing_type=2, max_iterations=20),
            "BasicMinSumDecoder (float64 input)": BasicMinSumDecoder(code, 0.7)}
    with torch.no_grad():
        for name, dec in decs.items():
            x = llr.double().numpy() if name.startswith("Basic") else llr
            fn = dec.decode if name.startswith("Basic") else dec
            for es in (True, False):
                for _ in range(5):
                    out = fn(x, early_stop=es)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(50):
                    out = fn(x, early_stop=es)
                torch.cuda.synchronize()
                it = out[2]
                print(json.dumps({"call": name, "snr_db": snr, "early_stop": es, "iterations": int(it) if not hasattr(it, "numel") else int(it.reshape(-1)[0]),
                                  "us_per_call": (time.perf_counter() - t0) / 50 * 1e6}))
