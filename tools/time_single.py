#!/usr/bin/env python3
"""Latency of the reference's own call shape: ONE codeword per call (CPU tensor in, CPU tensors out)."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import numpy as np, torch  # noqa: E402
import codes  # noqa: E402
from ldpc_decoder import BasicMinSumDecoder  # noqa: E402
from neural_2d_decoder import Neural2DMinSumDecoder  # noqa: E402
from rcq_decoder import RCQMinSumDecoder  # noqa: E402

code = codes.load_code("ira_1998_1512", max_iterations=10)
llr32 = bench.make_llr(1, code.n, 5.0, 1, torch.device("cuda", 0))[0].cpu()
cases = {"BasicMinSumDecoder.decode(np.float64[n])": (BasicMinSumDecoder(code, 0.7).decode, llr32.double().numpy()),
         "Neural2DMinSumDecoder(t[n]) no_grad": (Neural2DMinSumDecoder(code, 2, 10), llr32),
         "RCQMinSumDecoder.decode(t[n])": (RCQMinSumDecoder(code, 3, 8, [(3.0, 1.3), (5.0, 1.3), (7.0, 1.3)], 10).decode, llr32)}
with torch.no_grad():
    for name, (fn, x) in cases.items():
        for _ in range(20):
            fn(x)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(200):
            fn(x)
        torch.cuda.synchronize()
        print(json.dumps({"call": name, "us_per_call": (time.perf_counter() - t0) / 200 * 1e6}))
