#!/usr/bin/env python3
"""Throughput of the reference-compatible layered RCQ schedule (RCQMinSumDecoder(layered=True))."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import torch  # noqa: E402
import codes  # noqa: E402
from rcq_decoder import RCQMinSumDecoder  # noqa: E402

dev = torch.device("cuda", 0)
code = codes.load_code("ira_1998_1512", max_iterations=10)
dec = RCQMinSumDecoder(code, 3, 8, [(3.0, 1.3), (5.0, 1.3), (7.0, 1.3)], max_iterations=10, layered=True)
for B in (4096, 65536):
    llr = bench.make_llr(B, code.n, 2.0, 1234, dev)
    eng = dec._get_engine(dev)
    for _ in range(2):
        eng.decode(llr, early_stop=False, want_posterior=False)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3):
        eng.decode(llr, early_stop=False, want_posterior=False)
    e1.record(); e1.synchronize()
    ms = e0.elapsed_time(e1) / 3
    print(json.dumps({"workload": "layered RCQ (reference schedule), (1998,1512), T=10", "B": B, "decode_ms": ms, "Mcw_s": B / ms / 1e3}))
