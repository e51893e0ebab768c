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

# where the time goes: the same one-codeword decode at three depths of the stack
import torch_ops  # noqa: E402
dec = Neural2DMinSumDecoder(code, 2, 10)
eng = dec._get_engine(torch.device("cuda", 0))
h = torch_ops.engine_handle(eng)
x1 = llr32.reshape(1, -1).contiguous()
xd = x1.cuda()


def timed(fn, reps=300):
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e6


def device_only():
    eng.decode(xd, early_stop=True)
    torch.cuda.synchronize()


with torch.no_grad():
    print(json.dumps({"call": "breakdown: ldpc_decode on device-resident buffers + synchronise (kernel latency of ONE workgroup)", "us_per_call": timed(device_only)}))
    print(json.dumps({"call": "breakdown: DecodeEngine.decode_host (staged copies, direct)", "us_per_call": timed(lambda: eng.decode_host(x1))}))
    print(json.dumps({"call": "breakdown: torch.ops.ldpc.decode_host (operator dispatch on top)", "us_per_call": timed(lambda: torch.ops.ldpc.decode_host(x1, h, True, True))}))
    print(json.dumps({"call": "breakdown: weight_tables() flattening of the ParameterDicts", "us_per_call": timed(lambda: dec.weight_tables(), 200)}))
