#!/usr/bin/env python3
"""One codeword of the (16200,7200) code on the streaming engine, device-resident buffers: eager launches against a captured
graph replayed (what a launch-bound caller gains from INTEGRATION.md section 2)."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import torch  # noqa: E402
import codes  # noqa: E402
from rcq_decoder import WeightedRCQDecoder  # noqa: E402

dev = torch.device("cuda", 0)
code = codes.load_code("dvbs2_like_16200_7200", max_iterations=20)
dec = WeightedRCQDecoder(code, 3, 8, [(3.0, 1.3), (5.0, 1.3), (7.0, 1.3)], weight_sharing_type=2, max_iterations=20)
eng = dec._get_engine(dev)
for B in (1, 64):
    x = bench.make_llr(B, code.n, 2.0, 1, dev)
    for es in (False, True):
        def eager():
            eng.decode(x, early_stop=es)
        for _ in range(5):
            eager()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(50):
            eager()
            torch.cuda.synchronize()
        t_eager = (time.perf_counter() - t0) / 50 * 1e6
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            eager()
        torch.cuda.current_stream(dev).wait_stream(side)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            res = eng.decode(x, early_stop=es)
        for _ in range(5):
            graph.replay()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(50):
            graph.replay()
            torch.cuda.synchronize()
        t_graph = (time.perf_counter() - t0) / 50 * 1e6
        print(json.dumps({"batch": B, "early_stop": es, "eager_us": t_eager, "graph_replay_us": t_graph}))
