import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench, torch, numpy as np, codes
from ldpc_decoder import BasicMinSumDecoder
dev = torch.device("cuda", 0)
code = codes.load_code("ira_1998_1512", max_iterations=10)
dec = BasicMinSumDecoder(code, 0.7)
eng = dec._engine(torch.float64, dev)
B = 32768
llr = bench.make_llr(B, code.n, 2.0, 1234, dev).double()
for mode in ("auto", "stream"):
    eng.set_mode(mode)
    for _ in range(2): eng.decode(llr, early_stop=False, want_posterior=False)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): eng.decode(llr, early_stop=False, want_posterior=False)
    e1.record(); e1.synchronize()
    ms = e0.elapsed_time(e1) / 5
    print(json.dumps({"workload": "basic fp64 (the reference's own dtype for BasicMinSumDecoder)", "B": B, "decode_ms": ms,
                      "Mcw_s": B / ms / 1e3, "engine": eng.info()}))
