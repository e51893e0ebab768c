import sys, os, json, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import bench, torch, codes
from ldpc_decoder import BasicMinSumDecoder
dev = torch.device("cuda", 0)
code = codes.load_code("ira_1998_1512", max_iterations=10)
eng = BasicMinSumDecoder(code, 0.7)._engine(torch.float32, dev)
B = 1 << 20
llr = bench.make_llr(B, code.n, 4.5, 7, dev)
idx = torch.tensor([0, 1, 65535, 65536, 524287, 524288, B - 2, B - 1], device=dev)
small = eng.decode(llr[idx].contiguous(), early_stop=True)
for mode in ("auto", "stream"):
    eng.set_mode(mode)
    torch.cuda.synchronize(); t0 = time.time()
    res = eng.decode(llr, early_stop=True, want_posterior=False, want_packed=True)
    torch.cuda.synchronize(); dt = time.time() - t0
    ok = torch.equal(res.bits[idx], small.bits) and torch.equal(res.iterations[idx], small.iterations) and torch.equal(res.success[idx], small.success)
    print(json.dumps({"mode": mode, "engine": eng.info()["engine"], "B": B, "seconds": dt, "Mcw_s": B / dt / 1e6, "rows_match_small_batch": bool(ok),
                      "converged": float(res.success.float().mean()), "ws_GB": eng.workspace_bytes(B) / 2**30}))
    del res
