#!/usr/bin/env python3
"""Time one training step (forward with saved messages + HIP backward sweeps + Adam) of the gradient path.

    python tools/time_train.py [--batch 4096] [--iters 10] [--steps 10]
Prints one JSON line: codewords/s of forward+backward, the split, and the algorithmic HBM bytes of the backward
sweeps (per codeword and iteration: check pass reads v2c 4E + gradient 4E, writes 4E; variable pass reads c2v 4E +
gradient 4E twice (two passes), writes 4E)."""
import argparse, json, os, sys
os.environ.setdefault("LDPC_TRAIN_MAX_SAVED_BYTES", str(64 << 30))     # 500 KB of messages per codeword at T = 10
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402  (puts the package on sys.path)
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=4096)
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--snr-db", type=float, default=3.0)
    a = ap.parse_args()
    import codes
    from neural_2d_decoder import Neural2DMinSumDecoder
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    code = codes.load_code("ira_1998_1512", max_iterations=a.iters)
    model = Neural2DMinSumDecoder(code, 2, a.iters)
    with torch.no_grad():
        for p in model.beta_weights.values():
            p.fill_(0.7)
        for p in model.alpha_weights.values():
            p.fill_(1.0)
    opt = torch.optim.Adam(model.parameters(), lr=1e-3)
    llr = bench.make_llr(a.batch, code.n, a.snr_db, 1234, dev)
    tgt = torch.zeros_like(llr)
    ev = lambda: torch.cuda.Event(enable_timing=True)

    def step(timers=None):
        opt.zero_grad()
        e = [ev() for _ in range(3)]
        e[0].record()
        bits, post, iters = model(llr)
        loss = F.binary_cross_entropy_with_logits(-post, tgt)
        e[1].record()
        loss.backward()
        e[2].record()
        opt.step()
        if timers is not None:
            e[2].synchronize()
            timers[0] += e[0].elapsed_time(e[1])
            timers[1] += e[1].elapsed_time(e[2])
        return loss
    for _ in range(2):
        step()
    torch.cuda.synchronize()
    timers = [0.0, 0.0]
    e0, e1 = ev(), ev()
    e0.record()
    for _ in range(a.steps):
        loss = step(timers)
    e1.record(); e1.synchronize()
    ms = e0.elapsed_time(e1) / a.steps
    g = code.tanner_graph()
    bwd_bytes = a.batch * a.iters * (12 * g.E + 20 * g.E)
    print(json.dumps({"workload": f"(1998,1512) Neural2D type 2, T={a.iters}, batch {a.batch}, training step",
                      "ms_per_step": ms, "codewords_per_s": a.batch / ms * 1e3,
                      "forward_saving_ms": timers[0] / a.steps, "backward_ms": timers[1] / a.steps,
                      "backward_algorithmic_GBps": bwd_bytes / (timers[1] / a.steps * 1e-3) / 1e9,
                      "saved_bytes_per_codeword": (2 * a.iters - 1) * g.E * 4, "loss": float(loss.item())}))


if __name__ == "__main__":
    main()
