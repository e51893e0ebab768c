"""
GPU tests at BASELINE.json's full sizes (batch 65536 on the (1998,1512) code, 32768 on the
(16200,7200) code) through size-independent properties, plus oracle spot checks of rows
sampled from the big batch:

  * row independence: rows of a 65536-batch equal the oracle's single-codeword results
  * sign symmetry of min-sum: decoding llr*(1-2c) for a codeword c gives bits XOR c and
    posterior*(1-2c), same iteration count (exact in floating point: every operation is odd)
  * success  <=>  H @ bits == 0 (mod 2); iterations within [1, T]; packed bits == bits
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
QP = [(3.0, 1.3), (5.0, 1.3), (7.0, 1.3)]


@pytest.fixture(autouse=True)
def inference_mode():
    """These are decode (inference) tests: with autograd enabled the trainable decoders keep their messages for
    backward and run the saving streaming path, which tests/test_gpu_training.py covers.  Here grad is off so that
    LDPC_ENGINE_MODE really selects the engine under test; check_neural re-enables it for one comparison."""
    with torch.no_grad():
        yield


@pytest.fixture(autouse=True, params=["auto", "stream", "sweeps", "gather"])
def engine_mode(request, monkeypatch):
    """'auto' = LDS-resident fused kernel where the code qualifies, 'stream' = HBM-streaming engine (RCQ: fused
    one-kernel-per-iteration form), 'sweeps' = streaming with one kernel per sweep"""
    monkeypatch.setenv("LDPC_ENGINE_MODE", request.param)
    return request.param


def awgn_gpu(B, n, snr_db, seed, dev):
    s2 = 10.0 ** (-snr_db / 10.0)
    gen = torch.Generator(device=dev)
    gen.manual_seed(seed)
    z = torch.randn((B, n), generator=gen, device=dev, dtype=torch.float32)
    return 2.0 * (1.0 + (s2 ** 0.5) * z) / s2


def dense_H(graph, dev):
    return torch.from_numpy(graph.to_dense(np.float32)).to(dev)


def ira_encode(graph, u, dev):
    """systematic IRA encoding on the committed staircase codes: p_i = p_{i-1} + (H_info u)_i
    (torch on the GPU as plain test plumbing; counts stay far below 2^24 so fp32 is exact)"""
    k = graph.n - graph.m
    H = dense_H(graph, dev)
    u = torch.from_numpy(u).to(dev).float()
    s = torch.remainder(u @ H[:, :k].T, 2)
    p = torch.remainder(torch.cumsum(s, dim=1), 2)
    cw = torch.cat([u, p], dim=1)
    assert not bool(torch.remainder(cw @ H.T, 2).any())
    return cw.to(torch.int32)


def syndrome_ok(graph, bits):
    """bits: int tensor [B, n] on the GPU (or numpy) -> numpy bool[B], True where H @ bits == 0 mod 2"""
    if not isinstance(bits, torch.Tensor):
        return ~(graph.syndrome(bits).any(axis=-1))
    H = dense_H(graph, bits.device)
    return (~torch.remainder(bits.float() @ H.T, 2).bool().any(dim=1)).detach().cpu().numpy()


def test_basic_65536_rows_match_oracle_and_properties(gpu_device, oracle_mod):
    import codes
    from ldpc_decoder import BasicMinSumDecoder
    B = 65536
    code = codes.load_code("ira_1998_1512", 10)
    g = code.tanner_graph()
    dec = BasicMinSumDecoder(code, 0.7)
    eng = dec._engine(torch.float32, gpu_device)
    llr = torch.cat([awgn_gpu(B // 2, g.n, 2.0, 1, gpu_device), awgn_gpu(B // 2, g.n, 4.5, 2, gpu_device)])
    llr = llr[torch.randperm(B, device=gpu_device, generator=torch.Generator(device=gpu_device).manual_seed(3))]
    og = oracle_mod.OracleGraph(n=g.n, check_ptr=g.check_ptr, var_idx=g.var_idx)
    rng = np.random.default_rng(0)
    rows = np.unique(np.r_[0, 255, 256, B - 1, rng.integers(0, B, 120)])
    for early in (True, False):
        res = eng.decode(llr, early_stop=early, want_packed=True)
        bits = res.bits.detach().cpu().numpy()
        iters = res.iterations.detach().cpu().numpy()
        succ = res.success.detach().cpu().numpy()
        assert iters.min() >= 1 and iters.max() <= 10
        if early:
            assert len(np.unique(iters)) >= 4                    # codewords of one wave stop at different times
            np.testing.assert_array_equal(succ, syndrome_ok(g, res.bits))
            assert np.all(iters[~succ] == 10)
        else:
            assert np.all(iters == 10)
            np.testing.assert_array_equal(succ, syndrome_ok(g, res.bits))
        ob, op, oi, os_ = oracle_mod.basic_minsum(og, llr[rows].detach().cpu().numpy(), 0.7, 10, early_stop=early, dtype=np.float32)
        np.testing.assert_array_equal(bits[rows], ob)
        np.testing.assert_array_equal(iters[rows], oi)
        np.testing.assert_array_equal(succ[rows], os_)
        np.testing.assert_allclose(res.posterior[rows].detach().cpu().numpy(), op, rtol=1e-5, atol=1e-5)
        packed = res.packed_bits.detach().cpu().numpy()
        unpacked = ((packed[:, :, None] >> np.arange(8)) & 1).reshape(B, -1)[:, :g.n]
        np.testing.assert_array_equal(unpacked, bits)


@pytest.mark.parametrize("family", ["neural2d", "rcq"])
def test_sign_symmetry_at_full_batch(family, gpu_device):
    import codes
    from neural_2d_decoder import Neural2DMinSumDecoder
    from rcq_decoder import RCQMinSumDecoder
    B = 65536
    code = codes.load_code("ira_1998_1512", 10)
    g = code.tanner_graph()
    rng = np.random.default_rng(5)
    u = rng.integers(0, 2, (B, g.n - g.m))
    cw = ira_encode(g, u, gpu_device)
    sgn = (1 - 2 * cw).to(torch.float32)
    llr = torch.cat([awgn_gpu(B // 2, g.n, 5.5, 11, gpu_device), awgn_gpu(B // 2, g.n, 3.0, 12, gpu_device)])
    if family == "neural2d":
        dec = Neural2DMinSumDecoder(code, 2, 10)
        with torch.no_grad():                                  # deterministic weights (no str hash: it is salted per process)
            for i, k in enumerate(sorted(dec.beta_weights.keys())):
                dec.beta_weights[k].fill_(0.6 + 0.03 * (i % 10))
            for i, k in enumerate(sorted(dec.alpha_weights.keys())):
                dec.alpha_weights[k].fill_(0.9 + 0.02 * (i % 10))
        eng = dec._get_engine(gpu_device)
    else:
        eng = RCQMinSumDecoder(code, 3, 8, QP, 10)._get_engine(gpu_device)
    a = eng.decode(llr, early_stop=True)
    b = eng.decode(llr * sgn, early_stop=True)
    # Every arithmetic step is odd-symmetric, so posteriors mirror exactly.  The one asymmetric step of the
    # reference is the hard decision `posterior < 0` at an EXACT zero (both +0 and -0 give bit 0): with ~1e9
    # fp32 sums per decode a handful of exact cancellations occur, and such a codeword may also stop at a
    # different iteration.  Those rows are excluded -- and must be a vanishing fraction.
    same_stop = a.iterations == b.iterations
    clean = same_stop & (a.posterior != 0).all(dim=1) & (b.posterior != 0).all(dim=1)
    assert float(clean.float().mean()) > 0.999
    assert torch.equal(a.success[clean], b.success[clean])
    assert torch.equal(b.bits[clean], (a.bits ^ cw.to(torch.int32))[clean])
    assert torch.equal(b.posterior[clean], (a.posterior * sgn)[clean])
    assert float(a.success.float().mean()) > 0.2               # the early-stop latch was exercised


def _oracle_rows(B, extra=120):
    rng = np.random.default_rng(0)
    return np.unique(np.r_[0, 1, 255, 256, 257, B // 2 - 1, B // 2, B - 2, B - 1, rng.integers(0, B, extra)])


def test_neural2d_65536_rows_match_oracle(gpu_device, oracle_mod):
    """BASELINE config 3 at its FULL batch: rows sampled from the 65536-codeword decode equal the oracle's single-codeword
    results (bits, iterations, success exact; posterior within 1e-5), both stop modes, every engine form"""
    import codes
    from neural_2d_decoder import Neural2DMinSumDecoder
    B = 65536
    code = codes.load_code("ira_1998_1512", 10)
    g = code.tanner_graph()
    dec = Neural2DMinSumDecoder(code, 2, 10)
    rng = np.random.default_rng(4321)                           # bench.py's synthetic "pretrained" tables
    with torch.no_grad():
        for k in sorted(dec.beta_weights.keys()):
            dec.beta_weights[k].fill_(float(np.float32(rng.uniform(0.5, 1.0))))
        for k in sorted(dec.alpha_weights.keys()):
            dec.alpha_weights[k].fill_(float(np.float32(rng.uniform(0.8, 1.2))))
    beta = {k: float(v.item()) for k, v in dec.beta_weights.items()}
    alpha = {k: float(v.item()) for k, v in dec.alpha_weights.items()}
    eng = dec._get_engine(gpu_device)
    llr = torch.cat([awgn_gpu(B // 2, g.n, 2.0, 21, gpu_device), awgn_gpu(B // 2, g.n, 4.5, 22, gpu_device)])
    llr = llr[torch.randperm(B, device=gpu_device, generator=torch.Generator(device=gpu_device).manual_seed(23))]
    og = oracle_mod.OracleGraph(n=g.n, check_ptr=g.check_ptr, var_idx=g.var_idx)
    rows = _oracle_rows(B)
    x = llr[rows].detach().cpu().numpy()
    for early in (True, False):
        res = eng.decode(llr, early_stop=early)
        ob, op, oi, os_ = oracle_mod.neural2d(og, x, 2, 10, beta, alpha, early_stop=early)
        np.testing.assert_array_equal(res.bits[rows].cpu().numpy(), ob)
        np.testing.assert_array_equal(res.iterations[rows].cpu().numpy(), oi)
        np.testing.assert_array_equal(res.success[rows].cpu().numpy(), os_)
        np.testing.assert_allclose(res.posterior[rows].cpu().numpy(), op, rtol=1e-5, atol=1e-5)
        if early:
            assert len(np.unique(oi)) >= 3                       # the sample spans stopped and unstopped codewords
        np.testing.assert_array_equal(res.success.cpu().numpy(), syndrome_ok(g, res.bits))


def test_rcq_65536_rows_match_oracle_with_edge_codes(gpu_device, oracle_mod, engine_mode):
    """BASELINE config 4 at its FULL batch: sampled rows equal the oracle bit for bit -- bits, iterations, success, and
    the 3-bit quantiser code of EVERY edge in the last executed iteration (CSR order) -- both stop modes, every engine form"""
    import codes
    from rcq_decoder import RCQMinSumDecoder, _quantizer_schedule, _threshold_table
    B = 65536
    code = codes.load_code("ira_1998_1512", 10)
    g = code.tanner_graph()
    dec = RCQMinSumDecoder(code, 3, 8, QP, 10)
    eng = dec._get_engine(gpu_device)
    llr = torch.cat([awgn_gpu(B // 2, g.n, 2.0, 31, gpu_device), awgn_gpu(B // 2, g.n, 5.0, 32, gpu_device)])
    llr = llr[torch.randperm(B, device=gpu_device, generator=torch.Generator(device=gpu_device).manual_seed(33))]
    og = oracle_mod.OracleGraph(n=g.n, check_ptr=g.check_ptr, var_idx=g.var_idx)
    rows = _oracle_rows(B)
    x = llr[rows].detach().cpu().numpy()
    thr = _threshold_table(dec.quantizers)
    q_of_iter = _quantizer_schedule(len(dec.quantizers), 10)
    L = thr.shape[1]
    for early in (True, False):
        res = eng.decode(llr, early_stop=early)
        ob, _, oi, os_, ocodes = oracle_mod.rcq(og, x, 3, QP, 10, early_stop=early, trace_codes=True)
        np.testing.assert_array_equal(res.bits[rows].cpu().numpy(), ob)
        np.testing.assert_array_equal(res.iterations[rows].cpu().numpy(), oi)
        np.testing.assert_array_equal(res.success[rows].cpu().numpy(), os_)
        want = np.stack([ocodes[r, int(oi[r]) - 1] for r in range(len(rows))])     # codes of the last executed iteration
        if eng.info()["engine"] == "stream":
            got = eng.debug_c2v(B)[torch.from_numpy(rows).to(gpu_device)].cpu().numpy()
            np.testing.assert_array_equal(got, want)
        else:
            # resident engine: reconstructed values out of LDS -> codes (sign bit of a zero tells code L from code 0; the
            # reference gives code 0 to w = -0.0, the value read-out sees -0.0 as code L: both reconstruct to zero)
            vals, _, it2 = eng.debug_resident_c2v(llr[torch.from_numpy(rows).to(gpu_device)], early_stop=early)
            vals, it2 = vals.cpu().numpy(), it2.cpu().numpy()
            np.testing.assert_array_equal(it2, oi)
            for r in range(len(rows)):
                tau = thr[q_of_iter[int(oi[r]) - 1]]
                level = np.searchsorted(tau, np.abs(vals[r]))
                assert np.all(level < L) and np.array_equal(tau[level], np.abs(vals[r])), "not a reconstruction level"
                got = np.where(np.signbit(vals[r]), L, 0) + level
                bad = got != want[r]
                assert not np.any(bad & ~((want[r] == 0) & (got == L)))
        assert res.success.cpu().numpy().sum() > 0 or not early


def test_dvbs2_wrcq_32768_properties(gpu_device, oracle_mod):
    """config 5's per-GPU shard: (16200,7200), W-RCQ type 2, T=20, 32768 codewords"""
    import codes
    from rcq_decoder import WeightedRCQDecoder
    B = 32768
    code = codes.load_code("dvbs2_like_16200_7200", 20)
    g = code.tanner_graph()
    dec = WeightedRCQDecoder(code, 3, 8, QP, weight_sharing_type=2, max_iterations=20)
    rng = np.random.default_rng(4321)
    with torch.no_grad():
        for k in sorted(dec.beta_weights.keys()):
            dec.beta_weights[k].fill_(float(np.float32(rng.uniform(0.5, 1.0))))
        for k in sorted(dec.alpha_weights.keys()):
            dec.alpha_weights[k].fill_(float(np.float32(rng.uniform(0.8, 1.2))))
    beta = {k: float(v.item()) for k, v in dec.beta_weights.items()}
    alpha = {k: float(v.item()) for k, v in dec.alpha_weights.items()}
    eng = dec._get_engine(gpu_device)
    llr = torch.cat([awgn_gpu(B // 2, g.n, 2.0, 7, gpu_device), awgn_gpu(B // 2, g.n, 5.0, 8, gpu_device)])
    res = eng.decode(llr, early_stop=True)
    iters = res.iterations.detach().cpu().numpy()
    succ = res.success.detach().cpu().numpy()
    assert iters.min() >= 1 and iters.max() <= 20
    rows = np.r_[0, B // 2 - 1, B // 2, B - 1, np.random.default_rng(1).integers(0, B, 12)]
    bits = res.bits[rows].detach().cpu().numpy()
    np.testing.assert_array_equal(succ[rows], syndrome_ok(g, bits))
    og = oracle_mod.OracleGraph(n=g.n, check_ptr=g.check_ptr, var_idx=g.var_idx)
    ob, op, oi, _ = oracle_mod.weighted_rcq(og, llr[rows].detach().cpu().numpy(), 3, QP, 2, 20, beta, alpha)
    np.testing.assert_array_equal(bits, ob)
    np.testing.assert_array_equal(iters[rows], oi)
    np.testing.assert_array_equal(res.posterior[rows].detach().cpu().numpy(), op)


def run_bench(extra, env_extra=None, timeout=900):
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ)
    for k in ("LDPC_ENGINE_MODE", "WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    env.update(env_extra or {})
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py")] + extra, capture_output=True, text=True,
                         timeout=timeout, env=env, cwd=root)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.strip()]
    assert len(lines) == 1 and lines[0].startswith("{"), out.stdout[-2000:]      # exactly ONE line on stdout
    return json.loads(lines[0])


@pytest.mark.parametrize("engine_mode", ["auto"], indirect=True)
def test_bench_contract_line(gpu_device):
    """bench.py prints ONE JSON line with the driver's keys plus the roofline and cpu_baseline objects"""
    d = run_bench(["--steps", "2", "--warmup", "1", "--batch", "4096", "--sweep-reps", "2"])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline", "stream_engine"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1 and d["unit"] == "codewords/s"
    assert d["dtype"] == "f32" and d["data"] == "synthetic" and d["vs_baseline"] is None and "workload" in d["config"]
    r = d["roofline"]                                   # dominant kernel of the step: the LDS-resident fused decode
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "hbm", "hbm_formulation_equiv"):
        assert k in r, k
    assert r["bound"] == "lds" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9 and 0 < r["frac"] <= 1.0
    assert 0 < r["hbm"]["frac"] <= 1.0 and r["hbm"]["peak"] == 8000.0            # a real HBM rate can never exceed the peak
    rs = d["stream_engine"]["roofline"]                 # the north_star's kernel: CN->VN sweep of the streaming engine
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in rs, k
    assert rs["bound"] == "hbm" and rs["peak"] == 8000.0 and abs(rs["frac"] - rs["achieved"] / rs["peak"]) < 1e-9
    c = d["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in c, k
    assert c["kind"] == "port" and c["value"] > 0 and d["value"] > c["value"]


@pytest.mark.parametrize("engine_mode", ["auto"], indirect=True)
def test_bench_default_run_carries_every_baseline_config(gpu_device):
    """the default invocation (what the driver runs) reports configs 3, 4, 5 and the float64 Basic decoder as legs"""
    d = run_bench(["--steps", "3", "--warmup", "1", "--leg-steps", "2", "--sweep-reps", "4", "--no-cpu-baseline"])
    assert d["config"]["batch_per_gpu"] == 65536 and "workloads" in d
    for w in ("neural2d", "rcq", "wrcq_dvbs2", "basic_f64", "rcq_layered"):
        leg = d["workloads"][w]
        assert leg["value"] > 0 and leg["ms_per_step"] > 0 and "roofline" in leg and "workload" in leg
    assert d["workloads"]["wrcq_dvbs2"]["batch"] == 32768 and d["workloads"]["wrcq_dvbs2"]["engine"]["engine"] == "stream"
    assert d["workloads"]["basic_f64"]["dtype"] == "f64"
    lay = d["workloads"]["rcq_layered"]                 # the LDS-resident layered kernel, 4 codewords per one-wave workgroup
    assert lay["engine"]["engine"] == "resident" and lay["engine"]["threads_per_workgroup"] == 64
    assert lay["roofline"]["bound"] == "latency" and lay["roofline"]["dependent_steps"] == 4860 and lay["value"] > 4e6


@pytest.mark.parametrize("engine_mode", ["auto"], indirect=True)
def test_bench_bare_multi_rank_launch(gpu_device):
    """`python bench.py --gpus N` started bare spawns its own ranks (children created before any GPU call).  A 2-rank
    gloo rehearsal on this one GPU and a 1-rank RCCL run each give exactly one JSON line that carries everything the
    north_star asks of a multi-GPU run: the rank count the process group reports, every rank's device, the all-gather
    timed alone, the content check of the gathered array, config 5 sharded weak and strong, and the CPU baseline."""
    d = run_bench(["--gpus", "2", "--batch", "8192", "--steps", "2", "--warmup", "1", "--no-stream-leg", "--no-legs"],
                  {"LDPC_BENCH_BACKEND": "gloo"})
    assert d["n_gpus"] == 2 and d["config"]["global_batch"] == 16384 and d["scaling"] == "weak"
    assert "all_gather" in d["config"]["collective"]
    dd = d["distributed"]
    assert dd["world_size"] == 2 and dd["backend"] == "gloo" and [r["rank"] for r in dd["ranks"]] == [0, 1]
    assert all("device_index" in r and "name" in r for r in dd["ranks"])
    assert dd["gather_check"]["ok"] and dd["gather_check"]["own_shard_equal_on_ranks"] == 2
    assert dd["gather_check"]["sum_of_shard_checksums"] == dd["gather_check"]["checksum_of_gathered"] > 0
    ga = dd["allgather"]
    assert ga["ms"] > 0 and ga["bytes_per_rank"] == 8192 * ((1998 + 7) // 8) and ga["bytes_gathered"] == 2 * ga["bytes_per_rank"]
    assert d["cpu_baseline"]["value"] > 0 and d["cpu_baseline"]["kind"] == "port"     # rank 0's host cores, also at N > 1
    assert d["roofline"]["bound"] == "lds"
    # the DEFAULT multi-rank line (no --batch / --workload): config 2 weak + config 5 sharded weak and strong in the same line
    d = run_bench(["--gpus", "2", "--steps", "1", "--warmup", "1", "--leg-steps", "1", "--no-stream-leg", "--no-cpu-baseline",
                   "--config5-total", "4096", "--sweep-reps", "2"], {"LDPC_BENCH_BACKEND": "gloo"}, timeout=1500)
    assert d["n_gpus"] == 2 and d["config"]["batch_per_gpu"] == 65536
    sw = d["sharded_workloads"]
    weak, strong = sw["wrcq_dvbs2_weak"], sw["wrcq_dvbs2_strong"]
    assert weak["scaling"] == "weak" and weak["batch_per_gpu"] == 32768 and weak["global_batch"] == 65536
    assert strong["scaling"] == "strong" and strong["batch_per_gpu"] == 2048 and strong["global_batch"] == 4096
    for leg in (weak, strong):
        assert leg["value"] > 0 and leg["gather_check"]["ok"] and leg["allgather"]["ms"] > 0
        assert leg["engine"]["engine"] == "stream"
        assert leg["allgather"]["bytes_per_rank"] == leg["batch_per_gpu"] * ((16200 + 7) // 8)
    d = run_bench(["--gpus", "2", "--workload", "wrcq_dvbs2", "--strong", "--batch", "2048", "--steps", "1", "--warmup", "1"],
                  {"LDPC_BENCH_BACKEND": "gloo"})
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["config"]["batch_per_gpu"] == 1024
    assert d["distributed"]["gather_check"]["ok"]
    # one rank over RCCL (backend "nccl"): the collective really is RCCL's, overlapped and un-overlapped
    d = run_bench(["--gpus", "1", "--force-dist", "--batch", "8192", "--steps", "2", "--warmup", "1", "--no-stream-leg",
                   "--no-cpu-baseline", "--no-legs"])
    assert d["n_gpus"] == 1 and "nccl" in d["config"]["collective"] and d["distributed"]["backend"] == "nccl"
    assert d["distributed"]["world_size"] == 1 and d["distributed"]["gather_check"]["ok"]
    assert d["distributed"]["allgather"]["ms"] > 0 and d["config"]["overlap"] is True
    d = run_bench(["--gpus", "1", "--force-dist", "--no-overlap", "--steps", "1", "--warmup", "1", "--leg-steps", "1",
                   "--no-stream-leg", "--no-cpu-baseline", "--config5-total", "2048", "--sweep-reps", "2"])
    assert d["config"]["overlap"] is False and d["sharded_workloads"]["wrcq_dvbs2_strong"]["batch_per_gpu"] == 2048
    assert d["sharded_workloads"]["wrcq_dvbs2_weak"]["gather_check"]["ok"]
