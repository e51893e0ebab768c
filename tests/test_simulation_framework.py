"""
Batched Monte-Carlo driver: the reference's stop rule frame by frame, result
container / JSON layout (CPU), and on the GPU the counters against a direct decode of the very
same LLR blocks.
"""
import json

import numpy as np
import pytest
import torch


def ref_loop(frame_error, max_frames, max_errors):
    """the reference's per-frame loop (simulation_framework.py:110-132) on a pre-drawn error sequence"""
    total = errs = 0
    while total < max_frames and errs < max_errors:
        errs += int(frame_error[total])
        total += 1
    return total, errs


def test_block_truncation_equals_the_per_frame_stop_rule():
    from simulation_framework import frames_to_count
    rng = np.random.default_rng(0)
    for trial in range(300):
        p = rng.choice([0.0, 0.01, 0.2, 0.9, 1.0])
        seq = rng.random(5000) < p
        max_frames = int(rng.integers(1, 5000))
        max_errors = int(rng.integers(1, 60))
        block = int(rng.integers(1, 700))
        total = errs = 0
        while total < max_frames and errs < max_errors:
            frames = min(block, max_frames - total)
            blk = seq[total:total + frames]
            take = frames_to_count(blk, total, errs, max_frames, max_errors)
            errs += int(blk[:take].sum())
            total += take
        assert (total, errs) == ref_loop(seq, max_frames, max_errors)


def test_result_container_and_json_layout(tmp_path):
    from simulation_framework import LDPSimulator, SimulationConfig, SimulationResult
    cfg = SimulationConfig(results_dir=str(tmp_path), save_results=True)
    assert (cfg.snr_range, cfg.snr_step, cfg.max_frames, cfg.max_errors, cfg.min_frames, cfg.parallel_workers) == \
        ((0.0, 6.0), 0.5, 10000, 100, 1000, 4)                    # reference defaults (simulation_framework.py:27-38)
    sim = LDPSimulator(cfg)
    r = SimulationResult("Basic MinSum", [0.0, 0.5, 1.0])
    r.add_result(2, 0.25, 0.01, 7.5, 1.5, 400, 100)              # out-of-order index pads with zeros
    assert r.frame_error_rates == [0.0, 0.0, 0.25] and r.total_frames == [0, 0, 400]
    r.add_result(0, 1.0, 0.2, 10.0, 0.5, 100, 100)
    sim.save_results({"Basic MinSum": r}, "res.json")
    raw = json.load(open(tmp_path / "res.json"))
    assert set(raw["Basic MinSum"]) == {"decoder_name", "snr_values", "frame_error_rates", "bit_error_rates",
                                         "average_iterations", "simulation_times", "total_frames", "total_errors"}
    back = sim.load_results("res.json")["Basic MinSum"]
    assert back.frame_error_rates == r.frame_error_rates and back.total_errors == r.total_errors
    assert back.snr_values == [0.0, 0.5, 1.0]


def test_error_counting_from_packed_decisions_and_stage_cap():
    """host logic of the block decode: ones per frame from the bit-packed rows (bits past n ignored) and the choice of the
    first stage's iteration cap from a block's stop iterations"""
    from simulation_framework import _next_cap, _ones_per_frame
    rng = np.random.default_rng(3)
    for n in (96, 1998, 13):
        bits = (rng.random((37, n)) < 0.3).astype(np.uint8)
        padded = np.concatenate([bits, np.ones((37, (-n) % 8), np.uint8)], axis=1)          # garbage past n must not count
        packed = np.packbits(padded, axis=1, bitorder="little")
        got = _ones_per_frame(torch.from_numpy(packed), n)
        np.testing.assert_array_equal(got.numpy(), bits.sum(axis=1))

    class Eng:
        iters = 20
        def __init__(self, kind): self.kind = kind
        def info(self): return {"engine": self.kind}
    it = torch.cat([torch.full((3000,), 5), torch.full((900,), 7), torch.full((90,), 12), torch.full((10,), 20)]).to(torch.int32)
    assert _next_cap(Eng("stream"), it) == 7                      # 7 + 0.025 * 22 beats 5 + 0.25 * 22 and 12 + 0.0025 * 22
    assert _next_cap(Eng("resident"), it) is None                 # that engine stops codeword by codeword already
    assert _next_cap(Eng("stream"), torch.full((4000,), 20, dtype=torch.int32)) is None       # nothing stops early
    assert _next_cap(Eng("stream"), it[:500]) is None             # small blocks are not staged
    assert _next_cap(Eng("stream"), it[:500], 64) is not None


def test_create_test_decoders_matches_reference_set():
    from ldpc_decoder import create_test_ldpc_code
    from simulation_framework import create_test_decoders
    d = create_test_decoders(create_test_ldpc_code())
    assert list(d) == ["Basic MinSum", "N-NMS", "N-OMS", "N-2D-NMS Type 1", "N-2D-NMS Type 2", "N-2D-NMS Type 3",
                       "N-2D-NMS Type 4", "N-2D-OMS Type 2", "RCQ MinSum", "W-RCQ Type 2"]


@pytest.mark.gpu
def test_simulator_counters_match_direct_decode(gpu_device, tmp_path):
    import codes
    from ldpc_decoder import BasicMinSumDecoder
    from simulation_framework import LDPSimulator, SimulationConfig
    code = codes.load_code("small_96_48", 10)
    cfg = SimulationConfig(snr_range=(2.0, 4.0), snr_step=2.0, max_frames=3000, max_errors=40, batch_frames=512,
                           seed=5, results_dir=str(tmp_path), save_results=False)
    sim = LDPSimulator(cfg)
    dec = BasicMinSumDecoder(code, 0.7)
    fer, ber, avg_it, t, frames, errs = sim.simulate_single_snr(dec, code, 3.0, cfg.max_frames, cfg.max_errors)
    # replay: same generator seed, same blocks, per-frame reference loop on the decoded outcomes
    gen = torch.Generator(device=gpu_device)
    gen.manual_seed(5 * 1_000_003 + 3000)
    eng = dec._engine(torch.float32, gpu_device)
    ferr_all, berr_all, it_all = [], [], []
    while sum(len(x) for x in ferr_all) < cfg.max_frames:
        llr = sim._draw_llr(gen, min(512, cfg.max_frames - sum(len(x) for x in ferr_all)), code.n, 3.0, gpu_device)
        res = eng.decode(llr, early_stop=True, want_posterior=False)
        ferr_all.append((res.bits != 0).any(dim=1).cpu().numpy())
        berr_all.append((res.bits != 0).sum(dim=1).cpu().numpy())
        it_all.append(res.iterations.cpu().numpy())
        if np.concatenate(ferr_all).sum() >= cfg.max_errors:
            break
    ferr, berr, its = np.concatenate(ferr_all), np.concatenate(berr_all), np.concatenate(it_all)
    total, e = ref_loop(ferr, cfg.max_frames, cfg.max_errors)
    assert (frames, errs) == (total, e)
    assert fer == e / total and ber == berr[:total].sum() / (total * code.n) and avg_it == its[:total].sum() / total
    assert 0.0 < fer < 1.0
    # the literal reference channel (bit 0 -> negative LLR) reproduces the reference's FER = 1.0
    sim_q = LDPSimulator(SimulationConfig(max_frames=200, max_errors=50, batch_frames=64, llr_convention="reference",
                                          save_results=False, results_dir=str(tmp_path)))
    fer_q, *_rest, frames_q, errs_q = sim_q.simulate_single_snr(dec, code, 6.0, 200, 50)
    assert fer_q == 1.0 and (frames_q, errs_q) == (50, 50)
    # sweep over decoders in worker threads, JSON written
    cfg2 = SimulationConfig(snr_range=(3.0, 5.0), snr_step=2.0, max_frames=600, max_errors=20, batch_frames=256,
                            parallel_workers=2, results_dir=str(tmp_path), save_results=True)
    from rcq_decoder import RCQMinSumDecoder
    res = LDPSimulator(cfg2).simulate_multiple_decoders(
        {"Basic": dec, "RCQ": RCQMinSumDecoder(code, 3, 8, [(3.0, 1.3), (5.0, 1.3), (7.0, 1.3)], 10)}, code)
    assert set(res) == {"Basic", "RCQ"} and len(res["Basic"].frame_error_rates) == 2
    assert res["Basic"].frame_error_rates[1] <= res["Basic"].frame_error_rates[0]
    assert (tmp_path / "simulation_results.json").exists()


@pytest.mark.gpu
@pytest.mark.parametrize("staged", [False, True])
@pytest.mark.parametrize("family", ["basic", "rcq", "neural2d"])
def test_simulator_counters_match_oracle_decoded_blocks(family, staged, gpu_device, oracle_mod, tmp_path):
    """SURVEY 8f-1 against the ORACLE: the blocks the simulator draws are decoded frame by frame on the CPU by the
    restatement of the reference's decoders, the reference's own per-frame loop (simulation_framework.py:110-132: errors,
    bit errors of erroneous frames, iterations, stop at max_frames / max_errors) is applied to those outcomes, and
    FER / BER / average iterations / frame and error counts must equal what LDPSimulator.simulate_single_snr returns."""
    import codes
    from ldpc_decoder import BasicMinSumDecoder
    from neural_2d_decoder import Neural2DMinSumDecoder
    from rcq_decoder import RCQMinSumDecoder
    from simulation_framework import LDPSimulator, SimulationConfig
    qp = [(3.0, 1.3), (5.0, 1.3), (7.0, 1.3)]
    code = codes.load_code("small_96_48", 10)
    g = code.tanner_graph()
    og = oracle_mod.OracleGraph(n=g.n, check_ptr=g.check_ptr, var_idx=g.var_idx)
    if family == "basic":
        dec = BasicMinSumDecoder(code, 0.7)
        cpu = lambda x: oracle_mod.basic_minsum(og, x, 0.7, 10, dtype=np.float32)
    elif family == "rcq":
        dec = RCQMinSumDecoder(code, 3, 8, qp, 10)
        cpu = lambda x: oracle_mod.rcq(og, x, 3, qp, 10)
    else:
        dec = Neural2DMinSumDecoder(code, 2, 10)
        rng = np.random.default_rng(7)
        with torch.no_grad():
            for k in sorted(dec.beta_weights.keys()):
                dec.beta_weights[k].fill_(float(np.float32(rng.uniform(0.5, 1.0))))
            for k in sorted(dec.alpha_weights.keys()):
                dec.alpha_weights[k].fill_(float(np.float32(rng.uniform(0.8, 1.2))))
        beta = {k: float(v.item()) for k, v in dec.beta_weights.items()}
        alpha = {k: float(v.item()) for k, v in dec.alpha_weights.items()}
        cpu = lambda x: oracle_mod.neural2d(og, x, 2, 10, beta, alpha)
    if staged:
        # the streaming engine, blocks decoded in two stages (capped first stage + stragglers as a small batch: _decode_block);
        # the cap comes from the previous block, so blocks 2.. of every point run staged wherever staging pays
        from simulation_framework import _engine_of, _next_cap
        eng = _engine_of(dec, gpu_device).set_mode("stream")
        assert eng.info()["engine"] == "stream"
        it = torch.cat([torch.full((900,), 3), torch.full((90,), 6), torch.full((10,), 10)]).to(torch.int32).to(gpu_device)
        assert _next_cap(eng, it, 64) in (3, 4, 5, 6) and _next_cap(eng, torch.full((1000,), 10, dtype=torch.int32, device=gpu_device), 64) is None
    for snr_db, max_frames, max_errors, block in ((3.0, 2500, 30, 384), (1.0, 700, 1000, 200), (5.0, 1500, 5, 512)):
        cfg = SimulationConfig(max_frames=max_frames, max_errors=max_errors, batch_frames=block, seed=9,
                               results_dir=str(tmp_path), save_results=False, staged_early_stop=staged, stage_min_block=64)
        sim = LDPSimulator(cfg)
        with torch.no_grad():
            fer, ber, avg_it, _t, frames, errs = sim.simulate_single_snr(dec, code, snr_db, max_frames, max_errors)
        # the same blocks (same generator seed and draw sizes), decoded by the oracle; then the reference's loop, frame by frame
        gen = torch.Generator(device=gpu_device)
        gen.manual_seed(9 * 1_000_003 + int(round(snr_db * 1000)))
        total = frame_errors = bit_errors = total_iterations = 0
        while total < max_frames and frame_errors < max_errors:
            x = sim._draw_llr(gen, min(block, max_frames - total), code.n, snr_db, gpu_device).cpu().numpy()
            ob, _, oi, _ = cpu(x)[:4]
            for r in range(len(x)):                              # simulation_framework.py:110-132
                if not (total < max_frames and frame_errors < max_errors):
                    break
                if ob[r].any():                                  # all-zero codeword was sent
                    frame_errors += 1
                    bit_errors += int(ob[r].sum())
                total_iterations += int(oi[r])
                total += 1
        assert (frames, errs) == (total, frame_errors)
        assert fer == frame_errors / total and ber == bit_errors / (total * code.n) and avg_it == total_iterations / total
