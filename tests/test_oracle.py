"""
CPU tests of the oracle (oracle/ldpc_oracle.c): the C restatement of the reference's
decode loops must reproduce every golden vector captured from the REAL reference
(tests/golden/*.npz, written by oracle/make_golden.py in the build container).
This is what pins the oracle; the GPU tests then compare the HIP engine with it.
"""
import os

import numpy as np
import pytest

from conftest import GOLDEN, golden_sub, load_golden, weights_dict

PKG_DATA = os.path.join(os.path.dirname(GOLDEN), "..", "implementation-of-neural-ldpc-decoders-with-degree-specific-weight-sharing-and-rcq-quantization_amd", "data")


def graph_of(oracle, gold):
    if "H" in gold:
        return oracle.OracleGraph(gold["H"].astype(np.int64))
    z = np.load(os.path.join(PKG_DATA, str(gold["graph"]) + ".npz"))
    return oracle.OracleGraph(n=int(z["n"]), check_ptr=z["check_ptr"], var_idx=z["var_idx"].astype(np.int64))


def bitwise_equal(a, b):
    a, b = np.ascontiguousarray(a), np.ascontiguousarray(b)
    return a.shape == b.shape and a.tobytes() == b.tobytes()


# ------------------------------------------------------------------ third-party arithmetic
def test_sum_association_orders(oracle_mod):
    """torch.sum fp32 / np.sum fp64 of 1-D contiguous arrays, N = 0..575"""
    g = load_golden("sums")
    for x32, y32, x64, y64, n in zip(g["x32"], g["y32"], g["x64"], g["y64"], g["n"]):
        if n >= 0:
            assert bitwise_equal(oracle_mod.sum_f32(x32[:n]), y32), f"torch order N={n}"
            assert bitwise_equal(oracle_mod.sum_f64(x64[:n]), y64), f"numpy order N={n}"
        else:
            m = -n - 1
            assert bitwise_equal(oracle_mod.sum_f32(x32[:m]), y32), f"torch order (alphabet) N={m}"


def test_quantizer_known_answers(oracle_mod):
    g = load_golden("quantizer")
    # the repo's only deterministic vector (rcq_decoder.py:607-611, comprehensive_test.py:259-263)
    thr = oracle_mod.quantizer_thresholds(3, 5.0, 1.5)
    np.testing.assert_array_equal(np.asarray(thr), g["kat_thresholds"])
    assert thr == [0.0, 0.9622504486493761, 2.721655269759087, 5.0]
    np.testing.assert_array_equal(oracle_mod.quantize(g["kat_x"], thr), g["kat_codes"])
    np.testing.assert_array_equal(g["kat_codes"], [6, 5, 0, 2, 2])
    assert bitwise_equal(oracle_mod.dequantize(g["kat_codes"], thr), g["kat_deq"])
    for ci, (bc, C_, gm) in enumerate(g["sweep_cfg"]):
        thr = oracle_mod.quantizer_thresholds(int(bc), float(C_), float(gm))
        np.testing.assert_array_equal(np.asarray(thr), g[f"sweep{ci}_thresholds"])
        np.testing.assert_array_equal(oracle_mod.quantize(g[f"sweep{ci}_x"], thr), g[f"sweep{ci}_codes"])
        assert bitwise_equal(oracle_mod.dequantize(g[f"sweep{ci}_codes"], thr), g[f"sweep{ci}_deq"])


def test_quantizer_schedule(oracle_mod):
    # T=10 -> iterations 0-2 / 3-5 / 6-9, T=20 -> 0-5 / 6-12 / 13-19
    np.testing.assert_array_equal(oracle_mod.quantizer_schedule(10, 3), [0, 0, 0, 1, 1, 1, 2, 2, 2, 2])
    np.testing.assert_array_equal(oracle_mod.quantizer_schedule(20, 3), [0] * 6 + [1] * 7 + [2] * 7)
    np.testing.assert_array_equal(oracle_mod.quantizer_schedule(7, 1), [0] * 7)
    np.testing.assert_array_equal(oracle_mod.quantizer_schedule(9, 2), [0, 0, 0, 1, 1, 1, 1, 1, 1])


# ------------------------------------------------------------------ decoders vs golden
@pytest.mark.parametrize("name", ["toy_basic", "small_basic", "ira_basic"])
def test_basic_golden(name, oracle_mod):
    g = load_golden(name)
    og = graph_of(oracle_mod, g)
    bits, post, iters, succ = oracle_mod.basic_minsum(og, g["llr"], float(g["factor"]), int(g["T"]))
    np.testing.assert_array_equal(bits, g["bits"])
    np.testing.assert_array_equal(iters, g["iters"])
    np.testing.assert_array_equal(succ, g["success"])


def _check_neural(oracle_mod, og, sub):
    beta = weights_dict(sub["beta_keys"], sub["beta_vals"])
    alpha = weights_dict(sub["alpha_keys"], sub["alpha_vals"])
    bits, post, iters, _ = oracle_mod.neural2d(og, sub["llr"], int(sub["wtype"]), int(sub["T"]), beta, alpha)
    np.testing.assert_array_equal(bits, sub["bits"])
    np.testing.assert_array_equal(iters, sub["iters"])
    assert bitwise_equal(post, sub["posterior"])          # bit-exact, sign of zero included


def test_neural2d_golden(oracle_mod):
    g = load_golden("toy_neural2d")
    og = graph_of(oracle_mod, g)
    for w in (1, 2, 3, 4):
        _check_neural(oracle_mod, og, golden_sub(g, f"t{w}"))
        _check_neural(oracle_mod, og, golden_sub(g, f"t{w}d"))      # randn*0.1 init: negative betas
    g = load_golden("small_neural2d")
    og = graph_of(oracle_mod, g)
    for w in (1, 2, 3, 4):
        _check_neural(oracle_mod, og, golden_sub(g, f"t{w}"))
    g = load_golden("ira_neural2d")
    _check_neural(oracle_mod, graph_of(oracle_mod, g), g)


def _check_rcq(oracle_mod, og, sub):
    qp = [tuple(x) for x in sub["qp"]]
    bits, post, iters, succ, codes = oracle_mod.rcq(og, sub["llr"], int(sub["bc"]), qp, int(sub["T"]), trace_codes=True)
    np.testing.assert_array_equal(bits, sub["bits"])
    np.testing.assert_array_equal(iters, sub["iters"])
    np.testing.assert_array_equal(succ, sub["success"])
    for r, it in enumerate(sub["iters"]):                            # every iteration's 3-bit codes, CSR order
        np.testing.assert_array_equal(codes[r, :it], sub["codes"][r, :it])


def _check_wrcq(oracle_mod, og, sub):
    qp = [tuple(x) for x in sub["qp"]]
    beta = weights_dict(sub["beta_keys"], sub["beta_vals"])
    alpha = weights_dict(sub["alpha_keys"], sub["alpha_vals"])
    bits, post, iters, _, codes = oracle_mod.weighted_rcq(og, sub["llr"], int(sub["bc"]), qp, int(sub["wtype"]),
                                                          int(sub["T"]), beta, alpha, trace_codes=True)
    np.testing.assert_array_equal(bits, sub["bits"])
    np.testing.assert_array_equal(iters, sub["iters"])
    assert bitwise_equal(post, sub["posterior"])
    for r, it in enumerate(sub["iters"]):
        np.testing.assert_array_equal(codes[r, :it], sub["codes"][r, :it])


def test_rcq_golden(oracle_mod):
    g = load_golden("toy_rcq")
    og = graph_of(oracle_mod, g)
    _check_rcq(oracle_mod, og, golden_sub(g, "rcq"))
    _check_rcq(oracle_mod, og, golden_sub(g, "rcq4"))
    for w in (1, 2, 3, 4):
        _check_wrcq(oracle_mod, og, golden_sub(g, f"w{w}"))
    _check_wrcq(oracle_mod, og, golden_sub(g, "w2d"))
    g = load_golden("small_rcq")
    og = graph_of(oracle_mod, g)
    _check_rcq(oracle_mod, og, golden_sub(g, "rcq"))
    _check_wrcq(oracle_mod, og, golden_sub(g, "w2"))
    _check_wrcq(oracle_mod, og, golden_sub(g, "w1"))
    g = load_golden("ira_rcq")
    _check_rcq(oracle_mod, graph_of(oracle_mod, g), g)
    g = load_golden("ira_wrcq")
    _check_wrcq(oracle_mod, graph_of(oracle_mod, g), g)


def test_dvbs2_wrcq_golden(oracle_mod):
    if not os.path.exists(os.path.join(GOLDEN, "dvbs2_wrcq.npz")):
        pytest.skip("dvbs2 golden not generated")
    g = load_golden("dvbs2_wrcq")
    _check_wrcq(oracle_mod, graph_of(oracle_mod, g), g)


@pytest.mark.parametrize("name", ["toy_offset_edge", "small_offset_edge"])
def test_offset_and_edge_weight_golden(name, oracle_mod):
    """Neural2DOffsetMinSumDecoder types 1-4, NeuralMinSumDecoder, NeuralOffsetMinSumDecoder"""
    g = load_golden(name)
    og = graph_of(oracle_mod, g)
    for w in (1, 2, 3, 4):
        sub = golden_sub(g, f"o{w}")
        beta = weights_dict(sub["beta_keys"], sub["beta_vals"])
        alpha = weights_dict(sub["alpha_keys"], sub["alpha_vals"])
        bits, post, iters, _ = oracle_mod.neural2d_offset(og, sub["llr"], w, int(sub["T"]), beta, alpha)
        np.testing.assert_array_equal(bits, sub["bits"])
        np.testing.assert_array_equal(iters, sub["iters"])
        np.testing.assert_array_equal(post, sub["posterior"])
    for tag, offset in (("nms", False), ("oms", True)):
        sub = golden_sub(g, tag)
        beta = weights_dict(sub["beta_keys"], sub["beta_vals"])
        bits, post, iters, _ = oracle_mod.neural_minsum(og, sub["llr"], int(sub["T"]), beta, offset=offset)
        np.testing.assert_array_equal(bits, sub["bits"])
        np.testing.assert_array_equal(iters, sub["iters"])
        np.testing.assert_array_equal(post, sub["posterior"])


def test_layered_rcq_golden(oracle_mod):
    """RCQMinSumDecoder(layered=True) as the reference executes it (rcq_decoder.py:281-350)"""
    g = load_golden("layered_rcq")
    for tag, og in (("toy", oracle_mod.OracleGraph(g["toy_H"].astype(np.int64))),
                    ("small", graph_of(oracle_mod, {"graph": "small_96_48"}))):
        bits, post, iters, succ = oracle_mod.rcq_layered(og, g[f"{tag}_llr"], 3, [(3.0, 1.3), (5.0, 1.3), (7.0, 1.3)],
                                                         int(g[f"{tag}_T"]))
        np.testing.assert_array_equal(bits, g[f"{tag}_bits"])
        np.testing.assert_array_equal(iters, g[f"{tag}_iters"])
        np.testing.assert_array_equal(succ, g[f"{tag}_success"])


# ------------------------------------------------------------------ oracle self-consistency
def test_fixed_iteration_mode_and_threads(oracle_mod):
    """early_stop=False runs T iterations; success = final syndrome; threads do not change results"""
    g = load_golden("small_basic")
    og = graph_of(oracle_mod, g)
    x = g["llr"]
    b1, p1, i1, s1 = oracle_mod.basic_minsum(og, x, 0.7, 12, early_stop=False, threads=1)
    b2, p2, i2, s2 = oracle_mod.basic_minsum(og, x, 0.7, 12, early_stop=False, threads=4)
    assert np.all(i1 == 12) and np.array_equal(b1, b2) and bitwise_equal(p1, p2) and np.array_equal(s1, s2)
    H = np.zeros((og.m, og.n), dtype=np.int64)
    H[og.rows, og.var_idx] = 1
    np.testing.assert_array_equal(s1, (H @ b1.T % 2).sum(axis=0) == 0)


def test_basic_fp32_posterior_equals_neural2d_type3(oracle_mod):
    """Basic in fp32 is Neural2D type 3 with every beta = 0.7 (SURVEY 8c cross-check)"""
    g = load_golden("small_basic")
    og = graph_of(oracle_mod, g)
    x = g["llr"].astype(np.float32)
    dcs = sorted(set(og.dc.tolist()))
    beta = {f"iter_{t}_dc{d}": float(np.float32(0.7)) for t in range(12) for d in dcs}
    b1, p1, i1, _ = oracle_mod.basic_minsum(og, x, 0.7, 12, dtype=np.float32)
    b2, p2, i2, _ = oracle_mod.neural2d(og, x, 3, 12, beta, {})
    assert np.array_equal(b1, b2) and np.array_equal(i1, i2) and bitwise_equal(p1, p2)
