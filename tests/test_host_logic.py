"""
CPU tests of the host side: Tanner-graph compiler, committed codes, reference-style API
surface (names, constructor arguments, attributes, state_dict keys, RNG-identical init),
weight-table flattening against the oracle's independent flattening, the quantiser
utility against the reference's known answers, the C-ABI library (loads, exports every
symbol include/ldpc_hip.h declares) and the no-CPU-fallback rule.
"""
import ctypes
import os
import sys
import re

import numpy as np
import pytest
import torch

from conftest import PKG, ROOT, golden_sub, load_golden, weights_dict

QP = [(3.0, 1.3), (5.0, 1.3), (7.0, 1.3)]


# ------------------------------------------------------------------ graph compiler / codes
def test_tanner_graph_orders_match_reference_scans():
    from tanner_graph import TannerGraph
    rng = np.random.default_rng(0)
    H = (rng.random((9, 17)) < 0.3).astype(np.int64)
    H[4, :] = 0          # empty check
    H[:, 11] = 0         # isolated variable
    g = TannerGraph.from_dense(H)
    assert (g.n, g.m, g.E) == (17, 9, int(H.sum()))
    for i in range(9):   # np.where(H[i, :] == 1)  (ldpc_decoder.py:92)
        np.testing.assert_array_equal(g.var_idx[g.check_ptr[i]:g.check_ptr[i + 1]], np.where(H[i, :] == 1)[0])
    for j in range(17):  # np.where(H[:, j] == 1)  (ldpc_decoder.py:124)
        e = g.csc_edge[g.var_ptr[j]:g.var_ptr[j + 1]]
        np.testing.assert_array_equal(g.check_of_edge[e], np.where(H[:, j] == 1)[0])
        assert np.all(g.var_idx[e] == j)
    np.testing.assert_array_equal(g.to_dense(np.int64), H)
    bits = rng.integers(0, 2, (5, 17))
    np.testing.assert_array_equal(g.syndrome(bits), (bits @ H.T) % 2)
    # edge-list constructor: any order in, CSR out; duplicates rejected
    r, c = np.nonzero(H)
    p = rng.permutation(len(r))
    g2 = TannerGraph(17, 9, r[p], c[p])
    np.testing.assert_array_equal(g2.var_idx, g.var_idx)
    np.testing.assert_array_equal(g2.csc_edge, g.csc_edge)
    with pytest.raises(ValueError):
        TannerGraph(17, 9, np.r_[r, r[:1]], np.r_[c, c[:1]])
    with pytest.raises(ValueError):
        TannerGraph(17, 9, [0], [17])


def test_oracle_graph_agrees_with_product_graph(oracle_mod):
    """two independent builders (np.lexsort in oracle.py, stable argsort in tanner_graph.py)"""
    import codes
    for name in ("small_96_48", "ira_1998_1512"):
        g = codes.load_graph(name)
        og = oracle_mod.OracleGraph(n=g.n, check_ptr=g.check_ptr, var_idx=g.var_idx)
        np.testing.assert_array_equal(og.csc_edge, g.csc_edge)
        np.testing.assert_array_equal(og.var_ptr, g.var_ptr)


def test_committed_codes_have_the_specified_profiles():
    import codes
    g = codes.load_graph("ira_1998_1512")
    assert (g.n, g.m, g.E) == (1998, 486, 6587)
    assert dict(zip(*np.unique(g.dv, return_counts=True))) == {1: 1, 2: 485, 3: 1296, 8: 216}
    assert set(np.unique(g.dc)) == {13, 14} and g.max_dv <= 8
    g = codes.load_graph("dvbs2_like_16200_7200")        # SURVEY 8d config 5 / paper Table II
    assert (g.n, g.m, g.E) == (16200, 9000, 48599)
    assert dict(zip(*np.unique(g.dv, return_counts=True))) == {1: 1, 2: 8999, 3: 5400, 8: 1800}
    assert dict(zip(*np.unique(g.dc, return_counts=True))) == {4: 1441, 5: 3239, 6: 3600, 7: 720}
    g = codes.load_graph("small_96_48")
    assert set(np.unique(g.dv)) == {1, 2, 3, 8}


def test_code_generator_is_deterministic_and_simple():
    import codes
    spec = dict(n=96, m=48, info_degrees={8: 8, 3: 40}, check_degrees=None, seed=96)
    a, b = codes.generate_ira_code(**spec), codes.generate_ira_code(**spec)
    np.testing.assert_array_equal(a.var_idx, b.var_idx)
    committed = codes.load_graph("small_96_48")
    np.testing.assert_array_equal(a.var_idx, committed.var_idx)
    np.testing.assert_array_equal(a.check_ptr, committed.check_ptr)
    # IRA staircase: parity column p touches checks p and p+1 -> systematic encoding works
    k = 48
    rng = np.random.default_rng(1)
    u = rng.integers(0, 2, k)
    H = a.to_dense(np.int64)
    s = H[:, :k] @ u % 2
    p = np.cumsum(s) % 2
    cw = np.r_[u, p]
    assert not (H @ cw % 2).any()


# ------------------------------------------------------------------ API surface
def test_ldpccode_and_toy_code_match_reference_facts():
    from ldpc_decoder import LDPCCode, create_test_ldpc_code
    code = create_test_ldpc_code()
    assert (code.n, code.k, code.max_iterations) == (7, 4, 10) and code.H.shape == (4, 7)
    assert code.rate == 4 / 7
    assert code.check_node_degrees == {0: 3, 1: 3, 2: 3, 3: 4}
    assert code.variable_node_degrees == {0: 3, 1: 3, 2: 3, 3: 1, 4: 1, 5: 1, 6: 1}
    assert int(np.sum(code.H)) == 13
    assert LDPCCode(7, 4, code.H).max_iterations == 50                              # dataclass default
    g = code.tanner_graph()
    assert g is code.tanner_graph()           # compiled once
    code.H = code.H.copy()
    assert code.tanner_graph() is not g       # replaced matrix -> recompiled


def test_simulate_awgn_channel_is_the_reference_recipe():
    from ldpc_decoder import simulate_awgn_channel
    g = load_golden("toy_basic")
    np.random.seed(0)                         # golden rows 0..63: np.random.seed(s); simulate(zeros(7), 2.0)
    np.testing.assert_array_equal(simulate_awgn_channel(np.zeros(7, dtype=int), 2.0), g["llr"][0])
    np.random.seed(5)
    np.testing.assert_array_equal(simulate_awgn_channel(np.zeros(7, dtype=int), 2.0), g["llr"][5])


@pytest.mark.parametrize("wtype,count", [(1, 40), (2, 40), (3, 20), (4, 20)])
def test_parameter_counts_keys_and_rng_identical_init(wtype, count):
    """40/40/20/20 parameters at T=10 on the toy code (IMPLEMENTATION_SUMMARY.md:165-172); same
    seed -> the very parameters the reference constructor draws (captured in the golden file)"""
    from ldpc_decoder import create_test_ldpc_code
    from neural_2d_decoder import Neural2DMinSumDecoder
    g = load_golden("toy_neural2d")
    sub = golden_sub(g, f"t{wtype}d")
    torch.manual_seed(100 + wtype)
    dec = Neural2DMinSumDecoder(create_test_ldpc_code(), weight_sharing_type=wtype, max_iterations=10)
    assert len(dec.beta_weights) + len(dec.alpha_weights) == count
    assert len(list(dec.parameters())) == count
    assert {k: float(v.item()) for k, v in dec.beta_weights.items()} == weights_dict(sub["beta_keys"], sub["beta_vals"])
    assert {k: float(v.item()) for k, v in dec.alpha_weights.items()} == weights_dict(sub["alpha_keys"], sub["alpha_vals"])
    for k, v in dec.state_dict().items():
        assert re.fullmatch(r"(beta|alpha)_weights\.iter_\d+_(dc\d+(_dv\d+)?|dv\d+)", k) and v.shape == (1,)
    assert sorted(dec.check_node_degrees) == [3, 4] and sorted(dec.variable_node_degrees) == [1, 3]


def test_wrcq_init_and_attributes_match_reference():
    from ldpc_decoder import create_test_ldpc_code
    from rcq_decoder import RCQMinSumDecoder, WeightedRCQDecoder
    g = load_golden("toy_rcq")
    sub = golden_sub(g, "w2d")
    torch.manual_seed(7)
    dec = WeightedRCQDecoder(create_test_ldpc_code(), bc=3, bv=8, quantizer_params=QP, weight_sharing_type=2,
                             max_iterations=10)
    assert {k: float(v.item()) for k, v in dec.beta_weights.items()} == weights_dict(sub["beta_keys"], sub["beta_vals"])
    assert (dec.bc, dec.bv, dec.layered, len(dec.quantizers)) == (3, 8, False, 3)
    assert [dec.quantizers.index(dec._get_quantizer(t)) for t in range(10)] == [0, 0, 0, 1, 1, 1, 2, 2, 2, 2]
    r = RCQMinSumDecoder(create_test_ldpc_code(), 3, 8, QP, max_iterations=20)
    assert [r.quantizers.index(r._get_quantizer(t)) for t in range(20)] == [0] * 6 + [1] * 7 + [2] * 7
    assert RCQMinSumDecoder(create_test_ldpc_code(), 3, 8, QP[:1], 9)._get_quantizer(8).C == 3.0
    with pytest.raises(ValueError):
        WeightedRCQDecoder(create_test_ldpc_code(), 3, 8, QP, weight_sharing_type=0, max_iterations=2)


def test_edge_weight_decoders_init_and_analysis():
    """NeuralMinSumDecoder / NeuralOffsetMinSumDecoder: 130 parameters at T=10 on the toy code
    (IMPLEMENTATION_SUMMARY.md:168), reference key names, seed-identical init"""
    from ldpc_decoder import create_test_ldpc_code
    from neural_minsum_decoder import NeuralMinSumDecoder, NeuralOffsetMinSumDecoder, analyze_weight_patterns
    g = load_golden("toy_offset_edge")
    code = create_test_ldpc_code()
    sub = golden_sub(g, "nms")
    torch.manual_seed(int(sub["seed"]))
    dec = NeuralMinSumDecoder(code, max_iterations=10)
    assert len(dec.beta_weights) == 130 and "iter_0_c0_v0" in dec.beta_weights and "iter_9_c3_v6" in dec.beta_weights
    assert {k: float(v.item()) for k, v in dec.beta_weights.items()} == weights_dict(sub["beta_keys"], sub["beta_vals"])
    W = dec.weight_table()
    assert W.shape == (10, 13) and abs(float(W.mean()) - 0.7) < 0.05
    an = analyze_weight_patterns(dec, code)
    assert set(an) == {"weight_statistics", "iteration_patterns", "node_degree_correlations"}
    assert sorted(an["iteration_patterns"]) == list(range(10))
    assert set(an["node_degree_correlations"]) == {"check_degree_3", "check_degree_4"}
    assert an["node_degree_correlations"]["check_degree_3"]["count"] == 9
    np.testing.assert_allclose(an["iteration_patterns"][0]["mean"], W[0].astype(np.float64).mean())
    sub = golden_sub(g, "oms")
    torch.manual_seed(int(sub["seed"]))
    dec = NeuralOffsetMinSumDecoder(code, max_iterations=10)
    with torch.no_grad():
        for p in dec.beta_weights.values():
            p.mul_(3.0).abs_()                       # the transformation the golden generator applied
    assert {k: float(v.item()) for k, v in dec.beta_weights.items()} == weights_dict(sub["beta_keys"], sub["beta_vals"])


def test_quantizer_utility_known_answers():
    from rcq_decoder import NonUniformQuantizer
    g = load_golden("quantizer")
    q = NonUniformQuantizer(bc=3, C=5.0, gamma=1.5)
    assert q.thresholds == [0.0, 0.9622504486493761, 2.721655269759087, 5.0] and (q.bc, q.C, q.gamma) == (3, 5.0, 1.5)
    x = torch.from_numpy(g["kat_x"])
    codes = q.quantize(x)
    assert codes.dtype == torch.int64 and codes.tolist() == [6, 5, 0, 2, 2]
    deq = q.dequantize(codes)
    assert deq.dtype == torch.float32
    np.testing.assert_array_equal(deq.numpy(), g["kat_deq"])
    for ci, (bc, C_, gm) in enumerate(g["sweep_cfg"]):
        q = NonUniformQuantizer(int(bc), float(C_), float(gm))
        np.testing.assert_array_equal(np.asarray(q.thresholds), g[f"sweep{ci}_thresholds"])
        c = q.quantize(torch.from_numpy(g[f"sweep{ci}_x"]))
        np.testing.assert_array_equal(c.numpy(), g[f"sweep{ci}_codes"])
        d = q.dequantize(torch.from_numpy(g[f"sweep{ci}_codes"]))
        assert d.numpy().tobytes() == g[f"sweep{ci}_deq"].tobytes()      # incl. code 2^(bc-1) -> -0.0
    with pytest.raises(ZeroDivisionError):
        NonUniformQuantizer(1, 1.0, 1.0)            # 2^(bc-1)-1 == 0, as in the reference (rcq_decoder.py:54)


# ------------------------------------------------------------------ table flattening vs oracle
@pytest.mark.parametrize("name", ["small_96_48", "ira_1998_1512"])
@pytest.mark.parametrize("wtype", [1, 2, 3, 4])
def test_weight_tables_equal_oracle_flattening(name, wtype, oracle_mod):
    """product (weight_sharing.py) and oracle (oracle.py) flatten the reference's dict lookups
    independently; per-edge beta and per-variable alpha must agree for every iteration"""
    import codes
    from neural_2d_decoder import Neural2DMinSumDecoder
    code = codes.load_code(name, 5)
    dec = Neural2DMinSumDecoder(code, weight_sharing_type=wtype, max_iterations=5)
    rng = np.random.default_rng(wtype)
    with torch.no_grad():
        for p in list(dec.beta_weights.values()) + list(dec.alpha_weights.values()):
            p.fill_(float(np.float32(rng.uniform(0.2, 1.5))))
    # drop one key: the lookup falls back to the constant (ParameterDict.get(key, 0.7 / 1.0))
    if len(dec.beta_weights):
        del dec.beta_weights[sorted(dec.beta_weights.keys())[0]]
    beta = {k: float(v.item()) for k, v in dec.beta_weights.items()}
    alpha = {k: float(v.item()) for k, v in dec.alpha_weights.items()}
    g = code.tanner_graph()
    og = oracle_mod.OracleGraph(n=g.n, check_ptr=g.check_ptr, var_idx=g.var_idx)
    obt, obs, oat, oas = oracle_mod.weight_tables(og, wtype, 5, beta, alpha)
    bt, at = dec.weight_tables()
    lay = dec._sharing_layout()
    np.testing.assert_array_equal(bt[:, lay.beta_slot], obt[:, obs])
    np.testing.assert_array_equal(at[:, lay.alpha_slot], oat[:, oas])


# ------------------------------------------------------------------ native library, no fallback
def header_functions(header="ldpc_hip.h"):
    text = open(os.path.join(ROOT, "include", header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(ldpc_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    import _native
    lib_path = os.path.join(PKG, "libldpc_hip.so")
    assert os.path.exists(lib_path), "run __graft_entry__.build() first"
    lib = ctypes.CDLL(lib_path)
    names = header_functions()
    assert len(names) >= 11 and set(names) == set(_native.PRODUCT_EXPORTS)
    assert not any("debug" in n for n in names), "measurement/test hooks belong in ldpc_hip_debug.h"
    dbg = header_functions("ldpc_hip_debug.h")
    assert set(dbg) == set(_native.DEBUG_EXPORTS)
    for name in names + dbg:
        assert hasattr(lib, name), f"{name} declared in include/ but not exported"
    assert _native.load().ldpc_abi_version() == 1


def test_product_library_reads_no_environment():
    """every getenv of the native sources sits inside an #ifdef LDPC_RESIDENT_PROBES block (tuning builds of tools/)"""
    for fn in os.listdir(os.path.join(PKG, "csrc")):
        depth = 0
        for line in open(os.path.join(PKG, "csrc", fn)):
            st = line.strip()
            if st.startswith("#ifdef LDPC_RESIDENT_PROBES"):
                depth += 1
            elif st.startswith("#endif") and depth:
                depth -= 1
            elif "getenv" in line and not st.startswith("//"):
                assert depth > 0, f"{fn}: {st}"
    import subprocess
    syms = subprocess.run(["nm", "-D", "--undefined-only", os.path.join(PKG, "libldpc_hip.so")],
                          capture_output=True, text=True).stdout
    assert "getenv" not in syms


@pytest.mark.skipif(torch.cuda.is_available(), reason="checks the no-GPU behaviour")
def test_no_cpu_fallback_decode_fails_loudly():
    import _native
    from ldpc_decoder import BasicMinSumDecoder, create_test_ldpc_code
    from neural_2d_decoder import Neural2DMinSumDecoder
    from rcq_decoder import RCQMinSumDecoder
    code = create_test_ldpc_code()
    with pytest.raises(_native.NativeEngineError):
        BasicMinSumDecoder(code).decode(np.zeros(7))
    with pytest.raises(_native.NativeEngineError):
        Neural2DMinSumDecoder(code, 2, 3)(torch.zeros(7))
    with pytest.raises(_native.NativeEngineError):
        RCQMinSumDecoder(code, 3, 8, QP, 3).decode(torch.zeros(7))
    # the C ABI itself reports the missing device instead of computing anything
    lib = _native.load()
    h = ctypes.c_void_p()
    cp, vi = np.array([0, 1], np.int32), np.array([0], np.int32)
    assert lib.ldpc_graph_create(ctypes.byref(h), 1, 1, 1, _native.ptr(cp), _native.ptr(vi)) == -2
    assert b"HIP" in lib.ldpc_last_error()


def test_product_never_imports_the_oracle():
    """only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may touch oracle/"""
    for fn in os.listdir(PKG):
        if fn.endswith(".py"):
            src = open(os.path.join(PKG, fn)).read()
            assert "import oracle" not in src and "libldpc_oracle" not in src, fn
    for fn in os.listdir(os.path.join(PKG, "csrc")):
        assert "oracle" not in open(os.path.join(PKG, "csrc", fn)).read().lower(), fn


def test_parameter_fingerprint_sees_every_kind_of_weight_change():
    """The degree-shared decoders re-flatten their ParameterDicts only when `_param_stamp()` changes: in-place updates (version
    counters), replaced storage, a replaced Parameter, load_state_dict and a changed iteration count must all show; reading must not."""
    import codes
    from neural_2d_decoder import Neural2DMinSumDecoder
    from rcq_decoder import WeightedRCQDecoder
    code = codes.load_code("small_96_48", 5)
    for dec in (Neural2DMinSumDecoder(code, 2, 5), WeightedRCQDecoder(code, 3, 8, QP, weight_sharing_type=3, max_iterations=5)):
        seen = {dec._param_stamp()}
        def changed():
            st = dec._param_stamp()
            new = st not in seen
            seen.add(st)
            return new
        assert not changed()                                              # reading twice: same stamp
        _ = dec.weight_tables()
        _ = [float(p.item()) for p in dec.beta_weights.values()]
        assert not changed()
        k = sorted(dec.beta_weights.keys())[0]
        with torch.no_grad():
            dec.beta_weights[k].fill_(0.5)
        assert changed()
        with torch.no_grad():
            list(dec.parameters())[-1].mul_(1.5)                          # an alpha where the sharing type has any
        assert changed()
        dec.beta_weights[k].data = torch.tensor([0.25])
        assert changed()
        dec.beta_weights[k] = torch.nn.Parameter(torch.tensor([0.75]))
        assert changed()
        dec.load_state_dict({n: torch.full_like(v, 0.625) for n, v in dec.state_dict().items()})
        assert changed()
        opt = torch.optim.SGD(dec.parameters(), lr=0.1)
        for p_ in dec.parameters():
            p_.grad = torch.ones_like(p_)
        opt.step()
        assert changed()


def test_decoder_objects_pickle_without_native_handles():
    import copy
    import codes
    from neural_2d_decoder import Neural2DMinSumDecoder
    dec = Neural2DMinSumDecoder(codes.load_code("small_96_48", 10), 2, 4)
    dec2 = copy.deepcopy(dec)
    assert dec2._engine is None and list(dec2.state_dict()) == list(dec.state_dict())


def test_reference_import_lines_work_verbatim():
    """the import statements of the reference's own callers (examples.py:17-22, comprehensive_test.py:15-20,
    simulation_framework.py:19-23, training_framework.py:14-16) resolve against this package unchanged"""
    ns = {}
    exec("from ldpc_decoder import LDPCCode, BasicMinSumDecoder, create_test_ldpc_code, simulate_awgn_channel\n"
         "from neural_minsum_decoder import NeuralMinSumDecoder, NeuralOffsetMinSumDecoder, analyze_weight_patterns\n"
         "from neural_2d_decoder import Neural2DMinSumDecoder, Neural2DOffsetMinSumDecoder\n"
         "from rcq_decoder import RCQMinSumDecoder, WeightedRCQDecoder, NonUniformQuantizer\n"
         "from training_framework import TrainingConfig, PosteriorJointTrainer, GradientExplosionAnalyzer, create_dvbs2_code\n"
         "from simulation_framework import SimulationConfig, LDPSimulator, create_test_decoders\n"
         "from ldpc_decoder import LDPCCode, simulate_awgn_channel\n"
         "from rcq_decoder import WeightedRCQDecoder\n", ns)
    code = ns["create_dvbs2_code"]()
    assert (code.n, code.k, code.max_iterations) == (16200, 7200, 50) and code.H.shape == (9000, 16200)
    # the reference's ldpc_decoder module carries its own NeuralMinSumDecoder (:155): weights randn*0.1, no +0.7
    import ldpc_decoder
    import neural_minsum_decoder
    toy = ns["create_test_ldpc_code"]()
    torch.manual_seed(3)
    a = ldpc_decoder.NeuralMinSumDecoder(toy, max_iterations=2)
    torch.manual_seed(3)
    b = neural_minsum_decoder.NeuralMinSumDecoder(toy, max_iterations=2)
    assert type(a).__name__ == "NeuralMinSumDecoder" and len(a.alpha_weights) == 0
    assert list(a.beta_weights.keys()) == list(b.beta_weights.keys()) and len(a.beta_weights) == 2 * 13
    for k in a.beta_weights.keys():
        assert b.beta_weights[k].item() == pytest.approx(a.beta_weights[k].item() + 0.7, abs=1e-6)


def test_native_library_staleness_is_decided_by_content_hash(tmp_path, monkeypatch):
    """file times say nothing after a copy to another box.  build_native / load decide by a hash of the
    sources' CONTENT (plus the compile recipe) EMBEDDED in the library (ldpc_source_hash()); a library that does not
    carry the hash of the sources on disk is stale and is never loaded silently."""
    import subprocess
    import _native
    lib_path = os.path.join(PKG, "libldpc_hip.so")
    assert os.path.exists(lib_path), "run __graft_entry__.build() first"
    assert _native.built_hash(lib_path) == _native.source_hash() and not _native.is_stale(lib_path)
    # touching a source (newer mtime, same content) does not make the library stale ...
    src = os.path.join(PKG, "csrc", "ldpc_train.hip")
    st = os.stat(src)
    try:
        os.utime(src, (st.st_atime + 1e6, st.st_mtime + 1e6))
        assert not _native.is_stale(lib_path)
        assert _native.build_native(force=False) == lib_path            # returns at once: no hipcc run
    finally:
        os.utime(src, (st.st_atime, st.st_mtime))
    # ... a different recipe or different content does
    assert _native.source_hash(defines=("LDPC_X=1",)) != _native.source_hash()
    junk = tmp_path / "junk.so"
    junk.write_bytes(b"not a library")
    assert _native.built_hash(str(junk)) is None and _native.is_stale(str(junk))

    def fake_lib(name, digest):                                          # a library that CLAIMS to be built from `digest`
        c = tmp_path / (name + ".c")
        c.write_text('const char *ldpc_source_hash(void) { return "%s"; }\n' % digest)
        out = tmp_path / (name + ".so")
        subprocess.run(["gcc", "-shared", "-fPIC", "-o", str(out), str(c)], check=True)
        return str(out)

    other = fake_lib("other", "0" * 64)
    assert _native.built_hash(other) == "0" * 64 and _native.is_stale(other)
    assert not _native.is_stale(fake_lib("same", _native.source_hash()))
    # load() refuses a stale library loudly (no silent use, no compile inside an import)
    monkeypatch.setattr(_native, "LIB_PATH", other)
    monkeypatch.setattr(_native, "_lib", None)
    monkeypatch.delenv("LDPC_HIP_LIB", raising=False)
    with pytest.raises(_native.NativeEngineError, match="stale"):
        _native.load()


def test_native_graph_cache_holds_graphs_weakly():
    """one device graph per (TannerGraph, device) while the host graph lives; entries go when it dies (no GPU needed:
    the factory is injected)"""
    import gc
    from engine import _NativeGraph
    from tanner_graph import TannerGraph
    made = []

    class Fake:
        def __init__(self, tag):
            made.append(tag)

    before = len(_NativeGraph._cache)
    g1 = TannerGraph.from_dense(np.array([[1, 1, 0], [0, 1, 1]]))
    g2 = TannerGraph.from_dense(np.array([[1, 1, 0], [0, 1, 1]]))
    a = _NativeGraph._get(g1, 0, lambda: Fake("g1/0"))
    assert _NativeGraph._get(g1, 0, lambda: Fake("again")) is a and made == ["g1/0"]
    b = _NativeGraph._get(g1, 1, lambda: Fake("g1/1"))
    c = _NativeGraph._get(g2, 0, lambda: Fake("g2/0"))                 # equal content, different object: its own entry
    assert b is not a and c is not a and len(_NativeGraph._cache) == before + 3
    del g1
    gc.collect()
    assert len(_NativeGraph._cache) == before + 1                      # both device entries of g1 evicted
    del g2
    gc.collect()
    assert len(_NativeGraph._cache) == before


def test_bench_bare_launch_reports_a_failing_rank_at_once():
    """`python bench.py --gpus 2` on a box without a GPU: every rank fails at start-up; the parent polls all children,
    exits non-zero within seconds (no waiting in a collective) and shows each rank's stderr"""
    import subprocess
    import sys
    import time
    if torch.cuda.is_available():
        pytest.skip("checks the no-GPU behaviour")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    t0 = time.time()
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1"],
                         capture_output=True, text=True, timeout=300, env=env)
    assert out.returncode != 0 and out.stdout.strip() == ""
    assert "rank 0 stderr" in out.stderr and "rank 1 stderr" in out.stderr and "no CPU fallback" in out.stderr
    assert time.time() - t0 < 120


def test_bench_issue_limits_and_layered_model_are_consistent():
    """bench.py's counter-based limits of the resident kernel (VALU issue, LDS pipe) and the step-time model of the layered
    kernel: pure arithmetic on files under profiles/ -- checked here so that a malformed counters.json cannot break the line"""
    import importlib
    import json
    sys.path.insert(0, ROOT)
    bench = importlib.import_module("bench")
    cj = os.path.join(ROOT, "profiles", "counters.json")
    ent = json.load(open(cj))["basic"]["resident_decode"]
    lim = bench.issue_limits("basic", "resident_decode", 65536, 2.8)
    assert lim["valu_issue"]["ms"] == pytest.approx(ent["SQ_INSTS_VALU"] * 4 / 1024 / 2.4e9 * 1e3)
    assert lim["lds_pipe"]["ms"] == pytest.approx(ent["SQ_LDS_IDX_ACTIVE"] / 256 / 2.4e9 * 1e3)
    assert lim["binding"] in ("valu_issue", "lds_pipe") and 0 < lim[lim["binding"]]["frac"] <= 1.0
    assert bench.issue_limits("basic", "resident_decode", 4096, 0.2) is None          # counters exist for the default batch only
    assert bench.issue_limits("nope", "resident_decode", 65536, 1.0) is None

    class Eng:
        def info(self):
            return {"engine": "resident", "codewords_per_workgroup": 4, "lds_bytes": 31984, "threads_per_workgroup": 64}

    class G:
        m, E = 486, 6587
    r = bench.layered_roofline(Eng(), G(), 10, 65536, 9.6)
    assert r["dependent_steps"] == 4860 and r["waves_per_cu"] == 5 and r["rounds"] == pytest.approx(65536 / 4 / (5 * 256))
    assert r["achieved"] == pytest.approx(9.6e6 / r["rounds"] / 4860) and r["bound"] == "latency"
