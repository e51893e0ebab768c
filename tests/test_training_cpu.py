"""
CPU tests of the gradient path's test infrastructure and host logic (no GPU, no compute through the C ABI):
the gradient oracle against the golden gradients captured from the REAL reference under torch autograd, the
differentiable table builders, the trainer's data generation and the shape of the training C ABI.
"""
import numpy as np
import pytest
import torch

from conftest import golden_sub, load_golden, weights_dict


def _graph(gold, oracle_mod):
    if "H" in gold:
        return oracle_mod.OracleGraph(gold["H"].astype(np.int64))
    import codes
    tg = codes.load_code(str(gold["graph"]), max_iterations=1).tanner_graph()
    return oracle_mod.OracleGraph(n=tg.n, check_ptr=tg.check_ptr, var_idx=tg.var_idx)


@pytest.mark.parametrize("name,tags", [("grad_toy", [f"t{w}_T{T}" for w in (1, 2, 3, 4) for T in (3, 6)]),
                                       ("grad_small", ["t2_T4", "t1_T3"]), ("grad_ira", ["t2_T3"])])
def test_gradient_oracle_equals_reference_autograd(name, tags, oracle_mod):
    import grad_oracle
    gold = load_golden(name)
    g = _graph(gold, oracle_mod)
    for tag in tags:
        sub = golden_sub(gold, tag)
        T, wtype = int(sub["T"]), int(sub["wtype"])
        beta = weights_dict(sub["beta_keys"], sub["beta_vals"])
        alpha = weights_dict(sub["alpha_keys"], sub["alpha_vals"])
        bt, bs, at, as_ = oracle_mod.weight_tables(g, wtype, T, beta, alpha)
        gb, ga, post, iters = grad_oracle.table_grads(g, sub["llr"], bt, bs, at, as_, T)
        np.testing.assert_array_equal(iters, sub["iters"])
        np.testing.assert_allclose(post, sub["posterior"], rtol=1e-5, atol=1e-5)
        # columns that are constants of the sharing type have no reference parameter, hence no reference gradient
        if len(sub["grad_beta_keys"]):
            np.testing.assert_allclose(gb, sub["grad_beta_table"], rtol=2e-4, atol=2e-6, err_msg=tag)
        if len(sub["grad_alpha_keys"]):
            np.testing.assert_allclose(ga, sub["grad_alpha_table"], rtol=2e-4, atol=2e-6, err_msg=tag)


@pytest.mark.parametrize("name,tags", [("grad_toy", ["o1_T4", "o2_T4", "o3_T4", "o4_T4"]), ("grad_small", ["o2_T4"])])
def test_gradient_oracle_offset_form_equals_reference_autograd(name, tags, oracle_mod):
    import grad_oracle
    gold = load_golden(name)
    g = _graph(gold, oracle_mod)
    for tag in tags:
        sub = golden_sub(gold, tag)
        T, wtype = int(sub["T"]), int(sub["wtype"])
        bt, bs, at, as_ = oracle_mod.weight_tables(g, wtype, T, weights_dict(sub["beta_keys"], sub["beta_vals"]),
                                                   weights_dict(sub["alpha_keys"], sub["alpha_vals"]),
                                                   beta_default=0.0, alpha_default=0.0)
        gb, ga, post, iters = grad_oracle.table_grads(g, sub["llr"], bt, bs, at, as_[g.var_idx], T, offset=True)
        np.testing.assert_array_equal(iters, sub["iters"])
        np.testing.assert_allclose(post, sub["posterior"], rtol=1e-5, atol=1e-5)
        if len(sub["grad_beta_keys"]):
            np.testing.assert_allclose(gb, sub["grad_beta_table"], rtol=2e-4, atol=2e-6, err_msg=tag)
        if len(sub["grad_alpha_keys"]):
            np.testing.assert_allclose(ga, sub["grad_alpha_table"], rtol=2e-4, atol=2e-6, err_msg=tag)
    if name == "grad_toy":
        sub = golden_sub(gold, "edgeoff")
        T = int(sub["T"])
        bt = oracle_mod.edge_weight_table(g, T, weights_dict(sub["beta_keys"], sub["beta_vals"]))
        gb, _, _, iters = grad_oracle.table_grads(g, sub["llr"], bt, np.arange(g.E), np.zeros((T, 1), np.float32),
                                                  np.zeros(g.E, np.int64), T, offset=True)
        np.testing.assert_array_equal(iters, sub["iters"])
        np.testing.assert_allclose(gb, sub["grad_beta_table"], rtol=2e-4, atol=2e-6)


def test_gradient_oracle_edge_weights_equal_reference_autograd(oracle_mod):
    import grad_oracle
    gold = load_golden("grad_toy")
    sub = golden_sub(gold, "edge")
    g = _graph(gold, oracle_mod)
    T = int(sub["T"])
    bt = oracle_mod.edge_weight_table(g, T, weights_dict(sub["beta_keys"], sub["beta_vals"]))
    gb, _, post, iters = grad_oracle.table_grads(g, sub["llr"], bt, np.arange(g.E), np.ones((T, 1), np.float32),
                                                 np.zeros(g.n, np.int64), T)
    np.testing.assert_array_equal(iters, sub["iters"])
    np.testing.assert_allclose(gb, sub["grad_beta_table"], rtol=2e-4, atol=2e-6)


def test_gradient_oracle_matches_finite_differences(oracle_mod):
    """independent of autograd: central differences of the loss in fp64 on the toy code"""
    import grad_oracle
    gold = load_golden("grad_toy")
    sub = golden_sub(gold, "t2_T3")
    g = _graph(gold, oracle_mod)
    T = 3
    bt, bs, at, as_ = oracle_mod.weight_tables(g, 2, T, weights_dict(sub["beta_keys"], sub["beta_vals"]),
                                               weights_dict(sub["alpha_keys"], sub["alpha_vals"]), dtype=np.float64)
    llr = sub["llr"][:6].astype(np.float64)
    gb, ga, _, iters = grad_oracle.table_grads(g, llr, bt, bs, at, as_, T, dtype=torch.float64)

    def loss_at(b, a):
        post, _, it = grad_oracle.forward(g, llr, torch.tensor(b), bs, torch.tensor(a), as_, T, True, torch.float64)
        assert np.array_equal(it.numpy(), iters)           # the perturbation must not change where codewords stop
        return float(grad_oracle.bce_loss_sum(post))
    h = 1e-6
    for t in range(T):
        for s in range(bt.shape[1]):
            bp, bm = bt.copy(), bt.copy()
            bp[t, s] += h
            bm[t, s] -= h
            assert abs((loss_at(bp, at) - loss_at(bm, at)) / (2 * h) - gb[t, s]) < 1e-6
        for s in range(at.shape[1]):
            ap, am = at.copy(), at.copy()
            ap[t, s] += h
            am[t, s] -= h
            assert abs((loss_at(bt, ap) - loss_at(bt, am)) / (2 * h) - ga[t, s]) < 1e-6


@pytest.mark.parametrize("wtype", [1, 2, 3, 4])
def test_differentiable_tables_equal_the_uploaded_tables(wtype):
    import codes
    from neural_2d_decoder import Neural2DMinSumDecoder
    code = codes.load_code("small_96_48", max_iterations=4)
    torch.manual_seed(wtype)
    dec = Neural2DMinSumDecoder(code, wtype, 4)
    bt, at = dec._sharing_layout().tables_torch(dec.beta_weights, dec.alpha_weights, 4, dec._beta_default, dec._alpha_default)
    b_np, a_np = dec.weight_tables()
    assert np.array_equal(bt.detach().numpy(), b_np) and np.array_equal(at.detach().numpy(), a_np)
    # autograd routes a table gradient to exactly the parameter behind each cell
    gb, ga = torch.randn_like(bt), torch.randn_like(at)
    ((bt * gb).sum() + (at * ga).sum()).backward()
    lay = dec._sharing_layout()
    for t in range(4):
        for s, suf in enumerate(lay.beta_suffix):
            if suf is not None:
                assert dec.beta_weights[f"iter_{t}_{suf}"].grad.item() == pytest.approx(gb[t, s].item())
        for s, suf in enumerate(lay.alpha_suffix):
            if suf is not None:
                assert dec.alpha_weights[f"iter_{t}_{suf}"].grad.item() == pytest.approx(ga[t, s].item())


def test_training_config_and_data_generation():
    from ldpc_decoder import create_test_ldpc_code
    from training_framework import TrainingConfig
    import training_framework as tf
    cfg = TrainingConfig()
    assert (cfg.batch_size, cfg.num_epochs, cfg.learning_rate, cfg.snr_range, cfg.snr_step, cfg.max_grad_norm,
            cfg.use_posterior_training, cfg.use_gradient_clipping, cfg.clip_threshold) == \
        (32, 100, 0.001, (0.0, 6.0), 0.5, 1.0, True, False, 1e-3)             # training_framework.py:23-35
    code = create_test_ldpc_code()

    class Holder(tf.PosteriorJointTrainer):                     # data generation needs no device
        def __init__(self, config):
            self.config = config
    llr, tgt = Holder(TrainingConfig(seed=3, snr_range=(2.0, 8.0))).generate_training_data(code, 400)
    assert llr.shape == (400, 7) and llr.dtype == torch.float32 and float(tgt.abs().sum()) == 0.0
    assert float(llr.mean()) > 0                                 # decoder convention: bit 0 <-> positive LLR
    llr2, _ = Holder(TrainingConfig(seed=3, snr_range=(2.0, 8.0))).generate_training_data(code, 400)
    assert torch.equal(llr, llr2)
    np.random.seed(0)
    ref_llr, _ = Holder(TrainingConfig(llr_convention="reference", snr_range=(2.0, 8.0))).generate_training_data(code, 50)
    assert float(ref_llr.mean()) < 0                             # the reference's channel convention, literally
    post = torch.tensor([[2.0, -1.0, 0.5, 3.0, 1.0, 0.2, 4.0]], requires_grad=True)
    loss = Holder(cfg).compute_loss(None, torch.zeros(1, 7), post)
    assert loss.item() == pytest.approx(torch.nn.functional.softplus(-post).mean().item())


def test_training_abi_is_declared_and_refuses_to_run_without_a_gpu():
    import _native as nat
    for sym in ("ldpc_train_saved_bytes", "ldpc_train_workspace_bytes", "ldpc_decode_saving", "ldpc_backward"):
        assert sym in nat.EXPORTS
    from ldpc_decoder import create_test_ldpc_code
    from neural_2d_decoder import Neural2DMinSumDecoder
    dec = Neural2DMinSumDecoder(create_test_ldpc_code(), 2, 3)
    if not torch.cuda.is_available():
        with pytest.raises(Exception) as e:                       # grad-enabled call: still no CPU fallback
            dec(torch.zeros(7))
        assert "GPU" in str(e.value) or "HIP" in str(e.value) or "cuda" in str(e.value).lower()


def test_gradient_oracle_input_gradient_equals_reference_autograd(oracle_mod):
    """d loss/d llr (the reference's forward is differentiable in its input too)"""
    import grad_oracle
    gold = load_golden("grad_llr_toy")
    g = _graph(gold, oracle_mod)
    T = int(gold["n2d_T"])
    bt, bs, at, as_ = oracle_mod.weight_tables(g, 2, T, weights_dict(gold["n2d_beta_keys"], gold["n2d_beta_vals"]),
                                               weights_dict(gold["n2d_alpha_keys"], gold["n2d_alpha_vals"]))
    out = grad_oracle.table_grads(g, gold["llr"], bt, bs, at, as_, T, want_llr=True)
    np.testing.assert_array_equal(out[3], gold["n2d_iters"])
    np.testing.assert_allclose(out[4], gold["n2d_grad_llr"], rtol=2e-4, atol=2e-6)
    T = int(gold["oms_T"])
    bt, bs, at, as_ = oracle_mod.weight_tables(g, 2, T, weights_dict(gold["oms_beta_keys"], gold["oms_beta_vals"]),
                                               weights_dict(gold["oms_alpha_keys"], gold["oms_alpha_vals"]),
                                               beta_default=0.0, alpha_default=0.0)
    out = grad_oracle.table_grads(g, gold["llr"], bt, bs, at, as_[g.var_idx], T, offset=True, want_llr=True)
    np.testing.assert_array_equal(out[3], gold["oms_iters"])
    np.testing.assert_allclose(out[4], gold["oms_grad_llr"], rtol=2e-4, atol=2e-6)


@pytest.mark.parametrize("tag", ["toy", "small"])
@pytest.mark.parametrize("kind", ["n2d", "oms"])
def test_gradient_oracle_on_exact_ties_equals_reference_autograd(tag, kind, oracle_mod):
    """half-integer LLRs: several edges of a check share the second-smallest magnitude; the reference's torch.min(temp_mags)
    splits that gradient evenly (neural_2d_decoder.py:179).  d loss/d llr and both table gradients against the reference's."""
    import codes
    import grad_oracle
    gold = load_golden("grad_ties")
    assert int(gold[f"{tag}_tied_checks"]) >= 10
    if tag == "toy":
        g = oracle_mod.OracleGraph(gold["toy_H"].astype(np.int64))
    else:
        tg = codes.load_code("small_96_48", 10).tanner_graph()
        g = oracle_mod.OracleGraph(n=tg.n, check_ptr=tg.check_ptr, var_idx=tg.var_idx)
    T, off = int(gold[f"{tag}_{kind}_T"]), kind == "oms"
    dflt = dict(beta_default=0.0, alpha_default=0.0) if off else {}
    bt, bs, at, as_ = oracle_mod.weight_tables(g, 2, T, weights_dict(gold[f"{tag}_{kind}_beta_keys"], gold[f"{tag}_{kind}_beta_vals"]),
                                               weights_dict(gold[f"{tag}_{kind}_alpha_keys"], gold[f"{tag}_{kind}_alpha_vals"]), **dflt)
    out = grad_oracle.table_grads(g, gold[f"{tag}_llr"], bt, bs, at, as_[g.var_idx] if off else as_, T, offset=off, want_llr=True)
    np.testing.assert_array_equal(out[3], gold[f"{tag}_{kind}_iters"])
    np.testing.assert_allclose(out[4], gold[f"{tag}_{kind}_grad_llr"], rtol=2e-4, atol=2e-6)
    np.testing.assert_allclose(out[0], gold[f"{tag}_{kind}_grad_beta_table"], rtol=2e-4, atol=2e-6)
    np.testing.assert_allclose(out[1], gold[f"{tag}_{kind}_grad_alpha_table"], rtol=2e-4, atol=2e-6)
