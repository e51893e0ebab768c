"""
GPU tests (-m gpu) of the gradient path: d loss / d beta, d loss / d alpha from the HIP
backward sweeps (ldpc_decode_saving / ldpc_backward through the host classes' autograd bridge) against
(1) the REAL reference under torch autograd -- tests/golden/grad_*.npz, made by oracle/make_golden.py --
and (2) the gradient oracle (oracle/grad_oracle.py, a differentiable CPU restatement pinned to those
fixtures) on fresh inputs at sizes the reference cannot reach.

Bar: floating point, tolerance rtol 1e-4 / atol 2e-6 against the golden gradients (fp32 sums in a different
association order than autograd's), rtol 2e-3 on the large random cases (sums over hundreds of codewords).
"""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import golden_sub, load_golden, weights_dict

pytestmark = pytest.mark.gpu


def make_code(gold, max_iterations):
    from ldpc_decoder import LDPCCode
    import codes
    if "H" in gold:
        H = gold["H"].astype(np.int64)
        return LDPCCode(n=H.shape[1], k=H.shape[1] - H.shape[0], H=H, max_iterations=int(max_iterations))
    return codes.load_code(str(gold["graph"]), max_iterations=int(max_iterations))


def codeword_loss_sum(post):
    """sum over codewords of the reference's per-codeword loss (training_framework.py:101)"""
    return F.binary_cross_entropy_with_logits(-post, torch.zeros_like(post), reduction="none").mean(dim=1).sum()


def param_grads(params):
    return {k: (0.0 if p.grad is None else float(p.grad.item())) for k, p in params.items()}


def assert_grads(got, want, what, rtol=1e-4, atol=2e-6):
    assert set(got) == set(want), what
    for k in want:
        assert abs(got[k] - want[k]) <= atol + rtol * abs(want[k]), f"{what} {k}: {got[k]} vs reference {want[k]}"


def check_neural2d_block(gold, sub, gpu, batched=True, offset=False):
    from neural_2d_decoder import Neural2DMinSumDecoder, Neural2DOffsetMinSumDecoder
    T, wtype = int(sub["T"]), int(sub["wtype"])
    dec = (Neural2DOffsetMinSumDecoder if offset else Neural2DMinSumDecoder)(make_code(gold, T), wtype, T)
    beta = weights_dict(sub["beta_keys"], sub["beta_vals"])
    alpha = weights_dict(sub["alpha_keys"], sub["alpha_vals"])
    sd = {f"beta_weights.{k}": torch.tensor([v]) for k, v in beta.items()}
    sd.update({f"alpha_weights.{k}": torch.tensor([v]) for k, v in alpha.items()})
    dec.load_state_dict(sd)
    llr = torch.from_numpy(sub["llr"])
    want_b = weights_dict(sub["grad_beta_keys"], sub["grad_beta_vals"])
    want_a = weights_dict(sub["grad_alpha_keys"], sub["grad_alpha_vals"])
    if batched:
        bits, post, iters = dec(llr.to(gpu))
        np.testing.assert_array_equal(iters.cpu().numpy(), sub["iters"])
        np.testing.assert_allclose(post.detach().cpu().numpy(), sub["posterior"], rtol=1e-5, atol=1e-5)
        if want_b or want_a:
            assert post.requires_grad and not bits.requires_grad
            codeword_loss_sum(post).backward()
    else:                       # the reference's own call shape: one CPU vector at a time, gradients accumulate
        for b in range(llr.shape[0]):
            _, post, it = dec(llr[b])
            assert post.device.type == "cpu" and it == int(sub["iters"][b])
            loss = F.binary_cross_entropy_with_logits(-post, torch.zeros_like(post))
            assert abs(loss.item() - float(sub["loss"][b])) <= 1e-5 * max(1.0, abs(float(sub["loss"][b])))
            if post.requires_grad:
                loss.backward()
    assert_grads(param_grads(dec.beta_weights), want_b, f"type {wtype} T {T} beta")
    assert_grads(param_grads(dec.alpha_weights), want_a, f"type {wtype} T {T} alpha")


@pytest.mark.parametrize("wtype", [1, 2, 3, 4])
@pytest.mark.parametrize("T", [3, 6])
def test_golden_toy_neural2d_gradients(gpu_device, wtype, T):
    gold = load_golden("grad_toy")
    check_neural2d_block(gold, golden_sub(gold, f"t{wtype}_T{T}"), gpu_device)


def test_golden_toy_single_vector_calls_accumulate(gpu_device):
    gold = load_golden("grad_toy")
    check_neural2d_block(gold, golden_sub(gold, "t2_T6"), gpu_device, batched=False)


@pytest.mark.parametrize("name,tag", [("grad_small", "t2_T4"), ("grad_small", "t1_T3"), ("grad_ira", "t2_T3")])
def test_golden_small_and_ira_neural2d_gradients(gpu_device, name, tag):
    """96x48 code and the (1998,1512) benchmark code: gradients of the real reference under autograd"""
    gold = load_golden(name)
    check_neural2d_block(gold, golden_sub(gold, tag), gpu_device)


@pytest.mark.parametrize("name,tag", [("grad_toy", "o1_T4"), ("grad_toy", "o2_T4"), ("grad_toy", "o3_T4"),
                                      ("grad_toy", "o4_T4"), ("grad_small", "o2_T4")])
def test_golden_offset_form_gradients(gpu_device, name, tag):
    """Neural2DOffsetMinSumDecoder: relu(min - beta) - alpha, both trainable"""
    gold = load_golden(name)
    check_neural2d_block(gold, golden_sub(gold, tag), gpu_device, offset=True)


def test_golden_toy_edge_offset_gradients(gpu_device):
    from neural_minsum_decoder import NeuralOffsetMinSumDecoder
    import oracle
    gold = load_golden("grad_toy")
    sub = golden_sub(gold, "edgeoff")
    T = int(sub["T"])
    dec = NeuralOffsetMinSumDecoder(make_code(gold, T), T)
    dec.load_state_dict({f"beta_weights.{k}": torch.tensor([v])
                         for k, v in weights_dict(sub["beta_keys"], sub["beta_vals"]).items()})
    bits, post, iters = dec(torch.from_numpy(sub["llr"]).to(gpu_device))
    np.testing.assert_array_equal(iters.cpu().numpy(), sub["iters"])
    np.testing.assert_allclose(post.detach().cpu().numpy(), sub["posterior"], rtol=1e-5, atol=1e-5)
    codeword_loss_sum(post).backward()
    got = oracle.edge_weight_table(oracle.OracleGraph(gold["H"]), T, param_grads(dec.beta_weights))
    np.testing.assert_allclose(got, sub["grad_beta_table"], rtol=1e-4, atol=2e-6)


def test_golden_toy_edge_weight_gradients(gpu_device):
    """NeuralMinSumDecoder: one beta per (iteration, edge)"""
    from neural_minsum_decoder import NeuralMinSumDecoder
    import oracle
    gold = load_golden("grad_toy")
    sub = golden_sub(gold, "edge")
    T = int(sub["T"])
    dec = NeuralMinSumDecoder(make_code(gold, T), T)
    beta = weights_dict(sub["beta_keys"], sub["beta_vals"])
    dec.load_state_dict({f"beta_weights.{k}": torch.tensor([v]) for k, v in beta.items()})
    bits, post, iters = dec(torch.from_numpy(sub["llr"]).to(gpu_device))
    np.testing.assert_array_equal(iters.cpu().numpy(), sub["iters"])
    codeword_loss_sum(post).backward()
    g = oracle.OracleGraph(gold["H"])
    got = oracle.edge_weight_table(g, T, param_grads(dec.beta_weights))
    np.testing.assert_allclose(got, sub["grad_beta_table"], rtol=1e-4, atol=2e-6)


@pytest.mark.parametrize("early_stop", [True, False])
@pytest.mark.parametrize("batch", [300, 37])
def test_engine_backward_vs_oracle_on_the_1998_code(gpu_device, early_stop, batch):
    """fresh inputs, arbitrary d loss/d posterior, both tile widths (VEC 4 with a ragged last tile; VEC 1)"""
    import codes
    import grad_oracle
    import oracle
    from neural_2d_decoder import Neural2DMinSumDecoder
    T = 5
    code = codes.load_code("ira_1998_1512", max_iterations=T)
    dec = Neural2DMinSumDecoder(code, 2, T)
    rng = np.random.default_rng(11 + batch)
    with torch.no_grad():
        for p in dec.beta_weights.values():
            p.fill_(float(rng.uniform(0.5, 1.0)))
        for p in dec.alpha_weights.values():
            p.fill_(float(rng.uniform(0.8, 1.2)))
    n = code.n
    snr = np.where(np.arange(batch) % 2 == 0, 2.0, 6.5)               # half never converge, half stop early
    s2 = 10.0 ** (-snr / 10.0)
    llr = (2.0 * (1.0 + np.sqrt(s2)[:, None] * rng.standard_normal((batch, n))) / s2[:, None]).astype(np.float32)
    eng = dec._get_engine(gpu_device)
    x = torch.from_numpy(llr).to(gpu_device)
    res, saved = eng.decode_saving(x, early_stop=early_stop)
    plain = eng.decode(x, early_stop=early_stop)                       # same arithmetic as the ordinary decode
    assert torch.equal(res.bits, plain.bits) and torch.equal(res.iterations, plain.iterations)
    assert torch.equal(res.posterior, plain.posterior)
    if early_stop:
        assert len(torch.unique(res.iterations)) >= 2
    gpost = rng.standard_normal((batch, n)).astype(np.float32)
    gb, ga, _ = eng.backward(saved, x, res.iterations, torch.from_numpy(gpost).to(gpu_device))

    g = oracle.OracleGraph(n=n, check_ptr=code.tanner_graph().check_ptr, var_idx=code.tanner_graph().var_idx)
    bt_np, at_np = dec.weight_tables()
    lay = dec._sharing_layout()
    bt = torch.tensor(bt_np, requires_grad=True)
    at = torch.tensor(at_np, requires_grad=True)
    post, _, iters = grad_oracle.forward(g, llr, bt, lay.beta_slot, at, lay.alpha_slot, T, early_stop)
    agree = (iters.numpy() == res.iterations.cpu().numpy()) & \
        np.all(np.abs(post.detach().numpy() - res.posterior.cpu().numpy()) <= 1e-4 * np.maximum(1, np.abs(post.detach().numpy())), axis=1)
    assert agree.mean() > 0.98        # a different summation order may flip a near-tie; those codewords are left out
    gpost_m = gpost * agree[:, None]
    (post * torch.from_numpy(gpost_m)).sum().backward()
    gb2, ga2, _ = eng.backward(saved, x, res.iterations, torch.from_numpy(gpost_m).to(gpu_device))
    for got, want, what in ((gb2, bt.grad, "beta"), (ga2, at.grad, "alpha")):
        got, want = got.cpu().numpy(), want.numpy()
        scale = np.abs(want).max()
        np.testing.assert_allclose(got, want, rtol=2e-3, atol=2e-4 * scale, err_msg=what)
    assert torch.isfinite(gb).all() and torch.isfinite(ga).all()
    assert float(ga[T - 1].abs().max()) == 0.0       # alpha of the last iteration feeds no returned posterior


def test_stale_weights_are_restored_for_backward(gpu_device):
    """forward A, weights change, forward B, then backward A: A's gradient uses A's tables, B's engine state survives"""
    from neural_2d_decoder import Neural2DMinSumDecoder
    gold = load_golden("grad_toy")
    sub = golden_sub(gold, "t2_T6")
    T = 6
    dec = Neural2DMinSumDecoder(make_code(gold, T), 2, T)
    beta = weights_dict(sub["beta_keys"], sub["beta_vals"])
    alpha = weights_dict(sub["alpha_keys"], sub["alpha_vals"])
    sd = {f"beta_weights.{k}": torch.tensor([v]) for k, v in beta.items()}
    sd.update({f"alpha_weights.{k}": torch.tensor([v]) for k, v in alpha.items()})
    dec.load_state_dict(sd)
    llr = torch.from_numpy(sub["llr"]).to(gpu_device)
    _, post_a, _ = dec(llr)
    with torch.no_grad():
        for p in dec.parameters():
            p.mul_(0.5)
    with torch.no_grad():
        _, post_b, _ = dec(llr)
    codeword_loss_sum(post_a).backward()
    # gradients flow to the (now modified) leaf parameters with the values of forward A's tables
    assert_grads(param_grads(dec.beta_weights), weights_dict(sub["grad_beta_keys"], sub["grad_beta_vals"]), "beta")
    with torch.no_grad():
        _, post_b2, _ = dec(llr)
    assert torch.equal(post_b, post_b2)


def test_backward_after_the_decoder_is_gone(gpu_device):
    """the autograd node keeps its engine alive -- backward works after the decoder object (and with it the
    only other reference to the engine) has been deleted, and after the decoder rebuilt its engine for another key"""
    import gc
    from neural_2d_decoder import Neural2DMinSumDecoder
    gold = load_golden("grad_toy")
    sub = golden_sub(gold, "t2_T6")
    T = 6
    dec = Neural2DMinSumDecoder(make_code(gold, T), 2, T)
    beta = weights_dict(sub["beta_keys"], sub["beta_vals"])
    alpha = weights_dict(sub["alpha_keys"], sub["alpha_vals"])
    sd = {f"beta_weights.{k}": torch.tensor([v]) for k, v in beta.items()}
    sd.update({f"alpha_weights.{k}": torch.tensor([v]) for k, v in alpha.items()})
    dec.load_state_dict(sd)
    llr = torch.from_numpy(sub["llr"]).to(gpu_device)
    _, post, _ = dec(llr)
    betas, alphas = dict(dec.beta_weights.items()), dict(dec.alpha_weights.items())     # the leaves outlive the module
    dec._engine = None                                  # what a device / max_iterations change does: the engine is rebuilt
    del dec
    gc.collect()
    codeword_loss_sum(post).backward()
    assert_grads(param_grads(betas), weights_dict(sub["grad_beta_keys"], sub["grad_beta_vals"]), "beta")
    assert_grads(param_grads(alphas), weights_dict(sub["grad_alpha_keys"], sub["grad_alpha_vals"]), "alpha")


def test_forms_without_a_gradient_path(gpu_device):
    """RCQ: the reference's quantiser passes no gradient either (posterior without grad_fn); the native call says so"""
    from ldpc_decoder import create_test_ldpc_code
    from rcq_decoder import WeightedRCQDecoder
    from neural_2d_decoder import Neural2DMinSumDecoder
    code = create_test_ldpc_code()
    dec = WeightedRCQDecoder(code, 3, 8, [(3.0, 1.3)], 2, 4)
    x = torch.randn(5, code.n, device=gpu_device)
    bits, post, iters = dec(x)
    assert not post.requires_grad
    with pytest.raises(NotImplementedError):
        dec._get_engine(gpu_device).decode_saving(x)
    nd = Neural2DMinSumDecoder(code, 2, 4)
    with torch.no_grad():
        _, p2, _ = nd(x)
    assert not p2.requires_grad
    for p in nd.parameters():
        p.requires_grad_(False)
    _, p3, _ = nd(x)
    assert not p3.requires_grad


def test_trainer_reduces_the_loss(gpu_device):
    """PosteriorJointTrainer end to end on the 96x48 code: Adam steps on HIP gradients lower the training loss"""
    import codes
    from neural_2d_decoder import Neural2DMinSumDecoder
    from training_framework import GradientExplosionAnalyzer, PosteriorJointTrainer, TrainingConfig
    torch.manual_seed(5)
    code = codes.load_code("small_96_48", max_iterations=5)
    model = Neural2DMinSumDecoder(code, 2, 5)
    with torch.no_grad():                     # a deliberately poor start: all weights 0.3
        for p in model.parameters():
            p.fill_(0.3)
    cfg = TrainingConfig(batch_size=64, num_epochs=6, learning_rate=0.05, snr_range=(1.0, 4.0), device="cuda", seed=9)
    trainer = PosteriorJointTrainer(model, cfg)
    hist = trainer.train(code, num_train_samples=512, num_val_samples=128)
    assert len(hist["train_losses"]) >= 2 and hist["train_losses"][-1] < hist["train_losses"][0]
    assert all(np.isfinite(hist["gradient_norms"])) and hist["gradient_norms"][0] > 0
    vals = torch.cat([p.detach().reshape(-1) for p in model.parameters()])
    assert float((vals - 0.3).abs().max()) > 0.05
    out = GradientExplosionAnalyzer(model, code).analyze_gradient_explosion(num_samples=8)
    assert len(out["gradient_magnitudes"]) == 8 and np.isfinite(out["mean_gradient"])


def test_zero_iterations_and_empty_batch(gpu_device):
    from ldpc_decoder import create_test_ldpc_code
    from neural_2d_decoder import Neural2DMinSumDecoder
    code = create_test_ldpc_code()
    dec = Neural2DMinSumDecoder(code, 2, 0)
    assert len(list(dec.parameters())) == 0
    dec3 = Neural2DMinSumDecoder(code, 2, 3)
    bits, post, iters = dec3(torch.zeros(0, code.n, device=gpu_device))
    assert post.shape == (0, code.n)
    if post.requires_grad:
        post.sum().backward()
        assert all(float(p.grad.abs().sum()) == 0.0 for p in dec3.parameters() if p.grad is not None)


def test_second_backward_follows_torch_semantics(gpu_device):
    """the saved messages are ordinary saved tensors of the registered operator's autograd node: a second backward
    works with retain_graph=True (gradients accumulate) and raises torch's usual error without it"""
    from ldpc_decoder import create_test_ldpc_code
    from neural_2d_decoder import Neural2DMinSumDecoder
    code = create_test_ldpc_code()
    dec = Neural2DMinSumDecoder(code, 2, 4)
    with torch.no_grad():
        for p in dec.parameters():
            p.fill_(0.8)
    x = torch.randn(9, code.n, device=gpu_device) + 1.0
    _, post, _ = dec(x)
    loss = codeword_loss_sum(post)
    loss.backward()
    g1 = param_grads(dec.beta_weights)
    with pytest.raises(RuntimeError, match="second time|already been freed"):
        loss.backward()
    dec.zero_grad()
    _, post, _ = dec(x)
    loss = codeword_loss_sum(post)
    loss.backward(retain_graph=True)
    loss.backward()                                   # accumulates: twice the gradient
    g2 = param_grads(dec.beta_weights)
    for k in g1:
        assert g2[k] == pytest.approx(2 * g1[k], rel=1e-5, abs=1e-7)


def test_registered_torch_ops(gpu_device, oracle_mod):
    """torch.ops.ldpc.* exist, pass torch.library.opcheck, and give exactly what the C-ABI path gives"""
    import codes
    import torch_ops
    from neural_2d_decoder import Neural2DMinSumDecoder
    from rcq_decoder import RCQMinSumDecoder
    assert hasattr(torch.ops.ldpc, "decode") and hasattr(torch.ops.ldpc, "minsum_decode_train")
    assert hasattr(torch.ops.ldpc, "minsum_backward")
    code = codes.load_code("small_96_48", 6)
    dec = Neural2DMinSumDecoder(code, 2, 6)
    with torch.no_grad():
        for p in dec.parameters():
            p.fill_(0.85)
    x = (torch.randn(37, code.n, device=gpu_device) * 1.3 + 1.5).contiguous()
    eng = dec._get_engine(gpu_device)
    h = torch_ops.engine_handle(eng)
    # inference op == DecodeEngine.decode (ctypes straight into ldpc_decode)
    direct = eng.decode(x, early_stop=True, want_packed=True)
    bits, post, iters, succ, packed = torch.ops.ldpc.decode(x, h, True, True, True)
    assert torch.equal(bits, direct.bits) and torch.equal(post, direct.posterior) and torch.equal(iters, direct.iterations)
    assert torch.equal(succ, direct.success) and torch.equal(packed, direct.packed_bits)
    torch.library.opcheck(torch.ops.ldpc.decode, (x, h, True, True, False))
    torch.library.opcheck(torch.ops.ldpc.decode, (x, h, False, False, True))
    rcq = RCQMinSumDecoder(code, 3, 8, [(3.0, 1.3), (5.0, 1.3), (7.0, 1.3)], max_iterations=6)
    hr = torch_ops.engine_handle(rcq._get_engine(gpu_device))
    torch.library.opcheck(torch.ops.ldpc.decode, (x, hr, True, False, False))
    # training op: forward identical, gradients == the engine's own backward, opcheck incl. autograd registration
    bt, at = dec._sharing_layout().tables_torch(dec.beta_weights, dec.alpha_weights, 6, 0.7, 1.0)
    bt = bt.detach().clone().requires_grad_(True)
    at = at.detach().clone().requires_grad_(True)
    post2, bits2, iters2, saved = torch.ops.ldpc.minsum_decode_train(x, bt, at, h, True, False)
    assert torch.equal(post2.detach(), direct.posterior) and torch.equal(bits2, direct.bits) and torch.equal(iters2, direct.iterations)
    gp = torch.randn_like(post2)
    post2.backward(gp)
    res, saved_d = eng.decode_saving(x, early_stop=True)
    gb, ga, _ = eng.backward(saved_d, x, res.iterations, gp)
    torch.testing.assert_close(bt.grad, gb.cpu(), rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(at.grad, ga.cpu(), rtol=1e-5, atol=1e-6)
    xs = x[:5].contiguous()
    torch.library.opcheck(torch.ops.ldpc.minsum_decode_train,
                          (xs, bt.detach().clone().requires_grad_(True), at.detach().clone().requires_grad_(True), h, True, False),
                          test_utils=("test_schema", "test_autograd_registration", "test_faketensor"))


@pytest.mark.parametrize("kind", ["n2d", "oms"])
def test_golden_input_gradient(gpu_device, kind):
    """d loss/d llr through loss.backward(), batched on the GPU and one CPU vector at a time, against the reference"""
    from neural_2d_decoder import Neural2DMinSumDecoder, Neural2DOffsetMinSumDecoder
    gold = load_golden("grad_llr_toy")
    T = int(gold[f"{kind}_T"])
    dec = (Neural2DMinSumDecoder if kind == "n2d" else Neural2DOffsetMinSumDecoder)(make_code(gold, T), 2, T)
    sd = {f"beta_weights.{k}": torch.tensor([v]) for k, v in weights_dict(gold[f"{kind}_beta_keys"], gold[f"{kind}_beta_vals"]).items()}
    sd.update({f"alpha_weights.{k}": torch.tensor([v])
               for k, v in weights_dict(gold[f"{kind}_alpha_keys"], gold[f"{kind}_alpha_vals"]).items()})
    dec.load_state_dict(sd)
    x = torch.from_numpy(gold["llr"]).to(gpu_device).requires_grad_(True)
    _, post, iters = dec(x)
    np.testing.assert_array_equal(iters.cpu().numpy(), gold[f"{kind}_iters"])
    codeword_loss_sum(post).backward()
    np.testing.assert_allclose(x.grad.cpu().numpy(), gold[f"{kind}_grad_llr"], rtol=1e-4, atol=2e-6)
    for p in dec.parameters():                       # frozen weights, trainable front end: only the input gradient flows
        p.requires_grad_(False)
    x1 = torch.from_numpy(gold["llr"][3].copy()).requires_grad_(True)
    _, p1, _ = dec(x1)
    assert p1.requires_grad and p1.device.type == "cpu"
    F.binary_cross_entropy_with_logits(-p1, torch.zeros_like(p1)).backward()
    np.testing.assert_allclose(x1.grad.numpy(), gold[f"{kind}_grad_llr"][3], rtol=1e-4, atol=2e-6)


@pytest.mark.parametrize("tag", ["toy", "small"])
@pytest.mark.parametrize("kind", ["n2d", "oms"])
def test_golden_gradients_on_exact_ties(gpu_device, tag, kind):
    """the reference under autograd on half-integer LLRs (exact ties for the second minimum of a check): its torch.min
    splits that gradient evenly among the tied edges -- d loss/d llr and the weight gradients of the HIP backward sweeps
    must follow (tests/golden/grad_ties.npz; round 2 sent the whole gradient to the first tied edge)"""
    import codes
    from ldpc_decoder import LDPCCode
    from neural_2d_decoder import Neural2DMinSumDecoder, Neural2DOffsetMinSumDecoder
    import oracle
    gold = load_golden("grad_ties")
    T = int(gold[f"{tag}_{kind}_T"])
    if tag == "toy":
        H = gold["toy_H"].astype(np.int64)
        code = LDPCCode(n=H.shape[1], k=H.shape[1] - H.shape[0], H=H, max_iterations=T)
    else:
        code = codes.load_code("small_96_48", T)
    dec = (Neural2DMinSumDecoder if kind == "n2d" else Neural2DOffsetMinSumDecoder)(code, 2, T)
    beta = weights_dict(gold[f"{tag}_{kind}_beta_keys"], gold[f"{tag}_{kind}_beta_vals"])
    alpha = weights_dict(gold[f"{tag}_{kind}_alpha_keys"], gold[f"{tag}_{kind}_alpha_vals"])
    sd = {f"beta_weights.{k}": torch.tensor([v]) for k, v in beta.items()}
    sd.update({f"alpha_weights.{k}": torch.tensor([v]) for k, v in alpha.items()})
    dec.load_state_dict(sd)
    x = torch.from_numpy(gold[f"{tag}_llr"]).to(gpu_device).requires_grad_(True)
    _, post, iters = dec(x)
    np.testing.assert_array_equal(iters.cpu().numpy(), gold[f"{tag}_{kind}_iters"])
    codeword_loss_sum(post).backward()
    np.testing.assert_allclose(x.grad.cpu().numpy(), gold[f"{tag}_{kind}_grad_llr"], rtol=1e-4, atol=2e-6)
    # weight gradients: the reference's per-key values flattened to [T][slots] tables by the oracle's independent flattening
    tg = code.tanner_graph()
    og = oracle.OracleGraph(n=tg.n, check_ptr=tg.check_ptr, var_idx=tg.var_idx)
    gb = {k: (0.0 if p.grad is None else float(p.grad.item())) for k, p in dec.beta_weights.items()}
    ga = {k: (0.0 if p.grad is None else float(p.grad.item())) for k, p in dec.alpha_weights.items()}
    gbt, _, gat, _ = oracle.weight_tables(og, 2, T, gb, ga, beta_default=0.0, alpha_default=0.0)
    np.testing.assert_allclose(gbt, gold[f"{tag}_{kind}_grad_beta_table"], rtol=1e-4, atol=2e-6)
    np.testing.assert_allclose(gat, gold[f"{tag}_{kind}_grad_alpha_table"], rtol=1e-4, atol=2e-6)


def test_input_gradient_vs_oracle_on_the_1998_code(gpu_device):
    import codes
    import grad_oracle
    import oracle
    from neural_2d_decoder import Neural2DMinSumDecoder
    T, B = 4, 70
    code = codes.load_code("ira_1998_1512", max_iterations=T)
    dec = Neural2DMinSumDecoder(code, 2, T)
    rng = np.random.default_rng(5)
    with torch.no_grad():
        for p in dec.beta_weights.values():
            p.fill_(float(rng.uniform(0.5, 1.0)))
        for p in dec.alpha_weights.values():
            p.fill_(float(rng.uniform(0.8, 1.2)))
    snr = np.where(np.arange(B) % 2 == 0, 2.0, 6.5)
    s2 = 10.0 ** (-snr / 10.0)
    llr = (2.0 * (1.0 + np.sqrt(s2)[:, None] * rng.standard_normal((B, code.n))) / s2[:, None]).astype(np.float32)
    eng = dec._get_engine(gpu_device)
    x = torch.from_numpy(llr).to(gpu_device)
    res, saved = eng.decode_saving(x, early_stop=True)
    tg = code.tanner_graph()
    g = oracle.OracleGraph(n=code.n, check_ptr=tg.check_ptr, var_idx=tg.var_idx)
    bt_np, at_np = dec.weight_tables()
    lay = dec._sharing_layout()
    xt = torch.tensor(llr, requires_grad=True)
    post, _, iters = grad_oracle.forward(g, xt, torch.tensor(bt_np), lay.beta_slot, torch.tensor(at_np), lay.alpha_slot, T, True)
    agree = (iters.numpy() == res.iterations.cpu().numpy()) & \
        np.all(np.abs(post.detach().numpy() - res.posterior.cpu().numpy()) <= 1e-4 * np.maximum(1, np.abs(post.detach().numpy())), axis=1)
    assert agree.mean() > 0.95
    gpost = (rng.standard_normal((B, code.n)).astype(np.float32)) * agree[:, None]
    (post * torch.from_numpy(gpost)).sum().backward()
    _, _, _, gl = eng.backward(saved, x, res.iterations, torch.from_numpy(gpost).to(gpu_device), want_grad_llr=True)
    want = xt.grad.numpy()
    np.testing.assert_allclose(gl.cpu().numpy(), want, rtol=2e-3, atol=2e-4 * np.abs(want).max())


@pytest.mark.parametrize("wtype", [1, 2])
def test_gradients_on_wide_checks_and_high_degree_variables(gpu_device, wtype):
    """check degrees > 32 (the backward sweep re-reads rows instead of using its 32-edge masks), variable degrees > 8
    (two-pass variable backward), a degree-1 check, an empty check, an isolated and a degree-1 variable"""
    import grad_oracle
    import oracle
    from ldpc_decoder import LDPCCode
    from neural_2d_decoder import Neural2DMinSumDecoder
    rng = np.random.default_rng(12)
    H = (rng.random((12, 40)) < 0.9).astype(np.int64)
    H[10, :] = 0; H[10, 5] = 1
    H[11, :] = 0
    H[:, 39] = 0
    H[:, 38] = 0; H[0, 38] = 1
    T, B = 3, 20
    code = LDPCCode(n=40, k=28, H=H, max_iterations=T)
    og = oracle.OracleGraph(H)
    assert og.dc.max() > 32 and og.dv.max() > 8
    dec = Neural2DMinSumDecoder(code, wtype, T)
    with torch.no_grad():
        for p in dec.beta_weights.values():
            p.fill_(float(rng.uniform(0.05, 0.2)))       # dense graph: small factors keep the messages bounded
        for p in dec.alpha_weights.values():
            p.fill_(float(rng.uniform(0.8, 1.2)))
    llr = (rng.standard_normal((B, 40)) * 2 + 0.5).astype(np.float32)
    x = torch.from_numpy(llr).to(gpu_device).requires_grad_(True)
    _, post, iters = dec(x)
    codeword_loss_sum(post).backward()
    bt_np, at_np = dec.weight_tables()
    lay = dec._sharing_layout()
    gb, ga, opost, oit, gl = grad_oracle.table_grads(og, llr, bt_np, lay.beta_slot, at_np, lay.alpha_slot, T, want_llr=True)
    np.testing.assert_array_equal(iters.cpu().numpy(), oit)
    np.testing.assert_allclose(post.detach().cpu().numpy(), opost, rtol=1e-4, atol=1e-4)
    for t in range(T):
        for s_, suf in enumerate(lay.beta_suffix):
            if suf is not None:
                got = dec.beta_weights[f"iter_{t}_{suf}"].grad.item()
                assert abs(got - gb[t, s_]) <= 1e-5 + 1e-3 * abs(gb[t, s_]), ("beta", t, suf, got, gb[t, s_])
        for s_, suf in enumerate(lay.alpha_suffix):
            if suf is not None:
                got = dec.alpha_weights[f"iter_{t}_{suf}"].grad.item()
                assert abs(got - ga[t, s_]) <= 1e-5 + 1e-3 * abs(ga[t, s_]), ("alpha", t, suf, got, ga[t, s_])
    np.testing.assert_allclose(x.grad.cpu().numpy(), gl, rtol=1e-3, atol=1e-5)
