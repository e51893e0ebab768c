"""
Gradient oracle: a batched, differentiable torch (CPU) restatement of the normalised min-sum forward pass,
so that torch autograd yields d loss / d beta-table and d loss / d alpha-table.

TEST INFRASTRUCTURE, NOT PRODUCT CODE (same rule as oracle.py): importable only from tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg.

What it restates: Neural2DMinSumDecoder.forward (neural_2d_decoder.py:133-225) and, with a per-edge table,
NeuralMinSumDecoder.forward -- the operations autograd differentiates there:
  magnitudes = |incoming|, signs = sign(incoming)                      (:170-171)
  min_val = magnitudes[argmin], min2_val = min(magnitudes with the arg-min set to inf); = min_val at dc == 1  (:174-182)
      (gradient of min_val: the first arg-min edge; of min2_val: split evenly among the edges tied for it)
  c2v = beta * (min2_val | min_val) * prod(other signs)                (:186-191)
  v2c = llr + alpha * sum(other c2v)                                   (:203)
  posterior = llr + sum(c2v)                                           (:206-209), returned at the first zero syndrome
The per-check / per-variable Python loops become padded gathers; sums are torch.sum over the padded axis (not
the reference's exact association order -- gradients are compared with a tolerance).

Parity pinning: oracle/make_golden.py runs the REAL reference under autograd (loss.backward() on
binary_cross_entropy_with_logits(-posterior, 0), training_framework.py:101) and checks this file against it;
the resulting fixtures are tests/golden/grad_*.npz.  Status: PINNED.
"""

from __future__ import annotations

import numpy as np
import torch


def _padded_neighbourhoods(g):
    """check -> its CSR edges [m, max_dc]; variable -> its edges in ascending check order [n, max_dv]; E = padding"""
    max_dc = int(g.dc.max()) if g.m else 1
    max_dv = int(g.dv.max()) if g.n else 1
    ce = np.full((g.m, max(max_dc, 1)), g.E, dtype=np.int64)
    for i in range(g.m):
        ce[i, :g.dc[i]] = np.arange(g.check_ptr[i], g.check_ptr[i + 1])
    ve = np.full((g.n, max(max_dv, 1)), g.E, dtype=np.int64)
    for j in range(g.n):
        ve[j, :g.dv[j]] = g.csc_edge[g.var_ptr[j]:g.var_ptr[j + 1]]
    return torch.from_numpy(ce), torch.from_numpy(ve)


def forward(g, llr, beta_table, beta_slot, alpha_table, alpha_slot, T, early_stop=True, dtype=torch.float32,
            offset=False):
    """llr [B, n]; beta_table [T, Sb], alpha_table [T, Sa] torch tensors (may require grad).
    offset=False: normalised form, alpha_slot per VARIABLE (multiplies the leave-one-out sums).
    offset=True : C2V = prod(signs) * (relu(minval - beta) - alpha), alpha_slot per EDGE, plain V2C sums
                  (Neural2DOffsetMinSumDecoder, neural_2d_decoder.py:389-412; NeuralOffsetMinSumDecoder with alpha = 0).
    Returns (posterior [B, n] with grad_fn, bits int32 [B, n], iterations int64 [B])."""
    llr = llr if isinstance(llr, torch.Tensor) else torch.as_tensor(llr, dtype=dtype)
    B, n = llr.shape
    E = g.E
    ce, ve = _padded_neighbourhoods(g)
    cmask = ce < E                                        # [m, max_dc]
    vmask = ve < E
    var_of_edge = torch.from_numpy(g.var_idx.astype(np.int64))
    bslot = torch.from_numpy(np.asarray(beta_slot, dtype=np.int64))
    aslot_e = torch.from_numpy(np.asarray(alpha_slot, dtype=np.int64))
    if not offset:
        aslot_e = aslot_e[var_of_edge]                                                     # alpha of the edge's variable
    dc = torch.from_numpy(g.dc.astype(np.int64))
    chk_of_edge = torch.from_numpy(g.rows.astype(np.int64))
    pos_of_edge = torch.arange(E) - torch.from_numpy(g.check_ptr.astype(np.int64))[chk_of_edge]
    kpos = np.empty(E, dtype=np.int64)                    # position of each CSR edge inside its variable's list
    kpos[g.csc_edge] = np.arange(E) - np.repeat(g.var_ptr[:-1].astype(np.int64), g.dv)
    kpos_of_edge = torch.from_numpy(kpos)
    max_dc, max_dv = ce.shape[1], ve.shape[1]
    eye_c = torch.eye(max_dc, dtype=torch.bool)
    eye_v = torch.eye(max_dv, dtype=torch.bool)
    inf = torch.tensor(float("inf"), dtype=dtype)

    v2c = llr[:, var_of_edge]                              # [B, E]   "initialize with channel LLRs" (:153-157)
    H_rows = torch.from_numpy(g.rows.astype(np.int64))
    final_post = llr.clone()
    iters = torch.full((B,), T, dtype=torch.int64)
    done = torch.zeros(B, dtype=torch.bool)
    post = llr
    for t in range(T):
        pad = torch.cat([v2c, torch.zeros(B, 1, dtype=dtype)], dim=1)            # column E = padding
        inc = pad[:, ce]                                                           # [B, m, max_dc]
        mags = torch.where(cmask, inc.abs(), inf)
        signs = torch.where(cmask, torch.sign(inc), torch.ones((), dtype=dtype))
        m1, i1 = mags.min(dim=-1, keepdim=True)
        mags2 = mags.scatter(-1, i1, float("inf"))
        # torch.min(temp_mags) in the reference is a FULL reduction (neural_2d_decoder.py:179): its backward splits the
        # gradient evenly among all elements that hold the minimum -- amin has exactly that rule (min(dim) would send
        # it to one index); pinned by tests/golden/grad_ties.npz
        m2 = mags2.amin(dim=-1, keepdim=True)
        m2 = torch.where((dc == 1).view(1, -1, 1), m1, m2)                         # min2_val = min_val (:181-182)
        pos = torch.arange(max_dc).view(1, 1, -1)
        minval = torch.where(pos == i1, m2, m1)                                    # [B, m, max_dc]
        # product of the OTHER signs: replace the own sign by 1, multiply along the check
        s_others = torch.where(eye_c.view(1, 1, max_dc, max_dc), torch.ones((), dtype=dtype), signs.unsqueeze(-2))
        prod_others = s_others.prod(dim=-1)                                        # [B, m, max_dc]
        beta_e = beta_table[t][bslot]                                              # [E]
        alpha_e = alpha_table[t][aslot_e]                                          # [E]
        if offset:
            raw = minval[:, chk_of_edge, pos_of_edge]                              # back to CSR edge order
            c2v = prod_others[:, chk_of_edge, pos_of_edge] * (torch.relu(raw - beta_e.view(1, -1)) - alpha_e.view(1, -1))
        else:
            c2v = (minval * prod_others)[:, chk_of_edge, pos_of_edge]
            c2v = beta_e.view(1, -1) * c2v
        c2v_pad = torch.cat([c2v, torch.zeros(B, 1, dtype=dtype)], dim=1)
        at_var = c2v_pad[:, ve]                                                    # [B, n, max_dv]
        post = llr + at_var.sum(dim=-1)                                            # no alpha (:206-209)
        bits = (post < 0)
        syn = torch.zeros(B, g.m, dtype=torch.int64).index_add_(1, H_rows, bits[:, var_of_edge].to(torch.int64)) % 2
        ok = syn.sum(dim=1) == 0
        newly = ok & ~done if early_stop else torch.zeros_like(done)
        final_post = torch.where(newly.view(-1, 1), post, final_post)
        iters = torch.where(newly, torch.full_like(iters, t + 1), iters)
        done = done | newly
        # variable update (:203): leave-one-out sums
        others = torch.where(eye_v.view(1, 1, max_dv, max_dv), torch.zeros((), dtype=dtype), at_var.unsqueeze(-2)).sum(dim=-1)
        new_v2c = others[:, var_of_edge, kpos_of_edge]
        v2c = llr[:, var_of_edge] + (new_v2c if offset else alpha_e.view(1, -1) * new_v2c)
    final_post = torch.where(done.view(-1, 1), final_post, post)                   # not converged: last posterior
    return final_post, (final_post < 0).to(torch.int32), iters


def bce_loss_sum(posterior, targets=None):
    """sum over codewords of binary_cross_entropy_with_logits(-posterior[b], target[b]) (mean over the n bits):
    the per-codeword loss of training_framework.py:101"""
    import torch.nn.functional as F
    if targets is None:
        targets = torch.zeros_like(posterior)
    return F.binary_cross_entropy_with_logits(-posterior, targets.to(posterior.dtype), reduction="none").mean(dim=1).sum()


def table_grads(g, llr, beta_table, beta_slot, alpha_table, alpha_slot, T, early_stop=True, targets=None,
                dtype=torch.float32, offset=False, want_llr=False):
    """-> (grad beta [T, Sb], grad alpha [T, Sa], posterior, iterations[, grad llr [B, n]]) for bce_loss_sum"""
    bt = torch.tensor(np.asarray(beta_table), dtype=dtype, requires_grad=True)
    at = torch.tensor(np.asarray(alpha_table), dtype=dtype, requires_grad=True)
    x = torch.tensor(np.asarray(llr), dtype=dtype, requires_grad=want_llr)
    post, bits, iters = forward(g, x, bt, beta_slot, at, alpha_slot, T, early_stop, dtype, offset)
    loss = bce_loss_sum(post, None if targets is None else torch.as_tensor(targets))
    wrt = (bt, at, x) if want_llr else (bt, at)
    grads = torch.autograd.grad(loss, wrt, allow_unused=True)
    gb = torch.zeros_like(bt) if grads[0] is None else grads[0]
    ga = torch.zeros_like(at) if grads[1] is None else grads[1]
    out = (gb.detach().numpy(), ga.detach().numpy(), post.detach().numpy(), iters.numpy())
    if want_llr:
        out += ((torch.zeros_like(x) if grads[2] is None else grads[2]).detach().numpy(),)
    return out
