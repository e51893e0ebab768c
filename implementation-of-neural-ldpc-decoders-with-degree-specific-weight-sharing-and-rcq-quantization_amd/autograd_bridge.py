"""
torch.autograd bridge of the gradient path: makes ``loss.backward()`` work on the posterior the
normalised min-sum decoders return, the way it does in the reference where ``forward`` is a chain of
differentiable torch operations on ``nn.Parameter``s (neural_2d_decoder.py:133-225; used by
training_framework.py:127-134 and :322-324).

The weight tables are assembled from the ``ParameterDict`` entries with differentiable torch
operations (``cat`` / ``index_put``), so autograd itself routes the table gradients the HIP backward
sweeps return (include/ldpc_hip.h: ldpc_decode_saving / ldpc_backward) to the individual parameters.
Parameters that do not influence the returned posterior get a zero gradient (the reference leaves
``.grad`` at None for them).

No CPU fallback: without the native engine ``forward`` raises like every other decode call.
"""

from __future__ import annotations

import logging
import os

import numpy as np
import torch

logger = logging.getLogger(__name__)

# forward state kept for backward is (2T-1)*E*4 bytes per codeword; above this many bytes the decoders
# return a posterior without grad_fn (and say so once) instead of exhausting HBM
MAX_SAVED_BYTES = int(os.environ.get("LDPC_TRAIN_MAX_SAVED_BYTES", str(8 << 30)))
_warned = False
RECYCLE_SAVED = True        # release the saved messages to the engine's spare slot after backward (see backward())


def wants_grad(module: torch.nn.Module, llr=None) -> bool:
    if not torch.is_grad_enabled():
        return False
    if isinstance(llr, torch.Tensor) and llr.requires_grad:
        return True
    return any(p.requires_grad for p in module.parameters())


def saved_state_fits(engine, batch: int) -> bool:
    global _warned
    need = engine.train_saved_bytes(batch)
    if need <= MAX_SAVED_BYTES:
        return True
    if not _warned:
        _warned = True
        logger.warning("decode of %d codewords under autograd would keep %.1f GiB of messages for backward "
                       "(limit LDPC_TRAIN_MAX_SAVED_BYTES = %.1f GiB): returning a posterior without grad_fn; "
                       "wrap inference in torch.no_grad() or train on smaller batches",
                       batch, need / 2**30, MAX_SAVED_BYTES / 2**30)
    return False


class MinSumDecodeFn(torch.autograd.Function):
    """(beta_table [T, Sb], alpha_table [T, Sa]) -> posterior [B, n]; bits and iterations ride along.
    `alpha_is_oms`: the alpha table is the check-side offset of the offset decoders (engine slot oms_alpha),
    otherwise the variable-side multiplier (engine slot alpha)."""

    @staticmethod
    def forward(ctx, beta_table, alpha_table, engine, xd, early_stop, alpha_is_oms=False, llr_in=None):
        """xd: the LLRs as the engine takes them (detached, fp32, on the GPU); llr_in: the caller's own tensor when it
        requires grad (a trainable front end feeding the decoder) -- its gradient is d loss/d llr from the same sweep"""
        ctx.llr_meta = None
        if llr_in is not None and llr_in.requires_grad:
            ctx.llr_meta = (llr_in.device, llr_in.dtype, tuple(llr_in.shape))
        res, saved = engine.decode_saving(xd, early_stop=early_stop)
        ctx.engine, ctx.saved, ctx.xd, ctx.iters = engine, saved, xd, res.iterations
        ctx.alpha_is_oms = bool(alpha_is_oms)
        ctx.tables = (beta_table.detach().to("cpu", torch.float32).numpy().copy(),
                      alpha_table.detach().to("cpu", torch.float32).numpy().copy())
        ctx.meta = (beta_table.device, beta_table.dtype, alpha_table.device, alpha_table.dtype)
        ctx.mark_non_differentiable(res.bits, res.iterations)
        return res.posterior, res.bits, res.iterations

    @staticmethod
    def _upload(eng, beta, alpha, alpha_is_oms):
        if alpha_is_oms:
            eng.set_weights(beta, None, alpha if eng.current_tables()[2] is not None else None)
        else:
            eng.set_weights(beta, alpha)

    @staticmethod
    def backward(ctx, g_post, _g_bits, _g_iters):
        eng = ctx.engine
        if ctx.saved is None:
            raise RuntimeError("the saved messages of this decode were released by its first backward; for "
                               "backward(retain_graph=True) set autograd_bridge.RECYCLE_SAVED = False")
        held = eng.current_tables()
        held_alpha = held[2] if ctx.alpha_is_oms else held[1]
        same = np.array_equal(held[0], ctx.tables[0]) and (held_alpha is None or np.array_equal(held_alpha, ctx.tables[1]))
        if not same:                       # the weights moved on since this forward: put its tables back for the sweep
            MinSumDecodeFn._upload(eng, ctx.tables[0], ctx.tables[1], ctx.alpha_is_oms)
        gl = None
        try:
            if ctx.llr_meta is not None:
                gb, ga, goa, gl = eng.backward(ctx.saved, ctx.xd, ctx.iters, g_post, want_grad_llr=True)
            else:
                gb, ga, goa = eng.backward(ctx.saved, ctx.xd, ctx.iters, g_post)
        finally:
            if not same:
                MinSumDecodeFn._upload(eng, held[0], held_alpha, ctx.alpha_is_oms)
        if RECYCLE_SAVED:                  # GBs of messages: hand the buffer to the next forward instead of the allocator
            eng.recycle_saved(ctx.saved)
            ctx.saved = None
        if ctx.alpha_is_oms:
            ga = goa if goa is not None else torch.zeros(ctx.tables[1].shape, dtype=torch.float32, device=gb.device)
        bdev, bdt, adev, adt = ctx.meta
        if gl is not None:
            ldev, ldt, lshape = ctx.llr_meta
            gl = gl.to(device=ldev, dtype=ldt).reshape(lshape)
        return gb.to(device=bdev, dtype=bdt), ga.to(device=adev, dtype=adt), None, None, None, None, gl


def table_from_params(params, where, shape, default: float) -> torch.Tensor:
    """Differentiable [rows, cols] table: `default` everywhere, params[k] (shape [1]) at where[k] = (row, col)"""
    out = torch.full(shape, float(default), dtype=torch.float32)
    if not params:
        return out
    vals = torch.cat([p.reshape(-1)[:1] for p in params]).to("cpu", torch.float32)
    rr = torch.tensor([w[0] for w in where], dtype=torch.long)
    cc = torch.tensor([w[1] for w in where], dtype=torch.long)
    return out.index_put((rr, cc), vals)
