"""
torch.autograd bridge of the gradient path: makes ``loss.backward()`` work on the posterior the
normalised min-sum decoders return, the way it does in the reference where ``forward`` is a chain of
differentiable torch operations on ``nn.Parameter``s (neural_2d_decoder.py:133-225; used by
training_framework.py:127-134 and :322-324).

The weight tables are assembled from the ``ParameterDict`` entries with differentiable torch
operations (``cat`` / ``index_put``) and handed to the registered operator
``torch.ops.ldpc.minsum_decode_train`` (torch_ops.py), so autograd itself routes the table gradients
the HIP backward sweeps return (include/ldpc_hip.h: ldpc_decode_saving / ldpc_backward) to the
individual parameters.
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


def decode_train(beta_table: torch.Tensor, alpha_table: torch.Tensor, engine, x: torch.Tensor, early_stop: bool,
                 alpha_is_oms: bool = False):
    """(beta_table [T, Sb], alpha_table [T, Sa], llr x [B, n]) -> (posterior [B, n], bits, iterations) through the
    registered operator ``torch.ops.ldpc.minsum_decode_train`` (torch_ops.py): its autograd formula runs the HIP
    backward sweeps (``torch.ops.ldpc.minsum_backward``) and returns d loss/d beta, d loss/d alpha and -- when `x`
    requires grad (a trainable front end feeding the decoder) -- d loss/d llr from the same sweep.
    `alpha_is_oms`: the alpha table is the check-side offset of the offset decoders (engine slot oms_alpha),
    otherwise the variable-side multiplier (engine slot alpha)."""
    import torch_ops
    xd = x.to(device=engine.device, dtype=torch.float32)       # differentiable: the LLR gradient flows back through it
    if not x.requires_grad:
        xd = xd.detach()
    post, bits, iters, _saved = torch.ops.ldpc.minsum_decode_train(xd.contiguous(), beta_table, alpha_table,
                                                                  torch_ops.engine_handle(engine), bool(early_stop),
                                                                  bool(alpha_is_oms))
    return post, bits, iters


def table_from_params(params, where, shape, default: float) -> torch.Tensor:
    """Differentiable [rows, cols] table: `default` everywhere, params[k] (shape [1]) at where[k] = (row, col)"""
    out = torch.full(shape, float(default), dtype=torch.float32)
    if not params:
        return out
    vals = torch.cat([p.reshape(-1)[:1] for p in params]).to("cpu", torch.float32)
    rr = torch.tensor([w[0] for w in where], dtype=torch.long)
    cc = torch.tensor([w[1] for w in where], dtype=torch.long)
    return out.index_put((rr, cc), vals)
