"""
PyTorch operator registration of the decode path (BASELINE.json north_star: "Python host code calling
hand-written CDNA4 HIP kernels through PyTorch-ROCm custom ops").

The reference has no operator registry -- its decoders are Python loops behind
``decoder.decode(llr)`` / ``decoder(llr)`` (ldpc_decoder.py:63, neural_2d_decoder.py:133,
rcq_decoder.py:169, :495).  Here those methods end in ``torch.ops.ldpc.*`` calls, which in turn
enter the C ABI of include/ldpc_hip.h (ldpc_decode / ldpc_decode_saving / ldpc_backward):

  ldpc::decode(Tensor llr, int engine, bool early_stop, bool want_posterior, bool want_packed)
        -> (Tensor bits, Tensor posterior, Tensor iterations, Tensor success, Tensor packed_bits)
     every decoder family (the engine handle carries the descriptor: C2V rule, tables, quantisers);
     outputs that were not asked for come back as empty tensors.  No autograd (inference).
  ldpc::decode_host(Tensor llr_cpu, int engine, bool early_stop, bool want_posterior)
        -> (Tensor bits, Tensor posterior, Tensor iterations, Tensor success)      all on the CPU
     the reference's own call shape (one vector, or a batch of at most 64, in host memory): one staged async copy
     each way around the same ldpc_decode.
  ldpc::minsum_decode_train(Tensor llr, Tensor beta, Tensor alpha, int engine, bool early_stop, bool alpha_is_oms)
        -> (Tensor posterior, Tensor bits, Tensor iterations, Tensor saved)
     the same decode keeping every iteration's messages (`saved`) for the backward sweeps; differentiable in
     beta [T, Sb], alpha [T, Sa] and llr through register_autograd, whose backward is
  ldpc::minsum_backward(Tensor saved, Tensor llr, Tensor iterations, Tensor grad_posterior, Tensor beta,
                        Tensor alpha, int engine, bool alpha_is_oms, bool want_grad_llr)
        -> (Tensor grad_beta, Tensor grad_alpha, Tensor grad_llr)

``engine`` is an integer handle of a live ``engine.DecodeEngine`` (``engine_handle(eng)``): operator schemas
carry tensors and scalars, and the native decoder handle is neither.  There is no CPU implementation: the
ops exist for ROCm tensors only and fail loudly otherwise (no fallback).
"""

from __future__ import annotations

import threading
import weakref
from typing import Tuple

import numpy as np
import torch
from torch import Tensor

_engines = weakref.WeakValueDictionary()
_lock = threading.Lock()
_next = [1]


def engine_handle(eng) -> int:
    """integer handle under which `eng` is known to the ops (stable for the engine's lifetime)"""
    h = getattr(eng, "_op_handle", None)
    if h is None:
        with _lock:
            h = _next[0]
            _next[0] += 1
            _engines[h] = eng
        eng._op_handle = h
    return h


def _engine(handle: int):
    eng = _engines.get(int(handle))
    if eng is None:
        raise RuntimeError(f"ldpc ops: no live decode engine with handle {handle}")
    return eng


def _empty(dev, dtype=torch.uint8):
    return torch.empty((0,), dtype=dtype, device=dev)


# ------------------------------------------------------------------------------------------ inference
@torch.library.custom_op("ldpc::decode", mutates_args=())
def decode(llr: Tensor, engine: int, early_stop: bool, want_posterior: bool,
           want_packed: bool) -> Tuple[Tensor, Tensor, Tensor, Tensor, Tensor]:
    eng = _engine(engine)
    res = eng.decode(llr, early_stop=early_stop, want_bits=True, want_posterior=want_posterior, want_packed=want_packed)
    dev = llr.device
    return (res.bits, res.posterior if want_posterior else _empty(dev, llr.dtype), res.iterations, res.success,
            res.packed_bits if want_packed else _empty(dev))


@decode.register_fake
def _(llr, engine, early_stop, want_posterior, want_packed):
    B, n = llr.shape
    dev = llr.device
    return (torch.empty((B, n), dtype=torch.int32, device=dev),
            torch.empty((B, n), dtype=llr.dtype, device=dev) if want_posterior else torch.empty((0,), dtype=llr.dtype, device=dev),
            torch.empty((B,), dtype=torch.int32, device=dev), torch.empty((B,), dtype=torch.bool, device=dev),
            torch.empty((B, (n + 7) // 8), dtype=torch.uint8, device=dev) if want_packed else torch.empty((0,), dtype=torch.uint8, device=dev))


@torch.library.custom_op("ldpc::decode_host", mutates_args=())
def decode_host(llr: Tensor, engine: int, early_stop: bool, want_posterior: bool) -> Tuple[Tensor, Tensor, Tensor, Tensor]:
    """the reference's call shape -- a small batch in HOST memory, results back in host memory -- through one staged
    copy each way (engine.DecodeEngine.decode_host); still the GPU engine, there is no CPU arithmetic"""
    eng = _engine(engine)
    bits, post, iters, succ = eng.decode_host(llr.to(eng.dtype), early_stop=early_stop, want_posterior=want_posterior)
    return bits, post if want_posterior else torch.empty((0,), dtype=eng.dtype), iters, succ


@decode_host.register_fake
def _(llr, engine, early_stop, want_posterior):
    B, n = llr.shape
    dt = _engine(engine).dtype
    return (torch.empty((B, n), dtype=torch.int32), torch.empty((B, n), dtype=dt) if want_posterior else torch.empty((0,), dtype=dt),
            torch.empty((B,), dtype=torch.int32), torch.empty((B,), dtype=torch.bool))


# ------------------------------------------------------------------------------------------ training path
def _np_table(t: Tensor) -> np.ndarray:
    return t.detach().to("cpu", torch.float32).numpy().copy()


def _with_tables(eng, beta: np.ndarray, alpha: np.ndarray, alpha_is_oms: bool):
    """make the device tables equal (beta, alpha) for the duration of a call; returns a restore() closure.
    (A backward may run after the parameters moved on: the sweep must see the tables its forward used.)"""
    held = eng.current_tables()
    held_alpha = held[2] if alpha_is_oms else held[1]
    same = np.array_equal(held[0], beta) and (held_alpha is None or np.array_equal(held_alpha, alpha))

    def upload(b, a):
        if alpha_is_oms:
            eng.set_weights(b, None, a if eng.current_tables()[2] is not None else None)
        else:
            eng.set_weights(b, a)

    if same:
        return lambda: None
    upload(beta, alpha)
    return lambda: upload(held[0], held_alpha)


@torch.library.custom_op("ldpc::minsum_decode_train", mutates_args=())
def minsum_decode_train(llr: Tensor, beta: Tensor, alpha: Tensor, engine: int, early_stop: bool,
                        alpha_is_oms: bool) -> Tuple[Tensor, Tensor, Tensor, Tensor]:
    eng = _engine(engine)
    restore = _with_tables(eng, _np_table(beta), _np_table(alpha), alpha_is_oms)
    try:
        res, saved = eng.decode_saving(llr.detach(), early_stop=early_stop)
    finally:
        restore()
    return res.posterior, res.bits, res.iterations, saved


@minsum_decode_train.register_fake
def _(llr, beta, alpha, engine, early_stop, alpha_is_oms):
    B, n = llr.shape
    dev = llr.device
    nbytes = _engine(engine).train_saved_bytes(int(B)) if not isinstance(B, torch.SymInt) else torch.library.get_ctx().new_dynamic_size()
    return (torch.empty((B, n), dtype=torch.float32, device=dev), torch.empty((B, n), dtype=torch.int32, device=dev),
            torch.empty((B,), dtype=torch.int32, device=dev), torch.empty((nbytes,), dtype=torch.uint8, device=dev))


@torch.library.custom_op("ldpc::minsum_backward", mutates_args=())
def minsum_backward(saved: Tensor, llr: Tensor, iterations: Tensor, grad_posterior: Tensor, beta: Tensor, alpha: Tensor,
                    engine: int, alpha_is_oms: bool, want_grad_llr: bool) -> Tuple[Tensor, Tensor, Tensor]:
    eng = _engine(engine)
    restore = _with_tables(eng, _np_table(beta), _np_table(alpha), alpha_is_oms)
    try:
        if want_grad_llr:
            gb, ga, goa, gl = eng.backward(saved, llr, iterations, grad_posterior, want_grad_llr=True)
        else:
            gb, ga, goa = eng.backward(saved, llr, iterations, grad_posterior)
            gl = torch.empty((0,), dtype=torch.float32, device=llr.device)
    finally:
        restore()
    if alpha_is_oms:
        ga = goa if goa is not None else torch.zeros(tuple(alpha.shape), dtype=torch.float32, device=gb.device)
    return gb.to(device=beta.device, dtype=beta.dtype), ga.to(device=alpha.device, dtype=alpha.dtype), gl


@minsum_backward.register_fake
def _(saved, llr, iterations, grad_posterior, beta, alpha, engine, alpha_is_oms, want_grad_llr):
    return (torch.empty_like(beta), torch.empty_like(alpha),
            torch.empty_like(llr, dtype=torch.float32) if want_grad_llr else torch.empty((0,), dtype=torch.float32, device=llr.device))


def _train_setup(ctx, inputs, output):
    llr, beta, alpha, engine, early_stop, alpha_is_oms = inputs
    _post, _bits, iters, saved = output
    ctx.engine, ctx.alpha_is_oms = engine, alpha_is_oms
    # the handle table holds engines weakly; the autograd node keeps ITS engine alive until the graph is freed (the
    # decoder may be rebuilt or go out of scope between forward and backward -- the reference's graph is self-contained)
    ctx._engine_obj = _engine(engine)
    ctx.save_for_backward(saved, llr, iters, beta, alpha)
    ctx.set_materialize_grads(False)


def _train_backward(ctx, g_post, _g_bits, _g_iters, _g_saved):
    saved, llr, iters, beta, alpha = ctx.saved_tensors
    if g_post is None:
        return None, None, None, None, None, None
    want_llr = bool(ctx.needs_input_grad[0])
    gb, ga, gl = torch.ops.ldpc.minsum_backward(saved, llr, iters, g_post.contiguous(), beta, alpha, ctx.engine,
                                                ctx.alpha_is_oms, want_llr)
    return (gl if want_llr else None), gb, ga, None, None, None


minsum_decode_train.register_autograd(_train_backward, setup_context=_train_setup)
