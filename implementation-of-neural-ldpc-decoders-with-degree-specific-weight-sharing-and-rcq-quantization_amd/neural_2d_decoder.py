"""
Drop-in for the reference module of the same name (neural_2d_decoder.py):
``Neural2DMinSumDecoder`` and ``Neural2DOffsetMinSumDecoder`` with node-degree-based
weight sharing, ``forward`` running on the MI355X engine.

Reference behaviour mirrored (file:line in /root/reference):
  constructor (code, weight_sharing_type=2, max_iterations=50), attributes
  ``beta_weights`` / ``alpha_weights`` (ParameterDict, reference key names, same init
  order and ``randn*0.1`` draws), ``check_node_degrees`` / ``variable_node_degrees``
                                                          neural_2d_decoder.py:27-82
  weight lookup incl. the constants of types without a weight
      NMS: beta 0.7 / alpha 1.0                            neural_2d_decoder.py:84-131
      OMS: beta 0.0 / alpha 0.0                            neural_2d_decoder.py:289-336
  forward(llr[n]) -> (int32[n], float32 posterior[n], int) neural_2d_decoder.py:133-225, 338-434
  invalid sharing type -> ValueError                       neural_2d_decoder.py:81-82

Extensions: ``llr`` of shape ``[B, n]`` -> ``(bits[B,n] int32, posterior[B,n] fp32,
iterations[B] int32)``; ``early_stop=False`` keyword.  With autograd enabled and parameters (or the
LLRs) requiring grad, the returned posterior carries a grad_fn back to beta / alpha (and the LLRs) exactly as
the reference's chain of torch operations does; the derivative is computed by the HIP backward sweeps behind
``torch.ops.ldpc.minsum_decode_train`` (torch_ops.py, autograd_bridge.py).  Under ``torch.no_grad()`` the
decode is the plain ``torch.ops.ldpc.decode``.
"""

from __future__ import annotations

import logging
from typing import Optional, Tuple

import numpy as np
import torch
import torch.nn as nn

from ldpc_decoder import LDPCCode, _as_batch
from weight_sharing import SharingLayout, init_parameter_dicts, unique_degrees

logger = logging.getLogger(__name__)


class _DegreeSharedDecoder(nn.Module):
    """Host logic common to the degree-shared decoders: parameter dicts with the
    reference's keys, flattening into engine tables, engine cache, I/O conventions."""

    _beta_default = 0.7
    _alpha_default = 1.0
    _alpha_is_oms = False          # True: alpha is the check-side offset of the offset form (engine slot oms_alpha)

    def _init_sharing(self, code: LDPCCode, weight_sharing_type: int, max_iterations: int, strict=True):
        self.code = code
        self.weight_sharing_type = weight_sharing_type
        self.max_iterations = max_iterations
        self.check_node_degrees, self.variable_node_degrees = unique_degrees(code)
        self.beta_weights, self.alpha_weights = init_parameter_dicts(
            weight_sharing_type, max_iterations, self.check_node_degrees, self.variable_node_degrees, strict)
        self._layout: Optional[SharingLayout] = None
        self._engine = None
        self._engine_key = None
        self._uploaded = None

    def __getstate__(self):
        # native handles never travel through pickle / deepcopy; they are rebuilt lazily
        state = self.__dict__.copy()
        state.update(_layout=None, _engine=None, _engine_key=None, _uploaded=None, _stamp=None)
        return state

    # ---- tables ---------------------------------------------------------------------
    def _sharing_layout(self) -> SharingLayout:
        g = self.code.tanner_graph()
        if self._layout is None or getattr(self, "_layout_graph", None) is not g:
            self._layout = SharingLayout(g, self.weight_sharing_type)
            self._layout_graph = g
        return self._layout

    def weight_tables(self) -> Tuple[np.ndarray, np.ndarray]:
        """Current parameters as (beta[T,Sb], alpha[T,Sa]) float32 tables."""
        return self._sharing_layout().tables(self.beta_weights, self.alpha_weights, int(self.max_iterations),
                                             self._beta_default, self._alpha_default)

    def _engine_kwargs(self, layout: SharingLayout, beta: np.ndarray, alpha: np.ndarray) -> dict:
        raise NotImplementedError

    def _tables_for_upload(self, beta, alpha):
        """(beta, alpha, oms_alpha) in the engine's meaning"""
        return beta, alpha, None

    def _get_engine(self, device):
        from engine import DecodeEngine, _require_gpu
        dev = _require_gpu(device)
        g = self.code.tanner_graph()
        layout = self._sharing_layout()
        key = (dev.index, id(g), int(self.max_iterations), self._extra_key())
        # Flattening the ParameterDicts costs ~40 us -- a third of a one-codeword call (the reference's own call shape) -- so it is
        # skipped while no parameter changed: every in-place update (optimizer step, load_state_dict, fill_) bumps the tensor's
        # version counter, a replaced tensor (`p.data = ...`, a new Parameter under the key) shows in its identity / storage.
        stamp = self._param_stamp()
        if self._engine is not None and self._engine_key == key and stamp == getattr(self, "_stamp", None):
            return self._engine
        beta, alpha = self.weight_tables()
        if self._engine is None or self._engine_key != key:
            self._engine = DecodeEngine(g, dtype=torch.float32, iters=int(self.max_iterations), device=dev,
                                        **self._engine_kwargs(layout, beta, alpha))
            self._engine_key = key
            self._uploaded = (beta, alpha)
        elif not (np.array_equal(self._uploaded[0], beta) and np.array_equal(self._uploaded[1], alpha)):
            self._engine.set_weights(*self._tables_for_upload(beta, alpha))
            self._uploaded = (beta, alpha)
        self._stamp = stamp
        return self._engine

    def _param_stamp(self):
        """cheap fingerprint of the parameter set: counts, identities, version counters and storage addresses"""
        ident = ver = ptr = cnt = 0
        for d in (self.beta_weights, self.alpha_weights):
            for p_ in d.values():
                cnt += 1
                ident += id(p_)
                ver += p_._version
                ptr += p_.data_ptr()
        return (cnt, ident, ver, ptr, int(self.max_iterations))

    def _extra_key(self):
        return ()

    # ---- I/O ------------------------------------------------------------------------
    def _decode(self, llr: torch.Tensor, early_stop: bool, device=None):
        if not isinstance(llr, torch.Tensor):
            raise TypeError("llr must be a torch.Tensor")   # the reference fails on numpy input too (llr.device)
        _, x, single = _as_batch(llr, self.code.n)
        eng = self._get_engine(x.device if x.is_cuda else device)
        if not x.is_cuda and x.shape[0] <= eng.HOST_BATCH_MAX:       # the reference's call shape: torch.ops.ldpc.decode_host
            return eng.decode_host_op(x.detach().to(torch.float32), early_stop=early_stop), single, llr.device
        xd = x.detach().to(device=eng.device, dtype=torch.float32)
        res = eng.decode_op(xd, early_stop=early_stop)            # torch.ops.ldpc.decode
        return res, single, llr.device


    def _decode_with_grad(self, llr, early_stop, device):
        """the reference's posterior carries a grad_fn back to beta/alpha (neural_2d_decoder.py:189-209);
        here the HIP backward sweeps provide it (autograd_bridge.py)"""
        import autograd_bridge as ab
        if not isinstance(llr, torch.Tensor):
            raise TypeError("llr must be a torch.Tensor")
        _, x, single = _as_batch(llr, self.code.n)
        eng = self._get_engine(x.device if x.is_cuda else device)       # uploads the current parameter values
        if not ab.saved_state_fits(eng, x.shape[0]):
            return None
        bt, at = self._sharing_layout().tables_torch(self.beta_weights, self.alpha_weights, int(self.max_iterations),
                                                     self._beta_default, self._alpha_default)
        post, bits, iters = ab.decode_train(bt, at, eng, x, bool(early_stop), self._alpha_is_oms)
        out_dev = llr.device
        if single:
            return bits[0].to(out_dev), post[0].to(out_dev), int(iters[0].item())
        return bits.to(out_dev), post.to(out_dev), iters.to(out_dev)


class Neural2DMinSumDecoder(_DegreeSharedDecoder):
    """
    Neural 2D MinSum decoder with node-degree-based weight sharing

    Weight sharing types:
    - Type 1: one beta per (check node degree, variable node degree) pair
    - Type 2: beta per check node degree, alpha per variable node degree
    - Type 3: beta per check node degree only
    - Type 4: alpha per variable node degree only
    """

    def __init__(self, code: LDPCCode, weight_sharing_type: int = 2, max_iterations: int = 50):
        super().__init__()
        self._init_sharing(code, weight_sharing_type, max_iterations)
        logger.info(f"Initialized N-2D-NMS decoder (Type {weight_sharing_type}) with "
                    f"{len(self.beta_weights)} beta weights and {len(self.alpha_weights)} alpha weights")

    def _engine_kwargs(self, layout, beta, alpha):
        import _native as nat
        return dict(c2v_form=nat.C2V_NMS, beta=beta, beta_slot=layout.beta_slot,
                    alpha=alpha, alpha_slot=layout.alpha_slot)

    def forward(self, llr: torch.Tensor, early_stop: bool = True, device=None):
        """
        Args:
            llr: log-likelihood ratios from the channel, ``[n]`` or ``[B, n]``

        Returns:
            decoded_bits (int32), posterior (float32), iterations (int, or int32[B])
        """
        import autograd_bridge as ab
        if ab.wants_grad(self, llr):
            out = self._decode_with_grad(llr, early_stop, device)
            if out is not None:
                return out
        res, single, out_dev = self._decode(llr, early_stop, device)
        if single:
            return res.bits[0].to(out_dev), res.posterior[0].to(out_dev), int(res.iterations[0].item())
        return res.bits.to(out_dev), res.posterior.to(out_dev), res.iterations.to(out_dev)


class Neural2DOffsetMinSumDecoder(_DegreeSharedDecoder):
    """
    Neural 2D Offset MinSum decoder: C2V = prod(signs) * (relu(min - beta) - alpha),
    plain V2C sums (neural_2d_decoder.py:389-412).
    """

    _beta_default = 0.0
    _alpha_default = 0.0
    _alpha_is_oms = True

    def __init__(self, code: LDPCCode, weight_sharing_type: int = 2, max_iterations: int = 50):
        super().__init__()
        self._init_sharing(code, weight_sharing_type, max_iterations)
        logger.info(f"Initialized N-2D-OMS decoder (Type {weight_sharing_type}) with "
                    f"{len(self.beta_weights)} beta weights and {len(self.alpha_weights)} alpha weights")

    def _engine_kwargs(self, layout, beta, alpha):
        import _native as nat
        g = self.code.tanner_graph()
        rows = beta.shape[0]
        return dict(c2v_form=nat.C2V_OMS, beta=beta, beta_slot=layout.beta_slot,
                    alpha=np.ones((rows, 1), np.float32), alpha_slot=np.zeros(g.n, np.int32),
                    oms_alpha=alpha, oms_alpha_slot=layout.alpha_edge_slot)

    def _tables_for_upload(self, beta, alpha):
        return beta, None, alpha

    def forward(self, llr: torch.Tensor, early_stop: bool = True, device=None):
        import autograd_bridge as ab
        if ab.wants_grad(self, llr):       # relu / offset are differentiable in the reference too (:396-401)
            out = self._decode_with_grad(llr, early_stop, device)
            if out is not None:
                return out
        res, single, out_dev = self._decode(llr, early_stop, device)
        if single:
            return res.bits[0].to(out_dev), res.posterior[0].to(out_dev), int(res.iterations[0].item())
        return res.bits.to(out_dev), res.posterior.to(out_dev), res.iterations.to(out_dev)
