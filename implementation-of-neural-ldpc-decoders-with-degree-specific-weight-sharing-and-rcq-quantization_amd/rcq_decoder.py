"""
Drop-in for the reference module of the same name (rcq_decoder.py):
``NonUniformQuantizer``, ``RCQMinSumDecoder``, ``WeightedRCQDecoder`` with the
quantise / reconstruct / min-sum loop running on the MI355X engine (C2V messages
live in HBM as the 1-byte quantiser codes, reconstruction is a LUT in LDS).

Reference behaviour mirrored (file:line in /root/reference):
  NonUniformQuantizer(bc, C, gamma): thresholds C*(j/(2^(bc-1)-1))^gamma as Python
      floats; quantize -> int64 codes sign_bit*2^(bc-1)+level; dequantize -> float32
                                                               rcq_decoder.py:22-121
  RCQMinSumDecoder(code, bc, bv, quantizer_params, max_iterations=50, layered=False)
      .decode(llr[n]) -> (int32[n], bool, int); ``bv`` stored and unused as in the
      reference; quantiser schedule by thirds                  rcq_decoder.py:123-279
  WeightedRCQDecoder(code, bc, bv, quantizer_params, weight_sharing_type=2,
      max_iterations=50, layered=False)(llr[n]) -> (int32[n], float32[n], int);
      ``layered`` stored and ignored by forward as in the reference
                                                               rcq_decoder.py:352-597

``RCQMinSumDecoder(layered=True)`` runs the reference's layered schedule exactly as the
reference executes it (rcq_decoder.py:281-350): checks in order on running
posteriors; because the reference re-creates its message matrix for every check, the
"previous message" it subtracts is always zero, and so it is here (bug-compatible by decision --
parity is the contract; a single-check code, the one case where the subtraction would be real,
is rejected).

Extensions: batched ``[B, n]`` input and ``early_stop=False`` as in the other decoders;
``RCQMinSumDecoder(layered="paper")`` runs the layered schedule that code sets out to implement (previous message
subtracted before a check's update, new one added -- the one change is the message matrix living across checks).
Nothing in the reference executes it, so its parity is UNPINNED: it is checked against an independent CPU
restatement only.
"""

from __future__ import annotations

import logging
from typing import List, Tuple

import numpy as np
import torch

from ldpc_decoder import LDPCCode, _as_batch
from neural_2d_decoder import _DegreeSharedDecoder

logger = logging.getLogger(__name__)


class NonUniformQuantizer:
    """
    Non-uniform quantiser with power-function thresholds
    tau_j = C * (j / (2^(bc-1) - 1))^gamma,  j = 0 .. 2^(bc-1)-1.
    Host-side utility (vector API of the reference); inside the decoders the same
    tables are applied by the HIP kernels.
    """

    def __init__(self, bc: int, C: float, gamma: float):
        self.bc = bc
        self.C = C
        self.gamma = gamma
        self.thresholds = self._calculate_thresholds()
        logger.info(f"Initialized quantizer: bc={bc}, C={C}, gamma={gamma}")

    def _calculate_thresholds(self) -> List[float]:
        levels = 2 ** (self.bc - 1)
        top = levels - 1
        return [self.C * (j / top) ** self.gamma for j in range(levels)]

    def thresholds_f32(self) -> np.ndarray:
        """float32(tau): the values the fp32 comparisons actually use."""
        return np.asarray(self.thresholds, dtype=np.float32)

    def quantize(self, x: torch.Tensor) -> torch.Tensor:
        """code = (sign(x) < 0) * 2^(bc-1) + max{j : |x| >= tau_j}   (int64)"""
        levels = 2 ** (self.bc - 1)
        mag = torch.abs(x)
        level = torch.zeros_like(mag, dtype=torch.long)
        for j, tau in enumerate(self.thresholds):      # later thresholds overwrite earlier ones
            level = torch.where(mag >= tau, torch.full_like(level, j), level)
        sign_bit = (torch.sign(x) < 0).long()
        return sign_bit * levels + level

    def dequantize(self, quantized: torch.Tensor) -> torch.Tensor:
        """(1 - 2*sign_bit) * tau[code mod 2^(bc-1)] as float32 (code 2^(bc-1) is -0.0)"""
        levels = 2 ** (self.bc - 1)
        sign_bit = (quantized >= levels).long()
        idx = quantized % levels
        table = torch.tensor(self.thresholds, dtype=torch.float32, device=quantized.device)
        mag = table[idx.clamp(0, levels - 1)]
        return (1 - 2 * sign_bit.float()) * mag


def _quantizer_index(n_quantizers: int, max_iterations: int, iteration: int) -> int:
    """which quantiser an iteration uses: a single quantiser always; otherwise the first
    for the first third of max_iterations, the second for the second third, the last
    afterwards (rcq_decoder.py:156-167, 482-493)."""
    if n_quantizers == 1:
        return 0
    if iteration < max_iterations // 3:
        return 0
    if iteration < 2 * max_iterations // 3:
        return 1 if n_quantizers > 1 else 0
    return n_quantizers - 1


def _quantizer_schedule(n_quantizers: int, max_iterations: int) -> np.ndarray:
    T = int(max_iterations)
    out = np.zeros(max(T, 1), dtype=np.int32)
    for it in range(T):
        out[it] = _quantizer_index(n_quantizers, T, it)
    return out


def _threshold_table(quantizers) -> np.ndarray:
    return np.stack([q.thresholds_f32() for q in quantizers]).astype(np.float32)


class RCQMinSumDecoder:
    """RCQ MinSum decoder with non-uniform quantisation (flooding schedule)."""

    def __init__(self, code: LDPCCode, bc: int, bv: int, quantizer_params: List[Tuple[float, float]],
                 max_iterations: int = 50, layered: bool = False):
        self.code = code
        self.bc = bc
        self.bv = bv                      # stored, never used (as in the reference)
        self.max_iterations = max_iterations
        self.layered = layered
        self.quantizers = [NonUniformQuantizer(bc, C, gamma) for C, gamma in quantizer_params]
        self._engine = None
        self._engine_key = None
        logger.info(f"Initialized RCQ MinSum decoder: bc={bc}, bv={bv}, "
                    f"layered={layered}, {len(self.quantizers)} quantizers")

    def _get_quantizer(self, iteration: int) -> NonUniformQuantizer:
        return self.quantizers[_quantizer_index(len(self.quantizers), self.max_iterations, iteration)]

    def _get_engine(self, device):
        import _native as nat
        from engine import DecodeEngine, _require_gpu
        dev = _require_gpu(device)
        g = self.code.tanner_graph()
        T = int(self.max_iterations)
        thr = _threshold_table(self.quantizers)
        key = (dev.index, id(g), T, thr.tobytes(), self.layered if isinstance(self.layered, str) else bool(self.layered))
        if self._engine is None or self._engine_key != key:
            rows = max(T, 1)
            self._engine = DecodeEngine(
                g, dtype=torch.float32, c2v_form=nat.C2V_RCQ, iters=T, device=dev,
                schedule=(nat.SCHED_LAYERED if self.layered == "paper" else
                          nat.SCHED_LAYERED_REF if self.layered else nat.SCHED_FLOODING),
                beta=np.ones((rows, 1), np.float32), beta_slot=np.zeros(g.E, np.int32),     # prod(signs) * min
                alpha=np.ones((rows, 1), np.float32), alpha_slot=np.zeros(g.n, np.int32),   # llr + sum(others)
                thresholds=thr, q_of_iter=_quantizer_schedule(len(self.quantizers), T))
            self._engine_key = key
        return self._engine

    def decode(self, llr: torch.Tensor, early_stop: bool = True, device=None):
        """
        Args:
            llr: log-likelihood ratios from the channel (torch tensor), ``[n]`` or ``[B, n]``
        Returns:
            decoded_bits (int32), success (bool / bool[B]), iterations (int / int32[B])
        """
        if not isinstance(llr, torch.Tensor):
            raise TypeError("llr must be a torch.Tensor")    # the reference needs llr.device as well
        _, x, single = _as_batch(llr, self.code.n)
        eng = self._get_engine(x.device if x.is_cuda else device)
        if not x.is_cuda and x.shape[0] <= eng.HOST_BATCH_MAX:       # the reference's call shape: torch.ops.ldpc.decode_host
            res = eng.decode_host_op(x.detach().to(torch.float32), early_stop=early_stop, want_posterior=False)
        else:
            res = eng.decode_op(x.detach().to(device=eng.device, dtype=torch.float32), early_stop=early_stop,
                                want_posterior=False)
        out_dev = llr.device
        if single:
            return res.bits[0].to(out_dev), bool(res.success[0].item()), int(res.iterations[0].item())
        return res.bits.to(out_dev), res.success.to(out_dev), res.iterations.to(out_dev)


class WeightedRCQDecoder(_DegreeSharedDecoder):
    """Weighted RCQ decoder: degree-shared neural weights + RCQ quantisation."""

    def __init__(self, code: LDPCCode, bc: int, bv: int, quantizer_params: List[Tuple[float, float]],
                 weight_sharing_type: int = 2, max_iterations: int = 50, layered: bool = False):
        super().__init__()
        self.bc = bc
        self.bv = bv
        self.layered = layered            # stored; forward ignores it exactly like the reference's
        self.quantizers = [NonUniformQuantizer(bc, C, gamma) for C, gamma in quantizer_params]
        self._init_sharing(code, weight_sharing_type, max_iterations)
        logger.info(f"Initialized Weighted RCQ decoder: bc={bc}, bv={bv}, "
                    f"weight_type={weight_sharing_type}, layered={layered}")

    def _get_quantizer(self, iteration: int) -> NonUniformQuantizer:
        return self.quantizers[_quantizer_index(len(self.quantizers), self.max_iterations, iteration)]

    def _extra_key(self):
        return _threshold_table(self.quantizers).tobytes()

    def _engine_kwargs(self, layout, beta, alpha):
        import _native as nat
        return dict(c2v_form=nat.C2V_RCQ, beta=beta, beta_slot=layout.beta_slot,
                    alpha=alpha, alpha_slot=layout.alpha_slot,
                    thresholds=_threshold_table(self.quantizers),
                    q_of_iter=_quantizer_schedule(len(self.quantizers), self.max_iterations))

    def forward(self, llr: torch.Tensor, early_stop: bool = True, device=None):
        """
        Returns:
            decoded_bits (int32), posterior (float32), iterations (int / int32[B])
        """
        res, single, out_dev = self._decode(llr, early_stop, device)
        if single:
            return res.bits[0].to(out_dev), res.posterior[0].to(out_dev), int(res.iterations[0].item())
        return res.bits.to(out_dev), res.posterior.to(out_dev), res.iterations.to(out_dev)
