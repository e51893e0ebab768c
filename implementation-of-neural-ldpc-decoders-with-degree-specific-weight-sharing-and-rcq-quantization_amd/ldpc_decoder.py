"""
Drop-in for the reference module of the same name (ldpc_decoder.py):
``LDPCCode``, ``BasicMinSumDecoder``, ``create_test_ldpc_code``,
``simulate_awgn_channel`` -- same names, constructor arguments, attributes and
return tuples -- with ``decode`` running on the MI355X engine.

Reference behaviour mirrored (file:line in /root/reference):
  LDPCCode dataclass, rate, degree dicts                ldpc_decoder.py:26-54
  BasicMinSumDecoder(code, factor=0.7).decode(llr)      ldpc_decoder.py:56-153
      -> (np.ndarray[n] int64, bool, int); T = code.max_iterations
  NeuralMinSumDecoder(code, max_iterations=50)          ldpc_decoder.py:155-272
      the edge-weight decoder this module ALSO defines in the reference (weights randn*0.1, no +0.7);
      resolved lazily from neural_minsum_decoder.py (which imports this module)
  create_test_ldpc_code(): 4x7 H, max_iterations=10     ldpc_decoder.py:274-284
  simulate_awgn_channel(codeword, snr_db)               ldpc_decoder.py:286-302

Extensions (no reference counterpart): ``llr`` may be a batch ``[B, n]`` (numpy or a
torch tensor already on the GPU) -> ``(bits[B, n], success[B], iterations[B])``, the
row-wise map of the single-codeword call; keyword ``early_stop=False`` runs exactly T
iterations.  Arithmetic type follows the input: float64 in -> fp64 kernels with
np.sum's association order (results identical to the reference), float32 in -> fp32.
"""

from __future__ import annotations

import logging
from dataclasses import dataclass
from typing import Dict, Tuple

import numpy as np

from tanner_graph import TannerGraph

logger = logging.getLogger(__name__)


@dataclass
class LDPCCode:
    """LDPC Code parameters (field order and defaults as ldpc_decoder.py:26-32)"""
    n: int                      # codeword length
    k: int                      # dataword length
    H: np.ndarray               # parity check matrix (dense 0/1)
    max_iterations: int = 50

    @property
    def rate(self) -> float:
        return self.k / self.n

    @property
    def check_node_degrees(self) -> Dict[int, int]:
        """{check: row sum}; the reference recomputes this with a Python loop on every
        access (ldpc_decoder.py:38-45) -- same content, one vectorised pass."""
        H = np.asarray(self.H)
        return {i: int(d) for i, d in enumerate(H.sum(axis=1))}

    @property
    def variable_node_degrees(self) -> Dict[int, int]:
        H = np.asarray(self.H)
        return {j: int(d) for j, d in enumerate(H.sum(axis=0))}

    # ---- engine side ---------------------------------------------------------------
    def tanner_graph(self) -> TannerGraph:
        """CSR/CSC edge lists of H, compiled once per code object (the decoders below
        never touch the dense matrix again).  Rebuilt if H is replaced."""
        cache = self.__dict__.get("_tanner")
        if cache is not None and cache[0] is self.H:
            return cache[1]
        g = TannerGraph.from_dense(self.H)
        self.__dict__["_tanner"] = (self.H, g)
        return g

    @classmethod
    def from_graph(cls, graph: TannerGraph, k: int, max_iterations: int = 50) -> "LDPCCode":
        """Code from a sparse edge list; ``H`` is materialised as dense int8 because the
        reference API exposes it, the graph itself is reused as is."""
        H = graph.to_dense(np.int8)
        code = cls(n=graph.n, k=k, H=H, max_iterations=max_iterations)
        code.__dict__["_tanner"] = (H, graph)
        return code


def _as_batch(llr, n: int):
    """-> (kind, 2-D view, was_single).  kind 'np' | 'torch'."""
    import torch
    if isinstance(llr, torch.Tensor):
        x, kind = llr, "torch"
    else:
        x, kind = np.asarray(llr), "np"
    if x.ndim == 1:
        if x.shape[0] != n:
            raise ValueError(f"llr has {x.shape[0]} entries, code length is {n}")
        return kind, x.reshape(1, n), True
    if x.ndim == 2 and x.shape[1] == n:
        return kind, x, False
    raise ValueError(f"llr must have shape [{n}] or [B, {n}], got {tuple(x.shape)}")


class BasicMinSumDecoder:
    """Basic (normalised) MinSum LDPC decoder, flooding schedule."""

    def __init__(self, code: LDPCCode, factor: float = 0.7):
        self.code = code
        self.factor = factor
        self._engines = {}

    def _engine(self, torch_dtype, device):
        import torch
        import _native as nat
        from engine import DecodeEngine, _require_gpu
        dev = _require_gpu(device)
        g = self.code.tanner_graph()
        T = int(self.code.max_iterations)
        key = (torch_dtype, dev.index, id(g), T, float(self.factor))
        eng = self._engines.get(key)
        if eng is None:
            np_dt = np.float32 if torch_dtype == torch.float32 else np.float64
            rows = max(T, 1)
            eng = DecodeEngine(g, dtype=torch_dtype, c2v_form=nat.C2V_NMS, iters=T,
                               beta=np.full((rows, 1), self.factor, dtype=np_dt),     # factor * min * prod(signs)
                               beta_slot=np.zeros(g.E, np.int32),
                               alpha=np.ones((rows, 1), dtype=np_dt),                 # llr + 1 * sum(others)
                               alpha_slot=np.zeros(g.n, np.int32), device=dev)
            self._engines = {key: eng}
        return eng

    def decode(self, llr, early_stop: bool = True, device=None):
        """
        Decode using the MinSum algorithm.

        Args:
            llr: log-likelihood ratios from the channel, ``[n]`` (reference) or ``[B, n]``

        Returns:
            decoded_bits, success, iterations -- for ``[n]`` input exactly the reference's
            ``(np.ndarray int64, bool, int)``; for ``[B, n]`` arrays of those (torch tensors
            on the GPU when the input was a GPU tensor).
        """
        import torch
        n = self.code.n
        kind, x, single = _as_batch(llr, n)
        if kind == "np":
            dt = torch.float32 if x.dtype == np.float32 else torch.float64
            eng = self._engine(dt, device)
            xh = torch.from_numpy(np.ascontiguousarray(x, dtype=eng.np_dtype))
            if xh.shape[0] <= eng.HOST_BATCH_MAX:
                # the reference's call shape: host vector in, host results out -- one staged copy each way
                # (torch.ops.ldpc.decode_host)
                res = eng.decode_host_op(xh, early_stop=early_stop, want_posterior=False)
                bits = res.bits.numpy().astype(np.int64)
                succ, its = res.success.numpy(), res.iterations.numpy()
                if single:
                    return bits[0], bool(succ[0]), int(its[0])
                return bits, succ, its
            xd = xh.to(eng.device)
        else:
            dt = torch.float64 if x.dtype == torch.float64 else torch.float32
            eng = self._engine(dt, x.device if x.is_cuda else device)
            xd = x.to(device=eng.device, dtype=dt)
        res = eng.decode_op(xd, early_stop=early_stop, want_posterior=False)     # torch.ops.ldpc.decode
        if kind == "torch" and llr.is_cuda and not single:
            return res.bits, res.success, res.iterations
        bits = res.bits.cpu().numpy().astype(np.int64)      # reference: (posterior < 0).astype(int)
        succ = res.success.cpu().numpy()
        its = res.iterations.cpu().numpy()
        if single:
            return bits[0], bool(succ[0]), int(its[0])
        return bits, succ, its


def create_test_ldpc_code() -> LDPCCode:
    """The reference's toy code: (7,4) parity-check matrix, 10 iterations
    (ldpc_decoder.py:274-284)."""
    H = np.array([
        [1, 1, 0, 1, 0, 0, 0],
        [0, 1, 1, 0, 1, 0, 0],
        [1, 0, 1, 0, 0, 1, 0],
        [1, 1, 1, 0, 0, 0, 1],
    ])
    return LDPCCode(n=7, k=4, H=H, max_iterations=10)


def simulate_awgn_channel(codeword: np.ndarray, snr_db: float) -> np.ndarray:
    """BPSK over AWGN exactly as the reference draws it (ldpc_decoder.py:286-302):
    symbols 2c-1, noise power 1/snr, llr = 2r/noise_power, numpy global RNG.
    Note the reference's sign quirk: bit 0 maps to a NEGATIVE LLR here
    while every decoder decides ``posterior < 0 -> 1``."""
    codeword = np.asarray(codeword)
    bpsk_symbols = 2 * codeword - 1
    snr_linear = 10 ** (snr_db / 10)
    noise_power = 1 / snr_linear
    noise = np.random.normal(0, np.sqrt(noise_power), len(bpsk_symbols))
    received = bpsk_symbols + noise
    return 2 * received / noise_power


def __getattr__(name):
    # `from ldpc_decoder import NeuralMinSumDecoder` (the reference defines one here too, :155); lazy because
    # neural_minsum_decoder imports LDPCCode from this module
    if name == "NeuralMinSumDecoder":
        from neural_minsum_decoder import LdpcDecoderNeuralMinSumDecoder
        return LdpcDecoderNeuralMinSumDecoder
    raise AttributeError(f"module {__name__!r} has no attribute {name!r}")
