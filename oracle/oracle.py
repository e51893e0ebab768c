"""
ctypes front-end of the CPU restatement (oracle/ldpc_oracle.c).

TEST INFRASTRUCTURE, NOT PRODUCT CODE: importable only from tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg.  It deliberately does
not import anything from the product package -- the Tanner graph is rebuilt
here from the dense matrix with numpy the way the reference scans it
(``np.where(H[i, :] == 1)``, ldpc_decoder.py:92,124), and the weight tables are
rebuilt from the reference's ``iter_{t}_dc{dc}`` / ``iter_{t}_dv{dv}`` keys
(neural_2d_decoder.py:46-131), so a disagreement between the product's host
logic and this file shows up in the parity tests.
"""

from __future__ import annotations

import ctypes as C
import os
import subprocess
from typing import Dict, Optional, Sequence, Tuple

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libldpc_oracle.so")

C2V_NMS, C2V_RCQ, C2V_OMS = 0, 1, 2
SUM_TORCH, SUM_NUMPY = 0, 1


def build(force: bool = False) -> str:
    """Compile the library with the committed Makefile (gcc only)."""
    import hashlib
    h = hashlib.sha256()
    for f in ("ldpc_oracle.c", "ldpc_oracle_impl.h", "Makefile"):          # content, not file times (they do not survive a copy)
        with open(os.path.join(_HERE, f), "rb") as fh:
            h.update(f.encode() + b"\0" + fh.read())
    want, rec = h.hexdigest(), _LIB_PATH + ".srchash"
    try:
        have = open(rec).read().strip()
    except OSError:
        have = None
    if force or not os.path.exists(_LIB_PATH) or have != want:
        subprocess.run(["make", "-C", _HERE, "-B"], check=True, capture_output=True)
        with open(rec, "w") as fh:
            fh.write(want + "\n")
    return _LIB_PATH


class _Graph(C.Structure):
    _fields_ = [("n", C.c_int32), ("m", C.c_int32), ("E", C.c_int32),
                ("check_ptr", C.c_void_p), ("var_idx", C.c_void_p),
                ("var_ptr", C.c_void_p), ("csc_edge", C.c_void_p)]


class _Params(C.Structure):
    _fields_ = [("iters", C.c_int32), ("early_stop", C.c_int32), ("c2v_form", C.c_int32),
                ("sum_order", C.c_int32), ("n_beta_slots", C.c_int32), ("n_alpha_slots", C.c_int32),
                ("beta_slot", C.c_void_p), ("alpha_slot", C.c_void_p),
                ("n_levels", C.c_int32), ("n_quantizers", C.c_int32),
                ("thresholds", C.c_void_p), ("q_of_iter", C.c_void_p),
                ("n_oms_alpha_slots", C.c_int32), ("oms_alpha_slot", C.c_void_p),
                ("oms_alpha", C.c_void_p)]


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        _lib = C.CDLL(_LIB_PATH)
        for name, real in (("oracle_decode_f32", C.c_float), ("oracle_decode_f64", C.c_double)):
            fn = getattr(_lib, name)
            fn.restype = C.c_int
            fn.argtypes = [C.POINTER(_Graph), C.POINTER(_Params), C.c_void_p, C.c_void_p, C.c_void_p,
                           C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        _lib.oracle_sum_f32.restype = C.c_float
        _lib.oracle_sum_f32.argtypes = [C.c_int, C.c_void_p, C.c_int]
        _lib.oracle_sum_f64.restype = C.c_double
        _lib.oracle_sum_f64.argtypes = [C.c_int, C.c_void_p, C.c_int]
        _lib.oracle_quantize_f32.restype = None
        _lib.oracle_quantize_f32.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p]
        _lib.oracle_dequantize_f32.restype = None
        _lib.oracle_dequantize_f32.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p]
        _lib.oracle_decode_layered_paper_f32.restype = C.c_int
        _lib.oracle_decode_layered_paper_f32.argtypes = [C.POINTER(_Graph), C.c_int, C.c_void_p, C.c_int, C.c_void_p,
                                                         C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        _lib.oracle_decode_layered_f32.restype = C.c_int
        _lib.oracle_decode_layered_f32.argtypes = [C.POINTER(_Graph), C.c_int, C.c_void_p, C.c_int, C.c_void_p,
                                                   C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        _lib.oracle_num_threads.restype = C.c_int
        _lib.oracle_set_num_threads.argtypes = [C.c_int]
    return _lib


def _p(a: Optional[np.ndarray]):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


# --------------------------------------------------------------------------- graph
class OracleGraph:
    """CSR/CSC built independently of the product (np.nonzero on ``H == 1``)."""

    def __init__(self, H=None, *, n=None, check_ptr=None, var_idx=None):
        if H is not None:
            H = np.asarray(H)
            m, n = H.shape
            rows, cols = np.nonzero(H == 1)          # row-major scan == CSR order
        else:
            check_ptr = np.asarray(check_ptr, dtype=np.int64)
            m = len(check_ptr) - 1
            rows = np.repeat(np.arange(m), np.diff(check_ptr))
            cols = np.asarray(var_idx, dtype=np.int64)
        self.n, self.m, self.E = int(n), int(m), int(len(rows))
        self.rows = rows.astype(np.int32)
        self.var_idx = np.ascontiguousarray(cols, dtype=np.int32)
        self.dc = np.bincount(rows, minlength=m).astype(np.int64)
        self.dv = np.bincount(cols, minlength=n).astype(np.int64)
        self.check_ptr = np.concatenate([[0], np.cumsum(self.dc)]).astype(np.int32)
        self.var_ptr = np.concatenate([[0], np.cumsum(self.dv)]).astype(np.int32)
        # for every variable, its edges in ascending check order
        self.csc_edge = np.lexsort((rows, cols)).astype(np.int32)
        self._c = _Graph(self.n, self.m, self.E, _p(self.check_ptr), _p(self.var_idx),
                         _p(self.var_ptr), _p(self.csc_edge))


# --------------------------------------------------------------------------- tables
def quantizer_thresholds(bc: int, C_: float, gamma: float):
    """rcq_decoder.py:48-57, Python float arithmetic."""
    max_idx = 2 ** (bc - 1) - 1
    return [C_ * (j / (2 ** (bc - 1) - 1)) ** gamma for j in range(max_idx + 1)]


def quantizer_schedule(T: int, n_quantizers: int):
    """rcq_decoder.py:156-167 / 482-493 -> quantiser index per iteration."""
    out = []
    for it in range(T):
        if n_quantizers == 1:
            out.append(0)
        elif it < T // 3:
            out.append(0)
        elif it < 2 * T // 3:
            out.append(1 if n_quantizers > 1 else 0)
        else:
            out.append(n_quantizers - 1)
    return np.asarray(out, dtype=np.int32)


def weight_tables(g: OracleGraph, weight_sharing_type: int, T: int,
                  beta: Dict[str, float], alpha: Dict[str, float],
                  beta_default: float = 0.7, alpha_default: float = 1.0,
                  dtype=np.float32):
    """Flatten the reference's ParameterDict lookups (_get_beta_weight /
    _get_alpha_weight, neural_2d_decoder.py:84-131) into [T][slots] tables plus
    per-edge / per-variable slot indices."""
    E, n = g.E, g.n
    dc_e = g.dc[g.rows]                 # degree of the check of each edge
    dv_e = g.dv[g.var_idx]              # degree of the variable of each edge
    # beta
    if weight_sharing_type == 1:
        pairs = sorted(set(zip(dc_e.tolist(), dv_e.tolist())))
        lut = {p: s for s, p in enumerate(pairs)}
        beta_slot = np.asarray([lut[(a, b)] for a, b in zip(dc_e.tolist(), dv_e.tolist())], dtype=np.int32)
        keys = [f"dc{a}_dv{b}" for a, b in pairs]
    elif weight_sharing_type in (2, 3):
        dcs = sorted(set(dc_e.tolist()))
        lut = {d: s for s, d in enumerate(dcs)}
        beta_slot = np.asarray([lut[a] for a in dc_e.tolist()], dtype=np.int32)
        keys = [f"dc{a}" for a in dcs]
    else:
        beta_slot = np.zeros(E, dtype=np.int32)
        keys = [None]
    bt = np.full((max(T, 1), max(len(keys), 1)), beta_default, dtype=dtype)
    for t in range(T):
        for s, k in enumerate(keys):
            if k is not None and f"iter_{t}_{k}" in beta:
                bt[t, s] = beta[f"iter_{t}_{k}"]
    # alpha
    if weight_sharing_type in (2, 4):
        dvs = sorted(set(g.dv.tolist()))
        lut = {d: s for s, d in enumerate(dvs)}
        alpha_slot = np.asarray([lut[a] for a in g.dv.tolist()], dtype=np.int32)
        akeys = [f"dv{a}" for a in dvs]
    else:
        alpha_slot = np.zeros(n, dtype=np.int32)
        akeys = [None]
    at = np.full((max(T, 1), max(len(akeys), 1)), alpha_default, dtype=dtype)
    for t in range(T):
        for s, k in enumerate(akeys):
            if k is not None and f"iter_{t}_{k}" in alpha:
                at[t, s] = alpha[f"iter_{t}_{k}"]
    return bt, beta_slot, at, alpha_slot


# --------------------------------------------------------------------------- decode
def decode(g: OracleGraph, llr: np.ndarray, *, T: int, early_stop: bool = True,
           c2v_form: int = C2V_NMS, sum_order: int = SUM_TORCH,
           beta: np.ndarray, beta_slot: np.ndarray, alpha: np.ndarray, alpha_slot: np.ndarray,
           thresholds: Optional[np.ndarray] = None, q_of_iter: Optional[np.ndarray] = None,
           oms_alpha: Optional[np.ndarray] = None, oms_alpha_slot: Optional[np.ndarray] = None,
           trace_codes: bool = False, threads: Optional[int] = None):
    """Row-wise map of the single-codeword reference over llr[B, n].

    Returns (bits int32[B,n], posterior[B,n], iterations int32[B], success bool[B][, codes uint8[B,T,E]])
    """
    L = lib()
    dt = llr.dtype
    if dt not in (np.float32, np.float64):
        raise TypeError("llr must be float32 or float64")
    llr2 = np.ascontiguousarray(llr.reshape(-1, g.n))
    B = llr2.shape[0]
    beta = np.ascontiguousarray(beta, dtype=dt)
    alpha = np.ascontiguousarray(alpha, dtype=dt)
    beta_slot = np.ascontiguousarray(beta_slot, dtype=np.int32)
    alpha_slot = np.ascontiguousarray(alpha_slot, dtype=np.int32)
    assert beta_slot.shape == (g.E,) and alpha_slot.shape == (g.n,)
    assert beta.ndim == 2 and beta.shape[0] >= T and alpha.ndim == 2 and alpha.shape[0] >= T
    assert g.E == 0 or beta_slot.max() < beta.shape[1]
    assert alpha_slot.max() < alpha.shape[1]
    prm = _Params()
    prm.iters, prm.early_stop, prm.c2v_form, prm.sum_order = T, int(early_stop), c2v_form, sum_order
    prm.n_beta_slots, prm.n_alpha_slots = beta.shape[1], alpha.shape[1]
    prm.beta_slot, prm.alpha_slot = _p(beta_slot), _p(alpha_slot)
    keep = [beta, alpha, beta_slot, alpha_slot]
    if c2v_form == C2V_RCQ:
        thresholds = np.ascontiguousarray(thresholds, dtype=np.float32)
        q_of_iter = np.ascontiguousarray(q_of_iter, dtype=np.int32)
        assert thresholds.ndim == 2 and q_of_iter.shape[0] >= T
        prm.n_quantizers, prm.n_levels = thresholds.shape
        prm.thresholds, prm.q_of_iter = _p(thresholds), _p(q_of_iter)
        keep += [thresholds, q_of_iter]
    if c2v_form == C2V_OMS and oms_alpha is not None:
        oms_alpha = np.ascontiguousarray(oms_alpha, dtype=dt)
        oms_alpha_slot = np.ascontiguousarray(oms_alpha_slot, dtype=np.int32)
        prm.n_oms_alpha_slots = oms_alpha.shape[1]
        prm.oms_alpha, prm.oms_alpha_slot = _p(oms_alpha), _p(oms_alpha_slot)
        keep += [oms_alpha, oms_alpha_slot]
    bits = np.zeros((B, g.n), dtype=np.int32)
    post = np.zeros((B, g.n), dtype=dt)
    iters = np.zeros(B, dtype=np.int32)
    succ = np.zeros(B, dtype=np.uint8)
    codes = np.zeros((B, T, g.E), dtype=np.uint8) if trace_codes else None
    if threads is not None:
        L.oracle_set_num_threads(int(threads))
    fn = L.oracle_decode_f32 if dt == np.float32 else L.oracle_decode_f64
    rc = fn(C.byref(g._c), C.byref(prm), _p(beta), _p(alpha), _p(llr2), B,
            _p(bits), _p(post), _p(iters), _p(succ), _p(codes))
    if rc != 0:
        raise RuntimeError(f"oracle_decode failed rc={rc}")
    out = (bits, post, iters, succ.astype(bool))
    return out + (codes,) if trace_codes else out


# convenience wrappers, one per reference decoder ---------------------------------
def basic_minsum(g: OracleGraph, llr, factor: float = 0.7, T: int = 50, early_stop=True, dtype=np.float64, **kw):
    """BasicMinSumDecoder.decode (ldpc_decoder.py:63-153).  fp64/np.sum order by
    default; dtype=float32 gives the fp32 engine's arithmetic (torch.sum order)."""
    llr = np.asarray(llr, dtype=dtype)
    return decode(g, llr, T=T, early_stop=early_stop, c2v_form=C2V_NMS,
                  sum_order=SUM_NUMPY if dtype == np.float64 else SUM_TORCH,
                  beta=np.full((max(T, 1), 1), factor, dtype=dtype), beta_slot=np.zeros(g.E, np.int32),
                  alpha=np.ones((max(T, 1), 1), dtype=dtype), alpha_slot=np.zeros(g.n, np.int32), **kw)


def neural2d(g: OracleGraph, llr, weight_sharing_type: int, T: int, beta: Dict[str, float],
             alpha: Dict[str, float], early_stop=True, **kw):
    """Neural2DMinSumDecoder.forward (neural_2d_decoder.py:133-225)."""
    bt, bs, at, as_ = weight_tables(g, weight_sharing_type, T, beta, alpha)
    return decode(g, np.asarray(llr, dtype=np.float32), T=T, early_stop=early_stop, c2v_form=C2V_NMS,
                  sum_order=SUM_TORCH, beta=bt, beta_slot=bs, alpha=at, alpha_slot=as_, **kw)


def rcq(g: OracleGraph, llr, bc: int, quantizer_params: Sequence[Tuple[float, float]], T: int,
        early_stop=True, **kw):
    """RCQMinSumDecoder._decode_flooding (rcq_decoder.py:190-279)."""
    thr = np.asarray([quantizer_thresholds(bc, c, gm) for c, gm in quantizer_params], dtype=np.float32)
    return decode(g, np.asarray(llr, dtype=np.float32), T=T, early_stop=early_stop, c2v_form=C2V_RCQ,
                  sum_order=SUM_TORCH, beta=np.ones((max(T, 1), 1), np.float32), beta_slot=np.zeros(g.E, np.int32),
                  alpha=np.ones((max(T, 1), 1), np.float32), alpha_slot=np.zeros(g.n, np.int32),
                  thresholds=thr, q_of_iter=quantizer_schedule(T, len(quantizer_params)), **kw)


def weighted_rcq(g: OracleGraph, llr, bc: int, quantizer_params, weight_sharing_type: int, T: int,
                 beta: Dict[str, float], alpha: Dict[str, float], early_stop=True, **kw):
    """WeightedRCQDecoder.forward (rcq_decoder.py:495-597)."""
    thr = np.asarray([quantizer_thresholds(bc, c, gm) for c, gm in quantizer_params], dtype=np.float32)
    bt, bs, at, as_ = weight_tables(g, weight_sharing_type, T, beta, alpha)
    return decode(g, np.asarray(llr, dtype=np.float32), T=T, early_stop=early_stop, c2v_form=C2V_RCQ,
                  sum_order=SUM_TORCH, beta=bt, beta_slot=bs, alpha=at, alpha_slot=as_,
                  thresholds=thr, q_of_iter=quantizer_schedule(T, len(quantizer_params)), **kw)


def neural2d_offset(g: OracleGraph, llr, weight_sharing_type: int, T: int, beta: Dict[str, float],
                    alpha: Dict[str, float], early_stop=True, **kw):
    """Neural2DOffsetMinSumDecoder.forward (neural_2d_decoder.py:338-434): C2V = prod(signs) *
    (relu(min - beta) - alpha) with both looked up per edge (defaults 0.0), plain V2C sums."""
    bt, bs, at, as_ = weight_tables(g, weight_sharing_type, T, beta, alpha, beta_default=0.0, alpha_default=0.0)
    return decode(g, np.asarray(llr, dtype=np.float32), T=T, early_stop=early_stop, c2v_form=C2V_OMS,
                  sum_order=SUM_TORCH, beta=bt, beta_slot=bs,
                  alpha=np.ones((max(T, 1), 1), np.float32), alpha_slot=np.zeros(g.n, np.int32),
                  oms_alpha=at, oms_alpha_slot=as_[g.var_idx], **kw)


def edge_weight_table(g: OracleGraph, T: int, beta: Dict[str, float]):
    """{"iter_{t}_c{i}_v{j}": w} -> [T][E] in CSR edge order (neural_minsum_decoder.py:45-53)"""
    out = np.zeros((max(T, 1), max(g.E, 1)), dtype=np.float32)
    for t in range(T):
        for e in range(g.E):
            out[t, e] = beta[f"iter_{t}_c{int(g.rows[e])}_v{int(g.var_idx[e])}"]
    return out


def neural_minsum(g: OracleGraph, llr, T: int, beta: Dict[str, float], offset: bool = False, early_stop=True, **kw):
    """NeuralMinSumDecoder.forward (neural_minsum_decoder.py:58-150) / NeuralOffsetMinSumDecoder.forward
    (:192-285, offset=True: C2V = prod(signs) * relu(min - beta))."""
    return decode(g, np.asarray(llr, dtype=np.float32), T=T, early_stop=early_stop,
                  c2v_form=C2V_OMS if offset else C2V_NMS, sum_order=SUM_TORCH,
                  beta=edge_weight_table(g, T, beta), beta_slot=np.arange(g.E, dtype=np.int32),
                  alpha=np.ones((max(T, 1), 1), np.float32), alpha_slot=np.zeros(g.n, np.int32), **kw)


def rcq_layered(g: OracleGraph, llr, bc: int, quantizer_params, T: int, paper: bool = False):
    """RCQMinSumDecoder(layered=True).decode (rcq_decoder.py:281-350), bug-compatible; paper=True: the schedule that
    code sets out to implement (message matrix kept across checks) -- PARITY UNPINNED, nothing in the reference runs it.
    Returns (bits, final posteriors, iterations, success)."""
    thr = np.ascontiguousarray([quantizer_thresholds(bc, c, gm) for c, gm in quantizer_params], dtype=np.float32)
    sched = quantizer_schedule(T, len(quantizer_params))
    x = np.ascontiguousarray(np.asarray(llr, dtype=np.float32).reshape(-1, g.n))
    B = x.shape[0]
    bits = np.zeros((B, g.n), np.int32); post = np.zeros((B, g.n), np.float32)
    iters = np.zeros(B, np.int32); succ = np.zeros(B, np.uint8)
    fn = lib().oracle_decode_layered_paper_f32 if paper else lib().oracle_decode_layered_f32
    rc = fn(C.byref(g._c), T, _p(thr), thr.shape[1], _p(sched), _p(x), B, _p(bits), _p(post), _p(iters), _p(succ))
    if rc != 0:
        raise RuntimeError("oracle_decode_layered failed")
    return bits, post, iters, succ.astype(bool)


def quantize(x, thresholds):
    x = np.ascontiguousarray(x, dtype=np.float32).ravel()
    thr = np.ascontiguousarray(thresholds, dtype=np.float32)
    out = np.zeros(x.size, dtype=np.int64)
    lib().oracle_quantize_f32(_p(x), x.size, _p(thr), thr.size, _p(out))
    return out


def dequantize(codes, thresholds):
    codes = np.ascontiguousarray(codes, dtype=np.int64).ravel()
    thr = np.ascontiguousarray(thresholds, dtype=np.float32)
    out = np.zeros(codes.size, dtype=np.float32)
    lib().oracle_dequantize_f32(_p(codes), codes.size, _p(thr), thr.size, _p(out))
    return out


def sum_f32(x, order=SUM_TORCH):
    x = np.ascontiguousarray(x, dtype=np.float32)
    return np.float32(lib().oracle_sum_f32(order, _p(x), x.size))


def sum_f64(x, order=SUM_NUMPY):
    x = np.ascontiguousarray(x, dtype=np.float64)
    return np.float64(lib().oracle_sum_f64(order, _p(x), x.size))


def num_threads() -> int:
    return int(lib().oracle_num_threads())
