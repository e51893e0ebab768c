"""
Batched decode engine: Python owner of the native handles (include/ldpc_hip.h).

One ``DecodeEngine`` = one Tanner graph + one decoder descriptor (C2V rule,
weight tables, quantiser tables) on one GPU.  PyTorch is used only as plumbing:
device memory (``torch.empty``), the current HIP stream, and dtype bookkeeping.
The arithmetic runs in the hand-written HIP kernels; there is no CPU path.

The host decoder classes (ldpc_decoder.py, neural_2d_decoder.py, rcq_decoder.py)
flatten their reference-style parameters into the tables this class uploads.
"""

from __future__ import annotations

import collections
import ctypes as C
import os
import threading
import weakref
from dataclasses import dataclass
from typing import Optional

import numpy as np
import torch

import _native as nat
from tanner_graph import TannerGraph


@dataclass
class DecodeResult:
    """Row-wise map of the reference's return tuple over a batch."""
    bits: Optional[torch.Tensor]          # int32 [B, n]     (posterior < 0)
    posterior: Optional[torch.Tensor]     # dtype [B, n]
    iterations: torch.Tensor              # int32 [B]        1-based, T when not converged
    success: torch.Tensor                 # bool  [B]
    packed_bits: Optional[torch.Tensor] = None   # uint8 [B, ceil(n/8)]


def _require_gpu(device) -> torch.device:
    if not torch.cuda.is_available():
        raise nat.NativeEngineError(
            "no HIP device visible: the LDPC decode path runs only on the MI355X engine "
            "(libldpc_hip.so); there is no CPU fallback")
    dev = torch.device(device if device is not None else "cuda")
    if dev.type != "cuda":
        raise nat.NativeEngineError(f"decode device must be a ROCm GPU, got {dev}")
    if dev.index is None:
        dev = torch.device("cuda", torch.cuda.current_device())
    return dev


class _NativeGraph:
    """ldpc_graph* for (graph, device); shared by all engines on that graph.  The cache holds the host graph WEAKLY: when
    the TannerGraph object dies its entries are evicted, and the device copy is destroyed (ldpc_graph_destroy) as soon as
    no engine uses it either -- a sweep over many codes does not accumulate device memory."""
    _cache = {}
    _lock = threading.Lock()

    def __init__(self, graph: TannerGraph, device: torch.device):
        lib = nat.load()
        self.handle = C.c_void_p()
        with torch.cuda.device(device):
            cp = np.ascontiguousarray(graph.check_ptr, dtype=np.int32)
            vi = np.ascontiguousarray(graph.var_idx, dtype=np.int32)
            nat.check(lib.ldpc_graph_create(C.byref(self.handle), graph.n, graph.m, graph.E,
                                            nat.ptr(cp), nat.ptr(vi)), "ldpc_graph_create")
        self._lib = lib

    def __del__(self):
        try:
            if getattr(self, "handle", None) is not None and self.handle.value:
                self._lib.ldpc_graph_destroy(self.handle)
                self.handle = C.c_void_p()
        except Exception:
            pass

    @classmethod
    def _evict(cls, key):
        with cls._lock:
            cls._cache.pop(key, None)

    @classmethod
    def get(cls, graph: TannerGraph, device: torch.device) -> "_NativeGraph":
        return cls._get(graph, device.index, lambda: cls(graph, device))

    @classmethod
    def _get(cls, graph, dev_index, make):
        key = (id(graph), dev_index)
        with cls._lock:
            hit = cls._cache.get(key)
            if hit is not None and hit[0]() is graph:
                return hit[1]
            ng = make()
            cls._cache[key] = (weakref.ref(graph), ng)
            weakref.finalize(graph, cls._evict, key)
            return ng


class DecodeEngine:
    def __init__(self, graph: TannerGraph, *, dtype: torch.dtype, c2v_form: int, iters: int,
                 beta: np.ndarray, beta_slot: np.ndarray, alpha: np.ndarray, alpha_slot: np.ndarray,
                 thresholds: Optional[np.ndarray] = None, q_of_iter: Optional[np.ndarray] = None,
                 oms_alpha: Optional[np.ndarray] = None, oms_alpha_slot: Optional[np.ndarray] = None,
                 schedule: int = nat.SCHED_FLOODING, device=None):
        if dtype not in (torch.float32, torch.float64):
            raise TypeError("engine dtype must be float32 or float64")
        self.device = _require_gpu(device)
        self.graph = graph
        self.dtype = dtype
        self.np_dtype = np.float32 if dtype == torch.float32 else np.float64
        self.iters = int(iters)
        self.c2v_form = int(c2v_form)
        lib = nat.load()
        self._lib = lib
        self._ng = _NativeGraph.get(graph, self.device)
        rows = max(self.iters, 1)
        beta = np.ascontiguousarray(beta, dtype=self.np_dtype).reshape(rows, -1)
        alpha = np.ascontiguousarray(alpha, dtype=self.np_dtype).reshape(rows, -1)
        beta_slot = np.ascontiguousarray(beta_slot, dtype=np.int32)
        alpha_slot = np.ascontiguousarray(alpha_slot, dtype=np.int32)
        if beta_slot.shape != (graph.E,) or alpha_slot.shape != (graph.n,):
            raise ValueError("slot arrays must have one entry per edge / per variable")
        desc = nat.DecoderDesc()
        desc.dtype = nat.LDPC_F32 if dtype == torch.float32 else nat.LDPC_F64
        desc.c2v_form, desc.iters = self.c2v_form, self.iters
        desc.schedule = int(schedule)
        desc.n_beta_slots, desc.beta, desc.beta_slot = beta.shape[1], nat.ptr(beta), nat.ptr(beta_slot)
        desc.n_alpha_slots, desc.alpha, desc.alpha_slot = alpha.shape[1], nat.ptr(alpha), nat.ptr(alpha_slot)
        keep = [beta, alpha, beta_slot, alpha_slot]
        if self.c2v_form == nat.C2V_RCQ:
            thresholds = np.ascontiguousarray(thresholds, dtype=np.float32)
            q_of_iter = np.ascontiguousarray(q_of_iter, dtype=np.int32)
            if thresholds.ndim != 2 or q_of_iter.shape[0] < self.iters:
                raise ValueError("bad quantiser tables")
            desc.n_quantizers, desc.n_levels = thresholds.shape
            desc.thresholds, desc.q_of_iter = nat.ptr(thresholds), nat.ptr(q_of_iter)
            keep += [thresholds, q_of_iter]
        if self.c2v_form == nat.C2V_OMS and oms_alpha is not None:
            oms_alpha = np.ascontiguousarray(oms_alpha, dtype=self.np_dtype).reshape(rows, -1)
            oms_alpha_slot = np.ascontiguousarray(oms_alpha_slot, dtype=np.int32)
            desc.n_oms_alpha_slots = oms_alpha.shape[1]
            desc.oms_alpha, desc.oms_alpha_slot = nat.ptr(oms_alpha), nat.ptr(oms_alpha_slot)
            keep += [oms_alpha, oms_alpha_slot]
        self._table_shapes = (beta.shape, alpha.shape, None if oms_alpha is None else oms_alpha.shape)
        # what the device holds (the gradient path restores them): beta, alpha, check-side alpha of the offset form
        self._tables = [beta.copy(), alpha.copy(), None if oms_alpha is None else oms_alpha.copy()]
        self.handle = C.c_void_p()
        with torch.cuda.device(self.device):
            nat.check(lib.ldpc_decoder_create(C.byref(self.handle), self._ng.handle, C.byref(desc)),
                      "ldpc_decoder_create")
        # scratch of ldpc_decode, one buffer PER STREAM: the C ABI wants one workspace per decode that may overlap
        # another on the device, and calls from different threads arrive on different (or the same) torch streams --
        # same stream = serialised by the stream, different streams = different buffers.  Guarded by _ws_lock.
        self._ws = collections.OrderedDict()       # stream handle -> buffer, least recently used first; at most _WS_MAX
        self._ws_lock = threading.Lock()
        self._host_lock = threading.Lock()         # staging buffers of decode_host (not held by _workspace callers)
        mode = os.environ.get("LDPC_ENGINE_MODE", "auto")
        try:
            self.set_mode(mode)
        except NotImplementedError:
            if mode not in ("gather", "pair"):
                raise
            self.set_mode("stream")          # the RCQ-only forms, asked for process-wide: other decoders stream

    def __del__(self):
        try:
            if getattr(self, "handle", None) is not None and self.handle.value:
                self._lib.ldpc_decoder_destroy(self.handle)
                self.handle = C.c_void_p()
        except Exception:
            pass

    # ------------------------------------------------------------------ engine choice
    _MODES = {"auto": nat.MODE_AUTO, "stream": nat.MODE_STREAM, "resident": nat.MODE_RESIDENT, "sweeps": nat.MODE_SWEEPS,
              "gather": nat.MODE_GATHER, "pair": nat.MODE_PAIR}

    def set_mode(self, mode: str):
        """'auto' (LDS-resident fused kernel when the code qualifies, else streaming), 'stream' (HBM-streaming
        engine; fp32 RCQ decoders run its cheapest applicable form: 1-byte codes both ways ('pair'), else the fused
        one-kernel-per-iteration form ('gather')), 'sweeps' (streaming, always one kernel per sweep with fp32
        variable->check rows), 'gather' / 'pair' (force that RCQ form; error when the decoder does not qualify),
        'resident' -- every choice gives identical results."""
        nat.check(self._lib.ldpc_decoder_set_mode(self.handle, self._MODES[mode]), "ldpc_decoder_set_mode")
        with self._ws_lock:
            self._ws.clear()
        return self

    def info(self) -> dict:
        out = np.zeros(4, dtype=np.int32)
        nat.check(self._lib.ldpc_decoder_info(self.handle, nat.ptr(out)), "ldpc_decoder_info")
        return {"engine": {2: "resident", 3: "stream", 4: "stream", 5: "stream"}[int(out[0])],
                "stream_form": {2: None, 3: "two-sweeps", 4: "fused-rcq-iteration", 5: "rcq-code-pair"}[int(out[0])],
                "codewords_per_workgroup": int(out[1]),
                "threads_per_workgroup": int(out[2]), "lds_bytes": int(out[3])}

    # ------------------------------------------------------------------ weights
    def set_weights(self, beta: Optional[np.ndarray], alpha: Optional[np.ndarray],
                    oms_alpha: Optional[np.ndarray] = None):
        """Re-upload weight tables (same shapes) -- e.g. after loading a state_dict."""
        if beta is not None:
            beta = np.ascontiguousarray(beta, dtype=self.np_dtype).reshape(self._table_shapes[0])
        if alpha is not None:
            alpha = np.ascontiguousarray(alpha, dtype=self.np_dtype).reshape(self._table_shapes[1])
        if oms_alpha is not None:
            oms_alpha = np.ascontiguousarray(oms_alpha, dtype=self.np_dtype).reshape(self._table_shapes[2])
        with torch.cuda.device(self.device):
            stream = torch.cuda.current_stream(self.device)
            nat.check(self._lib.ldpc_decoder_set_weights(self.handle, nat.ptr(beta), nat.ptr(alpha),
                                                         nat.ptr(oms_alpha), C.c_void_p(stream.cuda_stream)),
                      "ldpc_decoder_set_weights")
            stream.synchronize()     # pageable host arrays: make the upload complete before they die
        if beta is not None:
            self._tables[0] = beta.copy()
        if alpha is not None:
            self._tables[1] = alpha.copy()
        if oms_alpha is not None:
            self._tables[2] = oms_alpha.copy()

    def current_tables(self):
        """(beta [T, Sb], alpha [T, Sa], oms_alpha [T, So] | None) numpy copies of the tables the device holds"""
        return tuple(self._tables)

    # ------------------------------------------------------------------ decode
    def workspace_bytes(self, batch: int) -> int:
        return int(self._lib.ldpc_decoder_workspace_bytes(self.handle, int(batch)))

    _WS_MAX = 4          # workspaces kept (one per recently used stream); a multi-GB buffer each on the big codes

    def _workspace(self, batch: int) -> torch.Tensor:
        """the current stream's scratch buffer (grown on demand, never shared between streams).  The cache is a small
        LRU: a caller cycling through short-lived torch streams does not pile up one buffer per stream -- an evicted buffer
        goes back to torch's caching allocator, which keeps it alive for the work already queued on its stream."""
        need = self.workspace_bytes(batch)
        key = torch.cuda.current_stream(self.device).cuda_stream
        with self._ws_lock:
            ws = self._ws.pop(key, None)
            if ws is None or ws.numel() < need:
                ws = None
                ws = torch.empty(need, dtype=torch.uint8, device=self.device)
            self._ws[key] = ws                     # most recently used last
            while len(self._ws) > self._WS_MAX:
                self._ws.popitem(last=False)
            return ws

    def decode(self, llr: torch.Tensor, *, early_stop: bool = True, want_bits: bool = True,
               want_posterior: bool = True, want_packed: bool = False, max_iters: Optional[int] = None) -> DecodeResult:
        """llr: [B, n] tensor on this engine's GPU, dtype == engine dtype.  max_iters: run at most that many of the decoder's
        iterations (ldpc_decode_capped; a codeword open at the cap reports iterations = cap, success = False)."""
        if llr.device != self.device:
            raise ValueError(f"llr is on {llr.device}, engine on {self.device}")
        if llr.dtype != self.dtype:
            raise TypeError(f"llr dtype {llr.dtype} != engine dtype {self.dtype}")
        if llr.dim() != 2 or llr.shape[1] != self.graph.n:
            raise ValueError(f"llr must have shape [B, {self.graph.n}], got {tuple(llr.shape)}")
        llr = llr.contiguous()
        B, n = llr.shape
        dev = self.device
        bits = torch.empty((B, n), dtype=torch.int32, device=dev) if want_bits else None
        post = torch.empty((B, n), dtype=self.dtype, device=dev) if want_posterior else None
        iters = torch.empty((B,), dtype=torch.int32, device=dev)
        succ = torch.empty((B,), dtype=torch.uint8, device=dev)
        packed = torch.empty((B, (n + 7) // 8), dtype=torch.uint8, device=dev) if want_packed else None
        if B > 0:
            ws = self._workspace(B)
            with torch.cuda.device(dev):
                stream = torch.cuda.current_stream(dev).cuda_stream
                p = lambda t: None if t is None else C.c_void_p(t.data_ptr())
                if max_iters is None:
                    nat.check(self._lib.ldpc_decode(self.handle, p(llr), B, int(bool(early_stop)), p(bits), p(post),
                                                    p(iters), p(succ), p(packed), p(ws), ws.numel(),
                                                    C.c_void_p(stream)), "ldpc_decode")
                else:
                    nat.check(self._lib.ldpc_decode_capped(self.handle, p(llr), B, int(bool(early_stop)), int(max_iters),
                                                           p(bits), p(post), p(iters), p(succ), p(packed), p(ws),
                                                           ws.numel(), C.c_void_p(stream)), "ldpc_decode_capped")
        return DecodeResult(bits, post, iters, succ.bool(), packed)

    # ------------------------------------------------------------------ small host batches (the reference's call shape)
    HOST_BATCH_MAX = 64

    def decode_host(self, llr_host: torch.Tensor, *, early_stop: bool = True, want_posterior: bool = True):
        """Decode a SMALL batch that lives in host memory (the reference's own call: one CPU vector in, CPU results out)
        with one staged copy each way: pinned host buffers and device buffers are kept per engine, the LLRs go up in one
        async copy, ldpc_decode writes bits / posterior / iterations / success into ONE device block laid out for THIS
        batch size (a one-codeword call moves ~2n words back, not the 64-row block), that block comes back in one async
        copy, one stream synchronise.  -> (bits int32 [B, n], posterior | None, iterations int32 [B], success bool [B])
        as CPU tensors (fresh copies).  The staging buffers have their own lock (decode() on other threads is not
        blocked while this call waits for the device)."""
        if llr_host.dim() != 2 or llr_host.shape[1] != self.graph.n or llr_host.shape[0] > self.HOST_BATCH_MAX:
            raise ValueError(f"decode_host takes [B <= {self.HOST_BATCH_MAX}, {self.graph.n}] host tensors")
        B, n = llr_host.shape
        es = 4 if self.dtype == torch.float32 else 8
        with self._host_lock:
            st = getattr(self, "_host_stage", None)
            if st is None:
                Bm = self.HOST_BATCH_MAX
                total = (Bm * n * 4 + Bm * n * es + Bm * 4 + Bm + 1024 + 255) // 256 * 256
                st = {"h_in": torch.empty((Bm, n), dtype=self.dtype, pin_memory=True),
                      "d_in": torch.empty((Bm, n), dtype=self.dtype, device=self.device),
                      "d_out": torch.empty(total, dtype=torch.uint8, device=self.device),
                      "h_out": torch.empty(total, dtype=torch.uint8, pin_memory=True), "ws": None}
                self._host_stage = st
            # block layout for B rows: bits | posterior | iterations | success, each 256-byte aligned (posterior: its dtype)
            al = lambda x: (x + 255) // 256 * 256
            o_bits = 0
            o_post = al(B * n * 4)
            o_it = o_post + (al(B * n * es) if want_posterior else 0)
            o_ok = o_it + al(B * 4)
            used = o_ok + B
            st["h_in"][:B].copy_(llr_host)
            with torch.cuda.device(self.device):
                stream = torch.cuda.current_stream(self.device)
                st["d_in"][:B].copy_(st["h_in"][:B], non_blocking=True)
                need = self.workspace_bytes(B)
                if st["ws"] is None or st["ws"].numel() < need:
                    st["ws"] = torch.empty(need, dtype=torch.uint8, device=self.device)
                base = st["d_out"].data_ptr()
                nat.check(self._lib.ldpc_decode(self.handle, C.c_void_p(st["d_in"].data_ptr()), B, int(bool(early_stop)),
                                                C.c_void_p(base + o_bits), C.c_void_p(base + o_post) if want_posterior else None,
                                                C.c_void_p(base + o_it), C.c_void_p(base + o_ok), None,
                                                C.c_void_p(st["ws"].data_ptr()), st["ws"].numel(),
                                                C.c_void_p(stream.cuda_stream)), "ldpc_decode")
                st["h_out"][:used].copy_(st["d_out"][:used], non_blocking=True)
                stream.synchronize()
            h = st["h_out"]
            bits = h[o_bits:o_bits + B * n * 4].view(torch.int32).view(B, n).clone()
            post = h[o_post:o_post + B * n * es].view(self.dtype).view(B, n).clone() if want_posterior else None
            iters = h[o_it:o_it + B * 4].view(torch.int32).clone()
            succ = h[o_ok:o_ok + B].clone().bool()
        return bits, post, iters, succ

    def decode_host_op(self, llr_host: torch.Tensor, *, early_stop: bool = True, want_posterior: bool = True) -> DecodeResult:
        """decode_host() through ``torch.ops.ldpc.decode_host`` -> DecodeResult of CPU tensors"""
        import torch_ops
        bits, post, iters, succ = torch.ops.ldpc.decode_host(llr_host, torch_ops.engine_handle(self), bool(early_stop),
                                                             bool(want_posterior))
        return DecodeResult(bits, post if want_posterior else None, iters, succ, None)

    def decode_op(self, llr: torch.Tensor, *, early_stop: bool = True, want_posterior: bool = True,
                  want_packed: bool = False) -> DecodeResult:
        """decode() entered through the registered PyTorch operator ``torch.ops.ldpc.decode`` (torch_ops.py) --
        the call the host decoder classes make"""
        import torch_ops
        bits, post, iters, succ, packed = torch.ops.ldpc.decode(llr, torch_ops.engine_handle(self), bool(early_stop),
                                                                bool(want_posterior), bool(want_packed))
        return DecodeResult(bits, post if want_posterior else None, iters, succ, packed if want_packed else None)

    # ------------------------------------------------------------------ gradients (training path)
    def _check_llr(self, llr: torch.Tensor) -> torch.Tensor:
        if llr.device != self.device:
            raise ValueError(f"llr is on {llr.device}, engine on {self.device}")
        if llr.dtype != self.dtype:
            raise TypeError(f"llr dtype {llr.dtype} != engine dtype {self.dtype}")
        if llr.dim() != 2 or llr.shape[1] != self.graph.n:
            raise ValueError(f"llr must have shape [B, {self.graph.n}], got {tuple(llr.shape)}")
        return llr.contiguous()

    def _train_workspace(self, batch: int) -> torch.Tensor:
        need = int(self._lib.ldpc_train_workspace_bytes(self.handle, batch))
        ws = getattr(self, "_train_ws", None)
        if ws is None or ws.numel() < need:
            self._train_ws = None
            self._train_ws = torch.empty(need, dtype=torch.uint8, device=self.device)
        return self._train_ws

    def train_saved_bytes(self, batch: int) -> int:
        return int(self._lib.ldpc_train_saved_bytes(self.handle, max(int(batch), 1)))

    def decode_saving(self, llr: torch.Tensor, *, early_stop: bool = True):
        """decode() that also keeps every iteration's messages for backward(); returns (DecodeResult, saved).
        fp32 normalised min-sum decoders only (NotImplementedError otherwise)."""
        llr = self._check_llr(llr)
        B, n = llr.shape
        dev = self.device
        bits = torch.empty((B, n), dtype=torch.int32, device=dev)
        post = torch.empty((B, n), dtype=self.dtype, device=dev)
        iters = torch.empty((B,), dtype=torch.int32, device=dev)
        succ = torch.empty((B,), dtype=torch.uint8, device=dev)
        # the saved messages become a saved tensor of the autograd node (torch_ops.py): allocated per call, freed with the graph
        saved = torch.empty(int(self._lib.ldpc_train_saved_bytes(self.handle, max(B, 1))), dtype=torch.uint8, device=dev)
        if B > 0:
            ws = self._train_workspace(B)
            with torch.cuda.device(dev):
                stream = torch.cuda.current_stream(dev).cuda_stream
                p = lambda t: C.c_void_p(t.data_ptr())
                nat.check(self._lib.ldpc_decode_saving(self.handle, p(llr), B, int(bool(early_stop)), p(bits), p(post),
                                                       p(iters), p(succ), p(saved), saved.numel(), p(ws), ws.numel(),
                                                       C.c_void_p(stream)), "ldpc_decode_saving")
        return DecodeResult(bits, post, iters, succ.bool(), None), saved

    def backward(self, saved: torch.Tensor, llr: torch.Tensor, iterations: torch.Tensor, grad_posterior: torch.Tensor,
                 want_grad_llr: bool = False):
        """(d loss/d beta [T, beta slots], d loss/d alpha [T, alpha slots], d loss/d oms_alpha [T, slots] | None)
        for a loss with d loss/d posterior = grad_posterior [B, n]; `saved`, `iterations` from decode_saving of the
        same llr with the same weight tables.  want_grad_llr: also return d loss/d llr [B, n]."""
        llr = self._check_llr(llr)
        B, n = llr.shape
        dev = self.device
        gp = grad_posterior.to(device=dev, dtype=torch.float32).contiguous()
        if gp.shape != llr.shape:
            raise ValueError(f"grad_posterior must have shape {tuple(llr.shape)}, got {tuple(gp.shape)}")
        iterations = iterations.to(device=dev, dtype=torch.int32).contiguous()
        if iterations.shape != (B,):
            raise ValueError("iterations must have one entry per codeword")
        gb = torch.zeros(self._table_shapes[0], dtype=torch.float32, device=dev)
        ga = torch.zeros(self._table_shapes[1], dtype=torch.float32, device=dev)
        goa = None if self._table_shapes[2] is None else torch.zeros(self._table_shapes[2], dtype=torch.float32, device=dev)
        gl = torch.zeros((B, n), dtype=torch.float32, device=dev) if want_grad_llr else None
        if B > 0:
            ws = self._train_workspace(B)
            with torch.cuda.device(dev):
                stream = torch.cuda.current_stream(dev).cuda_stream
                p = lambda t: None if t is None else C.c_void_p(t.data_ptr())
                nat.check(self._lib.ldpc_backward(self.handle, p(saved), saved.numel(), p(llr), B, p(iterations), p(gp),
                                                  p(gb), p(ga), p(goa), p(gl), p(ws), ws.numel(), C.c_void_p(stream)),
                          "ldpc_backward")
        if want_grad_llr:
            return gb, ga, goa, gl
        return gb, ga, goa

    def debug_sweep(self, batch: int, which: int, it: int):
        """Launch one CN (which=0) or VN (which=1) sweep on the state a previous
        decode(batch) left in the workspace -- bench.py's per-kernel timing hook."""
        ws = self._workspace(batch)
        with torch.cuda.device(self.device):
            stream = torch.cuda.current_stream(self.device).cuda_stream
            nat.check(self._lib.ldpc_debug_sweep(self.handle, int(batch), int(which), int(it),
                                                 C.c_void_p(ws.data_ptr()), ws.numel(), C.c_void_p(stream)),
                      "ldpc_debug_sweep")

    def debug_resident_c2v(self, llr: torch.Tensor, *, early_stop: bool = True):
        """LDS-resident engine: decode `llr` and also return every codeword's C2V messages of its last executed
        iteration -> (c2v [B, E] dtype values in CSR edge order, posterior [B, n], iterations [B]).  Test hook
        (include/ldpc_hip_debug.h); RCQ decoders hold reconstructed values, see `rcq_codes_from_values`."""
        llr = self._check_llr(llr)
        B, n = llr.shape
        dev = self.device
        c2v = torch.zeros((B, self.graph.E), dtype=self.dtype, device=dev)
        post = torch.empty((B, n), dtype=self.dtype, device=dev)
        iters = torch.empty((B,), dtype=torch.int32, device=dev)
        with torch.cuda.device(dev):
            stream = torch.cuda.current_stream(dev).cuda_stream
            p = lambda t: C.c_void_p(t.data_ptr())
            nat.check(self._lib.ldpc_debug_resident_c2v(self.handle, p(llr), B, int(bool(early_stop)), p(post), p(iters),
                                                        p(c2v), C.c_void_p(stream)), "ldpc_debug_resident_c2v")
        return c2v, post, iters

    def debug_c2v(self, batch: int) -> torch.Tensor:
        """Streaming engine: C2V state left by the last decode(batch): [B, E] uint8 quantiser codes (RCQ) or
        dtype values, CSR edge order.  Test hook (per-edge code parity with the reference)."""
        out8 = np.zeros(8, dtype=np.int64)
        nat.check(self._lib.ldpc_debug_workspace_layout(self.handle, int(batch), nat.ptr(out8)),
                  "ldpc_debug_workspace_layout")
        vec, tiles, off = int(out8[0]), int(out8[1]), int(out8[4])
        W, E = 64 * vec, self.graph.E
        ws = self._workspace(batch)
        if self.c2v_form == nat.C2V_RCQ:
            raw = ws[off: off + tiles * E * W].view(tiles, E, W)
        else:
            es = 4 if self.dtype == torch.float32 else 8
            raw = ws[off: off + tiles * E * W * es].view(self.dtype).view(tiles, E, W)
        return raw.permute(0, 2, 1).reshape(tiles * W, E)[:batch].contiguous()
