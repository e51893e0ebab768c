"""
Degree-based weight sharing: reference-style ParameterDicts  <->  flat device tables.

The reference keeps one shape-[1] ``nn.Parameter`` per (iteration, degree class) in
``beta_weights`` / ``alpha_weights`` under keys

    type 1: beta  iter_{t}_dc{dc}_dv{dv}
    type 2: beta  iter_{t}_dc{dc}      alpha iter_{t}_dv{dv}
    type 3: beta  iter_{t}_dc{dc}
    type 4:                            alpha iter_{t}_dv{dv}

(neural_2d_decoder.py:46-82, rcq_decoder.py:398-431) and looks them up per edge with
``_get_beta_weight`` / ``_get_alpha_weight`` (neural_2d_decoder.py:84-131), falling back
to a constant when a type has no such weight.  The engine wants ``[T][slots]`` tables
plus a slot index per edge / per variable; this module builds both, keeping the
reference's key names so ``state_dict``s are interchangeable.
"""

from __future__ import annotations

from typing import Dict, List, Optional, Tuple

import numpy as np
import torch
import torch.nn as nn

from tanner_graph import TannerGraph


def unique_degrees(code) -> Tuple[List[int], List[int]]:
    """``list(set(code.<...>_degrees.values()))`` exactly as the reference builds them
    (neural_2d_decoder.py:34-35): the iteration order of that list fixes the order in
    which parameters are created, hence which ``torch.randn`` draw each one gets."""
    return (list(set(code.check_node_degrees.values())),
            list(set(code.variable_node_degrees.values())))


def init_parameter_dicts(weight_sharing_type: int, max_iterations: int, check_degrees: List[int],
                         variable_degrees: List[int], strict: bool = True):
    """Create beta/alpha ParameterDicts in the reference's order with its init
    (``torch.randn(1) * 0.1``), so equal seeds give equal weights."""
    beta, alpha = nn.ParameterDict(), nn.ParameterDict()
    if weight_sharing_type == 1:
        for t in range(max_iterations):
            for dc in check_degrees:
                for dv in variable_degrees:
                    beta[f"iter_{t}_dc{dc}_dv{dv}"] = nn.Parameter(torch.randn(1) * 0.1)
    elif weight_sharing_type == 2:
        for t in range(max_iterations):
            for dc in check_degrees:
                beta[f"iter_{t}_dc{dc}"] = nn.Parameter(torch.randn(1) * 0.1)
            for dv in variable_degrees:
                alpha[f"iter_{t}_dv{dv}"] = nn.Parameter(torch.randn(1) * 0.1)
    elif weight_sharing_type == 3:
        for t in range(max_iterations):
            for dc in check_degrees:
                beta[f"iter_{t}_dc{dc}"] = nn.Parameter(torch.randn(1) * 0.1)
    elif weight_sharing_type == 4:
        for t in range(max_iterations):
            for dv in variable_degrees:
                alpha[f"iter_{t}_dv{dv}"] = nn.Parameter(torch.randn(1) * 0.1)
    elif strict:
        raise ValueError(f"Invalid weight sharing type: {weight_sharing_type}")
    return beta, alpha


class SharingLayout:
    """Slot maps of one (graph, sharing type).

    ``beta_suffix[s]`` / ``alpha_suffix[s]`` is the key suffix (after ``iter_{t}_``) of
    table column ``s``, or ``None`` for the constant column of a type without that
    weight.  ``beta_slot`` is per CSR edge, ``alpha_slot`` per variable; for the offset
    decoders (alpha applied in the check update) ``alpha_edge_slot`` is per CSR edge.
    """

    def __init__(self, graph: TannerGraph, weight_sharing_type: int):
        self.type = int(weight_sharing_type)
        dc_e = graph.dc[graph.check_of_edge].astype(np.int64)
        dv_e = graph.dv[graph.var_idx].astype(np.int64)
        if self.type == 1:
            pairs = sorted(set(zip(dc_e.tolist(), dv_e.tolist())))
            index = {p: s for s, p in enumerate(pairs)}
            self.beta_slot = np.fromiter((index[p] for p in zip(dc_e.tolist(), dv_e.tolist())),
                                         dtype=np.int32, count=graph.E)
            self.beta_suffix: List[Optional[str]] = [f"dc{a}_dv{b}" for a, b in pairs]
        elif self.type in (2, 3):
            dcs = sorted(set(dc_e.tolist()))
            index = {d: s for s, d in enumerate(dcs)}
            self.beta_slot = np.fromiter((index[d] for d in dc_e.tolist()), dtype=np.int32, count=graph.E)
            self.beta_suffix = [f"dc{d}" for d in dcs]
        else:
            self.beta_slot = np.zeros(graph.E, dtype=np.int32)
            self.beta_suffix = [None]
        if not self.beta_suffix:               # graph without edges
            self.beta_suffix = [None]
        if self.type in (2, 4):
            dvs = sorted(set(graph.dv.tolist()))
            index = {d: s for s, d in enumerate(dvs)}
            self.alpha_slot = np.fromiter((index[d] for d in graph.dv.tolist()), dtype=np.int32, count=graph.n)
            self.alpha_suffix: List[Optional[str]] = [f"dv{d}" for d in dvs]
            self.alpha_edge_slot = self.alpha_slot[graph.var_idx].astype(np.int32)
        else:
            self.alpha_slot = np.zeros(graph.n, dtype=np.int32)
            self.alpha_suffix = [None]
            self.alpha_edge_slot = np.zeros(graph.E, dtype=np.int32)

    def _table(self, params, suffixes, T: int, default: float) -> np.ndarray:
        """[max(T,1), len(suffixes)] float32.  WHICH parameter feeds which cell is resolved once per parameter set
        (the key lookups dominate a one-codeword call otherwise); the VALUES are re-read on every call, so in-place
        updates of any kind (optimizer, fill_, .data) are always seen."""
        rows = max(T, 1)
        out = np.full((rows, len(suffixes)), default, dtype=np.float32)
        ids = tuple(map(id, params.values()))
        cache = self.__dict__.setdefault("_cells", {})
        key = (id(params), tuple(suffixes), T)
        hit = cache.get(key)
        if hit is None or hit[0] != ids:
            picks, tt, ss = [], [], []
            for t in range(T):
                for s, suf in enumerate(suffixes):
                    if suf is None:
                        continue
                    k = f"iter_{t}_{suf}"
                    if k in params:               # reference: ParameterDict.get(key, default)
                        picks.append(params[k])
                        tt.append(t)
                        ss.append(s)
            hit = (ids, picks, np.asarray(tt, dtype=np.intp), np.asarray(ss, dtype=np.intp))
            cache[key] = hit
        _, picks, tt, ss = hit
        if picks:
            with torch.no_grad():
                try:                                          # the usual case: every parameter has shape [1]
                    flat = torch.cat(picks)
                    if flat.numel() != len(picks):
                        raise RuntimeError("parameter with more than one element")
                except RuntimeError:
                    flat = torch.cat([p.reshape(-1)[:1] for p in picks])
                out[tt, ss] = flat.to(dtype=torch.float32, device="cpu").numpy()
        return out

    @staticmethod
    def _table_torch(params, suffixes, T: int, default: float) -> torch.Tensor:
        """the same table as a differentiable function of the parameters (gradient path)"""
        from autograd_bridge import table_from_params
        picks, where = [], []
        for t in range(T):
            for s, suf in enumerate(suffixes):
                if suf is not None and f"iter_{t}_{suf}" in params:
                    picks.append(params[f"iter_{t}_{suf}"])
                    where.append((t, s))
        return table_from_params(picks, where, (max(T, 1), len(suffixes)), default)

    def tables_torch(self, beta_params, alpha_params, T: int, beta_default: float, alpha_default: float):
        return (self._table_torch(beta_params, self.beta_suffix, T, beta_default),
                self._table_torch(alpha_params, self.alpha_suffix, T, alpha_default))

    def tables(self, beta_params, alpha_params, T: int, beta_default: float, alpha_default: float):
        """-> (beta[T, Sb], alpha[T, Sa]) float32, one device->host copy per dict."""
        return (self._table(beta_params, self.beta_suffix, T, beta_default),
                self._table(alpha_params, self.alpha_suffix, T, alpha_default))
