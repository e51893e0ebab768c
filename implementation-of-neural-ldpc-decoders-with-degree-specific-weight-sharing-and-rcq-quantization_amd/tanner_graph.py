"""
Tanner-graph compiler: dense parity-check matrix  ->  CSR / CSC edge lists.

The reference never builds a graph: every decoder re-scans the dense matrix
with ``np.where(H[i, :] == 1)`` / ``np.where(H[:, j] == 1)`` for every node in
every iteration (ldpc_decoder.py:92,124; neural_2d_decoder.py:162,195;
rcq_decoder.py:212,250).  Those two scans define the only thing the engine
has to preserve -- the *neighbour order*:

* CSR  (check-major): check ``i`` ascending, inside a check variable ``j``
  ascending.  This is the order in which C2V messages are produced and the
  order ``argmin`` sees the magnitudes.
* CSC  (variable-major): variable ``j`` ascending, inside a variable check
  ``i`` ascending.  This is the order of the operands of the leave-one-out
  sums and of the posterior sum, which fixes their floating-point association.

Edge ids are CSR positions.  ``csc_edge[var_ptr[j] + k]`` is the CSR edge id of
the k-th (ascending check index) neighbour of variable ``j``.
"""

from __future__ import annotations

import numpy as np


class TannerGraph:
    """Immutable CSR+CSC view of a binary parity-check matrix (host, int32)."""

    __slots__ = ("n", "m", "E", "check_ptr", "var_idx", "var_ptr", "csc_edge",
                 "check_of_edge", "dc", "dv", "__weakref__")

    def __init__(self, n: int, m: int, rows: np.ndarray, cols: np.ndarray):
        rows = np.asarray(rows, dtype=np.int64).ravel()
        cols = np.asarray(cols, dtype=np.int64).ravel()
        if rows.shape != cols.shape:
            raise ValueError("rows/cols length mismatch")
        if rows.size and (rows.min() < 0 or rows.max() >= m or cols.min() < 0 or cols.max() >= n):
            raise ValueError("edge endpoint out of range")
        # CSR order: (row, col) lexicographic
        key = rows * n + cols
        order = np.argsort(key, kind="stable")
        key = key[order]
        if key.size > 1 and np.any(key[1:] == key[:-1]):
            raise ValueError("duplicate edge in edge list")
        rows = rows[order]
        cols = cols[order]
        E = int(rows.size)
        if E >= 2**31 - 1:
            raise ValueError("too many edges for int32 edge ids")
        self.n, self.m, self.E = int(n), int(m), E
        self.dc = np.bincount(rows, minlength=m).astype(np.int32)
        self.dv = np.bincount(cols, minlength=n).astype(np.int32)
        self.check_ptr = np.zeros(m + 1, dtype=np.int32)
        np.cumsum(self.dc, out=self.check_ptr[1:])
        self.var_ptr = np.zeros(n + 1, dtype=np.int32)
        np.cumsum(self.dv, out=self.var_ptr[1:])
        self.var_idx = cols.astype(np.int32)
        self.check_of_edge = rows.astype(np.int32)
        # CSC permutation: stable sort of CSR edges by column keeps rows ascending.
        self.csc_edge = np.argsort(cols, kind="stable").astype(np.int32)

    # ------------------------------------------------------------------ builders
    @classmethod
    def from_dense(cls, H) -> "TannerGraph":
        """Edges are the entries that compare equal to 1, exactly as the
        reference's ``H[i, :] == 1`` scans (ldpc_decoder.py:92)."""
        H = np.asarray(H)
        if H.ndim != 2:
            raise ValueError("H must be 2-D")
        rows, cols = np.nonzero(H == 1)
        return cls(H.shape[1], H.shape[0], rows, cols)

    @classmethod
    def from_csr(cls, n: int, check_ptr, var_idx) -> "TannerGraph":
        check_ptr = np.asarray(check_ptr, dtype=np.int64)
        m = check_ptr.size - 1
        rows = np.repeat(np.arange(m, dtype=np.int64), np.diff(check_ptr))
        return cls(n, m, rows, np.asarray(var_idx, dtype=np.int64))

    # ------------------------------------------------------------------ helpers
    def to_dense(self, dtype=np.int8) -> np.ndarray:
        H = np.zeros((self.m, self.n), dtype=dtype)
        H[self.check_of_edge, self.var_idx] = 1
        return H

    @property
    def max_dc(self) -> int:
        return int(self.dc.max()) if self.m else 0

    @property
    def max_dv(self) -> int:
        return int(self.dv.max()) if self.n else 0

    def check_degree_dict(self):
        """{check: degree}; same content as LDPCCode.check_node_degrees
        (ldpc_decoder.py:38-45)."""
        return {i: int(d) for i, d in enumerate(self.dc)}

    def variable_degree_dict(self):
        return {j: int(d) for j, d in enumerate(self.dv)}

    def syndrome(self, bits: np.ndarray) -> np.ndarray:
        """H @ bits mod 2 for bits of shape [..., n] (host helper for tests)."""
        bits = np.asarray(bits).astype(np.int64)
        contrib = bits[..., self.var_idx]
        cs = np.concatenate([np.zeros(bits.shape[:-1] + (1,), dtype=np.int64),
                             np.cumsum(contrib, axis=-1)], axis=-1)
        out = cs[..., self.check_ptr[1:]] - cs[..., self.check_ptr[:-1]]
        return out % 2

    def four_cycles(self) -> int:
        """Number of column pairs sharing >= 2 checks (diagnostic for generators)."""
        from collections import Counter
        cnt = Counter()
        for i in range(self.m):
            vs = self.var_idx[self.check_ptr[i]:self.check_ptr[i + 1]]
            for a in range(len(vs)):
                for b in range(a + 1, len(vs)):
                    cnt[(int(vs[a]), int(vs[b]))] += 1
        return sum(1 for v in cnt.values() if v >= 2)
