"""
Deterministic sparse LDPC code construction for the benchmark configurations.

The reference holds no usable matrix for the sizes BASELINE.json names: its only
real code is the 4x7 toy of ``create_test_ldpc_code`` (ldpc_decoder.py:274-284)
and ``create_dvbs2_code`` (training_framework.py:379-400) is a *dense random*
9000x16200 matrix (~7.3e7 edges).  SURVEY.md section 8(d) therefore specifies
builder-generated IRA-style codes:

* ``ira_1998_1512``  : n=1998, m=486, variable degrees {8:216, 3:1296, 2:485, 1:1}
* ``dvbs2_like_16200_7200`` : n=16200, m=9000, E=48599, node-perspective degree
  profile of the paper's Table II: VN {8:1800, 3:5400, 2:8999, 1:1},
  CN {4:1441, 5:3239, 6:3600, 7:720}
* ``small_96_48`` : n=96, m=48 test code with variable degrees {1,2,3,8}

Generated edge lists are committed under ``data/`` so every run (here and on
the GPU box) decodes the very same graph; ``generate_ira_code`` is the script
that made them (``python codes.py`` regenerates and verifies the files).

Construction: the last ``m`` columns are the IRA staircase (parity column p
touches checks p and p+1; the last one only check m-1).  Information columns
take their edges from a shuffled multiset of check "sockets" sized so every
check ends at its target degree; duplicate sockets inside one column and
4-cycles are repaired by socket swaps.
"""

from __future__ import annotations

import os
from typing import Dict, Optional, Sequence

import numpy as np

from tanner_graph import TannerGraph

_DATA_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data")


def generate_ira_code(n: int, m: int, info_degrees: Dict[int, int],
                      check_degrees: Optional[Dict[int, int]] = None,
                      seed: int = 0, max_repair_sweeps: int = 200) -> TannerGraph:
    """Build an IRA-style irregular code.

    info_degrees  {dv: number of information columns with that degree}
    check_degrees {dc: number of checks with that (total) degree}; ``None``
                  spreads the edges as evenly as possible.
    """
    k = n - m
    if sum(info_degrees.values()) != k:
        raise ValueError("info_degrees must cover exactly n-m columns")
    rng = np.random.default_rng(seed)

    # parity staircase
    par_rows = [np.arange(m, dtype=np.int64)]            # p -> check p
    par_cols = [k + np.arange(m, dtype=np.int64)]
    par_rows.append(np.arange(1, m, dtype=np.int64))     # p -> check p+1
    par_cols.append(k + np.arange(m - 1, dtype=np.int64))
    par_deg = np.full(m, 2, dtype=np.int64)
    par_deg[0] = 1

    col_deg = np.concatenate([np.full(cnt, dv, dtype=np.int64)
                              for dv, cnt in sorted(info_degrees.items(), reverse=True)])
    # interleave degrees over the column index range so that degree classes are
    # not contiguous blocks (keeps the VN sweep's gather pattern realistic)
    col_deg = col_deg[rng.permutation(k)]
    e_info = int(col_deg.sum())

    if check_degrees is None:
        total = e_info + int(par_deg.sum())
        base, extra = divmod(total, m)
        target = np.full(m, base, dtype=np.int64)
        target[rng.permutation(m)[:extra]] += 1
    else:
        if sum(check_degrees.values()) != m:
            raise ValueError("check_degrees must cover exactly m checks")
        target = np.concatenate([np.full(cnt, dc, dtype=np.int64)
                                 for dc, cnt in sorted(check_degrees.items())])
        target = target[rng.permutation(m)]
        # check 0 has one staircase edge only: give it the smallest target left so
        # capacities stay positive
    cap = target - par_deg
    if cap.min() < 0 or int(cap.sum()) != e_info:
        raise ValueError(f"degree profiles inconsistent: info edges {e_info}, capacity {int(cap.sum())}")

    sockets = np.repeat(np.arange(m, dtype=np.int64), cap)
    rng.shuffle(sockets)
    col_of_socket = np.repeat(np.arange(k, dtype=np.int64), col_deg)
    col_start = np.zeros(k + 1, dtype=np.int64)
    np.cumsum(col_deg, out=col_start[1:])

    def col_checks(c):
        return sockets[col_start[c]:col_start[c + 1]]

    # check -> set of info columns (for 4-cycle detection)
    def build_members():
        mem = [set() for _ in range(m)]
        for s in range(e_info):
            mem[sockets[s]].add(int(col_of_socket[s]))
        return mem

    def bad_sockets(members):
        """socket positions that are duplicates inside their column, or that
        close a 4-cycle with another information column or the staircase."""
        bad = []
        for c in range(k):
            chk = col_checks(c)
            seen = {}
            for t, i in enumerate(chk):
                i = int(i)
                if i in seen:
                    bad.append(col_start[c] + t)
                seen[i] = t
            # staircase 4-cycle: column touches checks i and i+1 -> shares two
            # checks with parity column i
            sset = set(int(x) for x in chk)
            for t, i in enumerate(chk):
                if int(i) + 1 in sset:
                    bad.append(col_start[c] + t)
            # info-info 4-cycles
            other = {}
            for t, i in enumerate(chk):
                for c2 in members[int(i)]:
                    if c2 == c:
                        continue
                    if c2 in other:
                        bad.append(col_start[c] + t)
                    else:
                        other[c2] = t
        return sorted(set(int(b) for b in bad))

    for sweep in range(max_repair_sweeps):
        members = build_members()
        bad = bad_sockets(members)
        if not bad:
            break
        # swap every offending socket with a uniformly random one
        for s in bad:
            t = int(rng.integers(0, e_info))
            sockets[s], sockets[t] = sockets[t], sockets[s]
    # duplicates are fatal, 4-cycles only undesirable: a final duplicate-only repair
    for _ in range(1000):
        dup = []
        for c in range(k):
            chk = col_checks(c)
            if len(set(int(x) for x in chk)) != len(chk):
                u, idx = np.unique(chk, return_index=True)
                mask = np.ones(len(chk), dtype=bool)
                mask[idx] = False
                dup.extend((col_start[c] + np.nonzero(mask)[0]).tolist())
        if not dup:
            break
        for s in dup:
            t = int(rng.integers(0, e_info))
            sockets[s], sockets[t] = sockets[t], sockets[s]
    else:
        raise RuntimeError("could not remove duplicate edges")

    rows = np.concatenate([sockets] + par_rows)
    cols = np.concatenate([col_of_socket] + par_cols)
    return TannerGraph(n, m, rows, cols)


# ---------------------------------------------------------------------------- named codes
_SPECS = {
    "ira_1998_1512": dict(n=1998, m=486, info_degrees={8: 216, 3: 1296}, check_degrees=None, seed=1998),
    "dvbs2_like_16200_7200": dict(n=16200, m=9000, info_degrees={8: 1800, 3: 5400},
                                  check_degrees={4: 1441, 5: 3239, 6: 3600, 7: 720}, seed=16200),
    "small_96_48": dict(n=96, m=48, info_degrees={8: 8, 3: 40}, check_degrees=None, seed=96),
}


def code_names() -> Sequence[str]:
    return tuple(_SPECS)


def _path(name: str) -> str:
    return os.path.join(_DATA_DIR, name + ".npz")


def save_graph(path: str, g: TannerGraph) -> None:
    np.savez_compressed(path, n=np.int32(g.n), m=np.int32(g.m),
                        check_ptr=g.check_ptr, var_idx=g.var_idx.astype(np.uint16 if g.n <= 65535 else np.int32))


def load_graph(name_or_path: str) -> TannerGraph:
    """Load a committed edge list (``data/<name>.npz``)."""
    path = name_or_path if os.path.exists(name_or_path) else _path(name_or_path)
    with np.load(path, allow_pickle=False) as z:
        return TannerGraph.from_csr(int(z["n"]), z["check_ptr"], z["var_idx"].astype(np.int32))


def load_code(name: str, max_iterations: int = 50):
    """Named code as an ``LDPCCode`` (dense ``H`` materialised lazily as int8)."""
    from ldpc_decoder import LDPCCode
    g = load_graph(name)
    return LDPCCode.from_graph(g, k=g.n - g.m, max_iterations=max_iterations)


if __name__ == "__main__":  # regenerate + verify the committed edge lists
    os.makedirs(_DATA_DIR, exist_ok=True)
    for name, spec in _SPECS.items():
        g = generate_ira_code(**spec)
        dvs = dict(zip(*np.unique(g.dv, return_counts=True)))
        dcs = dict(zip(*np.unique(g.dc, return_counts=True)))
        print(name, "n", g.n, "m", g.m, "E", g.E, "dv", dvs, "dc", dcs,
              "4-cycles", g.four_cycles() if g.E < 10000 else "n/a")
        p = _path(name)
        if os.path.exists(p):
            old = load_graph(p)
            same = np.array_equal(old.check_ptr, g.check_ptr) and np.array_equal(old.var_idx, g.var_idx)
            print("   committed file", "matches" if same else "DIFFERS (not overwritten)")
        else:
            save_graph(p, g)
            print("   wrote", p)
