import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "implementation-of-neural-ldpc-decoders-with-degree-specific-weight-sharing-and-rcq-quantization_amd")
GOLDEN = os.path.join(ROOT, "tests", "golden")
# flat module layout of the reference: the package directory itself goes on sys.path
for p in (PKG, os.path.join(ROOT, "oracle"), ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)
os.environ.setdefault("MPLBACKEND", "Agg")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: longer-running CPU test")


def load_golden(name):
    """npz fixture written by oracle/make_golden.py from the real reference"""
    path = os.path.join(GOLDEN, name + ".npz")
    with np.load(path, allow_pickle=False) as z:
        return {k: z[k] for k in z.files}


def golden_sub(d, prefix):
    """keys 'prefix_xxx' -> {'xxx': value}"""
    pl = len(prefix) + 1
    return {k[pl:]: v for k, v in d.items() if k.startswith(prefix + "_")}


def weights_dict(keys, vals):
    return {str(k): float(v) for k, v in zip(keys, vals)}


@pytest.fixture(scope="session")
def oracle_mod():
    import oracle  # oracle/oracle.py
    oracle.build()
    return oracle


@pytest.fixture(scope="session")
def gpu_device():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda", 0)
