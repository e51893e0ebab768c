"""
ctypes binding of the C ABI in include/ldpc_hip.h (libldpc_hip.so, built from
csrc/ by build_native()).

There is NO CPU fallback: when the library is missing or no HIP device is
usable every decode raises ``NativeEngineError``.  The host classes only add
Python-side bookkeeping around these calls.
"""

from __future__ import annotations

import ctypes as C
import os
import subprocess
import threading
from typing import Optional

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_NAME = "libldpc_hip.so"
LIB_PATH = os.environ.get("LDPC_HIP_LIB") or os.path.join(_HERE, LIB_NAME)   # override: A/B kernel variants
CSRC = os.path.join(_HERE, "csrc")
HEADER = os.path.join(os.path.dirname(_HERE), "include", "ldpc_hip.h")
DEBUG_HEADER = os.path.join(os.path.dirname(_HERE), "include", "ldpc_hip_debug.h")

LDPC_F32, LDPC_F64 = 0, 1
C2V_NMS, C2V_RCQ, C2V_OMS = 0, 1, 2
MODE_AUTO, MODE_STREAM, MODE_RESIDENT, MODE_SWEEPS, MODE_GATHER, MODE_PAIR = 0, 1, 2, 3, 4, 5
SCHED_FLOODING, SCHED_LAYERED_REF, SCHED_LAYERED = 0, 1, 2

HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
               "-ffp-contract=off",       # the reference never fuses llr + alpha*sum
               "-Wall", "-Wextra", "-Wno-unused-parameter", "-Wno-pragma-once-outside-header"]


class NativeEngineError(RuntimeError):
    pass


SOURCES = ("ldpc_hip.hip", "ldpc_kernels.hip", "ldpc_resident.hip", "ldpc_train.hip", "ldpc_layered.hip")


def _source_files():
    return [os.path.join(CSRC, f) for f in SOURCES] + [HEADER, DEBUG_HEADER]


def source_hash(defines=()) -> str:
    """sha256 over the compile recipe (flags, defines) and the CONTENT of every source the library is built from.
    File times say nothing after a copy to another machine; this does."""
    import hashlib
    h = hashlib.sha256()
    h.update(("\0".join(HIPCC_FLAGS) + "\1" + "\0".join(sorted(defines))).encode())
    for path in _source_files():
        h.update(b"\2" + os.path.basename(path).encode() + b"\3")
        with open(path, "rb") as f:
            h.update(f.read())
    return h.hexdigest()


def built_hash(target: Optional[str] = None) -> Optional[str]:
    """the source hash EMBEDDED in a built library (ldpc_source_hash(); None: not a library of this project, or a build
    that predates the symbol).  Read through a private dlopen handle -- loading a stale library to ask what it is is harmless."""
    target = target or os.path.join(_HERE, LIB_NAME)
    try:
        lib = C.CDLL(target)
        fn = lib.ldpc_source_hash
        fn.restype = C.c_char_p
        fn.argtypes = []
        return fn().decode() or None
    except (OSError, AttributeError):
        return None


def is_stale(target: Optional[str] = None, defines=()) -> bool:
    target = target or os.path.join(_HERE, LIB_NAME)
    return not os.path.exists(target) or built_hash(target) != source_hash(defines)


def build_native(force: bool = False, verbose: bool = False, defines=(), out: Optional[str] = None) -> str:
    """hipcc cross-compile of csrc/ldpc_hip.hip for gfx950 into the package dir; skipped when the library on disk was
    built from exactly these sources with exactly this recipe (content hash embedded in the library, not file times).
    `defines`/`out` build tuning variants (tools/sweep_variants.py)."""
    srcs = _source_files()
    target = out or os.path.join(_HERE, LIB_NAME)
    want = source_hash(defines)
    if not force and os.path.exists(target) and built_hash(target) == want:
        return target
    hipcc = os.environ.get("HIPCC") or ("/opt/rocm/bin/hipcc" if os.path.exists("/opt/rocm/bin/hipcc") else "hipcc")
    tmp = target + ".tmp%d" % os.getpid()
    cmd = [hipcc] + HIPCC_FLAGS + [f"-D{d}" for d in defines] + [f"-DLDPC_SRC_HASH={want}", "-o", tmp, srcs[0]]
    res = subprocess.run(cmd, capture_output=True, text=True)
    if verbose or res.returncode != 0:
        print(" ".join(cmd))
        print(res.stdout + res.stderr)
    if res.returncode != 0:
        try:
            os.unlink(tmp)
        except OSError:
            pass
        raise NativeEngineError("hipcc failed:\n" + res.stderr[-4000:])
    os.replace(tmp, target)                      # never a half-written library under the real name
    return target


class DecoderDesc(C.Structure):
    _fields_ = [("dtype", C.c_int32), ("c2v_form", C.c_int32), ("iters", C.c_int32),
                ("n_beta_slots", C.c_int32), ("beta", C.c_void_p), ("beta_slot", C.c_void_p),
                ("n_alpha_slots", C.c_int32), ("alpha", C.c_void_p), ("alpha_slot", C.c_void_p),
                ("n_levels", C.c_int32), ("n_quantizers", C.c_int32), ("thresholds", C.c_void_p),
                ("q_of_iter", C.c_void_p),
                ("n_oms_alpha_slots", C.c_int32), ("oms_alpha", C.c_void_p), ("oms_alpha_slot", C.c_void_p),
                ("schedule", C.c_int32)]


# every symbol include/ldpc_hip.h declares (tests check the library exports them all) ...
PRODUCT_EXPORTS = ("ldpc_graph_create", "ldpc_graph_destroy", "ldpc_graph_info", "ldpc_decoder_create",
                   "ldpc_decoder_set_mode", "ldpc_decoder_info",
                   "ldpc_decoder_set_weights", "ldpc_decoder_destroy", "ldpc_decoder_workspace_bytes",
                   "ldpc_decode", "ldpc_decode_capped", "ldpc_last_error", "ldpc_abi_version", "ldpc_source_hash",
                   "ldpc_train_saved_bytes", "ldpc_train_workspace_bytes", "ldpc_decode_saving", "ldpc_backward")
# ... and the measurement / test hooks of include/ldpc_hip_debug.h (bench.py's per-kernel timing, the tests' state dumps)
DEBUG_EXPORTS = ("ldpc_debug_sweep", "ldpc_debug_workspace_layout", "ldpc_debug_resident_c2v", "ldpc_debug_key4")
EXPORTS = PRODUCT_EXPORTS + DEBUG_EXPORTS

_lib = None
_lock = threading.Lock()


def load():
    """dlopen the engine; raises NativeEngineError (never falls back)."""
    global _lib
    if _lib is not None:
        return _lib
    with _lock:
        if _lib is not None:
            return _lib
        if not os.path.exists(LIB_PATH):
            raise NativeEngineError(
                f"{LIB_NAME} is not built: run `python -c 'import __graft_entry__ as g; g.build()'` "
                f"(hipcc --offload-arch=gfx950) -- there is no CPU fallback for the decode path")
        if "LDPC_HIP_LIB" not in os.environ and is_stale(LIB_PATH):
            raise NativeEngineError(
                f"{LIB_PATH} was not built from the sources in csrc/ (content hash {built_hash(LIB_PATH)} != "
                f"{source_hash()}): run `python -c 'import __graft_entry__ as g; g.build()'` -- a stale engine is never loaded")
        try:
            lib = C.CDLL(LIB_PATH)
        except OSError as e:
            raise NativeEngineError(f"cannot load {LIB_PATH}: {e}") from e
        vp, i32, i64 = C.c_void_p, C.c_int32, C.c_int64
        lib.ldpc_graph_create.restype = C.c_int
        lib.ldpc_graph_create.argtypes = [C.POINTER(vp), i32, i32, i32, vp, vp]
        lib.ldpc_graph_destroy.restype = None
        lib.ldpc_graph_destroy.argtypes = [vp]
        lib.ldpc_graph_info.restype = C.c_int
        lib.ldpc_graph_info.argtypes = [vp, vp]
        lib.ldpc_decoder_create.restype = C.c_int
        lib.ldpc_decoder_create.argtypes = [C.POINTER(vp), vp, C.POINTER(DecoderDesc)]
        lib.ldpc_decoder_set_mode.restype = C.c_int
        lib.ldpc_decoder_set_mode.argtypes = [vp, i32]
        lib.ldpc_decoder_info.restype = C.c_int
        lib.ldpc_decoder_info.argtypes = [vp, vp]
        lib.ldpc_decoder_set_weights.restype = C.c_int
        lib.ldpc_decoder_set_weights.argtypes = [vp, vp, vp, vp, vp]
        lib.ldpc_decoder_destroy.restype = None
        lib.ldpc_decoder_destroy.argtypes = [vp]
        lib.ldpc_decoder_workspace_bytes.restype = C.c_size_t
        lib.ldpc_decoder_workspace_bytes.argtypes = [vp, i64]
        lib.ldpc_decode.restype = C.c_int
        lib.ldpc_decode.argtypes = [vp, vp, i64, i32, vp, vp, vp, vp, vp, vp, C.c_size_t, vp]
        lib.ldpc_decode_capped.restype = C.c_int
        lib.ldpc_decode_capped.argtypes = [vp, vp, i64, i32, i32, vp, vp, vp, vp, vp, vp, C.c_size_t, vp]
        lib.ldpc_debug_sweep.restype = C.c_int
        lib.ldpc_debug_sweep.argtypes = [vp, i64, i32, i32, vp, C.c_size_t, vp]
        lib.ldpc_debug_workspace_layout.restype = C.c_int
        lib.ldpc_debug_workspace_layout.argtypes = [vp, i64, vp]
        lib.ldpc_debug_key4.restype = C.c_int
        lib.ldpc_debug_key4.argtypes = [vp, i64, C.c_float, vp, vp, vp, vp]
        lib.ldpc_debug_resident_c2v.restype = C.c_int
        lib.ldpc_debug_resident_c2v.argtypes = [vp, vp, i64, i32, vp, vp, vp, vp]
        lib.ldpc_train_saved_bytes.restype = C.c_size_t
        lib.ldpc_train_saved_bytes.argtypes = [vp, i64]
        lib.ldpc_train_workspace_bytes.restype = C.c_size_t
        lib.ldpc_train_workspace_bytes.argtypes = [vp, i64]
        lib.ldpc_decode_saving.restype = C.c_int
        lib.ldpc_decode_saving.argtypes = [vp, vp, i64, i32, vp, vp, vp, vp, vp, C.c_size_t, vp, C.c_size_t, vp]
        lib.ldpc_backward.restype = C.c_int
        lib.ldpc_backward.argtypes = [vp, vp, C.c_size_t, vp, i64, vp, vp, vp, vp, vp, vp, vp, C.c_size_t, vp]
        lib.ldpc_last_error.restype = C.c_char_p
        lib.ldpc_last_error.argtypes = []
        lib.ldpc_abi_version.restype = C.c_int
        lib.ldpc_abi_version.argtypes = []
        lib.ldpc_source_hash.restype = C.c_char_p
        lib.ldpc_source_hash.argtypes = []
        if lib.ldpc_abi_version() != 1:
            raise NativeEngineError("libldpc_hip.so ABI version mismatch")
        _lib = lib
    return _lib


def check(rc: int, what: str):
    if rc != 0:
        msg = load().ldpc_last_error().decode(errors="replace")
        if rc == -3:
            raise NotImplementedError(f"{what}: {msg}")
        if rc == -1:
            raise ValueError(f"{what}: {msg}")
        raise NativeEngineError(f"{what} failed (rc={rc}): {msg}")


def ptr(a: Optional[np.ndarray]):
    return None if a is None else a.ctypes.data_as(C.c_void_p)
