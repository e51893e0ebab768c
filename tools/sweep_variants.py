#!/usr/bin/env python3
"""Build kernel tuning variants (here, cross-compiled) or time them (on the GPU box).

    python tools/sweep_variants.py build            # -> build_variants/<name>.so
    python tools/sweep_variants.py run [workload]   # one subprocess per variant, JSON lines
"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "implementation-of-neural-ldpc-decoders-with-degree-specific-weight-sharing-and-rcq-quantization_amd")
sys.path.insert(0, PKG)
OUT = os.path.join(ROOT, "build_variants")

VARIANTS = {
    "base": [],
    "unroll8": ["LDPC_CN_UNROLL=8"],
    "unroll16": ["LDPC_CN_UNROLL=16"],
    "nt_store": ["LDPC_NT_STORE=1"],
    "nt_load": ["LDPC_NT_LOAD=1"],
    "nt_both": ["LDPC_NT_STORE=1", "LDPC_NT_LOAD=1"],
    "nt_both_u8": ["LDPC_NT_STORE=1", "LDPC_NT_LOAD=1", "LDPC_CN_UNROLL=8"],
}

if __name__ == "__main__":
    mode = sys.argv[1]
    if mode == "build":
        import _native
        os.makedirs(OUT, exist_ok=True)
        for name, defs in VARIANTS.items():
            print(name, _native.build_native(force=True, defines=defs, out=os.path.join(OUT, name + ".so")))
    else:
        wl = sys.argv[2:] or ["basic"]
        for w in wl:
            for name in VARIANTS:
                env = dict(os.environ, LDPC_HIP_LIB=os.path.join(OUT, name + ".so"))
                subprocess.run([sys.executable, os.path.join(ROOT, "tools", "time_sweeps.py"), "--workload", w], env=env)
