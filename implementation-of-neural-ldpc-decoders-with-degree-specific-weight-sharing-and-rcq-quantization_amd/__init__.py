"""
MI355X-native neural-MinSum / RCQ LDPC decode engine.

The directory mirrors the reference's flat module layout: put it on ``sys.path`` (or
import this package, which does so) and the reference's import lines keep working,

    from ldpc_decoder import LDPCCode, BasicMinSumDecoder, create_test_ldpc_code, simulate_awgn_channel
    from neural_2d_decoder import Neural2DMinSumDecoder, Neural2DOffsetMinSumDecoder
    from rcq_decoder import NonUniformQuantizer, RCQMinSumDecoder, WeightedRCQDecoder
    from neural_minsum_decoder import NeuralMinSumDecoder, NeuralOffsetMinSumDecoder, analyze_weight_patterns

with ``decode`` / ``forward`` executed by hand-written gfx950 kernels behind the C ABI
of ``include/ldpc_hip.h`` (``libldpc_hip.so`` in this directory, built by
``__graft_entry__.build()``).  There is no CPU fallback for the decode path.
"""
import os as _os
import sys as _sys

_here = _os.path.dirname(_os.path.abspath(__file__))
if _here not in _sys.path:
    _sys.path.insert(0, _here)

from ldpc_decoder import (LDPCCode, BasicMinSumDecoder, create_test_ldpc_code,  # noqa: E402,F401
                          simulate_awgn_channel)
from neural_2d_decoder import Neural2DMinSumDecoder, Neural2DOffsetMinSumDecoder  # noqa: E402,F401
from rcq_decoder import NonUniformQuantizer, RCQMinSumDecoder, WeightedRCQDecoder  # noqa: E402,F401
from neural_minsum_decoder import (NeuralMinSumDecoder, NeuralOffsetMinSumDecoder,  # noqa: E402,F401
                                   analyze_weight_patterns)
from tanner_graph import TannerGraph  # noqa: E402,F401
from engine import DecodeEngine, DecodeResult  # noqa: E402,F401
from _native import NativeEngineError, build_native  # noqa: E402,F401
import codes  # noqa: E402,F401
from simulation_framework import (SimulationConfig, SimulationResult, LDPSimulator,  # noqa: E402,F401
                                  create_test_decoders)

PACKAGE_DIR = _here
