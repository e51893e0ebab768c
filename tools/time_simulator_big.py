#!/usr/bin/env python3
"""Throughput of the batched Monte-Carlo driver on the code that does not fit LDS: (16200,7200), streaming engine,
early-stop decode per block.  JSON lines."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: F401,E402
import codes  # noqa: E402
from ldpc_decoder import BasicMinSumDecoder  # noqa: E402
from rcq_decoder import RCQMinSumDecoder  # noqa: E402
from simulation_framework import LDPSimulator, SimulationConfig  # noqa: E402

code = codes.load_code("dvbs2_like_16200_7200", max_iterations=20)
sim = LDPSimulator(SimulationConfig(save_results=False, batch_frames=32768, seed=1))
for name, dec in (("Basic MinSum", BasicMinSumDecoder(code, 0.7)), ("RCQ MinSum bc=3", RCQMinSumDecoder(code, 3, 8, [(3.0, 1.3), (5.0, 1.3), (7.0, 1.3)], 20))):
    sim.simulate_single_snr(dec, code, 5.0, 32768, 10 ** 9)          # warm-up
    for snr in (2.0, 4.0, 6.0):
        fer, ber, avg_it, secs, frames, errs = sim.simulate_single_snr(dec, code, snr, 262144, 10 ** 9)
        print(json.dumps({"decoder": name, "snr_db": snr, "frames": frames, "frames_per_s": frames / secs, "fer": fer, "ber": ber,
                          "avg_iterations": avg_it, "seconds": secs}))
