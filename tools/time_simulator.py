#!/usr/bin/env python3
"""Throughput of the batched Monte-Carlo driver: LDPSimulator.simulate_single_snr on the (1998,1512)
code -- on-device AWGN draw + early-stop decode + error counters with the reference's stop rule.  JSON lines."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: F401,E402  (package path)
import codes  # noqa: E402
from ldpc_decoder import BasicMinSumDecoder  # noqa: E402
from simulation_framework import LDPSimulator, SimulationConfig  # noqa: E402

code = codes.load_code("ira_1998_1512", max_iterations=10)
dec = BasicMinSumDecoder(code, 0.7)
sim = LDPSimulator(SimulationConfig(save_results=False, batch_frames=65536, seed=1))
sim.simulate_single_snr(dec, code, 5.0, 65536, 10 ** 9)          # warm-up: engine build, allocator
for snr in (3.0, 4.5, 5.5, 6.5):
    fer, ber, avg_it, secs, frames, errs = sim.simulate_single_snr(dec, code, snr, 2_000_000, 10 ** 9)
    print(json.dumps({"snr_db": snr, "frames": frames, "frames_per_s": frames / secs, "fer": fer, "ber": ber,
                      "avg_iterations": avg_it, "seconds": secs}))
