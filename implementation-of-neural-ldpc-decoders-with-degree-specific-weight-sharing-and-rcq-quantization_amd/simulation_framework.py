"""
Batched Monte-Carlo driver -- the natural caller of the batched engine.
Mirrors the reference module of the same name: ``SimulationConfig``, ``SimulationResult``,
``LDPSimulator`` (``simulate_single_snr`` / ``simulate_decoder`` / ``simulate_multiple_decoders``
/ ``save_results`` / ``load_results``) and ``create_test_decoders``, same field names, return
tuples and JSON layout (simulation_framework.py:27-69, 85-208, 338-420).  Plotting and the
training hooks of the reference are out of scope.

What changes underneath: the reference draws ONE noise vector with numpy and decodes ONE frame per
Python call (simulation_framework.py:110-132).  Here a block of ``batch_frames`` frames is drawn on
the GPU (``torch.randn``), decoded in one engine call with per-codeword early exit, and the error /
iteration counters are reduced on the device.  The stop rule is the reference's, applied frame by
frame: frames are taken in order until ``max_frames`` frames or ``max_errors`` frame errors have
been seen, so a block is truncated at the frame that reaches the limit.

Channel convention (``SimulationConfig.llr_convention``):
  "decoder"   (default) all-zero codeword with positive mean LLR, llr = 2(1 + sigma z)/sigma^2 --
              consistent with every decoder's ``posterior < 0 -> 1`` decision;
  "reference" the literal recipe of simulate_awgn_channel (ldpc_decoder.py:286-302): bit 0 -> -1,
              i.e. NEGATIVE mean LLR, under which the reference's own simulations report FER = 1.0
              (SURVEY 8a-9).
"""

from __future__ import annotations

import json
import logging
import os
import time
from concurrent.futures import ThreadPoolExecutor, as_completed
from dataclasses import dataclass
from typing import Callable, Dict, List, Tuple, Union

import numpy as np
import torch

from ldpc_decoder import BasicMinSumDecoder, LDPCCode
from neural_2d_decoder import Neural2DMinSumDecoder, Neural2DOffsetMinSumDecoder
from neural_minsum_decoder import NeuralMinSumDecoder, NeuralOffsetMinSumDecoder
from rcq_decoder import RCQMinSumDecoder, WeightedRCQDecoder

logger = logging.getLogger(__name__)


@dataclass
class SimulationConfig:
    """Simulation configuration (reference fields first, same defaults except `device`)"""
    snr_range: Tuple[float, float] = (0.0, 6.0)
    snr_step: float = 0.5
    max_frames: int = 10000
    max_errors: int = 100
    min_frames: int = 1000
    parallel_workers: int = 4
    device: str = "cuda"
    save_results: bool = True
    results_dir: str = "simulation_results"
    # extensions
    batch_frames: int = 8192
    seed: int = 0
    llr_convention: str = "decoder"
    staged_early_stop: bool = True          # streaming engine: cap the block's first stage, finish stragglers as a small batch
    stage_min_block: int = 1024             # ... only for blocks of at least this many frames


class SimulationResult:
    """Container for simulation results (same attributes as the reference's)"""

    def __init__(self, decoder_name: str, snr_values: List[float]):
        self.decoder_name = decoder_name
        self.snr_values = snr_values
        self.frame_error_rates: List[float] = []
        self.bit_error_rates: List[float] = []
        self.average_iterations: List[float] = []
        self.simulation_times: List[float] = []
        self.total_frames: List[int] = []
        self.total_errors: List[int] = []

    def add_result(self, snr_idx: int, fer: float, ber: float, avg_iter: float, sim_time: float,
                   total_frames: int, total_errors: int):
        for lst, zero in ((self.frame_error_rates, 0.0), (self.bit_error_rates, 0.0), (self.average_iterations, 0.0),
                          (self.simulation_times, 0.0), (self.total_frames, 0), (self.total_errors, 0)):
            while len(lst) <= snr_idx:
                lst.append(zero)
        self.frame_error_rates[snr_idx] = fer
        self.bit_error_rates[snr_idx] = ber
        self.average_iterations[snr_idx] = avg_iter
        self.simulation_times[snr_idx] = sim_time
        self.total_frames[snr_idx] = total_frames
        self.total_errors[snr_idx] = total_errors


def frames_to_count(frame_error: np.ndarray, frames_so_far: int, errors_so_far: int,
                    max_frames: int, max_errors: int) -> int:
    """How many frames of a block the reference's loop ``while total_frames < max_frames and
    frame_errors < max_errors`` (simulation_framework.py:110) would still have simulated: frames are
    consumed in order and the loop stops right after the frame that reaches either limit."""
    room = max(0, max_frames - frames_so_far)
    take = min(len(frame_error), room)
    need = max_errors - errors_so_far
    if need <= 0:
        return 0
    cum = np.cumsum(frame_error[:take].astype(np.int64))
    hit = np.nonzero(cum >= need)[0]
    if hit.size:
        take = int(hit[0]) + 1
    return take


def _engine_of(decoder, device):
    """the batched engine behind a host decoder object (fp32)"""
    if isinstance(decoder, BasicMinSumDecoder):
        return decoder._engine(torch.float32, device)
    if hasattr(decoder, "_get_engine"):
        return decoder._get_engine(device)
    raise TypeError(f"{type(decoder).__name__} is not one of this package's decoders")


def _decode_block(eng, llr: torch.Tensor, cap):
    """Early-stop decode of one block -> (bits, iterations), exactly what eng.decode(llr, early_stop=True) returns.
    The streaming engine freezes whole 256-codeword tiles only, so a block whose codewords stop after 6 iterations on average
    but has a straggler in every tile costs all T iterations.  With `cap` (chosen by _next_cap from the previous block) the
    block runs in two stages: every codeword up to `cap` iterations (ldpc_decode_capped: the decoder's own per-iteration tables),
    then the few still open ones again from their LLRs as a small batch with the full iteration count.  A codeword's decode does
    not depend on its neighbours in the batch, and a codeword that stops within the cap stops at the same iteration with the
    same decisions either way: the result is identical to the one-stage decode.
    The decisions come back bit-packed (uint8 [B, ceil(n/8)], the multi-GPU wire format): the driver only counts them, and the
    engine then stores neither int32 decision rows nor posterior rows."""
    kw = dict(early_stop=True, want_bits=False, want_posterior=False, want_packed=True)
    if cap is None:
        res = eng.decode(llr, **kw)
        return res.packed_bits, res.iterations
    res = eng.decode(llr, max_iters=int(cap), **kw)
    packed, iters = res.packed_bits, res.iterations
    open_idx = torch.nonzero(~res.success, as_tuple=False).reshape(-1)
    if open_idx.numel():
        rest = eng.decode(llr.index_select(0, open_idx).contiguous(), **kw)
        packed.index_copy_(0, open_idx, rest.packed_bits)
        iters.index_copy_(0, open_idx, rest.iterations)
    return packed, iters


_POPCOUNT = {}


def _ones_per_frame(packed: torch.Tensor, n: int) -> torch.Tensor:
    """number of set decision bits of every frame (int64 [B]) from the bit-packed rows"""
    lut = _POPCOUNT.get(packed.device)
    if lut is None:
        lut = torch.tensor([bin(v).count("1") for v in range(256)], dtype=torch.int16, device=packed.device)
        _POPCOUNT[packed.device] = lut
    if n % 8:                                       # bits past n in the last byte do not belong to the codeword
        packed = packed.clone()
        packed[:, -1] &= (1 << (n % 8)) - 1
    return lut[packed.to(torch.int64)].sum(dim=1, dtype=torch.int64)


def _next_cap(eng, iters: torch.Tensor, min_block: int = 1024):
    """Iteration cap for the next block from this block's stop iterations: the cap c minimising c + open(c) * (T + 2), the cost
    in full-block iterations of stage one plus the restart of the codewords still open after c (open(c) = share with more than c
    iterations; one that never converges counts as open for every c < T).  None when the engine stops codeword by codeword
    anyway (LDS-resident) or when staging would not save a quarter of the block's work."""
    info = eng.info()
    T = int(eng.iters)
    if info["engine"] != "stream" or T < 4 or iters.numel() < min_block:
        return None
    hist = torch.bincount(iters.to(torch.int64).clamp_(0, T), minlength=T + 1).to(torch.float64)
    open_after = 1.0 - torch.cumsum(hist, 0) / float(iters.numel())            # open_after[c] = share with iterations > c
    c = torch.arange(T + 1, dtype=torch.float64, device=open_after.device)
    cost = c + open_after * (T + 2.0)
    cost[0] = float("inf")
    best = int(torch.argmin(cost[:T]).item()) if T > 1 else T
    return best if float(cost[best]) <= 0.75 * T else None


class LDPSimulator:
    """LDPC decoder simulator on the batched GPU engine"""

    def __init__(self, config: SimulationConfig):
        self.config = config
        self.results: Dict[str, SimulationResult] = {}
        if config.save_results:
            os.makedirs(config.results_dir, exist_ok=True)

    # ------------------------------------------------------------------------------ one SNR point
    def _draw_llr(self, gen: torch.Generator, frames: int, n: int, snr_db: float, device) -> torch.Tensor:
        snr_linear = 10 ** (snr_db / 10)
        noise_power = 1 / snr_linear
        z = torch.randn((frames, n), generator=gen, device=device, dtype=torch.float32)
        symbol = 1.0 if self.config.llr_convention == "decoder" else -1.0      # all-zero codeword
        # llr = 2 * (symbol + sigma * z) / sigma^2 (simulation_framework.py:95-103 of the reference), as ONE scale-and-shift in
        # place: two passes over the block instead of four element-wise kernels with a temporary each
        return z.mul_(2.0 * (noise_power ** 0.5) / noise_power).add_(2.0 * symbol / noise_power)

    def simulate_single_snr(self, decoder: Callable, code: LDPCCode, snr_db: float, max_frames: int,
                            max_errors: int) -> Tuple[float, float, float, float, int, int]:
        """-> fer, ber, avg_iterations, simulation_time, total_frames, total_errors (frame errors)"""
        start_time = time.time()
        from engine import _require_gpu
        device = _require_gpu(self.config.device)
        eng = _engine_of(decoder, device)
        gen = torch.Generator(device=device)
        gen.manual_seed(int(self.config.seed) * 1_000_003 + int(round(snr_db * 1000)))
        frame_errors = bit_errors = total_iterations = total_frames = 0
        block = max(1, int(self.config.batch_frames))
        cap = None                                                              # iteration cap of the next block's first stage
        while total_frames < max_frames and frame_errors < max_errors:
            frames = min(block, max_frames - total_frames)
            llr = self._draw_llr(gen, frames, code.n, float(snr_db), device)
            packed, iters = _decode_block(eng, llr, cap)
            cap = _next_cap(eng, iters, int(self.config.stage_min_block)) if self.config.staged_early_stop else None
            wrong = _ones_per_frame(packed, code.n)                             # all-zero codeword was sent: every 1 is an error
            ferr = wrong > 0
            take = frames_to_count(ferr.cpu().numpy(), total_frames, frame_errors, max_frames, max_errors)
            frame_errors += int(ferr[:take].sum().item())
            bit_errors += int(wrong[:take].sum().item())
            total_iterations += int(iters[:take].sum().item())
            total_frames += take
        fer = frame_errors / total_frames if total_frames > 0 else 0.0
        ber = bit_errors / (total_frames * code.n) if total_frames > 0 else 0.0
        avg_iterations = total_iterations / total_frames if total_frames > 0 else 0.0
        return fer, ber, avg_iterations, time.time() - start_time, total_frames, frame_errors

    # ------------------------------------------------------------------------------ sweeps
    def simulate_decoder(self, decoder: Union[Callable, torch.nn.Module], code: LDPCCode,
                         decoder_name: str) -> SimulationResult:
        logger.info(f"Starting simulation for {decoder_name}")
        snr_values = np.arange(self.config.snr_range[0], self.config.snr_range[1] + self.config.snr_step,
                               self.config.snr_step)
        result = SimulationResult(decoder_name, snr_values.tolist())
        for snr_idx, snr_db in enumerate(snr_values):
            fer, ber, avg_iter, sim_time, total_frames, total_errors = self.simulate_single_snr(
                decoder, code, snr_db, self.config.max_frames, self.config.max_errors)
            result.add_result(snr_idx, fer, ber, avg_iter, sim_time, total_frames, total_errors)
            logger.info(f"SNR {snr_db:.1f}dB: FER={fer:.2e}, BER={ber:.2e}, Avg Iter={avg_iter:.1f}, Time={sim_time:.1f}s")
        self.results[decoder_name] = result
        return result

    def simulate_multiple_decoders(self, decoders: Dict[str, Union[Callable, torch.nn.Module]],
                                   code: LDPCCode) -> Dict[str, SimulationResult]:
        """one worker thread per decoder, as the reference does (each decoder owns its engine and
        workspace; the GPU serialises the kernels)"""
        logger.info(f"Starting simulation for {len(decoders)} decoders")
        results: Dict[str, SimulationResult] = {}
        if self.config.parallel_workers > 1:
            with ThreadPoolExecutor(max_workers=self.config.parallel_workers) as executor:
                futures = {executor.submit(self.simulate_decoder, dec, code, name): name for name, dec in decoders.items()}
                for future in as_completed(futures):
                    name = futures[future]
                    try:
                        results[name] = future.result()
                        logger.info(f"Completed simulation for {name}")
                    except Exception as e:            # the reference logs and carries on (simulation_framework.py:203-208)
                        logger.error(f"Error simulating {name}: {e}")
        else:
            for name, dec in decoders.items():
                results[name] = self.simulate_decoder(dec, code, name)
        if self.config.save_results:
            self.save_results(results, "simulation_results.json")
        return results

    # ------------------------------------------------------------------------------ persistence
    def save_results(self, results: Dict[str, SimulationResult], filename: str):
        serializable = {name: {"decoder_name": r.decoder_name, "snr_values": r.snr_values,
                               "frame_error_rates": r.frame_error_rates, "bit_error_rates": r.bit_error_rates,
                               "average_iterations": r.average_iterations, "simulation_times": r.simulation_times,
                               "total_frames": r.total_frames, "total_errors": r.total_errors}
                        for name, r in results.items()}
        os.makedirs(self.config.results_dir, exist_ok=True)
        filepath = f"{self.config.results_dir}/{filename}"
        with open(filepath, "w") as f:
            json.dump(serializable, f, indent=2)
        logger.info(f"Results saved to {filepath}")

    def load_results(self, filename: str) -> Dict[str, SimulationResult]:
        filepath = f"{self.config.results_dir}/{filename}"
        with open(filepath, "r") as f:
            data = json.load(f)
        results = {}
        for name, d in data.items():
            r = SimulationResult(d["decoder_name"], d["snr_values"])
            r.frame_error_rates = d["frame_error_rates"]
            r.bit_error_rates = d["bit_error_rates"]
            r.average_iterations = d["average_iterations"]
            r.simulation_times = d["simulation_times"]
            r.total_frames = d["total_frames"]
            r.total_errors = d["total_errors"]
            results[name] = r
        logger.info(f"Results loaded from {filepath}")
        return results


def create_test_decoders(code: LDPCCode) -> Dict[str, Union[Callable, torch.nn.Module]]:
    """the reference's comparison set (simulation_framework.py:384-420)"""
    decoders: Dict[str, Union[Callable, torch.nn.Module]] = {}
    decoders["Basic MinSum"] = BasicMinSumDecoder(code, factor=0.7)
    decoders["N-NMS"] = NeuralMinSumDecoder(code, max_iterations=10)
    decoders["N-OMS"] = NeuralOffsetMinSumDecoder(code, max_iterations=10)
    for weight_type in [1, 2, 3, 4]:
        decoders[f"N-2D-NMS Type {weight_type}"] = Neural2DMinSumDecoder(code, weight_sharing_type=weight_type,
                                                                         max_iterations=10)
    decoders["N-2D-OMS Type 2"] = Neural2DOffsetMinSumDecoder(code, weight_sharing_type=2, max_iterations=10)
    quantizer_params = [(3.0, 1.3), (5.0, 1.3), (7.0, 1.3)]
    decoders["RCQ MinSum"] = RCQMinSumDecoder(code, bc=3, bv=8, quantizer_params=quantizer_params, max_iterations=10)
    decoders["W-RCQ Type 2"] = WeightedRCQDecoder(code, bc=3, bv=8, quantizer_params=quantizer_params,
                                                  weight_sharing_type=2, max_iterations=10)
    return decoders
