"""
Drop-in for the reference module of the same name (training_framework.py): posterior joint training of the
neural min-sum decoders, with the forward AND backward passes on the MI355X engine.

Reference surface mirrored (file:line in /root/reference/training_framework.py):
  TrainingConfig                                  :23-35   same fields and defaults (device default "cuda" here)
  PosteriorJointTrainer(model, config)            :37-291  Adam, generate_training_data, compute_loss,
                                                           train_epoch, train, validate, plot_training_history
  GradientExplosionAnalyzer(model, code)          :293-378 analyze_gradient_explosion, plot_gradient_analysis
  loss = binary_cross_entropy_with_logits(-posteriors, targets)                                       :101

The reference file does not run as shipped (`F` is never imported, :101/:328; the decoders' forward takes one
codeword while the DataLoader hands it [batch, n], :123-127).  This module implements what it sets out to do:
the decoders here accept [B, n], and their posterior carries a grad_fn whose backward is the HIP gradient
sweep (autograd_bridge.py), so the loop below is the reference's loop, batched.

Deviation, stated: training LLRs are generated in the DECODER's sign convention (positive LLR = bit 0) by
default.  The reference calls simulate_awgn_channel, whose opposite convention makes the all-zero codeword
undecodable; `TrainingConfig.llr_convention = "reference"` reproduces that literally.
`train_epoch` returns (loss, accuracy, gradient norm) -- the three values the reference's own caller unpacks
(:208), although its annotation says two.
"""

from __future__ import annotations

import logging
import time
from dataclasses import dataclass
from typing import Dict, List, Optional, Tuple

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F
import torch.optim as optim
from torch.utils.data import DataLoader, TensorDataset

from ldpc_decoder import LDPCCode, simulate_awgn_channel

logger = logging.getLogger(__name__)


@dataclass
class TrainingConfig:
    """Training configuration"""
    batch_size: int = 32
    num_epochs: int = 100
    learning_rate: float = 0.001
    snr_range: Tuple[float, float] = (0.0, 6.0)
    snr_step: float = 0.5
    max_grad_norm: float = 1.0
    use_posterior_training: bool = True
    use_gradient_clipping: bool = False
    clip_threshold: float = 1e-3
    device: str = "cuda"
    llr_convention: str = "decoder"          # "reference": simulate_awgn_channel literally (see module docstring)
    data_parallel: bool = False              # one process per GPU: average the gradients over the ranks every step
    seed: Optional[int] = None               # training-data noise seed (decoder convention only)


def _grad_norm(model: nn.Module) -> float:
    total = 0.0
    for p in model.parameters():
        if p.grad is not None:
            total += p.grad.data.norm(2).item() ** 2
    return total ** 0.5


class PosteriorJointTrainer:
    """Trainer implementing posterior joint training (loss on the returned posterior only)"""

    def __init__(self, model: nn.Module, config: TrainingConfig):
        self.model = model
        self.config = config
        self.device = torch.device(config.device)
        self.model.to(self.device)
        self.optimizer = optim.Adam(self.model.parameters(), lr=config.learning_rate)
        self.train_losses: List[float] = []
        self.train_accuracies: List[float] = []
        self.gradient_norms: List[float] = []
        logger.info(f"Initialized trainer with {sum(p.numel() for p in model.parameters())} parameters")

    def generate_training_data(self, code: LDPCCode, num_samples: int) -> Tuple[torch.Tensor, torch.Tensor]:
        """All-zero codewords through AWGN, one SNR per sample on linspace(snr_min, snr_max) (:57-84)."""
        codewords = torch.zeros(num_samples, code.n, dtype=torch.float32)
        snr_min, snr_max = self.config.snr_range
        snrs = torch.linspace(snr_min, snr_max, num_samples)
        if self.config.llr_convention == "reference":
            llrs = torch.zeros_like(codewords)
            for i in range(num_samples):
                llrs[i] = torch.tensor(simulate_awgn_channel(codewords[i].numpy(), snrs[i].item()), dtype=torch.float32)
            return llrs, codewords
        gen = torch.Generator()
        if self.config.seed is not None:
            gen.manual_seed(int(self.config.seed))
        else:
            gen.seed()
        sigma2 = 10.0 ** (-snrs.double() / 10.0)                       # noise variance at unit symbol energy
        z = torch.randn(num_samples, code.n, generator=gen, dtype=torch.float64)
        llrs = 2.0 * (1.0 + sigma2.sqrt().unsqueeze(1) * z) / sigma2.unsqueeze(1)
        return llrs.to(torch.float32), codewords

    def compute_loss(self, outputs: torch.Tensor, targets: torch.Tensor, posteriors: torch.Tensor) -> torch.Tensor:
        """binary cross entropy of the posterior LLRs against the transmitted bits (:86-104)"""
        return F.binary_cross_entropy_with_logits(-posteriors, targets.float())

    def train_epoch(self, train_loader: DataLoader) -> Tuple[float, float, float]:
        self.model.train()
        total_loss, total_correct, total_samples = 0.0, 0, 0
        epoch_grad_norms = []
        for batch_idx, (llrs, targets) in enumerate(train_loader):
            llrs, targets = llrs.to(self.device), targets.to(self.device)
            self.optimizer.zero_grad()
            decoded, posteriors, iterations = self.model(llrs)
            loss = self.compute_loss(decoded, targets, posteriors)
            loss.backward()
            if self.config.data_parallel:              # each rank trained on its own shard of the batch
                import sharding
                sharding.all_reduce_gradients(self.model.parameters())
            total_norm = _grad_norm(self.model)
            epoch_grad_norms.append(total_norm)
            if self.config.use_gradient_clipping:
                torch.nn.utils.clip_grad_norm_(self.model.parameters(), self.config.clip_threshold)
            self.optimizer.step()
            correct = (decoded == targets).all(dim=1).sum().item()
            total_correct += correct
            total_samples += llrs.size(0)
            total_loss += loss.item()
            if batch_idx % 10 == 0:
                logger.info(f"Batch {batch_idx}, Loss: {loss.item():.6f}, "
                            f"Grad Norm: {total_norm:.6f}, Acc: {correct / llrs.size(0):.4f}")
        avg_grad_norm = float(np.mean(epoch_grad_norms)) if epoch_grad_norms else 0.0
        return total_loss / max(len(train_loader), 1), total_correct / max(total_samples, 1), avg_grad_norm

    def train(self, code: LDPCCode, num_train_samples: int = 1000, num_val_samples: int = 200) -> Dict[str, List[float]]:
        logger.info("Generating training data...")
        train_llrs, train_targets = self.generate_training_data(code, num_train_samples)
        val_llrs, val_targets = self.generate_training_data(code, num_val_samples)
        train_loader = DataLoader(TensorDataset(train_llrs, train_targets), batch_size=self.config.batch_size, shuffle=True)
        val_loader = DataLoader(TensorDataset(val_llrs, val_targets), batch_size=self.config.batch_size, shuffle=False)
        logger.info(f"Starting training for {self.config.num_epochs} epochs...")
        for epoch in range(self.config.num_epochs):
            start_time = time.time()
            train_loss, train_acc, train_grad_norm = self.train_epoch(train_loader)
            val_loss, val_acc, _ = self.validate(val_loader)
            self.train_losses.append(train_loss)
            self.train_accuracies.append(train_acc)
            self.gradient_norms.append(train_grad_norm)
            logger.info(f"Epoch {epoch + 1}/{self.config.num_epochs}: "
                        f"Train Loss: {train_loss:.6f}, Train Acc: {train_acc:.4f}, "
                        f"Val Loss: {val_loss:.6f}, Val Acc: {val_acc:.4f}, "
                        f"Grad Norm: {train_grad_norm:.6f}, Time: {time.time() - start_time:.2f}s")
            if train_acc > 0.99:                                     # :222-224
                logger.info(f"Early stopping at epoch {epoch + 1} due to high accuracy")
                break
        return {"train_losses": self.train_losses, "train_accuracies": self.train_accuracies,
                "gradient_norms": self.gradient_norms}

    def validate(self, val_loader: DataLoader) -> Tuple[float, float, float]:
        self.model.eval()
        total_loss, total_correct, total_samples = 0.0, 0, 0
        with torch.no_grad():
            for llrs, targets in val_loader:
                llrs, targets = llrs.to(self.device), targets.to(self.device)
                decoded, posteriors, iterations = self.model(llrs)
                total_loss += self.compute_loss(decoded, targets, posteriors).item()
                total_correct += (decoded == targets).all(dim=1).sum().item()
                total_samples += llrs.size(0)
        return total_loss / max(len(val_loader), 1), total_correct / max(total_samples, 1), 0.0

    def plot_training_history(self, save_path: Optional[str] = None):
        import matplotlib.pyplot as plt
        fig, axes = plt.subplots(1, 3, figsize=(15, 5))
        for ax, series, title, ylabel in ((axes[0], self.train_losses, "Training Loss", "Loss"),
                                          (axes[1], self.train_accuracies, "Training Accuracy", "Accuracy"),
                                          (axes[2], self.gradient_norms, "Gradient Norms", "Gradient Norm")):
            ax.plot(series)
            ax.set_title(title)
            ax.set_xlabel("Epoch")
            ax.set_ylabel(ylabel)
            ax.grid(True)
        plt.tight_layout()
        if save_path:
            plt.savefig(save_path)
        plt.show()


class GradientExplosionAnalyzer:
    """Gradient magnitudes of the decoder for random inputs (:293-378)"""

    def __init__(self, model: nn.Module, code: LDPCCode):
        self.model = model
        self.code = code

    def analyze_gradient_explosion(self, num_samples: int = 100) -> Dict[str, List[float]]:
        self.model.eval()
        gradient_magnitudes, iteration_gradients = [], []
        for _ in range(num_samples):
            llr = torch.randn(self.code.n) * 2
            decoded, posteriors, iterations = self.model(llr)
            loss = F.binary_cross_entropy_with_logits(-posteriors, torch.zeros_like(decoded).float())
            if posteriors.requires_grad:        # no parameter on the path (e.g. sharing type 4 stopping at once): gradient 0
                loss.backward()
            gradient_magnitudes.append(_grad_norm(self.model))
            iteration_gradients.append(iterations)
            self.model.zero_grad()
        return {"gradient_magnitudes": gradient_magnitudes, "iteration_counts": iteration_gradients,
                "mean_gradient": np.mean(gradient_magnitudes), "std_gradient": np.std(gradient_magnitudes),
                "max_gradient": np.max(gradient_magnitudes)}

    def plot_gradient_analysis(self, results: Dict[str, List[float]], save_path: Optional[str] = None):
        import matplotlib.pyplot as plt
        fig, axes = plt.subplots(1, 2, figsize=(12, 5))
        axes[0].hist(results["gradient_magnitudes"], bins=20, alpha=0.7)
        axes[0].set_title("Gradient Magnitude Distribution")
        axes[0].set_xlabel("Gradient Magnitude")
        axes[0].set_ylabel("Frequency")
        axes[0].grid(True)
        axes[1].scatter(results["iteration_counts"], results["gradient_magnitudes"], alpha=0.6)
        axes[1].set_title("Gradient Magnitude vs Iterations")
        axes[1].set_xlabel("Iterations")
        axes[1].set_ylabel("Gradient Magnitude")
        axes[1].grid(True)
        plt.tight_layout()
        if save_path:
            plt.savefig(save_path)
        plt.show()
