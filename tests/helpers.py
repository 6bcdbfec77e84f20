"""Shared helpers for the parity tests (oracle side only)."""

from __future__ import annotations

import math

import torch

from oracle.hifigan_ref import GeneratorRef, HiFiGANModelConfigRef


def signal_preserving_init_(gen: GeneratorRef, seed: int) -> GeneratorRef:
    """Re-draw every (already folded) weight so activations stay O(1) through the network — the
    upstream N(0, 0.01) init makes the output collapse to tanh(bias), which would hide errors."""
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for name, p in gen.named_parameters():
            if name.endswith("bias"):
                p.copy_(torch.randn(p.shape, generator=g) * 0.05)
                continue
            if name.startswith("ups."):
                fan_in = p.shape[0] * p.shape[2] / gen.cfg.upsample_rates[int(name.split(".")[1])]
            else:
                fan_in = p.shape[1] * p.shape[2]
            gain = 0.6 if ".convs2." in name else 1.0
            p.copy_(torch.randn(p.shape, generator=g) * gain / math.sqrt(fan_in))
    return gen


def make_ref_generator(cfg: HiFiGANModelConfigRef | None = None, seed: int = 1234) -> GeneratorRef:
    torch.manual_seed(seed)
    gen = GeneratorRef(cfg).eval()
    gen.remove_weight_norm()
    return signal_preserving_init_(gen, seed)


def synthetic_mel(B: int, T: int, n_mels: int = 80, seed: int = 1234) -> torch.Tensor:
    """SURVEY.md §8d C2 input: clamp(N(-5, 2^2), -11.5129, 2.0)."""
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(B, n_mels, T, generator=g) * 2.0 - 5.0).clamp(-11.5129, 2.0)


def rel_l2(a: torch.Tensor, b: torch.Tensor) -> float:
    return float((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-30))
