#!/usr/bin/env python3
"""Run N optimiser steps of both trainers in a given precision on fresh synthetic batches and report the loss trajectory
(finite, no blow-up): a robustness check of the bf16-operand path under real training dynamics."""
import math
import os
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
sys.path.insert(0, str(Path(__file__).resolve().parent))
import torch  # noqa: E402

from everyvoice_amd.spectral import MelSpectrogram  # noqa: E402
from everyvoice_amd.train.fs2 import FastSpeech2Trainer  # noqa: E402
from everyvoice_amd.train.hifigan import HiFiGANTrainer  # noqa: E402
from fs2_train_bench import training_batch  # noqa: E402

prec = os.environ.get("OPERANDS", "bf16")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
dev = torch.device("cuda:0")
tr = HiFiGANTrainer(device=dev, precision=prec)
mel_fn = MelSpectrogram()
g = torch.Generator().manual_seed(7)
for i in range(n):
    y = (0.3 * torch.tanh(torch.randn(16, 1, 8192, generator=g))).to(dev)
    mel = mel_fn(y.squeeze(1), log=True)[:, :, :32].contiguous()
    out = tr.training_step(mel, y)
    assert all(math.isfinite(v) for v in out.values()), (i, out)
    if i % max(1, n // 8) == 0 or i == n - 1:
        print(f"GAN[{prec}] step {i:4d}: " + " ".join(f"{k}={v:.4f}" for k, v in out.items()))
fs = FastSpeech2Trainer(device=dev, precision=prec)
for i in range(n // 2):
    batch, _ = training_batch(32, 100 + i, device=dev)
    out = {k: float(v) for k, v in fs.training_step(batch).items()}
    assert all(math.isfinite(v) for v in out.values()), (i, out)
    if i % max(1, n // 16) == 0 or i == n // 2 - 1:
        print(f"FS2[{prec}] step {i:4d}: " + " ".join(f"{k}={v:.4f}" for k, v in out.items()))
print("stable")
