#!/usr/bin/env python3
"""Per-step wall time of the first FastSpeech2 training steps (allocator warm-up, side streams)."""
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch  # noqa: E402

from everyvoice_amd.train.fs2 import FastSpeech2Trainer  # noqa: E402
from fs2_train_bench import training_batch  # noqa: E402

dev = torch.device("cuda:0")
tr = FastSpeech2Trainer(device=dev, precision="bf16")
batch, _ = training_batch(32, 1234, device=dev)
ts = []
for i in range(24):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    tr.training_step(batch)
    torch.cuda.synchronize()
    ts.append((time.perf_counter() - t0) * 1e3)
print(" ".join(f"{t:.1f}" for t in ts))
print("reserved MB", torch.cuda.memory_reserved() / 1e6, "allocated MB", torch.cuda.memory_allocated() / 1e6)
