#!/usr/bin/env python3
"""Is the FastSpeech2 training step bound by the host?  Time to ENQUEUE 20 steps vs time until the device has finished them."""
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
for _ in range(3):
    tr.training_step(batch)
torch.cuda.synchronize()
n = 20
t0 = time.perf_counter()
for _ in range(n):
    tr.training_step(batch)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"enqueue {1e3*(t1-t0)/n:.2f} ms / step, device done {1e3*(t2-t0)/n:.2f} ms / step")
