#!/usr/bin/env python3
"""cProfile of the FastSpeech2 training step at batch 2 (device work negligible: what the host spends per step)."""
import cProfile
import pstats
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch  # noqa: E402

from everyvoice_amd.train.fs2 import FastSpeech2Trainer  # noqa: E402
from fs2_train_bench import training_batch  # noqa: E402

dev = torch.device("cuda:0")
tr = FastSpeech2Trainer(device=dev, precision="bf16")
batch, _ = training_batch(2, 1234, device=dev)
for _ in range(3):
    tr.training_step(batch)
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(10):
    tr.training_step(batch)
torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(32)
