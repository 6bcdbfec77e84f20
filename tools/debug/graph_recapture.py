#!/usr/bin/env python3
"""Does the GAN step's time depend on the capture?  Capture the step graph several times in one process, time 40 replays of each."""
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
import torch  # noqa: E402

from everyvoice_amd.spectral import MelSpectrogram  # noqa: E402
from everyvoice_amd.train.hifigan import HiFiGANTrainer  # noqa: E402

dev = torch.device("cuda:0")
B, S = 16, 8192
g = torch.Generator().manual_seed(1234)
y = (0.3 * torch.tanh(torch.randn(B, 1, S, generator=g))).to(dev)
mel = MelSpectrogram()(y.squeeze(1), log=True)[:, :, : S // 256].contiguous()
tr = HiFiGANTrainer(device=dev, precision="bf16", use_graph=True)
for rnd in range(int(sys.argv[1]) if len(sys.argv) > 1 else 5):
    tr._graphs.clear()
    tr._graph_warm.clear()
    for i in range(4):
        tr.training_step(mel, y)
    assert len(tr._graphs) == 1 and tr._graph_failed is None, tr._graph_failed
    torch.cuda.synchronize()
    ts = []
    for rep in range(3):
        t0 = time.perf_counter()
        for i in range(40):
            tr.training_step(mel, y)
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) / 40 * 1e3)
    print(f"capture {rnd}: " + " ".join(f"{t:.2f}" for t in ts) + " ms")
