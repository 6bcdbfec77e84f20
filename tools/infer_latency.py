#!/usr/bin/env python3
"""HiFiGAN-V1 generator latency at serving shapes (one or a few utterances): ms per forward, median of 30, for a list of (B, frames)."""
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch  # noqa: E402

import bench  # noqa: E402

dev = torch.device("cuda:0")
model = bench.upstream_init_generator("bf16").to(dev).eval()
shapes = [(1, 100), (1, 400), (1, 800), (4, 400), (8, 768), (32, 768)]
out = []
for B, T in shapes:
    mel = bench.synthetic_mel(B, T, 7).to(dev)
    for _ in range(5):
        model.generator(mel)
    torch.cuda.synchronize()
    ts = []
    for _ in range(30):
        t0 = time.perf_counter()
        model.generator(mel)
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    ts.sort()
    out.append(f"{B}x{T}: {ts[15] * 1e3:.3f} ms")
print(" | ".join(out))
