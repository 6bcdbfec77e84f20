#!/usr/bin/env python3
"""Generator forward at T = 768 frames for batches 32 ... 1: ms per forward and per utterance.  If small batches cost less PER UTTERANCE,
the stage tensors of a small batch live in the 256 MiB Infinity Cache between launches and a slab-major schedule of the big batch pays.
usage: python tools/batch_slab_probe.py [frames]"""
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch  # noqa: E402

import bench  # noqa: E402

T = int(sys.argv[1]) if len(sys.argv) > 1 else 768
dev = torch.device("cuda:0")
model = bench.upstream_init_generator("bf16").to(dev).eval()
gen = model.generator
for B in (32, 16, 8, 4, 2, 1):
    mel = bench.synthetic_mel(B, T, 1234).to(dev)
    for _ in range(3):
        gen(mel)
    torch.cuda.synchronize()
    n = max(5, 160 // B)
    t0 = time.perf_counter()
    for _ in range(n):
        gen(mel)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / n * 1e3
    print(f"B={B:3d}: {ms:8.3f} ms / forward, {ms / B:7.4f} ms / utterance, {B * T * 256 / ms / 1e3:7.1f} M samples/s")
