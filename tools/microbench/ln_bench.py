#!/usr/bin/env python3
"""LayerNorm forward / backward alone at the FastSpeech2 shapes (channel-major fp32 [C, B, T]): us per call, GB/s of the tensors moved.
usage: python tools/microbench/ln_bench.py [iters]"""
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
import torch  # noqa: E402

from everyvoice_amd.train import ops  # noqa: E402

dev = torch.device("cuda:0")
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 50


def timed(fn):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


for C, B, T in ((256, 32, 814), (256, 32, 141), (64, 4, 60)):
    x, dy = torch.randn(C, B, T, device=dev), torch.randn(C, B, T, device=dev)
    g, b = torch.rand(C, device=dev) + 0.5, torch.randn(C, device=dev)
    dg, db = torch.zeros(C, device=dev), torch.zeros(C, device=dev)
    t_f = timed(lambda: ops.layernorm(x, g, b))
    t_b = timed(lambda: ops.layernorm_bwd(x, g, dy, dg, db))
    n = x.numel() * 4
    print(f"[{C} x {B} x {T}]: forward {t_f:6.1f} us ({2 * n / t_f * 1e-3:5.0f} GB/s)   backward {t_b:6.1f} us ({3 * n / t_b * 1e-3:5.0f} GB/s)")
