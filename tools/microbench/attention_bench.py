#!/usr/bin/env python3
"""The training attention kernels alone at the FastSpeech2 bench shapes (decoder: 32 x 814 frames, encoder: 32 x 141 symbols; 256
channels in 2 heads, attention dropout 0.1): us per forward / backward call, bf16 and fp32 operands.
usage: python tools/microbench/attention_bench.py [iters]"""
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


g = torch.Generator().manual_seed(3)
for name, B, T, lo in (("decoder 32 x 814", 32, 814, 500), ("encoder 32 x 141", 32, 141, 60)):
    D, H = 256, 2
    qkv = torch.randn(3 * D, B, T, generator=g).to(dev)
    lens = torch.randint(lo, T + 1, (B,), generator=g).to(torch.int32)
    lens[0] = T
    lens = lens.to(dev)
    dout = torch.randn(D, B, T, generator=g).to(dev)
    for prec in ("bf16", "f32"):
        ops.CONV_BACKEND["operands"] = prec
        out, saved = ops.attention_train_fwd(qkv, lens, H, 0.1, 7)
        t_f = timed(lambda: ops.attention_train_fwd(qkv, lens, H, 0.1, 7))
        t_b = timed(lambda: ops.attention_train_bwd(qkv, saved, dout, H, 0.1, 7))
        n = float((lens.double() ** 2).sum())
        fl_f, fl_b = 4.0 * n * D, 10.0 * n * D  # QK^T + PV;  S, dP, dQ, dK, dV recomputed / formed
        print(f"{name:18s} {prec:5s} forward {t_f:7.1f} us ({fl_f / t_f * 1e-6:6.1f} TFLOP/s)   backward {t_b:7.1f} us ({fl_b / t_b * 1e-6:6.1f} TFLOP/s)")
ops.CONV_BACKEND["operands"] = "f32"
