#!/usr/bin/env python3
"""fp32 transposed convolution through the polyphase input-gradient kernels vs torch, at the generator's upsampler shapes."""
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
import torch
import torch.nn.functional as F

from everyvoice_amd.train import ops

dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(3)
shapes = [(512, 256, 16, 8, 32), (256, 128, 16, 8, 256), (128, 64, 4, 2, 2048), (64, 32, 4, 2, 4096), (512, 256, 16, 8, 8), (64, 32, 4, 2, 1024)]
for B in (16, 2):
    for cin, cout, k, u, T in shapes:
        p = (k - u) // 2
        x = torch.randn(B, cin, T, generator=g)
        w = torch.randn(cin, cout, k, generator=g) * 0.05
        b = torch.randn(cout, generator=g)
        want = F.conv_transpose1d(x, w, b, u, p)
        xd = x.permute(1, 0, 2).contiguous().to(dev)
        got = ops.conv_transpose1d_fwd(xd, w.to(dev), b.to(dev), u, p).cpu().permute(1, 0, 2)
        err = (got - want).abs()
        bad = (err > 1e-3).nonzero()
        print(f"B={B} {cin}->{cout} k{k} s{u} T={T}: max err {float(err.max()):.3e}, bad {len(bad)}", bad[:8].tolist(), flush=True)
