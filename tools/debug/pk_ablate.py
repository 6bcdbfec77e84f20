#!/usr/bin/env python3
"""Timing ablations (EVMI_PK_ABLATE) of the packed bf16 training convolution at the generator's narrow shapes."""
import os
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
import torch

from everyvoice_amd.train import ops

ops.CONV_BACKEND.update(fwd="mfma", dgrad="mfma", operands="bf16")
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(1)
shapes = [(32, 32, 11, 1, 16, 8192), (64, 64, 11, 1, 16, 4096), (32, 32, 3, 1, 16, 8192), (128, 128, 11, 1, 16, 2048)]
for cin, cout, k, dil, B, T in shapes:
    x = torch.randn(cin, B, T, generator=g).to(dev)
    w = (torch.randn(cout, cin, k, generator=g) * 0.05).to(dev)
    b = torch.zeros(cout, device=dev)
    flops = 2.0 * B * T * cout * cin * k
    print(f"== {cin}->{cout} k{k} B{B} T{T}: {flops/1e9:.2f} GF")
    for name, abl in (("full", 0), ("no window loads", 1), ("no weight loads", 2), ("no MFMA loop", 4), ("no stores", 8), ("no K loop", 16), ("no loads no stores", 11), ("prep only (no K loop, no stores)", 24)):
        os.environ["EVMI_PK_ABLATE"] = str(abl)
        for _ in range(3):
            ops.conv1d_fwd(x, w, b, 1, dil * (k - 1) // 2, dil, 1)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        n = 20
        for _ in range(n):
            ops.conv1d_fwd(x, w, b, 1, dil * (k - 1) // 2, dil, 1)
        e1.record()
        e1.synchronize()
        ms = e0.elapsed_time(e1) / n
        print(f"   {name:34s} {ms*1e3:8.1f} us   ({flops/ms/1e9:7.1f} TF/s)")
os.environ["EVMI_PK_ABLATE"] = "0"

# the same shapes through the in-LDS-rounding variant of the fp32 kernels (no packing pass, one launch) and in exact fp32
for packed, operands in ((True, "bf16"), (False, "bf16"), (False, "f32")):
    ops.CONV_BACKEND.update(packed=packed, operands=operands)
    for cin, cout, k, dil, B, T in shapes:
        x = torch.randn(cin, B, T, generator=g).to(dev)
        w = (torch.randn(cout, cin, k, generator=g) * 0.05).to(dev)
        b = torch.zeros(cout, device=dev)
        flops = 2.0 * B * T * cout * cin * k
        for _ in range(3):
            ops.conv1d_fwd(x, w, b, 1, dil * (k - 1) // 2, dil, 1)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            ops.conv1d_fwd(x, w, b, 1, dil * (k - 1) // 2, dil, 1)
        e1.record()
        e1.synchronize()
        ms = e0.elapsed_time(e1) / 20
        print(f"packed={packed} {operands}: {cin}->{cout} k{k} T{T}: {ms*1e3:8.1f} us ({flops/ms/1e9:7.1f} TF/s)")
