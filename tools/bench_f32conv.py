#!/usr/bin/env python3
"""fp32 conv backends on the layer shapes of the GAN training step: matrix-core implicit GEMM vs unfold + rocBLAS."""
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch  # noqa: E402

from everyvoice_amd.train import ops  # noqa: E402

dev = torch.device("cuda:0")
B = 16
# (name, cin, cout, k, stride, pad, dil, groups, batch, T_in)
SHAPES = [
    ("G rb c256 k11 d5", 256, 256, 11, 1, 25, 5, 1, B, 256),
    ("G rb c128 k11 d5", 128, 128, 11, 1, 25, 5, 1, B, 2048),
    ("G rb c64 k7 d3", 64, 64, 7, 1, 9, 3, 1, B, 4096),
    ("G rb c32 k3 d1", 32, 32, 3, 1, 1, 1, 1, B, 8192),
    ("MSD L0 1->128 k15", 1, 128, 15, 1, 7, 1, 1, B, 8192),
    ("MSD L1 128->128 k41 s2 g4", 128, 128, 41, 2, 20, 1, 4, B, 8192),
    ("MSD L2 128->256 k41 s2 g16", 128, 256, 41, 2, 20, 1, 16, B, 4096),
    ("MSD L3 256->512 k41 s4 g16", 256, 512, 41, 4, 20, 1, 16, B, 2048),
    ("MSD L4 512->1024 k41 s4 g16", 512, 1024, 41, 4, 20, 1, 16, B, 512),
    ("MSD L5 1024->1024 k41 g16", 1024, 1024, 41, 1, 20, 1, 16, B, 128),
    ("MSD L6 1024->1024 k5", 1024, 1024, 5, 1, 2, 1, 1, B, 128),
    ("MPD p2 L1 32->128 k5 s3", 32, 128, 5, 3, 2, 1, 1, B * 2, 1366),
    ("MPD p2 L3 512->1024 k5 s3", 512, 1024, 5, 3, 2, 1, 1, B * 2, 152),
    ("MPD p2 L4 1024->1024 k5", 1024, 1024, 5, 1, 2, 1, 1, B * 2, 51),
    ("MPD p11 L3 512->1024 k5 s3", 512, 1024, 5, 3, 2, 1, 1, B * 11, 28),
    ("MPD p11 L4 1024->1024 k5", 1024, 1024, 5, 1, 2, 1, 1, B * 11, 10),
]
import os
flt = os.environ.get("F32_LAYERS", "")
if flt:
    SHAPES = [sh for sh in SHAPES if any(f in sh[0] for f in flt.split(","))]
print(f"{'layer':32s} {'GFLOP':>8s} | {'mfma ms':>8s} {'TF/s':>7s} | {'bf16 ms':>8s} {'TF/s':>7s} | {'gemm ms':>8s} {'TF/s':>7s}")
for name, cin, cout, k, s, p, d, g, b, t in SHAPES:
    x = torch.randn(cin, b, t, device=dev)
    w = torch.randn(cout, cin // g, k, device=dev) * 0.1
    bias = torch.zeros(cout, device=dev)
    t_out = ops.conv_out_len(t, k, s, p, d)
    fl = 2.0 * b * t_out * cout * (cin // g) * k
    res = {}
    for backend in ("mfma", "bf16", "gemm"):
        ops.CONV_BACKEND["fwd"] = "mfma" if backend == "bf16" else backend
        ops.CONV_BACKEND["operands"] = "bf16" if backend == "bf16" else "f32"
        for _ in range(2):
            ops.conv1d_fwd(x, w, bias, s, p, d, g)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            ops.conv1d_fwd(x, w, bias, s, p, d, g)
        torch.cuda.synchronize()
        res[backend] = (time.perf_counter() - t0) / 5 * 1e3
    print(f"{name:32s} {fl/1e9:8.2f} | {res['mfma']:8.3f} {fl/res['mfma']/1e9:7.1f} | {res['bf16']:8.3f} {fl/res['bf16']/1e9:7.1f} | {res['gemm']:8.3f} {fl/res['gemm']/1e9:7.1f}")
