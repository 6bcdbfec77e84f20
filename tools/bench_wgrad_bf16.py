#!/usr/bin/env python3
"""Weight gradient per layer shape of the GAN step: fp32 path ("auto": implicit GEMM / unfold + rocBLAS) vs packed bf16."""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
from everyvoice_amd.train import ops
dev = torch.device("cuda:0")
B = 16
SHAPES = [
    ("G rb c256 k11 d5", 256, 256, 11, 1, 25, 5, 1, B, 256),
    ("G rb c128 k11 d5", 128, 128, 11, 1, 25, 5, 1, B, 2048),
    ("G rb c64 k7 d3", 64, 64, 7, 1, 9, 3, 1, B, 4096),
    ("G rb c32 k3 d1", 32, 32, 3, 1, 1, 1, 1, B, 8192),
    ("G rb c32 k7 d3", 32, 32, 7, 1, 9, 3, 1, B, 8192),
    ("G rb c32 k11 d5", 32, 32, 11, 1, 25, 5, 1, B, 8192),
    ("G rb c64 k11 d5", 64, 64, 11, 1, 25, 5, 1, B, 4096),
    ("MSD L1 128->128 k41 s2 g4", 128, 128, 41, 2, 20, 1, 4, B, 8192),
    ("MSD L4 512->1024 k41 s4 g16", 512, 1024, 41, 4, 20, 1, 16, B, 512),
    ("MSD L5 1024->1024 k41 g16", 1024, 1024, 41, 1, 20, 1, 16, B, 128),
    ("MSD L6 1024->1024 k5", 1024, 1024, 5, 1, 2, 1, 1, B, 128),
    ("MPD p2 L1 32->128 k5 s3", 32, 128, 5, 3, 2, 1, 1, B * 2, 1366),
    ("MPD p2 L2 128->512 k5 s3", 128, 512, 5, 3, 2, 1, 1, B * 2, 456),
    ("MPD p2 L3 512->1024 k5 s3", 512, 1024, 5, 3, 2, 1, 1, B * 2, 152),
    ("MPD p2 L4 1024->1024 k5", 1024, 1024, 5, 1, 2, 1, 1, B * 2, 51),
    ("MPD p11 L3 512->1024 k5 s3", 512, 1024, 5, 3, 2, 1, 1, B * 11, 28),
    ("MPD p11 L4 1024->1024 k5", 1024, 1024, 5, 1, 2, 1, 1, B * 11, 10),
    ("FS2 ffn 256->1024 k1", 256, 1024, 1, 1, 0, 1, 1, 32, 947),
]
print(f"{'layer':32s} {'GFLOP':>8s} | {'f32 ms':>8s} {'TF/s':>7s} | {'bf16 ms':>8s} {'TF/s':>7s}")
for name, cin, cout, k, s, p, d, g, b, t in SHAPES:
    x = torch.randn(cin, b, t, device=dev)
    w = torch.randn(cout, cin // g, k, device=dev) * 0.1
    t_out = ops.conv_out_len(t, k, s, p, d)
    dy = torch.randn(cout, b, t_out, device=dev)
    fl = 2.0 * b * t_out * cout * (cin // g) * k
    res = {}
    for mode in ("f32", "bf16"):
        ops.CONV_BACKEND["operands"] = mode
        for _ in range(2):
            ops.conv1d_bwd(x, w, dy, s, p, d, g, need_dx=False)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            ops.conv1d_bwd(x, w, dy, s, p, d, g, need_dx=False)
        torch.cuda.synchronize()
        res[mode] = (time.perf_counter() - t0) / 5 * 1e3
    print(f"{name:32s} {fl/1e9:8.2f} | {res['f32']:8.3f} {fl/res['f32']/1e9:7.1f} | {res['bf16']:8.3f} {fl/res['bf16']/1e9:7.1f}")
