#!/usr/bin/env python3
"""Forward convolution of the packed bf16 kernel at the large layer shapes of both training legs, every tile forced in turn
(EVMI_PK_TILE is read per call): ms (pack + convolution) and TFLOP/s per tile, the planner's own choice marked.
usage: python tools/pk_tile_sweep.py [filter]"""
import os
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch  # noqa: E402

from everyvoice_amd import _lib  # noqa: E402
from everyvoice_amd.train import ops  # noqa: E402

dev = torch.device("cuda:0")
# (name, cin, cout, k, stride, pad, dil, groups, items, T_in)
SHAPES = [
    ("fs2 ffn 256->1024 32x814", 256, 1024, 1, 1, 0, 1, 1, 32, 814),
    ("fs2 ffn 1024->256 32x814", 1024, 256, 1, 1, 0, 1, 1, 32, 814),
    ("fs2 qkv 256->768 32x814", 256, 768, 1, 1, 0, 1, 1, 32, 814),
    ("fs2 postnet 512->512 k5", 512, 512, 5, 1, 2, 1, 1, 32, 814),
    ("fs2 enc ffn 256->1024 32x141", 256, 1024, 1, 1, 0, 1, 1, 32, 141),
    ("mpd p2 512->1024 k5 s3", 512, 1024, 5, 3, 2, 1, 1, 64, 152),
    ("mpd p2 1024->1024 k5", 1024, 1024, 5, 1, 2, 1, 1, 64, 51),
    ("mpd p5 1024->1024 k5", 1024, 1024, 5, 1, 2, 1, 1, 160, 21),
    ("mpd p11 1024->1024 k5", 1024, 1024, 5, 1, 2, 1, 1, 352, 10),
    ("mpd p3 128->512 k5 s3", 128, 512, 5, 3, 2, 1, 1, 96, 304),
    ("msd 1024->1024 k5 32x128", 1024, 1024, 5, 1, 2, 1, 1, 32, 128),
    ("msd 512->1024 k41 s4 g16", 512, 1024, 41, 4, 20, 1, 16, 32, 512),
    ("msd 1024->1024 k41 g16", 1024, 1024, 41, 1, 20, 1, 16, 32, 128),
]
flt = sys.argv[1] if len(sys.argv) > 1 else ""
ops.CONV_BACKEND["operands"] = "bf16"
lib = _lib.load()
tiles = [None, 0, 1, 2, 4, 6, 7, 8, 9]
print(f"{'layer':32s} {'GFLOP':>7s} | " + " | ".join(f"{'auto' if t is None else 'tile ' + str(t):>13s}" for t in tiles))
for name, cin, cout, k, s, p, d, g, b, t in SHAPES:
    if flt and flt not in name:
        continue
    x = torch.randn(cin, b, t, device=dev)
    w = torch.randn(cout, cin // g, k, device=dev) * 0.1
    bias = torch.zeros(cout, device=dev)
    t_out = ops.conv_out_len(t, k, s, p, d)
    fl = 2.0 * b * t_out * cout * (cin // g) * k
    cells = []
    for tile in tiles:
        if tile is None:
            os.environ.pop("EVMI_PK_TILE", None)
        else:
            os.environ["EVMI_PK_TILE"] = str(tile)
        plan = lib.evmi_conv1d_cbt_bf16pk_plan(b, cin, t, cout, t_out, k, s, p, d, g)
        if plan < 0:
            cells.append(f"{'-':>13s}")
            continue
        for _ in range(3):
            ops.conv1d_fwd(x, w, bias, s, p, d, g)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = 20
        for _ in range(n):
            ops.conv1d_fwd(x, w, bias, s, p, d, g)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / n * 1e3
        cells.append(f"{ms * 1e3:6.1f}us {fl / ms / 1e9:5.0f}" + (f" t{plan % 16}k{plan // 16}" if tile is None else ""))
    print(f"{name:32s} {fl / 1e9:7.2f} | " + " | ".join(f"{c:>13s}" for c in cells))
os.environ.pop("EVMI_PK_TILE", None)
