#!/usr/bin/env python3
"""TEMP: forward training attention under ablation bits (1 no global refetch, 2 no LDS commit, 4 no exp, 8 no S MFMAs, 16 no PV MFMAs)."""
import ctypes
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch  # noqa: E402

from everyvoice_amd import _lib  # noqa: E402
from everyvoice_amd.train import ops  # noqa: E402
from fs2_bench import synthetic_batch  # noqa: E402

dev = torch.device("cuda:0")
ops.CONV_BACKEND["operands"] = "bf16"
lib = _lib.load()
B, D, H = 32, 256, 2
_, lens, _, T_i = synthetic_batch(B, 1234)
T = int(T_i.max())
l32 = T_i.to(dev, torch.int32).contiguous()
qkv = torch.randn(3 * D, B, T, device=dev)
for abl in [0, 1, 2, 4, 8, 16, 3, 7, 24, 31, 0]:
    lib.evmi_debug_attn_ablation(ctypes.c_int(abl))
    for _ in range(3):
        ops.attention_train_fwd(qkv, l32, H, 0.0, 5)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        ops.attention_train_fwd(qkv, l32, H, 0.0, 5)
    e1.record()
    torch.cuda.synchronize()
    print(f"abl {abl:2d}: {e0.elapsed_time(e1)/20*1e3:.0f} us")
