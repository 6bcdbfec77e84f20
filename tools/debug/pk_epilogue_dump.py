#!/usr/bin/env python3
"""Outputs of the packed convolution's fused forms on fixed inputs -> gpurun_out/pk_<tag>.pt (compare two library builds bitwise)."""
import os
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
import torch  # noqa: E402

from everyvoice_amd.train import ops  # noqa: E402

dev = torch.device("cuda:0")
ops.CONV_BACKEND["operands"] = "bf16"
g = torch.Generator().manual_seed(3)
out = {}
for name, (C, B, T, k, dil) in {"a": (128, 2, 256, 3, 1), "b": (64, 2, 512, 7, 3), "c": (32, 2, 1024, 11, 5), "d": (256, 2, 64, 3, 1)}.items():
    x = torch.randn(C, B, T, generator=g).to(dev)
    w = (torch.randn(C, C, k, generator=g) * 0.05).to(dev)
    bias = torch.randn(C, generator=g).to(dev)
    res = torch.randn(C, B, T, generator=g).to(dev)
    pad = dil * (k - 1) // 2
    out[name + "_fwd"] = ops.conv1d_fused_fwd(x, w, bias, 1, pad, dil, 1, pre_slope=0.1, residual=res).cpu()
    out[name + "_fwd_plain"] = ops.conv1d_fwd(x, w, bias, 1, pad, dil, 1).cpu()
    dy = torch.randn(C, B, T, generator=g).to(dev)
    out[name + "_dgrad"] = ops.conv1d_fused_dgrad(dy, w, T, 1, pad, dil, 1, dy_mask=x, dy_mask_slope=0.1, dx_mask=res, dx_mask_slope=0.1, residual=x).cpu()
    out[name + "_dgrad_plain"] = ops.conv1d_bwd_data_mfma(dy, w, T, 1, pad, dil, 1).cpu()
torch.save(out, f"gpurun_out/pk_{sys.argv[1]}.pt")
