#!/usr/bin/env python3
"""LayerNorm forward / backward over channels at the FastSpeech2 decoder's shape ([256][32][814] fp32)."""
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
import torch  # noqa: E402

from everyvoice_amd.train import ops  # noqa: E402

dev = torch.device("cuda:0")
for C, B, T in ((256, 32, 814), (256, 32, 141)):
    x, dy = torch.randn(C, B, T, device=dev), torch.randn(C, B, T, device=dev)
    g, b = torch.randn(C, device=dev), torch.randn(C, device=dev)
    dg, db = torch.zeros(C, device=dev), torch.zeros(C, device=dev)
    for _ in range(3):
        ops.layernorm(x, g, b)
        ops.layernorm_bwd(x, g, dy, dg, db)
    torch.cuda.synchronize()
    e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    e[0].record()
    for _ in range(50):
        ops.layernorm(x, g, b)
    e[1].record()
    for _ in range(50):
        ops.layernorm_bwd(x, g, dy, dg, db)
    e[2].record()
    torch.cuda.synchronize()
    mb = C * B * T * 4 / 1e6
    tf, tb = e[0].elapsed_time(e[1]) / 50 * 1e3, e[1].elapsed_time(e[2]) / 50 * 1e3
    print(f"[{C}][{B}][{T}]: fwd {tf:.1f} us ({2*mb/tf/1e3:.2f} TB/s)  bwd {tb:.1f} us ({3*mb/tb/1e3:.2f} TB/s)")
