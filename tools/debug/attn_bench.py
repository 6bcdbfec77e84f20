#!/usr/bin/env python3
"""Training attention (forward, dQ, dK/dV) at the decoder's shape of BASELINE config 3, with and without probability dropout.
usage: python tools/debug/attn_bench.py   (env OPERANDS=bf16|f32)"""
import os
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch  # noqa: E402

from everyvoice_amd.train import ops  # noqa: E402
from fs2_bench import synthetic_batch  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    ops.CONV_BACKEND["operands"] = os.environ.get("OPERANDS", "bf16")
    B, D, H = 32, 256, 2
    _, lens, _, T_i = synthetic_batch(B, 1234)
    for name, ln in (("decoder", T_i), ("encoder", lens)):
        T = int(ln.max())
        l32 = ln.to(dev, torch.int32).contiguous()
        qkv = torch.randn(3 * D, B, T, device=dev)
        dout = torch.randn(D, B, T, device=dev)
        for p in (0.0, 0.1):
            out, saved = ops.attention_train_fwd(qkv, l32, H, p, 5)
            ops.attention_train_bwd(qkv, saved, dout, H, p, 5)
            torch.cuda.synchronize()
            e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
            n = 10
            e[0].record()
            for _ in range(n):
                out, saved = ops.attention_train_fwd(qkv, l32, H, p, 5)
            e[1].record()
            for _ in range(n):
                ops.attention_train_bwd(qkv, saved, dout, H, p, 5)
            e[2].record()
            torch.cuda.synchronize()
            fl = 4.0 * float((ln.double() ** 2).sum()) * D  # QK^T + PV over the live keys x live queries
            tf, tb = e[0].elapsed_time(e[1]) / n, e[1].elapsed_time(e[2]) / n
            print(f"{name} T={T} mean len {float(ln.float().mean()):.0f} p={p}: fwd {tf*1e3:.0f} us ({fl/tf/1e9:.1f} TF/s), bwd {tb*1e3:.0f} us ({2.5*fl/tb/1e9:.1f} TF/s)")


if __name__ == "__main__":
    main()
