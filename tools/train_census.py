#!/usr/bin/env python3
"""Per-shape census of every convolution call of one GAN training step (forward, input gradient, weight gradient): calls, summed ms
(each call timed alone with events: no overlap between streams), TFLOP/s.  Eager step; OPERANDS=bf16|f32.
usage: python tools/train_census.py [top_n]"""
import os
import sys
from collections import defaultdict
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch  # noqa: E402

from everyvoice_amd.spectral import MelSpectrogram  # noqa: E402
from everyvoice_amd.train import ops  # noqa: E402
from everyvoice_amd.train.hifigan import HiFiGANTrainer  # noqa: E402

dev = torch.device("cuda:0")
B, S = 16, 8192
g = torch.Generator().manual_seed(1234)
y = (0.3 * torch.tanh(torch.randn(B, 1, S, generator=g))).to(dev)
mel = MelSpectrogram()(y.squeeze(1), log=True)[:, :, : S // 256].contiguous()
tr = HiFiGANTrainer(device=dev, precision=os.environ.get("OPERANDS", "bf16"), use_graph=False, parallel_streams=False)
for _ in range(2):
    tr.training_step(mel, y)
torch.cuda.synchronize()
ops.SIDE_WGRAD["on"] = False

stats = defaultdict(lambda: [0, 0.0, 0.0])
depth = [0]


def wrap(name, shape_of):
    orig = getattr(ops, name)

    def timed(*a, **kw):
        if depth[0]:
            return orig(*a, **kw)
        depth[0] += 1
        try:
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            r = orig(*a, **kw)
            e1.record()
            e1.synchronize()
        finally:
            depth[0] -= 1
        key, fl = shape_of(*a, **kw)
        st = stats[key]
        st[0] += 1
        st[1] += e0.elapsed_time(e1)
        st[2] += fl
        return r

    setattr(ops, name, timed)


def fwd_shape(x, w, bias, stride=1, pad=0, dil=1, groups=1, *a, **kw):
    cin, Bx, t = x.shape
    cout, cg, k = w.shape
    n = ops.conv_out_len(t, k, stride, pad, dil)
    return ("fwd", cin, cout, k, stride, dil, groups, Bx, t), 2.0 * Bx * n * cout * cg * k


def dgrad_shape(dy, w, t_in, stride=1, pad=0, dil=1, groups=1, *a, **kw):
    cout, Bx, n = dy.shape
    _, cg, k = w.shape
    return ("dgrad", cg * groups, cout, k, stride, dil, groups, Bx, t_in), 2.0 * Bx * n * cout * cg * k


def wgrad_shape(x, w_shape, dy, dw_out, stride=1, pad=0, dil=1, groups=1, *a, **kw):
    cin, Bx, t = x.shape
    cout, cg, k = w_shape
    return ("wgrad", cin, cout, k, stride, dil, groups, Bx, t), 2.0 * Bx * dy.shape[2] * cout * cg * k


def bwd_shape(x, w, dy, stride=1, pad=0, dil=1, groups=1, need_dx=True, dw_out=None, db_out=None, accumulate=False, need_dw=True):
    cin, Bx, t = x.shape
    cout, cg, k = w.shape
    kind = "bwd:" + ("dx" if need_dx else "") + ("dw" if need_dw else "")
    return (kind, cin, cout, k, stride, dil, groups, Bx, t), 2.0 * Bx * dy.shape[2] * cout * cg * k * (int(need_dx) + int(need_dw))


wrap("conv1d_fused_fwd", fwd_shape)
wrap("conv1d_fused_dgrad", dgrad_shape)
wrap("conv1d_fused_wgrad", wgrad_shape)
wrap("conv1d_fwd", fwd_shape)
wrap("conv1d_bwd", bwd_shape)
wrap("conv1d_mfma", fwd_shape)
tr.training_step(mel, y)
torch.cuda.synchronize()
tot = sum(v[1] for v in stats.values())
top = int(sys.argv[1]) if len(sys.argv) > 1 else 60
print(f"{'kind':>7} {'cin':>5} {'cout':>5} {'k':>3} {'s':>2} {'d':>2} {'g':>3} {'B':>4} {'t_in':>6} | calls     ms  us/call   TF/s")
by_kind = defaultdict(lambda: [0.0, 0.0])
for key, v in sorted(stats.items(), key=lambda kv: -kv[1][1]):
    by_kind[key[0]][0] += v[1]
    by_kind[key[0]][1] += v[2]
for key, v in sorted(stats.items(), key=lambda kv: -kv[1][1])[:top]:
    print("%7s %5d %5d %3d %2d %2d %3d %4d %6d | %4d %7.3f %8.1f %6.1f" % (*key, v[0], v[1], v[1] / v[0] * 1e3, v[2] / v[1] / 1e9))
print(f"total {tot:.2f} ms over {sum(v[0] for v in stats.values())} calls, {sum(v[2] for v in stats.values()) / 1e12:.3f} TFLOP")
for k, (ms, fl) in by_kind.items():
    print(f"  {k:8s} {ms:7.2f} ms {fl / 1e12:6.3f} TFLOP -> {fl / ms / 1e9:6.1f} TF/s")
