#!/usr/bin/env python3
"""Per-shape census of the fp32 matrix-core convolutions of one GAN training step (calls, time, TFLOP/s)."""
import sys
from collections import defaultdict
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch  # noqa: E402

from everyvoice_amd.spectral import MelSpectrogram  # noqa: E402
from everyvoice_amd.train import ops  # noqa: E402
from everyvoice_amd.train.hifigan import HiFiGANTrainer  # noqa: E402

import os
ops.CONV_BACKEND.update(fwd="mfma", dgrad="mfma", operands=os.environ.get("OPERANDS", "f32"))
dev = torch.device("cuda:0")
B, S = 16, 8192
g = torch.Generator().manual_seed(1234)
y = (0.3 * torch.tanh(torch.randn(B, 1, S, generator=g))).to(dev)
mel = MelSpectrogram()(y.squeeze(1), log=True)[:, :, : S // 256].contiguous()
tr = HiFiGANTrainer(device=dev)
for _ in range(2):
    tr.training_step(mel, y)
torch.cuda.synchronize()

stats = defaultdict(lambda: [0, 0.0, 0.0])
orig = ops.conv1d_mfma


def timed(x, w, bias, stride=1, pad=0, dil=1, groups=1, out=None, n_out=None, out_stride=1, out_offset=0, accumulate=False, **kw):
    cin, Bx, t_in = x.shape
    cout, cin_g, k = w.shape
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    r = orig(x, w, bias, stride, pad, dil, groups, out, n_out, out_stride, out_offset, accumulate, **kw)
    e1.record()
    e1.synchronize()
    n = r.shape[2] if n_out is None else n_out
    key = (cin, cout, k, stride, dil, groups, Bx, t_in, n, out_stride)
    st = stats[key]
    st[0] += 1
    st[1] += e0.elapsed_time(e1)
    st[2] += 2.0 * Bx * n * cout * cin_g * k
    return r


ops.conv1d_mfma = timed
tr.training_step(mel, y)
torch.cuda.synchronize()
tot = sum(v[1] for v in stats.values())
print(f"{'cin':>5} {'cout':>5} {'k':>3} {'s':>2} {'d':>2} {'g':>3} {'B':>4} {'t_in':>6} {'n_out':>6} {'os':>2} | calls   ms    TF/s")
for key, v in sorted(stats.items(), key=lambda kv: -kv[1][1]):
    print("%5d %5d %3d %2d %2d %3d %4d %6d %6d %2d | %4d %7.3f %6.1f" % (*key, v[0], v[1], v[2] / v[1] / 1e9))
print(f"total {tot:.2f} ms, {sum(v[0] for v in stats.values())} calls, {sum(v[2] for v in stats.values())/1e12:.3f} TFLOP")
