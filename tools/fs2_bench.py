#!/usr/bin/env python3
"""FastSpeech2 inference on one GPU: LJSpeech-shaped synthetic batch (SURVEY.md 8d C3 shapes), mel frames / s."""
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch  # noqa: E402

from everyvoice_amd.fs2 import FastSpeech2  # noqa: E402


def synthetic_batch(B=32, seed=1234):
    g = torch.Generator().manual_seed(seed)
    L_i = torch.clamp(torch.round(torch.normal(99.9, 33.95, (B,), generator=g)), 12, 187).long()
    T_i = torch.maximum(L_i, torch.clamp(torch.round(5.67 * L_i + torch.normal(0.0, 20.0, (B,), generator=g)), max=947).long())
    L = int(L_i.max())
    ids = torch.randint(2, 80, (B, L), generator=g)
    ids = ids.masked_fill(torch.arange(L)[None] >= L_i[:, None], 0)
    durs = torch.zeros(B, L, dtype=torch.long)
    for b in range(B):  # positive durations summing to T_i
        n = int(L_i[b])
        extra = torch.multinomial(torch.ones(n), int(T_i[b]) - n, replacement=True, generator=g) if T_i[b] > n else torch.empty(0, dtype=torch.long)
        durs[b, :n] = 1 + torch.bincount(extra, minlength=n)
    return ids, L_i, durs, T_i


def forward_flops(lens, mel_lens, L_pad, T_pad, B, d=256, f=1024, k=9, heads=2, layers=4, n_mels=80, pn_ch=512, pn_k=5, vp_layers=5, vp_k=3):
    """Algorithmic FLOPs of one inference forward at the default sizes (2 per multiply-add).  Dense layers are computed on the
    padded [B, L_pad] / [B, T_pad] grids (as the reference does); attention per item over its padded queries and real keys."""
    def conformer(n_cols, q_len, k_lens):
        dense = 2 * (2 * d * f) + (3 * d * d + d * d) + (2 * d * d + d * d) + k * d  # FFN x2, QKV + out, pointwise x2, depthwise
        attn = sum(2 * q_len * int(kl) * d for kl in k_lens)                           # QK^T and PV over all heads
        return layers * (n_cols * dense + attn)
    enc = conformer(B * L_pad, L_pad, lens)
    dec = conformer(B * T_pad, T_pad, mel_lens)
    predictors = 3 * B * L_pad * (vp_layers * (vp_k * d + d * d) + d)
    mel = B * T_pad * d * n_mels
    postnet = B * T_pad * pn_k * (n_mels * pn_ch + 3 * pn_ch * pn_ch + pn_ch * n_mels)
    return 2.0 * (enc + dec + predictors + mel + postnet)


def main():
    dev = torch.device("cuda:0")
    import os
    model = FastSpeech2(device=dev, precision=os.environ.get("OPERANDS", "f32")).init_random(1234)
    ids, lens, durs, T_i = synthetic_batch()
    ids, lens, durs = ids.to(dev), lens.to(dev), durs.to(dev)
    for _ in range(3):
        out = model(ids, lens, durations=durs)
    torch.cuda.synchronize()
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 10
    t0 = time.perf_counter()
    for _ in range(n):
        out = model(ids, lens, durations=durs)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    frames = int(T_i.sum())
    fl = forward_flops(lens.cpu(), T_i, ids.shape[1], out[0].shape[1], ids.shape[0])
    print(f"algorithmic {fl/1e9:.1f} GFLOP per batch -> {fl/dt/1e12:.1f} TFLOP/s ({fl/dt/157e12*100:.1f} % of the 157 TFLOP/s fp32 matrix peak)")
    print(f"B={ids.shape[0]} L={ids.shape[1]} T={out[0].shape[1]} frames={frames}: {dt*1e3:.2f} ms / batch -> {frames/dt/1e6:.3f} M mel frames/s, "
          f"{ids.shape[0]/dt:.0f} utterances/s, {frames*256/22050/dt:.0f} x real time")


if __name__ == "__main__":
    main()
