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


def main():
    dev = torch.device("cuda:0")
    model = FastSpeech2(device=dev).init_random(1234)
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
    print(f"B={ids.shape[0]} L={ids.shape[1]} T={out[0].shape[1]} frames={frames}: {dt*1e3:.2f} ms / batch -> {frames/dt/1e6:.3f} M mel frames/s, "
          f"{ids.shape[0]/dt:.0f} utterances/s, {frames*256/22050/dt:.0f} x real time")


if __name__ == "__main__":
    main()
