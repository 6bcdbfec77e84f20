#!/usr/bin/env python3
"""Golden vectors of the FastSpeech2 forward path, generated from the build's own oracle (oracle/fs2_ref.py; the
reference's module is an absent submodule -> parity unpinned, see that file's header).  Self-contained: the small
model's parameters travel with the inputs and outputs (float16-exact values so that the file stays small).

    python tests/golden/make_fs2_golden.py      ->  tests/golden/fs2_small.npz
"""
import sys
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
from oracle.fs2_ref import FastSpeech2ConfigRef, FastSpeech2Ref, randomize_norm_stats_  # noqa: E402


def main():
    torch.manual_seed(20260101)
    cfg = FastSpeech2ConfigRef.small()
    m = FastSpeech2Ref(cfg).eval()
    g = torch.Generator().manual_seed(5)
    randomize_norm_stats_(m, g)
    with torch.no_grad():
        for n, p in m.named_parameters():
            if n.endswith("bias"):
                p.copy_(torch.randn(p.shape, generator=g) * 0.05)
        m.duration_predictor.linear.bias.fill_(1.2)
        # parameters rounded to fp16-representable values: the archive stores them in 2 bytes, exactly
        sd = {k: (v.half().float() if v.dtype == torch.float32 else v) for k, v in m.state_dict().items()}
        m.load_state_dict(sd)
    B, L = 3, 14
    lens = torch.tensor([14, 9, 11])
    ids = torch.randint(1, cfg.n_symbols, (B, L), generator=g)
    ids = ids.masked_fill(torch.arange(L)[None] >= lens[:, None], 0)
    durs = torch.randint(0, 6, (B, L), generator=g)
    durs[:, 0] += 1
    out = {}
    teacher = m(ids, lens, durations=durs)
    free = m(ids, lens, duration_control=1.0, pitch_control=1.2, energy_control=0.9)
    for tag, o in (("tf", teacher), ("free", free)):
        for name, t in zip(("mel", "post", "durations", "pitch", "energy", "mel_lens"), o):
            out[f"{tag}_{name}"] = t.numpy()
    arrays = {"ids": ids.numpy(), "lens": lens.numpy(), "given_durations": durs.numpy(), **out}
    for k, v in sd.items():
        arrays["param:" + k] = v.numpy().astype(np.float16) if v.dtype == torch.float32 else v.numpy()
    path = ROOT / "tests" / "golden" / "fs2_small.npz"
    np.savez_compressed(path, **arrays)
    print(path, path.stat().st_size, "bytes;", sum(v.numel() for v in sd.values()), "parameters")


if __name__ == "__main__":
    main()
