#!/usr/bin/env python3
"""LayerNorm forward / backward (channels-first [C][B][T] fp32, train/ops.py) timed alone at the FastSpeech2 step's two sizes
(decoder: 32 x 814 columns, encoder: 32 x 150) -- per call, HIP events, median of 5 rounds of 20 calls."""
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
import torch  # noqa: E402

from everyvoice_amd.train import ops  # noqa: E402

dev = torch.device("cuda:0")
for B, T in ((32, 814), (32, 150)):
    C = 256
    x = torch.randn(C, B, T, device=dev)
    dy = torch.randn(C, B, T, device=dev)
    g, b = torch.ones(C, device=dev), torch.zeros(C, device=dev)
    dg, db = torch.zeros(C, device=dev), torch.zeros(C, device=dev)

    def timed(fn, n=20, rounds=5):
        fn()
        torch.cuda.synchronize()
        res = []
        for _ in range(rounds):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(n):
                fn()
            e1.record()
            torch.cuda.synchronize()
            res.append(e0.elapsed_time(e1) / n * 1e3)
        return sorted(res)[len(res) // 2]

    mb = C * B * T * 4 / 1e6
    t_f = timed(lambda: ops.layernorm(x, g, b))
    t_b = timed(lambda: ops.layernorm_bwd(x, g, dy, dg, db))
    t_e = timed(lambda: ops.axpby(1.0, x, 1.0, dy))
    acc = torch.zeros_like(x)
    t_a = timed(lambda: ops.layernorm_bwd(x, g, dy, dg, db, acc_into=acc))
    print(f"{B} x {T} ({mb:.1f} MB per tensor): layernorm {t_f:.1f} us ({2 * mb / t_f:.2f} TB/s), backward {t_b:.1f} us ({3 * mb / t_b:.2f} TB/s), "
          f"backward adding into the gradient {t_a:.1f} us ({4 * mb / t_a:.2f} TB/s), a + b {t_e:.1f} us ({3 * mb / t_e:.2f} TB/s)")
