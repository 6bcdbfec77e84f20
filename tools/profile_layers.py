#!/usr/bin/env python3
"""Per-kernel-family table of one generator forward (HIP-event timed through the C ABI)."""
import argparse
import sys
from collections import defaultdict
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch  # noqa: E402

import bench  # noqa: E402

p = argparse.ArgumentParser()
p.add_argument("--batch", type=int, default=32)
p.add_argument("--frames", type=int, default=768)
p.add_argument("--passes", type=int, default=3)
p.add_argument("--layers", action="store_true")
a = p.parse_args()
dev = torch.device("cuda:0")
model = bench.upstream_init_generator("bf16").to(dev).eval()
mel = bench.synthetic_mel(a.batch, a.frames, 1234).to(dev)
gen = model.generator
gen(mel)
tot = defaultdict(lambda: [0, 0.0, 0.0, 0.0])
for _ in range(a.passes):
    _, recs = gen.forward_profiled(mel)
    for r in recs:
        t = tot[(r["kernel"], r["layer"]) if a.layers else r["kernel"]]
        t[0] += 1; t[1] += r["ms"]; t[2] += r["flops"]; t[3] += r["bytes"]
allms = sum(t[1] for t in tot.values()) / a.passes
print(f"{'kernel':60s} {'n':>4s} {'ms/fwd':>8s} {'avg ms':>8s} {'TF/s':>8s} {'GB/s':>8s} {'%':>6s}")
for k, t in sorted(tot.items(), key=lambda kv: -kv[1][1]):
    n, ms, fl, by = t
    print(f"{str(k):60s} {n // a.passes:4d} {ms / a.passes:8.3f} {ms / n:8.4f} {fl / ms / 1e9:8.1f} {by / ms / 1e6:8.1f} {100 * ms / a.passes / allms:6.1f}")
print(f"total event ms / forward: {allms:.3f}")
