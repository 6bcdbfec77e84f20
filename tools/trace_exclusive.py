#!/usr/bin/env python3
"""From a rocprofv3 kernel trace csv of training steps: for the LAST full step (between two bursts of the boundary kernel), how
the wall time splits by the number of kernels running (0, 1, 2, 3+), and per kernel family the wall time during which it ran ALONE
-- the serial part of the step, i.e. what a faster or fused version of that family would take off the step directly.
usage: trace_exclusive.py trace.csv [boundary kernel substring = optimizer_step]"""
import csv
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
mark = sys.argv[2] if len(sys.argv) > 2 else "optimizer_step"
short = lambda n: n.split("(")[0].replace("void ", "").replace("evmi::", "")[:64]  # noqa: E731
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"])) for r in rows)
ends = [e for s, e, n in ev if mark in n]
bounds = [ends[i] for i in range(len(ends)) if i + 1 == len(ends) or ends[i + 1] - ends[i] > 2e6]
if len(bounds) < 2:
    sys.exit("fewer than two step boundaries found")
t0, t1 = bounds[-2], bounds[-1]
step = [(s, e, n) for s, e, n in ev if s >= t0 and e <= t1]
pts = []
for i, (s, e, n) in enumerate(step):
    pts.append((s, 1, i))
    pts.append((e, -1, i))
pts.sort()
live, last = set(), t0
by_conc = defaultdict(int)
alone, alone_n, total = defaultdict(int), defaultdict(int), defaultdict(int)
for t, d, i in pts:
    dt = t - last
    if dt > 0:
        by_conc[min(len(live), 3)] += dt
        if len(live) == 1:
            alone[step[next(iter(live))][2]] += dt
    last = t
    if d > 0:
        live.add(i)
    else:
        live.discard(i)
by_conc[0] += t1 - last
for s, e, n in step:
    total[n] += e - s
    alone_n[n] += 1
wall = t1 - t0
print(f"step {wall / 1e6:.2f} ms wall, {len(step)} launches; wall time with 0 / 1 / 2 / 3+ kernels running: "
      + " / ".join(f"{by_conc[c] / 1e6:.2f}" for c in range(4)) + " ms")
print(f"{'kernel family':64s} {'launches':>8s} {'alone ms':>9s} {'summed ms':>9s}")
for n, t in sorted(alone.items(), key=lambda kv: -kv[1])[:45]:
    print(f"{n:64s} {alone_n[n]:8d} {t / 1e6:9.3f} {total[n] / 1e6:9.3f}")
