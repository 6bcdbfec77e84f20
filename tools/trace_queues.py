#!/usr/bin/env python3
"""From a rocprofv3 kernel trace csv of training steps: the LAST full step in 1 ms bins -- the fraction of each bin every hardware
queue had a kernel running, and the commonest kernels started in the bin.  Shows which stream's work runs beside which (a side
stream whose launches all run behind the main chain shows as a tail of bins with one queue).
usage: trace_queues.py trace.csv [boundary kernel substring = optimizer_step]"""
import csv
import sys
from collections import Counter, defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
mark = sys.argv[2] if len(sys.argv) > 2 else "optimizer_step"
short = lambda n: n.split("(")[0].replace("void ", "").replace("evmi::", "")[:40]  # noqa: E731
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"]), r["Queue_Id"]) for r in rows)
ends = [e for s, e, n, q in ev if mark in n]
bounds = [ends[i] for i in range(len(ends)) if i + 1 == len(ends) or ends[i + 1] - ends[i] > 2e6]
if len(bounds) < 2:
    sys.exit("fewer than two step boundaries found")
t0, t1 = bounds[-2], bounds[-1]
step = [x for x in ev if x[0] >= t0 and x[1] <= t1]
bins = defaultdict(lambda: defaultdict(float))
for s, e, n, q in step:
    a, b = (s - t0) / 1e6, (e - t0) / 1e6
    i = int(a)
    while i < b:
        bins[i][q] += min(b, i + 1) - max(a, i)
        i += 1
queues = sorted({x[3] for x in step})
print(f"step {(t1 - t0) / 1e6:.2f} ms wall, {len(step)} launches, queues {queues}")
print("  ms | " + " ".join(f"q{q:>3s}" for q in queues) + " | kernels started in the bin")
for i in range(int((t1 - t0) / 1e6) + 1):
    c = Counter(x[2] for x in step if int((x[0] - t0) / 1e6) == i)
    print(f"{i:4d} | " + " ".join(f"{bins[i].get(q, 0.0):4.2f}" for q in queues) + " | " + ", ".join(f"{n} x{k}" for n, k in c.most_common(3)))
