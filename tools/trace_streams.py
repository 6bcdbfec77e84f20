#!/usr/bin/env python3
"""From a rocprofv3 kernel trace csv of training steps: the LAST full step split by hardware queue (= stream).  Per queue: launches,
busy time, and per kernel family the launches, the run time and the idle time of the queue in front of those launches (what the
queue waited for: a dependency on another queue, or the dispatch latency of a short launch).  The queue with the most run time is
the step's critical chain: run + idle of its rows add up to the step.
usage: trace_streams.py trace.csv [boundary kernel substring = optimizer_step] [rows per queue = 40]"""
import csv
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
mark = sys.argv[2] if len(sys.argv) > 2 else "optimizer_step"
top = int(sys.argv[3]) if len(sys.argv) > 3 else 40
short = lambda n: n.split("(")[0].replace("void ", "").replace("evmi::", "")[:72]  # noqa: E731
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"]), r["Queue_Id"]) for r in rows)
ends = [e for s, e, n, q in ev if mark in n]
bounds = [ends[i] for i in range(len(ends)) if i + 1 == len(ends) or ends[i + 1] - ends[i] > 2e6]
if len(bounds) < 2:
    sys.exit("fewer than two step boundaries found")
t0, t1 = bounds[-2], bounds[-1]
step = [x for x in ev if x[0] >= t0 and x[1] <= t1]
print(f"step {(t1 - t0) / 1e6:.2f} ms wall, {len(step)} launches")
by_q = defaultdict(list)
for x in step:
    by_q[x[3]].append(x)
for q, xs in sorted(by_q.items(), key=lambda kv: -sum(e - s for s, e, _, _ in kv[1])):
    run = sum(e - s for s, e, _, _ in xs)
    fam = defaultdict(lambda: [0, 0, 0])
    prev = t0
    for s, e, n, _ in xs:
        f = fam[n]
        f[0] += 1
        f[1] += e - s
        f[2] += max(0, s - prev)
        prev = max(prev, e)
    idle = sum(f[2] for f in fam.values())
    print(f"== queue {q}: {len(xs)} launches, run {run / 1e6:.2f} ms, idle in front of its launches {idle / 1e6:.2f} ms, "
          f"first start {(xs[0][0] - t0) / 1e6:.2f} ms, last end {(max(e for _, e, _, _ in xs) - t0) / 1e6:.2f} ms")
    print(f"   {'kernel':72s} {'n':>5s} {'run us':>9s} {'idle us':>9s} {'avg us':>7s}")
    for n, (c, r, i) in sorted(fam.items(), key=lambda kv: -(kv[1][1] + kv[1][2]))[:top]:
        print(f"   {n:72s} {c:5d} {r / 1e3:9.1f} {i / 1e3:9.1f} {r / c / 1e3:7.1f}")
