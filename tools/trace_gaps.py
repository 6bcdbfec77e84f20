#!/usr/bin/env python3
"""From a rocprofv3 kernel trace csv of training steps: the LAST full step (between two launches of the optimiser kernel), its wall
time, the time at least one kernel runs, the idle gaps above a threshold with the kernels on either side, and the longest kernels.
usage: trace_gaps.py trace.csv [boundary kernel substring = optimizer_step] [gap us = 15]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
mark = sys.argv[2] if len(sys.argv) > 2 else "optimizer_step"
thr = float(sys.argv[3]) * 1e3 if len(sys.argv) > 3 else 15e3
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", "").replace("evmi::", "")[:70]) for r in rows)
ends = [e for s, e, n in ev if mark in n]
# steps end with the LAST optimiser launch of a burst: bursts separated by more than 2 ms
bounds = [ends[i] for i in range(len(ends)) if i + 1 == len(ends) or ends[i + 1] - ends[i] > 2e6]
if len(bounds) < 2:
    sys.exit("fewer than two step boundaries found")
t0, t1 = bounds[-2], bounds[-1]
step = [(s, e, n) for s, e, n in ev if s >= t0 and e <= t1]
busy, cs, ce, gaps = 0, None, None, []
last_name = "(step start)"
for s, e, n in step:
    if ce is None:
        cs, ce = s, e
        if s - t0 > thr:
            gaps.append((s - t0, "(previous step)", n, s - t0))
    elif s > ce:
        busy += ce - cs
        if s - ce > thr:
            gaps.append((s - ce, last_name, n, ce - t0))
        cs, ce = s, e
    else:
        ce = max(ce, e)
    if e >= ce:
        last_name = n
busy += ce - cs
total = sum(e - s for s, e, _ in step)
print(f"step {(t1 - t0) / 1e6:.2f} ms wall, {busy / 1e6:.2f} ms with a kernel running, summed {total / 1e6:.2f} ms, {len(step)} launches")
print(f"idle gaps above {thr / 1e3:.0f} us: {len(gaps)}, {sum(g[0] for g in gaps) / 1e6:.2f} ms in total")
for g, a, b, at in sorted(gaps, reverse=True)[:15]:
    print(f"  {g / 1e3:7.1f} us at {at / 1e6:6.2f} ms   after {a}   before {b}")
print("longest kernels:")
for s, e, n in sorted(step, key=lambda t: t[0] - t[1])[:12]:
    print(f"  {(e - s) / 1e3:8.1f} us  from {(s - t0) / 1e6:6.2f} ms  {n}")
