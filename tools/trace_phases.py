#!/usr/bin/env python3
"""Phases of ONE replayed GAN training step out of a rocprofv3 kernel trace (tools/prof_train.sh): wall time, summed kernel time,
launches and the largest kernels of [generator forward | discriminator step | discriminator update + generator-step discriminator
pass | generator backward], delimited by the step's own marker kernels (optimizer steps, the generator's tanh forward / backward).
usage: python tools/trace_phases.py gpurun_out/<tag>/<tag>_kernel_trace.csv"""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
opt = [i for i, r in enumerate(rows) if "optimizer_step_kernel" in r["Kernel_Name"]]
step = rows[opt[-3] + 1: opt[-1] + 1]  # between the generator update of the previous step and this step's
t0 = int(step[0]["Start_Timestamp"])
ms = lambda r, key="Start_Timestamp": (int(r[key]) - t0) / 1e6  # noqa: E731
span = max(ms(r, "End_Timestamp") for r in step)
iv = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in step)
busy, (cs, ce) = 0, iv[0]
for s, e in iv[1:]:
    if s > ce:
        busy += ce - cs
        cs, ce = s, e
    else:
        ce = max(ce, e)
busy += ce - cs
ksum = sum(e - s for s, e in iv) / 1e6
print(f"step: {span:.2f} ms wall, device busy {busy / 1e6:.2f} ms, summed kernel time {ksum:.2f} ms (concurrency {ksum / (busy / 1e6):.2f}), {len(step)} launches")
tanh_f = next(ms(r) for r in step if "ew_kernel<2>" in r["Kernel_Name"])
tanh_b = next(ms(r) for r in step if "ew_kernel<3>" in r["Kernel_Name"])
d_opt = next(ms(r) for r in step if "optimizer_step_kernel" in r["Kernel_Name"])
bounds = [("generator forward", 0.0, tanh_f), ("discriminator step (forward + backward, 2B items)", tanh_f, d_opt),
          ("discriminator update + generator-step discriminator pass + losses", d_opt, tanh_b), ("generator backward + update", tanh_b, span + 1)]
for label, lo, hi in bounds:
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in step:
        if lo <= ms(r) < hi:
            n = r["Kernel_Name"].split("(")[0].replace("void evmi::", "").replace("evmi::", "")[:64]
            agg[n][0] += 1
            agg[n][1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
    print(f"== {label}: {lo:.2f} -> {min(hi, span):.2f} ms ({min(hi, span) - lo:.2f} ms wall), summed kernel time {sum(v[1] for v in agg.values()):.2f} ms, "
          f"{sum(v[0] for v in agg.values())} launches")
    for n, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:12]:
        print(f"     {n:64s} {v[0]:4d} {v[1]:7.3f} ms")
