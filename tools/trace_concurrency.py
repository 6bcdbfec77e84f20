#!/usr/bin/env python3
"""From a rocprofv3 kernel trace csv: wall time covered by at least one kernel, summed kernel time, average concurrency, launches,
and the per-family share of the summed time.  usage: trace_concurrency.py trace.csv [n_steps]"""
import csv
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 1
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows)
# keep the last `steps`/total fraction?  the caller passes the number of steps the trace covers
busy, cur_s, cur_e = 0, None, None
for s, e, _ in ev:
    if cur_e is None or s > cur_e:
        if cur_e is not None:
            busy += cur_e - cur_s
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
total = sum(e - s for s, e, _ in ev)
span = ev[-1][1] - ev[0][0]
print(f"launches {len(ev)} ({len(ev)/steps:.0f}/step)  span {span/1e6:.1f} ms  busy {busy/1e6:.1f} ms ({busy/steps/1e6:.2f}/step)  summed {total/1e6:.1f} ms ({total/steps/1e6:.2f}/step)  concurrency {total/busy:.2f}")
fam = defaultdict(lambda: [0, 0])
for s, e, n in ev:
    k = n.split("(")[0].replace("void ", "").replace("evmi::", "")[:60]
    fam[k][0] += 1
    fam[k][1] += e - s
for k, (c, t) in sorted(fam.items(), key=lambda kv: -kv[1][1])[:40]:
    print(f"{k:60s} {c/steps:8.1f}/step {t/steps/1e3:9.1f} us/step  avg {t/c/1e3:7.1f} us")
