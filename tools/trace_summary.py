#!/usr/bin/env python3
"""Average duration per kernel (and grid) from a rocprofv3 kernel trace csv."""
import csv
import sys
from collections import defaultdict

acc = defaultdict(lambda: [0, 0.0])
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Kernel_Name"]
    if len(sys.argv) > 2 and sys.argv[2] not in n:
        continue
    key = (n[:70], r.get("Grid_Size_X", ""), r.get("Grid_Size_Y", ""))
    acc[key][0] += 1
    acc[key][1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
for k, v in acc.items():
    print(f"{k[0]:70s} grid {k[1]:>7s}x{k[2]:<4s} calls {v[0]:4d} avg {v[1]/v[0]:9.1f} us")
