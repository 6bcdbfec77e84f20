#!/usr/bin/env python3
"""Print the per-phase cycle timeline of workgroup 0 of the fused pair kernel."""
import ctypes as C
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch  # noqa: E402,F401

from everyvoice_amd import _lib  # noqa: E402

lib = _lib.load()
torch.zeros(1, device="cuda")
for c, ks, T in ((64, 11, 98304), (32, 11, 196608), (32, 3, 196608)):
    buf = (C.c_longlong * 256)()
    rc = lib.evmi_debug_pair_timeline(c, ks, 5, 32, T, buf, 256)
    if rc:
        print("failed", lib.evmi_last_error().decode()); continue
    st = [int(v) for v in buf if v]
    d = [b - a for a, b in zip(st, st[1:])]
    print(f"c{c} k{ks}: {len(st)} stamps; deltas (cycles):")
    print("  ", d[:80])
