#!/usr/bin/env python3
"""s_memtime timeline of one workgroup of a conv_tc debug variant (bench_kernels.hip, ABL bit 128): per-wave cycle deltas."""
import ctypes as C
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch  # noqa: E402,F401

from everyvoice_amd import _lib  # noqa: E402

_lib.load()
lib = C.CDLL(str(Path(__file__).resolve().parent / "microbench" / "libevmi_bench.so"))
T_BY_C = {256: 6144, 128: 49152, 64: 98304, 32: 196608}
torch.zeros(1, device="cuda")
for name in sys.argv[1:]:
    c = int(name[1:].split("k")[0])
    for dil, pre, res in ((5, 0.1, 0), (1, 1.0, 1)):
        buf = (C.c_longlong * (8 * 128))()
        rc = lib.evmi_bench_conv_tc_timeline(name.encode(), 32, T_BY_C[c], dil, res, C.c_float(pre), buf)
        if rc:
            print(name, "FAILED", rc)
            continue
        print(f"== {name} dil={dil} pre={pre} res={res}")
        for w in range(8):
            st = [buf[w * 128 + i] for i in range(128)]
            st = [x for x in st if x]
            if not st:
                continue
            d = [st[i + 1] - st[i] for i in range(len(st) - 1)]
            print(f" wave {w}: total {st[-1] - st[0]}  start_offset {st[0] - min(buf[v * 128] for v in range(8) if buf[v * 128])}")
            # stamp order: start | per chunk: X tile written, then per step (barrier passed, MFMAs issued) | loop done, staged, stores issued
            print("   deltas:", " ".join(str(x) for x in d))
