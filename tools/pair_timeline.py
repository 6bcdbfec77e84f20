#!/usr/bin/env python3
"""s_memtime timeline of workgroup 0 of the fused residual-pair kernel (debug instantiation in bench_kernels.hip).
Stamps per tile: tile start, x committed, one per weight step (2 * NG), loops done, stores issued."""
import ctypes as C
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch  # noqa: E402,F401

from everyvoice_amd import _lib  # noqa: E402

_lib.load()
lib = C.CDLL(str(Path(__file__).resolve().parent / "microbench" / "libevmi_bench.so"))
T_BY_C = {64: 98304, 32: 196608}
torch.zeros(1, device="cuda")
for spec in sys.argv[1:]:  # c64k11 ...
    c, ks = (int(x) for x in spec[1:].split("k"))
    for dil in (1, 5):
        buf = (C.c_longlong * 256)()
        rc = lib.evmi_debug_pair_timeline(c, ks, dil, 32, T_BY_C[c], buf, 256)
        if rc:
            print(spec, "FAILED", rc)
            continue
        st = [x for x in buf if x]
        d = [st[i + 1] - st[i] for i in range(len(st) - 1)]
        print(f"== {spec} dil={dil}: {len(st)} stamps, first 3 tiles:")
        print("  ", " ".join(str(x) for x in d[:90]))
