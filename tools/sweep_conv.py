#!/usr/bin/env python3
"""Sweep conv_tc tuning variants / ablations (bench_kernels.hip) at the bench shapes."""
import ctypes as C
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch  # noqa: E402,F401  (initialises the HIP runtime the same way the product does)

from everyvoice_amd import _lib  # noqa: E402

lib = _lib.load()
lib.evmi_bench_variant_name.restype = C.c_char_p
lib.evmi_bench_conv_tc.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_int,
                                   C.POINTER(C.c_float), C.POINTER(C.c_double)]
T_BY_C = {256: 6144, 128: 49152, 64: 98304, 32: 196608}
only = sys.argv[1:] 
torch.zeros(1, device="cuda")
names = [lib.evmi_bench_variant_name(i).decode() for i in range(lib.evmi_bench_num_variants())]
print(f"{'variant':28s} {'dil':>3s} {'pre':>4s} {'res':>3s} {'ms':>8s} {'TF/s':>8s}")
for n in names:
    if only and not any(o in n for o in only):
        continue
    c = int(n[1:].split("k")[0])
    for dil, pre, res in (((5, 0.1, 0), (1, 1.0, 1)) if "_md1" not in n else ((1, 0.1, 0), (1, 1.0, 1))):
        ms, fl = C.c_float(), C.c_double()
        rc = lib.evmi_bench_conv_tc(n.encode(), 32, T_BY_C[c], 0, dil, res, pre, 5, C.byref(ms), C.byref(fl))
        if rc:
            print(n, "FAILED", lib.evmi_last_error().decode())
            continue
        print(f"{n:28s} {dil:3d} {pre:4.1f} {res:3d} {ms.value:8.4f} {fl.value / ms.value / 1e9:8.1f}")
