#!/usr/bin/env python3
"""Sweep conv_tc tuning variants / ablations (bench_kernels.hip) at the bench shapes.
Variants are interleaved over several rounds in ONE process and the median is reported (run-to-run and
clock noise is several percent: single numbers from separate runs are not comparable)."""
import ctypes as C
import statistics
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch  # noqa: E402,F401  (initialises the HIP runtime the same way the product does)

from everyvoice_amd import _lib  # noqa: E402

_lib.load()  # the product library first (the bench library links against it)
_bench = Path(__file__).resolve().parent / "microbench" / "libevmi_bench.so"
if not _bench.exists():
    sys.exit("build the tuning variants first: make bench-kernels")
lib = C.CDLL(str(_bench))
lib.evmi_bench_variant_name.restype = C.c_char_p
lib.evmi_bench_conv_tc.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_int,
                                   C.POINTER(C.c_float), C.POINTER(C.c_double)]
T_BY_C = {256: 6144, 128: 49152, 64: 98304, 32: 196608}
args = [a for a in sys.argv[1:] if not a.startswith("--")]
rounds = next((int(a.split("=")[1]) for a in sys.argv[1:] if a.startswith("--rounds=")), 1)
exact = "--exact" in sys.argv
torch.zeros(1, device="cuda")
names = [lib.evmi_bench_variant_name(i).decode() for i in range(lib.evmi_bench_num_variants())]
names = [n for n in names if not args or (n in args if exact else any(o in n for o in args))]
res = {}
for _ in range(rounds):
    for n in names:
        c = int(n[1:].split("k")[0])
        cases = ((5, 0.1, 0), (1, 1.0, 1)) if "_md1" not in n else ((1, 0.1, 0), (1, 1.0, 1))
        for dil, pre, r in cases:
            ms, fl = C.c_float(), C.c_double()
            rc = lib.evmi_bench_conv_tc(n.encode(), 32, T_BY_C[c], 0, dil, r, pre, 5, C.byref(ms), C.byref(fl))
            if rc:
                print(n, "FAILED", lib.evmi_last_error().decode())
                continue
            res.setdefault((n, dil, pre, r), []).append((ms.value, fl.value))
print(f"{'variant':28s} {'dil':>3s} {'pre':>4s} {'res':>3s} {'ms(med)':>8s} {'min':>8s} {'TF/s':>8s}   rounds={rounds}")
for (n, dil, pre, r), v in res.items():
    ms = statistics.median(x[0] for x in v)
    print(f"{n:28s} {dil:3d} {pre:4.1f} {r:3d} {ms:8.4f} {min(x[0] for x in v):8.4f} {v[0][1] / ms / 1e9:8.1f}")
