#!/usr/bin/env python3
"""Host-side cost of the per-launch helpers (stream pointer, allocation, a trivial launch)."""
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
import torch  # noqa: E402

from everyvoice_amd import _lib  # noqa: E402
from everyvoice_amd.train import ops  # noqa: E402

dev = torch.device("cuda:0")
x = torch.randn(1024, device=dev)
n = 50000
t0 = time.perf_counter()
for _ in range(n):
    _lib.current_stream_ptr(dev)
t1 = time.perf_counter()
for _ in range(n):
    torch.cuda.current_stream(dev).cuda_stream
t2 = time.perf_counter()
for _ in range(n):
    torch.empty_like(x)
t3 = time.perf_counter()
for _ in range(n):
    ops.elementwise(ops.EW_SCALE, x, p0=1.0, out=x)
torch.cuda.synchronize()
t4 = time.perf_counter()
print(f"stream ptr {1e6*(t1-t0)/n:.2f} us (torch.cuda.current_stream: {1e6*(t2-t1)/n:.2f} us), empty_like {1e6*(t3-t2)/n:.2f} us, "
      f"elementwise launch {1e6*(t4-t3)/n:.2f} us")
