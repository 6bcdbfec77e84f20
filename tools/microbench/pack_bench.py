#!/usr/bin/env python3
"""The pack pass of the packed bf16 convolutions alone (stage 1 of the staged input gradient: fp32 [C, B, T] -> 16-byte units of 8
channels), plain and with the SiLU / dropout fusion, at the FastSpeech2 decoder shapes: us per call and GB/s of (bytes read + written).
usage: python tools/microbench/pack_bench.py [iters]"""
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
import torch  # noqa: E402

from everyvoice_amd import _lib  # noqa: E402

dev = torch.device("cuda:0")
lib = _lib.load()
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 50
st = _lib.current_stream_ptr(dev)


def timed(fn):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


B, T = 32, 814
for C_dy, C_x in ((1024, 256), (256, 1024), (512, 512)):
    dy = torch.randn(C_dy, B, T, device=dev)
    pre = torch.randn(C_dy, B, T, device=dev)
    w = torch.randn(C_dy, C_x, 1, device=dev) * 0.05
    dx = torch.empty(C_x, B, T, device=dev)
    n = lib.evmi_conv1d_dgrad_cbt_bf16pk_ws_elems(B, C_x, T, C_dy, T, 1, 1, 0, 1, 1)
    ws = torch.empty(n, device=dev)
    geo = (B, C_x, T, C_dy, T, 1, 1, 0, 1, 1)
    t_plain = timed(lambda: _lib.check(lib.evmi_conv1d_dgrad_cbt_bf16pk_staged(1, dy.data_ptr(), w.data_ptr(), dx.data_ptr(), ws.data_ptr(), n, *geo, st), "pack"))
    t_fused = timed(lambda: _lib.check(lib.evmi_conv1d_dgrad_cbt_bf16pk_staged_silu_dropout(1, dy.data_ptr(), pre.data_ptr(), 0.1, 7, 0, w.data_ptr(), dx.data_ptr(),
                                                                                           ws.data_ptr(), n, *geo, st), "pack fused"))
    by = dy.numel() * 6
    print(f"pack [{C_dy} x {B} x {T}]: plain {t_plain:6.1f} us ({by / t_plain * 1e-3:6.0f} GB/s)   dropout(x) * silu'(aux) {t_fused:6.1f} us ({(by + dy.numel() * 4) / t_fused * 1e-3:6.0f} GB/s)")
