#!/usr/bin/env python3
"""The embedding-table backward kernels alone at the FastSpeech2 bench shape (B = 32, L = 141, D = 256): us per call.
usage: python tools/microbench/table_bwd_bench.py [iters]"""
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


g = torch.Generator().manual_seed(1)
for B, L, D in ((32, 141, 256), (32, 814, 256)):
    dx = torch.randn(D, B, L, generator=g).to(dev)
    ids = torch.randint(1, 80, (B, L), generator=g).to(torch.int32).to(dev)
    lens = torch.full((B,), L, dtype=torch.int32, device=dev)
    table = torch.zeros(80, D, device=dev)
    t_text = timed(lambda: _lib.check(lib.evmi_fs2_embed_bwd_f32(dx.data_ptr(), ids.data_ptr(), lens.data_ptr(), table.data_ptr(), 80, B, L, D, 0, st), "embed_bwd"))
    vals = torch.randn(B, L, generator=g).to(dev)
    bins = torch.linspace(-3, 3, 255).to(dev)
    tab2 = torch.zeros(256, D, device=dev)
    idx = torch.empty(B, L, dtype=torch.int32, device=dev)
    t_bucket = timed(lambda: _lib.check(lib.evmi_fs2_bucket_embed_bwd_f32(dx.data_ptr(), vals.data_ptr(), bins.data_ptr(), tab2.data_ptr(), idx.data_ptr(), 256, B, L, D, 1.0, st),
                                        "bucket_embed_bwd"))
    print(f"B {B} L {L} D {D}: text table (80 rows) {t_text:6.1f} us   bucket table (256 rows, incl. bucketise) {t_bucket:6.1f} us")
