#!/usr/bin/env python3
"""Packed bf16 convolution at the FastSpeech2 feed-forward shapes under a forced tile (EVMI_PK_TILE, one process per tile)."""
import os
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[2]
if len(sys.argv) > 1 and sys.argv[1] == "child":
    sys.path.insert(0, str(ROOT))
    import torch

    from everyvoice_amd.train import ops

    dev = torch.device("cuda:0")
    ops.CONV_BACKEND["operands"] = "bf16"
    for cin, cout, k in ((256, 1024, 1), (1024, 256, 1), (256, 256, 1), (256, 768, 1), (512, 512, 5)):
        x = torch.randn(cin, 32, 814, device=dev)
        w = torch.randn(cout, cin, k, device=dev) * 0.05
        b = torch.randn(cout, device=dev)
        for _ in range(3):
            ops.conv1d_fwd(x, w, b, 1, (k - 1) // 2, 1, 1)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            ops.conv1d_fwd(x, w, b, 1, (k - 1) // 2, 1, 1)
        e1.record()
        torch.cuda.synchronize()
        t = e0.elapsed_time(e1) / 20 * 1e3
        print(f"  {cin}->{cout} k{k}: {t:.1f} us (pack + conv), {2.0*cin*cout*k*32*814/t/1e6:.0f} TFLOP/s")
else:
    for tile in ("-1", "0", "1", "4", "6"):
        print("EVMI_PK_TILE", tile, flush=True)
        env = dict(os.environ, EVMI_PK_TILE=tile)
        subprocess.run([sys.executable, __file__, "child"], env=env, check=False)
