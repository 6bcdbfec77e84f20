"""bf16-operand convolution vs torch on rounded operands: FastSpeech2 inference shapes (few columns, activations)."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch, torch.nn.functional as F
from everyvoice_amd.train import ops
dev = torch.device("cuda:0")
ops.CONV_BACKEND["operands"] = "bf16"
bf = lambda t: t.to(torch.bfloat16).to(torch.float32)
cbt = lambda t: t.permute(1, 0, 2).contiguous()
g = torch.Generator().manual_seed(0)
acts = {0: lambda v: v, 2: F.silu, 3: F.relu, 4: torch.tanh}
for B, T in ((2, 40), (2, 230), (1, 5), (32, 141)):
    for cin, cout, k, act in ((256, 1024, 1, 2), (1024, 256, 1, 0), (256, 512, 1, 0), (256, 256, 1, 3), (256, 80, 1, 0), (80, 512, 5, 4), (512, 512, 5, 4),
                              (512, 80, 5, 0), (256, 1, 1, 0), (256, 768, 1, 0)):
        x = torch.randn(B, cin, T, generator=g); w = torch.randn(cout, cin, k, generator=g) * 0.1; b = torch.randn(cout, generator=g)
        rounded = cout > 4
        want = acts[act](F.conv1d(bf(x) if rounded else x, bf(w) if rounded else w, b, 1, (k - 1) // 2))
        got = ops.conv1d_mfma(cbt(x).to(dev), w.to(dev), b.to(dev), 1, (k - 1) // 2, 1, 1, act=act)
        e = float((cbt(got.cpu()) - want).abs().max() / want.abs().max())
        print(f"B {B} T {T} cin {cin} cout {cout} k {k} act {act}: {e:.1e}" + ("   <------" if not e < 1e-4 else ""))
