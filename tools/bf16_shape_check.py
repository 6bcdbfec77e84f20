"""bf16-operand convolution (forward and input gradient) vs torch on rounded operands, on the generator's exact shapes."""
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
B = 2
shapes = [(80, 512, 7, 1, 3, 1, 8)]
for C, T in ((256, 64), (128, 512), (64, 1024), (32, 2048)):
    for k in (3, 7, 11):
        for d in (1, 3, 5):
            shapes.append((C, C, k, 1, d * (k - 1) // 2, d, T))
for cin, cout, k, s, p, d, T in shapes:
    x = torch.randn(B, cin, T, generator=g, requires_grad=True); w = torch.randn(cout, cin, k, generator=g) * 0.1; b = torch.randn(cout, generator=g)
    want = F.conv1d(bf(x), bf(w), b, s, p, d)
    got = ops.conv1d_fwd(cbt(x.detach()).to(dev), w.to(dev), b.to(dev), s, p, d, 1)
    e1 = float((cbt(got.cpu()) - want).abs().max() / want.abs().max())
    got2 = ops.conv1d_fwd(cbt(x.detach()).to(dev), w.to(dev), b.to(dev), s, p, d, 1, lrelu_slope=0.1)
    e3 = float((cbt(got2.cpu()) - F.leaky_relu(want, 0.1)).abs().max() / want.abs().max())
    dy = torch.randn(want.shape, generator=g)
    gi = torch.nn.grad.conv1d_input(x.shape, bf(w), bf(dy), s, p, d, 1)
    dx = ops.conv1d_bwd_data_mfma(cbt(dy).to(dev), w.to(dev), T, s, p, d, 1)
    e2 = float((cbt(dx.cpu()) - gi).abs().max() / gi.abs().max())
    flag = "  <-----" if max(e1, e2, e3) > 1e-4 else ""
    print(f"cin {cin} cout {cout} k {k} d {d} T {T}: fwd {e1:.1e} fwd+lrelu {e3:.1e} dgrad {e2:.1e}{flag}")
