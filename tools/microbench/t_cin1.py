import sys
sys.path.insert(0, "/root/repo")
import torch, torch.nn.functional as F
from everyvoice_amd.train import ops
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
for (B, T, cin, cout, k, s, p) in [(6, 683, 1, 32, 5, 3, 2), (4, 1024, 1, 32, 5, 3, 2), (12, 683, 1, 32, 5, 3, 2), (6, 228, 32, 128, 5, 3, 2), (6, 76, 128, 512, 5, 3, 2)]:
    x = torch.randn(B, cin, T, generator=g); w = torch.randn(cout, cin, k, generator=g) * 0.2; b = torch.randn(cout, generator=g)
    want = F.conv1d(x, w, b, s, p)
    got = ops.conv1d_fwd(x.permute(1, 0, 2).contiguous().to(dev), w.to(dev), b.to(dev), s, p, 1, 1).cpu().permute(1, 0, 2)
    err = (got - want).abs()
    print((B, T, cin, cout), "max err", float(err.max()), "at", [int(i) for i in torch.nonzero(err == err.max())[0]], want.shape)
    # dgrad
    dy = torch.randn(want.shape, generator=g)
    xr = x.clone().requires_grad_(); F.conv1d(xr, w, b, s, p).backward(dy)
    dx = ops.conv1d_bwd_data_mfma(dy.permute(1, 0, 2).contiguous().to(dev), w.to(dev), T, s, p, 1, 1).cpu().permute(1, 0, 2)
    e2 = (dx - xr.grad).abs()
    print("   dgrad max err", float(e2.max()), "at", [int(i) for i in torch.nonzero(e2 == e2.max())[0]])
