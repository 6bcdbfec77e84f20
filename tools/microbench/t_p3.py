import sys
sys.path.insert(0, "/root/repo")
import torch, torch.nn.functional as F
from everyvoice_amd.train import ops
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
def check(B, T, cin, cout, k, s, p):
    x = torch.randn(B, cin, T, generator=g); w = torch.randn(cout, cin, k, generator=g) * 0.2; b = torch.randn(cout, generator=g)
    xr = x.clone().requires_grad_(); wr = w.clone().requires_grad_(); br = b.clone().requires_grad_()
    y = F.conv1d(xr, wr, br, s, p); dy = torch.randn(y.shape, generator=g); y.backward(dy)
    xd, wd, bd, dyd = x.permute(1, 0, 2).contiguous().to(dev), w.to(dev), b.to(dev), dy.permute(1, 0, 2).contiguous().to(dev)
    yg = ops.conv1d_fwd(xd, wd, bd, s, p, 1, 1)
    dbuf = torch.zeros(cout, device=dev)
    dx, dw, db = ops.conv1d_bwd(xd, wd, dyd, s, p, 1, 1, db_out=dbuf)
    rel = lambda a, b_: float((a - b_).abs().max() / (b_.abs().max() + 1e-30))
    print((B, T, cin, cout, k, s), "y %.2e dx %.2e dw %.2e db %.2e" % (rel(yg.cpu().permute(1, 0, 2), y.detach()), rel(dx.cpu().permute(1, 0, 2), xr.grad), rel(dw.cpu(), wr.grad), rel(db.cpu(), br.grad)))
for B in (12,):
    check(B, 683, 1, 32, 5, 3, 2); check(B, 228, 32, 128, 5, 3, 2); check(B, 76, 128, 512, 5, 3, 2); check(B, 26, 512, 1024, 5, 3, 2)
    check(B, 9, 1024, 1024, 5, 1, 2); check(B, 9, 1024, 1, 3, 1, 1)
