import sys, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from everyvoice_amd.train.hifigan import HiFiGANTrainer
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(8)
B, S = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (2, 2048)
prec = sys.argv[3] if len(sys.argv) > 3 else "f32"
y = (0.3 * torch.tanh(torch.randn(B, 1, S, generator=g))).to(dev)
mel = torch.randn(B, 80, S // 256, generator=g).to(dev)
runs = {}
for name, par in (("par1", True), ("par2", True), ("seq1", False), ("seq2", False)):
    tr = HiFiGANTrainer(device=dev, seed=5, parallel_streams=par, precision=prec)
    tr.keep_grads = True
    out = tr.training_step(mel, y)
    torch.cuda.synchronize()
    runs[name] = (out, {k: v.clone() for k, v in tr.last_grads["d"].items()}, {k: v.clone() for k, v in tr.last_grads["g"].items()}, tr.last_grads["y_hat"].clone())
def cmp(a, b):
    ra, rb = runs[a], runs[b]
    print(a, b, "losses", ra[0] == rb[0], {k: (ra[0][k], rb[0][k]) for k in ra[0] if ra[0][k] != rb[0][k]})
    print("  y_hat equal", torch.equal(ra[3], rb[3]))
    for side in (1, 2):
        bad = [(k, float((ra[side][k] - rb[side][k]).abs().max()), float(ra[side][k].abs().max())) for k in ra[side] if not torch.equal(ra[side][k], rb[side][k])]
        print("  side", side, "mismatching tensors", len(bad), bad[:10])
cmp("par1", "par2"); cmp("seq1", "seq2"); cmp("par1", "seq1")
