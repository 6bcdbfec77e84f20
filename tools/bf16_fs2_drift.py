"""Per-call drift of the bf16-operand FastSpeech2 forward against the fp32 one (same model, same inputs)."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
from everyvoice_amd.fs2 import FastSpeech2
from everyvoice_amd.train import ops
dev = torch.device("cuda:0")
model = FastSpeech2(device=dev).init_random(1234)
g = torch.Generator().manual_seed(11)
B, L = 2, 40
lens = torch.tensor([40, 23])
ids = torch.randint(1, 80, (B, L), generator=g).masked_fill(torch.arange(L)[None] >= lens[:, None], 0)
durs = torch.randint(2, 9, (B, L), generator=g)
orig = ops.conv1d_mfma
rec = {}
def hook(mode):
    def f(x, w, *a, **kw):
        out = orig(x, w, *a, **kw)
        rec.setdefault(mode, []).append((tuple(x.shape), tuple(w.shape), x.is_contiguous(), w.is_contiguous(), kw.get("act", 0), x.clone(), out.clone(), w.data_ptr() % 16, x.data_ptr() % 16))
        return out
    return f
outs = {}
for mode in ("f32", "bf16"):
    model.precision = mode
    ops.conv1d_mfma = hook(mode)
    outs[mode] = model(ids.to(dev), lens.to(dev), durations=durs.to(dev))
ops.conv1d_mfma = orig
for i, (a, b) in enumerate(zip(rec["f32"], rec["bf16"])):
    dx = float((a[5] - b[5]).norm() / (a[5].norm() + 1e-30)); dy = float((a[6] - b[6]).norm() / (a[6].norm() + 1e-30))
    flag = "   <-----" if dy > 10 * max(dx, 3e-3) else ""
    print(f"{i:3d} x{a[0]} w{a[1]} contig {a[2]},{a[3]} act {a[4]} align {a[7]},{a[8]} in-drift {dx:.2e} out-drift {dy:.2e}{flag}")
print("mel drift", float((outs["f32"][1] - outs["bf16"][1]).norm() / outs["f32"][1].norm()))
