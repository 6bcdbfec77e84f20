"""Per-convolution drift of the bf16-operand generator forward against the fp32 one (same trainer, same inputs)."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
from everyvoice_amd.spectral import MelSpectrogram
from everyvoice_amd.train import ops, autograd as ag
from everyvoice_amd.train.hifigan import HiFiGANTrainer, _to_cbt

dev = torch.device("cuda:0")
B, S = 2, 2048
g = torch.Generator().manual_seed(11)
y = (0.3 * torch.tanh(torch.randn(B, 1, S, generator=g))).to(dev)
mel = MelSpectrogram()(y.squeeze(1), log=True)[:, :, : S // 256].contiguous()
tr = HiFiGANTrainer(device=dev)
tr._materialize(tr.generator.layers())
orig = ops.conv1d_mfma
rec = {}
def hook(mode):
    def f(x, w, *a, **kw):
        out = orig(x, w, *a, **kw)
        rec.setdefault(mode, []).append((tuple(x.shape), tuple(w.shape), a[1:5] if len(a) > 1 else kw, x.clone(), out.clone()))
        return out
    return f
for mode in ("f32", "bf16"):
    ops.CONV_BACKEND["operands"] = mode
    ops.conv1d_mfma = hook(mode)
    tape = ag.Tape()
    yh = tr.generator.forward(tape, ag.Var(_to_cbt(mel), needs_grad=False), training=True) if False else tr.generator.forward(tape, ag.Var(_to_cbt(mel), needs_grad=False))
    rec[mode + "_y"] = yh.data.clone()
ops.conv1d_mfma = orig
for i, (a, b) in enumerate(zip(rec["f32"], rec["bf16"])):
    dx = float((a[3] - b[3]).norm() / (a[3].norm() + 1e-30))
    dy = float((a[4] - b[4]).norm() / (a[4].norm() + 1e-30))
    # the bf16 layer on the fp32 layer's input: isolates this layer's own error
    ops.CONV_BACKEND["operands"] = "bf16"
    print(f"{i:3d} x{a[0]} w{a[1]} {a[2]}  in-drift {dx:.2e} out-drift {dy:.2e}")
print("y_hat drift", float((rec["f32_y"] - rec["bf16_y"]).norm() / rec["f32_y"].norm()), "norm", float(rec["f32_y"].norm()))
