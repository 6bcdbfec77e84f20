import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
sys.path.insert(0, str(Path(__file__).resolve().parents[2] / "tests"))
import torch
import test_gpu_disc_chain as T

for which, B, TT in (("msd2", 3, 2400), ("msd2", 2, 8192), ("msd0", 2, 8192), ("mpd4", 2, 8192)):
    tr = T._trainer()
    d = tr.mpd[int(which[3])] if which.startswith("mpd") else tr.msd[int(which[3])]
    g = torch.Generator().manual_seed(3)
    y = (0.5 * torch.tanh(torch.randn(1, 2 * B, TT, generator=g))).cuda()
    if which.startswith("msd") and which != "msd0":
        from everyvoice_amd.train import ops
        for _ in range(int(which[3])):
            y = ops.avgpool4s2(y)
    st = T._sn_state(d)
    wl, want = T._run_d_step(tr, d, y, 11, chain=False)
    T._restore(st)
    gl, got = T._run_d_step(tr, d, y, 11, chain=True)
    print(which, B, TT, "logits max err / scale", float((gl - wl).abs().max() / wl.abs().max()))
    for n, w in want.items():
        if float(w.norm()) == 0:
            continue
        print("   %-45s cos %.6f  ratio %.5f  numel %d" % (n, T._cos(got[n], w), float(got[n].norm() / w.norm()), w.numel()))
