#!/usr/bin/env python3
"""Two HiFiGAN trainers in lockstep on the bench batch: parameters bitwise equal after every step, or the tensors that differ.
usage: [GRAPH=0] [OPERANDS=bf16|f32] python tools/lockstep_gan.py [steps = 150]"""
import os, sys, collections
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
from everyvoice_amd.spectral import MelSpectrogram
from everyvoice_amd.train.hifigan import HiFiGANTrainer
dev = torch.device("cuda:0")
B, S = 16, 8192
g = torch.Generator().manual_seed(1234)
y = (0.3 * torch.tanh(torch.randn(B, 1, S, generator=g))).to(dev)
mel = MelSpectrogram()(y.squeeze(1), log=True)[:, :, : S // 256].contiguous()
graph = os.environ.get("GRAPH", "1") == "1"
mk = lambda: HiFiGANTrainer(device=dev, seed=3, precision=os.environ.get("OPERANDS", "bf16"), use_graph=graph)
a, b = mk(), mk()
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 150
ev = 0
for step in range(steps):
    la = a.training_step(mel, y, sync=False)
    lb = b.training_step(mel, y, sync=False)
    torch.cuda.synchronize()
    same = torch.equal(a.g_params.flat, b.g_params.flat) and torch.equal(a.d_params.flat, b.d_params.flat)
    if not same or not torch.equal(la, lb):
        sa, sb = a.state_dict(), b.state_dict()
        bad = [k for k in sa if torch.is_tensor(sa[k]) and sa[k].shape == sb[k].shape and not torch.equal(sa[k], sb[k])]
        print("step", step, "losses equal", torch.equal(la, lb), "differing tensors", len(bad), bad[:8], flush=True)
        ev += 1
        b.load_state_dict(a.state_dict()) if hasattr(b, "load_state_dict") else None
        for pa, pb in ((a.g_params, b.g_params), (a.d_params, b.d_params)):
            pb.flat.copy_(pa.flat); pb.m.copy_(pa.m); pb.v.copy_(pa.v)
        if ev > 6:
            break
print("graph", graph, "steps", step + 1, "events", ev)
