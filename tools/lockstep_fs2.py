#!/usr/bin/env python3
"""Two FastSpeech2 trainers in lockstep on the bench batch: after every step the parameters must be bitwise equal; where they are
not, the tensors that differ name the layer whose gradient came out differently (no clipping, so nothing spreads it), and trainer B is
put back on A.  How the stale-tile race of the bf16 attention backward was found.
usage: [LEARN=1] [SIDE=1] [GRAPH=1] [OPERANDS=bf16|f32] python tools/lockstep_fs2.py [steps = 200]"""
import os, sys, collections
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1])); sys.path.insert(0, str(Path(__file__).resolve().parent))
import torch
from fs2_train_bench import training_batch
from everyvoice_amd.train.fs2 import FastSpeech2Trainer, FastSpeech2TrainingConfig
from everyvoice_amd.fs2 import FastSpeech2ModelConfig
dev = torch.device("cuda:0")
learn = os.environ.get("LEARN", "0") == "1"
prec = os.environ.get("OPERANDS", "bf16")
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
batch, T_i = training_batch(32, learn_alignment=learn, device=dev)
mk = lambda: FastSpeech2Trainer(FastSpeech2ModelConfig(learn_alignment=learn), training=FastSpeech2TrainingConfig(gradient_clip_val=None), device=dev,
                                precision=prec, use_graph=os.environ.get("GRAPH", "0") == "1", side_wgrad=os.environ.get("SIDE", "0") == "1")
a, b = mk(), mk()
events = collections.Counter()
for step in range(steps):
    a.training_step(batch)
    b.training_step(batch)
    torch.cuda.synchronize()
    if not torch.equal(a.params.flat, b.params.flat):
        sa, sb = a.state_dict(), b.state_dict()
        bad = [k for k in sa if sa[k].shape == sb[k].shape and not torch.equal(sa[k], sb[k])]
        print("step", step, "differ:", bad[:10], len(bad), flush=True)
        for k in bad:
            events[k] += 1
        b.params.flat.copy_(a.params.flat); b.params.m.copy_(a.params.m); b.params.v.copy_(a.params.v)
        sb2 = a.state_dict()
        b.load_state_dict(sb2)
        b.params.m.copy_(a.params.m); b.params.v.copy_(a.params.v); b.params.step = a.params.step
        if hasattr(b.params, "step_dev"): b.params.step_dev.copy_(a.params.step_dev)
        b.global_step = a.global_step
print("events per tensor:", dict(events))
