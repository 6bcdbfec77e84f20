import os, sys, hashlib, collections
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[2])); sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
from fs2_train_bench import training_batch
from everyvoice_amd.train.fs2 import FastSpeech2Trainer
from everyvoice_amd.fs2 import FastSpeech2ModelConfig
from everyvoice_amd.train import ops
dev = torch.device("cuda:0")
def run(learn, ready, prec, steps=4, sync=False):
    batch, T_i = training_batch(32, learn_alignment=learn, device=dev)
    tr = FastSpeech2Trainer(FastSpeech2ModelConfig(learn_alignment=learn), device=dev, precision=prec, use_graph=False, side_wgrad=False)
    tr.batch_ready = ready
    for _ in range(steps):
        l = tr.training_step(batch)
        if sync:
            torch.cuda.synchronize()
    torch.cuda.synchronize()
    return hashlib.md5(tr.params.flat.cpu().numpy().tobytes()).hexdigest()[:8]
N = int(sys.argv[1]) if len(sys.argv) > 1 else 12
for name, kw in (("no-learn, ready", dict(learn=False, ready=True, prec="bf16")),
                 ("no-learn, not ready", dict(learn=False, ready=False, prec="bf16")),
                 ("no-learn, ready, sync", dict(learn=False, ready=True, prec="bf16", sync=True)),
                 ("no-learn, ready, f32", dict(learn=False, ready=True, prec="f32")),
                 ("no-learn, ready, 1 step", dict(learn=False, ready=True, prec="bf16", steps=1)),
                 ("no-learn, ready, 2 steps", dict(learn=False, ready=True, prec="bf16", steps=2))):
    c = collections.Counter(run(**kw) for _ in range(N))
    print(f"{name:28s}", dict(c), flush=True)
