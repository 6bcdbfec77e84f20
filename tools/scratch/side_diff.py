import os, sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[2])); sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
from fs2_train_bench import training_batch
from everyvoice_amd.train.fs2 import FastSpeech2Trainer
from everyvoice_amd.fs2 import FastSpeech2ModelConfig
from everyvoice_amd.train import ops
dev = torch.device("cuda:0")
batch, T_i = training_batch(32, learn_alignment=True, device=dev)
def run(side, group, steps=1, graph=False):
    ops.SIDE_GROUP[0] = group
    from everyvoice_amd.train.fs2 import FastSpeech2TrainingConfig
    tr = FastSpeech2Trainer(FastSpeech2ModelConfig(learn_alignment=True), training=FastSpeech2TrainingConfig(gradient_clip_val=None), device=dev, precision="bf16", use_graph=graph, side_wgrad=side)
    tr.batch_ready = True
    for _ in range(steps):
        tr.training_step(batch)
    torch.cuda.synchronize()
    return {k: v.clone() for k, v in tr.state_dict().items()}
for graph, steps in ((False, 4), (False, 4)):
    ref = run(True, 1, steps, graph)
    for g in (3, 32):
        got = run(True, g, steps, graph)
        bad = [(k, float((got[k].float() - ref[k].float()).abs().max())) for k in ref if got[k].shape == ref[k].shape and not torch.equal(got[k], ref[k])]
        print("graph", graph, "steps", steps, "group", g, "differing tensors:", len(bad), "of", len(ref), [b[0] for b in bad][:60], flush=True)
