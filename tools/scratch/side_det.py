import os, sys, hashlib
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[2])); sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
from fs2_train_bench import training_batch
from everyvoice_amd.train.fs2 import FastSpeech2Trainer
from everyvoice_amd.fs2 import FastSpeech2ModelConfig
from everyvoice_amd.train import ops
dev = torch.device("cuda:0")
batch, T_i = training_batch(32, learn_alignment=True, device=dev)
def run(side, group, graph, steps=4):
    ops.SIDE_GROUP[0] = group
    tr = FastSpeech2Trainer(FastSpeech2ModelConfig(learn_alignment=True), device=dev, precision="bf16", use_graph=graph, side_wgrad=side)
    tr.batch_ready = True
    for _ in range(steps):
        l = tr.training_step(batch)
    torch.cuda.synchronize()
    return hashlib.md5(tr.params.flat.cpu().numpy().tobytes()).hexdigest()[:10], round(float(l["total"]), 6)
for graph in (True,):
    for side, group in ((False, 8), (True, 1), (True, 8), (True, 3), (True, 32), (True, 5)):
        print("graph", graph, "side", side, "group", group, run(side, group, graph), flush=True)
