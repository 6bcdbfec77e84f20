import os, sys, hashlib, collections
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
    sd = tr.state_dict()
    return hashlib.md5(tr.params.flat.cpu().numpy().tobytes()).hexdigest()[:8], hashlib.md5(sd["text_input_layer.weight"].cpu().numpy().tobytes()).hexdigest()[:6]
for side, group in ((False, 1), (True, 1), (True, 8)):
    c = collections.Counter(run(side, group, False) for _ in range(7))
    print("side", side, "group", group, dict(c), flush=True)
