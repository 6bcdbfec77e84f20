import sys, time, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from everyvoice_amd.train.hifigan import HiFiGANTrainer
from everyvoice_amd.train import ops
dev = torch.device("cuda:0")
tr = HiFiGANTrainer(device=dev, precision="bf16")
mel = torch.randn(16, 80, 32, device=dev)
for _ in range(3): tr.generate(mel)
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    for _ in range(2): tr.generate(mel)
    torch.cuda.synchronize()
    with torch.cuda.graph(g, stream=s):
        y = tr.generate(mel)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20): g.replay()
torch.cuda.synchronize()
print("generator forward (graph replay, no MRF streams): %.3f ms" % ((time.perf_counter() - t0) / 20 * 1e3))
t0 = time.perf_counter()
for _ in range(20): tr.generate(mel)
torch.cuda.synchronize()
print("generator forward (eager): %.3f ms" % ((time.perf_counter() - t0) / 20 * 1e3))
