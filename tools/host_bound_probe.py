#!/usr/bin/env python3
"""Is a replayed training step bound by the HOST side of hipGraphLaunch?  Per step: the time until training_step(sync=False) returns
(the runtime has walked the captured graph and written its packets) against the time until the device has finished it.  If the two
are close, the step is paced by the host thread, and launches -- not kernel time -- are what to remove."""
import os
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
sys.path.insert(0, str(Path(__file__).resolve().parent))
import torch  # noqa: E402

dev = torch.device("cuda:0")
which = sys.argv[1] if len(sys.argv) > 1 else "gan"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 40
if which == "gan":
    from everyvoice_amd.spectral import MelSpectrogram
    from everyvoice_amd.train.hifigan import HiFiGANTrainer

    g = torch.Generator().manual_seed(1234)
    y = (0.3 * torch.tanh(torch.randn(16, 1, 8192, generator=g))).to(dev)
    mel = MelSpectrogram()(y.squeeze(1), log=True)[:, :, :32].contiguous()
    tr = HiFiGANTrainer(device=dev, precision="bf16", use_graph=True)
    step = lambda: tr.training_step(mel, y, sync=False)  # noqa: E731
else:
    from fs2_train_bench import training_batch

    from everyvoice_amd.train.fs2 import FastSpeech2Trainer

    tr = FastSpeech2Trainer(device=dev, precision="bf16", use_graph=True)
    batch, _ = training_batch(32, device=dev)
    tr.batch_ready = True
    step = lambda: tr.training_step(batch)  # noqa: E731
for _ in range(6):
    step()
torch.cuda.synchronize()
host, total = [], []
for _ in range(n):
    t0 = time.perf_counter()
    step()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    host.append(t1 - t0)
    total.append(t2 - t0)
host.sort(); total.sort()
# back to back (no sync between steps): what a training loop sees
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(n):
    step()
torch.cuda.synchronize()
b2b = (time.perf_counter() - t0) / n
print(f"{which}: host enqueue median {host[n // 2] * 1e3:.2f} ms (min {host[0] * 1e3:.2f}, max {host[-1] * 1e3:.2f}); enqueue + device median {total[n // 2] * 1e3:.2f} ms; "
      f"back to back {b2b * 1e3:.2f} ms per step; cpus {len(os.sched_getaffinity(0))}")
