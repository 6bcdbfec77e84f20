#!/usr/bin/env python3
"""The GAN step's two modes: time blocks of ten replayed steps for a while in ONE process and sample the GPU's clocks
beside them (rocm-smi, if the box lets an ordinary user read them): does a process change mode, and do the clocks say why?"""
import subprocess
import sys
import threading
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch  # noqa: E402

from everyvoice_amd.spectral import MelSpectrogram  # noqa: E402
from everyvoice_amd.train.hifigan import HiFiGANTrainer  # noqa: E402

dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(1234)
y = (0.3 * torch.tanh(torch.randn(16, 1, 8192, generator=g))).to(dev)
mel = MelSpectrogram()(y.squeeze(1), log=True)[:, :, :32].contiguous()
tr = HiFiGANTrainer(device=dev, precision="bf16", use_graph=True)
for _ in range(4):
    tr.training_step(mel, y)
torch.cuda.synchronize()
clocks, stop = [], [False]


def sample():
    while not stop[0]:
        try:
            out = subprocess.run(["rocm-smi", "--showclocks", "--showpower", "--csv"], capture_output=True, text=True, timeout=5).stdout
            rows = [ln for ln in out.splitlines() if ln and not ln.startswith("WARN")]
            clocks.append((time.perf_counter(), rows[-1] if rows else ""))
        except Exception as e:  # noqa: BLE001
            clocks.append((time.perf_counter(), f"rocm-smi: {e}"))
            return
        time.sleep(0.4)


th = threading.Thread(target=sample, daemon=True)
th.start()
blocks = int(sys.argv[1]) if len(sys.argv) > 1 else 40
t_start = time.perf_counter()
for b in range(blocks):
    t0 = time.perf_counter()
    for _ in range(10):
        tr.training_step(mel, y, sync=False)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 10
    near = [c for t, c in clocks if t >= t0 - 0.5]
    print(f"t={t0 - t_start:5.1f}s  {dt * 1e3:6.2f} ms/step  {near[-1][:160] if near else ''}", flush=True)
stop[0] = True
