#!/usr/bin/env python3
"""Time the GAN training step (BASELINE config 4: bs 16, 8192-sample segments) on one GPU."""
import os
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch  # noqa: E402

from everyvoice_amd.spectral import MelSpectrogram  # noqa: E402
from everyvoice_amd.train.hifigan import HiFiGANTrainer  # noqa: E402

import os  # noqa: E402
from everyvoice_amd.train import ops  # noqa: E402
if os.environ.get("EVMI_CONV_BACKEND"):  # e.g. "mfma,mfma" / "gemm,gemm" / "mfma,gemm"
    parts = os.environ["EVMI_CONV_BACKEND"].split(",")
    ops.CONV_BACKEND.update(fwd=parts[0], dgrad=parts[1], wgrad=parts[2] if len(parts) > 2 else "mfma")
from everyvoice_amd.train import ops as _ops  # noqa: E402
_ops.CONV_BACKEND["operands"] = os.environ.get("OPERANDS", "f32")
dev = torch.device("cuda:0")
B, S = int(os.environ.get("EVMI_TRAIN_B", "16")), 8192
g = torch.Generator().manual_seed(1234)
y = (0.3 * torch.tanh(torch.randn(B, 1, S, generator=g))).to(dev)
mel = MelSpectrogram()(y.squeeze(1), log=True)[:, :, : S // 256].contiguous()
pg = None
if os.environ.get("DP1") == "1":  # the data-parallel code path on a one-rank RCCL group: what its captured stretches cost on one GPU
    import socket
    import torch.distributed as dist
    with socket.socket() as s_:
        s_.bind(("127.0.0.1", 0))
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(s_.getsockname()[1]))
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    pg = True
tr = HiFiGANTrainer(device=dev, precision=os.environ.get("OPERANDS", "f32"), use_graph=os.environ.get("GRAPH", "0") == "1", process_group=pg,
                    reconstruction_loss=os.environ.get("RECON", "mel"),
                    parallel_streams=os.environ.get("STREAMS", "1") == "1", side_wgrad=os.environ.get("SIDE_WGRAD", "0") == "1")
for i in range(4):
    out = tr.training_step(mel, y)
print("graph:", [len(e["graphs"]) for e in tr._graphs.values()], tr._graph_failed)
torch.cuda.synchronize()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 5
t0 = time.perf_counter()
for i in range(n):
    out = tr.training_step(mel, y)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / n
print(f"step {dt*1e3:.1f} ms -> {1/dt:.2f} steps/s; losses {out}")
if tr.phase_times:
    print("phases (ms):", tr.phase_times)
    for sec in tr.branch_times or []:
        if len(sec) > 3:
            print("  branches (ms):", sec)
print(f"G params {tr.g_params.numel():,}  D params {tr.d_params.numel():,}")
if pg:
    import torch.distributed as dist
    dist.destroy_process_group()
