import os, socket, sys, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, torch.distributed as dist
from everyvoice_amd.train.hifigan import HiFiGANTrainer
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(8)
B, S = 2, 2048
y = (0.3 * torch.tanh(torch.randn(B, 1, S, generator=g))).to(dev)
mel = torch.randn(B, 80, S // 256, generator=g).to(dev)
with socket.socket() as s:
    s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
dp = HiFiGANTrainer(device=dev, seed=5, process_group=True, use_graph=True)
orig = dp._capture
def cap(*a, **k):
    try:
        return orig(*a, **k)
    except Exception:
        traceback.print_exc()
        raise
dp._capture = cap
import everyvoice_amd.train.hifigan as H
orig_group = H.HiFiGANTrainer._phase_d_group
def dbg_group(self, ctx, idxs, reducer):
    print("  d_group", list(idxs), "capturing", torch.cuda.is_current_stream_capturing(), flush=True)
    return orig_group(self, ctx, idxs, reducer)
H.HiFiGANTrainer._phase_d_group = dbg_group
orig_run = H.Branches.run_indexed
def dbg_run(self, items):
    if torch.cuda.is_current_stream_capturing():
        print("    branches.run_indexed", [j for j, _ in items], flush=True)
    return orig_run(self, items)
H.Branches.run_indexed = dbg_run
for i in range(5):
    try:
        out = dp.training_step(mel, y, sync=False)
        print("step", i, "graph_failed:", dp._graph_failed, "graphs:", [len(e["graphs"]) for e in dp._graphs.values()], flush=True)
    except Exception as e:
        print("step", i, "EXC", type(e).__name__, str(e)[:300], flush=True)
        break
