"""Data parallelism executed for real: TWO ranks sharing GPU 0 (one process each, gloo moving the gradient buckets between
them -- RCCL refuses two ranks on one device, and the boxes these tests run on have one), against ONE rank on the concatenated
batch.  HiFiGAN has no BatchNorm, every loss is a mean over the batch and the shards are equal: the mean of the two ranks'
gradients IS the gradient of the big batch (summation order aside), so after the exchange every rank must hold it -- in eager
mode (buckets launched inside backward) and in graph mode (stretches cut at bucket boundaries).  SURVEY.md 8e; the reference:
Lightning DDP, everyvoice/base_cli/interfaces.py:90-97 -> helpers.py:252-270."""

import os
import socket
import sys
from pathlib import Path

import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = Path(__file__).resolve().parents[1]


def _batch(B, S, seed=8):
    g = torch.Generator().manual_seed(seed)
    y = 0.3 * torch.tanh(torch.randn(B, 1, S, generator=g))
    mel = torch.randn(B, 80, S // 256, generator=g)
    return mel, y


def _rank_main(rank, world, port, use_graph, steps, out_dir, B, S):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    sys.path.insert(0, str(ROOT))
    import torch.distributed as dist

    from everyvoice_amd.train.hifigan import HiFiGANTrainer

    dist.init_process_group("gloo", rank=rank, world_size=world)
    dev = torch.device("cuda:0")
    try:
        tr = HiFiGANTrainer(device=dev, seed=5, process_group=True, use_graph=use_graph)
        mel, y = _batch(B * world, S)
        mel, y = mel[rank * B:(rank + 1) * B].to(dev), y[rank * B:(rank + 1) * B].to(dev)
        losses = [tr.training_step(mel, y) for _ in range(steps)]
        torch.cuda.synchronize(dev)
        torch.save({"d_grad": tr.d_params.grad.cpu(), "g_grad": tr.g_params.grad.cpu(), "d": tr.d_params.flat.cpu(), "g": tr.g_params.flat.cpu(),
                    "losses": losses, "graph_failed": tr._graph_failed, "graphs": [len(e["graphs"]) for e in tr._graphs.values()]},
                   Path(out_dir) / f"rank{rank}.pt")
    finally:
        dist.destroy_process_group()


def _run_two_ranks(tmp_path, use_graph, steps, B, S):
    import torch.multiprocessing as mp

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    procs = [ctx.Process(target=_rank_main, args=(r, 2, port, use_graph, steps, str(tmp_path), B, S)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(600)
        assert p.exitcode == 0, f"rank exited with {p.exitcode}"
    return [torch.load(tmp_path / f"rank{r}.pt", weights_only=False) for r in range(2)]


@pytest.mark.parametrize("use_graph", [False, True])
def test_two_ranks_on_one_gpu_equal_one_rank_on_the_concatenated_batch(cuda_device, tmp_path, use_graph):
    from everyvoice_amd.train.hifigan import HiFiGANTrainer

    B, S = 2, 2048
    steps = 4 if use_graph else 1  # graph mode: two eager steps, the capture, one pure replay
    r0, r1 = _run_two_ranks(tmp_path, use_graph, steps, B, S)
    assert r0["graph_failed"] is None and r1["graph_failed"] is None
    if use_graph:
        assert r0["graphs"] and r0["graphs"][0] >= 5  # stretches cut at bucket boundaries, exchanges between them
    # both ranks hold the same averaged gradients and the same parameters
    for k in ("d_grad", "g_grad", "d", "g"):
        assert torch.equal(r0[k], r1[k]), k
    # ... and they are the one-rank step's on the concatenated batch (first step: same initial parameters on every side)
    if not use_graph:
        one = HiFiGANTrainer(device=cuda_device, seed=5)
        mel, y = _batch(2 * B, S)
        one.training_step(mel.to(cuda_device), y.to(cuda_device))
        for k, got, want in (("d_grad", r0["d_grad"], one.d_params.grad.cpu()), ("g_grad", r0["g_grad"], one.g_params.grad.cpu())):
            scale = float(want.abs().max())
            err = float((got - want).abs().max())
            assert scale > 0 and err <= 2e-4 * scale, (k, err, scale)  # summation order (two half-batch sums vs one)
        # parameters: AdamW's first step is lr * g / (|g| + eps) -- elements with |g| ~ eps may land anywhere in +-lr; the rest agree
        for k, got, want in (("d", r0["d"], one.d_params.flat.cpu()), ("g", r0["g"], one.g_params.flat.cpu())):
            diff = (got - want).abs()
            assert float(diff.max()) <= 2.5 * one.opt["lr"] and float((diff > 1e-7).float().mean()) < 0.02, k
    else:
        # graph mode: four steps of two ranks vs four eager steps of two ranks would be the bitwise comparison; here: the losses of
        # every step agree with the eager two-rank run's to summation order (same shards, same arithmetic, exchanges moved)
        tmp2 = tmp_path / "eager"
        tmp2.mkdir()
        e0, _ = _run_two_ranks(tmp2, False, steps, B, S)
        for a, b in zip(r0["losses"], e0["losses"]):
            for k in a:
                assert a[k] == pytest.approx(b[k], rel=1e-6, abs=1e-7), k
        assert torch.equal(r0["d"], e0["d"]) and torch.equal(r0["g"], e0["g"])  # graph replays == eager steps, exchange included
