"""The N > 1 path of bench.py on CPU: two ranks over gloo run the same barrier / max-over-ranks
timing and whole-job aggregation that the GPU replicas use (inference shards nothing between
ranks: "replicas only")."""

import json
import os
import socket
import sys
import time
from pathlib import Path

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import bench  # noqa: E402


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    calls = {"n": 0}

    def step():
        calls["n"] += 1
        time.sleep(0.01 * (rank + 1))  # rank 1 is the slow one: the max over ranks must win

    def max_reduce(x):
        t = torch.tensor([x], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    elapsed = bench.timed_region(step, steps=5, warmup=2, sync_fn=lambda: None, barrier_fn=dist.barrier, max_reduce_fn=max_reduce)
    samples = 1000
    value = world * samples * 5 / elapsed
    Path(out_dir, f"rank{rank}.json").write_text(json.dumps({"elapsed": elapsed, "calls": calls["n"], "value": value}))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_timing_and_aggregation(tmp_path):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    r = [json.loads(Path(tmp_path, f"rank{i}.json").read_text()) for i in range(world)]
    assert r[0]["calls"] == r[1]["calls"] == 7  # W warm-up + exactly K timed steps
    assert r[0]["elapsed"] == r[1]["elapsed"]  # every rank reports the max over ranks
    assert r[0]["elapsed"] >= 5 * 0.02 * 0.9  # the slow rank (20 ms / step) sets the time
    assert abs(r[0]["value"] - 2 * 1000 * 5 / r[0]["elapsed"]) < 1e-6  # whole-job aggregate, not per-GPU


def _allreduce_worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from everyvoice_amd.train.hifigan import allreduce_mean_

    flat = torch.arange(10, dtype=torch.float32) * (rank + 1)  # the optimiser's flat gradient buffer of this rank
    allreduce_mean_(flat, None, lambda t, s: t.mul_(s))        # CPU stand-in for the device scaling kernel
    torch.save(flat, Path(out_dir, f"g{rank}.pt"))
    dist.destroy_process_group()


def test_gradient_allreduce_is_a_mean_over_ranks(tmp_path):
    """The data-parallel exchange of the training step (one all-reduce per optimiser's flat gradient buffer)."""
    world = 2
    mp.spawn(_allreduce_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    g0, g1 = (torch.load(Path(tmp_path, f"g{i}.pt")) for i in range(world))
    want = torch.arange(10, dtype=torch.float32) * 1.5
    assert torch.equal(g0, g1) and torch.allclose(g0, want)


def _bucket_worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from everyvoice_amd.train.hifigan import BucketReducer

    flat = torch.arange(23, dtype=torch.float32) * (rank + 1)
    red = BucketReducer(flat, None, lambda t, s: t.mul_(s))
    hi = flat.numel()
    for lo in (17, 9, 4):  # backward finishes the buffer suffix-first, bucket by bucket
        red.launch(lo, hi)
        hi = lo
    red.launch(0, hi)      # what is left in front of the first bucket
    red.launch(0, 0)       # empty ranges are ignored
    red.finish()
    torch.save(flat, Path(out_dir, f"b{rank}.pt"))
    dist.destroy_process_group()


def test_bucketed_overlapped_allreduce_is_a_mean_over_ranks(tmp_path):
    """The exchange the training step uses on N > 1 GPUs: asynchronous all-reduces of contiguous buckets launched during
    backward, one wait + 1/world scaling at the end (gloo stands in for RCCL; the side stream is GPU-only)."""
    world = 2
    mp.spawn(_bucket_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    b0, b1 = (torch.load(Path(tmp_path, f"b{i}.pt")) for i in range(world))
    assert torch.equal(b0, b1) and torch.allclose(b0, torch.arange(23, dtype=torch.float32) * 1.5)


def test_dist_env_and_roofline_aggregation(monkeypatch):
    monkeypatch.setenv("RANK", "3")
    monkeypatch.setenv("LOCAL_RANK", "1")
    monkeypatch.setenv("WORLD_SIZE", "4")
    assert bench.dist_env() == (3, 1, 4)
    recs = [
        dict(kernel="a", layer="l0", ms=1.0, flops=2e9, bytes=1e6),
        dict(kernel="a", layer="l1", ms=3.0, flops=2e9, bytes=1e6),
        dict(kernel="b", layer="l2", ms=1.0, flops=1e9, bytes=4e6),
    ]
    roof = bench.roofline_from_records([recs, recs])
    assert roof["kernel"] == "a" and roof["launches_per_forward"] == 2
    assert abs(roof["avg_launch_ms"] - 2.0) < 1e-9
    assert abs(roof["achieved"] - 2e9 / 2e-3 / 1e12) < 1e-2  # algorithmic FLOP per launch / mean duration
    assert roof["peak"] == 2500.0 and roof["bound"] == "mfma"
    assert abs(roof["frac"] - roof["achieved"] / 2500.0) < 1e-3


def test_bare_multi_gpu_command_launches_its_own_ranks(monkeypatch, capfd):
    """`python bench.py --gpus 2` with no launcher in the environment: the parent starts two ranks through
    torch.distributed.run, relays rank 0's JSON line and returns the children's exit code (--selftest-cpu replaces the GPU
    legs by a sleep and RCCL by gloo; the launcher path is the one the GPU run takes)."""
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        monkeypatch.delenv(k, raising=False)
    rc = bench.main(["--gpus", "2", "--steps", "3", "--warmup", "1", "--selftest-cpu"])
    out = capfd.readouterr().out
    assert rc == 0, out
    line = json.loads([ln for ln in out.splitlines() if ln.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["ranks_seen"] == 2 and line["steps"] == 3
    assert line["ms_per_step"] >= 4.0 * 0.9  # the slower rank (2 x 2 ms) sets the time


def test_rank_count_mismatch_is_a_hard_error(monkeypatch, capfd):
    monkeypatch.setenv("WORLD_SIZE", "4")
    monkeypatch.setenv("RANK", "0")
    monkeypatch.setenv("LOCAL_RANK", "0")
    assert bench.main(["--gpus", "8", "--selftest-cpu"]) == 2
    assert "WORLD_SIZE=4" in capfd.readouterr().err
