"""Repeats the GAN step's graph == eager comparison and says WHERE two runs first differ.

Every job is a set of `world` rank processes sharing GPU 0 (gloo moves the gradient buckets, as in tests/test_gpu_ddp.py); inside
one job the ranks build a fresh trainer `trials` times and run `steps` steps each time, and after every step they store a checksum of
every named parameter and of its gradient (int32 view summed in int64 on the device: any flipped bit shows).  Jobs named in
`--jobs` run side by side (the test runs the eager and the graph job together: four processes on one GPU).  The parent then checks
that EVERY trial of EVERY job holds the same checksums -- the step is deterministic by design, so graph trials, eager trials and
repeats must all agree -- and prints the first (step, buffer, parameter) that does not.

    python tools/ddp_repeat.py --jobs graph,eager --world 2 --trials 10
    python tools/ddp_repeat.py --jobs graph,eager --world 1 --trials 10      # one process per job, no exchange
    python tools/ddp_repeat.py --fresh 12 --parallel 3 --trials 1            # the test's situation: every comparison in fresh processes
    python tools/ddp_repeat.py --jobs graphP,eagerP,eager --world 1 --trials 2   # NaN-poisoned torch.empty / workspaces vs plain
    python tools/ddp_repeat.py --jobs graphF,eagerF,eager --world 2 --trials 3   # schedule fuzzing: forked chains start behind random busy-waits
    EVMI_DISC_CHAIN=0 python tools/ddp_repeat.py ...                          # switches read by the trainer pass through
"""

from __future__ import annotations

import argparse
import json
import os
import socket
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]


def _batch(B, S, seed=8):
    import torch

    g = torch.Generator().manual_seed(seed)
    y = 0.3 * torch.tanh(torch.randn(B, 1, S, generator=g))
    mel = torch.randn(B, 80, S // 256, generator=g)
    return mel, y


def _checksums(group):
    import torch

    out = {}
    for kind, buf in (("grad", group.grad), ("value", group.flat)):
        ints = buf.view(torch.int32).to(torch.int64)
        cs = torch.cat([torch.zeros(1, dtype=torch.int64, device=buf.device), ints.cumsum(0)])
        bounds = [(name, off, off + int(torch.tensor(shape).prod())) for name, shape, off in group._specs]
        lo = torch.tensor([b[1] for b in bounds], device=buf.device)
        hi = torch.tensor([b[2] for b in bounds], device=buf.device)
        sums = (cs[hi] - cs[lo]).tolist()
        out[kind] = {b[0]: s for b, s in zip(bounds, sums)}
    return out


def _poison():
    """Jobs "graphP" / "eagerP": every torch.empty / empty_like / workspace view starts as NaN (floats) or 0x7f7f7f7f (integers).  A kernel that
    reads memory nobody wrote then shows as NaNs or as checksums that differ from the unpoisoned run's -- whatever the allocator hands out."""
    import torch

    from everyvoice_amd.train import ops

    def fill(t):
        if t.is_cuda and t.numel():
            if t.dtype.is_floating_point:
                t.fill_(float("nan"))
            elif t.dtype in (torch.int32, torch.int64):
                t.fill_(0x7F7F7F7F)
        return t

    real_empty, real_like, real_get = torch.empty, torch.empty_like, ops.Workspace.get
    torch.empty = lambda *a, **k: fill(real_empty(*a, **k))
    torch.empty_like = lambda *a, **k: fill(real_like(*a, **k))
    ops.Workspace.get = lambda self, key, numel, device: fill(real_get(self, key, numel, device))


def _fuzz(seed: int):
    sys.path.insert(0, str(ROOT / "tests"))
    from helpers import fuzz_gan_schedule

    fuzz_gan_schedule(seed)


def _rank_main(rank, world, port, mode, trials, steps, out_dir, B, S, precision):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    sys.path.insert(0, str(ROOT))
    import torch

    from everyvoice_amd.train.hifigan import HiFiGANTrainer

    tag = mode
    if mode.endswith("P"):  # job "graphP" / "eagerP": the poisoned twin of "graph" / "eager"
        _poison()
        mode = mode[:-1]
    if mode.endswith("F"):  # job "graphF" / "eagerF": the same steps with every forked chain delayed at random
        _fuzz(1000 * rank + 17)
        mode = mode[:-1]

    dev = torch.device("cuda:0")
    if world > 1:
        import torch.distributed as dist

        dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        mel, y = _batch(B * world, S)
        mel, y = mel[rank * B:(rank + 1) * B].to(dev), y[rank * B:(rank + 1) * B].to(dev)
        for trial in range(trials):
            tr = HiFiGANTrainer(device=dev, seed=5, process_group=True if world > 1 else None, use_graph=(mode == "graph"), precision=precision)
            rec = []
            for _ in range(steps):
                losses = tr.training_step(mel, y)
                torch.cuda.synchronize(dev)
                rec.append({"losses": losses, "d": _checksums(tr.d_params), "g": _checksums(tr.g_params)})
            info = {"graph_failed": tr._graph_failed, "graphs": [len(e["graphs"]) for e in tr._graphs.values()], "steps": rec}
            (Path(out_dir) / f"{tag}_rank{rank}_trial{trial}.json").write_text(json.dumps(info))
            del tr
            torch.cuda.synchronize(dev)
    finally:
        if world > 1:
            dist.destroy_process_group()


def _first_difference(a, b):
    """(step, group, kind, parameter) of the first checksum that differs between two trial records, or None."""
    for s, (ra, rb) in enumerate(zip(a["steps"], b["steps"])):
        for grp in ("d", "g"):
            for kind in ("grad", "value"):
                bad = [n for n in ra[grp][kind] if ra[grp][kind][n] != rb[grp][kind][n]]
                if bad:
                    return s, grp, kind, bad
    return None


def _fresh(args):
    import subprocess

    base = [sys.executable, str(Path(__file__).resolve()), "--jobs", args.jobs, "--world", str(args.world), "--trials", str(args.trials), "--steps", str(args.steps),
            "--batch", str(args.batch), "--samples", str(args.samples), "--precision", args.precision]
    pending, running, bad, t0 = list(range(args.fresh)), [], 0, time.time()
    while pending or running:
        while pending and len(running) < args.parallel:
            i = pending.pop(0)
            Path(args.out).parent.mkdir(parents=True, exist_ok=True)
            log = open(Path(args.out).parent / f"{Path(args.out).name}_fresh{i}.log", "w")
            running.append((i, subprocess.Popen(base + ["--out", f"{args.out}_fresh{i}"], stdout=log, stderr=subprocess.STDOUT), log))
        for item in list(running):
            i, p, log = item
            if p.poll() is not None:
                log.close()
                running.remove(item)
                lines = [ln for ln in open(log.name).read().splitlines() if ln.startswith(("ddp_repeat", "(", "a rank", "no rec"))]
                print(f"fresh {i}: rc {p.returncode}: " + " | ".join(lines[-4:]), flush=True)
                bad += p.returncode != 0
        time.sleep(0.5)
    print(f"ddp_repeat --fresh: {args.fresh - bad} / {args.fresh} comparisons clean, {args.parallel} side by side, {time.time() - t0:.0f} s")
    return 1 if bad else 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--jobs", default="graph,eager")
    ap.add_argument("--world", type=int, default=2)
    ap.add_argument("--trials", type=int, default=10)
    ap.add_argument("--steps", type=int, default=4)
    ap.add_argument("--batch", type=int, default=2)
    ap.add_argument("--samples", type=int, default=2048)
    ap.add_argument("--precision", default="bf16")
    ap.add_argument("--out", default="gpurun_out/ddp_repeat")
    ap.add_argument("--fresh", type=int, default=0, help="run the whole comparison this many times in FRESH processes (the test's situation: first "
                    "steps of a process, workspaces growing, code objects loading), --parallel of them side by side")
    ap.add_argument("--parallel", type=int, default=1)
    args = ap.parse_args()
    if args.fresh:
        return _fresh(args)
    import torch.multiprocessing as mp

    out = Path(args.out)
    out.mkdir(parents=True, exist_ok=True)
    for f in out.glob("*_trial*.json"):
        f.unlink()
    ctx = mp.get_context("spawn")
    procs = []
    t0 = time.time()
    for mode in args.jobs.split(","):
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            port = s.getsockname()[1]
        for r in range(args.world):
            p = ctx.Process(target=_rank_main, args=(r, args.world, port, mode, args.trials, args.steps, str(out), args.batch, args.samples, args.precision))
            p.start()
            procs.append(p)
    rc = 0
    for p in procs:
        p.join(3000)
        if p.exitcode != 0:
            print(f"a rank exited with {p.exitcode}")
            rc = 2
    recs = {}
    for f in sorted(out.glob("*_trial*.json")):
        mode, rank, trial = f.stem.split("_")
        recs[(mode, int(rank[4:]), int(trial[5:]))] = json.loads(f.read_text())
    if not recs:
        print("no records")
        return 2
    ref_key = min(k for k in recs if k[0] == args.jobs.split(",")[-1])  # the first trial of the last job (eager, by default)
    ref = recs[ref_key]
    n_bad = 0
    for k in sorted(recs):
        r = recs[k]
        if k[0].startswith("graph") and (r["graph_failed"] is not None or not r["graphs"]):
            print(f"{k}: graph mode did not capture: {r['graph_failed']}")
            rc = 2
        diff = _first_difference(ref, r)
        if diff is not None:
            n_bad += 1
            s, grp, kind, bad = diff
            print(f"{k} differs from {ref_key}: first at step {s}, {grp} {kind}: {len(bad)} parameters, e.g. {bad[:6]}")
    total = len(recs)
    print(f"ddp_repeat: jobs={args.jobs} world={args.world} precision={args.precision} trials={args.trials} steps={args.steps}: "
          f"{total - n_bad} / {total} records equal to {ref_key}; {time.time() - t0:.0f} s; switches: "
          + " ".join(f"{k}={v}" for k, v in sorted(os.environ.items()) if k.startswith("EVMI_")))
    return rc or (1 if n_bad else 0)


if __name__ == "__main__":
    sys.exit(main())
