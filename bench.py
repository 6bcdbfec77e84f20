#!/usr/bin/env python3
"""Headline benchmark: HiFiGAN-V1 generator inference, 22.05 kHz audio samples / second.

  python bench.py --gpus N --steps K --warmup W

N > 1: either launched by `python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...` (RANK / LOCAL_RANK /
WORLD_SIZE in the environment), or started bare -- then this process, WITHOUT touching the GPU, starts that launcher as a
child (one rank per GPU over RCCL), relays rank 0's JSON line and exits with the children's return code (the one-command
form of the reference's `everyvoice train ... --devices N --strategy ddp`, base_cli/interfaces.py:84-97,
base_cli/helpers.py:252-270).

Workload (BASELINE.json configs[1]): mel = clamp(N(-5, 2^2), -11.5129, 2.0) of
shape [32, 80, 768] per GPU, seed 1234 + rank, already resident in HBM; weights N(0, 0.01) upstream
init with weight norm folded; one step = one generator forward = 32 x 768 x 256 = 6,291,456 samples.
Inference does not shard anything between GPUs ("replicas only"): N ranks run N
independent replicas, `value` is the sum over ranks, scaling is weak.

The JSON line carries, besides the driver's contract fields:
  roofline      the dominant kernel family of the forward, timed live with HIP events on the launch
                stream (evmi_generator_forward_profiled); achieved = algorithmic FLOP per launch /
                mean launch duration, against the dense bf16 MFMA peak (2.5 PFLOP/s)
  cpu_baseline  the CPU oracle (oracle/hifigan_ref.py, a port: the reference's own vocoder code is an
                un-vendored submodule) timed on this box's host cores on a bounded slice of the
                same workload (rank 0, N = 1 only)
"""

from __future__ import annotations

import argparse
import json
import os
import socket
import subprocess
import sys
import time
from collections import defaultdict
from pathlib import Path

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

B_PER_GPU = 32
T_FRAMES = 768
HOP = 256
MFMA_PEAK_TFLOPS_BF16 = 2500.0  # dense; /opt/skills/guides/MI355X_MICROARCH.md
HBM_PEAK_GBS = 8000.0
METRIC = "hifigan_v1_infer_audio_samples_per_sec"


def parse_args(argv=None):
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=20)
    p.add_argument("--warmup", type=int, default=3)
    p.add_argument("--batch", type=int, default=B_PER_GPU)
    p.add_argument("--frames", type=int, default=T_FRAMES)
    p.add_argument("--precision", default="bf16", choices=["bf16", "f32"])
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--cpu-frames", type=int, default=768, help="mel frames per item of the CPU-baseline sample")
    p.add_argument("--cpu-batch", type=int, default=16, help="items of the bench batch the CPU baseline runs (~20 s of CPU work per pass, two passes)")
    p.add_argument("--profile-passes", type=int, default=7)
    p.add_argument("--no-train", action="store_true", help="skip the GAN-training leg (second half of the metric)")
    p.add_argument("--no-fs2", action="store_true", help="skip the FastSpeech2 feature-prediction inference leg")
    p.add_argument("--no-side-legs", action="store_true",
                   help="skip the other-precision and length-sensitivity inference legs (profiling runs: per-kernel counters then average over "
                        "launches of ONE shape)")
    p.add_argument("--train-precision", default="bf16", choices=["bf16", "f32"],
                   help="training legs: bf16 convolution operands with fp32 accumulation / master weights (BASELINE config 3 names bf16), "
                        "or the exact fp32 path; the other one is timed beside it with fewer steps")
    p.add_argument("--train-steps", type=int, default=50)
    p.add_argument("--train-warmup", type=int, default=10)
    p.add_argument("--selftest-cpu", action="store_true",
                   help="launcher self-test: ranks rendezvous over gloo on the CPU, time a dummy step and report the rank count "
                        "(tests/test_bench_dist.py; measures nothing)")
    return p.parse_args(argv)


def _free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launch_ranks(n: int, argv: list[str]) -> int:
    """`python bench.py --gpus N` without a launcher: start N ranks as children of this process through
    torch.distributed.run (never os.exec*: this also has to work under rocprofv3), pass their stderr through, relay the
    JSON line rank 0 prints and return the children's exit code.  This process never initialises the GPU."""
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: RCCL across processes needs it on this driver
    env.setdefault("OMP_NUM_THREADS", "8")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), str(Path(__file__).resolve()), *argv]
    proc = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
    line = None
    for ln in proc.stdout.splitlines():
        if ln.startswith("{") and '"metric"' in ln:
            line = ln
        else:
            print(ln, file=sys.stderr)
    if proc.returncode != 0 or line is None:
        print(f"bench.py: the {n}-rank launch failed (rc {proc.returncode}, JSON line {'present' if line else 'missing'})", file=sys.stderr)
        return proc.returncode or 1
    print(line, flush=True)
    return 0


def selftest_cpu(args, rank: int, world: int) -> int:
    """The launcher path with the GPU legs replaced by a sleep: rendezvous, barrier / max-over-ranks timing, one JSON line."""
    import torch
    import torch.distributed as dist

    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo")
    ones = torch.ones(1)
    if world > 1:
        dist.all_reduce(ones)

    def max_reduce(x):
        t = torch.tensor([x], dtype=torch.float64)
        if world > 1:
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    elapsed = timed_region(lambda: time.sleep(0.002 * (rank + 1)), args.steps, args.warmup, lambda: None,
                           dist.barrier if world > 1 else (lambda: None), max_reduce)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps({"metric": "selftest_cpu", "value": world * args.steps / elapsed, "n_gpus": world, "ranks_seen": int(ones.item()),
                          "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3}))
    return 0


def dist_env():
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    return rank, local_rank, world


def timed_region(step_fn, steps: int, warmup: int, sync_fn, barrier_fn, max_reduce_fn) -> float:
    """W untimed warm-up steps, then EXACTLY `steps` steps bracketed by barrier + device sync on both
    sides; returns the MAX over ranks of the elapsed seconds."""
    import gc

    for _ in range(warmup):
        step_fn()
    sync_fn()
    # the cyclic collector stays out of the timed steps (as `timeit` does): late in the process a full collection walks the
    # previous legs' objects for tens of milliseconds, and whether one fell into a 60 ms region made the FastSpeech2 inference leg
    # read 5.9 or 8.8 ms per batch from run to run on the same box
    gc.collect()
    was_enabled = gc.isenabled()
    gc.disable()
    try:
        barrier_fn()
        sync_fn()
        t0 = time.perf_counter()
        for _ in range(steps):
            step_fn()
        sync_fn()
        barrier_fn()
        sync_fn()
        elapsed = time.perf_counter() - t0
    finally:
        if was_enabled:
            gc.enable()
    return max_reduce_fn(elapsed)


def upstream_init_generator(precision: str):
    """HiFi-GAN V1 (schema defaults), upstream N(0, 0.01) init, fixed seed."""
    import torch

    from everyvoice_amd.config import HiFiGANConfig
    from everyvoice_amd.vocoder import HiFiGANGenerator

    torch.manual_seed(1234)
    return HiFiGANGenerator(HiFiGANConfig(), precision=precision)


def synthetic_mel(batch: int, frames: int, seed: int):
    import torch

    g = torch.Generator().manual_seed(seed)
    return (torch.randn(batch, 80, frames, generator=g) * 2.0 - 5.0).clamp(-11.5129, 2.0)


def _median(xs):
    xs = sorted(xs)
    n = len(xs)
    return xs[n // 2] if n % 2 else 0.5 * (xs[n // 2 - 1] + xs[n // 2])


def roofline_from_records(passes: list[list[dict]]) -> dict:
    """Aggregate HIP-event launch records by kernel family; describe the dominant one.  Every figure is a MEDIAN (over the launches
    of a family for the per-launch duration, over the passes for the per-forward sums): one launch that a pre-emption or a clock
    dip stretched twenty-fold does not move it (round 5's driver line was a 3-pass mean and named the wrong kernel)."""
    launches, per_pass = defaultdict(list), defaultdict(list)
    for recs in passes:
        tot = defaultdict(float)
        for r in recs:
            k = r["kernel"]
            launches[k].append(r["ms"])
            tot[k] += r["ms"]
        for k, v in tot.items():
            per_pass[k].append(v)
    # algorithmic work per launch: the mean over a family's launches of one pass (its launches may differ in length)
    fl_launch, by_launch, n_launch = {}, {}, {}
    for k in launches:
        rs = [r for r in passes[0] if r["kernel"] == k]
        n_launch[k] = len(rs)
        fl_launch[k] = sum(r["flops"] for r in rs) / max(len(rs), 1)
        by_launch[k] = sum(r["bytes"] for r in rs) / max(len(rs), 1)
    fam_ms = {k: _median(v) for k, v in per_pass.items()}          # per forward
    all_ms = _median([sum(r["ms"] for r in recs) for recs in passes])
    all_flops = sum(r["flops"] for r in passes[0])
    dom = max(fam_ms, key=fam_ms.get)
    avg_ms = _median(launches[dom])
    achieved = fl_launch[dom] / (avg_ms * 1e-3) / 1e12
    fams = sorted(fam_ms, key=fam_ms.get, reverse=True)
    return {
        "bound": "mfma",
        "kernel": dom,
        "achieved": round(achieved, 2),
        "peak": MFMA_PEAK_TFLOPS_BF16,
        "unit": "TFLOP/s",
        "frac": round(achieved / MFMA_PEAK_TFLOPS_BF16, 4),
        "traffic": None,
        "avg_launch_ms": round(avg_ms, 4),
        "statistic": "median over launches (per-launch figures) and over passes (per-forward sums)",
        # the plain mean beside it: what `rocprofv3 --stats` prints as AverageNs for the same kernel (profiles/*_kernel_stats.csv)
        "mean_launch_ms": round(sum(launches[dom]) / len(launches[dom]), 4),
        "frac_by_mean": round(fl_launch[dom] / (sum(launches[dom]) / len(launches[dom]) * 1e-3) / 1e12 / MFMA_PEAK_TFLOPS_BF16, 4),
        "launches_per_forward": n_launch[dom],
        "share_of_forward_time": round(fam_ms[dom] / all_ms, 4),
        "algorithmic_gflop_per_launch": round(fl_launch[dom] / 1e9, 3),
        "algorithmic_hbm_gbs": round(by_launch[dom] / (avg_ms * 1e-3) / 1e9, 1),
        "whole_forward": {
            "event_ms": round(all_ms, 3),
            "tflops": round(all_flops / (all_ms * 1e-3) / 1e12, 2),
            "by_kernel_ms": {k: round(fam_ms[k], 3) for k in fams[:8]},
        },
    }


def measured_roofline(gen, mel, timed_ms: float, n_passes: int, tolerance: float = 0.10, max_rounds: int = 4) -> dict:
    """The roofline block, measured directly behind the headline's timed region (same clocks, same allocator state) and CHECKED
    against it: a pass whose per-kernel HIP-event durations do not add up to the timed step within `tolerance` is rejected (the
    events serialise the forward, so a sound pass reads a few per cent ABOVE the timed step, never far from it) and measured again,
    in this process.  `check` says what was kept."""
    kept, rejected, rounds = [], 0, 0
    while len(kept) < n_passes and rounds < max_rounds:
        rounds += 1
        for _ in range(n_passes - len(kept)):
            _, recs = gen.forward_profiled(mel)
            ev = sum(r["ms"] for r in recs)
            if abs(ev - timed_ms) <= tolerance * timed_ms:
                kept.append(recs)
            else:
                rejected += 1
    ok = len(kept) >= max(3, n_passes // 2)
    if not kept:  # nothing agreed with the timed step: report what the last pass read and say it is not evidence
        kept = [recs]
    roof = roofline_from_records(kept)
    roof["check"] = {"event_ms": roof["whole_forward"]["event_ms"], "timed_ms": round(timed_ms, 3), "passes": len(kept), "rejected": rejected,
                     "tolerance": tolerance, "consistent": bool(ok and abs(roof["whole_forward"]["event_ms"] - timed_ms) <= tolerance * timed_ms)}
    return roof


def git_head() -> str | None:
    """The commit this tree was built from: `git rev-parse HEAD`, or the `.git_head` file a gpurun snapshot carries (no .git there)."""
    import subprocess

    try:
        r = subprocess.run(["git", "-C", str(ROOT), "rev-parse", "HEAD"], capture_output=True, text=True, timeout=10)
        if r.returncode == 0 and r.stdout.strip():
            return r.stdout.strip()
    except (OSError, subprocess.SubprocessError):
        pass
    f = ROOT / ".git_head"
    return f.read_text().strip() if f.exists() else None


def profile_commit(summary_path, key: str = "commit") -> str | None:
    """The commit (or product-code fingerprint, key "code") a committed PMC summary was collected at (its .meta.json), so a reader sees
    whether `traffic` is from this tree."""
    meta = Path(str(summary_path).replace("_pmc_summary.json", "_pmc_summary.meta.json"))
    try:
        return json.loads(meta.read_text()).get(key)
    except (OSError, ValueError):
        return None


def code_fingerprint() -> str:
    """sha256[:16] of the product's sources (tools/code_fingerprint.py): equal between this line and a counter summary's ``code`` means
    the counters were collected on the code this line timed, whatever commits (profiles, docs) came in between."""
    sys.path.insert(0, str(ROOT / "tools"))
    from code_fingerprint import code_fingerprint as fp

    return fp(ROOT)


def recorded_pmc_traffic(kernel: str):
    """HBM bytes per launch of `kernel` from the newest committed PMC summary (profiles/*_pmc_summary.json,
    produced by tools/gpu_profile.sh + tools/pmc_summarize.py from separate rocprofv3 --pmc passes of this
    same bench command; FETCH_SIZE doubled per the gfx950 correction).  PMC counters cannot be read from
    inside the process, so this is the recorded measurement, or None when the kernel has none."""
    best = None
    for f in sorted((ROOT / "profiles").glob("*_pmc_summary.json")):
        try:
            e = json.loads(f.read_text()).get(kernel)
        except (OSError, ValueError):
            continue
        if e and "hbm_bytes_per_launch" in e:
            best = (e["hbm_bytes_per_launch"], f.name)
    return best


def cpu_baseline(frames: int, batch: int) -> dict:
    """The CPU oracle on this box's host cores, on a bounded slice of the same workload."""
    import torch

    from oracle.hifigan_ref import GeneratorRef

    torch.manual_seed(1234)
    ref = GeneratorRef().eval().remove_weight_norm()
    mel = synthetic_mel(batch, frames, 1234)
    # host cores this process may use; torch's intra-op pool oversubscribes badly past ~64 threads on
    # these convolutions, so pick the fastest of a few thread counts on a tiny probe and say which
    avail = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    best = None
    with torch.no_grad():
        probe = mel[:, :, : min(128, mel.shape[2])]  # the sample's own batch size, long enough that the thread count matters as on the sample
        for n in sorted({min(avail, c) for c in (16, 32, 64)}):  # (8 and > 64 threads never won at this batch size on these hosts)
            torch.set_num_threads(n)
            ref(probe)
            t0 = time.perf_counter()
            ref(probe)
            dt = time.perf_counter() - t0
            if best is None or dt < best[0]:
                best = (dt, n)
        cores = best[1]
        torch.set_num_threads(cores)
        times = []
        for _ in range(2):  # the better of two passes (the probe above warmed the pools): the baseline is given its best case
            t0 = time.perf_counter()
            wav = ref(mel)
            times.append(time.perf_counter() - t0)
        dt = min(times)
    return {
        "value": round(wav.numel() / dt, 1),
        "unit": "samples/s",
        "cores": cores,
        "host_cores_available": avail,
        "kind": "port",
        "sample": f"oracle/hifigan_ref.py GeneratorRef fp32, torch {torch.__version__} CPU, {cores} threads, "
                  f"mel [{batch},80,{frames}] slice of the bench input ({wav.numel()} samples in {dt:.2f} s, better of 2 passes; thread count probed over 16 / 32 / 64 at this batch size)",
    }


def _median_time(fn, warmup: int, repeats: int) -> float:
    for _ in range(warmup):
        fn()
    ts = []
    for _ in range(repeats):
        t0 = time.perf_counter()
        fn()
        ts.append(time.perf_counter() - t0)
    return sorted(ts)[len(ts) // 2]


def cpu_baseline_train(cores: int, batch: int = 8) -> dict:
    """One full GAN step (D step, then G step, AdamW on both) of the torch-CPU oracle on `batch` 8192-sample segments;
    steps/s scaled to the bench's 16 segments per step (the step is linear in the batch)."""
    import torch
    import torch.nn.functional as F

    from oracle import mel_ref
    from oracle.hifigan_ref import (GeneratorRef, MultiPeriodDiscriminatorRef, MultiScaleDiscriminatorRef, discriminator_loss_ref,
                                    feature_loss_ref, generator_loss_ref)

    torch.manual_seed(1234)
    torch.set_num_threads(cores)
    g, mpd, msd = GeneratorRef().train(), MultiPeriodDiscriminatorRef().train(), MultiScaleDiscriminatorRef().train()
    kw = dict(lr=2e-4, betas=(0.8, 0.99), eps=1e-8, weight_decay=0.01)
    opt_g = torch.optim.AdamW(g.parameters(), **kw)
    opt_d = torch.optim.AdamW(list(mpd.parameters()) + list(msd.parameters()), **kw)
    y = 0.3 * torch.tanh(torch.randn(batch, 1, 8192))
    mel = mel_ref.mel_spectrogram_ref(y.squeeze(1))[:, :, :32]

    def step():
        y_hat = g(mel)
        opt_d.zero_grad()
        r1, g1, _, _ = mpd(y, y_hat.detach())
        r2, g2, _, _ = msd(y, y_hat.detach())
        (discriminator_loss_ref(r1, g1) + discriminator_loss_ref(r2, g2)).backward()
        opt_d.step()
        opt_g.zero_grad()
        loss_mel = F.l1_loss(mel_ref.mel_spectrogram_ref(y.squeeze(1)), mel_ref.mel_spectrogram_ref(y_hat.squeeze(1))) * 45
        _, g1, fr1, fg1 = mpd(y, y_hat)
        _, g2, fr2, fg2 = msd(y, y_hat)
        (generator_loss_ref(g1) + generator_loss_ref(g2) + feature_loss_ref(fr1, fg1) + feature_loss_ref(fr2, fg2) + loss_mel).backward()
        opt_g.step()

    dt = _median_time(step, warmup=1, repeats=3)
    return {"value": round(batch / 16.0 / dt, 4), "unit": "steps/s", "cores": cores, "kind": "port",
            "sample": f"oracle/hifigan_ref.py full GAN step (torch autograd + AdamW) on {batch} segments of 8192 samples, {cores} threads: "
                      f"{dt:.2f} s (median of 3 after 1 warm-up); scaled to 16 segments per step"}


def cpu_baseline_fs2(cores: int, batch: int = 32, repeats: int = 10) -> dict:
    import torch

    from oracle.fs2_ref import FastSpeech2Ref
    sys.path.insert(0, str(ROOT / "tools"))
    from fs2_bench import synthetic_batch

    torch.manual_seed(1234)
    torch.set_num_threads(cores)
    ref = FastSpeech2Ref().eval()
    ids, lens, durs, t_i = synthetic_batch(32, 1234)
    ids, lens, durs = ids[:batch], lens[:batch], durs[:batch]
    L = int(lens.max())
    ref(ids[:, :L], lens, durations=durs[:, :L])
    t0 = time.perf_counter()
    for _ in range(repeats):
        ref(ids[:, :L], lens, durations=durs[:, :L])
    dt = (time.perf_counter() - t0) / repeats
    frames = int(t_i[:batch].sum())
    return {"value": round(frames / dt, 1), "unit": "frames/s", "cores": cores, "kind": "port",
            "sample": f"oracle/fs2_ref.py FastSpeech2Ref fp32, first {batch} utterances of the bench batch ({frames} frames in {dt:.2f} s, mean of {repeats} passes), {cores} threads"}


def _graph_info(trainer) -> dict:
    """Whether the timed steps were HIP-graph replays (and of how many captured stretches), or why not."""
    failed = getattr(trainer, "_graph_failed", None)
    entries = list(getattr(trainer, "_graphs", {}).values())
    used = bool(entries) and failed is None
    return {"used": used, "reason": failed if failed else (None if used else "no shape was captured (use_graph off or too few steps)"),
            "stretches": [len(e["graphs"]) for e in entries]}


def train_leg(args, dev, rank, world, use_dist, barrier, max_reduce) -> dict:
    """Second half of BASELINE.json's metric: HiFiGAN-V1 GAN training steps/s at bs 16 per GPU (config 4:
    generator + MPD + MSD, LSGAN + feature matching + 45 x mel L1, two AdamW optimisers), data parallel with one
    RCCL all-reduce per optimiser (discriminator 283 MB, generator 56 MB of fp32 gradients) when N > 1.
    Synthetic segments y = 0.3 * tanh(N(0,1)) [16, 1, 8192] per rank (seed 1234 + rank), mel from the device
    front-end.  --train-precision bf16 (default): bf16 operands, fp32 accumulation / master weights / activations -- forward,
    input gradient and weight gradient of every convolution on the packed-input kernels (conv_cbt_bf16_pk.hip,
    conv_wgrad_bf16_pk.hip); f32: the exact path on the fp32 matrix cores (conv_cbt_f32_mfma.hip, conv_wgrad_f32_mfma.hip);
    the other one is timed beside it.  The roofline object prices the step with the FLOPs its own launches issue (counted by
    train/ops.py: every convolution / GEMM call of one eager step, 3.44 TFLOP at this shape; SURVEY.md 8(d)'s estimate was 25.8 MFLOP
    per segment sample) against the dense bf16 MFMA peak (bf16) or the 157 TFLOP/s fp32 matrix peak."""
    import torch

    from everyvoice_amd.spectral import MelSpectrogram
    from everyvoice_amd.train.hifigan import HiFiGANTrainer

    B, S = 16, 8192
    torch.cuda.empty_cache()
    g = torch.Generator().manual_seed(1234 + rank)
    y = (0.3 * torch.tanh(torch.randn(B, 1, S, generator=g))).to(dev)
    mel = MelSpectrogram()(y.squeeze(1), log=True)[:, :, : S // 256].contiguous()
    prec = args.train_precision
    trainer = HiFiGANTrainer(device=dev, process_group=True if use_dist else None, precision=prec, use_graph=True)
    losses = {}

    def step():
        losses.update(trainer.training_step(mel, y))

    from everyvoice_amd.train import autograd as ag
    from everyvoice_amd.train import ops as train_ops

    ag.activation_elements(reset=True)
    train_ops.flop_counter(reset=True)
    step()  # (eager: the graph is captured on a later warm-up step) -- counts the activations and the contractions one step issues
    act_elems = ag.activation_elements(reset=True)
    counted_flops = train_ops.flop_counter(reset=True)
    elapsed = timed_region(step, args.train_steps, args.train_warmup, lambda: torch.cuda.synchronize(dev), barrier, max_reduce)
    graph_info = _graph_info(trainer)
    comm_ms = None
    if use_dist:  # the same steps again with events around every bucket's all-reduce (on its side stream): exchange ms per step
        reds = getattr(trainer, "_dp_reducers", ())
        for r in reds:
            r.timing = []
        n_comm = max(3, args.train_steps // 5)
        for _ in range(n_comm):
            step()
        torch.cuda.synchronize(dev)
        comm_ms = round(sum(r.comm_ms() for r in reds) / n_comm, 3) if reds else None
        for r in reds:
            r.timing = None
    # the other precision beside it (same trainer object, fewer steps)
    other = "f32" if prec == "bf16" else "bf16"
    trainer.precision = other
    n_other = 5
    elapsed_other = timed_region(step, n_other, 3, lambda: torch.cuda.synchronize(dev), barrier, max_reduce)
    trainer.precision = prec
    # BASELINE config 4 as written names the multi-resolution STFT loss: the same step with it added to the 45 x mel-L1 term
    del trainer
    trainer_mr = HiFiGANTrainer(device=dev, process_group=True if use_dist else None, precision=prec, reconstruction_loss="mel+mrstft", use_graph=True)
    losses_mr = {}
    n_mr = args.train_steps
    elapsed_mr = timed_region(lambda: losses_mr.update(trainer_mr.training_step(mel, y)), n_mr, args.train_warmup, lambda: torch.cuda.synchronize(dev), barrier, max_reduce)
    graph_info_mr = _graph_info(trainer_mr)
    params = {"generator": trainer_mr.g_params.numel(), "discriminators": trainer_mr.d_params.numel()}
    del trainer_mr
    # algorithmic FLOPs of one step per GPU: the contractions the tape issued in one eager step, 2 per multiply-add, convolutions at
    # their dense B * T_out * C_out * (C_in / groups) * k products (ops.flop_counter) -- SURVEY.md 8(d)'s estimate was 25.8 MFLOP per
    # segment sample (3.38 TFLOP per step)
    flop_per_step = counted_flops
    tflops = flop_per_step * args.train_steps / elapsed / 1e12
    # bf16 mode: forward and input-gradient convolutions on the bf16 matrix cores (2.5 PFLOP/s dense), weight gradients still on
    # the fp32 ones (157 TFLOP/s): priced against the bf16 peak, the stricter denominator
    peak = MFMA_PEAK_TFLOPS_BF16 if prec == "bf16" else 157.0
    n_params = params["generator"] + params["discriminators"]
    # algorithmic HBM bytes of one step in the fp32 storage the tape uses: every activation written once and read once going
    # forward, read once more + its gradient written and read going backward (5 passes); every parameter: weight norm (3),
    # three forward reads, weight-gradient write, weight-norm backward (5), optimiser (7 streams) -- 19 passes
    algorithmic_bytes = 4 * (5 * act_elems + 19 * n_params)
    roof = {"bound": "mfma", "achieved": round(tflops, 2), "peak": peak, "unit": "TFLOP/s", "frac": round(tflops / peak, 4),
            "traffic": None, "flop_per_step_per_gpu": flop_per_step, "flop_count": "exact: counted from the step's own launches (ops.flop_counter)",
            "survey_estimate_flop_per_step": 25.8e6 * B * S, "scope": "whole training step",
            "algorithmic_bytes_per_step": algorithmic_bytes, "activation_elements_per_step": act_elems,
            "algorithmic_hbm_gbs": round(algorithmic_bytes * args.train_steps / elapsed / 1e9, 1)}
    pmc_files = sorted((ROOT / "profiles").glob("*train_pmc_summary.json"))
    if pmc_files:  # recorded rocprofv3 --pmc passes (tools/gpu_profile_train.sh): the kernel with the most HBM reads per step
        pmc = json.loads(pmc_files[-1].read_text())
        # the kernel FAMILY (template instantiations together) with the most active GPU cycles over the profiled steps
        fam = defaultdict(lambda: {"cycles": 0.0, "mfma": 0.0, "lds": 0.0, "bytes": 0.0, "launches": 0})
        for kname, v in pmc.items():
            f = fam[kname.split("<")[0].replace("evmi::", "")]
            cyc = v.get("active_cycles_per_launch", 0.0) * v["launches"]
            f["cycles"] += cyc
            f["mfma"] += v.get("mfma_busy_frac", 0.0) * cyc
            f["lds"] += v.get("lds_bank_conflict_frac", 0.0) * cyc
            f["bytes"] += (v["fetch_bytes_per_launch"] + v["write_bytes_per_launch"]) * v["launches"]
            f["launches"] += v["launches"]
        name, top = max(fam.items(), key=lambda kv: kv[1]["cycles"])
        steps_profiled = json.loads(pmc_files[-1].with_suffix(".meta.json").read_text())["steps"] if pmc_files[-1].with_suffix(".meta.json").exists() else 3
        roof["traffic"] = round(sum(v["launches"] * (v["fetch_bytes_per_launch"] + v["write_bytes_per_launch"]) for v in pmc.values()) / steps_profiled)
        roof["traffic_over_algorithmic"] = round(roof["traffic"] / algorithmic_bytes, 2)
        meta = json.loads(pmc_files[-1].with_suffix(".meta.json").read_text()) if pmc_files[-1].with_suffix(".meta.json").exists() else {}
        if "hbm_bytes_per_step" in meta:  # (the summariser's own total: counters with the calibration of profiles/*_pmc_calibration.json applied)
            roof["traffic"] = round(meta["hbm_bytes_per_step"])
            roof["traffic_over_algorithmic"] = round(roof["traffic"] / algorithmic_bytes, 2)
        roof["traffic_source"] = (f"profiles/{pmc_files[-1].name}: HBM bytes per step, all kernels (FETCH_SIZE + WRITE_SIZE, separate --pmc passes; "
                                  f"{meta.get('convention', 'raw KiB counters')})")
        roof["traffic_commit"] = meta.get("commit")
        roof["traffic_code"] = meta.get("code")
        roof["dominant_kernel"] = {"name": name, "mfma_busy_frac": round(top["mfma"] / max(top["cycles"], 1.0), 4),
                                   "lds_bank_conflict_frac": round(top["lds"] / max(top["cycles"], 1.0), 4),
                                   "hbm_bytes_per_launch": round(top["bytes"] / max(top["launches"], 1)),
                                   "share_of_active_cycles": round(top["cycles"] / max(sum(f["cycles"] for f in fam.values()), 1.0), 4),
                                   "profiled_precision": "bf16" if "pk_kernel" in name else "f32"}
    return {
        "roofline": roof,
        "metric": "hifigan_v1_gan_train_steps_per_sec",
        "value": round(args.train_steps / elapsed, 3),
        "unit": "steps/s",
        "ms_per_step": round(elapsed / args.train_steps * 1e3, 2),
        "steps": args.train_steps,
        "warmup": args.train_warmup,
        "batch_per_gpu": B,
        "global_batch": B * world,
        "segment_samples": S,
        "scaling": "weak",
        "dtype": prec,
        "other_precision": {"dtype": other, "value": round(n_other / elapsed_other, 3), "unit": "steps/s", "ms_per_step": round(elapsed_other / n_other * 1e3, 2), "steps": n_other},
        "with_mrstft_loss": {"reconstruction_loss": "mel+mrstft", "value": round(n_mr / elapsed_mr, 3), "unit": "steps/s",
                             "ms_per_step": round(elapsed_mr / n_mr * 1e3, 2), "steps": n_mr, "warmup": args.train_warmup,
                             "g_stft": round(losses_mr.get("g_stft", 0.0), 4), "graph": graph_info_mr},
        "graph": graph_info,
        "allreduce_ms_per_step": comm_ms,
        "parallelism": f"dp{world}" + (" (RCCL all-reduce of 2 flat gradient buffers per step, bucketed, overlapped with backward)" if world > 1 else ""),
        "params": params,
        "last_losses": {k: round(v, 4) for k, v in losses.items()},
    }


def fs2_leg(args, dev, rank, world, barrier, max_reduce) -> dict:
    """FastSpeech2 feature prediction, inference: text ids -> mel on an LJSpeech-shaped synthetic
    batch (8d C3 shapes: B = 32, L ~ N(99.9, 34) in [12, 187], T ~ 5.67 L, durations given so the length is fixed),
    default model sizes, random parameters; dense layers with bf16 operands (--train-precision, default) or exact fp32, the
    other timed beside it.  Replicas only across GPUs."""
    import torch

    from everyvoice_amd.fs2 import FastSpeech2
    sys.path.insert(0, str(ROOT / "tools"))
    from fs2_bench import forward_flops, synthetic_batch

    prec = args.train_precision
    # every leg starts from an empty allocator cache, as the same workload does in a process of its own: served from the blocks the
    # previous legs left behind, this leg read 5.9 to 8.8 ms per batch from run to run (fresh allocations: 5.9, every time)
    torch.cuda.empty_cache()
    model = FastSpeech2(device=dev, precision=prec).init_random(1234)
    ids, lens, durs, t_i = synthetic_batch(32, 1234 + rank)
    ids, lens, durs = ids.to(dev), lens.to(dev), durs.to(dev)
    steps, warmup = 30, 5
    elapsed = timed_region(lambda: model(ids, lens, durations=durs), steps, warmup, lambda: torch.cuda.synchronize(dev), barrier, max_reduce)
    other = "f32" if prec == "bf16" else "bf16"
    model.precision = other
    elapsed_other = timed_region(lambda: model(ids, lens, durations=durs), 3, 1, lambda: torch.cuda.synchronize(dev), barrier, max_reduce)
    model.precision = prec
    frames = int(t_i.sum())
    flops = forward_flops(lens.cpu(), t_i, int(ids.shape[1]), int(t_i.max()), 32)
    tflops = flops * steps / elapsed / 1e12
    peak = MFMA_PEAK_TFLOPS_BF16 if prec == "bf16" else 157.0
    traffic, traffic_src = None, None
    metas = sorted((ROOT / "profiles").glob("*fs2infer_pmc_summary.meta.json"))
    if metas:  # recorded rocprofv3 --pmc passes over this forward (tools/gpu_profile_fs2_infer.sh): HBM bytes per forward, all kernels
        meta = json.loads(metas[-1].read_text())
        traffic = round(meta["hbm_bytes_per_step"])
        traffic_src = (f"profiles/{metas[-1].name.replace('.meta', '')}: FETCH_SIZE + WRITE_SIZE of every kernel of a forward (separate --pmc passes; "
                       f"{meta.get('convention', 'raw KiB counters')}; collected at commit {meta.get('commit')}, code {meta.get('code')})")
    return {"roofline": {"bound": "mfma", "achieved": round(tflops, 2), "peak": peak, "unit": "TFLOP/s", "frac": round(tflops / peak, 4),
                         "traffic": traffic, "traffic_source": traffic_src, "flop_per_batch": flops, "scope": "whole forward (dense layers on the padded grids)"},
            "metric": "fastspeech2_infer_mel_frames_per_sec", "value": round(world * frames * steps / elapsed, 1), "unit": "frames/s",
            "ms_per_batch": round(elapsed / steps * 1e3, 3), "batch": 32, "max_tokens": int(ids.shape[1]), "frames_per_batch": frames,
            "dtype": prec, "steps": steps, "warmup": warmup, "parallelism": f"replicas x{world}",
            "other_precision": {"dtype": other, "ms_per_batch": round(elapsed_other / 3 * 1e3, 3), "value": round(world * frames * 3 / elapsed_other, 1), "unit": "frames/s"},
            "realtime_factor": round(world * frames * 256 / 22050.0 * steps / elapsed, 1)}


def fs2_train_leg(args, dev, rank, world, use_dist, barrier, max_reduce) -> dict:
    """BASELINE config 3: FastSpeech2 feature-prediction training, LJSpeech-shaped synthetic batch of 32 per GPU, default model
    (learn_alignment on: aligner + monotonic search + CTC loss inside the step), step = forward + losses + backward + clipped
    Noam AdamW; data parallel with one RCCL
    all-reduce of the flat gradient buffer when N > 1.  --train-precision bf16 (what the config names; default): bf16 operands of
    the dense layers, fp32 accumulation, master weights and activations; f32: the exact path, timed beside it.
    Algorithmic FLOPs = 3 x the forward's (dx and dw of every product)."""
    import torch

    from everyvoice_amd.train.fs2 import FastSpeech2Trainer
    sys.path.insert(0, str(ROOT / "tools"))
    from fs2_bench import forward_flops
    from fs2_train_bench import training_batch

    prec = args.train_precision
    torch.cuda.empty_cache()
    tr = FastSpeech2Trainer(device=dev, process_group=True if use_dist else None, precision=prec, use_graph=True)  # default config: learn_alignment on
    batch, t_i = training_batch(32, 1234 + rank, device=dev)
    torch.cuda.synchronize(dev)
    tr.batch_ready = True  # inputs resident and complete before the timed region: the next batch's layout passes run under the current step
    out = {}

    def step():
        out["losses"] = tr.training_step(batch)

    from everyvoice_amd.train import autograd as ag

    steps, warmup = 30, 5
    ag.activation_elements(reset=True)
    step()  # (eager: shapes are captured on a later warm-up step) -- counts the activation tensors one step creates
    act_elems = ag.activation_elements(reset=True)
    elapsed = timed_region(step, steps, warmup, lambda: torch.cuda.synchronize(dev), barrier, max_reduce)
    # the same step run eagerly (what a shape costs until it has been captured, and what EVMI / a caller without graphs gets)
    tr.use_graph = False
    elapsed_eager = timed_region(step, 10, 2, lambda: torch.cuda.synchronize(dev), barrier, max_reduce)
    tr.use_graph = True
    other = "f32" if prec == "bf16" else "bf16"
    tr.precision = other
    graph_info = _graph_info(tr)
    n_other = 10
    elapsed_other = timed_region(step, n_other, 4, lambda: torch.cuda.synchronize(dev), barrier, max_reduce)
    tr.precision = prec
    flops = 3.0 * forward_flops(batch["lens"], t_i, int(batch["ids"].shape[1]), int(t_i.max()), 32)
    tflops = flops * steps / elapsed / 1e12
    peak = MFMA_PEAK_TFLOPS_BF16 if prec == "bf16" else 157.0
    # algorithmic HBM bytes, priced as the GAN leg's: every activation written once and read once forward, read once more and its
    # gradient written and read backward (5 passes of 4 bytes); every parameter through weight use, gradient and AdamW (19 passes)
    n_params = tr.params.numel()
    algorithmic_bytes = 4 * (5 * act_elems + 19 * n_params)
    traffic, traffic_src = None, None
    metas = sorted((ROOT / "profiles").glob("*fs2_pmc_summary.meta.json"))
    if metas:  # recorded rocprofv3 --pmc passes over this step (tools/gpu_profile_fs2_train.sh): HBM bytes per step, all kernels
        meta = json.loads(metas[-1].read_text())
        traffic = round(meta["hbm_bytes_per_step"])
        traffic_src = (f"profiles/{metas[-1].name.replace('.meta', '')}: FETCH_SIZE + WRITE_SIZE of every kernel of the step (separate --pmc passes; "
                       f"{meta.get('convention', 'raw KiB counters')}; collected at commit {meta.get('commit')}, code {meta.get('code')})")
    return {"metric": "fastspeech2_train_steps_per_sec_bs32", "value": round(steps / elapsed, 3), "unit": "steps/s",
            "ms_per_step": round(elapsed / steps * 1e3, 2), "steps": steps, "warmup": warmup, "batch_per_gpu": 32, "global_batch": 32 * world,
            "frames_per_sec": round(world * int(t_i.sum()) * steps / elapsed, 1), "scaling": "weak", "dtype": prec,
            "other_precision": {"dtype": other, "value": round(n_other / elapsed_other, 3), "unit": "steps/s", "ms_per_step": round(elapsed_other / n_other * 1e3, 2), "steps": n_other},
            "graph": graph_info,
            "eager": {"value": round(10 / elapsed_eager, 3), "unit": "steps/s", "ms_per_step": round(elapsed_eager / 10 * 1e3, 2), "steps": 10,
                      "note": "the same resident batch without graph replay: the rate of shapes not (yet) captured; train_base_command runs "
                              "lightning.FastSpeech2(use_graph=True): captured shapes replay; graph_buckets=(16, 64) (opt-in) pads batches to multiples so that a "
                              "loader's variable lengths fall on few shapes"},
            "parallelism": f"dp{world}" + (" (RCCL all-reduce of the flat gradient buffer in two buckets, the first under the rest of backward)" if world > 1 else ""),
            "params": tr.params.numel(), "last_losses": {k: round(float(v), 4) for k, v in out["losses"].items()},
            "roofline": {"bound": "mfma", "achieved": round(tflops, 2), "peak": peak, "unit": "TFLOP/s", "frac": round(tflops / peak, 4),
                         "traffic": traffic, "traffic_source": traffic_src, "flop_per_step": flops, "scope": "whole step (forward + backward + optimiser)",
                         "algorithmic_bytes_per_step": algorithmic_bytes, "activation_elements_per_step": act_elems,
                         "traffic_over_algorithmic": round(traffic / algorithmic_bytes, 2) if traffic else None}}


def cpu_baseline_fs2_train(cores: int, batch: int = 8) -> dict:
    import torch

    from oracle.fs2_ref import FastSpeech2Ref, training_losses_ref
    sys.path.insert(0, str(ROOT / "tools"))
    from fs2_train_bench import training_batch

    torch.manual_seed(1234)
    torch.set_num_threads(cores)
    ref = FastSpeech2Ref().train()
    full, t_i = training_batch(32, 1234, learn_alignment=False)  # the CPU sample uses given durations (no aligner): a lower bound on its work
    L = int(full["lens"][:batch].max())
    b = {k: (v[:batch, :L] if v.dim() >= 2 and k != "mel" else v[:batch]) for k, v in full.items()}
    opt = torch.optim.AdamW(ref.parameters(), lr=1e-4)

    def step():
        opt.zero_grad()
        training_losses_ref(ref, b)["total"].backward()
        torch.nn.utils.clip_grad_norm_(ref.parameters(), 1.0)
        opt.step()

    dt = _median_time(step, warmup=1, repeats=3)
    return {"value": round(batch / 32.0 / dt, 4), "unit": "steps/s", "cores": cores, "kind": "port",
            "sample": f"oracle/fs2_ref.py training step (torch autograd + clipping + AdamW) on the first {batch} utterances of the bench batch, "
                      f"{cores} threads: {dt:.2f} s (median of 3 after 1 warm-up); scaled to 32 utterances per step"}


def _side_legs(args, other, gen, mel, samples_per_step, dev, rank, world, barrier, max_reduce):
    """The other arithmetic beside the headline (the reference computes in fp32; bf16 operands with fp32 accumulation are SURVEY 8(d)
    C2's contract) and the length sensitivity SURVEY.md 8(d) C2 asks for."""
    import torch

    model_o = upstream_init_generator(other).to(dev).eval()
    n_other, w_other = 20, 5

    elapsed_o = timed_region(lambda: model_o.generator(mel), n_other, w_other, lambda: torch.cuda.synchronize(dev), barrier, max_reduce)
    flops_step = 2.0 * gen.macs_per_sample() * samples_per_step
    peak_o = 157.0 if other == "f32" else MFMA_PEAK_TFLOPS_BF16
    tf_o = flops_step * n_other / elapsed_o / 1e12
    other_precision = {"dtype": other, "value": round(world * samples_per_step * n_other / elapsed_o, 1), "unit": "samples/s",
                       "ms_per_step": round(elapsed_o / n_other * 1e3, 3), "steps": n_other, "warmup": w_other,
                       "roofline": {"bound": "mfma", "achieved": round(tf_o, 2), "peak": peak_o, "unit": "TFLOP/s", "frac": round(tf_o / peak_o, 4),
                                    "scope": "whole forward" + (" on the fp32-input matrix cores (v_mfma_f32_32x32x2_f32: 157 TFLOP/s nominal)" if other == "f32" else "")}}
    del model_o
    # length sensitivity: the same batch of 32 at 128 and 566 frames, in the headline precision
    lengths = {}
    for frames_l in (128, 566):
        mel_l = synthetic_mel(args.batch, frames_l, 1234 + rank).to(dev)
        el = timed_region(lambda: gen(mel_l), 20, 5, lambda: torch.cuda.synchronize(dev), barrier, max_reduce)
        lengths[str(frames_l)] = {"value": round(world * args.batch * frames_l * gen.hop * 20 / el, 1), "unit": "samples/s", "ms_per_step": round(el / 20 * 1e3, 3),
                                  "steps": 20, "warmup": 5}
        del mel_l

    return other_precision, lengths


def main(argv=None) -> int:
    argv = list(sys.argv[1:] if argv is None else argv)
    args = parse_args(argv)
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        return launch_ranks(args.gpus, argv)  # before anything touches the GPU
    rank, local_rank, world = dist_env()
    if world != args.gpus:
        print(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks", file=sys.stderr)
        return 2
    if args.selftest_cpu:
        return selftest_cpu(args, rank, world)
    t_start = time.perf_counter()
    import torch

    if not torch.cuda.is_available():
        print("bench.py: no GPU visible; the product path has no CPU fallback", file=sys.stderr)
        return 2
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)

    use_dist = world > 1
    if use_dist:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)

        def barrier():
            dist.barrier(device_ids=[local_rank])

        def max_reduce(x):
            t = torch.tensor([x], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            return float(t.item())

        ones = torch.ones(1, device=dev)
        dist.all_reduce(ones)  # the rank count RCCL itself sees
        rccl_ranks = int(ones.item())
    else:
        barrier = lambda: None  # noqa: E731
        max_reduce = lambda x: x  # noqa: E731
        rccl_ranks = 0

    model = upstream_init_generator(args.precision).to(dev).eval()
    mel = synthetic_mel(args.batch, args.frames, 1234 + rank).to(dev)
    gen = model.generator
    samples_per_step = args.batch * args.frames * gen.hop
    out = {}

    def step():
        out["wav"] = gen(mel)

    elapsed = timed_region(step, args.steps, args.warmup, lambda: torch.cuda.synchronize(dev), barrier, max_reduce)
    value = world * samples_per_step * args.steps / elapsed
    # the roofline block's per-kernel events: directly behind the timed region, before any other leg touches clocks or the allocator
    roof = measured_roofline(gen, mel, elapsed / args.steps * 1e3, args.profile_passes)
    # the other arithmetic beside it (the reference computes in fp32; bf16 operands with fp32 accumulation are SURVEY 8(d) C2's contract)
    other = "f32" if args.precision == "bf16" else "bf16"
    other_precision, lengths = None, {}

    leg_seconds = {"headline": round(time.perf_counter() - t_start, 1)}

    def guarded(name, fn, *a):
        """The secondary legs never take the headline line down with them: a failure is reported in the leg's object (and on stderr)."""
        t0 = time.perf_counter()
        try:
            return fn(*a)
        except Exception as e:  # noqa: BLE001
            import traceback

            traceback.print_exc(file=sys.stderr)
            return {"error": f"{name}: {type(e).__name__}: {e}"}
        finally:
            leg_seconds[name] = round(time.perf_counter() - t0, 1)

    if not args.no_side_legs:
        side = guarded("side_legs", _side_legs, args, other, gen, mel, samples_per_step, dev, rank, world, barrier, max_reduce)
        other_precision, lengths = (side, {}) if isinstance(side, dict) else side
    train = None
    if not args.no_train:
        train = guarded("train_leg", train_leg, args, dev, rank, world, use_dist, barrier, max_reduce)
    fs2 = None if args.no_fs2 else guarded("fs2_leg", fs2_leg, args, dev, rank, world, barrier, max_reduce)
    fs2_train = None if args.no_fs2 else guarded("fs2_train_leg", fs2_train_leg, args, dev, rank, world, use_dist, barrier, max_reduce)

    result = None
    if rank == 0:
        pmc = recorded_pmc_traffic(roof["kernel"])
        if pmc:
            roof["traffic"] = pmc[0]
            roof["traffic_source"] = f"profiles/{pmc[1]} (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes, bytes per launch)"
            roof["traffic_commit"] = profile_commit(ROOT / "profiles" / pmc[1])
            roof["traffic_code"] = profile_commit(ROOT / "profiles" / pmc[1], "code")
            roof["algorithmic_bytes_per_launch"] = round(roof["algorithmic_hbm_gbs"] * 1e9 * roof["avg_launch_ms"] * 1e-3)
        flops_per_sample = 2.0 * gen.macs_per_sample()
        result = {
            "metric": METRIC,
            "value": round(value, 1),
            "unit": "samples/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": args.precision,
            "data": "synthetic",
            "config": {
                "workload": f"HiFiGAN-V1 generator inference, mel[{args.batch},80,{args.frames}] -> wav[{args.batch},1,{args.frames * gen.hop}] "
                            f"per GPU, {world} replica(s), upstream N(0,0.01) init, weight norm folded",
                "global_batch": args.batch * world,
                "frames": args.frames,
                "samples_per_step_per_gpu": samples_per_step,
                "parallelism": f"replicas x{world}",
                "flop_per_sample": flops_per_sample,
            },
            "whole_job_tflops": round(value * flops_per_sample / 1e12, 2),
            "realtime_factor": round(value / 22050.0, 1),
            "other_precision": other_precision,
            "length_sensitivity": lengths,
            "roofline": roof,
            "commit": git_head(),
            "code": code_fingerprint(),
        }
        if use_dist:
            result["rccl_ranks"] = rccl_ranks
        if train is not None:
            result["train"] = train
        if fs2 is not None:
            result["fs2"] = fs2
            result["fs2_train"] = fs2_train
        if world == 1 and not args.no_cpu_baseline:
            result["cpu_baseline"] = guarded("cpu_baseline", cpu_baseline, args.cpu_frames, args.cpu_batch)
            cores = result["cpu_baseline"].get("cores", 16)
            if "value" in result["cpu_baseline"]:
                result["gpu_over_cpu"] = round(value / result["cpu_baseline"]["value"], 1)
            if train is not None and "error" not in train:
                train["cpu_baseline"] = guarded("cpu_baseline_train", cpu_baseline_train, cores)
            if fs2 is not None and "error" not in fs2:
                fs2["cpu_baseline"] = guarded("cpu_baseline_fs2", cpu_baseline_fs2, cores)
            if fs2_train is not None and "error" not in fs2_train:
                fs2_train["cpu_baseline"] = guarded("cpu_baseline_fs2_train", cpu_baseline_fs2_train, cores)
    if use_dist:
        import torch.distributed as dist

        dist.barrier(device_ids=[local_rank])
        dist.destroy_process_group()
    if rank == 0:
        result["leg_seconds"] = leg_seconds  # wall time of every leg of this process (import and set-up included in "headline")
        print(json.dumps(result))
    return 0


if __name__ == "__main__":
    sys.exit(main())
