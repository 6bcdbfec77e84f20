"""Shared helpers for the parity tests (oracle side only)."""

from __future__ import annotations

import math

import torch

from oracle.hifigan_ref import GeneratorRef, HiFiGANModelConfigRef


def signal_preserving_init_(gen: GeneratorRef, seed: int) -> GeneratorRef:
    """Re-draw every (already folded) weight so activations stay O(1) through the network — the
    upstream N(0, 0.01) init makes the output collapse to tanh(bias), which would hide errors."""
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for name, p in gen.named_parameters():
            if name.endswith("bias"):
                p.copy_(torch.randn(p.shape, generator=g) * 0.05)
                continue
            if name.startswith("ups."):
                fan_in = p.shape[0] * p.shape[2] / gen.cfg.upsample_rates[int(name.split(".")[1])]
            else:
                fan_in = p.shape[1] * p.shape[2]
            gain = 0.6 if ".convs2." in name else 1.0
            p.copy_(torch.randn(p.shape, generator=g) * gain / math.sqrt(fan_in))
    return gen


def make_ref_generator(cfg: HiFiGANModelConfigRef | None = None, seed: int = 1234) -> GeneratorRef:
    torch.manual_seed(seed)
    gen = GeneratorRef(cfg).eval()
    gen.remove_weight_norm()
    return signal_preserving_init_(gen, seed)


def synthetic_mel(B: int, T: int, n_mels: int = 80, seed: int = 1234) -> torch.Tensor:
    """SURVEY.md §8d C2 input: clamp(N(-5, 2^2), -11.5129, 2.0)."""
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(B, n_mels, T, generator=g) * 2.0 - 5.0).clamp(-11.5129, 2.0)


def rel_l2(a: torch.Tensor, b: torch.Tensor) -> float:
    return float((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-30))


_CHILD_RESULTS: dict = {}


def child_pytest_results(group: str, jobs: dict, parallel: int = 4, timeout: int = 1500) -> dict:
    """Run the child ``pytest`` processes of one test family side by side and cache the outcome per family.

    ``jobs``: key -> (pytest arguments, extra environment).  The kernels' A/B switches are read once per process, so every switch /
    forced tile needs a process of its own; run one after the other those children were 40 % of the GPU suite's wall time (start-up
    and ``import torch`` mostly), although each uses a sliver of the GPU.  The first test of a family that asks starts ALL of the
    family's children, ``parallel`` at a time; every parametrised test then asserts on its own child's result."""
    import os
    import subprocess
    import sys
    import tempfile
    import time

    if group in _CHILD_RESULTS:
        return _CHILD_RESULTS[group]
    # the children compute torch references on the CPU: side by side each gets its share of the cores (N processes x all cores
    # oversubscribes the OpenMP pools -- the first version of this helper took > 20 minutes for what runs in 5 one after the other)
    threads = str(max(1, min(32, (os.cpu_count() or 8) // max(1, parallel))))
    pending = list(jobs.items())
    running: list = []
    results: dict = {}
    while pending or running:
        while pending and len(running) < parallel:
            key, (args, env) = pending.pop(0)
            # output goes to a file, read after exit: a child that prints more than a pipe holds (several long tracebacks) would
            # otherwise block on write until the timeout kills it, and the real failure would read as rc -9
            log = tempfile.TemporaryFile(mode="w+", errors="replace")
            proc = subprocess.Popen([sys.executable, "-m", "pytest", *args], env=dict(os.environ, OMP_NUM_THREADS=threads, MKL_NUM_THREADS=threads, **env),
                                    stdout=log, stderr=subprocess.STDOUT, text=True)
            running.append((key, proc, time.time(), log))
        for item in list(running):
            key, proc, t0, log = item
            if proc.poll() is None:
                if time.time() - t0 > timeout:
                    proc.kill()
                    proc.wait()
                else:
                    continue
            log.seek(0)
            out = log.read()
            log.close()
            results[key] = (proc.returncode, out[-4000:])
            running.remove(item)
        time.sleep(0.2)
    _CHILD_RESULTS[group] = results
    return results


def fuzz_gan_schedule(seed: int, setattr_fn=setattr):
    """Schedule fuzzing of the GAN step (tools/ddp_repeat.py jobs "graphF" / "eagerF"; tests/test_gpu_train_step.py).  Every chain that is forked onto a side stream -- the discriminators' branches, the
    spectral-norm preparation, the fragment stream, the gradient exchange's side streams -- starts behind a busy-wait kernel of a random
    length (0 - 3 ms, seeded), so the streams of a step finish in orders a quiet GPU never produces.  A consumer that reads a tensor
    without an edge from its producer's stream then reads it EARLY, every time, instead of once in fifty runs under contention; with
    every edge in place the step's results cannot change (same kernels, same operands) and the checksums equal the plain job's.
    Captured steps hold the busy-wait kernels as graph nodes: replays are perturbed the same way.
    ``setattr_fn``: pytest's ``monkeypatch.setattr`` in a test (undone at its end)."""
    import random

    import torch

    from everyvoice_amd.train import hifigan

    rng = random.Random(seed)

    def nap():
        ms = rng.choice([0.0, 0.0, 0.3, 1.0, 3.0])
        if ms:
            torch.cuda._sleep(int(ms * 2.0e6))  # (cycles of the ~2 GHz shader clock)

    real_run = hifigan.Branches.run_indexed

    def run_indexed(self, items):
        def wrap(fn):
            def go():
                nap()
                fn()
            return go
        return real_run(self, [(j, wrap(fn)) for j, fn in items])

    setattr_fn(hifigan.Branches, "run_indexed", run_indexed)
    real_prepare = hifigan.HiFiGANTrainer._prepare_chain_fragments

    def prepare(self, probe, generator_step, which="all"):
        nap()
        return real_prepare(self, probe, generator_step, which)

    setattr_fn(hifigan.HiFiGANTrainer, "_prepare_chain_fragments", prepare)
    real_launch = hifigan.BucketReducer.launch

    def launch(self, lo, hi):
        if self.stream is not None and hi > lo:
            with torch.cuda.stream(self.stream):
                nap()
        return real_launch(self, lo, hi)

    setattr_fn(hifigan.BucketReducer, "launch", launch)
