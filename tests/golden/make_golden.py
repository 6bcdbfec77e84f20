#!/usr/bin/env python3
"""Generate golden vectors by IMPORTING THE REFERENCE (build container only).

Run:  python tests/golden/make_golden.py      (needs /root/reference mounted)

The reference is Python, so its own functions are executed here and only DATA (inputs and
expected outputs) is committed, as ``tests/golden/*.npz``.  torchaudio and email-validator are
not installed in this image; ``everyvoice.utils.heavy`` imports torchaudio at module top without
using it in the functions captured here, so an empty stand-in module is injected into
``sys.modules`` for the duration of this script only (nothing is written to disk).

Captured (reference file:line):
  expand                          everyvoice/utils/heavy.py:12-21           -> expand.npz
  collate_fn                      everyvoice/utils/heavy.py:24-36           -> collate.npz
  dynamic_range_(de)compression   everyvoice/utils/heavy.py:39-44           -> drc.npz
  get_segments (explicit start)   everyvoice/utils/heavy.py:122-148         -> segments.npz
  BetaBinomialInterpolator        everyvoice/preprocessor/attention_prior.py:34-67 -> attn_prior.npz
  create_depthwise_separable_convolution  everyvoice/model/utils.py:5-48    -> dwsep.npz
  original_hifigan_leaky_relu     everyvoice/utils/__init__.py:178-181      -> lrelu.npz
Also copied as DATA from the reference's test fixtures:
  the five LJ duration tensors    everyvoice/tests/data/lj/preprocessed/duration/*.pt (inside expand.npz)
  LJ010-0008.wav samples + the ming024 mel array of the same utterance      -> mel_anchor.npz
"""

from __future__ import annotations

import importlib.util
import sys
import types
import wave
from pathlib import Path

import numpy as np
import torch

REF = Path("/root/reference")
OUT = Path(__file__).resolve().parent


def _inject_stubs():
    ta = types.ModuleType("torchaudio")
    ta.transforms = types.ModuleType("torchaudio.transforms")
    sys.modules.setdefault("torchaudio", ta)
    sys.modules.setdefault("torchaudio.transforms", ta.transforms)
    try:
        import email_validator  # noqa: F401
    except ImportError:
        ev = types.ModuleType("email_validator")

        class EmailNotValidError(ValueError):
            pass

        def validate_email(email, *a, **k):
            return types.SimpleNamespace(normalized=email, email=email, local_part=email.split("@")[0])

        ev.EmailNotValidError = EmailNotValidError
        ev.validate_email = validate_email
        sys.modules["email_validator"] = ev
        import importlib.metadata as md

        _orig = md.version

        def _version(name):
            if name.replace("_", "-") == "email-validator":
                return "2.1.0"
            return _orig(name)

        md.version = _version


def main():
    assert REF.exists(), "run in the build container: /root/reference is not mounted"
    _inject_stubs()
    sys.path.insert(0, str(REF))
    from everyvoice.utils import original_hifigan_leaky_relu
    from everyvoice.utils.heavy import (
        collate_fn,
        dynamic_range_compression_torch,
        dynamic_range_decompression_torch,
        expand,
        get_segments,
    )

    g = torch.Generator().manual_seed(1234)

    # ---- expand on the five real LJ duration tensors + edge cases ---------------------------
    store = {}
    dur_dir = REF / "everyvoice/tests/data/lj/preprocessed/duration"
    for i, p in enumerate(sorted(dur_dir.glob("*.pt"))):
        d = torch.load(p, weights_only=True)
        v = torch.randn(d.numel(), 256, generator=g)
        store[f"lj{i}_dur"] = d.numpy().astype(np.int64)
        store[f"lj{i}_val"] = v.numpy()
        store[f"lj{i}_out"] = expand(v, d).numpy()
    edge_d = torch.tensor([2, 0, 3, -1, 1, 0, 4])
    edge_v = torch.randn(7, 8, generator=g)
    store["edge_dur"], store["edge_val"] = edge_d.numpy().astype(np.int64), edge_v.numpy()
    store["edge_out"] = expand(edge_v, edge_d).numpy()
    frac_d = torch.tensor([1.9, 0.4, 2.0, 3.7, -0.5, 1.0])
    frac_v = torch.randn(6, 4, generator=g)
    store["frac_dur"], store["frac_val"] = frac_d.numpy(), frac_v.numpy()
    store["frac_out"] = expand(frac_v, frac_d).numpy()
    store["list_out"] = expand([10, 20, 30], [2, 0, 3])
    store["np1d_out"] = expand(np.arange(5, dtype=np.float32), np.array([1, 2, 0, 1, 3]))
    np.savez_compressed(OUT / "expand.npz", **store)

    # ---- collate_fn on a seeded ragged batch -------------------------------------------------
    lens = [7, 3, 11, 5]
    batch = []
    for i, n in enumerate(lens):
        batch.append(
            {
                "mel": torch.randn(n * 4, 80, generator=g),
                "text": torch.randint(2, 80, (n,), generator=g),
                "nested": {"pitch": torch.randn(n, generator=g).numpy(), "speaker_id": i},
                "basename": f"utt{i}",
            }
        )
    coll = collate_fn(batch)
    store = {}
    for i, b in enumerate(batch):
        store[f"in{i}_mel"] = b["mel"].numpy()
        store[f"in{i}_text"] = b["text"].numpy()
        store[f"in{i}_pitch"] = b["nested"]["pitch"]
        store[f"in{i}_speaker_id"] = np.int64(b["nested"]["speaker_id"])
    store["out_keys"] = np.array(sorted(coll.keys()))
    store["out_mel"] = coll["mel"].numpy()
    store["out_text"] = coll["text"].numpy()
    store["out_nested_pitch"] = coll["nested_pitch"].numpy()
    store["out_nested_speaker_id"] = coll["nested_speaker_id"].numpy()
    assert coll["nested_speaker_id"].dtype == torch.int32
    np.savez_compressed(OUT / "collate.npz", **store)

    # ---- dynamic range (de)compression on edge values ----------------------------------------
    x = torch.tensor([0.0, 1e-6, 1e-5, 9.999e-6, 1.0, 2.0, 1e3, -1.0, 3.3e-4], dtype=torch.float32)
    xr = torch.rand(64, generator=g) * 5
    x = torch.cat([x, xr])
    np.savez_compressed(
        OUT / "drc.npz",
        x=x.numpy(),
        drc=dynamic_range_compression_torch(x).numpy(),
        drd=dynamic_range_decompression_torch(dynamic_range_compression_torch(x)).numpy(),
    )

    # ---- get_segments with explicit starts ---------------------------------------------------
    mel = torch.randn(80, 100, generator=g)
    wav = torch.randn(1, 100 * 256, generator=g)
    s_mel, st = get_segments(mel, 32, start=17)
    s_wav, st2 = get_segments(wav, 8192, start=17 * 256)
    short = torch.randn(80, 20, generator=g)
    s_short, st3 = get_segments(short, 32)
    exact = torch.randn(80, 33, generator=g)
    s_exact, st4 = get_segments(exact, 32, start=0)
    np.savez_compressed(
        OUT / "segments.npz",
        mel=mel.numpy(), wav=wav.numpy(), short=short.numpy(), exact=exact.numpy(),
        seg_mel=s_mel.numpy(), seg_wav=s_wav.numpy(), seg_short=s_short.numpy(), seg_exact=s_exact.numpy(),
        starts=np.array([st, st2, st3, st4]),
    )

    # ---- attention prior ---------------------------------------------------------------------
    spec = importlib.util.spec_from_file_location("ap", REF / "everyvoice/preprocessor/attention_prior.py")
    ap = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ap)
    interp = ap.BetaBinomialInterpolator()
    np.savez_compressed(
        OUT / "attn_prior.npz",
        shapes=np.array([[443, 60], [150, 33], [32, 7], [101, 19]]),
        p0=interp(443, 60), p1=interp(150, 33), p2=interp(32, 7), p3=interp(101, 19),
    )

    # ---- depthwise-separable conv factory: structure + seeded forward -----------------------
    from everyvoice.model.utils import create_depthwise_separable_convolution

    # NB: as written at everyvoice/model/utils.py:33-43 the non-transposed branch passes
    # ``output_padding`` to Conv1d and raises TypeError on every torch release; only the
    # transposed branch is callable, so that is what is captured (plus the error text).
    try:
        create_depthwise_separable_convolution(16, 24, 3, padding=1)
        conv_branch_error = ""
    except TypeError as e:
        conv_branch_error = str(e)
    torch.manual_seed(1234)
    seq = create_depthwise_separable_convolution(16, 24, 3, padding=1, transpose=True)
    xin = torch.randn(2, 16, 21, generator=g)
    with torch.no_grad():
        yout = seq(xin)
    sd = {k.replace(".", "__"): v.detach().numpy() for k, v in seq.state_dict().items()}
    np.savez_compressed(OUT / "dwsep.npz", x=xin.numpy(), y=yout.numpy(),
                        conv_branch_error=np.array(conv_branch_error),
                        names=np.array(sorted(seq.state_dict().keys())), **sd)

    # ---- hifigan activation ------------------------------------------------------------------
    xa = torch.randn(257, generator=g) * 3
    np.savez_compressed(OUT / "lrelu.npz", x=xa.numpy(), y=original_hifigan_leaky_relu(xa).numpy())

    # ---- mel anchor: reference test wav + the ming024 mel the reference's test data holds -----
    with wave.open(str(REF / "everyvoice/tests/data/LJ010-0008.wav"), "rb") as w:
        assert w.getframerate() == 22050 and w.getnchannels() == 1 and w.getsampwidth() == 2
        pcm = np.frombuffer(w.readframes(w.getnframes()), dtype=np.int16)
    mel_ming = np.load(REF / "everyvoice/tests/data/ming024/eng-LJSpeech-mel-LJ010-0008.npy")
    np.savez_compressed(OUT / "mel_anchor.npz", pcm=pcm, mel_ming024=mel_ming.astype(np.float32))
    print("golden vectors written to", OUT)
    for p in sorted(OUT.glob("*.npz")):
        print(f"  {p.name:20s} {p.stat().st_size/1024:8.1f} KiB")


if __name__ == "__main__":
    main()
