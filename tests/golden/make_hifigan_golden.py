#!/usr/bin/env python3
"""Regression vectors for the vocoder from the ORACLE (oracle/hifigan_ref.py), not from the
reference: the reference's model code is an un-vendored submodule, so no reference output exists
("parity unpinned").  Weights are re-drawn from a fixed seed at test time
(tests/helpers.py), only the input mel and the expected waveform are stored."""
import sys
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))
from helpers import make_ref_generator, synthetic_mel  # noqa: E402

torch.set_num_threads(4)
gen = make_ref_generator(seed=1234)
mel = synthetic_mel(2, 6, seed=99)
with torch.no_grad():
    wav = gen(mel)
np.savez_compressed(Path(__file__).parent / "hifigan_v1_small.npz", mel=mel.numpy(), wav=wav.numpy(),
                    seed=np.int64(1234), torch_version=np.array(torch.__version__))
print("wav", wav.shape, "abs max", float(wav.abs().max()), "std", float(wav.std()))
