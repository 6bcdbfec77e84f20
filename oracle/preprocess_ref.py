"""TEST INFRASTRUCTURE (oracle): CPU restatements of the audio steps of the reference's preprocessor whose arithmetic lives in
torchaudio (absent from this image -> **parity unpinned**: the published algorithms are restated):

  resample_ref   torchaudio.functional.resample(waveform, orig, new) with its defaults (sinc_interp_hann, lowpass_filter_width 6,
                 rolloff 0.99) -- called at everyvoice/preprocessor/preprocessor.py:196-198
  loudness_ref   torchaudio.transforms.Loudness(sr) = torchaudio.functional.loudness (ITU-R BS.1770-4) -- the "audio_empty" gate
                 at everyvoice/preprocessor/preprocessor.py:177-185
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline legs may import this module.
"""

from __future__ import annotations

import math

import numpy as np
import torch
import torch.nn.functional as F


def sinc_kernel_ref(orig_freq: int, new_freq: int, lowpass_filter_width: int = 6, rolloff: float = 0.99):
    g = math.gcd(int(orig_freq), int(new_freq))
    orig, new = int(orig_freq) // g, int(new_freq) // g
    base_freq = min(orig, new) * rolloff
    width = math.ceil(lowpass_filter_width * orig / base_freq)
    idx = torch.arange(-width, width + orig, dtype=torch.float64)[None, None] / orig
    t = torch.arange(0, -new, -1, dtype=torch.float64)[:, None, None] / new + idx
    t = (t * base_freq).clamp_(-lowpass_filter_width, lowpass_filter_width)
    window = torch.cos(t * math.pi / lowpass_filter_width / 2) ** 2
    t = t * math.pi
    kernels = torch.where(t == 0, torch.tensor(1.0, dtype=torch.float64), t.sin() / t) * window * (base_freq / orig)
    return kernels.to(torch.float32), width, orig, new


def resample_ref(waveform: torch.Tensor, orig_freq: int, new_freq: int) -> torch.Tensor:
    """[..., S] -> [..., ceil(new * S / orig)]"""
    if orig_freq == new_freq:
        return waveform
    kernel, width, orig, new = sinc_kernel_ref(orig_freq, new_freq)
    shape = waveform.shape
    x = waveform.reshape(-1, shape[-1]).to(torch.float32)
    length = x.shape[-1]
    x = F.pad(x, (width, width + orig))
    y = F.conv1d(x[:, None], kernel, stride=orig).transpose(1, 2).reshape(x.shape[0], -1)
    return y[..., : math.ceil(new * length / orig)].reshape(shape[:-1] + (-1,))


def _biquad(x: np.ndarray, b0, b1, b2, a0, a1, a2) -> np.ndarray:
    """torchaudio.functional.biquad -> lfilter(clamp=True), float32 arithmetic, direct form I."""
    b0, b1, b2, a1, a2 = (np.float32(v / a0) for v in (b0, b1, b2, a1, a2))
    y = np.zeros_like(x, dtype=np.float32)
    x1 = x2 = y1 = y2 = np.float32(0)
    for n in range(x.shape[-1]):
        xv = x[n]
        yv = np.float32(b0 * xv + b1 * x1 + b2 * x2 - a1 * y1 - a2 * y2)
        x2, x1, y2, y1 = x1, xv, y1, yv
        y[n] = yv
    return np.clip(y, -1.0, 1.0)


def loudness_ref(waveform: torch.Tensor, sample_rate: int) -> float:
    """[channels, S] -> LKFS (float; NaN when nothing passes the gates)."""
    x = waveform.to(torch.float32).numpy()
    w0 = 2 * math.pi * 1500.0 / sample_rate
    alpha = math.sin(w0) / 2 / (1 / math.sqrt(2))
    A = math.exp(4.0 / 40 * math.log(10))
    t1, t2, t3 = 2 * math.sqrt(A) * alpha, (A - 1) * math.cos(w0), (A + 1) * math.cos(w0)
    s1 = (A * ((A + 1) + t2 + t1), -2 * A * ((A - 1) + t3), A * ((A + 1) + t2 - t1), (A + 1) - t2 + t1, 2 * ((A - 1) - t3), (A + 1) - t2 - t1)
    w0 = 2 * math.pi * 38.0 / sample_rate
    alpha = math.sin(w0) / 2 / 0.5
    s2 = ((1 + math.cos(w0)) / 2, -1 - math.cos(w0), (1 + math.cos(w0)) / 2, 1 + alpha, -2 * math.cos(w0), 1 - alpha)
    y = np.stack([_biquad(_biquad(ch, *s1), *s2) for ch in x])
    gate, bias = int(round(0.4 * sample_rate)), -0.691
    step = int(gate * 0.25)
    if y.shape[-1] < gate:
        return float("nan")
    e = torch.from_numpy(y * y).unfold(-1, gate, step).mean(-1).numpy().astype(np.float64)  # [ch, blocks]
    g = np.array([1.0, 1.0, 1.0, 1.41, 1.41][: e.shape[0]])[:, None]
    with np.errstate(divide="ignore", invalid="ignore"):
        loud = bias + 10 * np.log10((g * e).sum(0))
        gated = loud > -70.0
        ef = (e * gated).sum(1) / gated.sum()
        gamma_rel = bias + 10 * np.log10((g[:, 0] * ef).sum()) - 10.0
        gated = gated & (loud > gamma_rel)
        ef = (e * gated).sum(1) / gated.sum()
        return float(bias + 10 * np.log10((g[:, 0] * ef).sum()))


def pitch_acf_ref(audio: np.ndarray, hop: int, sr: int, f0_floor: float = 71.0, f0_ceil: float = 800.0, threshold: float = 0.8) -> np.ndarray:
    """The product's own F0 estimator restated (NOT pyworld -- see csrc/preprocess_ops.hip: pitch_acf_kernel): normalised
    autocorrelation over a window of two periods of f0_floor centred on each frame, smallest local maximum within 95 % of the
    best, parabolic refinement; 0 where max r < threshold.  [S] -> [S // hop + 1] Hz."""
    x = np.asarray(audio, dtype=np.float32)
    n = len(x)
    lag_lo, lag_hi = int(np.floor(np.float32(sr) / np.float32(f0_ceil))), int(np.ceil(np.float32(sr) / np.float32(f0_floor)))
    win = 2 * lag_hi
    out = np.zeros(n // hop + 1, dtype=np.float32)
    for f in range(len(out)):
        start = f * hop - win // 2
        idx = np.arange(start, start + win + lag_hi + 1)
        seg = np.where((idx >= 0) & (idx < n), x[np.clip(idx, 0, n - 1)], 0.0).astype(np.float64)
        a = seg[:win]
        e0 = float(a @ a)
        lags = np.arange(lag_lo - 1, lag_hi + 2)
        r = np.zeros(len(lags))
        for li, lag in enumerate(lags):
            b = seg[lag : lag + win]
            den = np.sqrt(e0 * float(b @ b))
            r[li] = float(a @ b) / den if den > 1e-12 else 0.0
        rmax = r[1:-1].max()
        if rmax < threshold:
            continue
        for li in range(1, len(r) - 1):
            if r[li] >= 0.95 * rmax and r[li] >= r[li - 1] and r[li] >= r[li + 1]:
                den = r[li - 1] - 2 * r[li] + r[li + 1]
                off = 0.5 * (r[li - 1] - r[li + 1]) / den if abs(den) > 1e-12 else 0.0
                out[f] = sr / (lags[li] + min(max(off, -0.5), 0.5))
                break
    return out
