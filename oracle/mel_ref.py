"""fp32 CPU restatement of the reference's default mel front-end ("mel-librosa").  TEST INFRASTRUCTURE.

Follows everyvoice/utils/heavy.py:69-100 (Spectrogram n_fft/win/hop, hann, power 2, centred,
reflect pad, one-sided -> sqrt(spec + 1e-9) -> librosa Slaney mel basis @ ) and
everyvoice/utils/heavy.py:39-40 (log(clamp(., 1e-5))), with the frame truncation of
everyvoice/preprocessor/preprocessor.py:870-929 ([:, :S // hop]).

torchaudio and librosa are not installed in this image, so the two third-party pieces are
restated from their published algorithms:
  * torchaudio.transforms.Spectrogram(power=2, center, reflect, onesided, normalized=False)
    == |torch.stft(...)|^2 with a periodic hann window;
  * librosa.filters.mel(sr, n_fft, n_mels, fmin, fmax) with htk=False, norm="slaney":
    Slaney mel scale (linear below 1 kHz, log above, 3 mel = 200 Hz), triangular filters from
    n_mels + 2 equally spaced mel points, each scaled by 2 / (f[i+2] - f[i]); float32 output.

Anchor (sanity, not a bit pin): the ming024 mel of LJ010-0008.wav that the reference's test
data holds (everyvoice/tests/data/ming024/eng-LJSpeech-mel-LJ010-0008.npy) — checked in
tests/test_oracle_golden.py on interior frames.
"""

from __future__ import annotations

import numpy as np
import torch


def _hz_to_mel_slaney(f):
    f = np.asanyarray(f, dtype=np.float64)
    f_sp = 200.0 / 3
    mels = f / f_sp
    min_log_hz = 1000.0
    min_log_mel = min_log_hz / f_sp
    logstep = np.log(6.4) / 27.0
    return np.where(f >= min_log_hz, min_log_mel + np.log(np.maximum(f, 1e-30) / min_log_hz) / logstep, mels)


def _mel_to_hz_slaney(m):
    m = np.asanyarray(m, dtype=np.float64)
    f_sp = 200.0 / 3
    min_log_hz = 1000.0
    min_log_mel = min_log_hz / f_sp
    logstep = np.log(6.4) / 27.0
    return np.where(m >= min_log_mel, min_log_hz * np.exp(logstep * (m - min_log_mel)), f_sp * m)


def slaney_mel_basis(sr: int, n_fft: int, n_mels: int = 80, fmin: float = 0.0, fmax: float | None = 8000.0) -> np.ndarray:
    """[n_mels, n_fft//2 + 1] float32 filterbank (librosa.filters.mel defaults)."""
    if fmax is None:
        fmax = sr / 2.0
    fftfreqs = np.linspace(0.0, sr / 2.0, n_fft // 2 + 1)
    mel_pts = np.linspace(_hz_to_mel_slaney(fmin), _hz_to_mel_slaney(fmax), n_mels + 2)
    hz_pts = _mel_to_hz_slaney(mel_pts)
    fdiff = np.diff(hz_pts)
    ramps = hz_pts[:, None] - fftfreqs[None, :]
    weights = np.zeros((n_mels, n_fft // 2 + 1), dtype=np.float64)
    for i in range(n_mels):
        lower = -ramps[i] / fdiff[i]
        upper = ramps[i + 2] / fdiff[i + 1]
        weights[i] = np.maximum(0.0, np.minimum(lower, upper))
    enorm = 2.0 / (hz_pts[2 : n_mels + 2] - hz_pts[:n_mels])
    weights *= enorm[:, None]
    return weights.astype(np.float32)


def magnitude_spectrogram_ref(audio: torch.Tensor, n_fft=1024, win=1024, hop=256) -> torch.Tensor:
    """sqrt(|STFT|^2 + 1e-9): [..., S] -> [..., n_fft//2+1, 1 + S//hop]."""
    window = torch.hann_window(win, dtype=audio.dtype)
    spec = torch.stft(
        audio, n_fft, hop_length=hop, win_length=win, window=window, center=True,
        pad_mode="reflect", normalized=False, onesided=True, return_complex=True,
    )
    power = spec.real**2 + spec.imag**2
    return torch.sqrt(power + 1e-9)


def mel_spectrogram_ref(audio, sr=22050, n_fft=1024, win=1024, hop=256, n_mels=80, fmin=0, fmax=8000,
                        log=True, truncate=False) -> torch.Tensor:
    """audio [..., S] fp32 -> (log-)mel [..., n_mels, frames]; ``truncate`` keeps S // hop frames
    as Preprocessor.process_spec does."""
    audio = torch.as_tensor(audio, dtype=torch.float32)
    mag = magnitude_spectrogram_ref(audio, n_fft, win, hop)
    basis = torch.from_numpy(slaney_mel_basis(sr, n_fft, n_mels, fmin, fmax))
    mel = torch.matmul(basis, mag)
    if log:
        mel = torch.log(torch.clamp(mel, min=1e-5))
    if truncate:
        mel = mel[..., : audio.shape[-1] // hop]
    return mel


# ---- the other branches of get_spectral_transform (everyvoice/utils/heavy.py:59-68, 101-118) -------------------------------------------
# torchaudio is not installed here; its transforms are thin wrappers whose arithmetic is torch.stft / torch.istft (both present) and
# torchaudio.functional.melscale_fbanks (restated below from its published algorithm: parity of the filterbank is unpinned).


def spectrogram_ref(audio: torch.Tensor, n_fft: int, win: int, hop: int, power=2.0) -> torch.Tensor:
    """torchaudio.transforms.Spectrogram(n_fft, win_length, hop_length, power): torch.stft(hann, centred, reflect, one-sided), then
    |.|^power (power None: the complex tensor)."""
    audio = torch.as_tensor(audio, dtype=torch.float32)
    lead = audio.shape[:-1]  # (torchaudio packs the leading dimensions into one batch axis and unpacks them afterwards)
    spec = torch.stft(audio.reshape(-1, audio.shape[-1]), n_fft, hop_length=hop, win_length=win, window=torch.hann_window(win), center=True,
                      pad_mode="reflect", normalized=False, onesided=True, return_complex=True)
    spec = spec.reshape(lead + spec.shape[-2:])
    if power is None:
        return spec
    return spec.abs() if power == 1 else spec.abs().pow(power)


def htk_slaney_fbanks(sr: int, n_freqs: int, n_mels: int, fmin: float, fmax: float) -> np.ndarray:
    """torchaudio.functional.melscale_fbanks(n_freqs, fmin, fmax, n_mels, sr, norm="slaney", mel_scale="htk") -> [n_freqs, n_mels]."""
    all_freqs = np.linspace(0, sr // 2, n_freqs)
    m_min = 2595.0 * np.log10(1.0 + fmin / 700.0)
    m_max = 2595.0 * np.log10(1.0 + fmax / 700.0)
    m_pts = np.linspace(m_min, m_max, n_mels + 2)
    f_pts = 700.0 * (10.0 ** (m_pts / 2595.0) - 1.0)
    fb = np.zeros((n_freqs, n_mels))
    for m in range(n_mels):
        lo, mid, hi = f_pts[m], f_pts[m + 1], f_pts[m + 2]
        for i, f in enumerate(all_freqs):
            down = (f - lo) / (mid - lo)
            up = (hi - f) / (hi - mid)
            fb[i, m] = max(0.0, min(down, up)) * 2.0 / (hi - lo)
    return fb.astype(np.float32)


def torchaudio_mel_ref(audio, sr, n_fft, win, hop, n_mels, fmin=0.0, fmax=None) -> torch.Tensor:
    """torchaudio.transforms.MelSpectrogram(..., norm="slaney", center=True): fb^T @ |STFT|^2."""
    fmax = float(sr // 2) if fmax is None else fmax
    fb = torch.from_numpy(htk_slaney_fbanks(sr, n_fft // 2 + 1, n_mels, fmin, fmax))
    spec = spectrogram_ref(audio, n_fft, win, hop, 2.0)
    return torch.matmul(spec.transpose(-1, -2), fb).transpose(-1, -2)


def inverse_spectrogram_ref(spec: torch.Tensor, n_fft: int, win: int, hop: int) -> torch.Tensor:
    """torchaudio.transforms.InverseSpectrogram(n_fft, win_length, hop_length): torch.istft(hann, centred, one-sided)."""
    return torch.istft(spec, n_fft, hop_length=hop, win_length=win, window=torch.hann_window(win), center=True, normalized=False,
                       onesided=True)
