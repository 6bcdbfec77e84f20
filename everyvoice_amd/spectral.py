"""GPU mirror of the reference's default spectral transform ("mel-librosa") and of the preprocessor's
feature extraction built on it:

  get_spectral_transform(...)            everyvoice/utils/heavy.py:47-119   ("mel-librosa" and "linear-magnitude")
  extract_spectral_features(audio, ...)  everyvoice/preprocessor/preprocessor.py:220-233 (+ frame truncation 870-929)
  extract_energy(logmel)                 everyvoice/preprocessor/preprocessor.py:302-309

The DFT basis (window folded in) and the librosa Slaney mel filterbank are host-built constants uploaded once;
all per-sample arithmetic runs in libevmi_hip (evmi_mel_spectrogram_f32).  CUDA tensors only.
"""

from __future__ import annotations

import numpy as np
import torch

from . import _lib


def slaney_mel_filterbank(sr: int, n_fft: int, n_mels: int = 80, fmin: float = 0.0, fmax: float | None = 8000.0) -> np.ndarray:
    """librosa.filters.mel(sr, n_fft, n_mels, fmin, fmax) defaults (htk=False, norm="slaney"), float32
    [n_mels, n_fft//2 + 1]: Slaney mel scale (linear to 1 kHz, log above), triangles scaled by 2 / bandwidth."""
    fmax = sr / 2.0 if fmax is None else fmax
    f_sp, min_log_hz, logstep = 200.0 / 3, 1000.0, np.log(6.4) / 27.0
    min_log_mel = min_log_hz / f_sp

    def hz_to_mel(f):
        f = np.asarray(f, dtype=np.float64)
        return np.where(f >= min_log_hz, min_log_mel + np.log(np.maximum(f, 1e-30) / min_log_hz) / logstep, f / f_sp)

    def mel_to_hz(m):
        m = np.asarray(m, dtype=np.float64)
        return np.where(m >= min_log_mel, min_log_hz * np.exp(logstep * (m - min_log_mel)), f_sp * m)

    freqs = np.linspace(0.0, sr / 2.0, n_fft // 2 + 1)
    pts = mel_to_hz(np.linspace(hz_to_mel(fmin), hz_to_mel(fmax), n_mels + 2))
    width = np.diff(pts)
    ramps = pts[:, None] - freqs[None, :]
    lower = -ramps[:-2] / width[:-1, None]
    upper = ramps[2:] / width[1:, None]
    fb = np.maximum(0.0, np.minimum(lower, upper)) * (2.0 / (pts[2:] - pts[:-2]))[:, None]
    return fb.astype(np.float32)


def windowed_dft_basis(n_fft: int, win_length: int | None = None) -> tuple[np.ndarray, int]:
    """[n_fft, 2 * n_bins_padded] float32: interleaved (w cos, -w sin) columns, periodic hann window of
    win_length centred in n_fft (torch.stft's convention); returns (basis, n_bins_padded)."""
    win_length = win_length or n_fft
    k = np.arange(n_fft, dtype=np.float64)
    w = np.zeros(n_fft)
    left = (n_fft - win_length) // 2
    w[left : left + win_length] = 0.5 - 0.5 * np.cos(2.0 * np.pi * np.arange(win_length) / win_length)
    n_bins = n_fft // 2 + 1
    nb_pad = (n_bins + 15) // 16 * 16
    basis = np.zeros((n_fft, 2 * nb_pad), dtype=np.float64)
    b = np.arange(n_bins, dtype=np.float64)
    ang = 2.0 * np.pi * ((k[:, None] * b[None, :]) % n_fft) / n_fft  # reduce the angle before cos/sin
    basis[:, 0 : 2 * n_bins : 2] = w[:, None] * np.cos(ang)
    basis[:, 1 : 2 * n_bins : 2] = -w[:, None] * np.sin(ang)
    return basis.astype(np.float32), nb_pad


class MelSpectrogram:
    """Callable mirror of the "mel-librosa" transform; ``log=True`` adds dynamic_range_compression_torch."""

    def __init__(self, n_fft=1024, win_length=1024, hop_length=256, sample_rate=22050, n_mels=80, f_min=0, f_max=8000):
        if win_length != n_fft:
            raise NotImplementedError("evmi_mel_spectrogram_f32 covers win_length == n_fft (the reference's defaults)")
        if n_fft % hop_length:
            raise NotImplementedError("n_fft must be a multiple of hop_length")
        self.n_fft, self.hop, self.n_mels = n_fft, hop_length, n_mels
        basis, self.nb_pad = windowed_dft_basis(n_fft, win_length)
        self._basis_host = torch.from_numpy(basis)
        self._mel_host = torch.from_numpy(slaney_mel_filterbank(sample_rate, n_fft, n_mels, f_min, f_max))
        self._dev = {}

    def _consts(self, device):
        if device not in self._dev:
            self._dev[device] = (self._basis_host.to(device), self._mel_host.to(device))
        return self._dev[device]

    def __call__(self, audio: torch.Tensor, log: bool = False, return_energy: bool = False, return_magnitude: bool = False,
                 lens: torch.Tensor | None = None):
        """``lens`` [B] (int32, device): a ragged batch -- rows of ``audio`` are zero-padded utterances; item b is transformed as if
        alone (reflection at its own end), its valid frames are the first ``1 + lens[b] // hop`` of its row."""
        if not audio.is_cuda:
            raise RuntimeError("everyvoice_amd.spectral computes on the GPU only (no CPU fallback)")
        squeeze = audio.dim() == 1
        x = audio.reshape(-1, audio.shape[-1]).to(torch.float32).contiguous()
        B, S = x.shape
        frames = 1 + S // self.hop
        basis, melb = self._consts(x.device)
        mel = torch.empty(B, self.n_mels, frames, device=x.device, dtype=torch.float32)
        energy = torch.empty(B, frames, device=x.device, dtype=torch.float32) if return_energy else None
        mag = torch.empty(B, self.n_fft // 2 + 1, frames, device=x.device, dtype=torch.float32) if return_magnitude else None
        lib = _lib.load()
        with torch.cuda.device(x.device):
            if lens is not None:
                lens = lens.to(x.device, torch.int32).contiguous()
                _lib.check(
                    lib.evmi_mel_spectrogram_ragged_f32(x.data_ptr(), lens.data_ptr(), basis.data_ptr(), melb.data_ptr(), mel.data_ptr(),
                                                        _lib.ptr(energy), _lib.ptr(mag), B, S, self.n_fft, self.hop, self.nb_pad, self.n_mels,
                                                        int(log), _lib.current_stream_ptr(x.device)),
                    "evmi_mel_spectrogram_ragged_f32",
                )
            else:
                _lib.check(
                    lib.evmi_mel_spectrogram_f32(x.data_ptr(), basis.data_ptr(), melb.data_ptr(), mel.data_ptr(), _lib.ptr(energy),
                                                 _lib.ptr(mag), B, S, self.n_fft, self.hop, self.nb_pad, self.n_mels, int(log),
                                                 _lib.current_stream_ptr(x.device)),
                    "evmi_mel_spectrogram_f32",
                )
        shape = tuple(audio.shape[:-1])
        outs = [mel[0] if squeeze else mel.reshape(shape + mel.shape[1:])]
        if return_energy:
            outs.append(energy[0] if squeeze else energy.reshape(shape + energy.shape[1:]))
        if return_magnitude:
            outs.append(mag[0] if squeeze else mag.reshape(shape + mag.shape[1:]))
        return outs[0] if len(outs) == 1 else tuple(outs)


def get_spectral_transform(spec_type, n_fft, win_length, hop_length, sample_rate=None, n_mels=None, f_min=0, f_max=8000):
    """Same signature as the reference; "mel-librosa" (the default spec_type) is implemented on the GPU."""
    if spec_type == "mel-librosa":
        return MelSpectrogram(n_fft, win_length, hop_length, sample_rate, n_mels, f_min, f_max)
    raise NotImplementedError(f"spec_type {spec_type!r}: only the reference's default 'mel-librosa' runs on libevmi_hip")


def extract_spectral_features(audio: torch.Tensor, transform: MelSpectrogram, normalize: bool = True, truncate: bool = True):
    """log-mel of ``audio`` [.., S]; ``truncate`` keeps S // hop frames as Preprocessor.process_spec does."""
    mel = transform(audio, log=normalize)
    return mel[..., : audio.shape[-1] // transform.hop] if truncate else mel


def extract_energy(audio: torch.Tensor, transform: MelSpectrogram, truncate: bool = True):
    """(log-mel, energy) with energy = ||log-mel||_2 over the mel bins (Preprocessor.extract_energy)."""
    mel, energy = transform(audio, log=True, return_energy=True)
    n = audio.shape[-1] // transform.hop
    return (mel[..., :n], energy[..., :n]) if truncate else (mel, energy)
