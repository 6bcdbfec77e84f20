"""GPU mirror of the reference's default spectral transform ("mel-librosa") and of the preprocessor's
feature extraction built on it:

  get_spectral_transform(...)            everyvoice/utils/heavy.py:47-119   (all branches: "mel-librosa", "mel", "linear", "raw", "istft")
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


def htk_mel_filterbank(sr: int, n_fft: int, n_mels: int, fmin: float = 0.0, fmax: float | None = None) -> np.ndarray:
    """torchaudio.functional.melscale_fbanks(n_fft // 2 + 1, fmin, fmax, n_mels, sr, norm="slaney", mel_scale="htk") as
    torchaudio.transforms.MelSpectrogram(norm="slaney") builds it: HTK mel scale (2595 log10(1 + f / 700)), triangles over
    linspace(0, sr // 2, n_freqs) scaled by 2 / bandwidth; float32 [n_mels, n_fft // 2 + 1] (torchaudio stores the transpose)."""
    fmax = float(sr // 2) if fmax is None else float(fmax)
    freqs = np.linspace(0.0, float(sr // 2), n_fft // 2 + 1)
    m_lo, m_hi = 2595.0 * np.log10(1.0 + fmin / 700.0), 2595.0 * np.log10(1.0 + fmax / 700.0)
    pts = 700.0 * (10.0 ** (np.linspace(m_lo, m_hi, n_mels + 2) / 2595.0) - 1.0)
    width = np.diff(pts)
    slopes = pts[:, None] - freqs[None, :]
    fb = np.maximum(0.0, np.minimum(-slopes[:-2] / width[:-1, None], slopes[2:] / width[1:, None]))
    return (fb * (2.0 / (pts[2:] - pts[:-2]))[:, None]).astype(np.float32)


class Spectrogram:
    """torchaudio.transforms.Spectrogram(n_fft, win_length, hop_length, power) with its defaults (periodic hann window, centred, reflect
    padding, one-sided, not normalised): spec types "linear" (power 2) and "raw" (power None: complex64) of
    everyvoice/utils/heavy.py:101-114.  audio [..., S] -> [..., n_fft // 2 + 1, 1 + S // hop].  The DFT is a fp32 GEMM with the windowed
    basis over the framed signal (evmi_stft_frames_f32, evmi_gemm_f32), then one layout pass (evmi_spectrogram_layout_f32)."""

    def __init__(self, n_fft=400, win_length=None, hop_length=None, power=2.0):
        self.n_fft = int(n_fft)
        self.win = int(win_length) if win_length is not None else self.n_fft
        self.hop = int(hop_length) if hop_length is not None else self.win // 2
        if power not in (None, 1, 1.0, 2, 2.0):
            raise NotImplementedError("Spectrogram: power is None (complex), 1 or 2")
        self.power = power
        self.nb = self.n_fft // 2 + 1
        basis, _ = windowed_dft_basis(self.n_fft, self.win)
        b = torch.from_numpy(basis)
        self._host = (b[:, 0 : 2 * self.nb : 2].t().contiguous(), b[:, 1 : 2 * self.nb : 2].t().contiguous())  # [nb, n_fft] each
        self._dev = {}

    def _consts(self, device):
        if device not in self._dev:
            self._dev[device] = tuple(t.to(device) for t in self._host)
        return self._dev[device]

    def _planes(self, audio):
        """-> (x [B, S], re, im [nb, B*F], F)"""
        from .train import ops

        if not audio.is_cuda:
            raise RuntimeError("everyvoice_amd.spectral computes on the GPU only (no CPU fallback)")
        if audio.shape[-1] <= self.n_fft // 2:
            raise ValueError("reflect padding needs more than n_fft // 2 samples")
        x = audio.reshape(-1, audio.shape[-1]).to(torch.float32).contiguous()
        cos, sin = self._consts(x.device)
        with torch.cuda.device(x.device):
            fr, F = ops.stft_frames(x, self.n_fft, self.hop)
            re = torch.empty(self.nb, fr.shape[1], device=x.device)
            im = torch.empty_like(re)
            ops.gemm(cos, fr, re)
            ops.gemm(sin, fr, im)
        return x, re, im, F

    @staticmethod
    def _layout(mode, re, im, B, C, F, complex_out=False):
        out = torch.empty((B, C, F, 2) if complex_out else (B, C, F), device=re.device, dtype=torch.float32)
        with torch.cuda.device(re.device):
            _lib.check(_lib.load().evmi_spectrogram_layout_f32(mode, re.data_ptr(), _lib.ptr(im), out.data_ptr(), B, C, F,
                                                               _lib.current_stream_ptr(re.device)), "evmi_spectrogram_layout_f32")
        return out

    def __call__(self, audio: torch.Tensor) -> torch.Tensor:
        x, re, im, F = self._planes(audio)
        B = x.shape[0]
        if self.power is None:
            out = torch.view_as_complex(self._layout(1, re, im, B, self.nb, F, complex_out=True))
        else:
            out = self._layout(0 if float(self.power) == 2.0 else 4, re, im, B, self.nb, F)
        return out.reshape(tuple(audio.shape[:-1]) + out.shape[1:])


class TorchaudioMelSpectrogram(Spectrogram):
    """torchaudio.transforms.MelSpectrogram(sample_rate, n_fft, win_length, hop_length, f_min, f_max, n_mels, norm="slaney", center=True)
    -- spec type "mel" (everyvoice/utils/heavy.py:59-68): power spectrogram, HTK-scale filterbank with Slaney area normalisation."""

    def __init__(self, sample_rate=16000, n_fft=400, win_length=None, hop_length=None, f_min=0.0, f_max=None, n_mels=128):
        super().__init__(n_fft, win_length, hop_length, power=2.0)
        self.n_mels = int(n_mels)
        self._fb_host = torch.from_numpy(htk_mel_filterbank(sample_rate, self.n_fft, self.n_mels, f_min, f_max))
        self._fb = {}

    def __call__(self, audio: torch.Tensor) -> torch.Tensor:
        from .train import ops

        x, re, im, F = self._planes(audio)
        B = x.shape[0]
        if x.device not in self._fb:
            self._fb[x.device] = self._fb_host.to(x.device)
        with torch.cuda.device(x.device):
            power = ops.elementwise(ops.EW_AXPBY, ops.elementwise(ops.EW_MUL, re, re), ops.elementwise(ops.EW_MUL, im, im), p0=1.0, p1=1.0)
            mel = torch.empty(self.n_mels, power.shape[1], device=x.device)
            ops.gemm(self._fb[x.device], power, mel)
        out = self._layout(2, mel, None, B, self.n_mels, F)
        return out.reshape(tuple(audio.shape[:-1]) + out.shape[1:])


class InverseSpectrogram:
    """torchaudio.transforms.InverseSpectrogram(n_fft, win_length, hop_length) -- "istft" (everyvoice/utils/heavy.py:115-118): torch.istft
    with a periodic hann window, centred, one-sided: complex [..., n_fft // 2 + 1, F] -> [..., hop * (F - 1)].  Overlap-add as a GEMM with
    the windowed inverse-DFT basis and a fold (evmi_gemm_f32, evmi_fold_cbt_f32), divided by the window envelope."""

    def __init__(self, n_fft=400, win_length=None, hop_length=None):
        self.n_fft = int(n_fft)
        self.win = int(win_length) if win_length is not None else self.n_fft
        self.hop = int(hop_length) if hop_length is not None else self.win // 2
        self.nb = self.n_fft // 2 + 1
        n = np.arange(self.n_fft, dtype=np.float64)
        w = np.zeros(self.n_fft)
        left = (self.n_fft - self.win) // 2
        w[left : left + self.win] = 0.5 - 0.5 * np.cos(2.0 * np.pi * np.arange(self.win) / self.win)
        h = np.arange(self.nb, dtype=np.float64)
        c = np.where((h == 0) | (h == self.n_fft // 2), 1.0, 2.0) / self.n_fft  # the mirrored bins of the one-sided spectrum
        ang = 2.0 * np.pi * ((h[:, None] * n[None, :]) % self.n_fft) / self.n_fft
        basis = np.concatenate([c[:, None] * np.cos(ang) * w[None, :], -c[:, None] * np.sin(ang) * w[None, :]])  # [2 nb, n_fft]
        if self.n_fft % 2 == 0:
            basis[self.nb + self.n_fft // 2] = 0.0  # the imaginary parts of bins 0 and n_fft / 2 do not reach a real signal
        basis[self.nb] = 0.0
        self._basis_host = torch.from_numpy(basis.astype(np.float32))
        self._win_sq = w * w
        self._dev, self._env = {}, {}

    def _inv_envelope(self, frames, device):
        key = (frames, device)
        if key not in self._env:
            env = np.zeros(self.n_fft + self.hop * (frames - 1))
            for t in range(frames):
                env[t * self.hop : t * self.hop + self.n_fft] += self._win_sq
            env = env[self.n_fft // 2 : self.n_fft // 2 + self.hop * (frames - 1)]
            if env.min() < 1e-11:
                raise ValueError("window overlap-add envelope reaches zero (torch.istft refuses the same input)")
            self._env[key] = torch.from_numpy((1.0 / env).astype(np.float32)).to(device)
        return self._env[key]

    def __call__(self, spec: torch.Tensor) -> torch.Tensor:
        from .train import ops

        if not spec.is_cuda:
            raise RuntimeError("everyvoice_amd.spectral computes on the GPU only (no CPU fallback)")
        if not spec.is_complex() or spec.shape[-2] != self.nb:
            raise ValueError(f"InverseSpectrogram: complex [..., {self.nb}, frames] expected")
        F = spec.shape[-1]
        z = torch.view_as_real(spec.reshape(-1, self.nb, F).to(torch.complex64).contiguous())  # [B, nb, F, 2]
        B = z.shape[0]
        dev = z.device
        if dev not in self._dev:
            self._dev[dev] = self._basis_host.to(dev)
        t_out = self.hop * (F - 1)
        with torch.cuda.device(dev):
            planes = torch.empty(2 * self.nb, B * F, device=dev, dtype=torch.float32)
            _lib.check(_lib.load().evmi_spectrogram_layout_f32(3, z.data_ptr(), None, planes.data_ptr(), B, self.nb, F,
                                                               _lib.current_stream_ptr(dev)), "evmi_spectrogram_layout_f32")
            col = torch.empty(self.n_fft, B * F, device=dev, dtype=torch.float32)
            ops.gemm(self._dev[dev], planes, col, ta=True)  # [n_fft, 2 nb] @ [2 nb, B F]: every frame's windowed inverse DFT
            y = ops.fold(col, 1, B, t_out, F, self.n_fft, self.hop, self.n_fft // 2, 1)  # overlap-add, n_fft // 2 trimmed per side
            y = y.view(B, t_out) * self._inv_envelope(F, dev)[None, :]
        return y.reshape(tuple(spec.shape[:-2]) + (t_out,))


def get_spectral_transform(spec_type, n_fft, win_length, hop_length, sample_rate=None, n_mels=None, f_min=0, f_max=8000):
    """Same signature and branches as the reference (everyvoice/utils/heavy.py:47-119); everything runs on libevmi_hip.  An unknown
    spec_type returns None, as there."""
    if spec_type == "mel-librosa":
        return MelSpectrogram(n_fft, win_length, hop_length, sample_rate, n_mels, f_min, f_max)
    if spec_type == "mel":
        return TorchaudioMelSpectrogram(sample_rate, n_fft, win_length, hop_length, f_min, f_max, n_mels)
    if spec_type == "linear":
        return Spectrogram(n_fft, win_length, hop_length)
    if spec_type == "raw":
        return Spectrogram(n_fft, win_length, hop_length, power=None)
    if spec_type == "istft":
        return InverseSpectrogram(n_fft, win_length, hop_length)
    return None


def extract_spectral_features(audio: torch.Tensor, transform: MelSpectrogram, normalize: bool = True, truncate: bool = True):
    """log-mel of ``audio`` [.., S]; ``truncate`` keeps S // hop frames as Preprocessor.process_spec does."""
    mel = transform(audio, log=normalize)
    return mel[..., : audio.shape[-1] // transform.hop] if truncate else mel


def extract_energy(audio: torch.Tensor, transform: MelSpectrogram, truncate: bool = True):
    """(log-mel, energy) with energy = ||log-mel||_2 over the mel bins (Preprocessor.extract_energy)."""
    mel, energy = transform(audio, log=True, return_energy=True)
    n = audio.shape[-1] // transform.hop
    return (mel[..., :n], energy[..., :n]) if truncate else (mel, energy)
