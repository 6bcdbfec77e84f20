"""Data formats on either side of the hot path: the audio gating + on-disk layout of
the reference's preprocessor for the features computed on the GPU (spec, energy, audio), phone-level averaging,
and the synthesis writers (PCM-16 wav, ``[n_mels, T]`` spec) with the reference's file naming.

Mirrors (paths relative to the reference):
  process_audio            everyvoice/preprocessor/preprocessor.py:131-218  (channel / length gates, peak normalise to
                           0.95, truncate to a multiple of the hop; the LUFS gate, SoX effects and resampling are the
                           reference's CPU steps and are not reproduced: inputs must already be at the target rate)
  create_path naming       everyvoice/preprocessor/preprocessor.py:502-508, 529-533, 633-639
                           ``<save_dir>/<kind>/<basename>--<speaker>--<language>--<kind-file>``
  average_data_by_durations  everyvoice/preprocessor/preprocessor.py:287-300 (host loop, as in the reference)
  save_wav / save_tensor   everyvoice/preprocessor/helpers.py:23-44  (PCM_S 16-bit)
  prediction file names    everyvoice/base_cli/prediction_writing_callback.py:35-41
File IO and per-utterance scalars stay on the host (they are IO-bound plumbing); every per-sample transform
(STFT, mel, log, energy, vocoding) runs in libevmi_hip.
"""

from __future__ import annotations

import wave
from pathlib import Path

import numpy as np
import torch

from .config import AudioConfig
from .spectral import MelSpectrogram

SEP = "--"


def load_wav(path) -> tuple[torch.Tensor, int, float]:
    """PCM wav -> (float32 [channels, samples] in [-1, 1), sampling rate, seconds), like torchaudio.load."""
    with wave.open(str(path), "rb") as w:
        sr, ch, width, n = w.getframerate(), w.getnchannels(), w.getsampwidth(), w.getnframes()
        raw = w.readframes(n)
    if width == 2:
        data = np.frombuffer(raw, dtype="<i2").astype(np.float32) / 32768.0
    elif width == 4:
        data = np.frombuffer(raw, dtype="<i4").astype(np.float32) / 2147483648.0
    elif width == 1:
        data = (np.frombuffer(raw, dtype=np.uint8).astype(np.float32) - 128.0) / 128.0
    else:
        raise ValueError(f"{path}: unsupported sample width {width}")
    audio = torch.from_numpy(data.reshape(-1, ch).T.copy()) if n else torch.zeros(ch, 0)
    return audio, sr, (n / sr if sr else 0.0)


def save_wav(audio: torch.Tensor, path, sr: int, bits_per_sample: int = 16) -> None:
    """float [-1, 1] [S] or [1, S] -> PCM_S wav (creates the directory), as preprocessor/helpers.py:31-44."""
    if bits_per_sample != 16:
        raise NotImplementedError("only 16-bit PCM (AudioConfig.target_bit_depth default)")
    path = Path(path)
    path.parent.mkdir(parents=True, exist_ok=True)
    x = audio.detach().to("cpu", torch.float32).reshape(-1).numpy()
    pcm = np.clip(np.round(x * 32768.0), -32768, 32767).astype("<i2")
    with wave.open(str(path), "wb") as w:
        w.setnchannels(1)
        w.setsampwidth(2)
        w.setframerate(sr)
        w.writeframes(pcm.tobytes())


def save_tensor(tensor: torch.Tensor, path) -> None:
    path = Path(path)
    path.parent.mkdir(parents=True, exist_ok=True)
    torch.save(tensor.detach().cpu(), path)


def feature_path(save_dir, kind: str, basename: str, speaker: str, language: str, fn: str) -> Path:
    return Path(save_dir) / kind / SEP.join([basename, speaker, language, fn])


def process_audio(wav_path, cfg: AudioConfig, normalize: bool = True):
    """(audio [S'], sr) with S' a multiple of the hop, or (None, reason) when the file is skipped."""
    audio, sr, seconds = load_wav(wav_path)
    if audio.shape[0] > 2:
        return None, "multichannel_files"
    if seconds > cfg.max_audio_length:
        return None, "audio_too_long"
    if seconds < cfg.min_audio_length:
        return None, "audio_too_short"
    if sr != cfg.input_sampling_rate:
        raise NotImplementedError(f"{wav_path}: {sr} Hz; resampling is the reference's CPU step (expected {cfg.input_sampling_rate})")
    if audio.shape[0] == 2:
        audio = audio.mean(0, keepdim=True)  # sox "channels 1" (the reference's default effect)
    peak = float(audio.abs().max())
    if not np.isfinite(peak) or peak == 0.0:
        return None, "audio_empty"
    if normalize:
        audio = audio / peak * 0.95
    audio = audio.squeeze(0)
    n = audio.numel() // cfg.fft_hop_size * cfg.fft_hop_size
    return audio[:n], sr


def average_data_by_durations(data: torch.Tensor, durations) -> torch.Tensor:
    """Per-phone mean of ``data`` over ``d_i`` consecutive frames, 1e-7 where d_i == 0."""
    out, pos = [], 0
    for d in torch.as_tensor(durations).tolist():
        d = int(d)
        out.append(float(data[pos : pos + d].mean()) if d > 0 else 1e-7)
        pos += d
    return torch.tensor(out, dtype=torch.float32)


class GpuPreprocessor:
    """spec + energy (+ normalised audio) of a list of wavs, in the reference's on-disk layout."""

    def __init__(self, cfg: AudioConfig | None = None, device="cuda:0"):
        self.cfg = cfg or AudioConfig()
        self.device = torch.device(device)
        self.transform = MelSpectrogram(self.cfg.n_fft, self.cfg.fft_window_size, self.cfg.fft_hop_size,
                                        self.cfg.input_sampling_rate, self.cfg.n_mels, self.cfg.f_min, self.cfg.f_max)
        self.counters: dict[str, int] = {}

    def features(self, audio: torch.Tensor):
        """audio [S] (host or device) -> (log-mel [n_mels, S // hop], energy [S // hop]) on the device."""
        x = audio.to(self.device)
        mel, energy = self.transform(x, log=True, return_energy=True)
        n = x.shape[-1] // self.cfg.fft_hop_size
        return mel[..., :n].contiguous(), energy[..., :n].contiguous()

    def process(self, items: list[dict], save_dir) -> list[dict]:
        """items: dicts with ``basename``, ``speaker``, ``language``, ``wav``.  Returns the items that were kept."""
        kept = []
        sr_tag, spec_fn = self.cfg.input_sampling_rate, f"spec-{self.cfg.input_sampling_rate}-{self.cfg.spec_type}.pt"
        for it in items:
            audio, info = process_audio(it["wav"], self.cfg)
            if audio is None:
                self.counters[info] = self.counters.get(info, 0) + 1
                continue
            mel, energy = self.features(audio)
            ids = (it["basename"], it.get("speaker", "default"), it.get("language", "default"))
            save_wav(audio, feature_path(save_dir, "audio", *ids, f"audio-{sr_tag}.wav"), sr_tag, self.cfg.target_bit_depth)
            save_tensor(mel, feature_path(save_dir, "spec", *ids, spec_fn))
            save_tensor(energy, feature_path(save_dir, "energy", *ids, "energy.pt"))
            self.counters["processed_files"] = self.counters.get("processed_files", 0) + 1
            kept.append(dict(it, frames=mel.shape[1], samples=audio.numel()))
        return kept


def synthesize_from_spec(spec: torch.Tensor, vocoder, out_dir, basename: str, speaker: str = "default",
                         language: str = "default", sr: int = 22050) -> Path:
    """``everyvoice synthesize from-spec``: a saved ``[n_mels, T]`` (or ``[B, n_mels, T]``) log-mel -> wav file named
    like the reference's prediction writers (``<basename>--<speaker>--<language>--pred.wav``)."""
    dev = next(vocoder.parameters()).device
    mel = spec.to(dev, torch.float32)
    if mel.dim() == 2:
        mel = mel.unsqueeze(0)
    wav = vocoder(mel)
    path = Path(out_dir) / SEP.join([basename, speaker, language, "pred.wav"])
    save_wav(wav[0, 0], path, sr)
    return path


class Scaler:
    """Dataset-level statistics of a per-utterance feature (pitch, energy) and its standardisation -- the behaviour of the
    reference's ``Scaler`` (``everyvoice/preprocessor/helpers.py:47-106``): values are collected with ``append``; ``calculate_stats``
    takes min / max / unbiased std over the non-NaN entries and ``nanmean`` over all of them, and reports the extrema after
    ``(x - mean) / std``; the collected list is read-only from outside.  Tensors may live on any device."""

    _FIELDS = ("min", "max", "std", "mean", "norm_min", "norm_max")

    def __init__(self):
        self.clear_data()

    def clear_data(self):
        self._chunks: list[torch.Tensor] = []
        self._all: torch.Tensor | None = None
        for f in self._FIELDS:
            setattr(self, f, None)

    def append(self, value: torch.Tensor):
        self._chunks.append(value)
        self._all = None

    @property
    def data(self):
        return self._chunks

    @data.setter
    def data(self, value):
        raise ValueError(f"Scaler.data is read-only (got {value!r}): use Scaler.append(...) or Scaler.clear_data()")

    def __len__(self):
        return len(self._chunks)

    def normalize(self, x):
        return (x - self.mean) / self.std

    def denormalize(self, x):
        return x * self.std + self.mean

    def calculate_stats(self):
        if not self._chunks:
            return None
        if self._all is None:
            self._all = torch.cat(self._chunks)
        finite = self._all[~torch.isnan(self._all)]
        self.min, self.max, self.std = finite.min(), finite.max(), finite.std()
        self.mean = torch.nanmean(self._all)
        self.norm_min, self.norm_max = self.normalize(self.min), self.normalize(self.max)
        return {"sample_size": len(self), **{f: float(getattr(self, f)) for f in self._FIELDS}}


def synthesize_from_text(ids: torch.Tensor, lens: torch.Tensor, fs2, vocoder, out_dir, basenames: list[str], speaker: str = "default",
                         language: str = "default", output_types=("wav", "spec"), sr: int = 22050, hop: int = 256,
                         duration_control: float = 1.0, global_step: int | None = None) -> list[dict]:
    """The device part of ``everyvoice synthesize from-text`` (``fs2.cli.synthesize.synthesize_helper``,
    ``everyvoice/demo/app.py:84-106``): token ids -> FastSpeech2 (postnet mel) -> HiFiGAN -> files named as the reference's
    prediction writers name them (``everyvoice/base_cli/prediction_writing_callback.py:35-41``):
    ``<out_dir>/wav/<basename>--<speaker>--<language>--pred.wav`` and ``<out_dir>/synthesized_spec/...--spec-pred....pt``
    holding ``[n_mels, T]``.  Text normalisation / g2p (CPU string work) stays with the caller: ``ids`` are symbol ids, 0 pads."""
    mel, post, durations, _, _, mel_lens = fs2(ids, lens, duration_control=duration_control)
    wav = vocoder(post.transpose(1, 2).contiguous()) if "wav" in output_types else None
    results = []
    for i, base in enumerate(basenames):
        T = int(mel_lens[i])
        rec = {"basename": base, "frames": T, "durations": durations[i, : int(lens[i])].cpu()}
        if "spec" in output_types:
            p = Path(out_dir) / "synthesized_spec" / SEP.join([base, speaker, language, f"spec-pred-{sr}-mel-librosa.pt"])
            save_tensor(post[i, :T].transpose(0, 1).contiguous(), p)
            rec["spec"] = p
        if wav is not None:
            p = Path(out_dir) / "wav" / SEP.join([base, speaker, language, "pred.wav"])
            save_wav(wav[i, 0, : T * hop], p, sr)
            rec["wav"] = p
        results.append(rec)
    return results
