"""Data formats on either side of the hot path: the audio gating + on-disk layout of
the reference's preprocessor for the features computed on the GPU (spec, energy, audio), phone-level averaging,
and the synthesis writers (PCM-16 wav, ``[n_mels, T]`` spec) with the reference's file naming.

Mirrors (paths relative to the reference):
  process_audio            everyvoice/preprocessor/preprocessor.py:131-218  (channel / length gates, the -36 LUFS "audio_empty"
                           gate, the default SoX effect "channels 1" (mix-down), resampling, peak normalise to 0.95, truncate
                           to a multiple of the hop).  Loudness, resampling and normalisation run on the device
                           (csrc/preprocess_ops.hip; torchaudio's algorithms restated, parity unpinned: torchaudio is not in
                           the image); other SoX effect chains are not reproduced
  preprocess / .config-lock  preprocessor.py:974-1082 (what the lock records, when a run refuses to continue)
  compute_stats / normalize_stats  preprocessor.py:378-490 (dataset statistics of energy / pitch -> Stats, files rewritten normalised)
  create_path naming       everyvoice/preprocessor/preprocessor.py:502-508, 529-533, 633-639
                           ``<save_dir>/<kind>/<basename>--<speaker>--<language>--<kind-file>``
  average_data_by_durations  everyvoice/preprocessor/preprocessor.py:287-300 (host loop, as in the reference)
  save_wav / save_tensor   everyvoice/preprocessor/helpers.py:23-44  (PCM_S 16-bit)
  prediction file names    everyvoice/base_cli/prediction_writing_callback.py:35-41
File IO and per-utterance scalars stay on the host (they are IO-bound plumbing); every per-sample transform
(STFT, mel, log, energy, vocoding) runs in libevmi_hip.
"""

from __future__ import annotations

import json
import math
import wave
from glob import glob
from pathlib import Path

import numpy as np
import torch

from .config import AudioConfig
from .spectral import MelSpectrogram, get_spectral_transform

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


def sinc_resample_kernel(orig_freq: int, new_freq: int, lowpass_filter_width: int = 6, rolloff: float = 0.99):
    """The polyphase windowed-sinc filter bank of torchaudio.functional.resample (defaults: sinc_interp_hann): ``new`` filters of
    ``2 * width + orig`` taps applied at stride ``orig`` -> (kernel [new, 1, taps] float32, width, orig, new), rates reduced
    by their gcd.  Built in float64 on the host, as torchaudio does; the convolution itself runs on the device."""
    g = math.gcd(int(orig_freq), int(new_freq))
    orig, new = int(orig_freq) // g, int(new_freq) // g
    base = min(orig, new) * rolloff
    width = math.ceil(lowpass_filter_width * orig / base)
    taps = np.arange(-width, width + orig, dtype=np.float64)[None, :] / orig
    t = (np.arange(0, -new, -1, dtype=np.float64)[:, None] / new + taps) * base
    t = np.clip(t, -lowpass_filter_width, lowpass_filter_width)
    window = np.cos(t * math.pi / lowpass_filter_width / 2) ** 2
    t = t * math.pi
    with np.errstate(invalid="ignore", divide="ignore"):
        sinc = np.where(t == 0, 1.0, np.sin(t) / t)
    return torch.from_numpy((sinc * window * (base / orig)).astype(np.float32))[:, None, :].contiguous(), width, orig, new


_RESAMPLE_KERNELS: dict = {}


def resample(audio: torch.Tensor, orig_freq: int, new_freq: int) -> torch.Tensor:
    """[B, S] (device) -> [B, ceil(new * S / orig)]: torchaudio.functional.resample as ONE strided convolution on the device
    (evmi_conv1d_f32: exact fp32 fmaf chains) with the filter bank above (preprocessor.py:196-198)."""
    from . import _lib

    if orig_freq == new_freq:
        return audio
    if not audio.is_cuda:
        raise RuntimeError("everyvoice_amd.pipeline.resample computes on the GPU only (no CPU fallback)")
    key = (int(orig_freq), int(new_freq), audio.device)
    if key not in _RESAMPLE_KERNELS:
        k, width, orig, new = sinc_resample_kernel(orig_freq, new_freq)
        _RESAMPLE_KERNELS[key] = (k.to(audio.device), width, orig, new)
    kernel, width, orig, new = _RESAMPLE_KERNELS[key]
    x = audio.to(torch.float32)
    B, S = x.shape
    xp = torch.nn.functional.pad(x, (width, width + orig)).contiguous()  # (memory plumbing: the zero margin of the filter)
    taps = kernel.shape[-1]
    frames = (xp.shape[1] - taps) // orig + 1
    y = torch.empty(B, new, frames, device=x.device, dtype=torch.float32)
    _lib.check(_lib.load().evmi_conv1d_f32(xp.data_ptr(), kernel.data_ptr(), 0, 0, y.data_ptr(), B, 1, xp.shape[1], new, taps, orig, 0, 1, 1,
                                           1.0, 1.0, 0, _lib.current_stream_ptr(x.device)), "evmi_conv1d_f32")
    return y.transpose(1, 2).reshape(B, -1)[:, : math.ceil(new * S / orig)].contiguous()


def loudness(audio: torch.Tensor, lens: torch.Tensor, sample_rate: int) -> torch.Tensor:
    """Integrated loudness (LKFS) of a zero-padded batch [items, channels, t_max] with lengths [items] -> [items] on the device
    (torchaudio.transforms.Loudness restated: K-weighting, 400 ms blocks, absolute / relative gates)."""
    from . import _lib

    if not audio.is_cuda:
        raise RuntimeError("everyvoice_amd.pipeline.loudness computes on the GPU only (no CPU fallback)")
    x = audio.to(torch.float32).contiguous()
    items, channels, t_max = x.shape
    lib = _lib.load()
    lens32 = lens.to(x.device, torch.int32).contiguous()
    y2 = torch.empty_like(x)
    z = torch.empty(max(1, lib.evmi_loudness_scratch_elems(items, channels, t_max, sample_rate)), device=x.device, dtype=torch.float32)
    out = torch.empty(items, device=x.device, dtype=torch.float32)
    _lib.check(lib.evmi_loudness_lkfs_f32(x.data_ptr(), lens32.data_ptr(), y2.data_ptr(), z.data_ptr(), out.data_ptr(), items, channels, t_max,
                                          int(sample_rate), _lib.current_stream_ptr(x.device)), "evmi_loudness_lkfs_f32")
    return out


def world_frames(n_samples: int, hop: int, sample_rate: int) -> int:
    """pyworld's frame count for n_samples: GetSamplesForDIO with frame_period = hop / fs * 1000, in the same double arithmetic."""
    frame_period = hop / sample_rate * 1000.0
    return int(1000.0 * n_samples / sample_rate / frame_period) + 1


def extract_pitch(audio: torch.Tensor, lens: torch.Tensor | None, hop: int, sample_rate: int, f0_floor: float = 71.0, f0_ceil: float = 800.0,
                  voicing_threshold: float = 0.8, interpolate: bool = True, estimator: str = "world", speed: int = 4) -> torch.Tensor:
    """Frame-level pitch (Hz) of a zero-padded batch [items, t_max] on the device: ``Preprocessor.extract_pitch``
    (preprocessor.py:244-285) -- ``pyworld.dio(x.f64, fs, frame_period = hop / fs * 1000, speed = 4)`` -> ``pyworld.stonemask`` -> unvoiced
    frames -> NaN -> linear interpolation across them, an utterance without any voiced frame -> zeros.

    ``estimator="world"`` (default, round 5): WORLD's DIO + StoneMask themselves, in float64 on the device (csrc/pitch_world.hip;
    oracle/pitch_world_ref.py restates the algorithm and reproduces the reference's pyworld fixture to 1e-13 Hz).  Output
    [items, world_frames(t_max)]: pyworld's frame count ((int)(1000 n / fs / frame_period) + 1, which is n // hop + 1 or one fewer when
    the double division lands below the integer); item i is valid up to world_frames(lens[i]), zero behind.
    ``estimator="acf"``: the normalised-autocorrelation tracker of rounds 2-4 (its own estimator: values differ from the reference's;
    output [items, t_max // hop + 1]; ``voicing_threshold`` applies to it only)."""
    from . import _lib

    if not audio.is_cuda:
        raise RuntimeError("everyvoice_amd.pipeline.extract_pitch computes on the GPU only (no CPU fallback)")
    if estimator not in ("world", "acf"):
        raise ValueError("extract_pitch: estimator 'world' (DIO + StoneMask, the reference's) or 'acf'")
    x = audio.to(torch.float32).reshape(-1, audio.shape[-1]).contiguous()
    items, t_max = x.shape
    lens32 = (torch.full((items,), t_max) if lens is None else lens).to(x.device, torch.int32).contiguous()
    lib = _lib.load()
    if estimator == "world":
        frames = world_frames(t_max, hop, sample_rate)
        f0 = torch.empty(items, frames, device=x.device, dtype=torch.float32)
        n_ws = lib.evmi_pitch_world_ws_elems(items, t_max, int(sample_rate), hop, int(speed), float(f0_floor), float(f0_ceil), 2.0)
        if n_ws <= 0:
            raise RuntimeError("extract_pitch: evmi_pitch_world_ws_elems rejected the shape")
        ws = torch.empty(n_ws, device=x.device, dtype=torch.float64)
        _lib.check(lib.evmi_pitch_world_f64(x.data_ptr(), lens32.data_ptr(), f0.data_ptr(), ws.data_ptr(), n_ws, items, t_max, int(sample_rate), hop, int(speed),
                                            float(f0_floor), float(f0_ceil), 2.0, 0.1, _lib.current_stream_ptr(x.device)), "evmi_pitch_world_f64")
        n_valid = [world_frames(int(n), hop, sample_rate) for n in lens32.cpu().tolist()]
    else:
        f0 = torch.empty(items, t_max // hop + 1, device=x.device, dtype=torch.float32)
        _lib.check(lib.evmi_pitch_acf_f32(x.data_ptr(), lens32.data_ptr(), f0.data_ptr(), items, t_max, hop, int(sample_rate), float(f0_floor),
                                          float(f0_ceil), float(voicing_threshold), _lib.current_stream_ptr(x.device)), "evmi_pitch_acf_f32")
        n_valid = [int(n) // hop + 1 for n in lens32.cpu().tolist()]
    if not interpolate:
        return f0
    out = f0.cpu().numpy().astype(np.float64)  # (per-utterance 1-D interpolation over a few hundred frames: host plumbing, as in the reference)
    for i in range(items):
        n = min(n_valid[i], out.shape[1])
        row = out[i, :n]
        voiced = row > 0
        if voiced.any():
            row[~voiced] = np.interp(np.nonzero(~voiced)[0], np.nonzero(voiced)[0], row[voiced])
        out[i, n:] = 0.0
    return torch.from_numpy(out.astype(np.float32)).to(x.device)


LOUDNESS_GATE_LKFS = -36.0  # preprocessor.py:180: "a conservative threshold"


def gate_audio(wav_path, cfg: AudioConfig):
    """The host-side gates of process_audio (file IO and counts): -> (audio [channels, S], sr) or (None, reason)."""
    audio, sr, seconds = load_wav(wav_path)
    if audio.shape[0] > 2:
        return None, "multichannel_files"
    if seconds > cfg.max_audio_length:
        return None, "audio_too_long"
    if seconds < cfg.min_audio_length:
        return None, "audio_too_short"
    return audio, sr


def process_audio_batch(wavs: list[torch.Tensor], sr: int, cfg: AudioConfig, device, normalize: bool = True, resample_rate: int | None = None):
    """The device part of process_audio for a batch of gated utterances at one source rate: loudness gate -> mix-down -> resample
    -> peak normalise -> truncate to a multiple of the hop.  wavs: [channels_i, S_i] host tensors.
    -> (audio [n, t_max] on the device, lens [n] python ints (multiples of the hop), kept indices, {reason: count})."""
    from . import _lib

    counters: dict[str, int] = {}
    n = len(wavs)
    if n == 0:
        return None, [], [], counters
    target_sr = resample_rate or sr
    kept_idx = []
    keep_mask = torch.ones(n, dtype=torch.bool)
    # loudness is measured on the file as loaded (all its channels, original rate), before any effect
    for ch in sorted({w.shape[0] for w in wavs}):
        idx = [i for i, w in enumerate(wavs) if w.shape[0] == ch]
        t_max = max(wavs[i].shape[1] for i in idx)
        batch = torch.zeros(len(idx), ch, t_max)
        for j, i in enumerate(idx):
            batch[j, :, : wavs[i].shape[1]] = wavs[i]
        lk = loudness(batch.to(device), torch.tensor([wavs[i].shape[1] for i in idx]), sr).cpu()
        for j, i in enumerate(idx):
            if torch.isnan(lk[j]) or float(lk[j]) < LOUDNESS_GATE_LKFS:
                keep_mask[i] = False
                counters["audio_empty"] = counters.get("audio_empty", 0) + 1
    kept_idx = [i for i in range(n) if keep_mask[i]]
    if not kept_idx:
        return None, [], [], counters
    t_max = max(wavs[i].shape[1] for i in kept_idx)
    mono = torch.zeros(len(kept_idx), t_max)
    for j, i in enumerate(kept_idx):
        mono[j, : wavs[i].shape[1]] = wavs[i].mean(0)  # SoX "channels 1" (the reference's default effect): the channels' mean
    x = mono.to(device)
    lens = [wavs[i].shape[1] for i in kept_idx]
    if target_sr != sr:
        x = resample(x, sr, target_sr)
        g = math.gcd(sr, target_sr)
        lens = [math.ceil((target_sr // g) * L / (sr // g)) for L in lens]
    if normalize:
        y = torch.empty_like(x)
        lens_t = torch.tensor(lens, dtype=torch.int32, device=x.device)
        _lib.check(_lib.load().evmi_peak_normalize_f32(x.data_ptr(), y.data_ptr(), lens_t.data_ptr(), x.shape[0], x.shape[1], 0.95,
                                                       _lib.current_stream_ptr(x.device)), "evmi_peak_normalize_f32")
        x = y
    lens = [L // cfg.fft_hop_size * cfg.fft_hop_size for L in lens]
    return x, lens, kept_idx, counters


def process_audio(wav_path, cfg: AudioConfig, normalize: bool = True, device="cuda:0", resample_rate: int | None = None):
    """(audio [S'] on the host, sr) with S' a multiple of the hop, or (None, reason) when the file is skipped
    (everyvoice/preprocessor/preprocessor.py:131-218)."""
    audio, info = gate_audio(wav_path, cfg)
    if audio is None:
        return None, info
    x, lens, kept, counters = process_audio_batch([audio], info, cfg, torch.device(device), normalize, resample_rate or cfg.input_sampling_rate)
    if not kept:
        return None, next(iter(counters))
    return x[0, : lens[0]].cpu(), resample_rate or cfg.input_sampling_rate


def average_data_by_durations(data: torch.Tensor, durations) -> torch.Tensor:
    """Per-phone mean of ``data`` over ``d_i`` consecutive frames, 1e-7 where d_i == 0."""
    out, pos = [], 0
    for d in torch.as_tensor(durations).tolist():
        d = int(d)
        out.append(float(data[pos : pos + d].mean()) if d > 0 else 1e-7)
        pos += d
    return torch.tensor(out, dtype=torch.float32)


class ConfigLockMismatch(RuntimeError):
    """The preprocessed directory was written with another configuration (or an interrupted run): refuse to mix outputs."""


class GpuPreprocessor:
    """spec + energy (+ normalised audio) of a list of wavs, in the reference's on-disk layout, several utterances per launch."""

    def __init__(self, cfg: AudioConfig | None = None, device="cuda:0", batch_items: int = 32, pitch: bool = True):
        self.cfg = cfg or AudioConfig()
        self.device = torch.device(device)
        self.batch_items = batch_items
        self.pitch = pitch  # also write pitch/<...>--pitch.pt (FastSpeech2's pitch targets: WORLD's DIO + StoneMask on the device, see extract_pitch)
        self.transform = MelSpectrogram(self.cfg.n_fft, self.cfg.fft_window_size, self.cfg.fft_hop_size,
                                        self.cfg.input_sampling_rate, self.cfg.n_mels, self.cfg.f_min, self.cfg.f_max)
        # spec_type "mel" / "linear" (heavy.py:59-68, 101-107): the generic transforms, one utterance at a time (the ragged one-launch
        # front end above is the default type's); "raw" is complex and has no log / energy -- the reference's process_spec fails on it too
        self.generic = None
        if self.cfg.spec_type != "mel-librosa":
            if self.cfg.spec_type not in ("mel", "linear"):
                raise ValueError(f"preprocessing.audio.spec_type {self.cfg.spec_type!r}: a real-valued spectrogram type is needed")
            self.generic = get_spectral_transform(self.cfg.spec_type, self.cfg.n_fft, self.cfg.fft_window_size, self.cfg.fft_hop_size,
                                                  self.cfg.input_sampling_rate, self.cfg.n_mels, self.cfg.f_min, self.cfg.f_max)
        self.counters: dict[str, int] = {}
        self._interp = None

    def features(self, audio: torch.Tensor):
        """audio [S] (host or device) -> (log-mel [n_mels, S // hop], energy [S // hop]) on the device."""
        x = audio.to(self.device)
        n = x.shape[-1] // self.cfg.fft_hop_size
        if self.generic is not None:  # extract_spectral_features + extract_energy (preprocessor.py:220-233, 302-309) on any real spec
            spec = torch.log(torch.clamp(self.generic(x), min=1e-5))[..., :n].contiguous()
            return spec, torch.linalg.norm(spec, dim=-2)
        mel, energy = self.transform(x, log=True, return_energy=True)
        return mel[..., :n].contiguous(), energy[..., :n].contiguous()

    # -- .config-lock (preprocessor.py:974-1082) ----------------------------------------------------------------------
    def get_config_lock(self, in_progress: bool = True) -> dict:
        return {"info": "This file has the configuration that was used to preprocess files. Do not edit.",
                "status": "in progress" if in_progress else "completed",
                "preprocessing.audio": self.cfg.model_dump(mode="json"), "preprocessing.source_data": {}, "text": {}}

    def save_config_lock(self, save_dir, in_progress: bool):
        save_dir = Path(save_dir)
        save_dir.mkdir(parents=True, exist_ok=True)
        lock = save_dir / ".config-lock"
        if lock.exists():
            lock.chmod(0o666)
        with open(lock, "w", encoding="utf8") as f:
            json.dump(self.get_config_lock(in_progress), f, indent=2, ensure_ascii=False)
            f.write("\n")
        lock.chmod(0o444)  # read-only: discourages edits

    def config_lock_has_conflicts(self, save_dir) -> bool:
        lock = Path(save_dir) / ".config-lock"
        try:
            saved = json.loads(lock.read_text(encoding="utf8"))
        except FileNotFoundError:
            return False
        except json.JSONDecodeError:
            return True
        if saved.get("status") != "completed":  # an interrupted run's partial results cannot be trusted
            return True
        return saved.get("preprocessing.audio") != self.get_config_lock()["preprocessing.audio"] or saved.get("text") != {}

    def process(self, items: list[dict], save_dir, overwrite: bool = False) -> list[dict]:
        """items: dicts with ``basename``, ``speaker``, ``language``, ``wav``.  Returns the items that were kept.  Utterances are
        gated on the host (file IO), then run through the device pipeline in ragged batches of ``batch_items``: loudness gate,
        mix-down, resampling to input_sampling_rate, peak normalisation, STFT -> mel -> log + energy in ONE launch per batch."""
        if self.config_lock_has_conflicts(save_dir) and not overwrite:
            raise ConfigLockMismatch(f"{save_dir}/.config-lock records another audio configuration or an interrupted run; "
                                     "preprocess into a new directory or pass overwrite=True")
        self.save_config_lock(save_dir, in_progress=True)
        kept = []
        sr_tag, spec_fn = self.cfg.input_sampling_rate, f"spec-{self.cfg.input_sampling_rate}-{self.cfg.spec_type}.pt"
        hop = self.cfg.fft_hop_size
        pending: dict[int, list] = {}  # source sampling rate -> [(item, audio)]

        def flush(sr):
            group = pending.pop(sr, [])
            if not group:
                return
            x, lens, kept_idx, counters = process_audio_batch([a for _, a in group], sr, self.cfg, self.device, True, sr_tag)
            for k, v in counters.items():
                self.counters[k] = self.counters.get(k, 0) + v
            if not kept_idx:
                return
            t_max = max(lens)
            x = x[:, :t_max].contiguous()
            if self.generic is None:
                mel, energy = self.transform(x, log=True, return_energy=True, lens=torch.tensor(lens, dtype=torch.int32))
            else:
                per_item = [self.features(x[j, : lens[j]]) for j in range(len(lens))]
                mel = torch.nn.utils.rnn.pad_sequence([m.t() for m, _ in per_item], batch_first=True).transpose(1, 2)
                energy = torch.nn.utils.rnn.pad_sequence([e for _, e in per_item], batch_first=True)
            pitch_host = extract_pitch(x, torch.tensor(lens), hop, sr_tag).cpu() if self.pitch else None
            x_host, mel_host, energy_host = x.cpu(), mel.cpu(), energy.cpu()
            for j, i in enumerate(kept_idx):
                it = group[i][0]
                ids = (it["basename"], it.get("speaker", "default"), it.get("language", "default"))
                n, frames = lens[j], lens[j] // hop
                save_wav(x_host[j, :n], feature_path(save_dir, "audio", *ids, f"audio-{sr_tag}.wav"), sr_tag, self.cfg.target_bit_depth)
                save_tensor(mel_host[j, :, :frames].clone(), feature_path(save_dir, "spec", *ids, spec_fn))
                save_tensor(energy_host[j, :frames].clone(), feature_path(save_dir, "energy", *ids, "energy.pt"))
                self.process_attn_prior(it, frames, save_dir, overwrite)  # (the reference's stage order: ... spec, attn, energy, pitch)
                if pitch_host is not None:
                    save_tensor(pitch_host[j, :frames].clone(), feature_path(save_dir, "pitch", *ids, "pitch.pt"))
                self.counters["processed_files"] = self.counters.get("processed_files", 0) + 1
                kept.append(dict(it, frames=frames, samples=n))

        for it in items:
            audio, info = gate_audio(it["wav"], self.cfg)
            if audio is None:
                self.counters[info] = self.counters.get(info, 0) + 1
                continue
            pending.setdefault(info, []).append((it, audio))
            if len(pending[info]) >= self.batch_items:
                flush(info)
        for sr in list(pending):
            flush(sr)
        self.save_config_lock(save_dir, in_progress=False)
        return kept

    def process_attn_prior(self, item: dict, frames: int, save_dir, overwrite: bool = False) -> list[Path]:
        """The ``attn`` stage (preprocessor.py:672-740): one beta-binomial prior [frames, tokens] (float64, computed on the device
        by evmi_attention_prior_f64) per text representation the item carries -- ``attn/<...>--characters-attn-prior.pt`` for
        ``character_tokens``, ``...--phones-attn-prior.pt`` for ``phone_tokens`` ("/"-joined token strings, as the reference's text
        stage writes them into the filelist; a list of tokens is accepted too).  Existing files are kept unless ``overwrite``."""
        from .heavy import BetaBinomialInterpolator

        written = []
        ids = (item["basename"], item.get("speaker", "default"), item.get("language", "default"))
        for key, name in (("character_tokens", "characters"), ("phone_tokens", "phones")):
            toks = item.get(key)
            if not toks:  # None in datasets of the other representation, "" in multi-source datasets
                continue
            n_tok = len(toks.split("/")) if isinstance(toks, str) else len(toks)
            path = feature_path(save_dir, "attn", *ids, f"{name}-attn-prior.pt")
            if path.exists() and not overwrite:
                continue
            if self._interp is None:
                self._interp = BetaBinomialInterpolator(device=self.device)
            prior = self._interp(frames, n_tok)
            assert tuple(prior.shape) == (frames, n_tok)
            save_tensor(prior, path)
            written.append(path)
            if len(self._interp._cache) > 256:
                self._interp._cache.clear()
        return written

    @staticmethod
    def write_filelist(items: list[dict], path, fields=("basename", "speaker", "language", "characters", "character_tokens", "phone_tokens")) -> Path:
        """The processed filelist (``<save_dir>/<name>.psv``, preprocessor.py:1113-1180: text lives in the filelist, not on disk):
        pipe-separated with a header line, the columns the items actually carry."""
        path = Path(path)
        path.parent.mkdir(parents=True, exist_ok=True)
        cols = [f for f in fields if any(it.get(f) not in (None, "") for it in items)] or ["basename"]
        with open(path, "w", encoding="utf8", newline="") as f:
            f.write("|".join(cols) + "\n")
            for it in items:
                vals = [it.get(c, "") for c in cols]
                vals = ["/".join(v) if isinstance(v, (list, tuple)) else ("" if v is None else str(v)) for v in vals]
                f.write("|".join(v.replace("\\", "\\\\").replace("|", "\\|") for v in vals) + "\n")
        return path

    # -- dataset statistics (preprocessor.py:378-490) -------------------------------------------------------------------
    def compute_stats(self, save_dir, energy: bool = True, pitch: bool = True):
        """(energy Scaler, pitch Scaler) over every saved ``energy/*energy*`` / ``pitch/*pitch*`` tensor (None where disabled)."""
        out = []
        for kind, on in (("energy", energy), ("pitch", pitch)):
            sc = None
            if on:
                sc = Scaler()
                for path in sorted(glob(str(Path(save_dir) / f"{kind}/**/*{kind}*"), recursive=True)):
                    sc.append(torch.load(path, weights_only=True).to(self.device))
            out.append(sc)
        return tuple(out)

    def normalize_stats(self, save_dir, energy_scaler, pitch_scaler) -> dict:
        """Standardise every saved energy / pitch tensor with the dataset statistics and return them as the ``stats`` dict that
        becomes FastSpeech2's ``Stats(pitch=StatsInfo(...), energy=StatsInfo(...))`` (tests/model_stubs.py:50-57)."""
        stats = {}
        for kind, sc in (("energy", energy_scaler), ("pitch", pitch_scaler)):
            if not sc or not len(sc):
                continue
            st = sc.calculate_stats()
            for path in sorted(glob(str(Path(save_dir) / f"{kind}/**/*{kind}*"), recursive=True)):
                save_tensor(sc.normalize(torch.load(path, weights_only=True).to(self.device)), path)
            stats[kind] = st
        return stats


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


class PredictionWriter:
    """A prediction writer of the reference's synthesis (``base_cli/prediction_writing_callback.py:14-41``): one per output format;
    ``get_filename(basename, speaker, language)`` = ``<save_dir>/<basename>--<speaker>--<language>[--ckpt=<step>]--<file_extension>``,
    ``last_file_written`` is what the demo hands back (``demo/app.py:107-108``)."""

    def __init__(self, save_dir: Path, file_extension: str, global_step: int = 0, include_global_step_in_filename: bool = False):
        self.file_extension = file_extension
        self.global_step = f"ckpt={global_step}"
        self.save_dir = Path(save_dir)
        self.sep = SEP
        self.include_global_step_in_filename = include_global_step_in_filename
        self.save_dir.mkdir(parents=True, exist_ok=True)
        self.last_file_written = None

    def get_filename(self, basename: str, speaker: str, language: str) -> str:
        parts = [basename, speaker, language, self.file_extension]
        if self.include_global_step_in_filename:
            parts.insert(-1, self.global_step)
        path = self.save_dir / self.sep.join(parts)
        path.parent.mkdir(parents=True, exist_ok=True)
        return str(path)


def synthesize_helper(model, texts: list, language: str | None, speaker: str | None, duration_control: float | None, global_step: int, output_type,
                      text_representation=None, accelerator: str = "auto", devices: str = "1", device=None, batch_size: int = 16, num_workers: int = 0,
                      filelist=None, filelist_data=None, output_dir: Path = Path("synthesis_output"), teacher_forcing_directory: Path | None = None,
                      vocoder_model=None, vocoder_config=None, vocoder_global_step: int | None = None, style_reference=None, return_scores: bool = False,
                      text_to_ids=None):
    """``fs2.cli.synthesize.synthesize_helper`` (call site ``everyvoice/demo/app.py:84-106``, same keyword names) for the path this
    library accelerates: texts -> symbol ids -> FastSpeech2 -> (vocoder) -> files through per-format writers.
    Returns ``(config, device, predictions, callbacks)`` with ``callbacks`` keyed by output format ("wav", "spec").

    ``texts`` are strings mapped to ids by ``text_to_ids`` (the reference's TextProcessor: CPU string work, out of scope) or already
    lists / tensors of symbol ids; ``filelist_data`` rows (dicts with basename / ids / speaker / language) replace ``texts``.
    ``teacher_forcing_directory``: read ``duration/<basename>--<speaker>--<language>--duration.pt`` from there and synthesise with
    those durations -- how the spectrograms for vocoder matching are produced (docs/guides/finetune.md:18-43)."""
    formats = [getattr(f, "value", f) for f in (output_type if isinstance(output_type, (list, tuple)) else [output_type])]
    unsupported = [f for f in formats if f not in ("wav", "spec")]
    if unsupported:
        raise NotImplementedError(f"output formats {unsupported}: textgrid / readalong outputs belong to the reference's text front-end (out of scope)")
    dev = torch.device(device) if device is not None else model.device
    output_dir = Path(output_dir)
    sr = vocoder_config.preprocessing.audio.output_sampling_rate if vocoder_config is not None else 22050
    in_sr = vocoder_config.preprocessing.audio.input_sampling_rate if vocoder_config is not None else 22050
    hop_out = (vocoder_config.preprocessing.audio.fft_hop_size * (sr // in_sr)) if vocoder_config is not None else 256
    callbacks = {}
    if "wav" in formats:
        if vocoder_model is None:
            raise ValueError("output_type 'wav' needs a vocoder_model")
        callbacks["wav"] = PredictionWriter(output_dir / "wav", "pred.wav", vocoder_global_step or 0, include_global_step_in_filename=False)
    if "spec" in formats:
        callbacks["spec"] = PredictionWriter(output_dir / "synthesized_spec", f"spec-pred-{in_sr}-mel-librosa.pt", global_step)
    rows = []
    if filelist_data is not None:
        rows = [dict(r) for r in filelist_data]
    else:
        for i, t in enumerate(texts):
            ids = text_to_ids(t) if isinstance(t, str) else t
            if isinstance(t, str) and text_to_ids is None:
                raise ValueError("synthesize_helper: pass text_to_ids (text -> symbol ids) or symbol ids; text processing is host-side string work")
            rows.append({"basename": f"utt-{i:04d}" if not isinstance(t, str) else "".join(c if c.isalnum() else "-" for c in t)[:20] or f"utt-{i:04d}",
                         "ids": ids, "speaker": speaker, "language": language})
    predictions = []
    for lo in range(0, len(rows), batch_size):
        chunk = rows[lo : lo + batch_size]
        lens = torch.tensor([len(r["ids"]) for r in chunk])
        ids = torch.zeros(len(chunk), int(lens.max()), dtype=torch.long)
        for j, r in enumerate(chunk):
            ids[j, : lens[j]] = torch.as_tensor(r["ids"], dtype=torch.long)
        kw = {}
        spk = [r.get("speaker") or speaker or "default" for r in chunk]
        lang = [r.get("language") or language or "default" for r in chunk]
        if getattr(model, "speaker2id", None) and getattr(model.config, "multispeaker", False):
            kw["speakers"] = torch.tensor([model.speaker2id[s] for s in spk])
        if getattr(model, "lang2id", None) and getattr(model.config, "multilingual", False):
            kw["languages"] = torch.tensor([model.lang2id[x] for x in lang])
        if teacher_forcing_directory is not None:
            durs = torch.zeros_like(ids)
            for j, r in enumerate(chunk):
                d = torch.load(Path(teacher_forcing_directory) / "duration" / SEP.join([r["basename"], spk[j], lang[j], "duration.pt"]), weights_only=True)
                durs[j, : lens[j]] = d[: lens[j]].long()
            kw["durations"] = durs
        _, post, durations, _, _, mel_lens = model(ids, lens, duration_control=1.0 if duration_control is None else duration_control, **kw)
        wav = vocoder_model(post.transpose(1, 2).contiguous()) if "wav" in formats else None
        for j, r in enumerate(chunk):
            T = int(mel_lens[j])
            rec = {"basename": r["basename"], "speaker": spk[j], "language": lang[j], "frames": T, "durations": durations[j, : int(lens[j])].cpu()}
            if "spec" in formats:
                path = callbacks["spec"].get_filename(r["basename"], spk[j], lang[j])
                save_tensor(post[j, :T].transpose(0, 1).contiguous(), path)
                callbacks["spec"].last_file_written = rec["spec"] = path
            if wav is not None:
                path = callbacks["wav"].get_filename(r["basename"], spk[j], lang[j])
                save_wav(wav[j, 0, : T * hop_out], path, sr)
                callbacks["wav"].last_file_written = rec["wav"] = path
            predictions.append(rec)
    return model.config, dev, predictions, callbacks


def generate_teacher_forced_specs(model, filelist_data: list[dict], preprocessed_dir, global_step: int = 0, batch_size: int = 16):
    """Vocoder matching, step 1 (docs/guides/finetune.md:18-43: ``everyvoice synthesize from-text ... -O spec
    --teacher-forcing-directory <preprocessed>``): the feature-prediction network's spectrogram of every training utterance under its
    ground-truth durations, written next to the real features as ``synthesized_spec/...--spec-pred-<sr>-mel-librosa.pt`` -- what
    ``training.finetune: true`` then feeds the vocoder (dataset.SpecDataset)."""
    _, _, preds, callbacks = synthesize_helper(model, [], None, None, 1.0, global_step, ["spec"], filelist_data=filelist_data,
                                               output_dir=Path(preprocessed_dir), teacher_forcing_directory=Path(preprocessed_dir), batch_size=batch_size)
    return preds
