"""Host-side data formats around the path (CPU) and the end-to-end preprocess -> copy-synthesis chain (GPU)."""

import wave

import numpy as np
import pytest
import torch

from everyvoice_amd.config import AudioConfig
from everyvoice_amd import pipeline
from oracle import heavy_ref


def _write_wav(path, x, sr=22050, ch=1):
    pcm = np.clip(np.round(np.asarray(x) * 32767), -32768, 32767).astype("<i2")
    with wave.open(str(path), "wb") as w:
        w.setnchannels(ch)
        w.setsampwidth(2)
        w.setframerate(sr)
        w.writeframes(pcm.tobytes())


def test_wav_round_trip_and_host_gates(tmp_path, golden_dir):
    g = np.load(golden_dir / "mel_anchor.npz")
    pcm = g["pcm"]
    p = tmp_path / "a.wav"
    _write_wav(p, pcm.astype(np.float32) / 32767.0)
    audio, sr, sec = pipeline.load_wav(p)
    assert sr == 22050 and audio.shape == (1, len(pcm)) and abs(sec - len(pcm) / 22050) < 1e-9
    assert np.array_equal(np.round(audio[0].numpy() * 32768).astype(np.int16), pcm)  # bit-exact PCM decode
    cfg = AudioConfig()
    got, sr2 = pipeline.gate_audio(p, cfg)
    assert sr2 == 22050 and torch.equal(got, audio)
    pipeline.save_wav(audio[0] * 0.5, tmp_path / "sub" / "o.wav", sr2)
    back, _, _ = pipeline.load_wav(tmp_path / "sub" / "o.wav")
    assert float((back[0] - audio[0] * 0.5).abs().max()) <= 1.0 / 32768 + 1e-7
    # host gates (test_preprocessing.py:109-160): too short, too long, > 2 channels
    _write_wav(tmp_path / "short.wav", np.zeros(2000) + 0.1)
    assert pipeline.gate_audio(tmp_path / "short.wav", cfg) == (None, "audio_too_short")
    _write_wav(tmp_path / "long.wav", np.zeros(22050 * 12) + 0.1)
    assert pipeline.gate_audio(tmp_path / "long.wav", cfg) == (None, "audio_too_long")
    _write_wav(tmp_path / "multi.wav", np.zeros(22050 * 4) + 0.1, ch=4)
    assert pipeline.gate_audio(tmp_path / "multi.wav", cfg) == (None, "multichannel_files")


def test_resample_filter_bank_and_config_lock(tmp_path):
    from oracle.preprocess_ref import sinc_kernel_ref

    for orig, new in ((44100, 22050), (48000, 22050), (16000, 22050)):
        k, width, o, n = pipeline.sinc_resample_kernel(orig, new)
        kr, wr, o_r, n_r = sinc_kernel_ref(orig, new)
        assert (width, o, n) == (wr, o_r, n_r) and k.shape == kr.shape == (n, 1, 2 * width + o)
        torch.testing.assert_close(k, kr, rtol=0, atol=1e-7)
    # .config-lock (preprocessor.py:974-1082): written read-only, "in progress" while running; a later run with another audio
    # configuration, or after an interrupted run, is refused
    pre = pipeline.GpuPreprocessor(AudioConfig(), device="cpu")
    assert not pre.config_lock_has_conflicts(tmp_path)
    pre.save_config_lock(tmp_path, in_progress=True)
    lock = tmp_path / ".config-lock"
    import json
    import stat

    saved = json.loads(lock.read_text())
    assert saved["status"] == "in progress" and saved["preprocessing.audio"]["n_fft"] == 1024 and "Do not edit" in saved["info"]
    assert not (lock.stat().st_mode & stat.S_IWUSR)
    assert pre.config_lock_has_conflicts(tmp_path)  # interrupted run
    pre.save_config_lock(tmp_path, in_progress=False)
    assert not pre.config_lock_has_conflicts(tmp_path)
    other = pipeline.GpuPreprocessor(AudioConfig(n_fft=2048, fft_window_size=2048, fft_hop_size=512), device="cpu")
    assert other.config_lock_has_conflicts(tmp_path)
    with pytest.raises(pipeline.ConfigLockMismatch):
        other.process([], tmp_path)


def test_paths_and_phone_average(tmp_path):
    p = pipeline.feature_path(tmp_path, "spec", "LJ010-0008", "default", "default", "spec-22050-mel-librosa.pt")
    assert p == tmp_path / "spec" / "LJ010-0008--default--default--spec-22050-mel-librosa.pt"
    data = torch.arange(10, dtype=torch.float32)
    d = [3, 0, 2, 5]
    got = pipeline.average_data_by_durations(data, d)
    assert np.allclose(got.numpy(), heavy_ref.average_by_durations_ref(data.numpy(), d))
    assert got[1] == pytest.approx(1e-7)


@pytest.mark.gpu
def test_preprocess_then_copy_synthesis(tmp_path, golden_dir, cuda_device):
    from everyvoice_amd.config import HiFiGANConfig
    from everyvoice_amd.vocoder import HiFiGANGenerator
    from oracle import mel_ref

    g = np.load(golden_dir / "mel_anchor.npz")
    wav = tmp_path / "LJ010-0008.wav"
    _write_wav(wav, g["pcm"].astype(np.float32) / 32767.0)
    pre = pipeline.GpuPreprocessor(device=cuda_device)
    kept = pre.process([dict(basename="LJ010-0008", speaker="default", language="default", wav=wav)], tmp_path / "pre")
    assert len(kept) == 1 and pre.counters["processed_files"] == 1
    spec = torch.load(tmp_path / "pre" / "spec" / "LJ010-0008--default--default--spec-22050-mel-librosa.pt")
    energy = torch.load(tmp_path / "pre" / "energy" / "LJ010-0008--default--default--energy.pt")
    audio, _, _ = pipeline.load_wav(tmp_path / "pre" / "audio" / "LJ010-0008--default--default--audio-22050.wav")
    assert spec.shape == (80, kept[0]["frames"]) and energy.shape == (kept[0]["frames"],)
    assert audio.shape[1] == kept[0]["samples"] == kept[0]["frames"] * 256
    want = mel_ref.mel_spectrogram_ref(pipeline.process_audio(wav, AudioConfig())[0], truncate=True)
    assert float((spec - want).abs().max()) <= 2e-3
    np.testing.assert_allclose(energy.numpy(), heavy_ref.energy_ref(want.numpy()), rtol=2e-4)
    # copy synthesis (everyvoice synthesize from-spec): spec file -> wav file with the reference's naming
    torch.manual_seed(0)
    voc = HiFiGANGenerator(HiFiGANConfig()).to(cuda_device)
    out = pipeline.synthesize_from_spec(spec, voc, tmp_path / "synth", "LJ010-0008")
    assert out.name == "LJ010-0008--default--default--pred.wav"
    y, sr, _ = pipeline.load_wav(out)
    assert sr == 22050 and y.shape == (1, spec.shape[1] * 256)


@pytest.mark.gpu
def test_process_audio_gates_and_normalisation(tmp_path, golden_dir, cuda_device):
    g = np.load(golden_dir / "mel_anchor.npz")
    pcm = g["pcm"]
    p = tmp_path / "a.wav"
    _write_wav(p, pcm.astype(np.float32) / 32767.0)
    cfg = AudioConfig()
    out, sr2 = pipeline.process_audio(p, cfg, device=cuda_device)
    assert out.numel() % 256 == 0 and len(pcm) - out.numel() < 256  # test_preprocessing.py:356-383 invariant
    assert abs(float(out.abs().max()) - 0.95) < 1e-6 and sr2 == 22050
    _write_wav(tmp_path / "zeros.wav", np.zeros(22050))
    assert pipeline.process_audio(tmp_path / "zeros.wav", cfg, device=cuda_device) == (None, "audio_empty")
    _write_wav(tmp_path / "quiet.wav", 1e-3 * np.sin(np.arange(22050) * 0.1))  # about -63 LUFS: below the -36 gate
    assert pipeline.process_audio(tmp_path / "quiet.wav", cfg, device=cuda_device) == (None, "audio_empty")


@pytest.mark.gpu
def test_loudness_and_resampling_match_the_restated_torchaudio(cuda_device):
    from oracle.preprocess_ref import loudness_ref, resample_ref

    gen = torch.Generator().manual_seed(5)
    items = [0.3 * torch.tanh(torch.randn(1, 30000, generator=gen)), 0.02 * torch.randn(1, 22050, generator=gen),
             torch.sin(torch.arange(40000) * 0.05)[None] * torch.linspace(0, 1, 40000)[None]]
    t_max = max(x.shape[1] for x in items)
    batch = torch.zeros(len(items), 1, t_max)
    for i, x in enumerate(items):
        batch[i, :, : x.shape[1]] = x
    got = pipeline.loudness(batch.to(cuda_device), torch.tensor([x.shape[1] for x in items]), 22050).cpu()
    for i, x in enumerate(items):
        assert float(got[i]) == pytest.approx(loudness_ref(x, 22050), abs=2e-3), i  # dB
    stereo = torch.stack([items[0][0, :22050], 0.5 * items[1][0]])[None]
    got2 = pipeline.loudness(stereo.to(cuda_device), torch.tensor([22050]), 22050).cpu()
    assert float(got2[0]) == pytest.approx(loudness_ref(stereo[0], 22050), abs=2e-3)
    for orig, new in ((44100, 22050), (48000, 22050), (16000, 22050)):
        x = 0.3 * torch.tanh(torch.randn(2, 12345, generator=gen))
        want = resample_ref(x, orig, new)
        out = pipeline.resample(x.to(cuda_device), orig, new).cpu()
        assert out.shape == want.shape
        torch.testing.assert_close(out, want, rtol=0, atol=2e-6)


@pytest.mark.gpu
def test_batched_preprocessing_equals_one_by_one_and_dataset_statistics(tmp_path, cuda_device):
    """N2: several utterances of different lengths (and one at 44.1 kHz, one stereo) per launch give exactly the files the
    one-utterance path writes; then compute_stats / normalize_stats (A8) over the energy files."""
    from oracle import mel_ref

    gen = torch.Generator().manual_seed(9)
    items = []
    for i, (n, sr, ch) in enumerate([(30000, 22050, 1), (12000, 22050, 1), (45000, 22050, 2), (50000, 44100, 1), (9000, 22050, 1)]):
        x = 0.3 * torch.tanh(torch.randn(n * ch, generator=gen)).numpy()
        _write_wav(tmp_path / f"u{i}.wav", x, sr=sr, ch=ch)
        items.append(dict(basename=f"u{i}", speaker="default", language="default", wav=tmp_path / f"u{i}.wav"))
    _write_wav(tmp_path / "silent.wav", np.zeros(22050))
    items.append(dict(basename="silent", speaker="default", language="default", wav=tmp_path / "silent.wav"))
    batched = pipeline.GpuPreprocessor(device=cuda_device, batch_items=8)
    kept = batched.process(items, tmp_path / "b")
    single = pipeline.GpuPreprocessor(device=cuda_device, batch_items=1)
    kept1 = single.process(items, tmp_path / "s")
    assert [k["basename"] for k in kept] and sorted(k["basename"] for k in kept) == sorted(k["basename"] for k in kept1) == [f"u{i}" for i in range(5)]
    assert batched.counters == single.counters and batched.counters["audio_empty"] == 1 and batched.counters["processed_files"] == 5
    for k in kept:
        for kind, fn in (("spec", "spec-22050-mel-librosa.pt"), ("energy", "energy.pt"), ("pitch", "pitch.pt")):
            a = torch.load(tmp_path / "b" / kind / f"{k['basename']}--default--default--{fn}")
            b = torch.load(tmp_path / "s" / kind / f"{k['basename']}--default--default--{fn}")
            assert torch.equal(a, b), (k["basename"], kind)
        wa, _, _ = pipeline.load_wav(tmp_path / "b" / "audio" / f"{k['basename']}--default--default--audio-22050.wav")
        wb, _, _ = pipeline.load_wav(tmp_path / "s" / "audio" / f"{k['basename']}--default--default--audio-22050.wav")
        assert torch.equal(wa, wb) and wa.shape[1] == k["samples"] == k["frames"] * 256
        spec = torch.load(tmp_path / "b" / "spec" / f"{k['basename']}--default--default--spec-22050-mel-librosa.pt")
        assert float((spec - mel_ref.mel_spectrogram_ref(wa[0], truncate=True)).abs().max()) <= 3e-3  # (the saved wav is PCM-16 quantised)
    assert (tmp_path / "b" / ".config-lock").exists()
    # dataset statistics of the energy files, then their standardisation in place
    es, ps = batched.compute_stats(tmp_path / "b", pitch=True)
    assert len(es) == 5 and len(ps) == 5  # pitch/<...>--pitch.pt is written next to energy (one value per frame)
    assert torch.load(tmp_path / "b" / "pitch" / "u0--default--default--pitch.pt").shape == torch.load(tmp_path / "b" / "energy" / "u0--default--default--energy.pt").shape
    allv = torch.cat([torch.load(p) for p in sorted((tmp_path / "b" / "energy").iterdir())])
    stats = batched.normalize_stats(tmp_path / "b", es, ps)
    assert set(stats) == {"energy", "pitch"} and stats["energy"]["sample_size"] == 5
    assert stats["energy"]["mean"] == pytest.approx(float(allv.mean()), rel=1e-5) and stats["energy"]["std"] == pytest.approx(float(allv.std()), rel=1e-5)
    normed = torch.cat([torch.load(p) for p in sorted((tmp_path / "b" / "energy").iterdir())])
    assert abs(float(normed.mean())) < 1e-4 and float(normed.std()) == pytest.approx(1.0, rel=1e-4)
    assert stats["energy"]["norm_min"] == pytest.approx(float(normed.min()), rel=1e-4)


@pytest.mark.gpu
@pytest.mark.parametrize("spec_type", ["linear", "mel"])
def test_preprocessing_with_the_other_spec_types(tmp_path, cuda_device, spec_type):
    """preprocessing.audio.spec_type "linear" / "mel" (everyvoice/utils/heavy.py:59-68, 101-107): the files carry the type in their
    name, hold log(clamp(transform(audio), 1e-5)) truncated to samples // hop frames, and energy is its norm over the bins."""
    from oracle import mel_ref

    gen = torch.Generator().manual_seed(4)
    items = []
    for i, n in enumerate([20000, 9000]):
        _write_wav(tmp_path / f"v{i}.wav", 0.3 * torch.tanh(torch.randn(n, generator=gen)).numpy())
        items.append(dict(basename=f"v{i}", speaker="default", language="default", wav=tmp_path / f"v{i}.wav"))
    pre = pipeline.GpuPreprocessor(AudioConfig(spec_type=spec_type), device=cuda_device, batch_items=4, pitch=False)
    kept = pre.process(items, tmp_path / "o")
    assert len(kept) == 2
    for k in kept:
        wa, _, _ = pipeline.load_wav(tmp_path / "o" / "audio" / f"{k['basename']}--default--default--audio-22050.wav")
        spec = torch.load(tmp_path / "o" / "spec" / f"{k['basename']}--default--default--spec-22050-{spec_type}.pt")
        if spec_type == "linear":
            want = mel_ref.spectrogram_ref(wa[0], 1024, 1024, 256, 2.0)
        else:
            want = mel_ref.torchaudio_mel_ref(wa[0], 22050, 1024, 1024, 256, 80, 0.0, 8000.0)
        want = torch.log(torch.clamp(want, min=1e-5))[:, : k["frames"]]
        assert spec.shape == want.shape == ((513 if spec_type == "linear" else 80), k["frames"])
        # compared before the logarithm, relative to the largest bin (the fp32 DFT's error is absolute: the log of a bin 60 dB below
        # the peak moves by 1e-2), and in the log domain on the bins within 30 dB of it
        assert float((spec.exp() - want.exp()).abs().max()) <= 1e-4 * float(want.exp().max())
        loud = want > want.max() - 7.0
        assert float((spec - want)[loud].abs().max()) <= 2e-3
        energy = torch.load(tmp_path / "o" / "energy" / f"{k['basename']}--default--default--energy.pt")
        torch.testing.assert_close(energy, torch.linalg.norm(spec, dim=0), rtol=1e-5, atol=1e-5)
    with pytest.raises(ValueError):
        pipeline.GpuPreprocessor(AudioConfig(spec_type="raw"), device=cuda_device)


@pytest.mark.gpu
def test_autocorrelation_pitch_tracker_on_known_tones_and_against_its_own_restatement(cuda_device):
    """extract_pitch(estimator="acf") is NOT the reference's estimator (pyworld dio + stonemask, preprocessor.py:244-285; that one is
    estimator="world", the default since round 5 -- tests below): it is this library's own normalised-autocorrelation tracker behind
    the reference's interface, kept selectable.  What this test establishes is therefore limited to:
    the interface (one value per hop; unvoiced stretches interpolated across; an utterance without voicing -> zeros,
    preprocessor.py:277-283), known fundamentals of synthetic harmonic tones within 1 %, and that the device kernel computes what
    its numpy restatement (oracle/preprocess_ref.py:pitch_acf_ref -- a SELF-comparison, not a pin against the reference) computes.
    The comparison with reference-side data is the speech anchor below."""
    from oracle.preprocess_ref import pitch_acf_ref

    sr, hop = 22050, 256
    t = torch.arange(sr, dtype=torch.float32) / sr
    tone = lambda f: sum(torch.sin(2 * np.pi * f * k * t) / k for k in (1, 2, 3, 4)) * 0.2  # noqa: E731
    gen = torch.Generator().manual_seed(0)
    glide = 0.2 * torch.sin(2 * np.pi * torch.cumsum(120.0 + 80.0 * t, 0) / sr)
    noise = 0.05 * torch.randn(sr, generator=gen)
    mixed = torch.cat([tone(220.0)[: sr // 3], torch.zeros(sr // 3), tone(330.0)[: sr - 2 * (sr // 3)]])
    batch = torch.stack([tone(110.0), tone(440.0), glide, noise, mixed])
    lens = torch.tensor([sr, sr, sr, sr, sr])
    raw = pipeline.extract_pitch(batch.to(cuda_device), lens, hop, sr, interpolate=False, estimator="acf").cpu()
    assert raw.shape == (5, sr // hop + 1)
    mid = slice(6, raw.shape[1] - 6)
    assert float((raw[0, mid] - 110.0).abs().max()) < 1.1 and float((raw[1, mid] - 440.0).abs().max()) < 4.4
    want_glide = (120.0 + 80.0 * torch.arange(raw.shape[1]) * hop / sr)
    assert float(((raw[2, mid] - want_glide[mid]).abs() / want_glide[mid]).max()) < 0.02
    assert float((raw[3] > 0).float().mean()) < 0.1  # white noise: (almost) no frame is voiced
    for i in (0, 2, 4):
        ref = pitch_acf_ref(batch[i].numpy(), hop, sr)
        voiced = (ref > 0) & (raw[i].numpy() > 0)
        assert (ref > 0).sum() == (raw[i].numpy() > 0).sum() and np.abs(ref[voiced] - raw[i].numpy()[voiced]).max() < 0.05
    out = pipeline.extract_pitch(batch.to(cuda_device), lens, hop, sr, estimator="acf").cpu()
    assert float(out[4].min()) > 200.0  # the silent third is bridged between 220 and 330 Hz, as the reference's _interpolate does
    gap = out[4, 30:57]
    assert bool(((gap[1:] - gap[:-1]) >= -1e-3).all()) and 220.0 <= float(gap.min()) and float(gap.max()) <= 331.0
    assert float(pipeline.extract_pitch(torch.zeros(1, sr, device=cuda_device), None, hop, sr, estimator="acf").abs().max()) == 0.0


def _phone_average(f0, durs):
    out, p = [], 0
    for d in durs:
        out.append(float(np.mean(f0[p:p + d])) if d > 0 else 0.0)
        p += int(d)
    return np.array(out)


@pytest.mark.gpu
def test_world_pitch_on_the_device_reproduces_the_reference_pyworld_fixture_and_its_oracle(cuda_device, golden_dir):
    """SURVEY.md 8a A7, the reference's own estimator: extract_pitch(estimator="world") = WORLD's DIO + StoneMask in float64 on the device
    (csrc/pitch_world.hip) for pyworld.dio(x.f64, fs, frame_period = hop / fs * 1000, speed) -> pyworld.stonemask
    (everyvoice/preprocessor/preprocessor.py:244-285).
    (a) A pin on reference-side data: LJ010-0008 (the reference's test wav) at speed 1 -- the setting of the codebase that wrote the
        reference's fixture everyvoice/tests/data/ming024/eng-LJSpeech-pitch-LJ010-0008.npy (pyworld dio + stonemask, unvoiced frames
        interpolated, averaged per phone with the fixture's durations, standardised with dataset statistics) -- must BE that fixture
        up to the standardisation's line: Pearson r >= 0.999999 and every one of the 67 phones within 2e-3 Hz of a * fixture + b
        (the oracle: 1e-13 Hz; the device result is stored as fp32: 1.5e-5 Hz per ulp at 250 Hz, averaged).
    (b) Against oracle/pitch_world_ref.py frame by frame at speed 1 and at the reference's speed 4 (the decimated path the fixture
        does not cover): the same frames voiced, values within 1e-5 relative."""
    from oracle.pitch_world_ref import dio, stonemask

    g = np.load(golden_dir / "data_side.npz")
    pcm = np.load(golden_dir / "mel_anchor.npz")["pcm"].astype(np.float32) / 32768.0
    durs, want = g["ming024_duration"], g["ming024_pitch"].astype(np.float64)
    hop, sr = 256, 22050
    x = torch.from_numpy(pcm)[None].to(cuda_device)
    for speed in (1, 4):
        raw = pipeline.extract_pitch(x, None, hop, sr, interpolate=False, speed=speed).cpu()[0].numpy().astype(np.float64)
        f0, t = dio(pcm.astype(np.float64), sr, frame_period=hop / sr * 1000.0, speed=speed)
        ref = stonemask(pcm.astype(np.float64), f0, t, sr)
        assert raw.shape == ref.shape == (pipeline.world_frames(len(pcm), hop, sr),)
        assert np.array_equal(raw > 0, ref > 0), (speed, int((raw > 0).sum()), int((ref > 0).sum()))
        v = ref > 0
        assert v.sum() > 250 and np.abs(raw[v] - ref[v]).max() <= 1e-5 * ref[v].max(), (speed, np.abs(raw[v] - ref[v]).max())
    full = pipeline.extract_pitch(x, None, hop, sr, speed=1).cpu()[0].numpy().astype(np.float64)
    frames = int(durs.sum())
    assert durs.shape[0] == 67 and frames == 497 and full.shape[0] >= frames
    got = _phone_average(full[:frames], durs)
    ok = durs > 0
    r = np.corrcoef(got[ok], want[ok])[0, 1]
    a, b = np.polyfit(want[ok], got[ok], 1)  # Hz = a * standardised + b: the dataset's pitch std and mean (46.75 / 207.62 Hz)
    res = np.abs(a * want[ok] + b - got[ok]).max()
    print(f"WORLD pitch on the device vs the reference's pyworld fixture: r = {r:.9f}, max residual {res:.2e} Hz (std {a:.2f}, mean {b:.2f} Hz)")
    assert r >= 0.999999 and res <= 2e-3 and 40.0 < a < 55.0 and 200.0 < b < 215.0


@pytest.mark.gpu
def test_world_pitch_batches_ragged_lengths_tones_and_silence(cuda_device):
    """A zero-padded batch of different lengths = every item alone (bitwise: items do not interact); known fundamentals of harmonic tones
    within 1 %; white noise (almost) unvoiced; silence -> zeros (preprocessor.py:277-283); frame counts follow pyworld's formula."""
    sr, hop = 22050, 256
    t = torch.arange(sr, dtype=torch.float32) / sr
    tone = lambda f: sum(torch.sin(2 * np.pi * f * k * t) / k for k in (1, 2, 3, 4)) * 0.2  # noqa: E731
    gen = torch.Generator().manual_seed(0)
    noise = 0.05 * torch.randn(sr, generator=gen)
    lens = torch.tensor([sr, sr - 4001, sr - 777, sr])
    batch = torch.stack([tone(110.0), tone(440.0), tone(220.0), noise])
    for i, n in enumerate(lens.tolist()):
        batch[i, n:] = 0.0
    raw = pipeline.extract_pitch(batch.to(cuda_device), lens, hop, sr, interpolate=False).cpu()
    assert raw.shape == (4, pipeline.world_frames(sr, hop, sr))
    for i, f in ((0, 110.0), (1, 440.0), (2, 220.0)):
        n = pipeline.world_frames(int(lens[i]), hop, sr)
        mid = raw[i, 8 : n - 8]
        assert float((mid > 0).float().mean()) > 0.95 and float((mid[mid > 0] - f).abs().max()) < 0.01 * f, (i, mid)
        assert float(raw[i, n:].abs().max() if n < raw.shape[1] else 0.0) == 0.0
        alone = pipeline.extract_pitch(batch[i : i + 1, : int(lens[i])].contiguous().to(cuda_device), None, hop, sr, interpolate=False).cpu()[0]
        assert torch.equal(alone, raw[i, : alone.shape[0]]), i
    assert float((raw[3] > 0).float().mean()) < 0.2
    assert float(pipeline.extract_pitch(torch.zeros(1, sr, device=cuda_device), None, hop, sr).abs().max()) == 0.0


@pytest.mark.gpu
def test_pitch_tracker_speech_anchor_against_the_reference_fixture(cuda_device, golden_dir):
    """The autocorrelation tracker (estimator="acf") on real speech -- a sanity anchor, not a pin: LJ010-0008 through extract_pitch, averaged per phone
    with the durations of the reference's ming024 fixture (everyvoice/tests/data/ming024/*-duration-*.npy, 67 phones, 497
    frames), against that fixture's phone-level pitch array -- which is another codebase's pyworld track, standardised with
    LJSpeech statistics.  Different estimator, so the tolerance is loose and on shape only: over the phones both call voiced the
    two contours must correlate (Pearson r >= 0.8) and, after undoing the standardisation with a least-squares line, agree within
    12 % in the median."""
    g = np.load(golden_dir / "data_side.npz")
    pcm = np.load(golden_dir / "mel_anchor.npz")["pcm"].astype(np.float32) / 32768.0
    durs, want = torch.from_numpy(g["ming024_duration"]), g["ming024_pitch"].astype(np.float64)
    hop, sr = 256, 22050
    f0 = pipeline.extract_pitch(torch.from_numpy(pcm)[None].to(cuda_device), None, hop, sr, estimator="acf").cpu()[0]
    frames = int(durs.sum())
    assert abs(f0.shape[0] - frames) <= 8 and durs.shape[0] == 67 and frames == 497  # (ming024 trims its utterances: a few frames fewer)
    f0 = torch.nn.functional.pad(f0, (0, max(0, frames - f0.shape[0])))[:frames]
    got = pipeline.average_data_by_durations(f0, durs).numpy().astype(np.float64)
    ok = (durs.numpy() > 0) & (got > 60.0)
    assert ok.sum() >= 55
    r = np.corrcoef(got[ok], want[ok])[0, 1]
    a, b = np.polyfit(want[ok], got[ok], 1)  # Hz = a * standardised + b: a ~ the dataset's pitch std, b ~ its mean
    rel = np.abs(a * want[ok] + b - got[ok]) / got[ok]
    print(f"pitch anchor: r = {r:.3f}, implied LJSpeech mean {b:.1f} Hz / std {a:.1f} Hz, median rel. deviation {np.median(rel):.3f}")
    assert r >= 0.8 and 100.0 < b < 300.0 and a > 0 and np.median(rel) <= 0.12
