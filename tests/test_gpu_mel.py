"""STFT/mel front-end on the GPU (fp32 matrix-core DFT) vs the CPU oracle (torch.stft restatement).

Tolerance: |log-mel diff| <= 2e-3 absolute on speech (values span about [-11.5, 2]); the only
differences are fp32 summation order (1024-term DFT as an fmaf chain vs torch's FFT) and logf."""

import numpy as np
import pytest
import torch

from oracle import heavy_ref, mel_ref

pytestmark = pytest.mark.gpu

LOGMEL_ATOL = 2e-3


def test_logmel_of_reference_wav(cuda_device, golden_dir):
    from everyvoice_amd.spectral import MelSpectrogram, extract_energy, extract_spectral_features

    g = np.load(golden_dir / "mel_anchor.npz")
    audio = torch.from_numpy(g["pcm"].astype(np.float32) / 32768.0)
    S = audio.numel() // 256 * 256  # Preprocessor.process_audio truncates to a multiple of hop
    audio = audio[:S]
    want = mel_ref.mel_spectrogram_ref(audio, truncate=True)
    tr = MelSpectrogram()
    got = extract_spectral_features(audio.to(cuda_device), tr).cpu()
    assert got.shape == want.shape == (80, S // 256)  # frames == S // hop (test_preprocessing.py:356-383)
    assert float((got - want).abs().max()) <= LOGMEL_ATOL
    mel, energy = extract_energy(audio.to(cuda_device), tr)
    np.testing.assert_allclose(energy.cpu().numpy(), heavy_ref.energy_ref(want.numpy()), rtol=2e-4)
    # the sanity anchor the reference's test data holds for this utterance (ming024 mel, interior frames)
    ref = g["mel_ming024"]
    n = min(got.shape[1], len(ref))
    assert np.abs(got.numpy().T[4 : n - 4] - ref[4 : n - 4]).max() < 0.02


@pytest.mark.parametrize("B,S", [(1, 8192), (16, 8192), (3, 5000), (2, 22050)])
def test_batched_random_audio(cuda_device, B, S):
    from everyvoice_amd.spectral import MelSpectrogram

    g = torch.Generator().manual_seed(S)
    audio = 0.3 * torch.tanh(torch.randn(B, S, generator=g))  # SURVEY.md §8d C4 signal
    tr = MelSpectrogram()
    got_log, got_mag = tr(audio.to(cuda_device), log=True, return_magnitude=True)
    want_log = mel_ref.mel_spectrogram_ref(audio)
    want_mag = mel_ref.magnitude_spectrogram_ref(audio)
    assert got_log.shape == want_log.shape == (B, 80, 1 + S // 256)
    assert float((got_log.cpu() - want_log).abs().max()) <= LOGMEL_ATOL
    torch.testing.assert_close(got_mag.cpu(), want_mag, rtol=1e-3, atol=2e-4)
    lin = tr(audio.to(cuda_device), log=False).cpu()
    torch.testing.assert_close(torch.log(torch.clamp(lin, min=1e-5)), want_log, rtol=0, atol=LOGMEL_ATOL)


@pytest.mark.parametrize("B,S", [(1, 16384), (3, 11000), (16, 8192)])
def test_config5_front_end_44k_nfft2048_hop512(cuda_device, B, S):
    """BASELINE config 5's STFT contract (AudioConfig with 44.1 kHz, n_fft = win = 2048, hop 512, f_max 8000): the magnitude
    tile of 1025 bins does not fit the LDS next to the audio segment, so the kernel walks the bins in chunks."""
    from everyvoice_amd.spectral import MelSpectrogram

    g = torch.Generator().manual_seed(S)
    audio = 0.3 * torch.tanh(torch.randn(B, S, generator=g))
    tr = MelSpectrogram(2048, 2048, 512, 44100, 80, 0, 8000)
    got_log, got_energy, got_mag = tr(audio.to(cuda_device), log=True, return_energy=True, return_magnitude=True)
    want_log = mel_ref.mel_spectrogram_ref(audio, sr=44100, n_fft=2048, win=2048, hop=512)
    assert got_log.shape == want_log.shape == (B, 80, 1 + S // 512)
    assert float((got_log.cpu() - want_log).abs().max()) <= LOGMEL_ATOL
    torch.testing.assert_close(got_mag.cpu(), mel_ref.magnitude_spectrogram_ref(audio, 2048, 2048, 512), rtol=1e-3, atol=4e-4)
    np.testing.assert_allclose(got_energy.cpu().numpy(), np.linalg.norm(want_log.numpy(), axis=1), rtol=2e-4)


def test_many_mel_bands_on_a_short_fft(cuda_device):
    """n_fft 256 / hop 64 with 128 mel bands: the per-frame log-mel scratch (32 x 129 words) is larger than the audio tile."""
    from everyvoice_amd.spectral import MelSpectrogram

    audio = 0.3 * torch.tanh(torch.randn(2, 5000, generator=torch.Generator().manual_seed(1)))
    tr = MelSpectrogram(256, 256, 64, 22050, 128, 0, 8000)
    got, energy = tr(audio.to(cuda_device), log=True, return_energy=True)
    want = mel_ref.mel_spectrogram_ref(audio, n_fft=256, win=256, hop=64, n_mels=128)
    assert float((got.cpu() - want).abs().max()) <= LOGMEL_ATOL
    np.testing.assert_allclose(energy.cpu().numpy(), np.linalg.norm(want.numpy(), axis=1), rtol=2e-4)


def test_silence_and_edges(cuda_device):
    from everyvoice_amd.spectral import MelSpectrogram

    tr = MelSpectrogram()
    z = tr(torch.zeros(2, 4096, device=cuda_device), log=True).cpu()
    want = mel_ref.mel_spectrogram_ref(torch.zeros(2, 4096))
    torch.testing.assert_close(z, want, rtol=0, atol=1e-4)  # sqrt(1e-9) floor through the mel basis, clamped at 1e-5
    with pytest.raises(RuntimeError):  # reflect padding needs more than n_fft/2 samples, like torch.stft
        tr(torch.zeros(1, 300, device=cuda_device))
    with pytest.raises(RuntimeError, match="GPU only"):
        tr(torch.zeros(1, 4096))


# ---- the other branches of get_spectral_transform (everyvoice/utils/heavy.py:59-68, 101-118) ----------------------------------------
# Tolerance: the DFT is a fp32 GEMM with a 1024-term (or shorter) dot product per bin where torch runs an FFT: relative to the largest
# value of the tensor 2e-5; power spectra square that.
SPEC_CASES = [(1024, 1024, 256, (2, 8192)), (512, 400, 100, (3, 5000)), (400, 400, 200, (1, 2, 4000)), (2048, 1200, 240, (2, 9600))]


@pytest.mark.parametrize("n_fft,win,hop,shape", SPEC_CASES)
def test_linear_and_raw_spectrograms_match_torch_stft(cuda_device, n_fft, win, hop, shape):
    from everyvoice_amd.spectral import get_spectral_transform

    audio = torch.randn(*shape, generator=torch.Generator().manual_seed(n_fft + hop)) * 0.3
    lin = get_spectral_transform("linear", n_fft, win, hop)(audio.to(cuda_device)).cpu()
    want = mel_ref.spectrogram_ref(audio, n_fft, win, hop, 2.0)
    assert lin.shape == want.shape == shape[:-1] + (n_fft // 2 + 1, 1 + shape[-1] // hop)
    assert float((lin - want).abs().max()) <= 1e-4 * float(want.max())
    raw = get_spectral_transform("raw", n_fft, win, hop)(audio.to(cuda_device)).cpu()
    want_c = mel_ref.spectrogram_ref(audio, n_fft, win, hop, None)
    assert raw.dtype == torch.complex64 and raw.shape == want_c.shape
    assert float((raw - want_c).abs().max()) <= 2e-5 * float(want_c.abs().max())


def test_torchaudio_mel_branch(cuda_device):
    from everyvoice_amd.spectral import get_spectral_transform

    audio = torch.randn(2, 12000, generator=torch.Generator().manual_seed(3)) * 0.2
    tr = get_spectral_transform("mel", 1024, 1024, 256, sample_rate=22050, n_mels=80, f_min=0, f_max=8000)
    got = tr(audio.to(cuda_device)).cpu()
    want = mel_ref.torchaudio_mel_ref(audio, 22050, 1024, 1024, 256, 80, 0.0, 8000.0)
    assert got.shape == want.shape == (2, 80, 1 + 12000 // 256)
    assert float((got - want).abs().max()) <= 1e-4 * float(want.max())


@pytest.mark.parametrize("n_fft,win,hop,shape", SPEC_CASES)
def test_inverse_spectrogram_matches_torch_istft_and_round_trips(cuda_device, n_fft, win, hop, shape):
    from everyvoice_amd.spectral import get_spectral_transform

    frames = 1 + shape[-1] // hop
    g = torch.Generator().manual_seed(hop)
    spec = torch.complex(torch.randn(*shape[:-1], n_fft // 2 + 1, frames, generator=g), torch.randn(*shape[:-1], n_fft // 2 + 1, frames, generator=g))
    inv = get_spectral_transform("istft", n_fft, win, hop)
    got = inv(spec.to(cuda_device)).cpu()
    want = mel_ref.inverse_spectrogram_ref(spec.reshape(-1, n_fft // 2 + 1, frames), n_fft, win, hop).reshape(shape[:-1] + (-1,))
    assert got.shape == want.shape == shape[:-1] + (hop * (frames - 1),)
    assert float((got - want).abs().max()) <= 2e-5 * float(want.abs().max())
    # analysis -> synthesis returns the signal (the property that holds at any size)
    audio = torch.randn(*shape, generator=g) * 0.3
    S = shape[-1] // hop * hop
    back = inv(get_spectral_transform("raw", n_fft, win, hop)(audio[..., :S].to(cuda_device))).cpu()
    assert back.shape[-1] == S and float((back - audio[..., :S]).abs().max()) <= 2e-5


def test_unknown_spec_type_returns_none_like_the_reference():
    from everyvoice_amd.spectral import get_spectral_transform

    assert get_spectral_transform("no-such-type", 1024, 1024, 256) is None
