"""Host-side mirror of the configuration surface the hot path reads.

Field names and defaults are those frozen in the reference's published JSON schemas
(``everyvoice/.schema/everyvoice-spec-to-wav-0.5.json``: ``HiFiGANModelConfig`` 293-415,
``AudioConfig``; ``everyvoice/config/preprocessing_config.py:25-91``), so a checkpoint's
``hyper_parameters["config"]`` dict of the reference validates here unchanged.  Only the parts
the path consumes are typed; the rest of the reference's config tree (training/logger/paths)
is carried as plain dicts (``extra="allow"`` on the containers that hold them).
"""

from __future__ import annotations

from typing import Any

from pydantic import BaseModel, ConfigDict, Field


class _Strict(BaseModel):
    model_config = ConfigDict(extra="forbid", validate_assignment=True)


class AudioConfig(_Strict):
    """everyvoice/config/preprocessing_config.py:25-91 — the STFT contract."""

    min_audio_length: float = 0.4
    max_audio_length: float = 11.0
    max_wav_value: float = 32767.0
    input_sampling_rate: int = 22050
    output_sampling_rate: int = 22050
    alignment_sampling_rate: int = 22050
    target_bit_depth: int = 16
    n_fft: int = 1024
    fft_window_size: int = 1024
    fft_hop_size: int = 256
    f_min: int = 0
    f_max: int = 8000
    n_mels: int = 80
    spec_type: str = "mel-librosa"
    vocoder_segment_size: int = 8192


class PreprocessingConfig(BaseModel):
    model_config = ConfigDict(extra="allow")
    audio: AudioConfig = Field(default_factory=AudioConfig)


class HiFiGANModelConfig(_Strict):
    """everyvoice/.schema/everyvoice-spec-to-wav-0.5.json:293-415."""

    resblock: str = "1"
    upsample_rates: list[int] = Field(default_factory=lambda: [8, 8, 2, 2])
    upsample_kernel_sizes: list[int] = Field(default_factory=lambda: [16, 16, 4, 4])
    upsample_initial_channel: int = 512
    resblock_kernel_sizes: list[int] = Field(default_factory=lambda: [3, 7, 11])
    resblock_dilation_sizes: list[list[int]] = Field(default_factory=lambda: [[1, 3, 5], [1, 3, 5], [1, 3, 5]])
    activation_function: str = "everyvoice.utils.original_hifigan_leaky_relu"
    istft_layer: bool = False
    msd_layers: int = 3
    mpd_layers: list[int] = Field(default_factory=lambda: [2, 3, 5, 7, 11])


class HiFiGANConfig(BaseModel):
    """The vocoder config as the path needs it: ``model`` + ``preprocessing.audio``.
    ``training``, ``contact``, ``VERSION`` and path fields ride along untouched."""

    model_config = ConfigDict(extra="allow")
    VERSION: str = "1.0"
    model: HiFiGANModelConfig = Field(default_factory=HiFiGANModelConfig)
    preprocessing: PreprocessingConfig = Field(default_factory=PreprocessingConfig)
    training: dict[str, Any] = Field(default_factory=dict)

    # iSTFTNet head size (tests/data/relative/config/everyvoice-text-to-wav.yaml:6-8)
    gen_istft_n_fft: int = 16
    gen_istft_hop_size: int = 4


#: leaky-relu slope of the only activation the reference ships for the vocoder
#: (everyvoice/utils/__init__.py:178-181)
ACTIVATION_SLOPES = {"everyvoice.utils.original_hifigan_leaky_relu": 0.1}
