"""Host-side mirror of the configuration surface the hot path reads.

Field names and defaults are those frozen in the reference's published JSON schemas
(``everyvoice/.schema/everyvoice-spec-to-wav-0.5.json``: ``HiFiGANModelConfig`` 293-415,
``AudioConfig``; ``everyvoice/config/preprocessing_config.py:25-91``), so a checkpoint's
``hyper_parameters["config"]`` dict of the reference validates here unchanged.  Only the parts
the path consumes are typed; the rest of the reference's config tree (training/logger/paths)
is carried as plain dicts (``extra="allow"`` on the containers that hold them).
"""

from __future__ import annotations

import json
from pathlib import Path
from typing import Any, Literal, Optional, Union

from pydantic import BaseModel, ConfigDict, Field, model_validator


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
    """everyvoice/config/preprocessing_config.py (dataset, save_dir, audio; ``source_data`` rides along untouched)."""

    model_config = ConfigDict(extra="allow")
    dataset: str = "YourDataSet"
    train_split: float = 0.9
    dataset_split_seed: int = 1234
    save_dir: Path = Path("preprocessed/YourDataSet")
    audio: AudioConfig = Field(default_factory=AudioConfig)


# ---- training configuration (everyvoice/config/shared_types.py:150-320; schema everyvoice-spec-to-wav-0.5.json:434-622) ----
class AdamOptimizer(_Strict):
    learning_rate: float = 1e-4
    eps: float = 1e-8
    weight_decay: float = 0.01
    betas: tuple[float, float] = (0.9, 0.98)
    name: Literal["adam"] = "adam"


class AdamWOptimizer(AdamOptimizer):
    name: Literal["adamw"] = "adamw"


class RMSOptimizer(_Strict):
    learning_rate: float = 1e-4
    eps: float = 1e-8
    weight_decay: float = 0.01
    alpha: float = 0.99
    name: Literal["rms"] = "rms"


class LoggerConfig(BaseModel):
    """Where logs and checkpoints go: <save_dir> / <name> / <version> / <sub_dir> (shared_types.py:143-177)."""

    model_config = ConfigDict(extra="allow")
    name: str = "BaseExperiment"
    save_dir: Path = Path("logs_and_checkpoints")
    sub_dir_callable: str = "everyvoice.utils.get_current_time"
    version: str = "base"

    @property
    def sub_dir(self) -> str:
        import time

        if not hasattr(self, "_sub_dir"):
            object.__setattr__(self, "_sub_dir", time.strftime("%Y-%m-%d-%H-%M-%S"))
        return self._sub_dir


class BaseTrainingConfig(BaseModel):
    """shared_types.py:180-258 (BaseTrainingConfig): the fields the driver reads (helpers.py:234-259, 300-307)."""

    model_config = ConfigDict(extra="allow", validate_assignment=True)
    batch_size: int = 16
    save_top_k_ckpts: int = 5
    ckpt_steps: Optional[int] = Field(default=None, ge=0)
    ckpt_epochs: Optional[int] = Field(default=1, ge=0)
    val_check_interval: Union[int, float, None] = 500
    check_val_every_n_epoch: Optional[int] = None
    max_epochs: int = 1000
    max_steps: int = 100000
    finetune_checkpoint: Optional[Path] = None
    training_filelist: Path = Path("path/to/your/preprocessed/training_filelist.psv")
    validation_filelist: Path = Path("path/to/your/preprocessed/validation_filelist.psv")
    filelist_loader: str = "everyvoice.utils.generic_psv_filelist_reader"
    logger: LoggerConfig = Field(default_factory=LoggerConfig)
    val_data_workers: int = 0
    train_data_workers: int = 4

    @model_validator(mode="after")
    def _mutually_exclusive_ckpt_options(self):
        if self.ckpt_epochs is not None and self.ckpt_steps is not None:
            raise ValueError("ckpt_epochs and ckpt_steps have to be mutually exclusive")
        return self


class HiFiGANTrainingConfig(BaseTrainingConfig):
    """everyvoice-spec-to-wav-0.5.json:434-622 (hfgl.config.HiFiGANTrainingConfig)."""

    generator_warmup_steps: int = 0
    gan_type: Literal["original", "wgan"] = "original"
    optimizer: Union[AdamOptimizer, AdamWOptimizer, RMSOptimizer] = Field(default_factory=AdamWOptimizer, discriminator="name")
    wgan_clip_value: float = 0.01
    use_weighted_sampler: bool = False
    finetune: bool = False


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
    training: HiFiGANTrainingConfig = Field(default_factory=HiFiGANTrainingConfig)

    # -- the two class-level entry points the reference's driver uses (base_cli/helpers.py:85, 111; shared_types.py:90-94) --
    @classmethod
    def load_config_from_path(cls, path) -> "HiFiGANConfig":
        """A YAML or JSON config file -> validated config (relative paths are resolved against the file's directory)."""
        path = Path(path)
        text = path.read_text(encoding="utf8")
        if path.suffix.lower() == ".json":
            data = json.loads(text)
        else:
            import yaml

            data = yaml.safe_load(text) or {}
        cfg = cls(**data)
        base = path.resolve().parent
        t = cfg.training
        for name in ("training_filelist", "validation_filelist", "finetune_checkpoint"):
            v = getattr(t, name)
            if v is not None and not Path(v).is_absolute():
                setattr(t, name, (base / v).resolve())
        if not cfg.preprocessing.save_dir.is_absolute():
            cfg.preprocessing.save_dir = (base / cfg.preprocessing.save_dir).resolve()
        if not t.logger.save_dir.is_absolute():
            t.logger.save_dir = (base / t.logger.save_dir).resolve()
        return cfg

    def update_config(self, new_config: dict) -> "HiFiGANConfig":
        """Nested update from ``-c key.sub=value`` overrides (shared_types.py:90-121): values merge into the dumped config
        and the whole thing is validated again."""
        def merge(a, b):
            out = dict(a)
            for k, v in b.items():
                old = out.get(k)
                if isinstance(v, dict) and isinstance(old, dict) and not ("name" in v and v["name"] != old.get("name")):
                    out[k] = merge(old, v)
                else:
                    out[k] = v  # (a tagged union switching to another member, e.g. optimizer.name: replaced, not merged)
            return out

        merged = merge(self.model_dump(), new_config)
        self.__init__(**merged)
        return self

    def model_checkpoint_dump(self) -> dict:
        """JSON-only, path-free dump for ``hyper_parameters["config"]`` (shared_types.py:56-88: checkpoints travel between
        machines, so every Path-valued field is dropped)."""
        def strip(v):
            if isinstance(v, dict):
                return {k: strip(x) for k, x in v.items() if not isinstance(x, Path)}
            if isinstance(v, (list, tuple)):
                return [strip(x) for x in v if not isinstance(x, Path)]
            return v

        return json.loads(json.dumps(strip(self.model_dump()), default=str))

    # iSTFTNet head size (tests/data/relative/config/everyvoice-text-to-wav.yaml:6-8)
    gen_istft_n_fft: int = 16
    gen_istft_hop_size: int = 4


#: leaky-relu slope of the only activation the reference ships for the vocoder
#: (everyvoice/utils/__init__.py:178-181)
ACTIVATION_SLOPES = {"everyvoice.utils.original_hifigan_leaky_relu": 0.1}
