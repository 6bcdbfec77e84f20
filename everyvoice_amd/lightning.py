"""The driver <-> model contract of the reference, without importing Lightning: module-shaped classes
``HiFiGAN(config)`` and ``FastSpeech2(config, stats=, lang2id=, speaker2id=)`` over the libevmi_hip trainers, and a
``train_base_command`` with the reference's signature and call sequence.

Mirrors (paths relative to the reference):
  train_base_command          everyvoice/base_cli/helpers.py:173-375   load config (+ ``-c key=value`` overrides) -> log dir with
                              hparams.yaml -> data_module(config) -> train from scratch, or ``finetune_checkpoint``: load_from_checkpoint,
                              refuse a different model architecture, restart the optimiser when its hyper-parameters changed,
                              otherwise resume weights + optimiser + step counters; checkpoints: ``last.ckpt`` every ckpt_steps /
                              ckpt_epochs and the save_top_k_ckpts best by the monitored validation loss (mode min);
                              max_epochs / max_steps; val_check_interval / check_val_every_n_epoch; ``--devices N --strategy ddp``
                              = one process per GPU (base_cli/interfaces.py:84-97), utterances sharded by rank (dataset.ShardedSampler)
  module contract             everyvoice/tests/test_model.py:85-151 (``config`` / ``hparams.config``; JSON-only, path-free
                              ``hyper_parameters["config"]``; ``model_info`` = {name, version} written by on_save_checkpoint),
                              :253-262 (foreign config -> TypeError "Unable to load config. ..."), :302-313 (wrong class -> TypeError
                              "Wrong model type (X), we are expecting a 'Y' model"), :454-459 (newer major version -> ValueError)
The reference's Trainer / loggers / callbacks (control plane) are reduced to this plain loop; the step, the optimisers and the
checkpoint contents belong to the module, as in the reference.
"""

from __future__ import annotations

import json
import math
import os
import sys
from dataclasses import asdict, dataclass, field, is_dataclass
from pathlib import Path
from types import SimpleNamespace

import torch

from .config import HiFiGANConfig


class InvalidConfiguration(ValueError):
    """Fine-tuning with a model architecture that differs from the checkpoint's (helpers.py:318-330)."""


def _check_model_info(ckpt: dict, expected: str, my_version: str):
    info = ckpt.get("model_info") if isinstance(ckpt, dict) else None
    if isinstance(info, dict):
        if info.get("name") != expected:
            raise TypeError(f"Wrong model type ({info.get('name')}), we are expecting a '{expected}' model")
        ck_major = int(str(info.get("version", "1.0")).split(".")[0])
        if ck_major > int(my_version.split(".")[0]):
            raise ValueError("Your model was created with a newer version of EveryVoice, please update your software.")


class _Module:
    """What both model classes share of the LightningModule surface the driver touches."""

    _VERSION = "1.0"
    current_epoch = 0

    def __init__(self):
        self.logged: dict[str, float] = {}
        self.device = torch.device("cpu")

    def log(self, name: str, value, **_):
        self.logged[name] = float(value)

    def log_dict(self, d: dict, **_):
        for k, v in d.items():
            self.log(k, v)

    @property
    def global_step(self) -> int:
        return self.trainer_.global_step if getattr(self, "trainer_", None) is not None else 0

    def save_checkpoint(self, path):
        ckpt = self.checkpoint()
        self.on_save_checkpoint(ckpt)
        Path(path).parent.mkdir(parents=True, exist_ok=True)
        torch.save(ckpt, path)
        return ckpt


# =====================================================================================================================
class HiFiGAN(_Module):
    """``hfgl.model.HiFiGAN``: generator + MPD + MSD, manual optimisation with two optimisers."""

    def __init__(self, config: HiFiGANConfig | dict, device=None, precision: str = "bf16", process_group=None, use_graph: bool = True):
        super().__init__()
        if not isinstance(config, HiFiGANConfig):
            try:
                config = HiFiGANConfig(**config)
            except Exception as e:  # noqa: BLE001 -- pydantic's ValidationError and anything a foreign dict provokes
                raise TypeError("Unable to load config.  Possible causes: is it really a VocoderConfig? or the correct version?") from e
        self.config = config
        self.hparams = SimpleNamespace(config=config)
        self.precision, self.process_group, self.use_graph = precision, process_group, use_graph
        self.trainer_ = None
        self._pending_ckpt = None
        if device is not None:
            self.to(device)

    # -- placement: the libevmi_hip trainer is built for a device -------------------------------------------------------
    def to(self, device):
        from .train.hifigan import HiFiGANTrainer

        device = torch.device(device)
        if self.trainer_ is not None and self.trainer_.device == device:
            return self
        old = self.trainer_.checkpoint() if self.trainer_ is not None else self._pending_ckpt
        t, o = self.config.training, self.config.training.optimizer
        kw = dict(lr=o.learning_rate, eps=o.eps, weight_decay=o.weight_decay, optimizer=o.name)
        if o.name == "rms":
            kw["alpha"] = o.alpha
        else:
            kw["betas"] = tuple(o.betas)
        self.trainer_ = HiFiGANTrainer(self.config, device=device, precision=self.precision, process_group=self.process_group,
                                       gan_type=t.gan_type, wgan_clip_value=t.wgan_clip_value, generator_warmup_steps=t.generator_warmup_steps,
                                       use_graph=self.use_graph and device.type == "cuda", **kw)
        if old is not None:
            self.trainer_.load_checkpoint(old, restore_optimizers=self._restore_optimizers)
        self._pending_ckpt = None
        self.device = device
        return self

    _restore_optimizers = True

    def update_config_settings(self):
        """The driver replaced ``self.config`` (helpers.py:336-338): re-read what the step uses from it."""
        self.hparams = SimpleNamespace(config=self.config)
        if self.trainer_ is not None:
            t, o = self.config.training, self.config.training.optimizer
            tr = self.trainer_
            tr.config = self.config
            tr.opt.update(lr=o.learning_rate, eps=o.eps, weight_decay=o.weight_decay)
            if o.name != "rms":
                tr.opt["betas"] = tuple(o.betas)
            else:
                tr.alpha = o.alpha
            tr.optimizer, tr.gan_type = o.name, t.gan_type
            tr.wgan_clip_value, tr.generator_warmup_steps = t.wgan_clip_value, t.generator_warmup_steps

    def restart_optimizers(self):
        """Moments, step counters and the global step back to zero (the driver's "optimizer hyperparameters changed" branch,
        helpers.py:341-356, when the trainer already exists)."""
        tr = self.trainer_
        for grp in (tr.g_params, tr.d_params):
            grp.m.zero_()
            grp.v.zero_()
            grp.set_step(0)
        tr.global_step = 0

    def configure_optimizers(self):
        """Two optimisers (generator, discriminators) of the configured kind; they live as flat buffers inside the trainer
        (one fused kernel each), described here the way Lightning would list them."""
        o = self.config.training.optimizer
        return [{"name": o.name, "params": "generator", **o.model_dump()}, {"name": o.name, "params": "discriminators", **o.model_dump()}]

    # -- steps ----------------------------------------------------------------------------------------------------------
    def _device_batch(self, batch):
        spec, audio, basenames, spec_from_audio = batch
        dev = self.trainer_.device
        return spec.to(dev, torch.float32), audio.to(dev, torch.float32).reshape(audio.shape[0], 1, -1), basenames, spec_from_audio

    def training_step(self, batch, batch_idx: int = 0):
        """batch = (spec [B, n_mels, F], audio [B, S], basenames, spec_from_audio [B, n_mels, F]) as the reference's
        SpecDataset + DataLoader deliver it (tests/test_dataloader.py:55-65)."""
        if self.trainer_ is None:
            raise RuntimeError("HiFiGAN.training_step: move the module to a GPU first (.to('cuda:0')); there is no CPU path")
        spec, audio, _, _ = self._device_batch(batch)
        out = self.trainer_.training_step(spec, audio)
        self.log_dict({"training/disc/d_loss_total": out["d"], "training/gen/loss_total": out["g_total"], "training/gen/mel_spec_error": out["g_mel"] / 45.0,
                       "training/gen/adv": out["g_adv"], "training/gen/feature_matching": out["g_fm"]})
        return out

    def validation_step(self, batch, batch_idx: int = 0):
        """Generator forward on a validation item; ``validation/mel_spec_error`` (L1 between log-mels) is what the vocoder's
        checkpoints are ranked by."""
        from .spectral import MelSpectrogram

        spec, audio, _, _ = self._device_batch(batch)
        wav = self.trainer_.generate(spec)
        a = self.config.preprocessing.audio
        tr = getattr(self, "_val_mel", None) or MelSpectrogram(a.n_fft, a.fft_window_size, a.fft_hop_size, a.input_sampling_rate, a.n_mels, a.f_min, a.f_max)
        self._val_mel = tr
        n = min(wav.shape[-1], audio.shape[-1])
        err = float((tr(wav[:, 0, :n], log=True) - tr(audio[:, 0, :n], log=True)).abs().mean())
        self.log("validation/mel_spec_error", err)
        return err

    # -- checkpoints ----------------------------------------------------------------------------------------------------
    def state_dict(self):
        return self.trainer_.state_dict() if self.trainer_ is not None else {}

    def checkpoint(self) -> dict:
        if self.trainer_ is None:
            from .train.hifigan import HiFiGANTrainer

            self.trainer_ = HiFiGANTrainer(self.config, device="cpu")  # host-only: parameters exist, nothing can run
        ck = self.trainer_.checkpoint()
        ck["epoch"] = self.current_epoch
        return ck

    def on_save_checkpoint(self, checkpoint: dict):
        """JSON-only, path-free config + the model's name and version (tests/test_model.py:85-151, 302-313)."""
        checkpoint.setdefault("hyper_parameters", {})["config"] = self.config.model_checkpoint_dump()
        checkpoint["model_info"] = {"name": type(self).__name__, "version": self._VERSION}

    @classmethod
    def load_from_checkpoint(cls, path, device=None, **kw):
        ckpt = torch.load(path, map_location="cpu", weights_only=True) if not isinstance(path, dict) else path
        _check_model_info(ckpt, cls.__name__, cls._VERSION)
        try:
            config = HiFiGANConfig(**ckpt["hyper_parameters"]["config"])
        except Exception as e:  # noqa: BLE001
            raise TypeError("Unable to load config.  Possible causes: is it really a VocoderConfig? or the correct version?") from e
        obj = cls(config, **kw)
        obj._pending_ckpt = ckpt
        obj.current_epoch = int(ckpt.get("epoch", 0))
        if device is not None:
            obj.to(device)
        return obj


# =====================================================================================================================
@dataclass
class FastSpeech2Config:
    """The feature-prediction config as the path needs it (``FeaturePredictionConfig``, everyvoice/model/feature_prediction/config.py):
    ``model`` (FastSpeech2ModelConfig) + ``training`` (loss weights over the driver-side BaseTrainingConfig fields) +
    ``preprocessing`` (save_dir, audio) and the symbol inventory the data side encodes token strings with (``symbols``: what the
    reference derives from ``text.symbols`` in its text front-end, out of scope here)."""

    model: object = None
    training: object = None
    preprocessing: object = None
    symbols: list | None = None
    VERSION: str = "1.0"

    def __post_init__(self):
        from .config import PreprocessingConfig
        from .fs2 import FastSpeech2ModelConfig
        from .train.fs2 import FastSpeech2TrainingConfig

        if isinstance(self.model, dict):
            self.model = _dataclass_from_dict(FastSpeech2ModelConfig, self.model)
        if isinstance(self.training, dict):
            self.training = FastSpeech2TrainingConfig(**self.training)
        if self.preprocessing is None or isinstance(self.preprocessing, dict):
            self.preprocessing = PreprocessingConfig(**(self.preprocessing or {}))
        self.model = self.model or FastSpeech2ModelConfig()
        self.training = self.training or FastSpeech2TrainingConfig()

    # -- the two class-level entry points the driver uses (base_cli/helpers.py:85, 111; shared_types.py:90-94) ------------------
    @classmethod
    def load_config_from_path(cls, path) -> "FastSpeech2Config":
        """A YAML or JSON config file -> config; relative paths are resolved against the file's directory."""
        path = Path(path)
        text = path.read_text(encoding="utf8")
        if path.suffix.lower() == ".json":
            data = json.loads(text)
        else:
            import yaml

            data = yaml.safe_load(text) or {}
        try:
            cfg = cls(**data)
        except TypeError as e:
            raise TypeError("Unable to load config.  Possible causes: is it really a FeaturePredictionConfig? or the correct version?") from e
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

    def model_dump(self, mode: str = "json") -> dict:
        return {"VERSION": self.VERSION, "model": asdict(self.model), "training": self.training.json_dict(paths=True),
                "preprocessing": self.preprocessing.model_dump(mode="json"), "symbols": list(self.symbols) if self.symbols is not None else None}

    def update_config(self, new_config: dict) -> "FastSpeech2Config":
        """Nested update from ``-c key.sub=value`` overrides (shared_types.py:90-121): merged into the dumped config, rebuilt."""
        def merge(a, b):
            out = dict(a)
            for k, v in b.items():
                out[k] = merge(out[k], v) if isinstance(v, dict) and isinstance(out.get(k), dict) else v
            return out

        self.__init__(**merge(self.model_dump(), new_config))
        return self

    def model_checkpoint_dump(self) -> dict:
        """JSON-only and path-free (tests/test_model.py:85-151)."""
        return json.loads(json.dumps({"VERSION": self.VERSION, "model": asdict(self.model), "training": self.training.json_dict(),
                                      "symbols": list(self.symbols) if self.symbols is not None else None}, default=str))


class FastSpeech2(_Module):
    """``fs2.model.FastSpeech2(config, stats, lang2id, speaker2id)`` (tests/model_stubs.py:44-58)."""

    def __init__(self, config: FastSpeech2Config | None = None, stats=None, lang2id: dict | None = None, speaker2id: dict | None = None,
                 device=None, precision: str = "bf16", process_group=None, use_graph: bool = True, graph_buckets: tuple | None = None):
        """``use_graph`` / ``graph_buckets``: steps replay as HIP graphs per padded batch shape (train/fs2.py): an LRU of captured shapes,
        unseen shapes run eagerly twice first.  ``graph_buckets=(symbols, frames)`` (e.g. (16, 64)) pads batches up to multiples so that
        the variable-length batches of a real loader fall on a small set of shapes -- OFF by default since round 5: the
        Conformer's BatchNorm takes its batch statistics over every column, padded ones included (as torchaudio's does), so extra
        padding shifts mean / variance and the running statistics away from a run padded to the batch maximum only; opt in where
        replay rate matters more than that drift.  ``model_kwargs`` of ``train_base_command`` reach here."""
        super().__init__()
        from .fs2 import Stats

        self.config = config or FastSpeech2Config()
        self.hparams = SimpleNamespace(config=self.config)
        self.stats = stats or Stats()
        self.lang2id, self.speaker2id = dict(lang2id or {}), dict(speaker2id or {})
        self.precision, self.process_group = precision, process_group
        self.use_graph, self.graph_buckets = bool(use_graph), tuple(graph_buckets) if graph_buckets else None
        self.trainer_ = None
        self._pending_ckpt = None
        if device is not None:
            self.to(device)

    def to(self, device):
        from .train.fs2 import FastSpeech2Trainer

        device = torch.device(device)
        if self.trainer_ is not None and self.trainer_.device == device:
            return self
        old = self.trainer_.checkpoint() if self.trainer_ is not None else self._pending_ckpt
        self.trainer_ = FastSpeech2Trainer(self.config.model, self.stats, self.config.training, device=device, lang2id=self.lang2id,
                                           speaker2id=self.speaker2id, process_group=self.process_group, precision=self.precision,
                                           use_graph=self.use_graph and device.type == "cuda", graph_buckets=self.graph_buckets)
        if old is not None:
            self.trainer_.load_checkpoint(old)
        self._pending_ckpt = None
        self.device = device
        return self

    def update_config_settings(self):
        self.hparams = SimpleNamespace(config=self.config)
        if self.trainer_ is not None:
            self.trainer_.training = self.config.training

    def restart_optimizers(self):
        tr = self.trainer_
        tr.params.m.zero_()
        tr.params.v.zero_()
        tr.params.step = 0
        tr.global_step = 0

    def configure_optimizers(self):
        o = self.config.training.optimizer
        return [{"name": "adamw", "schedule": "noam", **asdict(o)}]

    def training_step(self, batch: dict, batch_idx: int = 0):
        if self.trainer_ is None:
            raise RuntimeError("FastSpeech2.training_step: move the module to a GPU first (.to('cuda:0')); there is no CPU path")
        self.trainer_.current_epoch = self.current_epoch
        out = {k: float(v) for k, v in self.trainer_.training_step(batch).items()}
        self.log_dict({f"training/{k}_loss": v for k, v in out.items()})
        return out

    def validation_step(self, batch: dict, batch_idx: int = 0):
        """The losses of a validation batch in evaluation mode, as the reference's loop runs them under ``model.eval()``: dropout
        off, BatchNorm on its running statistics (not updated), no backward (``validation/mel_loss`` is the monitored one)."""
        losses = {k: float(v) for k, v in self.trainer_.evaluate(batch).items()}
        self.log_dict({f"validation/{k}_loss": v for k, v in losses.items()})
        return losses.get("mel", losses.get("total"))

    def state_dict(self):
        return self.trainer_.state_dict() if self.trainer_ is not None else {}

    def checkpoint(self) -> dict:
        if self.trainer_ is None:
            raise RuntimeError("FastSpeech2.checkpoint: the libevmi_hip trainer exists on a GPU only (.to('cuda:0') first)")
        ck = self.trainer_.checkpoint()
        ck["epoch"] = self.current_epoch
        return ck

    def on_save_checkpoint(self, checkpoint: dict):
        hp = checkpoint.setdefault("hyper_parameters", {})
        hp["config"] = self.config.model_checkpoint_dump()
        hp["stats"] = asdict(self.stats) if is_dataclass(self.stats) else dict(self.stats)
        hp["lang2id"], hp["speaker2id"] = dict(self.lang2id), dict(self.speaker2id)
        checkpoint["model_info"] = {"name": type(self).__name__, "version": self._VERSION}

    @classmethod
    def load_from_checkpoint(cls, path, device=None, **kw):
        from .fs2 import FastSpeech2ModelConfig, Stats, StatsInfo
        from .train.fs2 import FastSpeech2TrainingConfig, NoamOptimizerConfig

        ckpt = torch.load(path, map_location="cpu", weights_only=True) if not isinstance(path, dict) else path
        _check_model_info(ckpt, cls.__name__, cls._VERSION)
        try:
            hp = ckpt["hyper_parameters"]
            c = hp["config"]
            model_cfg = FastSpeech2ModelConfig.from_dict(c["model"]) if hasattr(FastSpeech2ModelConfig, "from_dict") else _dataclass_from_dict(FastSpeech2ModelConfig, c["model"])
            config = FastSpeech2Config(model=model_cfg, training=FastSpeech2TrainingConfig(**dict(c.get("training", {}))), symbols=c.get("symbols"))
            st = hp.get("stats")
            stats = Stats(pitch=StatsInfo(**st["pitch"]), energy=StatsInfo(**st["energy"])) if st else None
        except (KeyError, TypeError, ValueError) as e:
            raise TypeError("Unable to load config.  Possible causes: is it really a FeaturePredictionConfig? or the correct version?") from e
        obj = cls(config, stats=stats, lang2id=hp.get("lang2id"), speaker2id=hp.get("speaker2id"), **kw)
        obj._pending_ckpt = ckpt
        obj.current_epoch = int(ckpt.get("epoch", 0))
        if device is not None:
            obj.to(device)
        return obj


def _dataclass_from_dict(cls, d: dict):
    """Nested dataclass from its asdict() form (fields that are dataclasses themselves are rebuilt recursively)."""
    import dataclasses
    import typing

    hints = typing.get_type_hints(cls)
    kw = {}
    for f in dataclasses.fields(cls):
        if f.name not in d:
            continue
        v, t = d[f.name], hints.get(f.name)
        kw[f.name] = _dataclass_from_dict(t, v) if dataclasses.is_dataclass(t) and isinstance(v, dict) else v
    return cls(**kw)


# =====================================================================================================================
def parse_config_args(config_args: list[str]) -> dict:
    """``-c training.batch_size=4 -c model.istft_layer=true`` -> nested dict (base_cli/helpers.py:111-132); values parse as JSON
    when they can ("4" -> 4, "true" -> True), else stay strings."""
    out: dict = {}
    for arg in config_args or []:
        key, sep, value = arg.partition("=")
        if not sep:
            raise ValueError(f"config override {arg!r} is not of the form key=value")
        try:
            parsed = json.loads(value)
        except ValueError:
            parsed = value
        node = out
        parts = key.split(".")
        for p in parts[:-1]:
            node = node.setdefault(p, {})
        node[parts[-1]] = parsed
    return out


def load_config_base_command(model_config, config_args: list[str], config_file: Path):
    config = model_config.load_config_from_path(config_file)
    overrides = parse_config_args(config_args)
    if overrides:
        config.update_config(overrides)
    return config


def save_configuration_to_log_dir(config) -> Path:
    """<save_dir>/<name>/<version>/<sub_dir>/hparams.yaml with the JSON form of the config (helpers.py:150-170).  Under
    ``--devices N`` every rank must agree on <sub_dir> (a timestamp): the parent takes it once and hands it to its children
    (EVMI_LOG_SUB_DIR); only rank 0 writes the file."""
    lg = config.training.logger
    sub = os.environ.get("EVMI_LOG_SUB_DIR")
    if sub:
        object.__setattr__(lg, "_sub_dir", sub)
    log_dir = Path(lg.save_dir) / lg.name / lg.version / lg.sub_dir
    log_dir.mkdir(parents=True, exist_ok=True)
    if int(os.environ.get("RANK", "0")) == 0:
        import yaml

        with (log_dir / "hparams.yaml").open("w", encoding="UTF-8") as f:
            yaml.dump(json.loads(json.dumps(config.model_dump(mode="json") if hasattr(config, "model_dump") else config.model_checkpoint_dump())), stream=f)
    return log_dir


def _config_diff(a: dict, b: dict, prefix="") -> list:
    diffs = []
    for k in sorted(set(a) | set(b)):
        va, vb = a.get(k), b.get(k)
        if isinstance(va, dict) and isinstance(vb, dict):
            diffs += _config_diff(va, vb, f"{prefix}{k}.")
        elif va != vb and not (isinstance(va, (list, tuple)) and isinstance(vb, (list, tuple)) and list(va) == list(vb)):
            diffs.append((f"{prefix}{k}", va, vb))
    return diffs


def _dump(obj) -> dict:
    if hasattr(obj, "model_dump"):
        return obj.model_dump(mode="json")
    return json.loads(json.dumps(asdict(obj), default=str)) if is_dataclass(obj) else dict(obj)


class _TopK:
    """ModelCheckpoint(monitor, mode="min", save_top_k=k): keeps the k best checkpoints by the monitored value."""

    def __init__(self, k: int, directory: Path):
        self.k, self.dir, self.best = k, directory, []  # (value, path)

    def offer(self, value: float, step: int, save_fn):
        if self.k == 0 or value is None or math.isnan(value):
            return None
        if len(self.best) >= self.k > 0 and value >= self.best[-1][0]:
            return None
        path = self.dir / f"step={step}-monitor={value:.5f}.ckpt"
        save_fn(path)
        self.best.append((value, path))
        self.best.sort(key=lambda t: t[0])
        while 0 < self.k < len(self.best):
            _, old = self.best.pop()
            old.unlink(missing_ok=True)
        return path


def fit(model_obj, data, config, monitor: str, log_dir: Path, gradient_clip_val=None, resume: dict | None = None, calls: list | None = None):
    """The training loop of this rank: epochs over ``data.train_dataloader()``, validation every ``val_check_interval`` steps
    (an int) or fraction of an epoch (a float) / every ``check_val_every_n_epoch`` epochs, ``last.ckpt`` every ckpt_steps or
    ckpt_epochs, the save_top_k_ckpts best by ``monitor``; stops at max_epochs or max_steps."""
    t = config.training
    note = (lambda *a: calls.append(a)) if calls is not None else (lambda *a: None)
    rank = int(os.environ.get("RANK", "0"))
    # what Lightning's Trainer.fit does with a data module: prepare_data() once (rank 0), then setup("fit") on every rank
    if rank == 0 and hasattr(data, "prepare_data"):
        data.prepare_data()
    if int(os.environ.get("WORLD_SIZE", "1")) > 1:
        import torch.distributed as dist

        dist.barrier()
    if hasattr(data, "setup"):
        data.setup("fit")
    train_loader, val_loader = data.train_dataloader(), data.val_dataloader()
    ckpt_dir = Path(log_dir) / "checkpoints"
    topk = _TopK(t.save_top_k_ckpts, ckpt_dir)
    if gradient_clip_val is not None and hasattr(getattr(model_obj, "trainer_", None), "training"):
        model_obj.trainer_.training.gradient_clip_val = gradient_clip_val
    epoch0 = int(resume.get("epoch", 0)) if resume else 0
    steps_per_epoch = max(1, len(train_loader))
    vci = t.val_check_interval
    val_every_steps = vci if isinstance(vci, int) and not isinstance(vci, bool) else (max(1, int(steps_per_epoch * vci)) if vci else None)

    world = int(os.environ.get("WORLD_SIZE", "1"))

    def validate():
        # every rank takes the validation batches i = rank (mod world); the monitored value is the mean over all of them
        # (Lightning: a DistributedSampler on the validation loader + sync_dist logging), so every rank ranks checkpoints alike
        vals = [model_obj.validation_step(b, i) for i, b in enumerate(val_loader) if i % world == rank]
        vals = [v for v in vals if v is not None]
        tot, cnt = float(sum(vals)), float(len(vals))
        if world > 1:
            import torch.distributed as dist

            dev = getattr(model_obj, "device", torch.device("cpu"))
            acc = torch.tensor([tot, cnt], dtype=torch.float64, device=dev if dist.get_backend() == "nccl" else "cpu")
            dist.all_reduce(acc)
            tot, cnt = float(acc[0]), float(acc[1])
        value = tot / cnt if cnt else float("nan")
        model_obj.log(monitor, value)
        note("validate", model_obj.global_step, value)
        if rank == 0:
            topk.offer(value, model_obj.global_step, model_obj.save_checkpoint)
        return value

    def save_last():
        if rank == 0:
            model_obj.save_checkpoint(ckpt_dir / "last.ckpt")
            note("save_last", model_obj.global_step)

    done = False
    for epoch in range(epoch0, t.max_epochs):
        model_obj.current_epoch = epoch
        sampler = getattr(data, "train_sampler", None)
        if sampler is not None:
            sampler.set_epoch(epoch)
        for i, batch in enumerate(train_loader):
            model_obj.training_step(batch, i)
            step = model_obj.global_step
            note("step", step)
            if val_every_steps and t.check_val_every_n_epoch is None and step % val_every_steps == 0:
                validate()
            if t.ckpt_steps and step % t.ckpt_steps == 0:
                save_last()
            if t.max_steps and t.max_steps > 0 and step >= t.max_steps:
                done = True
                break
        if not done:
            model_obj.current_epoch = epoch + 1  # the epoch is complete: a checkpoint written now resumes with the next one
        if t.check_val_every_n_epoch and (epoch + 1) % t.check_val_every_n_epoch == 0:
            validate()
        if t.ckpt_epochs and (epoch + 1) % t.ckpt_epochs == 0:
            save_last()
        if done:
            break
    if rank == 0 and not (ckpt_dir / "last.ckpt").exists():
        save_last()
    return model_obj


def train_base_command(model_config, data_module, model, monitor: str, config_args: list[str], config_file: Path, accelerator: str = "auto",
                       devices="auto", nodes: int = 1, strategy: str = "ddp", gradient_clip_val: float | None = None, model_kwargs=None,
                       calls: list | None = None):
    """Same parameters, same order of operations as the reference's (base_cli/helpers.py:173-375).  ``devices`` > 1 outside a
    launcher: this process starts one rank per GPU (torch.distributed.run) running the same command, and returns their exit code."""
    model_kwargs = dict(model_kwargs or {})
    n_dev = _resolve_devices(devices)
    if n_dev > 1 and "WORLD_SIZE" not in os.environ:
        return _launch_ranks(n_dev, model_config, data_module, model, monitor, config_args, config_file, gradient_clip_val, accelerator, model_kwargs)
    config = load_config_base_command(model_config, config_args, Path(config_file))
    log_dir = save_configuration_to_log_dir(config)
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    device = torch.device("cuda", int(os.environ.get("LOCAL_RANK", "0"))) if accelerator != "cpu" else torch.device("cpu")
    if world > 1:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if not dist.is_initialized():
            if device.type == "cuda":
                torch.cuda.set_device(device)
            if device.type == "cuda":
                dist.init_process_group("nccl", device_id=device)
            else:
                dist.init_process_group("gloo")
        model_kwargs.setdefault("process_group", True)
    try:
        data = data_module(config, rank=rank, world=world)
    except TypeError:
        data = data_module(config)
    last_ckpt = config.training.finetune_checkpoint
    last_ckpt = last_ckpt if last_ckpt is not None and os.path.exists(last_ckpt) else None
    resume = None
    if last_ckpt is None:  # train from scratch
        model_obj = model(config, **model_kwargs)
    else:
        try:
            model_obj = model.load_from_checkpoint(last_ckpt, **model_kwargs)
        except (TypeError, ValueError) as e:
            print(f"Unable to load {last_ckpt}: {e}", file=sys.stderr)
            sys.exit(1)
        model_diff = _config_diff(_dump(model_obj.config.model), _dump(config.model))
        if model_diff:
            raise InvalidConfiguration(
                "Sorry, you are a trying to fine-tune a model with a different architecture defined in your configuration than was "
                f"used during pre-training.\n\nPlease fix your configuration or use a different model.\n\nValues Changed: {model_diff}")
        optimizer_diff = _config_diff(_dump(model_obj.config.training.optimizer), _dump(config.training.optimizer))
        model_obj.config = config
        if optimizer_diff:  # weights from the checkpoint, optimiser and step counters restarted with the new hyper-parameters
            print(f"Some of your optimizer hyperparameters have changed from your checkpoint at '{last_ckpt}', so we will override your "
                  f"checkpoint hyperparameters and restart the optimizer.\n\nValues Changed: {optimizer_diff}", file=sys.stderr)
            model_obj._restore_optimizers = False
            if model_obj._pending_ckpt is not None:
                model_obj._pending_ckpt = {**model_obj._pending_ckpt, "global_step": 0, "epoch": 0, "optimizer_states": []}
            if getattr(model_obj, "trainer_", None) is not None and hasattr(model_obj, "restart_optimizers"):
                model_obj.restart_optimizers()  # load_from_checkpoint(device=...) already built the trainer and restored its moments
            model_obj.current_epoch = 0
        else:
            resume = {"epoch": model_obj.current_epoch}
        if hasattr(model_obj, "update_config_settings"):
            model_obj.update_config_settings()
    if device.type == "cuda":
        model_obj.to(device)
    if hasattr(model_obj, "update_config_settings") and last_ckpt is not None:
        model_obj.update_config_settings()
    fit(model_obj, data, config, monitor, log_dir, gradient_clip_val, resume, calls)
    if world > 1:
        import torch.distributed as dist

        dist.barrier()
        dist.destroy_process_group()
    return model_obj


def _resolve_devices(devices) -> int:
    if isinstance(devices, int):
        return devices
    if str(devices) == "auto":
        return 1 if "WORLD_SIZE" not in os.environ else int(os.environ["WORLD_SIZE"])
    return int(devices)


def _launch_ranks(n, model_config, data_module, model, monitor, config_args, config_file, gradient_clip_val, accelerator="auto", model_kwargs=None) -> int:
    """``--devices N --strategy ddp``: one process per GPU over RCCL, each running this same command (never exec: the parent
    has not touched the GPU and stays alive to return the children's exit code).  What the children need travels as one JSON
    document: the three classes by qualified name, the config arguments, the accelerator and the model keyword arguments
    (JSON values only: a process group or a device cannot cross a process boundary and is refused)."""
    import socket
    import subprocess
    import time

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("EVMI_LOG_SUB_DIR", time.strftime("%Y-%m-%d-%H-%M-%S"))  # one log directory for all ranks
    for cls in (model_config, data_module, model):
        if "<locals>" in cls.__qualname__:
            raise ValueError(f"{cls.__module__}:{cls.__qualname__}: a class defined inside a function cannot be found by the ranks of --devices {n}; "
                             "define it at module level")
    kwargs = dict(model_kwargs or {})
    try:
        json.dumps(kwargs)
    except TypeError as e:
        raise TypeError(f"model_kwargs must be JSON values to reach the ranks started by --devices {n}: {e}") from e
    spec = {"model_config": f"{model_config.__module__}:{model_config.__qualname__}", "data_module": f"{data_module.__module__}:{data_module.__qualname__}",
            "model": f"{model.__module__}:{model.__qualname__}", "monitor": monitor, "config_args": list(config_args or []), "config_file": str(config_file),
            "gradient_clip_val": gradient_clip_val, "accelerator": accelerator, "model_kwargs": kwargs}
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1", "--master-port", str(port),
           "-m", "everyvoice_amd.lightning", json.dumps(spec)]
    return subprocess.run(cmd, env=env).returncode


def _resolve_class(spec: str):
    """"module:Outer.Inner" -> the class (qualified names of nested classes are walked attribute by attribute)."""
    import importlib

    mod, _, qual = spec.partition(":")
    if "<locals>" in qual.split("."):
        raise ValueError(f"{spec}: a class defined inside a function cannot be found by the ranks of --devices N; define it at module level")
    obj = importlib.import_module(mod)
    for part in qual.split("."):
        obj = getattr(obj, part)
    return obj


def _main(argv):
    spec = json.loads(argv[0])
    train_base_command(_resolve_class(spec["model_config"]), _resolve_class(spec["data_module"]), _resolve_class(spec["model"]), spec["monitor"],
                       spec["config_args"], Path(spec["config_file"]), accelerator=spec.get("accelerator", "auto"), devices="auto",
                       gradient_clip_val=spec["gradient_clip_val"], model_kwargs=spec.get("model_kwargs") or {})
    return 0


if __name__ == "__main__":
    sys.exit(_main(sys.argv[1:]))
