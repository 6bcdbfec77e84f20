"""Data side of the vocoder training path (8e): filelists, the spec / audio dataset, the data module with
the reference's loader policy, and the rank-sharded sampler that data-parallel training draws its utterances through.

Mirrors (paths relative to the reference):
  generic_psv_filelist_reader   everyvoice/utils/__init__.py:306-365   pipe-separated filelist with a header line -> list of dicts
  BaseDataModule                everyvoice/dataloader/__init__.py:16-106  train: batch_size, train_data_workers, pin_memory False,
                                drop_last True, sampler None (Lightning then injects a DistributedSampler under DDP); val:
                                batch_size 1, 0 workers; datasets saved to / loaded from <logger.save_dir>/<logger.name>/{train,val}_data.pth
  SpecDataset, HiFiGANDataModule  hfgl.dataset (absent submodule); semantics pinned by everyvoice/tests/test_dataloader.py:48-70:
                                items are (spec [n_mels, F], audio [S], basename, spec_from_audio [n_mels, F]); with use_segments
                                F = vocoder_segment_size / (hop * output_sr // input_sr) and spec / audio are cropped TOGETHER
                                (get_segments, utils/heavy.py:122-148); ``training.finetune`` reads the input spec from
                                ``synthesized_spec/...spec-pred-<sr>-<type>.pt`` (docs/guides/finetune.md:18-43)
On-disk layout: <save_dir>/<kind>/<basename>--<speaker>--<language>--<file> (preprocessor.py:502-508).
Host-side plumbing only (file IO, cropping, batching); the features themselves come from the GPU preprocessor.
"""

from __future__ import annotations

import csv
import math
import os
import random
from pathlib import Path

import torch
from torch.utils.data import DataLoader, Dataset, Sampler

from .config import HiFiGANConfig
from .heavy import get_segments
from .pipeline import SEP, load_wav


def generic_psv_filelist_reader(path, delimiter="|", fieldnames=None, file_has_header_line=True, record_limit: int = 0) -> list[dict]:
    """Rows of a *sv filelist as dicts (QUOTE_NONE, backslash escapes), as the reference's generic_dict_loader."""
    assert fieldnames is not None or file_has_header_line
    with open(path, "r", newline="", encoding="utf8") as f:
        reader = csv.DictReader(f, fieldnames=fieldnames, delimiter=delimiter, quoting=csv.QUOTE_NONE, escapechar="\\")
        rows = []
        for i, row in enumerate(reader):
            if fieldnames is not None and file_has_header_line and i == 0:
                continue
            rows.append(dict(row))
            if record_limit and len(rows) >= record_limit:
                break
    return rows


def resolve_filelist_loader(name_or_callable):
    """``training.filelist_loader`` is a dotted name in configs; the reference's default maps to the reader above."""
    if callable(name_or_callable):
        return name_or_callable
    if str(name_or_callable).endswith("generic_psv_filelist_reader"):
        return generic_psv_filelist_reader
    import importlib

    mod, _, fn = str(name_or_callable).rpartition(".")
    return getattr(importlib.import_module(mod), fn)


class SpecDataset(Dataset):
    """(spec, audio, basename, spec_from_audio) per utterance for vocoder training."""

    def get_labels(self):
        """What ``use_weighted_sampler`` balances over (dataloader/imbalanced_sampler.py:51-58): the speaker of every item."""
        return [item.get("speaker", "default") for item in self.audio_files]

    def __init__(self, audio_files: list[dict], config: HiFiGANConfig, use_segments: bool = False, finetune: bool | None = None):
        self.config = config
        self.audio_files = audio_files
        self.use_segments = use_segments
        self.finetune = bool(config.training.finetune) if finetune is None else finetune
        a = config.preprocessing.audio
        self.save_dir = Path(config.preprocessing.save_dir)
        self.input_sr, self.output_sr = a.input_sampling_rate, a.output_sampling_rate
        self.spec_type = a.spec_type
        self.segment_size = a.vocoder_segment_size
        self.upsample = self.output_sr // self.input_sr
        self.hop_out = a.fft_hop_size * self.upsample
        self.frames_per_seg = math.ceil(self.segment_size / self.hop_out)

    def __len__(self):
        return len(self.audio_files)

    def _path(self, item: dict, kind: str, fn: str) -> Path:
        return self.save_dir / kind / SEP.join([item["basename"], item.get("speaker", "default"), item.get("language", "default"), fn])

    def __getitem__(self, index: int):
        item = self.audio_files[index]
        audio, sr, _ = load_wav(self._path(item, "audio", f"audio-{self.output_sr}.wav"))
        y = audio[0]
        spec_out = torch.load(self._path(item, "spec", f"spec-{self.output_sr}-{self.spec_type}.pt"), weights_only=True)
        if self.finetune:  # the vocoder learns to invert what the feature-prediction network actually produces
            spec_in = torch.load(self._path(item, "synthesized_spec", f"spec-pred-{self.input_sr}-{self.spec_type}.pt"), weights_only=True)
        elif self.input_sr == self.output_sr:
            spec_in = spec_out
        else:
            spec_in = torch.load(self._path(item, "spec", f"spec-{self.input_sr}-{self.spec_type}.pt"), weights_only=True)
        if self.use_segments:
            # one random frame offset, applied to the input spec, the output spec and (times the hop) to the waveform
            n = min(spec_in.shape[1], spec_out.shape[1])
            if n > self.frames_per_seg:
                spec_in, start = get_segments(spec_in[:, :n], self.frames_per_seg)
                spec_out, _ = get_segments(spec_out[:, :n], self.frames_per_seg, start)
                y, _ = get_segments(y.unsqueeze(0), self.segment_size, start * self.hop_out)
                y = y.squeeze(0)
            else:  # no room for a random offset: the utterance from its start, zero padded on the right up to a segment
                fit = lambda t, size: torch.nn.functional.pad(t[..., :size], (0, max(0, size - t.shape[-1])))  # noqa: E731
                spec_in, spec_out, y = fit(spec_in, self.frames_per_seg), fit(spec_out, self.frames_per_seg), fit(y, self.segment_size)
        return spec_in, y, item["basename"], spec_out


class ShardedSampler(Sampler):
    """DistributedSampler semantics (what Lightning injects when ``train_dataloader`` passes ``sampler=None`` under DDP,
    dataloader/__init__.py:54-68): every epoch a seeded permutation of the dataset, padded by wrap-around to a multiple of the
    world size, of which rank r takes positions r, r + world, ... -- the ranks' shards are disjoint and cover the dataset."""

    def __init__(self, dataset_len: int, rank: int = 0, world: int = 1, shuffle: bool = True, seed: int = 0, drop_last: bool = False):
        if not 0 <= rank < world:
            raise ValueError(f"rank {rank} outside a world of {world}")
        self.n, self.rank, self.world, self.shuffle, self.seed, self.drop_last = dataset_len, rank, world, shuffle, seed, drop_last
        self.epoch = 0
        self.num_samples = self.n // world if drop_last else (self.n + world - 1) // world
        self.total = self.num_samples * world

    def set_epoch(self, epoch: int):
        self.epoch = int(epoch)

    def __len__(self):
        return self.num_samples

    def __iter__(self):
        if self.shuffle:
            g = torch.Generator().manual_seed(self.seed + self.epoch)
            idx = torch.randperm(self.n, generator=g).tolist()
        else:
            idx = list(range(self.n))
        if self.drop_last:
            idx = idx[: self.total]
        elif len(idx) < self.total and idx:
            idx += (idx * math.ceil((self.total - len(idx)) / len(idx)))[: self.total - len(idx)]
        return iter(idx[self.rank : self.total : self.world])


class ImbalancedDatasetSampler(Sampler):
    """everyvoice/dataloader/imbalanced_sampler.py:14-68 (ufoym's imbalanced-dataset-sampler as the reference adapts it): draws
    ``num_samples`` indices with replacement, item i with probability proportional to 1 / (number of items carrying i's label),
    so every label is drawn equally often.  Labels: ``labels``, else ``callback_get_label(dataset)``, else a TensorDataset's
    second tensor, else ``dataset.get_labels()``.  ``seed`` / ``set_epoch`` / ``rank``: a reproducible stream per epoch and per
    data-parallel rank (the reference draws from torch's global generator)."""

    def __init__(self, dataset, labels=None, indices=None, num_samples=None, callback_get_label=None, seed: int | None = None, rank: int = 0):
        from collections import Counter

        self.indices = list(range(len(dataset))) if indices is None else list(indices)
        self.callback_get_label = callback_get_label
        self.num_samples = len(self.indices) if num_samples is None else num_samples
        labels = self._get_labels(dataset) if labels is None else labels
        labels = [x.item() if torch.is_tensor(x) else x for x in labels]
        if len(labels) != len(self.indices):
            raise ValueError(f"{len(labels)} labels for {len(self.indices)} indices")
        # the reference sorts its (index, label) frame by index before taking the weights but draws through the UNSORTED index
        # list (imbalanced_sampler.py:43-49, 61-64): reproduced as is, so that an explicit unsorted `indices` behaves the same
        order = sorted(range(len(self.indices)), key=lambda j: self.indices[j])
        labels = [labels[j] for j in order]
        count = Counter(labels)
        self.weights = torch.tensor([1.0 / count[x] for x in labels], dtype=torch.float64)
        self.seed, self.rank, self.epoch = seed, rank, 0

    def _get_labels(self, dataset):
        if self.callback_get_label:
            return self.callback_get_label(dataset)
        if isinstance(dataset, torch.utils.data.TensorDataset):
            return dataset.tensors[1]
        if isinstance(dataset, Dataset):
            return dataset.get_labels()
        raise NotImplementedError

    def set_epoch(self, epoch: int):
        self.epoch = int(epoch)

    def __iter__(self):
        g = None
        if self.seed is not None:
            g = torch.Generator().manual_seed((self.seed * 1000003 + self.epoch) * 4099 + self.rank)
        return (self.indices[i] for i in torch.multinomial(self.weights, self.num_samples, replacement=True, generator=g).tolist())

    def __len__(self):
        return self.num_samples


def vocoder_collate(batch):
    """list of (spec [M, F], audio [S], basename, spec_from_audio [M, F]) -> (spec [B, M, F], audio [B, S], basenames, spec [B, M, F])."""
    spec, audio, names, spec_out = zip(*batch)
    return torch.stack(spec), torch.stack(audio), list(names), torch.stack(spec_out)


class BaseDataModule:
    """The reference's loader policy without the Lightning base class (duck-typed LightningDataModule: prepare_data, setup,
    train_dataloader, val_dataloader)."""

    def __init__(self, config, inference_output_dir: Path | None = None, rank: int = 0, world: int = 1, seed: int = 1234):
        self.collate_fn = None
        self.config = config
        self.use_weighted_sampler = False
        self.batch_size = config.training.batch_size
        self.rank, self.world, self.seed = rank, world, seed
        self.inference_output_dir = inference_output_dir
        if inference_output_dir is not None:
            Path(inference_output_dir).mkdir(exist_ok=True, parents=True)
            self.predict_path = Path(inference_output_dir) / "latest_predict_data.pth"
        lg = config.training.logger
        self.train_path = os.path.join(lg.save_dir, lg.name, "train_data.pth")
        self.val_path = os.path.join(lg.save_dir, lg.name, "val_data.pth")
        self.train_sampler = None

    def setup(self, stage: str | None = None):
        if stage == "fit":  # datasets prepared and saved by this software (prepare_data)
            self.train_dataset = torch.load(self.train_path, weights_only=False)
            self.val_dataset = torch.load(self.val_path, weights_only=False)
        if stage == "predict":
            self.predict_dataset = torch.load(self.predict_path, weights_only=False)

    def train_dataloader(self):
        # sampler=None in the reference; under DDP every rank must see its own shard, which Lightning arranges by injecting a
        # DistributedSampler: the same thing is done here explicitly.  use_weighted_sampler: label-balanced draws
        # (dataloader/__init__.py:54-58), each rank its own len / world of them
        if self.use_weighted_sampler:
            n = len(self.train_dataset)
            self.train_sampler = ImbalancedDatasetSampler(self.train_dataset, num_samples=(n + self.world - 1) // self.world, seed=self.seed, rank=self.rank)
        else:
            self.train_sampler = ShardedSampler(len(self.train_dataset), self.rank, self.world, shuffle=True, seed=self.seed) if self.world > 1 else None
        return DataLoader(self.train_dataset, batch_size=self.batch_size, num_workers=self.config.training.train_data_workers,
                          pin_memory=False, drop_last=True, collate_fn=self.collate_fn, sampler=self.train_sampler,
                          shuffle=False if self.train_sampler is not None else None)

    def predict_dataloader(self):
        return DataLoader(self.predict_dataset, batch_size=self.batch_size, num_workers=self.config.training.train_data_workers,
                          pin_memory=False, drop_last=False, collate_fn=self.collate_fn)

    def val_dataloader(self):
        sampler = ImbalancedDatasetSampler(self.val_dataset, seed=self.seed) if self.use_weighted_sampler else None  # dataloader/__init__.py:80-84
        return DataLoader(self.val_dataset, batch_size=1, num_workers=0, pin_memory=False, drop_last=True, collate_fn=self.collate_fn, sampler=sampler)

    def prepare_data(self):
        raise NotImplementedError("This method should be implemented by the child class")

    def load_dataset(self):
        raise NotImplementedError("The base data module does not have a method implemented for loading a dataset. "
                                  "Please use another Data Loader that inherits the BaseDataModule class.")


class HiFiGANDataModule(BaseDataModule):
    def __init__(self, config: HiFiGANConfig, **kw):
        super().__init__(config=config, **kw)
        self.use_weighted_sampler = config.training.use_weighted_sampler
        self.collate_fn = vocoder_collate
        self.load_dataset()
        if config.training.finetune:  # only utterances with a synthesised spectrogram can be used (docs/guides/finetune.md)
            keep = lambda item: (Path(config.preprocessing.save_dir) / "synthesized_spec" / SEP.join(  # noqa: E731
                [item["basename"], item.get("speaker", "default"), item.get("language", "default"),
                 f"spec-pred-{config.preprocessing.audio.input_sampling_rate}-{config.preprocessing.audio.spec_type}.pt"])).exists()
            self.train_dataset = [x for x in self.train_dataset if keep(x)]
            self.val_dataset = [x for x in self.val_dataset if keep(x)]

    def load_dataset(self):
        loader = resolve_filelist_loader(self.config.training.filelist_loader)
        self.train_dataset = loader(self.config.training.training_filelist)
        self.val_dataset = loader(self.config.training.validation_filelist)

    def prepare_data(self):
        train = SpecDataset(self.train_dataset, self.config, use_segments=True)
        val = SpecDataset(self.val_dataset, self.config, use_segments=True)
        Path(self.train_path).parent.mkdir(parents=True, exist_ok=True)
        torch.save(train, self.train_path)
        torch.save(val, self.val_path)


def seed_data_workers(seed: int):
    """Python's ``random`` drives the segment crops (get_segments): seed it per process for reproducible runs."""
    random.seed(seed)
