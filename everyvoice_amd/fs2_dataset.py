"""Data side of FastSpeech2 training (the ``FastSpeech2DataModule`` class ``train_base_command`` is handed,
everyvoice/base_cli/helpers.py:26-27, 181): the dataset over a preprocessed directory, its collate function, and the data module
with the reference's loader policy, rank-sharded for data-parallel training.

The reference's classes live in the absent submodule (``fs2.dataset``: FastSpeech2_lightning, .gitmodules:1-3); what is pinned
in-tree and followed here:
  on-disk layout           everyvoice/preprocessor/preprocessor.py:502-508 (``<save_dir>/<kind>/<basename>--<speaker>--<language>--<file>``),
                           :633-670 (``energy/...energy.pt``, ``pitch/...pitch.pt``: frame level, or averaged per symbol when the level is
                           "phone" and durations are given), :672-740 (``attn/...{characters|phones}-attn-prior.pt``: float64
                           [frames, tokens]), :1176 (``pfs/...pfs.pt``), duration files ``duration/...duration.pt``
                           (tests/data/lj/preprocessed/duration/*.pt)
  filelist columns         ``character_tokens`` / ``phone_tokens``: "/"-joined token strings written by the text stage
                           (preprocessor.py:745-870; text/text_processor.py:501-511 split_tokens, "<SLASH>" substitution)
  token ids                text/text_processor.py:119-134: pad symbol "\\x80" = 0, " " = 1, then the symbols by (-len, symbol)
  dataset filtering        everyvoice/utils/__init__.py:61-97 (items without the target representation are dropped; fewer items than a
                           batch is an error -> SystemExit(1))
  collate                  everyvoice/utils/heavy.py:24-36 (dict of lists; tensors padded with 0 to the longest; ints -> IntTensor)
  loader policy            everyvoice/dataloader/__init__.py:16-106 (BaseDataModule)
Text normalisation, g2p and the symbol inventory itself belong to the reference's text front-end (out of scope): the dataset
receives token strings and a ``TokenTable``.  Host-side plumbing only; the features come from the GPU preprocessor
(``pipeline.GpuPreprocessor``: spec, energy, pitch, attention priors by ``evmi_attention_prior_f64``).
"""

from __future__ import annotations

import sys
from pathlib import Path

import torch
from torch.utils.data import Dataset

from .dataset import BaseDataModule, resolve_filelist_loader
from .heavy import collate_fn
from .pipeline import SEP

PAD_SYMBOL, CHARACTER_JOINER, JOINER_SUBSTITUTION = "\x80", "/", "<SLASH>"
TEXT_KEYS = {"characters": "character_tokens", "ipa_phones": "phone_tokens", "phones": "phone_tokens", "phonological_features": "phone_tokens"}
PRIOR_NAMES = {"character_tokens": "characters", "phone_tokens": "phones"}


def split_tokens(joined: str) -> list[str]:
    """text/text_processor.py:501-511."""
    return [x.replace(JOINER_SUBSTITUTION, CHARACTER_JOINER) for x in joined.split(CHARACTER_JOINER)]


class OutOfVocabularySymbolError(KeyError):
    pass


class TokenTable:
    """Symbol -> id with the reference's id assignment (text_processor.py:119-134): [pad, " "] + sorted(rest, key=(-len, symbol))."""

    def __init__(self, symbols):
        rest = set(symbols) - {PAD_SYMBOL, " "}
        self.symbols = [PAD_SYMBOL, " "] + sorted(rest, key=lambda s: (-len(s), s))
        self._id = {s: i for i, s in enumerate(self.symbols)}

    def __len__(self):
        return len(self.symbols)

    def encode_string_tokens(self, tokens: list[str]) -> list[int]:
        try:
            return [self._id[t] for t in tokens]
        except KeyError as e:
            raise OutOfVocabularySymbolError(f"Sequence {tokens} contains item {e.args[0]!r}") from e

    def to_json(self) -> list[str]:
        return list(self.symbols)


def filter_dataset_based_on_target_text_representation_level(level: str, dataset: list[dict], name: str, batch_size: int) -> list[dict]:
    """everyvoice/utils/__init__.py:61-97."""
    try:
        key = TEXT_KEYS[str(getattr(level, "value", level))]
    except KeyError:
        raise NotImplementedError(f"{level} have not yet been implemented.") from None
    kept = [item for item in dataset if key in item and item[key]]
    if len(kept) != len(dataset):
        print(f"Removing {len(dataset) - len(kept)} from your {name} set because they do not have text values for the target training "
              f"representation level {level}.", file=sys.stderr)
    if batch_size > len(kept):
        print(f"Sorry you do not have enough {level} data in your current {name} filelist to run the model with a batch size of {batch_size}.",
              file=sys.stderr)
        sys.exit(1)
    return kept


class FastSpeech2Dataset(Dataset):
    """One utterance -> the dict the FastSpeech2 step consumes (before collation):
    ``text`` [L] ids (or ``pfs`` [L, 43]), ``mel`` [T, n_mels], ``duration`` [L] | ``attn_prior`` [T, L] float64,
    ``pitch`` / ``energy`` ([L] phone level with given durations, [T] frame level under alignment learning), ``speaker_id`` /
    ``language_id`` ints, ``basename`` / ``speaker`` / ``language`` / ``raw_text`` strings."""

    def __init__(self, dataset: list[dict], config, token_table: TokenTable | None, lang2id: dict | None = None, speaker2id: dict | None = None):
        self.dataset, self.config, self.tokens = dataset, config, token_table
        m, a = config.model, config.preprocessing.audio
        self.level = str(getattr(m.target_text_representation_level, "value", m.target_text_representation_level))
        self.text_key = TEXT_KEYS[self.level]
        self.pfs = self.level == "phonological_features"
        self.learn_alignment = bool(m.learn_alignment)
        # phone- or frame-level variance files: decided by the configuration (the preprocessor writes per-symbol averages only with
        # given durations AND a "phone"-level predictor, preprocessor.py:641-669), never guessed from a tensor's length
        vp = m.variance_predictors
        self.phone_level = {key: (not self.learn_alignment) and str(getattr(getattr(vp, key).level, "value", getattr(vp, key).level)) == "phone"
                            for key in ("pitch", "energy")}
        self.save_dir = Path(config.preprocessing.save_dir)
        self.spec_fn = f"spec-{a.input_sampling_rate}-{a.spec_type}.pt"
        self.lang2id, self.speaker2id = dict(lang2id or {}), dict(speaker2id or {})
        if not self.pfs and token_table is None:
            raise ValueError("FastSpeech2Dataset needs a TokenTable for character / phone training (ids come from the text front-end's symbol set)")

    def __len__(self):
        return len(self.dataset)

    def get_labels(self):
        """What ImbalancedDatasetSampler balances over (dataloader/imbalanced_sampler.py:51-58): the speaker of every item."""
        return [item.get("speaker", "default") for item in self.dataset]

    def _load(self, item: dict, kind: str, fn: str) -> torch.Tensor:
        path = self.save_dir / kind / SEP.join([item["basename"], item.get("speaker", "default"), item.get("language", "default"), fn])
        return torch.load(path, weights_only=True)

    def __getitem__(self, index: int) -> dict:
        item = self.dataset[index]
        speaker, language = item.get("speaker", "default"), item.get("language", "default")
        mel = self._load(item, "spec", self.spec_fn).transpose(0, 1).contiguous()  # [n_mels, T] on disk -> [T, n_mels]
        T = mel.shape[0]
        out = {"basename": item["basename"], "speaker": speaker, "language": language, "raw_text": item.get("characters", item.get("raw_text", "")),
               "speaker_id": int(self.speaker2id.get(speaker, 0)), "language_id": int(self.lang2id.get(language, 0)), "mel": mel}
        tokens = split_tokens(item[self.text_key])
        if self.pfs:
            out["pfs"] = self._load(item, "pfs", "pfs.pt").to(torch.float32)
            L = out["pfs"].shape[0]
        else:
            out["text"] = torch.tensor(self.tokens.encode_string_tokens(tokens), dtype=torch.long)
            L = out["text"].shape[0]
        if self.learn_alignment:
            prior = self._load(item, "attn", f"{PRIOR_NAMES[self.text_key]}-attn-prior.pt")
            if tuple(prior.shape) != (T, L):
                raise ValueError(f"{item['basename']}: attention prior {tuple(prior.shape)} does not match (frames {T}, tokens {L})")
            out["attn_prior"] = prior.to(torch.float64)
        else:
            dur = self._load(item, "duration", "duration.pt").to(torch.long)
            if dur.shape[0] != L or abs(int(dur.sum()) - T) > 10:  # tests/test_preprocessing.py:527 allows the aligner 10 frames of slack
                raise ValueError(f"{item['basename']}: durations ({dur.shape[0]} symbols, {int(dur.sum())} frames) do not fit {L} tokens / {T} frames")
            over = int(dur.sum()) - T
            if over > 0:  # within the slack, but the step would count zero-padded frames as targets: take the excess off the tail
                for i in range(L - 1, -1, -1):
                    cut = min(over, int(dur[i]))
                    dur[i] -= cut
                    over -= cut
                    if over == 0:
                        break
            out["duration"] = dur
        # frame-level files under alignment learning (durations are not known at preprocessing time, preprocessor.py:641-669);
        # with given durations the files hold one value per symbol when the predictor's level is "phone", one per frame otherwise
        for key in ("pitch", "energy"):
            v = self._load(item, key, f"{key}.pt").to(torch.float32)
            want = L if self.phone_level[key] else T
            if not self.phone_level[key] and v.shape[0] != want and abs(v.shape[0] - want) <= 1 and v.shape[0] > 0:
                # frame-level values from pyworld's DIO have int(n / fs / frame_period_ms * 1000) + 1 frames, computed in float
                # milliseconds (preprocessor.py:244-285): one more or one fewer than the centred STFT's n // hop + 1 on some
                # files.  The reference trains on those; here: drop the extra value / repeat the last one.
                v = v[:want] if v.shape[0] > want else torch.cat([v, v[-1:]])
            if v.shape[0] != want:
                raise ValueError(f"{item['basename']}: {key} has {v.shape[0]} values, the configuration "
                                 f"({'phone' if self.phone_level[key] else 'frame'} level) wants {want}")
            out[key if self.phone_level[key] else key + "_frames"] = v
        return out


def fs2_collate(items: list[dict]) -> dict:
    """heavy.collate_fn (utils/heavy.py:24-36) + the names and lengths the FastSpeech2 step reads: ``ids`` / ``pfs``, ``lens``,
    ``mel`` [B, T, n_mels], ``mel_lens``, ``durations`` | ``attn_prior`` [B, T, L], ``pitch`` / ``energy`` (phone level [B, L]) or
    ``pitch_frames`` / ``energy_frames`` [B, T], ``speakers`` / ``languages`` [B]."""
    lens = torch.tensor([int((it["pfs"] if "pfs" in it else it["text"]).shape[0]) for it in items], dtype=torch.long)
    mel_lens = torch.tensor([int(it["mel"].shape[0]) for it in items], dtype=torch.long)
    priors = [it.get("attn_prior") for it in items]  # ragged along both axes: padded below, not by pad_sequence
    b = collate_fn([{k: v for k, v in it.items() if k != "attn_prior"} for it in items])
    out = {"lens": lens, "mel_lens": mel_lens, "mel": b["mel"], "basename": b["basename"], "speaker": b["speaker"], "language": b["language"],
           "speakers": b["speaker_id"], "languages": b["language_id"]}
    if "pfs" in b:
        out["pfs"] = b["pfs"]
    else:
        out["ids"] = b["text"]
    L, T = int(lens.max()), int(mel_lens.max())
    if priors[0] is not None:
        out["attn_prior"] = torch.zeros(len(items), T, L, dtype=torch.float64)
        for i, p in enumerate(priors):
            out["attn_prior"][i, : p.shape[0], : p.shape[1]] = p
    else:
        out["durations"] = b["duration"]
    for key in ("pitch", "energy", "pitch_frames", "energy_frames"):
        if key in b:
            out[key] = b[key]
    return out


class FastSpeech2DataModule(BaseDataModule):
    """``FastSpeech2DataModule(config)`` as ``train_base_command`` constructs it (helpers.py:271); ``rank`` / ``world`` shard the
    utterances across data-parallel ranks (the DistributedSampler Lightning injects)."""

    def __init__(self, config, token_table: TokenTable | None = None, lang2id: dict | None = None, speaker2id: dict | None = None, **kw):
        super().__init__(config=config, **kw)
        self.collate_fn = fs2_collate
        self.use_weighted_sampler = bool(getattr(config.training, "use_weighted_sampler", False))
        self.token_table = token_table
        if token_table is None and getattr(config, "symbols", None):
            self.token_table = TokenTable(config.symbols)
        self.lang2id, self.speaker2id = lang2id, speaker2id
        self.load_dataset()
        level = config.model.target_text_representation_level
        self.train_dataset = filter_dataset_based_on_target_text_representation_level(level, self.train_dataset, "training", self.batch_size)
        self.val_dataset = filter_dataset_based_on_target_text_representation_level(level, self.val_dataset, "validation", 1)
        if self.speaker2id is None:
            self.speaker2id = {s: i for i, s in enumerate(sorted({x.get("speaker", "default") for x in self.train_dataset + self.val_dataset}))}
        if self.lang2id is None:
            self.lang2id = {s: i for i, s in enumerate(sorted({x.get("language", "default") for x in self.train_dataset + self.val_dataset}))}

    def load_dataset(self):
        loader = resolve_filelist_loader(self.config.training.filelist_loader)
        self.train_dataset = loader(self.config.training.training_filelist)
        self.val_dataset = loader(self.config.training.validation_filelist)

    def prepare_data(self):
        train = FastSpeech2Dataset(self.train_dataset, self.config, self.token_table, self.lang2id, self.speaker2id)
        val = FastSpeech2Dataset(self.val_dataset, self.config, self.token_table, self.lang2id, self.speaker2id)
        Path(self.train_path).parent.mkdir(parents=True, exist_ok=True)
        torch.save(train, self.train_path)
        torch.save(val, self.val_path)
