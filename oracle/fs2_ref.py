"""CPU restatement (torch, fp32) of the FastSpeech2 feature-prediction forward path -- TEST INFRASTRUCTURE ONLY.

PARITY UNPINNED: the reference's model code lives in the un-vendored submodule EveryVoiceTTS/FastSpeech2_lightning
(package ``fs2``; ``.gitmodules:1-6``), absent from /root/reference.  What the reference tree does pin, and what this
file follows:
  * hyper-parameters: ``ConformerConfig`` (layers 4, heads 2, input_dim 256, feedforward_dim 1024, conv_kernel_size 9,
    dropout 0.2), ``VariancePredictorConfig`` (n_layers 5, kernel_size 3, dropout 0.5, input_dim 256, n_bins 256,
    depthwise True, level phone), ``FastSpeech2ModelConfig`` (use_postnet True, max_length 1000, learn_alignment True)
    -- ``everyvoice/.schema/everyvoice-text-to-spec-0.5.json`` ($defs ConformerConfig / VariancePredictorConfig /
    FastSpeech2ModelConfig) and ``everyvoice/tests/data/test.ckpt`` ``hyper_parameters``;
  * the ConformerConfig field set is exactly the constructor of ``torchaudio.models.Conformer`` (input_dim, num_heads,
    ffn_dim, num_layers, depthwise_conv_kernel_size, dropout): ConformerRef restates that public module (macaron
    half-step FFNs, pre-norm MHSA, conv module LayerNorm -> pointwise(2d) -> GLU -> depthwise(k) -> BatchNorm -> SiLU
    -> pointwise, final LayerNorm) with its parameter names;
  * ``position_embedding.inv_freq`` [128] -- the only tensor in ``tests/data/test.ckpt``: the FastPitch-style sinusoid
    ``cat(sin(pos * inv_freq), cos(pos * inv_freq))`` with ``inv_freq = 10000^(-2i/256)``;
  * the depthwise-separable convolution of the variance predictors: ``everyvoice/model/utils.py:5-48``
    (weight-normed depthwise Conv1d(C, C, k, groups=C) then weight-normed pointwise Conv1d(C, out, 1));
  * the length regulator primitive ``expand``: ``everyvoice/utils/heavy.py:12-21`` (oracle/heavy_ref.py);
  * variance quantisation between ``stats.{pitch,energy}.norm_min/norm_max`` (``tests/model_stubs.py:50-57``).
Everything else (sub-layer order inside the variance adaptor, postnet = the 5-layer Tacotron-2 postnet, positional
embedding added before encoder and decoder) follows the public upstreams the reference credits (ming024/FastSpeech2,
NVIDIA FastPitch) and is this build's own oracle: golden vectors are generated from it under fixed seeds.
"""

from __future__ import annotations

import math
from dataclasses import dataclass, field

import torch
import torch.nn as nn
import torch.nn.functional as F
from torch.nn.utils import weight_norm

from .heavy_ref import expand_ref


@dataclass
class ConformerConfigRef:
    layers: int = 4
    heads: int = 2
    input_dim: int = 256
    feedforward_dim: int = 1024
    conv_kernel_size: int = 9
    dropout: float = 0.2


@dataclass
class VariancePredictorConfigRef:
    n_layers: int = 5
    kernel_size: int = 3
    dropout: float = 0.5
    input_dim: int = 256
    n_bins: int = 256
    depthwise: bool = True
    level: str = "phone"


@dataclass
class StatsInfoRef:
    min: float = -3.0
    max: float = 3.0
    std: float = 1.0
    mean: float = 0.0
    norm_min: float = -3.0
    norm_max: float = 3.0


N_PHONOLOGICAL_FEATURES = 43  # everyvoice/text/features.py:7


@dataclass
class FastSpeech2ConfigRef:
    encoder: ConformerConfigRef = field(default_factory=ConformerConfigRef)
    decoder: ConformerConfigRef = field(default_factory=ConformerConfigRef)
    energy: VariancePredictorConfigRef = field(default_factory=VariancePredictorConfigRef)
    duration: VariancePredictorConfigRef = field(default_factory=VariancePredictorConfigRef)
    pitch: VariancePredictorConfigRef = field(default_factory=VariancePredictorConfigRef)
    n_symbols: int = 80
    # "characters" / "phones": symbol ids through an embedding table; "phonological_features": 43-dim multi-hot vectors
    # (everyvoice/text/features.py:7 N_PHONOLOGICAL_FEATURES) through a bias-free Linear(43 -> input_dim)
    # (TargetTrainingTextRepresentationLevel, everyvoice/config/type_definitions.py:16-19; the layer itself is in the absent
    # fs2 submodule: restated, parity unpinned)
    target_text_representation_level: str = "characters"
    n_mels: int = 80
    n_speakers: int = 0   # > 0: multispeaker (an embedding added to the encoder output, as ming024/FastSpeech2 does)
    n_languages: int = 0  # > 0: multilingual
    use_postnet: bool = True
    max_length: int = 1000
    postnet_channels: int = 512
    postnet_kernel: int = 5
    postnet_layers: int = 5

    @classmethod
    def small(cls):
        """A configuration the CPU oracle finishes in well under a second (tests)."""
        c = ConformerConfigRef(layers=2, heads=2, input_dim=64, feedforward_dim=128, conv_kernel_size=5)
        v = VariancePredictorConfigRef(n_layers=2, kernel_size=3, input_dim=64, n_bins=16)
        return cls(encoder=c, decoder=ConformerConfigRef(**c.__dict__), energy=v, duration=VariancePredictorConfigRef(**v.__dict__),
                   pitch=VariancePredictorConfigRef(**v.__dict__), n_symbols=20, n_mels=16, postnet_channels=32)


# ---- torchaudio.models.Conformer, restated ------------------------------------------------------------------------
class _ConvolutionModuleRef(nn.Module):
    def __init__(self, d, k, dropout):
        super().__init__()
        self.layer_norm = nn.LayerNorm(d)
        self.sequential = nn.Sequential(
            nn.Conv1d(d, 2 * d, 1, bias=True), nn.GLU(dim=1),
            nn.Conv1d(d, d, k, padding=(k - 1) // 2, groups=d, bias=True),
            nn.BatchNorm1d(d), nn.SiLU(), nn.Conv1d(d, d, 1, bias=True), nn.Dropout(dropout))

    def forward(self, x):  # [B, T, D]
        return self.sequential(self.layer_norm(x).transpose(1, 2)).transpose(1, 2)


class _FeedForwardModuleRef(nn.Module):
    def __init__(self, d, hidden, dropout):
        super().__init__()
        self.sequential = nn.Sequential(nn.LayerNorm(d), nn.Linear(d, hidden), nn.SiLU(), nn.Dropout(dropout),
                                        nn.Linear(hidden, d), nn.Dropout(dropout))

    def forward(self, x):
        return self.sequential(x)


class ConformerLayerRef(nn.Module):
    def __init__(self, d, ffn, heads, k, dropout):
        super().__init__()
        self.ffn1 = _FeedForwardModuleRef(d, ffn, dropout)
        self.self_attn_layer_norm = nn.LayerNorm(d)
        self.self_attn = nn.MultiheadAttention(d, heads, dropout=dropout)
        self.self_attn_dropout = nn.Dropout(dropout)
        self.conv_module = _ConvolutionModuleRef(d, k, dropout)
        self.ffn2 = _FeedForwardModuleRef(d, ffn, dropout)
        self.final_layer_norm = nn.LayerNorm(d)

    def forward(self, x, key_padding_mask):  # x [T, B, D]
        x = self.ffn1(x) * 0.5 + x
        r = x
        x = self.self_attn_layer_norm(x)
        x, _ = self.self_attn(x, x, x, key_padding_mask=key_padding_mask, need_weights=False)
        x = self.self_attn_dropout(x) + r
        x = x + self.conv_module(x.transpose(0, 1)).transpose(0, 1)
        x = self.ffn2(x) * 0.5 + x
        return self.final_layer_norm(x)


class ConformerRef(nn.Module):
    def __init__(self, cfg: ConformerConfigRef):
        super().__init__()
        self.conformer_layers = nn.ModuleList(
            [ConformerLayerRef(cfg.input_dim, cfg.feedforward_dim, cfg.heads, cfg.conv_kernel_size, cfg.dropout) for _ in range(cfg.layers)])

    def forward(self, x, lengths):  # x [B, T, D]
        mask = torch.arange(x.shape[1])[None, :] >= lengths[:, None]
        x = x.transpose(0, 1)
        for layer in self.conformer_layers:
            x = layer(x, mask)
        return x.transpose(0, 1), lengths


# ---- the rest of the model -----------------------------------------------------------------------------------------
class PositionalEmbeddingRef(nn.Module):
    def __init__(self, d):
        super().__init__()
        self.register_buffer("inv_freq", 1.0 / (10000 ** (torch.arange(0.0, d, 2.0) / d)))

    def forward(self, n):
        s = torch.outer(torch.arange(n, dtype=torch.float32), self.inv_freq)
        return torch.cat([s.sin(), s.cos()], dim=1)  # [n, d]


def depthwise_separable_ref(cin, cout, k):
    """everyvoice/model/utils.py:5-48 (non-transposed branch, as the reference intends it)."""
    return nn.Sequential(weight_norm(nn.Conv1d(cin, cin, k, padding=(k - 1) // 2, groups=cin)), weight_norm(nn.Conv1d(cin, cout, 1)))


class VariancePredictorRef(nn.Module):
    def __init__(self, cfg: VariancePredictorConfigRef):
        super().__init__()
        d, k = cfg.input_dim, cfg.kernel_size
        self.convs = nn.ModuleList([depthwise_separable_ref(d, d, k) if cfg.depthwise else nn.Conv1d(d, d, k, padding=(k - 1) // 2)
                                    for _ in range(cfg.n_layers)])
        self.norms = nn.ModuleList([nn.LayerNorm(d) for _ in range(cfg.n_layers)])
        self.dropout = nn.Dropout(cfg.dropout)
        self.linear = nn.Linear(d, 1)

    def forward(self, x, mask):  # x [B, L, D], mask [B, L] True = padding
        for conv, norm in zip(self.convs, self.norms):
            x = self.dropout(norm(F.relu(conv(x.transpose(1, 2)).transpose(1, 2))))
        return self.linear(x).squeeze(-1).masked_fill(mask, 0.0)


class PostNetRef(nn.Module):
    def __init__(self, n_mels, ch, k, layers):
        super().__init__()
        dims = [n_mels] + [ch] * (layers - 1) + [n_mels]
        self.convolutions = nn.ModuleList(
            [nn.Sequential(nn.Conv1d(dims[i], dims[i + 1], k, padding=(k - 1) // 2), nn.BatchNorm1d(dims[i + 1])) for i in range(layers)])

    def forward(self, x):  # [B, T, n_mels]
        x = x.transpose(1, 2)
        for i, c in enumerate(self.convolutions):
            x = c(x)
            if i < len(self.convolutions) - 1:
                x = torch.tanh(x)
        return x.transpose(1, 2)


class FastSpeech2Ref(nn.Module):
    """Inference forward: ids [B, L] (0 = pad), lens [B] -> (mel [B, T, n_mels], postnet mel, durations, pitch, energy, mel_lens)."""

    def __init__(self, cfg: FastSpeech2ConfigRef | None = None, pitch_stats: StatsInfoRef | None = None,
                 energy_stats: StatsInfoRef | None = None):
        super().__init__()
        self.cfg = cfg = cfg or FastSpeech2ConfigRef()
        d = cfg.encoder.input_dim
        self.pfs = cfg.target_text_representation_level == "phonological_features"
        self.text_input_layer = nn.Linear(N_PHONOLOGICAL_FEATURES, d, bias=False) if self.pfs else nn.Embedding(cfg.n_symbols, d, padding_idx=0)
        self.position_embedding = PositionalEmbeddingRef(d)
        self.encoder = ConformerRef(cfg.encoder)
        self.speaker_embedding = nn.Embedding(cfg.n_speakers, d) if cfg.n_speakers else None
        self.language_embedding = nn.Embedding(cfg.n_languages, d) if cfg.n_languages else None
        self.duration_predictor = VariancePredictorRef(cfg.duration)
        self.pitch_predictor = VariancePredictorRef(cfg.pitch)
        self.energy_predictor = VariancePredictorRef(cfg.energy)
        ps, es = pitch_stats or StatsInfoRef(), energy_stats or StatsInfoRef()
        self.register_buffer("pitch_bins", torch.linspace(ps.norm_min, ps.norm_max, cfg.pitch.n_bins - 1))
        self.register_buffer("energy_bins", torch.linspace(es.norm_min, es.norm_max, cfg.energy.n_bins - 1))
        self.pitch_embedding = nn.Embedding(cfg.pitch.n_bins, d)
        self.energy_embedding = nn.Embedding(cfg.energy.n_bins, d)
        self.decoder = ConformerRef(cfg.decoder)
        self.mel_linear = nn.Linear(d, cfg.n_mels)
        self.postnet = PostNetRef(cfg.n_mels, cfg.postnet_channels, cfg.postnet_kernel, cfg.postnet_layers) if cfg.use_postnet else None

    @torch.no_grad()
    def forward(self, ids, lens, duration_control=1.0, pitch_control=1.0, energy_control=1.0, durations=None, speakers=None, languages=None):
        B, L = ids.shape[:2]  # ids [B, L] symbol ids, or [B, L, 43] phonological feature vectors
        pad = torch.arange(L)[None, :] >= lens[:, None]
        x = self.text_input_layer(ids) + self.position_embedding(L)[None]
        x = x.masked_fill(pad[..., None], 0.0)
        x, _ = self.encoder(x, lens)
        if self.speaker_embedding is not None:
            x = x + self.speaker_embedding(speakers)[:, None, :].masked_fill(pad[..., None], 0.0)
        if self.language_embedding is not None:
            x = x + self.language_embedding(languages)[:, None, :].masked_fill(pad[..., None], 0.0)
        log_d = self.duration_predictor(x, pad)
        pitch = self.pitch_predictor(x, pad) * pitch_control
        x = x + self.pitch_embedding(torch.bucketize(pitch, self.pitch_bins))
        energy = self.energy_predictor(x, pad) * energy_control
        x = x + self.energy_embedding(torch.bucketize(energy, self.energy_bins))
        if durations is None:
            durations = torch.clamp(torch.round(torch.exp(log_d) - 1.0) * duration_control, min=0).long()
        durations = durations.masked_fill(pad, 0)
        mel_lens = durations.sum(1)
        T = int(mel_lens.max())
        frames = torch.zeros(B, T, x.shape[2])
        for b in range(B):  # the length regulator: expand (utils/heavy.py:12-21) per item, zero padded to the batch max
            e = torch.from_numpy(expand_ref(x[b].numpy(), durations[b].numpy()))
            frames[b, : e.shape[0]] = e
        fpad = torch.arange(T)[None, :] >= mel_lens[:, None]
        y = (frames + self.position_embedding(T)[None]).masked_fill(fpad[..., None], 0.0)
        y, _ = self.decoder(y, mel_lens)
        mel = self.mel_linear(y).masked_fill(fpad[..., None], 0.0)
        post = mel + self.postnet(mel) if self.postnet is not None else mel
        post = post.masked_fill(fpad[..., None], 0.0)
        return mel, post, durations, pitch, energy, mel_lens


def randomize_norm_stats_(model: nn.Module, gen: torch.Generator):
    """Non-trivial BatchNorm running statistics / affine parameters so that eval-mode folding is actually exercised."""
    for m in model.modules():
        if isinstance(m, nn.BatchNorm1d):
            m.running_mean.copy_(torch.randn(m.num_features, generator=gen) * 0.1)
            m.running_var.copy_(torch.rand(m.num_features, generator=gen) * 0.5 + 0.75)
            m.weight.data.copy_(torch.rand(m.num_features, generator=gen) * 0.5 + 0.75)
            m.bias.data.copy_(torch.randn(m.num_features, generator=gen) * 0.1)


def training_losses_ref(model: FastSpeech2Ref, batch: dict, weights: dict | None = None, aligner=None, hard=None) -> dict:
    """Teacher-forced training forward with torch autograd (``model.train()`` decides dropout / BatchNorm mode):
    ground-truth durations through the length regulator, ground-truth pitch / energy into the embeddings, MSE on
    log-durations / pitch / energy over the unpadded symbols and on mel / postnet mel over the unpadded frames;
    weights = FastSpeech2TrainingConfig defaults of the reference's schema (mel 1, postnet 1, pitch / energy / duration 0.1).
    PARITY UNPINNED like the rest of this file (the training step lives in the absent submodule)."""
    w = {"mel": 1.0, "postnet": 1.0, "pitch": 0.1, "energy": 0.1, "duration": 0.1, "attn_ctc": 0.1, "attn_bin": 0.0}
    w.update(weights or {})
    ids, lens = (batch["pfs"] if getattr(model, "pfs", False) else batch["ids"]), batch["lens"]  # symbol ids, or feature vectors [B, L, 43]
    B, L = ids.shape[:2]
    pad = torch.arange(L)[None, :] >= lens[:, None]
    losses = {}
    if aligner is not None:
        # learn_alignment: soft attention between the target mel and the bare symbol embeddings, hard path by monotonic search
        # (no gradient), durations = frames per symbol; CTC forward-sum + binarisation losses (oracle/alignment_ref.py)
        from .alignment_ref import binarization_loss_ref, forward_sum_loss_ref
        from .mas_ref import maximum_path_batch_ref
        mel_lens_in = batch["mel_lens"]
        Tm = int(mel_lens_in.max())
        soft, logprob = aligner(batch["mel"][:, :Tm].transpose(1, 2), model.text_input_layer(ids).masked_fill(pad[..., None], 0.0).transpose(1, 2), lens,
                                batch.get("attn_prior"))
        if hard is None:
            hard, _ = maximum_path_batch_ref(torch.log(soft.detach()).numpy(), mel_lens_in.numpy(), lens.numpy())
            hard = torch.from_numpy(hard)
        durations = hard.sum(1).long()
        losses["attn_ctc"] = w["attn_ctc"] * forward_sum_loss_ref(logprob, lens, mel_lens_in)
        if w["attn_bin"] > 0:
            losses["attn_bin"] = w["attn_bin"] * binarization_loss_ref(hard, soft)
        cum = torch.cumsum(durations, 1)
        def phone_level(key):  # average_data_by_durations (preprocessor.py:287-300): mean over the symbol's frames, 1e-7 for none
            fr = F.pad(torch.cumsum(batch[key + "_frames"][:, :Tm], 1), (1, 0))
            sums = torch.gather(fr, 1, cum) - torch.gather(fr, 1, cum - durations)
            return torch.where(durations > 0, sums / durations.clamp_min(1), torch.full_like(sums, 1e-7))
        batch = dict(batch, durations=durations, pitch=phone_level("pitch") if "pitch" not in batch else batch["pitch"],
                     energy=phone_level("energy") if "energy" not in batch else batch["energy"])
    durations = batch["durations"]
    x = model.text_input_layer(ids) + model.position_embedding(L)[None]
    x = x.masked_fill(pad[..., None], 0.0)
    x, _ = model.encoder(x, lens)
    if model.speaker_embedding is not None:
        x = x + model.speaker_embedding(batch["speakers"])[:, None, :].masked_fill(pad[..., None], 0.0)
    if model.language_embedding is not None:
        x = x + model.language_embedding(batch["languages"])[:, None, :].masked_fill(pad[..., None], 0.0)
    durations = durations.clamp_min(0).masked_fill(pad, 0)
    pitch_t, energy_t = batch["pitch"].masked_fill(pad, 0.0), batch["energy"].masked_fill(pad, 0.0)
    n_tok = lens.sum()
    log_d = model.duration_predictor(x, pad)
    losses["duration"] = w["duration"] * ((log_d - torch.log(durations.float() + 1.0)) ** 2).sum() / n_tok
    pitch = model.pitch_predictor(x, pad)
    losses["pitch"] = w["pitch"] * ((pitch - pitch_t) ** 2).sum() / n_tok
    x = x + model.pitch_embedding(torch.bucketize(pitch_t, model.pitch_bins))
    energy = model.energy_predictor(x, pad)
    losses["energy"] = w["energy"] * ((energy - energy_t) ** 2).sum() / n_tok
    x = x + model.energy_embedding(torch.bucketize(energy_t, model.energy_bins))
    mel_lens = durations.sum(1)
    T = int(mel_lens.max())
    frames = torch.stack([F.pad(torch.repeat_interleave(x[b], durations[b], dim=0), (0, 0, 0, T - int(mel_lens[b]))) for b in range(B)])
    fpad = torch.arange(T)[None, :] >= mel_lens[:, None]
    y = (frames + model.position_embedding(T)[None]).masked_fill(fpad[..., None], 0.0)
    y, _ = model.decoder(y, mel_lens)
    mel = model.mel_linear(y).masked_fill(fpad[..., None], 0.0)
    target = batch["mel"][:, :T]
    n_el = mel_lens.sum() * mel.shape[2]
    losses["mel"] = w["mel"] * ((mel - target) ** 2).sum() / n_el
    if model.postnet is not None:
        post = (mel + model.postnet(mel)).masked_fill(fpad[..., None], 0.0)
        losses["postnet"] = w["postnet"] * ((post - target) ** 2).sum() / n_el
    losses["total"] = sum(losses.values())
    return losses
