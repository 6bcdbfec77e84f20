"""FastSpeech2 feature prediction (text ids -> mel), inference forward path on libevmi_hip.

Host-side mirror of what the reference constructs as ``FastSpeech2(config, stats=..., lang2id=..., speaker2id=...)``
(``everyvoice/tests/model_stubs.py:44-58``; the module itself lives in the absent submodule
EveryVoiceTTS/FastSpeech2_lightning).  Configuration names follow the reference's schema
(``everyvoice/.schema/everyvoice-text-to-spec-0.5.json``: ConformerConfig {layers, heads, input_dim, feedforward_dim,
conv_kernel_size, dropout}, VariancePredictorConfig {n_layers, kernel_size, input_dim, n_bins, depthwise, level}); the
state-dict layout is the one of ``torchaudio.models.Conformer`` for encoder / decoder (the ConformerConfig field set is
exactly that constructor) plus ``text_input_layer``, ``position_embedding.inv_freq`` (the tensor
``everyvoice/tests/data/test.ckpt`` holds), ``{duration,pitch,energy}_predictor``, ``{pitch,energy}_embedding``,
``mel_linear`` and ``postnet``.

Everything runs channel-major ``x[c][b][t]`` in fp32: Linear / pointwise / postnet layers on the fp32 matrix-core
implicit GEMM (``evmi_conv1d_cbt_f32``), attention, LayerNorm, depthwise convolutions (eval-mode BatchNorm and weight norm
folded at load time), embeddings, bucketisation, the integer length regulator in ``csrc/fs2_ops.hip``.  No CPU fallback.
"""

from __future__ import annotations

from dataclasses import dataclass, field

import torch

from . import _lib
from .train import ops


@dataclass
class ConformerConfig:
    layers: int = 4
    heads: int = 2
    input_dim: int = 256
    feedforward_dim: int = 1024
    conv_kernel_size: int = 9
    dropout: float = 0.2


@dataclass
class VariancePredictorConfig:
    loss: str = "mse"
    n_layers: int = 5
    kernel_size: int = 3
    dropout: float = 0.5
    input_dim: int = 256
    n_bins: int = 256
    depthwise: bool = True
    level: str = "phone"


@dataclass
class StatsInfo:
    """``everyvoice/tests/model_stubs.py:50-57``."""
    min: float = -3.0
    max: float = 3.0
    std: float = 1.0
    mean: float = 0.0
    norm_min: float = -3.0
    norm_max: float = 3.0


@dataclass
class Stats:
    pitch: StatsInfo = field(default_factory=StatsInfo)
    energy: StatsInfo = field(default_factory=StatsInfo)


@dataclass
class VariancePredictors:
    energy: VariancePredictorConfig = field(default_factory=VariancePredictorConfig)
    duration: VariancePredictorConfig = field(default_factory=VariancePredictorConfig)
    pitch: VariancePredictorConfig = field(default_factory=VariancePredictorConfig)


@dataclass
class FastSpeech2ModelConfig:
    encoder: ConformerConfig = field(default_factory=ConformerConfig)
    decoder: ConformerConfig = field(default_factory=ConformerConfig)
    variance_predictors: VariancePredictors = field(default_factory=VariancePredictors)
    learn_alignment: bool = True
    max_length: int = 1000
    mel_loss: str = "mse"
    use_postnet: bool = True
    multilingual: bool = False
    multispeaker: bool = False
    # everyvoice-text-to-spec-0.5.json:265-272 (TargetTrainingTextRepresentationLevel): "characters" / "phones" = symbol ids through
    # the embedding table; "phonological_features" = 43-dim multi-hot vectors (text/features.py:7) through a bias-free Linear
    target_text_representation_level: str = "characters"
    # not in the reference's schema (fixed inside the absent module): sizes of the embedding table and the postnet
    n_symbols: int = 80
    n_mels: int = 80
    postnet_channels: int = 512
    postnet_kernel: int = 5
    postnet_layers: int = 5
    n_speakers: int = 0   # table sizes when multispeaker / multilingual (len(speaker2id) / len(lang2id) in the reference)
    n_languages: int = 0


N_PHONOLOGICAL_FEATURES = 43  # everyvoice/text/features.py:7

_LN_EPS = 1e-5
_BN_EPS = 1e-5


def _chk(rc, what):
    _lib.check(rc, what)


def _s(t):
    return _lib.current_stream_ptr(t.device)


def _conv(x, w, b, k=1, act=ops.ACT_NONE):
    return ops.conv1d_mfma(x, w, b, 1, (k - 1) // 2, 1, 1, act=act)


def _conv_add(x, w, b, res, k=1):
    """res + conv(x): the residual is added in the convolution's epilogue on the packed bf16 kernels (one pass over the output
    instead of three); elsewhere the convolution and the sum run as before."""
    return ops.conv1d_fused_fwd(x, w, b, 1, (k - 1) // 2, 1, 1, residual=res)


def _layernorm(x, g, b):
    y = torch.empty_like(x)
    _chk(_lib.load().evmi_layernorm_cbt_f32(x.data_ptr(), g.data_ptr(), b.data_ptr(), y.data_ptr(), x.shape[0], x.shape[1] * x.shape[2],
                                            _LN_EPS, _s(x)), "evmi_layernorm_cbt_f32")
    return y


def _ln_conv(x, g, b, w, bias, act=ops.ACT_NONE):
    """act(conv1x1(LayerNorm(x))): on the packed bf16 kernels the normalised tensor is written straight into the convolution's packed input
    (ops.layernorm_dense_fwd: one launch and one fp32 round trip of the tensor less); elsewhere the two operators."""
    C, B, T = x.shape
    # (a position-wise layer: all columns as ONE item -- any B * T then shares the packed layout, csrc/conv_pk_common.h)
    if ops._packed() and ops.ln_dense_fused_supported(1, B * T, C, w.shape[0]):
        return ops.layernorm_dense_fwd(x.reshape(C, 1, B * T), g, b, w, bias, {}, act=act, eps=_LN_EPS).view(-1, B, T)
    if ops.ln_dense_fused_supported(B, T, C, w.shape[0]):
        return ops.layernorm_dense_fwd(x, g, b, w, bias, {}, act=act, eps=_LN_EPS)
    return _conv(_layernorm(x, g, b), w, bias, act=act)


def _dwconv(x, w, b, k, act):
    y = torch.empty_like(x)
    C, B, T = x.shape
    _chk(_lib.load().evmi_dwconv1d_cbt_f32(x.data_ptr(), w.data_ptr(), b.data_ptr(), y.data_ptr(), C, B, T, k, (k - 1) // 2, act, _s(x)),
         "evmi_dwconv1d_cbt_f32")
    return y


def _fold_weight_norm(sd, prefix):
    g, v = sd[prefix + ".weight_g"].float(), sd[prefix + ".weight_v"].float()
    return g * v / v.flatten(1).norm(dim=1).view(-1, *([1] * (v.dim() - 1)))


def _fold_bn(w, b, sd, prefix):
    """Eval-mode BatchNorm1d after a convolution: y = (conv - mean) / sqrt(var + eps) * gamma + beta."""
    s = sd[prefix + ".weight"].float() / torch.sqrt(sd[prefix + ".running_var"].float() + _BN_EPS)
    return w * s.view(-1, *([1] * (w.dim() - 1))), (b - sd[prefix + ".running_mean"].float()) * s + sd[prefix + ".bias"].float()


class _Conformer:
    def __init__(self, cfg: ConformerConfig, sd: dict, prefix: str, dev):
        self.cfg = cfg
        self.layers = []
        d = cfg.input_dim
        put = lambda t: t.to(dev, torch.float32).contiguous()
        for i in range(cfg.layers):
            p = f"{prefix}.conformer_layers.{i}."
            L = {}
            for name in ("ffn1", "ffn2"):
                L[name] = dict(ln_g=put(sd[p + name + ".sequential.0.weight"]), ln_b=put(sd[p + name + ".sequential.0.bias"]),
                               w1=put(sd[p + name + ".sequential.1.weight"].unsqueeze(-1)), b1=put(sd[p + name + ".sequential.1.bias"]),
                               # the block's 0.5 (x + 0.5 * ffn(x)) is folded into the second layer: scaling by a power of two is
                               # exact, so res + conv(h, w2 / 2, b2 / 2) rounds exactly as x + 0.5 * conv(h, w2, b2) did
                               w2=put(0.5 * sd[p + name + ".sequential.4.weight"].float().unsqueeze(-1)),
                               b2=put(0.5 * sd[p + name + ".sequential.4.bias"].float()))
            L["attn"] = dict(ln_g=put(sd[p + "self_attn_layer_norm.weight"]), ln_b=put(sd[p + "self_attn_layer_norm.bias"]),
                             w_in=put(sd[p + "self_attn.in_proj_weight"].unsqueeze(-1)), b_in=put(sd[p + "self_attn.in_proj_bias"]),
                             w_out=put(sd[p + "self_attn.out_proj.weight"].unsqueeze(-1)), b_out=put(sd[p + "self_attn.out_proj.bias"]))
            c = p + "conv_module."
            wdw, bdw = _fold_bn(sd[c + "sequential.2.weight"].float(), sd[c + "sequential.2.bias"].float(), sd, c + "sequential.3")
            L["conv"] = dict(ln_g=put(sd[c + "layer_norm.weight"]), ln_b=put(sd[c + "layer_norm.bias"]),
                             w_pw1=put(sd[c + "sequential.0.weight"]), b_pw1=put(sd[c + "sequential.0.bias"]),
                             w_dw=put(wdw.reshape(d, -1)), b_dw=put(bdw),
                             w_pw2=put(sd[c + "sequential.5.weight"]), b_pw2=put(sd[c + "sequential.5.bias"]))
            L["final"] = dict(g=put(sd[p + "final_layer_norm.weight"]), b=put(sd[p + "final_layer_norm.bias"]))
            self.layers.append(L)

    def _ffn(self, x, P):
        C, B, T = x.shape
        if ops._packed() and ops.ffn_packed_supported(1, B * T, C, P["w1"].shape[0], P["w2"].shape[0]):
            # the block as a packed chain (train/ops.py: ffn_packed_infer): the 1024-channel tensor exists as packed bf16 only
            xr = x.reshape(C, 1, B * T)
            return ops.ffn_packed_infer(xr, P["ln_g"], P["ln_b"], P["w1"], P["b1"], P["w2"], P["b2"], xr, eps=_LN_EPS).view(C, B, T)
        h = _ln_conv(x, P["ln_g"], P["ln_b"], P["w1"], P["b1"], act=ops.ACT_SILU)
        return _conv_add(h, P["w2"], P["b2"], x)

    def forward(self, x, lens):
        """x [D, B, T] (padded positions are computed, not masked: the convolution module sees them, as in the reference)."""
        lib = _lib.load()
        D, B, T = x.shape
        for L in self.layers:
            x = self._ffn(x, L["ffn1"])
            A = L["attn"]
            qkv = _ln_conv(x, A["ln_g"], A["ln_b"], A["w_in"], A["b_in"])
            att = torch.empty_like(x)
            attn = lib.evmi_attention_cbt_bf16 if ops.CONV_BACKEND["operands"] == "bf16" else lib.evmi_attention_cbt_f32
            _chk(attn(qkv.data_ptr(), lens.data_ptr(), att.data_ptr(), B, T, D, self.cfg.heads, _s(x)), "evmi_attention_cbt")
            x = _conv_add(att, A["w_out"], A["b_out"], x)
            Cm = L["conv"]
            p = _ln_conv(x, Cm["ln_g"], Cm["ln_b"], Cm["w_pw1"], Cm["b_pw1"])
            g = ops.elementwise(15, p[:D], p[D:])  # GLU over the channel halves
            h = _dwconv(g, Cm["w_dw"], Cm["b_dw"], self.cfg.conv_kernel_size, 1)  # depthwise + folded BatchNorm + SiLU
            x = _conv_add(h, Cm["w_pw2"], Cm["b_pw2"], x)
            x = self._ffn(x, L["ffn2"])
            x = _layernorm(x, L["final"]["g"], L["final"]["b"])
        return x


class _VariancePredictor:
    def __init__(self, cfg: VariancePredictorConfig, sd: dict, prefix: str, dev):
        self.cfg = cfg
        put = lambda t: t.to(dev, torch.float32).contiguous()
        self.layers = []
        d = cfg.input_dim
        for i in range(cfg.n_layers):
            p = f"{prefix}.convs.{i}"
            if cfg.depthwise:
                conv = dict(w_dw=put(_fold_weight_norm(sd, p + ".0").reshape(d, -1)), b_dw=put(sd[p + ".0.bias"]),
                            w_pw=put(_fold_weight_norm(sd, p + ".1")), b_pw=put(sd[p + ".1.bias"]))
            else:
                conv = dict(w=put(sd[p + ".weight"]), b=put(sd[p + ".bias"]))
            conv.update(ln_g=put(sd[f"{prefix}.norms.{i}.weight"]), ln_b=put(sd[f"{prefix}.norms.{i}.bias"]))
            self.layers.append(conv)
        self.w_lin, self.b_lin = put(sd[prefix + ".linear.weight"].unsqueeze(-1)), put(sd[prefix + ".linear.bias"])

    def forward(self, x, lens):
        """x [D, B, L] -> [B, L] (zero at padded positions)."""
        k = self.cfg.kernel_size
        for P in self.layers:
            h = (_conv(_dwconv(x, P["w_dw"], P["b_dw"], k, 0), P["w_pw"], P["b_pw"], act=ops.ACT_RELU) if self.cfg.depthwise
                 else _conv(x, P["w"], P["b"], k, act=ops.ACT_RELU))
            x = _layernorm(h, P["ln_g"], P["ln_b"])
        y = _conv(x, self.w_lin, self.b_lin)  # [1, B, L]
        _chk(_lib.load().evmi_mask_cols_f32(y.data_ptr(), lens.data_ptr(), 1, y.shape[1], y.shape[2], _s(y)), "evmi_mask_cols_f32")
        return y[0]


class FastSpeech2:
    """``model = FastSpeech2(config, stats); model.load_state_dict(sd); mel, post, durations, pitch, energy, mel_lens = model(ids, lens)``"""

    def __init__(self, config: FastSpeech2ModelConfig | None = None, stats: Stats | None = None, device="cuda:0",
                 lang2id: dict | None = None, speaker2id: dict | None = None, precision: str = "f32"):
        if precision not in ("f32", "bf16"):
            raise ValueError("precision: 'f32' (exact fp32 arithmetic) or 'bf16' (bf16 operands of the dense layers, fp32 accumulation)")
        self.precision = precision
        self.config = config or FastSpeech2ModelConfig()
        self.stats = stats or Stats()
        self.device = torch.device(device)
        self.lang2id, self.speaker2id = lang2id or {}, speaker2id or {}
        if self.device.type != "cuda":
            raise RuntimeError("FastSpeech2 runs on libevmi_hip (MI355X) only; there is no CPU path")
        _lib.load()
        self._ready = False

    # -- parameters -----------------------------------------------------------------------------------------------
    def load_state_dict(self, sd: dict):
        c, dev = self.config, self.device
        put = lambda t: t.to(dev, torch.float32).contiguous()
        self.table = put(sd["text_input_layer.weight"])
        self.inv_freq = put(sd["position_embedding.inv_freq"])
        self.encoder = _Conformer(c.encoder, sd, "encoder", dev)
        self.decoder = _Conformer(c.decoder, sd, "decoder", dev)
        self.speaker_table = put(sd["speaker_embedding.weight"]) if c.multispeaker else None
        self.language_table = put(sd["language_embedding.weight"]) if c.multilingual else None
        vp = c.variance_predictors
        self.duration_predictor = _VariancePredictor(vp.duration, sd, "duration_predictor", dev)
        self.pitch_predictor = _VariancePredictor(vp.pitch, sd, "pitch_predictor", dev)
        self.energy_predictor = _VariancePredictor(vp.energy, sd, "energy_predictor", dev)
        lin = lambda st, n: torch.linspace(st.norm_min, st.norm_max, n - 1)
        self.pitch_bins = put(sd["pitch_bins"] if "pitch_bins" in sd else lin(self.stats.pitch, vp.pitch.n_bins))
        self.energy_bins = put(sd["energy_bins"] if "energy_bins" in sd else lin(self.stats.energy, vp.energy.n_bins))
        self.pitch_table, self.energy_table = put(sd["pitch_embedding.weight"]), put(sd["energy_embedding.weight"])
        self.w_mel, self.b_mel = put(sd["mel_linear.weight"].unsqueeze(-1)), put(sd["mel_linear.bias"])
        self.postnet = []
        if c.use_postnet:
            for i in range(c.postnet_layers):
                p = f"postnet.convolutions.{i}"
                w, b = _fold_bn(sd[p + ".0.weight"].float(), sd[p + ".0.bias"].float(), sd, p + ".1")
                self.postnet.append((put(w), put(b)))
        self._ready = True
        return self

    @staticmethod
    def state_dict_shapes(c: FastSpeech2ModelConfig) -> dict:
        """Names and shapes of the state dict (``torchaudio.models.Conformer`` layout for encoder / decoder)."""
        pfs = c.target_text_representation_level == "phonological_features"
        shapes = {"text_input_layer.weight": (c.encoder.input_dim, N_PHONOLOGICAL_FEATURES) if pfs else (c.n_symbols, c.encoder.input_dim),
                  "position_embedding.inv_freq": (c.encoder.input_dim // 2,)}
        for name, cf in (("encoder", c.encoder), ("decoder", c.decoder)):
            d, f, k = cf.input_dim, cf.feedforward_dim, cf.conv_kernel_size
            for i in range(cf.layers):
                p = f"{name}.conformer_layers.{i}."
                for ffn in ("ffn1", "ffn2"):
                    shapes.update({p + ffn + ".sequential.0.weight": (d,), p + ffn + ".sequential.0.bias": (d,),
                                   p + ffn + ".sequential.1.weight": (f, d), p + ffn + ".sequential.1.bias": (f,),
                                   p + ffn + ".sequential.4.weight": (d, f), p + ffn + ".sequential.4.bias": (d,)})
                shapes.update({p + "self_attn_layer_norm.weight": (d,), p + "self_attn_layer_norm.bias": (d,),
                               p + "self_attn.in_proj_weight": (3 * d, d), p + "self_attn.in_proj_bias": (3 * d,),
                               p + "self_attn.out_proj.weight": (d, d), p + "self_attn.out_proj.bias": (d,),
                               p + "conv_module.layer_norm.weight": (d,), p + "conv_module.layer_norm.bias": (d,),
                               p + "conv_module.sequential.0.weight": (2 * d, d, 1), p + "conv_module.sequential.0.bias": (2 * d,),
                               p + "conv_module.sequential.2.weight": (d, 1, k), p + "conv_module.sequential.2.bias": (d,),
                               p + "conv_module.sequential.3.weight": (d,), p + "conv_module.sequential.3.bias": (d,),
                               p + "conv_module.sequential.3.running_mean": (d,), p + "conv_module.sequential.3.running_var": (d,),
                               p + "conv_module.sequential.5.weight": (d, d, 1), p + "conv_module.sequential.5.bias": (d,),
                               p + "final_layer_norm.weight": (d,), p + "final_layer_norm.bias": (d,)})
        vp = c.variance_predictors
        for name, cf in (("duration_predictor", vp.duration), ("pitch_predictor", vp.pitch), ("energy_predictor", vp.energy)):
            d, k = cf.input_dim, cf.kernel_size
            for i in range(cf.n_layers):
                p = f"{name}.convs.{i}"
                shapes.update({p + ".0.weight_g": (d, 1, 1), p + ".0.weight_v": (d, 1, k), p + ".0.bias": (d,),
                               p + ".1.weight_g": (d, 1, 1), p + ".1.weight_v": (d, d, 1), p + ".1.bias": (d,),
                               f"{name}.norms.{i}.weight": (d,), f"{name}.norms.{i}.bias": (d,)})
            shapes.update({name + ".linear.weight": (1, d), name + ".linear.bias": (1,)})
        d = c.encoder.input_dim
        if c.multispeaker:
            shapes["speaker_embedding.weight"] = (max(1, c.n_speakers), d)
        if c.multilingual:
            shapes["language_embedding.weight"] = (max(1, c.n_languages), d)
        shapes.update({"pitch_embedding.weight": (vp.pitch.n_bins, d), "energy_embedding.weight": (vp.energy.n_bins, d),
                       "mel_linear.weight": (c.n_mels, c.decoder.input_dim), "mel_linear.bias": (c.n_mels,)})
        if c.use_postnet:
            dims = [c.n_mels] + [c.postnet_channels] * (c.postnet_layers - 1) + [c.n_mels]
            for i in range(c.postnet_layers):
                p = f"postnet.convolutions.{i}"
                shapes.update({p + ".0.weight": (dims[i + 1], dims[i], c.postnet_kernel), p + ".0.bias": (dims[i + 1],),
                               p + ".1.weight": (dims[i + 1],), p + ".1.bias": (dims[i + 1],),
                               p + ".1.running_mean": (dims[i + 1],), p + ".1.running_var": (dims[i + 1],)})
        return shapes

    def init_random(self, seed: int = 1234):
        """Random parameters of the right shapes (benchmarks; there is no network for checkpoints)."""
        g = torch.Generator().manual_seed(seed)
        sd = {}
        for name, shape in self.state_dict_shapes(self.config).items():
            if name.endswith("inv_freq"):
                d = self.config.encoder.input_dim
                sd[name] = 1.0 / (10000 ** (torch.arange(0.0, d, 2.0) / d))
            elif name.endswith("running_var") or name.endswith("weight_g") or (len(shape) == 1 and name.endswith(".weight")):
                sd[name] = torch.ones(shape)
            elif name.endswith("bias") or name.endswith("running_mean"):
                sd[name] = torch.zeros(shape)
            else:
                fan_in = max(1, int(torch.tensor(shape[1:]).prod())) if len(shape) > 1 else shape[0]
                sd[name] = torch.randn(shape, generator=g) / fan_in ** 0.5
        return self.load_state_dict(sd)

    # -- checkpoints (conventions of the reference: everyvoice/tests/test_model.py:85-151, 302-313, 454-459) -----------
    _VERSION = "1.0"

    def hyper_parameters(self) -> dict:
        from dataclasses import asdict
        return {"config": asdict(self.config), "stats": asdict(self.stats), "lang2id": dict(self.lang2id), "speaker2id": dict(self.speaker2id)}

    def to_checkpoint(self, state_dict: dict) -> dict:
        """Lightning-shaped dict: state_dict, JSON-only hyper_parameters (config, stats, lang2id, speaker2id), model_info."""
        return {"state_dict": {k: v.detach().cpu() for k, v in state_dict.items()}, "hyper_parameters": self.hyper_parameters(),
                "model_info": {"name": "FastSpeech2", "version": self._VERSION}}

    @classmethod
    def from_checkpoint(cls, ckpt: dict, device="cuda:0"):
        info = ckpt.get("model_info") if isinstance(ckpt, dict) else None
        if isinstance(info, dict) and info.get("name") != "FastSpeech2":
            raise TypeError(f"Wrong model type ({info.get('name')}), we are expecting a 'FastSpeech2' model")
        if isinstance(info, dict) and int(str(info.get("version", "1.0")).split(".")[0]) > int(cls._VERSION.split(".")[0]):
            raise ValueError("Your model was created with a newer version of EveryVoice, please update your software.")
        try:
            hp = ckpt["hyper_parameters"]
            c = hp["config"]
            cfg = FastSpeech2ModelConfig(
                encoder=ConformerConfig(**c["encoder"]), decoder=ConformerConfig(**c["decoder"]),
                variance_predictors=VariancePredictors(**{k: VariancePredictorConfig(**v) for k, v in c["variance_predictors"].items()}),
                **{k: v for k, v in c.items() if k not in ("encoder", "decoder", "variance_predictors")})
            stats = Stats(pitch=StatsInfo(**hp["stats"]["pitch"]), energy=StatsInfo(**hp["stats"]["energy"]))
        except (KeyError, TypeError) as e:
            raise TypeError("Unable to load config.  Possible causes: is it really a FastSpeech2Config? or the correct version?") from e
        model = cls(cfg, stats, device=device, lang2id=hp.get("lang2id"), speaker2id=hp.get("speaker2id"))
        return model.load_state_dict(ckpt["state_dict"])

    # -- forward --------------------------------------------------------------------------------------------------
    @torch.no_grad()
    def __call__(self, ids: torch.Tensor, lens: torch.Tensor, duration_control=1.0, pitch_control=1.0, energy_control=1.0,
                 durations: torch.Tensor | None = None, speakers: torch.Tensor | None = None, languages: torch.Tensor | None = None):
        """ids [B, L] (0 = padding), lens [B] -> (mel [B, T, n_mels], postnet mel, durations [B, L], pitch [B, L],
        energy [B, L], mel_lens [B]) on the device."""
        if not self._ready:
            raise RuntimeError("load_state_dict() or init_random() first")
        prev = ops.CONV_BACKEND["operands"]
        ops.CONV_BACKEND["operands"] = self.precision
        try:
            return self._forward(ids, lens, duration_control, pitch_control, energy_control, durations, speakers, languages)
        finally:
            ops.CONV_BACKEND["operands"] = prev

    def _forward(self, ids, lens, duration_control, pitch_control, energy_control, durations, speakers, languages):
        lib, dev, c = _lib.load(), self.device, self.config
        B, L = ids.shape[:2]
        if L > c.max_length:
            raise ValueError(f"text of {L} symbols exceeds max_length {c.max_length}")
        D = c.encoder.input_dim
        lens32 = lens.to(dev, torch.int32).contiguous()
        if c.target_text_representation_level == "phonological_features":
            # ids [B, L, 43] feature vectors: Linear(43 -> D) as a pointwise convolution over [43, B, L], then the positional term
            # (which also zeroes the padded columns, as the embedding kernel does)
            if ids.dim() != 3 or ids.shape[2] != N_PHONOLOGICAL_FEATURES:
                raise ValueError(f"phonological features: expected [B, L, {N_PHONOLOGICAL_FEATURES}], got {tuple(ids.shape)}")
            feats = ids.to(dev, torch.float32).permute(2, 0, 1).contiguous()
            x = _conv(feats, self.table.view(D, N_PHONOLOGICAL_FEATURES, 1), None)
            _chk(lib.evmi_fs2_add_posemb_f32(x.data_ptr(), lens32.data_ptr(), self.inv_freq.data_ptr(), B, L, D, _s(x)), "evmi_fs2_add_posemb_f32")
        else:
            ids32 = ids.to(dev, torch.int32).contiguous()
            x = torch.empty(D, B, L, device=dev, dtype=torch.float32)
            _chk(lib.evmi_fs2_embed_f32(ids32.data_ptr(), lens32.data_ptr(), self.table.data_ptr(), self.inv_freq.data_ptr(), x.data_ptr(),
                                        B, L, D, _s(x)), "evmi_fs2_embed_f32")
        x = self.encoder.forward(x, lens32)
        for table, item_ids, what in ((self.speaker_table, speakers, "speakers"), (self.language_table, languages, "languages")):
            if table is not None:
                if item_ids is None:
                    raise ValueError(f"this model needs `{what}` ids [B]")
                item32 = item_ids.to(dev, torch.int32).contiguous()
                _chk(lib.evmi_fs2_add_item_embedding_f32(x.data_ptr(), item32.data_ptr(), lens32.data_ptr(), table.data_ptr(), B, L, D, _s(x)),
                     "evmi_fs2_add_item_embedding_f32")
        log_d = self.duration_predictor.forward(x, lens32)
        pitch = self.pitch_predictor.forward(x, lens32)
        vp = c.variance_predictors
        _chk(lib.evmi_fs2_bucket_embed_add_f32(x.data_ptr(), pitch.data_ptr(), self.pitch_bins.data_ptr(), self.pitch_table.data_ptr(),
                                               vp.pitch.n_bins, B, L, D, float(pitch_control), _s(x)), "evmi_fs2_bucket_embed_add_f32")
        energy = self.energy_predictor.forward(x, lens32)
        _chk(lib.evmi_fs2_bucket_embed_add_f32(x.data_ptr(), energy.data_ptr(), self.energy_bins.data_ptr(), self.energy_table.data_ptr(),
                                               vp.energy.n_bins, B, L, D, float(energy_control), _s(x)), "evmi_fs2_bucket_embed_add_f32")
        if durations is None:
            dur = torch.empty(B, L, device=dev, dtype=torch.int32)
            _chk(lib.evmi_fs2_durations_i32(log_d.data_ptr(), lens32.data_ptr(), dur.data_ptr(), B, L, float(duration_control), _s(x)),
                 "evmi_fs2_durations_i32")
        else:
            pad = torch.arange(L, device=dev)[None, :] >= lens32[:, None]
            dur = durations.to(dev, torch.int32).clamp_min(0).masked_fill(pad, 0).contiguous()
        cum = torch.cumsum(dur, 1, dtype=torch.int32).contiguous()
        mel_lens = cum[:, -1].contiguous()
        T = int(mel_lens.max())  # the one host sync of the forward: the output length
        if T <= 0:
            raise ValueError("all predicted durations are zero")
        frames = torch.empty(D, B, T, device=dev, dtype=torch.float32)
        _chk(lib.evmi_length_regulate_cbt_f32(x.data_ptr(), cum.data_ptr(), frames.data_ptr(), D, B, L, T, _s(x)), "evmi_length_regulate_cbt_f32")
        _chk(lib.evmi_fs2_add_posemb_f32(frames.data_ptr(), mel_lens.data_ptr(), self.inv_freq.data_ptr(), B, T, D, _s(x)), "evmi_fs2_add_posemb_f32")
        y = self.decoder.forward(frames, mel_lens)
        mel = _conv(y, self.w_mel, self.b_mel)
        _chk(lib.evmi_mask_cols_f32(mel.data_ptr(), mel_lens.data_ptr(), c.n_mels, B, T, _s(x)), "evmi_mask_cols_f32")
        post = mel
        if self.postnet:
            h = mel
            for i, (w, b) in enumerate(self.postnet):
                h = _conv(h, w, b, c.postnet_kernel, act=ops.ACT_TANH if i < len(self.postnet) - 1 else ops.ACT_NONE)
            post = ops.axpby(1.0, mel, 1.0, h)
            _chk(lib.evmi_mask_cols_f32(post.data_ptr(), mel_lens.data_ptr(), c.n_mels, B, T, _s(x)), "evmi_mask_cols_f32")
        to_btc = lambda t: t.permute(1, 2, 0).contiguous()
        mult = lambda v, s: v if s == 1.0 else v * s
        return to_btc(mel), to_btc(post), dur.to(torch.int64), mult(pitch, pitch_control), mult(energy, energy_control), mel_lens.to(torch.int64)
