"""Host-side mirror of the reference's vocoder inference interface, computing on libevmi_hip.

Mirrors (absent submodule ``hfgl``; call sites in the reference):
  * ``HiFiGANGenerator(config)`` with ``.generator`` / ``.config`` — everyvoice/demo/app.py:28-33,
    everyvoice/base_cli/checkpoint.py:92-103; exported-generator checkpoints
    (everyvoice/cli.py:372-390, everyvoice/tests/test_cli.py:342-363: 13,254,034 parameters with
    weight norm folded for the test config);
  * ``load_hifigan_from_checkpoint(ckpt, device) -> (model, config)`` — everyvoice/demo/app.py:457-463
    (raises TypeError for a checkpoint that is not a HiFiGAN one).

The modules hold parameters only (upstream state-dict names, so reference checkpoints load
unchanged, ``weight_g``/``weight_v`` pairs are folded on load); ``forward`` hands device pointers
to ``evmi_generator_forward``.  There is no eager-PyTorch compute path here.
"""

from __future__ import annotations

import ctypes as C
from collections import OrderedDict

import torch
from torch import nn

from . import _lib
from .config import ACTIVATION_SLOPES, HiFiGANConfig

# "bf16": bf16 operands / activations on the bf16 matrix cores (the headline path, native generator object)
# "f32":  exact fp32 arithmetic on the fp32-input matrix cores (v_mfma_f32_32x32x2_f32 = fmaf chains), channel-major kernels
# "f32-direct": exact fp32 on the vector ALUs (the native object's direct fmaf kernels; the slowest and simplest path)
PRECISIONS = {"bf16": _lib.EVMI_PREC_BF16, "f32": _lib.EVMI_PREC_F32, "f32-direct": _lib.EVMI_PREC_F32}


class _ConvParams(nn.Module):
    """weight + bias of one convolution (no forward: the arithmetic lives in the HIP library)."""

    def __init__(self, *weight_shape: int, bias: int):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(*weight_shape))
        self.bias = nn.Parameter(torch.zeros(bias))


class _ResBlockParams(nn.Module):
    def __init__(self, kind: str, channels: int, kernel: int, n_dil: int):
        super().__init__()
        mk = lambda: nn.ModuleList(_ConvParams(channels, channels, kernel, bias=channels) for _ in range(n_dil))  # noqa: E731
        if kind == "1":
            self.convs1 = mk()
            self.convs2 = mk()
        else:
            self.convs = mk()


def _model_cfg_to_c(config: HiFiGANConfig) -> _lib.GeneratorConfig:
    m = config.model
    c = _lib.GeneratorConfig()
    c.n_mels = config.preprocessing.audio.n_mels
    c.upsample_initial_channel = m.upsample_initial_channel
    c.num_upsamples = len(m.upsample_rates)
    if len(m.upsample_kernel_sizes) != c.num_upsamples:
        raise ValueError("upsample_rates and upsample_kernel_sizes differ in length")
    if c.num_upsamples > _lib.EVMI_MAX_UPSAMPLES or len(m.resblock_kernel_sizes) > _lib.EVMI_MAX_RESBLOCK_KERNELS:
        raise ValueError("too many upsampling stages / resblock kernels")
    for i, (u, k) in enumerate(zip(m.upsample_rates, m.upsample_kernel_sizes)):
        c.upsample_rates[i] = u
        c.upsample_kernel_sizes[i] = k
    c.resblock_type = int(str(getattr(m.resblock, "value", m.resblock)))
    c.num_kernels = len(m.resblock_kernel_sizes)
    for j, (k, dils) in enumerate(zip(m.resblock_kernel_sizes, m.resblock_dilation_sizes)):
        c.resblock_kernel_sizes[j] = k
        c.num_dilations[j] = len(dils)
        for n, d in enumerate(dils):
            c.resblock_dilations[j][n] = d
    if m.activation_function not in ACTIVATION_SLOPES:
        raise ValueError(
            f"activation_function {m.activation_function!r} has no HIP epilogue; supported: {sorted(ACTIVATION_SLOPES)}"
        )
    c.lrelu_slope = ACTIVATION_SLOPES[m.activation_function]
    c.post_lrelu_slope = 0.01  # F.leaky_relu default before conv_post, as upstream
    c.istft_layer = int(m.istft_layer)
    c.istft_n_fft = config.gen_istft_n_fft
    c.istft_hop = config.gen_istft_hop_size
    return c


class Generator(nn.Module):
    """HiFiGAN / iSTFTNet generator: ``forward(mel[B, n_mels, T]) -> wav[B, 1, T*hop]`` on the GPU."""

    def __init__(self, config: HiFiGANConfig, precision: str = "bf16"):
        super().__init__()
        self.config = config
        self.precision = precision
        m = config.model
        ch0 = m.upsample_initial_channel
        n_mels = config.preprocessing.audio.n_mels
        kind = str(getattr(m.resblock, "value", m.resblock))
        self.conv_pre = _ConvParams(ch0, n_mels, 7, bias=ch0)
        self.ups = nn.ModuleList(
            _ConvParams(ch0 >> i, ch0 >> (i + 1), k, bias=ch0 >> (i + 1))
            for i, k in enumerate(m.upsample_kernel_sizes)
        )
        self.resblocks = nn.ModuleList()
        for i in range(len(m.upsample_rates)):
            for k, d in zip(m.resblock_kernel_sizes, m.resblock_dilation_sizes):
                self.resblocks.append(_ResBlockParams(kind, ch0 >> (i + 1), k, len(d)))
        ch_last = ch0 >> len(m.upsample_rates)
        post_out = config.gen_istft_n_fft + 2 if m.istft_layer else 1
        self.conv_post = _ConvParams(post_out, ch_last, 7, bias=post_out)
        self._c_cfg = _model_cfg_to_c(config)
        object.__setattr__(self, "_handle", None)
        self._uploaded_version = None
        self.reset_parameters()

    # -- parameters -----------------------------------------------------------------------------
    def reset_parameters(self, std: float = 0.01) -> None:
        """Upstream init: N(0, 0.01) for ups / resblocks / conv_post, torch Conv1d default for conv_pre."""
        with torch.no_grad():
            for name, p in self.named_parameters():
                if name.endswith("bias"):
                    fan_in = dict(self.named_parameters())[name[:-4] + "weight"][0].numel()
                    bound = 1.0 / (fan_in**0.5)
                    p.uniform_(-bound, bound)
                elif name.startswith("conv_pre"):
                    nn.init.kaiming_uniform_(p, a=5**0.5)
                else:
                    p.normal_(0.0, std)

    def remove_weight_norm(self) -> "Generator":
        """Weights are held folded already; kept for interface parity with the reference."""
        return self

    def _load_from_state_dict(self, state_dict, prefix, *args, **kwargs):
        fold_weight_norm_(state_dict, prefix)
        return super()._load_from_state_dict(state_dict, prefix, *args, **kwargs)

    def load_state_dict(self, state_dict, strict: bool = True, assign: bool = False):
        sd = OrderedDict(state_dict)
        fold_weight_norm_(sd, "")
        out = super().load_state_dict(sd, strict=strict, assign=assign)
        self._uploaded_version = None
        return out

    def _weights_version(self):
        return tuple((p.data_ptr(), p._version) for p in self.parameters())

    # -- native object ----------------------------------------------------------------------------
    def _new_handle(self, device_index: int):
        lib = _lib.load()
        old = self.__dict__.get("_handle")
        if old is not None:
            lib.evmi_generator_destroy(old)
        h = C.c_void_p()
        _lib.check(lib.evmi_generator_create(C.byref(self._c_cfg), device_index, C.byref(h)), "evmi_generator_create")
        object.__setattr__(self, "_handle", h)
        object.__setattr__(self, "_handle_device", device_index)
        self._uploaded_version = None

    def _ensure_native(self, device: torch.device):
        """The native object (weights in kernel layouts + workspace) lives on ONE device: it is created for the device of the
        first input and re-created -- weights uploaded again -- when the module is used on another one."""
        lib = _lib.load()
        index = device.index if device.index is not None else torch.cuda.current_device()
        if self._handle is None or self.__dict__.get("_handle_device") != index:
            self._new_handle(index)
        ver = self._weights_version()
        if self._uploaded_version != ver:
            for name, p in self.named_parameters():
                host = p.detach().to("cpu", torch.float32).contiguous()
                _lib.check(
                    lib.evmi_generator_set_weight(self._handle, name.encode(), host.data_ptr(), host.numel()),
                    f"evmi_generator_set_weight({name})",
                )
            with torch.cuda.device(index):  # finalize uploads to the handle's device: the caller's current device is left alone
                _lib.check(lib.evmi_generator_finalize(self._handle), "evmi_generator_finalize")
            self._uploaded_version = ver
        return lib

    def __del__(self):
        try:
            h = self.__dict__.pop("_handle", None)
            if h is not None and _lib._lib is not None:
                _lib._lib.evmi_generator_destroy(h)
        except Exception:  # interpreter shutdown: module globals may already be gone
            pass

    @property
    def hop(self) -> int:
        h = 1
        for u in self.config.model.upsample_rates:
            h *= u
        return h * self.config.gen_istft_hop_size if self.config.model.istft_layer else h

    def macs_per_sample(self) -> float:
        """Multiply-accumulates per output sample of this configuration (host arithmetic: a throw-away native object when the
        module has not run yet, so asking does not pin the module to a device)."""
        lib = _lib.load()
        if self._handle is not None:
            return float(lib.evmi_generator_macs_per_sample(self._handle))
        h = C.c_void_p()
        _lib.check(lib.evmi_generator_create(C.byref(self._c_cfg), 0, C.byref(h)), "evmi_generator_create")
        try:
            return float(lib.evmi_generator_macs_per_sample(h))
        finally:
            lib.evmi_generator_destroy(h)

    def _check_input(self, mel: torch.Tensor) -> torch.Tensor:
        if not mel.is_cuda:
            raise RuntimeError("everyvoice_amd.Generator computes on the GPU only (no CPU fallback): move the input to cuda")
        if mel.dim() == 2:
            mel = mel.unsqueeze(0)
        n_mels = self.config.preprocessing.audio.n_mels
        if mel.dim() != 3 or mel.shape[1] != n_mels:
            raise ValueError(f"expected mel of shape [B, {n_mels}, T], got {tuple(mel.shape)}")
        return mel.to(torch.float32).contiguous()

    @torch.no_grad()
    def _forward_f32_mfma(self, mel: torch.Tensor) -> torch.Tensor:
        """The exact-fp32 forward on the fp32 matrix cores: the channel-major [C][B][T] convolution kernels of the training path
        (csrc/conv_cbt_f32_mfma.hip: implicit GEMM, products and sums are exact fp32 fmaf chains) driven layer by layer with the
        folded weights.  ~15 x the direct vector-ALU kernels' rate at the benchmark shape."""
        from .train import autograd as ag
        from .train import ops

        m = self.config.model
        slope = ACTIVATION_SLOPES[m.activation_function]
        B, n_mels, T = mel.shape
        prev = ops.CONV_BACKEND["operands"]
        ops.CONV_BACKEND["operands"] = "f32"
        try:
            x = torch.empty(n_mels, B, T, device=mel.device, dtype=torch.float32)
            _lib.check(_lib.load().evmi_transpose_bct_cbt_f32(mel.data_ptr(), x.data_ptr(), B, n_mels, T, _lib.current_stream_ptr(mel.device)),
                       "evmi_transpose_bct_cbt_f32")
            x = ops.conv1d_fwd(x, self.conv_pre.weight, self.conv_pre.bias, 1, 3, 1, 1)
            nk = len(m.resblock_kernel_sizes)
            for i, (u, ku) in enumerate(zip(m.upsample_rates, m.upsample_kernel_sizes)):
                x = ops.conv_transpose1d_fwd(ops.lrelu(x, slope), self.ups[i].weight, self.ups[i].bias, u, (ku - u) // 2)
                xs = None
                for j, (k, dils) in enumerate(zip(m.resblock_kernel_sizes, m.resblock_dilation_sizes)):
                    rb = self.resblocks[i * nk + j]
                    y = x
                    for q, d in enumerate(dils):
                        if hasattr(rb, "convs1"):
                            t = ops.conv1d_fwd(ops.lrelu(y, slope), rb.convs1[q].weight, rb.convs1[q].bias, 1, d * (k - 1) // 2, d, 1, lrelu_slope=slope)
                            t = ops.conv1d_fwd(t, rb.convs2[q].weight, rb.convs2[q].bias, 1, (k - 1) // 2, 1, 1)
                        else:
                            t = ops.conv1d_fwd(ops.lrelu(y, slope), rb.convs[q].weight, rb.convs[q].bias, 1, d * (k - 1) // 2, d, 1)
                        y = ops.axpby(1.0, t, 1.0, y, out=t)
                    xs = y if xs is None else ops.axpby(1.0, xs, 1.0, y, out=xs)
                x = ops.elementwise(ops.EW_SCALE, xs, out=xs, p0=1.0 / nk)
            x = ops.lrelu(x, 0.01)
            if m.istft_layer:
                n_fft, hop = self.config.gen_istft_n_fft, self.config.gen_istft_hop_size
                consts = self.__dict__.get("_istft_consts")
                if consts is None or consts.device != mel.device:
                    consts = ag.ISTFTConstants(n_fft, hop, mel.device)
                    object.__setattr__(self, "_istft_consts", consts)
                x = ops.conv1d_fwd(ops.reflect_pad_left1(x), self.conv_post.weight, self.conv_post.bias, 1, 3, 1, 1)
                s = ops.istft_polar(x, n_fft // 2 + 1)
                raw = ops.conv_transpose1d_fwd(s, consts.weight, None, hop, n_fft // 2)
                x = ops.elementwise(ops.EW_MUL, raw, consts.inv_envelope(B, x.shape[2]), out=raw)
            else:
                x = ops.tanh(ops.conv1d_fwd(x, self.conv_post.weight, self.conv_post.bias, 1, 3, 1, 1))
        finally:
            ops.CONV_BACKEND["operands"] = prev
        return x.view(B, 1, -1)  # [1, B, T'] and [B, 1, T'] are the same bytes

    @torch.no_grad()
    def forward(self, mel: torch.Tensor) -> torch.Tensor:
        mel = self._check_input(mel)
        if self.precision == "f32":
            return self._forward_f32_mfma(mel)
        lib = self._ensure_native(mel.device)
        B, _, T = mel.shape
        wav = torch.empty(B, 1, T * self.hop, device=mel.device, dtype=torch.float32)
        with torch.cuda.device(mel.device):
            _lib.check(
                lib.evmi_generator_forward(self._handle, mel.data_ptr(), wav.data_ptr(), B, T, PRECISIONS[self.precision],
                                           _lib.current_stream_ptr(mel.device)),
                "evmi_generator_forward",
            )
        return wav

    @torch.no_grad()
    def forward_profiled(self, mel: torch.Tensor):
        """One forward with every launch bracketed by HIP events on the current stream.
        Returns (wav, [dict(kernel, layer, ms, flops, bytes)])."""
        mel = self._check_input(mel)
        lib = self._ensure_native(mel.device)
        B, _, T = mel.shape
        wav = torch.empty(B, 1, T * self.hop, device=mel.device, dtype=torch.float32)
        cap = 512
        recs = (_lib.LaunchRecord * cap)()
        n = C.c_int(0)
        with torch.cuda.device(mel.device):
            _lib.check(
                lib.evmi_generator_forward_profiled(self._handle, mel.data_ptr(), wav.data_ptr(), B, T,
                                                    PRECISIONS[self.precision], _lib.current_stream_ptr(mel.device),
                                                    recs, cap, C.byref(n)),
                "evmi_generator_forward_profiled",
            )
        out = [
            dict(kernel=r.kernel.decode(), layer=r.layer.decode(), ms=float(r.ms), flops=float(r.flops), bytes=float(r.bytes))
            for r in recs[: min(n.value, cap)]
        ]
        return wav, out


def fold_weight_norm_(state_dict: dict, prefix: str = "") -> None:
    """Replace every ``<name>.weight_g`` / ``<name>.weight_v`` pair under ``prefix`` by the folded
    ``<name>.weight = g * v / ||v||`` (norm over all dims but 0, torch.nn.utils.weight_norm's dim=0)."""
    for key in [k for k in state_dict if k.startswith(prefix) and k.endswith("weight_g")]:
        base = key[: -len("weight_g")]
        g = state_dict.pop(key).to(torch.float32)
        v = state_dict.pop(base + "weight_v").to(torch.float32)
        dims = tuple(range(1, v.dim()))
        norm = v.pow(2).sum(dim=dims, keepdim=True).sqrt()
        state_dict[base + "weight"] = v * (g / norm)


class HiFiGANGenerator(nn.Module):
    """Generator-only vocoder module (what ``everyvoice export spec-to-wav`` writes and
    ``load_hifigan_from_checkpoint`` returns for an exported checkpoint)."""

    _VERSION = "1.0"

    def __init__(self, config: dict | HiFiGANConfig, precision: str = "bf16"):
        super().__init__()
        if isinstance(config, dict):
            config = HiFiGANConfig(**config)
        self.config = config
        self.generator = Generator(config, precision=precision)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return self.generator(x)

    def on_save_checkpoint(self, checkpoint: dict) -> None:
        """Checkpoint conventions of the reference (everyvoice/tests/test_model.py:85-151,302-313):
        JSON-only config under hyper_parameters, model_info name/version."""
        checkpoint["hyper_parameters"] = {"config": self.config.model_dump(mode="json")}
        checkpoint["model_info"] = {"name": type(self).__name__, "version": self._VERSION}

    def to_checkpoint(self) -> dict:
        ckpt = {"state_dict": OrderedDict((k, v.detach().cpu()) for k, v in self.state_dict().items())}
        self.on_save_checkpoint(ckpt)
        return ckpt


_VOCODER_NAMES = ("HiFiGAN", "HiFiGANGenerator")


def load_hifigan_from_checkpoint(ckpt: dict, device, precision: str = "bf16"):
    """(model, config) from a reference-format checkpoint dict: a full ``HiFiGAN`` training checkpoint
    (generator + discriminators; only ``generator.*`` is used) or an exported ``HiFiGANGenerator`` one."""
    info = ckpt.get("model_info") if isinstance(ckpt, dict) else None
    if isinstance(info, dict) and info.get("name") not in _VOCODER_NAMES:
        raise TypeError(
            f"Wrong model type ({info.get('name')}), we are expecting a 'HiFiGAN' or 'HiFiGANGenerator' model"
        )
    try:
        config = ckpt["hyper_parameters"]["config"]
        state = ckpt["state_dict"]
    except (KeyError, TypeError) as e:
        raise TypeError("Unable to load config.  Possible causes: is it really a VocoderConfig? or the correct version?") from e
    if isinstance(config, dict):
        try:
            config = HiFiGANConfig(**config)
        except Exception as e:  # pydantic.ValidationError
            raise TypeError(
                "Unable to load config.  Possible causes: is it really a VocoderConfig? or the correct version?"
            ) from e
    gen_state = OrderedDict((k, v) for k, v in state.items() if k.startswith("generator."))
    if not gen_state:
        raise TypeError("checkpoint has no 'generator.*' tensors: maybe it's not actually a HiFiGAN model")
    model = HiFiGANGenerator(config, precision=precision)
    model.load_state_dict(gen_state)
    model = model.to(device)
    model.eval()
    return model, config
