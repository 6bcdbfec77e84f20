"""Plain-PyTorch fp32 CPU restatement of the HiFiGAN / iSTFTNet vocoder.  TEST INFRASTRUCTURE.

PARITY UNPINNED (numerically): the reference keeps this model in the un-vendored
submodule ``everyvoice/model/vocoder/HiFiGAN_iSTFT_lightning`` (package ``hfgl``,
``/root/reference/.gitmodules:4-6``, directory empty), pinned only to "0.5.x"
(``everyvoice/tests/test_cli.py:75-91``).  What IS pinned, and asserted in
``tests/test_oracle_hifigan.py``:

* every hyper-parameter default: ``everyvoice/.schema/everyvoice-spec-to-wav-0.5.json:293-415``
  (``HiFiGANModelConfig``: upsample_rates [8,8,2,2], kernels [16,16,4,4], initial channel 512,
  resblock "1", kernels [3,7,11], dilations [[1,3,5]]*3, mpd_layers [2,3,5,7,11], msd_layers 3);
* the activation: ``everyvoice/utils/__init__.py:178-181`` (leaky_relu slope 0.1);
* exact parameter counts: generator+MPD+MSD of the test config (iSTFT, ups [8,8]) with
  weight-norm g/v = 83,986,835 (``everyvoice/tests/test_cli.py:340``); exported generator of
  the same config with weight norm folded = 13,254,034 (``everyvoice/tests/test_cli.py:363``).

The layer structure follows the public code bases the reference credits in
``README.md:94-104`` (jik876/hifi-gan for G/MPD/MSD, rishikksh20/iSTFTNet-pytorch for the
iSTFT head), restated here from their published architecture — state-dict keys keep the
upstream names (``conv_pre``, ``ups.N``, ``resblocks.N.convs1.M``, ``conv_post``,
``weight_g``/``weight_v``) so that a real EveryVoice checkpoint loads unchanged.
"""

from __future__ import annotations

import math
import warnings
from dataclasses import dataclass, field

import torch
import torch.nn.functional as F
from torch import nn

LRELU_SLOPE = 0.1  # everyvoice/utils/__init__.py:178-181


@dataclass
class HiFiGANModelConfigRef:
    """Defaults: everyvoice/.schema/everyvoice-spec-to-wav-0.5.json:293-415."""

    resblock: str = "1"
    upsample_rates: list = field(default_factory=lambda: [8, 8, 2, 2])
    upsample_kernel_sizes: list = field(default_factory=lambda: [16, 16, 4, 4])
    upsample_initial_channel: int = 512
    resblock_kernel_sizes: list = field(default_factory=lambda: [3, 7, 11])
    resblock_dilation_sizes: list = field(default_factory=lambda: [[1, 3, 5]] * 3)
    istft_layer: bool = False
    msd_layers: int = 3
    mpd_layers: list = field(default_factory=lambda: [2, 3, 5, 7, 11])
    # not in the schema: fixed by the preprocessing config (AudioConfig.n_mels) and the
    # iSTFTNet head size seen in tests/data/relative/config/everyvoice-text-to-wav.yaml:6-8
    n_mels: int = 80
    gen_istft_n_fft: int = 16
    gen_istft_hop_size: int = 4

    @classmethod
    def test_config(cls) -> "HiFiGANModelConfigRef":
        """everyvoice/tests/data/relative/config/everyvoice-spec-to-wav.yaml:5-19."""
        return cls(istft_layer=True, upsample_rates=[8, 8], upsample_kernel_sizes=[16, 16])


def _wn(module: nn.Module) -> nn.Module:
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        return nn.utils.weight_norm(module)


def _sn(module: nn.Module) -> nn.Module:
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        return nn.utils.spectral_norm(module)


def _same_padding(kernel: int, dilation: int = 1) -> int:
    return (kernel * dilation - dilation) // 2


def _normal_init(m: nn.Module, std: float = 0.01) -> None:
    if isinstance(m, (nn.Conv1d, nn.ConvTranspose1d)):
        m.weight.data.normal_(0.0, std)


class ResBlock1Ref(nn.Module):
    """3 x [lrelu -> dilated conv -> lrelu -> conv] with a residual after each pair."""

    def __init__(self, channels: int, kernel: int, dilations=(1, 3, 5)):
        super().__init__()
        self.convs1 = nn.ModuleList(
            _wn(nn.Conv1d(channels, channels, kernel, 1, dilation=d, padding=_same_padding(kernel, d)))
            for d in dilations
        )
        self.convs2 = nn.ModuleList(
            _wn(nn.Conv1d(channels, channels, kernel, 1, dilation=1, padding=_same_padding(kernel, 1)))
            for _ in dilations
        )
        self.convs1.apply(_normal_init)
        self.convs2.apply(_normal_init)

    def forward(self, x):
        for c1, c2 in zip(self.convs1, self.convs2):
            xt = F.leaky_relu(x, LRELU_SLOPE)
            xt = c1(xt)
            xt = F.leaky_relu(xt, LRELU_SLOPE)
            xt = c2(xt)
            x = xt + x
        return x

    def remove_weight_norm(self):
        for c in list(self.convs1) + list(self.convs2):
            nn.utils.remove_weight_norm(c)


class ResBlock2Ref(nn.Module):
    """2 x [lrelu -> dilated conv] with a residual each (the "2" resblock option)."""

    def __init__(self, channels: int, kernel: int, dilations=(1, 3)):
        super().__init__()
        self.convs = nn.ModuleList(
            _wn(nn.Conv1d(channels, channels, kernel, 1, dilation=d, padding=_same_padding(kernel, d)))
            for d in dilations
        )
        self.convs.apply(_normal_init)

    def forward(self, x):
        for c in self.convs:
            xt = F.leaky_relu(x, LRELU_SLOPE)
            xt = c(xt)
            x = xt + x
        return x

    def remove_weight_norm(self):
        for c in self.convs:
            nn.utils.remove_weight_norm(c)


class GeneratorRef(nn.Module):
    """HiFiGAN generator (V1 by default) with the optional iSTFTNet head.

    forward(mel[B, n_mels, T]) -> wav[B, 1, T * hop]  (hop = prod(upsample_rates) [* istft hop]).
    """

    def __init__(self, cfg: HiFiGANModelConfigRef | None = None):
        super().__init__()
        cfg = cfg or HiFiGANModelConfigRef()
        self.cfg = cfg
        self.num_kernels = len(cfg.resblock_kernel_sizes)
        self.num_upsamples = len(cfg.upsample_rates)
        ch0 = cfg.upsample_initial_channel
        self.conv_pre = _wn(nn.Conv1d(cfg.n_mels, ch0, 7, 1, padding=3))
        block = ResBlock1Ref if str(cfg.resblock) == "1" else ResBlock2Ref
        self.ups = nn.ModuleList()
        for i, (u, k) in enumerate(zip(cfg.upsample_rates, cfg.upsample_kernel_sizes)):
            self.ups.append(
                _wn(nn.ConvTranspose1d(ch0 // (2**i), ch0 // (2 ** (i + 1)), k, u, padding=(k - u) // 2))
            )
        self.resblocks = nn.ModuleList()
        ch = ch0
        for i in range(self.num_upsamples):
            ch = ch0 // (2 ** (i + 1))
            for k, d in zip(cfg.resblock_kernel_sizes, cfg.resblock_dilation_sizes):
                self.resblocks.append(block(ch, k, tuple(d)))
        self.out_channels = ch
        if cfg.istft_layer:
            self.post_n_fft = cfg.gen_istft_n_fft
            self.conv_post = _wn(nn.Conv1d(ch, self.post_n_fft + 2, 7, 1, padding=3))
            self.reflection_pad = nn.ReflectionPad1d((1, 0))
        else:
            self.conv_post = _wn(nn.Conv1d(ch, 1, 7, 1, padding=3))
        self.ups.apply(_normal_init)
        self.conv_post.apply(_normal_init)

    @property
    def hop(self) -> int:
        h = math.prod(self.cfg.upsample_rates)
        return h * self.cfg.gen_istft_hop_size if self.cfg.istft_layer else h

    def trunk(self, x):
        x = self.conv_pre(x)
        for i in range(self.num_upsamples):
            x = F.leaky_relu(x, LRELU_SLOPE)
            x = self.ups[i](x)
            xs = None
            for j in range(self.num_kernels):
                y = self.resblocks[i * self.num_kernels + j](x)
                xs = y if xs is None else xs + y
            x = xs / self.num_kernels
        return F.leaky_relu(x)  # NB default slope 0.01 before conv_post, as upstream

    def forward(self, mel):
        x = self.trunk(mel)
        if not self.cfg.istft_layer:
            return torch.tanh(self.conv_post(x))
        x = self.conv_post(self.reflection_pad(x))
        half = self.post_n_fft // 2 + 1
        mag = torch.exp(x[:, :half, :])
        phase = torch.sin(x[:, half:, :])
        return istft_ref(mag, phase, self.post_n_fft, self.cfg.gen_istft_hop_size)

    def remove_weight_norm(self):
        for up in self.ups:
            nn.utils.remove_weight_norm(up)
        for rb in self.resblocks:
            rb.remove_weight_norm()
        nn.utils.remove_weight_norm(self.conv_pre)
        nn.utils.remove_weight_norm(self.conv_post)
        return self


def istft_ref(mag: torch.Tensor, phase: torch.Tensor, n_fft: int, hop: int) -> torch.Tensor:
    """Inverse STFT of mag*exp(i*phase) (hann window n_fft, centred), as
    torchaudio.transforms.InverseSpectrogram(n_fft, win_length=n_fft, hop_length=hop) computes it
    (everyvoice/utils/heavy.py:115-118).  Returns [B, 1, hop*(frames-1)]."""
    spec = torch.complex(mag * torch.cos(phase), mag * torch.sin(phase))
    window = torch.hann_window(n_fft, dtype=mag.dtype, device=mag.device)
    wav = torch.istft(spec, n_fft, hop_length=hop, win_length=n_fft, window=window, center=True)
    return wav.unsqueeze(1)


class DiscriminatorPRef(nn.Module):
    def __init__(self, period: int, kernel: int = 5, stride: int = 3):
        super().__init__()
        self.period = period
        chans = [1, 32, 128, 512, 1024]
        pad = (_same_padding(kernel, 1), 0)
        self.convs = nn.ModuleList(
            _wn(nn.Conv2d(chans[i], chans[i + 1], (kernel, 1), (stride, 1), padding=pad)) for i in range(4)
        )
        self.convs.append(_wn(nn.Conv2d(1024, 1024, (kernel, 1), 1, padding=(2, 0))))
        self.conv_post = _wn(nn.Conv2d(1024, 1, (3, 1), 1, padding=(1, 0)))

    def forward(self, x):
        fmap = []
        b, c, t = x.shape
        if t % self.period != 0:
            n_pad = self.period - (t % self.period)
            x = F.pad(x, (0, n_pad), "reflect")
            t = t + n_pad
        x = x.view(b, c, t // self.period, self.period)
        for conv in self.convs:
            x = F.leaky_relu(conv(x), LRELU_SLOPE)
            fmap.append(x)
        x = self.conv_post(x)
        fmap.append(x)
        return torch.flatten(x, 1, -1), fmap


class MultiPeriodDiscriminatorRef(nn.Module):
    def __init__(self, periods=(2, 3, 5, 7, 11)):
        super().__init__()
        self.discriminators = nn.ModuleList(DiscriminatorPRef(p) for p in periods)

    def forward(self, y, y_hat):
        outs = [[], [], [], []]
        for d in self.discriminators:
            r, fr = d(y)
            g, fg = d(y_hat)
            for lst, v in zip(outs, (r, g, fr, fg)):
                lst.append(v)
        return tuple(outs)


class DiscriminatorSRef(nn.Module):
    def __init__(self, use_spectral_norm: bool = False):
        super().__init__()
        norm = _sn if use_spectral_norm else _wn
        self.convs = nn.ModuleList(
            [
                norm(nn.Conv1d(1, 128, 15, 1, padding=7)),
                norm(nn.Conv1d(128, 128, 41, 2, groups=4, padding=20)),
                norm(nn.Conv1d(128, 256, 41, 2, groups=16, padding=20)),
                norm(nn.Conv1d(256, 512, 41, 4, groups=16, padding=20)),
                norm(nn.Conv1d(512, 1024, 41, 4, groups=16, padding=20)),
                norm(nn.Conv1d(1024, 1024, 41, 1, groups=16, padding=20)),
                norm(nn.Conv1d(1024, 1024, 5, 1, padding=2)),
            ]
        )
        self.conv_post = norm(nn.Conv1d(1024, 1, 3, 1, padding=1))

    def forward(self, x):
        fmap = []
        for conv in self.convs:
            x = F.leaky_relu(conv(x), LRELU_SLOPE)
            fmap.append(x)
        x = self.conv_post(x)
        fmap.append(x)
        return torch.flatten(x, 1, -1), fmap


class MultiScaleDiscriminatorRef(nn.Module):
    def __init__(self, n_scales: int = 3):
        super().__init__()
        self.discriminators = nn.ModuleList(
            DiscriminatorSRef(use_spectral_norm=(i == 0)) for i in range(n_scales)
        )
        self.meanpools = nn.ModuleList(nn.AvgPool1d(4, 2, padding=2) for _ in range(n_scales - 1))

    def forward(self, y, y_hat):
        outs = [[], [], [], []]
        for i, d in enumerate(self.discriminators):
            if i != 0:
                y = self.meanpools[i - 1](y)
                y_hat = self.meanpools[i - 1](y_hat)
            r, fr = d(y)
            g, fg = d(y_hat)
            for lst, v in zip(outs, (r, g, fr, fg)):
                lst.append(v)
        return tuple(outs)


# ---- losses of the GAN step (LSGAN "original" gan_type) --------------------------------------


def feature_loss_ref(fmap_r, fmap_g):
    loss = 0.0
    for dr, dg in zip(fmap_r, fmap_g):
        for rl, gl in zip(dr, dg):
            loss = loss + torch.mean(torch.abs(rl - gl))
    return loss * 2


def discriminator_loss_ref(real_outs, gen_outs):
    loss = 0.0
    for dr, dg in zip(real_outs, gen_outs):
        loss = loss + torch.mean((1 - dr) ** 2) + torch.mean(dg**2)
    return loss


def generator_loss_ref(gen_outs):
    loss = 0.0
    for dg in gen_outs:
        loss = loss + torch.mean((1 - dg) ** 2)
    return loss


def count_params(module: nn.Module, trainable_only: bool = True) -> int:
    return sum(p.numel() for p in module.parameters() if p.requires_grad or not trainable_only)


def seeded_generator(cfg: HiFiGANModelConfigRef | None = None, seed: int = 1234, fold: bool = True) -> GeneratorRef:
    """Random-init generator under a fixed seed (torch.manual_seed(1234)),
    weight norm folded for inference when ``fold``."""
    gen_state = torch.random.get_rng_state()
    torch.manual_seed(seed)
    g = GeneratorRef(cfg)
    # make biases and conv_pre non-trivial but small so every term of every layer is exercised
    torch.random.set_rng_state(gen_state)
    g.eval()
    if fold:
        g.remove_weight_norm()
    return g


def mrstft_loss_ref(y, y_hat, resolutions=((1024, 120, 600), (2048, 240, 1200), (512, 50, 240)), eps: float = 1e-7):
    """Multi-resolution STFT loss (Yamamoto et al. 2020, Parallel WaveGAN: spectral convergence + log-magnitude L1, averaged
    over the resolutions (n_fft, hop, win)), the selectable alternative BASELINE.json's config 4 names next to the 45 x mel-L1
    term.  y, y_hat [B, T].  Magnitudes are sqrt(re^2 + im^2 + eps) (additive epsilon: smooth everywhere)."""
    import torch

    total = y.new_zeros(())
    for n_fft, hop, win in resolutions:
        window = torch.hann_window(win, periodic=True)
        def mag(x):
            spec = torch.stft(x, n_fft, hop, win, window, center=True, pad_mode="reflect", return_complex=True)
            return torch.sqrt(spec.real ** 2 + spec.imag ** 2 + eps)
        my, mg = mag(y), mag(y_hat)
        sc = torch.norm(my - mg, p="fro") / torch.norm(my, p="fro")
        lm = (torch.log(my) - torch.log(mg)).abs().mean()
        total = total + sc + lm
    return total / len(resolutions)
