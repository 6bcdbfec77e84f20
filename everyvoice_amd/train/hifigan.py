"""GAN training step of the HiFiGAN vocoder on libevmi_hip — the host-side mirror of the hot loop of
``hfgl.model.HiFiGAN.training_step`` (absent submodule): manual optimisation with
two optimisers, ``gan_type = "original"`` (LSGAN):

    y_hat = G(mel)
    D step:  loss_d = sum_i mean((1 - D_i(y))^2) + mean(D_i(y_hat.detach())^2)            -> AdamW(D)
    G step:  loss_g = sum_i mean((1 - D_i(y_hat))^2) + 2 * sum L1(fmaps) + 45 * L1(logmel(y), logmel(y_hat)) -> AdamW(G)

Layer structure and state-dict names follow upstream (jik876 HiFi-GAN generator / MPD / MSD; the first MSD scale
is spectral-normalised), see oracle/hifigan_ref.py for the pins.  All arithmetic runs in libevmi_hip kernels
(fp32, channel-major activations); torch owns memory and, for data parallel training, the RCCL all-reduce of
the two flat gradient buffers.
"""

from __future__ import annotations

import torch

from ..config import ACTIVATION_SLOPES, HiFiGANConfig
from ..spectral import slaney_mel_filterbank, windowed_dft_basis
from . import autograd as ag
from . import ops
from .layers import ParamGroup, SNConv, WNBatch, WNConv, kaiming_uniform_conv_init_


def _to_cbt(x_bct: torch.Tensor) -> torch.Tensor:
    return x_bct.permute(1, 0, 2).contiguous()  # layout change only (memory plumbing)


class GeneratorT:
    def __init__(self, cfg: HiFiGANConfig, group: ParamGroup):
        m = cfg.model
        self.resblock2 = str(getattr(m.resblock, "value", m.resblock)) == "2"
        self.istft = bool(m.istft_layer)
        self._istft_consts = None
        self._istft_cfg = (cfg.gen_istft_n_fft, cfg.gen_istft_hop_size)
        self.slope = ACTIVATION_SLOPES[m.activation_function]
        ch0, n_mels = m.upsample_initial_channel, cfg.preprocessing.audio.n_mels
        # declaration order = forward order: backward then finishes the flat gradient buffer suffix-first (BucketReducer)
        self.conv_pre = WNConv(group, "conv_pre", n_mels, ch0, 7, pad=3)
        self.ups, self.resblocks = [], []
        for i, (u, ku) in enumerate(zip(m.upsample_rates, m.upsample_kernel_sizes)):
            self.ups.append(WNConv(group, f"ups.{i}", ch0 >> i, ch0 >> (i + 1), ku, stride=u, pad=(ku - u) // 2, transposed=True))
            c = ch0 >> (i + 1)
            for j, (k, dils) in enumerate(zip(m.resblock_kernel_sizes, m.resblock_dilation_sizes)):
                n = i * len(m.resblock_kernel_sizes) + j
                if self.resblock2:  # 2 x [lrelu -> dilated conv] with a residual each (upstream ResBlock2: parameters convs.q)
                    self.resblocks.append([(WNConv(group, f"resblocks.{n}.convs.{q}", c, c, k, pad=d * (k - 1) // 2, dil=d), None)
                                           for q, d in enumerate(dils)])
                    continue
                self.resblocks.append([
                    (WNConv(group, f"resblocks.{n}.convs1.{q}", c, c, k, pad=d * (k - 1) // 2, dil=d),
                     WNConv(group, f"resblocks.{n}.convs2.{q}", c, c, k, pad=(k - 1) // 2))
                    for q, d in enumerate(dils)
                ])
        self.num_kernels = len(m.resblock_kernel_sizes)
        self.conv_post = WNConv(group, "conv_post", ch0 >> len(m.upsample_rates), cfg.gen_istft_n_fft + 2 if self.istft else 1, 7, pad=3)

    def layers(self):
        out = [self.conv_pre, *self.ups, self.conv_post]
        for rb in self.resblocks:
            for c1, c2 in rb:
                out += [c1] if c2 is None else [c1, c2]
        return out

    def stage_layers(self, i):
        out = [self.ups[i]]
        for j in range(self.num_kernels):
            for c1, c2 in self.resblocks[i * self.num_kernels + j]:
                out += [c1] if c2 is None else [c1, c2]
        return out

    def forward(self, tape: ag.Tape, mel: ag.Var, bucket_hook=None) -> ag.Var:
        """`bucket_hook(layers)` is called before the forward of each group of layers whose parameters form one gradient bucket."""
        hook = bucket_hook or (lambda layers: None)
        hook([self.conv_pre])
        x = ag.conv1d(tape, mel, self.conv_pre)
        for i, up in enumerate(self.ups):
            hook(self.stage_layers(i))
            x = ag.lrelu(tape, x, self.slope)
            x = ag.conv_transpose1d(tape, x, up)
            xs = None
            for j in range(self.num_kernels):
                y = x
                for c1, c2 in self.resblocks[i * self.num_kernels + j]:
                    t = ag.lrelu(tape, y, self.slope)
                    if c2 is None:
                        t = ag.conv1d(tape, t, c1)
                    else:
                        t = ag.conv1d_lrelu(tape, t, c1, self.slope)
                        t = ag.conv1d(tape, t, c2)
                    y = ag.add(tape, t, y)
                xs = y if xs is None else ag.add(tape, xs, y)
            x = ag.scale(tape, xs, 1.0 / self.num_kernels)
        hook([self.conv_post])
        x = ag.lrelu(tape, x, 0.01)
        if self.istft:  # iSTFTNet head: reflection pad -> conv_post (n_fft + 2 channels) -> exp / sin -> inverse STFT
            if self._istft_consts is None:
                self._istft_consts = ag.ISTFTConstants(*self._istft_cfg, x.data.device)
            x = ag.reflect_pad_left1(tape, x)
            x = ag.conv1d(tape, x, self.conv_post)
            return ag.istft(tape, x, self._istft_consts)
        x = ag.conv1d(tape, x, self.conv_post)
        return ag.tanh(tape, x)


class DiscriminatorPT:
    def __init__(self, group: ParamGroup, prefix: str, period: int):
        self.period = period
        chans = [1, 32, 128, 512, 1024]
        self.convs = [WNConv(group, f"{prefix}.convs.{i}", chans[i], chans[i + 1], 5, stride=3, pad=2, conv2d=True) for i in range(4)]
        self.convs.append(WNConv(group, f"{prefix}.convs.4", 1024, 1024, 5, pad=2, conv2d=True))
        self.conv_post = WNConv(group, f"{prefix}.conv_post", 1024, 1, 3, pad=1, conv2d=True)

    def layers(self):
        return [*self.convs, self.conv_post]

    def forward(self, tape, audio: ag.Var, training=True):
        x = ag.period_view(tape, audio, self.period)  # [1, B*p, H]: Conv2d((k,1)) == Conv1d over H per column
        fmap = []
        for conv in self.convs:
            x = ag.conv1d_lrelu(tape, x, conv, 0.1, training)
            fmap.append(x)
        x = ag.conv1d(tape, x, self.conv_post, training)
        fmap.append(x)
        return x, fmap


class DiscriminatorST:
    SPEC = [(1, 128, 15, 1, 1, 7), (128, 128, 41, 2, 4, 20), (128, 256, 41, 2, 16, 20), (256, 512, 41, 4, 16, 20),
            (512, 1024, 41, 4, 16, 20), (1024, 1024, 41, 1, 16, 20), (1024, 1024, 5, 1, 1, 2)]

    def __init__(self, group: ParamGroup, prefix: str, spectral: bool):
        cls = SNConv if spectral else WNConv
        self.convs = [cls(group, f"{prefix}.convs.{i}", ci, co, k, stride=s, pad=p, groups=g)
                      for i, (ci, co, k, s, g, p) in enumerate(self.SPEC)]
        self.conv_post = cls(group, f"{prefix}.conv_post", 1024, 1, 3, pad=1)

    def layers(self):
        return [*self.convs, self.conv_post]

    def forward(self, tape, x: ag.Var, training=True):
        fmap = []
        for conv in self.convs:
            x = ag.conv1d_lrelu(tape, x, conv, 0.1, training)
            fmap.append(x)
        x = ag.conv1d(tape, x, self.conv_post, training)
        fmap.append(x)
        return x, fmap


class MelLoss:
    """45 * L1(logmel(y), logmel(y_hat)) with the reference's mel-librosa front-end (heavy.py:69-100, 39-40), as
    GEMMs: frames [n_fft, B*F] -> (cos | sin) DFT -> magnitude -> mel basis -> log-clamp."""

    def __init__(self, audio_cfg, device):
        self.n_fft, self.hop = audio_cfg.n_fft, audio_cfg.fft_hop_size
        basis, nb_pad = windowed_dft_basis(self.n_fft, audio_cfg.fft_window_size)
        nb = self.n_fft // 2 + 1
        b = torch.from_numpy(basis)  # [n_fft, 2*nb_pad] interleaved (w cos, -w sin)
        self.cos = b[:, 0 : 2 * nb : 2].t().contiguous().to(device)  # [nb, n_fft]
        self.sin = b[:, 1 : 2 * nb : 2].t().contiguous().to(device)
        self.melb = torch.from_numpy(slaney_mel_filterbank(audio_cfg.input_sampling_rate, self.n_fft, audio_cfg.n_mels,
                                                           audio_cfg.f_min, audio_cfg.f_max)).to(device)
        self.nb, self.n_mels = nb, audio_cfg.n_mels

    def logmel(self, audio_bt: torch.Tensor):
        fr, F = ops.stft_frames(audio_bt, self.n_fft, self.hop)
        N = fr.shape[1]
        re = torch.empty(self.nb, N, device=fr.device)
        im = torch.empty(self.nb, N, device=fr.device)
        ops.gemm(self.cos, fr, re)
        ops.gemm(self.sin, fr, im)
        mag = ops.elementwise(ops.EW_MAG, re, im, p0=1e-9)
        mel = torch.empty(self.n_mels, N, device=fr.device)
        ops.gemm(self.melb, mag, mel)
        return ops.elementwise(ops.EW_LOG_CLAMP, mel, p0=1e-5), (re, im, mag, mel)

    def loss_and_grad(self, y_bt, yhat_bt, weight, loss_out):
        """loss_out[0] += weight * mean|logmel(y) - logmel(y_hat)|; returns d loss / d y_hat [B, T]."""
        B, T = yhat_bt.shape
        lm_y, _ = self.logmel(y_bt)
        lm_g, (re, im, mag, mel) = self.logmel(yhat_bt)
        n = lm_g.numel()
        ops.scalar_reduce(0, lm_g, lm_y, loss_out, scale=weight / n, accumulate=True)
        dlog = ops.elementwise(ops.EW_SIGN_DIFF, lm_g, lm_y, p0=weight / n)
        dmel = ops.elementwise(ops.EW_DIV_MASK, dlog, mel, p0=1e-5)
        dmag = torch.empty_like(mag)
        ops.gemm(self.melb, dmel, dmag, ta=True)
        dre = ops.elementwise(ops.EW_MUL_DIV, dmag, re, mag)
        dim = ops.elementwise(ops.EW_MUL_DIV, dmag, im, mag)
        dfr = torch.empty(self.n_fft, dre.shape[1], device=dre.device)
        ops.gemm(self.cos, dre, dfr, ta=True)
        ops.gemm(self.sin, dim, dfr, ta=True, beta=1.0)
        return ops.stft_frames_bwd(dfr, B, T, self.n_fft, self.hop)


class MultiResolutionSTFTLoss:
    """Multi-resolution STFT loss (spectral convergence + log-magnitude L1, mean over the resolutions; Yamamoto et al. 2020) --
    the selectable alternative to the 45 x mel-L1 term that BASELINE.json's config 4 names.  Same GEMM
    formulation as MelLoss: frames -> windowed DFT (cos | sin) -> sqrt(re^2 + im^2 + eps).  The two Frobenius norms of the
    spectral-convergence term are read back to the host (one sync per resolution)."""

    def __init__(self, device, resolutions=((1024, 120, 600), (2048, 240, 1200), (512, 50, 240)), eps=1e-7):
        self.eps, self.res = eps, []
        for n_fft, hop, win in resolutions:
            basis, _ = windowed_dft_basis(n_fft, win)
            nb = n_fft // 2 + 1
            b = torch.from_numpy(basis)
            self.res.append((n_fft, hop, nb, b[:, 0 : 2 * nb : 2].t().contiguous().to(device), b[:, 1 : 2 * nb : 2].t().contiguous().to(device)))

    def _mag(self, audio_bt, n_fft, hop, nb, cos, sin):
        fr, _ = ops.stft_frames(audio_bt, n_fft, hop)
        re = torch.empty(nb, fr.shape[1], device=fr.device)
        im = torch.empty_like(re)
        ops.gemm(cos, fr, re)
        ops.gemm(sin, fr, im)
        return ops.elementwise(ops.EW_MAG, re, im, p0=self.eps), re, im

    def loss_and_grad(self, y_bt, yhat_bt, weight, loss_out):
        """loss_out[0] += weight * mean_r (sc_r + logmag_r); returns d loss / d y_hat [B, T]."""
        B, T = yhat_bt.shape
        grad = None
        w = weight / len(self.res)
        for n_fft, hop, nb, cos, sin in self.res:
            my, _, _ = self._mag(y_bt, n_fft, hop, nb, cos, sin)
            mg, re, im = self._mag(yhat_bt, n_fft, hop, nb, cos, sin)
            n = mg.numel()
            sq = torch.zeros(2, device=mg.device)
            ops.scalar_reduce(1, ops.axpby(1.0, mg, -1.0, my), None, sq[0:1], p=0.0)   # ||mg - my||^2
            ops.scalar_reduce(1, my, None, sq[1:2], p=0.0)                              # ||my||^2
            nd, ny = (float(v) ** 0.5 for v in sq.tolist())
            loss_out += w * nd / ny
            ops.scalar_reduce(0, ops.elementwise(16, mg), ops.elementwise(16, my), loss_out, scale=w / n, accumulate=True)
            # d/dmg: spectral convergence (mg - my) / (||mg - my|| ||my||), log-magnitude sign(mg - my) / (n mg)
            dmag = ops.elementwise(17, mg, my, p0=w / (nd * ny) if nd > 0 else 0.0, p1=w / n)
            dre = ops.elementwise(ops.EW_MUL_DIV, dmag, re, mg)
            dim = ops.elementwise(ops.EW_MUL_DIV, dmag, im, mg)
            dfr = torch.empty(n_fft, dre.shape[1], device=dre.device)
            ops.gemm(cos, dre, dfr, ta=True)
            ops.gemm(sin, dim, dfr, ta=True, beta=1.0)
            g = ops.stft_frames_bwd(dfr, B, T, n_fft, hop)
            grad = g if grad is None else ops.axpby(1.0, grad, 1.0, g)
        return grad


def allreduce_mean_(flat_grad: torch.Tensor, process_group, scale_fn) -> torch.Tensor:
    """flat_grad <- mean over ranks (sum all-reduce, then ``scale_fn(flat_grad, 1 / world)``)."""
    import torch.distributed as dist

    dist.all_reduce(flat_grad, op=dist.ReduceOp.SUM, group=process_group)
    scale_fn(flat_grad, 1.0 / dist.get_world_size(process_group))
    return flat_grad


class BucketReducer:
    """Data-parallel gradient exchange overlapped with backward: the flat gradient buffer of one optimiser is
    reduced in contiguous buckets, each launched -- asynchronously, on a side stream when the buffer lives on a GPU -- the
    moment backward has finished the last layer that writes into it; ``finish()`` waits for all of them and applies the
    1/world scaling.  Backward visits the layers in reverse declaration order, so finished gradients form a growing suffix
    of the buffer: ``launch(lo, hi)`` is called with adjacent, descending ranges.  RCCL over xGMI under backend "nccl"."""

    def __init__(self, flat_grad: torch.Tensor, process_group, scale_fn):
        self.flat, self.pg, self.scale_fn = flat_grad, process_group, scale_fn
        self.works = []
        self.stream = torch.cuda.Stream(flat_grad.device) if flat_grad.is_cuda else None

    def launch(self, lo: int, hi: int) -> None:
        import torch.distributed as dist

        if hi <= lo:
            return
        chunk = self.flat[lo:hi]
        if self.stream is not None:
            ready = torch.cuda.Event()
            ready.record(torch.cuda.current_stream(self.flat.device))  # gradients of this bucket are final from here on
            with torch.cuda.stream(self.stream):
                self.stream.wait_event(ready)
                self.works.append(dist.all_reduce(chunk, op=dist.ReduceOp.SUM, group=self.pg, async_op=True))
        else:
            self.works.append(dist.all_reduce(chunk, op=dist.ReduceOp.SUM, group=self.pg, async_op=True))

    def finish(self) -> None:
        import torch.distributed as dist

        for w in self.works:
            w.wait()
        self.works.clear()
        if self.stream is not None:
            torch.cuda.current_stream(self.flat.device).wait_stream(self.stream)
        self.scale_fn(self.flat, 1.0 / dist.get_world_size(self.pg))


class HiFiGANTrainer:
    """Generator + MPD + MSD with two AdamW optimisers; ``training_step`` is one full GAN step."""

    def __init__(self, config: HiFiGANConfig | None = None, device="cuda:0", lr=2e-4, betas=(0.8, 0.99), eps=1e-8,
                 weight_decay=0.01, seed=1234, process_group=None, reconstruction_loss="mel", stft_loss_weight=45.0,
                 precision="f32"):
        if precision not in ("f32", "bf16"):
            raise ValueError("precision: 'f32' (exact fp32 arithmetic) or 'bf16' (bf16 convolution operands, fp32 accumulation, "
                             "fp32 master weights and activations: the mixed-precision counterpart of Lightning's bf16-mixed)")
        self.precision = precision
        self.config = config or HiFiGANConfig()
        self.device = torch.device(device)
        self.opt = dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay)
        self.pg = process_group
        self.g_params, self.d_params = ParamGroup(self.device), ParamGroup(self.device)
        self.generator = GeneratorT(self.config, self.g_params)
        m = self.config.model
        self.mpd = [DiscriminatorPT(self.d_params, f"mpd.discriminators.{i}", p) for i, p in enumerate(m.mpd_layers)]
        self.msd = [DiscriminatorST(self.d_params, f"msd.discriminators.{i}", spectral=(i == 0)) for i in range(m.msd_layers)]
        self.g_params.finalize()
        self.d_params.finalize()
        # weight norm of all layers of an optimiser in one launch (forward), one launch per gradient bucket (backward)
        self._wn_batches = [WNBatch(self.g_params, self.generator.layers()), WNBatch(self.d_params, self.d_layers())]
        self.mel_loss = MelLoss(self.config.preprocessing.audio, self.device)
        if reconstruction_loss not in ("mel", "mrstft", "mel+mrstft"):
            raise ValueError("reconstruction_loss: 'mel' (45 x mel-L1, the upstream default), 'mrstft' or 'mel+mrstft'")
        self.reconstruction_loss, self.stft_loss_weight = reconstruction_loss, stft_loss_weight
        self.stft_loss = MultiResolutionSTFTLoss(self.device) if "mrstft" in reconstruction_loss else None
        gen = torch.Generator().manual_seed(seed)
        for layer in self.generator.layers():
            std = None if layer.name == "conv_pre" else 0.01  # upstream init_weights: N(0, 0.01) except conv_pre
            kaiming_uniform_conv_init_(layer, gen, std)
        for d in [*self.mpd, *self.msd]:
            for layer in d.layers():
                kaiming_uniform_conv_init_(layer, gen)
        self.global_step = 0
        self.keep_grads = False  # tests: keep copies of both gradient buffers of the last step
        self.last_grads = {}

    # -- state ------------------------------------------------------------------------------------------
    def d_layers(self):
        return [layer for d in [*self.mpd, *self.msd] for layer in d.layers()]

    def load_reference_state(self, gen_sd=None, mpd_sd=None, msd_sd=None):
        """Load upstream-named state dicts (weight_g / weight_v / weight_orig / bias, SN buffers weight_u / weight_v)."""
        def put(group, prefix, sd, layers):
            by_name = {l.name: l for l in layers}
            for k, v in sd.items():
                name = prefix + k
                base = name.rsplit(".", 1)[0]
                layer = by_name.get(base)
                if layer is None:
                    raise KeyError(name)
                leaf = name.rsplit(".", 1)[1]
                if isinstance(layer, SNConv) and leaf in ("weight_u", "weight_v"):
                    (layer.u if leaf == "weight_u" else layer.v).copy_(v.to(self.device))
                else:
                    group.load(name, v)
        if gen_sd is not None:
            put(self.g_params, "", gen_sd, self.generator.layers())
        if mpd_sd is not None:
            put(self.d_params, "mpd.", mpd_sd, self.d_layers())
        if msd_sd is not None:
            put(self.d_params, "msd.", msd_sd, self.d_layers())

    # -- checkpoint / resume / export (reference conventions: everyvoice/tests/test_model.py:85-151, 302-313, 454-459) --
    _VERSION = "1.0"

    def state_dict(self) -> dict:
        """Reference layout of the ``HiFiGAN`` LightningModule: ``generator.*``, ``mpd.*``, ``msd.*`` with the upstream
        parameter names (weight_g / weight_v, and weight_orig + weight_u / weight_v buffers of the spectral-norm scale)."""
        sd = {"generator." + k: v.cpu() for k, v in self.g_params.state_dict().items()}
        sd.update({k: v.cpu() for k, v in self.d_params.state_dict().items()})
        for layer in self.d_layers():
            if isinstance(layer, SNConv):
                sd[layer.name + ".weight_u"] = layer.u.detach().cpu().clone()
                sd[layer.name + ".weight_v"] = layer.v.detach().cpu().clone()
        return sd

    def checkpoint(self) -> dict:
        """A Lightning-shaped checkpoint dict: ``state_dict``, JSON-only ``hyper_parameters["config"]``, ``model_info``, the
        step counters and both optimisers' moments (flat buffers: this trainer's own optimiser-state format)."""
        return {
            "epoch": 0, "global_step": self.global_step, "state_dict": self.state_dict(),
            "hyper_parameters": {"config": self.config.model_dump(mode="json")},
            "model_info": {"name": "HiFiGAN", "version": self._VERSION},
            "optimizer_states": [
                {"evmi_flat_adamw": {"group": name, "step": grp.step, "exp_avg": grp.m.cpu(), "exp_avg_sq": grp.v.cpu(), **self.opt}}
                for name, grp in (("generator", self.g_params), ("discriminators", self.d_params))],
        }

    def load_checkpoint(self, ckpt: dict, restore_optimizers: bool = True):
        info = ckpt.get("model_info") if isinstance(ckpt, dict) else None
        if isinstance(info, dict) and info.get("name") != "HiFiGAN":
            raise TypeError(f"Wrong model type ({info.get('name')}), we are expecting a 'HiFiGAN' model")
        if isinstance(info, dict):
            ck_major, my_major = str(info.get("version", "1.0")).split(".")[0], self._VERSION.split(".")[0]
            if int(ck_major) > int(my_major):
                raise ValueError("Your model was created with a newer version of EveryVoice, please update your software.")
        try:
            sd = ckpt["state_dict"]
        except (KeyError, TypeError) as e:
            raise TypeError("Unable to load config.  Possible causes: is it really a VocoderConfig? or the correct version?") from e
        strip = lambda pre: {k[len(pre):]: v for k, v in sd.items() if k.startswith(pre)}
        self.load_reference_state(strip("generator."), strip("mpd."), strip("msd."))
        self.global_step = int(ckpt.get("global_step", 0))
        if restore_optimizers:
            for st in ckpt.get("optimizer_states", []):
                st = st.get("evmi_flat_adamw") if isinstance(st, dict) else None
                if st is None:
                    continue
                grp = self.g_params if st["group"] == "generator" else self.d_params
                grp.m.copy_(st["exp_avg"].to(self.device))
                grp.v.copy_(st["exp_avg_sq"].to(self.device))
                grp.step = int(st["step"])
        return self

    def export_generator_checkpoint(self) -> dict:
        """What ``everyvoice export spec-to-wav`` writes (cli.py:381-386): the generator alone, loadable by
        ``everyvoice_amd.vocoder.load_hifigan_from_checkpoint`` for inference."""
        return {"state_dict": {"generator." + k: v.cpu() for k, v in self.g_params.state_dict().items()},
                "hyper_parameters": {"config": self.config.model_dump(mode="json")},
                "model_info": {"name": "HiFiGANGenerator", "version": self._VERSION}}

    def _materialize(self, layers):
        batches = []
        for layer in layers:
            b = getattr(layer, "_batch", None)
            if b is None:
                layer.materialize()
            elif b not in batches:
                batches.append(b)
        for b in batches:
            b.materialize()

    def _reducer(self, group: ParamGroup):
        """Bucketed all-reduce of one optimiser's flat gradient buffer, overlapped with backward (None on one GPU)."""
        if self.pg is None:
            return None
        return BucketReducer(group.grad, self.pg if self.pg is not True else None,
                             lambda t, sc: ops.elementwise(ops.EW_SCALE, t, out=t, p0=sc))

    @staticmethod
    def _bucket_hook(tape, group: ParamGroup, layers, reducer, state):
        """Record, BEFORE the forward of `layers`, the closure that backward runs AFTER all of their gradient closures:
        turn the effective-weight gradients into parameter gradients and hand the now-final slice of the flat buffer to the
        reducer.  `state["hi"]` is the start of the suffix already handed over."""
        def done():
            batches = {}
            for layer in layers:
                b = getattr(layer, "_batch", None)
                if b is None:
                    layer.finish_grads()
                else:
                    batches.setdefault(id(b), (b, []))[1].append(layer)
            for b, ls in batches.values():
                b.finish(ls)
            if reducer is not None:
                lo = min(group.offset_of(n) for layer in layers for n in layer.param_names())
                reducer.launch(lo, state["hi"])
                state["hi"] = lo
        tape.record(done)

    # -- discriminators on one waveform ----------------------------------------------------------------------
    def _discriminate(self, tape, audio: ag.Var, training=True):
        logits, fmaps = [], []
        for d in self.mpd:
            o, f = d.forward(tape, audio, training)
            logits.append(o)
            fmaps.append(f)
        x = audio
        for i, d in enumerate(self.msd):
            if i > 0:
                x = ag.avgpool4s2(tape, x)
            o, f = d.forward(tape, x, training)
            logits.append(o)
            fmaps.append(f)
        return logits, fmaps

    def _discriminate_pair(self, tape, y: torch.Tensor, y_hat: torch.Tensor, reducer=None, state=None):
        """Discriminator step: real and generated waveforms as ONE batch of 2B items (columns of the same GEMMs), so
        every convolution, its input gradient and its weight gradient run once over twice the columns.  The
        spectral-norm scale discriminator keeps the reference's two forward calls: each call runs its own power
        iteration and sees its own sigma.  Returns [(logits Var, "pair" | "real" | "fake")]."""
        pair = ag.Var(torch.cat([y, y_hat], dim=1), needs_grad=False)
        outs = []
        for d in self.mpd:
            self._bucket_hook(tape, self.d_params, d.layers(), reducer, state)
            outs.append((d.forward(tape, pair)[0], "pair"))
        x = pair
        for i, d in enumerate(self.msd):
            if i > 0:
                x = ag.avgpool4s2(tape, x)
            self._bucket_hook(tape, self.d_params, d.layers(), reducer, state)
            if any(isinstance(layer, SNConv) for layer in d.layers()):
                outs.append((d.forward(tape, ag.Var(y, needs_grad=False))[0], "real"))
                outs.append((d.forward(tape, ag.Var(y_hat, needs_grad=False))[0], "fake"))
            else:
                outs.append((d.forward(tape, x)[0], "pair"))
        return outs

    # -- one GAN step -----------------------------------------------------------------------------------------
    def training_step(self, mel_bct: torch.Tensor, audio_bct: torch.Tensor) -> dict:
        """mel [B, n_mels, T/hop], audio [B, 1, T] on the device.  Returns the scalar losses (python floats)."""
        prev = ops.CONV_BACKEND["operands"]
        ops.CONV_BACKEND["operands"] = self.precision
        try:
            return self._training_step(mel_bct, audio_bct)
        finally:
            ops.CONV_BACKEND["operands"] = prev

    def _training_step(self, mel_bct: torch.Tensor, audio_bct: torch.Tensor) -> dict:
        dev = self.device
        B = audio_bct.shape[0]
        y = audio_bct.to(torch.float32).reshape(1, B, -1).contiguous()  # [B,1,T] and [1,B,T] are the same bytes
        mel = _to_cbt(mel_bct.to(torch.float32))
        g_layers, d_layers = self.generator.layers(), self.d_layers()
        self._materialize(g_layers)
        self._materialize(d_layers)
        losses = {k: torch.zeros(1, device=dev) for k in ("d", "g_adv", "g_fm", "g_mel", "g_stft")}

        # ---- generator forward (tape kept for the generator step) ----
        g_tape = ag.Tape()
        g_reducer, g_state = self._reducer(self.g_params), {"hi": self.g_params.grad.numel()}
        y_hat = self.generator.forward(g_tape, ag.Var(mel, needs_grad=False),
                                       lambda layers: self._bucket_hook(g_tape, self.g_params, layers, g_reducer, g_state))

        # ---- discriminator step ----
        self.d_params.zero_grad()
        for layer in d_layers:
            layer.frozen = False
        d_tape = ag.Tape()
        d_reducer, d_state = self._reducer(self.d_params), {"hi": self.d_params.grad.numel()}
        for o, kind in self._discriminate_pair(d_tape, y, y_hat.data, d_reducer, d_state):  # y_hat.detach()
            if kind == "pair":  # real items first, generated items second along the batch axis
                n, h = o.data.numel() // 2, o.data.shape[1] // 2  # period discriminators: the batch axis is (item, column)
                o.grad = torch.empty_like(o.data)
                parts = ((o.data[:, :h], o.grad[:, :h], 1.0), (o.data[:, h:], o.grad[:, h:], 0.0))
            else:
                n = o.data.numel()
                o.grad = torch.empty_like(o.data)
                parts = ((o.data, o.grad, 1.0 if kind == "real" else 0.0),)
            for logits, grad, target in parts:
                ops.scalar_reduce(1, logits, None, losses["d"], scale=1.0 / n, p=target, accumulate=True)
                ops.elementwise(ops.EW_SQ_GRAD, logits, out=grad, p0=1.0 / n, p1=target)
        d_tape.backward()  # every discriminator's bucket is finished (and its all-reduce launched) as backward leaves it
        if d_reducer is not None:
            d_reducer.launch(0, d_state["hi"])  # alignment padding in front of the first parameter, if any
            d_reducer.finish()
        if self.keep_grads:
            self.last_grads["d"] = {k: v.clone() for k, v in self.d_params.gradients().items()}
        self.d_params.adamw(**self.opt)
        self._materialize(d_layers)  # the generator step sees the updated discriminators

        # ---- generator step ----
        self.g_params.zero_grad()
        for layer in d_layers:
            layer.frozen = True  # gradients flow through the discriminators to y_hat only
        gd_tape = ag.Tape()
        y_hat_in = ag.Var(y_hat.data)  # boundary between the discriminator tape and the generator tape
        _, fmaps_r = self._discriminate(gd_tape, ag.Var(y, needs_grad=False))
        fake_logits, fmaps_g = self._discriminate(gd_tape, y_hat_in)
        for dg in fake_logits:
            n = dg.data.numel()
            ops.scalar_reduce(1, dg.data, None, losses["g_adv"], scale=1.0 / n, p=1.0, accumulate=True)
            dg.grad = ops.elementwise(ops.EW_SQ_GRAD, dg.data, p0=1.0 / n, p1=1.0)
        for fr_list, fg_list in zip(fmaps_r, fmaps_g):
            for fr, fg in zip(fr_list, fg_list):
                n = fg.data.numel()
                ops.scalar_reduce(0, fg.data, fr.data, losses["g_fm"], scale=2.0 / n, accumulate=True)
                fg.accumulate(ops.elementwise(ops.EW_SIGN_DIFF, fg.data, fr.data, p0=2.0 / n))
        gd_tape.backward()
        for layer in d_layers:
            if isinstance(layer, SNConv):
                layer._calls.clear()  # frozen: no parameter gradients from this pass
        if "mel" in self.reconstruction_loss.split("+"):
            total = self.mel_loss.loss_and_grad(y.view(B, -1), y_hat.data.view(B, -1), 45.0, losses["g_mel"]).view(1, B, -1)
        else:
            total = torch.zeros(1, B, y.shape[-1], device=dev)
        if self.stft_loss is not None:
            d_stft = self.stft_loss.loss_and_grad(y.view(B, -1), y_hat.data.view(B, -1), self.stft_loss_weight, losses["g_stft"])
            total = ops.axpby(1.0, total, 1.0, d_stft.view(1, B, -1))
        if y_hat_in.grad is not None:
            total = ops.axpby(1.0, total, 1.0, y_hat_in.grad)
        y_hat.grad = total
        g_tape.backward()  # buckets: conv_post, the four upsampling stages, conv_pre
        if g_reducer is not None:
            g_reducer.launch(0, g_state["hi"])
            g_reducer.finish()
        if self.keep_grads:
            self.last_grads["g"] = {k: v.clone() for k, v in self.g_params.gradients().items()}
            self.last_grads["y_hat"] = y_hat.data.clone()
        self.g_params.adamw(**self.opt)
        self.global_step += 1
        out = {k: float(v.item()) for k, v in losses.items()}
        out["g_total"] = out["g_adv"] + out["g_fm"] + out["g_mel"] + out["g_stft"]
        return out
