"""GAN training step of the HiFiGAN vocoder on libevmi_hip — the host-side mirror of the hot loop of
``hfgl.model.HiFiGAN.training_step`` (absent submodule): manual optimisation with
two optimisers, ``gan_type = "original"`` (LSGAN):

    y_hat = G(mel)
    D step:  loss_d = sum_i mean((1 - D_i(y))^2) + mean(D_i(y_hat.detach())^2)            -> AdamW(D)
    G step:  loss_g = sum_i mean((1 - D_i(y_hat))^2) + 2 * sum L1(fmaps) + 45 * L1(logmel(y), logmel(y_hat)) -> AdamW(G)

``gan_type = "wgan"`` (critic losses mean D(y_hat) - mean D(y) / -mean D(y_hat), weight clipping to +-wgan_clip_value after
the critic's optimiser step), ``generator_warmup_steps`` (the first steps train the generator on the reconstruction loss
alone) and the optimiser union Adam / AdamW / RMSprop are the options the reference's training schema freezes
(everyvoice/.schema/everyvoice-spec-to-wav-0.5.json:434-622); their arithmetic inside the absent submodule is restated from
the published algorithms (parity unpinned, like the rest of the model code).

Layer structure and state-dict names follow upstream (jik876 HiFi-GAN generator / MPD / MSD; the first MSD scale
is spectral-normalised), see oracle/hifigan_ref.py for the pins.  All arithmetic runs in libevmi_hip kernels
(fp32, channel-major activations); torch owns memory and, for data parallel training, the RCCL all-reduce of
the two flat gradient buffers.

Execution: the step makes no host round trip until its losses are read (spectral-norm scales, loss norms and the optimisers'
step counters stay on the device).  The eight discriminators -- and the three MRF branches of every generator stage -- are
independent chains of small kernels: each runs on its own HIP stream (fork / join around them, forward and backward), and with
fixed segment shapes the whole step is captured once into a HIP graph and replayed (``use_graph``), so launch-bound layers of
different discriminators overlap on the device instead of queueing behind each other.
"""

from __future__ import annotations

import torch

from .. import _lib
from ..config import ACTIVATION_SLOPES, HiFiGANConfig
from ..spectral import slaney_mel_filterbank, windowed_dft_basis
from . import autograd as ag
from . import ops
from .layers import ParamGroup, SNConv, WNBatch, WNConv, kaiming_uniform_conv_init_


def _to_cbt(x_bct: torch.Tensor) -> torch.Tensor:
    return x_bct.permute(1, 0, 2).contiguous()  # layout change only (memory plumbing)


class Branches:
    """Fork / join of independent chains onto side HIP streams.  ``run(fns)``: fn i runs on stream i (mod the pool), ordered
    after everything already queued on the current stream; the current stream continues after all of them.  Work queued
    this way is captured into a HIP graph like any other (the fork / join events become graph edges)."""

    def __init__(self, device, n: int, enabled: bool = True):
        self.device = torch.device(device)
        self.streams = [torch.cuda.Stream(self.device) for _ in range(n)] if (enabled and self.device.type == "cuda") else []
        self.timing = None  # a list: every fork / join appends the (start, end) events of its branches (diagnostics)

    def run(self, fns) -> None:
        self.run_indexed(list(enumerate(fns)))

    def run_indexed(self, items) -> None:
        """items: (stream index, fn) in HOST issue order; fn runs on stream (index mod pool size)."""
        if not self.streams or len(items) < 2:
            for _, fn in items:
                fn()
            return
        main = torch.cuda.current_stream(self.device)
        fork = torch.cuda.Event()
        fork.record(main)
        used = self.streams[: min(max(j for j, _ in items) + 1, len(self.streams))]
        for st in used:
            st.wait_event(fork)
        evs = []
        try:
            for i, fn in items:
                with torch.cuda.stream(used[i % len(used)]):
                    if self.timing is not None:
                        e0 = torch.cuda.Event(enable_timing=True)
                        e0.record()
                    fn()
                    if self.timing is not None:
                        e1 = torch.cuda.Event(enable_timing=True)
                        e1.record()
                        evs.append((e0, e1))
        finally:  # (also when a branch raised: a stream capture must not be left with forked streams unjoined)
            for st in used:
                main.wait_stream(st)
        if self.timing is not None:
            self.timing.append(evs)


def parallel_section(tape: ag.Tape, branches: Branches, fns):
    """fns[i](sub_tape) -> result.  Every fn records on its own tape and runs on its own stream; the recorded backward runs the
    sub-tapes the same way.  Inputs shared between branches must enter them as separate leaf Vars (see ``fan_out``)."""
    subs = [ag.Tape() for _ in fns]
    results = [None] * len(fns)

    def fwd(i):
        def go():
            results[i] = fns[i](subs[i])
        return go

    branches.run([fwd(i) for i in range(len(fns))])
    tape.record(lambda: branches.run([sub.backward for sub in subs]))
    return results


def fan_out(tape: ag.Tape, x: ag.Var, n: int):
    """n leaf Vars over x's data, one per parallel branch; their gradients are added into x (in branch order) after the
    branches' backward has been joined.  Call BEFORE ``parallel_section`` (the tape runs in reverse)."""
    leaves = [ag.Var(x.data, needs_grad=x.needs_grad) for _ in range(n)]

    def gather():
        for leaf in leaves:
            if leaf.grad is not None:
                x.accumulate(leaf.grad)

    tape.record(gather)
    return leaves


class GeneratorT:
    def __init__(self, cfg: HiFiGANConfig, group: ParamGroup):
        m = cfg.model
        self.resblock2 = str(getattr(m.resblock, "value", m.resblock)) == "2"
        self.fused_pairs = True  # residual pairs as ag.resblock_pair (False: the unfused op sequence, for A/B tests)
        # precision="bf16": the residual stacks in time-major bf16 on the inference convolution kernels (train/mrf_tm.py) where
        # those kernels take the stage's shapes; False: the channel-major packed path everywhere (A/B switch, other configurations)
        import os

        self.time_major = os.environ.get("EVMI_TRAIN_TM", "1") == "1"
        self._tm_stages: dict = {}
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

    def _mrf_branch(self, tape: ag.Tape, y: ag.Var, i: int, j: int) -> ag.Var:
        for c1, c2 in self.resblocks[i * self.num_kernels + j]:
            if c2 is not None and self.fused_pairs:
                y = ag.resblock_pair(tape, y, c1, c2, self.slope)
                continue
            t = ag.lrelu(tape, y, self.slope)
            if c2 is None:
                t = ag.conv1d(tape, t, c1)
            else:
                t = ag.conv1d_lrelu(tape, t, c1, self.slope)
                t = ag.conv1d(tape, t, c2)
            y = ag.add(tape, t, y)
        return y

    def _tm_stage(self, i: int, x: torch.Tensor):
        """The time-major MRF of stage i when it applies: bf16 operands, ResBlock1 pairs, shapes the inference kernels take."""
        if not (self.time_major and self.fused_pairs and not self.resblock2 and x.is_cuda and ops.CONV_BACKEND["operands"] == "bf16"
                and ops.CONV_BACKEND["packed"] and 1 < self.num_kernels <= 3):
            return None
        # the decision is per (stage, length): the kernels take row counts that are multiples of 16 (a decision cached from
        # the first shape would send other lengths down the time-major path whatever their length)
        if x.shape[2] % 16:
            return None
        st = self._tm_stages.get(i)
        if st is None:
            from .mrf_tm import MRFStageTM, stage_supported

            pairs = [self.resblocks[i * self.num_kernels + j] for j in range(self.num_kernels)]
            ok = stage_supported(x.shape[0], [p[0][0].k for p in pairs], [[c1.dil for c1, _ in p] for p in pairs])
            st = self._tm_stages[i] = MRFStageTM(x.shape[0], pairs, self.slope, x.device) if ok else False
        return st or None

    def forward(self, tape: ag.Tape, mel: ag.Var, bucket_hook=None, branches: Branches | None = None) -> ag.Var:
        """`bucket_hook(layers)` is called before the forward of each group of layers whose parameters form one gradient bucket.
        ``branches``: the MRF branches of a stage (kernel sizes 3 / 7 / 11: independent residual chains) run side by side."""
        hook = bucket_hook or (lambda layers: None)
        hook([self.conv_pre])
        x = ag.conv1d(tape, mel, self.conv_pre)
        for i, up in enumerate(self.ups):
            hook(self.stage_layers(i))
            x = ag.lrelu(tape, x, self.slope)
            x = ag.conv_transpose1d(tape, x, up)
            tm = self._tm_stage(i, x.data)
            if tm is not None:  # the whole MRF of this stage in the inference layout: one op, its own backward
                x = tm.apply(tape, x, branches)
                continue
            if branches is not None and self.num_kernels > 1:  # (also without side streams: same summation order either way)
                leaves = fan_out(tape, x, self.num_kernels)
                ys = parallel_section(tape, branches, [(lambda sub, j=j: self._mrf_branch(sub, leaves[j], i, j)) for j in range(self.num_kernels)])
            else:
                ys = [self._mrf_branch(tape, x, i, j) for j in range(self.num_kernels)]
            xs = ys[0]
            for y in ys[1:]:
                xs = ag.add(tape, xs, y)
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


def _chain_for(disc, period: int, x: torch.Tensor):
    """The flat packed bf16 chain of a discriminator (train/disc_chain.py) when it applies: precision "bf16" on the packed kernels,
    on a GPU; ``EVMI_DISC_CHAIN=0`` keeps the op-by-op channel-major path (A/B switch; the tests run both)."""
    if not (x.is_cuda and ops._packed() and _DISC_CHAIN):
        return None
    ch = getattr(disc, "_chain", None)
    if ch is None:
        from .disc_chain import DiscChain

        ch = disc._chain = DiscChain(disc, period, x.device)
    return ch if ch.ok else None


import os as _os

_DISC_CHAIN = _os.environ.get("EVMI_DISC_CHAIN", "1") == "1"


class DiscriminatorPT:
    def __init__(self, group: ParamGroup, prefix: str, period: int):
        self.period = period
        chans = [1, 32, 128, 512, 1024]
        self.convs = [WNConv(group, f"{prefix}.convs.{i}", chans[i], chans[i + 1], 5, stride=3, pad=2, conv2d=True) for i in range(4)]
        self.convs.append(WNConv(group, f"{prefix}.convs.4", 1024, 1024, 5, pad=2, conv2d=True))
        self.conv_post = WNConv(group, f"{prefix}.conv_post", 1024, 1, 3, pad=1, conv2d=True)

    def layers(self):
        return [*self.convs, self.conv_post]

    def forward(self, tape, audio: ag.Var, training=True, role="pair", grad_from=None):
        chain = _chain_for(self, self.period, audio.data)
        if chain is not None:
            return chain.forward(tape, audio, training, role, grad_from)
        assert grad_from is None, "a [real | generated] batch is a packed chain's (train/disc_chain.py)"
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

    def forward(self, tape, x: ag.Var, training=True, role="pair", grad_from=None):
        chain = _chain_for(self, 1, x.data)
        if chain is not None:
            return chain.forward(tape, x, training, role, grad_from)
        assert grad_from is None, "a [real | generated] batch is a packed chain's (train/disc_chain.py)"
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
    spectral-convergence term stay on the device (the loss and its gradient kernels read them there)."""

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
            sq = torch.empty(2, device=mg.device)
            ops.scalar_reduce(1, ops.axpby(1.0, mg, -1.0, my), None, sq[0:1], p=0.0)   # ||mg - my||^2
            ops.scalar_reduce(1, my, None, sq[1:2], p=0.0)                              # ||my||^2
            _lib.check(_lib.load().evmi_ratio_accumulate_f32(loss_out.data_ptr(), sq.data_ptr(), w, _lib.current_stream_ptr(mg.device)),
                       "evmi_ratio_accumulate_f32")                                     # loss += w ||mg - my|| / ||my||
            ops.scalar_reduce(0, ops.elementwise(16, mg), ops.elementwise(16, my), loss_out, scale=w / n, accumulate=True)
            # d/dmg: spectral convergence (mg - my) / (||mg - my|| ||my||), log-magnitude sign(mg - my) / (n mg)
            dmag = ops.elementwise(ops.EW_STFT_GRAD_DEV, mg, my, sq, p0=w, p1=w / n)
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
        self.timing = None  # a list: every launch appends (start, end) events on the side stream (bench.py: all-reduce ms per step)

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
                if self.timing is not None:
                    e0 = torch.cuda.Event(enable_timing=True)
                    e0.record(self.stream)
                # RCCL: enqueued behind `ready` on the side stream, runs while the launching stream goes on with backward
                work = dist.all_reduce(chunk, op=dist.ReduceOp.SUM, group=self.pg, async_op=True)
                if self.timing is not None:  # (measurement runs only: order the side stream behind the collective, then stamp)
                    work.wait()
                    e1 = torch.cuda.Event(enable_timing=True)
                    e1.record(self.stream)
                    self.timing.append((e0, e1))
                else:
                    self.works.append(work)
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

    def comm_ms(self, reset: bool = True) -> float:
        """Summed device time of the recorded all-reduces (synchronises)."""
        if not self.timing:
            return 0.0
        self.timing[-1][1].synchronize()
        ms = sum(a.elapsed_time(b) for a, b in self.timing)
        if reset:
            self.timing.clear()
        return ms


class HiFiGANTrainer:
    """Generator + MPD + MSD with two optimisers; ``training_step`` is one full GAN step."""

    LOSS_KEYS = ("d", "g_adv", "g_fm", "g_mel", "g_stft")

    def __init__(self, config: HiFiGANConfig | None = None, device="cuda:0", lr=2e-4, betas=(0.8, 0.99), eps=1e-8,
                 weight_decay=0.01, seed=1234, process_group=None, reconstruction_loss="mel", stft_loss_weight=45.0,
                 precision="f32", optimizer="adamw", alpha=0.99, gan_type="original", wgan_clip_value=0.01,
                 generator_warmup_steps=0, use_graph=False, parallel_streams=True, side_wgrad=False):
        if precision not in ("f32", "bf16"):
            raise ValueError("precision: 'f32' (exact fp32 arithmetic) or 'bf16' (bf16 convolution operands, fp32 accumulation, "
                             "fp32 master weights and activations: the mixed-precision counterpart of Lightning's bf16-mixed)")
        if optimizer not in ParamGroup.OPTIMIZERS:
            raise ValueError(f"optimizer: one of {sorted(ParamGroup.OPTIMIZERS)} (AdamOptimizer / AdamWOptimizer / RMSOptimizer of the reference's config)")
        if gan_type not in ("original", "wgan"):
            raise ValueError("gan_type: 'original' (least-squares GAN, the default) or 'wgan'")
        self.precision = precision
        self.config = config or HiFiGANConfig()
        self.device = torch.device(device)
        self.opt = dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay)
        self.optimizer, self.alpha = optimizer, alpha
        self.gan_type, self.wgan_clip_value, self.generator_warmup_steps = gan_type, float(wgan_clip_value), int(generator_warmup_steps)
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
        # one stream per discriminator (the MRF branches of the generator reuse the first few)
        self.branches = Branches(self.device, 2 * (len(self.mpd) + len(self.msd)) + 2, enabled=parallel_streams)
        self._stream = torch.cuda.Stream(self.device) if self.device.type == "cuda" else None
        n_d = len(self.mpd) + len(self.msd)
        self._loss_buf = torch.zeros(len(self.LOSS_KEYS), device=self.device)      # d, g_adv, g_fm, g_mel, g_stft
        self._slots = torch.zeros(4, n_d, device=self.device)                      # per-discriminator partial d / g_adv / g_fm / d (generated call of the spectral-norm scale)
        self.use_graph = bool(use_graph) and self.device.type == "cuda"
        # weight gradients on sibling streams beside the input-gradient chains (ops.side_wgrad).  Off by default: correct in eager
        # mode (bitwise the same step), but capturing a step that forks a stream from an already forked stream ends in a
        # segmentation fault inside hipStreamEndCapture on ROCm 7.0, and graph mode is the faster one
        self.side_wgrad = bool(side_wgrad) and parallel_streams and not self.use_graph
        self._graphs, self._graph_warm = {}, {}
        self._graph_failed = None
        import os

        self.phase_times = {} if os.environ.get("EVMI_PHASE_TIMES") else None
        self.branch_times = None
        if self.phase_times is not None:
            self.branches.timing = []

    # -- state ------------------------------------------------------------------------------------------
    def d_layers(self):
        return [layer for d in [*self.mpd, *self.msd] for layer in d.layers()]

    def discriminators(self):
        return [*self.mpd, *self.msd]

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
        parameter names AND shapes (weight_g / weight_v, Conv2d((k, 1)) tensors for the period discriminators, and
        weight_orig + weight_u / weight_v buffers of the spectral-norm scale)."""
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
                {"evmi_flat_adamw": {"group": name, "step": grp.step, "exp_avg": grp.m.cpu(), "exp_avg_sq": grp.v.cpu(),
                                     "optimizer": self.optimizer, "alpha": self.alpha, **self.opt}}
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
                grp.set_step(int(st["step"]))
        return self

    def export_generator_checkpoint(self) -> dict:
        """What ``everyvoice export spec-to-wav`` writes (cli.py:381-386): the generator alone, loadable by
        ``everyvoice_amd.vocoder.load_hifigan_from_checkpoint`` for inference."""
        return {"state_dict": {"generator." + k: v.cpu() for k, v in self.g_params.state_dict().items()},
                "hyper_parameters": {"config": self.config.model_dump(mode="json")},
                "model_info": {"name": "HiFiGANGenerator", "version": self._VERSION}}

    def generate(self, mel_bct: torch.Tensor) -> torch.Tensor:
        """The generator alone with the current training weights (validation / synthesis during training): mel [B, n_mels, F]
        -> wav [B, 1, F * hop], no tape kept."""
        prev = ops.CONV_BACKEND["operands"]
        ops.CONV_BACKEND["operands"] = self.precision
        try:
            self._materialize(self.generator.layers())
            y = self.generator.forward(ag.Tape(), ag.Var(_to_cbt_kernel(mel_bct.to(self.device, torch.float32)), needs_grad=False))
        finally:
            ops.CONV_BACKEND["operands"] = prev
        return y.data.view(mel_bct.shape[0], 1, -1)  # [1, B, T] and [B, 1, T] are the same bytes

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
        d_ids = getattr(self, "_d_layer_ids", None)
        if d_ids is None:
            d_ids = self._d_layer_ids = {id(l) for l in self.d_layers()}
        if any(id(l) in d_ids for l in layers):
            for d in self.discriminators():  # the discriminators' weights may have changed: fragments made from older ones are stale
                ch = getattr(d, "_chain", None)
                if ch is not None:
                    ch.epoch += 1

    def _reducer(self, group: ParamGroup):
        """Bucketed all-reduce of one optimiser's flat gradient buffer, overlapped with backward (None on one GPU)."""
        if self.pg is None:
            return None
        return BucketReducer(group.grad, self.pg if self.pg is not True else None,
                             lambda t, sc: ops.elementwise(ops.EW_SCALE, t, out=t, p0=sc))

    @staticmethod
    def _bucket_range(group: ParamGroup, layers):
        """[lo, hi) of the flat buffers that the parameters of `layers` (a contiguous run in declaration order) occupy."""
        names = [n for layer in layers for n in layer.param_names()]
        lo = min(group.offset_of(n) for n in names)
        hi = max(group.offset_of(n) + (group.gradient(group._index[n]).numel() + 3) // 4 * 4 for n in names)
        return lo, hi

    def _bucket_hook(self, tape, group: ParamGroup, layers, reducer):
        """Record, BEFORE the forward of `layers`, the closure that backward runs AFTER all of their gradient closures: turn
        the effective-weight gradients into parameter gradients and hand the now-final slice of the flat gradient buffer to the
        reducer (on whatever stream this bucket's backward ran on)."""
        def done():
            ops.wgrad_join()  # this bucket's weight-gradient kernels (queued beside the chain) before their results are used
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
                reducer.launch(*self._bucket_range(group, layers))
        tape.record(done)

    # -- discriminators --------------------------------------------------------------------------------------
    def _scale_inputs(self, tape, audio: ag.Var):
        """Inputs of the scale discriminators: the waveform and its AvgPool1d(4, 2, 2) pyramid."""
        xs = [audio]
        for _ in self.msd[1:]:
            xs.append(ag.avgpool4s2(tape, xs[-1]))
        return xs

    def _discriminate(self, tape, audio: ag.Var, training=True):
        """All eight discriminators on one waveform, side by side on their streams -> (logits, feature maps)."""
        xs = self._scale_inputs(tape, audio)
        mpd_in = fan_out(tape, audio, len(self.mpd) + 1)  # + the first scale, which also reads the waveform itself
        ins = [*mpd_in[: len(self.mpd)], mpd_in[-1], *[fan_out(tape, x, 1)[0] for x in xs[1:]]]
        ds = self.discriminators()
        outs = parallel_section(tape, self.branches, [(lambda sub, i=i: ds[i].forward(sub, ins[i], training)) for i in range(len(ds))])
        return [o for o, _ in outs], [f for _, f in outs]

    def _logit_loss(self, logits, grad, target_real: bool, n: int, slot, sign=1.0):
        """One term of an adversarial loss on a logits tensor: accumulates the value into `slot`, writes d loss / d logits.
        original: mean((logit - target)^2), target 1 (real) or 0 (generated).  wgan: -/+ mean(logit)."""
        if self.gan_type == "wgan":
            sc = (-1.0 if target_real else 1.0) * sign / n
            ops.scalar_reduce(2, logits, None, slot, scale=sc, accumulate=True)
            ops.elementwise(ops.EW_FILL, logits, out=grad, p0=sc)
        else:
            tgt = 1.0 if target_real else 0.0
            ops.scalar_reduce(1, logits, None, slot, scale=1.0 / n, p=tgt, accumulate=True)
            ops.elementwise(ops.EW_SQ_GRAD, logits, out=grad, p0=1.0 / n, p1=tgt)

    def _sn_layers(self):
        return [layer for layer in self.d_layers() if isinstance(layer, SNConv)]

    def _fragments_beside(self, probe: torch.Tensor, generator_step: bool):
        """The weight-normed chains' fragments (280 MB of weights re-laid: ~0.6 ms when it runs alone) on a side stream, forked here;
        returns ``join()`` for the point where the chains are about to fork.  One fork from the step's stream, one join: no nesting."""
        if self.device.type != "cuda":
            self._prepare_chain_fragments(probe, generator_step, "wn")
            return lambda: None
        if getattr(self, "_frag_stream", None) is None:
            self._frag_stream = torch.cuda.Stream(self.device)
        main = torch.cuda.current_stream(self.device)
        self._frag_stream.wait_stream(main)
        with torch.cuda.stream(self._frag_stream):
            self._prepare_chain_fragments(probe, generator_step, "wn")
        return lambda: torch.cuda.current_stream(self.device).wait_stream(self._frag_stream)

    def _prepare_chain_fragments(self, probe: torch.Tensor, generator_step: bool, which: str = "all"):
        """Packed discriminator chains (train/disc_chain.py): the bf16 weight fragments of every matrix-core layer of every chain for the
        coming phase, BEFORE the chains fork onto their streams -- a few launches for all eight discriminators, and the generator
        step's real and generated calls share one set.  ``which``: "wn" (the weight-normed discriminators: their weights stand since
        the optimiser step, so this part runs early, beside other work: ``_fragments_beside``), "sn" (the spectral-norm scale: its
        per-call weights exist once ``_prepare_spectral_norm`` has run) or "all"."""
        jobs = []
        for d in self.discriminators():
            chain = _chain_for(d, getattr(d, "period", 1), probe)
            if chain is None:
                continue
            sn = any(isinstance(layer, SNConv) for layer in d.layers())
            if which != "all" and sn != (which == "sn"):
                continue
            # spectral norm: two calls (real, generated) with their own weights; in the generator step only the generated call runs backward
            jobs += chain.frag_jobs(((False, True) if generator_step else (True, True)) if sn else (True,))
        if jobs:
            from .disc_chain import launch_fragments

            launch_fragments(jobs, self.device)

    def _prepare_spectral_norm(self, n_calls: int):
        """Power iterations + effective weights of the spectral-norm scale's next forward calls, every layer on its own stream:
        ~8 small dependent launches per layer and call that would otherwise sit in front of each of its convolutions."""
        layers = self._sn_layers()
        self.branches.run([(lambda l=l: l.prepare(n_calls)) for l in layers])

    def _d_losses(self, o: ag.Var, kind: str, slot):
        o.grad = torch.empty_like(o.data)
        if kind == "pair":  # real items first, generated items second along the batch axis ((item, column) for the period view)
            n, h = o.data.numel() // 2, o.data.shape[1] // 2
            self._logit_loss(o.data[:, :h], o.grad[:, :h], True, n, slot)
            self._logit_loss(o.data[:, h:], o.grad[:, h:], False, n, slot)
        else:
            self._logit_loss(o.data, o.grad, kind == "real", o.data.numel(), slot)

    def _d_branch(self, tape, i: int, d, pair: ag.Var, reducer):
        """Discriminator step of one weight-normed discriminator: real and generated waveforms as ONE batch of 2B items (columns
        of the same GEMMs), its loss terms (into slot i); the recorded backward ends with its gradient bucket."""
        self._bucket_hook(tape, self.d_params, d.layers(), reducer)
        self._d_losses(d.forward(tape, pair)[0], "pair", self._slots[0, i : i + 1])

    def _d_branch_sn(self, tape, i: int, d, audio: torch.Tensor, kind: str):
        """One of the two forward calls of the spectral-norm scale (each sees its own power iteration): a chain of its own."""
        self._d_losses(d.forward(tape, ag.Var(audio, needs_grad=False), role=kind)[0], kind, self._slots[3 if kind == "fake" else 0, i : i + 1])

    def _g_forward(self, tape, d, x: ag.Var, role: str):
        return d.forward(tape, x, role=role)

    def _g_losses(self, i: int, real, fake):
        """Adversarial and feature-matching terms of one discriminator (into slots i) with their gradients."""
        if real is None:  # a packed chain over [real | generated] (train/disc_chain.py: grad_from): the generated half's terms
            dg, fm = fake
            h = fm.cfg.off_g
            dg.grad = ops.zeros(*dg.data.shape, device=dg.data.device)
            d_f, g_f = dg.data[:, h:], dg.grad[:, h:]
            self._logit_loss(d_f, g_f, True, d_f.numel(), self._slots[1, i : i + 1])
            fm.feature_matching(None, self._slots[2, i : i + 1])
            return
        (_, fr_list), (dg, fg_list) = real, fake
        n = dg.data.numel()
        dg.grad = torch.empty_like(dg.data)
        # original: mean((1 - D(y_hat))^2); wgan: -mean(D(y_hat))  (= the "real" form of the critic term)
        self._logit_loss(dg.data, dg.grad, True, n, self._slots[1, i : i + 1])
        if not isinstance(fg_list, list):  # a packed chain (train/disc_chain.py): one launch pair; gradients ride in its backward
            fg_list.feature_matching(fr_list, self._slots[2, i : i + 1])
            return
        for fr, fg in zip(fr_list, fg_list):
            n = fg.data.numel()
            ops.scalar_reduce(0, fg.data, fr.data, self._slots[2, i : i + 1], scale=2.0 / n, accumulate=True)
            fg.accumulate(ops.elementwise(ops.EW_SIGN_DIFF, fg.data, fr.data, p0=2.0 / n))

    # -- one GAN step -----------------------------------------------------------------------------------------
    def training_step(self, mel_bct: torch.Tensor, audio_bct: torch.Tensor, sync: bool = True):
        """mel [B, n_mels, T/hop], audio [B, 1, T] on the device.  Returns the scalar losses: python floats (ONE device-to-host
        read at the end of the step), or with ``sync=False`` a device tensor in LOSS_KEYS order (no host synchronisation)."""
        prev, prev_side = ops.CONV_BACKEND["operands"], ops.SIDE_WGRAD["on"]
        ops.CONV_BACKEND["operands"] = self.precision
        ops.SIDE_WGRAD["on"] = self.side_wgrad and self.device.type == "cuda"
        ops.side_reset()  # nothing an aborted step left collected reaches this one (ops.side_reset)
        try:
            if self.device.type != "cuda":
                buf = self._eager_step(mel_bct, audio_bct)
            else:
                # the step always runs on the trainer's own stream -- eagerly, while capturing and when replaying -- so that the
                # per-stream scratch buffers grown by the eager warm-up are the ones the captured graph uses
                caller = torch.cuda.current_stream(self.device)
                self._stream.wait_stream(caller)
                with torch.cuda.stream(self._stream):
                    if self.use_graph and self._graph_failed is None:
                        buf = self._graph_step(mel_bct, audio_bct)
                    else:
                        buf = self._eager_step(mel_bct, audio_bct)
                caller.wait_stream(self._stream)
        except BaseException:
            ops.side_reset(abort=True)
            raise
        finally:
            ops.CONV_BACKEND["operands"] = prev
            ops.SIDE_WGRAD["on"] = prev_side
        ops.side_check_drained()
        self.global_step += 1
        if not sync:
            return buf
        vals = buf.tolist()
        out = dict(zip(self.LOSS_KEYS, vals))
        out["g_total"] = out["g_adv"] + out["g_fm"] + out["g_mel"] + out["g_stft"]
        return out

    def _opt_kw(self):
        return dict(name=self.optimizer, alpha=self.alpha, **self.opt)

    def _eager_step(self, mel_bct, audio_bct):
        warm = self.global_step < self.generator_warmup_steps
        if self.pg is not None:
            # data parallel: ALWAYS the captured step's schedule -- a rank that steps eagerly (a shape not captured yet, a failed
            # capture) must issue the collective sequence of the ranks that replay.  EVMI_PHASE_TIMES is a single-process
            # measurement; under data parallelism it is ignored (the per-phase path's bucket hooks are another sequence)
            return self._eager_data_parallel_step(mel_bct, audio_bct, warm)
        marks = []

        def mark(name):  # EVMI_PHASE_TIMES=1: device time of every phase of the step (HIP events on the step's stream)
            if self.phase_times is not None:
                ev = torch.cuda.Event(enable_timing=True)
                ev.record(torch.cuda.current_stream(self.device))
                marks.append((name, ev))

        mark("start")
        ctx = self._phase_generator_forward(mel_bct, audio_bct, d_step=not warm)
        mark("generator forward")
        if not warm:
            d_red = self._reducer(self.d_params)
            self._phase_d_backward(ctx, d_red)
            if d_red is not None:
                d_red.launch(0, min(self.d_params.offset_of(n) for n in self.d_params.names()))  # alignment padding in front, if any
                d_red.finish()
            mark("discriminator step: forward + backward")
            self._phase_d_update(ctx)
            mark("discriminator update + weight norm")
        g_red = self._reducer(self.g_params)
        ctx["g_reducer"][0] = g_red
        self._phase_g_backward(ctx, adversarial=not warm)
        if g_red is not None:
            g_red.launch(0, min(self.g_params.offset_of(n) for n in self.g_params.names()))
            g_red.finish()
        mark("generator step: discriminators + losses + generator backward")
        self._phase_g_update(ctx)
        mark("generator update")

        if marks:
            marks[-1][1].synchronize()
            if self.branches.timing:
                self.branch_times = [[round(a.elapsed_time(b), 3) for a, b in sec] for sec in self.branches.timing]
                self.branches.timing.clear()
            self.phase_times = {b[0]: round(a[1].elapsed_time(b[1]), 3) for a, b in zip(marks, marks[1:])}
        return self._loss_buf

    # ---- the phases of a step (each one a stretch without communication: what a HIP graph captures) ----
    def _phase_generator_forward(self, mel_bct, audio_bct, d_step: bool = True):
        B = audio_bct.shape[0]
        y = audio_bct.to(torch.float32).reshape(1, B, -1)  # [B,1,T] and [1,B,T] are the same bytes
        if not y.is_contiguous():
            y = ops.copy(y.contiguous())
        mel = _to_cbt_kernel(mel_bct.to(torch.float32))  # [B, C, T] -> [C, B, T]: layout change only
        ops.fill_(self._loss_buf, 0.0)
        ops.fill_(self._slots, 0.0)
        g_layers, d_layers = self.generator.layers(), self.d_layers()
        self._materialize(g_layers)
        # Every step re-derives what it reads from the discriminators' parameters -- effective weights here, the weight-normed chains'
        # fragments beside the generator's forward -- although both stand since the previous step's update: a captured step is then
        # self-contained (parameters written between two replays -- a checkpoint loaded into a live trainer, a lock-step tool -- are
        # picked up by the next replay; round 5's host-side version gate baked "nothing to do" into the graph: ADVICE r05) at 0.04 ms.
        self._materialize(d_layers)
        # the discriminator step's weight fragments run UNDER the generator's forward
        frag_join = self._fragments_beside(y, generator_step=False) if d_step else None
        g_tape = ag.Tape()
        g_reducer = [None]  # filled in before the generator's backward (the reducer object is created per phase)
        y_hat = self.generator.forward(
            g_tape, ag.Var(mel, needs_grad=False),
            lambda layers: self._bucket_hook_late(g_tape, self.g_params, layers, g_reducer), branches=self.branches)
        return dict(B=B, y=y, y_hat=y_hat, g_tape=g_tape, g_reducer=g_reducer, d_layers=d_layers, frag_join=frag_join)

    def _bucket_hook_late(self, tape, group, layers, reducer_box):
        """As `_bucket_hook`, with the reducer looked up when backward runs (the generator's tape is recorded long before it)."""
        def done():
            ops.wgrad_join()
            batches = {}
            for layer in layers:
                b = getattr(layer, "_batch", None)
                if b is None:
                    layer.finish_grads()
                else:
                    batches.setdefault(id(b), (b, []))[1].append(layer)
            for b, ls in batches.values():
                b.finish(ls)
            if reducer_box[0] is not None:
                reducer_box[0].launch(*self._bucket_range(group, layers))
        tape.cut(tag=self._bucket_range(group, layers))  # (backward reaches `done` first, then this point)
        tape.record(done)

    def _phase_d_backward(self, ctx, reducer):
        """Discriminator step up to its gradients: the discriminators side by side, the spectral-norm scale's two calls too."""
        self._phase_d_prelude(ctx)
        self._phase_d_group(ctx, range(len(self.discriminators())), reducer)
        self._phase_d_epilogue(ctx)

    def d_bucket_groups(self):
        """The discriminators as gradient-bucket groups for a captured data-parallel step: [period discriminators] (164 MB of
        gradients: reduced UNDER the second group's forward + backward), [scale discriminators] (118 MB)."""
        n_p = len(self.mpd)
        return [list(range(n_p)), list(range(n_p, n_p + len(self.msd)))]

    def _phase_d_prelude(self, ctx):
        y, y_hat, B = ctx["y"], ctx["y_hat"], ctx["B"]
        self.d_params.zero_grad()
        for layer in ctx["d_layers"]:
            layer.frozen = False
        self._prepare_spectral_norm(2)  # real call, then generated call
        self._prepare_chain_fragments(y, generator_step=False, which="sn")
        join = ctx.pop("frag_join", None)
        if join is not None:
            join()  # (the weight-normed chains' fragments, started beside the generator's forward)
        else:
            self._prepare_chain_fragments(y, generator_step=False, which="wn")
        T = y.shape[-1]
        pair_t = torch.empty(1, 2 * B, T, device=self.device, dtype=torch.float32)
        ops.copy(y, out=pair_t[:, :B])
        ops.copy(y_hat.data, out=pair_t[:, B:])  # y_hat.detach()
        pair = ag.Var(pair_t, needs_grad=False)
        pairs = [pair]
        pool_tape = ag.Tape()  # (the pyramid of a gradient-free input: nothing to run backward)
        for _ in self.msd[1:]:
            pairs.append(ag.avgpool4s2(pool_tape, pairs[-1]))
        ctx["d_ins"] = [pair] * len(self.mpd) + pairs

    def _phase_d_group(self, ctx, idxs, reducer):
        """Forward + backward of the discriminators `idxs`, side by side; every one's gradient bucket is finished (and, eagerly,
        its all-reduce launched) as its stream leaves it."""
        y, y_hat, ins = ctx["y"], ctx["y_hat"], ctx["d_ins"]
        d_tape = ag.Tape()
        ds = self.discriminators()
        idxs = set(idxs)
        items = []  # (stream index, fn): every chain keeps the stream it has when all discriminators run together -- the
        slot = 0    # per-stream workspaces the eager warm-up steps grew are the ones a captured stretch of a subset finds
        for i, d in enumerate(ds):
            sn = any(isinstance(layer, SNConv) for layer in d.layers())
            if i in idxs:
                if sn:
                    # its bucket closes on the main stream once both chains' backward has been joined
                    self._bucket_hook(d_tape, self.d_params, d.layers(), reducer)
                    items.append((slot, lambda sub, i=i, d=d: self._d_branch_sn(sub, i, d, y, "real")))
                    items.append((slot + 1, lambda sub, i=i, d=d: self._d_branch_sn(sub, i, d, y_hat.data, "fake")))
                else:
                    items.append((slot, lambda sub, i=i, d=d: self._d_branch(sub, i, d, ins[i], reducer)))
            slot += 2 if sn else 1
        subs = [ag.Tape() for _ in items]
        self.branches.run_indexed([(j, (lambda k=k, fn=fn: fn(subs[k]))) for k, (j, fn) in enumerate(items)])
        d_tape.record(lambda: self.branches.run_indexed([(j, subs[k].backward) for k, (j, _) in enumerate(items)]))
        d_tape.backward()

    def _phase_d_epilogue(self, ctx):
        for layer in self._sn_layers():
            layer.release()  # every stream has been joined: the prepared spectral-norm tensors may go
        ops.scalar_reduce(2, self._slots[0], None, self._loss_buf[0:1])
        ops.scalar_reduce(2, self._slots[3], None, self._loss_buf[0:1], accumulate=True)
        ctx.pop("d_ins", None)
        if self.keep_grads:
            self.last_grads["d"] = {k: v.clone() for k, v in self.d_params.gradients().items()}

    def _phase_d_update(self, ctx):
        self.d_params.optimizer_step(clip=self.wgan_clip_value if self.gan_type == "wgan" else 0.0, **self._opt_kw())
        self._materialize(ctx["d_layers"])  # the generator step sees the updated discriminators

    def _recon_grad(self, y, y_hat_t, B):
        """Reconstruction losses (45 x mel-L1 and / or the multi-resolution STFT loss) and their gradient wrt y_hat [1, B, T]."""
        total = None
        if "mel" in self.reconstruction_loss.split("+"):
            total = self.mel_loss.loss_and_grad(y.view(B, -1), y_hat_t.view(B, -1), 45.0, self._loss_buf[3:4]).view(1, B, -1)
        if self.stft_loss is not None:
            d_stft = self.stft_loss.loss_and_grad(y.view(B, -1), y_hat_t.view(B, -1), self.stft_loss_weight, self._loss_buf[4:5]).view(1, B, -1)
            total = d_stft if total is None else ops.axpby(1.0, total, 1.0, d_stft, out=total)
        return total

    def _phase_g_backward(self, ctx, adversarial=True):
        y, y_hat, B = ctx["y"], ctx["y_hat"], ctx["B"]
        self.g_params.zero_grad()
        for layer in ctx["d_layers"]:
            layer.frozen = True  # gradients flow through the discriminators to y_hat only
        y_hat_in = ag.Var(y_hat.data)  # boundary between the discriminator tape and the generator tape
        recon = [None]
        if adversarial:
            frag_join = self._fragments_beside(y, generator_step=True)  # (under the spectral-norm power iterations)
            self._prepare_spectral_norm(2)  # real call, then generated call
            self._prepare_chain_fragments(y, generator_step=True, which="sn")
            frag_join()
            gd_tape = ag.Tape()
            real = ag.Var(y, needs_grad=False)
            ds = self.discriminators()
            nd = len(ds)
            n_p = len(self.mpd)
            is_sn = [any(isinstance(l, SNConv) for l in d.layers()) for d in ds]
            # Packed chains (precision "bf16"): every weight-normed discriminator sees [real | generated] as ONE batch -- forward once
            # over 2B waveforms (half the launches of two B-waveform passes, each better filled), backward over the generated half,
            # feature matching between the two halves of the same tensors.  The spectral-norm scale keeps its two calls (each call
            # runs its own power iteration: two different effective weights).
            pair_mode = all(sn or _chain_for(d, getattr(d, "period", 1), y) is not None for d, sn in zip(ds, is_sn)) and not all(is_sn)
            res_r, res_f = [None] * nd, [None] * nd
            pair_in = None
            if pair_mode:
                T = y.shape[-1]
                pair_t = torch.empty(1, 2 * B, T, device=self.device, dtype=torch.float32)
                ops.copy(y, out=pair_t[:, :B])
                ops.copy(y_hat.data, out=pair_t[:, B:])
                pair_in = ag.Var(pair_t)
                xs_p = self._scale_inputs(gd_tape, pair_in)
                sn_pooled = any(is_sn[i] and i > n_p for i in range(nd))  # (a spectral-norm scale behind the pooling: not in the upstream model)
                xs_r = self._scale_inputs(gd_tape, real) if sn_pooled else [real]
                xs_f = self._scale_inputs(gd_tape, y_hat_in) if sn_pooled else [y_hat_in]
                n_wave = sum(1 for i in range(nd) if not is_sn[i] and i <= n_p)  # chains reading the waveform itself: the periods (+ scale 0 when weight-normed)
                p_leaves = fan_out(gd_tape, pair_in, n_wave)
                f_leaves = fan_out(gd_tape, y_hat_in, sum(1 for i in range(nd) if is_sn[i] and i <= n_p))
                ins_p, ins_r, ins_f = [None] * nd, [None] * nd, [None] * nd
                pi = fi = 0
                for i in range(nd):
                    scale = max(0, i - n_p)
                    if is_sn[i]:
                        ins_r[i] = xs_r[scale]
                        if scale == 0:
                            ins_f[i] = f_leaves[fi]
                            fi += 1
                        else:
                            ins_f[i] = fan_out(gd_tape, xs_f[scale], 1)[0]
                    elif scale == 0:
                        ins_p[i] = p_leaves[pi]
                        pi += 1
                    else:
                        ins_p[i] = fan_out(gd_tape, xs_p[scale], 1)[0]

                def fwd_pair(i):
                    return lambda sub: res_f.__setitem__(i, ds[i].forward(sub, ins_p[i], role="g_both", grad_from=B))

                def fwd_real(i):
                    return lambda sub: res_r.__setitem__(i, self._g_forward(sub, ds[i], ins_r[i], "g_real"))

                def fwd_fake(i):
                    return lambda sub: res_f.__setitem__(i, self._g_forward(sub, ds[i], ins_f[i], "g_fake"))

                sn_idx = [i for i in range(nd) if is_sn[i]]
                order_a = [fwd_fake(i) if is_sn[i] else fwd_pair(i) for i in range(nd)]
                order_b = [fwd_real(i) for i in sn_idx]
                # host order decides which prepared spectral-norm call a chain receives: issue the real chains of those scales first
                self._ordered_section(gd_tape, order_a, order_b + [lambda sub: recon.__setitem__(0, self._recon_grad(y, y_hat.data, B))],
                                      first=[nd + j for j in range(len(sn_idx))])
            else:
                xs_r = self._scale_inputs(gd_tape, real)
                xs_f = self._scale_inputs(gd_tape, y_hat_in)
                fake_leaves = fan_out(gd_tape, y_hat_in, n_p + 1)
                ins_r = [real] * (n_p + 1) + xs_r[1:]
                ins_f = [*fake_leaves[:n_p], fake_leaves[-1], *[fan_out(gd_tape, x, 1)[0] for x in xs_f[1:]]]
                # generated-waveform chains on streams 0 .. nd-1 (their backward too), real-waveform chains (forward only) and the
                # reconstruction loss on the streams after them; the spectral-norm scale hands out its real call first

                def fwd_real(i):
                    return lambda sub: res_r.__setitem__(i, self._g_forward(sub, ds[i], ins_r[i], "g_real"))

                def fwd_fake(i):
                    return lambda sub: res_f.__setitem__(i, self._g_forward(sub, ds[i], ins_f[i], "g_fake"))

                sn_first = [i for i in range(nd) if is_sn[i]]
                order_fake = [fwd_fake(i) for i in range(nd)]
                order_real = [fwd_real(i) for i in range(nd)]
                # host order decides which prepared spectral-norm call a chain receives: issue the real chains of those scales first
                self._ordered_section(gd_tape, order_fake, order_real + [lambda sub: recon.__setitem__(0, self._recon_grad(y, y_hat.data, B))], first=[nd + i for i in sn_first])
            self.branches.run([(lambda i=i: self._g_losses(i, res_r[i], res_f[i])) for i in range(nd)])
            gd_tape.backward()
            if pair_in is not None and pair_in.grad is not None:  # the generated half of the batched chains' waveform gradient
                y_hat_in.accumulate(ops.copy(pair_in.grad[:, B:].contiguous()))
            ops.scalar_reduce(2, self._slots[1], None, self._loss_buf[1:2])
            ops.scalar_reduce(2, self._slots[2], None, self._loss_buf[2:3])
            for layer in ctx["d_layers"]:
                if isinstance(layer, SNConv):
                    layer._calls.clear()  # frozen: no parameter gradients from this pass
                    layer.release()
            total = recon[0]
        else:
            total = self._recon_grad(y, y_hat.data, B)
        if y_hat_in.grad is not None:
            total = ops.axpby(1.0, total, 1.0, y_hat_in.grad, out=total)
        y_hat.grad = total
        if ctx.get("g_segmented"):  # captured data-parallel step: the generator's backward is run bucket by bucket by the caller
            ctx["g_segments"] = ctx["g_tape"].backward_segments(ctx.get("g_stop"))
            return
        ctx["g_tape"].backward()  # buckets: conv_post, the upsampling stages, conv_pre
        if self.keep_grads:
            self.last_grads["g"] = {k: v.clone() for k, v in self.g_params.gradients().items()}
            self.last_grads["y_hat"] = y_hat.data.clone()

    def _ordered_section(self, tape, fns_a, fns_b, first=()):
        """A parallel section over fns_a + fns_b (branch j on stream j) whose HOST issue order starts with the indices in `first`."""
        fns = list(fns_a) + list(fns_b)
        subs = [ag.Tape() for _ in fns]
        order = list(first) + [j for j in range(len(fns)) if j not in first]
        self.branches.run_indexed([(j, (lambda j=j: fns[j](subs[j]))) for j in order])
        tape.record(lambda: self.branches.run_indexed([(j, subs[j].backward) for j in range(len(fns))]))

    def _phase_g_update(self, ctx):
        self.g_params.optimizer_step(**self._opt_kw())

    # ---- HIP-graph execution of the step ----------------------------------------------------------------------------
    GRAPH_WARMUP_STEPS = 2

    def _graph_step(self, mel_bct, audio_bct):
        """Fixed-shape steps (vocoder segments are: batch x vocoder_segment_size) run as HIP graph replays: the first steps at a
        shape run eagerly (workspaces grow, kernel attributes are set), the next one is captured -- on one GPU as ONE graph,
        under data parallelism as three (up to the discriminators' gradients | their update and the generator's backward | the
        generator's update) with the two gradient all-reduces issued between them -- and every later step replays.
        Any failure while capturing falls back to eager execution for good (``_graph_failed`` holds the reason)."""
        warm = self.global_step < self.generator_warmup_steps
        key = (tuple(mel_bct.shape), tuple(audio_bct.shape), self.precision, warm, self.keep_grads)
        entry = self._graphs.get(key)
        if entry is None:
            n = self._graph_warm.get(key, 0)
            if n < self.GRAPH_WARMUP_STEPS or self.keep_grads:
                self._graph_warm[key] = n + 1
                return self._eager_step(mel_bct, audio_bct)
            steps = (self.g_params.step, self.d_params.step)
            failure = None
            try:
                if getattr(self, "_force_capture_failure", False):  # (tests/test_gpu_ddp.py: one rank eager beside one that replays)
                    raise RuntimeError("capture failure forced by a test")
                entry = self._capture(key, mel_bct, audio_bct, warm)
            except Exception as e:  # noqa: BLE001 -- whatever the runtime objected to: the eager path is always available
                failure = f"{type(e).__name__}: {e}"
            # data parallel: the eager step runs the captured step's schedule (`_eager_data_parallel_step`: same stretches, same
            # buckets, same collective sequence), so a rank whose capture failed goes on eagerly beside ranks that replay --
            # no agreement between the ranks is needed (round 4 all-reduced a flag here, a collective only capturing ranks issued)
            if failure is not None:
                self._graph_failed = failure
                torch.cuda.synchronize(self.device)
                ops.side_reset()  # the aborted capture's collected weight-gradient launches and events must not reach the eager step
                # nothing of the aborted capture has run, but its host-side bookkeeping has: the spectral-norm layers hold
                # prepared (weight, sigma, u, v) tuples that live in the dead graph's pool and were never computed, and the
                # optimisers' host counters were bumped.  Put both back before the eager step.
                for layer in self._sn_layers():
                    layer._ready.clear()
                    layer._held.clear()
                    layer._calls.clear()
                self.g_params._step, self.d_params._step = steps
                return self._eager_step(mel_bct, audio_bct)
            self._graphs[key] = entry
        ops.copy(mel_bct.to(torch.float32).contiguous(), out=entry["mel"])
        ops.copy(audio_bct.to(torch.float32).contiguous(), out=entry["audio"])
        for g, after in zip(entry["graphs"], entry["after"]):
            g.replay()
            if after is not None:
                after()  # the gradient exchange at this bucket boundary: RCCL calls sit between the captured stretches
        # the host-side step counters follow the device-side ones the graph increments
        if not warm:
            self.d_params._step += 1
        self.g_params._step += 1
        return self._loss_buf

    def _allreduce_whole(self, group: ParamGroup):
        if self.pg is not None:
            allreduce_mean_(group.grad, self.pg if self.pg is not True else None, lambda t, sc: ops.elementwise(ops.EW_SCALE, t, out=t, p0=sc))

    # smallest generator-side bucket (floats) worth a stretch boundary of its own: smaller ones ride with the next
    MIN_G_BUCKET = 1 << 20

    def _capture(self, key, mel_bct, audio_bct, warm):
        """One GPU: the whole step is ONE graph.  Data parallel: the step is captured in stretches that end where a gradient
        bucket becomes final -- [generator forward + period discriminators] [scale discriminators] [discriminator update +
        generator-step discriminator pass + generator backward down to the first large bucket] ... [generator update] -- and
        between two stretches that bucket's all-reduce is launched on a side stream (RCCL calls are not captured), so it runs
        UNDER the next stretch; only an optimiser waits for its buckets (the reference: DDP buckets firing inside
        backward, base_cli/helpers.py:252-270)."""
        mel_s = mel_bct.to(torch.float32).contiguous().clone()
        audio_s = audio_bct.to(torch.float32).contiguous().clone()
        steps = (self.g_params.step, self.d_params.step)
        torch.cuda.synchronize(self.device)
        pool = torch.cuda.graph_pool_handle()
        graphs, after = [], []
        ctx = {}

        def cap(fn, then=None):
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, pool=pool, stream=self._stream, capture_error_mode="thread_local"):
                fn()
            graphs.append(g)
            after.append(then)

        def whole():
            ctx.update(self._phase_generator_forward(mel_s, audio_s, d_step=not warm))
            if not warm:
                self._phase_d_backward(ctx, None)
                self._phase_d_update(ctx)
            self._phase_g_backward(ctx, adversarial=not warm)
            self._phase_g_update(ctx)

        try:
            if self.pg is None:
                cap(whole)
            else:
                self._capture_data_parallel(cap, ctx, mel_s, audio_s, warm)
        finally:
            ctx.clear()
            # capturing does not execute: the host-side counters the phases bumped are put back (replay bumps them again)
            self.g_params._step, self.d_params._step = steps
        return dict(graphs=graphs, after=after, mel=mel_s, audio=audio_s)

    def _capture_data_parallel(self, cap_raw, ctx, mel_s, audio_s, warm):
        """The data-parallel schedule with every stretch captured and every exchange stored for the replay loop."""
        thens = []

        def run(fn):
            cap_raw(fn, lambda i=len(thens): thens[i] and thens[i]())
            thens.append(None)

        def emit(then):
            thens[-1] = then

        self._data_parallel_schedule(run, emit, ctx, mel_s, audio_s, warm)

    def _eager_data_parallel_step(self, mel_bct, audio_bct, warm):
        """The data-parallel schedule executed eagerly: the SAME stretches and the SAME exchanges between them as the captured step
        (`_capture_data_parallel`), so a rank that runs eagerly -- a shape not captured yet, a capture that failed on this rank only --
        issues exactly the collective sequence of a rank that replays (the reference: every rank runs the same DDP bucket sequence,
        everyvoice/base_cli/interfaces.py:90-97 -> base_cli/helpers.py:252-270).  No agreement between ranks is needed any more."""
        ctx = {}
        try:
            self._data_parallel_schedule(lambda fn: fn(), lambda then: then(), ctx, mel_bct, audio_bct, warm)
        finally:
            ctx.clear()
        return self._loss_buf

    def _data_parallel_schedule(self, run, emit, ctx, mel_s, audio_s, warm):
        """``run(fn)``: execute (or capture) one stretch; ``emit(then)``: the exchange that follows it (called at once, or stored)."""
        if getattr(self, "_dp_reducers", None) is None:
            scale = lambda t, sc: ops.elementwise(ops.EW_SCALE, t, out=t, p0=sc)  # noqa: E731
            pg = self.pg if self.pg is not True else None
            self._dp_reducers = (BucketReducer(self.d_params.grad, pg, scale), BucketReducer(self.g_params.grad, pg, scale))  # (their side streams live across steps)
        d_red, g_red = self._dp_reducers

        def first():
            ctx.update(self._phase_generator_forward(mel_s, audio_s, d_step=not warm))
            if not warm:
                self._phase_d_prelude(ctx)

        if warm:
            run(first)
        else:
            groups = self.d_bucket_groups()
            ds = self.discriminators()
            d_front = min(self.d_params.offset_of(n) for n in self.d_params.names())  # alignment padding in front, if any
            for gi, idxs in enumerate(groups):
                lo, hi = self._bucket_range(self.d_params, [layer for i in idxs for layer in ds[i].layers()])
                last = gi == len(groups) - 1

                def stretch(gi=gi, idxs=idxs, last=last):
                    if gi == 0:
                        first()
                    self._phase_d_group(ctx, idxs, None)
                    if last:
                        self._phase_d_epilogue(ctx)

                run(stretch)
                if last:  # the discriminators' optimiser comes next: wait for every bucket, scale by 1 / world
                    emit(lambda lo=lo, hi=hi: (d_red.launch(lo, hi), d_red.launch(0, d_front), d_red.finish()))
                else:  # runs on the side stream UNDER the next group's forward + backward
                    emit(lambda lo=lo, hi=hi: d_red.launch(lo, hi))
        # generator step: the first stretch runs the discriminators' update, the generator-step discriminator pass, the losses
        # and the generator's backward down to the first bucket boundary worth a cut; then one stretch per further bucket
        ctx["g_segmented"] = True
        pending = [0]

        def stop(tag):  # small buckets (conv_post, the narrow late stages) ride with the next one
            pending[0] += tag[1] - tag[0]
            if pending[0] >= self.MIN_G_BUCKET:
                pending[0] = 0
                return True
            return False

        ctx["g_stop"] = stop
        state = {"cut": None, "done": False}

        def advance():
            try:
                state["cut"] = next(ctx["g_segments"])[0]  # gradients from here to the previous cut are final
            except StopIteration:
                state["cut"], state["done"] = 0, True

        def g_first():
            if not warm:
                self._phase_d_update(ctx)
            self._phase_g_backward(ctx, adversarial=not warm)
            advance()

        hi = self.g_params.grad.numel()
        run(g_first)
        while True:
            lo, done = state["cut"], state["done"]
            emit((lambda lo=lo, hi=hi: (g_red.launch(lo, hi), g_red.finish())) if done else (lambda lo=lo, hi=hi: g_red.launch(lo, hi)))
            hi = lo
            if done:
                break
            run(advance)

        def g_last():
            if self.keep_grads:  # (tests; never while capturing: keep_grads keeps a step eager)
                self.last_grads["g"] = {k: v.clone() for k, v in self.g_params.gradients().items()}
                self.last_grads["y_hat"] = ctx["y_hat"].data.clone()
            self._phase_g_update(ctx)

        run(g_last)


def _to_cbt_kernel(x_bct: torch.Tensor) -> torch.Tensor:
    """[B, C, T] -> [C, B, T] with library copy kernels (one strided row-block copy per item: layout change only)."""
    B, C, T = x_bct.shape
    out = torch.empty(C, B, T, device=x_bct.device, dtype=torch.float32)
    if x_bct.is_cuda:
        _lib.check(_lib.load().evmi_transpose_bct_cbt_f32(x_bct.contiguous().data_ptr(), out.data_ptr(), B, C, T, _lib.current_stream_ptr(x_bct.device)),
                   "evmi_transpose_bct_cbt_f32")
        return out
    return x_bct.permute(1, 0, 2).contiguous()
