"""One full GAN training step on the GPU vs the torch-CPU oracle (autograd + torch.optim.AdamW).

fp32 on both sides; differences are summation order only.  Tolerances: losses rtol 2e-4; gradients are
compared per tensor with max|diff| <= 2e-3 * max|grad| + 1e-7 (relative to the tensor's scale)."""

import os

import pytest
import torch
import torch.nn.functional as F

from oracle import mel_ref
from oracle.hifigan_ref import (GeneratorRef, MultiPeriodDiscriminatorRef, MultiScaleDiscriminatorRef,
                                discriminator_loss_ref, feature_loss_ref, generator_loss_ref)

pytestmark = pytest.mark.gpu


def _params_close(name, got, want, grad, lr=2e-4, solid_frac=1e-2):
    """Updated parameters after one AdamW step.  The first step moves every element by lr * g / (|g| + eps), i.e.
    by +-lr whatever |g| is, so an element whose gradient sits at rounding-noise level can legitimately move the
    other way: compare where the oracle's gradient is clearly above noise, bound the rest by 2 * lr."""
    diff = (got.reshape(want.shape) - want).abs()
    assert float(diff.max()) <= 2.2 * lr, name
    if grad is not None:
        solid = grad.abs() > solid_frac * grad.abs().max()
        if solid.any():
            assert float(diff[solid].max()) <= 5e-6, name


def _grad_close(name, got, want, rel=1e-2):
    """Gradients against the oracle's.  Typical deviation is a few 1e-4 of the tensor's largest entry (fp32 summation
    order); 1e-2 leaves room for one pre-activation within rounding distance of the leaky-ReLU kink taking the other
    slope (0.1 vs 1) in one of the two implementations, which moves a layer's gradients by a few 1e-3 -- observed
    when only the summation order of the logit convolution changed.  A wrong tap, stride or scale is off by >> 1e-2."""
    scale = float(want.abs().max())
    err = float((got.reshape(want.shape) - want).abs().max())
    if os.environ.get("EVMI_TEST_REPORT"):
        print(f"GRADERR {err / (scale + 1e-30):.3e} {name} err {err:.3e} scale {scale:.3e}")
        return
    assert err <= rel * scale + 1e-7, f"{name}: err {err:.3e} vs scale {scale:.3e}"


@pytest.fixture(scope="module")
def oracle_models():
    torch.manual_seed(1234)
    torch.set_num_threads(8)
    g = GeneratorRef().train()
    # livelier than the N(0, 0.01) init so every layer's gradient is well above rounding noise
    with torch.no_grad():
        for n, p in g.named_parameters():
            if n.endswith("weight_v"):
                p.mul_(8.0)
            if n.endswith("weight_g"):
                p.mul_(8.0)
    return g, MultiPeriodDiscriminatorRef().train(), MultiScaleDiscriminatorRef().train()


def test_discriminators_forward_match(cuda_device, oracle_models):
    from everyvoice_amd.train import autograd as ag
    from everyvoice_amd.train.hifigan import HiFiGANTrainer

    _, mpd0, msd0 = oracle_models
    mpd, msd = MultiPeriodDiscriminatorRef().eval(), MultiScaleDiscriminatorRef().eval()  # eval: no power iteration
    mpd.load_state_dict(mpd0.state_dict())
    msd.load_state_dict(msd0.state_dict())
    tr = HiFiGANTrainer(device=cuda_device)
    tr.load_reference_state(None, mpd.state_dict(), msd.state_dict())
    tr._materialize(tr.d_layers())
    g = torch.Generator().manual_seed(9)
    y = 0.3 * torch.tanh(torch.randn(2, 1, 2048, generator=g))
    with torch.no_grad():
        r_p, _, f_p, _ = mpd(y, y)
        r_s, _, f_s, _ = msd(y, y)
    logits, fmaps = tr._discriminate(ag.Tape(), ag.Var(y.reshape(1, 2, -1).to(cuda_device), needs_grad=False), training=False)
    want_logits, want_fmaps = r_p + r_s, f_p + f_s
    assert len(logits) == 8 and [len(f) for f in fmaps] == [6] * 5 + [8] * 3
    for i, (lg, wl) in enumerate(zip(logits, want_logits)):
        got = lg.data.cpu()
        if i < 5:  # MPD: ours is [1, B*p, H]; torch flattens [B, 1, H, p]
            p = tr.mpd[i].period
            got = got.view(2, p, -1).permute(0, 2, 1).reshape(2, -1)
        else:
            got = got.view(2, -1)
        _grad_close(f"logits[{i}]", got, wl, rel=5e-5)
    for i in range(8):
        for fm, wf in zip(fmaps[i], want_fmaps[i]):
            got = fm.data.cpu()
            if i < 5:
                p = tr.mpd[i].period
                C = got.shape[0]
                got = got.view(C, 2, p, -1).permute(1, 0, 3, 2)  # -> [B, C, H, p]
            else:
                got = got.permute(1, 0, 2)
            _grad_close(f"fmap[{i}]", got.contiguous(), wf, rel=2e-5)


def _bf(t):
    return t.to(torch.bfloat16).to(torch.float32)


class _RoundedConv(torch.autograd.Function):
    """The arithmetic of precision="bf16" restated for the oracle: forward and input gradient use bf16-rounded operands
    (round to nearest even) with fp32 accumulation; so does the weight gradient where the product's bf16 kernel takes the shape."""

    @staticmethod
    def forward(ctx, x, w, b, stride, padding, dilation, groups, nd):
        ctx.save_for_backward(x, w)
        ctx.cfg = (stride, padding, dilation, groups, nd, b is not None)
        conv = F.conv1d if nd == 1 else F.conv2d
        return conv(_bf(x), _bf(w), b, stride, padding, dilation, groups)

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        stride, padding, dilation, groups, nd, has_b = ctx.cfg
        gi = torch.nn.grad.conv1d_input if nd == 1 else torch.nn.grad.conv2d_input
        gw = torch.nn.grad.conv1d_weight if nd == 1 else torch.nn.grad.conv2d_weight
        dx = gi(x.shape, _bf(w), _bf(dy), stride, padding, dilation, groups)
        one = lambda v: v[0] if isinstance(v, (tuple, list)) else v  # noqa: E731
        items = x.shape[0] * (x.shape[3] if nd == 2 else 1)  # Conv2d((k, 1)) on [B, C, H, p] = Conv1d on B * p items of length H
        if _wgrad_rounded(items, x.shape[1], x.shape[2], w.shape[0], dy.shape[2], w.shape[2], one(stride), one(padding), one(dilation), groups):
            dw = gw(_bf(x), w.shape, _bf(dy), stride, padding, dilation, groups)
        else:
            dw = gw(x, w.shape, dy, stride, padding, dilation, groups)
        db = dy.sum(dim=[0] + list(range(2, dy.dim()))) if has_b else None
        return dx, dw, db, None, None, None, None, None


def _wgrad_rounded(items, cin, t_in, cout, t_out, k, stride, pad, dil, groups):
    from everyvoice_amd.train import ops

    return ops.wgrad_takes_bf16(items, cin, t_in, cout, t_out, k, stride, pad, dil, groups)  # the product's own rule


class _RoundedConvT(torch.autograd.Function):
    """conv_transpose1d in precision="bf16": the input-gradient kernel of the strided convolution with the same weights."""

    @staticmethod
    def forward(ctx, x, w, b, stride, padding):
        ctx.save_for_backward(x, w)
        ctx.cfg = (stride, padding, b is not None)
        return torch.nn.functional.conv_transpose1d(_bf(x), _bf(w), b, stride, padding)

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        stride, padding, has_b = ctx.cfg
        dx = _ORIG_CONV1D(_bf(dy), _bf(w), None, stride, padding)
        # the weight gradient of the strided convolution dy -> x with the same weight tensor (input dy, output gradient x)
        rd = _wgrad_rounded(dy.shape[0], dy.shape[1], dy.shape[2], x.shape[1], x.shape[2], w.shape[2], stride, padding, 1, 1)
        dw = torch.nn.grad.conv1d_weight(_bf(dy) if rd else dy, w.shape, _bf(x) if rd else x, stride, padding)
        return dx, dw, dy.sum(dim=(0, 2)) if has_b else None, None, None


_ORIG_CONV1D = F.conv1d


class _bf16_operand_oracle:
    """Context: torch's conv1d / conv2d run with rounded operands wherever the product takes its bf16 kernels -- asked from the
    product (ops.fwd_takes_bf16 / ops.wgrad_takes_bf16), so a dispatch change cannot silently change what "match" means."""

    def __enter__(self):
        self.c1, self.c2 = F.conv1d, F.conv2d
        c1, c2 = self.c1, self.c2

        def takes_bf16(x, w, stride, padding, dilation, groups):
            """The product's own forward dispatch rule (ops.fwd_takes_bf16 -> evmi_conv1d_cbt_bf16_rounds), not a restatement:
            Conv2d((k, 1)) on [B, C, H, p] is the Conv1d on B * p items of length H that the product runs."""
            from everyvoice_amd.train import ops

            one = lambda v: v[0] if isinstance(v, (tuple, list)) else v  # noqa: E731
            s, p, d = one(stride), one(padding), one(dilation)
            items = x.shape[0] * (x.shape[3] if x.dim() == 4 else 1)
            t_in, k = x.shape[2], w.shape[2]
            t_out = (t_in + 2 * p - d * (k - 1) - 1) // s + 1
            return ops.fwd_takes_bf16(items, x.shape[1], t_in, w.shape[0], t_out, k, s, p, d, groups)

        def conv1d(x, w, b=None, stride=1, padding=0, dilation=1, groups=1):
            rounded = takes_bf16(x, w, stride, padding, dilation, groups)
            if torch.is_grad_enabled() and rounded:
                return _RoundedConv.apply(x, w, b, stride, padding, dilation, groups, 1)
            if rounded:
                return c1(_bf(x), _bf(w), b, stride, padding, dilation, groups)
            return c1(x, w, b, stride, padding, dilation, groups)

        def conv2d(x, w, b=None, stride=1, padding=0, dilation=1, groups=1):
            rounded = takes_bf16(x, w, stride, padding, dilation, groups)
            if torch.is_grad_enabled() and rounded:
                return _RoundedConv.apply(x, w, b, stride, padding, dilation, groups, 2)
            if rounded:
                return c2(_bf(x), _bf(w), b, stride, padding, dilation, groups)
            return c2(x, w, b, stride, padding, dilation, groups)

        self.ct = F.conv_transpose1d
        ct = self.ct

        def conv_transpose1d(x, w, b=None, stride=1, padding=0, output_padding=0, groups=1, dilation=1):
            s1 = stride[0] if isinstance(stride, (tuple, list)) else stride
            p1 = padding[0] if isinstance(padding, (tuple, list)) else padding
            op = output_padding[0] if isinstance(output_padding, (tuple, list)) else output_padding
            if torch.is_grad_enabled() and groups == 1 and op == 0 and w.shape[0] >= 8 and w.shape[1] > 4:
                return _RoundedConvT.apply(x, w, b, s1, p1)
            return ct(x, w, b, stride, padding, output_padding, groups, dilation)

        F.conv1d, F.conv2d, F.conv_transpose1d = conv1d, conv2d, conv_transpose1d
        return self

    def __exit__(self, *exc):
        F.conv1d, F.conv2d, F.conv_transpose1d = self.c1, self.c2, self.ct


# The fixture's generator gain of 8 per layer drives conv_post to ~1e14: every output sample sits at tanh = +-1, the generator's
# gradients are exactly zero and y_hat only records signs (one of which flips with the fp32 summation order wherever the
# pre-activation cancels to 1e-6 of its scale).  The fp32 whole-step tests therefore run the generator at gain 2 (F32_G_GAIN x 8):
# |conv_post| ~ 0.4 (max 1.8), every generator gradient tensor well above rounding noise.
F32_G_GAIN = 0.25


def test_full_gan_step_matches_oracle(cuda_device, oracle_models):
    _full_gan_step(cuda_device, oracle_models, "f32", g_gain=F32_G_GAIN)


def test_full_gan_step_bf16_operands_match_rounded_oracle(cuda_device, oracle_models):
    """precision="bf16": the same step with bf16 convolution operands (fp32 accumulation, master weights, activations) against
    the oracle with the SAME roundings restated at the same call sites -- so the tolerances stay those of the fp32 test."""
    # Generator gains of 1 instead of the fixture's 8 (at 8 every output sample sits in the saturated tail of tanh).
    # What "match" can mean here: two correct implementations differ in fp32 summation order (1e-6), which moves single operands
    # across a bf16 rounding boundary (4e-3), which flips leaky-ReLU / L1-loss kinks downstream.  Measured on the oracle ALONE
    # (same roundings, weights perturbed by 1e-6 relative): y_hat moves by 5e-5, the generator's gradient tensors by 2.5 %
    # (median) to 11 % (max) of their largest entry; without the roundings by 2e-5 / 0.25 %.  So: losses and y_hat tight, the
    # gradient tensors by direction (cosine >= 0.99; measured >= 0.9978) and size (norm within 5 %); the arithmetic itself is pinned per operator in test_gpu_train_ops.py (1e-4).
    with _bf16_operand_oracle():
        _full_gan_step(cuda_device, oracle_models, "bf16", g_gain=1.0 / 8.0)


def test_full_gan_step_istft_generator_matches_oracle(cuda_device, oracle_models):
    """BASELINE config 5's vocoder: the iSTFTNet head in training (reference test configuration C8C8I: upsampling 8 x 8, then
    conv_post -> exp / sin -> inverse STFT 16 / 4): the whole GAN step against torch autograd through torch.istft."""
    _full_gan_step(cuda_device, oracle_models, "f32", istft=True)


def test_full_gan_step_at_bench_size_matches_oracle(cuda_device, oracle_models):
    """BASELINE config 4 at its own size: 16 segments of 8192 samples per GPU -- the shape bench.py times, where the planner
    picks the 128 x 128 split-K tiles and the one-launch polyphase input gradients -- whole step against the CPU oracle in fp32."""
    _full_gan_step(cuda_device, oracle_models, "f32", g_gain=F32_G_GAIN, B=16, S=8192)


def test_full_gan_step_at_bench_size_bf16_operands(cuda_device, oracle_models):
    """The timed configuration itself (bs 16 x 8192, precision="bf16") against the oracle with the same roundings restated."""
    with _bf16_operand_oracle():
        _full_gan_step(cuda_device, oracle_models, "bf16", g_gain=1.0 / 8.0, B=16, S=8192)


def test_full_gan_step_config5_vocoder_matches_oracle(cuda_device, oracle_models):
    """BASELINE config 5's vocoder as a whole GAN step: 44.1 kHz front-end (n_fft 2048, hop 512) for the mel input and the
    45 x mel-L1 term, generator upsampling 8 x 8 x 2 + iSTFT(16, 4) = hop 512."""
    _full_gan_step(cuda_device, oracle_models, "f32", istft="c5", B=2, S=4096)


def test_full_gan_step_wgan_rmsprop_clipping_matches_oracle(cuda_device, oracle_models):
    """gan_type "wgan" (everyvoice-spec-to-wav-0.5.json:573-605): critic losses mean D(y_hat) - mean D(y) and -mean D(y_hat),
    RMSprop on both sides, critic weights clipped to +-wgan_clip_value after its step -- against torch autograd + torch.optim.RMSprop."""
    _full_gan_step(cuda_device, oracle_models, "f32", g_gain=F32_G_GAIN, gan_type="wgan")


def test_step_without_side_streams_is_bitwise_the_same(cuda_device):
    """The fork / join of the discriminators and MRF branches onto side streams changes the schedule, not the arithmetic."""
    from everyvoice_amd.train.hifigan import HiFiGANTrainer

    g = torch.Generator().manual_seed(8)
    B, S = 2, 2048
    y = (0.3 * torch.tanh(torch.randn(B, 1, S, generator=g))).to(cuda_device)
    mel = torch.randn(B, 80, S // 256, generator=g).to(cuda_device)
    outs, sds = [], []
    for par in (True, False):
        tr = HiFiGANTrainer(device=cuda_device, seed=5, parallel_streams=par)
        outs.append([tr.training_step(mel, y) for _ in range(2)])
        sds.append(tr.state_dict())
    assert outs[0] == outs[1]
    for k in sds[0]:
        assert torch.equal(sds[0][k], sds[1][k]), k


@pytest.mark.parametrize("precision", ["f32", "bf16"])
def test_fused_residual_pairs_equal_the_unfused_sequence(cuda_device, precision):
    """ag.resblock_pair (activations in the packs, residual and activation backward in the epilogues) against the op-by-op
    sequence lrelu -> conv -> lrelu -> conv -> add: the same roundings in the same order, so two steps end bitwise equal -- in
    bf16 through the fused kernels, in f32 through the fall-back that runs the same passes separately."""
    from everyvoice_amd.train.hifigan import HiFiGANTrainer

    g = torch.Generator().manual_seed(8)
    B, S = 2, 2048
    y = (0.3 * torch.tanh(torch.randn(B, 1, S, generator=g))).to(cuda_device)
    mel = torch.randn(B, 80, S // 256, generator=g).to(cuda_device)
    outs, sds = [], []
    for fused in (True, False):
        tr = HiFiGANTrainer(device=cuda_device, seed=5, precision=precision)
        tr.generator.fused_pairs = fused
        tr.generator.time_major = False  # (this test is about the channel-major pair fusion; the time-major stacks: test_gpu_train_tm.py)
        outs.append([tr.training_step(mel, y) for _ in range(2)])
        sds.append(tr.state_dict())
    assert outs[0] == outs[1]
    for k in sds[0]:
        assert torch.equal(sds[0][k], sds[1][k]), k


@pytest.mark.parametrize("precision", ["f32", "bf16"])
def test_graph_replay_equals_eager_steps(cuda_device, precision):
    """use_graph=True: two eager warm-up steps, one captured, then replays -- five steps end bitwise where five eager steps do
    (same kernels, same order per tensor; the optimisers' step counters live on the device)."""
    from everyvoice_amd.train.hifigan import HiFiGANTrainer

    g = torch.Generator().manual_seed(18)
    B, S = 2, 2048
    ys = [(0.3 * torch.tanh(torch.randn(B, 1, S, generator=g))).to(cuda_device) for _ in range(5)]
    mels = [torch.randn(B, 80, S // 256, generator=g).to(cuda_device) for _ in range(5)]
    res = {}
    for graph in (False, True):
        tr = HiFiGANTrainer(device=cuda_device, seed=5, precision=precision, use_graph=graph)
        losses = [tr.training_step(m, y) for m, y in zip(mels, ys)]
        if graph:
            assert tr._graph_failed is None, tr._graph_failed
            assert len(tr._graphs) == 1
        res[graph] = (losses, tr.state_dict(), tr.checkpoint()["optimizer_states"])
    assert res[True][0] == res[False][0]
    for k in res[False][1]:
        assert torch.equal(res[False][1][k], res[True][1][k]), k
    for a, b in zip(res[False][2], res[True][2]):
        assert a["evmi_flat_adamw"]["step"] == b["evmi_flat_adamw"]["step"] == 5
        assert torch.equal(a["evmi_flat_adamw"]["exp_avg_sq"], b["evmi_flat_adamw"]["exp_avg_sq"])


def test_parameters_written_between_replays_reach_the_next_replay(cuda_device):
    """round 5 decided on the HOST whether the discriminators' effective weights and the chains' weight fragments
    are remade, and a captured step baked "nothing to do" in -- a checkpoint loaded into a live trainer (or any write into the flat
    parameter buffer) between two replays left the next discriminator phase on the pre-load weights.  Every step derives both again:
    a trainer that replays, has another trainer's discriminators and generator loaded into it and replays again ends bit for bit where
    an eager trainer treated the same way does."""
    from everyvoice_amd.train.hifigan import HiFiGANTrainer

    g = torch.Generator().manual_seed(19)
    B, S = 2, 2048
    ys = [(0.3 * torch.tanh(torch.randn(B, 1, S, generator=g))).to(cuda_device) for _ in range(6)]
    mels = [torch.randn(B, 80, S // 256, generator=g).to(cuda_device) for _ in range(6)]
    donor = HiFiGANTrainer(device=cuda_device, seed=77, precision="bf16").checkpoint()
    res = {}
    for graph in (False, True):
        tr = HiFiGANTrainer(device=cuda_device, seed=5, precision="bf16", use_graph=graph)
        for m, y in zip(mels[:4], ys[:4]):  # two eager steps, the capture, one replay
            tr.training_step(m, y)
        tr.load_checkpoint(donor, restore_optimizers=False)  # (a live trainer: its captured step stays)
        tr.d_params.data(0).mul_(1.5)  # ... and a direct write into the flat buffer, the lock-step tool's kind
        losses = [tr.training_step(m, y) for m, y in zip(mels[4:], ys[4:])]
        if graph:
            assert tr._graph_failed is None and len(tr._graphs) == 1
        res[graph] = (losses, tr.d_params.flat.clone(), tr.g_params.flat.clone())
    assert res[True][0] == res[False][0]
    assert torch.equal(res[True][1], res[False][1]) and torch.equal(res[True][2], res[False][2])


@pytest.mark.parametrize("graph", [False, True])
def test_gan_step_under_schedule_fuzzing_is_bitwise_the_same(cuda_device, monkeypatch, graph):
    """a bit-for-bit claim that holds on one box and not on another is a missing edge between streams.  Here every
    chain that is forked onto a side stream (the eight discriminators' branches forward and backward, the spectral-norm preparation,
    the weight fragments' stream) starts behind a busy-wait kernel of a random length (tests/helpers.py: fuzz_gan_schedule), so the
    streams finish in orders a quiet GPU never produces and a consumer without an edge from its producer reads early EVERY time.
    Four steps in the bench's precision -- eagerly, and as two eager steps, a capture and a replay (the busy-waits are graph nodes
    then) -- end bit for bit where the unperturbed steps end."""
    from helpers import fuzz_gan_schedule

    from everyvoice_amd.train.hifigan import HiFiGANTrainer

    g = torch.Generator().manual_seed(21)
    B, S = 2, 2048
    ys = [(0.3 * torch.tanh(torch.randn(B, 1, S, generator=g))).to(cuda_device) for _ in range(4)]
    mels = [torch.randn(B, 80, S // 256, generator=g).to(cuda_device) for _ in range(4)]

    def run():
        tr = HiFiGANTrainer(device=cuda_device, seed=5, precision="bf16", use_graph=graph)
        losses = [tr.training_step(m, y) for m, y in zip(mels, ys)]
        assert tr._graph_failed is None and bool(tr._graphs) == graph
        return losses, tr.d_params.flat.clone(), tr.g_params.flat.clone(), tr.d_params.grad.clone(), tr.g_params.grad.clone()

    plain = run()
    fuzz_gan_schedule(3, monkeypatch.setattr)
    fuzzed = run()
    assert fuzzed[0] == plain[0]
    for a, b, name in zip(fuzzed[1:], plain[1:], ("d", "g", "d_grad", "g_grad")):
        assert torch.equal(a, b), name


def test_generator_warmup_steps_train_the_generator_alone(cuda_device):
    """generator_warmup_steps (same schema): during the warm-up the discriminators are neither stepped nor consulted -- the
    generator follows the reconstruction loss only; afterwards the full GAN step runs."""
    from everyvoice_amd.train.hifigan import HiFiGANTrainer

    g = torch.Generator().manual_seed(28)
    y = (0.3 * torch.tanh(torch.randn(2, 1, 2048, generator=g))).to(cuda_device)
    mel = torch.randn(2, 80, 8, generator=g).to(cuda_device)
    tr = HiFiGANTrainer(device=cuda_device, seed=5, generator_warmup_steps=2)
    d0, g0 = tr.d_params.flat.clone(), tr.g_params.flat.clone()
    out = tr.training_step(mel, y)
    assert out["d"] == 0.0 and out["g_adv"] == 0.0 and out["g_fm"] == 0.0 and out["g_mel"] > 0
    assert torch.equal(tr.d_params.flat, d0) and not torch.equal(tr.g_params.flat, g0) and tr.d_params.step == 0
    tr.training_step(mel, y)
    out = tr.training_step(mel, y)  # third step: past the warm-up
    assert out["d"] > 0 and out["g_adv"] > 0 and out["g_fm"] > 0 and not torch.equal(tr.d_params.flat, d0) and tr.d_params.step == 1


@pytest.mark.parametrize("name", ["adam", "adamw", "rms"])
def test_optimizer_kernel_matches_torch_optim(cuda_device, name):
    """The flat-buffer optimiser kernel against torch.optim.{Adam, AdamW, RMSprop} over three steps (weight decay on)."""
    from everyvoice_amd.train.layers import ParamGroup

    g = torch.Generator().manual_seed(1)
    grp = ParamGroup(cuda_device)
    grp.declare("w", (257,))
    grp.finalize()
    w0 = torch.randn(257, generator=g)
    grp.load("w", w0)
    ref = torch.nn.Parameter(w0.clone())
    kw = dict(lr=1e-2, eps=1e-8, weight_decay=0.05)
    opt = {"adam": lambda: torch.optim.Adam([ref], betas=(0.8, 0.99), **kw), "adamw": lambda: torch.optim.AdamW([ref], betas=(0.8, 0.99), **kw),
           "rms": lambda: torch.optim.RMSprop([ref], alpha=0.9, **kw)}[name]()
    for _ in range(3):
        gr = torch.randn(257, generator=g)
        ref.grad = gr.clone()
        opt.step()
        grp.gradient(0).copy_(gr.to(cuda_device))
        grp.optimizer_step(name, lr=1e-2, betas=(0.8, 0.99), eps=1e-8, weight_decay=0.05, alpha=0.9)
    torch.testing.assert_close(grp.data(0).cpu(), ref.detach(), rtol=2e-5, atol=2e-6)
    assert grp.step == 3 and int(grp.step_dev.item()) == 3


def test_full_gan_step_resblock2_generator_matches_oracle(cuda_device, oracle_models):
    """The "2" resblock option (upstream V3 shape: 256 initial channels, upsampling 8 x 8 x 4, kernels 3 / 5 / 7 with two
    dilations each): the whole GAN step against the oracle."""
    _full_gan_step(cuda_device, oracle_models, "f32", istft="v3")


C5_AUDIO = dict(input_sampling_rate=44100, output_sampling_rate=44100, n_fft=2048, fft_window_size=2048, fft_hop_size=512)
C5_MEL = dict(sr=44100, n_fft=2048, win=2048, hop=512)


def _full_gan_step(cuda_device, oracle_models, precision, g_gain=1.0, istft=False, B=2, S=2048, gan_type="original", **trainer_kw):
    from everyvoice_amd.config import HiFiGANConfig
    from everyvoice_amd.train.hifigan import HiFiGANTrainer

    mpd_ref, msd_ref = MultiPeriodDiscriminatorRef().train(), MultiScaleDiscriminatorRef().train()
    if istft:
        from oracle.hifigan_ref import HiFiGANModelConfigRef

        torch.manual_seed(77)
        v3 = dict(resblock="2", upsample_rates=[8, 8, 4], upsample_kernel_sizes=[16, 16, 8], upsample_initial_channel=256,
                  resblock_kernel_sizes=[3, 5, 7], resblock_dilation_sizes=[[1, 2], [2, 6], [3, 12]])
        c5 = dict(istft_layer=True, upsample_rates=[8, 8, 2], upsample_kernel_sizes=[16, 16, 4])  # x 4 (iSTFT hop) = hop 512
        model_kw = v3 if istft == "v3" else c5 if istft == "c5" else dict(istft_layer=True, upsample_rates=[8, 8], upsample_kernel_sizes=[16, 16])
        g_ref = GeneratorRef(HiFiGANModelConfigRef(**model_kw)).train()
        with torch.no_grad():  # livelier than N(0, 0.01), short of saturating exp()
            for n, p in g_ref.named_parameters():
                if n.endswith("weight_g") and not n.startswith("conv_post"):
                    p.mul_(2.0)
        for new, old in zip((mpd_ref, msd_ref), oracle_models[1:]):
            new.load_state_dict(old.state_dict())
        config = HiFiGANConfig(model=model_kw, preprocessing=dict(audio=C5_AUDIO) if istft == "c5" else {})
    else:
        g_ref = GeneratorRef().train()
        for new, old in zip((g_ref, mpd_ref, msd_ref), oracle_models):  # weight-normed modules do not deepcopy
            new.load_state_dict(old.state_dict())
        config = None
    if g_gain != 1.0:
        with torch.no_grad():
            for n, p in g_ref.named_parameters():
                if n.endswith("weight_g"):
                    p.mul_(g_gain)
    opt_kw = dict(lr=2e-4, betas=(0.8, 0.99), eps=1e-8, weight_decay=0.01)
    wgan = gan_type == "wgan"
    if wgan:  # critic losses, RMSprop (the optimiser WGAN is defined with), weight clipping
        trainer_kw = dict(gan_type="wgan", optimizer="rms", alpha=0.99, wgan_clip_value=0.01, **trainer_kw)
    tr = HiFiGANTrainer(config, device=cuda_device, precision=precision, **opt_kw, **trainer_kw)
    tr.load_reference_state(g_ref.state_dict(), mpd_ref.state_dict(), msd_ref.state_dict())
    tr.keep_grads = True
    loss_rel = 2e-4 if precision == "f32" else 2e-3

    gen = torch.Generator().manual_seed(11)
    y = 0.3 * torch.tanh(torch.randn(B, 1, S, generator=gen))
    mel_kw = C5_MEL if istft == "c5" else {}
    hop = mel_kw.get("hop", 256)
    mel = mel_ref.mel_spectrogram_ref(y.squeeze(1), **mel_kw)[:, :, : S // hop]
    y32, mel32 = y, mel
    if precision == "f32":
        # The fp32 product is compared with the oracle run in FLOAT64 from the same fp32 parameters and inputs.  Two fp32 runs of
        # this step differ from each other by more than either differs from the exact result: a leaky-ReLU / L1 kink that one of them
        # takes on the other side moves a gradient tensor by up to 4e-2 of its largest entry at B = 2 (measured: fp32 oracle vs fp64
        # oracle 4.0e-2 on resblocks.3.convs1.2.weight_v, and the product's deviation from the fp32 oracle was that same 4.0e-2).
        g_ref, mpd_ref, msd_ref = g_ref.double(), mpd_ref.double(), msd_ref.double()
        y, mel = y.double(), mel.double()

    # ---- oracle step (jik876 training loop order: D step, then G step) ----
    d_params_ref = list(mpd_ref.parameters()) + list(msd_ref.parameters())
    if wgan:
        rms = dict(lr=2e-4, alpha=0.99, eps=1e-8, weight_decay=0.01)
        opt_g, opt_d = torch.optim.RMSprop(g_ref.parameters(), **rms), torch.optim.RMSprop(d_params_ref, **rms)
    else:
        opt_g, opt_d = torch.optim.AdamW(g_ref.parameters(), **opt_kw), torch.optim.AdamW(d_params_ref, **opt_kw)
    y_hat = g_ref(mel)
    opt_d.zero_grad()
    r1, g1, _, _ = mpd_ref(y, y_hat.detach())
    r2, g2, _, _ = msd_ref(y, y_hat.detach())
    if wgan:
        loss_d = sum(gg.mean() - rr.mean() for rr, gg in zip(r1 + r2, g1 + g2))
    else:
        loss_d = discriminator_loss_ref(r1, g1) + discriminator_loss_ref(r2, g2)
    loss_d.backward()
    d_grads = {"mpd." + k: v.grad.clone().float() for k, v in mpd_ref.named_parameters()}
    d_grads.update({"msd." + k: v.grad.clone().float() for k, v in msd_ref.named_parameters()})
    opt_d.step()
    if wgan:
        with torch.no_grad():
            for prm in d_params_ref:
                prm.clamp_(-0.01, 0.01)
    opt_g.zero_grad()
    lm_y = mel_ref.mel_spectrogram_ref(y.squeeze(1), **mel_kw)
    lm_g = mel_ref.mel_spectrogram_ref(y_hat.squeeze(1), **mel_kw)
    loss_mel = F.l1_loss(lm_y, lm_g) * 45
    _, g1, fr1, fg1 = mpd_ref(y, y_hat)
    _, g2, fr2, fg2 = msd_ref(y, y_hat)
    loss_fm = feature_loss_ref(fr1, fg1) + feature_loss_ref(fr2, fg2)
    loss_adv = -sum(gg.mean() for gg in g1 + g2) if wgan else generator_loss_ref(g1) + generator_loss_ref(g2)
    (loss_adv + loss_fm + loss_mel).backward()
    g_grads = {k: v.grad.clone().float() for k, v in g_ref.named_parameters()}
    opt_g.step()
    y_hat = y_hat.float()
    g_ref, mpd_ref, msd_ref = g_ref.float(), mpd_ref.float(), msd_ref.float()  # the updated parameters, for the comparisons below

    # ---- the same step on the GPU ----
    out = tr.training_step(mel32.to(cuda_device), y32.to(cuda_device))
    if precision == "f32":
        got_y = tr.last_grads["y_hat"].cpu().view(B, 1, S)
        bad = ((got_y - y_hat.detach()).abs() > 1e-4).nonzero()
        if len(bad):
            print("y_hat mismatches:", [(tuple(i.tolist()), float(got_y[tuple(i)]), float(y_hat.detach()[tuple(i)])) for i in bad[:20]])
        torch.testing.assert_close(got_y, y_hat.detach(), rtol=1e-4, atol=1e-5)
    else:
        dyh = (tr.last_grads["y_hat"].cpu().view(B, 1, S) - y_hat.detach()).abs()
        assert float(dyh.max()) <= 2e-2 * float(y_hat.detach().abs().max()), float(dyh.max())
    assert out["d"] == pytest.approx(float(loss_d.detach()), rel=loss_rel)
    assert out["g_adv"] == pytest.approx(float(loss_adv.detach()), rel=loss_rel)
    assert out["g_fm"] == pytest.approx(float(loss_fm.detach()), rel=loss_rel)
    assert out["g_mel"] == pytest.approx(float(loss_mel.detach()), rel=loss_rel)
    if precision != "f32":
        # Per tensor: cosine with the oracle's gradient and ratio of the norms.  Bounds = measured worst minus a margin
        # (profiles/r05_gan_bf16_gradient_cosines.txt: 388 tensors per test):
        #   at the TIMED configuration (16 x 8192, what bench.py runs): discriminators worst cosine 0.9999, norms within 0.1 %;
        #   generator worst 0.9989, norms within 1.6 %  ->  0.999 / 1 % and 0.998 / 3 %;
        #   at 2 x 2048 (few samples: single leaky-ReLU / L1 kinks that flip with a bf16 rounding weigh more; the generator's residual
        #   stacks store activations AND gradients in time-major bf16, train/mrf_tm.py): discriminators 0.9995 / 0.4 %, generator worst
        #   0.9907 (a weight_g gradient of the 32-channel stage: one scalar per channel, a sum of near-cancelling terms), median
        #   0.9975, 11 % of the tensors below 0.995, norms within 5.2 %  ->  0.999 / 1.5 % and 0.988 / 6 %, median >= 0.996, at most
        #   20 % below 0.995.  A wrong tap in one stack drags its tensors to 0.9 or below and fails the floor at either size; a
        #   systematic loss of precision fails the median.
        timed = B * S >= 65536 and os.environ.get("EVMI_TRAIN_TM", "1") != "0"  # (the channel-major A/B path keeps the wider bounds)
        cosines = {"d": [], "g": []}
        for grads, key in ((d_grads, "d"), (g_grads, "g")):
            for name, want in grads.items():
                got = tr.last_grads[key][name].cpu().reshape(want.shape).double().flatten()
                w = want.double().flatten()
                if float(w.norm()) < 1e-12:
                    continue
                cos = float(torch.dot(got, w) / (got.norm() * w.norm() + 1e-300))
                ratio = float(got.norm() / w.norm())
                if os.environ.get("EVMI_TEST_REPORT"):
                    print(f"COS {cos:.4f} ratio {ratio:.3f} {key}.{name}")
                cosines[key].append(cos)
                if key == "d":
                    floor, tol = 0.999, (0.01 if timed else 0.015)
                else:
                    floor, tol = (0.998, 0.03) if timed else (0.988, 0.06)
                assert cos >= floor and 1.0 - tol <= ratio <= 1.0 + tol, f"{key}.{name}: cos {cos:.4f} norm ratio {ratio:.3f}"
        gc = sorted(cosines["g"])
        med, low = gc[len(gc) // 2], sum(c < 0.995 for c in gc)
        assert med >= (0.9995 if timed else 0.996) and low <= (0 if timed else 0.20 * len(gc)), (med, low, len(gc))
        return
    # fp32 evaluation noise of the generator's gradients against the exact (fp64) step: the fp32 ORACLE itself is 1.15e-2 away from
    # the fp64 oracle on resblocks.8.convs1.0.weight_v at 2 x 2048 samples (kink flips), 1.3e-3 at 16 x 8192 where they average out
    g_rel = 1e-2 if B * S >= 65536 else 2e-2
    for name, want in d_grads.items():
        _grad_close(name, tr.last_grads["d"][name].cpu(), want)
    for name, want in g_grads.items():
        _grad_close(name, tr.last_grads["g"][name].cpu(), want, rel=g_rel)
    # updated parameters after AdamW on both sides
    if wgan:  # RMSprop's first step is lr * g / (sqrt(0.01 g^2) + eps) = 10 lr sign(g); the clipped critic is compared exactly below
        sd_d = tr.d_params.state_dict()
        ref_d = {**{"mpd." + k: v for k, v in mpd_ref.state_dict().items()}, **{"msd." + k: v for k, v in msd_ref.state_dict().items()}}
        for k, got in sd_d.items():  # the trainer's parameters (the spectral-norm buffers are not among them)
            got, want = got.cpu().reshape(ref_d[k].shape), ref_d[k]
            assert float(got.abs().max()) <= 0.01 + 1e-7, k                       # wgan_clip_value
            assert float(((got - want).abs() > 2.2e-3).float().mean()) < 0.02, k  # entries whose tiny gradient changed sign: +-10 lr
        sd_g = tr.g_params.state_dict()
        for k, v in g_ref.state_dict().items():
            # RMSprop's first step is +-10 lr by the gradient's SIGN: only entries well above the 2e-2 gradient tolerance are pinned
            _params_close(k, sd_g[k].cpu(), v, g_grads.get(k), lr=2e-3, solid_frac=1e-1)
        return
    sd_g = tr.g_params.state_dict()
    for k, v in g_ref.state_dict().items():
        _params_close(k, sd_g[k].cpu(), v, g_grads.get(k))
    sd_d = tr.d_params.state_dict()
    ref_d = {"mpd." + k: v for k, v in mpd_ref.state_dict().items()}
    ref_d.update({"msd." + k: v for k, v in msd_ref.state_dict().items()})
    for k, v in ref_d.items():
        if k.endswith("weight_u") or k.endswith("weight_v") and "discriminators.0" in k and k.startswith("msd."):
            continue  # spectral-norm buffers: checked below
        if k in sd_d:
            _params_close(k, sd_d[k].cpu(), v, d_grads.get(k))
    # power-iteration state of the spectral-norm discriminator after the step's four forward calls
    for i, conv in enumerate(tr.msd[0].layers()):
        name = f"discriminators.0.convs.{i}" if i < 7 else "discriminators.0.conv_post"
        torch.testing.assert_close(conv.u.cpu(), msd_ref.state_dict()[name + ".weight_u"], rtol=1e-3, atol=1e-5)


def test_checkpoint_resume_and_export(cuda_device, tmp_path):
    """Save after one step, resume in a fresh trainer: the second step is bitwise the same as without the interruption;
    the exported generator checkpoint loads into the inference vocoder and reproduces the trainer's generator."""
    from everyvoice_amd.train import autograd as ag
    from everyvoice_amd.train.hifigan import HiFiGANTrainer, _to_cbt
    from everyvoice_amd.vocoder import load_hifigan_from_checkpoint

    g = torch.Generator().manual_seed(4)
    B, S = 2, 2048
    y = (0.3 * torch.tanh(torch.randn(B, 1, S, generator=g))).to(cuda_device)
    mel = torch.randn(B, 80, S // 256, generator=g).to(cuda_device)
    a = HiFiGANTrainer(device=cuda_device, seed=7)
    a.training_step(mel, y)
    path = tmp_path / "step1.ckpt"
    torch.save(a.checkpoint(), path)
    out_a = a.training_step(mel, y)

    b = HiFiGANTrainer(device=cuda_device, seed=99)  # different init: everything must come from the file
    ckpt = torch.load(path, map_location="cpu", weights_only=False)
    assert ckpt["model_info"] == {"name": "HiFiGAN", "version": "1.0"}
    import json
    json.dumps(ckpt["hyper_parameters"]["config"])  # JSON-only, as the reference requires
    b.load_checkpoint(ckpt)
    assert b.global_step == 1
    out_b = b.training_step(mel, y)
    assert out_a == out_b
    for (k, va), vb in zip(a.state_dict().items(), b.state_dict().values()):
        assert torch.equal(va, vb), k

    with pytest.raises(TypeError, match="Wrong model type"):
        b.load_checkpoint({**ckpt, "model_info": {"name": "FastSpeech2", "version": "1.0"}})
    with pytest.raises(ValueError, match="newer version"):
        b.load_checkpoint({**ckpt, "model_info": {"name": "HiFiGAN", "version": "2.0"}})

    model, _ = load_hifigan_from_checkpoint(a.export_generator_checkpoint(), cuda_device, precision="f32")
    a._materialize(a.generator.layers())
    want = a.generator.forward(ag.Tape(), ag.Var(_to_cbt(mel), needs_grad=False)).data.view(B, 1, -1)
    torch.testing.assert_close(model(mel), want, rtol=1e-4, atol=1e-5)


def test_bucketed_allreduce_path_on_rccl_world_of_one(cuda_device):
    """The N > 1 code path (BucketReducer: side stream, events, async RCCL all-reduces launched during backward, 1/world scaling)
    on a one-rank "nccl" group: the step must equal the plain single-GPU step bit for bit."""
    import os
    import socket

    import torch.distributed as dist

    from everyvoice_amd.train.hifigan import HiFiGANTrainer

    g = torch.Generator().manual_seed(8)
    B, S = 2, 2048
    y = (0.3 * torch.tanh(torch.randn(B, 1, S, generator=g))).to(cuda_device)
    mel = torch.randn(B, 80, S // 256, generator=g).to(cuda_device)
    plain = HiFiGANTrainer(device=cuda_device, seed=5)
    plain.keep_grads = True
    out_plain = plain.training_step(mel, y)

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=cuda_device)
    try:
        dp = HiFiGANTrainer(device=cuda_device, seed=5, process_group=True)
        dp.keep_grads = True
        out_dp = dp.training_step(mel, y)
    finally:
        dist.destroy_process_group()
    assert out_plain == out_dp
    for side in ("d", "g"):
        for k, v in plain.last_grads[side].items():
            assert torch.equal(v, dp.last_grads[side][k]), (side, k)


def test_graph_mode_under_data_parallelism_on_rccl_world_of_one(cuda_device):
    """use_graph=True with a process group: the step is captured in stretches that end at gradient-bucket boundaries (period
    discriminators | scale discriminators | discriminator update + generator backward down to its first large bucket | ... |
    generator update), each bucket's RCCL all-reduce launched on a side stream between two replays so that it runs under the next
    stretch -- on a one-rank "nccl" group five steps must end exactly where five plain single-GPU steps do."""
    import os
    import socket

    import torch.distributed as dist

    from everyvoice_amd.train.hifigan import HiFiGANTrainer

    g = torch.Generator().manual_seed(8)
    B, S = 2, 2048
    y = (0.3 * torch.tanh(torch.randn(B, 1, S, generator=g))).to(cuda_device)
    mel = torch.randn(B, 80, S // 256, generator=g).to(cuda_device)
    plain = HiFiGANTrainer(device=cuda_device, seed=5)
    out_plain = [plain.training_step(mel, y) for _ in range(5)]
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=cuda_device)
    try:
        dp = HiFiGANTrainer(device=cuda_device, seed=5, process_group=True, use_graph=True)
        out_dp = [dp.training_step(mel, y) for _ in range(5)]
        assert dp._graph_failed is None, dp._graph_failed
        (entry,) = dp._graphs.values()
        assert len(entry["graphs"]) >= 5 and sum(a is not None for a in entry["after"]) == len(entry["graphs"])  # cut at bucket boundaries
    finally:
        dist.destroy_process_group()
    assert out_plain == out_dp
    for k, v in plain.state_dict().items():
        assert torch.equal(v, dp.state_dict()[k]), k


def test_multi_resolution_stft_loss_value_and_gradient(cuda_device):
    """The selectable multi-resolution STFT loss (BASELINE config 4) vs torch.stft + autograd on the CPU."""
    from everyvoice_amd.train.hifigan import MultiResolutionSTFTLoss
    from oracle.hifigan_ref import mrstft_loss_ref

    g = torch.Generator().manual_seed(31)
    B, T = 3, 8192
    y = 0.3 * torch.tanh(torch.randn(B, T, generator=g))
    y_hat = (y + 0.1 * torch.randn(B, T, generator=g)).requires_grad_()
    want = mrstft_loss_ref(y, y_hat) * 2.5
    want.backward()
    out = torch.zeros(1, device=cuda_device)
    grad = MultiResolutionSTFTLoss(cuda_device).loss_and_grad(y.to(cuda_device), y_hat.detach().to(cuda_device), 2.5, out)
    assert float(out) == pytest.approx(float(want), rel=2e-5)
    # the log-magnitude term's gradient is sign / (n |Y^|): bins with tiny magnitude amplify the fp32 differences between
    # torch's FFT and the DFT-as-GEMM by 1 / |Y^|, so the bound is on the L2 norm (and a looser one on the worst sample)
    diff = grad.cpu() - y_hat.grad
    assert float(diff.norm() / y_hat.grad.norm()) <= 3e-3
    assert float(diff.abs().max()) <= 1e-2 * float(y_hat.grad.abs().max())


def test_training_step_with_mrstft_option_runs(cuda_device):
    from everyvoice_amd.train.hifigan import HiFiGANTrainer

    g = torch.Generator().manual_seed(2)
    y = (0.3 * torch.tanh(torch.randn(2, 1, 2048, generator=g))).to(cuda_device)
    mel = torch.randn(2, 80, 8, generator=g).to(cuda_device)
    out = HiFiGANTrainer(device=cuda_device, reconstruction_loss="mel+mrstft").training_step(mel, y)
    assert out["g_stft"] > 0 and out["g_mel"] > 0 and out["g_total"] == pytest.approx(out["g_adv"] + out["g_fm"] + out["g_mel"] + out["g_stft"])
    with pytest.raises(ValueError):
        HiFiGANTrainer(device=cuda_device, reconstruction_loss="l2")


def test_weight_gradients_on_sibling_streams_do_not_change_the_step(cuda_device):
    """side_wgrad=True (eager mode): weight / bias gradient kernels queued beside the input-gradient chains and joined before
    anything reads the parameter gradients -- the same arithmetic, so two steps end bitwise where the default schedule does."""
    from everyvoice_amd.train.hifigan import HiFiGANTrainer

    g = torch.Generator().manual_seed(8)
    B, S = 2, 2048
    y = (0.3 * torch.tanh(torch.randn(B, 1, S, generator=g))).to(cuda_device)
    mel = torch.randn(B, 80, S // 256, generator=g).to(cuda_device)
    outs, sds = [], []
    for side in (True, False):
        tr = HiFiGANTrainer(device=cuda_device, seed=5, precision="bf16", side_wgrad=side)
        outs.append([tr.training_step(mel, y) for _ in range(2)])
        sds.append(tr.state_dict())
    assert outs[0] == outs[1]
    for k in sds[0]:
        assert torch.equal(sds[0][k], sds[1][k]), k


def test_bench_size_gan_steps_are_reproducible_run_to_run(cuda_device):
    """Two trainers in lockstep on a bench-size batch (16 x 8192 samples, bf16, the captured step with its eight discriminator
    streams): 60 steps each, generator and discriminator parameters and the losses bitwise equal after every one -- no atomics and
    fixed-order reductions everywhere, so a difference would be a race between streams or inside a kernel."""
    from everyvoice_amd.spectral import MelSpectrogram
    from everyvoice_amd.train.hifigan import HiFiGANTrainer

    g = torch.Generator().manual_seed(1234)
    B, S = 16, 8192
    y = (0.3 * torch.tanh(torch.randn(B, 1, S, generator=g))).to(cuda_device)
    mel = MelSpectrogram()(y.squeeze(1), log=True)[:, :, : S // 256].contiguous()
    a = HiFiGANTrainer(device=cuda_device, seed=3, precision="bf16", use_graph=True)
    b = HiFiGANTrainer(device=cuda_device, seed=3, precision="bf16", use_graph=True)
    for step in range(60):
        la, lb = a.training_step(mel, y, sync=False), b.training_step(mel, y, sync=False)
        assert torch.equal(la, lb), f"losses differ at step {step}"
        assert torch.equal(a.g_params.flat, b.g_params.flat) and torch.equal(a.d_params.flat, b.d_params.flat), f"parameters differ after step {step}"
    assert a._graph_failed is None and b._graph_failed is None


def test_channel_major_generator_path_behind_its_switch():
    """precision="bf16" trains the generator's residual stacks in time-major bf16 on the inference kernels by default
    (train/mrf_tm.py); EVMI_TRAIN_TM=0 keeps every layer on the channel-major packed kernels.  The switch is read when a trainer is
    built: the bf16 whole-step comparisons of this file run once more in a child process with it off, so that path cannot rot."""
    import os
    import subprocess
    import sys

    r = subprocess.run([sys.executable, "-m", "pytest", __file__, "-q", "-x", "-k", "bf16 and not behind_its_switch"],
                       env=dict(os.environ, EVMI_TRAIN_TM="0"), capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
