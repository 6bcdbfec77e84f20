"""FastSpeech2 training (csrc/fs2_train_ops.hip + train/fs2.py) against torch-CPU autograd of the oracle module.

Operator tests compare each backward kernel with torch autograd of the same op (fp32, tolerance 2e-4 of the tensor's largest
entry: only the summation order differs).  The step test loads the trainer's parameters into ``oracle.fs2_ref.FastSpeech2Ref``
in ``train()`` mode (dropout 0 so both sides see the same network; BatchNorm on batch statistics), runs
``training_losses_ref`` + ``backward()`` and checks every loss, every parameter gradient, the BatchNorm running statistics and
the parameters after one clipped AdamW step.  Gradient tolerance 2e-3 (L2, relative): 8 conformer layers of fp32 re-association.
"""

import math

import pytest
import torch
import torch.nn.functional as F

from oracle.fs2_ref import FastSpeech2ConfigRef, FastSpeech2Ref, training_losses_ref

pytestmark = pytest.mark.gpu


def _close(got, want, rel=2e-4):
    scale = float(want.abs().max()) + 1e-12
    err = float((got.cpu() - want).abs().max())
    assert err <= rel * scale, f"err {err:.3e} vs scale {scale:.3e}"


def _l2close(got, want, rel, name=""):
    num = float((got.cpu().reshape(-1) - want.reshape(-1)).norm())
    den = float(want.norm()) + 1e-8
    floor = 1e-6 * want.numel() ** 0.5  # gradients that are exactly zero in exact arithmetic (a bias in front of BatchNorm) are noise on both sides
    assert num <= rel * den + floor, f"{name}: |diff| {num:.3e} vs |want| {den:.3e}"


def _cbt(x):  # [B, C, T] -> [C, B, T]
    return x.permute(1, 0, 2).contiguous()


# ---- operators -----------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("C,B,T", [(64, 3, 37), (256, 2, 130), (10, 1, 5)])
def test_layernorm_backward(cuda_device, C, B, T):
    from everyvoice_amd.train import ops

    g = torch.Generator().manual_seed(C + T)
    x = torch.randn(B, C, T, generator=g, requires_grad=True)
    gamma = (torch.rand(C, generator=g) + 0.5).requires_grad_()
    beta = torch.randn(C, generator=g).requires_grad_()
    dy = torch.randn(B, C, T, generator=g)
    y = F.layer_norm(x.transpose(1, 2), (C,), gamma, beta).transpose(1, 2)
    y.backward(dy)
    dev = cuda_device
    dgamma = torch.full((C,), 1.0, device=dev)  # accumulated into: start from a known non-zero value
    dbeta = torch.full((C,), -2.0, device=dev)
    dx = ops.layernorm_bwd(_cbt(x.detach()).to(dev), gamma.detach().to(dev), _cbt(dy).to(dev), dgamma, dbeta)
    _close(dx.cpu().permute(1, 0, 2), x.grad)
    _close(dgamma.cpu() - 1.0, gamma.grad)
    _close(dbeta.cpu() + 2.0, beta.grad)


@pytest.mark.parametrize("act", [0, 2, 4])
def test_batchnorm_train_forward_backward(cuda_device, act):
    from everyvoice_amd.train import ops

    C, B, T = 24, 3, 50
    g = torch.Generator().manual_seed(act)
    x = (torch.randn(B, C, T, generator=g) * 2 + 0.5).requires_grad_()
    bn = torch.nn.BatchNorm1d(C).train()
    with torch.no_grad():
        bn.weight.copy_(torch.rand(C, generator=g) + 0.5)
        bn.bias.copy_(torch.randn(C, generator=g) * 0.3)
        bn.running_mean.copy_(torch.randn(C, generator=g))
        bn.running_var.copy_(torch.rand(C, generator=g) + 0.5)
    dev = cuda_device
    rm, rv = bn.running_mean.clone().to(dev), bn.running_var.clone().to(dev)
    z = bn(x)
    y = F.silu(z) if act == 2 else (torch.tanh(z) if act == 4 else z)
    dy = torch.randn(B, C, T, generator=g)
    y.backward(dy)
    gam, bet = bn.weight.detach().to(dev), bn.bias.detach().to(dev)
    out, mean, rstd = ops.batchnorm_fwd(_cbt(x.detach()).to(dev), gam, bet, rm, rv, act)
    _close(out.cpu().permute(1, 0, 2), y.detach())
    _close(rm, bn.running_mean, 1e-5)
    _close(rv, bn.running_var, 1e-5)
    dg, db = torch.zeros(C, device=dev), torch.zeros(C, device=dev)
    dx = ops.batchnorm_bwd(_cbt(x.detach()).to(dev), gam, bet, mean, rstd, _cbt(dy).to(dev), dg, db, act)
    _close(dx.cpu().permute(1, 0, 2), x.grad)
    _close(dg, bn.weight.grad)
    _close(db, bn.bias.grad)


@pytest.mark.parametrize("k", [3, 9])
def test_depthwise_conv_backward(cuda_device, k):
    from everyvoice_amd.train import ops

    C, B, T = 20, 3, 41
    g = torch.Generator().manual_seed(k)
    x = torch.randn(B, C, T, generator=g, requires_grad=True)
    w = torch.randn(C, 1, k, generator=g, requires_grad=True)
    b = torch.randn(C, generator=g, requires_grad=True)
    y = F.conv1d(x, w, b, padding=(k - 1) // 2, groups=C)
    dy = torch.randn(B, C, T, generator=g)
    y.backward(dy)
    dev = cuda_device
    got = ops.dwconv_fwd(_cbt(x.detach()).to(dev), w.detach().to(dev), b.detach().to(dev), k)
    _close(got.cpu().permute(1, 0, 2), y.detach())
    dw, db = torch.zeros(C, 1, k, device=dev), torch.zeros(C, device=dev)
    dx = ops.dwconv_bwd(_cbt(x.detach()).to(dev), w.detach().to(dev), _cbt(dy).to(dev), dw, db, k)
    _close(dx.cpu().permute(1, 0, 2), x.grad)
    _close(dw, w.grad)
    _close(db, b.grad)


def _attention_ref(qkv, lens, H, keep_mask=None, p=0.0):
    """torch.nn.MultiheadAttention's arithmetic on [B, 3D, T]: key padding mask, softmax, dropout on the probabilities."""
    B, D3, T = qkv.shape
    D = D3 // 3
    dh = D // H
    q, k, v = [t.reshape(B, H, dh, T) for t in qkv.split(D, dim=1)]
    s = torch.einsum("bhdq,bhdk->bhqk", q, k) * dh ** -0.5
    s = s.masked_fill((torch.arange(T)[None, :] >= lens[:, None])[:, None, None, :], float("-inf"))
    pr = torch.softmax(s, -1)
    if keep_mask is not None:
        pr = pr * keep_mask / (1.0 - p)
    return torch.einsum("bhqk,bhdk->bhdq", pr, v).reshape(B, D, T)


@pytest.mark.parametrize("D,H,B,T", [(64, 2, 3, 29), (256, 2, 2, 70), (256, 2, 2, 947), (128, 2, 2, 161)])
def test_attention_training_forward_backward(cuda_device, D, H, B, T):
    """evmi_mha_fwd_f32 / evmi_mha_bwd_f32 (flash-style, probabilities recomputed in the backward) against torch autograd of
    the explicit softmax attention; (256, 2, B, 947) is BASELINE config 3's decoder shape: d_head 128, the longest utterance."""
    from everyvoice_amd.train import ops

    g = torch.Generator().manual_seed(D + T)
    qkv = torch.randn(B, 3 * D, T, generator=g, requires_grad=True)
    lens = torch.randint(T // 2, T + 1, (B,), generator=g)
    lens[0] = T
    o = _attention_ref(qkv, lens, H)
    do = torch.randn(B, D, T, generator=g)
    o.backward(do)
    dev = cuda_device
    x = _cbt(qkv.detach()).to(dev)
    lens32 = lens.to(dev, torch.int32)
    out, saved = ops.attention_train_fwd(x, lens32, H)
    _close(out.cpu().permute(1, 0, 2), o.detach())
    # the inference kernel (fused, fp32 matrix cores) computes the same thing
    from everyvoice_amd import _lib
    fused = torch.empty_like(out)
    _lib.check(_lib.load().evmi_attention_cbt_f32(x.data_ptr(), lens32.data_ptr(), fused.data_ptr(), B, T, D, H, torch.cuda.current_stream().cuda_stream), "attention")
    _close(fused.cpu(), out.cpu())
    dqkv = ops.attention_train_bwd(x, saved, _cbt(do).to(dev), H)
    _close(dqkv.cpu().permute(1, 0, 2), qkv.grad)
    again = ops.attention_train_bwd(x, saved, _cbt(do).to(dev), H)
    assert torch.equal(again, dqkv)  # one writer per element, fixed summation order


@pytest.mark.parametrize("D,H,B,T,p", [(256, 2, 2, 300, 0.0), (64, 2, 3, 70, 0.2), (128, 2, 2, 947, 0.1), (256, 2, 3, 333, 0.1)])
def test_attention_training_bf16_operands(cuda_device, D, H, B, T, p):
    """evmi_mha_{fwd,bwd}_bf16: Q / K / V / dO, probabilities and score gradients rounded to bf16 into the matrix cores, fp32
    statistics and accumulation -- against torch autograd in fp32 with the same dropout mask: output within 1e-2 of its scale,
    every gradient block (dq, dk, dv) cosine >= 0.999 and norm within 1 %."""
    from everyvoice_amd.train import ops

    g = torch.Generator().manual_seed(D + T)
    dev = cuda_device
    qkv = torch.randn(B, 3 * D, T, generator=g, requires_grad=True)
    lens = torch.randint(T // 2, T + 1, (B,), generator=g)
    lens[0] = T
    if B > 2:
        lens[-1] = 5  # an item shorter than one key tile (whole query blocks past its length, walked tiles mostly padding)
    keep = None
    if p > 0:
        ones = torch.ones(B * T * T, device=dev)
        keep = torch.stack([(ops.dropout(ones, p, 7 + h) > 0).float().view(B, T, T) for h in range(H)], dim=1).cpu()
    o = _attention_ref(qkv, lens, H, keep, p)
    do = torch.randn(B, D, T, generator=g)
    o.backward(do)
    x, lens32 = _cbt(qkv.detach()).to(dev), lens.to(dev, torch.int32)
    ops.CONV_BACKEND["operands"] = "bf16"
    try:
        out, saved = ops.attention_train_fwd(x, lens32, H, p, seed=7)
        dqkv = ops.attention_train_bwd(x, saved, _cbt(do).to(dev), H, p, seed=7)
        again = ops.attention_train_bwd(x, saved, _cbt(do).to(dev), H, p, seed=7)
    finally:
        ops.CONV_BACKEND["operands"] = "f32"
    assert torch.equal(dqkv, again)
    _close(out.cpu().permute(1, 0, 2), o.detach(), 1e-2)
    got = dqkv.cpu().permute(1, 0, 2)
    for name, sl in (("dq", slice(0, D)), ("dk", slice(D, 2 * D)), ("dv", slice(2 * D, 3 * D))):
        a, w = got[:, sl].double().flatten(), qkv.grad[:, sl].double().flatten()
        cos, ratio = float(torch.dot(a, w) / (a.norm() * w.norm())), float(a.norm() / w.norm())
        assert cos >= 0.999 and 0.99 <= ratio <= 1.01, (name, cos, ratio)


def test_attention_dropout_matches_torch_with_the_same_mask(cuda_device):
    """With p > 0: the mask is element ((b T + q) T + k) of the counter-based stream seeded with seed + head -- the same
    stream evmi_dropout_f32 draws from, so the test reads the mask back through it -- applied to the normalised probabilities
    (scaled by 1 / (1 - p)); forward and backward then equal torch autograd with that mask."""
    from everyvoice_amd.train import ops

    D, H, B, T, p, seed = 64, 2, 2, 48, 0.3, 99
    g = torch.Generator().manual_seed(3)
    dev = cuda_device
    qkv = torch.randn(B, 3 * D, T, generator=g, requires_grad=True)
    lens = torch.tensor([T, T - 7])
    ones = torch.ones(B * T * T, device=dev)
    keep = torch.stack([(ops.dropout(ones, p, seed + h) > 0).float().view(B, T, T) for h in range(H)], dim=1).cpu()  # [B, H, Tq, Tk]
    assert abs(float(keep.mean()) - (1 - p)) < 0.03
    o = _attention_ref(qkv, lens, H, keep, p)
    do = torch.randn(B, D, T, generator=g)
    o.backward(do)
    x = _cbt(qkv.detach()).to(dev)
    lens32 = lens.to(dev, torch.int32)
    out, saved = ops.attention_train_fwd(x, lens32, H, p, seed=seed)
    _close(out.cpu().permute(1, 0, 2), o.detach())
    out2, _ = ops.attention_train_fwd(x, lens32, H, p, seed=seed)
    assert torch.equal(out, out2)  # same seed, same mask
    dqkv = ops.attention_train_bwd(x, saved, _cbt(do).to(dev), H, p, seed=seed)
    _close(dqkv.cpu().permute(1, 0, 2), qkv.grad)


def test_glu_silu_relu_dropout(cuda_device):
    from everyvoice_amd.train import ops

    g = torch.Generator().manual_seed(0)
    dev = cuda_device
    p = torch.randn(2 * 16, 3, 21, generator=g, requires_grad=True)
    dy = torch.randn(16, 3, 21, generator=g)
    F.glu(p, dim=0).backward(dy)
    _close(ops.glu_bwd(p.detach().to(dev), dy.to(dev)), p.grad)
    z = torch.randn(1000, generator=g, requires_grad=True)
    dz = torch.randn(1000, generator=g)
    F.silu(z).backward(dz)
    _close(ops.elementwise(ops.EW_SILU_BWD, dz.to(dev), z.detach().to(dev)), z.grad)
    x = torch.randn(100000, generator=g).to(dev)
    y = ops.dropout(x, 0.25, 7)
    kept = y != 0
    assert abs(float(kept.float().mean()) - 0.75) < 0.01
    _close(y[kept].cpu(), (x[kept] / 0.75).cpu(), 1e-6)
    assert torch.equal(y, ops.dropout(x, 0.25, 7)) and not torch.equal(y, ops.dropout(x, 0.25, 8))


def test_dropout_fused_with_its_neighbours_equals_the_separate_passes(cuda_device):
    """evmi_dropout_fused_f32 (x + s * dropout(h), dropout(silu(h)) and both backwards) against the one-operator kernels with
    the same seed -- same mask, same arithmetic up to the rounding of one fused multiply-add -- at a length that is not a
    multiple of the four elements a thread takes."""
    from everyvoice_amd.train import ops

    g = torch.Generator().manual_seed(1)
    dev = cuda_device
    for n in (4096, 100003):
        a = torch.randn(n, generator=g).to(dev)
        b = torch.randn(n, generator=g).to(dev)
        p, seed = 0.1, 77
        d = ops.dropout(a, p, seed)
        _close(ops.dropout_fused(1, a, b, p, seed, 0.5), (b + 0.5 * d).cpu(), 1e-6)
        _close(ops.dropout_fused(4, a, None, p, seed, 0.5), (0.5 * d).cpu(), 1e-6)
        _close(ops.dropout_fused(2, a, None, p, seed), ops.dropout(ops.elementwise(ops.EW_SILU, a), p, seed).cpu(), 1e-6)
        _close(ops.dropout_fused(3, a, b, p, seed), ops.elementwise(ops.EW_SILU_BWD, d, b).cpu(), 1e-6)
        assert torch.equal(ops.dropout_fused(4, a, None, p, seed, 1.0) == 0, d == 0)  # the same mask


# ---- the whole step ------------------------------------------------------------------------------------------------------
def _ref_cfg(dropout=0.0, speakers=0, default_size=False):
    """``default_size``: the model BASELINE config 3 names and bench.py times (256-dim conformers with 2 x 128 heads and 1024-wide
    feed-forward, 4 + 4 layers, 5-layer variance predictors, 80 mels, 512-channel postnet) instead of the 64-dim test model."""
    c = FastSpeech2ConfigRef() if default_size else FastSpeech2ConfigRef.small()
    c.encoder.dropout = c.decoder.dropout = dropout
    c.duration.dropout = c.pitch.dropout = c.energy.dropout = dropout
    c.n_speakers = speakers
    return c


def _trainer(ref_cfg, cuda_device, seed=11, learn_alignment=False, **kw):
    from everyvoice_amd.train.fs2 import FastSpeech2Trainer
    from tests.test_gpu_fs2 import _product_config

    cfg = _product_config(ref_cfg)
    cfg.learn_alignment = learn_alignment
    for enc, rc in ((cfg.encoder, ref_cfg.encoder), (cfg.decoder, ref_cfg.decoder)):
        enc.dropout = rc.dropout
    for name in ("duration", "pitch", "energy"):
        getattr(cfg.variance_predictors, name).dropout = getattr(ref_cfg, name).dropout
    if ref_cfg.n_speakers:
        cfg.multispeaker, cfg.n_speakers = True, ref_cfg.n_speakers
    tr = FastSpeech2Trainer(cfg, device=cuda_device, seed=seed, **kw)
    g = torch.Generator().manual_seed(seed)
    sd = tr.state_dict()
    for n in tr.params.names():  # livelier than the default init: biases, norms and weight-norm gains that matter
        if n.endswith("bias"):
            sd[n] = torch.randn(sd[n].shape, generator=g) * 0.05
        elif n.endswith("weight_g") or (sd[n].dim() == 1 and n.endswith(".weight")):
            sd[n] = torch.rand(sd[n].shape, generator=g) * 0.5 + 0.75
    tr.load_state_dict(sd)
    return tr


def _train_batch(ref_cfg, B, L, seed):
    g = torch.Generator().manual_seed(seed)
    lens = torch.randint(max(1, L // 2), L + 1, (B,), generator=g)
    lens[0] = L
    pad = torch.arange(L)[None] >= lens[:, None]
    ids = torch.randint(1, ref_cfg.n_symbols, (B, L), generator=g).masked_fill(pad, 0)
    durs = torch.randint(0, 5, (B, L), generator=g)
    durs[:, 0] += 1
    durs = durs.masked_fill(pad, 0)
    mel_lens = durs.sum(1)
    T = int(mel_lens.max())
    mel = torch.randn(B, T, ref_cfg.n_mels, generator=g)
    mel = mel.masked_fill((torch.arange(T)[None] >= mel_lens[:, None])[..., None], 0.0)
    batch = dict(ids=ids, lens=lens, durations=durs, mel=mel, pitch=torch.randn(B, L, generator=g), energy=torch.randn(B, L, generator=g))
    if ref_cfg.n_speakers:
        batch["speakers"] = torch.randint(0, ref_cfg.n_speakers, (B,), generator=g)
    return batch


def _oracle_from(tr, ref_cfg):
    ref = FastSpeech2Ref(ref_cfg).train()
    sd = {k: v.detach().cpu() for k, v in tr.state_dict().items()}
    missing, unexpected = ref.load_state_dict(sd, strict=False)
    assert not missing and not unexpected, (missing, unexpected)
    return ref


def test_training_step_bf16_operands_follow_the_exact_step(cuda_device):
    """precision="bf16" (BASELINE config 3's dtype): the same step -- same parameters, same batch, dropout off -- with bf16
    operands in the dense layers' forward, input gradient and weight gradient, against the exact fp32 step of the same trainer
    class.  The per-operator arithmetic is pinned in test_gpu_train_ops.py (= fp32 product of the rounded operands, 1e-4);
    here: every loss within 2e-2, the flat gradient within 5 % in norm and cosine >= 0.99 (multi-speaker model, so every
    embedding path is on it)."""
    ref_cfg = _ref_cfg(0.0, 3)
    batch = _train_batch(ref_cfg, 4, 23, seed=5)
    out = {}
    for prec in ("f32", "bf16"):
        from everyvoice_amd.train import ops

        tr = _trainer(ref_cfg, cuda_device, precision=prec)
        ops.CONV_BACKEND["operands"] = prec  # what training_step does around forward_backward
        try:
            losses = tr.forward_backward(batch)
        finally:
            ops.CONV_BACKEND["operands"] = "f32"
        out[prec] = ({k: float(v) for k, v in losses.items()}, tr.params.grad.clone())
    for k, v in out["f32"][0].items():
        assert out["bf16"][0][k] == pytest.approx(v, rel=2e-2, abs=1e-4), k
    g32, g16 = out["f32"][1].double(), out["bf16"][1].double()
    cos = float(torch.dot(g32, g16) / (g32.norm() * g16.norm()))
    ratio = float(g16.norm() / g32.norm())
    assert cos >= 0.99 and 0.95 <= ratio <= 1.05, (cos, ratio)
    assert float((g32 - g16).abs().max()) > 0.0  # the bf16 kernels did run


@pytest.mark.parametrize("B,L,speakers", [(3, 14, 0), (2, 23, 3)])
def test_training_step_losses_gradients_and_update(cuda_device, B, L, speakers):
    ref_cfg = _ref_cfg(0.0, speakers)
    tr = _trainer(ref_cfg, cuda_device)
    batch = _train_batch(ref_cfg, B, L, seed=L)
    ref = _oracle_from(tr, ref_cfg)
    want = training_losses_ref(ref, batch)
    want["total"].backward()
    got = tr.forward_backward(batch)
    for k, v in want.items():
        assert float(got[k]) == pytest.approx(float(v), rel=2e-4), k
    grads = tr.params.gradients()
    named = dict(ref.named_parameters())
    assert set(grads) == set(named)
    for name, p in named.items():
        want_g = p.grad if p.grad is not None else torch.zeros_like(p)
        _l2close(grads[name], want_g, 2e-3, name)
    for name, buf in ref.named_buffers():
        if name.endswith("running_mean") or name.endswith("running_var"):
            _close(tr.state_dict()[name], buf, 1e-4)

    # one optimiser step: clip_grad_norm_(1.0) + AdamW at the Noam learning rate of step 1
    o = tr.training.optimizer
    # a bias in front of BatchNorm has a zero gradient in exact arithmetic: Adam turns the rounding noise into +-lr steps
    noise = {n for n, p in named.items() if p.grad is None or float(p.grad.norm()) < 1e-5 * p.numel() ** 0.5}
    assert all(n.endswith(".bias") for n in noise) and len(noise) <= 2 * ref_cfg.encoder.layers + ref_cfg.postnet_layers
    torch.nn.utils.clip_grad_norm_(ref.parameters(), tr.training.gradient_clip_val)
    opt = torch.optim.AdamW(ref.parameters(), lr=tr.learning_rate(1), betas=tuple(o.betas), eps=o.eps, weight_decay=o.weight_decay)
    opt.step()
    tr2 = _trainer(ref_cfg, cuda_device)
    tr2.training_step(batch)
    sd = tr2.state_dict()
    for name, p in ref.named_parameters():
        if name in noise:
            continue
        # Adam's first step moves every entry by ~lr * sign(g): single entries whose gradient is numerically zero may differ in sign
        diff = (sd[name].cpu() - p.detach()).abs()
        assert float((diff > 1e-6 + 2e-3 * tr.learning_rate(1)).float().mean()) < 0.01, name


def test_training_step_phonological_features_input(cuda_device):
    """The bias-free Linear(43 -> d) input layer of target_text_representation_level = "phonological_features" in training: losses
    and every gradient (its own weight included) against torch autograd of the oracle."""
    ref_cfg = _ref_cfg(0.0)
    ref_cfg.target_text_representation_level = "phonological_features"
    tr = _trainer(ref_cfg, cuda_device)
    batch = _train_batch(ref_cfg, 3, 10, seed=4)
    g = torch.Generator().manual_seed(21)
    pad = torch.arange(10)[None] >= batch["lens"][:, None]
    batch["pfs"] = (torch.rand(3, 10, 43, generator=g) < 0.3).float().masked_fill(pad[..., None], 0.0)
    ref = _oracle_from(tr, ref_cfg)
    want = training_losses_ref(ref, batch)
    want["total"].backward()
    got = tr.forward_backward(batch)
    for k, v in want.items():
        assert float(got[k]) == pytest.approx(float(v), rel=2e-4), k
    grads = tr.params.gradients()
    named = dict(ref.named_parameters())
    assert set(grads) == set(named) and tuple(named["text_input_layer.weight"].shape) == (ref_cfg.encoder.input_dim, 43)
    for name, p in named.items():
        _l2close(grads[name], p.grad if p.grad is not None else torch.zeros_like(p), 2e-3, name)


def test_training_step_default_model_size_matches_oracle(cuda_device):
    """BASELINE config 3 at its own model size (multi-speaker, as config 5 has it): every loss, every parameter gradient and
    the BatchNorm statistics of one fp32 step against torch-CPU autograd of the oracle.  Gradient tolerance 4e-3 (L2, relative)
    instead of the small model's 2e-3: the contractions are four times longer (256 / 1024 channels) and the variance
    predictors' first-layer bias gradients are sums of near-cancelling terms (measured 2.7e-3 on the worst one)."""
    ref_cfg = _ref_cfg(0.0, 4, default_size=True)
    tr = _trainer(ref_cfg, cuda_device)
    batch = _train_batch(ref_cfg, 4, 48, seed=48)
    ref = _oracle_from(tr, ref_cfg)
    want = training_losses_ref(ref, batch)
    want["total"].backward()
    got = tr.forward_backward(batch)
    for k, v in want.items():
        assert float(got[k]) == pytest.approx(float(v), rel=2e-4), k
    grads = tr.params.gradients()
    named = dict(ref.named_parameters())
    assert set(grads) == set(named)
    for name, p in named.items():
        _l2close(grads[name], p.grad if p.grad is not None else torch.zeros_like(p), 4e-3, name)
    for name, buf in ref.named_buffers():
        if name.endswith("running_mean") or name.endswith("running_var"):
            _close(tr.state_dict()[name], buf, 1e-4)


def test_training_step_default_model_size_bf16_follows_oracle(cuda_device):
    """The timed arithmetic of config 3 (precision="bf16": bf16 operands, fp32 accumulation / master weights) at the default
    model size against the fp32 oracle: losses within 2e-2, the whole gradient within 5 % in norm and cosine >= 0.99, every
    large parameter tensor's gradient cosine >= 0.97 (per-operator arithmetic is pinned at 1e-4 in test_gpu_train_ops.py)."""
    from everyvoice_amd.train import ops

    ref_cfg = _ref_cfg(0.0, 4, default_size=True)
    tr = _trainer(ref_cfg, cuda_device, precision="bf16")
    batch = _train_batch(ref_cfg, 4, 48, seed=48)
    ref = _oracle_from(tr, ref_cfg)
    want = training_losses_ref(ref, batch)
    want["total"].backward()
    ops.CONV_BACKEND["operands"] = "bf16"  # what training_step does around forward_backward
    try:
        got = tr.forward_backward(batch)
    finally:
        ops.CONV_BACKEND["operands"] = "f32"
    for k, v in want.items():
        assert float(got[k]) == pytest.approx(float(v), rel=2e-2, abs=1e-4), k
    grads = tr.params.gradients()
    flat_g, flat_w = [], []
    for name, p in ref.named_parameters():
        if p.grad is None:
            continue
        g_, w_ = grads[name].cpu().double().flatten(), p.grad.double().flatten()
        flat_g.append(g_)
        flat_w.append(w_)
        if w_.numel() >= 4096 and float(w_.norm()) > 1e-6:
            cos = float(torch.dot(g_, w_) / (g_.norm() * w_.norm() + 1e-300))
            assert cos >= 0.97, f"{name}: cos {cos:.4f}"
    g_, w_ = torch.cat(flat_g), torch.cat(flat_w)
    cos, ratio = float(torch.dot(g_, w_) / (g_.norm() * w_.norm())), float(g_.norm() / w_.norm())
    assert cos >= 0.99 and 0.95 <= ratio <= 1.05, (cos, ratio)


def _bench_batch(B, take=None):
    """The synthetic LJSpeech-shaped batch bench.py's FastSpeech2 training leg times (tools/fs2_bench.py: batch 32, <= 187 symbols,
    <= 947 frames, seed 1234) with given durations; ``take``: its first items only, trimmed to their own longest row."""
    import sys
    from pathlib import Path

    sys.path.insert(0, str(Path(__file__).resolve().parents[1] / "tools"))
    from fs2_train_bench import training_batch

    batch, _ = training_batch(B, seed=1234, learn_alignment=False)
    if take is not None:
        lens = batch["lens"][:take]
        L = int(lens.max())
        T = int(batch["durations"][:take].sum(1).max())
        batch = dict(ids=batch["ids"][:take, :L], lens=lens, durations=batch["durations"][:take, :L], mel=batch["mel"][:take, :T],
                     pitch=batch["pitch"][:take, :L], energy=batch["energy"][:take, :L])
    return batch


def _grad_agreement(got_flat, want_flat):
    g_, w_ = got_flat.double(), want_flat.double()
    return float(torch.dot(g_, w_) / (g_.norm() * w_.norm())), float(g_.norm() / w_.norm())


PER_TENSOR_COS_FLOOR = 0.98  # (measured worst large tensor: text_input_layer.weight 0.9875; whole gradient 0.99983 -- the test prints both)


def test_training_step_at_the_bench_shape_bf16_follows_fp32(cuda_device):
    """BASELINE config 3 at the batch bench.py times (32 utterances, <= 187 symbols, <= 947 frames): the planner's tile and
    split-K choices depend on the column count, so the B = 4 tests do not run the timed kernels (conv_pk_kernel<128, 256>,
    <128, 128> from 1.5 workgroups per CU, the split-K input gradients, wgrad_pk_kernel<4 | 8> over 26 k positions).  Here the
    timed arithmetic (precision="bf16") runs the whole forward + backward at that size against the exact fp32 step of the same
    trainer class on the same parameters and batch -- that fp32 path is pinned against the torch-CPU oracle (tests above), and
    per operator the bench-shape kernels are pinned against torch in test_gpu_train_ops.py.  Every loss within 2e-2, whole
    gradient cosine >= 0.999 and norm within 3 %, every large tensor's gradient cosine >= 0.98 (measured 0.99983 / 0.9875)."""
    from everyvoice_amd.train import ops

    ref_cfg = _ref_cfg(0.0, 0, default_size=True)
    batch = _bench_batch(32)
    assert batch["ids"].shape[0] == 32 and int(batch["durations"].sum(1).max()) > 800
    out = {}
    for prec in ("f32", "bf16"):
        tr = _trainer(ref_cfg, cuda_device, precision=prec)
        ops.CONV_BACKEND["operands"] = prec  # what training_step does around forward_backward
        try:
            losses = tr.forward_backward(batch)
        finally:
            ops.CONV_BACKEND["operands"] = "f32"
        out[prec] = ({k: float(v) for k, v in losses.items()}, {k: v.clone() for k, v in tr.params.gradients().items()}, tr.params.grad.clone())
        del tr
    for k, v in out["f32"][0].items():
        assert math.isfinite(out["bf16"][0][k]) and out["bf16"][0][k] == pytest.approx(v, rel=2e-2, abs=1e-4), k
    cos, ratio = _grad_agreement(out["bf16"][2], out["f32"][2])
    assert cos >= 0.999 and 0.97 <= ratio <= 1.03, (cos, ratio)
    worst = (2.0, None)
    for name, g32 in out["f32"][1].items():
        if g32.numel() >= 4096 and float(g32.norm()) > 1e-6:
            c, _ = _grad_agreement(out["bf16"][1][name].flatten(), g32.flatten())
            worst = min(worst, (c, name))
            assert c >= PER_TENSOR_COS_FLOOR, f"{name}: cos {c:.4f}"
    print(f"bench shape bf16 vs fp32: whole cos {cos:.6f}, worst large tensor {worst}")
    assert float((out["bf16"][2] - out["f32"][2]).abs().max()) > 0.0  # the bf16 kernels did run


def test_training_step_bench_slice_bf16_follows_oracle(cuda_device):
    """The first 8 utterances of the bench batch (full-length rows: up to 187 symbols / 947 frames, 8 x more columns per item
    than the B = 4 / L = 48 oracle test) in precision="bf16" against torch-CPU autograd of the oracle module: losses within
    2e-2, whole gradient cosine >= 0.99, norm within 5 %."""
    from everyvoice_amd.train import ops

    ref_cfg = _ref_cfg(0.0, 0, default_size=True)
    tr = _trainer(ref_cfg, cuda_device, precision="bf16")
    batch = _bench_batch(32, take=8)
    ref = _oracle_from(tr, ref_cfg)
    want = training_losses_ref(ref, batch)
    want["total"].backward()
    ops.CONV_BACKEND["operands"] = "bf16"
    try:
        got = tr.forward_backward(batch)
    finally:
        ops.CONV_BACKEND["operands"] = "f32"
    for k, v in want.items():
        assert float(got[k]) == pytest.approx(float(v), rel=2e-2, abs=1e-4), k
    grads = tr.params.gradients()
    flat_g, flat_w = [], []
    for name, p in ref.named_parameters():
        if p.grad is not None:
            flat_g.append(grads[name].cpu().flatten())
            flat_w.append(p.grad.flatten())
    cos, ratio = _grad_agreement(torch.cat(flat_g), torch.cat(flat_w))
    assert cos >= 0.99 and 0.95 <= ratio <= 1.05, (cos, ratio)


@pytest.mark.parametrize("prec", ["f32", "bf16"])
def test_training_step_at_the_bench_batch_follows_oracle(cuda_device, prec):
    """the WHOLE bench batch (32 utterances x <= 947 frames, default-size model) against torch-CPU autograd of the
    oracle module -- not the trainer's own fp32 step, and not a slice.  fp32: every loss 2e-4, whole gradient within 4e-3 (relative L2);
    bf16: losses 2e-2, whole-gradient cosine >= 0.999 (measured 0.99983), norm within 3 %."""
    from everyvoice_amd.train import ops

    torch.set_num_threads(8)
    ref_cfg = _ref_cfg(0.0, 0, default_size=True)
    tr = _trainer(ref_cfg, cuda_device, precision=prec)
    batch = _bench_batch(32)
    ref = _oracle_from(tr, ref_cfg)
    want = training_losses_ref(ref, batch)
    want["total"].backward()
    ops.CONV_BACKEND["operands"] = prec
    try:
        got = tr.forward_backward(batch)
    finally:
        ops.CONV_BACKEND["operands"] = "f32"
    for k, v in want.items():
        assert float(got[k]) == pytest.approx(float(v), rel=2e-4 if prec == "f32" else 2e-2, abs=1e-5 if prec == "f32" else 1e-4), k
    grads = tr.params.gradients()
    flat_g, flat_w = [], []
    for name, p in ref.named_parameters():
        if p.grad is not None:
            flat_g.append(grads[name].cpu().flatten())
            flat_w.append(p.grad.flatten())
    g, w = torch.cat(flat_g).double(), torch.cat(flat_w).double()
    cos, ratio = _grad_agreement(g, w)
    print(f"bench batch vs oracle [{prec}]: cos {cos:.6f} ratio {ratio:.5f} rel L2 {float((g - w).norm() / w.norm()):.3e}")
    if prec == "f32":
        assert float((g - w).norm() / w.norm()) <= 4e-3, float((g - w).norm() / w.norm())
    else:
        assert cos >= 0.999 and 0.97 <= ratio <= 1.03, (cos, ratio)


@pytest.mark.parametrize("B,T", [(4, 112), (32, 814), (1, 4513)])  # (one item: any length shares its packed operands)
def test_feed_forward_middle_fused_into_the_packs_equals_the_separate_passes(cuda_device, B, T):
    """ops.conv1d_fwd_silu_dropout / conv1d_bwd_silu_dropout_dy (train/fs2.py: ffn_core): dense2(dropout(silu(a))) and its backward with
    the activation and the mask applied while the operands are packed, against the three separate operators on the same seed -- the
    same values are rounded to bf16 either way, so outputs, input gradients and weight gradients agree to summation order; the first
    layer's bias gradient is summed from the ROUNDED output gradient in the fused form (2e-3)."""
    from everyvoice_amd.train import ops

    D, F_, p, seed = 256, 1024, 0.1, 12345
    g = torch.Generator().manual_seed(B + T)
    h = torch.randn(D, B, T, generator=g).to(cuda_device)
    w1 = (torch.randn(F_, D, 1, generator=g) * D ** -0.5).to(cuda_device)
    w2 = (torch.randn(D, F_, 1, generator=g) * F_ ** -0.5).to(cuda_device)
    b1, b2 = torch.randn(F_, generator=g).to(cuda_device) * 0.1, torch.randn(D, generator=g).to(cuda_device) * 0.1
    dy = torch.randn(D, B, T, generator=g).to(cuda_device)
    prev = ops.CONV_BACKEND["operands"]
    ops.CONV_BACKEND["operands"] = "bf16"
    try:
        assert ops.ffn_fused_supported(B, T, F_, D)
        k1, k2 = {}, {}
        a = ops.conv1d_fwd(h, w1, b1, 1, 0, 1, 1, keep=k1)
        s = ops.dropout_fused(2, a, None, p, seed)
        y = ops.conv1d_fwd(s, w2, b2, 1, 0, 1, 1, keep=k2)
        dw1, dw2, db1, db2 = torch.zeros_like(w1), torch.zeros_like(w2), torch.zeros_like(b1), torch.zeros_like(b2)
        ds, _, _ = ops.conv1d_bwd(s, w2, dy, 1, 0, 1, 1, need_dx=True, dw_out=dw2, db_out=db2, accumulate=True, packed=k2)
        da = ops.dropout_fused(3, ds, a, p, seed)
        dh, _, _ = ops.conv1d_bwd(h, w1, da, 1, 0, 1, 1, need_dx=True, dw_out=dw1, db_out=db1, accumulate=True, packed=k1)
        # fused
        f1, f2 = {}, {}
        a_f = ops.conv1d_fwd(h, w1, b1, 1, 0, 1, 1, keep=f1)
        y_f = ops.conv1d_fwd_silu_dropout(a_f, w2, b2, p, seed, f2)
        ew1, ew2, eb1, eb2 = torch.zeros_like(w1), torch.zeros_like(w2), torch.zeros_like(b1), torch.zeros_like(b2)
        ds_f, _, _ = ops.conv1d_bwd(a_f, w2, dy, 1, 0, 1, 1, need_dx=True, dw_out=ew2, db_out=eb2, accumulate=True, packed=f2)
        dh_f = ops.conv1d_bwd_silu_dropout_dy(h, w1, ds_f, a_f, p, seed, ew1, eb1, f1)
        ops.wgrad_join(cuda_device)
        torch.cuda.synchronize()
    finally:
        ops.CONV_BACKEND["operands"] = prev
    rel = lambda got, want: float((got - want).abs().max() / want.abs().max())  # noqa: E731
    assert float((s == 0).float().mean()) == pytest.approx(p, abs=0.01)  # (the mask is there)
    for name, got, want, tol in (("y", y_f, y, 2e-6), ("ds", ds_f, ds, 2e-6), ("dh", dh_f, dh, 2e-6), ("dw2", ew2, dw2, 2e-6), ("dw1", ew1, dw1, 2e-5),
                                 ("db2", eb2, db2, 2e-6), ("db1", eb1, db1, 2e-3)):
        assert rel(got, want) <= tol, (name, rel(got, want))


@pytest.mark.parametrize("B,T,cin,cout,sb", [(32, 814, 256, 256, 1.0), (4, 112, 1024, 256, 0.5), (1, 4513, 256, 256, 1.0)])
def test_residual_add_and_dropout_in_the_dense_layers_epilogue_equal_the_separate_passes(cuda_device, B, T, cin, cout, sb):
    """ops.conv1d_fwd_resdrop / conv1d_bwd_dropout_dy (train/fs2.py: dense_residual_dropout, ffn_core with a residual): a + sb *
    dropout(dense(h)) with the add and the mask in the layer's epilogue, and sb * dropout(dy) formed while dy is packed, against
    dense -> evmi_dropout_fused_f32 (modes 1 / 4) -> the layer's backward on the same seed.  The same values are rounded to bf16 either
    way: outputs and the three gradients agree to rounding of one fused scale (2e-6); the bias gradient is summed from the ROUNDED
    gradient in the fused form (2e-3).  Also against torch with the exported mask: y = a + sb * keep / (1 - p) * (W bf16(h) + b)."""
    from everyvoice_amd.train import ops

    p, seed = 0.2, 977
    g = torch.Generator().manual_seed(B + T + cin)
    h = torch.randn(cin, B, T, generator=g).to(cuda_device)
    a = torch.randn(cout, B, T, generator=g).to(cuda_device)
    w = (torch.randn(cout, cin, 1, generator=g) * cin ** -0.5).to(cuda_device)
    b = (torch.randn(cout, generator=g) * 0.1).to(cuda_device)
    dy = torch.randn(cout, B, T, generator=g).to(cuda_device)
    prev = ops.CONV_BACKEND["operands"]
    ops.CONV_BACKEND["operands"] = "bf16"
    try:
        assert ops.resdrop_fused_supported(B, T, cin, cout)
        k1, k2 = {}, {}
        z = ops.conv1d_fwd(h, w, b, 1, 0, 1, 1, keep=k1)
        y = ops.dropout_fused(1, z, a, p, seed, sb)
        dz = ops.dropout_fused(4, dy, None, p, seed, sb)
        dw, db, ew, eb = torch.zeros_like(w), torch.zeros_like(b), torch.zeros_like(w), torch.zeros_like(b)
        dh, _, _ = ops.conv1d_bwd(h, w, dz, 1, 0, 1, 1, need_dx=True, dw_out=dw, db_out=db, accumulate=True, packed=k1)
        y_f = ops.conv1d_fwd_resdrop(h, w, b, a, p, seed, sb, k2)
        dh_f = ops.conv1d_bwd_dropout_dy(h, w, dy, p, seed, sb, ew, eb, k2)
        keep = (ops.dropout(torch.ones(cout * B * T, device=cuda_device), p, seed) > 0).float().view(cout, B * T).cpu()
        ops.wgrad_join(cuda_device)
        torch.cuda.synchronize()
    finally:
        ops.CONV_BACKEND["operands"] = prev
    rel = lambda got, want: float((got - want).abs().max() / want.abs().max())  # noqa: E731
    for name, got, want, tol in (("y", y_f, y, 2e-6), ("dh", dh_f, dh, 2e-6), ("dw", ew, dw, 2e-6), ("db", eb, db, 2e-3)):
        assert rel(got, want) <= tol, (name, rel(got, want))
    bf = lambda t: t.to(torch.bfloat16).to(torch.float32)  # noqa: E731
    z_ref = bf(w.cpu().view(cout, cin)) @ bf(h.cpu().view(cin, B * T)) + b.cpu()[:, None]
    y_ref = a.cpu().view(cout, B * T) + sb * keep / (1 - p) * z_ref
    assert float((y_f.cpu().view(cout, B * T) - y_ref).abs().max() / y_ref.abs().max()) <= 2e-5
    dz_ref = bf(sb * (keep * dy.cpu().view(cout, B * T) / (1 - p)))
    dh_ref = bf(w.cpu().view(cout, cin)).t() @ dz_ref
    # (a gradient within fp32 rounding of a bf16 boundary rounds the other way on one side: 4e-3 of that ONE term of a 256-term sum)
    assert float((dh_f.cpu().view(cin, B * T) - dh_ref).norm() / dh_ref.norm()) <= 1e-3


def test_fused_feed_forward_block_against_torch_with_the_kernels_own_mask(cuda_device):
    """the fused feed-forward block (train/fs2.py: ffn_core = LayerNorm written packed -> dense1 -> SiLU + dropout
    applied while dense2's input is packed -> dense2; backward with dropout(ds) * silu'(a) applied while dense1's output gradient is
    packed) at the BENCH shape (32 x 814 columns, 256 -> 1024 -> 256) against TORCH -- not against the library's own separate passes --
    with the kernels' mask exported through the one-operator dropout kernel (same counter-based stream: element i of the [1024, B, T]
    tensor) and the path's rounding points restated: every matrix operand is rounded to bf16 (LayerNorm output, dropout(silu(a)), both
    output gradients, the weights), products accumulate in fp32, the first layer's bias gradient sums the ROUNDED gradient.
    Tolerances: a value that lands within summation-order noise of a bf16 rounding boundary rounds the other way on one side (4e-3 of
    that element); relative L2 <= 1e-3 per tensor (measured 1e-4 .. 4e-4) and every element within 1 % of the tensor's largest."""
    from everyvoice_amd.train import ops

    D, F_, B, T, p, seed = 256, 1024, 32, 814, 0.2, 4321
    g = torch.Generator().manual_seed(11)
    x = torch.randn(D, B, T, generator=g) * 1.3 + 0.2
    gamma, beta = torch.rand(D, generator=g) + 0.5, torch.randn(D, generator=g) * 0.1
    w1 = torch.randn(F_, D, 1, generator=g) * D ** -0.5
    w2 = torch.randn(D, F_, 1, generator=g) * F_ ** -0.5
    b1, b2 = torch.randn(F_, generator=g) * 0.1, torch.randn(D, generator=g) * 0.1
    dy = torch.randn(D, B, T, generator=g)
    dev = cuda_device
    bf = lambda t: t.to(torch.bfloat16).to(torch.float32)  # noqa: E731
    prev = ops.CONV_BACKEND["operands"]
    ops.CONV_BACKEND["operands"] = "bf16"
    try:
        assert ops.ffn_fused_supported(B, T, F_, D) and ops.ln_dense_fused_supported(B, T, D, F_)
        xd, w1d, w2d = x.to(dev), w1.to(dev), w2.to(dev)
        gd, bd = gamma.to(dev), beta.to(dev)
        keep = (ops.dropout(torch.ones(F_ * B * T, device=dev), p, seed) > 0).float().view(F_, B * T).cpu()
        f1, f2 = {}, {}
        a = ops.layernorm_dense_fwd(xd, gd, bd, w1d, b1.to(dev), f1)
        y = ops.conv1d_fwd_silu_dropout(a, w2d, b2.to(dev), p, seed, f2)
        ew1, ew2, eb1, eb2 = torch.zeros_like(w1d), torch.zeros_like(w2d), torch.zeros(F_, device=dev), torch.zeros(D, device=dev)
        dgam, dbet = torch.zeros(D, device=dev), torch.zeros(D, device=dev)
        ds, _, _ = ops.conv1d_bwd(a, w2d, dy.to(dev), 1, 0, 1, 1, need_dx=True, dw_out=ew2, db_out=eb2, accumulate=True, packed=f2, x_standin=True)
        dh = ops.conv1d_bwd_silu_dropout_dy(xd, w1d, ds, a, p, seed, ew1, eb1, f1)
        dx = ops.layernorm_bwd(xd, gd, dh, dgam, dbet)
        ops.wgrad_join(dev)
        torch.cuda.synchronize()
    finally:
        ops.CONV_BACKEND["operands"] = prev
    assert float(keep.mean()) == pytest.approx(1 - p, abs=0.005)
    # ---- torch, with the rounding points of the packed path
    N = B * T
    xm = x.view(D, N)
    mu, var = xm.mean(0, keepdim=True), xm.var(0, unbiased=False, keepdim=True)
    xhat = (xm - mu) * torch.rsqrt(var + 1e-5)
    hn = bf(xhat * gamma[:, None] + beta[:, None])
    W1, W2 = bf(w1.view(F_, D)), bf(w2.view(D, F_))
    a_ref = W1 @ hn + b1[:, None]
    sig = torch.sigmoid(a_ref)
    s_ref = bf(a_ref * sig * keep / (1 - p))
    y_ref = W2 @ s_ref + b2[:, None]
    dyr = bf(dy.view(D, N))
    ds_ref = W2.t() @ dyr
    dw2_ref, db2_ref = dyr @ s_ref.t(), dy.view(D, N).sum(1)
    da_ref = bf(ds_ref * keep / (1 - p) * (sig * (1 + a_ref * (1 - sig))))
    dh_ref = W1.t() @ da_ref
    dw1_ref, db1_ref = da_ref @ hn.t(), da_ref.sum(1)
    dxhat = dh_ref * gamma[:, None]
    dx_ref = torch.rsqrt(var + 1e-5) * (dxhat - dxhat.mean(0, keepdim=True) - xhat * (dxhat * xhat).mean(0, keepdim=True))
    dgam_ref, dbet_ref = (dh_ref * xhat).sum(1), dh_ref.sum(1)
    for name, got, want in (("a", a, a_ref), ("y", y, y_ref), ("ds", ds, ds_ref), ("dh", dh, dh_ref), ("dx", dx, dx_ref), ("dw2", ew2, dw2_ref), ("dw1", ew1, dw1_ref),
                            ("db2", eb2, db2_ref), ("db1", eb1, db1_ref), ("dgamma", dgam, dgam_ref), ("dbeta", dbet, dbet_ref)):
        got = got.cpu().reshape(want.shape)
        l2 = float((got - want).norm() / want.norm())
        mx = float((got - want).abs().max() / want.abs().max())
        print(f"fused feed-forward vs torch, {name}: rel L2 {l2:.2e}, max {mx:.2e}")
        assert l2 <= 1e-3 and mx <= 1e-2, (name, l2, mx)


@pytest.mark.parametrize("B,T", [(32, 814), (4, 112), (2, 32), (1, 4513), (1, 77)])  # (one item: the row pitch is rounded up to 64)
def test_feed_forward_block_as_a_packed_chain_against_torch_with_the_kernels_own_masks(cuda_device, B, T):
    """The feed-forward block with its 1024-channel tensors packed end to end (train/fs2.py: ffn_core -> ops.ffn_packed_fwd / _bwd:
    LayerNorm written packed -> dense1, whose epilogue writes bf16(a) and the packed dropout(silu(a)) -> dense2 with the residual add and
    the outer dropout in its epilogue; backward: scale * dropout(dy) packed -> dense2's input gradient, whose epilogue writes the packed
    dropout(ds) * silu'(bf16(a)) -> dense1's input gradient -> LayerNorm backward) against TORCH with both masks exported through the
    one-operator dropout kernel and the chain's rounding points restated (bench shape, and two small ones that take the split-K
    launches whose reduce pass carries the same tails).  Tolerances as the test above: relative L2 <= 1e-3, elements within 1 %."""
    from everyvoice_amd.train import ops

    D, F_, p, seed, seed_out, sb = 256, 1024, 0.2, 4321, 977, 0.5
    g = torch.Generator().manual_seed(11 + B)
    x = torch.randn(D, B, T, generator=g) * 1.3 + 0.2
    gamma, beta = torch.rand(D, generator=g) + 0.5, torch.randn(D, generator=g) * 0.1
    w1 = torch.randn(F_, D, 1, generator=g) * D ** -0.5
    w2 = torch.randn(D, F_, 1, generator=g) * F_ ** -0.5
    b1, b2 = torch.randn(F_, generator=g) * 0.1, torch.randn(D, generator=g) * 0.1
    dy = torch.randn(D, B, T, generator=g)
    dev = cuda_device
    bf = lambda t: t.to(torch.bfloat16).to(torch.float32)  # noqa: E731
    prev = ops.CONV_BACKEND["operands"]
    ops.CONV_BACKEND["operands"] = "bf16"
    try:
        assert ops.ffn_packed_supported(B, T, D, F_, D)
        xd, w1d, w2d = x.to(dev), w1.to(dev), w2.to(dev)
        gd, bd = gamma.to(dev), beta.to(dev)
        keep1 = (ops.dropout(torch.ones(F_ * B * T, device=dev), p, seed) > 0).float().view(F_, B * T).cpu()
        keep2 = (ops.dropout(torch.ones(D * B * T, device=dev), p, seed_out) > 0).float().view(D, B * T).cpu()
        kp = {}
        y = ops.ffn_packed_fwd(xd, gd, bd, w1d, b1.to(dev), w2d, b2.to(dev), xd, p, seed, seed_out, sb, kp)
        ew1, ew2, eb1, eb2 = torch.zeros_like(w1d), torch.zeros_like(w2d), torch.zeros(F_, device=dev), torch.zeros(D, device=dev)
        dgam, dbet = torch.zeros(D, device=dev), torch.zeros(D, device=dev)
        dh = ops.ffn_packed_bwd(xd, w1d, w2d, dy.to(dev), p, seed, seed_out, sb, ew1, eb1, ew2, eb2, kp)
        dx = ops.layernorm_bwd(xd, gd, dh, dgam, dbet)
        ops.wgrad_join(dev)
        torch.cuda.synchronize()
        a_pk = kp["a_pk"].view(torch.bfloat16).view(F_ // 8, ops.pk_pitch(B, T), 8)[:, : B * T].permute(0, 2, 1).reshape(F_, B * T).float().cpu()
    finally:
        ops.CONV_BACKEND["operands"] = prev
    N = B * T
    xm = x.view(D, N)
    mu, var = xm.mean(0, keepdim=True), xm.var(0, unbiased=False, keepdim=True)
    xhat = (xm - mu) * torch.rsqrt(var + 1e-5)
    hn = bf(xhat * gamma[:, None] + beta[:, None])
    W1, W2 = bf(w1.view(F_, D)), bf(w2.view(D, F_))
    a_ref = W1 @ hn + b1[:, None]
    s_ref = bf(a_ref * torch.sigmoid(a_ref) * keep1 / (1 - p))
    y_ref = xm + sb * keep2 / (1 - p) * (W2 @ s_ref + b2[:, None])
    dz = bf(sb * keep2 / (1 - p) * dy.view(D, N))
    ds_ref = W2.t() @ dz
    dw2_ref, db2_ref = dz @ s_ref.t(), dz.sum(1)
    ab = bf(a_ref)  # the backward's silu' is taken at the stored bf16 pre-activation
    sig = torch.sigmoid(ab)
    da_ref = bf(ds_ref * keep1 / (1 - p) * (sig * (1 + ab * (1 - sig))))
    dh_ref = W1.t() @ da_ref
    dw1_ref, db1_ref = da_ref @ hn.t(), da_ref.sum(1)
    dxhat = dh_ref * gamma[:, None]
    dx_ref = torch.rsqrt(var + 1e-5) * (dxhat - dxhat.mean(0, keepdim=True) - xhat * (dxhat * xhat).mean(0, keepdim=True))
    dgam_ref, dbet_ref = (dh_ref * xhat).sum(1), dh_ref.sum(1)
    for name, got, want in (("a_pk", a_pk, ab), ("y", y, y_ref), ("dh", dh, dh_ref), ("dx", dx, dx_ref), ("dw2", ew2, dw2_ref), ("dw1", ew1, dw1_ref),
                            ("db2", eb2, db2_ref), ("db1", eb1, db1_ref), ("dgamma", dgam, dgam_ref), ("dbeta", dbet, dbet_ref)):
        got = got.cpu().reshape(want.shape)
        l2 = float((got - want).norm() / want.norm())
        mx = float((got - want).abs().max() / want.abs().max())
        print(f"packed feed-forward chain vs torch ({B} x {T}), {name}: rel L2 {l2:.2e}, max {mx:.2e}")
        assert l2 <= 1e-3 and mx <= 1e-2, (name, l2, mx)


@pytest.mark.parametrize("B,T,C,F_", [(4, 112, 256, 1024), (32, 814, 256, 768), (8, 64, 128, 256), (1, 4513, 256, 1024)])
def test_layernorm_written_as_the_packed_input_of_the_dense_layer_behind_it(cuda_device, B, T, C, F_):
    """ops.layernorm_dense_fwd (train/fs2.py: ln_dense, ffn_core): LayerNorm -> pointwise layer with the normalised tensor written
    straight into the layer's packed bf16 input, against LayerNorm then the layer (which packs the same values): same output, and the
    layer's backward from the kept packed copy gives the same input / weight / bias gradients."""
    from everyvoice_amd.train import ops

    g = torch.Generator().manual_seed(B + T + C)
    x = (torch.randn(C, B, T, generator=g) * 1.7 + 0.3).to(cuda_device)
    gamma, beta = (torch.rand(C, generator=g) + 0.5).to(cuda_device), (torch.randn(C, generator=g) * 0.1).to(cuda_device)
    w = (torch.randn(F_, C, 1, generator=g) * C ** -0.5).to(cuda_device)
    b = (torch.randn(F_, generator=g) * 0.1).to(cuda_device)
    dy = torch.randn(F_, B, T, generator=g).to(cuda_device)
    prev = ops.CONV_BACKEND["operands"]
    ops.CONV_BACKEND["operands"] = "bf16"
    try:
        assert ops.ln_dense_fused_supported(B, T, C, F_)
        k1, k2 = {}, {}
        h = ops.layernorm(x, gamma, beta)
        y = ops.conv1d_fwd(h, w, b, 1, 0, 1, 1, keep=k1)
        y_f = ops.layernorm_dense_fwd(x, gamma, beta, w, b, k2)
        dw, db, ew, eb = torch.zeros_like(w), torch.zeros_like(b), torch.zeros_like(w), torch.zeros_like(b)
        dh, _, _ = ops.conv1d_bwd(h, w, dy, 1, 0, 1, 1, need_dx=True, dw_out=dw, db_out=db, accumulate=True, packed=k1)
        dh_f, _, _ = ops.conv1d_bwd(x, w, dy, 1, 0, 1, 1, need_dx=True, dw_out=ew, db_out=eb, accumulate=True, packed=k2)
        ops.wgrad_join(cuda_device)
        torch.cuda.synchronize()
    finally:
        ops.CONV_BACKEND["operands"] = prev
    rel = lambda got, want: float((got - want).abs().max() / want.abs().max())  # noqa: E731
    for name, got, want in (("y", y_f, y), ("dh", dh_f, dh), ("dw", ew, dw), ("db", eb, db)):
        assert rel(got, want) <= 2e-6, (name, rel(got, want))


def test_bench_size_steps_are_reproducible_run_to_run(cuda_device):
    """Two trainers in lockstep on the bench batch (default-size model, precision="bf16", dropout on): 120 steps each, the parameters
    bitwise equal after every one.  Nothing in the step is order-dependent (no atomics, fixed-order reductions), so any difference is a
    race.  This is the test that would have caught the dK/dV attention kernel reading its LDS-direct dO tile with no vmcnt wait (a
    stale tile about once per 60 steps at this size -- common.h: lds_dma_barrier): with that bug the chance of 120 clean steps is ~2 %."""
    ref_cfg = _ref_cfg(0.1, 0, default_size=True)
    batch = _bench_batch(32)
    a = _trainer(ref_cfg, cuda_device, precision="bf16")
    b = _trainer(ref_cfg, cuda_device, precision="bf16")
    assert torch.equal(a.params.flat, b.params.flat)
    for step in range(120):
        la, lb = a.training_step(batch), b.training_step(batch)
        assert torch.equal(a.params.flat, b.params.flat), f"parameters differ after step {step}"
        assert all(torch.equal(la[k], lb[k]) for k in la), f"losses differ at step {step}"


def test_noam_schedule(cuda_device):
    tr = _trainer(_ref_cfg(), cuda_device)
    o = tr.training.optimizer
    assert tr.learning_rate(o.warmup_steps) == pytest.approx(o.learning_rate)
    assert tr.learning_rate(1) == pytest.approx(o.learning_rate / o.warmup_steps)
    assert tr.learning_rate(4 * o.warmup_steps) == pytest.approx(o.learning_rate / 2)


def test_training_reduces_the_loss_and_inference_loads_the_result(cuda_device):
    from everyvoice_amd.fs2 import FastSpeech2
    from everyvoice_amd.train.fs2 import FastSpeech2TrainingConfig, NoamOptimizerConfig

    ref_cfg = _ref_cfg(0.1)  # with dropout on
    tr = _trainer(ref_cfg, cuda_device, training=FastSpeech2TrainingConfig(optimizer=NoamOptimizerConfig(learning_rate=2e-3, warmup_steps=5)))
    batch = _train_batch(ref_cfg, 4, 16, seed=2)
    first = float(tr.training_step(batch)["total"])
    for _ in range(25):
        last = float(tr.training_step(batch)["total"])
    assert math.isfinite(last) and last < 0.7 * first, (first, last)
    model = FastSpeech2(tr.config, device=cuda_device).load_state_dict(tr.state_dict())
    mel, post, dur, pitch, energy, mel_lens = model(batch["ids"], batch["lens"], durations=batch["durations"])
    assert torch.isfinite(post).all() and post.shape[0] == 4 and int(mel_lens.max()) == post.shape[1]
    # the same state in the oracle (eval mode: running statistics) gives the same mel
    ref = FastSpeech2Ref(ref_cfg).eval()
    ref.load_state_dict({k: v.cpu() for k, v in tr.state_dict().items()}, strict=False)
    want = ref(batch["ids"], batch["lens"], durations=batch["durations"])
    _close(post, want[1], 5e-4)


def test_checkpoint_resume_is_bitwise(cuda_device):
    ref_cfg = _ref_cfg(0.1)
    batch = _train_batch(ref_cfg, 2, 12, seed=4)
    a = _trainer(ref_cfg, cuda_device)
    a.training_step(batch)
    ck = a.checkpoint()
    assert ck["model_info"] == {"name": "FastSpeech2", "version": "1.0"}
    import json
    json.dumps(ck["hyper_parameters"])  # JSON-only, path-free
    a.training_step(batch)
    b = _trainer(ref_cfg, cuda_device).load_checkpoint(ck)
    b.training_step(batch)
    sa, sb = a.state_dict(), b.state_dict()
    for k in sa:  # every reduction of the step has a fixed order (no atomics): resuming is bitwise
        assert torch.equal(sa[k], sb[k]), k
    with pytest.raises(TypeError, match="Wrong model type"):
        b.load_checkpoint({"model_info": {"name": "HiFiGAN"}})


def test_data_parallel_path_on_rccl_world_of_one(cuda_device):
    """The N > 1 code path (RCCL all-reduce of the flat gradient buffer + 1/world scaling) on a one-rank "nccl" group equals the
    single-GPU step bit for bit."""
    import os
    import socket

    import torch.distributed as dist

    ref_cfg = _ref_cfg(0.1)
    batch = _train_batch(ref_cfg, 2, 12, seed=9)
    plain = _trainer(ref_cfg, cuda_device)
    plain.training_step(batch)
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=cuda_device)
    try:
        dp = _trainer(ref_cfg, cuda_device, process_group=True)
        dp.training_step(batch)
        torch.cuda.synchronize()
    finally:
        dist.destroy_process_group()
    sa, sb = plain.state_dict(), dp.state_dict()
    for k in sa:
        assert torch.equal(sa[k], sb[k]), k


# ---- alignment learning (learn_alignment: true, the reference's default) -------------------------------------------------------
def test_forward_sum_loss_gradient(cuda_device):
    from everyvoice_amd.train import ops
    from oracle.alignment_ref import forward_sum_loss_ref

    g = torch.Generator().manual_seed(21)
    B, T, L = 3, 40, 9
    text_lens, mel_lens = torch.tensor([9, 5, 7]), torch.tensor([40, 22, 31])
    lp = (torch.randn(B, T, L, generator=g) * 2).requires_grad_()
    want = 0.1 * forward_sum_loss_ref(lp, text_lens, mel_lens)
    want.backward()
    dev = cuda_device
    loss, grad = ops.forward_sum_loss_and_grad(lp.detach().to(dev), text_lens.to(dev, torch.int32), mel_lens.to(dev, torch.int32), 0.1)
    assert float(loss) == pytest.approx(float(want), rel=2e-5)
    _close(grad, lp.grad, 2e-4)
    # an item with fewer frames than symbols is infeasible: zero loss and zero gradient (zero_infinity)
    loss2, grad2 = ops.forward_sum_loss_and_grad(lp.detach().to(dev), torch.tensor([9, 5, 7], dtype=torch.int32, device=dev),
                                                 torch.tensor([40, 3, 31], dtype=torch.int32, device=dev), 1.0)
    assert torch.isfinite(loss2).all() and float(grad2[1].abs().sum()) == 0.0


@pytest.mark.parametrize("with_prior", [True, False])
def test_alignment_attention_backward(cuda_device, with_prior):
    from everyvoice_amd.heavy import BetaBinomialInterpolator, maximum_path
    from everyvoice_amd.train import ops
    from oracle.alignment_ref import alignment_attention_ref, binarization_loss_ref, forward_sum_loss_ref

    g = torch.Generator().manual_seed(5)
    A, B, T, L = 12, 2, 26, 7
    text_lens, mel_lens = torch.tensor([7, 5]), torch.tensor([26, 17])
    q = torch.randn(B, A, T, generator=g, requires_grad=True)
    k = torch.randn(B, A, L, generator=g, requires_grad=True)
    dev = cuda_device
    prior = None
    if with_prior:
        interp = BetaBinomialInterpolator(device=dev)
        prior = torch.zeros(B, T, L, dtype=torch.float64)
        for b in range(B):
            prior[b, : mel_lens[b], : text_lens[b]] = interp(int(mel_lens[b]), int(text_lens[b])).cpu()
    temp = 0.05
    soft, logprob = alignment_attention_ref(q, k, text_lens, prior, temp)
    tl, ml = text_lens.to(dev, torch.int32), mel_lens.to(dev, torch.int32)
    qd, kd = _cbt(q.detach()).to(dev), _cbt(k.detach()).to(dev)
    pd = None if prior is None else prior.to(dev)
    soft_g, logprob_g = ops.align_attention_fwd(qd, kd, pd, tl, temp)
    _close(soft_g, soft.detach())
    hard, dur = maximum_path(torch.log(soft_g), ml, tl)
    hard_c = hard.cpu()
    loss = 0.3 * forward_sum_loss_ref(logprob, text_lens, mel_lens) + 0.7 * binarization_loss_ref(hard_c, soft)
    loss.backward()
    _, dlogprob = ops.forward_sum_loss_and_grad(logprob_g, tl, ml, 0.3)
    dq, dk = ops.align_attention_bwd(qd, kd, soft_g, logprob_g, pd, hard, dlogprob, tl, temp, 0.7 / float(mel_lens.sum()))
    _l2close(dq.cpu().permute(1, 0, 2), q.grad, 1e-3, "dq")
    _l2close(dk.cpu().permute(1, 0, 2), k.grad, 1e-3, "dk")


def _align_batch(ref_cfg, B, L, seed, dev):
    from everyvoice_amd.heavy import BetaBinomialInterpolator

    batch = _train_batch(ref_cfg, B, L, seed)
    mel_lens = batch["durations"].sum(1)
    T = int(mel_lens.max())
    g = torch.Generator().manual_seed(seed + 100)
    interp = BetaBinomialInterpolator(device=dev)
    prior = torch.zeros(B, T, L, dtype=torch.float64)
    for b in range(B):
        prior[b, : mel_lens[b], : batch["lens"][b]] = interp(int(mel_lens[b]), int(batch["lens"][b])).cpu()
    out = dict(ids=batch["ids"], lens=batch["lens"], mel=batch["mel"], mel_lens=mel_lens, attn_prior=prior,
               pitch_frames=torch.randn(B, T, generator=g), energy_frames=torch.randn(B, T, generator=g))
    return out


def test_training_step_with_alignment_learning(cuda_device):
    """learn_alignment: true -- the aligner's soft attention, the hard path (monotonic search), durations and frame-averaged
    pitch / energy targets derived from it, CTC + binarisation losses and every gradient (aligner projections included)."""
    from oracle.alignment_ref import AlignerRef

    ref_cfg = _ref_cfg(0.0)
    tr = _trainer(ref_cfg, cuda_device, learn_alignment=True)
    tr.current_epoch = 50  # half of the binarisation warm-up: weight 0.05
    batch = _align_batch(ref_cfg, 3, 10, seed=6, dev=cuda_device)
    sd = {k: v.detach().cpu() for k, v in tr.state_dict().items()}
    ref = FastSpeech2Ref(ref_cfg).train()
    ref.load_state_dict({k: v for k, v in sd.items() if not k.startswith("attention.")}, strict=True)
    aligner = AlignerRef(ref_cfg.encoder.input_dim, ref_cfg.n_mels)
    aligner.load_state_dict({k[len("attention."):]: v for k, v in sd.items() if k.startswith("attention.")}, strict=True)
    got = tr.forward_backward(batch)
    hard_gpu = tr.last_alignment.cpu()
    want = training_losses_ref(ref, batch, weights={"attn_bin": 0.05}, aligner=aligner)
    want_hard = training_losses_ref(ref, batch, weights={"attn_bin": 0.05}, aligner=aligner, hard=hard_gpu)
    assert float(want_hard["total"]) == pytest.approx(float(want["total"]), rel=1e-6)  # the oracle's own search finds the same path
    want["total"].backward()
    assert set(got) == set(want)
    for k, v in want.items():
        assert float(got[k]) == pytest.approx(float(v), rel=2e-4), k
    grads = tr.params.gradients()
    named = dict(ref.named_parameters())
    named.update({"attention." + n: p for n, p in aligner.named_parameters()})
    assert set(grads) == set(named)
    for name, p in named.items():
        want_g = p.grad if p.grad is not None else torch.zeros_like(p)
        _l2close(grads[name], want_g, 2e-3, name)


def test_alignment_learning_trains(cuda_device):
    from everyvoice_amd.train.fs2 import FastSpeech2TrainingConfig, NoamOptimizerConfig

    ref_cfg = _ref_cfg(0.1)
    tr = _trainer(ref_cfg, cuda_device, learn_alignment=True,
                  training=FastSpeech2TrainingConfig(optimizer=NoamOptimizerConfig(learning_rate=2e-3, warmup_steps=5)))
    batch = _align_batch(ref_cfg, 4, 12, seed=3, dev=cuda_device)
    first = tr.training_step(batch)
    for _ in range(20):
        last = tr.training_step(batch)
    assert float(last["total"]) < 0.8 * float(first["total"])
    assert float(last["attn_ctc"]) <= float(first["attn_ctc"]) * 1.001
    assert int(tr.last_alignment.sum()) == int(batch["mel_lens"].sum())  # one symbol per frame


# ---- HIP-graph execution of the step ----------------------------------------------------------------------------------------
def _graph_pair(cuda_device, learn_alignment, precision="f32", **kw):
    ref_cfg = _ref_cfg(0.1, 0)
    return (_trainer(ref_cfg, cuda_device, learn_alignment=learn_alignment, precision=precision, use_graph=False, **kw),
            _trainer(ref_cfg, cuda_device, learn_alignment=learn_alignment, precision=precision, use_graph=True, **kw), ref_cfg)


def _shaped_batch(ref_cfg, seed, learn_alignment, dev, B=4, L=19, T=60):
    """A batch whose padded shape is exactly (B, L, T) whatever the seed: item 0 has L symbols and T frames, the others fewer."""
    g = torch.Generator().manual_seed(seed)
    lens = torch.randint(L // 2, L, (B,), generator=g)
    lens[0] = L
    pad = torch.arange(L)[None] >= lens[:, None]
    ids = torch.randint(1, ref_cfg.n_symbols, (B, L), generator=g).masked_fill(pad, 0)
    durs = torch.randint(1, 4, (B, L), generator=g).masked_fill(pad, 0)
    durs[0, L - 1] = T - int(durs[0, : L - 1].sum())
    mel_lens = durs.sum(1)
    assert int(mel_lens.max()) == T and int(durs.min()) >= 0 and int(durs[0, L - 1]) > 0
    mel = torch.randn(B, T, ref_cfg.n_mels, generator=g).masked_fill((torch.arange(T)[None] >= mel_lens[:, None])[..., None], 0.0)
    if not learn_alignment:
        return dict(ids=ids, lens=lens, durations=durs, mel=mel, pitch=torch.randn(B, L, generator=g), energy=torch.randn(B, L, generator=g))
    from everyvoice_amd.heavy import BetaBinomialInterpolator

    interp = BetaBinomialInterpolator(device=dev)
    prior = torch.zeros(B, T, L, dtype=torch.float64)
    for b in range(B):
        prior[b, : mel_lens[b], : lens[b]] = interp(int(mel_lens[b]), int(lens[b])).cpu()
    return dict(ids=ids, lens=lens, mel=mel, mel_lens=mel_lens, attn_prior=prior, pitch_frames=torch.randn(B, T, generator=g),
                energy_frames=torch.randn(B, T, generator=g))


@pytest.mark.parametrize("learn_alignment,precision", [(False, "f32"), (True, "f32"), (True, "bf16")])
def test_graph_replays_equal_eager_steps_bitwise(cuda_device, learn_alignment, precision):
    """use_graph=True: two eager steps, then the step is captured and every later one is a replay.  With dropout ON the replays
    must draw new masks each step, follow the Noam schedule and the per-batch token / frame counts -- all of which live on the
    device -- and stay bit for bit on the eager trainer's trajectory: parameters, optimiser moments, BatchNorm statistics and
    losses after six steps over three different batches of one padded shape."""
    eager, graph, ref_cfg = _graph_pair(cuda_device, learn_alignment, precision)
    batches = [_shaped_batch(ref_cfg, seed, learn_alignment, cuda_device) for seed in (5, 6, 7)]
    assert len({int(b["lens"].sum()) for b in batches}) >= 2  # different counts inside one padded shape
    used = []
    for step in range(6):
        b = batches[step % 3]
        le, lg = eager.training_step(b), graph.training_step(b)
        used.append(graph.last_step_was_graph)
        for k in le:
            assert torch.equal(le[k], lg[k]), (step, k, float(le[k]), float(lg[k]))
    assert graph._graph_failed is None and used == [False, False, True, True, True, True] and len(graph._graphs) == 1
    assert torch.equal(eager.params.flat, graph.params.flat) and torch.equal(eager.params.m, graph.params.m) and torch.equal(eager.params.v, graph.params.v)
    se, sg = eager.state_dict(), graph.state_dict()
    assert all(torch.equal(se[k], sg[k]) for k in se), "BatchNorm statistics / counters differ"
    assert eager.global_step == graph.global_step == 6 and graph.params.step == 6 and int(graph.params.step_dev.item()) == 6
    # dropout really differs from step to step inside the replays: the same batch twice gives different losses
    a = float(graph.training_step(batches[0])["mel"])
    b_ = float(graph.training_step(batches[0])["mel"])
    assert a != b_


@pytest.mark.parametrize("use_graph", [False, True])
def test_weight_gradient_schedules_give_the_same_step(cuda_device, use_graph):
    """The weight / bias gradients of a backward run (a) in place, (b) on the sibling stream with a fork per layer, (c) collected
    into groups of 8 layers per fork (the default: what a captured step can overlap, ops.SIDE_GROUP) or (d) groups of 3 (groups
    that end inside a block, a last partial group): the same launches in the same order per gradient buffer, so four steps end bit for
    bit in the same place -- eager and captured."""
    from everyvoice_amd.train import ops

    ref_cfg = _ref_cfg(0.1, 0)
    # (T = 64: B * T a multiple of 64, so the decoder's feed-forward blocks take the fused form of ffn_core under every schedule too)
    batches = [_shaped_batch(ref_cfg, seed, True, cuda_device, T=64) for seed in (5, 6)]
    prev_ops = ops.CONV_BACKEND["operands"]
    ops.CONV_BACKEND["operands"] = "bf16"
    try:
        assert ops.ffn_fused_supported(4, 64, ref_cfg.decoder.feedforward_dim, ref_cfg.decoder.input_dim)
    finally:
        ops.CONV_BACKEND["operands"] = prev_ops
    prev = ops.SIDE_GROUP[0]
    finals = []
    try:
        for side, group in ((False, 8), (True, 1), (True, 8), (True, 3)):
            ops.SIDE_GROUP[0] = group
            tr = _trainer(ref_cfg, cuda_device, learn_alignment=True, precision="bf16", use_graph=use_graph, side_wgrad=side)
            losses = [tr.training_step(batches[i % 2]) for i in range(4)]
            assert tr.last_step_was_graph == use_graph and tr._graph_failed is None
            finals.append((tr.params.flat.clone(), [{k: float(v) for k, v in l.items()} for l in losses]))
    finally:
        ops.SIDE_GROUP[0] = prev
    for flat, losses in finals[1:]:
        assert losses == finals[0][1]
        assert torch.equal(flat, finals[0][0])


@pytest.mark.parametrize("use_graph", [False, True])
def test_side_branches_on_their_stream_leave_the_same_bits_as_on_the_chain(cuda_device, use_graph):
    """The variance predictors (forward and backward) and the aligner's backward run on a stream of their own beside the main chain
    (train/fs2.py: _step); the tape structure -- their own tapes, aliases of their inputs, the joins -- is the same with the branch
    stream off (`_pred_branch = False`: everything on the chain), so the two schedules must agree bit for bit: parameters, optimiser
    moments and losses after five steps in bf16 with dropout on, eager and captured, at a shape large enough for the streams to really
    overlap.  A tensor released while the other stream still reads it, or a reduction flushed on the wrong stream, shows up here
    (the aligner's backward did: DESIGN.md 12.9, item 9)."""
    ref_cfg = _ref_cfg(0.1, 0)
    batches = [_shaped_batch(ref_cfg, seed, True, cuda_device, B=8, L=40, T=128) for seed in (3, 4)]
    finals = []
    for branch in (True, False):
        tr = _trainer(ref_cfg, cuda_device, learn_alignment=True, precision="bf16", use_graph=use_graph)
        tr._pred_branch = branch
        losses = [tr.training_step(batches[i % 2]) for i in range(5)]
        torch.cuda.synchronize()
        assert tr.last_step_was_graph == use_graph and tr._graph_failed is None
        finals.append((tr.params.flat.clone(), [{k: float(v) for k, v in l.items()} for l in losses]))
    assert finals[0][1] == finals[1][1]
    assert torch.equal(finals[0][0], finals[1][0])


def test_graph_buckets_pad_to_a_small_set_of_shapes(cuda_device):
    """graph_buckets=(8, 32): symbol and frame axes are padded up to multiples, so batches of different raw shapes share one
    captured step; the padded step equals the eager step on the same batch padded by hand (padding is explicit zeros: ids 0,
    mel 0, prior 0), and the counts that normalise the losses are the real ones."""
    ref_cfg = _ref_cfg(0.0, 0)
    tr = _trainer(ref_cfg, cuda_device, learn_alignment=True, use_graph=True, graph_buckets=(8, 32))
    plain = _trainer(ref_cfg, cuda_device, learn_alignment=True, use_graph=False, graph_buckets=(8, 32))
    shapes = set()
    for i, (L, seed) in enumerate([(18, 1), (20, 2), (23, 3), (19, 4), (22, 5)]):
        b = _align_batch(ref_cfg, 3, L, seed, cuda_device)
        lg, le = tr.training_step(b), plain.training_step(b)
        d, meta = tr._prepare(b)
        shapes.add((meta["L"], meta["T"]))
        assert meta["L"] % 8 == 0 and meta["T"] % 32 == 0 and d["ids"].shape[1] == meta["L"] and d["mel_t"].shape[2] == meta["T"]
        for k in le:
            assert torch.equal(le[k], lg[k]), (i, k)
    assert len(tr._graphs) <= len(shapes) <= 2 and tr._graph_failed is None and tr.last_step_was_graph
    assert torch.equal(tr.params.flat, plain.params.flat)


@pytest.mark.parametrize("learn_alignment", [False, True])
def test_graph_mode_under_data_parallelism_on_rccl_world_of_one(cuda_device, learn_alignment):
    """use_graph=True with a process group (what bench.py runs on N > 1 GPUs): the step is captured in three stretches -- forward +
    backward down to the decoder | the rest of the backward | clipping + optimiser -- with the two bucket all-reduces launched
    between the replays (the first one on a side stream, under the second stretch).  On a one-rank "nccl" group five steps must
    end exactly where five single-GPU steps end (dropout on: the seed base carries the rank and the world size)."""
    import os
    import socket

    import torch.distributed as dist

    ref_cfg = _ref_cfg(0.1)
    batches = [_shaped_batch(ref_cfg, seed, learn_alignment, cuda_device) for seed in (5, 6)]
    plain = _trainer(ref_cfg, cuda_device, learn_alignment=learn_alignment)
    for i in range(5):
        plain.training_step(batches[i % 2])
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=cuda_device)
    try:
        dp = _trainer(ref_cfg, cuda_device, learn_alignment=learn_alignment, process_group=True, use_graph=True)
        used = []
        for i in range(5):
            dp.training_step(batches[i % 2])
            used.append(dp.last_step_was_graph)
        torch.cuda.synchronize()
        assert dp._graph_failed is None, dp._graph_failed
        assert used == [False, False, True, True, True] and [len(e["graphs"]) for e in dp._graphs.values()] == [3]
    finally:
        dist.destroy_process_group()
    sa, sb = plain.state_dict(), dp.state_dict()
    for k in sa:
        assert torch.equal(sa[k], sb[k]), k
    assert torch.equal(plain.params.m, dp.params.m)
