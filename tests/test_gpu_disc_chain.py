"""The discriminator chains on flat packed bf16 tensors (train/disc_chain.py, csrc/disc_chain.hip, evmi_conv_pkflat_*) against the
op-by-op channel-major path they replace (itself pinned on the oracle by test_gpu_train_step.py / test_gpu_train_ops.py): same bf16
operands into the same matrix-core products; what differs is where fp32 values meet their bf16 rounding (the chain stores every
activation and every activation gradient in bf16 once; the op-by-op path rounds them when each consumer packs them), the logit layer
(reads the rounded activation) and the bias gradients / feature-matching sums (over rounded values).  Tolerances are set from that:
logits and losses 1e-2 of their scale, gradient tensors cosine >= 0.999 and norm within 2 %."""

import pytest
import torch

pytestmark = pytest.mark.gpu


def _trainer():
    from everyvoice_amd.train.hifigan import HiFiGANTrainer

    return HiFiGANTrainer(device="cuda:0", precision="bf16", use_graph=False, parallel_streams=False, seed=7)


def _cos(a, b):
    a, b = a.double().flatten(), b.double().flatten()
    return float((a @ b) / (a.norm() * b.norm() + 1e-30))


def _run_d_step(tr, d, audio, dl_seed, chain):
    """Forward + backward of one discriminator on a gradient-free batch with a fixed logits gradient -> (logits, {name: grad})."""
    from everyvoice_amd.train import autograd as ag
    from everyvoice_amd.train import hifigan as H
    from everyvoice_amd.train import ops

    H._DISC_CHAIN = chain
    ops.CONV_BACKEND["operands"] = "bf16"
    try:
        tr.d_params.zero_grad()
        for layer in d.layers():
            layer.frozen = False
        tr._materialize(d.layers())
        tape = ag.Tape()
        tr._bucket_hook(tape, tr.d_params, d.layers(), None)
        out, fm = d.forward(tape, ag.Var(audio, needs_grad=False))
        g = torch.Generator(device="cpu").manual_seed(dl_seed)
        out.grad = (torch.randn(out.data.shape, generator=g) / out.data.numel()).to(audio.device)
        tape.backward()
        torch.cuda.synchronize()
        names = [n for layer in d.layers() for n in layer.param_names()]
        grads = {n: tr.d_params.gradients()[n].clone() for n in names}
        return out.data.clone(), grads
    finally:
        ops.CONV_BACKEND["operands"] = "f32"
        H._DISC_CHAIN = True


def _sn_state(d):
    from everyvoice_amd.train.layers import SNConv

    return [(layer, layer.u.clone(), layer.v.clone()) for layer in d.layers() if isinstance(layer, SNConv)]


def _restore(state):
    for layer, u, v in state:
        layer.u.copy_(u)
        layer.v.copy_(v)
        layer._ready.clear()
        layer._calls.clear()
        layer._held.clear()


@pytest.mark.parametrize("which", ["mpd0", "mpd2", "mpd4", "msd0", "msd1", "msd2"])
@pytest.mark.parametrize("B,T", [(2, 8192), (3, 2400)])
def test_discriminator_step_of_a_chain_equals_the_op_by_op_path(which, B, T):
    tr = _trainer()
    d = tr.mpd[int(which[3])] if which.startswith("mpd") else tr.msd[int(which[3])]
    g = torch.Generator().manual_seed(3)
    y = (0.5 * torch.tanh(torch.randn(1, 2 * B, T, generator=g))).cuda()
    if which.startswith("msd") and which != "msd0":  # the pooled inputs of the later scales
        from everyvoice_amd.train import ops

        for _ in range(int(which[3])):
            y = ops.avgpool4s2(y)
    st = _sn_state(d)
    want_logits, want = _run_d_step(tr, d, y, 11, chain=False)
    _restore(st)
    got_logits, got = _run_d_step(tr, d, y, 11, chain=True)
    assert d._chain.ok and d._chain._cfgs, "the chain did not run"
    scale = float(want_logits.abs().max())
    assert float((got_logits - want_logits).abs().max()) <= 1e-2 * scale, (float((got_logits - want_logits).abs().max()), scale)
    for name, w in want.items():
        gt = got[name]
        if float(w.norm()) == 0.0:
            assert float(gt.norm()) == 0.0, name
            continue
        c, r = _cos(gt, w), float(gt.norm() / w.norm())
        # (bias gradients and the one-input-channel first layer's weight gradient are long sums with heavy cancellation over the
        # activation gradients, which the chain has rounded to bf16 and the op-by-op path has not: their own tolerance)
        floor = 0.995 if name.endswith(".bias") or ".convs.0." in name else 0.999
        assert c >= floor and abs(r - 1) <= 2e-2, (name, c, r)


def _run_g_step(tr, i, d, y, y_hat, chain):
    from everyvoice_amd.train import autograd as ag
    from everyvoice_amd.train import hifigan as H
    from everyvoice_amd.train import ops

    H._DISC_CHAIN = chain
    ops.CONV_BACKEND["operands"] = "bf16"
    try:
        for layer in d.layers():
            layer.frozen = True
        tr._materialize(d.layers())
        ops.fill_(tr._slots, 0.0)
        tape = ag.Tape()
        xf = ag.Var(y_hat)
        real = d.forward(tape, ag.Var(y, needs_grad=False), role="g_real")
        fake = d.forward(tape, xf, role="g_fake")
        tr._g_losses(i, real, fake)
        tape.backward()
        torch.cuda.synchronize()
        return tr._slots[:, i].clone(), xf.grad.clone()
    finally:
        ops.CONV_BACKEND["operands"] = "f32"
        H._DISC_CHAIN = True
        for layer in d.layers():
            layer.frozen = False


@pytest.mark.parametrize("which", ["mpd1", "mpd3", "msd0", "msd1"])
def test_generator_step_pass_of_a_chain_equals_the_op_by_op_path(which):
    tr = _trainer()
    i = int(which[3]) if which.startswith("mpd") else len(tr.mpd) + int(which[3])
    d = tr.discriminators()[i]
    B, T = 2, 8192
    g = torch.Generator().manual_seed(5)
    y = (0.5 * torch.tanh(torch.randn(1, B, T, generator=g))).cuda()
    y_hat = (0.5 * torch.tanh(torch.randn(1, B, T, generator=g))).cuda()
    if which == "msd1":
        from everyvoice_amd.train import ops

        y, y_hat = ops.avgpool4s2(y), ops.avgpool4s2(y_hat)
    st = _sn_state(d)
    want_slots, want_dx = _run_g_step(tr, i, d, y, y_hat, chain=False)
    _restore(st)
    got_slots, got_dx = _run_g_step(tr, i, d, y, y_hat, chain=True)
    # slots: [d, g_adv, g_fm, d(generated call of the spectral-norm scale)]
    for row in (1, 2):
        assert abs(float(got_slots[row]) - float(want_slots[row])) <= 5e-3 * max(1e-6, abs(float(want_slots[row]))), (row, got_slots, want_slots)
    c, r = _cos(got_dx, want_dx), float(got_dx.norm() / want_dx.norm())
    assert c >= 0.995 and abs(r - 1) <= 3e-2, (c, r)
