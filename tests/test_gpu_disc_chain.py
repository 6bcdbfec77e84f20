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
@pytest.mark.parametrize("B,T", [(2, 8192)])
def test_discriminator_step_of_a_chain_equals_the_op_by_op_path(which, B, T):
    """(At the training segment length both paths round the same operands at the same points.  On short inputs the op-by-op path
    serves some layers with exact-fp32 kernels -- ops.fwd_takes_bf16 -- and the comparison would measure bf16 against fp32: short and
    odd sizes are compared with the torch emulation of the chain's own roundings below.)"""
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


@pytest.mark.parametrize("which", ["mpd0", "mpd4", "msd1", "msd2"])
def test_generator_step_on_one_real_and_generated_batch_equals_two_passes(which):
    """The generator step of a weight-normed discriminator as ONE forward over [real | generated] with the backward over the generated
    half (DiscChain.forward(grad_from=B)) against the two separate passes: same kernels on the same values, only the column counts
    (hence tiles and split-K sums) differ."""
    from everyvoice_amd.train import autograd as ag
    from everyvoice_amd.train import ops

    tr = _trainer()
    i = int(which[3]) if which.startswith("mpd") else len(tr.mpd) + int(which[3])
    d = tr.discriminators()[i]
    B, T = 2, 8192
    g = torch.Generator().manual_seed(9)
    y = (0.5 * torch.tanh(torch.randn(1, B, T, generator=g))).cuda()
    y_hat = (0.5 * torch.tanh(torch.randn(1, B, T, generator=g))).cuda()
    for _ in range(int(which[3]) if which.startswith("msd") else 0):
        y, y_hat = ops.avgpool4s2(y), ops.avgpool4s2(y_hat)
    want_slots, want_dx = _run_g_step(tr, i, d, y, y_hat, chain=True)
    ops.CONV_BACKEND["operands"] = "bf16"
    try:
        for layer in d.layers():
            layer.frozen = True
        tr._materialize(d.layers())
        ops.fill_(tr._slots, 0.0)
        tape = ag.Tape()
        both = ag.Var(torch.cat([y, y_hat], 1).contiguous())
        res = d.forward(tape, both, role="g_both", grad_from=B)
        tr._g_losses(i, None, res)
        tape.backward()
        torch.cuda.synchronize()
        got_slots, got_dx = tr._slots[:, i].clone(), both.grad.clone()
    finally:
        ops.CONV_BACKEND["operands"] = "f32"
        for layer in d.layers():
            layer.frozen = False
    assert float(got_dx[:, :B].abs().max()) == 0.0  # nothing flows into the real half
    for row in (1, 2):
        assert abs(float(got_slots[row]) - float(want_slots[row])) <= 1e-4 * abs(float(want_slots[row])), (row, got_slots, want_slots)
    c, r = _cos(got_dx[:, B:], want_dx), float(got_dx[:, B:].norm() / want_dx.norm())
    assert c >= 0.9999 and abs(r - 1) <= 2e-3, (c, r)


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


# ---- the flat packed kernels themselves against torch on the same bf16 operands (fp32 accumulation: only the summation order differs) ----
def _pf_from(x, T, dev):
    """x [C, n_items, len] fp32 -> PF with item pitch T holding bf16(x)."""
    from everyvoice_amd.train.disc_chain import FRONT, PF

    C, n, ln = x.shape
    pf = PF(C, n, T, ln, dev)
    v = pf.buf.view(torch.bfloat16).view(C // 8, pf.plane, 8)
    xb = x.to(torch.bfloat16).to(dev).view(C // 8, 8, n, ln).permute(0, 2, 3, 1)  # [octet][item][pos][8]
    for i in range(n):
        v[:, FRONT + i * T : FRONT + i * T + ln, :] = xb[:, i]
    return pf


def _pf_to(pf):
    """-> ([C, n_items, valid] fp32 of the data units, max |.| of everything outside them -- which must stay zero)"""
    from everyvoice_amd.train.disc_chain import FRONT

    v = pf.buf.view(torch.bfloat16).view(pf.C // 8, pf.plane, 8).float()
    items = torch.stack([v[:, FRONT + i * pf.T : FRONT + i * pf.T + pf.valid, :] for i in range(pf.n_items)], 1)  # [oct][item][pos][8]
    out = items.permute(0, 3, 1, 2).reshape(pf.C, pf.n_items, pf.valid)
    rest = v.clone()
    for i in range(pf.n_items):
        rest[:, FRONT + i * pf.T : FRONT + i * pf.T + pf.valid, :] = 0
    return out.cpu(), float(rest.abs().max())


@pytest.mark.parametrize("n,t_in,cin,cout,k,s,pad,g", [
    (6, 38, 512, 1024, 41, 4, 20, 16), (6, 10, 1024, 1024, 41, 1, 20, 16), (6, 10, 1024, 1024, 5, 1, 2, 1), (22, 28, 512, 1024, 5, 3, 2, 1),
    (6, 150, 256, 512, 41, 4, 20, 16), (6, 300, 128, 256, 41, 2, 20, 16), (6, 600, 128, 128, 41, 2, 20, 4), (10, 83, 128, 512, 5, 3, 2, 1),
    (5, 249, 32, 128, 5, 3, 2, 1), (3, 4096, 128, 256, 41, 2, 20, 16)])
def test_flat_packed_kernels_against_torch(n, t_in, cin, cout, k, s, pad, g):
    import torch.nn.functional as F

    from everyvoice_amd import _lib
    from everyvoice_amd.train.disc_chain import PF, _conv_len

    lib = _lib.load()
    dev = torch.device("cuda:0")
    st = _lib.current_stream_ptr(dev)
    gen = torch.Generator().manual_seed(n * 1000 + t_in)
    bf = lambda t: t.to(torch.bfloat16).float()  # noqa: E731
    t_out = _conv_len(t_in, k, s, pad)
    right = (t_out - 1) * s + (k - 1) - pad - (t_in - 1)
    gdy = -(-max(0, k - 1 - pad) // s)
    Tc = max(t_out + gdy, -(-(t_in + max(pad, right, 0)) // s))
    x = torch.randn(cin, n, t_in, generator=gen)
    w = torch.randn(cout, cin // g, k, generator=gen) / (cin // g * k) ** 0.5
    b = torch.randn(cout, generator=gen) * 0.1
    X = _pf_from(x, s * Tc, dev)
    Y = PF(cout, n, Tc + 3, t_out, dev)  # (stored at another pitch than it is computed at)
    shf, shd = (0, n, s * Tc, cin, cout, k, s, pad, 1, g), (1, n, Tc, cin, cout, k, s, pad, 1, g)
    wsf = torch.empty(lib.evmi_conv_pkflat_ws_elems(*shf), device=dev)
    wsd = torch.empty(lib.evmi_conv_pkflat_ws_elems(*shd), device=dev)
    assert wsf.numel() > 0 and wsd.numel() > 0
    _lib.check(lib.evmi_conv_pkflat_tab(*shf, wsf.data_ptr(), wsf.numel(), st), "tab")
    _lib.check(lib.evmi_conv_pkflat_tab(*shd, wsd.data_ptr(), wsd.numel(), st), "tab")
    wd = w.to(dev)
    bd = b.to(dev)
    jobs = (_lib.PkFlatJob * 2)()
    wfs = []
    for j, mode in enumerate((0, 1)):
        wf = torch.empty(lib.evmi_conv_pkflat_frag_elems(mode, cin, cout, k, s, g), device=dev)
        wfs.append(wf)
        jobs[j].mode, jobs[j].c_in, jobs[j].c_out, jobs[j].k, jobs[j].stride, jobs[j].groups = mode, cin, cout, k, s, g
        jobs[j].w, jobs[j].wf, jobs[j].wf_elems = wd.data_ptr(), wf.data_ptr(), wf.numel()
    _lib.check(lib.evmi_conv_pkflat_fragments(2, jobs, st), "fragments")
    # forward: lrelu(conv + bias)
    _lib.check(lib.evmi_conv_pkflat_fwd(X.ptr, X.plane, wfs[0].data_ptr(), bd.data_ptr(), Y.ptr, Y.plane, wsf.data_ptr(), wsf.numel(), n, s * Tc, cin, cout, k, s,
                                        pad, 1, g, t_out, Y.T, 1, 0.1, st), "fwd")
    got, stray = _pf_to(Y)
    want = F.leaky_relu(F.conv1d(bf(x).permute(1, 0, 2), bf(w), b, s, pad, 1, g), 0.1).permute(1, 0, 2)
    assert stray == 0.0, "the forward wrote outside the data units"
    err = float((got - bf(want)).abs().max() / want.abs().max())
    assert err <= 1e-2, ("fwd", err)  # (one bf16 rounding of the output: 2^-8 relative)
    assert float((got - want).norm() / want.norm()) <= 4e-3
    # input gradient with the activation backward: dx = conv_input_grad(dy) * lrelu'(x)
    dy = torch.randn(cout, n, t_out, generator=gen)
    DY = _pf_from(dy, Tc, dev)
    DX = PF(cin, n, s * Tc + 5, t_in, dev)
    _lib.check(lib.evmi_conv_pkflat_dgrad(DY.ptr, DY.plane, wfs[1].data_ptr(), DX.ptr, DX.plane, wsd.data_ptr(), wsd.numel(), n, Tc, cin, cout, k, s, pad, 1, g,
                                          t_in, DX.T, X.ptr, 0, X.plane, X.T, 0.1, 0.0, st), "dgrad")
    got, stray = _pf_to(DX)
    dx = torch.nn.grad.conv1d_input((n, cin, t_in), bf(w), bf(dy).permute(1, 0, 2), s, pad, 1, g).permute(1, 0, 2)
    want = dx * torch.where(bf(x) > 0, 1.0, 0.1)
    assert stray == 0.0, "the input gradient wrote outside the data units"
    assert float((got - want).norm() / want.norm()) <= 4e-3, ("dgrad", float((got - want).norm() / want.norm()))
    # weight gradient
    nw = lib.evmi_conv_pkflat_wgrad_ws_elems(n, Tc, cin, cout, k, s, 1, g)
    assert nw >= 0
    wsw = torch.empty(max(nw, 4), device=dev)
    dw = torch.zeros(cout, cin // g, k, device=dev)
    _lib.check(lib.evmi_conv_pkflat_wgrad(X.ptr, X.plane, DY.ptr, DY.plane, dw.data_ptr(), wsw.data_ptr(), wsw.numel(), n, Tc, cin, cout, k, s, pad, 1, g, 0, st), "wgrad")
    want = torch.nn.grad.conv1d_weight(bf(x).permute(1, 0, 2), w.shape, bf(dy).permute(1, 0, 2), s, pad, 1, g)
    assert float((dw.cpu() - want).abs().max() / want.abs().max()) <= 1e-4, ("wgrad", float((dw.cpu() - want).abs().max() / want.abs().max()))


# ---- a whole chain against torch autograd with the chain's rounding points restated ------------------------------------------------
class _RoundFwd(torch.autograd.Function):  # a tensor stored in bf16; its gradient passes
    @staticmethod
    def forward(ctx, x):
        return x.to(torch.bfloat16).float()

    @staticmethod
    def backward(ctx, g):
        return g


class _RoundBwd(torch.autograd.Function):  # identity whose gradient is stored in bf16
    @staticmethod
    def forward(ctx, x):
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        return g.to(torch.bfloat16).float()


def _torch_chain(d, period, audio, ws, bs, real_acts=None):
    """audio [n, T] -> (logits [n * p, len], activations): first and logit layers in exact fp32 on their stored operands, matrix-core
    layers on bf16-rounded weights; every activation and every pre-activation gradient rounded once."""
    import torch.nn.functional as F

    n, T = audio.shape
    x = audio
    if period > 1:
        H = (T + period - 1) // period
        if H * period > T:
            x = F.pad(x.unsqueeze(1), (0, H * period - T), mode="reflect").squeeze(1)
        x = x.view(n, H, period).permute(0, 2, 1).reshape(n * period, 1, H)
    else:
        x = x.unsqueeze(1)
    acts = []
    for i, c in enumerate(d.convs):
        w = ws[i] if i == 0 else ws[i] + (ws[i].to(torch.bfloat16).float() - ws[i]).detach()
        pre = _RoundBwd.apply(F.conv1d(x, w, bs[i], c.stride, c.pad, c.dil, c.groups))
        x = _RoundFwd.apply(F.leaky_relu(pre, 0.1))
        acts.append(x)
    post = d.conv_post
    return F.conv1d(x, ws[-1], bs[-1], 1, post.pad), acts


@pytest.mark.parametrize("which,n,T", [("mpd4", 3, 2400), ("mpd1", 2, 4001), ("msd1", 3, 2400), ("msd2", 6, 600), ("msd1", 2, 8192)])
@pytest.mark.parametrize("generator_step", [False, True])
def test_chain_against_torch_autograd_with_its_roundings(which, n, T, generator_step):
    from everyvoice_amd.train import autograd as ag
    from everyvoice_amd.train import hifigan as H
    from everyvoice_amd.train import ops

    tr = _trainer()
    i_d = int(which[3]) if which.startswith("mpd") else len(tr.mpd) + int(which[3])
    d = tr.discriminators()[i_d]
    period = getattr(d, "period", 1)
    g = torch.Generator().manual_seed(n * 100 + T)
    y = 0.5 * torch.tanh(torch.randn(n, T, generator=g))
    y_real = 0.5 * torch.tanh(torch.randn(n, T, generator=g))
    layers = d.layers()
    ops.CONV_BACKEND["operands"] = "bf16"
    try:
        tr.d_params.zero_grad()
        for layer in layers:
            layer.frozen = generator_step
            ops.fill_(layer._dw, 0.0)
        tr._materialize(layers)
        ops.fill_(tr._slots, 0.0)
        tape = ag.Tape()
        xf = ag.Var(y.view(1, n, T).cuda(), needs_grad=generator_step)
        if generator_step:
            real = d.forward(tape, ag.Var(y_real.view(1, n, T).cuda(), needs_grad=False), role="g_real")
            fake = d.forward(tape, xf, role="g_fake")
            tr._g_losses(i_d, real, fake)
            out = fake[0]
        else:
            out, _ = d.forward(tape, xf, role="pair")
            out.grad = (torch.randn(out.data.shape, generator=g) / out.data.numel()).cuda()
        assert d._chain.ok and d._chain._cfgs, "the chain did not run"
        dl = out.grad.clone()
        tape.backward()
        torch.cuda.synchronize()
        ws = [layer._w.detach().cpu().clone().requires_grad_(True) for layer in layers]
        bs = [layer.bias_data().detach().cpu().clone().requires_grad_(True) for layer in layers]
        got_logits = out.data.cpu()[0]
        got_dw = [layer._dw.cpu().clone() for layer in layers]
        got_db = [tr.d_params.gradients()[layer.name + ".bias"].cpu().clone() for layer in layers]
        got_dx = xf.grad.cpu()[0] if generator_step else None
        slots = tr._slots[:, i_d].cpu()
    finally:
        ops.CONV_BACKEND["operands"] = "f32"
        for layer in layers:
            layer.frozen = False
    # torch side
    ya = y.clone().requires_grad_(True)
    logits, acts = _torch_chain(d, period, ya, ws, bs)
    if generator_step:
        with torch.no_grad():
            logits_r, acts_r = _torch_chain(d, period, y_real, ws, bs)
        fm = sum(2.0 * (a - b).abs().mean() for a, b in zip(acts + [logits], acts_r + [logits_r]))
        adv = ((logits - 1.0) ** 2).mean()
        (fm + adv).backward()
        assert abs(float(slots[2]) - float(fm)) <= 2e-3 * float(fm), (float(slots[2]), float(fm))
        assert abs(float(slots[1]) - float(adv)) <= 2e-3 * float(adv), (float(slots[1]), float(adv))
        c, r = _cos(got_dx, ya.grad), float(got_dx.norm() / ya.grad.norm())
        assert c >= 0.999 and abs(r - 1) <= 1e-2, ("d audio", c, r)
        return
    logits.backward(dl.cpu()[0].view_as(logits))
    scale = float(logits.abs().max())
    assert float((got_logits - logits.detach().view_as(got_logits)).abs().max()) <= 2e-3 * scale
    for layer, dw, db, w, b in zip(layers, got_dw, got_db, ws, bs):
        for name, gt, want in ((layer.name + ".weight", dw, w.grad), (layer.name + ".bias", db, b.grad)):
            c, r = _cos(gt, want), float(gt.norm() / want.norm())
            # (what is left between the two: fp32 summation order moving single values across a bf16 rounding boundary -- an ulp of
            # 2^-8 on that element, which the layers behind it amplify on the shortest inputs (items of 10 positions under 41 taps): measured
            # 0.9988-0.9994 (weights) / 0.9995 (biases, sums with heavy cancellation) there from run to run, 0.99999 at the training
            # segment length -- so the short inputs get a floor that catches a wrong tap or a wrong item boundary (those cost percents),
            # the long ones the tight one)
            small = n * T < 16000  # (items of 10-20 positions in the late layers)
            floor = 0.995 if small else (0.999 if name.endswith(".bias") else 0.9995)
            assert c >= floor and abs(r - 1) <= 1e-2, (name, c, r)


@pytest.mark.parametrize("switch", ["EVMI_WG_TAPSPLIT=0", "EVMI_WG_WIDE=0 EVMI_WG_XCD=0"])
def test_flat_kernels_behind_their_switches(switch):
    """The flat weight gradient without the tap split over a row's waves; (round 6) on 64 x 64 tiles only, workgroups in launch order
    instead of XCD order: the comparisons with torch of this file once more in a child process each (the switches are read once per
    process)."""
    import os
    import subprocess
    import sys

    env = dict(kv.split("=") for kv in switch.split())
    r = subprocess.run([sys.executable, "-m", "pytest", __file__, "-q", "-x", "-k", "flat_packed_kernels_against_torch or chain_against_torch_autograd"],
                       env=dict(os.environ, **env), capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
