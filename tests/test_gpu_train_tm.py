"""Time-major bf16 training of the generator's residual stacks (csrc/train_tm.hip, train/mrf_tm.py) against torch on the CPU.

Per kernel: the forward / input-gradient convolution entry (leaky ReLU on load and in the epilogue, residual, activation-backward
mask) and the weight / bias gradients equal torch's fp32 convolution of the SAME bf16 operands to summation order, outputs
rounded to bf16 once (tolerance: one bf16 ulp of the tensor's scale for bf16 outputs, 1e-4 for fp32 outputs).  Per stage: the
MRF op (forward + recorded backward) against torch autograd of the upstream ResBlock1 average on bf16-rounded parameters, within
bf16 storage noise.  Whole step: test_gpu_train_step.py's bf16 tests run this path by default (and the packed path behind
EVMI_TRAIN_TM=0 in a child process)."""

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _bf(t):
    return t.to(torch.bfloat16).to(torch.float32)


def _fill(buf, x_btc):
    """x [B, T, C] fp32 -> the valid rows of a TMBuf (rounded to bf16)."""
    buf.valid().copy_(x_btc.to(buf.store.device, torch.bfloat16))
    return _bf(x_btc)


def _read(buf):
    return buf.valid().to(torch.float32).cpu()


CASES = [(32, 11, 5), (32, 3, 1), (64, 7, 3), (128, 11, 5), (128, 3, 1), (256, 7, 3), (256, 11, 1)]


@pytest.mark.parametrize("C,k,d", CASES)
def test_tm_conv_forward_input_gradient_weight_gradient(cuda_device, C, k, d):
    from everyvoice_amd import _lib
    from everyvoice_amd.train import ops
    from everyvoice_amd.train.mrf_tm import TMBuf

    lib = _lib.load()
    assert lib.evmi_conv_tc_supported(C, C, k, d)
    g = torch.Generator().manual_seed(C + k + d)
    B, T = 3, 400 if C >= 128 else 1200
    slope = 0.1
    x = torch.randn(B, T, C, generator=g)
    w = torch.randn(C, C, k, generator=g) * (1.5 / (C * k) ** 0.5)
    b = torch.randn(C, generator=g) * 0.1
    dev = cuda_device
    xb, tb, yb = TMBuf(C, B, T, dev), TMBuf(C, B, T, dev), TMBuf(C, B, T, dev)
    xr = _fill(xb, x)
    wd = w.to(dev)
    laid = torch.empty(C * C * k // 2, device=dev)
    laid_t = torch.empty_like(laid)
    s = torch.cuda.current_stream().cuda_stream
    _lib.check(lib.evmi_conv_tc_relayout_f32(wd.data_ptr(), laid.data_ptr(), C, C, k, d, 0, s), "relayout")
    _lib.check(lib.evmi_conv_tc_relayout_f32(wd.data_ptr(), laid_t.data_ptr(), C, C, k, d, 1, s), "relayout^T")
    pad = d * (k - 1) // 2
    bd = b.to(dev)

    def conv(xin, wl, bias, out, pre=1.0, post=1.0, res=None, mask=None, mslope=1.0):
        _lib.check(lib.evmi_conv_tc_tm_bf16(xin.ptr, wl.data_ptr(), bias.data_ptr(), res.ptr if res else 0, mask.ptr if mask else 0, out.ptr, B, T, xin.Tp, xin.PL,
                                            C, C, k, d, pre, post, mslope, 1.0, s), "conv_tc_tm")

    # forward of a pair's first convolution: t = lrelu(conv(lrelu(x)) + b)
    conv(xb, laid, bd, tb, pre=slope, post=slope)
    want_t = F.leaky_relu(F.conv1d(F.leaky_relu(xr, slope).transpose(1, 2), _bf(w), b, 1, pad, d), slope).transpose(1, 2)
    got_t = _read(tb)
    scale = float(want_t.abs().max())
    assert float((got_t - want_t).abs().max()) <= 2 ** -7 * scale  # one bf16 rounding of the output (+ summation order)
    # ... and of its second: y = x + conv(t) + b, no activations
    conv(tb, laid, bd, yb, res=xb)
    want_y = (F.conv1d(got_t.transpose(1, 2), _bf(w), b, 1, pad, d).transpose(1, 2) + xr)
    assert float((_read(yb) - want_y).abs().max()) <= 2 ** -7 * float(want_y.abs().max())
    assert float(yb.store.view(torch.bfloat16)[: 64 * C].float().abs().max()) == 0.0  # guard rows untouched
    pad_rows = yb.store.view(torch.bfloat16)[64 * C:].view(-1, C)[: yb.PL]
    assert float(pad_rows.float().abs().max()) == 0.0  # the zero rows in front of item 0 too
    # input gradient with the activation backward and the skip path in the epilogue: dx = conv^T(dy) * lrelu'(x) + dy
    dy = torch.randn(B, T, C, generator=g)
    dyb, dxb = TMBuf(C, B, T, dev), TMBuf(C, B, T, dev)
    dyr = _fill(dyb, dy)
    zero = torch.zeros(C, device=dev)
    conv(dyb, laid_t, zero, dxb, mask=xb, mslope=slope, res=dyb)
    want_dx = torch.nn.grad.conv1d_input((B, C, T), _bf(w), dyr.transpose(1, 2).contiguous(), 1, pad, d).transpose(1, 2)
    want_dx = want_dx * torch.where(xr > 0, 1.0, slope) + dyr
    assert float((_read(dxb) - want_dx).abs().max()) <= 2 ** -7 * float(want_dx.abs().max())
    # weight gradient (fp32 result) and bias gradient
    n = lib.evmi_conv1d_wgrad_tm_bf16_ws_elems(xb.rows, C, C, k, d)
    assert n >= 0
    ws = torch.empty(n, device=dev)
    base = torch.randn(C, C, k, generator=g)
    dw = base.clone().to(dev)
    _lib.check(lib.evmi_conv1d_wgrad_tm_bf16(xb.ptr, dyb.ptr, dw.data_ptr(), ws.data_ptr(), n, xb.rows, C, C, k, pad, d, 1, s), "wgrad_tm")
    want_dw = torch.nn.grad.conv1d_weight(xr.transpose(1, 2).contiguous(), (C, C, k), dyr.transpose(1, 2).contiguous(), 1, pad, d)
    assert float((dw.cpu() - base - want_dw).abs().max()) <= 1e-4 * float(want_dw.abs().max())
    nb = lib.evmi_tm_colsum_bf16_ws_elems(dyb.rows, C)
    wsb = torch.empty(nb, device=dev)
    db = torch.ones(C, device=dev)
    _lib.check(lib.evmi_tm_colsum_bf16(dyb.ptr, db.data_ptr(), wsb.data_ptr(), nb, dyb.rows, C, 1, s), "colsum")
    torch.testing.assert_close(db.cpu() - 1.0, dyr.sum(dim=(0, 1)), rtol=1e-4, atol=1e-3)
    assert ops.flop_counter() >= 0


def test_layout_changes_round_trip(cuda_device):
    from everyvoice_amd import _lib
    from everyvoice_amd.train.mrf_tm import TMBuf

    lib = _lib.load()
    g = torch.Generator().manual_seed(3)
    C, B, T = 64, 3, 208
    x = torch.randn(C, B, T, generator=g)
    s = torch.cuda.current_stream().cuda_stream
    a, b2 = TMBuf(C, B, T, cuda_device), TMBuf(C, B, T, cuda_device)
    _lib.check(lib.evmi_cbt_f32_to_tm_bf16(x.to(cuda_device).data_ptr(), a.ptr, C, B, T, a.Tp, a.PL, 0.1, 0.5, s), "to_tm")
    want = _bf(F.leaky_relu(x, 0.1) * 0.5).permute(1, 2, 0)
    assert torch.equal(_read(a), want)
    _lib.check(lib.evmi_tm_lrelu_bf16(a.ptr, b2.ptr, a.numel_body, 0.1, s), "lrelu")
    assert torch.equal(_read(b2), _bf(F.leaky_relu(want, 0.1)))
    out = torch.empty(C, B, T, device=cuda_device)
    _lib.check(lib.evmi_tm_bf16_to_cbt_f32(a.ptr, b2.ptr, 0, out.data_ptr(), C, B, T, a.Tp, a.PL, 0.25, s), "to_cbt")
    torch.testing.assert_close(out.cpu(), ((want + _read(b2)) * 0.25).permute(2, 0, 1), rtol=1e-6, atol=1e-7)


@pytest.mark.parametrize("C,T", [(32, 512), (128, 128)])
def test_mrf_stage_forward_backward_against_torch_autograd(cuda_device, C, T):
    """MRFStageTM.apply (three ResBlock1 branches k = 3 / 7 / 11 x dilations 1 / 3 / 5, averaged) against torch autograd on the
    bf16-rounded weights: output within 1 % (bf16 storage of every intermediate), input gradient and every weight / bias gradient
    cosine >= 0.999 and norm within 2 %."""
    from everyvoice_amd.train import autograd as ag
    from everyvoice_amd.train import ops
    from everyvoice_amd.train.layers import ParamGroup, WNConv, kaiming_uniform_conv_init_
    from everyvoice_amd.train.mrf_tm import MRFStageTM, stage_supported

    ks, dils, slope, B = (3, 7, 11), (1, 3, 5), 0.1, 2
    assert stage_supported(C, ks, [dils] * 3)
    grp = ParamGroup(cuda_device)
    pairs = [[(WNConv(grp, f"rb{j}.convs1.{m}", C, C, k, pad=d * (k - 1) // 2, dil=d), WNConv(grp, f"rb{j}.convs2.{m}", C, C, k, pad=(k - 1) // 2))
              for m, d in enumerate(dils)] for j, k in enumerate(ks)]
    grp.finalize()
    gen = torch.Generator().manual_seed(C)
    layers = [c for p in pairs for pr in p for c in pr]
    for layer in layers:
        kaiming_uniform_conv_init_(layer, gen)
        layer.materialize()
    st = MRFStageTM(C, pairs, slope, cuda_device)
    x = torch.randn(C, B, T, generator=gen)
    dout = torch.randn(C, B, T, generator=gen)
    tape = ag.Tape()
    xv = ag.Var(x.to(cuda_device))
    grp.zero_grad()
    ops.CONV_BACKEND["operands"] = "bf16"
    try:
        out = st.apply(tape, xv)
        out.grad = dout.to(cuda_device)
        tape.backward()
        for layer in layers:
            layer.finish_grads()
    finally:
        ops.CONV_BACKEND["operands"] = "f32"
    # torch: the same network on the effective weights (rounded to bf16, as the kernels read them), fp32 activations
    xt = x.permute(1, 0, 2).clone().requires_grad_()
    ws = {}
    ys = []
    for j, p in enumerate(pairs):
        cur = xt
        for m, (c1, c2) in enumerate(p):
            w1 = _bf(c1._w.cpu()).requires_grad_()
            w2 = _bf(c2._w.cpu()).requires_grad_()
            b1 = c1.bias_data().cpu().clone().requires_grad_()
            b2 = c2.bias_data().cpu().clone().requires_grad_()
            ws[(j, m)] = (w1, w2, b1, b2)
            t = F.leaky_relu(F.conv1d(F.leaky_relu(cur, slope), w1, b1, 1, c1.pad, c1.dil), slope)
            cur = cur + F.conv1d(t, w2, b2, 1, c2.pad, 1)
        ys.append(cur)
    want = sum(ys) / 3.0
    want.backward(dout.permute(1, 0, 2))

    def agree(got, ref, what, cos_min=0.999):
        got, ref = got.double().flatten(), ref.double().flatten()
        cos = float(torch.dot(got, ref) / (got.norm() * ref.norm()))
        ratio = float(got.norm() / ref.norm())
        assert cos >= cos_min and 0.98 <= ratio <= 1.02, (what, cos, ratio)

    agree(out.data.cpu().permute(1, 0, 2), want.detach(), "output", 0.9999)
    agree(xv.grad.cpu().permute(1, 0, 2), xt.grad, "input gradient")
    for (j, m), (w1, w2, b1, b2) in ws.items():
        c1, c2 = pairs[j][m]
        # d loss / d effective weight sits in the layer's sink before finish_grads folds it into (g, v): compare through the bias
        # gradients (plain parameters) and the weight-norm'd parameter gradients' direction
        agree(grp.gradient(c1.i_bias).cpu(), b1.grad, f"bias {j}.{m}.1", 0.995)  # (sums of near-cancelling bf16-stored gradients)
        agree(grp.gradient(c2.i_bias).cpu(), b2.grad, f"bias {j}.{m}.2", 0.995)
        for c, w in ((c1, w1), (c2, w2)):
            g_, v_ = grp.data(c.i_g).cpu(), grp.data(c.i_v).cpu()
            norm = v_.flatten(1).norm(dim=1).view(-1, 1, 1)
            dv_ref = (g_ / norm) * (w.grad - (w.grad * v_).flatten(1).sum(1).view(-1, 1, 1) * v_ / norm ** 2)
            agree(grp.gradient(c.i_v).cpu(), dv_ref, f"weight_v {j}.{m}", 0.998)
