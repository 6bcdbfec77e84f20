"""Training building blocks on the GPU vs torch-CPU autograd (the oracle for gradients).
fp32 everywhere: tolerances are summation-order noise (rtol 1e-4 / atol 1e-5 unless noted)."""

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def cbt(t):  # torch [B, C, T] -> CBT [C, B, T]
    return t.permute(1, 0, 2).contiguous()


def bct(t):
    return t.permute(1, 0, 2).contiguous()


CONVS = [
    dict(cin=8, cout=12, k=3, stride=1, pad=1, dil=1, groups=1),
    dict(cin=16, cout=16, k=11, stride=1, pad=25, dil=5, groups=1),
    dict(cin=16, cout=32, k=41, stride=4, pad=20, dil=1, groups=4),   # MSD-style grouped strided
    dict(cin=1, cout=8, k=15, stride=1, pad=7, dil=1, groups=1),
    dict(cin=4, cout=6, k=5, stride=3, pad=2, dil=1, groups=1),       # MPD-style (k,1)/(3,1) on the period view
    dict(cin=24, cout=8, k=1, stride=1, pad=0, dil=1, groups=1),      # pointwise: no unfold
]


@pytest.mark.parametrize("c", CONVS)
def test_conv1d_fwd_bwd(cuda_device, c):
    from everyvoice_amd.train import ops

    g = torch.Generator().manual_seed(1)
    B, T = 3, 101
    x = torch.randn(B, c["cin"], T, generator=g, requires_grad=True)
    w = (torch.randn(c["cout"], c["cin"] // c["groups"], c["k"], generator=g) * 0.2).requires_grad_()
    b = torch.randn(c["cout"], generator=g, requires_grad=True)
    y = F.conv1d(x, w, b, c["stride"], c["pad"], c["dil"], c["groups"])
    dy = torch.randn(y.shape, generator=g)
    y.backward(dy)
    xd, wd, bd, dyd = cbt(x.detach()).to(cuda_device), w.detach().to(cuda_device), b.detach().to(cuda_device), cbt(dy).to(cuda_device)
    yg = ops.conv1d_fwd(xd, wd, bd, c["stride"], c["pad"], c["dil"], c["groups"])
    torch.testing.assert_close(bct(yg.cpu()), y.detach(), rtol=1e-4, atol=1e-5)
    dbuf = torch.zeros(c["cout"], device=cuda_device)
    dx, dw, db = ops.conv1d_bwd(xd, wd, dyd, c["stride"], c["pad"], c["dil"], c["groups"], db_out=dbuf)
    torch.testing.assert_close(bct(dx.cpu()), x.grad, rtol=1e-4, atol=1e-5)
    torch.testing.assert_close(dw.cpu(), w.grad, rtol=1e-4, atol=1e-4)
    torch.testing.assert_close(db.cpu(), b.grad, rtol=1e-4, atol=1e-4)
    # accumulation into existing gradient buffers
    dw2 = dw.clone()
    ops.conv1d_bwd(xd, wd, dyd, c["stride"], c["pad"], c["dil"], c["groups"], need_dx=False, dw_out=dw2, db_out=dbuf, accumulate=True)
    torch.testing.assert_close(dw2.cpu(), 2 * w.grad, rtol=1e-4, atol=2e-4)
    torch.testing.assert_close(dbuf.cpu(), 2 * b.grad, rtol=1e-4, atol=2e-4)


MFMA_CONVS = CONVS + [
    dict(cin=128, cout=128, k=11, stride=1, pad=25, dil=5, groups=1),   # generator resblock shape
    dict(cin=40, cout=100, k=7, stride=1, pad=3, dil=1, groups=1),      # partial M tile, channels not a multiple of 16
    dict(cin=128, cout=256, k=41, stride=2, pad=20, dil=1, groups=16),  # MSD layer 2
    dict(cin=32, cout=128, k=5, stride=3, pad=2, dil=1, groups=1),      # MPD layer 1
    dict(cin=64, cout=1, k=3, stride=1, pad=1, dil=1, groups=1),        # conv_post of a discriminator (direct GEMV kernel)
    dict(cin=1024, cout=1, k=3, stride=1, pad=1, dil=1, groups=1),      # ... at full width: channel-split + reduce pass
    dict(cin=130, cout=3, k=15, stride=1, pad=7, dil=1, groups=1),      # few outputs, ragged channel chunks
    dict(cin=1, cout=128, k=15, stride=1, pad=7, dil=1, groups=1),      # first scale-discriminator layer (outer product)
    dict(cin=1, cout=40, k=5, stride=3, pad=2, dil=1, groups=1),        # first period-discriminator layer
]


@pytest.mark.parametrize("c", MFMA_CONVS)
@pytest.mark.parametrize("T", [101, 700])
def test_conv1d_mfma_fwd(cuda_device, c, T):
    """fp32 matrix-core implicit GEMM vs torch: exact fp32 fmaf chains, only the summation order differs."""
    from everyvoice_amd.train import ops

    g = torch.Generator().manual_seed(T)
    B = 3
    x = torch.randn(B, c["cin"], T, generator=g)
    w = torch.randn(c["cout"], c["cin"] // c["groups"], c["k"], generator=g) * 0.2
    b = torch.randn(c["cout"], generator=g)
    want = F.conv1d(x, w, b, c["stride"], c["pad"], c["dil"], c["groups"])
    got = ops.conv1d_mfma(cbt(x).to(cuda_device), w.to(cuda_device), b.to(cuda_device), c["stride"], c["pad"], c["dil"], c["groups"])
    torch.testing.assert_close(bct(got.cpu()), want, rtol=1e-4, atol=2e-5)
    # accumulate into an existing tensor on a strided output grid (the polyphase placement used for input gradients)
    base = torch.randn(c["cout"], B, want.shape[2] * 2 + 1, generator=g)
    out = base.clone().to(cuda_device)
    ops.conv1d_mfma(cbt(x).to(cuda_device), w.to(cuda_device), None, c["stride"], c["pad"], c["dil"], c["groups"], out=out,
                    n_out=want.shape[2], out_stride=2, out_offset=1, accumulate=True)
    ref = base.clone()
    ref[:, :, 1::2] += cbt(F.conv1d(x, w, None, c["stride"], c["pad"], c["dil"], c["groups"]))
    torch.testing.assert_close(out.cpu(), ref, rtol=1e-4, atol=2e-5)


@pytest.mark.parametrize("u,k", [(8, 16), (2, 4), (3, 7)])
def test_conv_transpose1d_fwd_bwd(cuda_device, u, k):
    from everyvoice_amd.train import ops

    g = torch.Generator().manual_seed(2)
    B, T, cin, cout = 2, 37, 12, 6
    p = (k - u) // 2
    x = torch.randn(B, cin, T, generator=g, requires_grad=True)
    w = (torch.randn(cin, cout, k, generator=g) * 0.2).requires_grad_()
    b = torch.randn(cout, generator=g, requires_grad=True)
    y = F.conv_transpose1d(x, w, b, u, p)
    dy = torch.randn(y.shape, generator=g)
    y.backward(dy)
    xd, wd, bd, dyd = cbt(x.detach()).to(cuda_device), w.detach().to(cuda_device), b.detach().to(cuda_device), cbt(dy).to(cuda_device)
    yg = ops.conv_transpose1d_fwd(xd, wd, bd, u, p)
    torch.testing.assert_close(bct(yg.cpu()), y.detach(), rtol=1e-4, atol=1e-5)
    dbuf = torch.zeros(cout, device=cuda_device)
    dx, dw, db = ops.conv_transpose1d_bwd(xd, wd, dyd, u, p, db_out=dbuf)
    torch.testing.assert_close(bct(dx.cpu()), x.grad, rtol=1e-4, atol=1e-5)
    torch.testing.assert_close(dw.cpu(), w.grad, rtol=1e-4, atol=1e-4)
    torch.testing.assert_close(db.cpu(), b.grad, rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("u,k,cin,cout", [(8, 16, 64, 32), (2, 4, 128, 64), (3, 7, 24, 40)])
def test_conv_transpose1d_bf16_operands(cuda_device, bf16_operands, u, k, cin, cout):
    """bf16 mode: the transposed convolution runs as the (polyphase, packed) input-gradient kernel of the strided convolution
    with the same weight tensor, its input gradient as that convolution, its weight gradient as that convolution's."""
    from everyvoice_amd.train import ops

    g = torch.Generator().manual_seed(5)
    B, T = 3, 41
    p = (k - u) // 2
    x = torch.randn(B, cin, T, generator=g)
    w = torch.randn(cin, cout, k, generator=g) * 0.2
    b = torch.randn(cout, generator=g)
    y = F.conv_transpose1d(_bf(x), _bf(w), b, u, p)
    dy = torch.randn(y.shape, generator=g)
    xd, wd, bd, dyd = cbt(x).to(cuda_device), w.to(cuda_device), b.to(cuda_device), cbt(dy).to(cuda_device)
    yg = ops.conv_transpose1d_fwd(xd, wd, bd, u, p)
    torch.testing.assert_close(bct(yg.cpu()), y, rtol=1e-4, atol=3e-5)
    dbuf = torch.zeros(cout, device=cuda_device)
    dx, dw, db = ops.conv_transpose1d_bwd(xd, wd, dyd, u, p, db_out=dbuf)
    torch.testing.assert_close(bct(dx.cpu()), F.conv1d(_bf(dy), _bf(w), None, u, p), rtol=1e-4, atol=3e-5)
    cands = [torch.nn.grad.conv1d_weight(_bf(dy), w.shape, _bf(x), u, p), torch.nn.grad.conv1d_weight(dy, w.shape, x, u, p)]
    errs = [float((dw.cpu() - c).abs().max() / c.abs().max()) for c in cands]   # bf16 or fp32 weight gradient by shape
    assert min(errs) <= 1e-4, errs
    torch.testing.assert_close(db.cpu(), dy.sum(dim=(0, 2)), rtol=1e-4, atol=1e-4)


def test_pool_period_view_activations(cuda_device):
    from everyvoice_amd.train import ops

    g = torch.Generator().manual_seed(3)
    x = torch.randn(2, 3, 50, generator=g, requires_grad=True)
    y = F.avg_pool1d(x, 4, 2, padding=2)
    dy = torch.randn(y.shape, generator=g)
    y.backward(dy)
    yg = ops.avgpool4s2(cbt(x.detach()).to(cuda_device))
    torch.testing.assert_close(bct(yg.cpu()), y.detach())
    torch.testing.assert_close(bct(ops.avgpool4s2_bwd(cbt(dy).to(cuda_device), 50).cpu()), x.grad)
    for period, T in ((2, 64), (3, 64), (5, 61), (7, 50), (11, 8192)):
        a = torch.randn(2, 1, T, generator=g, requires_grad=True)
        n_pad = (period - T % period) % period
        ap = F.pad(a, (0, n_pad), "reflect") if n_pad else a
        v = ap.view(2, 1, -1, period)  # [B, 1, H, p]
        want = v.permute(1, 0, 3, 2).reshape(1, 2 * period, -1)  # [1, B*p, H]
        got = ops.period_view(cbt(a.detach()).to(cuda_device), period)
        torch.testing.assert_close(got.cpu(), want.detach())
        dv = torch.randn(want.shape, generator=g)
        want.backward(dv)
        torch.testing.assert_close(bct(ops.period_view_bwd(dv.to(cuda_device), 2, T, period).cpu()), a.grad)
    t = torch.randn(1000, generator=g)
    torch.testing.assert_close(ops.lrelu(t.to(cuda_device), 0.1).cpu(), F.leaky_relu(t, 0.1))
    torch.testing.assert_close(ops.lrelu_bwd(t.to(cuda_device), (-t).to(cuda_device), 0.1).cpu(), t * torch.where(-t > 0, 1.0, 0.1))
    torch.testing.assert_close(ops.tanh(t.to(cuda_device)).cpu(), torch.tanh(t), rtol=1e-5, atol=1e-6)
    th = torch.tanh(t)
    torch.testing.assert_close(ops.tanh_bwd(t.to(cuda_device), th.to(cuda_device)).cpu(), t * (1 - th * th), rtol=1e-5, atol=1e-6)


def test_weight_norm_and_adamw(cuda_device):
    from everyvoice_amd.train import ops

    torch.manual_seed(4)
    for conv in (torch.nn.Conv1d(6, 10, 5), torch.nn.ConvTranspose1d(6, 4, 4, 2, padding=1)):
        wn = torch.nn.utils.weight_norm(conv)
        gw, vw = wn.weight_g.detach().clone(), wn.weight_v.detach().clone()
        w, norm = ops.weight_norm_fwd(gw.to(cuda_device), vw.to(cuda_device))
        torch.testing.assert_close(w.cpu(), wn.weight.detach(), rtol=1e-5, atol=1e-6)
        dw = torch.randn(wn.weight.shape)
        wn.weight_g.grad = wn.weight_v.grad = None
        torch._weight_norm(wn.weight_v, wn.weight_g, 0).backward(dw)
        dg, dv = torch.empty_like(gw, device=cuda_device), torch.empty_like(vw, device=cuda_device)
        ops.weight_norm_bwd(gw.to(cuda_device), vw.to(cuda_device), norm, dw.to(cuda_device), dg, dv)
        torch.testing.assert_close(dg.cpu(), wn.weight_g.grad, rtol=1e-4, atol=1e-5)
        torch.testing.assert_close(dv.cpu(), wn.weight_v.grad, rtol=1e-4, atol=1e-5)
    p = torch.nn.Parameter(torch.randn(1000))
    opt = torch.optim.AdamW([p], lr=2e-4, betas=(0.8, 0.99), eps=1e-8, weight_decay=0.01)
    pd = p.detach().clone().to(cuda_device)
    m, v = torch.zeros_like(pd), torch.zeros_like(pd)
    for step in range(1, 4):
        gr = torch.randn(1000)
        p.grad = gr.clone()
        opt.step()
        ops.adamw_step(pd, gr.to(cuda_device), m, v, 2e-4, (0.8, 0.99), 1e-8, 0.01, step)
        torch.testing.assert_close(pd.cpu(), p.detach(), rtol=1e-5, atol=1e-7)


def test_stft_frames_and_adjoint(cuda_device):
    from everyvoice_amd.train import ops

    g = torch.Generator().manual_seed(5)
    B, T, n_fft, hop = 3, 2048, 1024, 256
    x = torch.randn(B, T, generator=g, requires_grad=True)
    xp = F.pad(x.unsqueeze(1), (n_fft // 2, n_fft // 2), "reflect").squeeze(1)
    fr = xp.unfold(1, n_fft, hop)  # [B, F, n_fft]
    want = fr.permute(2, 0, 1).reshape(n_fft, -1)
    got, Fr = ops.stft_frames(x.detach().to(cuda_device), n_fft, hop)
    assert Fr == 1 + T // hop
    torch.testing.assert_close(got.cpu(), want.detach())
    d = torch.randn(want.shape, generator=g)
    want.backward(d)
    dx = ops.stft_frames_bwd(d.to(cuda_device), B, T, n_fft, hop)
    torch.testing.assert_close(dx.cpu(), x.grad, rtol=1e-5, atol=1e-5)


WGRAD_CASES = [
    # (B, T, cin, cout, k, stride, pad, dil, groups)
    (16, 300, 128, 128, 11, 1, 25, 5, 1),     # generator resblock: split over workgroups along the contraction
    (176, 10, 96, 160, 5, 1, 2, 1, 1),        # period discriminator tail: rows of 10 samples, several items per step, ragged M tile
    (22, 28, 64, 96, 5, 3, 2, 1, 1),          # strided (k,1) convolution on short rows
    (4, 513, 64, 128, 41, 4, 20, 1, 16),      # scale discriminator: grouped, stride 4, odd length
    (3, 65, 40, 1, 3, 1, 1, 1, 1),            # logit convolution (one output channel)
    (5, 333, 1, 32, 15, 1, 7, 1, 1),          # first layer (one input channel): the direct kernel of conv_cbt_direct.hip
    (64, 1366, 1, 32, 5, 3, 2, 1, 1),         # ... period discriminator's, strided: several column blocks, partials added in order
    (3, 9000, 1, 128, 15, 1, 7, 1, 1),        # ... scale discriminator's
    (2, 100, 1, 20, 16, 2, 3, 2, 1),          # ... the longest kernel it takes, dilated, channels not a multiple of its block of 8
    (2, 1000, 32, 32, 3, 1, 1, 1, 1),         # 42 channels per column tile would exceed cin: clipped tile
]


@pytest.mark.parametrize("case", WGRAD_CASES)
def test_conv1d_wgrad_mfma(cuda_device, case):
    """Implicit-GEMM weight gradient vs torch autograd (fp32 fmaf chains; only the summation order differs)."""
    from everyvoice_amd import _lib
    from everyvoice_amd.train import ops

    B, T, cin, cout, k, s, p, d, groups = case
    g = torch.Generator().manual_seed(B * 7 + T)
    x = torch.randn(B, cin, T, generator=g)
    w = (torch.randn(cout, cin // groups, k, generator=g) * 0.2).requires_grad_()
    y = F.conv1d(x, w, None, s, p, d, groups)
    dy = torch.randn(y.shape, generator=g)
    y.backward(dy)
    lib = _lib.load()
    n_out = y.shape[2]
    ws_elems = lib.evmi_conv1d_wgrad_cbt_f32_ws_elems(B, cin, T, cout, n_out, k, s, p, d, groups)
    assert ws_elems > 0
    xd, dyd = cbt(x).to(cuda_device), cbt(dy).to(cuda_device)
    ws = torch.empty(ws_elems, device=cuda_device)
    base = torch.randn(w.shape, generator=g)
    for accumulate in (0, 1):
        dw = base.clone().to(cuda_device)
        _lib.check(lib.evmi_conv1d_wgrad_cbt_f32(xd.data_ptr(), dyd.data_ptr(), dw.data_ptr(), ws.data_ptr(), ws_elems, B, cin, T, cout, n_out,
                                                 k, s, p, d, groups, accumulate, torch.cuda.current_stream().cuda_stream), "wgrad")
        want = w.grad + (base if accumulate else 0)
        scale = float(w.grad.abs().max())
        assert float((dw.cpu() - want).abs().max()) <= 2e-5 * scale + 1e-6, case


DGRAD_CASES = [
    # (B, T, cin, cout, k, stride, pad, dil, groups)
    (22, 28, 64, 96, 5, 3, 2, 1, 1),         # strided on short rows: three phases of different lengths in one launch
    (12, 9, 128, 160, 5, 3, 2, 1, 1),        # rows of 9 -> phases of 3 samples: the staged row is planned for the shortest phase
    (4, 513, 64, 128, 41, 4, 20, 1, 16),     # grouped, stride 4, odd length
    (3, 100, 48, 40, 2, 3, 0, 1, 1),         # kernel shorter than the stride: one residue class is never written (stays zero)
    (16, 300, 128, 128, 11, 1, 25, 5, 1),    # dilated stride-1
    (5, 64, 40, 72, 7, 2, 3, 1, 1),
    (4, 40, 128, 512, 5, 3, 2, 1, 1),        # deep contraction on few columns, three phases: the bf16 kernel splits K over workgroups
    (2, 33, 256, 1024, 5, 1, 2, 1, 1),       # ... and the stride-1 case (one phase)
]


@pytest.mark.parametrize("case", DGRAD_CASES)
def test_conv1d_dgrad_fused_phases(cuda_device, case):
    """evmi_conv1d_dgrad_cbt_f32 (all polyphase components in one launch) vs torch autograd."""
    from everyvoice_amd.train import ops

    B, T, cin, cout, k, s, p, d, groups = case
    g = torch.Generator().manual_seed(T * 3 + k)
    x = torch.randn(B, cin, T, generator=g, requires_grad=True)
    w = torch.randn(cout, cin // groups, k, generator=g) * 0.2
    y = F.conv1d(x, w, None, s, p, d, groups)
    dy = torch.randn(y.shape, generator=g)
    y.backward(dy)
    dx = ops.conv1d_bwd_data_mfma(cbt(dy).to(cuda_device), w.to(cuda_device), T, s, p, d, groups)
    scale = float(x.grad.abs().max())
    assert float((bct(dx.cpu()) - x.grad).abs().max()) <= 2e-5 * scale + 1e-6, case


def _bf(t):
    return t.to(torch.bfloat16).to(torch.float32)


@pytest.fixture
def bf16_operands():
    from everyvoice_amd.train import ops

    ops.CONV_BACKEND["operands"] = "bf16"
    yield
    ops.CONV_BACKEND["operands"] = "f32"


@pytest.mark.parametrize("c", MFMA_CONVS)
@pytest.mark.parametrize("T", [101, 700])
def test_conv1d_bf16_operands_fwd(cuda_device, bf16_operands, c, T):
    """bf16-operand mode: both operands rounded to bf16 (nearest even), products accumulated in fp32 -- so the oracle is the
    fp32 convolution of the ROUNDED operands and only the summation order differs (same tolerance as the fp32 test)."""
    from everyvoice_amd.train import ops

    g = torch.Generator().manual_seed(T + 7)
    B = 3
    x = torch.randn(B, c["cin"], T, generator=g)
    w = torch.randn(c["cout"], c["cin"] // c["groups"], c["k"], generator=g) * 0.2
    b = torch.randn(c["cout"], generator=g)
    # GEMV / outer-product shapes and groups of fewer than 8 channels stay on the exact fp32 kernels
    cin_g, cout_g = c["cin"] // c["groups"], c["cout"] // c["groups"]
    rounded = c["cout"] > 4 and cout_g > 4 and cin_g >= 8
    xr, wr = (_bf(x), _bf(w)) if rounded else (x, w)
    want = F.conv1d(xr, wr, b, c["stride"], c["pad"], c["dil"], c["groups"])
    got = ops.conv1d_mfma(cbt(x).to(cuda_device), w.to(cuda_device), b.to(cuda_device), c["stride"], c["pad"], c["dil"], c["groups"])
    torch.testing.assert_close(bct(got.cpu()), want, rtol=1e-4, atol=3e-5)
    # strided output grid + accumulation (the polyphase placement of input gradients)
    base = torch.randn(c["cout"], B, want.shape[2] * 2 + 1, generator=g)
    out = base.clone().to(cuda_device)
    ops.conv1d_mfma(cbt(x).to(cuda_device), w.to(cuda_device), None, c["stride"], c["pad"], c["dil"], c["groups"], out=out,
                    n_out=want.shape[2], out_stride=2, out_offset=1, accumulate=True)
    ref = base.clone()
    ref[:, :, 1::2] += cbt(F.conv1d(xr, wr, None, c["stride"], c["pad"], c["dil"], c["groups"]))
    torch.testing.assert_close(out.cpu(), ref, rtol=1e-4, atol=3e-5)
    # and within the operand-rounding distance of the exact convolution (2^-9 per operand, random signs)
    exact = F.conv1d(x, w, b, c["stride"], c["pad"], c["dil"], c["groups"])
    assert float((bct(got.cpu()) - exact).norm() / exact.norm()) < 6e-3


@pytest.mark.parametrize("case", DGRAD_CASES)
def test_conv1d_bf16_operands_dgrad(cuda_device, bf16_operands, case):
    from everyvoice_amd.train import ops

    B, T, cin, cout, k, s, p, d, groups = case
    g = torch.Generator().manual_seed(T * 3 + k + 1)
    x = torch.randn(B, cin, T, generator=g, requires_grad=True)
    w = torch.randn(cout, cin // groups, k, generator=g) * 0.2
    dy = torch.randn(F.conv1d(x, w, None, s, p, d, groups).shape, generator=g)
    # shapes outside the bf16 staging (GEMV / outer products, fewer than 16 channels per group) run the exact fp32 kernels
    F.conv1d(x, _bf(w), None, s, p, d, groups).backward(_bf(dy))
    g_bf = x.grad.clone()
    x.grad = None
    F.conv1d(x, w, None, s, p, d, groups).backward(dy)
    dx = bct(ops.conv1d_bwd_data_mfma(cbt(dy).to(cuda_device), w.to(cuda_device), T, s, p, d, groups).cpu())
    scale = float(x.grad.abs().max())
    err = min(float((dx - g_bf).abs().max()), float((dx - x.grad).abs().max()))
    assert err <= 2e-5 * scale + 1e-6, case


WGRAD_BF16_CASES = [
    # (B, T, cin, cout, k, stride, pad, dil, groups)
    (4, 300, 64, 64, 3, 1, 1, 1, 1),
    (16, 256, 128, 128, 11, 1, 25, 5, 1),     # generator resblock: two tap groups, dilation 5
    (22, 28, 64, 96, 5, 3, 2, 1, 1),          # period discriminator: stride 3, rows of 10 outputs, ragged channel tile
    (12, 9, 128, 160, 5, 3, 2, 1, 1),         # rows of 3 outputs: K steps span many items
    (3, 130, 512, 512, 41, 4, 20, 1, 8),      # scale discriminator: 41 taps in six groups, stride 4, 64 channels per group
    (5, 64, 72, 104, 7, 2, 3, 1, 1),          # channel counts that are no multiple of 32 / 64
    (2, 1000, 64, 96, 1, 1, 0, 1, 1),         # pointwise (the FastSpeech2 dense layers)
    (7, 51, 96, 256, 5, 1, 2, 1, 1),          # odd item count: rows padded to a K-step boundary
    (16, 100, 512, 384, 5, 1, 2, 1, 1),       # 196,608 (co, ci) pairs over three K splits: the LDS-transposing reduce
]


@pytest.mark.parametrize("case", WGRAD_BF16_CASES)
def test_conv1d_wgrad_bf16_packed(cuda_device, case):
    """evmi_conv1d_wgrad_cbt_bf16pk (transposing LDS reads on packed operands) vs torch's weight gradient of the ROUNDED
    operands: fp32 accumulation of exact bf16 products, only the summation order differs."""
    from everyvoice_amd import _lib

    B, T, cin, cout, k, s, p, d, groups = case
    g = torch.Generator().manual_seed(B * 11 + T)
    x = torch.randn(B, cin, T, generator=g)
    w = torch.randn(cout, cin // groups, k, generator=g)
    n_out = (T + 2 * p - d * (k - 1) - 1) // s + 1
    dy = torch.randn(B, cout, n_out, generator=g)
    want = torch.nn.grad.conv1d_weight(_bf(x), w.shape, _bf(dy), s, p, d, groups)
    lib = _lib.load()
    ws_elems = lib.evmi_conv1d_wgrad_cbt_bf16pk_ws_elems(B, cin, T, cout, n_out, k, s, p, d, groups)
    assert ws_elems > 0, case
    xd, dyd = cbt(x).to(cuda_device), cbt(dy).to(cuda_device)
    ws = torch.empty(ws_elems, device=cuda_device)
    base = torch.randn(w.shape, generator=g)
    for accumulate in (0, 1):
        dw = base.clone().to(cuda_device)
        _lib.check(lib.evmi_conv1d_wgrad_cbt_bf16pk(xd.data_ptr(), dyd.data_ptr(), dw.data_ptr(), ws.data_ptr(), ws_elems, B, cin, T, cout,
                                                    n_out, k, s, p, d, groups, accumulate, torch.cuda.current_stream().cuda_stream), "wgrad")
        ref = want + (base if accumulate else 0)
        scale = float(want.abs().max())
        assert float((dw.cpu() - ref).abs().max()) <= 3e-5 * scale + 1e-6, (case, accumulate)


BF16_EDGE_CASES = [
    # (B, T, cin, cout, k, stride, pad, dil, groups)
    (1, 7, 8, 7, 3, 1, 1, 1, 1),          # one item, one octet, ragged output rows
    (1, 1, 16, 10, 1, 1, 0, 1, 1),        # a single position
    (37, 3, 9, 33, 3, 3, 0, 1, 1),        # output length 1 per item, 37 items, 9 channels (two octets, 7 of them padding)
    (2, 50, 24, 18, 5, 1, 2, 2, 3),       # three groups of 8 channels, dilation 2
    (1, 8192, 8, 8, 7, 1, 3, 1, 1),       # one long row: 64 column tiles of one item
    (3, 40, 40, 24, 2, 3, 0, 1, 1),       # kernel shorter than the stride (one residue class of dx stays zero)
    (5, 33, 136, 72, 41, 1, 20, 1, 1),    # 41 taps on rows shorter than the kernel
    (2, 300, 64, 64, 3, 1, 1, 1, 1),
    (65, 10, 1024, 64, 5, 1, 2, 1, 1),    # many short items, deep contraction (84 ring steps)
]


@pytest.mark.parametrize("case", BF16_EDGE_CASES)
def test_bf16_packed_kernels_edge_shapes(cuda_device, bf16_operands, case):
    """Forward, input gradient and weight gradient through ops.conv1d_fwd / conv1d_bwd in bf16-operand mode vs torch on the
    rounded operands (where a shape falls outside a packed kernel the exact fp32 kernel runs: either oracle may be the match)."""
    from everyvoice_amd.train import ops

    B, T, cin, cout, k, s, p, d, groups = case
    g = torch.Generator().manual_seed(B * 13 + T + k)
    x = torch.randn(B, cin, T, generator=g)
    w = torch.randn(cout, cin // groups, k, generator=g) * 0.3
    b = torch.randn(cout, generator=g)
    n_out = (T + 2 * p - d * (k - 1) - 1) // s + 1
    dy = torch.randn(B, cout, n_out, generator=g)
    xd, wd, dyd = cbt(x).to(cuda_device), w.to(cuda_device), cbt(dy).to(cuda_device)
    yg = bct(ops.conv1d_fwd(xd, wd, b.to(cuda_device), s, p, d, groups).cpu())
    dxg, dwg, _ = ops.conv1d_bwd(xd, wd, dyd, s, p, d, groups)
    dxg, dwg = bct(dxg.cpu()), dwg.cpu()

    def closest(got, cands, tol):
        errs = [float((got - c).abs().max() / (c.abs().max() + 1e-30)) for c in cands]
        assert min(errs) <= tol, (case, errs)

    closest(yg, [F.conv1d(_bf(x), _bf(w), b, s, p, d, groups), F.conv1d(x, w, b, s, p, d, groups)], 1e-4)
    closest(dxg, [torch.nn.grad.conv1d_input(x.shape, _bf(w), _bf(dy), s, p, d, groups),
                  torch.nn.grad.conv1d_input(x.shape, w, dy, s, p, d, groups)], 1e-4)
    closest(dwg, [torch.nn.grad.conv1d_weight(_bf(x), w.shape, _bf(dy), s, p, d, groups),
                  torch.nn.grad.conv1d_weight(x, w.shape, dy, s, p, d, groups)], 1e-4)


# ---- the packed kernels at the shapes bench.py times (every instantiation a profile shows must have met torch) ----
# (name, B, T, cin, cout, k, stride, pad, dil, groups, forward tile, input-gradient tile, weight-gradient taps): the expected
# instantiations are what evmi_conv1d_*_bf16pk_plan reports with no EVMI_PK_* switch set (tile index: see include/evmi.h)
BENCH_SHAPE_CASES = [
    # (weight-gradient taps 5 = wgrad_pk_kernel<5, false, true>: the 128 x 128 tile on eight waves, round 6)
    ("fs2 postnet 512->512 k5 on 32 x 814", 32, 814, 512, 512, 5, 1, 2, 1, 1, 6, 6, 5),        # conv_pk_kernel<128, 256>
    ("fs2 ffn 256->1024 on 32 x 814", 32, 814, 256, 1024, 1, 1, 0, 1, 1, 8, 9, 5),             # the eight-wave <128, 128> for short contractions
    ("fs2 ffn 1024->256 on 32 x 814", 32, 814, 1024, 256, 1, 1, 0, 1, 1, 9, 8, 5),             # 9: <128, 128> with the weight fragments in registers
    ("fs2 postnet 80->512 k5 on 32 x 947", 32, 947, 80, 512, 5, 1, 2, 1, 1, 9, 9, 8),          # dgrad: split-K <128, 128>
    ("fs2 encoder 256->768 on 32 x 187", 32, 187, 256, 768, 1, 1, 0, 1, 1, 1, 2, 4),           # <64, 128> / <64, 64>
    ("gan generator c32 k11 d5 on 16 x 8192", 16, 8192, 32, 32, 11, 1, 25, 5, 1, 3, 3, 8),     # <32, 128>
    ("gan generator c64 k7 d3 on 16 x 4096", 16, 4096, 64, 64, 7, 1, 9, 3, 1, 1, 1, 8),        # <64, 128>
    ("gan generator c128 k3 on 16 x 2048", 16, 2048, 128, 128, 3, 1, 1, 1, 1, 1, 1, 4),
    ("gan generator c512 k11 d5 on 16 x 32", 16, 32, 512, 512, 11, 1, 25, 5, 1, 9, 9, 8),      # split-K over workgroups
    ("mpd p2 32->128 s3 on 64 x 4096", 64, 4096, 32, 128, 5, 3, 2, 1, 1, 9, 3, 8),
    ("mpd p11 1024->1024 on 352 x 28", 352, 28, 1024, 1024, 5, 1, 2, 1, 1, 9, 9, 5),           # short items: <128, 128>
    ("msd 512->1024 k41 s4 g16 on 32 x 512", 32, 512, 512, 1024, 41, 4, 20, 1, 16, 1, 3, 8),
    ("msd 128->128 k41 s2 g4 on 32 x 8192", 32, 8192, 128, 128, 41, 2, 20, 1, 4, 3, 3, 8),
]


_BENCH_REF: dict = {}


def _bench_shape_reference(case, with_exact):
    """Seeded operands of a bench shape and torch's answers on the rounded operands (and, for the forced-tile runs, on the exact ones: a
    FORCED tile that cannot stage a shape leaves it to the exact fp32 kernels) -- computed once per shape and process."""
    name, B, T, cin, cout, k, s, p, d, groups = case[:10]
    ent = _BENCH_REF.get(name)
    if ent is None:
        n_out = (T + 2 * p - d * (k - 1) - 1) // s + 1
        g = torch.Generator().manual_seed(B * 7 + T + k)
        x = torch.randn(B, cin, T, generator=g)
        w = torch.randn(cout, cin // groups, k, generator=g) * (2.0 / (cin // groups * k) ** 0.5)
        b = torch.randn(cout, generator=g)
        dy = torch.randn(B, cout, n_out, generator=g)
        ent = _BENCH_REF[name] = dict(x=x, w=w, b=b, dy=dy, want_y=F.conv1d(_bf(x), _bf(w), b, s, p, d, groups),
                                      want_dx=torch.nn.grad.conv1d_input(x.shape, _bf(w), _bf(dy), s, p, d, groups),
                                      want_dw=torch.nn.grad.conv1d_weight(_bf(x), w.shape, _bf(dy), s, p, d, groups), exact=None)
    if with_exact and ent["exact"] is None:
        x, w, b, dy = ent["x"], ent["w"], ent["b"], ent["dy"]
        ent["exact"] = (F.conv1d(x, w, b, s, p, d, groups), torch.nn.grad.conv1d_input(x.shape, w, dy, s, p, d, groups),
                        torch.nn.grad.conv1d_weight(x, w.shape, dy, s, p, d, groups))
    return ent["x"], ent["w"], ent["b"], ent["dy"], ent["want_y"], ent["want_dx"], ent["want_dw"], ent["exact"] if with_exact else (None, None, None)


@pytest.mark.parametrize("case", BENCH_SHAPE_CASES, ids=[c[0] for c in BENCH_SHAPE_CASES])
def test_bf16_packed_kernels_at_bench_shapes(cuda_device, bf16_operands, case):
    """Forward, input gradient and weight gradient of the packed bf16 kernels at the layer shapes bench.py's training legs run
    (batch 32 x <= 947 frames for FastSpeech2, 16 x 8192 samples for the GAN step) against torch on the rounded operands -- the
    planner's tile and split-K choices depend on the column count, so the small-shape tests above do not reach these
    instantiations.  The test also asserts WHICH instantiation the planner picks (unless a switch forces one: the child runs of
    test_packed_conv_every_tile_forced), so that a kernel named in profiles/*_kernel_stats.csv is one a test has compared."""
    import os

    from everyvoice_amd import _lib
    from everyvoice_amd.train import ops

    name, B, T, cin, cout, k, s, p, d, groups, tile_f, tile_d, taps_w = case
    lib = _lib.load()
    n_out = (T + 2 * p - d * (k - 1) - 1) // s + 1
    if not any(os.environ.get(v) for v in ("EVMI_PK_TILE", "EVMI_PK_SPLITK", "EVMI_PK_WIDE", "EVMI_PK_ADIR", "EVMI_PK_XCD_HB", "EVMI_WG_WIDE")):
        geo = (B, cin, T, cout, n_out, k, s, p, d, groups)
        assert lib.evmi_conv1d_cbt_bf16pk_plan(*geo) % 16 == tile_f, name
        assert lib.evmi_conv1d_dgrad_cbt_bf16pk_plan(*geo) % 16 == tile_d, name
        assert lib.evmi_conv1d_wgrad_cbt_bf16pk_plan(*geo) % 16 == taps_w, name
    forced = bool(os.environ.get("EVMI_PK_TILE"))
    x, w, b, dy, want_y, want_dx, want_dw, exact = _bench_shape_reference(case, forced)
    xd, wd, dyd = cbt(x).to(cuda_device), w.to(cuda_device), cbt(dy).to(cuda_device)
    y = bct(ops.conv1d_fwd(xd, wd, b.to(cuda_device), s, p, d, groups).cpu())
    dx, dw, _ = ops.conv1d_bwd(xd, wd, dyd, s, p, d, groups)
    for got, want, want_exact, what in ((y, want_y, exact[0], "forward"), (bct(dx.cpu()), want_dx, exact[1], "input gradient"),
                                        (dw.cpu(), want_dw, exact[2], "weight gradient")):
        err = float((got - want).abs().max() / want.abs().max())
        if forced:
            err = min(err, float((got - want_exact).abs().max() / want_exact.abs().max()))
        assert err <= 1e-4, (name, what, err)  # fp32 accumulation of exact bf16 products: summation order only


_PK_CHILD_TESTS = "bf16_packed_kernels_at_bench_shapes or bf16_packed_kernels_edge_shapes"
_PK_SWITCHES = ["EVMI_PK_SPLITK=0", "EVMI_PK_WIDE=1", "EVMI_WG_SPLITS=1", "EVMI_WG_NST=2", "EVMI_PK_ADIR=0", "EVMI_PK_XCD_HB=1", "EVMI_WG_WIDE=0"]


@pytest.mark.parametrize("tile", range(11))
def test_packed_conv_every_tile_forced(cuda_device, bf16_operands, monkeypatch, tile):
    """conv_pk_kernel<128,128 | 64,128 | 64,64 | 32,128 | 64,256 | 32,256 | 128,256>, the eight-wave <128,256> / <128,128> (indices 7, 8) and
    <128,128> with the weight fragments in registers (index 9), <32,512> (index 10): the planner picks one per shape; here every one
    of them is FORCED through the bf16 comparisons with torch of this file -- the bench shapes, the edge shapes and the strided /
    grouped input-gradient shapes -- so a tile the planner starts choosing tomorrow (as <128, 256> was switched on at the end of
    round 2) has already met the oracle.  EVMI_PK_TILE is read at every planning call (csrc/conv_cbt_bf16_pk.hip), so the tiles are
    forced in THIS process (round 4 started a child process per tile: 220 s of the GPU suite, most of it torch's reference
    convolutions recomputed eleven times -- they are cached per shape now).  Shapes a forced tile cannot stage fall back to the exact
    fp32 kernels, which the comparisons accept (closest of the two oracles)."""
    monkeypatch.setenv("EVMI_PK_TILE", str(tile))
    for case in BENCH_SHAPE_CASES:
        test_bf16_packed_kernels_at_bench_shapes(cuda_device, bf16_operands, case)
    for case in BF16_EDGE_CASES:
        test_bf16_packed_kernels_edge_shapes(cuda_device, bf16_operands, case)
    for case in DGRAD_CASES:
        test_conv1d_bf16_operands_dgrad(cuda_device, bf16_operands, case)


@pytest.mark.parametrize("switch", _PK_SWITCHES)
def test_packed_conv_planner_switches(switch):
    """The planner's other A/B switches (no split-K, 256-column tiles for narrow layers, unsplit / two-slot weight gradients, weight
    fragments through the LDS everywhere, the plain m-tile-major XCD order) are read ONCE per process:
    the same comparisons in a child process each."""
    from helpers import child_pytest_results

    jobs = {}
    for sw in _PK_SWITCHES:
        key, val = sw.split("=")
        jobs[sw] = ([__file__, "-q", "-x", "-k", _PK_CHILD_TESTS + " or conv1d_wgrad_bf16_packed"], {key: val})
    rc, out = child_pytest_results("train_ops_pk", jobs, parallel=3, timeout=600)[switch]
    assert rc == 0, out


def test_conv_kernels_edge_shapes(cuda_device):
    """Single item, single output position, output length 1 per item with many items, channels not a multiple of anything."""
    from everyvoice_amd.train import ops

    g = torch.Generator().manual_seed(99)
    for (B, T, cin, cout, k, s, p, d, groups) in ((1, 7, 5, 7, 3, 1, 1, 1, 1), (1, 1, 6, 10, 1, 1, 0, 1, 1), (37, 3, 9, 33, 3, 3, 0, 1, 1),
                                                  (2, 50, 6, 9, 5, 1, 2, 2, 3), (1, 8192, 2, 2, 7, 1, 3, 1, 1)):
        x = torch.randn(B, cin, T, generator=g, requires_grad=True)
        w = (torch.randn(cout, cin // groups, k, generator=g) * 0.3).requires_grad_()
        b = torch.randn(cout, generator=g)
        y = F.conv1d(x, w, b, s, p, d, groups)
        dy = torch.randn(y.shape, generator=g)
        y.backward(dy)
        xd, wd, dyd = cbt(x.detach()).to(cuda_device), w.detach().to(cuda_device), cbt(dy).to(cuda_device)
        yg = ops.conv1d_fwd(xd, wd, b.to(cuda_device), s, p, d, groups)
        torch.testing.assert_close(bct(yg.cpu()), y.detach(), rtol=1e-4, atol=1e-5)
        dx, dw, _ = ops.conv1d_bwd(xd, wd, dyd, s, p, d, groups)
        torch.testing.assert_close(bct(dx.cpu()), x.grad, rtol=1e-4, atol=1e-5)
        torch.testing.assert_close(dw.cpu(), w.grad, rtol=1e-4, atol=1e-4)


def test_allreduce_bucket_c_abi_on_a_world_of_one(cuda_device):
    """evmi_allreduce_bucket (the C-ABI form of the data-parallel gradient exchange): RCCL resolved at run time, a one-rank
    communicator, sum + scale in place on the caller's stream.  (N > 1 needs N GPUs; the torch.distributed path is covered by
    the gloo tests and the one-rank "nccl" tests.)"""
    import ctypes as C

    from everyvoice_amd import _lib

    lib = _lib.load()
    uid = C.create_string_buffer(128)
    _lib.check(lib.evmi_comm_unique_id(uid), "evmi_comm_unique_id")
    comm = C.c_void_p()
    torch.cuda.set_device(cuda_device)
    _lib.check(lib.evmi_comm_init_rank(C.byref(comm), 1, uid, 0), "evmi_comm_init_rank")
    try:
        g = torch.arange(1000, dtype=torch.float32, device=cuda_device)
        _lib.check(lib.evmi_allreduce_bucket(comm, g[100:].data_ptr(), 900, 0.5, _lib.current_stream_ptr(cuda_device)), "evmi_allreduce_bucket")
        torch.cuda.synchronize()
        want = torch.arange(1000, dtype=torch.float32)
        want[100:] *= 0.5
        assert torch.equal(g.cpu(), want)
    finally:
        _lib.check(lib.evmi_comm_destroy(comm), "evmi_comm_destroy")


@pytest.mark.parametrize("B,T,cin,cout", [(4, 336, 96, 160), (32, 814, 256, 1024), (3, 64, 1024, 256), (2, 32, 40, 48)])
def test_pointwise_layers_pack_their_operands_once(cuda_device, bf16_operands, B, T, cin, cout):
    """k = 1 / stride 1 layers on the packed bf16 kernels: the forward's packed input and the input gradient's packed dy are in the
    layout the weight gradient reads (conv_pk_common.h: tight items, rows that end on one of its K steps), so `conv1d_fwd(..., keep=d)` + `conv1d_bwd(..., packed=d)`
    pack every operand once.  Same kernels on the same packed bits: the results must be IDENTICAL to the path that packs again, and
    match torch on the rounded operands."""
    from everyvoice_amd.train import ops

    g = torch.Generator().manual_seed(B * 131 + T)
    x = torch.randn(B, cin, T, generator=g)
    w = torch.randn(cout, cin, 1, generator=g) * 0.1
    bias = torch.randn(cout, generator=g)
    dy = torch.randn(B, cout, T, generator=g)
    xd, wd, bd, dyd = cbt(x).to(cuda_device), w.to(cuda_device), bias.to(cuda_device), cbt(dy).to(cuda_device)
    assert ops.shares_packed(B, T, 1, 1, 0, 1, 1) and not ops.shares_packed(B, T, 3, 1, 1, 1, 1) and not ops.shares_packed(B, T, 1, 2, 0, 1, 1)
    assert not ops.shares_packed(5, 333, 1, 1, 0, 1, 1)  # rows that do not end on a K step of the weight gradient: packed twice, as before
    keep = {}
    y_keep = ops.conv1d_fwd(xd, wd, bd, 1, 0, 1, 1, keep=keep)
    y_plain = ops.conv1d_fwd(xd, wd, bd, 1, 0, 1, 1)
    assert "x_packed" in keep and torch.equal(y_keep, y_plain)
    dw_a, dw_b = torch.zeros_like(wd), torch.zeros_like(wd)
    db_a, db_b = torch.zeros_like(bd), torch.zeros_like(bd)
    dx_a, _, _ = ops.conv1d_bwd(xd, wd, dyd, 1, 0, 1, 1, need_dx=True, dw_out=dw_a, db_out=db_a, accumulate=True, packed=keep)
    assert "dy_packed" in keep
    dx_b, _, _ = ops.conv1d_bwd(xd, wd, dyd, 1, 0, 1, 1, need_dx=True, dw_out=dw_b, db_out=db_b, accumulate=True)
    ops.wgrad_join(cuda_device)
    torch.cuda.synchronize()
    assert torch.equal(dx_a, dx_b) and torch.equal(dw_a, dw_b) and torch.equal(db_a, db_b)
    # without an input gradient only x arrives packed
    dw_c = torch.zeros_like(wd)
    ops.conv1d_bwd(xd, wd, dyd, 1, 0, 1, 1, need_dx=False, dw_out=dw_c, accumulate=True, packed={"x_packed": keep["x_packed"]})
    torch.cuda.synchronize()
    assert torch.equal(dw_c, dw_a)
    # and against torch on the bf16-rounded operands
    xr, wr, dyr = x.bfloat16().float(), w.bfloat16().float().requires_grad_(), dy.bfloat16().float()
    F.conv1d(xr, wr, None).backward(dyr)
    scale = float(wr.grad.abs().max())
    assert float((dw_a.cpu() - wr.grad).abs().max()) <= 1e-4 * scale + 1e-5
