"""A minimal reverse-mode tape over the libevmi_hip training operators.

``Var`` carries a CBT activation and its (lazily accumulated) gradient; every functional op appends a
closure to the tape that, given the output gradient, accumulates into its inputs' gradients and into the
parameter-gradient buffers of the layer it used.  No torch autograd, no torch math: torch only owns the
memory.  This is the host-side glue the reference gets from PyTorch's autograd engine.
"""

from __future__ import annotations

import torch

from . import ops


_ACTIVATION_ELEMS = [0]


def activation_elements(reset: bool = False) -> int:
    """Elements of every activation tensor (Var) created since the last reset: what bench.py prices a step's algorithmic HBM
    traffic with (each activation written once and read once forward; read once more, its gradient written and read, backward)."""
    n = _ACTIVATION_ELEMS[0]
    if reset:
        _ACTIVATION_ELEMS[0] = 0
    return n


class Var:
    __slots__ = ("data", "grad", "needs_grad")

    def __init__(self, data: torch.Tensor, needs_grad: bool = True):
        self.data = data
        self.grad = None
        self.needs_grad = needs_grad
        _ACTIVATION_ELEMS[0] += data.numel()

    def accumulate(self, g: torch.Tensor) -> None:
        if not self.needs_grad:
            return
        if self.grad is None:
            self.grad = g
        else:
            ops.axpby(1.0, self.grad, 1.0, g, out=self.grad)


class _Cut:
    """A point of the tape where backward may be interrupted (``Tape.backward_segments``); ``join`` runs before the interruption."""

    def __init__(self, join=None, tag=None):
        self.join, self.tag = join, tag


class Tape:
    def __init__(self):
        self._ops = []

    def record(self, fn) -> None:
        self._ops.append(fn)

    def cut(self, join=None, tag=None) -> None:
        """Mark this point: ``backward_segments`` stops here (after everything recorded LATER has run its backward).  Used where a
        gradient bucket becomes final and its all-reduce -- which is not part of a captured HIP graph -- has to be issued.
        ``tag`` is handed to the consumer (``backward_segments`` yields it), e.g. the bucket's range in the flat buffer."""
        self._ops.append(_Cut(join, tag))

    def backward(self) -> None:
        for fn in reversed(self._ops):
            if not isinstance(fn, _Cut):
                fn()
        ops.wgrad_join()  # weight-gradient kernels queued beside this chain (ops.side_wgrad) are part of this backward
        self._ops.clear()

    def backward_segments(self, stop=None):
        """Generator form: every ``next()`` runs the backward up to the next cut (weight-gradient kernels queued beside the chain
        are joined first, then the cut's own ``join``) and yields the cut's tag; the last one runs to the start of the tape and
        ends the generator.  ``stop(tag) -> bool`` chooses the cuts to honour (default: all)."""
        for fn in reversed(self._ops):
            if isinstance(fn, _Cut):
                if stop is not None and not stop(fn.tag):
                    continue
                ops.wgrad_join()
                if fn.join is not None:
                    fn.join()
                yield fn.tag
            else:
                fn()
        ops.wgrad_join()
        self._ops.clear()


def conv1d(tape: Tape, x: Var, layer, training: bool = True) -> Var:
    """``layer`` supplies hyper-parameters, bias, the effective weight and the sink of its gradient."""
    w, dw_sink = layer.effective(training)
    db_sink = layer.call_db_sink()
    y = Var(ops.conv1d_fwd(x.data, w, layer.bias_data(), layer.stride, layer.pad, layer.dil, layer.groups))

    def bwd():
        if y.grad is None:
            return
        dx, _, _ = ops.conv1d_bwd(x.data, w, y.grad, layer.stride, layer.pad, layer.dil, layer.groups, need_dx=x.needs_grad,
                                  dw_out=dw_sink, db_out=db_sink, accumulate=True, need_dw=not layer.frozen)
        if dx is not None:
            x.accumulate(dx)

    tape.record(bwd)
    return y


def conv1d_lrelu(tape: Tape, x: Var, layer, slope: float, training: bool = True) -> Var:
    """leaky_relu(conv1d(x)) with the activation in the convolution's epilogue: the pre-activation is never stored; the
    backward takes its sign from the output (same sign for slope > 0)."""
    w, dw_sink = layer.effective(training)
    db_sink = layer.call_db_sink()
    y = Var(ops.conv1d_fwd(x.data, w, layer.bias_data(), layer.stride, layer.pad, layer.dil, layer.groups, lrelu_slope=slope))

    def bwd():
        if y.grad is None:
            return
        if layer.frozen:  # input gradient only: the activation's backward rides in the pack of dy
            if x.needs_grad:
                x.accumulate(ops.conv1d_fused_dgrad(y.grad, w, x.data.shape[2], layer.stride, layer.pad, layer.dil, layer.groups, dy_mask=y.data,
                                                    dy_mask_slope=slope, x_for_fallback=x.data))
            return
        else:  # activation backward and bias gradient in one pass over dy
            dpre, db_out = ops.lrelu_bwd_rowsum(y.grad, y.data, slope, db_sink, accumulate=True), None
        dx, _, _ = ops.conv1d_bwd(x.data, w, dpre, layer.stride, layer.pad, layer.dil, layer.groups, need_dx=x.needs_grad,
                                  dw_out=dw_sink, db_out=db_out, accumulate=True, need_dw=not layer.frozen)
        if dx is not None:
            x.accumulate(dx)

    tape.record(bwd)
    return y


def resblock_pair(tape: Tape, x: Var, c1, c2, slope: float, training: bool = True) -> Var:
    """x + conv2(leaky_relu(conv1(leaky_relu(x)))) -- one residual pair of a HiFi-GAN ResBlock1 -- with the activations and the
    residual taken into the convolutions' packs and epilogues (ops.conv1d_fused_*): forward = 2 convolution calls; backward =
    bias row sums + 2 weight gradients + 2 input gradients, the second of which delivers the pair's complete input gradient
    (activation backward and skip path included).  SURVEY.md 8b names this fusion evmi_resblock1_fused_{fwd,bwd}."""
    w1, dw1 = c1.effective(training)
    db1 = c1.call_db_sink()
    w2, dw2 = c2.effective(training)
    db2 = c2.call_db_sink()
    t = ops.conv1d_fused_fwd(x.data, w1, c1.bias_data(), c1.stride, c1.pad, c1.dil, c1.groups, act=ops.ACT_LRELU, act_param=slope, pre_slope=slope)
    y = Var(ops.conv1d_fused_fwd(t, w2, c2.bias_data(), c2.stride, c2.pad, c2.dil, c2.groups, residual=x.data))

    def bwd():
        dy = y.grad
        if dy is None:
            return
        C2, N2 = dy.shape[0], dy.shape[1] * dy.shape[2]
        if not c2.frozen:
            ops.side_wgrad(dy, db2).run(lambda: ops.row_reduce(0, dy, None, db2, C2, N2, accumulate=True))
            ops.conv1d_fused_wgrad(t, w2.shape, dy, dw2, c2.stride, c2.pad, c2.dil, c2.groups)
        dt = ops.conv1d_fused_dgrad(dy, w2, t.shape[2], c2.stride, c2.pad, c2.dil, c2.groups, x_for_fallback=t)
        if c1.frozen:
            dpre = ops.lrelu_bwd(dt, t, slope)
        else:  # backward of the activation between the two convolutions + conv1's bias gradient, one pass
            dpre = ops.lrelu_bwd_rowsum(dt, t, slope, db1, accumulate=True)
            ops.conv1d_fused_wgrad(x.data, w1.shape, dpre, dw1, c1.stride, c1.pad, c1.dil, c1.groups, x_pre_slope=slope)
        if x.needs_grad:
            x.accumulate(ops.conv1d_fused_dgrad(dpre, w1, x.data.shape[2], c1.stride, c1.pad, c1.dil, c1.groups, dx_mask=x.data, dx_mask_slope=slope,
                                                residual=dy, x_for_fallback=x.data))

    tape.record(bwd)
    return y


def conv_transpose1d(tape: Tape, x: Var, layer, training: bool = True) -> Var:
    w, dw_sink = layer.effective(training)
    y = Var(ops.conv_transpose1d_fwd(x.data, w, layer.bias_data(), layer.stride, layer.pad))

    def bwd():
        if y.grad is None:
            return
        dx, _, _ = ops.conv_transpose1d_bwd(x.data, w, y.grad, layer.stride, layer.pad, need_dx=x.needs_grad,
                                            dw_out=dw_sink, db_out=layer.db_sink(), accumulate=True)
        if dx is not None:
            x.accumulate(dx)

    tape.record(bwd)
    return y


def lrelu(tape: Tape, x: Var, slope: float) -> Var:
    y = Var(ops.lrelu(x.data, slope))
    tape.record(lambda: y.grad is not None and x.accumulate(ops.lrelu_bwd(y.grad, x.data, slope)))
    return y


def tanh(tape: Tape, x: Var) -> Var:
    y = Var(ops.tanh(x.data))
    tape.record(lambda: y.grad is not None and x.accumulate(ops.tanh_bwd(y.grad, y.data)))
    return y


def add(tape: Tape, a: Var, b: Var) -> Var:
    y = Var(ops.axpby(1.0, a.data, 1.0, b.data))

    def bwd():
        if y.grad is None:
            return
        a.accumulate(y.grad)
        b.accumulate(ops.copy(y.grad) if a.needs_grad and a.grad is y.grad else y.grad)

    tape.record(bwd)
    return y


def scale(tape: Tape, x: Var, s: float) -> Var:
    y = Var(ops.elementwise(ops.EW_SCALE, x.data, p0=s))
    tape.record(lambda: y.grad is not None and x.accumulate(ops.elementwise(ops.EW_SCALE, y.grad, p0=s)))
    return y


def avgpool4s2(tape: Tape, x: Var) -> Var:
    y = Var(ops.avgpool4s2(x.data), needs_grad=x.needs_grad)  # (a view of a gradient-free input needs none: the first convolution
    t_in = x.data.shape[2]                                      #  behind it skips its input gradient)
    tape.record(lambda: y.grad is not None and x.needs_grad and x.accumulate(ops.avgpool4s2_bwd(y.grad, t_in)))
    return y


def period_view(tape: Tape, x: Var, period: int) -> Var:
    _, B, T = x.data.shape
    y = Var(ops.period_view(x.data, period), needs_grad=x.needs_grad)
    tape.record(lambda: y.grad is not None and x.needs_grad and x.accumulate(ops.period_view_bwd(y.grad, B, T, period)))
    return y


def reflect_pad_left1(tape: Tape, x: Var) -> Var:
    """torch.nn.ReflectionPad1d((1, 0)): the iSTFTNet head pads one frame in front of conv_post."""
    y = Var(ops.reflect_pad_left1(x.data))
    tape.record(lambda: y.grad is not None and x.accumulate(ops.reflect_pad_left1_bwd(y.grad)))
    return y


class ISTFTConstants:
    """Windowed inverse-DFT basis as transposed-convolution weights [n_fft + 2, 1, n_fft] (real rows then imaginary rows; hann
    window, 1/N and the factor 2 of the mirrored bins folded in) and the reciprocal window envelope of torch.istft(center=True)."""

    def __init__(self, n_fft: int, hop: int, device):
        import math

        self.n_fft, self.hop, self.device = n_fft, hop, device
        H = n_fft // 2 + 1
        n = torch.arange(n_fft, dtype=torch.float64)
        win = torch.hann_window(n_fft, dtype=torch.float64)  # periodic, as torch.hann_window's default
        w = torch.zeros(2 * H, 1, n_fft, dtype=torch.float64)
        for h in range(H):
            c = (1.0 if h in (0, n_fft // 2) else 2.0) / n_fft
            ang = 2.0 * math.pi * h * n / n_fft
            w[h, 0] = c * torch.cos(ang) * win
            w[H + h, 0] = -c * torch.sin(ang) * win
        self.weight = w.to(torch.float32).to(device)
        self._win_sq = (win * win).to(torch.float32)
        self._inv_env = {}

    def inv_envelope(self, B: int, frames: int) -> torch.Tensor:
        key = (B, frames)
        if key not in self._inv_env:
            n_fft, hop = self.n_fft, self.hop
            env = torch.zeros(n_fft + hop * (frames - 1))
            for t in range(frames):
                env[t * hop : t * hop + n_fft] += self._win_sq
            env = env[n_fft // 2 : n_fft // 2 + hop * (frames - 1)]
            self._inv_env[key] = (1.0 / env).reshape(1, 1, -1).repeat(1, B, 1).contiguous().to(self.device)
        return self._inv_env[key]


def istft(tape: Tape, x: Var, consts: ISTFTConstants) -> Var:
    """x [n_fft + 2, B, frames] (conv_post output) -> wav [1, B, hop * (frames - 1)] = istft(exp(x_lo) * exp(i sin(x_hi)))."""
    H = consts.n_fft // 2 + 1
    _, B, frames = x.data.shape
    s = ops.istft_polar(x.data, H)
    inv_env = consts.inv_envelope(B, frames)
    raw = ops.conv_transpose1d_fwd(s, consts.weight, None, consts.hop, consts.n_fft // 2)
    y = Var(ops.elementwise(ops.EW_MUL, raw, inv_env))

    def bwd():
        if y.grad is None:
            return
        dyr = ops.elementwise(ops.EW_MUL, y.grad, inv_env)
        ds, _, _ = ops.conv_transpose1d_bwd(s, consts.weight, dyr, consts.hop, consts.n_fft // 2, need_dx=True, need_dw=False)
        x.accumulate(ops.istft_polar_bwd(x.data, ds, H))

    tape.record(bwd)
    return y
