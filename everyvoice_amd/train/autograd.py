"""A minimal reverse-mode tape over the libevmi_hip training operators.

``Var`` carries a CBT activation and its (lazily accumulated) gradient; every functional op appends a
closure to the tape that, given the output gradient, accumulates into its inputs' gradients and into the
parameter-gradient buffers of the layer it used.  No torch autograd, no torch math: torch only owns the
memory.  This is the host-side glue the reference gets from PyTorch's autograd engine.
"""

from __future__ import annotations

import torch

from . import ops


class Var:
    __slots__ = ("data", "grad", "needs_grad")

    def __init__(self, data: torch.Tensor, needs_grad: bool = True):
        self.data = data
        self.grad = None
        self.needs_grad = needs_grad

    def accumulate(self, g: torch.Tensor) -> None:
        if not self.needs_grad:
            return
        if self.grad is None:
            self.grad = g
        else:
            ops.axpby(1.0, self.grad, 1.0, g, out=self.grad)


class Tape:
    def __init__(self):
        self._ops = []

    def record(self, fn) -> None:
        self._ops.append(fn)

    def backward(self) -> None:
        for fn in reversed(self._ops):
            fn()
        self._ops.clear()


def conv1d(tape: Tape, x: Var, layer, training: bool = True) -> Var:
    """``layer`` supplies hyper-parameters, bias, the effective weight and the sink of its gradient."""
    w, dw_sink = layer.effective(training)
    y = Var(ops.conv1d_fwd(x.data, w, layer.bias_data(), layer.stride, layer.pad, layer.dil, layer.groups))

    def bwd():
        if y.grad is None:
            return
        dx, _, _ = ops.conv1d_bwd(x.data, w, y.grad, layer.stride, layer.pad, layer.dil, layer.groups, need_dx=x.needs_grad,
                                  dw_out=dw_sink, db_out=layer.db_sink(), accumulate=True, need_dw=not layer.frozen)
        if dx is not None:
            x.accumulate(dx)

    tape.record(bwd)
    return y


def conv1d_lrelu(tape: Tape, x: Var, layer, slope: float, training: bool = True) -> Var:
    """leaky_relu(conv1d(x)) with the activation in the convolution's epilogue: the pre-activation is never stored; the
    backward takes its sign from the output (same sign for slope > 0)."""
    w, dw_sink = layer.effective(training)
    y = Var(ops.conv1d_fwd(x.data, w, layer.bias_data(), layer.stride, layer.pad, layer.dil, layer.groups, lrelu_slope=slope))

    def bwd():
        if y.grad is None:
            return
        dpre = ops.lrelu_bwd(y.grad, y.data, slope)
        dx, _, _ = ops.conv1d_bwd(x.data, w, dpre, layer.stride, layer.pad, layer.dil, layer.groups, need_dx=x.needs_grad,
                                  dw_out=dw_sink, db_out=layer.db_sink(), accumulate=True, need_dw=not layer.frozen)
        if dx is not None:
            x.accumulate(dx)

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
        b.accumulate(y.grad.clone() if a.needs_grad and a.grad is y.grad else y.grad)

    tape.record(bwd)
    return y


def scale(tape: Tape, x: Var, s: float) -> Var:
    y = Var(ops.elementwise(ops.EW_SCALE, x.data, p0=s))
    tape.record(lambda: y.grad is not None and x.accumulate(ops.elementwise(ops.EW_SCALE, y.grad, p0=s)))
    return y


def avgpool4s2(tape: Tape, x: Var) -> Var:
    y = Var(ops.avgpool4s2(x.data))
    t_in = x.data.shape[2]
    tape.record(lambda: y.grad is not None and x.accumulate(ops.avgpool4s2_bwd(y.grad, t_in)))
    return y


def period_view(tape: Tape, x: Var, period: int) -> Var:
    _, B, T = x.data.shape
    y = Var(ops.period_view(x.data, period))
    tape.record(lambda: y.grad is not None and x.accumulate(ops.period_view_bwd(y.grad, B, T, period)))
    return y
