"""Parameters and (weight- / spectral-) normalised convolution layers of the training path.

Parameters of one optimiser live in one flat fp32 buffer (``ParamGroup``): AdamW is a single kernel over it
and data-parallel training all-reduces one contiguous gradient buffer per optimiser (the generator side and
the discriminator side of the GAN step separately).  State-dict names follow upstream
(``conv_pre.weight_g`` / ``weight_v`` / ``bias``, ``...weight_orig`` / ``weight_u`` / ``weight_v`` for the
spectral-norm discriminator) so reference checkpoints map one to one.
"""

from __future__ import annotations

import math

import torch

from . import ops


class ParamGroup:
    def __init__(self, device):
        self.device = device
        self._specs = []  # (name, shape, offset)
        self._export = {}  # name -> shape in an exported state dict, where it differs
        self._n = 0
        self.flat = self.grad = self.m = self.v = self.step_dev = None
        self._step = 0
        self._views = {}

    def declare(self, name: str, shape, export_shape=None) -> int:
        """``export_shape``: the tensor shape upstream stores under this name when it differs from the one the kernels use
        (the period discriminators' Conv2d((k, 1)) weights are [c_out, c_in, k, 1] upstream, [c_out, c_in, k] here)."""
        n = math.prod(shape)
        if export_shape is not None:
            assert math.prod(export_shape) == n
            self._export[name] = tuple(export_shape)
        self._specs.append((name, tuple(shape), self._n))
        self._n += (n + 3) // 4 * 4  # keep every tensor 16-byte aligned
        return len(self._specs) - 1

    def finalize(self):
        self.flat = torch.zeros(self._n, device=self.device, dtype=torch.float32)
        self.grad = torch.zeros_like(self.flat)
        self.m = torch.zeros_like(self.flat)
        self.v = torch.zeros_like(self.flat)
        self.step_dev = torch.zeros(1, device=self.device, dtype=torch.int32)  # the step counter the optimiser kernel reads
        self._index = {name: i for i, (name, _, _) in enumerate(self._specs)}

    def _view(self, buf, i):
        # views of the two flat buffers the layers ask for on every call are made once (a slice + view costs ~2 us, ~950 of them
        # per FastSpeech2 step); the buffers themselves are never re-allocated after finalize()
        key = (id(buf), i)
        v = self._views.get(key)
        if v is None or v[0] is not buf:
            _, shape, off = self._specs[i]
            v = self._views[key] = (buf, buf[off : off + math.prod(shape)].view(shape))
        return v[1]

    def data(self, i):
        return self._view(self.flat, i)

    def gradient(self, i):
        return self._view(self.grad, i)

    def names(self):
        return [s[0] for s in self._specs]

    def offset_of(self, name: str) -> int:
        """Start of a parameter in the flat buffers (parameters are laid out in declaration order)."""
        return self._specs[self._index[name]][2]

    def numel(self):
        return sum(math.prod(s[1]) for s in self._specs)

    def state_dict(self):
        return {name: self._view(self.flat, i).detach().clone().reshape(self._export.get(name, shape))
                for i, (name, shape, _) in enumerate(self._specs)}

    def gradients(self):
        return {name: self._view(self.grad, i) for i, (name, _, _) in enumerate(self._specs)}

    def load(self, name: str, value: torch.Tensor):
        self._view(self.flat, self._index[name]).copy_(value.to(self.device, torch.float32).reshape(self._specs[self._index[name]][1]))

    def zero_grad(self):
        ops.fill_(self.grad, 0.0)

    @property
    def step(self) -> int:
        """Optimiser steps taken.  The optimiser kernel reads the DEVICE copy (``step_dev``); assigning here (checkpoint
        restore) sets both, the step itself only bumps the host mirror next to the device-side increment it launches."""
        return self._step

    @step.setter
    def step(self, value: int):
        self._step = int(value)
        if self.step_dev is not None:
            self.step_dev.fill_(self._step)

    def set_step(self, step: int):
        self.step = step

    OPTIMIZERS = {"adamw": 0, "adam": 1, "rms": 2}

    def optimizer_step(self, name="adamw", lr=1e-4, betas=(0.9, 0.98), eps=1e-8, weight_decay=0.01, alpha=0.99, clip=0.0, lr_dev=None):
        """One step of the configured optimiser (the reference's union: AdamOptimizer / AdamWOptimizer / RMSOptimizer,
        everyvoice/.schema/everyvoice-spec-to-wav-0.5.json:434-622) over the flat buffer; ``clip`` > 0 clamps the updated
        parameters (WGAN).  The step number lives on the device, so a captured HIP graph of the step replays correctly."""
        from .. import _lib

        kind = self.OPTIMIZERS[name]
        self._step += 1
        lib = _lib.load()
        st = _lib.current_stream_ptr(self.flat.device)
        _lib.check(lib.evmi_counter_add_i32(self.step_dev.data_ptr(), 1, st), "evmi_counter_add_i32")
        b1, b2 = (alpha, 0.0) if kind == 2 else betas
        if lr_dev is not None:  # a scheduled rate the caller stored on the device (graph replays pick up the new value)
            _lib.check(lib.evmi_optimizer_step_lrdev_f32(kind, self.flat.data_ptr(), self.grad.data_ptr(), self.m.data_ptr(), self.v.data_ptr(),
                                                         self.flat.numel(), lr_dev.data_ptr(), b1, b2, eps, weight_decay, self.step_dev.data_ptr(),
                                                         float(clip), st), "evmi_optimizer_step_lrdev_f32")
            return
        _lib.check(lib.evmi_optimizer_step_f32(kind, self.flat.data_ptr(), self.grad.data_ptr(), self.m.data_ptr(), self.v.data_ptr(),
                                               self.flat.numel(), lr, b1, b2, eps, weight_decay, self._step, self.step_dev.data_ptr(),
                                               float(clip), st), "evmi_optimizer_step_f32")

    def adamw(self, lr, betas, eps, weight_decay, lr_dev=None):
        self.optimizer_step("adamw", lr, betas, eps, weight_decay, lr_dev=lr_dev)


class _ConvBase:
    transposed = False

    def __init__(self, group: ParamGroup, name, cin, cout, k, stride=1, pad=0, dil=1, groups=1, transposed=False, conv2d=False):
        self.group, self.name = group, name
        self.conv2d = conv2d  # upstream module is Conv2d((k, 1)): exported tensors carry the trailing unit axis
        self.cin, self.cout, self.k = cin, cout, k
        self.stride, self.pad, self.dil, self.groups = stride, pad, dil, groups
        self.transposed = transposed
        self.wshape = (cin, cout, k) if transposed else (cout, cin // groups, k)
        self.i_bias = group.declare(name + ".bias", (cout,))
        self._dw = None
        self.frozen = False  # True: backward propagates to the input only (discriminators during the generator step)

    def bias_data(self):
        return self.group.data(self.i_bias)

    def param_names(self):
        return [n for n in self.group.names() if n.startswith(self.name + ".")]

    def db_sink(self):
        return self.group.gradient(self.i_bias)

    def call_db_sink(self):
        """Bias-gradient sink of the forward call `effective` just served (weight-normed layers: the parameter gradient itself)."""
        return self.group.gradient(self.i_bias)

    def _shared_sink(self):
        """Gradient wrt the EFFECTIVE weight accumulates here during backward; ``finish_grads`` maps it to the
        stored parameters (g, v) or weight_orig."""
        if self._dw is None:
            self._dw = ops.zeros(*self.wshape, device=self.group.device)
        return self._dw


class WNConv(_ConvBase):
    """weight_norm(Conv1d / ConvTranspose1d): parameters weight_g [rows,1,1], weight_v [rows, ...], bias."""

    def __init__(self, group, name, cin, cout, k, **kw):
        super().__init__(group, name, cin, cout, k, **kw)
        rows = self.wshape[0]
        self.i_g = group.declare(name + ".weight_g", (rows, 1, 1), (rows, 1, 1, 1) if self.conv2d else None)
        self.i_v = group.declare(name + ".weight_v", self.wshape, (*self.wshape, 1) if self.conv2d else None)
        self._w = self._norm = None

    def materialize(self):
        """Recompute w = g * v / ||v|| from the current parameters (once per optimiser step)."""
        self._w, self._norm = ops.weight_norm_fwd(self.group.data(self.i_g), self.group.data(self.i_v))

    def effective(self, training=True):
        """(effective weight, gradient sink) for one forward call."""
        if self._w is None:
            self.materialize()
        return self._w, self._shared_sink()

    def finish_grads(self):
        if self._dw is None:
            return
        ops.weight_norm_bwd(self.group.data(self.i_g), self.group.data(self.i_v), self._norm, self._dw,
                            self.group.gradient(self.i_g), self.group.gradient(self.i_v))
        ops.fill_(self._dw, 0.0)


class WNBatch:
    """All weight-normed layers of one optimiser: effective weights, norms and gradient sinks live in three flat buffers, so
    ``materialize`` is ONE launch per optimiser step and ``finish`` one launch per gradient bucket (instead of one per layer)."""

    def __init__(self, group: ParamGroup, layers):
        from .. import _lib

        self.group = group
        self.layers = sorted([l for l in layers if isinstance(l, WNConv)], key=lambda l: group.offset_of(l.name + ".weight_g"))
        L = len(self.layers)
        rows = [l.wshape[0] for l in self.layers]
        n = [math.prod(l.wshape[1:]) for l in self.layers]
        row_start, w_off, norm_off = [0], [0], [0]
        for r, k in zip(rows, n):
            row_start.append(row_start[-1] + r)
            w_off.append(w_off[-1] + (r * k + 3) // 4 * 4)
            norm_off.append(norm_off[-1] + r)
        dev = group.device
        self.eff = torch.zeros(max(w_off[-1], 1), device=dev, dtype=torch.float32)
        self.dw_eff = torch.zeros_like(self.eff)
        self.norms = torch.ones(max(norm_off[-1], 1), device=dev, dtype=torch.float32)
        tab = torch.zeros(6, L + 1, dtype=torch.int64)
        tab[0] = torch.tensor(row_start)
        for i, l in enumerate(self.layers):
            tab[1, i] = n[i]
            tab[2, i] = group.offset_of(l.name + ".weight_g")
            tab[3, i] = group.offset_of(l.name + ".weight_v")
            tab[4, i] = w_off[i]
            tab[5, i] = norm_off[i]
            l._w = self.eff[w_off[i] : w_off[i] + rows[i] * n[i]].view(l.wshape)
            l._dw = self.dw_eff[w_off[i] : w_off[i] + rows[i] * n[i]].view(l.wshape)
            l._norm = self.norms[norm_off[i] : norm_off[i] + rows[i]]
            l._batch, l._bi = self, i
        self.table = tab.to(dev)
        self.row_start = row_start
        self._lib = _lib

    def materialize(self):
        if not self.layers:
            return
        g = self.group
        self._lib.check(self._lib.load().evmi_weight_norm_fwd_batched_f32(g.flat.data_ptr(), self.eff.data_ptr(), self.norms.data_ptr(),
                                                                       self.table.data_ptr(), len(self.layers), 0, self.row_start[-1],
                                                                       self._lib.current_stream_ptr(g.device)), "evmi_weight_norm_fwd_batched_f32")

    def finish(self, layers):
        """Parameter gradients of a gradient bucket's weight-normed layers (a contiguous range in declaration order)."""
        idx = sorted(l._bi for l in layers if getattr(l, "_batch", None) is self)
        if not idx:
            return
        assert idx == list(range(idx[0], idx[-1] + 1)), "a gradient bucket must be a contiguous layer range"
        g = self.group
        self._lib.check(self._lib.load().evmi_weight_norm_bwd_batched_f32(g.flat.data_ptr(), g.grad.data_ptr(), self.norms.data_ptr(),
                                                                       self.dw_eff.data_ptr(), self.table.data_ptr(), len(self.layers),
                                                                       self.row_start[idx[0]], self.row_start[idx[-1] + 1],
                                                                       self._lib.current_stream_ptr(g.device)), "evmi_weight_norm_bwd_batched_f32")


class SNConv(_ConvBase):
    """spectral_norm(Conv1d) as torch.nn.utils.spectral_norm: parameter weight_orig, buffers weight_u / weight_v;
    in training mode every forward call runs one power iteration (and therefore sees its own sigma).

    The effective weights depend on (weight_orig, u) only, not on activations: ``prepare(n)`` runs the power iterations of the
    next n forward calls ahead of time (each layer's chain of small matrix-vector kernels is independent of the other layers',
    so the trainer runs them side by side, off the convolutions' critical path); ``effective`` then hands them out in order.
    Every call owns its weight- and bias-gradient buffers, so two calls (real / generated waveform) may run their backward on
    different streams; ``finish_grads`` adds them into the parameter gradients in call order."""

    def __init__(self, group, name, cin, cout, k, **kw):
        super().__init__(group, name, cin, cout, k, **kw)
        self.i_w = group.declare(name + ".weight_orig", self.wshape)
        h, wdt = self.wshape[0], math.prod(self.wshape[1:])
        self.u = torch.zeros(h, device=group.device)
        self.v = torch.zeros(wdt, device=group.device)
        self._calls = []  # (sigma tensor [1], u, v, dw buffer, db buffer) per forward call of this step
        self._ready = []  # prepared, not yet handed out
        self._held = []   # everything prepare() allocated, kept alive until release(): see prepare()
        self._call_db = None

    def materialize(self):
        pass  # the effective weight depends on the power-iteration state: computed per forward call

    def _iterate(self, u_prev, training):
        """One forward call's (w, sigma, u, v) from the state u_prev: one power iteration in training mode."""
        W = self.group.data(self.i_w)
        h, wdt = self.wshape[0], math.prod(self.wshape[1:])
        Wm = W.view(h, wdt)
        dev = W.device
        sigma = torch.empty(1, device=dev)
        if training:
            tmp_v = torch.empty(wdt, device=dev)
            ops.gemm(Wm, u_prev.view(h, 1), tmp_v.view(wdt, 1), ta=True)   # W^T u
            v = ops.normalize_vec(tmp_v, torch.empty(wdt, device=dev))
            wv = torch.empty(h, device=dev)
            ops.gemm(Wm, v.view(wdt, 1), wv.view(h, 1))                     # W v
            u = ops.normalize_vec(wv, torch.empty(h, device=dev))
        else:
            u, v = u_prev, self.v
            wv = torch.empty(h, device=dev)
            ops.gemm(Wm, v.view(wdt, 1), wv.view(h, 1))
        ops.row_reduce(1, u, wv, sigma, 1, h)                               # sigma = u . (W v), left on the device
        w = ops.elementwise(ops.EW_DIV_SCALAR, W, c=sigma)
        return w, sigma, u, v

    def prepare(self, n_calls: int, training=True):
        """Power iterations + effective weights of the next `n_calls` forward calls (the buffers end at the last call's state)."""
        # The tensors made here are allocated on the CURRENT stream but consumed by convolutions on other streams (the
        # discriminator's chains).  The caching allocator hands a freed block back to its allocation stream at once, so dropping
        # the last reference while a consumer stream still reads it would let the next allocation here overwrite live data:
        # they stay referenced in `_held` until the trainer calls release() behind the join that ends the phase.
        u = self.u
        v = self.v
        for _ in range(n_calls):
            w, sigma, u, v = self._iterate(u, training)
            self._ready.append((w, sigma, u, v))
            self._held.append((w, sigma, u, v))
        if training and n_calls:
            ops.copy(u, out=self.u)
            ops.copy(v, out=self.v)

    def effective(self, training=True):
        """One forward call: (one power iteration in training mode,) sigma = u^T W v, w = W / sigma."""
        if not self._ready:
            self.prepare(1, training)
        w, sigma, u, v = self._ready.pop(0)
        dw = db = None
        if not self.frozen:
            dw = ops.zeros(*self.wshape, device=w.device)
            db = ops.zeros(self.cout, device=w.device)
        self._calls.append((sigma, u, v, dw, db))
        self._call_db = db
        return w, dw

    def call_db_sink(self):
        """Bias-gradient sink of the forward call `effective` just served (read right after it)."""
        return self._call_db

    def release(self):
        """Drop the prepared tensors (call once every stream that used them has been joined)."""
        self._held.clear()

    def finish_grads(self):
        from .. import _lib

        W = self.group.data(self.i_w)
        gW = self.group.gradient(self.i_w)
        gb = self.group.gradient(self.i_bias)
        h, wdt = self.wshape[0], math.prod(self.wshape[1:])
        for sigma, u, v, dw, db in self._calls:
            if dw is None:
                continue
            # d weight_orig += dw / sigma - (<dw, W> / sigma^2) u v^T   (sigma and the inner product stay on the device)
            dot = torch.empty(1, device=W.device)
            ops.scalar_reduce(2, ops.elementwise(ops.EW_MUL, dw, W), None, dot)
            _lib.check(_lib.load().evmi_spectral_norm_grad_f32(gW.data_ptr(), dw.data_ptr(), u.data_ptr(), v.data_ptr(), sigma.data_ptr(),
                                                               dot.data_ptr(), h, wdt, _lib.current_stream_ptr(W.device)), "evmi_spectral_norm_grad_f32")
            ops.axpby(1.0, gb, 1.0, db, out=gb)
        self._calls.clear()


def kaiming_uniform_conv_init_(layer: _ConvBase, gen: torch.Generator, std: float | None = None):
    """torch's default Conv1d init (kaiming_uniform a=sqrt(5)) or N(0, std) like upstream init_weights, expressed on
    (g, v): v = w0, g = ||w0|| per row, so the effective weight starts as w0 exactly like torch's weight_norm."""
    shape = layer.wshape
    fan_in = (shape[1] if not layer.transposed else shape[0]) * shape[2] if not layer.transposed else shape[1] * shape[2]
    bound = 1.0 / math.sqrt(fan_in)
    if std is None:
        w0 = (torch.rand(shape, generator=gen) * 2 - 1) * bound
    else:
        w0 = torch.randn(shape, generator=gen) * std
    b0 = (torch.rand(layer.cout, generator=gen) * 2 - 1) * bound
    g = layer.group
    g.load(layer.name + ".bias", b0)
    if isinstance(layer, WNConv):
        g.load(layer.name + ".weight_v", w0)
        g.load(layer.name + ".weight_g", w0.reshape(shape[0], -1).norm(dim=1).reshape(shape[0], 1, 1))
    else:
        g.load(layer.name + ".weight_orig", w0)
        layer.u.copy_(torch.nn.functional.normalize(torch.randn(shape[0], generator=gen), dim=0))
        layer.v.copy_(torch.nn.functional.normalize(torch.randn(math.prod(shape[1:]), generator=gen), dim=0))
