"""FastSpeech2 feature-prediction TRAINING on libevmi_hip (BASELINE config 3; SURVEY.md 8a F1-F4, 8b driver contract).

The reference trains ``fs2.model.FastSpeech2`` (absent submodule FastSpeech2_lightning) under Lightning:
``train_base_command(model_config, data_module, model, monitor, ...)`` (``everyvoice/base_cli/helpers.py:173-195``) owns the
Trainer; the module owns the optimiser, the step and ``on_save_checkpoint``.  ``FastSpeech2Trainer`` is that module-side
contract without Lightning: ``training_step(batch) -> losses`` (forward in training mode, every loss, backward, gradient
all-reduce, clipping, Noam-scheduled AdamW), ``state_dict()`` with the state-dict names ``everyvoice_amd.fs2.FastSpeech2``
loads, ``checkpoint()`` with ``model_info`` and JSON-only hyper-parameters (``everyvoice/tests/test_model.py:85-151``).

Teacher forcing as FastSpeech2 trains: ground-truth durations drive the length regulator, ground-truth (phone-level) pitch
and energy are bucketised into the embeddings, the predictors are regressed on them; loss weights are the reference's
``FastSpeech2TrainingConfig`` defaults (``everyvoice/.schema/everyvoice-text-to-spec-0.5.json``: mel 1.0, postnet 1.0, pitch /
energy / duration 0.1; optimiser ``noam``: lr 1e-3, betas (0.9, 0.999), eps 1e-8, weight decay 1e-6, warm-up 1000).
Durations come from the dataset (``learn_alignment: false``) or from the alignment module's monotonic search
(``everyvoice_amd.heavy.maximum_path``); the gradient of the alignment losses themselves is not part of this step yet.

Everything is channel-major fp32 ``x[c][b][t]``; no torch autograd and no torch math in the step.
"""

from __future__ import annotations

import math
import os
from dataclasses import asdict, dataclass, field

import torch

from .. import _lib
from ..fs2 import N_PHONOLOGICAL_FEATURES, FastSpeech2ModelConfig, Stats
from . import ops
from .autograd import _ACTIVATION_ELEMS, Tape, Var
from .layers import ParamGroup, WNBatch, WNConv


@dataclass
class NoamOptimizerConfig:
    """``everyvoice/config/shared_types.py:311-320`` with the FastSpeech2TrainingConfig defaults."""
    learning_rate: float = 1e-3
    eps: float = 1e-8
    weight_decay: float = 1e-6
    betas: tuple = (0.9, 0.999)
    name: str = "noam"
    warmup_steps: int = 1000


@dataclass
class FastSpeech2TrainingConfig:
    """``FastSpeech2TrainingConfig`` (everyvoice/.schema/everyvoice-text-to-spec-0.5.json:499-551) over the driver-side fields of
    ``BaseTrainingConfig`` (everyvoice/config/shared_types.py:180-258: what ``train_base_command`` reads, helpers.py:234-259)."""
    batch_size: int = 16
    optimizer: NoamOptimizerConfig = field(default_factory=NoamOptimizerConfig)
    mel_loss_weight: float = 1.0
    postnet_loss_weight: float = 1.0
    pitch_loss_weight: float = 0.1
    energy_loss_weight: float = 0.1
    duration_loss_weight: float = 0.1
    attn_ctc_loss_weight: float = 0.1
    attn_bin_loss_weight: float = 0.1
    attn_bin_loss_warmup_epochs: int = 100
    gradient_clip_val: float | None = 1.0
    # -- BaseTrainingConfig --
    save_top_k_ckpts: int = 5
    ckpt_steps: int | None = None
    ckpt_epochs: int | None = 1
    val_check_interval: int | float | None = 500
    check_val_every_n_epoch: int | None = None
    max_epochs: int = 1000
    max_steps: int = 100000
    finetune_checkpoint: object = None   # Path | None
    training_filelist: object = "path/to/your/preprocessed/training_filelist.psv"
    validation_filelist: object = "path/to/your/preprocessed/validation_filelist.psv"
    filelist_loader: str = "everyvoice.utils.generic_psv_filelist_reader"
    logger: object = None                # config.LoggerConfig
    val_data_workers: int = 0
    train_data_workers: int = 4
    use_weighted_sampler: bool = False

    def __post_init__(self):
        from pathlib import Path

        from ..config import LoggerConfig

        if isinstance(self.optimizer, dict):
            od = dict(self.optimizer)
            od["betas"] = tuple(od.get("betas", (0.9, 0.999)))
            self.optimizer = NoamOptimizerConfig(**od)
        if self.logger is None or isinstance(self.logger, dict):
            self.logger = LoggerConfig(**(self.logger or {}))
        if self.ckpt_epochs is not None and self.ckpt_steps is not None:
            raise ValueError("ckpt_epochs and ckpt_steps have to be mutually exclusive")
        for name in ("finetune_checkpoint", "training_filelist", "validation_filelist"):
            v = getattr(self, name)
            if v is not None and not isinstance(v, Path):
                setattr(self, name, Path(v))

    def json_dict(self, paths: bool = False) -> dict:
        """JSON-only form; ``paths=False`` drops every path-valued field and the logger (checkpoints travel between machines:
        shared_types.py:56-88, tests/test_model.py:85-151)."""
        from pathlib import Path

        out = {}
        for k, v in self.__dict__.items():
            if k == "logger":
                if paths:
                    out[k] = v.model_dump(mode="json")
            elif isinstance(v, Path):
                if paths:
                    out[k] = str(v)
            elif k == "optimizer":
                out[k] = {**asdict(v), "betas": list(v.betas)}
            else:
                out[k] = v
        return out


def _s(t):
    return _lib.current_stream_ptr(t.device)


def _chk(rc, what):
    _lib.check(rc, what)


# ---- parameter holders (names = the reference state dict) ---------------------------------------------------------------
class Dense:
    """Unnormalised Conv1d / Linear weights: the gradient of the effective weight IS the parameter gradient."""
    stride, dil, groups, frozen, transposed = 1, 1, 1, False, False

    def __init__(self, group: ParamGroup, wname: str, bname: str, cin: int, cout: int, k: int = 1, linear: bool = False):
        self.group, self.cin, self.cout, self.k = group, cin, cout, k
        self.pad = (k - 1) // 2
        self.wshape = (cout, cin, k)
        self.i_w = group.declare(wname, (cout, cin) if linear else (cout, cin, k))
        self.i_bias = group.declare(bname, (cout,)) if bname else None  # bname None: a bias-free layer

    def effective(self, training=True):
        return self.group.data(self.i_w).view(self.wshape), self.group.gradient(self.i_w).view(self.wshape)

    def bias_data(self):
        return None if self.i_bias is None else self.group.data(self.i_bias)

    def db_sink(self):
        return None if self.i_bias is None else self.group.gradient(self.i_bias)

    def materialize(self):
        pass

    def finish_grads(self):
        pass


class Affine:
    """LayerNorm / BatchNorm scale and shift."""

    def __init__(self, group: ParamGroup, prefix: str, C: int, batchnorm: bool = False):
        self.group, self.C = group, C
        self.i_g = group.declare(prefix + ".weight", (C,))
        self.i_b = group.declare(prefix + ".bias", (C,))
        self.prefix = prefix
        if batchnorm:
            self.running_mean = torch.zeros(C, device=group.device)
            self.running_var = torch.ones(C, device=group.device)
            self.batches = 0

    def gamma(self):
        return self.group.data(self.i_g)

    def beta(self):
        return self.group.data(self.i_b)

    def dgamma(self):
        return self.group.gradient(self.i_g)

    def dbeta(self):
        return self.group.gradient(self.i_b)


class Table:
    def __init__(self, group: ParamGroup, name: str, rows: int, D: int):
        self.group, self.rows, self.D = group, rows, D
        self.i = group.declare(name, (rows, D))

    def data(self):
        return self.group.data(self.i)

    def grad(self):
        return self.group.gradient(self.i)


# ---- tape operators ---------------------------------------------------------------------------------------------------
def _row(t):
    """[C, B, T] -> [C, 1, B * T] (a view): a position-wise layer sees all its columns as ONE item, the form in which the packed bf16
    kernels share one packed copy of every operand between forward, input gradient and weight gradient for any B * T
    (csrc/conv_pk_common.h: pk_shared_items) -- with B items that needs B * T to be a multiple of 64."""
    return t if (t.shape[1] == 1 or not ops._packed()) else t.reshape(t.shape[0], 1, -1)


def _held_tensors(fns, depth: int = 4) -> list:
    """Every tensor the closures `fns` (tape operators) can reach through their cells: Vars, dicts, lists, nested closures.  A branch
    that runs on another stream keeps this list until the main chain has joined it: an operator that drops a tensor when it has
    LAUNCHED its kernels (`packed.clear()`, a Var going out of scope) hands the block back to the pool of the stream that allocated it,
    and that stream's next allocation may write it while the branch's kernels still read it."""
    out, seen = [], set()

    def visit(o, d):
        if id(o) in seen or d < 0:
            return
        seen.add(id(o))
        if torch.is_tensor(o):
            out.append(o)
        elif isinstance(o, Var):
            visit(o.data, d)
            visit(o.grad, d)
        elif isinstance(o, dict):
            for v in o.values():
                visit(v, d - 1)
        elif isinstance(o, (list, tuple)):
            for v in o:
                visit(v, d - 1)
        elif callable(o) and getattr(o, "__closure__", None):
            for c in o.__closure__:
                try:
                    visit(c.cell_contents, d - 1)
                except ValueError:  # (an empty cell)
                    pass

    for f in fns:
        visit(f, depth)
    return out


def _items(B, T):
    """(items, columns per item) as `_row` hands a [C, B, T] tensor to a position-wise layer."""
    return (1, B * T) if ops._packed() else (B, T)


_EVAL = [False]  # FastSpeech2Trainer.evaluate: dropout off, BatchNorm on its running statistics (and not updating them)


def dense(tape: Tape, x: Var, layer, act=ops.ACT_NONE) -> Var:
    """conv1d / Linear (+ ReLU or tanh in the epilogue: their backward only needs the output)."""
    assert act in (ops.ACT_NONE, ops.ACT_RELU, ops.ACT_TANH)
    w, dw_sink = layer.effective(True)
    packed = {}  # pointwise layers on the packed bf16 kernels: x is packed once (here), dy once (input gradient), for all three products
    shape = x.data.shape
    row = (lambda t: _row(t)) if layer.k == 1 else (lambda t: t)  # (pointwise: one item)
    y = Var(ops.conv1d_fwd(row(x.data), w, layer.bias_data(), 1, layer.pad, 1, 1, act=act, keep=packed).view(-1, shape[1], shape[2] + 2 * layer.pad - layer.k + 1))

    def bwd():
        if y.grad is None:
            return
        dy = y.grad
        if act == ops.ACT_RELU:
            dy = ops.elementwise(ops.EW_RELU_BWD, dy, y.data)
        elif act == ops.ACT_TANH:
            dy = ops.tanh_bwd(dy, y.data)
        dx, _, _ = ops.conv1d_bwd(row(x.data), w, row(dy), 1, layer.pad, 1, 1, need_dx=x.needs_grad, dw_out=dw_sink, db_out=layer.db_sink(), accumulate=True,
                                  packed=packed)
        packed.clear()
        if dx is not None:
            x.accumulate(dx.view(shape))

    tape.record(bwd)
    return y


def silu(tape: Tape, x: Var) -> Var:
    y = Var(ops.elementwise(ops.EW_SILU, x.data))
    tape.record(lambda: y.grad is not None and x.accumulate(ops.elementwise(ops.EW_SILU_BWD, y.grad, x.data)))
    return y


def glu(tape: Tape, p: Var) -> Var:
    D = p.data.shape[0] // 2
    y = Var(ops.elementwise(ops.EW_GLU, p.data[:D], p.data[D:]))
    tape.record(lambda: y.grad is not None and p.accumulate(ops.glu_bwd(p.data, y.grad)))
    return y


def _ln_bwd_into(x: Var, ln: Affine, dh) -> None:
    """x.grad (+)= LayerNorm's backward of dh.  Where x already has a gradient -- the residual path's, in every Conformer sub-layer -- the
    kernel adds to it in place (one read more instead of an add pass over three tensors: 49 launches per FastSpeech2 step)."""
    if not x.needs_grad:
        return
    g = x.grad
    if g is not None and g.shape == x.data.shape and g.is_contiguous() and g.dtype == torch.float32 and g.is_cuda:
        ops.layernorm_bwd(x.data, ln.gamma(), dh, ln.dgamma(), ln.dbeta(), acc_into=g)
    else:
        x.accumulate(ops.layernorm_bwd(x.data, ln.gamma(), dh, ln.dgamma(), ln.dbeta()))


def layernorm(tape: Tape, x: Var, ln: Affine) -> Var:
    y = Var(ops.layernorm(x.data, ln.gamma(), ln.beta()))
    tape.record(lambda: y.grad is not None and _ln_bwd_into(x, ln, y.grad))
    return y


def batchnorm(tape: Tape, x: Var, bn: Affine, act=ops.ACT_NONE) -> Var:
    if _EVAL[0]:
        return Var(ops.batchnorm_fwd(x.data, bn.gamma(), bn.beta(), bn.running_mean, bn.running_var, act, momentum=-1.0)[0])
    out, mean, rstd = ops.batchnorm_fwd(x.data, bn.gamma(), bn.beta(), bn.running_mean, bn.running_var, act)
    bn.batches += 1
    y = Var(out)
    tape.record(lambda: y.grad is not None and x.accumulate(ops.batchnorm_bwd(x.data, bn.gamma(), bn.beta(), mean, rstd, y.grad, bn.dgamma(), bn.dbeta(), act)))
    return y


def dwconv(tape: Tape, x: Var, layer) -> Var:
    """Depthwise convolution; ``layer``: Dense or WNConv with wshape (C, 1, k)."""
    w, dw_sink = layer.effective(True)
    k = layer.k
    y = Var(ops.dwconv_fwd(x.data, w, layer.bias_data(), k))

    def bwd():
        if y.grad is None:
            return
        dx = ops.dwconv_bwd(x.data, w, y.grad, dw_sink, layer.db_sink(), k, need_dx=x.needs_grad)
        if dx is not None:
            x.accumulate(dx)

    tape.record(bwd)
    return y


def dropout(tape: Tape, x: Var, p: float, seed: int) -> Var:
    if p <= 0.0:
        return x
    y = Var(ops.dropout(x.data, p, seed))
    tape.record(lambda: y.grad is not None and x.accumulate(ops.dropout(y.grad, p, seed)))
    return y


def residual_dropout(tape: Tape, a: Var, b: Var, p: float, seed: int, sb: float = 1.0) -> Var:
    """a + sb * dropout(b, p): one pass forward, one pass for b's gradient (sb * dropout(dy) with the same mask)."""
    if p <= 0.0:
        return residual(tape, a, b, sb)
    y = Var(ops.dropout_fused(1, b.data, a.data, p, seed, sb))

    def bwd():
        if y.grad is None:
            return
        b.accumulate(ops.dropout_fused(4, y.grad, None, p, seed, sb))
        a.accumulate(y.grad)

    tape.record(bwd)
    return y


def dense_residual_dropout(tape: Tape, a: Var, h: Var, layer, p: float, seed: int, sb: float = 1.0) -> Var:
    """a + sb * dropout(dense(h), p) as ONE tape operator.  On the packed bf16 kernels the residual add and the mask sit in the
    layer's epilogue (ops.conv1d_fwd_resdrop) and, in the backward, sb * dropout(dy) is formed while dy is packed for the layer's
    three products (ops.conv1d_bwd_dropout_dy): per site one elementwise launch and one fp32 tensor write + read less in each
    direction (32 sites per FastSpeech2 step).  Elsewhere: the two operators it stands for."""
    C, B, T = h.data.shape
    if p <= 0.0 or _EVAL[0] or layer.k != 1 or not ops.resdrop_fused_supported(*_items(B, T), C, layer.cout):
        return residual_dropout(tape, a, dense(tape, h, layer), p, seed, sb)
    w, dw_sink = layer.effective(True)
    packed = {}
    y = Var(ops.conv1d_fwd_resdrop(_row(h.data), w, layer.bias_data(), _row(a.data), p, seed, sb, packed).view(a.data.shape))

    def bwd():
        if y.grad is None:
            return
        dh = ops.conv1d_bwd_dropout_dy(_row(h.data), w, _row(y.grad), p, seed, sb, dw_sink, layer.db_sink(), packed)
        packed.clear()
        h.accumulate(dh.view(h.data.shape))
        a.accumulate(y.grad)

    tape.record(bwd)
    return y


def silu_dropout(tape: Tape, x: Var, p: float, seed: int) -> Var:
    """dropout(silu(x), p) in one pass; backward dropout(dy) * silu'(x) in one pass."""
    if p <= 0.0:
        return silu(tape, x)
    y = Var(ops.dropout_fused(2, x.data, None, p, seed))
    tape.record(lambda: y.grad is not None and x.accumulate(ops.dropout_fused(3, y.grad, x.data, p, seed)))
    return y


def ln_dense(tape: Tape, x: Var, ln: Affine, layer) -> Var:
    """dense(LayerNorm(x)) as one tape operator: on the packed bf16 kernels the normalised tensor is written straight into the layer's
    packed input (ops.layernorm_dense_fwd) and never exists in fp32 -- its only other reader, the layer's weight gradient, takes the
    packed copy; LayerNorm's own backward needs x, not its output.  Elsewhere: the two operators."""
    C, B, T = x.data.shape
    if not ops.ln_dense_fused_supported(*_items(B, T), C, layer.cout) or layer.k != 1:
        return dense(tape, layernorm(tape, x, ln), layer)
    w, dw_sink = layer.effective(True)
    packed = {}
    y = Var(ops.layernorm_dense_fwd(_row(x.data), ln.gamma(), ln.beta(), w, layer.bias_data(), packed).view(-1, B, T))

    def bwd():
        if y.grad is None:
            return
        # (x.data stands in for the normalised tensor, which was never stored: only its shape is read -- the weight gradient takes the packed copy)
        dh, _, _ = ops.conv1d_bwd(_row(x.data), w, _row(y.grad), 1, 0, 1, 1, need_dx=True, dw_out=dw_sink, db_out=layer.db_sink(), accumulate=True, packed=packed,
                                   x_standin=True)
        packed.clear()
        _ln_bwd_into(x, ln, dh.view(x.data.shape))

    tape.record(bwd)
    return y


def ffn_core(tape: Tape, x: Var, ln: Affine, l1, l2, p: float, seed: int, res: Var | None = None, seed_out: int = 0, sb: float = 1.0) -> Var:
    """dense2(dropout(silu(dense1(LayerNorm(x))), p)): a Conformer feed-forward block up to its residual add as ONE tape operator.  On the
    packed bf16 kernels LayerNorm writes the first layer's packed input, the activation and the mask are applied while the second
    layer's input is packed, and in the backward while the first layer's output gradient is packed: neither the normalised tensor nor
    dropout(silu(a)) nor its gradient is stored in fp32 (per direction two elementwise passes over the block's widest tensor and one
    fp32 copy of it less).  Elsewhere: the four operators it stands for.
    ``res``: the block's residual input -- the operator then returns res + sb * dropout(that, p) with the add and the mask (seed_out) in
    the second layer's epilogue and sb * dropout(dy) formed while dy is packed (dense_residual_dropout's fusion)."""
    C, B, T = x.data.shape
    N = B * T  # (position-wise layers: all columns as one item, _row)
    if p <= 0.0 or _EVAL[0] or not ops.ffn_fused_supported(*_items(B, T), l1.cout, l2.cout):
        h = dense(tape, silu_dropout(tape, ln_dense(tape, x, ln, l1), p, seed), l2)
        return h if res is None else residual_dropout(tape, res, h, p, seed_out, sb)
    w1, dw1 = l1.effective(True)
    w2, dw2 = l2.effective(True)
    xr = _row(x.data)
    if res is not None and ops.ffn_packed_supported(*_items(B, T), C, l1.cout, l2.cout):
        # the whole block as a packed chain: the c_mid-channel tensors (pre-activation, activated + masked, and its gradient) exist as
        # packed bf16 only, written by the producing layers' epilogues (ops.ffn_packed_fwd)
        keep = {}
        yp = Var(ops.ffn_packed_fwd(xr, ln.gamma(), ln.beta(), w1, l1.bias_data(), w2, l2.bias_data(), _row(res.data), p, seed, seed_out, sb, keep).view(-1, B, T))
        _ACTIVATION_ELEMS[0] += l1.cout * N  # (the pre-activation: the tensor the separate operators count as dense1's output)

        def bwd_packed():
            if yp.grad is None:
                return
            dh = ops.ffn_packed_bwd(xr, w1, w2, _row(yp.grad), p, seed, seed_out, sb, dw1, l1.db_sink(), dw2, l2.db_sink(), keep)
            keep.clear()
            res.accumulate(yp.grad)  # (first: LayerNorm's backward then adds into it -- res is x in the Conformer's blocks)
            _ln_bwd_into(x, ln, dh.view(x.data.shape))

        tape.record(bwd_packed)
        return yp
    packed1, packed2 = {}, {}
    ln_fused = ops.ln_dense_fused_supported(*_items(B, T), C, l1.cout)
    if ln_fused:
        h = None
        a = ops.layernorm_dense_fwd(xr, ln.gamma(), ln.beta(), w1, l1.bias_data(), packed1)
    else:
        h = layernorm(tape, x, ln)
        a = ops.conv1d_fwd(_row(h.data), w1, l1.bias_data(), 1, l1.pad, 1, 1, keep=packed1)
    fuse_out = res is not None and ops.resdrop_fused_supported(*_items(B, T), l1.cout, l2.cout)
    if fuse_out:
        y = Var(ops.conv1d_fwd_resdrop(a, w2, l2.bias_data(), _row(res.data), p, seed_out, sb, packed2, in_p=p, in_seed=seed).view(-1, B, T))
    else:
        y = Var(ops.conv1d_fwd_silu_dropout(a, w2, l2.bias_data(), p, seed, packed2).view(-1, B, T))
    _ACTIVATION_ELEMS[0] += a.numel()  # (the pre-activation: the tensor the separate operators count as dense1's output)

    def bwd():
        if y.grad is None:
            return
        # second layer: its packed input is the forward's (a stands in for the fp32 tensor that was never stored: only its shape is read)
        if fuse_out:
            ds = ops.conv1d_bwd_dropout_dy(a, w2, _row(y.grad), p, seed_out, sb, dw2, l2.db_sink(), packed2, x_standin=True)
        else:
            ds, _, _ = ops.conv1d_bwd(a, w2, _row(y.grad), 1, 0, 1, 1, need_dx=True, dw_out=dw2, db_out=l2.db_sink(), accumulate=True, packed=packed2,
                                       x_standin=True)
        packed2.clear()
        # first layer: x.data / h.data only lend their shape (the weight gradient reads the packed copy)
        dh = ops.conv1d_bwd_silu_dropout_dy(xr if ln_fused else _row(h.data), w1, ds, a, p, seed, dw1, l1.db_sink(), packed1).view(x.data.shape)
        packed1.clear()
        if fuse_out:
            res.accumulate(y.grad)
        if ln_fused:
            _ln_bwd_into(x, ln, dh)
        else:
            h.accumulate(dh)

    tape.record(bwd)
    return y if (res is None or fuse_out) else residual_dropout(tape, res, y, p, seed_out, sb)


def residual(tape: Tape, a: Var, b: Var, sb: float = 1.0) -> Var:
    """a + sb * b"""
    y = Var(ops.axpby(1.0, a.data, sb, b.data))

    def bwd():
        if y.grad is None:
            return
        b.accumulate(ops.elementwise(ops.EW_SCALE, y.grad, p0=sb))
        a.accumulate(y.grad)  # y.grad is not used again: handing it over is safe, b got its own tensor

    tape.record(bwd)
    return y


def attention(tape: Tape, qkv: Var, lens32, heads: int, p: float, seed: int) -> Var:
    out, saved = ops.attention_train_fwd(qkv.data, lens32, heads, p, seed)
    y = Var(out)
    tape.record(lambda: y.grad is not None and qkv.accumulate(ops.attention_train_bwd(qkv.data, saved, y.grad, heads, p, seed)))
    return y


def masked(tape: Tape, x: Var, lens32) -> Var:
    """Zero the columns t >= len[b] (a copy: earlier operators may still need the unmasked values)."""
    y = Var(ops.mask_cols_(x.data.clone(), lens32))
    tape.record(lambda: y.grad is not None and x.accumulate(ops.mask_cols_(y.grad.clone(), lens32)))
    return y


def mse_loss(tape: Tape, pred: Var, target: torch.Tensor, count_dev: torch.Tensor, weight: float) -> torch.Tensor:
    """weight * sum((pred - target)^2) / count -> device scalar; both operands are zero outside the valid region.  ``count_dev``:
    the number of valid elements as a DEVICE scalar (it changes from batch to batch; a captured step must not bake it in)."""
    diff = ops.axpby(1.0, pred.data, -1.0, target)
    out = torch.empty(1, device=diff.device, dtype=torch.float32)
    ops.scalar_reduce(1, diff, None, out, scale=weight, p=0.0)
    ops.elementwise(ops.EW_SCALE_DIV_SCALAR, out, c=count_dev, out=out, p0=1.0)
    tape.record(lambda: pred.accumulate(ops.elementwise(ops.EW_SCALE_DIV_SCALAR, diff, c=count_dev, p0=2.0 * weight)))
    return out


# ---- the model ---------------------------------------------------------------------------------------------------------
class _ConformerT:
    def __init__(self, g: ParamGroup, cfg, prefix: str):
        self.cfg = cfg
        d, f, k = cfg.input_dim, cfg.feedforward_dim, cfg.conv_kernel_size
        self.layers = []
        for i in range(cfg.layers):
            p = f"{prefix}.conformer_layers.{i}."
            L = {}
            for name in ("ffn1",):
                L[name] = self._ffn(g, p + name, d, f)
            L["attn_ln"] = Affine(g, p + "self_attn_layer_norm", d)
            L["in_proj"] = Dense(g, p + "self_attn.in_proj_weight", p + "self_attn.in_proj_bias", d, 3 * d, linear=True)
            L["out_proj"] = Dense(g, p + "self_attn.out_proj.weight", p + "self_attn.out_proj.bias", d, d, linear=True)
            c = p + "conv_module."
            L["conv_ln"] = Affine(g, c + "layer_norm", d)
            L["pw1"] = Dense(g, c + "sequential.0.weight", c + "sequential.0.bias", d, 2 * d)
            L["dw"] = Dense(g, c + "sequential.2.weight", c + "sequential.2.bias", 1, d, k)
            L["bn"] = Affine(g, c + "sequential.3", d, batchnorm=True)
            L["pw2"] = Dense(g, c + "sequential.5.weight", c + "sequential.5.bias", d, d)
            L["ffn2"] = self._ffn(g, p + "ffn2", d, f)
            L["final_ln"] = Affine(g, p + "final_layer_norm", d)
            self.layers.append(L)

    @staticmethod
    def _ffn(g, p, d, f):
        return dict(ln=Affine(g, p + ".sequential.0", d), l1=Dense(g, p + ".sequential.1.weight", p + ".sequential.1.bias", d, f, linear=True),
                    l2=Dense(g, p + ".sequential.4.weight", p + ".sequential.4.bias", f, d, linear=True))

    def batchnorms(self):
        return [L["bn"] for L in self.layers]

    def forward(self, tape: Tape, x: Var, lens32, seeds) -> Var:
        p = 0.0 if _EVAL[0] else self.cfg.dropout
        for L in self.layers:
            x = self._ffn_fwd(tape, x, L["ffn1"], p, seeds)
            h = ln_dense(tape, x, L["attn_ln"], L["in_proj"])
            h = attention(tape, h, lens32, self.cfg.heads, p, seeds(self.cfg.heads))
            x = dense_residual_dropout(tape, x, h, L["out_proj"], p, seeds())
            h = glu(tape, ln_dense(tape, x, L["conv_ln"], L["pw1"]))
            h = batchnorm(tape, dwconv(tape, h, L["dw"]), L["bn"], ops.ACT_SILU)
            x = dense_residual_dropout(tape, x, h, L["pw2"], p, seeds())
            x = self._ffn_fwd(tape, x, L["ffn2"], p, seeds)
            x = layernorm(tape, x, L["final_ln"])
        return x

    @staticmethod
    def _ffn_fwd(tape, x, F, p, seeds):
        s1, s2 = seeds(), seeds()  # (the order of the separate operators: the block's inner mask, then the residual's)
        return ffn_core(tape, x, F["ln"], F["l1"], F["l2"], p, s1, res=x, seed_out=s2, sb=0.5)


class _VariancePredictorT:
    def __init__(self, g: ParamGroup, cfg, prefix: str):
        self.cfg = cfg
        d, k = cfg.input_dim, cfg.kernel_size
        self.layers = []
        for i in range(cfg.n_layers):
            p = f"{prefix}.convs.{i}"
            if cfg.depthwise:
                conv = (WNConv(g, p + ".0", d, d, k, pad=(k - 1) // 2, groups=d), WNConv(g, p + ".1", d, d, 1))
            else:
                conv = (Dense(g, p + ".weight", p + ".bias", d, d, k),)
            self.layers.append((conv, Affine(g, f"{prefix}.norms.{i}", d)))
        self.linear = Dense(g, prefix + ".linear.weight", prefix + ".linear.bias", d, 1, linear=True)

    def convs(self):
        return [c for conv, _ in self.layers for c in conv]

    def forward(self, tape: Tape, x: Var, lens32, seeds) -> Var:
        """x [D, B, L] -> [1, B, L], zero at the padded positions."""
        for conv, ln in self.layers:
            h = dense(tape, dwconv(tape, x, conv[0]), conv[1], ops.ACT_RELU) if self.cfg.depthwise else dense(tape, x, conv[0], ops.ACT_RELU)
            x = dropout(tape, layernorm(tape, h, ln), 0.0 if _EVAL[0] else self.cfg.dropout, seeds())
        return masked(tape, dense(tape, x, self.linear), lens32)


class _AlignerT:
    """The aligner of ``learn_alignment: true`` (Badlani et al. 2021; FastPitch ``ConvAttention``): keys = the symbol embeddings
    through Conv(k=3) - ReLU - Conv(k=1), queries = the target mel through Conv(k=3) - ReLU - Conv(1) - ReLU - Conv(1), scores
    ``-temperature * ||q - k||^2`` with the beta-binomial prior (A9) added in the log domain.  PARITY UNPINNED (absent submodule)."""

    n_att, temperature = 80, 0.0005

    def __init__(self, g: ParamGroup, d_text: int, n_mels: int):
        name = lambda side, i: (f"attention.{side}_proj.{i}.conv.weight", f"attention.{side}_proj.{i}.conv.bias")
        self.key = [Dense(g, *name("key", 0), d_text, 2 * d_text, 3), Dense(g, *name("key", 2), 2 * d_text, self.n_att, 1)]
        self.query = [Dense(g, *name("query", 0), n_mels, 2 * n_mels, 3), Dense(g, *name("query", 2), 2 * n_mels, n_mels, 1),
                      Dense(g, *name("query", 4), n_mels, self.n_att, 1)]

    def join_side(self):
        """The main stream waits for the CTC side stream (what the aligner's backward does anyway; a captured stretch that ends
        before that backward must not leave the forked stream unjoined)."""
        if getattr(self, "_pending_done", None) is not None:
            torch.cuda.current_stream(self._pending_done_device).wait_event(self._pending_done)
            self._pending_done = None  # joined: the aligner's backward (possibly in the NEXT captured stretch) must not wait on an
                                       # event that belongs to this one

    def forward(self, tape: Tape, text_emb: Var, mel: Var, prior, text_lens32, mel_lens32, n_frames_dev, ctc_weight: float, bin_on: bool,
                bin_w_dev=None, bin_div_dev=None):
        """-> ``join``; ``join() -> (losses dict, hard durations [B, L] int32, hard alignment)`` once the main stream needs them.
        Records the backward of both losses into the projections / embedding."""
        from ..heavy import maximum_path
        k = dense(tape, dense(tape, text_emb, self.key[0], ops.ACT_RELU), self.key[1])
        q = dense(tape, dense(tape, dense(tape, mel, self.query[0], ops.ACT_RELU), self.query[1], ops.ACT_RELU), self.query[2])
        soft, logprob = ops.align_attention_fwd(q.data, k.data, prior, text_lens32, self.temperature)
        # The CTC forward-sum loss and its gradient (one workgroup per utterance walking the frames: ~2 ms of latency, a fraction of
        # the chip) are only needed when backward reaches the aligner: they run on a side stream beside the variance adaptor and
        # the decoder, and the aligner's backward waits for them.
        main = torch.cuda.current_stream(logprob.device)
        if getattr(self, "_side", None) is None:
            self._side = torch.cuda.Stream(logprob.device)
        fork, done = torch.cuda.Event(), torch.cuda.Event()
        fork.record(main)
        self._side.wait_event(fork)
        with torch.cuda.stream(self._side):
            ctc, dlogprob = ops.forward_sum_loss_and_grad(logprob, text_lens32, mel_lens32, ctc_weight)
            done.record(self._side)
        # The monotonic search over log(soft) (no gradient; one wave per utterance, ~0.7 ms) gives the durations the length regulator
        # and the variance targets need -- but the encoder does not: it runs on a second side stream beside the encoder's forward,
        # and `join()` (called by the trainer in front of the first use of the durations) makes the main stream wait for it.
        if getattr(self, "_side_mas", None) is None:
            self._side_mas = torch.cuda.Stream(logprob.device)
        mas_done = torch.cuda.Event()
        self._side_mas.wait_event(fork)
        with torch.cuda.stream(self._side_mas):
            hard, dur = maximum_path(ops.elementwise(16, soft), mel_lens32, text_lens32)
            dur = dur.to(torch.int32)
            mas_done.record(self._side_mas)
        if not torch.cuda.is_current_stream_capturing():
            for t in (hard, dur):  # allocated on the side stream's pool, used (and released) under the main stream
                t.record_stream(main)
        losses = {"attn_ctc": ctc}
        self._pending_done, self._pending_done_device = done, logprob.device

        def join():
            from ..heavy import binarization_loss
            torch.cuda.current_stream(logprob.device).wait_event(mas_done)
            if bin_on:  # the weight ramps with the epoch: it is a device scalar, so the launch sequence (and a captured graph) stays
                losses["attn_bin"] = ops.elementwise(ops.EW_MUL, binarization_loss(hard, soft).reshape(1), bin_w_dev)
            return losses, dur, hard

        def bwd():
            self.join_side()  # (the CTC side stream: its gradient is needed now)
            # (binarisation weight / frame count: one device scalar, frames / weight, divides a unit scale)
            dq, dk = ops.align_attention_bwd(q.data, k.data, soft, logprob, prior, hard if bin_on else None, dlogprob, text_lens32,
                                             self.temperature, 1.0 if bin_on else 0.0, bin_count=bin_div_dev if bin_on else n_frames_dev)
            q.accumulate(dq)
            k.accumulate(dk)

        tape.record(bwd)
        return join


class FastSpeech2Trainer:
    """``tr = FastSpeech2Trainer(config, stats); losses = tr.training_step(batch)``.

    batch: ``ids [B, L]`` (0 = padding), ``lens [B]``, ``durations [B, L]`` (frames per symbol), ``mel [B, T, n_mels]`` (zero padded,
    ``T = max sum(durations)``), ``pitch`` / ``energy`` ``[B, L]`` (phone level, normalised), optional ``speakers`` / ``languages`` ``[B]``.
    """

    _VERSION = "1.0"

    def __init__(self, config: FastSpeech2ModelConfig | None = None, stats: Stats | None = None, training: FastSpeech2TrainingConfig | None = None,
                 device="cuda:0", seed: int = 1234, lang2id: dict | None = None, speaker2id: dict | None = None, process_group=None,
                 precision: str = "f32", side_wgrad: bool | None = None, use_graph: bool = False, graph_buckets: tuple | None = None):
        if precision not in ("f32", "bf16"):
            raise ValueError("precision: 'f32' or 'bf16' (bf16 operands of the dense layers, fp32 accumulation / master weights)")
        self.precision = precision
        # weight / bias gradients on a sibling stream beside the input-gradient chain (ops.side_wgrad): the step is one eager stream
        # of mostly small launches, so the two fill each other's gaps
        self.side_wgrad = (os.environ.get("EVMI_FS2_SIDE_WGRAD", "1") == "1") if side_wgrad is None else bool(side_wgrad)
        self._stream = None  # set at the end of __init__ (with its sibling streams)
        self._pred_branch = os.environ.get("EVMI_FS2_PRED_STREAM", "1") == "1"  # the variance predictors beside the decoder (_step)
        self._align_branch = os.environ.get("EVMI_FS2_ALIGN_STREAM", "1") == "1"  # the aligner's backward beside the encoder's (_step)
        self._dp_branch = os.environ.get("EVMI_FS2_DP_BRANCH", "1") == "1"  # both branches under data parallelism too (stretches cut behind their joins)
        self.last_step_branch_on_stream = False
        self.use_graph = bool(use_graph)
        self.graph_buckets = tuple(int(v) for v in graph_buckets) if graph_buckets else None  # (symbols, frames) multiples to pad to
        self._graphs, self._graph_warm, self._graph_failed, self.last_step_was_graph = {}, {}, None, False
        self.config = c = config or FastSpeech2ModelConfig()
        self.stats = stats or Stats()
        self.training = training or FastSpeech2TrainingConfig()
        self.device = torch.device(device)
        self.lang2id, self.speaker2id = lang2id or {}, speaker2id or {}
        self.pg = process_group  # None: one GPU; True: the default torch.distributed group; else a group (RCCL under "nccl")
        if self.device.type != "cuda":
            raise RuntimeError("FastSpeech2Trainer runs on libevmi_hip (MI355X) only; there is no CPU path")
        _lib.load()
        g = self.params = ParamGroup(self.device)
        d = c.encoder.input_dim
        self.pfs = c.target_text_representation_level == "phonological_features"
        # symbol ids -> embedding table, or 43-dim phonological feature vectors -> bias-free Linear (everyvoice/text/features.py:7)
        self.text_table = None if self.pfs else Table(g, "text_input_layer.weight", c.n_symbols, d)
        self.text_linear = Dense(g, "text_input_layer.weight", None, N_PHONOLOGICAL_FEATURES, d, linear=True) if self.pfs else None
        self.aligner = _AlignerT(g, d, c.n_mels) if c.learn_alignment else None
        self.encoder = _ConformerT(g, c.encoder, "encoder")
        self.speaker_table = Table(g, "speaker_embedding.weight", max(1, c.n_speakers), d) if c.multispeaker else None
        self.language_table = Table(g, "language_embedding.weight", max(1, c.n_languages), d) if c.multilingual else None
        vp = c.variance_predictors
        self.duration_predictor = _VariancePredictorT(g, vp.duration, "duration_predictor")
        self.pitch_predictor = _VariancePredictorT(g, vp.pitch, "pitch_predictor")
        self.energy_predictor = _VariancePredictorT(g, vp.energy, "energy_predictor")
        self.pitch_table = Table(g, "pitch_embedding.weight", vp.pitch.n_bins, d)
        self.energy_table = Table(g, "energy_embedding.weight", vp.energy.n_bins, d)
        self.decoder = _ConformerT(g, c.decoder, "decoder")
        self.mel_linear = Dense(g, "mel_linear.weight", "mel_linear.bias", c.decoder.input_dim, c.n_mels, linear=True)
        self.postnet = []
        if c.use_postnet:
            dims = [c.n_mels] + [c.postnet_channels] * (c.postnet_layers - 1) + [c.n_mels]
            for i in range(c.postnet_layers):
                p = f"postnet.convolutions.{i}"
                self.postnet.append((Dense(g, p + ".0.weight", p + ".0.bias", dims[i], dims[i + 1], c.postnet_kernel), Affine(g, p + ".1", dims[i + 1], batchnorm=True)))
        g.finalize()
        self.inv_freq = (1.0 / (10000 ** (torch.arange(0.0, d, 2.0) / d))).to(self.device)
        lin = lambda st, n: torch.linspace(st.norm_min, st.norm_max, n - 1).to(self.device)
        self.pitch_bins, self.energy_bins = lin(self.stats.pitch, vp.pitch.n_bins), lin(self.stats.energy, vp.energy.n_bins)
        self._wn = [cv for vpred in (self.duration_predictor, self.pitch_predictor, self.energy_predictor) for cv in vpred.convs() if isinstance(cv, WNConv)]
        # effective weights, norms and gradient sinks of the 30 weight-normed layers in three flat buffers: ONE launch per step computes
        # w = g v / ||v|| for all of them and one turns the sinks into (dg, dv) -- 90 launches of ~4 us at the two ends of the step otherwise
        self._wn_batch = WNBatch(self.params, self._wn)
        self._bn = self.encoder.batchnorms() + self.decoder.batchnorms() + [bn for _, bn in self.postnet]
        self.global_step = 0
        self.current_epoch = 0  # the driver advances it; only the binarisation-loss warm-up reads it
        self._prior = None
        self._seed = seed
        self._grad_norm = torch.zeros(1, device=self.device)
        self._seed_base = torch.zeros(1, device=self.device, dtype=torch.int64)  # (seed, step, rank) << 16: see _store_step_scalars
        self._scal = torch.zeros(8, device=self.device, dtype=torch.float32)     # [learning rate, tokens, frames, mel elements]
        self._reducer, self._tail_lo = None, None
        self.init_random(seed)
        # the step's stream and its three siblings, taken from the pool back to back (four different hardware queues: training_step)
        self._stream = torch.cuda.Stream(self.device)
        with torch.cuda.stream(self._stream):
            ops._side_state(self.device)  # the weight-gradient sibling of self._stream
        if self.aligner is not None:
            self.aligner._side = torch.cuda.Stream(self.device)
            self.aligner._side_mas = torch.cuda.Stream(self.device)
        if self.device.type == "cuda":
            self._pred_stream = torch.cuda.Stream(self.device)
            self._pred_fork, self._pred_done = torch.cuda.Event(), torch.cuda.Event()
            self._align_fork, self._align_done = torch.cuda.Event(), torch.cuda.Event()

    def _tail_offset(self) -> int:
        """First element of the flat buffers that belongs to the decoder / mel_linear / postnet (declared last, in this order)."""
        if self._tail_lo is None:
            tail = ("decoder.", "mel_linear.", "postnet.")
            names = self.params.names()
            lo = min(self.params.offset_of(n) for n in names if n.startswith(tail))
            assert all(n.startswith(tail) == (self.params.offset_of(n) >= lo) for n in names), "decoder-side parameters are not a suffix of the buffer"
            self._tail_lo = lo
        return self._tail_lo

    # -- parameters ---------------------------------------------------------------------------------------------------
    def init_random(self, seed: int = 1234):
        """torch-like initial values under a fixed seed (there is no network for checkpoints)."""
        gen = torch.Generator().manual_seed(seed)
        sd = {}
        for name, shape, _ in self.params._specs:
            if name.endswith("weight_g") or (len(shape) == 1 and name.endswith(".weight")):
                sd[name] = torch.ones(shape)
            elif name.endswith("bias"):
                sd[name] = torch.zeros(shape)
            else:
                fan_in = max(1, math.prod(shape[1:])) if len(shape) > 1 else shape[0]
                sd[name] = torch.randn(shape, generator=gen) / fan_in ** 0.5
        self.load_state_dict(sd, strict=False)

    def load_state_dict(self, sd: dict, strict: bool = True):
        names = set(self.params.names())
        for name in names:
            if name in sd:
                self.params.load(name, sd[name])
            elif strict:
                raise KeyError(f"missing parameter {name}")
        for bn in self._bn:
            if bn.prefix + ".running_mean" in sd:
                bn.running_mean.copy_(sd[bn.prefix + ".running_mean"])
                bn.running_var.copy_(sd[bn.prefix + ".running_var"])
        return self  # (the effective weights are recomputed at the start of every step: WNBatch.materialize)

    def state_dict(self) -> dict:
        """Exactly what ``everyvoice_amd.fs2.FastSpeech2.load_state_dict`` (and the oracle module) take."""
        sd = self.params.state_dict()
        for bn in self._bn:
            sd[bn.prefix + ".running_mean"] = bn.running_mean.clone()
            sd[bn.prefix + ".running_var"] = bn.running_var.clone()
            sd[bn.prefix + ".num_batches_tracked"] = torch.tensor(bn.batches)
        sd["position_embedding.inv_freq"] = self.inv_freq.clone()
        sd["pitch_bins"], sd["energy_bins"] = self.pitch_bins.clone(), self.energy_bins.clone()
        return sd

    def checkpoint(self) -> dict:
        """Lightning-shaped: state_dict, optimiser state, JSON-only hyper_parameters, model_info (tests/test_model.py:85-151)."""
        g = self.params
        return {"state_dict": {k: v.cpu() for k, v in self.state_dict().items()},
                "optimizer_states": [{"m": g.m.cpu(), "v": g.v.cpu(), "step": g.step}],
                "global_step": self.global_step,
                "hyper_parameters": {"config": asdict(self.config), "stats": asdict(self.stats), "training": self.training.json_dict(),
                                     "lang2id": dict(self.lang2id), "speaker2id": dict(self.speaker2id)},
                "model_info": {"name": "FastSpeech2", "version": self._VERSION}}

    def load_checkpoint(self, ckpt: dict):
        info = ckpt.get("model_info", {})
        if info.get("name") != "FastSpeech2":
            raise TypeError(f"Wrong model type ({info.get('name')}), we are expecting a 'FastSpeech2' model")
        self.load_state_dict(ckpt["state_dict"])
        for bn in self._bn:
            bn.batches = int(ckpt["state_dict"].get(bn.prefix + ".num_batches_tracked", 0))
        o = ckpt["optimizer_states"][0]
        self.params.m.copy_(o["m"])
        self.params.v.copy_(o["v"])
        self.params.step = int(o["step"])
        self.global_step = int(ckpt["global_step"])
        return self

    # -- schedule -----------------------------------------------------------------------------------------------------
    def learning_rate(self, step: int) -> float:
        """Noam: linear warm-up to ``learning_rate`` at ``warmup_steps``, then inverse square root decay."""
        o = self.training.optimizer
        w = max(1, o.warmup_steps)
        return o.learning_rate * w ** 0.5 * min(step ** -0.5, step * w ** -1.5)

    # -- the step -------------------------------------------------------------------------------------------------------
    def _world_rank(self):
        if self.pg is None:
            return 1, 0
        import torch.distributed as dist

        grp = self.pg if self.pg is not True else None
        return dist.get_world_size(grp), dist.get_rank(grp)

    def _prepare(self, batch: dict):
        """Host batch -> (device batch in the step's dtypes / layouts, host-side facts about it).  Everything that needs the host
        (lengths, counts, the padded shape) is decided HERE, before the step: the step itself then is a fixed launch sequence for a
        given padded shape, which is what lets it be captured into a HIP graph and replayed on other batches of that shape."""
        dev = self.device
        lens_host = batch["lens"].to("cpu", torch.int64)
        learn = self.aligner is not None
        L = int(batch["pfs"].shape[1] if self.pfs else batch["ids"].shape[1])
        B = int(lens_host.shape[0])
        if learn:  # durations come out of the aligner; the frame counts are the data's
            mel_lens_host = batch["mel_lens"].to("cpu", torch.int64)
        else:
            pad_host = torch.arange(L)[None, :] >= lens_host[:, None]
            dur_host = batch["durations"].to("cpu").clamp_min(0).masked_fill(pad_host, 0)
            mel_lens_host = dur_host.sum(1)
        T = int(mel_lens_host.max())
        Lp, Tp = L, T
        if self.graph_buckets is not None:  # padded shapes from a small set: captured steps get reused (see training_step)
            Lp = -(-L // self.graph_buckets[0]) * self.graph_buckets[0]
            Tp = -(-T // self.graph_buckets[1]) * self.graph_buckets[1]

        def fit(t: torch.Tensor, sizes: dict) -> torch.Tensor:
            """Slice / zero-pad the given axes of a batch tensor to the step's padded shape."""
            for axis, size in sizes.items():
                if t.shape[axis] > size:
                    t = t.narrow(axis, 0, size)
                elif t.shape[axis] < size:
                    padding = [0, 0] * t.dim()
                    padding[2 * (t.dim() - 1 - axis) + 1] = size - t.shape[axis]
                    t = torch.nn.functional.pad(t, padding)
            return t

        def up(t: torch.Tensor, dtype) -> torch.Tensor:
            """To the device in the step's dtype.  Host tensors go through pinned memory without blocking: a pageable copy makes
            the host wait for everything queued on the stream in front of it."""
            if t.is_cuda or dev.type != "cuda":
                return t.to(dev, dtype)
            return t.to(dtype).pin_memory().to(dev, non_blocking=True)

        d = {"lens": up(batch["lens"], torch.int32).contiguous(), "mel_lens": up(mel_lens_host, torch.int32).contiguous()}
        if self.pfs:  # batch["pfs"] [B, L, 43] multi-hot feature vectors (the reference's `pfs` files) -> [43, B, L]
            d["pfs"] = fit(up(batch["pfs"], torch.float32), {1: Lp}).permute(2, 0, 1).contiguous()
        else:
            d["ids"] = fit(up(batch["ids"], torch.int32), {1: Lp}).contiguous()
        if not learn:
            d["durations"] = fit(up(dur_host, torch.int32), {1: Lp}).contiguous()
        d["mel_t"] = fit(up(batch["mel"], torch.float32), {1: Tp}).permute(2, 0, 1).contiguous()  # [n_mels, B, T]
        if learn and batch.get("attn_prior") is not None:
            d["attn_prior"] = fit(up(batch["attn_prior"], torch.float64), {1: Tp, 2: Lp}).contiguous()
        for key in ("pitch", "energy"):
            if key in batch:
                d[key] = fit(up(batch[key], torch.float32), {1: Lp}).contiguous()
            else:
                d[key + "_frames"] = fit(up(batch[key + "_frames"], torch.float32), {1: Tp}).contiguous()
        for table, key in ((self.speaker_table, "speakers"), (self.language_table, "languages")):
            if table is not None:
                if batch.get(key) is None:
                    raise ValueError(f"this model needs `{key}` ids [B]")
                d[key] = up(batch[key], torch.int32).contiguous()
        n_frames = float(mel_lens_host.sum())
        meta = dict(B=B, L=Lp, T=Tp, n_tok=float(lens_host.sum()), n_frames=n_frames, n_el=n_frames * self.config.n_mels)
        return d, meta

    def _prepare_on_upload_stream(self, batch: dict):
        """`_prepare` on a stream of its own, uploads through pinned memory.  A pageable host -> device copy on the step's stream made
        the host wait for the WHOLE previous step, after which the device idled while the host padded, permuted and converted the
        next batch (0.7 ms of a 24 ms step in the trace: `tools/trace_gaps.py`).  Now the host never waits, and a host batch (a data
        loader's) is uploaded and laid out UNDER step n while the step's stream only waits for the upload stream's event.  A batch
        that already holds device tensors may have been produced by work still queued on the caller's stream: the upload stream
        then queues behind it (a device-side dependency; the host still runs ahead) -- unless the caller sets `batch_ready = True`
        (a prefetcher that synchronised its own copies, or a batch resident on the device from the start, as in bench.py)."""
        if self.device.type != "cuda":
            return self._prepare(batch)
        if getattr(self, "_upload", None) is None:
            self._upload = torch.cuda.Stream(self.device)
        step = torch.cuda.current_stream(self.device)
        if any(torch.is_tensor(v) and v.is_cuda for v in batch.values()) and not getattr(self, "batch_ready", False):
            self._upload.wait_stream(step)  # (`batch_ready = True`: the caller vouches that the batch's device tensors are complete)
        with torch.cuda.stream(self._upload):
            d, meta = self._prepare(batch)
        step.wait_stream(self._upload)
        for v in d.values():  # allocated under the upload stream, read by the step's
            if v.is_cuda:
                v.record_stream(step)
        return d, meta

    def _store_step_scalars(self, meta: dict) -> None:
        """What changes from step to step without changing the launch sequence lives on the device: the dropout seed base
        (seed, step, rank) and [learning rate of this step, token count, frame count, mel element count]."""
        world, rank = self._world_rank()
        base = ((((self._seed * 1000003 + self.global_step) * world + rank) << 16)) & 0x7FFFFFFFFFFFFFFF
        ops.store_u64(self._seed_base, base)
        bin_w = self._bin_weight()
        ops.store_f32(self._scal, [self.learning_rate(self.global_step + 1), meta["n_tok"], meta["n_frames"], meta["n_el"], bin_w,
                                   meta["n_frames"] / bin_w if bin_w > 0.0 else 1.0])

    def _bin_weight(self) -> float:
        """Weight of the binarisation loss at the current epoch (linear warm-up over ``attn_bin_loss_warmup_epochs``; 0: the term is off)."""
        if self.aligner is None:
            return 0.0
        tr = self.training
        return tr.attn_bin_loss_weight * min(self.current_epoch / max(1, tr.attn_bin_loss_warmup_epochs), 1.0)

    def forward_backward(self, batch: dict) -> dict:
        """Forward in training mode + every loss + backward; gradients are left in ``self.params.grad``."""
        d, meta = self._prepare(batch)
        self._store_step_scalars(meta)
        prev = ops.SEED_BASE[0]
        ops.SEED_BASE[0] = self._seed_base
        try:
            return self._forward_backward(d, meta)
        finally:
            ops.SEED_BASE[0] = prev

    def _forward_backward(self, batch: dict, meta: dict, segmented: bool = False):
        """The launch sequence of one step on a prepared (device) batch: no host read, no host-dependent argument.
        ``segmented``: -> (backward generator, losses) instead of running the backward: every ``next()`` runs it up to the next
        gradient-bucket boundary (data-parallel graph capture: the all-reduces sit between the captured stretches), and
        ``_finish_backward(losses)`` completes the step's gradient side."""
        lib, dev, c, tr = _lib.load(), self.device, self.config, self.training
        lens, mel_lens, mel_t = batch["lens"], batch["mel_lens"], batch["mel_t"]
        feats, ids = batch.get("pfs"), batch.get("ids")
        B, L, T = meta["B"], meta["L"], meta["T"]
        D = c.encoder.input_dim
        n_tok, n_frames, n_el = self._scal[1:2], self._scal[2:3], self._scal[3:4]  # device scalars (see _store_step_scalars)
        pad = torch.arange(L, device=dev)[None, :] >= lens[:, None]
        learn = self.aligner is not None

        if not _EVAL[0]:
            self.params.zero_grad()
        self._wn_batch.materialize()
        tape = Tape()
        counter = [0]

        def seeds(n=1):
            """Counter-based dropout seeds: the draw's index inside the step; the kernels add the device-resident base
            (seed, step, rank) << 16 -- distinct per (seed, step, rank, draw): every data-parallel rank masks its shard
            independently, as per-process RNG streams do under DDP."""
            counter[0] += n
            assert counter[0] < 65536, "more dropout draws in one step than the seed layout reserves"
            return counter[0] - n

        def embed(with_position: bool) -> Var:
            if self.pfs:  # Linear(43 -> D) over the feature vectors, padded columns zeroed (+ the positional sinusoid)
                v = dense(tape, Var(feats, needs_grad=False), self.text_linear)
                if not with_position:
                    return masked(tape, v, lens)
                e = v.data.clone()
                _chk(lib.evmi_fs2_add_posemb_f32(e.data_ptr(), lens.data_ptr(), self.inv_freq.data_ptr(), B, L, D, _s(e)), "evmi_fs2_add_posemb_f32")
                out = Var(e)
                tape.record(lambda: out.grad is not None and v.accumulate(ops.mask_cols_(out.grad.clone(), lens)))
                return out
            e = torch.empty(D, B, L, device=dev, dtype=torch.float32)
            _chk(lib.evmi_fs2_embed_f32(ids.data_ptr(), lens.data_ptr(), self.text_table.data().data_ptr(), self.inv_freq.data_ptr() if with_position else 0,
                                        e.data_ptr(), B, L, D, _s(e)), "evmi_fs2_embed_f32")
            v = Var(e)
            tape.record(lambda: v.grad is not None and _chk(lib.evmi_fs2_embed_bwd_f32(
                v.grad.data_ptr(), ids.data_ptr(), lens.data_ptr(), self.text_table.grad().data_ptr(), self.text_table.rows, B, L, D, 0, _s(e)), "evmi_fs2_embed_bwd_f32"))
            return v

        losses = {}
        # The variance predictors are side branches under teacher forcing (see below): they get a tape of their own and ALIASES of their
        # inputs, in every training schedule (one order of additions into the encoder output's gradient whatever stream runs them).
        struct = not _EVAL[0]
        # (data parallel: the captured step is cut into stretches around the gradient exchanges and a forked stream has to come back
        # inside one stretch -- so the stretch boundary sits BEHIND the branch's join (below: `dp_cut`), not across the branch, and the
        # step a data-parallel run times is the step one GPU times.  EVMI_FS2_DP_BRANCH=0: the predictors on the chain there, as in round 5.)
        data_parallel = segmented or self._reducer is not None
        on_stream = struct and self._pred_branch and dev.type == "cuda" and (self._dp_branch or not data_parallel)
        self.last_step_branch_on_stream = bool(on_stream)
        pjoin = {"done": None, "tape": Tape()} if struct else None

        def join_branch(var, alias):
            """(a tape operator) the main chain waits for the branch and takes the gradient it left on its alias of `var`"""
            if pjoin["done"] is not None:
                torch.cuda.current_stream(dev).wait_event(pjoin["done"])
                pjoin["done"] = None
            if alias.grad is not None:
                var.accumulate(alias.grad)

        # The aligner's backward depends on its own losses only (attention, five k = 3 convolutions over the mel frames, fp32 products:
        # ~1 ms at the END of the chain).  On the predictors' stream from the start of backward it gave their 1.2 ms back (17.7 vs 16.5 ms:
        # its kernels are large enough to take CUs from the decoder's backward); started where the ENCODER's backward starts -- 4.5 k
        # columns, latency-bound launches that leave most of the chip idle -- it runs beside that.  Its own tape, an alias of the text
        # embedding whose gradient joins the chain in front of the embedding's backward: the same additions in every schedule.
        astruct = struct and learn

        def start_aligner_backward():
            if not astruct or pjoin.get("astarted"):
                return
            pjoin["astarted"] = True
            if on_stream and self._align_branch:
                main = torch.cuda.current_stream(dev)
                self._align_fork.record(main)
                self._pred_stream.wait_event(self._align_fork)
                with torch.cuda.stream(self._pred_stream):
                    side_on, ops.SIDE_WGRAD["on"] = ops.SIDE_WGRAD["on"], False
                    pjoin["akeep"] = _held_tensors(pjoin["atape"]._ops)  # (outlive the KERNELS: the aligner's forward ran on the main stream)
                    try:
                        pjoin["atape"].backward()
                        self._align_done.record(self._pred_stream)
                    finally:
                        ops.SIDE_WGRAD["on"] = side_on
                pjoin["adone"] = self._align_done
            else:
                pjoin["atape"].backward()

        def join_aligner(var, alias):
            start_aligner_backward()  # (if the encoder output had no gradient to trigger it)
            if pjoin.get("adone") is not None:
                torch.cuda.current_stream(dev).wait_event(pjoin["adone"])
                pjoin["adone"] = None
            if alias.grad is not None:
                var.accumulate(alias.grad)

        def dp_cut():
            """Data parallel: where the tail bucket (decoder / mel_linear / postnet: ~half of the parameters, final once backward has left
            the decoder) goes to its all-reduce, which runs on a side stream under the rest of the backward.  Recorded in front of the
            decoder's forward = reached after its backward.  With the predictors on their stream the point sits behind their join
            (the bucket leaves a length-regulator backward and two embedding backwards later; every fork is closed inside a stretch)."""
            if segmented:
                # (captured step: the stretch ends here; every stream forked so far must be back on the main one)
                tape.cut(self.aligner.join_side if learn else None)
            else:
                lo_tail, red = self._tail_offset(), self._reducer
                tape.record(lambda: (ops.wgrad_join(dev), red.launch(lo_tail, self.params.grad.numel())))

        if learn:
            te = embed(False)
            if astruct:
                pjoin["atape"] = Tape()
                ta = Var(te.data)
                _ACTIVATION_ELEMS[0] -= te.data.numel()  # (an alias, not another activation)
                tape.record(lambda: join_aligner(te, ta))
            align_join = self.aligner.forward(pjoin["atape"] if astruct else tape, ta if astruct else te, Var(mel_t, needs_grad=False), batch.get("attn_prior"),
                                              lens, mel_lens, n_frames, tr.attn_ctc_loss_weight, self._bin_weight() > 0.0, self._scal[4:5], self._scal[5:6])
        else:
            dur = batch["durations"]

        x0 = embed(True)
        x = self.encoder.forward(tape, x0, lens, seeds)

        if learn:  # the alignment search ran beside the encoder
            align_losses, dur, self.last_alignment = align_join()
            losses.update(align_losses)
        cum = torch.cumsum(dur, 1, dtype=torch.int32).contiguous()
        log_d_t = torch.log(dur.float() + 1.0).contiguous()
        pitch_t = self._phone_level(batch, "pitch", cum, dur, pad, T)
        energy_t = self._phone_level(batch, "energy", cum, dur, pad, T)

        for table, key in ((self.speaker_table, "speakers"), (self.language_table, "languages")):
            if table is not None:
                x = self._add_item_embedding(tape, x, table, batch[key], lens)

        w = tr.duration_loss_weight
        # The three variance predictors are side branches under teacher forcing (the decoder takes the TARGETS' embeddings): ~170 launches
        # of ~8 us at 4.5 k columns, forward and backward.  They run as ONE chain on a stream of their own beside the length regulator, the
        # decoder and its backward, on aliases of the encoder output whose gradients join the main chain where that output's backward
        # starts: 17.7 -> 16.5 ms per step.  (Three chains on three streams, each forked per predictor, were measured in round 3 and lost
        # 2.6 ms: ~600 small launches then, every one a cross-stream edge of the captured graph.)
        if struct:
            xe, xa = x, Var(x.data)  # the predictors' view of the encoder output: same data, its own gradient
            _ACTIVATION_ELEMS[0] -= x.data.numel()  # (an alias, not another activation)
            # (runs in backward when everything behind x has contributed: in front of the encoder's backward -- where the aligner's starts)
            tape.record(start_aligner_backward)
            if data_parallel and on_stream:
                dp_cut()  # backward: ... join_branch(xe, xa) | exchange of the tail bucket | aligner beside the encoder ...
            tape.record(lambda: join_branch(xe, xa))
            x1 = self._add_bucket_embedding(tape, x, pitch_t, self.pitch_bins, self.pitch_table)
            x1a = Var(x1.data)
            _ACTIVATION_ELEMS[0] -= x1.data.numel()
            tape.record(lambda: join_branch(x1, x1a))
            tp = pjoin["tape"]

            def predictors():
                losses["duration"] = mse_loss(tp, self.duration_predictor.forward(tp, xa, lens, seeds), log_d_t.view(1, B, L), n_tok, w)
                losses["pitch"] = mse_loss(tp, self.pitch_predictor.forward(tp, xa, lens, seeds), pitch_t.view(1, B, L), n_tok, tr.pitch_loss_weight)
                losses["energy"] = mse_loss(tp, self.energy_predictor.forward(tp, x1a, lens, seeds), energy_t.view(1, B, L), n_tok, tr.energy_loss_weight)

            if on_stream:
                main = torch.cuda.current_stream(dev)
                self._pred_fork.record(main)
                self._pred_stream.wait_event(self._pred_fork)
                with torch.cuda.stream(self._pred_stream):
                    predictors()
            else:
                predictors()
            x = self._add_bucket_embedding(tape, x1, energy_t, self.energy_bins, self.energy_table)
        else:
            losses["duration"] = mse_loss(tape, self.duration_predictor.forward(tape, x, lens, seeds), log_d_t.view(1, B, L), n_tok, w)
            losses["pitch"] = mse_loss(tape, self.pitch_predictor.forward(tape, x, lens, seeds), pitch_t.view(1, B, L), n_tok, tr.pitch_loss_weight)
            x = self._add_bucket_embedding(tape, x, pitch_t, self.pitch_bins, self.pitch_table)
            losses["energy"] = mse_loss(tape, self.energy_predictor.forward(tape, x, lens, seeds), energy_t.view(1, B, L), n_tok, tr.energy_loss_weight)
            x = self._add_bucket_embedding(tape, x, energy_t, self.energy_bins, self.energy_table)

        frames = torch.empty(D, B, T, device=dev, dtype=torch.float32)
        _chk(lib.evmi_length_regulate_cbt_f32(x.data.data_ptr(), cum.data_ptr(), frames.data_ptr(), D, B, L, T, _s(frames)), "evmi_length_regulate_cbt_f32")
        _chk(lib.evmi_fs2_add_posemb_f32(frames.data_ptr(), mel_lens.data_ptr(), self.inv_freq.data_ptr(), B, T, D, _s(frames)), "evmi_fs2_add_posemb_f32")
        f = Var(frames)
        x_enc = x

        def lr_bwd():
            if f.grad is None:
                return
            dfr = ops.mask_cols_(f.grad, mel_lens)  # the positional sinusoid is constant; padded frames were zeroed
            dx = torch.empty(D, B, L, device=dev, dtype=torch.float32)
            _chk(lib.evmi_length_regulate_bwd_cbt_f32(dfr.data_ptr(), cum.data_ptr(), dx.data_ptr(), D, B, L, T, _s(dx)), "evmi_length_regulate_bwd_cbt_f32")
            x_enc.accumulate(dx)

        tape.record(lr_bwd)
        if data_parallel and not on_stream:
            dp_cut()
        y = self.decoder.forward(tape, f, mel_lens, seeds)
        mel = masked(tape, dense(tape, y, self.mel_linear), mel_lens)
        losses["mel"] = mse_loss(tape, mel, mel_t, n_el, tr.mel_loss_weight)
        if self.postnet:
            h = mel
            for i, (conv, bn) in enumerate(self.postnet):
                h = batchnorm(tape, dense(tape, h, conv), bn, ops.ACT_TANH if i < len(self.postnet) - 1 else ops.ACT_NONE)
            post = masked(tape, residual(tape, mel, h), mel_lens)
            losses["postnet"] = mse_loss(tape, post, mel_t, n_el, tr.postnet_loss_weight)
        if _EVAL[0]:
            ops.wgrad_join(dev)
            return self._finish_backward(losses, grads=False)
        def branch_backward():
            """The predictors' backward starts with the step's: on their stream, behind their forward, beside the decoder's backward.
            -> what the branch's operators hold (it must outlive the branch's KERNELS, not just their launches: kept until the chain has joined)"""
            main = torch.cuda.current_stream(dev)
            self._pred_fork.record(main)
            self._pred_stream.wait_event(self._pred_fork)
            with torch.cuda.stream(self._pred_stream):
                side_on, ops.SIDE_WGRAD["on"] = ops.SIDE_WGRAD["on"], False  # (their weight gradients stay on this stream: it is a side chain already)
                keep = _held_tensors(pjoin["tape"]._ops)
                try:
                    pjoin["tape"].backward()
                    self._pred_done.record(self._pred_stream)
                finally:
                    ops.SIDE_WGRAD["on"] = side_on
            pjoin["done"] = self._pred_done
            if not torch.cuda.is_current_stream_capturing():
                for t in [v for v in losses.values() if torch.is_tensor(v)]:
                    t.record_stream(main)
            return keep

        def branch_close(keep):
            main = torch.cuda.current_stream(dev)
            for which in ("done", "adone"):  # (a join that did not run: nothing needed that gradient)
                if pjoin.get(which) is not None:
                    main.wait_event(pjoin[which])
                    pjoin[which] = None
            del keep
            pjoin.pop("akeep", None)

        if segmented:
            def segments():
                if on_stream:
                    keep = branch_backward()
                    for tag in tape.backward_segments():
                        # (a stretch ends here: the cut sits behind the predictors' join, so their stream is back; the aligner's starts behind it)
                        yield tag
                    branch_close(keep)
                    return
                pjoin["tape"].backward()  # (the predictors' backward: in front of the chain's first stretch, on its stream)
                yield from tape.backward_segments()

            return segments(), losses
        if on_stream:
            keep = branch_backward()
            tape.backward()
            branch_close(keep)
            return self._finish_backward(losses)
        if struct:
            pjoin["tape"].backward()
        tape.backward()
        return self._finish_backward(losses)

    def _finish_backward(self, losses: dict, grads: bool = True) -> dict:
        if grads:
            self._wn_batch.finish(self._wn)
        total = torch.zeros(1, device=self.device)
        for v in losses.values():
            ops.axpby(1.0, total, 1.0, v, out=total)
        losses["total"] = total
        return losses

    def evaluate(self, batch: dict) -> dict:
        """The losses of a batch in evaluation mode (the reference validates under ``model.eval()``): dropout off, BatchNorm
        normalising with its running statistics and leaving them (and ``num_batches_tracked``) untouched, no backward, no
        gradients written, no optimiser state touched."""
        prev = ops.CONV_BACKEND["operands"]
        ops.CONV_BACKEND["operands"] = self.precision
        _EVAL[0] = True
        try:
            return self.forward_backward(batch)
        finally:
            _EVAL[0] = False
            ops.CONV_BACKEND["operands"] = prev

    def _phone_level(self, batch, key, cum, dur, pad, T):
        """Phone-level variance targets [B, L]: given as such (``pitch``), or frame-level (``pitch_frames`` [B, T]) averaged over
        each symbol's frames -- ``average_data_by_durations``, everyvoice/preprocessor/preprocessor.py:287-300 (1e-7 for zero frames)."""
        dev = self.device
        if key in batch:
            return batch[key].masked_fill(pad, 0.0).contiguous()
        fr = batch[key + "_frames"]
        B, L = dur.shape
        sums = torch.empty(B, L, device=dev, dtype=torch.float32)
        _chk(_lib.load().evmi_length_regulate_bwd_cbt_f32(fr.data_ptr(), cum.data_ptr(), sums.data_ptr(), 1, B, L, T, _s(fr)), "evmi_length_regulate_bwd_cbt_f32")
        return torch.where(dur > 0, sums / dur.clamp_min(1), torch.full_like(sums, 1e-7)).masked_fill(pad, 0.0).contiguous()

    def _add_item_embedding(self, tape, x: Var, table: Table, item_ids, lens):
        lib = _lib.load()
        D, B, L = x.data.shape
        out = x.data.clone()
        _chk(lib.evmi_fs2_add_item_embedding_f32(out.data_ptr(), item_ids.data_ptr(), lens.data_ptr(), table.data().data_ptr(), B, L, D, _s(out)),
             "evmi_fs2_add_item_embedding_f32")
        y = Var(out)

        def bwd():
            if y.grad is None:
                return
            _chk(lib.evmi_fs2_item_embedding_bwd_f32(y.grad.data_ptr(), item_ids.data_ptr(), lens.data_ptr(), table.grad().data_ptr(), table.rows, B, L, D, _s(out)),
                 "evmi_fs2_item_embedding_bwd_f32")
            x.accumulate(y.grad)

        tape.record(bwd)
        return y

    def _add_bucket_embedding(self, tape, x: Var, values, bins, table: Table):
        lib = _lib.load()
        D, B, L = x.data.shape
        out = x.data.clone()
        _chk(lib.evmi_fs2_bucket_embed_add_f32(out.data_ptr(), values.data_ptr(), bins.data_ptr(), table.data().data_ptr(), table.rows, B, L, D, 1.0, _s(out)),
             "evmi_fs2_bucket_embed_add_f32")
        y = Var(out)
        tape.record(lambda: y.grad is not None and self._bucket_embedding_backward(y, x, values, bins, table))
        return y

    def _bucket_embedding_backward(self, y: Var, x: Var, values, bins, table: Table):
        lib = _lib.load()
        D, B, L = y.data.shape
        idx = torch.empty(B, L, device=y.data.device, dtype=torch.int32)
        _chk(lib.evmi_fs2_bucket_embed_bwd_f32(y.grad.data_ptr(), values.data_ptr(), bins.data_ptr(), table.grad().data_ptr(), idx.data_ptr(), table.rows,
                                               B, L, D, 1.0, _s(y.data)), "evmi_fs2_bucket_embed_bwd_f32")
        x.accumulate(y.grad)

    def training_step(self, batch: dict) -> dict:
        """One optimiser step; returns the losses as device scalars (no host synchronisation inside the step).

        The step runs on the trainer's own stream (the caller's stream is joined on both sides): that stream and the three
        sibling streams beside it (weight gradients, CTC loss, alignment search) were taken from the stream pool back to back, so
        they sit on four different hardware queues whatever the process created before -- on the caller's stream the overlap
        depended on where the pool's round-robin stood (a GAN trainer earlier in the process cost this step 4.7 ms).

        ``use_graph``: the step is ~1,650 launches whose host cost (Python + ctypes, ~10 us each) is twice its device time.  For a
        padded batch shape (B, L, T) that has been seen GRAPH_WARMUP_STEPS times the launch sequence is captured into a HIP graph
        (``torch.cuda.CUDAGraph`` on the step's stream, the sibling streams forked from it) and every later batch of that shape is
        one replay: inputs are copied into the graph's static buffers, and what changes from step to step without changing the
        sequence -- dropout seed base, learning rate, token / frame counts -- is stored on the device first
        (``_store_step_scalars``).  Shapes seen once or twice run eagerly; ``graph_buckets=(l, t)`` pads L and T up to multiples,
        so a real data stream lands on a small set of shapes (the extra padded columns enter the Conformer's BatchNorm statistics,
        as the reference's own padding does).  Eager and replayed steps are bit for bit the same arithmetic."""
        if self._stream is None:
            return self._training_step(batch)
        caller = torch.cuda.current_stream(self.device)
        self._stream.wait_stream(caller)
        with torch.cuda.stream(self._stream):
            losses = self._training_step(batch)
        caller.wait_stream(self._stream)
        for v in losses.values():
            v.record_stream(caller)
        return losses

    GRAPH_WARMUP_STEPS = 2
    GRAPH_CACHE = 24  # captured shapes kept (least recently used out first)

    def _training_step(self, batch: dict) -> dict:
        prev, prev_side, prev_base, prev_ln = ops.CONV_BACKEND["operands"], ops.SIDE_WGRAD["on"], ops.SEED_BASE[0], ops.LN_DEFER["on"]
        ops.CONV_BACKEND["operands"] = self.precision
        ops.SIDE_WGRAD["on"] = self.side_wgrad and self.device.type == "cuda"
        ops.LN_DEFER["on"] = self.device.type == "cuda"  # (reduced where the backward chains end: ops.wgrad_join)
        ops.SEED_BASE[0] = self._seed_base
        ops.side_reset()  # nothing an aborted step left collected reaches this one (ops.side_reset)
        try:
            d, meta = self._prepare_on_upload_stream(batch)
            self._store_step_scalars(meta)
            self.last_step_was_graph = False
            entry = None
            if self.use_graph and self._graph_failed is None:
                entry = self._graph_entry(d, meta)
            if entry is None:
                losses = self._step_body(d, meta)
            else:
                # into the captured step's static inputs (same shapes and types by construction of the key): one multi-tensor launch
                keys = list(d)
                torch._foreach_copy_([entry["inputs"][k] for k in keys], [d[k] for k in keys])
                self._replay(entry)
                # copies: the graph's own loss tensors are overwritten by the next replay of this shape, and a caller may keep
                # losses across steps (running means, deferred logging) -- eager steps hand out fresh tensors too
                losses = {k: v.clone() for k, v in entry["losses"].items()}
                self.last_step_was_graph = True
        except BaseException:
            ops.side_reset(abort=True)
            raise
        finally:
            ops.CONV_BACKEND["operands"] = prev
            ops.SIDE_WGRAD["on"] = prev_side
            ops.SEED_BASE[0] = prev_base
            ops.LN_DEFER["on"] = prev_ln
        ops.side_check_drained()
        self.global_step += 1
        return losses

    def _step_body(self, d: dict, meta: dict) -> dict:
        """Forward + backward + gradient exchange + clipping + optimiser on a prepared batch (eager; also what gets captured)."""
        from .hifigan import BucketReducer

        # data parallel: utterances are sharded across ranks, gradients averaged by a bucketed all-reduce that
        # overlaps backward (two buckets: see _forward_backward)
        self._reducer = (BucketReducer(self.params.grad, self.pg if self.pg is not True else None,
                                       lambda t, sc: ops.elementwise(ops.EW_SCALE, t, out=t, p0=sc)) if self.pg is not None else None)
        losses = self._forward_backward(d, meta)
        if self._reducer is not None:  # the head of the buffer (everything in front of the decoder), then wait + 1/world scaling
            self._reducer.launch(0, self._tail_offset())
            self._reducer.finish()
            self._reducer = None
        self._clip_and_update()
        return losses

    def _clip_and_update(self):
        g = self.params
        clip = self.training.gradient_clip_val
        if clip is not None:
            # g *= min(1, clip / (||g|| + 1e-6)), the norm staying on the device (torch.nn.utils.clip_grad_norm_)
            ops.scalar_reduce(1, g.grad, None, self._grad_norm, p=0.0)
            ops.elementwise(ops.EW_CLIP_SCALE, g.grad, None, self._grad_norm, out=g.grad, p0=float(clip))
        o = self.training.optimizer
        g.adamw(0.0, tuple(o.betas), o.eps, o.weight_decay, lr_dev=self._scal[0:1])  # the rate of this step: _store_step_scalars

    # -- HIP-graph execution -----------------------------------------------------------------------------------------------------
    def _graph_key(self, d: dict, meta: dict):
        tr = self.training
        # (the binarisation weight itself is a device scalar: only whether the term exists shapes the launch sequence)
        return (meta["B"], meta["L"], meta["T"], tuple(sorted(d)), self.precision, self._bin_weight() > 0.0, tr.gradient_clip_val, self.pg is not None)

    def _graph_entry(self, d: dict, meta: dict):
        """The captured step for this batch's padded shape, or None (not seen often enough yet, or capturing failed: eager)."""
        key = self._graph_key(d, meta)
        entry = self._graphs.get(key)
        if entry is not None:
            self._graphs[key] = self._graphs.pop(key)  # most recently used last
            return entry
        n = self._graph_warm.get(key, 0)
        if n < self.GRAPH_WARMUP_STEPS:  # eager steps grow the per-stream workspaces and set the kernels' launch attributes
            self._graph_warm[key] = n + 1
            if len(self._graph_warm) > 4096:
                self._graph_warm.clear()
            return None
        try:
            entry = self._capture(d, meta)
        except Exception as e:  # noqa: BLE001 -- whatever the runtime objected to: the eager path is always available
            self._graph_failed = f"{type(e).__name__}: {e}"
            torch.cuda.synchronize(self.device)
            ops.side_reset()  # the aborted capture's collected weight-gradient launches and its events must not reach the eager step
            return None
        self._graphs[key] = entry
        while len(self._graphs) > self.GRAPH_CACHE:
            self._graphs.pop(next(iter(self._graphs)))
        return entry

    def _capture(self, d: dict, meta: dict) -> dict:
        """Record the step on static copies of the inputs.  Capturing executes nothing: host-side counters the step's code bumps
        (optimiser step, BatchNorm batch counts) are put back -- `_replay` advances them."""
        inputs = {k: v.clone() for k, v in d.items()}
        step0, batches0 = self.params._step, [bn.batches for bn in self._bn]
        torch.cuda.synchronize(self.device)
        graphs, holder = [], {}
        pool = torch.cuda.graph_pool_handle()

        def cap(fn):
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, pool=pool, stream=self._stream, capture_error_mode="thread_local"):
                fn()
            graphs.append(g)

        try:
            if self.pg is None:
                cap(lambda: holder.__setitem__("losses", self._step_body(inputs, meta)))
            else:
                self._capture_data_parallel(cap, inputs, meta, holder)
        finally:
            self.params._step = step0
            for bn, b in zip(self._bn, batches0):
                bn.batches = b
            self._reducer = None
        return dict(graphs=graphs, inputs=inputs, losses=holder["losses"], cuts=holder.get("cuts", []))

    def _replay(self, entry: dict) -> None:
        graphs, cuts = entry["graphs"], entry["cuts"]
        for i, g in enumerate(graphs):
            g.replay()
            if i < len(cuts):
                cuts[i]()  # the gradient exchange that sits between two captured stretches (RCCL calls are not captured)
        self.params._step += 1
        for bn in self._bn:
            bn.batches += 1

    def _capture_data_parallel(self, cap, inputs, meta, holder):
        """Under data parallelism the step is three captured stretches with the two bucket all-reduces issued between them, the
        first one on a side stream so that it runs UNDER the second stretch (the backward of the variance adaptor, the aligner
        and the encoder): forward + backward down to the decoder | rest of the backward | clipping + optimiser."""
        from .hifigan import BucketReducer

        red = BucketReducer(self.params.grad, self.pg if self.pg is not True else None, lambda t, sc: ops.elementwise(ops.EW_SCALE, t, out=t, p0=sc))
        lo_tail, n_all = self._tail_offset(), self.params.grad.numel()
        state = {}

        def part_a():
            state["segments"], holder["losses"] = self._forward_backward(inputs, meta, segmented=True)  # cut where the tail bucket is final
            next(state["segments"])

        def part_b():
            for _ in state["segments"]:
                pass
            self._finish_backward(holder["losses"])

        cap(part_a)
        cap(part_b)
        cap(self._clip_and_update)
        holder["cuts"] = [lambda: red.launch(lo_tail, n_all), lambda: (red.launch(0, lo_tail), red.finish())]
