"""Training-time operators on libevmi_hip (fp32, channel-major "CBT" activations [C, B, T]).

Thin host wrappers: torch allocates device memory and provides the stream, every arithmetic step is a
libevmi_hip call (convolution / unfold / fold / activation / loss / norm kernels, the library's own fp32 GEMM for plain products).
Forward functions return what their backward needs; nothing here uses torch autograd or torch math.
"""

from __future__ import annotations

import torch

from .. import _lib

# elementwise op codes (csrc/train_ops.hip)
EW_LRELU, EW_LRELU_BWD, EW_TANH, EW_TANH_BWD, EW_AXPBY, EW_SCALE, EW_MUL = 0, 1, 2, 3, 4, 5, 6
EW_SIGN_DIFF, EW_SQ_GRAD, EW_LOG_CLAMP, EW_DIV_MASK, EW_MAG, EW_MUL_DIV = 7, 8, 9, 10, 11, 12


def _s(t: torch.Tensor) -> int:
    return _lib.current_stream_ptr(t.device)


def _chk(rc: int, what: str) -> None:
    _lib.check(rc, what)


# Algorithmic FLOPs of the contractions issued since the last reset (2 per multiply-add; convolutions count their dense
# B * T_out * C_out * (C_in / groups) * k products whichever kernel serves them): bench.py prices a training step with the count
# of one eager step instead of an estimate.
_FLOPS = [0.0]


def flop_counter(reset: bool = False) -> float:
    n = _FLOPS[0]
    if reset:
        _FLOPS[0] = 0.0
    return n


def _count_conv(B, t_out, cout, cin_g, k):
    _FLOPS[0] += 2.0 * B * t_out * cout * cin_g * k


class Workspace:
    """Grow-only scratch buffers keyed by (role, stream): launches on different HIP streams may overlap on the device, so every
    stream has its own (unfold matrices, packed operands and weight fragments are large and short-lived).  They are grown
    with plain allocations outside any graph capture: run every shape once eagerly before capturing."""

    def __init__(self):
        self._bufs: dict[tuple, torch.Tensor] = {}

    def get(self, key: str, numel: int, device) -> torch.Tensor:
        k = (key, _lib.current_stream_ptr(device) if device.type == "cuda" else 0)
        buf = self._bufs.get(k)
        if buf is None or buf.numel() < numel or buf.device != device:
            if device.type == "cuda" and torch.cuda.is_current_stream_capturing():
                raise RuntimeError(f"workspace {key!r} would have to grow during graph capture: warm the step up eagerly first")
            buf = torch.empty(max(numel, 1) * 5 // 4, device=device, dtype=torch.float32)
            self._bufs[k] = buf
        return buf[:numel]


WS = Workspace()


def gemm(a, b, out, ta=False, tb=False, alpha=1.0, beta=0.0, M=None, N=None, K=None, lda=None, ldb=None, ldc=None):
    """Row-major out[M, N] = alpha * op(a) @ op(b) + beta * out (a, b, out: 2-D views or raw buffers + explicit dims)."""
    lib = _lib.load()
    if M is None:
        M, K = (a.shape[1], a.shape[0]) if ta else (a.shape[0], a.shape[1])
        N = b.shape[0] if tb else b.shape[1]
        lda, ldb, ldc = a.stride(0), b.stride(0), out.stride(0)
    _FLOPS[0] += 2.0 * M * N * K
    _chk(lib.evmi_gemm_f32(int(ta), int(tb), M, N, K, alpha, a.data_ptr(), lda, b.data_ptr(), ldb, beta, out.data_ptr(), ldc, _s(out)), "evmi_gemm_f32")
    return out


def gemm_groups(a, b, out, groups, M, N, K, lda, ldb, ldc, sa, sb, sc, ta=False, tb=False, alpha=1.0, beta=0.0):
    """``groups`` independent GEMMs at fixed element strides (one per convolution group)."""
    _FLOPS[0] += 2.0 * groups * M * N * K
    _chk(_lib.load().evmi_gemm_batched_f32(int(ta), int(tb), M, N, K, alpha, a.data_ptr(), lda, sa, b.data_ptr(), ldb, sb, beta,
                                           out.data_ptr(), ldc, sc, groups, _s(out)), "evmi_gemm_batched_f32")
    return out


def conv_out_len(t_in, k, stride, pad, dil):
    return (t_in + 2 * pad - dil * (k - 1) - 1) // stride + 1


def unfold(x, k, stride, pad, dil, key="col"):
    """x [C, B, T] -> col [C*k, B*T_out] (a workspace view; valid until the next unfold with the same key)."""
    C, B, t_in = x.shape
    t_out = conv_out_len(t_in, k, stride, pad, dil)
    if k == 1 and stride == 1 and pad == 0:
        return x.reshape(C, B * t_in), t_out
    col = WS.get(key, C * k * B * t_out, x.device)
    _chk(_lib.load().evmi_unfold_cbt_f32(x.data_ptr(), col.data_ptr(), C, B, t_in, t_out, k, stride, pad, dil, _s(x)), "evmi_unfold_cbt_f32")
    return col.view(C * k, B * t_out), t_out


def fold(dcol, C, B, t_in, t_out, k, stride, pad, dil, out=None, accumulate=False):
    """Adjoint of unfold: dcol [C*k, B*T_out] -> dx [C, B, T_in]."""
    if out is None:
        out = torch.empty(C, B, t_in, device=dcol.device, dtype=torch.float32)
    if k == 1 and stride == 1 and pad == 0 and not accumulate:
        return copy(dcol.reshape(-1), out=out.view(-1)).view(C, B, t_in)
    _chk(_lib.load().evmi_fold_cbt_f32(dcol.data_ptr(), out.data_ptr(), C, B, t_in, t_out, k, stride, pad, dil, int(accumulate), _s(out)), "evmi_fold_cbt_f32")
    return out


def elementwise(op, a, b=None, c=None, out=None, p0=0.0, p1=0.0):
    if out is None:
        out = torch.empty_like(a)
    _chk(_lib.load().evmi_elementwise_f32(op, a.data_ptr(), _lib.ptr(b), _lib.ptr(c), out.data_ptr(), a.numel(), p0, p1, _s(a)), "evmi_elementwise_f32")
    return out


EW_DIV_SCALAR, EW_STFT_GRAD_DEV, EW_FILL, EW_SCALE_DIV_SCALAR = 21, 22, 23, 24


def fill_(x, value=0.0):
    """x[:] = value with a library kernel (x contiguous)."""
    return elementwise(EW_FILL, x, out=x, p0=value)


def copy(x, out=None):
    """A contiguous copy made by a library kernel (the step issues no torch copy / fill kernels)."""
    return elementwise(EW_SCALE, x, out=out, p0=1.0)


def zeros(*shape, device):
    return fill_(torch.empty(*shape, device=device, dtype=torch.float32))


def row_reduce(mode, a, b, out, rows, n_per_row, scale=1.0, accumulate=False):
    _chk(_lib.load().evmi_row_reduce_f32(mode, a.data_ptr(), _lib.ptr(b), out.data_ptr(), rows, n_per_row, scale, int(accumulate), _s(a)), "evmi_row_reduce_f32")
    return out


def scalar_reduce(mode, a, b, out, scale=1.0, p=0.0, accumulate=False):
    _chk(_lib.load().evmi_scalar_reduce_f32(mode, a.data_ptr(), _lib.ptr(b), out.data_ptr(), a.numel(), scale, p, int(accumulate), _s(a)), "evmi_scalar_reduce_f32")
    return out


# ---- weight gradients beside the input-gradient chain -----------------------------------------------------
# Backward is a chain layer -> layer through the INPUT gradients; a layer's weight / bias gradient feeds nothing but its
# parameter-gradient buffer.  With SIDE_WGRAD["on"] (the trainers switch it on around their step) those kernels are queued on a
# sibling stream of the current one -- forked behind the tensors they read, joined by `wgrad_join` before anything consumes the
# parameter gradients (every Tape.backward ends with it) -- so the chain does not wait for them.  The tensors a side launch reads
# stay referenced until the join (see the allocator note in train/layers.py: SNConv.prepare).
SIDE_WGRAD = {"on": False}
_SIDE: dict = {}
_SIDE_STREAMS: set = set()


class _SideState:
    def __init__(self, device):
        self.stream = torch.cuda.Stream(device)
        self.pending = None
        self.keep = []      # tensors read by the launches of the group now on the sibling stream
        self.queue = []     # launches of the group being collected (closures), their tensors in keep_next
        self.keep_next = []
        # one event object per direction, re-recorded on every use (a wait takes the record that precedes it in host order);
        # creating two events and entering torch.cuda.stream() per weight gradient cost ~33 us of host time each, 3.8 ms per
        # FastSpeech2 step
        self.ready, self.done = torch.cuda.Event(), torch.cuda.Event()


def _side_state(device):
    key = _lib.current_stream_ptr(device)
    st = _SIDE.get(key)
    if st is None:
        st = _SIDE[key] = _SideState(device)
        _SIDE_STREAMS.add(st.stream.cuda_stream)
    return st


# Launches per fork: the weight gradients of SIDE_GROUP[0] layers are collected on the host and put on the sibling stream together --
# ONE fork behind the last of those layers, ONE join (of the group before) in front of it.  A fork per layer is what a captured step
# cannot use: a chain of 300 side launches each waiting for its own node of the main chain replays SERIALLY on ROCm 7.0 (the whole
# side chain after the main chain: 6.5 ms for two chains of 2.8 ms -- the FastSpeech2 step's weight gradients all ran behind its
# backward, profiles/r04x_fs2_queues_before.txt), groups of 8 / 16 in 3.2 ms, eager 7.8 -> 3.9 ms (tools/microbench/graph_side_chain.py).
# <= 1: the per-layer fork.
import os as _os

SIDE_GROUP = [int(_os.environ.get("EVMI_SIDE_GROUP", "8"))]


def _side_flush(st, cur):
    """The collected launches go to the sibling stream: the chain joins the previous group, forks here."""
    if st.pending is not None:
        cur.wait_event(st.pending)
    st.keep = st.keep_next
    st.keep_next = []
    queue, st.queue = st.queue, []
    st.ready.record(cur)
    st.stream.wait_event(st.ready)
    torch.cuda.set_stream(st.stream)
    try:
        for fn in queue:
            fn()
        st.done.record(st.stream)
    finally:
        torch.cuda.set_stream(cur)
    st.pending = st.done


class side_wgrad:
    """``side_wgrad(x, dy, ...).run(fn)``: fn launches weight-gradient kernels (on the current stream) -- on the sibling stream when
    enabled (at once with a per-layer fork, or collected into groups: SIDE_GROUP), in place otherwise.  Also a context manager
    (``with side_wgrad(x, dy): ...``) for the per-layer form."""

    def __init__(self, *tensors):
        self.tensors = [t for t in tensors if t is not None]
        self.ctx = None
        self.state = None
        self.grouped = False

    def _enabled_state(self):
        dev = self.tensors[0].device if self.tensors else None
        if not SIDE_WGRAD["on"] or dev is None or dev.type != "cuda" or _lib.current_stream_ptr(dev) in _SIDE_STREAMS:
            return None
        self.prev = torch.cuda.current_stream(dev)
        return _side_state(dev)

    def mark(self):
        """Fork point of the per-layer form: the sibling stream will wait for what the current stream holds NOW, whatever is queued
        on it before the launches are issued -- so a caller can put the chain's next kernel (the input gradient) on the device first
        and queue the weight-gradient launches, which nothing waits for, behind it in HOST order (the main queue otherwise sits idle
        while the host issues them: 85-130 us per layer in the FastSpeech2 trace).  (Grouped form: nothing to do here.)"""
        if SIDE_GROUP[0] > 1:
            self.grouped = True
            return self
        st = self.state = self._enabled_state()
        if st is None:
            return self
        st.ready.record(self.prev)
        st.stream.wait_event(st.ready)
        st.keep.extend(self.tensors)
        return self

    def run(self, fn):
        if SIDE_GROUP[0] > 1:
            st = self._enabled_state()
            if st is None:  # not enabled / already beside a chain: here
                return fn()
            st.queue.append(fn)
            st.keep_next.extend(self.tensors)
            if len(st.queue) >= SIDE_GROUP[0]:
                _side_flush(st, self.prev)
            return None
        with self:
            return fn()

    def __enter__(self):
        if self.grouped:
            raise RuntimeError("side_wgrad: the grouped form takes its launches through run()")
        if self.state is None:
            self.mark()
        if self.state is None:  # not enabled / already beside a chain: stay here
            return self
        torch.cuda.set_stream(self.state.stream)
        self.ctx = True
        return self

    def __exit__(self, *exc):
        if self.ctx is not None:
            st = self.state
            st.done.record(st.stream)
            torch.cuda.set_stream(self.prev)
            st.pending = st.done

        return False


def side_check_drained():
    """Raises if weight-gradient launches are still collected on the host (a chain ended without its ``wgrad_join``): the trainers
    call this at the end of every step."""
    if any(_LN_PENDING.values()):
        n_ln = sum(len(v) for v in _LN_PENDING.values())
        _LN_PENDING.clear()
        raise RuntimeError(f"layernorm_bwd: {n_ln} deferred parameter-gradient reductions were never issued (a backward chain ended without wgrad_join)")
    left = sum(len(st.queue) for st in _SIDE.values())
    if left:
        for st in _SIDE.values():
            st.queue.clear()
            st.keep_next.clear()
        raise RuntimeError(f"side_wgrad: {left} weight-gradient launches were never issued (a backward chain ended without wgrad_join on its stream)")


def side_reset(abort: bool = False):
    """Forget every weight-gradient launch that was collected but not issued, and every pending event.  A step that aborts half way
    -- a graph capture that fails and falls back to the eager body, an out-of-memory error a caller retries -- leaves closures in
    ``queue`` whose tensors were never written (they live in the aborted capture's pool) and a ``pending`` event recorded inside that
    capture: the next step's first flush would wait on that event and launch the stale closures with accumulate=1 into the live
    weight-gradient buffers (silent gradient corruption; ``side_check_drained`` does not see it because the flush empties
    the queue).  Called at the start of every step, and where a step aborts -- there with ``abort=True``: weight-gradient kernels
    already ISSUED on a sibling stream may still be reading the tensors ``keep`` holds (allocated on the chain's stream); the device is
    drained before they are let go, so the allocator cannot hand their blocks to the caller's retry under running kernels."""
    if abort and _SIDE and torch.cuda.is_available():
        try:
            torch.cuda.synchronize()
        except RuntimeError:  # (a capture that is still being torn down: nothing was issued)
            pass
    for st in _SIDE.values():
        st.queue.clear()
        st.keep_next.clear()
        st.keep.clear()
        st.pending = None
    _LN_PENDING.clear()


def wgrad_join(device=None):
    """The current stream waits for the weight-gradient kernels its sibling stream still has queued (the launches still collected on
    the host are issued first); their inputs may go."""
    layernorm_flush()
    if not _SIDE:
        return
    cur = torch.cuda.current_stream(device)
    st = _SIDE.get(cur.cuda_stream)
    if st is None:
        return
    if st.queue:
        _side_flush(st, cur)
    if st.pending is not None:
        cur.wait_event(st.pending)
        st.pending = None
    st.keep.clear()


# ---- convolutions ---------------------------------------------------------------------------------------
ACT_NONE, ACT_LRELU, ACT_SILU, ACT_RELU, ACT_TANH = 0, 1, 2, 3, 4


def pk_pitch(B, t) -> int:
    """Units per channel-octet row of a shared packed operand (csrc/conv_pk_common.h: pk_shared_pitch): tight items, or -- one item -- its
    length rounded up to the weight gradient's K step (the units behind the last column are zero)."""
    return (t + 63) // 64 * 64 if B == 1 else B * t


def shares_packed(B, t, k, stride, pad, dil, groups) -> bool:
    """True for the layers whose packed bf16 operands serve forward, input gradient AND weight gradient (pointwise, stride 1, B * t a
    multiple of the weight gradient's K step)."""
    return bool(_packed() and _lib.load().evmi_conv1d_bf16pk_shares_packed(B, t, k, stride, pad, dil, groups))


def conv1d_mfma(x, w, bias, stride=1, pad=0, dil=1, groups=1, out=None, n_out=None, out_stride=1, out_offset=0,
                accumulate=False, act=ACT_NONE, act_param=0.0, keep=None):
    """fp32 matrix-core implicit GEMM (evmi_conv1d_cbt_f32): x [Cin, B, T], w [Cout, Cin/groups, k] -> y [Cout, B, T_out].
    ``keep`` (a dict): on the packed bf16 path of a pointwise stride-1 layer the call gets a workspace of its own, left in
    ``keep["x_packed"]`` -- its head is the packed input, which the layer's weight gradient reads again (no second pack)."""
    cin, B, t_in = x.shape
    cout, cin_g, k = w.shape
    t_conv = conv_out_len(t_in, k, stride, pad, dil)
    if out is None:
        out = torch.empty(cout, B, t_conv, device=x.device, dtype=torch.float32)
    lib = _lib.load()
    _count_conv(B, t_conv if n_out is None else n_out, cout, cin_g, k)
    if CONV_BACKEND["operands"] == "bf16" and CONV_BACKEND["packed"]:
        n_eff = t_conv if n_out is None else n_out
        pk_elems = lib.evmi_conv1d_cbt_bf16pk_ws_elems(B, cin, t_in, cout, n_eff, k, stride, pad, dil, groups)
        if pk_elems > 0:  # packed bf16 copy of x + A fragments (conv_cbt_bf16_pk.hip)
            if keep is not None and n_out is None and shares_packed(B, t_in, k, stride, pad, dil, groups):
                ws = keep["x_packed"] = torch.empty(pk_elems, device=x.device, dtype=torch.float32)
            else:
                ws = WS.get("pk", pk_elems, x.device)
            _chk(lib.evmi_conv1d_cbt_bf16pk(x.data_ptr(), w.data_ptr(), _lib.ptr(bias), out.data_ptr(), ws.data_ptr(), pk_elems, B, cin, t_in,
                                            cout, out.shape[2], n_eff, k, stride, pad, dil, groups, out_stride, out_offset,
                                            int(accumulate), act, float(act_param), _s(x)), "evmi_conv1d_cbt_bf16pk")
            return out
    wf_elems = lib.evmi_conv1d_cbt_f32_ws_elems(B, cin, cout, t_conv if n_out is None else n_out, k, groups)
    wf = WS.get("wfrag", wf_elems, x.device)
    fn = lib.evmi_conv1d_cbt_bf16 if CONV_BACKEND["operands"] == "bf16" else lib.evmi_conv1d_cbt_f32
    _chk(fn(x.data_ptr(), w.data_ptr(), _lib.ptr(bias), out.data_ptr(), wf.data_ptr(), wf_elems, B, cin, t_in, cout,
            out.shape[2], t_conv if n_out is None else n_out, k, stride, pad, dil, groups,
            out_stride, out_offset, int(accumulate), act, float(act_param), _s(x)), "evmi_conv1d_cbt")
    return out



# ---- fused forms (packed bf16 kernels take the neighbours of a convolution into their pack / epilogue; every other path runs
# the same arithmetic as separate passes, so callers use ONE signature whatever kernel serves the shape) --------------------
def _packed() -> bool:
    return CONV_BACKEND["operands"] == "bf16" and CONV_BACKEND["packed"]


def conv1d_fused_fwd(x, w, bias, stride=1, pad=0, dil=1, groups=1, act=ACT_NONE, act_param=0.0, pre_slope=1.0, residual=None):
    """y = act(conv(leaky_relu(x, pre_slope)) + bias) + residual."""
    cin, B, t_in = x.shape
    cout, _, k = w.shape
    t_out = conv_out_len(t_in, k, stride, pad, dil)
    lib = _lib.load()
    if _packed() and CONV_BACKEND["fwd"] == "mfma":
        pk = lib.evmi_conv1d_cbt_bf16pk_ws_elems(B, cin, t_in, cout, t_out, k, stride, pad, dil, groups)
        if pk > 0:
            _count_conv(B, t_out, cout, cin // groups, k)
            y = torch.empty(cout, B, t_out, device=x.device, dtype=torch.float32)
            ws = WS.get("pk", pk, x.device)
            _chk(lib.evmi_conv1d_cbt_bf16pk_fused(x.data_ptr(), w.data_ptr(), _lib.ptr(bias), y.data_ptr(), ws.data_ptr(), pk, B, cin, t_in, cout, t_out,
                                                  t_out, k, stride, pad, dil, groups, act, float(act_param), float(pre_slope), _lib.ptr(residual),
                                                  _s(x)), "evmi_conv1d_cbt_bf16pk_fused")
            return y
    xa = x if pre_slope == 1.0 else lrelu(x, pre_slope)
    y = conv1d_fwd(xa, w, bias, stride, pad, dil, groups, lrelu_slope=act_param if act == ACT_LRELU else None, act=act if act != ACT_LRELU else ACT_NONE)
    return y if residual is None else axpby(1.0, y, 1.0, residual, out=y)


def conv1d_fused_dgrad(dy, w, t_in, stride=1, pad=0, dil=1, groups=1, dy_mask=None, dy_mask_slope=1.0, dx_mask=None, dx_mask_slope=1.0,
                       residual=None, x_for_fallback=None):
    """dx = conv_input_grad(dy * lrelu'(dy_mask)) * lrelu'(dx_mask) + residual  (lrelu'(m) = 1 where m > 0, else the slope)."""
    cout, B, t_out = dy.shape
    _, cin_g, k = w.shape
    cin = cin_g * groups
    lib = _lib.load()
    if _packed() and CONV_BACKEND["dgrad"] == "mfma" and not ((dx_mask is not None or residual is not None) and k < stride):
        pk = lib.evmi_conv1d_dgrad_cbt_bf16pk_ws_elems(B, cin, t_in, cout, t_out, k, stride, pad, dil, groups)
        if pk > 0:
            _count_conv(B, t_out, cout, cin_g, k)
            dx = zeros(cin, B, t_in, device=dy.device) if k < stride else torch.empty(cin, B, t_in, device=dy.device, dtype=torch.float32)
            ws = WS.get("pk", pk, dy.device)
            _chk(lib.evmi_conv1d_dgrad_cbt_bf16pk_fused(dy.data_ptr(), w.data_ptr(), dx.data_ptr(), ws.data_ptr(), pk, B, cin, t_in, cout, t_out, k, stride,
                                                        pad, dil, groups, 1.0, _lib.ptr(dy_mask), float(dy_mask_slope), _lib.ptr(dx_mask),
                                                        float(dx_mask_slope), _lib.ptr(residual), _s(dy)), "evmi_conv1d_dgrad_cbt_bf16pk_fused")
            return dx
    dpre = dy if dy_mask is None else lrelu_bwd(dy, dy_mask, dy_mask_slope)
    x_dummy = x_for_fallback if x_for_fallback is not None else torch.empty(cin, B, t_in, device=dy.device, dtype=torch.float32)
    dx, _, _ = conv1d_bwd(x_dummy, w, dpre, stride, pad, dil, groups, need_dx=True, need_dw=False)
    if dx_mask is not None:
        dx = lrelu_bwd(dx, dx_mask, dx_mask_slope)
    return dx if residual is None else axpby(1.0, dx, 1.0, residual, out=dx)


def conv1d_fused_wgrad(x, w_shape, dy, dw_out, stride=1, pad=0, dil=1, groups=1, x_pre_slope=1.0, accumulate=True):
    """dw (+)= conv_weight_grad(leaky_relu(x, x_pre_slope), dy)."""
    cin, B, t_in = x.shape
    cout, _, k = w_shape
    t_out = dy.shape[2]
    lib = _lib.load()
    if _packed() and CONV_BACKEND["wgrad"] != "gemm":
        pk = lib.evmi_conv1d_wgrad_cbt_bf16pk_ws_elems(B, cin, t_in, cout, t_out, k, stride, pad, dil, groups)
        if pk > 0:
          _count_conv(B, t_out, cout, cin // groups, k)
          def launch():
            ws = WS.get("pkw", pk, x.device)
            _chk(lib.evmi_conv1d_wgrad_cbt_bf16pk_fused(x.data_ptr(), dy.data_ptr(), dw_out.data_ptr(), ws.data_ptr(), pk, B, cin, t_in, cout, t_out, k,
                                                        stride, pad, dil, groups, int(accumulate), float(x_pre_slope), 0, 1.0, _s(x)),
                 "evmi_conv1d_wgrad_cbt_bf16pk_fused")
          side_wgrad(x, dy, dw_out).run(launch)
          return dw_out
    xa = x if x_pre_slope == 1.0 else lrelu(x, x_pre_slope)
    w_dummy = dw_out  # (only its shape is read on this path)
    conv1d_bwd(xa, w_dummy, dy, stride, pad, dil, groups, need_dx=False, dw_out=dw_out, accumulate=accumulate)
    return dw_out


def dgrad_mfma_supported(B, cin, t_in, cout, t_out, k, stride, dil, groups) -> bool:
    """The input gradient runs as stride-1 convolutions of dy (c_out channels) with at most ceil(k / stride) taps."""
    if stride > 1 and dil != 1:
        return False
    M = k if stride == 1 else (k + stride - 1) // stride
    n_out = t_in if stride == 1 else max(1, t_in // stride)
    return mfma_conv_supported(B, cout, t_out, cin, n_out, M, 1, dil if stride == 1 else 1, groups)


def conv1d_bwd_data_mfma(dy, w, t_in, stride=1, pad=0, dil=1, groups=1, keep=None):
    """Input gradient of conv1d on the fp32 matrix cores: `stride` polyphase stride-1 convolutions of dy with
    re-indexed weights, each writing its own residue class of dx (strided layers all have dilation 1)."""
    cout, B, t_out = dy.shape
    _, cin_g, k = w.shape
    cin = cin_g * groups
    lib = _lib.load()
    if CONV_BACKEND["operands"] == "bf16" and CONV_BACKEND["packed"]:
        pk_elems = lib.evmi_conv1d_dgrad_cbt_bf16pk_ws_elems(B, cin, t_in, cout, t_out, k, stride, pad, dil, groups)
        if pk_elems > 0:
            _count_conv(B, t_out, cout, cin_g, k)
            if keep is not None and shares_packed(B, t_out, k, stride, pad, dil, groups):  # (``keep["dy_packed"]``: see conv1d_mfma)
                ws = keep["dy_packed"] = torch.empty(pk_elems, device=dy.device, dtype=torch.float32)
            else:
                ws = WS.get("pk", pk_elems, dy.device)
            dx = zeros(cin, B, t_in, device=dy.device) if k < stride else torch.empty(cin, B, t_in, device=dy.device, dtype=torch.float32)
            _chk(lib.evmi_conv1d_dgrad_cbt_bf16pk(dy.data_ptr(), w.data_ptr(), dx.data_ptr(), ws.data_ptr(), pk_elems, B, cin, t_in, cout,
                                                  t_out, k, stride, pad, dil, groups, _s(dy)), "evmi_conv1d_dgrad_cbt_bf16pk")
            return dx
    ws_elems = lib.evmi_conv1d_dgrad_cbt_f32_ws_elems(B, cin, t_in, cout, t_out, k, stride, pad, dil, groups)
    if ws_elems > 0:  # every phase in one launch, weight fragments straight from w
        _count_conv(B, t_out, cout, cin_g, k)
        ws = WS.get("wfrag", ws_elems, dy.device)
        dx = zeros(cin, B, t_in, device=dy.device) if k < stride else torch.empty(cin, B, t_in, device=dy.device, dtype=torch.float32)
        fn = lib.evmi_conv1d_dgrad_cbt_bf16 if CONV_BACKEND["operands"] == "bf16" else lib.evmi_conv1d_dgrad_cbt_f32
        _chk(fn(dy.data_ptr(), w.data_ptr(), dx.data_ptr(), ws.data_ptr(), ws_elems, B, cin, t_in, cout, t_out,
                k, stride, pad, dil, groups, _s(dy)), "evmi_conv1d_dgrad_cbt")
        return dx
    if stride == 1:
        wt = WS.get("wt", cin * (cout // groups) * k, dy.device)
        _chk(lib.evmi_dgrad_weights_f32(w.data_ptr(), wt.data_ptr(), cin, cout, k, groups, 1, 0, _s(dy)), "evmi_dgrad_weights_f32")
        dx = torch.empty(cin, B, t_in, device=dy.device, dtype=torch.float32)
        return conv1d_mfma(dy, wt.view(cin, cout // groups, k), None, 1, dil * (k - 1) - pad, dil, groups, out=dx, n_out=t_in)
    assert dil == 1, "strided convolutions of this model are not dilated"
    dx = zeros(cin, B, t_in, device=dy.device)
    for phi in range(min(stride, k)):
        M = (k - phi + stride - 1) // stride
        q0 = max(0, -((phi - pad) // stride))          # first q with stride*q + phi - pad >= 0
        q_hi = (t_in - 1 + pad - phi) // stride         # last q with stride*q + phi - pad <= t_in - 1
        n_out = q_hi - q0 + 1
        if n_out <= 0:
            continue
        wt = WS.get("wt", cin * (cout // groups) * M, dy.device)
        _chk(lib.evmi_dgrad_weights_f32(w.data_ptr(), wt.data_ptr(), cin, cout, k, groups, stride, phi, _s(dy)), "evmi_dgrad_weights_f32")
        # dx[stride*q + phi - pad] = sum_m' wt[m'] * dy[q - (M-1) + m'],  q = q0 + to
        conv1d_mfma(dy, wt.view(cin, cout // groups, M), None, 1, (M - 1) - q0, 1, groups, out=dx, n_out=n_out,
                    out_stride=stride, out_offset=stride * q0 + phi - pad)
    return dx


# "mfma": the hand-written fp32 matrix-core convolution kernels; "gemm": unfold + the plain fp32 GEMM (A/B and reference variant).
# wgrad "auto": implicit GEMM for grouped layers (2-5x over unfold + per-group GEMMs), unfold + ONE library GEMM for dense
# ones in fp32 mode (one long-K product, split over K).
# "operands": "f32" = exact fp32 fmaf chains on the fp32-input matrix cores; "bf16" = the same kernels round both operands to
# bf16 on their way into v_mfma_f32_32x32x16_bf16 (fp32 accumulation, fp32 tensors in HBM, fp32 master weights).
# "packed": bf16 operands go through the packed-input kernel (conv_cbt_bf16_pk.hip) where it takes the shape; False = always the
# in-LDS rounding variant of the fp32 kernels (A/B switch).
CONV_BACKEND = {"fwd": "mfma", "dgrad": "mfma", "wgrad": "mfma", "operands": "f32", "packed": True}


def wgrad_takes_bf16(B, cin, t_in, cout, t_out, k, stride, pad, dil, groups) -> bool:
    """True where precision="bf16" computes this layer's weight gradient from bf16-rounded operands (conv_wgrad_bf16_pk.hip takes
    the shape); narrower groups keep exact fp32 operands.  The ONE statement of that rule: tests restating the arithmetic for
    the oracle ask here instead of copying the predicate."""
    return (CONV_BACKEND["packed"] and CONV_BACKEND["wgrad"] != "gemm"
            and _lib.load().evmi_conv1d_wgrad_cbt_bf16pk_ws_elems(B, cin, t_in, cout, t_out, k, stride, pad, dil, groups) > 0)


def fwd_takes_bf16(B, cin, t_in, cout, t_out, k, stride, pad, dil, groups) -> bool:
    """True where precision="bf16" rounds the operands of this layer's forward convolution (and, with the roles of the channels
    swapped, of its input gradient) to bf16; GEMV / outer-product shapes and groups narrower than the staging stay exact.
    The library's own dispatch rule (evmi_conv1d_cbt_bf16_rounds): tests restating the arithmetic ask here."""
    if not CONV_BACKEND["packed"]:
        raise RuntimeError("fwd_takes_bf16 states the rule of the default (packed) configuration")
    return bool(_lib.load().evmi_conv1d_cbt_bf16_rounds(B, cin, t_in, cout, t_out, k, stride, pad, dil, groups))


def mfma_conv_supported(B, cin, t_in, cout, n_out, k, stride, dil, groups) -> bool:
    """False for the degenerate shapes the matrix-core kernel does not stage (rows of a few samples under a long kernel)."""
    return bool(_lib.load().evmi_conv1d_cbt_f32_supported(B, cin, t_in, cout, n_out, k, stride, dil, groups))


_ACT_EW = {ACT_SILU: 13, ACT_RELU: 14, ACT_TANH: EW_TANH}


def conv1d_fwd(x, w, bias, stride=1, pad=0, dil=1, groups=1, lrelu_slope=None, act=ACT_NONE, keep=None):
    """x [Cin, B, T], w [Cout, Cin/groups, k] -> y [Cout, B, T_out] (leaky-relu applied in the epilogue when a slope is given;
    ``act``: one of the other epilogue activations)."""
    cin, B, t_in = x.shape
    cout, cin_g, k = w.shape
    if act != ACT_NONE:
        if CONV_BACKEND["fwd"] == "mfma" and mfma_conv_supported(B, cin, t_in, cout, conv_out_len(t_in, k, stride, pad, dil), k, stride, dil, groups):
            return conv1d_mfma(x, w, bias, stride, pad, dil, groups, act=act, keep=keep)
        y = conv1d_fwd(x, w, bias, stride, pad, dil, groups)
        return elementwise(_ACT_EW[act], y, out=y)
    if CONV_BACKEND["fwd"] == "mfma" and mfma_conv_supported(B, cin, t_in, cout, conv_out_len(t_in, k, stride, pad, dil), k, stride, dil, groups):
        if lrelu_slope is None:
            return conv1d_mfma(x, w, bias, stride, pad, dil, groups, keep=keep)
        return conv1d_mfma(x, w, bias, stride, pad, dil, groups, act=ACT_LRELU, act_param=lrelu_slope, keep=keep)
    if lrelu_slope is not None:
        y = conv1d_fwd(x, w, bias, stride, pad, dil, groups)
        return elementwise(EW_LRELU, y, out=y, p0=lrelu_slope)
    col, t_out = unfold(x, k, stride, pad, dil)
    y = torch.empty(cout, B, t_out, device=x.device, dtype=torch.float32)
    N = B * t_out
    cout_g = cout // groups
    kg = cin_g * k
    # Y_g [cout_g, N] = W_g [cout_g, kg] . col_g [kg, N]
    gemm_groups(w, col, y, groups, cout_g, N, kg, kg, N, N, cout_g * kg, kg * N, cout_g * N)
    if bias is not None:
        _chk(_lib.load().evmi_bias_add_rows_f32(y.data_ptr(), bias.data_ptr(), cout, N, _s(y)), "evmi_bias_add_rows_f32")
    return y


def _weight_and_bias_grad(x, w_shape, dy, dw, db_out, stride, pad, dil, groups, accumulate, packed=None, x_standin=False):
    """dw (+)= conv_weight_grad(x, dy) and, when asked, db_out (+)= row sums of dy -- on whichever kernel takes the shape.
    ``x_standin``: x only lends its shape (the layer's real input exists as ``packed["x_packed"]`` alone: fused LayerNorm -> dense,
    fused feed-forward middle); any path that would read x itself is an error then, not a fallback."""
    cin, B, t_in = x.shape
    cout, cin_g, k = w_shape
    t_out = dy.shape[2]
    N = B * t_out
    cout_g = cout // groups
    lib = _lib.load()
    pk_elems = (lib.evmi_conv1d_wgrad_cbt_bf16pk_ws_elems(B, cin, t_in, cout, t_out, k, stride, pad, dil, groups)
                if CONV_BACKEND["operands"] == "bf16" and CONV_BACKEND["packed"] and CONV_BACKEND["wgrad"] != "gemm" else 0)
    ws_elems = 0
    if pk_elems == 0 and (CONV_BACKEND["wgrad"] == "mfma" or (CONV_BACKEND["wgrad"] == "auto" and groups > 1)):
        ws_elems = lib.evmi_conv1d_wgrad_cbt_f32_ws_elems(B, cin, t_in, cout, t_out, k, stride, pad, dil, groups)
    if pk_elems > 0 or ws_elems > 0:
        _count_conv(B, t_out, cout, cin_g, k)
    if pk_elems > 0:  # bf16 operands: packed dy and x, transposing LDS reads (conv_wgrad_bf16_pk.hip)
        ws = WS.get("pkw", pk_elems, x.device)
        xp = packed.get("x_packed") if packed else None
        dyp = packed.get("dy_packed") if packed else None
        prepacked = (xp is not None or dyp is not None) and shares_packed(B, t_in, k, stride, pad, dil, groups)
        if x_standin and not (prepacked and xp is not None):
            raise RuntimeError("weight gradient of a fused layer: the packed input is gone or the shape left the pre-packed path "
                               "(x is a stand-in for a tensor that was never stored)")
        if prepacked:
            # the forward's packed input / the input gradient's packed dy, read where they lie (no second pack)
            _chk(lib.evmi_conv1d_wgrad_cbt_bf16pk_prepacked(x.data_ptr(), _lib.ptr(xp), dy.data_ptr(), _lib.ptr(dyp), dw.data_ptr(), ws.data_ptr(),
                                                            pk_elems, B, cin, t_in, cout, t_out, k, stride, pad, dil, groups, int(accumulate),
                                                            _s(x)), "evmi_conv1d_wgrad_cbt_bf16pk_prepacked")
        else:
            _chk(lib.evmi_conv1d_wgrad_cbt_bf16pk(x.data_ptr(), dy.data_ptr(), dw.data_ptr(), ws.data_ptr(), pk_elems, B, cin, t_in, cout, t_out,
                                                  k, stride, pad, dil, groups, int(accumulate), _s(x)), "evmi_conv1d_wgrad_cbt_bf16pk")
    elif x_standin:
        raise RuntimeError("weight gradient of a fused layer fell off the packed bf16 kernels (backend switched since the forward?): "
                           "x is a stand-in for a tensor that was never stored")
    elif ws_elems > 0:  # implicit GEMM on the fp32 matrix cores (conv_wgrad_f32_mfma.hip)
        ws = WS.get("wgrad", ws_elems, x.device)
        _chk(lib.evmi_conv1d_wgrad_cbt_f32(x.data_ptr(), dy.data_ptr(), dw.data_ptr(), ws.data_ptr(), ws_elems, B, cin, t_in, cout, t_out,
                                           k, stride, pad, dil, groups, int(accumulate), _s(x)), "evmi_conv1d_wgrad_cbt_f32")
    else:
        col = x if (k == 1 and stride == 1 and pad == 0) else unfold(x, k, stride, pad, dil)[0]
        kg = cin_g * k
        # dW_g [cout_g, kg] (+)= dY_g [cout_g, N] . col_g^T
        gemm_groups(dy, col, dw, groups, cout_g, kg, N, N, N, kg, cout_g * N, kg * N, cout_g * kg, tb=True, beta=1.0 if accumulate else 0.0)
    if db_out is not None:
        return row_reduce(0, dy, None, db_out, cout, N, accumulate=accumulate)
    return None


def conv1d_bwd(x, w, dy, stride=1, pad=0, dil=1, groups=1, need_dx=True, dw_out=None, db_out=None, accumulate=False,
               need_dw=True, packed=None, x_standin=False):
    """Returns (dx, dw, db); dw/db are written (or accumulated) into the given buffers when provided.
    ``packed``: the dict the layer's forward call filled (``conv1d_fwd(..., keep=packed)``): a pointwise stride-1 layer on the packed
    bf16 kernels then packs x once (forward) and dy once (input gradient) for all three of its products."""
    cin, B, t_in = x.shape
    cout, cin_g, k = w.shape
    t_out = dy.shape[2]
    N = B * t_out
    cout_g = cout // groups
    dym = dy.view(cout, N)
    wm = w.reshape(cout, cin_g * k)
    dw = db = None
    side = None
    share = packed is not None and need_dw and shares_packed(B, t_in, k, stride, pad, dil, groups)
    share_dy = (share and need_dx and CONV_BACKEND["dgrad"] == "mfma" and dgrad_mfma_supported(B, cin, t_in, cout, t_out, k, stride, dil, groups)
                and _lib.load().evmi_conv1d_dgrad_cbt_bf16pk_ws_elems(B, cin, t_in, cout, t_out, k, stride, pad, dil, groups) > 0)
    staged = None
    if share_dy:
        # dy is packed FIRST, as a launch of its own, so that the weight gradient's stream can fork behind the pack and in front of
        # the input gradient: both read the one packed dy side by side.  (Forking behind the whole input gradient instead -- one more
        # cross-stream edge per layer on the chain's critical path -- cost a captured FastSpeech2 step 4 ms: 21.9 -> 25.8 ms.)
        lib = _lib.load()
        pk_elems = lib.evmi_conv1d_dgrad_cbt_bf16pk_ws_elems(B, cin, t_in, cout, t_out, k, stride, pad, dil, groups)
        ws = packed["dy_packed"] = torch.empty(pk_elems, device=dy.device, dtype=torch.float32)
        dx = torch.empty(cin, B, t_in, device=dy.device, dtype=torch.float32)
        staged = (lib, ws, pk_elems, dx)
        _chk(lib.evmi_conv1d_dgrad_cbt_bf16pk_staged(1, dy.data_ptr(), w.data_ptr(), dx.data_ptr(), ws.data_ptr(), pk_elems, B, cin, t_in, cout, t_out,
                                                     k, stride, pad, dil, groups, _s(dy)), "evmi_conv1d_dgrad_cbt_bf16pk_staged")
    if need_dw:
        dw = dw_out if dw_out is not None else torch.empty_like(w)
        side = side_wgrad(x, dy, dw, db_out, *(packed.values() if share else ())).mark()  # fork here; the launches are issued behind the input gradient's (below)
    dx = None
    if staged is not None:
        lib, ws, pk_elems, dx = staged
        _count_conv(B, t_out, cout, cin_g, k)
        _chk(lib.evmi_conv1d_dgrad_cbt_bf16pk_staged(2, dy.data_ptr(), w.data_ptr(), dx.data_ptr(), ws.data_ptr(), pk_elems, B, cin, t_in, cout, t_out,
                                                     k, stride, pad, dil, groups, _s(dy)), "evmi_conv1d_dgrad_cbt_bf16pk_staged")
    elif need_dx and CONV_BACKEND["dgrad"] == "mfma" and dgrad_mfma_supported(B, cin, t_in, cout, t_out, k, stride, dil, groups):
        dx = conv1d_bwd_data_mfma(dy, w, t_in, stride, pad, dil, groups)
    elif need_dx:
        pointwise = k == 1 and stride == 1 and pad == 0
        dcol = (torch.empty(cin, B * t_in, device=x.device, dtype=torch.float32) if pointwise
                else WS.get("dcol", cin * k * N, x.device).view(cin * k, N))
        kg = cin_g * k
        # dcol_g [kg, N] = W_g^T . dY_g
        gemm_groups(w, dy, dcol, groups, kg, N, cout_g, kg, N, N, cout_g * kg, cout_g * N, kg * N, ta=True)
        dx = dcol.view(cin, B, t_in) if pointwise else fold(dcol, cin, B, t_in, t_out, k, stride, pad, dil)
    if side is not None:
        db = db_out  # (what _weight_and_bias_grad returns: the caller's bias-gradient buffer, filled by the launch)
        pk = dict(packed) if share else None  # (a snapshot: the caller clears its dict when this returns, the launch may come later)
        if x_standin and not (share and pk.get("x_packed") is not None):
            raise RuntimeError("conv1d_bwd(x_standin=True): the forward's packed input is required")
        side.run(lambda: _weight_and_bias_grad(x, w.shape, dy, dw, db_out, stride, pad, dil, groups, accumulate, packed=pk, x_standin=x_standin))
    return dx, dw, db


# ---- dense2(dropout(silu(a))) and its backward with the activation / mask applied while the operands are packed ---------------------
def ffn_fused_supported(B, t, c_mid, c_out) -> bool:
    """True where both halves below run on the packed bf16 kernels with shared packed operands (pointwise layers, B * t a multiple of
    the weight gradient's K step): the FastSpeech2 feed-forward blocks at precision="bf16"."""
    if not (_packed() and CONV_BACKEND["fwd"] == "mfma" and CONV_BACKEND["dgrad"] == "mfma" and CONV_BACKEND["wgrad"] != "gemm"):
        return False
    lib = _lib.load()
    geo = (t, 1, 1, 0, 1, 1)
    return bool(shares_packed(B, t, 1, 1, 0, 1, 1)
                # layer 1 (c_out -> c_mid): forward, input gradient, weight gradient
                and lib.evmi_conv1d_cbt_bf16pk_ws_elems(B, c_out, t, c_mid, *geo) > 0
                and lib.evmi_conv1d_dgrad_cbt_bf16pk_ws_elems(B, c_out, t, c_mid, *geo) > 0
                and lib.evmi_conv1d_wgrad_cbt_bf16pk_ws_elems(B, c_out, t, c_mid, *geo) > 0
                and dgrad_mfma_supported(B, c_out, t, c_mid, t, 1, 1, 1, 1)
                # layer 2 (c_mid -> c_out): its weight gradient reads the packed dropout(silu(a)) -- the fp32 tensor handed to
                # conv1d_bwd is the pre-activation a, a stand-in (probe these too)
                and lib.evmi_conv1d_cbt_bf16pk_ws_elems(B, c_mid, t, c_out, *geo) > 0
                and lib.evmi_conv1d_dgrad_cbt_bf16pk_ws_elems(B, c_mid, t, c_out, *geo) > 0
                and lib.evmi_conv1d_wgrad_cbt_bf16pk_ws_elems(B, c_mid, t, c_out, *geo) > 0
                and dgrad_mfma_supported(B, c_mid, t, c_out, t, 1, 1, 1, 1))


def ln_dense_fused_supported(B, t, c_in, c_out) -> bool:
    """True where LayerNorm -> pointwise layer runs with the normalised tensor written straight into the layer's packed input."""
    if not (_packed() and CONV_BACKEND["fwd"] == "mfma" and CONV_BACKEND["dgrad"] == "mfma" and CONV_BACKEND["wgrad"] != "gemm"):
        return False
    lib = _lib.load()
    return bool(c_in in (128, 256) and shares_packed(B, t, 1, 1, 0, 1, 1)
                and lib.evmi_conv1d_cbt_bf16pk_ws_elems(B, c_in, t, c_out, t, 1, 1, 0, 1, 1) > 0
                and lib.evmi_conv1d_wgrad_cbt_bf16pk_ws_elems(B, c_in, t, c_out, t, 1, 1, 0, 1, 1) > 0)


def layernorm_dense_fwd(x, gamma, beta, w, bias, keep, act=ACT_NONE, eps=1e-5):
    """y = act(dense(LayerNorm(x))) (pointwise w [cout, cin, 1]); LayerNorm(x) exists only packed, in ``keep["x_packed"]`` (the layer's
    weight gradient reads it there).  Caller: ln_dense_fused_supported."""
    cin, B, t = x.shape
    cout = w.shape[0]
    lib = _lib.load()
    pk_elems = lib.evmi_conv1d_cbt_bf16pk_ws_elems(B, cin, t, cout, t, 1, 1, 0, 1, 1)
    ws = keep["x_packed"] = torch.empty(pk_elems, device=x.device, dtype=torch.float32)
    out = torch.empty(cout, B, t, device=x.device, dtype=torch.float32)
    # (the layer's weight fragments are prepared by LayerNorm's launch: the convolution starts without a preparation launch of its own)
    _chk(lib.evmi_layernorm_pack_bf16pk_w(x.data_ptr(), gamma.data_ptr(), beta.data_ptr(), ws.data_ptr(), pk_elems, B, cin, t, cout, eps, w.data_ptr(), None, None,
                                          0, 0, _s(x)), "evmi_layernorm_pack_bf16pk_w")
    _count_conv(B, t, cout, cin, 1)
    _chk(lib.evmi_conv1d_cbt_bf16pk_prepacked(w.data_ptr(), _lib.ptr(bias), out.data_ptr(), ws.data_ptr(), pk_elems, B, cin, t, cout, act, 0.0, 1, _s(x)),
         "evmi_conv1d_cbt_bf16pk_prepacked")
    return out


def conv1d_fwd_silu_dropout(a, w, bias, p, seed, keep):
    """y = dense(dropout(silu(a), p)) (pointwise w [cout, cin, 1]); the activated, masked tensor exists only packed, in ``keep["x_packed"]``
    (the layer's weight gradient reads it there).  Caller: ffn_fused_supported."""
    cin, B, t = a.shape
    cout = w.shape[0]
    lib = _lib.load()
    pk_elems = lib.evmi_conv1d_cbt_bf16pk_ws_elems(B, cin, t, cout, t, 1, 1, 0, 1, 1)
    ws = keep["x_packed"] = torch.empty(pk_elems, device=a.device, dtype=torch.float32)
    out = torch.empty(cout, B, t, device=a.device, dtype=torch.float32)
    _count_conv(B, t, cout, cin, 1)
    _chk(lib.evmi_conv1d_cbt_bf16pk_silu_dropout(a.data_ptr(), w.data_ptr(), _lib.ptr(bias), out.data_ptr(), ws.data_ptr(), pk_elems, B, cin, t, cout, 1, 0,
                                                 float(p), int(seed), _lib.ptr(SEED_BASE[0]), _s(a)), "evmi_conv1d_cbt_bf16pk_silu_dropout")
    return out


def conv1d_bwd_silu_dropout_dy(x, w, ds, pre, p, seed, dw_out, db_out, packed):
    """Backward of a pointwise layer y = w x + b whose output gradient is dy = dropout(ds, p) * silu'(pre) (y = pre feeds the activation
    behind it): dy is formed while ds is packed -- it exists only as the packed bf16 operand that the input gradient, the weight gradient
    and the bias gradient (row sums of the packed rows) all read.  Returns dx; dw_out / db_out are accumulated into.  ``packed``: the
    forward's ``keep`` dict (x_packed).  Caller: ffn_fused_supported."""
    cin, B, t = x.shape
    cout = w.shape[0]
    lib = _lib.load()
    pk_elems = lib.evmi_conv1d_dgrad_cbt_bf16pk_ws_elems(B, cin, t, cout, t, 1, 1, 0, 1, 1)
    ws = torch.empty(pk_elems, device=ds.device, dtype=torch.float32)
    dx = torch.empty(cin, B, t, device=ds.device, dtype=torch.float32)
    args = (ds.data_ptr(), pre.data_ptr(), float(p), int(seed), _lib.ptr(SEED_BASE[0]), w.data_ptr(), dx.data_ptr(), ws.data_ptr(), pk_elems, B, cin, t, cout, t,
            1, 1, 0, 1, 1)
    _chk(lib.evmi_conv1d_dgrad_cbt_bf16pk_staged_silu_dropout(1, *args, _s(ds)), "evmi_conv1d_dgrad_cbt_bf16pk_staged_silu_dropout")
    xp = packed.get("x_packed") if packed else None
    side = side_wgrad(x, ds, pre, dw_out, db_out, ws, xp).mark()  # fork behind the pack, in front of the input gradient
    _count_conv(B, t, cout, cin, 1)
    _chk(lib.evmi_conv1d_dgrad_cbt_bf16pk_staged_silu_dropout(2, *args, _s(ds)), "evmi_conv1d_dgrad_cbt_bf16pk_staged_silu_dropout")

    def launch():
        n_w = lib.evmi_conv1d_wgrad_cbt_bf16pk_ws_elems(B, cin, t, cout, t, 1, 1, 0, 1, 1)
        wsw = WS.get("pkw", n_w, x.device)
        _count_conv(B, t, cout, cin, 1)
        _chk(lib.evmi_conv1d_wgrad_cbt_bf16pk_prepacked(x.data_ptr(), _lib.ptr(xp), ds.data_ptr(), ws.data_ptr(), dw_out.data_ptr(), wsw.data_ptr(), n_w, B, cin, t,
                                                        cout, t, 1, 1, 0, 1, 1, 1, _s(x)), "evmi_conv1d_wgrad_cbt_bf16pk_prepacked")
        if db_out is not None:  # row sums of the packed dy: [cout / 8 octet rows][B * t units]
            job = (_lib.PkFlatRows * 1)()
            job[0].dy, job[0].plane, job[0].units, job[0].C, job[0].db = ws.data_ptr(), pk_pitch(B, t), pk_pitch(B, t), cout, db_out.data_ptr()
            n_r = lib.evmi_pkflat_rowsum_ws_elems(1, job)
            wsr = WS.get("pkrow", n_r, x.device)
            _chk(lib.evmi_pkflat_rowsum(1, job, wsr.data_ptr(), n_r, _s(x)), "evmi_pkflat_rowsum")

    side.run(launch)
    return dx


# ---- a feed-forward block whose wide middle tensor exists packed only ----------------------------------------------------------
FFN_PACKED = [_os.environ.get("EVMI_FS2_FFN_PACKED", "1") != "0"]  # A/B switch (tools/fs2_train_bench.py)


def ffn_packed_supported(B, t, c_in, c_mid, c_out) -> bool:
    """True where LayerNorm -> dense(c_in -> c_mid) -> SiLU -> dropout -> dense(c_mid -> c_out) -> (+ residual, dropout) runs as the packed
    chain of ffn_packed_fwd / ffn_packed_bwd (every fusion it is built from is available)."""
    return bool(FFN_PACKED[0] and c_mid % 8 == 0 and ffn_fused_supported(B, t, c_mid, c_out) and ln_dense_fused_supported(B, t, c_in, c_mid)
                and resdrop_fused_supported(B, t, c_mid, c_out))


def ffn_packed_fwd(x, gamma, beta, w1, b1, w2, b2, res, p, seed, seed_out, scale, keep, eps=1e-5):
    """res + scale * dropout(dense2(dropout(silu(dense1(LayerNorm(x))), p)), p) with NO fp32 copy of the c_mid-channel tensors: LayerNorm
    writes dense1's packed input, dense1's epilogue writes bf16(a) (``keep["a_pk"]``, for the backward) and dense2's packed input
    dropout(silu(a)) (``keep["s_packed"]``), dense2's epilogue adds the residual.  ``keep["x_packed"]``: LayerNorm(x), packed."""
    cin, B, t = x.shape
    cmid, cout = w1.shape[0], w2.shape[0]
    lib = _lib.load()
    geo = (t, 1, 1, 0, 1, 1)
    n1 = lib.evmi_conv1d_cbt_bf16pk_ws_elems(B, cin, t, cmid, *geo)
    n2 = lib.evmi_conv1d_cbt_bf16pk_ws_elems(B, cmid, t, cout, *geo)
    ws1 = keep["x_packed"] = torch.empty(n1, device=x.device, dtype=torch.float32)
    ws2 = keep["s_packed"] = torch.empty(n2, device=x.device, dtype=torch.float32)
    a_pk = keep["a_pk"] = torch.empty(cmid // 8 * pk_pitch(B, t) * 4, device=x.device, dtype=torch.float32)  # 16-byte units of 8 bf16 channels
    out = torch.empty(cout, B, t, device=x.device, dtype=torch.float32)
    st = _s(x)
    # three launches: LayerNorm (+ the weight fragments of both layers), dense1, dense2
    _chk(lib.evmi_layernorm_pack_bf16pk_w(x.data_ptr(), gamma.data_ptr(), beta.data_ptr(), ws1.data_ptr(), n1, B, cin, t, cmid, eps, w1.data_ptr(), w2.data_ptr(),
                                          ws2.data_ptr(), n2, cout, st), "evmi_layernorm_pack_bf16pk_w")
    _count_conv(B, t, cmid, cin, 1)
    _chk(lib.evmi_conv1d_cbt_bf16pk_ffn_up(w1.data_ptr(), _lib.ptr(b1), ws1.data_ptr(), n1, a_pk.data_ptr(), ws2.data_ptr(), n2, B, cin, t, cmid, cout, float(p),
                                           int(seed), _lib.ptr(SEED_BASE[0]), 1, st), "evmi_conv1d_cbt_bf16pk_ffn_up")
    _count_conv(B, t, cout, cmid, 1)
    _chk(lib.evmi_conv1d_cbt_bf16pk_resdrop(3, None, w2.data_ptr(), _lib.ptr(b2), res.data_ptr(), out.data_ptr(), ws2.data_ptr(), n2, B, cmid, t, cout, 0.0, 0,
                                            float(p), int(seed_out), float(scale), _lib.ptr(SEED_BASE[0]), st), "evmi_conv1d_cbt_bf16pk_resdrop")
    return out


def ffn_packed_infer(x, gamma, beta, w1, b1, w2, b2, res, eps=1e-5):
    """res + dense2(silu(dense1(LayerNorm(x)))) without dropout (inference): three launches -- LayerNorm (+ the weight fragments of both
    layers), dense1 (its epilogue writes silu(a) as dense2's packed input; the pre-activation is not stored), dense2 (+ residual)."""
    cin, B, t = x.shape
    cmid, cout = w1.shape[0], w2.shape[0]
    lib = _lib.load()
    geo = (t, 1, 1, 0, 1, 1)
    n1 = lib.evmi_conv1d_cbt_bf16pk_ws_elems(B, cin, t, cmid, *geo)
    n2 = lib.evmi_conv1d_cbt_bf16pk_ws_elems(B, cmid, t, cout, *geo)
    ws1 = torch.empty(n1, device=x.device, dtype=torch.float32)
    ws2 = torch.empty(n2, device=x.device, dtype=torch.float32)
    out = torch.empty(cout, B, t, device=x.device, dtype=torch.float32)
    st = _s(x)
    _chk(lib.evmi_layernorm_pack_bf16pk_w(x.data_ptr(), gamma.data_ptr(), beta.data_ptr(), ws1.data_ptr(), n1, B, cin, t, cmid, eps, w1.data_ptr(), w2.data_ptr(),
                                          ws2.data_ptr(), n2, cout, st), "evmi_layernorm_pack_bf16pk_w")
    _count_conv(B, t, cmid, cin, 1)
    _chk(lib.evmi_conv1d_cbt_bf16pk_ffn_up(w1.data_ptr(), _lib.ptr(b1), ws1.data_ptr(), n1, None, ws2.data_ptr(), n2, B, cin, t, cmid, cout, 0.0, 0, None, 1, st),
         "evmi_conv1d_cbt_bf16pk_ffn_up")
    _count_conv(B, t, cout, cmid, 1)
    _chk(lib.evmi_conv1d_cbt_bf16pk_resdrop(3, None, w2.data_ptr(), _lib.ptr(b2), res.data_ptr(), out.data_ptr(), ws2.data_ptr(), n2, B, cmid, t, cout, 0.0, 0,
                                            0.0, 0, 1.0, None, st), "evmi_conv1d_cbt_bf16pk_resdrop")
    return out


def ffn_packed_bwd(x, w1, w2, dy, p, seed, seed_out, scale, dw1, db1, dw2, db2, keep):
    """Backward of ffn_packed_fwd up to LayerNorm: returns d LayerNorm(x) [c_in, B, t] (fp32); the weight and bias gradients of both
    layers go to the sibling stream (side_wgrad) reading the packed operands.  dz = scale * dropout(dy) is packed once; the second
    layer's input gradient leaves its epilogue as the first layer's packed dy = dropout(ds) * silu'(a)."""
    cin, B, t = x.shape
    cmid, cout = w1.shape[0], w2.shape[0]
    lib = _lib.load()
    geo = (t, 1, 1, 0, 1, 1)
    n_d2 = lib.evmi_conv1d_dgrad_cbt_bf16pk_ws_elems(B, cmid, t, cout, *geo)
    n_d1 = lib.evmi_conv1d_dgrad_cbt_bf16pk_ws_elems(B, cin, t, cmid, *geo)
    wsd2 = torch.empty(n_d2, device=dy.device, dtype=torch.float32)
    wsd1 = torch.empty(n_d1, device=dy.device, dtype=torch.float32)
    dh = torch.empty(cin, B, t, device=dy.device, dtype=torch.float32)
    ws1, ws2, a_pk = keep["x_packed"], keep["s_packed"], keep["a_pk"]
    st = _s(dy)
    # second layer: pack dz (stage 1), fork its weight gradient, input gradient into the first layer's packed dy
    args2 = (dy.data_ptr(), float(p), int(seed_out), _lib.ptr(SEED_BASE[0]), float(scale), w2.data_ptr(), wsd2.data_ptr(), wsd2.data_ptr(), n_d2, B, cmid, t, cout, t,
             1, 1, 0, 1, 1)
    _chk(lib.evmi_conv1d_dgrad_cbt_bf16pk_staged_dropout(1, *args2, st), "evmi_conv1d_dgrad_cbt_bf16pk_staged_dropout")
    side2 = side_wgrad(dy, dw2, db2, wsd2, ws2).mark()
    _count_conv(B, t, cout, cmid, 1)
    _chk(lib.evmi_conv1d_dgrad_cbt_bf16pk_ffn_down(w2.data_ptr(), wsd2.data_ptr(), n_d2, a_pk.data_ptr(), wsd1.data_ptr(), n_d1, B, cin, t, cmid, cout, float(p),
                                                   int(seed), _lib.ptr(SEED_BASE[0]), st), "evmi_conv1d_dgrad_cbt_bf16pk_ffn_down")

    def wgrad(c_i, c_o, xp, dyp, dw, db):
        def launch():
            n_w = lib.evmi_conv1d_wgrad_cbt_bf16pk_ws_elems(B, c_i, t, c_o, *geo)
            wsw = WS.get("pkw", n_w, x.device)
            _count_conv(B, t, c_o, c_i, 1)
            # (the fp32 operands are stand-ins: both packed copies are given)
            _chk(lib.evmi_conv1d_wgrad_cbt_bf16pk_prepacked(xp.data_ptr(), xp.data_ptr(), dyp.data_ptr(), dyp.data_ptr(), dw.data_ptr(), wsw.data_ptr(), n_w, B, c_i, t,
                                                            c_o, t, 1, 1, 0, 1, 1, 1, _s(x)), "evmi_conv1d_wgrad_cbt_bf16pk_prepacked")
            if db is not None:  # row sums of the packed output gradient: [c_o / 8 octet rows][B * t units]
                job = (_lib.PkFlatRows * 1)()
                job[0].dy, job[0].plane, job[0].units, job[0].C, job[0].db = dyp.data_ptr(), pk_pitch(B, t), pk_pitch(B, t), c_o, db.data_ptr()
                n_r = lib.evmi_pkflat_rowsum_ws_elems(1, job)
                wsr = WS.get("pkrow", n_r, x.device)
                _chk(lib.evmi_pkflat_rowsum(1, job, wsr.data_ptr(), n_r, _s(x)), "evmi_pkflat_rowsum")
        return launch

    side2.run(wgrad(cmid, cout, ws2, wsd2, dw2, db2))
    side1 = side_wgrad(x, dw1, db1, wsd1, ws1, a_pk).mark()  # fork behind the epilogue that wrote the packed dy, in front of the input gradient
    _count_conv(B, t, cmid, cin, 1)
    _chk(lib.evmi_conv1d_dgrad_cbt_bf16pk_staged(3, wsd1.data_ptr(), w1.data_ptr(), dh.data_ptr(), wsd1.data_ptr(), n_d1, B, cin, t, cmid, t, 1, 1, 0, 1, 1, st),
         "evmi_conv1d_dgrad_cbt_bf16pk_staged")
    side1.run(wgrad(cin, cmid, ws1, wsd1, dw1, db1))
    return dh


# ---- a + s * dropout(dense(h)): the residual add and the dropout in the layer's epilogue, their backward in the pack of dy ---------------
RESDROP_FUSION = [_os.environ.get("EVMI_FS2_RESDROP", "1") != "0"]  # A/B switch of the fusion below (tools/fs2_train_bench.py)


def resdrop_fused_supported(B, t, c_in, c_out) -> bool:
    """True where a pointwise layer c_in -> c_out runs on the packed bf16 kernels with shared packed operands in all three products
    (the FastSpeech2 sub-layers' last dense layers at precision="bf16")."""
    if not RESDROP_FUSION[0] or not (_packed() and CONV_BACKEND["fwd"] == "mfma" and CONV_BACKEND["dgrad"] == "mfma" and CONV_BACKEND["wgrad"] != "gemm"):
        return False
    lib = _lib.load()
    geo = (t, 1, 1, 0, 1, 1)
    return bool(shares_packed(B, t, 1, 1, 0, 1, 1)
                and lib.evmi_conv1d_cbt_bf16pk_ws_elems(B, c_in, t, c_out, *geo) > 0
                and lib.evmi_conv1d_dgrad_cbt_bf16pk_ws_elems(B, c_in, t, c_out, *geo) > 0
                and lib.evmi_conv1d_wgrad_cbt_bf16pk_ws_elems(B, c_in, t, c_out, *geo) > 0
                and dgrad_mfma_supported(B, c_in, t, c_out, t, 1, 1, 1, 1))


def conv1d_fwd_resdrop(x, w, bias, res, p, seed, scale, keep, in_p=None, in_seed=0):
    """res + scale * dropout(dense(in) + bias, p) in ONE launch pair (pack + convolution): in = x, or dropout(silu(x), in_p) when
    ``in_p`` is given (the second layer of a feed-forward block: conv1d_fwd_silu_dropout's input fusion).  The packed input stays in
    ``keep["x_packed"]`` for the weight gradient.  Caller: resdrop_fused_supported (and ffn_fused_supported for the input fusion)."""
    cin, B, t = x.shape
    cout = w.shape[0]
    lib = _lib.load()
    pk_elems = lib.evmi_conv1d_cbt_bf16pk_ws_elems(B, cin, t, cout, t, 1, 1, 0, 1, 1)
    ws = keep["x_packed"] = torch.empty(pk_elems, device=x.device, dtype=torch.float32)
    out = torch.empty(cout, B, t, device=x.device, dtype=torch.float32)
    _count_conv(B, t, cout, cin, 1)
    _chk(lib.evmi_conv1d_cbt_bf16pk_resdrop(0 if in_p is None else 1, x.data_ptr(), w.data_ptr(), _lib.ptr(bias), res.data_ptr(), out.data_ptr(), ws.data_ptr(),
                                            pk_elems, B, cin, t, cout, float(in_p or 0.0), int(in_seed), float(p), int(seed), float(scale),
                                            _lib.ptr(SEED_BASE[0]), _s(x)), "evmi_conv1d_cbt_bf16pk_resdrop")
    return out


def conv1d_bwd_dropout_dy(x, w, dy, p, seed, scale, dw_out, db_out, packed, x_standin=False):
    """Backward of a pointwise layer z = w x + b behind which sits y = a + scale * dropout(z, p): the layer's output gradient
    dz = scale * dropout(dy, p) is formed while dy is packed -- it exists only as the packed bf16 operand that the input gradient, the
    weight gradient and the bias gradient (row sums of the packed rows) read.  Returns dx; dw_out / db_out are accumulated into.
    ``packed``: the forward's ``keep`` dict (x_packed)."""
    cin, B, t = x.shape
    cout = w.shape[0]
    lib = _lib.load()
    pk_elems = lib.evmi_conv1d_dgrad_cbt_bf16pk_ws_elems(B, cin, t, cout, t, 1, 1, 0, 1, 1)
    ws = torch.empty(pk_elems, device=dy.device, dtype=torch.float32)
    dx = torch.empty(cin, B, t, device=dy.device, dtype=torch.float32)
    args = (dy.data_ptr(), float(p), int(seed), _lib.ptr(SEED_BASE[0]), float(scale), w.data_ptr(), dx.data_ptr(), ws.data_ptr(), pk_elems, B, cin, t, cout, t,
            1, 1, 0, 1, 1)
    _chk(lib.evmi_conv1d_dgrad_cbt_bf16pk_staged_dropout(1, *args, _s(dy)), "evmi_conv1d_dgrad_cbt_bf16pk_staged_dropout")
    xp = packed.get("x_packed") if packed else None
    if xp is None:
        raise RuntimeError("conv1d_bwd_dropout_dy: the forward's packed input is required")
    side = side_wgrad(x, dy, dw_out, db_out, ws, xp).mark()  # fork behind the pack, in front of the input gradient
    _count_conv(B, t, cout, cin, 1)
    _chk(lib.evmi_conv1d_dgrad_cbt_bf16pk_staged_dropout(2, *args, _s(dy)), "evmi_conv1d_dgrad_cbt_bf16pk_staged_dropout")

    def launch():
        n_w = lib.evmi_conv1d_wgrad_cbt_bf16pk_ws_elems(B, cin, t, cout, t, 1, 1, 0, 1, 1)
        wsw = WS.get("pkw", n_w, x.device)
        _count_conv(B, t, cout, cin, 1)
        _chk(lib.evmi_conv1d_wgrad_cbt_bf16pk_prepacked(x.data_ptr(), _lib.ptr(xp), dy.data_ptr(), ws.data_ptr(), dw_out.data_ptr(), wsw.data_ptr(), n_w, B, cin, t,
                                                        cout, t, 1, 1, 0, 1, 1, 1, _s(x)), "evmi_conv1d_wgrad_cbt_bf16pk_prepacked")
        if db_out is not None:  # row sums of the packed dz: [cout / 8 octet rows][B * t units]
            job = (_lib.PkFlatRows * 1)()
            job[0].dy, job[0].plane, job[0].units, job[0].C, job[0].db = ws.data_ptr(), pk_pitch(B, t), pk_pitch(B, t), cout, db_out.data_ptr()
            n_r = lib.evmi_pkflat_rowsum_ws_elems(1, job)
            wsr = WS.get("pkrow", n_r, x.device)
            _chk(lib.evmi_pkflat_rowsum(1, job, wsr.data_ptr(), n_r, _s(x)), "evmi_pkflat_rowsum")

    side.run(launch)
    return dx


def _convt_via_dgrad():
    import os
    if CONV_BACKEND["dgrad"] != "mfma":
        return False
    if CONV_BACKEND["operands"] == "bf16":
        return CONV_BACKEND["packed"]
    return os.environ.get("EVMI_CONVT_F32_DGRAD", "1") == "1"


def conv_transpose1d_fwd(x, w, bias, stride, pad):
    """x [Cin, B, T], w [Cin, Cout, k] -> y [Cout, B, (T-1)*stride - 2*pad + k]  (= dgrad of a strided conv)."""
    cin, B, t_in = x.shape
    _, cout, k = w.shape
    t_out = (t_in - 1) * stride - 2 * pad + k
    if _convt_via_dgrad() and dgrad_mfma_supported(B, cout, t_out, cin, t_in, k, stride, 1, 1):
        # a transposed convolution IS the input gradient of the strided convolution with the same weight tensor
        # (w [c_in, c_out, k] read as [conv c_out][conv c_in][k]): the polyphase matrix-core kernels (packed bf16 or fp32), one launch
        y = conv1d_bwd_data_mfma(x, w, t_out, stride, pad, 1, 1)
        if bias is not None:
            _chk(_lib.load().evmi_bias_add_rows_f32(y.data_ptr(), bias.data_ptr(), cout, B * t_out, _s(y)), "evmi_bias_add_rows_f32")
        return y
    col = WS.get("dcol", cout * k * B * t_in, x.device).view(cout * k, B * t_in)
    gemm(w.reshape(cin, cout * k), x.view(cin, B * t_in), col, ta=True)
    y = fold(col, cout, B, t_out, t_in, k, stride, pad, 1)
    if bias is not None:
        _chk(_lib.load().evmi_bias_add_rows_f32(y.data_ptr(), bias.data_ptr(), cout, B * t_out, _s(y)), "evmi_bias_add_rows_f32")
    return y


def conv_transpose1d_bwd(x, w, dy, stride, pad, need_dx=True, dw_out=None, db_out=None, accumulate=False, need_dw=True):
    cin, B, t_in = x.shape
    _, cout, k = w.shape
    t_out = dy.shape[2]
    if _convt_via_dgrad() and dgrad_mfma_supported(B, cout, t_out, cin, t_in, k, stride, 1, 1):
        # the adjoint pair of the above: dx = conv1d(dy, w), dw = weight gradient of that convolution with (input, output
        # gradient) = (dy, x) -- both in the transposed convolution's own weight layout [c_in, c_out, k]
        db = row_reduce(0, dy, None, db_out, cout, B * t_out, accumulate=accumulate) if db_out is not None else None
        dw = None
        if need_dw:
            _, dw, _ = conv1d_bwd(dy, w, x, stride, pad, 1, 1, need_dx=False, dw_out=dw_out, accumulate=accumulate)
        dx = conv1d_fwd(dy, w, None, stride, pad, 1, 1) if need_dx else None
        return dx, dw, db
    col, t_chk = unfold(dy, k, stride, pad, 1)  # [Cout*k, B*T_in]
    assert t_chk == t_in
    dw = None
    if need_dw:
        dw = dw_out if dw_out is not None else torch.empty_like(w)
        gemm(x.view(cin, B * t_in), col, dw.view(cin, cout * k), tb=True, beta=1.0 if accumulate else 0.0)
    db = None
    if db_out is not None:
        db = row_reduce(0, dy, None, db_out, cout, B * t_out, accumulate=accumulate)
    dx = None
    if need_dx:
        dx = torch.empty(cin, B, t_in, device=x.device, dtype=torch.float32)
        gemm(w.reshape(cin, cout * k), col, dx.view(cin, B * t_in))
    return dx, dw, db


# ---- iSTFTNet head ---------------------------------------------------------------------------------------------
def istft_polar(a, H):
    """a [2H, B, T] (log-magnitude rows, phase rows) -> [2H, B, T] real rows, imaginary rows of exp(a) * exp(i sin(b))."""
    s = torch.empty_like(a)
    _chk(_lib.load().evmi_istft_polar_f32(a.data_ptr(), s.data_ptr(), H, a.numel() // (2 * H), _s(a)), "evmi_istft_polar_f32")
    return s


def istft_polar_bwd(a, ds, H):
    da = torch.empty_like(a)
    _chk(_lib.load().evmi_istft_polar_bwd_f32(a.data_ptr(), ds.data_ptr(), da.data_ptr(), H, a.numel() // (2 * H), _s(a)), "evmi_istft_polar_bwd_f32")
    return da


def reflect_pad_left1(x):
    C, B, T = x.shape
    y = torch.empty(C, B, T + 1, device=x.device, dtype=torch.float32)
    _chk(_lib.load().evmi_reflect_pad_left1_f32(x.data_ptr(), y.data_ptr(), C * B, T, 0, _s(x)), "evmi_reflect_pad_left1_f32")
    return y


def reflect_pad_left1_bwd(dy):
    C, B, T1 = dy.shape
    dx = torch.empty(C, B, T1 - 1, device=dy.device, dtype=torch.float32)
    _chk(_lib.load().evmi_reflect_pad_left1_f32(dy.data_ptr(), dx.data_ptr(), C * B, T1 - 1, 1, _s(dy)), "evmi_reflect_pad_left1_f32")
    return dx


# ---- activations / pooling / views -------------------------------------------------------------------------
def lrelu(x, slope):
    return elementwise(EW_LRELU, x, p0=slope)


def lrelu_bwd_rowsum(dy, y, slope, db_out, accumulate=True):
    """dpre = dy * (y > 0 ? 1 : slope) and db_out (+)= row sums of dpre, in one pass (dy, y: [C, B, T])."""
    dpre = torch.empty_like(dy)
    C = dy.shape[0]
    _chk(_lib.load().evmi_lrelu_bwd_rowsum_f32(dy.data_ptr(), y.data_ptr(), dpre.data_ptr(), db_out.data_ptr(), C, dy.numel() // C, slope,
                                               int(accumulate), _s(dy)), "evmi_lrelu_bwd_rowsum_f32")
    return dpre


def lrelu_bwd(dy, x, slope):
    return elementwise(EW_LRELU_BWD, dy, x, p0=slope)


def tanh(x):
    return elementwise(EW_TANH, x)


def tanh_bwd(dy, y):
    return elementwise(EW_TANH_BWD, dy, y)


def axpby(a, x, b, y, out=None):
    return elementwise(EW_AXPBY, x, y, out=out, p0=a, p1=b)


def avgpool4s2(x):
    C, B, t_in = x.shape
    y = torch.empty(C, B, t_in // 2 + 1, device=x.device, dtype=torch.float32)
    _chk(_lib.load().evmi_avgpool4s2_f32(x.data_ptr(), y.data_ptr(), C * B, t_in, 0, _s(x)), "evmi_avgpool4s2_f32")
    return y


def avgpool4s2_bwd(dy, t_in):
    C, B, _ = dy.shape
    dx = torch.empty(C, B, t_in, device=dy.device, dtype=torch.float32)
    _chk(_lib.load().evmi_avgpool4s2_f32(dy.data_ptr(), dx.data_ptr(), C * B, t_in, 1, _s(dy)), "evmi_avgpool4s2_f32")
    return dx


def period_view(x, period):
    """x [1, B, T] -> [1, B*period, ceil(T/period)] (reflect padding on the right, as DiscriminatorP)."""
    _, B, T = x.shape
    H = (T + period - 1) // period
    y = torch.empty(1, B * period, H, device=x.device, dtype=torch.float32)
    _chk(_lib.load().evmi_period_view_f32(x.data_ptr(), y.data_ptr(), B, T, period, 0, _s(x)), "evmi_period_view_f32")
    return y


def period_view_bwd(dy, B, T, period):
    dx = torch.empty(1, B, T, device=dy.device, dtype=torch.float32)
    _chk(_lib.load().evmi_period_view_f32(dy.data_ptr(), dx.data_ptr(), B, T, period, 1, _s(dy)), "evmi_period_view_f32")
    return dx


# ---- norms / optimiser -----------------------------------------------------------------------------------------
def weight_norm_fwd(g, v):
    rows = v.shape[0]
    n = v.numel() // rows
    w = torch.empty_like(v)
    norm = torch.empty(rows, device=v.device, dtype=torch.float32)
    _chk(_lib.load().evmi_weight_norm_fwd_f32(g.data_ptr(), v.data_ptr(), w.data_ptr(), norm.data_ptr(), rows, n, _s(v)), "evmi_weight_norm_fwd_f32")
    return w, norm


def weight_norm_bwd(g, v, norm, dw, dg_out, dv_out):
    rows = v.shape[0]
    n = v.numel() // rows
    _chk(_lib.load().evmi_weight_norm_bwd_f32(g.data_ptr(), v.data_ptr(), norm.data_ptr(), dw.data_ptr(), dg_out.data_ptr(), dv_out.data_ptr(), rows, n, _s(v)), "evmi_weight_norm_bwd_f32")


def normalize_vec(x, out, eps=1e-12):
    _chk(_lib.load().evmi_normalize_vec_f32(x.data_ptr(), out.data_ptr(), x.numel(), eps, _s(x)), "evmi_normalize_vec_f32")
    return out


def adamw_step(p, g, m, v, lr, betas, eps, weight_decay, step):
    _chk(_lib.load().evmi_adamw_f32(p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), p.numel(), lr, betas[0], betas[1], eps, weight_decay, step, _s(p)), "evmi_adamw_f32")


def stft_frames(x, n_fft, hop):
    """x [B, T] -> frames [n_fft, B*F], F = 1 + T // hop (centred, reflect padding; window lives in the DFT basis)."""
    B, T = x.shape
    F = 1 + T // hop
    fr = torch.empty(n_fft, B * F, device=x.device, dtype=torch.float32)
    _chk(_lib.load().evmi_stft_frames_f32(x.data_ptr(), fr.data_ptr(), B, T, n_fft, hop, 0, _s(x)), "evmi_stft_frames_f32")
    return fr, F


def stft_frames_bwd(dfr, B, T, n_fft, hop):
    dx = torch.empty(B, T, device=dfr.device, dtype=torch.float32)
    _chk(_lib.load().evmi_stft_frames_f32(dfr.data_ptr(), dx.data_ptr(), B, T, n_fft, hop, 1, _s(dfr)), "evmi_stft_frames_f32")
    return dx


# ---- FastSpeech2 training operators (csrc/fs2_train_ops.hip) -----------------------------------------------------------
EW_SILU, EW_RELU, EW_GLU, EW_SILU_BWD, EW_RELU_BWD, EW_CLIP_SCALE = 13, 14, 15, 18, 19, 20


def layernorm(x, gamma, beta, eps=1e-5):
    y = torch.empty_like(x)
    _chk(_lib.load().evmi_layernorm_cbt_f32(x.data_ptr(), gamma.data_ptr(), beta.data_ptr(), y.data_ptr(), x.shape[0], x.shape[1] * x.shape[2],
                                            eps, _s(x)), "evmi_layernorm_cbt_f32")
    return y


# LayerNorm parameter gradients reduced once per backward chain: with LN_DEFER["on"] (the FastSpeech2 trainer, around its step) a
# LayerNorm backward leaves its per-workgroup partial sums in a buffer of its own and ``wgrad_join`` -- where every chain ends --
# reduces all the lists collected since in one launch (55 small launches on the step's critical chain otherwise).
LN_DEFER = {"on": False}
_LN_PENDING = {}  # stream -> [(partials, dgamma, dbeta, C, N)]: a chain's partials are reduced on the stream that produced them


def layernorm_flush():
    """Reduce the partial lists the deferred LayerNorm backwards of the CURRENT stream left."""
    if not _LN_PENDING:
        return
    for key in list(_LN_PENDING):
        pend = _LN_PENDING[key]
        if not pend:
            del _LN_PENDING[key]
            continue
        dev = pend[0][0].device
        if _lib.current_stream_ptr(dev) != key:
            continue
        # one launch adds every job's column sums into its targets without atomics, the jobs side by side: two deferred backwards of
        # ONE LayerNorm (a layer applied twice in a chain) go into separate, stream-ordered launches
        rounds: list = []
        for job in pend:
            for r in rounds:
                if all(job[1].data_ptr() != o[1].data_ptr() and job[2].data_ptr() != o[2].data_ptr() for o in r):
                    r.append(job)
                    break
            else:
                rounds.append([job])
        try:
            for r in rounds:
                jobs = (_lib.LnPartials * len(r))()
                for j, (ws, dg, db, C, N) in zip(jobs, r):
                    j.ws, j.dgamma, j.dbeta, j.C, j.n_cols = ws.data_ptr(), dg.data_ptr(), db.data_ptr(), C, N
                _chk(_lib.load().evmi_layernorm_bwd_partials_reduce(len(r), jobs, key), "evmi_layernorm_bwd_partials_reduce")
        finally:
            del _LN_PENDING[key]


def layernorm_bwd(x, gamma, dy, dgamma, dbeta, eps=1e-5, acc_into=None):
    """dx; dgamma / dbeta are accumulated into (with LN_DEFER["on"]: by the chain's ``wgrad_join``).  ``acc_into``: a tensor of x's shape
    the kernel ADDS dx to (and which is returned) -- the input's gradient so far, e.g. the residual path's: no separate add pass."""
    lib = _lib.load()
    C, N = x.shape[0], x.shape[1] * x.shape[2]
    n = lib.evmi_layernorm_bwd_cbt_f32_ws_elems(C, N)
    acc = 0
    if acc_into is not None:
        assert acc_into.shape == x.shape and acc_into.is_contiguous() and acc_into.dtype == torch.float32
        dx, acc = acc_into, 1
    else:
        dx = torch.empty_like(x)
    if LN_DEFER["on"] and x.is_cuda:
        ws = torch.empty(n, device=x.device, dtype=torch.float32)
        _chk(lib.evmi_layernorm_bwd_cbt_f32(x.data_ptr(), gamma.data_ptr(), dy.data_ptr(), dx.data_ptr(), None, None, ws.data_ptr(), n, C, N, eps, acc, _s(x)),
             "evmi_layernorm_bwd_cbt_f32")
        _LN_PENDING.setdefault(_s(x), []).append((ws, dgamma, dbeta, C, N))
        return dx
    ws = WS.get("ln_bwd", n, x.device)
    _chk(lib.evmi_layernorm_bwd_cbt_f32(x.data_ptr(), gamma.data_ptr(), dy.data_ptr(), dx.data_ptr(), dgamma.data_ptr(), dbeta.data_ptr(),
                                        ws.data_ptr(), n, C, N, eps, acc, _s(x)), "evmi_layernorm_bwd_cbt_f32")
    return dx


def batchnorm_fwd(x, gamma, beta, running_mean, running_var, act=ACT_NONE, eps=1e-5, momentum=0.1):
    C, N = x.shape[0], x.shape[1] * x.shape[2]
    y = torch.empty_like(x)
    mean = torch.empty(C, device=x.device, dtype=torch.float32)
    rstd = torch.empty_like(mean)
    # momentum < 0: evaluation mode (running statistics normalise and stay as they are)
    _chk(_lib.load().evmi_batchnorm_fwd_cbt_f32(x.data_ptr(), gamma.data_ptr(), beta.data_ptr(), y.data_ptr(), mean.data_ptr(), rstd.data_ptr(),
                                                _lib.ptr(running_mean), _lib.ptr(running_var), C, N, eps, momentum, act, _s(x)), "evmi_batchnorm_fwd_cbt_f32")
    return y, mean, rstd


def batchnorm_bwd(x, gamma, beta, mean, rstd, dy, dgamma, dbeta, act=ACT_NONE):
    dx = torch.empty_like(x)
    _chk(_lib.load().evmi_batchnorm_bwd_cbt_f32(x.data_ptr(), gamma.data_ptr(), beta.data_ptr(), mean.data_ptr(), rstd.data_ptr(), dy.data_ptr(),
                                                dx.data_ptr(), dgamma.data_ptr(), dbeta.data_ptr(), x.shape[0], x.shape[1] * x.shape[2], act, _s(x)),
         "evmi_batchnorm_bwd_cbt_f32")
    return dx


def dwconv_fwd(x, w, bias, k):
    y = torch.empty_like(x)
    C, B, T = x.shape
    _chk(_lib.load().evmi_dwconv1d_cbt_f32(x.data_ptr(), w.data_ptr(), _lib.ptr(bias), y.data_ptr(), C, B, T, k, (k - 1) // 2, 0, _s(x)), "evmi_dwconv1d_cbt_f32")
    return y


def dwconv_bwd(x, w, dy, dw, db, k, need_dx=True):
    C, B, T = x.shape
    dx = torch.empty_like(x) if need_dx else None
    lib = _lib.load()
    n = lib.evmi_dwconv1d_bwd_cbt_f32_ws_elems(C, B, k)
    side = side_wgrad(x, dy, dw, db).mark()
    if need_dx:
        _chk(lib.evmi_dwconv1d_bwd_cbt_f32(x.data_ptr(), w.data_ptr(), dy.data_ptr(), dx.data_ptr(), 0, 0, 0, 0, C, B, T, k, (k - 1) // 2, _s(x)),
             "evmi_dwconv1d_bwd_cbt_f32")
    def launch():  # the filter / bias gradient (partial sums per item, then the reduction) beside the chain, issued behind dx
        ws = WS.get("dw_bwd", n, x.device)
        _chk(lib.evmi_dwconv1d_bwd_cbt_f32(x.data_ptr(), w.data_ptr(), dy.data_ptr(), 0, dw.data_ptr(), db.data_ptr(), ws.data_ptr(), n, C, B, T, k,
                                           (k - 1) // 2, _s(x)), "evmi_dwconv1d_bwd_cbt_f32")
    side.run(launch)
    return dx


# A device-resident base added to every dropout seed ([1] int64 tensor, or None): the FastSpeech2 trainer stores
# (seed, step, rank) << 16 there before each step and passes the draw's index as `seed`, so a step captured into a HIP graph
# draws fresh masks on every replay (include/evmi.h: evmi_dropout_f32).
SEED_BASE = [None]


def store_f32(dst: torch.Tensor, values) -> None:
    """dst[: len(values)] = values (<= 8 floats), carried in a kernel's argument block (no host buffer to keep alive)."""
    import ctypes as C

    vals = [float(v) for v in values]
    arr = (C.c_float * len(vals))(*vals)
    _chk(_lib.load().evmi_store_f32(dst.data_ptr(), len(vals), arr, _s(dst)), "evmi_store_f32")


def store_u64(dst: torch.Tensor, value: int) -> None:
    _chk(_lib.load().evmi_store_u64(dst.data_ptr(), int(value) & 0xFFFFFFFFFFFFFFFF, _s(dst)), "evmi_store_u64")


def dropout(x, p, seed):
    y = torch.empty_like(x)
    _chk(_lib.load().evmi_dropout_f32(x.data_ptr(), y.data_ptr(), x.numel(), p, seed, _lib.ptr(SEED_BASE[0]), _s(x)), "evmi_dropout_f32")
    return y


def dropout_fused(mode, a, b, p, seed, scale=1.0):
    """evmi_dropout_fused_f32: 1: b + scale * drop(a); 2: drop(silu(a)); 3: drop(a) * silu'(b); 4: scale * drop(a)."""
    y = torch.empty_like(a)
    _chk(_lib.load().evmi_dropout_fused_f32(mode, a.data_ptr(), _lib.ptr(b), y.data_ptr(), a.numel(), p, seed, _lib.ptr(SEED_BASE[0]), float(scale), _s(a)),
         "evmi_dropout_fused_f32")
    return y


def glu_bwd(p, dy):
    dp = torch.empty_like(p)
    _chk(_lib.load().evmi_glu_bwd_f32(p.data_ptr(), dy.data_ptr(), dp.data_ptr(), dy.numel(), _s(p)), "evmi_glu_bwd_f32")
    return dp


def mask_cols_(x, lens32):
    C, B, T = x.shape
    _chk(_lib.load().evmi_mask_cols_f32(x.data_ptr(), lens32.data_ptr(), C, B, T, _s(x)), "evmi_mask_cols_f32")
    return x


def attention_train_fwd(qkv, lens32, heads, p=0.0, seed=0):
    """Multi-head self-attention in training mode (evmi_mha_fwd_f32, flash-style): qkv [3D, B, T] -> (out [D, B, T], saved).
    Only the per-query log-sum-exp is kept for the backward; attention dropout as torch.nn.MultiheadAttention applies it
    (on the normalised probabilities), mask regenerated from (seed + head, element index) in the backward."""
    D3, B, T = qkv.shape
    D = D3 // 3
    out = torch.empty(D, B, T, device=qkv.device, dtype=torch.float32)
    lse = torch.empty(B, heads, T, device=qkv.device, dtype=torch.float32)
    lib = _lib.load()
    fn = lib.evmi_mha_fwd_bf16 if CONV_BACKEND["operands"] == "bf16" else lib.evmi_mha_fwd_f32
    _chk(fn(qkv.data_ptr(), lens32.data_ptr(), out.data_ptr(), lse.data_ptr(), B, T, D, heads, float(p), int(seed), _lib.ptr(SEED_BASE[0]), _s(qkv)), "evmi_mha_fwd")
    return out, (out, lse, lens32)


def attention_train_bwd(qkv, saved, dout, heads, p=0.0, seed=0):
    out, lse, lens32 = saved
    D3, B, T = qkv.shape
    dqkv = torch.empty_like(qkv)
    dsum = torch.empty_like(lse)
    lib = _lib.load()
    fn = lib.evmi_mha_bwd_bf16 if CONV_BACKEND["operands"] == "bf16" else lib.evmi_mha_bwd_f32
    _chk(fn(qkv.data_ptr(), lens32.data_ptr(), out.data_ptr(), dout.data_ptr(), lse.data_ptr(), dsum.data_ptr(), dqkv.data_ptr(), B, T, D3 // 3, heads,
            float(p), int(seed), _lib.ptr(SEED_BASE[0]), _s(qkv)), "evmi_mha_bwd")
    return dqkv


# ---- alignment learning, training side (csrc/align_train_ops.hip) -------------------------------------------------------
def forward_sum_loss_and_grad(logprob, text_lens32, mel_lens32, weight, blank_logprob=-1.0):
    """(weight * mean over the batch of the per-item CTC forward-sum losses [device scalar], d that / d logprob [B, T, L])."""
    lib = _lib.load()
    B, T, L = logprob.shape
    n = lib.evmi_forward_sum_grad_f32_ws_elems(B, T, L)
    ws = WS.get("ctc", n, logprob.device)
    per_item = torch.empty(B, device=logprob.device, dtype=torch.float32)
    grad = torch.empty_like(logprob)
    _chk(lib.evmi_forward_sum_grad_f32(logprob.data_ptr(), text_lens32.data_ptr(), mel_lens32.data_ptr(), per_item.data_ptr(), grad.data_ptr(),
                                       ws.data_ptr(), n, B, T, L, blank_logprob, weight, _s(logprob)), "evmi_forward_sum_grad_f32")
    loss = torch.empty(1, device=logprob.device, dtype=torch.float32)
    scalar_reduce(2, per_item, None, loss, scale=weight / B)
    return loss, grad


def align_attention_fwd(q, k, prior, text_lens32, temperature):
    """q [A, B, T], k [A, B, L] (projected mel / text), prior [B, T, L] float64 or None -> (soft, logprob) [B, T, L]."""
    A, B, T = q.shape
    L = k.shape[2]
    soft = torch.empty(B, T, L, device=q.device, dtype=torch.float32)
    logprob = torch.empty_like(soft)
    _chk(_lib.load().evmi_align_attention_f32(q.data_ptr(), k.data_ptr(), _lib.ptr(prior), text_lens32.data_ptr(), soft.data_ptr(), logprob.data_ptr(),
                                              A, B, T, L, temperature, _s(q)), "evmi_align_attention_f32")
    return soft, logprob


def align_attention_bwd(q, k, soft, logprob, prior, hard, dlogprob, text_lens32, temperature, bin_scale, bin_count=None):
    """-> (dq [A, B, T], dk [A, B, L]); ``bin_count``: device scalar the binarisation weight is divided by (the frame count)."""
    lib = _lib.load()
    A, B, T = q.shape
    L = k.shape[2]
    dev = q.device
    da = torch.empty(B, T, L, device=dev, dtype=torch.float32)
    rs = torch.empty(B, T, device=dev, dtype=torch.float32)
    cs = torch.empty(B, L, device=dev, dtype=torch.float32)
    _chk(lib.evmi_align_attention_bwd_f32(soft.data_ptr(), logprob.data_ptr(), _lib.ptr(prior), _lib.ptr(hard), _lib.ptr(dlogprob), text_lens32.data_ptr(),
                                          da.data_ptr(), rs.data_ptr(), cs.data_ptr(), B, T, L, bin_scale, _lib.ptr(bin_count), _s(q)), "evmi_align_attention_bwd_f32")
    dq = torch.empty_like(q)
    dk = torch.empty_like(k)
    gemm_groups(k, da, dq, B, A, T, L, B * L, L, B * T, L, T * L, T, tb=True)  # K_b . da_b^T
    gemm_groups(q, da, dk, B, A, L, T, B * T, L, B * L, T, T * L, L)           # Q_b . da_b
    _chk(lib.evmi_align_qk_grad_f32(q.data_ptr(), rs.data_ptr(), dq.data_ptr(), A, B * T, -2.0 * temperature, _s(q)), "evmi_align_qk_grad_f32")
    _chk(lib.evmi_align_qk_grad_f32(k.data_ptr(), cs.data_ptr(), dk.data_ptr(), A, B * L, -2.0 * temperature, _s(q)), "evmi_align_qk_grad_f32")
    return dq, dk
