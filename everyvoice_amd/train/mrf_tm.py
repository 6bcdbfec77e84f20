"""The generator's residual stacks (upstream MRF: per stage three ResBlock1 branches of three (conv1, conv2) pairs, averaged) trained
in the INFERENCE layout -- time-major bf16 ``[B][Tp][C]`` -- on the inference convolution kernels (csrc/train_tm.hip):

    forward   t = lrelu(conv1(lrelu(x)) + b1)   y = x + conv2(t) + b2              two conv_tc launches per pair, nothing between them
    backward  dc1 = convT2(dy) * lrelu'(t)      dx = convT1(dc1) * lrelu'(x) + dy   two conv_tc launches (mask / residual in the epilogue)
              dw2 += wgrad(t, dy)   dw1 += wgrad(lrelu(x), dc1)   db = column sums  the TM weight-gradient kernel, no packed copies

against ``precision="bf16"``'s channel-major path (fp32 ``[C][B][T]`` tensors re-packed in front of every convolution:
conv_cbt_bf16_pk.hip) -- which remains the path of every other layer and of shapes the time-major kernels do not take.  Storage
differs: activations AND gradients of a stack live in bf16 here (fp32 there, rounded to bf16 on their way into the matrix cores
either way); accumulation is fp32 in both.  Layout changes happen once per stage and direction (fp32 channel-major <-> TM).

Reference: the module is hfgl.model's generator (absent submodule; upstream jik876 hifi-gan models.py ResBlock1 / Generator.forward:
``xs += resblocks[i*num_kernels+j](x)``, ``x = xs / num_kernels``); this file is host orchestration only, every arithmetic step is a
libevmi_hip call.
"""

from __future__ import annotations

import torch

from .. import _lib
from . import autograd as ag
from . import ops

GUARD_FRONT, GUARD_BACK, PAD_ROWS = 64, 320, 32


class TMBuf:
    """bf16 [B][Tp][C] with zero rows around every item's T valid rows and zero guard rows around the whole (allocated zeroed
    once; kernels write valid rows only)."""

    def __init__(self, C: int, B: int, T: int, device):
        self.C, self.B, self.T = C, B, T
        self.PL = PAD_ROWS
        self.Tp = (T + 2 * PAD_ROWS + 15) // 16 * 16
        self.rows = B * self.Tp
        n = (GUARD_FRONT + self.rows + GUARD_BACK) * C
        self.store = torch.zeros(n // 2, device=device, dtype=torch.float32)  # (two bf16 per word; 16-byte aligned base)
        self.ptr = self.store.data_ptr() + GUARD_FRONT * C * 2
        self.numel_body = self.rows * C

    def valid(self) -> torch.Tensor:
        """[B, T, C] bf16 view of the valid rows (tests)."""
        body = self.store.view(torch.bfloat16)[GUARD_FRONT * self.C: GUARD_FRONT * self.C + self.rows * self.C].view(self.B, self.Tp, self.C)
        return body[:, self.PL: self.PL + self.T]


def _s(device):
    return _lib.current_stream_ptr(device)


def stage_supported(C: int, kernel_sizes, dilations) -> bool:
    lib = _lib.load()
    if C % 8 or C < 32 or C > 256:
        return False
    for k, dils in zip(kernel_sizes, dilations):
        for d in list(dils) + [1]:
            if not lib.evmi_conv_tc_supported(C, C, k, d):
                return False
            if lib.evmi_conv1d_wgrad_tm_bf16_ws_elems(1 << 16, C, C, k, d) < 0:
                return False
    return True


class MRFStageTM:
    """One stage's MRF.  ``branches``: per branch the list of (conv1, conv2) WNConv pairs; persistent TM buffers per
    (batch, length) so that a captured HIP graph of the step sees static addresses."""

    def __init__(self, C: int, branch_pairs, slope: float, device):
        self.C, self.pairs, self.slope, self.device = C, branch_pairs, slope, torch.device(device)
        self._bufs: dict = {}
        self._shape_lru: list = []
        self._pinned: set = set()
        self._table = self._table_base = self._arena = None
        self.zero_bias = torch.zeros(max(C, 8), device=self.device, dtype=torch.float32)

    # ---- storage ---------------------------------------------------------------------------------------------------
    MAX_SHAPES = 3  # (batch, length) buffer sets kept: the training segment shape + validation / synthesis lengths in turn

    def buf(self, key, B, T) -> TMBuf:
        """Buffers are persistent per (batch, length) -- a captured graph needs static addresses -- but only for the MAX_SHAPES most
        recently used shapes: ``HiFiGANTrainer.generate`` on utterances of many lengths would otherwise keep ~19 buffers per stage
        for every length it ever saw.  A shape a graph capture has touched is pinned."""
        shape = (B, T)
        capturing = torch.cuda.is_current_stream_capturing()
        if capturing:
            self._pinned.add(shape)  # a captured graph holds these addresses: replays do not pass through here, so the shape stays for good
        if shape in self._shape_lru:
            if self._shape_lru[-1] != shape:
                self._shape_lru.remove(shape)
                self._shape_lru.append(shape)
        else:
            if capturing:
                raise RuntimeError("MRFStageTM: a time-major buffer would have to be allocated during graph capture: warm the step up eagerly first")
            self._shape_lru.append(shape)
            loose = [sh for sh in self._shape_lru if sh not in self._pinned]
            while len(loose) > self.MAX_SHAPES:
                old = loose.pop(0)
                self._shape_lru.remove(old)
                for k in [k for k in self._bufs if (k[1], k[2]) == old]:
                    del self._bufs[k]
        k = (key, B, T)
        b = self._bufs.get(k)
        if b is None:
            if torch.cuda.is_current_stream_capturing():
                raise RuntimeError("MRFStageTM: a time-major buffer would have to be allocated during graph capture: warm the step up eagerly first")
            b = self._bufs[k] = TMBuf(self.C, B, T, self.device)
        return b

    def _relayout_all(self):
        """Every convolution's effective weight into its kernel's tile layout (bf16), forward form and input-gradient form
        (channels swapped, taps reversed), in ONE launch: weights change every step, the table that drives the kernel does not.
        The effective weights of an optimiser's weight-normed layers live in one flat buffer (layers.WNBatch): offsets from
        its lowest address."""
        import ctypes as C

        lib = _lib.load()
        layers = [c for br in self.pairs for pr in br for c in pr]
        ws = [c.effective(True)[0] for c in layers]
        base = min(w.data_ptr() for w in ws)
        if self._table is None or self._table_base != base:
            rows, off = [], 0
            self._w_off = {}
            for c, w in zip(layers, ws):
                bm, kc, wl = C.c_int(), C.c_int(), C.c_int()
                _lib.check(lib.evmi_conv_tc_tile_layout(self.C, self.C, c.k, c.dil, C.byref(bm), C.byref(kc), C.byref(wl)), "evmi_conv_tc_tile_layout")
                for mode in (0, 1):
                    assert (w.data_ptr() - base) % 4 == 0
                    rows.append([(w.data_ptr() - base) // 4, off, c.k, bm.value, kc.value, wl.value, mode, 0])
                    self._w_off[(id(c), bool(mode))] = off
                    off += self.C * self.C * c.k
            self._arena = torch.empty((off + 1) // 2, device=self.device, dtype=torch.float32)
            self._table = torch.tensor(rows, dtype=torch.int64).to(self.device)
            self._table_base, self._base_w = base, min(ws, key=lambda w: w.data_ptr())
            self._ks_max = max(c.k for c in layers)
        _lib.check(lib.evmi_conv_tc_relayout_batched_f32(base, self._arena.data_ptr(), self._table.data_ptr(), self._table.shape[0], self.C, self._ks_max,
                                                         _s(self.device)), "evmi_conv_tc_relayout_batched_f32")

    def _laid_ptr(self, layer, transpose: bool) -> int:
        return self._arena.data_ptr() + 2 * self._w_off[(id(layer), transpose)]

    # ---- kernels ---------------------------------------------------------------------------------------------------
    def _conv(self, x: TMBuf, w_laid, bias, out: TMBuf, k, dil, pre=1.0, post=1.0, res: TMBuf | None = None, mask: TMBuf | None = None, mask_slope=1.0):
        ops._count_conv(x.B, x.T, self.C, self.C, k)
        _lib.check(_lib.load().evmi_conv_tc_tm_bf16(x.ptr, w_laid, bias.data_ptr(), res.ptr if res is not None else 0,
                                                    mask.ptr if mask is not None else 0, out.ptr, x.B, x.T, x.Tp, x.PL, self.C, self.C, k, dil,
                                                    float(pre), float(post), float(mask_slope), 1.0, _s(self.device)), "evmi_conv_tc_tm_bf16")

    def _wgrad(self, x: TMBuf, dy: TMBuf, dw: torch.Tensor, k, dil):
        lib = _lib.load()
        ops._count_conv(x.B, x.T, self.C, self.C, k)
        n = lib.evmi_conv1d_wgrad_tm_bf16_ws_elems(x.rows, self.C, self.C, k, dil)
        ws = ops.WS.get("tm_wgrad", n, self.device)
        _lib.check(lib.evmi_conv1d_wgrad_tm_bf16(x.ptr, dy.ptr, dw.data_ptr(), ws.data_ptr(), n, x.rows, self.C, self.C, k, dil * (k - 1) // 2, dil, 1,
                                                 _s(self.device)), "evmi_conv1d_wgrad_tm_bf16")

    def _colsum(self, dy: TMBuf, db: torch.Tensor):
        lib = _lib.load()
        n = lib.evmi_tm_colsum_bf16_ws_elems(dy.rows, self.C)
        ws = ops.WS.get("tm_colsum", n, self.device)
        _lib.check(lib.evmi_tm_colsum_bf16(dy.ptr, db.data_ptr(), ws.data_ptr(), n, dy.rows, self.C, 1, _s(self.device)), "evmi_tm_colsum_bf16")

    def _colsum_batch(self, jobs):
        """jobs: (TMBuf, bias-gradient sink) of one shape: db += column sums, 24 tensors per launch pair."""
        import ctypes as C

        lib = _lib.load()
        for i in range(0, len(jobs), 24):
            part = jobs[i:i + 24]
            n = len(part)
            rows = part[0][0].rows
            dys = (C.c_void_p * n)(*[b.ptr for b, _ in part])
            dbs = (C.c_void_p * n)(*[d.data_ptr() for _, d in part])
            ne = n * lib.evmi_tm_colsum_bf16_ws_elems(rows, self.C)
            ws = ops.WS.get("tm_colsum_b", ne, self.device)
            _lib.check(lib.evmi_tm_colsum_batch_bf16(n, dys, dbs, ws.data_ptr(), ne, rows, self.C, 1, _s(self.device)), "evmi_tm_colsum_batch_bf16")

    def _lrelu(self, x: TMBuf, y: TMBuf):
        _lib.check(_lib.load().evmi_tm_lrelu_bf16(x.ptr, y.ptr, x.numel_body, self.slope, _s(self.device)), "evmi_tm_lrelu_bf16")

    def _to_tm(self, x_cbt: torch.Tensor, out: TMBuf, scale=1.0):
        _lib.check(_lib.load().evmi_cbt_f32_to_tm_bf16(x_cbt.data_ptr(), out.ptr, self.C, out.B, out.T, out.Tp, out.PL, 1.0, float(scale), _s(self.device)),
                   "evmi_cbt_f32_to_tm_bf16")

    def _to_cbt(self, parts, scale=1.0) -> torch.Tensor:
        a = parts[0]
        out = torch.empty(self.C, a.B, a.T, device=self.device, dtype=torch.float32)
        ptrs = [p.ptr for p in parts] + [0, 0]
        _lib.check(_lib.load().evmi_tm_bf16_to_cbt_f32(ptrs[0], ptrs[1], ptrs[2], out.data_ptr(), self.C, a.B, a.T, a.Tp, a.PL, float(scale), _s(self.device)),
                   "evmi_tm_bf16_to_cbt_f32")
        return out

    # ---- the op ----------------------------------------------------------------------------------------------------
    def apply(self, tape: ag.Tape, x: ag.Var, branches=None) -> ag.Var:
        """mean over the branches of ResBlock1(x): forward now, backward recorded on the tape."""
        C, B, T = x.data.shape
        assert C == self.C and len(self.pairs) <= 3
        nb = len(self.pairs)
        slope = self.slope
        u = self.buf("u", B, T)
        self._to_tm(x.data, u)
        self._relayout_all()
        outs = [None] * nb

        def fwd_branch(j):
            def go():
                cur = u
                for m, (c1, c2) in enumerate(self.pairs[j]):
                    t = self.buf(("t", j, m), B, T)
                    y = self.buf(("y", j, m), B, T)
                    self._conv(cur, self._laid_ptr(c1, False), c1.bias_data(), t, c1.k, c1.dil, pre=slope, post=slope)
                    self._conv(t, self._laid_ptr(c2, False), c2.bias_data(), y, c2.k, 1, res=cur)
                    ag._ACTIVATION_ELEMS[0] += t.numel_body  # (two bf16 tensors = one fp32 tensor's bytes in the bench's pricing)
                    cur = y
                outs[j] = cur
            return go

        run = branches.run if branches is not None else (lambda fns: [f() for f in fns])
        run([fwd_branch(j) for j in range(nb)])
        out = ag.Var(self._to_cbt(outs, 1.0 / nb))

        def bwd():
            if out.grad is None:
                return
            dy0 = self.buf("dy0", B, T)
            self._to_tm(out.grad, dy0, 1.0 / nb)
            dxs = [None] * nb
            bias_jobs = [[] for _ in range(nb)]  # (gradient tensor, bias-gradient sink) of every convolution: column sums of the whole
                                                 # stack in ONE launch pair behind the branches (18 tensors: 36 launches of 5-7 us on
                                                 # the branch chains as single calls); every pair keeps its own gradient buffers for it

            def bwd_branch(j):
                def go():
                    dcur = dy0
                    act = self.buf(("a", j), B, T)
                    for m in reversed(range(len(self.pairs[j]))):
                        c1, c2 = self.pairs[j][m]
                        cur = u if m == 0 else self.buf(("y", j, m - 1), B, T)
                        t = self.buf(("t", j, m), B, T)
                        _, dw1 = c1.effective(True)
                        _, dw2 = c2.effective(True)
                        bias_jobs[j].append((dcur, c2.call_db_sink()))
                        self._wgrad(t, dcur, dw2, c2.k, 1)
                        dc1 = self.buf(("dc1", j, m), B, T)
                        self._conv(dcur, self._laid_ptr(c2, True), self.zero_bias, dc1, c2.k, 1, mask=t, mask_slope=slope)
                        bias_jobs[j].append((dc1, c1.call_db_sink()))
                        self._lrelu(cur, act)
                        self._wgrad(act, dc1, dw1, c1.k, c1.dil)
                        dx = self.buf(("dx", j, m), B, T)
                        self._conv(dc1, self._laid_ptr(c1, True), self.zero_bias, dx, c1.k, c1.dil, mask=cur, mask_slope=slope, res=dcur)
                        dcur = dx
                    dxs[j] = dcur
                return go

            run([bwd_branch(j) for j in range(nb)])
            self._colsum_batch([job for jobs in bias_jobs for job in jobs])
            x.accumulate(self._to_cbt(dxs, 1.0))

        tape.record(bwd)
        return out
