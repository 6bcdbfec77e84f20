"""One discriminator of the GAN step as a chain on flat packed bf16 tensors (precision="bf16").

Upstream's DiscriminatorP / DiscriminatorS (jik876 hifi-gan models.py; the reference reaches them through
``hfgl.model.HiFiGAN.training_step``) are ``conv -> leaky_relu`` chains whose every activation is read three
times (next convolution, feature matching, its own backward).  Op by op on the channel-major fp32 path each of those reads went
through a re-layout pass (``prep_pk`` / ``pack2`` / ``pad_x`` / ``pad_dy``: a third of the step's HBM bytes and ~500 launches per
step).  Here an activation exists ONCE, as the flat packed bf16 tensor (include/evmi.h, "Discriminator chains") the
matrix-core kernels load directly:

    A_1 = first(audio)                          evmi_disc_first_fwd      (one input channel: direct kernel, reads the waveform itself)
    A_{i+1} = lrelu(conv_i(A_i))  i = 1..L-1    evmi_conv_pkflat_fwd     (epilogue writes the consumer's packed layout)
    logits = post(A_L)                          evmi_disc_post_fwd
    G_L = (post^T dlogits [+ fm]) * lrelu'(A_L) evmi_disc_post_dgrad
    G_i = (conv_i^T G_{i+1} [+ fm]) * lrelu'(A_i)   evmi_conv_pkflat_dgrad   (mask = A_i itself; fm = the real waveform's A_i)
    dw_i += wgrad(A_i, G_{i+1})                 evmi_conv_pkflat_wgrad   (reads the same two tensors)
    db_i += rowsum(G_{i+1})                     evmi_pkflat_rowsum       (all layers of the chain in one launch pair)

Geometry: the items of a tensor lie end to end with a zero gap behind each -- the next convolution's padding -- so every kernel
runs over ONE long row.  A convolution with stride s computes output items ``Tc`` units apart from input items ``s * Tc`` apart and
stores them at the pitch its consumer wants; the gradient tensors take the transposed pitches.  ``_geometry`` derives all of it.
"""

from __future__ import annotations

import ctypes as C
import math

import torch

from .. import _lib
from . import autograd as ag
from . import ops

FRONT, TAIL = 64, 2048  # zero guard units in front of / behind every row


class PF:
    """A flat packed tensor: C channels, n_items items T units apart, `valid` data units per item (zeroed once; kernels write data only)."""

    __slots__ = ("C", "n_items", "T", "valid", "plane", "buf", "ptr")

    def __init__(self, C_, n_items, T, valid, device):
        self.C, self.n_items, self.T, self.valid = C_, n_items, T, valid
        self.plane = (FRONT + n_items * T + TAIL + 63) // 64 * 64
        self.buf = torch.zeros((C_ // 8) * self.plane * 4, device=device, dtype=torch.float32)  # 16-byte units as 4 floats
        self.ptr = self.buf.data_ptr() + FRONT * 16

    @property
    def units(self) -> int:
        return (self.C // 8) * self.plane


def _conv_len(t, k, s, p, d=1):
    return (t + 2 * p - d * (k - 1) - 1) // s + 1


class _Cfg:
    """Everything static about one (chain, item count, length, role): tensor geometry, buffers, per-call workspaces."""

    def __init__(self, chain, n_audio, t_audio, with_grad, device, grad_audio=None):
        lib = _lib.load()
        self.n_audio, self.t_audio = n_audio, t_audio
        p = chain.period
        self.n_items = n_audio * p
        # items gradients flow into: all of them, or the LAST grad_audio waveforms (generator step: real and generated waveforms run
        # forward as one batch -- half the launches, better filled -- and only the generated half runs backward)
        self.n_items_g = (n_audio if grad_audio is None else grad_audio) * p
        self.off_g = self.n_items - self.n_items_g
        self.H = (t_audio + p - 1) // p
        convs = chain.convs
        L = len(convs)
        self.lens = [self.H]
        for c in convs:
            self.lens.append(_conv_len(self.lens[-1], c.k, c.stride, c.pad, c.dil))
        # compute pitch of layer i (1..L-1): items of its output Tc apart, of its input s * Tc apart
        self.Tc = [0] * L
        for i in range(1, L):
            c = convs[i]
            right = (self.lens[i + 1] - 1) * c.stride + (c.k - 1) * c.dil - c.pad - (self.lens[i] - 1)
            need_gap = max(c.pad, right, 0)
            gdy = -(-max(0, (c.k - 1) * c.dil - c.pad) // c.stride)
            self.Tc[i] = max(self.lens[i + 1] + gdy, -(-(self.lens[i] + need_gap) // c.stride))
        post = chain.conv_post
        # activation A_i (i = 1..L): pitch dictated by its consumer
        T_A = [0] * (L + 1)
        for i in range(1, L):
            T_A[i] = convs[i].stride * self.Tc[i]
        T_A[L] = max(self.Tc[L - 1] if L > 1 else 0, self.lens[L] + max(post.pad, post.k - 1 - post.pad))
        self.A = [None] + [PF(convs[i - 1].cout, self.n_items, T_A[i], self.lens[i], device) for i in range(1, L + 1)]
        # gradient G_i (shape of A_i): pitch = compute pitch of the layer that produced A_i (its dgrad / wgrad read it flat)
        self.G = [None] * (L + 1)
        if with_grad:
            for i in range(1, L + 1):
                T_G = self.Tc[i - 1] if i >= 2 else T_A[1]
                self.G[i] = PF(convs[i - 1].cout, self.n_items_g, T_G, self.lens[i], device)
        # per-call workspaces: the call's static K-block offset table and split counters (written once, here) + its split partial tiles
        self.ws_f, self.ws_d = [None] * L, [None] * L
        st = _lib.current_stream_ptr(device)
        for i in range(1, L):
            c = convs[i]
            for mode, T, dst in ((0, T_A[i], self.ws_f), (1, self.Tc[i], self.ws_d)):
                if mode == 1 and not with_grad:
                    continue
                shape = (mode, self.n_items if mode == 0 else self.n_items_g, T, c.cin, c.cout, c.k, c.stride, c.pad, c.dil, c.groups)
                n = lib.evmi_conv_pkflat_ws_elems(*shape)
                if n <= 0:
                    raise RuntimeError(f"disc chain: layer {c.name} (direction {mode}) not taken by the flat packed kernel")
                dst[i] = torch.empty(n, device=device, dtype=torch.float32)
                _lib.check(lib.evmi_conv_pkflat_tab(*shape, dst[i].data_ptr(), n, st), "evmi_conv_pkflat_tab")
        self.logits_n = _conv_len(self.lens[L], post.k, post.stride, post.pad, post.dil)
        assert post.stride == 1 and self.logits_n == self.lens[L], "the logit layer is a 'same' convolution"
        # feature-matching scales: 2 / numel of every feature map (upstream feature_loss: mean |.| per map, times 2)
        self.fm_scale = [0.0] + [2.0 / (convs[i - 1].cout * self.n_items_g * self.lens[i]) for i in range(1, L + 1)]


class DiscChain:
    """Host side of one discriminator's chain.  ``supported(d)``: all shapes are taken by the flat packed kernels."""

    def __init__(self, disc, period: int, device):
        self.disc, self.period, self.device = disc, period, torch.device(device)
        self.convs, self.conv_post = disc.convs, disc.conv_post
        self._cfgs: dict = {}
        self._pinned: set = set()
        c0 = self.convs[0]
        ok = (c0.cin == 1 and c0.groups == 1 and c0.k <= 16 and c0.cout % 8 == 0 and c0.dil == 1 and self.conv_post.cout == 1
              and self.conv_post.k <= 8 and self.conv_post.groups == 1 and self.conv_post.stride == 1 and self.conv_post.dil == 1)
        for c in self.convs[1:]:
            ok = ok and (c.cin // c.groups) % 8 == 0 and (c.cout // c.groups) % 8 == 0 and c.k >= c.stride and not c.transposed
        self.ok = bool(ok)
        self.epoch = 0
        self._frag_bufs: dict = {}   # (layer, direction, slot) -> fragment buffer
        self._frag_map: dict = {}    # (layer, direction, weight pointer) -> (epoch, fragment buffer)

    # ---- weight fragments: per (layer, direction, weights), shared by every call of the layer while the weights stand ----------
    # (``self.epoch``: bumped by the chain's trainer whenever its discriminator's weights change (HiFiGANTrainer._materialize): older
    # fragments are stale.  Per chain: a second trainer in the process must not age this one's fragments.)

    def _frag_buf(self, i, mode, slot):
        key = (i, mode, slot)
        buf = self._frag_bufs.get(key)
        if buf is None:
            if torch.cuda.is_current_stream_capturing():
                raise RuntimeError("disc chain: fragment buffers would have to be created during graph capture: warm the step up eagerly first")
            c = self.convs[i]
            n = _lib.load().evmi_conv_pkflat_frag_elems(mode, c.cin, c.cout, c.k, c.stride, c.groups)
            buf = self._frag_bufs[key] = torch.empty(n, device=self.device, dtype=torch.float32)
        return buf

    def _call_weights(self, i):
        """The effective weights of the coming forward calls of layer i, in call order."""
        c = self.convs[i]
        if hasattr(c, "_ready"):  # spectral norm: one weight tensor per prepared call
            return [r[0] for r in c._ready]
        if c._w is None:
            c.materialize()
        return [c._w]

    def frag_jobs(self, dgrad_slots=(True,)):
        """(direction, layer, weights, buffer) of every fragment set the coming phase needs; ``dgrad_slots[s]``: call s of a layer
        also runs its input gradient (the last entry stands for further calls)."""
        jobs = []
        for i in range(1, len(self.convs)):
            for s, w in enumerate(self._call_weights(i)):
                for mode in (0, 1):
                    if mode == 1 and not dgrad_slots[min(s, len(dgrad_slots) - 1)]:
                        continue
                    wf = self._frag_buf(i, mode, s)
                    self._frag_map[(i, mode, w.data_ptr())] = (self.epoch, wf)
                    jobs.append((mode, self.convs[i], w, wf))
        return jobs

    def _frag(self, i, mode, w, role):
        """The fragments of `w` for layer i: prepared by the trainer at the start of the phase, or made here (direct calls)."""
        e = self._frag_map.get((i, mode, w.data_ptr()))
        if e is not None and e[0] == self.epoch:
            return e[1]
        wf = self._frag_buf(i, mode, ("local", role))
        launch_fragments([(mode, self.convs[i], w, wf)], self.device)
        self._frag_map[(i, mode, w.data_ptr())] = (self.epoch, wf)
        return wf

    MAX_CFGS = 8  # buffer sets kept (least recently used out first; a set a graph capture has touched stays)

    def cfg(self, n_audio, t_audio, role, with_grad, grad_audio=None) -> _Cfg:
        key = (n_audio, t_audio, role, with_grad, grad_audio)
        capturing = torch.cuda.is_current_stream_capturing()
        c = self._cfgs.get(key)
        if c is None:
            if capturing:
                raise RuntimeError("disc chain: buffers would have to be created during graph capture: warm the step up eagerly first")
            c = _Cfg(self, n_audio, t_audio, with_grad, self.device, grad_audio)
            loose = [k for k in self._cfgs if k not in self._pinned]
            while len(loose) >= self.MAX_CFGS:
                del self._cfgs[loose.pop(0)]
        else:
            del self._cfgs[key]  # (re-inserted below: most recently used last)
        self._cfgs[key] = c
        if capturing:
            self._pinned.add(key)
        return c

    # ---- forward ------------------------------------------------------------------------------------------------------------------
    def forward(self, tape: ag.Tape, audio: ag.Var, training=True, role="pair", grad_from=None):
        """audio.data [1, n_audio, t_audio] fp32 -> (logits Var [1, n_items, n], ChainFmaps).  ``grad_from`` = B (generator step, frozen
        layers): the batch is [real (B waveforms) | generated]; the forward runs over all of it, the backward over the generated half
        only, its feature-matching reference being the real half of the same tensors."""
        lib = _lib.load()
        x = audio.data
        _, n_audio, t_audio = x.shape
        frozen = all(layer.frozen for layer in self.convs)
        with_grad = audio.needs_grad or not frozen
        if grad_from is not None and not (frozen and n_audio == 2 * grad_from and grad_from > 0):
            # (feature matching pairs generated item i with real item i row by row: only an even split pairs the right items)
            raise ValueError("disc chain: a split batch is the generator step's [real | generated] (frozen layers, n_audio == 2 * grad_from)")
        cfg = self.cfg(n_audio, t_audio, role, with_grad, None if grad_from is None else n_audio - grad_from)
        convs, post, L = self.convs, self.conv_post, len(self.convs)
        st = ops._s(x)
        eff = [c.effective(training) + (c.call_db_sink(),) for c in convs]  # (w, dw sink, db sink) per call, in layer order
        w_post, dw_post = post.effective(training)
        db_post = post.call_db_sink()
        wf_f = [None] + [self._frag(i, 0, eff[i][0], role) for i in range(1, L)]
        wf_d = [None] + [self._frag(i, 1, eff[i][0], role) if with_grad else None for i in range(1, L)]
        c0 = convs[0]
        A = cfg.A
        _lib.check(lib.evmi_disc_first_fwd(x.data_ptr(), n_audio, t_audio, self.period, eff[0][0].data_ptr(), c0.bias_data().data_ptr(), A[1].ptr, A[1].plane,
                                           A[1].T, cfg.lens[1], c0.cout, c0.k, c0.stride, c0.pad, 0.1, st), "evmi_disc_first_fwd")
        ops._count_conv(cfg.n_items, cfg.lens[1], c0.cout, 1, c0.k)
        for i in range(1, L):
            c = convs[i]
            _lib.check(lib.evmi_conv_pkflat_fwd(A[i].ptr, A[i].plane, wf_f[i].data_ptr(), c.bias_data().data_ptr(), A[i + 1].ptr, A[i + 1].plane,
                                                cfg.ws_f[i].data_ptr(), cfg.ws_f[i].numel(), cfg.n_items, A[i].T, c.cin, c.cout, c.k, c.stride, c.pad, c.dil,
                                                c.groups, cfg.lens[i + 1], A[i + 1].T, ops.ACT_LRELU, 0.1, st), "evmi_conv_pkflat_fwd")
            ops._count_conv(cfg.n_items, cfg.lens[i + 1], c.cout, c.cin // c.groups, c.k)
        n = cfg.logits_n
        logits = torch.empty(1, cfg.n_items, n, device=x.device, dtype=torch.float32)
        ws = ops.WS.get("dc_post", lib.evmi_disc_post_fwd_ws_elems(cfg.n_items, n, post.cin), x.device)
        _lib.check(lib.evmi_disc_post_fwd(A[L].ptr, A[L].plane, A[L].T, cfg.n_items, n, w_post.data_ptr(), post.bias_data().data_ptr(), logits.data_ptr(),
                                          ws.data_ptr(), ws.numel(), post.cin, post.k, post.pad, st), "evmi_disc_post_fwd")
        ops._count_conv(cfg.n_items, n, 1, post.cin, post.k)
        ag._ACTIVATION_ELEMS[0] += sum(a.C * cfg.n_items * a.valid for a in A[1:]) // 2  # (bf16: half an fp32 element each)
        out = ag.Var(logits)
        fm = ChainFmaps(self, cfg, out)
        fm.split = grad_from is not None

        def bwd():
            if out.grad is None:
                return
            self._backward(cfg, fm, audio, out.grad, eff, (w_post, dw_post, db_post), frozen, wf_d)

        tape.record(bwd)
        return out, fm

    # ---- backward -----------------------------------------------------------------------------------------------------------------
    def _backward(self, cfg: _Cfg, fm: "ChainFmaps", audio: ag.Var, dlogits, eff, post_eff, frozen, wf_d):
        lib = _lib.load()
        convs, post, L = self.convs, self.conv_post, len(self.convs)
        A, G = cfg.A, cfg.G
        dev = dlogits.device
        st = ops._s(dlogits)
        ref = fm.ref  # the real waveform's activations (generator step): feature-matching gradients ride in the epilogues
        w_post, dw_post, db_post = post_eff
        n = cfg.logits_n

        ng, off = cfg.n_items_g, cfg.off_g  # items that run backward: the last ng of the batch
        split = fm.split and ref is not None

        def fm_args(i):
            mask = A[i].ptr + off * A[i].T * 16  # the activations of the items that run backward
            if ref is None:
                return (mask, 0, A[i].plane, A[i].T, 0.1, 0.0)
            return (mask, A[i].ptr if split else ref.A[i].ptr, A[i].plane, A[i].T, 0.1, cfg.fm_scale[i])

        if ref is not None and not split:
            assert all(ref.A[i].T == A[i].T and ref.A[i].plane == A[i].plane for i in range(1, L + 1)), "feature-matching pair: different geometry"
        dl_ptr = dlogits.data_ptr() + off * n * 4
        _lib.check(lib.evmi_disc_post_dgrad(dl_ptr, w_post.data_ptr(), G[L].ptr, G[L].plane, G[L].T, ng, n, post.cin, post.k, post.pad,
                                            *fm_args(L), st), "evmi_disc_post_dgrad")
        ops._count_conv(ng, n, 1, post.cin, post.k)
        if not frozen:
            ws = ops.WS.get("dc_postw", lib.evmi_disc_post_wgrad_ws_elems(cfg.n_items, n, post.cin, post.k), dev)
            _lib.check(lib.evmi_disc_post_wgrad(A[L].ptr, A[L].plane, A[L].T, cfg.n_items, n, dlogits.data_ptr(), dw_post.data_ptr(), ws.data_ptr(), ws.numel(),
                                                post.cin, post.k, post.pad, 1, st), "evmi_disc_post_wgrad")
            ops.scalar_reduce(2, dlogits, None, db_post, accumulate=True)
            ops._count_conv(cfg.n_items, n, 1, post.cin, post.k)
        for i in range(L - 1, 0, -1):
            c = convs[i]
            need_dx = i > 1 or audio.needs_grad or not frozen  # G_1 feeds the first layer's weight gradient / the waveform's gradient
            if need_dx:
                _lib.check(lib.evmi_conv_pkflat_dgrad(G[i + 1].ptr, G[i + 1].plane, wf_d[i].data_ptr(), G[i].ptr, G[i].plane, cfg.ws_d[i].data_ptr(),
                                                      cfg.ws_d[i].numel(), ng, G[i + 1].T, c.cin, c.cout, c.k, c.stride, c.pad, c.dil, c.groups,
                                                      cfg.lens[i], G[i].T, *fm_args(i), st), "evmi_conv_pkflat_dgrad")
                ops._count_conv(ng, cfg.lens[i + 1], c.cout, c.cin // c.groups, c.k)
            if not frozen:
                nws = lib.evmi_conv_pkflat_wgrad_ws_elems(cfg.n_items, G[i + 1].T, c.cin, c.cout, c.k, c.stride, c.dil, c.groups)
                if nws < 0:
                    raise RuntimeError(f"disc chain: weight gradient of {c.name} not taken by the flat packed kernel")
                ws = ops.WS.get("dc_wg", nws, dev)
                _lib.check(lib.evmi_conv_pkflat_wgrad(A[i].ptr, A[i].plane, G[i + 1].ptr, G[i + 1].plane, eff[i][1].data_ptr(), ws.data_ptr(), ws.numel(),
                                                      cfg.n_items, G[i + 1].T, c.cin, c.cout, c.k, c.stride, c.pad, c.dil, c.groups, 1, st), "evmi_conv_pkflat_wgrad")
                ops._count_conv(cfg.n_items, cfg.lens[i + 1], c.cout, c.cin // c.groups, c.k)
        c0 = convs[0]
        x = audio.data
        if not frozen:
            nws = lib.evmi_disc_first_wgrad_ws_elems(cfg.n_items, cfg.lens[1], c0.cout, c0.k)
            ws = ops.WS.get("dc_w0", nws, dev)
            _lib.check(lib.evmi_disc_first_wgrad(x.data_ptr(), cfg.n_audio, cfg.t_audio, self.period, G[1].ptr, G[1].plane, G[1].T, cfg.lens[1],
                                                 eff[0][1].data_ptr(), eff[0][2].data_ptr(), ws.data_ptr(), ws.numel(), c0.cout, c0.k, c0.stride, c0.pad, 1, st),
                       "evmi_disc_first_wgrad")
            ops._count_conv(cfg.n_items, cfg.lens[1], c0.cout, 1, c0.k)
            # bias gradients of the matrix-core layers: row sums of G_2 .. G_L, one launch pair
            rows = (_lib.PkFlatRows * (L - 1))()
            for i in range(1, L):
                r = rows[i - 1]
                r.dy, r.plane, r.units, r.C, r.db = G[i + 1].ptr, G[i + 1].plane, cfg.n_items * G[i + 1].T, convs[i].cout, eff[i][2].data_ptr()
            ws = ops.WS.get("dc_rows", lib.evmi_pkflat_rowsum_ws_elems(L - 1, rows), dev)
            _lib.check(lib.evmi_pkflat_rowsum(L - 1, rows, ws.data_ptr(), ws.numel(), st), "evmi_pkflat_rowsum")
        if audio.needs_grad:
            # (a split batch: zeros for the waveforms that take no gradient)
            dxv = ops.zeros(1, cfg.n_items, cfg.H, device=dev) if off else torch.empty(1, cfg.n_items, cfg.H, device=dev, dtype=torch.float32)
            _lib.check(lib.evmi_disc_first_dgrad(G[1].ptr, G[1].plane, G[1].T, cfg.lens[1], eff[0][0].data_ptr(), dxv.data_ptr() + off * cfg.H * 4, ng, cfg.H,
                                                 c0.cout, c0.k, c0.stride, c0.pad, st), "evmi_disc_first_dgrad")
            ops._count_conv(ng, cfg.lens[1], c0.cout, 1, c0.k)
            audio.accumulate(dxv if self.period == 1 else ops.period_view_bwd(dxv, cfg.n_audio, cfg.t_audio, self.period))


def launch_fragments(jobs, device) -> None:
    """jobs: (direction, layer, weights, buffer) as ``DiscChain.frag_jobs`` returns them -- any number of chains' jobs together; the
    library packs 32 of them into a launch."""
    if not jobs:
        return
    arr = (_lib.PkFlatJob * len(jobs))()
    for j, (mode, c, w, wf) in zip(arr, jobs):
        j.mode, j.c_in, j.c_out, j.k, j.stride, j.groups = mode, c.cin, c.cout, c.k, c.stride, c.groups
        j.w, j.wf, j.wf_elems = w.data_ptr(), wf.data_ptr(), wf.numel()
    _lib.check(_lib.load().evmi_conv_pkflat_fragments(len(jobs), arr, _lib.current_stream_ptr(device)), "evmi_conv_pkflat_fragments")


class ChainFmaps:
    """The feature maps of one chain call: the packed activations A_1 .. A_L (owned by the chain's buffers) and the logits Var."""

    def __init__(self, chain: DiscChain, cfg: _Cfg, logits: ag.Var):
        self.chain, self.cfg, self.logits = chain, cfg, logits
        self.A = cfg.A
        self.ref = None  # set by feature_matching(): the real-waveform call whose activations the backward compares with
        self.split = False  # the batch is [real | generated]: the reference is the first half of this call's own tensors

    def feature_matching(self, real: "ChainFmaps", slot: torch.Tensor) -> None:
        """slot[0] += 2 * sum over the feature maps of mean |fake - real| (upstream feature_loss); the gradients are produced by
        this call's backward (packed maps: in the input-gradient epilogues; logits: here, as on the op-by-op path)."""
        lib = _lib.load()
        cfg = self.cfg
        L = len(self.chain.convs)
        pairs = (_lib.PkFlatPair * L)()
        for i in range(1, L + 1):
            p = pairs[i - 1]
            if self.split:  # the generated half against the real half of the same tensor: row by row
                a = self.A[i]
                p.a, p.b, p.rows, p.plane, p.units = a.ptr + cfg.off_g * a.T * 16, a.ptr, a.C // 8, a.plane, cfg.n_items_g * a.T
            else:
                p.a, p.b, p.rows, p.plane, p.units = self.A[i].buf.data_ptr(), real.A[i].buf.data_ptr(), 1, 0, self.A[i].units
            p.scale = cfg.fm_scale[i]
        dev = slot.device
        ws = ops.WS.get("dc_fm", lib.evmi_pkflat_absdiff_ws_elems(L), dev)
        _lib.check(lib.evmi_pkflat_absdiff(L, pairs, slot.data_ptr(), ws.data_ptr(), ws.numel(), ops._s(slot)), "evmi_pkflat_absdiff")
        if self.split:
            h = cfg.off_g
            fg_d, fr_d = self.logits.data[:, h:], self.logits.data[:, :h]
            n = fg_d.numel()
            ops.scalar_reduce(0, fg_d.contiguous(), fr_d.contiguous(), slot, scale=2.0 / n, accumulate=True)
            g = self.logits.grad[:, h:]
            ops.axpby(1.0, g, 1.0, ops.elementwise(ops.EW_SIGN_DIFF, fg_d.contiguous(), fr_d.contiguous(), p0=2.0 / n), out=g)
            self.ref = self
            return
        fg, fr = self.logits, real.logits
        n = fg.data.numel()
        ops.scalar_reduce(0, fg.data, fr.data, slot, scale=2.0 / n, accumulate=True)
        fg.accumulate(ops.elementwise(ops.EW_SIGN_DIFF, fg.data, fr.data, p0=2.0 / n))
        self.ref = real
