#!/usr/bin/env python3
"""The flat packed convolution kernels (evmi_conv_pkflat_fwd / _dgrad / _wgrad) at the layer shapes of the GAN step's discriminators
(16 x 8192 samples: 32 items in the discriminator step), each call timed alone: us per call, TFLOP/s of the valid outputs, the
planner's tile / split.  usage: python tools/pkflat_bench.py [iters]     (EVMI_PK_TILE etc. pass through to the planner)"""
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch  # noqa: E402

from everyvoice_amd import _lib  # noqa: E402
from everyvoice_amd.train.disc_chain import PF, _conv_len  # noqa: E402

dev = torch.device("cuda:0")
lib = _lib.load()
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 20
st = _lib.current_stream_ptr(dev)

# (name, n_items, t_in, cin, cout, k, stride, pad, groups)
SHAPES = []
for p, n in ((2, 64), (3, 96), (11, 352)):
    H = (8192 + p - 1) // p
    l1 = _conv_len(H, 5, 3, 2)
    l2 = _conv_len(l1, 5, 3, 2)
    l3 = _conv_len(l2, 5, 3, 2)
    l4 = _conv_len(l3, 5, 3, 2)
    SHAPES += [(f"mpd{p}.1", n, l1, 32, 128, 5, 3, 2, 1), (f"mpd{p}.2", n, l2, 128, 512, 5, 3, 2, 1), (f"mpd{p}.3", n, l3, 512, 1024, 5, 3, 2, 1),
               (f"mpd{p}.4", n, l4, 1024, 1024, 5, 1, 2, 1)]
for s, T in ((0, 8192), (2, 2049)):
    SHAPES += [(f"msd{s}.1", 32, T, 128, 128, 41, 2, 20, 4), (f"msd{s}.2", 32, _conv_len(T, 41, 2, 20), 128, 256, 41, 2, 20, 16)]
    t3 = _conv_len(_conv_len(T, 41, 2, 20), 41, 2, 20)
    t4 = _conv_len(t3, 41, 4, 20)
    t5 = _conv_len(t4, 41, 4, 20)
    SHAPES += [(f"msd{s}.3", 32, t3, 256, 512, 41, 4, 20, 16), (f"msd{s}.4", 32, t4, 512, 1024, 41, 4, 20, 16),
               (f"msd{s}.5", 32, t5, 1024, 1024, 41, 1, 20, 16), (f"msd{s}.6", 32, t5, 1024, 1024, 5, 1, 2, 1)]


def timed(fn):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


print(f"{'layer':9s} {'items':>5s} {'t_in':>5s} {'cin':>5s} {'cout':>5s} {'k':>3s} {'s':>2s} {'g':>3s} | {'fwd us':>8s} {'TF/s':>6s} {'plan':>5s} | {'dgrad us':>8s} {'TF/s':>6s} {'plan':>5s} | {'wgrad us':>8s} {'TF/s':>6s} | frag us")
tot = [0.0, 0.0, 0.0, 0.0]
import os  # noqa: E402

ONLY = os.environ.get("PKFLAT_ONLY")  # e.g. "mpd2.4": one shape (for a profiler run)
for name, n, t_in, cin, cout, k, s, pad, g in SHAPES:
    if ONLY and name != ONLY:
        continue
    t_out = _conv_len(t_in, k, s, pad)
    right = (t_out - 1) * s + (k - 1) - pad - (t_in - 1)
    gdy = -(-max(0, k - 1 - pad) // s)
    Tc = max(t_out + gdy, -(-(t_in + max(pad, right, 0)) // s))
    X, Y = PF(cin, n, s * Tc, t_in, dev), PF(cout, n, Tc, t_out, dev)
    DX = PF(cin, n, s * Tc, t_in, dev)
    X.buf.normal_()
    Y.buf.normal_()
    w = torch.randn(cout, cin // g, k, device=dev) * 0.05
    b = torch.randn(cout, device=dev)
    dw = torch.zeros_like(w)
    shf, shd = (0, n, s * Tc, cin, cout, k, s, pad, 1, g), (1, n, Tc, cin, cout, k, s, pad, 1, g)
    nf, nd = lib.evmi_conv_pkflat_ws_elems(*shf), lib.evmi_conv_pkflat_ws_elems(*shd)
    nw = lib.evmi_conv_pkflat_wgrad_ws_elems(n, Tc, cin, cout, k, s, 1, g)
    wsf, wsd, wsw = (torch.empty(max(m, 4), device=dev) for m in (nf, nd, nw))
    _lib.check(lib.evmi_conv_pkflat_tab(*shf, wsf.data_ptr(), wsf.numel(), st), "tab")
    _lib.check(lib.evmi_conv_pkflat_tab(*shd, wsd.data_ptr(), wsd.numel(), st), "tab")
    wff, wfd = (torch.empty(lib.evmi_conv_pkflat_frag_elems(m, cin, cout, k, s, g), device=dev) for m in (0, 1))
    flop = 2.0 * n * t_out * cout * (cin // g) * k
    jobs = (_lib.PkFlatJob * 2)()
    for j, (mode, wf) in enumerate(((0, wff), (1, wfd))):
        jb = jobs[j]
        jb.mode, jb.c_in, jb.c_out, jb.k, jb.stride, jb.groups = mode, cin, cout, k, s, g
        jb.w, jb.wf, jb.wf_elems = w.data_ptr(), wf.data_ptr(), wf.numel()

    def prep():
        _lib.check(lib.evmi_conv_pkflat_fragments(2, jobs, st), "fragments")

    def fwd():
        _lib.check(lib.evmi_conv_pkflat_fwd(X.ptr, X.plane, wff.data_ptr(), b.data_ptr(), Y.ptr, Y.plane, wsf.data_ptr(), wsf.numel(), n, s * Tc, cin, cout, k, s,
                                            pad, 1, g, t_out, Tc, 1, 0.1, st), "fwd")

    def dgrad():
        _lib.check(lib.evmi_conv_pkflat_dgrad(Y.ptr, Y.plane, wfd.data_ptr(), DX.ptr, DX.plane, wsd.data_ptr(), wsd.numel(), n, Tc, cin, cout, k, s, pad, 1, g,
                                              t_in, s * Tc, X.ptr, 0, X.plane, s * Tc, 0.1, 0.0, st), "dgrad")

    def wgrad():
        _lib.check(lib.evmi_conv_pkflat_wgrad(X.ptr, X.plane, Y.ptr, Y.plane, dw.data_ptr(), wsw.data_ptr(), wsw.numel(), n, Tc, cin, cout, k, s, pad, 1, g, 0, st),
                   "wgrad")

    t_p = timed(prep)
    t_f, t_d, t_w = timed(fwd), timed(dgrad), timed(wgrad)
    pf = lib.evmi_conv_pkflat_plan(0, n, s * Tc, cin, cout, k, s, pad, 1, g)
    pd = lib.evmi_conv_pkflat_plan(1, n, Tc, cin, cout, k, s, pad, 1, g)
    fmt = lambda p: f"{p % 16}/{p // 16}"  # noqa: E731
    print(f"{name:9s} {n:5d} {t_in:5d} {cin:5d} {cout:5d} {k:3d} {s:2d} {g:3d} | {t_f:8.1f} {flop / t_f * 1e-6:6.0f} {fmt(pf):>5s} | {t_d:8.1f} {flop / t_d * 1e-6:6.0f} {fmt(pd):>5s} | "
          f"{t_w:8.1f} {flop / t_w * 1e-6:6.0f} | {t_p:6.1f}")
    for i, t in enumerate((t_f, t_d, t_w, t_p)):
        tot[i] += t
print("totals (us): fwd %.0f  dgrad %.0f  wgrad %.0f  fragments %.0f" % tuple(tot))
