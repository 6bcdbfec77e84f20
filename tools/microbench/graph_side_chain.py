#!/usr/bin/env python3
"""How a captured HIP graph runs a SIDE chain whose nodes each depend on one node of the main chain (the weight gradients beside the
input-gradient chain of a backward pass): main k_1 .. k_N, side s_1 .. s_N with s_i after k_i.  Each kernel is a one-workgroup spin of
~20 us, so two chains that overlap take N x 20 us and two that do not 2N x 20 us.  Patterns:
  A  one side stream, s_i waits for an event recorded after k_i; one join at the end           (what ops.side_wgrad captures)
  B  per-layer fork / join: s_i beside k_{i+1}, main waits for s_i before k_{i+2}
  C  as A over 4 side streams in rotation, all joined at the end
  D  groups of g layers: after k_1 .. k_g one fork, s_1 .. s_g on the side beside k_{g+1} .. k_{2g}; the side group is joined before the next fork
usage: python tools/microbench/graph_side_chain.py [N=100] [spin cycles=40000]"""
import sys
import time

import torch

N = int(sys.argv[1]) if len(sys.argv) > 1 else 100
CYC = int(sys.argv[2]) if len(sys.argv) > 2 else 40000
dev = torch.device("cuda:0")
main = torch.cuda.Stream(dev)
sides = [torch.cuda.Stream(dev) for _ in range(4)]


def spin():
    torch.cuda._sleep(CYC)


def pattern_a(n_side=1):
    cur = torch.cuda.current_stream(dev)
    used = set()
    for i in range(N):
        spin()
        ev = torch.cuda.Event()
        ev.record(cur)
        s = sides[i % n_side]
        s.wait_event(ev)
        used.add(i % n_side)
        with torch.cuda.stream(s):
            spin()
    for j in used:
        ev = torch.cuda.Event()
        ev.record(sides[j])
        cur.wait_event(ev)


def pattern_b():
    cur = torch.cuda.current_stream(dev)
    pending = None
    for i in range(N):
        spin()  # k_i
        if pending is not None:  # s_{i-1} ran beside k_i: join it now
            cur.wait_event(pending)
        ev = torch.cuda.Event()
        ev.record(cur)
        sides[0].wait_event(ev)
        with torch.cuda.stream(sides[0]):
            spin()
            pending = torch.cuda.Event()
            pending.record(sides[0])
    cur.wait_event(pending)


def pattern_d(g):
    cur = torch.cuda.current_stream(dev)
    pending = None
    for i0 in range(0, N, g):
        n = min(g, N - i0)
        for _ in range(n):
            spin()
        if pending is not None:
            cur.wait_event(pending)
        ev = torch.cuda.Event()
        ev.record(cur)
        sides[0].wait_event(ev)
        with torch.cuda.stream(sides[0]):
            for _ in range(n):
                spin()
            pending = torch.cuda.Event()
            pending.record(sides[0])
    cur.wait_event(pending)


def timed(fn, reps=5):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


def captured(body):
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(main):
        body()  # warm-up
        torch.cuda.synchronize()
        with torch.cuda.graph(g, stream=main):
            body()
    return g.replay


with torch.cuda.stream(main):
    t_one = timed(lambda: [spin() for _ in range(N)])
print(f"{N} spins in one stream, eager: {t_one:.2f} ms ({t_one / N * 1e3:.1f} us each)")
for name, body in (("A one side stream", lambda: pattern_a(1)), ("B per-layer fork / join", pattern_b), ("C four side streams", lambda: pattern_a(4)),
                   ("D groups of 2", lambda: pattern_d(2)), ("D groups of 4", lambda: pattern_d(4)), ("D groups of 8", lambda: pattern_d(8)),
                   ("D groups of 16", lambda: pattern_d(16))):
    with torch.cuda.stream(main):
        t_e = timed(body)
    t_g = timed(captured(body))
    print(f"{name:26s} eager {t_e:6.2f} ms   graph replay {t_g:6.2f} ms   (overlapped = {t_one:.2f}, serial = {2 * t_one:.2f})")
