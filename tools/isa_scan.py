#!/usr/bin/env python3
"""Static scan of the compiled kernels for serialised global loads: a `s_waitcnt vmcnt(0)` directly behind a SINGLE global load
(the signature of a load under a per-lane condition, or of a value parked in accumulation registers) costs one memory round trip
per element.  usage: python tools/isa_scan.py <dir with .s files from hipcc -S --cuda-device-only>"""
import re
import subprocess
import sys
from pathlib import Path


def demangle(n):
    try:
        return subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-cxxfilt", n], capture_output=True, text=True).stdout.strip() or n
    except OSError:
        return n


def lds_dma_handover_findings(path):
    """Second rule: in a kernel that uses LDS-direct loads (global_load_lds), a loop whose header runs into an s_barrier
    must wait `vmcnt(0)` in front of that barrier -- the barrier is where the tiles requested a step ago are handed over, the compiler does
    not order an LDS-direct load against a later ds_read, and a __syncthreads() alone only waits while stores may be outstanding.
    Returns [(kernel, loop label, the instructions from the label to the barrier)] of the loops that do not."""
    lines = open(path).read().split("\n")
    fn, bodies = None, {}
    for line in lines:
        m = re.match(r"^(_Z\w+):", line)
        if m:
            fn = m.group(1)
            bodies[fn] = []
        elif fn is not None:
            t = line.strip()
            if t.startswith(".LBB") or (t and not t.startswith(";") and not t.startswith(".")):
                bodies[fn].append(t)
            if "s_endpgm" in t:
                fn = None
    bad = []
    for name, body in bodies.items():
        if not any("global_load_lds" in x for x in body):
            continue
        for i, x in enumerate(body):
            if x.startswith(".LBB") and "Loop Header" in x:
                win = body[i + 1:i + 8]
                for j, y in enumerate(win):
                    if y.startswith("s_barrier"):
                        if "vmcnt(0)" not in " ".join(win[:j]):
                            bad.append((demangle(name), x.split(":")[0], win[:j + 1]))
                        break
    return bad


def main():
    rows = []
    loops = []
    for f in sorted(Path(sys.argv[1]).glob("*.s")):
        lines = f.read_text().split("\n")
        i = 0
        while i < len(lines):
            m = re.match(r"^(_Z\w+):", lines[i])
            if not m:
                i += 1
                continue
            name, j = m.group(1), i + 1
            seq = []
            while j < len(lines) and not lines[j].startswith(".Lfunc_end"):
                s = lines[j].strip()
                if s.startswith(("global_load_lds", "buffer_load")) and "lds" in s:
                    seq.append("D")
                elif s.startswith(("global_load", "buffer_load", "flat_load")):
                    seq.append("L")
                elif s.startswith("s_waitcnt") and "vmcnt(0)" in s:
                    seq.append("W")
                elif s.startswith(("v_mfma", "s_barrier", "global_store", "ds_", "s_cbranch", "s_branch")):
                    seq.append(".")
                j += 1
            # inner loops that drain behind one or two loads per trip (serial reductions: one memory round trip per element)
            body = lines[i + 1:j]
            labels = {mm.group(1): n for n, l in enumerate(body) if (mm := re.match(r"^(\.LBB\d+_\d+):", l))}
            thin = 0
            for n, l in enumerate(body):
                mm = re.search(r"s_cbranch\w+\s+(\.LBB\d+_\d+)", l)
                if mm and mm.group(1) in labels and labels[mm.group(1)] < n:
                    seg = [x.strip() for x in body[labels[mm.group(1)]:n]]
                    if any(re.match(r"^\.LBB", x) for x in seg[1:]):
                        continue  # not an innermost loop
                    nl = sum(x.startswith(("global_load", "buffer_load", "flat_load")) and "lds" not in x for x in seg)
                    nw = sum(x.startswith("s_waitcnt") and "vmcnt(0)" in x for x in seg)
                    if 1 <= nl <= 2 and nw >= 1:
                        thin += 1
            if thin:
                loops.append((thin, f.stem, demangle(name)[:110]))
            t = "".join(seq)
            t = re.sub(r"\.+", ".", t)
            singles = len(re.findall(r"(?<!L)L\.?W", t))  # one load, then a full drain
            loads = t.count("L")
            if singles >= 4:
                rows.append((singles, loads, f.stem, demangle(name)[:110]))
            i = j
    for r in sorted(rows, reverse=True):
        print(f"{r[0]:4d} single-load drains of {r[1]:4d} loads  {r[2]:22s} {r[3]}")
    for r in sorted(loops, reverse=True):
        print(f"{r[0]:4d} thin loop(s): <= 2 loads then vmcnt(0) per trip  {r[1]:22s} {r[2]}")


def main_lds_dma(d):
    for f in sorted(Path(d).glob("*.s")):
        for kernel, label, win in lds_dma_handover_findings(f):
            print(f"{f.name}: {kernel[:90]}: loop {label}: barrier without vmcnt(0) in front of it: {win}")


if __name__ == "__main__":
    main()
    main_lds_dma(sys.argv[1])
