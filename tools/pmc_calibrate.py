#!/usr/bin/env python3
"""Known-byte kernels for calibrating rocprofv3's FETCH_SIZE / WRITE_SIZE on this stack, in the access patterns this library uses:
  ew_kernel<5>           fp32, 4 bytes per lane, coalesced: reads N * 4 bytes, writes N * 4 bytes          (channel-major fp32 tensors)
  pkflat_absdiff_kernel  16 bytes per lane, coalesced: reads 2 * units * 16 bytes, writes ~nothing          (packed / time-major bf16 tensors)
  pkflat_zero_kernel     16 bytes per lane: writes units * 16 bytes
Buffers are 1 GiB each (four times the 256 MiB Infinity Cache), every kernel runs three times.
`tools/pmc_calibrate.sh <tag>` runs this under separate --pmc passes and writes profiles/<tag>_pmc_calibration.json.
`python tools/pmc_calibrate.py summarize <tag>` turns the counter files into that JSON."""
import csv
import json
import sys
from collections import defaultdict
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
N_FLOATS = 1 << 28   # 1 GiB of fp32
N_UNITS = 1 << 26    # 1 GiB of 16-byte units
KNOWN = {"ew_kernel<5>": {"read": N_FLOATS * 4, "write": N_FLOATS * 4, "pattern": "fp32, 4 B per lane"},
         "pkflat_absdiff_kernel": {"read": 2 * N_UNITS * 16, "write": 0, "pattern": "16 B per lane"},
         "pkflat_zero_kernel": {"read": 0, "write": N_UNITS * 16, "pattern": "16 B per lane"}}


def run():
    sys.path.insert(0, str(ROOT))
    import torch

    from everyvoice_amd import _lib
    from everyvoice_amd.train import ops

    dev = torch.device("cuda:0")
    lib = _lib.load()
    st = _lib.current_stream_ptr(dev)
    a = torch.empty(N_FLOATS, device=dev)
    b = torch.empty(N_FLOATS, device=dev)
    a.normal_()
    b.normal_()
    slot = torch.zeros(1, device=dev)
    ws = torch.empty(lib.evmi_pkflat_absdiff_ws_elems(1), device=dev)
    pair = (_lib.PkFlatPair * 1)()
    pair[0].a, pair[0].b, pair[0].units, pair[0].plane, pair[0].rows, pair[0].scale = a.data_ptr(), b.data_ptr(), N_UNITS, 0, 1, 1.0
    for _ in range(3):
        ops.copy(a, out=b)
        torch.cuda.synchronize()
        _lib.check(lib.evmi_pkflat_absdiff(1, pair, slot.data_ptr(), ws.data_ptr(), ws.numel(), st), "absdiff")
        torch.cuda.synchronize()
        _lib.check(lib.evmi_pkflat_zero(b.data_ptr(), N_UNITS, st), "zero")
        torch.cuda.synchronize()


def summarize(tag):
    src = ROOT / "gpurun_out"
    out = {}
    for sub, counter, key in (("cal_fetch", "FETCH_SIZE", "read"), ("cal_write", "WRITE_SIZE", "write")):
        f = next((src / f"{tag}_{sub}").rglob("*counter_collection.csv"), None)
        if f is None:
            continue
        agg, n = defaultdict(float), defaultdict(set)
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != counter:
                continue
            name = r["Kernel_Name"].split("(")[0].replace("void evmi::", "").replace("evmi::", "")
            agg[name] += float(r["Counter_Value"])
            n[name].add(r["Dispatch_Id"])
        for name, k in KNOWN.items():
            if name in agg and k[key] > 0:
                per = agg[name] * 1024 / len(n[name])
                out.setdefault(name, {"pattern": k["pattern"]})[f"{counter}_bytes_per_launch_raw"] = round(per)
                out[name][f"known_{key}_bytes"] = k[key]
                out[name][f"{counter}_factor"] = round(k[key] / per, 4)  # multiply the raw counter (KiB * 1024) by this
    res = {"kernels": out,
           "fetch_factor_4B_per_lane": out.get("ew_kernel<5>", {}).get("FETCH_SIZE_factor"),
           "fetch_factor_16B_per_lane": out.get("pkflat_absdiff_kernel", {}).get("FETCH_SIZE_factor"),
           "write_factor_4B_per_lane": out.get("ew_kernel<5>", {}).get("WRITE_SIZE_factor"),
           "write_factor_16B_per_lane": out.get("pkflat_zero_kernel", {}).get("WRITE_SIZE_factor"),
           "note": "bytes = counter * 1024 * factor; measured on 1 GiB buffers (4x the Infinity Cache), three launches each"}
    head = ROOT / ".git_head"
    if head.exists():
        res["commit"] = head.read_text().strip()
    dst = ROOT / "profiles" / f"{tag}_pmc_calibration.json"
    dst.write_text(json.dumps(res, indent=1))
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "summarize":
        summarize(sys.argv[2])
    else:
        run()
