#!/usr/bin/env python3
"""Per-kernel PMC summary of tools/gpu_profile_train.sh -> profiles/<tag>_pmc_summary.json.

FETCH_SIZE / WRITE_SIZE are in KiB (bytes = counter * 1024, /opt/skills/guides/MI355X_MICROARCH.md); the guide's gfx950 note
(FETCH_SIZE counts half the bytes of wide 16 B / lane streaming reads) is NOT applied here because these kernels mix
4-byte and 16-byte per-lane loads: `fetch_bytes` is the raw counter, `fetch_bytes_x2` the corrected upper bound."""
import csv
import json
import sys
from collections import defaultdict
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
tag = sys.argv[1]
src = ROOT / "gpurun_out"


def counters(sub):
    f = next((src / f"{tag}_{sub}").rglob("*counter_collection.csv"), None)
    agg = defaultdict(lambda: defaultdict(float))
    calls = defaultdict(set)
    if f is None:
        return agg, calls
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"].split("(")[0].replace("void evmi::", "")
        agg[name][r["Counter_Name"]] += float(r["Counter_Value"])
        calls[name].add(r["Dispatch_Id"])
    return agg, calls


fetch, calls = counters("pmc_fetch")
write, _ = counters("pmc_write")
sq, _ = counters("pmc_sq")
out = {}
for name in sorted(fetch, key=lambda n: -fetch[n].get("FETCH_SIZE", 0)):
    n = max(1, len(calls[name]))
    e = {"launches": n, "fetch_bytes_per_launch": fetch[name].get("FETCH_SIZE", 0) * 1024 / n,
         "write_bytes_per_launch": write.get(name, {}).get("WRITE_SIZE", 0) * 1024 / n}
    e["fetch_bytes_x2_per_launch"] = 2 * e["fetch_bytes_per_launch"]
    s = sq.get(name, {})
    if s.get("GRBM_GUI_ACTIVE"):
        # SQ_VALU_MFMA_BUSY_CYCLES sums the busy cycles of all 1024 SIMDs; GRBM_GUI_ACTIVE sums the 8 XCDs' active cycles:
        # fraction of the clocked SIMD-cycles with the matrix pipe busy = MFMA_BUSY / (GUI_ACTIVE / 8 * 1024)
        e["mfma_busy_frac"] = s.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / (s["GRBM_GUI_ACTIVE"] / 8 * 1024)
        e["lds_bank_conflict_frac"] = s.get("SQ_LDS_BANK_CONFLICT", 0) / max(1.0, s.get("SQ_LDS_IDX_ACTIVE", 0))
        e["active_cycles_per_launch"] = s["GRBM_GUI_ACTIVE"] / 8 / n
    out[name] = e
dst = ROOT / "profiles" / f"{tag}_pmc_summary.json"
json.dump(out, open(dst, "w"), indent=1)
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 5  # tools/train_bench.py 1: four warm-up steps + one timed
total = sum(v["launches"] * (v["fetch_bytes_per_launch"] + v["write_bytes_per_launch"]) for v in out.values())
json.dump({"steps": steps, "hbm_bytes_per_step": total / steps, "launches_per_step": sum(v["launches"] for v in out.values()) / steps},
          open(dst.with_suffix(".meta.json"), "w"))
print(f"steps {steps}: {total / steps / 1e9:.2f} GB of HBM traffic per step (raw counters)")
for k, v in list(out.items())[:12]:
    print(k[:70], {kk: (round(vv, 4) if isinstance(vv, float) and vv < 10 else round(vv)) for kk, vv in v.items()})
print("->", dst)
