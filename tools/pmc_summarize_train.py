#!/usr/bin/env python3
"""Per-kernel PMC summary of tools/gpu_profile_train.sh (and the FastSpeech2 variants) -> profiles/<tag>_pmc_summary.json.

FETCH_SIZE / WRITE_SIZE are in KiB (bytes = counter * 1024, /opt/skills/guides/MI355X_MICROARCH.md).  ONE convention for every leg
(VERDICT r03 item 4): the raw counters are multiplied by the factors MEASURED on known-byte kernels in this library's two access
patterns (tools/pmc_calibrate.sh -> profiles/*_pmc_calibration.json: fp32 4 bytes per lane, packed / time-major 16 bytes per lane);
a kernel takes the 16-byte factors when its bulk traffic is 16-byte units (the matrix-core kernels' LDS-direct loads, the packed /
time-major elementwise kernels), the 4-byte factors otherwise.  `*_raw` keeps the uncorrected counter."""
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


def calibration():
    files = sorted((ROOT / "profiles").glob("*_pmc_calibration.json"))
    if not files:
        return None, {"fetch4": 1.0, "fetch16": 1.0, "write4": 1.0, "write16": 1.0}
    c = json.loads(files[-1].read_text())
    g = lambda k: float(c.get(k) or 1.0)  # noqa: E731
    return files[-1].name, {"fetch4": g("fetch_factor_4B_per_lane"), "fetch16": g("fetch_factor_16B_per_lane"),
                            "write4": g("write_factor_4B_per_lane"), "write16": g("write_factor_16B_per_lane")}


WIDE = ("conv_pk_kernel", "wgrad_pk_kernel", "conv_tc", "resblock_", "pkflat_", "wfrag_flat", "disc_post", "disc_first_wgrad", "disc_first_dgrad",
        "tm_", "relayout_tc", "attention_", "conv_cbt_f32_mfma", "conv_wgrad_f32_mfma", "gemm_f32_mfma", "pack2_kernel", "prep_pk_kernel")
cal_file, cal = calibration()


def wide(name):
    return any(w in name for w in WIDE)


fetch, calls = counters("pmc_fetch")
write, _ = counters("pmc_write")
sq, _ = counters("pmc_sq")
out = {}
for name in sorted(fetch, key=lambda n: -fetch[n].get("FETCH_SIZE", 0)):
    n = max(1, len(calls[name]))
    w16 = wide(name)
    raw_f = fetch[name].get("FETCH_SIZE", 0) * 1024 / n
    raw_w = write.get(name, {}).get("WRITE_SIZE", 0) * 1024 / n
    e = {"launches": n, "fetch_bytes_per_launch": raw_f * cal["fetch16" if w16 else "fetch4"],
         "write_bytes_per_launch": raw_w * cal["write16" if w16 else "write4"],
         "fetch_bytes_per_launch_raw": raw_f, "write_bytes_per_launch_raw": raw_w, "pattern": "16 B per lane" if w16 else "4 B per lane"}
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
total_raw = sum(v["launches"] * (v["fetch_bytes_per_launch_raw"] + v["write_bytes_per_launch_raw"]) for v in out.values())
sys.path.insert(0, str(ROOT / "tools"))
head = ROOT / ".git_head"
json.dump({"steps": steps, "hbm_bytes_per_step": total / steps, "hbm_bytes_per_step_raw": total_raw / steps,
           "launches_per_step": sum(v["launches"] for v in out.values()) / steps,
           "commit": head.read_text().strip() if head.exists() else None,
           "code": __import__("code_fingerprint").code_fingerprint(ROOT),
           "convention": f"counters x factors of profiles/{cal_file}" if cal_file else "raw KiB counters (no calibration file)",
           "calibration": cal},
          open(dst.with_suffix(".meta.json"), "w"), indent=1)
print(f"steps {steps}: {total / steps / 1e9:.2f} GB of HBM traffic per step ({total_raw / steps / 1e9:.2f} raw)")
for k, v in list(out.items())[:12]:
    print(k[:70], {kk: (round(vv, 4) if isinstance(vv, float) and vv < 10 else round(vv)) for kk, vv in v.items()})
print("->", dst)
