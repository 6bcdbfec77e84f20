#!/usr/bin/env python3
"""Summarise the rocprofv3 outputs of tools/gpu_profile.sh into profiles/<tag>_*.

  python tools/pmc_summarize.py <tag>

HBM traffic follows /opt/skills/guides/MI355X_MICROARCH.md §HBM: FETCH_SIZE / WRITE_SIZE are in KiB
(hbm_bytes = counter * 1024) and on gfx950 FETCH_SIZE reports half the bytes of a wide (16 B / lane)
coalesced streaming read, so the read side is doubled for these kernels, whose global reads are all
16-B-per-lane row-contiguous loads.  WRITE_SIZE is uncalibrated in the guide and taken as is.
"""
import csv
import json
import re
import shutil
import sys
from collections import defaultdict
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
src = ROOT / "gpurun_out"
dst = ROOT / "profiles"
dst.mkdir(exist_ok=True)


def short(name: str) -> str:
    m = re.search(r"ConvTcCfg<([^>]*)>", name)
    if m:
        p = [x.strip() for x in m.group(1).split(",")]
        return f"conv_tc_mfma<c{p[0]},k{p[6]},bm{p[2]},bn{p[3]},kc{p[1]},t{p[7]}>"
    m = re.search(r"ConvDmaCfg<([^>]*)>", name)
    if m:
        p = [x.strip() for x in m.group(1).split(",")]
        return f"conv_tc_dma<c{p[0]},k{p[1]},bm128,bn256,kc64>"
    m = re.search(r"PairCfg<([^>]*)>", name)
    if m:
        p = [x.strip() for x in m.group(1).split(",")]
        return f"resblock_pair_mfma<c{p[0]},k{p[1]},bn{p[2]},t{p[3]}>"
    return re.sub(r"\(.*", "", name).replace("void evmi::", "")[:60]


def counters(sub):
    f = next((src / f"{tag}_{sub}").rglob("*counter_collection.csv"), None)
    agg = defaultdict(lambda: defaultdict(float))
    n = defaultdict(lambda: defaultdict(int))
    if f is None:
        return agg, n
    for r in csv.DictReader(open(f)):
        k = short(r["Kernel_Name"])
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        n[k][r["Counter_Name"]] += 1
    return agg, n


_cal = sorted((ROOT / "profiles").glob("*_pmc_calibration.json"))
_c = json.loads(_cal[-1].read_text()) if _cal else {}
FETCH16 = float(_c.get("fetch_factor_16B_per_lane") or 2.0)
WRITE16 = float(_c.get("write_factor_16B_per_lane") or 1.0)
stats = next((src / f"{tag}_trace").rglob("*kernel_stats.csv"), None)
if stats:
    shutil.copy(stats, dst / f"{tag}_kernel_stats.csv")
out = {}
fetch, nf = counters("pmc_fetch")
write, nw = counters("pmc_write")
sq, nsq = counters("pmc_sq")
l2, nl2 = counters("pmc_l2")
for k in sorted(set(fetch) | set(write) | set(sq)):
    e = {}
    if "FETCH_SIZE" in fetch[k]:
        per = fetch[k]["FETCH_SIZE"] / nf[k]["FETCH_SIZE"]
        e["fetch_size_kib_per_launch_raw"] = round(per, 1)
        e["hbm_read_bytes_per_launch"] = round(per * 1024 * FETCH16)  # measured factor for 16 B per lane streaming reads (2.0 without a calibration file: the guide's gfx950 note)
    if "WRITE_SIZE" in write[k]:
        per = write[k]["WRITE_SIZE"] / nw[k]["WRITE_SIZE"]
        e["write_size_kib_per_launch_raw"] = round(per, 1)
        e["hbm_write_bytes_per_launch"] = round(per * 1024 * WRITE16)
    if "hbm_read_bytes_per_launch" in e and "hbm_write_bytes_per_launch" in e:
        e["hbm_bytes_per_launch"] = e["hbm_read_bytes_per_launch"] + e["hbm_write_bytes_per_launch"]
    if "SQ_WAVE_CYCLES" in sq[k]:
        wc = sq[k]["SQ_WAVE_CYCLES"]
        e["wave_cycles_frac"] = {
            "wait_any": round(sq[k]["SQ_WAIT_ANY"] / wc, 3),
            "wait_inst_any": round(sq[k]["SQ_WAIT_INST_ANY"] / wc, 3),
            "active_inst_any": round(sq[k]["SQ_ACTIVE_INST_ANY"] / wc, 3),
        }
        if sq[k].get("SQ_LDS_IDX_ACTIVE"):
            e["lds_bank_conflict_frac"] = round(sq[k]["SQ_LDS_BANK_CONFLICT"] / sq[k]["SQ_LDS_IDX_ACTIVE"], 4)
        if l2[k].get("GRBM_GUI_ACTIVE"):
            # MFMA utilisation against the chip: SQ_VALU_MFMA_BUSY_CYCLES sums the busy cycles of all 1024 SIMDs (256 CUs x 4),
            # GRBM_GUI_ACTIVE the 8 XCDs' active cycles (separate passes of the same command: per-launch averages of each)
            mf = sq[k]["SQ_VALU_MFMA_BUSY_CYCLES"] / max(1, nsq[k]["SQ_VALU_MFMA_BUSY_CYCLES"])
            ga = l2[k]["GRBM_GUI_ACTIVE"] / max(1, nl2[k]["GRBM_GUI_ACTIVE"])
            e["mfma_busy_frac"] = round(mf / (ga / 8 * 1024), 4)
            e["active_cycles_per_launch"] = round(ga / 8)
    if l2[k].get("TCC_HIT_sum") is not None and (l2[k]["TCC_HIT_sum"] + l2[k]["TCC_MISS_sum"]) > 0:
        e["l2_hit_rate"] = round(l2[k]["TCC_HIT_sum"] / (l2[k]["TCC_HIT_sum"] + l2[k]["TCC_MISS_sum"]), 3)
    out[k] = e
(dst / f"{tag}_pmc_summary.json").write_text(json.dumps(out, indent=1, sort_keys=True))
import sys as _sys
_sys.path.insert(0, str(ROOT / "tools"))
_head = ROOT / ".git_head"
(dst / f"{tag}_pmc_summary.meta.json").write_text(json.dumps({
    "commit": _head.read_text().strip() if _head.exists() else None,
    "code": __import__("code_fingerprint").code_fingerprint(ROOT),
    "convention": f"counters x factors of profiles/{_cal[-1].name}" if _cal else "FETCH_SIZE x 2 (guide), WRITE_SIZE raw",
    "fetch_factor": FETCH16, "write_factor": WRITE16}, indent=1))
print(json.dumps({k: v for k, v in out.items() if "conv_tc_mfma<c128,k11" in k or "pair_mfma<c64,k11" in k}, indent=1))
