#!/bin/bash
# Stall-reason counters of the kernels a command launches (run on the GPU box): three `rocprofv3 --pmc` passes (8 SQ slots each, no
# tracing domains besides --kernel-trace), then per kernel: counters summed over its dispatches and the derived fractions.
# usage: bash tools/pmc_kernels.sh <tag> <kernel-name-filter-regex> -- python3 script.py args...
TAG=$1; FILTER=$2; shift 3
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > $OUT/${TAG}_counters_list.txt 2>&1
have() { for c in "$@"; do grep -qw "$c" $OUT/${TAG}_counters_list.txt && echo -n "$c "; done; }
P1=$(have SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE)
P2=$(have SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU GRBM_GUI_ACTIVE)
P3=$(have SQ_WAVES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_INSTS_SMEM SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_BUSY_CU_CYCLES)
i=0
for P in "$P1" "$P2" "$P3"; do
  i=$((i+1))
  rocprofv3 --pmc $P --kernel-trace --output-format csv -d $OUT/${TAG}_pmck_$i -o p -- "$@" > $OUT/${TAG}_pmck_$i.log 2>&1
done
python3 - <<PY
import csv, glob, collections, json, re
agg = collections.defaultdict(lambda: collections.defaultdict(float)); calls = collections.defaultdict(int)
dur = collections.defaultdict(float)
for i in (1, 2, 3):
    for f in glob.glob("$OUT/${TAG}_pmck_%d/**/*counter_collection.csv" % i, recursive=True):
        seen = set()
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if not re.search(r"""$FILTER""", k): continue
            k = re.sub(r"^void ", "", k).replace("evmi::", "").split("(")[0][:100]
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
            if i == 1 and r["Counter_Name"] == "SQ_WAVE_CYCLES":
                calls[k] += 1
out = {}
for k, v in sorted(agg.items()):
    n = max(calls[k], 1)
    wc = v.get("SQ_WAVE_CYCLES", 0.0) or 1.0
    d = {"dispatches": n, "per_dispatch": {c: x / n for c, x in sorted(v.items())}}
    gui = v.get("GRBM_GUI_ACTIVE", 0.0) / n
    der = {}
    if gui:
        der["mfma_busy_frac_of_clocked_simd_cycles"] = v.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / n / (1024.0 * gui / 8.0)
        der["waves_resident_avg_per_simd"] = (v.get("SQ_WAVE_CYCLES", 0.0) / n * 4.0) / (1024.0 * gui / 8.0) / 8.0 if False else None
    der["wait_inst_any_over_wave_cycles"] = v.get("SQ_WAIT_INST_ANY", 0.0) / wc
    der["wait_any_over_wave_cycles"] = v.get("SQ_WAIT_ANY", 0.0) / wc
    der["active_inst_any_over_wave_cycles"] = v.get("SQ_ACTIVE_INST_ANY", 0.0) / wc
    der["wait_inst_lds_over_wave_cycles"] = v.get("SQ_WAIT_INST_LDS", 0.0) / wc
    der["lds_bank_conflict_frac"] = v.get("SQ_LDS_BANK_CONFLICT", 0.0) / (v.get("SQ_LDS_IDX_ACTIVE", 0.0) or 1.0)
    if v.get("SQ_INSTS_MFMA"):
        der["valu_insts_per_mfma"] = (v.get("SQ_INSTS_VALU", 0.0) - v["SQ_INSTS_MFMA"]) / v["SQ_INSTS_MFMA"]
        der["lds_insts_per_mfma"] = v.get("SQ_INSTS_LDS", 0.0) / v["SQ_INSTS_MFMA"]
        der["salu_insts_per_mfma"] = v.get("SQ_INSTS_SALU", 0.0) / v["SQ_INSTS_MFMA"]
    if v.get("SQ_WAVES"):
        der["wave_cycles_per_wave"] = v.get("SQ_WAVE_CYCLES", 0.0) / v["SQ_WAVES"]
    d["derived"] = {a: b for a, b in der.items() if b is not None}
    out[k] = d
json.dump(out, open("$OUT/${TAG}_pmck_summary.json", "w"), indent=1)
for k, d in out.items():
    print(k, d["dispatches"])
    for a, b in d["derived"].items():
        print(f"    {a:44s} {b:10.4f}")
PY
