#!/bin/bash
# PMC passes over tools/sweep_conv.py for a few conv_tc variants (run on the GPU box): gpurun_out/<tag>_pmcv_*.
# usage: bash tools/pmc_conv_variants.sh <tag> variant...
TAG=$1; shift
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > $OUT/${TAG}_counters_list.txt 2>&1
CMD="python3 $R/tools/sweep_conv.py --rounds=1 --exact $*"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d $OUT/${TAG}_pmcv_sq -o p -- $CMD > $OUT/${TAG}_pmcv_sq.log 2>&1
rocprofv3 --pmc SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/${TAG}_pmcv_sq2 -o p -- $CMD > $OUT/${TAG}_pmcv_sq2.log 2>&1
python3 - <<PY
import csv, glob, collections
for sub in ("sq", "sq2"):
    for f in glob.glob("$OUT/${TAG}_pmcv_%s/**/*counter_collection.csv" % sub, recursive=True):
        agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(int)
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if "conv_tc" not in k: continue
            k = k.split("<evmi::")[1].split(">")[0]
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        for k, v in agg.items():
            print(sub, k)
            base = v.get("SQ_WAVE_CYCLES") or 1.0
            for c, x in sorted(v.items()):
                print(f"    {c:28s} {x:16.0f}  /wave_cycles {x / base:8.4f}")
PY
