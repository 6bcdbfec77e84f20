#!/bin/bash
# LDS bank-conflict share of the packed training convolutions for two builds of the library (EVMI_LIB), one --pmc pass each
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out; cd /tmp && export TMPDIR=/tmp
export OPERANDS=bf16 GRAPH=0 STREAMS=1
for v in old new; do
  EVMI_LIB=$R/tools/debug/libs/libevmi_$v.so rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $OUT/r02z_lds_$v -o p -- python3 $R/tools/train_bench.py 1 > $OUT/r02z_lds_$v.log 2>&1
done
python3 - <<PY
import csv, glob, collections
for v in ("old", "new"):
    f = glob.glob("$OUT/r02z_lds_%s/**/*counter_collection.csv" % v, recursive=True)[0]
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "conv_pk_kernel" not in k: continue
        agg[k.split("(")[0]][r["Counter_Name"]] += float(r["Counter_Value"])
    for k, c in sorted(agg.items()):
        print(v, k, "conflict/active = %.3f" % (c["SQ_LDS_BANK_CONFLICT"] / max(c["SQ_LDS_IDX_ACTIVE"], 1)), "mfma busy/sq busy = %.3f" % (c["SQ_VALU_MFMA_BUSY_CYCLES"] / max(c["SQ_BUSY_CYCLES"], 1)))
PY
