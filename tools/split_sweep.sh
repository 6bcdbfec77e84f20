#!/bin/bash
# A/B of the split planners inside the WHOLE training steps (the kernels' own benches time every call alone; in a step eight chains share the chip).
# usage (gpurun): bash tools/split_sweep.sh <out file>      each line: switches | GAN step ms | FastSpeech2 step ms
R=${GRAFT_REPO_ROOT:-.}
OUT=${1:-gpurun_out/split_sweep.txt}
: > $OUT
run() {
  g=$(env "$@" OPERANDS=bf16 GRAPH=1 python3 $R/tools/train_bench.py 40 2>/dev/null | grep "^step" | sed 's/ ms.*//; s/step //')
  f=$(env "$@" OPERANDS=bf16 python3 $R/tools/fs2_train_bench.py 40 2>/dev/null | grep "^step" | sed 's/ ms.*//; s/step //')
  echo "$* | GAN $g ms | FS2 $f ms" | tee -a $OUT
}
run X=0
run EVMI_PK_SPLITK=0
run EVMI_PK_SPLIT_WANT=256
run EVMI_WG_WANT=256
run EVMI_WG_WANT=128
run EVMI_WG_WANT=1024
run EVMI_PK_SPLITK=0 EVMI_WG_WANT=256
run EVMI_WG_TAPSPLIT=0
run X=1
