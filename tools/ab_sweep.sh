#!/bin/bash
# A/B of environment switches inside the WHOLE training steps, interleaved (box clocks drift): every setting twice.
# usage (gpurun): bash tools/ab_sweep.sh <out file> "<settings A>" "<settings B>" ...     settings: space-separated VAR=value (or X=0 for none)
R=${GRAFT_REPO_ROOT:-.}
OUT=$1; shift
: > $OUT
for rep in 1 2; do
  for set in "$@"; do
    g=$(env $set OPERANDS=bf16 GRAPH=1 python3 $R/tools/train_bench.py 40 2>/dev/null | grep "^step" | sed 's/ ms.*//; s/step //')
    f=$(env $set OPERANDS=bf16 python3 $R/tools/fs2_train_bench.py 40 2>/dev/null | grep "^step" | sed 's/ ms.*//; s/step //')
    echo "$set | GAN $g ms | FS2 $f ms" | tee -a $OUT
  done
done
