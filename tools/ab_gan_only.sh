#!/bin/bash
# usage: bash tools/ab_gan_only.sh <reps> "<settings A>" "<settings B>" ...   -- the GAN step alone (tools/train_bench.py 40), interleaved
R=${GRAFT_REPO_ROOT:-.}
reps=$1; shift
for rep in $(seq 1 $reps); do
  for set in "$@"; do
    g=$(env $set OPERANDS=bf16 GRAPH=1 python3 $R/tools/train_bench.py 40 2>/dev/null | grep "^step" | sed 's/ ms.*//; s/step //')
    echo "$rep | $set | GAN $g ms"
  done
done
