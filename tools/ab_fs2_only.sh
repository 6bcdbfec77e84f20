#!/bin/bash
# usage: bash tools/ab_fs2_only.sh <reps> "<settings A>" "<settings B>" ...   -- the FastSpeech2 step alone (tools/fs2_train_bench.py 40), interleaved
R=${GRAFT_REPO_ROOT:-.}
reps=$1; shift
for rep in $(seq 1 $reps); do
  for set in "$@"; do
    f=$(env $set OPERANDS=bf16 python3 $R/tools/fs2_train_bench.py 40 2>/dev/null | grep "^step" | sed 's/ ms.*//; s/step //')
    echo "$rep | $set | FS2 $f ms"
  done
done
