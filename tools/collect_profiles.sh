#!/bin/bash
# One round's profile set on the GPU box, reduced to what gets committed (summaries, not raw traces: gpurun merges <= 64 MiB back).
# usage (through gpurun): bash tools/collect_profiles.sh <tag> <what...>   what: bench | stats | pmc_infer | pmc_train | pmc_fs2 | pmc_fs2infer
# Everything ends up in gpurun_out/<tag>_*; copy what should be judged into profiles/.
TAG=$1; shift
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out
mkdir -p $OUT
for what in "$@"; do
  case $what in
    bench)
      python3 $R/bench.py > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_bench.err ;;
    stats)
      OPERANDS=bf16 GRAPH=1 bash $R/tools/prof_train.sh ${TAG}_train 10
      python3 $R/tools/trace_phases.py $OUT/${TAG}_train/${TAG}_train_kernel_trace.csv > $OUT/${TAG}_train_phases.txt
      cp $OUT/${TAG}_train/${TAG}_train_kernel_stats.csv $OUT/${TAG}_train_bf16_graph_kernel_stats.csv; rm -rf $OUT/${TAG}_train
      OPERANDS=bf16 bash $R/tools/prof_fs2_train.sh ${TAG}_fs2
      cp $OUT/${TAG}_fs2/${TAG}_fs2_kernel_stats.csv $OUT/${TAG}_fs2_train_bf16_kernel_stats.csv; rm -rf $OUT/${TAG}_fs2 ;;
    pmc_infer)
      BENCH_FLAGS="--no-train --no-fs2" bash $R/tools/gpu_profile.sh $TAG > $OUT/${TAG}_gpu_profile.log 2>&1
      python3 $R/tools/pmc_summarize.py $TAG > $OUT/${TAG}_pmc_summarize.log 2>&1
      cp $R/profiles/${TAG}_kernel_stats.csv $R/profiles/${TAG}_pmc_summary.json $R/profiles/${TAG}_pmc_summary.meta.json $OUT/ 2>/dev/null
      rm -rf $OUT/${TAG}_trace $OUT/${TAG}_pmc_fetch $OUT/${TAG}_pmc_write $OUT/${TAG}_pmc_sq $OUT/${TAG}_pmc_l2 ;;
    pmc_train)
      GRAPH=0 bash $R/tools/gpu_profile_train.sh ${TAG}_train > $OUT/${TAG}_train_pmc.log 2>&1
      python3 $R/tools/pmc_summarize_train.py ${TAG}_train 5 > $OUT/${TAG}_train_pmc_summarize.log 2>&1
      cp $R/profiles/${TAG}_train_pmc_summary.json $R/profiles/${TAG}_train_pmc_summary.meta.json $OUT/ 2>/dev/null
      rm -rf $OUT/${TAG}_train_pmc_fetch $OUT/${TAG}_train_pmc_write $OUT/${TAG}_train_pmc_sq ;;
    pmc_fs2)
      bash $R/tools/gpu_profile_fs2_train.sh ${TAG}_fs2 > $OUT/${TAG}_fs2_pmc.log 2>&1
      python3 $R/tools/pmc_summarize_train.py ${TAG}_fs2 5 > $OUT/${TAG}_fs2_pmc_summarize.log 2>&1
      cp $R/profiles/${TAG}_fs2_pmc_summary.json $R/profiles/${TAG}_fs2_pmc_summary.meta.json $OUT/ 2>/dev/null
      rm -rf $OUT/${TAG}_fs2_pmc_fetch $OUT/${TAG}_fs2_pmc_write $OUT/${TAG}_fs2_pmc_sq ;;
    pmc_fs2infer)
      bash $R/tools/gpu_profile_fs2_infer.sh ${TAG}_fs2infer > $OUT/${TAG}_fs2infer_pmc.log 2>&1
      python3 $R/tools/pmc_summarize_train.py ${TAG}_fs2infer 5 > $OUT/${TAG}_fs2infer_pmc_summarize.log 2>&1
      cp $R/profiles/${TAG}_fs2infer_pmc_summary.json $R/profiles/${TAG}_fs2infer_pmc_summary.meta.json $OUT/ 2>/dev/null
      rm -rf $OUT/${TAG}_fs2infer_pmc_fetch $OUT/${TAG}_fs2infer_pmc_write $OUT/${TAG}_fs2infer_pmc_sq ;;
  esac
done
du -sh $OUT; ls $OUT | grep $TAG
