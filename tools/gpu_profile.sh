#!/bin/bash
# Run on the GPU box (gpurun): kernel-trace stats of the bench command + separate PMC passes.
# usage: bash tools/gpu_profile.sh <tag>     -> gpurun_out/<tag>_*
TAG=${1:-r01}
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
FLAGS="${BENCH_FLAGS:---no-train --no-fs2} --no-side-legs"
BENCH="python3 $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline --profile-passes 1 $FLAGS"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_trace -o t -- $BENCH > $OUT/${TAG}_trace.log 2>&1
# PMC passes: counters only (no other trace domains), one TCC-heavy counter per pass
BENCH1="python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --profile-passes 1 $FLAGS"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/${TAG}_pmc_fetch -o p -- $BENCH1 > $OUT/${TAG}_pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/${TAG}_pmc_write -o p -- $BENCH1 > $OUT/${TAG}_pmc_write.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d $OUT/${TAG}_pmc_sq -o p -- $BENCH1 > $OUT/${TAG}_pmc_sq.log 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d $OUT/${TAG}_pmc_l2 -o p -- $BENCH1 > $OUT/${TAG}_pmc_l2.log 2>&1
ls $OUT/${TAG}_*
