#!/bin/bash
# PMC passes over FastSpeech2 training steps (counters only, one pass per counter group).  usage: bash tools/gpu_profile_fs2_train.sh <tag>
# (eager steps: the same kernels as the graph replays, and every dispatch is visible to the counter collector)
TAG=${1:-r03fs2}
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
export OPERANDS=${OPERANDS:-bf16} EVMI_FS2_GRAPH=0
CMD="python3 $R/tools/fs2_train_bench.py 1"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/${TAG}_pmc_fetch -o p -- $CMD > $OUT/${TAG}_pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/${TAG}_pmc_write -o p -- $CMD > $OUT/${TAG}_pmc_write.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/${TAG}_pmc_sq -o p -- $CMD > $OUT/${TAG}_pmc_sq.log 2>&1
ls $OUT/${TAG}_pmc_*
