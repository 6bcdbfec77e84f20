#!/bin/bash
# PMC passes over FastSpeech2 inference forwards (3 warm-up + 2 timed = 5 forwards).  usage: bash tools/gpu_profile_fs2_infer.sh <tag>
TAG=${1:-r04fs2infer}
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
export OPERANDS=${OPERANDS:-bf16}
CMD="python3 $R/tools/fs2_bench.py 2"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/${TAG}_pmc_fetch -o p -- $CMD > $OUT/${TAG}_pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/${TAG}_pmc_write -o p -- $CMD > $OUT/${TAG}_pmc_write.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/${TAG}_pmc_sq -o p -- $CMD > $OUT/${TAG}_pmc_sq.log 2>&1
ls $OUT/${TAG}_pmc_*
