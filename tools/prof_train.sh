# usage: bash tools/prof_train.sh <tag> [backend]   -> gpurun_out/<tag>_kernel_stats.csv
cd /tmp && export TMPDIR=/tmp
export EVMI_CONV_BACKEND=${2:-mfma,mfma,auto}
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/$1 -o $1 -- python3 $GRAFT_REPO_ROOT/tools/train_bench.py 5 > $GRAFT_REPO_ROOT/gpurun_out/$1.log 2>&1
