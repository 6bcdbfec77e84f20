# usage: bash tools/prof_fs2_train.sh <tag>   -> gpurun_out/<tag>/<tag>_kernel_stats.csv
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/$1 -o $1 -- python3 $GRAFT_REPO_ROOT/tools/fs2_train_bench.py 3 > $GRAFT_REPO_ROOT/gpurun_out/$1.log 2>&1
