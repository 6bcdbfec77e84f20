# usage: bash tools/prof_train.sh <tag> [steps]   -> gpurun_out/<tag>/...kernel_stats.csv, kernel_trace.csv  (env OPERANDS / GRAPH / STREAMS pass through)
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/$1 -o $1 -- python3 $GRAFT_REPO_ROOT/tools/train_bench.py ${2:-5} > $GRAFT_REPO_ROOT/gpurun_out/$1.log 2>&1
