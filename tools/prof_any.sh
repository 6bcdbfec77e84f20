# usage: bash tools/prof_any.sh <tag> <python script> [args...]  -> gpurun_out/<tag>/<tag>_kernel_stats.csv
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/$tag -o $tag -- python3 $GRAFT_REPO_ROOT/$@ > $GRAFT_REPO_ROOT/gpurun_out/$tag.log 2>&1
