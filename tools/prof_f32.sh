# usage: bash tools/prof_f32.sh <tag> [layer-filter]; env EVMI_F32_* pass through.  -> gpurun_out/<tag>/<tag>_kernel_trace.csv
cd /tmp && export TMPDIR=/tmp
export F32_LAYERS="$2"
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/$1 -o $1 -- python3 $GRAFT_REPO_ROOT/tools/bench_f32conv.py > $GRAFT_REPO_ROOT/gpurun_out/$1.log 2>&1
