mkdir -p gpurun_out/r05c
nproc > gpurun_out/r05c/nproc.txt
timeout 900 python -m pytest tests/test_gpu_train_ops.py -q -x -k "planner_switches or every_tile" --durations=20 > gpurun_out/r05c/ops_children.log 2>&1; echo rc=$? >> gpurun_out/r05c/ops_children.log
tail -30 gpurun_out/r05c/ops_children.log; cat gpurun_out/r05c/nproc.txt
