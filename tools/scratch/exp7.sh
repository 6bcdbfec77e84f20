python -m pytest tests/test_gpu_train_ops.py -q -x -k "every_tile_forced" 2>&1 | tail -2
python -m pytest tests/test_gpu_fs2_train.py -q -x 2>&1 | grep -E "passed|failed" | tail -2
for i in 1 2; do OPERANDS=bf16 python tools/fs2_train_bench.py 30 2>&1 | grep "^step" | cut -c1-50; done
EVMI_FS2_GRAPH=0 OPERANDS=bf16 python tools/fs2_train_bench.py 30 2>&1 | grep "^step" | cut -c1-50
LEARN=1 SIDE=1 GRAPH=1 python tools/scratch/lockstep.py 150 2>&1 | grep -v amdgpu | cut -c1-200 | tail -2
