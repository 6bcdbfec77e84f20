python tools/scratch/lockstep.py 300 2>&1 | grep -v amdgpu | cut -c1-300 | tail -4
python -m pytest tests/test_gpu_fs2_train.py -q -x 2>&1 | tail -3
python -m pytest tests/test_gpu_generator.py tests/test_gpu_fs2.py -q -x 2>&1 | tail -2
python bench.py --no-train --no-fs2 --no-side-legs --no-cpu-baseline 2>&1 | tail -1 | cut -c1-400
OPERANDS=bf16 python tools/fs2_train_bench.py 20 2>&1 | grep "^step"
