python tools/microbench/attention_bench.py 2>&1 | grep -v amdgpu
python -m pytest tests/test_gpu_fs2_train.py -q -x -k "attention" 2>&1 | tail -2
OPERANDS=bf16 python tools/fs2_train_bench.py 20 2>&1 | grep "^step"
