python tools/microbench/table_bwd_bench.py 2>&1 | grep -v amdgpu
python -m pytest tests/test_gpu_fs2_train.py -q 2>&1 | grep -E "passed|failed" | tail -2
for i in 1 2; do OPERANDS=bf16 python tools/fs2_train_bench.py 30 2>&1 | grep "^step" | cut -c1-50; done
