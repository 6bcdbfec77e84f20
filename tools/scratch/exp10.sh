python -m pytest tests/test_gpu_fs2_train.py -q -x -k "layernorm_written or feed_forward_middle" 2>&1 | tail -6
python -m pytest tests/test_gpu_fs2_train.py -q 2>&1 | grep -E "passed|failed|FAILED" | tail -5
for i in 1 2; do OPERANDS=bf16 python tools/fs2_train_bench.py 30 2>&1 | grep "^step" | cut -c1-50; done
