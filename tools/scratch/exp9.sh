python -m pytest tests/test_gpu_fs2_train.py -q -x -k "feed_forward_middle" 2>&1 | tail -8
python -m pytest tests/test_gpu_fs2_train.py -q -x 2>&1 | grep -E "passed|failed|Error|assert" | tail -5
for i in 1 2; do OPERANDS=bf16 python tools/fs2_train_bench.py 30 2>&1 | grep "^step" | cut -c1-50; done
