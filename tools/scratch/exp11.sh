python tools/microbench/ln_bench.py
python -m pytest tests/test_gpu_fs2_train.py -q -k "layernorm" 2>&1 | tail -2
for i in 1 2; do OPERANDS=bf16 python tools/fs2_train_bench.py 30 2>&1 | grep "^step" | cut -c1-50; done
