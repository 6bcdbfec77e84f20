export OPERANDS=bf16 GRAPH=1
for v in 0 1 0 1; do echo "small tiles $v"; EVMI_TM_SMALL_TILES=$v python tools/train_bench.py 40 2>&1 | grep "^step" | cut -c1-50; done
python -m pytest tests/test_gpu_train_step.py -q -x 2>&1 | tail -2
