export OPERANDS=bf16 GRAPH=1
for v in 0 1 0 1; do echo "defer $v"; EVMI_TM_DEFER_WGRAD=$v python tools/train_bench.py 40 2>&1 | grep "^step\|graph:" | cut -c1-60; done
python -m pytest tests/test_gpu_train_step.py -q -x 2>&1 | grep -E "passed|failed|Error" | tail -3
