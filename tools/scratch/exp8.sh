export OPERANDS=bf16 GRAPH=1
for v in 1 0 1 0; do echo "splitk $v"; EVMI_PK_SPLITK=$v python tools/train_bench.py 40 2>&1 | grep "^step" | cut -c1-50; done
