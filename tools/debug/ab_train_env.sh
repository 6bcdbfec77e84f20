# GAN step (graph mode, bf16) under planner switches of the packed training convolution; every setting twice, interleaved
for i in 1 2; do
for cfg in "X=0" "EVMI_PK_WANT=256" "EVMI_PK_WANT=1024" "EVMI_PK_SPLITK=0" "EVMI_PK_WIDE=1" "EVMI_PK_SPLIT_WANT=256" "EVMI_PK_SPLIT_WANT=512"; do
  echo -n "$cfg: "; env $cfg OPERANDS=bf16 GRAPH=1 python tools/train_bench.py 40 2>&1 | grep "^step" | cut -c1-40
done; done
