# same-box interleaved A/B of two library builds on the GAN step (graph mode, bf16)
for i in 1 2 3; do for v in old new; do
  EVMI_LIB=$GRAFT_REPO_ROOT/tools/debug/libs/libevmi_$v.so OPERANDS=bf16 GRAPH=1 python tools/train_bench.py 40 2>&1 | grep "^step" | cut -c1-40 | sed "s/^/$v /"
done; done
