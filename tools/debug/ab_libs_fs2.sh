# same-box interleaved A/B of two library builds on the FastSpeech2 training step (bf16, batch 32)
for i in 1 2 3; do for v in old new; do
  EVMI_LIB=$GRAFT_REPO_ROOT/tools/debug/libs/libevmi_$v.so OPERANDS=bf16 python tools/fs2_train_bench.py 20 2>&1 | grep "^step" | cut -c1-40 | sed "s/^/$v /"
done; done
