export F32_LAYERS="${LAYERS:-}"
for cfg in "" ; do
  echo "== $cfg"
  env $cfg python tools/bench_f32conv.py 2>&1 | tail -16 | cut -c1-78
done
OPERANDS=bf16 python tools/train_bench.py 2>&1 | tail -2 | cut -c1-100
OPERANDS=bf16 python tools/fs2_train_bench.py 2>&1 | tail -1 | cut -c1-100
