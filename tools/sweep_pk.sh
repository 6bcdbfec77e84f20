export F32_LAYERS="${LAYERS:-G rb c128,G rb c64,MSD L6,MPD p2 L4,MSD L1,MSD L3,MPD p11 L3}"
for cfg in "" "EVMI_PK_WANT=256" "EVMI_PK_WANT=1024"; do
  echo "== $cfg"
  env $cfg python tools/bench_f32conv.py 2>&1 | tail -7 | cut -c1-78
done
OPERANDS=bf16 python tools/train_bench.py 2>&1 | tail -2 | cut -c1-100
