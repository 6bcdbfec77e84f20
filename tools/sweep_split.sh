export F32_LAYERS="MSD L6,MPD p2 L4,MPD p2 L3,MPD p11 L4,G rb c256,G rb c128"
for cfg in "" "EVMI_PK_SPLIT_WANT=512" "EVMI_PK_SPLIT_WANT=256" "EVMI_PK_SPLIT_MINKB=48" "EVMI_PK_SPLIT_MINKB=12 EVMI_PK_SPLIT_WANT=512" "EVMI_PK_SPLIT_BELOW=512"; do
  echo "== $cfg"
  env $cfg python tools/bench_f32conv.py 2>&1 | tail -6 | cut -c1-78
done
