export F32_LAYERS="MPD p2 L4,MSD L6,G rb c128,G rb c256,MPD p11 L3"
for cfg in "" "EVMI_BF_NST=3 EVMI_BF_PS=8" "EVMI_BF_NST=3 EVMI_BF_PS=16" "EVMI_BF_NST=2 EVMI_BF_PS=32" "EVMI_BF_TILE=0" "EVMI_BF_TILE=0 EVMI_BF_NST=3 EVMI_BF_PS=8" "EVMI_BF_TILE=1" "EVMI_BF_TILE=1 EVMI_BF_NST=3 EVMI_BF_PS=8" "EVMI_F32_WP=0"; do
  echo "== $cfg"
  env $cfg python tools/bench_f32conv.py 2>&1 | tail -5 | cut -c1-70
done
