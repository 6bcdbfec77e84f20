export F32_LAYERS="${LAYERS:-MSD L6,MPD p2 L4,MPD p2 L3,MPD p11 L3,MPD p11 L4}"
for cfg in "" "EVMI_PK_TILE=0" "EVMI_PK_TILE=1" "EVMI_PK_TILE=1 EVMI_PK_KBS=8 EVMI_PK_NST=2" "EVMI_PK_TILE=0 EVMI_PK_KBS=4 EVMI_PK_NST=2" "EVMI_PK_TILE=2 EVMI_PK_KBS=16 EVMI_PK_NST=2"; do
  echo "== $cfg"
  env $cfg python tools/bench_f32conv.py 2>&1 | tail -5 | cut -c1-78
done
