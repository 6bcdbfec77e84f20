for cfg in "" "EVMI_WG_MIN_STEPS=16" "EVMI_WG_MIN_STEPS=4" "EVMI_WG_WANT=256" "EVMI_WG_NST=2"; do
  echo "== $cfg"
  env $cfg python tools/bench_wgrad_bf16.py 2>&1 | tail -14 | cut -c1-80
done
