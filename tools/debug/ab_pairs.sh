python -m pytest tests/test_gpu_generator.py -q -x 2>&1 | tail -2
EVMI_PAIR_OVL=3 python -m pytest tests/test_gpu_generator.py -q -x 2>&1 | tail -2
for cfg in "EVMI_PAIR_OVL=0" "EVMI_PAIR_OVL=1" "EVMI_PAIR_OVL=3" "EVMI_PAIR_OVL=2" "EVMI_PAIR_C128=0" "EVMI_PAIR_OVL=0" "EVMI_PAIR_OVL=1"; do
  echo "== $cfg"
  env $cfg python bench.py --no-train --no-fs2 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step']); print({k:v for k,v in d['roofline']['whole_forward']['by_kernel_ms'].items() if 'pair' in k})"
done
