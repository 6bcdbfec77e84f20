python -m pytest tests/test_gpu_generator.py -q -x 2>&1 | grep -E "passed|failed|Error|error" | tail -3
for i in 1 2; do for v in 0 1; do
  EVMI_PAIR32=$v python bench.py --no-train --no-fs2 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('PAIR32=$v', d['value'], d['ms_per_step'], {k:v for k,v in d['roofline']['whole_forward']['by_kernel_ms'].items() if 'pair' in k})"
done; done
