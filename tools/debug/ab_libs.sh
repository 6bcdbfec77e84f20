for i in 1 2 3; do
for v in old new; do
  EVMI_LIB=$GRAFT_REPO_ROOT/tools/debug/libs/libevmi_$v.so python bench.py --no-train --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); f=d['fs2']; t=d['fs2_train']; print('$v', 'fs2 infer', f['ms_per_batch'], f['other_precision']['ms_per_batch'], 'train', t['ms_per_step'], t['other_precision']['ms_per_step'])"
done; done

