export OPERANDS=bf16
python tools/scratch/side_det.py 2>&1 | grep -v amdgpu
for g in 1 8 16; do echo "group $g"; EVMI_SIDE_GROUP=$g python tools/fs2_train_bench.py 20 2>&1 | grep "^step"; done
