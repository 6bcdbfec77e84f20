for i in 1 2; do
EVMI_FS2_RESDROP=0 OPERANDS=bf16 timeout 600 python tools/fs2_train_bench.py 60 2>&1 | tail -1 | cut -c1-60
EVMI_FS2_RESDROP=1 OPERANDS=bf16 timeout 600 python tools/fs2_train_bench.py 60 2>&1 | tail -1 | cut -c1-60
done
