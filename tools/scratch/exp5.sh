python -m pytest tests/test_gpu_disc_chain.py -q -x 2>&1 | tail -2
python tools/pkflat_bench.py 30 > gpurun_out/r04y_pkflat.txt 2>&1; tail -22 gpurun_out/r04y_pkflat.txt | cut -c1-60,100-140
OPERANDS=bf16 GRAPH=1 python tools/train_bench.py 30 2>&1 | grep "^step"
for g in 1 8; do EVMI_FS2_GRAPH=0 EVMI_SIDE_GROUP=$g OPERANDS=bf16 python tools/fs2_train_bench.py 20 2>&1 | grep "^step" | cut -c1-60; done
