export OPERANDS=bf16
GRAPH=1 python tools/train_bench.py 30 2>&1 | tail -2
bash tools/prof_fs2_train.sh r04x_fs2
f=$(ls gpurun_out/r04x_fs2/*kernel_trace.csv | head -1)
python tools/trace_gaps.py $f optimizer_step 2 > gpurun_out/r04x_fs2_gaps.txt 2>&1
head -24 gpurun_out/r04x_fs2_gaps.txt
python tools/trace_concurrency.py $f 2>&1 | tail -25
python tools/trace_exclusive.py $f > gpurun_out/r04x_fs2_exclusive.txt 2>&1
