cd $GRAFT_REPO_ROOT
python tools/profile_layers.py > gpurun_out/layers_r1a.txt 2>&1
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > $GRAFT_REPO_ROOT/gpurun_out/counters_list.txt 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_r1a -o r1a -- python3 $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 1 --no-cpu-baseline --profile-passes 1 > $GRAFT_REPO_ROOT/gpurun_out/prof_r1a.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_LDS_BANK_CONFLICT --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_r1a -o pmc1 -- python3 $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline --profile-passes 1 > $GRAFT_REPO_ROOT/gpurun_out/pmc_r1a.log 2>&1
ls -R $GRAFT_REPO_ROOT/gpurun_out | head -40
