bash tools/pmc_kernels.sh r05d_attn 'attention_train' -- python3 $GRAFT_REPO_ROOT/tools/microbench/attention_bench.py 5 2>&1 | tail -60
bash tools/pmc_kernels.sh r05d_pk 'conv_pk_kernel|wgrad_pk_kernel' -- python3 $GRAFT_REPO_ROOT/tools/pkflat_bench.py 3 2>&1 | tail -80
