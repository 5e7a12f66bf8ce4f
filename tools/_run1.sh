cd $GRAFT_REPO_ROOT
SBA_C=96 bash tools/pmc_kernel.sh gpurun_out/w96/pmc_new swin_win96 python tools/time_swin_block_attn.py 64 | tail -40
SBA_C=96 FLAGS=swin_c96_shared bash tools/pmc_kernel.sh gpurun_out/w96/pmc_old "swin_block_attn_kernel<96" python tools/time_swin_block_attn.py 64 | tail -40
