cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 900 python tools/gpu_check.py swin_block_attn/c384 > gpurun_out/w96_check.log 2>&1; grep -c PASS gpurun_out/w96_check.log; grep -v PASS gpurun_out/w96_check.log | cut -c1-300 | tail -4
SBA_C=384 timeout 300 python tools/time_swin_block_attn.py 64 2>&1 | grep fused
