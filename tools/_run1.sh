cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 900 python tools/gpu_check.py swin_block_attn/c96 model/swin golden/committed_swin > gpurun_out/w96_check.log 2>&1; grep -c PASS gpurun_out/w96_check.log; grep -v PASS gpurun_out/w96_check.log | cut -c1-300 | tail -4
SBA_C=96 timeout 300 python tools/time_swin_block_attn.py 64 2>&1 | grep fused
for a in 0 4 15; do echo -n "w96_ablate=$a: "; EQV_LIB=$GRAFT_REPO_ROOT/eqxvision_amd/csrc/libeqxvision_amd_prof.so FLAGS=w96_ablate=$a SBA_C=96 timeout 300 python tools/time_swin_block_attn.py 64 2>&1 | grep "fused" | cut -c1-60; done
timeout 300 python tools/ab_flag.py swin_c96_shared swin_t 128 2>&1 | tail -6
