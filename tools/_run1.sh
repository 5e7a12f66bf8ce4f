cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3a
timeout 900 python tools/gpu_check.py swin_block_attn model/swin > gpurun_out/r3a/check.log 2>&1
(for c in 384 192 96; do SBA_PROF=1 SBA_C=$c timeout 300 python tools/time_swin_block_attn.py 64; done) > gpurun_out/r3a/time_sba.log 2>&1
grep -c PASS gpurun_out/r3a/check.log; grep "FAIL" gpurun_out/r3a/check.log | cut -c1-260; grep "fused\|total" gpurun_out/r3a/time_sba.log
