cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3r
timeout 900 python tools/gpu_check.py stem/vit model/vit golden/committed_vit golden/committed_small > gpurun_out/r3r/check.log 2>&1
grep -c PASS gpurun_out/r3r/check.log; grep "FAIL" gpurun_out/r3r/check.log | cut -c1-400; grep "f32out\|full_config\|large_logit" gpurun_out/r3r/check.log | cut -c1-260
(LANES=2 timeout 300 python tools/ab_flag.py no_patch_f32out vit_base 256 3) 2>&1 | tail -6
