cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3a
timeout 900 python tools/gpu_check.py patch4_ln model/swin golden/committed_swin golden/committed_small > gpurun_out/r3a/check.log 2>&1
(LANES=2 timeout 300 python tools/ab_flag.py no_patch4_ln swin_t 128 3) > gpurun_out/r3a/ab.log 2>&1
grep -c PASS gpurun_out/r3a/check.log; grep "FAIL\|full_config\|patch4\|golden" gpurun_out/r3a/check.log | cut -c1-300; cat gpurun_out/r3a/ab.log
