cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3a
timeout 900 python tools/gpu_check.py ln_mlp/stream_c192 model/swin > gpurun_out/r3a/check.log 2>&1
LMS_C=192 timeout 300 python tools/time_ln_mlp_stream.py > gpurun_out/r3a/time_lms.log 2>&1
(LANES=2 timeout 300 python tools/ab_flag.py no_ln_mlp_stream swin_t 128 2) > gpurun_out/r3a/ab.log 2>&1
grep -c PASS gpurun_out/r3a/check.log; grep "FAIL\|full_config\|c192" gpurun_out/r3a/check.log | cut -c1-260; grep -v variant gpurun_out/r3a/time_lms.log; cat gpurun_out/r3a/ab.log
