cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3e
(LANES=2 timeout 300 python tools/ab_flag.py no_tuned resnet50 256 4) > gpurun_out/r3e/ab_tuned.log 2>&1
timeout 600 python tools/gpu_check.py model/resnet50 golden/committed > gpurun_out/r3e/check.log 2>&1
grep -c PASS gpurun_out/r3e/check.log; grep "FAIL" gpurun_out/r3e/check.log | cut -c1-300; cat gpurun_out/r3e/ab_tuned.log | tail -12
