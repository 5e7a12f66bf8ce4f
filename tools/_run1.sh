cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3a
timeout 1200 python tools/gpu_check.py golden/ trained_scale > gpurun_out/r3a/check.log 2>&1
cat gpurun_out/r3a/check.log | cut -c1-420
