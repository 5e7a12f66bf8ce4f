cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3a
BNECK_2STREAM=1 timeout 300 python tools/time_bneck.py 128 256 > gpurun_out/r3a/time_bneck.log 2>&1
cat gpurun_out/r3a/time_bneck.log
