cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3a
(echo "== lanes 2"; LANES=2 timeout 300 python tools/ab_flag.py no_bneck_tail resnet50 256 3; echo "== lanes 1"; LANES=1 timeout 300 python tools/ab_flag.py no_bneck_tail resnet50 256 3) > gpurun_out/r3a/ab.log 2>&1
cat gpurun_out/r3a/ab.log
