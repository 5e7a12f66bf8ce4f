cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r04_s2; mkdir -p $O
timeout 900 python tools/mall_probe.py > $O/mall_probe.txt 2>&1
cat $O/mall_probe.txt
