cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 300 python tools/lane_offset.py swin_t 128 2>&1 | tail -6
timeout 300 python tools/lane_offset.py alexnet 256 2>&1 | tail -6
