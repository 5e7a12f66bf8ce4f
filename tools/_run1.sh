cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 300 python tools/time_fc_stream.py 128 2>&1 | tail -4
