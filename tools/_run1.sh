cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
(LANES=2 timeout 300 python tools/ab_flag.py stem_pool11_tpw1 alexnet 256 3) 2>&1 | tail -6
