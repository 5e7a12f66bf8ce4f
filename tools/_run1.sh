cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r04_full; mkdir -p $O
timeout 3000 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; tail -5 $O/pytest_gpu.log
