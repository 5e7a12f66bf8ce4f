cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 3000 python -m pytest tests -m gpu -x -q 2>&1 | tail -8 | tee gpurun_out/pytest_gpu.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3 | tee gpurun_out/smoke.txt
