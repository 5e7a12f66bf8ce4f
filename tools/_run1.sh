cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r04_s10; mkdir -p $O
timeout 1500 python tools/gpu_check.py grad/swin > $O/check.log 2>&1; grep -E "PASS|FAIL|passed" $O/check.log | cut -c1-420
timeout 600 python -m pytest tests/test_c_host.py -m gpu -x -q 2>&1 | tail -25
