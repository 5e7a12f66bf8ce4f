cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r04_s9; mkdir -p $O
timeout 1500 python tools/gpu_check.py grad/ > $O/check_grad.log 2>&1; cat $O/check_grad.log | cut -c1-1500
