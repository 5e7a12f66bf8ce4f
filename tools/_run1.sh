cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 1500 python tools/gpu_check.py grad/ > gpurun_out/grad_check.log 2>&1; grep -c PASS gpurun_out/grad_check.log; grep -v PASS gpurun_out/grad_check.log | cut -c1-300 | tail -5; grep -o "grad/[a-z0-9_A-Z]*  {\"s\": [0-9.]*, \"err\": [0-9.e-]*, \"rel_l2\": [0-9.e-]*, \"tensors_over_lim\": [0-9]*" gpurun_out/grad_check.log
timeout 1500 python -m pytest tests -m gpu -q --deselect "tests/test_gpu_parity.py::test_grad" 2>&1 | tail -3
