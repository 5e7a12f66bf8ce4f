#!/bin/bash
# round 6, session P: ViT LayerNorms folded into the GEMM epilogues (mv_linear_lnout / lnin, residual stream as two bf16 planes) --
# parity of the pair and the model, the kernels alone, then the same-box A/B against the LayerNorm launches (switch no_ln_fold)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r6p; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -m gpu -k "ln_fold or layernorm_fold" > $O/pytest_fold.log 2>&1; tail -25 $O/pytest_fold.log
timeout 300 python tools/time_ln_fold.py 128 2>&1 | grep -v amdgpu.ids | tee $O/kernels_alone.txt
timeout 600 python tools/ab_flag.py no_ln_fold vit_base 256 4 2>&1 | grep -v amdgpu.ids | tee $O/ab_vit_base.txt
