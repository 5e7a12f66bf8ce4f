#!/bin/bash
# round 6, session D: three same-box A/Bs -- conv3x3c64 fragment prefetch two taps ahead (default) vs one (_c3pf2), igemm8 lin_f32out
# with the first residual rows fetched before the main loop (_early) vs in the epilogue (default), the layer-1 recompute plan on / off
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r6d; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_parity.py -q -x -m gpu -k "conv or resnet" > $O/pytest_default.log 2>&1; tail -2 $O/pytest_default.log
EQV_LIB=$GRAFT_REPO_ROOT/eqxvision_amd/csrc/libeqxvision_amd_early.so timeout 600 python -m pytest tests/test_gpu_parity.py -q -x -m gpu -k "linear or vit or swin_t_B1 or head" > $O/pytest_early.log 2>&1; tail -2 $O/pytest_early.log
bash tools/ab_lib.sh _c3pf2 resnet50 256 conv3x3c64 2>&1 | grep -v amdgpu.ids | tee $O/ab_c3pf2_resnet50.txt
bash tools/ab_lib.sh _early vit_base 256 lin_f32out 2>&1 | grep -v amdgpu.ids | tee $O/ab_early_vit_base.txt
timeout 300 python tools/ab_flag.py no_chain_rc resnet50 256 3 2>&1 | grep -v amdgpu.ids | tee $O/ab_no_chain_rc.txt
timeout 300 python tools/time_chain_rc.py 128 2>&1 | grep -v amdgpu.ids | tee $O/time_chain_rc.txt
