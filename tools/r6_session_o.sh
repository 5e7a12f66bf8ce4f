#!/bin/bash
# round 6, session O: the GEMM cores' bf16 epilogue staged in bf16 (default) vs the fp32 patch (lib_f32patch) -- parity, then same-box A/Bs
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r6o; mkdir -p $O
timeout 2400 python -m pytest tests/test_gpu_parity.py -q -x -m gpu -k "not grad and not bwd" > $O/pytest.log 2>&1; grep -E "passed|failed" $O/pytest.log | tail -2; grep -E "^FAILED" $O/pytest.log | head
bash tools/ab_lib.sh _f32patch vit_base 256 "256x256_lin " 2>&1 | grep -v amdgpu.ids | tee $O/ab_vit_base.txt
bash tools/ab_lib.sh _f32patch resnet50 256 "igemm8_bf16_256x256_dense" 2>&1 | grep -v amdgpu.ids | tee $O/ab_resnet50.txt
bash tools/ab_lib.sh _f32patch swin_t 128 "igemm8" 2>&1 | grep -v amdgpu.ids | tee $O/ab_swin_t.txt
