#!/bin/bash
# round 6, session G: chain_rc v2 (shifts through the matrix pipe, scales folded, identity as the C operand, packed ReLU)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r6g; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_parity.py -q -x -m gpu -k "chain or resnet50" > $O/pytest.log 2>&1; tail -3 $O/pytest.log
timeout 300 python tools/time_chain_rc.py 128 2>&1 | grep -v amdgpu.ids | tee $O/time_chain_rc.txt
timeout 300 python tools/ab_flag.py no_chain_rc resnet50 256 3 2>&1 | grep -v amdgpu.ids | tee $O/ab_no_chain_rc.txt
