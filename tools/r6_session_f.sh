#!/bin/bash
# round 6, session F: chain_rc1 with 8 waves (32-channel store patches) vs 6 waves (64-channel patches)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r6f; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_parity.py -q -x -m gpu -k "chain or resnet50_B" > $O/pytest.log 2>&1; tail -2 $O/pytest.log
timeout 300 python tools/time_chain_rc.py 128 2>&1 | grep -v amdgpu.ids | tee $O/time_chain_rc.txt
FLAGVAL=6 timeout 300 python tools/ab_flag.py chain_rc1_waves resnet50 256 3 2>&1 | grep -v amdgpu.ids | tee $O/ab_rc1_waves6.txt
