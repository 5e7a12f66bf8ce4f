#!/bin/bash
# round 6, session K: chain_res (the stage's last boundary in the accumulator-layout style)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r6k; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_parity.py -q -x -m gpu -k "chain or resnet50" > $O/pytest.log 2>&1; tail -3 $O/pytest.log
timeout 300 python tools/time_chain_rc.py 128 2>&1 | grep -v amdgpu.ids | tee $O/time_chain_rc.txt
timeout 300 python tools/ab_flag.py no_chain_res resnet50 256 3 2>&1 | grep -v amdgpu.ids | tee $O/ab_no_chain_res.txt
