#!/bin/bash
# round 6, session E: chain_rc0 (the first layer-1 boundary without its output map, 12 waves) -- parity, alone, in-process A/B
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r6e; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_parity.py -q -x -m gpu -k "chain or resnet50" > $O/pytest.log 2>&1; tail -2 $O/pytest.log
timeout 300 python tools/time_chain_rc.py 128 2>&1 | grep -v amdgpu.ids | tee $O/time_chain_rc.txt
timeout 300 python tools/ab_flag.py no_chain_rc0 resnet50 256 3 2>&1 | grep -v amdgpu.ids | tee $O/ab_no_chain_rc0.txt
FLAGVAL=8 timeout 300 python tools/ab_flag.py chain_rc0_waves resnet50 256 3 2>&1 | grep -v amdgpu.ids | tee $O/ab_rc0_waves8.txt
timeout 300 python tools/ab_flag.py no_chain_rc resnet50 256 3 2>&1 | grep -v amdgpu.ids | tee $O/ab_no_chain_rc.txt
