#!/bin/bash
# round 6, session C: same-box A/B of the chain1x1 loop with both tiles unconditional (default lib) against round 5's conditional
# second tile (libeqxvision_amd_condb.so = chain1x1.hip compiled with -DMV_CHAIN_COND_B, other objects shared); new model cases
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r6c; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_parity.py -q -x -m gpu -k "recompute_plan or chain" > $O/pytest.log 2>&1; tail -3 $O/pytest.log
bash tools/ab_lib.sh _condb resnet50 256 chain1x1 2>&1 | grep -v amdgpu.ids | tee $O/ab_condb_resnet50.txt
