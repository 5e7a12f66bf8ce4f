#!/bin/bash
# round 6, session B: the layer-1 recompute plan -- parity, kernels alone, in-process A/B on resnet50
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r6b; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -m gpu -k "chain or dual or resnet or jitted_training" > $O/pytest.log 2>&1
tail -5 $O/pytest.log
timeout 300 python tools/time_chain_rc.py 128 > $O/time_chain_rc.txt 2>&1; cat $O/time_chain_rc.txt
timeout 300 python tools/ab_flag.py no_chain_rc resnet50 256 3 > $O/ab_no_chain_rc.txt 2>&1; cat $O/ab_no_chain_rc.txt
timeout 300 python tools/ab_flag.py no_chain_sub resnet50 256 3 > $O/ab_no_chain_sub.txt 2>&1; cat $O/ab_no_chain_sub.txt
