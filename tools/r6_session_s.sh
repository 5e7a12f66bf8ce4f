#!/bin/bash
# round 6, session S: more shapes of the LayerNorm-fold pair; lane counts of vit_base with the fold; parity margins of the final tree
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r6s; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "ln_fold" > $O/pytest_fold.log 2>&1; tail -8 $O/pytest_fold.log
for L in 1 2 3; do
  timeout 300 python bench.py --model vit_base --batch 256 --steps 40 --warmup 5 --no-cpu --no-lanes1 --extra none --lanes $L --no-pipelined 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('lanes $L', d['value'], d['ms_per_step'])"
done | tee $O/vit_lanes.txt
timeout 1500 python tools/parity_margins.py 2>&1 | grep -v amdgpu.ids | tee $O/parity_margins.txt | tail -40
