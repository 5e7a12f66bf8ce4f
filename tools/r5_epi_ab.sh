#!/bin/bash
# Round 5: A/B of the branch-free igemm8 epilogue (csrc/igemm_pipe.h: epilogue_rows) against the round-4 epilogue, same box.
# The round-4 library is csrc/libeqxvision_amd_r4epi.so: `git show 96090ca:eqxvision_amd/csrc/igemm8.hip` (+ that commit's igemm_pipe.h /
# mfma_common.h) compiled to an object and linked with today's other objects (csrc/build/*.o); it is not kept in the tree.
# tools/ab_lib.sh SUFFIX MODEL BATCH KERNEL is the generic form of this A/B.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r5epi; mkdir -p $O
OLD=$GRAFT_REPO_ROOT/eqxvision_amd/csrc/libeqxvision_amd_r4epi.so
cat > /tmp/gemms.py <<'PY'
import sys, os
sys.path.insert(0, "tools")
from gemm_sweep_lib import run
for name, M, N, K, act, res, f32 in [("qkv", 25216, 2304, 768, 0, False, False), ("fc1 gelu", 25216, 3072, 768, 2, False, False),
                                     ("proj f32+res", 25216, 768, 768, 0, True, True), ("fc2 f32+res", 25216, 768, 3072, 0, True, True),
                                     ("qkv B256", 50432, 2304, 768, 0, False, False), ("fc2 B256", 50432, 768, 3072, 0, True, True),
                                     ("r50 l3 conv1", 25088, 256, 1024, 1, False, False), ("r50 l4 conv3", 6272, 2048, 512, 1, True, False)]:
    us, k = run(M, N, K, act=act, res=res, f32=f32)
    print(f"{name:14s} M{M} N{N} K{K} {k}: {us:.1f} us {2.0*M*N*K/us/1e6:.1f} TF", flush=True)
PY
grep -n "MV_ACT_GELU_TANH" include/eqxvision_amd.h | head -2
for rep in 1 2; do
  echo "--- new epilogue (rep $rep)"; timeout 300 python /tmp/gemms.py 2>&1 | grep -v amdgpu.ids
  echo "--- round-4 epilogue (rep $rep)"; EQV_LIB=$OLD timeout 300 python /tmp/gemms.py 2>&1 | grep -v amdgpu.ids
done > $O/gemm_alone.txt 2>&1
cat $O/gemm_alone.txt
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; tail -3 $O/pytest_gpu.log
for m in vit_base:256 resnet50:256 swin_t:128 alexnet:256; do
  n=${m%%:*}; b=${m##*:}
  for rep in 1 2; do
    timeout 300 python bench.py --model $n --batch $b --steps 30 --warmup 5 --no-cpu --no-lanes1 --extra none > $O/bench_${n}_new$rep.json 2> $O/bench_${n}_new$rep.err
    EQV_LIB=$OLD timeout 300 python bench.py --model $n --batch $b --steps 30 --warmup 5 --no-cpu --no-lanes1 --extra none > $O/bench_${n}_old$rep.json 2> $O/bench_${n}_old$rep.err
  done
  for v in new1 old1 new2 old2; do python - <<PY
import json
try:
    d=json.loads(open("$O/bench_${n}_$v.json").read().strip().splitlines()[-1]); print("$n $v", d["value"], d["ms_per_step"])
except Exception as e: print("$n $v failed", e)
PY
  done
done 2>&1 | tee $O/ab_summary.txt
