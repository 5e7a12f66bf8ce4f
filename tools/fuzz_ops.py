#!/usr/bin/env python
"""Randomised shapes through the C ABI with the DEFAULT dispatch, checked against the numpy oracle: convolutions
(any tap count / stride / padding / dilation / residual / activation / output dtype) and linears.  Finds the shapes
where a specialised kernel is chosen but mishandles an edge (ragged rows, channel tails, tiny reductions).
usage: fuzz_ops.py [N_CASES] [SEED]   (MISC=1: attention / pools / fused entries; FAMILY=1: depthwise, grouped, resize, training-mode
moments and dropout; STOCHASTIC=1: attention dropout, window-layout dropout, key splits; LNFOLD=1: the LayerNorm-fold Linear pair)"""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests import _cases as T

n = int(sys.argv[1]) if len(sys.argv) > 1 else 120
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
fails = 0
kernels = {}
flags = tuple(f for f in os.environ.get("FUZZ_FLAGS", "").split(",") if f)      # e.g. FUZZ_FLAGS=igemm8=3 forces a kernel family
cases = T.fuzz_ln_fold_cases(n, seed) if os.environ.get("LNFOLD") else T.fuzz_stochastic_cases(n, seed) if os.environ.get("STOCHASTIC") else T.fuzz_family_cases(n, seed) if os.environ.get("FAMILY") else T.fuzz_misc_cases(n, seed) if os.environ.get("MISC") else T.fuzz_cases(n, seed, mfma_only=bool(os.environ.get("MFMA_ONLY")),
                                                                               flags=flags)
for name, fn in cases:
    try:
        info = fn()
        ok = bool(info["ok"])
        kernels[info.get("kernel", "?")] = kernels.get(info.get("kernel", "?"), 0) + 1
    except Exception as e:  # noqa: BLE001
        ok, info = False, {"error": f"{type(e).__name__}: {e}"}
    if not ok:
        fails += 1
        print("FAIL", name, json.dumps({k: v for k, v in info.items() if k != "tb"})[:300], flush=True)
print(f"{n} shapes, {fails} failures; kernels hit: {json.dumps(kernels)}")
sys.exit(1 if fails else 0)
