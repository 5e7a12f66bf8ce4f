#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r6p; mkdir -p $O
timeout 300 python bench.py --model vit_base --batch 256 --steps 30 --warmup 5 --no-cpu --no-lanes1 --extra none --layers $O/layers_fold.txt > $O/bench_fold.json 2>$O/bench_fold.err; tail -1 $O/bench_fold.json | cut -c1-300
head -24 $O/layers_fold.txt
python - <<'PY'
import sys; sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import numpy as np, torch
import _model_cases as MC
from eqxvision_amd import _lib
import eqxvision_amd as eqv
from oracle import state as S
sd = S.vit_state(1, 224, 16, 768, 2, 12, 4, 1000)
fac = lambda torch_weights=None, **kw: eqv.utils.load_torch_weights(eqv.models.VisionTransformer(**kw), torch_weights)
net = MC._load(fac, sd, img_size=224, patch_size=16, embed_dim=768, depth=2, num_heads=12, num_classes=1000)
x = np.tile(np.asarray(S.synthetic_images(8,224,seed=3)), (8,1,1,1))
rec=[]; old=_lib.set_recording(rec)
out = MC._run(net, x, "bf16")
_lib.set_recording(old)
print([r[-1] if isinstance(r,(tuple,list)) else r for r in rec])
print(rec[0])
PY
