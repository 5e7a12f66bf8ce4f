#!/bin/bash
# round 6, session R: the tree with the ViT LayerNorm fold -- whole GPU suite, then the round's profile session again (all four models)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r6r; mkdir -p $O
timeout 2400 python -m pytest tests -q -x -m gpu > $O/pytest_gpu.log 2>&1; tail -4 $O/pytest_gpu.log
bash tools/profile_session.sh r6prof2 r06 > gpurun_out/r6prof2.log 2>&1; tail -5 gpurun_out/r6prof2.log
timeout 900 python bench.py > gpurun_out/r6prof2/bench_default_line.json 2> $O/bench_default.err
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r6prof2/bench_default_line.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], {k: v.get("value") for k, v in d["extra"].items()})
PY
