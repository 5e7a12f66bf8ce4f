#!/bin/bash
# round 6, session Q: the LayerNorm fold in the tree -- whole GPU suite, default bench line
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r6q; mkdir -p $O
timeout 2400 python -m pytest tests -q -x -m gpu > $O/pytest_gpu.log 2>&1; tail -6 $O/pytest_gpu.log
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err; python - <<'PY'
import json
d=json.loads(open("gpurun_out/r6q/bench_default.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], {k: v.get("value") for k, v in d.get("extra_models", {}).items()} if isinstance(d.get("extra_models"), dict) else "")
print({k: d[k] for k in d if k not in ("config","roofline","cpu_baseline","extra_models")})
PY
