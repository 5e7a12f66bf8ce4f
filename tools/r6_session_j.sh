#!/bin/bash
# round 6, session J: the whole GPU suite + the default bench line
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r6j; mkdir -p $O
timeout 2400 python -m pytest tests -q -x -m gpu > $O/pytest_gpu.log 2>&1; tail -4 $O/pytest_gpu.log
timeout 600 python bench.py > $O/bench_default.json 2> $O/bench_default.err; python - <<'PY'
import json
d=json.loads(open("gpurun_out/r6j/bench_default.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], {k:v["value"] for k,v in d["extra"].items()})
PY
