#!/bin/bash
# round 6, final check: smoke(), the whole GPU suite, the default bench line
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r6n; mkdir -p $O
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -2 $O/smoke.log
timeout 2700 python -m pytest tests -q -x -m gpu > $O/pytest_gpu.log 2>&1; grep -E "passed|failed" $O/pytest_gpu.log | tail -2
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err; python - <<'PY'
import json
d=json.loads(open("gpurun_out/r6n/bench_default.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["roofline"]["whole_forward"]["frac"], d.get("lanes_not_joined", {}).get("value"), {k:v["value"] for k,v in d["extra"].items()})
PY
