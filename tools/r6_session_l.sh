#!/bin/bash
# round 6, session L: filter_jit(join="stream") -- the GPU case, and the default bench line with the lanes-not-joined figure beside it
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r6l; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -m gpu -k "filter_jit or lanes or jit" > $O/pytest.log 2>&1; tail -3 $O/pytest.log
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err; python - <<'PY'
import json
d=json.loads(open("gpurun_out/r6l/bench_default.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d.get("lanes_not_joined"))
for k,v in d["extra"].items(): print(k, v.get("value"), v.get("lanes_not_joined", {}).get("value"), v.get("lanes_not_joined", {}).get("logits_bit_identical_to_the_joined_forward"), v.get("error"))
PY
tail -3 $O/bench_default.err
