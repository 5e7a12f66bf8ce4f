cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 2500 python -m pytest tests -m gpu -q 2>&1 | grep -E "passed|failed|FAILED" | tail -3
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
( time timeout 600 python bench.py > gpurun_out/bench_default_line.json 2> gpurun_out/bench_default.err ) 2>&1 | grep real
cut -c1-900 gpurun_out/bench_default_line.json
python - <<'PY'
import json
d=json.load(open("gpurun_out/bench_default_line.json"))
print({k:(v.get("value") if isinstance(v,dict) else v) for k,v in d.get("extra",{}).items()} if isinstance(d.get("extra"),dict) else d.get("extra"))
print("roofline", d["roofline"]["frac"], d["roofline"].get("rocprof",{}).get("frac"), "cpu_baseline", d.get("cpu_baseline"))
PY
