cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3a
timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/r3a/bench_full.json 2> gpurun_out/r3a/bench_full.err
tail -3 gpurun_out/r3a/bench_full.err; python -c "
import json
d=json.load(open('gpurun_out/r3a/bench_full.json'))
print(json.dumps({k:v for k,v in d.items() if k!='extra'}, indent=1)[:3500])
for m,r in d.get('extra',{}).items(): print(m, json.dumps(r)[:1800])
"
