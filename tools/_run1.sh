cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r04_s7; mkdir -p $O
timeout 1200 python tools/gpu_check.py linear/ layernorm igemm8 conv/ stream/ dual/ model/vit_tiny model/resnet50_B2 model/swin_t_B1 golden/ > $O/check.log 2>&1; grep -c PASS $O/check.log; grep FAIL $O/check.log | cut -c1-300
timeout 400 python bench.py --no-cpu > $O/bench_default.json 2> $O/bench_default.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r04_s7/bench_default.json').read().strip().splitlines()[-1])
print('resnet50', d['value'], d['ms_per_step'])
for k,v in d.get('extra',{}).items(): print(k, v.get('value'), v.get('ms_per_step'))
PY
tail -3 $O/bench_default.err
