#!/bin/bash
# whole-model benches (3 models x lanes 1/2) + optional per-launch worksheets.  usage: tools/models_session.sh TAG
TAG=${1:-run}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/$TAG; mkdir -p $O
for L in 2 1; do
timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu --lanes $L --layers $O/layers_resnet50_l$L.txt > $O/bench_resnet50_l$L.json 2> $O/bench_resnet50_l$L.err
timeout 300 python bench.py --model vit_base --steps 10 --warmup 3 --no-cpu --lanes $L --layers $O/layers_vit_l$L.txt > $O/bench_vit_l$L.json 2> $O/bench_vit_l$L.err
timeout 300 python bench.py --model swin_t --batch 128 --steps 10 --warmup 3 --no-cpu --lanes $L --layers $O/layers_swin_l$L.txt > $O/bench_swin_l$L.json 2> $O/bench_swin_l$L.err
done
for f in $O/bench_*.json; do echo $f; python -c "import json,sys; d=json.load(open('$f')); print(d['value'], d['ms_per_step'], d['config']['launches_per_step'])" 2>/dev/null || tail -3 ${f%.json}.err; done
