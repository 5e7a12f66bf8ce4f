cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for m in resnet50:256 swin_t:128 alexnet:256 vit_base:256; do
  M=${m%%:*}; B=${m##*:}
  timeout 600 python bench.py --model $M --batch $B --steps 1500 --warmup 20 --no-cpu --extra none --soak 2 --no-lanes1 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$M', d['value'], d['ms_per_step'], d['steps'])"
done
rocm-smi --showmeminfo vram 2>/dev/null | grep -i "used" | head -2
