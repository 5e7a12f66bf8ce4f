#!/bin/bash
# usage: ab_lib.sh LIBSUFFIX MODEL BATCH KERNELPAT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
S=$1; M=$2; B=$3; K=$4
for rep in 1 2; do
  for v in "" $S; do
    EQV_LIB=$GRAFT_REPO_ROOT/eqxvision_amd/csrc/libeqxvision_amd$v.so timeout 300 python bench.py --model $M --batch $B --steps 30 --warmup 5 --no-cpu --no-lanes1 --extra none --layers /tmp/l$v$rep.txt > /tmp/b$v$rep.json 2>/dev/null
    python - <<PY
import json
d=json.loads(open("/tmp/b$v$rep.json").read().strip().splitlines()[-1])
ls=[l for l in open("/tmp/l$v$rep.txt") if "$K" in l]
us=[float(l.split()[-6]) for l in ls]
print("lib$v rep$rep: %.0f img/s  %.4f ms   $K alone: %s us" % (d["value"], d["ms_per_step"], " ".join("%.1f"%u for u in us[:4])))
PY
  done
done
