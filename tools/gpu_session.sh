#!/bin/bash
# One GPU-box session: parity sweep, benches with per-launch worksheets, rocprofv3 kernel stats.
# usage: tools/gpu_session.sh TAG [check-filter]
TAG=${1:-run}; FILT=${2:-}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/$TAG; mkdir -p $O
timeout 900 python tools/gpu_check.py $FILT > $O/check.log 2>&1
cp gpurun_out/gpu_check.json $O/ 2>/dev/null
timeout 300 python bench.py --steps 30 --warmup 5 --layers $O/layers_resnet50.txt > $O/bench_resnet50.json 2> $O/bench_resnet50.err
timeout 300 python bench.py --model vit_base --steps 10 --warmup 3 --no-cpu --layers $O/layers_vit.txt > $O/bench_vit.json 2> $O/bench_vit.err
timeout 300 python bench.py --model swin_t --batch 128 --steps 10 --warmup 3 --no-cpu --layers $O/layers_swin.txt > $O/bench_swin.json 2> $O/bench_swin.err
if [ -n "$PROF" ]; then
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_resnet50 -o r50 -- python bench.py --steps 20 --warmup 5 --no-cpu > $O/prof_resnet50.log 2>&1
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_vit -o vit -- python bench.py --model vit_base --steps 10 --warmup 3 --no-cpu > $O/prof_vit.log 2>&1
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_swin -o swin -- python bench.py --model swin_t --batch 128 --steps 10 --warmup 3 --no-cpu > $O/prof_swin.log 2>&1
  for m in resnet50:r50 vit:vit swin:swin; do d=${m%%:*}; n=${m##*:}; t=$(find $O/prof_$d -name "*kernel_trace.csv" | head -1); [ -n "$t" ] && python tools/rocprof_trim.py $t $O/prof_${d}_warm_stats.txt; done
  find $O -name "*kernel_stats*" | head; find $O -name "*.db" -delete; find $O -name "*kernel_trace*" -size +8M -delete
fi
grep -c PASS $O/check.log; grep FAIL $O/check.log | cut -c1-300
cat $O/bench_resnet50.json $O/bench_vit.json $O/bench_swin.json
tail -2 $O/bench_resnet50.err $O/bench_vit.err
