cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r3b; mkdir -p $O
for ord in lane interleave; do
for spec in swin_t:128 vit_base:256; do
  M=${spec%%:*}; B=${spec##*:}
  EQV_INSITU_ORDER=$ord timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${M}_$ord -o t -- python bench.py --model $M --batch $B --steps 20 --warmup 5 --no-cpu --extra none --soak 1 --no-lanes1 > $O/${M}_$ord.log 2>&1
  grep '^{' $O/${M}_$ord.log | tail -1 | python -c "import json,sys; r=json.loads(sys.stdin.read())['roofline']; print('$M $ord', r['kernel'], r['avg_launch_us'], r['avg_launch_us_eager_all_events'])"
  t=$(find $O/${M}_$ord -name "*kernel_trace.csv" | head -1); python tools/rocprof_trim.py $t $O/${M}_${ord}_warm.txt; head -8 $O/${M}_${ord}_warm.txt | cut -c1-130
done; done
find $O -size +3M -delete
