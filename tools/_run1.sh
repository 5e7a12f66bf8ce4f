cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r3h; mkdir -p $O
for spec in swin_t:128:88 resnet50:256:82; do
  M=${spec%%:*}; r=${spec#*:}; B=${r%%:*}; N=${r##*:}
  timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/${M}_t -o t -- python bench.py --model $M --batch $B --steps 4 --warmup 1 --no-cpu --extra none --soak 0 --no-lanes1 > $O/$M.log 2>&1
  t=$(find $O/${M}_t -name "*kernel_trace.csv" | head -1)
  python tools/graph_timeline.py $t $N 7 > $O/${M}_timeline.txt
  python tools/graph_timeline.py $t $N 0 > $O/${M}_timeline_eager.txt
  head -3 $O/${M}_timeline.txt
done
find $O -size +3M -delete
