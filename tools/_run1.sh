cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r3m; mkdir -p $O
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/alex_t -o t -- python bench.py --model alexnet --batch 256 --steps 20 --warmup 5 --no-cpu --extra none --soak 1 --lanes 1 > $O/alex.log 2>&1
t=$(find $O/alex_t -name "*kernel_trace.csv" | head -1); python tools/rocprof_trim.py $t $O/alexnet_lanes1_warm.txt; head -16 $O/alexnet_lanes1_warm.txt | cut -c1-150
find $O -size +3M -delete
