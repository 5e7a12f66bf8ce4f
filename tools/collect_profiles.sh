#!/bin/bash
# Copy ONE profile session (gpurun_out/TAG, written by tools/profile_session.sh TAG ROUND) into profiles/ROUND and profiles/traffic.json
# in the same step, so that traffic.json can never cite files of another session (tests/test_profiles.py checks exactly that).
# usage: tools/collect_profiles.sh TAG ROUND
TAG=$1; ROUND=$2; S=gpurun_out/$TAG; D=profiles/$ROUND
[ -f $S/traffic.json ] || { echo "no $S/traffic.json"; exit 1; }
mkdir -p $D
for M in resnet50 vit_base swin_t alexnet; do
  for f in bench.json bench_layers.json per_launch.txt rocprofv3_warm_stats.txt rocprofv3_kernel_stats.csv lanes1_bench.json \
           lanes1_rocprofv3_warm_stats.txt pmc_sq.txt graph_timeline_under_rocprofv3.txt; do
    [ -f $S/${M}_$f ] && cp $S/${M}_$f $D/${M}_$f
  done
done
for f in roofline_vs_rocprof.txt hbm_traffic_pmc.txt bench_default_line.json; do [ -f $S/$f ] && cp $S/$f $D/$f; done
cp $S/traffic.json profiles/traffic.json
python -m pytest tests/test_profiles.py -q 2>&1 | tail -2
