#!/bin/bash
# round 6, session A: L2-resident stream probe, the new parity cases, the round's baseline line on this box
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r6a; mkdir -p $O
timeout 300 tools/ubench/l2_stream > $O/l2_stream.txt 2>&1
timeout 1500 python -m pytest tests/test_gpu_parity.py -q -x -m gpu -k "fp32 or factory_swin or jitted_training or device_status or make_step or committed" > $O/pytest_new.log 2>&1
tail -5 $O/pytest_new.log
timeout 600 python bench.py --layers $O/layers_resnet50.txt > $O/bench_default.json 2> $O/bench_default.err
tail -c 1500 $O/bench_default.json
