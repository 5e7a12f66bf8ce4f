cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3k
timeout 900 python tools/gpu_check.py fc_stream/ model/alexnet model/vgg > gpurun_out/r3k/check.log 2>&1
grep -c PASS gpurun_out/r3k/check.log; grep "FAIL" gpurun_out/r3k/check.log | cut -c1-400; grep "fc_stream" gpurun_out/r3k/check.log | cut -c1-220
timeout 300 python bench.py --model alexnet --batch 256 --steps 30 --warmup 5 --no-cpu --extra none --soak 1 --no-lanes1 --layers gpurun_out/r3k/alexnet_per_launch.txt > gpurun_out/r3k/alexnet.json 2> gpurun_out/r3k/alexnet.err
python -c "import json; d=json.load(open('gpurun_out/r3k/alexnet.json')); print(d['value'], d['ms_per_step'])"; tail -3 gpurun_out/r3k/alexnet.err
head -14 gpurun_out/r3k/alexnet_per_launch.txt | cut -c1-150
(LANES=2 timeout 300 python tools/ab_flag.py no_fc_stream alexnet 256 3) 2>&1 | tail -6
(LANES=2 timeout 300 python tools/ab_flag.py no_fc_stream vgg16 128 2) 2>&1 | tail -4
