cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/sk
timeout 900 python tools/gpu_check.py splitk igemm8/ linear_split > gpurun_out/sk/check.log 2>&1; grep -c PASS gpurun_out/sk/check.log; grep -v PASS gpurun_out/sk/check.log | cut -c1-400 | tail -20
timeout 600 python tools/time_splitk.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/sk/time_splitk.txt
for M in resnet50:256 swin_t:128; do
  timeout 300 python tools/ab_flag.py no_splitk ${M%%:*} ${M##*:} 2>&1 | tail -6 | tee -a gpurun_out/sk/ab.txt
done
