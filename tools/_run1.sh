cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 1500 python tools/gpu_check.py bwd/ grad/ > gpurun_out/bwd_check.log 2>&1; grep -c PASS gpurun_out/bwd_check.log; grep -v PASS gpurun_out/bwd_check.log | cut -c1-500 | tail -8
for m in "vit_tiny 8" "vit_base 4" "resnet18 8"; do timeout 600 python tools/time_train_step.py $m 3 2>&1 | grep -v "amdgpu.ids\|Warning\|warn" | tail -1; done | tee gpurun_out/train_step_times5.txt
