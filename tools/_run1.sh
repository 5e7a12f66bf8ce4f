cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for m in "alexnet 8" "resnet18 8" "resnet50 4" "vit_tiny 8" "swin_t 4"; do timeout 600 python tools/time_train_step.py $m 3 2>&1 | grep -v "amdgpu.ids\|Warning\|warn" | tail -1; done | tee gpurun_out/train_step_times.txt
