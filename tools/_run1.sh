cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/tune
for m in "resnet50 256" "swin_t 128"; do
  timeout 1500 python tools/tune_tiles.py $m 2 2>&1 | grep -v amdgpu.ids | tee gpurun_out/tune/${m%% *}.txt | tail -25
done
