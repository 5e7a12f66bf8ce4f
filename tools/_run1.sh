cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3a
(timeout 300 python tools/lane_offset.py resnet50 256; timeout 300 python tools/lane_offset.py vit_base 256; timeout 300 python tools/lane_offset.py swin_t 128) > gpurun_out/r3a/lane_offset.log 2>&1
cat gpurun_out/r3a/lane_offset.log
