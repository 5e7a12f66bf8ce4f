cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for i in 1 2; do
for lib in libeqxvision_amd.so libeqxvision_amd_oldstream.so; do
  for m in efficientnet_b0 mobilenet_v2 mobilenet_v3_large regnet_y_400mf; do
    echo -n "$lib $m: "; EQV_LIB=$GRAFT_REPO_ROOT/eqxvision_amd/csrc/$lib timeout 300 python bench.py --model $m --batch 256 --extra none --no-cpu --steps 40 --warmup 8 2>/dev/null | python -c "import sys,json; print(json.loads(sys.stdin.read())['value'])"
  done
done
done | tee gpurun_out/stream_ab.txt
