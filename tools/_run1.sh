cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r04_s8; mkdir -p $O
timeout 600 python tools/time_i8h.py 3 > $O/time_i8h.txt 2>&1; cat $O/time_i8h.txt
FLAGVAL=1 timeout 300 python tools/ab_flag.py i8h vit_base 256 3 > $O/ab_i8h_vit.txt 2>&1; cat $O/ab_i8h_vit.txt
