cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r04_s5; mkdir -p $O
SKEWS=0,500,1000,1500 EQV_LIB=$PWD/eqxvision_amd/csrc/libeqxvision_amd_prof.so timeout 300 python tools/time_bneck_strip.py 128 > $O/time_strip_prof.txt 2>&1; cat $O/time_strip_prof.txt
SKEWS=0,500,1000,1500 timeout 300 python tools/time_bneck_strip.py 128 > $O/time_strip.txt 2>&1; cat $O/time_strip.txt
FLAGVAL=1000 timeout 300 python tools/ab_flag.py strip_skew resnet50 256 3 > $O/ab_skew.txt 2>&1; cat $O/ab_skew.txt
