"""The expanding 1x1 layer of a ResNet bottleneck with residual + ReLU (conv3: C -> 4C, resnet.py:144-162) through mv_linear_fwd under
different kernel choices.  usage: time_expand.py [M] [C]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from gemm_sweep_lib import run
M = int(sys.argv[1]) if len(sys.argv) > 1 else 25088
C = int(sys.argv[2]) if len(sys.argv) > 2 else 256
N = 4 * C
by = 2.0 * (M * C + 2 * M * N + N * C)
for flags in ((), ((f"ov:{M}:{C}:{N}:1:1:1", 10),), ((f"ov:{M}:{C}:{N}:1:1:1", 11),), ((f"ov:{M}:{C}:{N}:1:1:1", 12),),
              ((f"ov:{M}:{C}:{N}:1:1:1", 3),), ((f"ov:{M}:{C}:{N}:1:1:1", 7),)):
    try:
        us, k = run(M, N, C, act=1, res=True, flags=flags)
        print(f"M{M} {C}->{N} +res relu {dict(flags)} [{k}]: {us:.1f} us  {by/us/1e3:.0f} GB/s  {2.0*M*N*C/us/1e6:.0f} TF")
    except Exception as e:
        print(flags, "failed:", str(e)[:100])
