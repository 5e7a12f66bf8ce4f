"""Run one mv_linear_fwd shape a few times (for rocprofv3 --pmc): usage one_gemm.py M N K [flag=value ...]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from gemm_sweep_lib import run
M, N, K = (int(a) for a in sys.argv[1:4])
flags = tuple((a.split("=")[0], int(a.split("=")[1])) for a in sys.argv[4:])
us, k = run(M, N, K, flags=flags)
print(f"M{M} N{N} K{K} {k}: {us:.1f} us {2.0*M*N*K/us/1e6:.1f} TF")
