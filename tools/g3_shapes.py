import sys, os
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from gemm_sweep_lib import run
for M, N, K in ((4096, 4096, 4096), (8192, 8192, 8192), (8192, 8192, 8064), (50432, 3072, 768), (50432, 3072, 3072), (50432, 3072, 12288)):
    out = []
    for flags in ((("igemm3", 2),), (("igemm2_tile", 3),)):
        us, k = run(M, N, K, flags=flags)
        out.append(f"{k}: {us:8.1f} us {2.0*M*N*K/us/1e6:7.1f} TF")
    print(f"M{M} N{N} K{K} | " + " | ".join(out), flush=True)
