"""GEMM sweep through mv_linear_fwd: time vs K for fixed M,N -> per-iteration slope and per-block intercept."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from gemm_sweep_lib import run

M = 50432
for flags in ((("igemm2_tile", 2),), (("igemm2_tile", 3),), (("no_igemm2", 1),)):
    for N in (768, 2304, 3072):
        row = []
        for K in (256, 768, 1536, 3072, 6144):
            us, k = run(M, N, K, flags=flags)
            row.append((K, us, 2.0 * M * N * K / us / 1e6))
        print(k, "N", N, " ".join(f"K{K}:{us:7.1f}us/{tf:6.1f}TF" for K, us, tf in row), flush=True)
for act, res, f32 in ((2, False, False), (0, True, False), (0, True, True)):
    us, k = run(M, 3072 if act else 768, 768 if act else 3072, act, res, f32, flags=(("igemm2_tile", 2),))
    print(k, "act", act, "res", res, "f32", f32, f"{us:7.1f}us")
