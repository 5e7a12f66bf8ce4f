"""Tile-choice sweep over the Swin-T linear shapes (mv_linear_fwd), default dispatch vs forced igemm2 tiles."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from gemm_sweep_lib import run

SHAPES = [  # M, K, N, act, res, f32
    (401408, 384, 96, 0, True, True), (100352, 768, 192, 0, True, True), (100352, 384, 192, 0, False, False),
    (25088, 384, 1152, 0, False, False), (25088, 384, 384, 0, True, True), (25088, 384, 1536, 2, False, False),
    (25088, 1536, 384, 0, True, True), (25088, 768, 384, 0, False, False),
    (6272, 768, 2304, 0, False, False), (6272, 768, 768, 0, True, True), (6272, 768, 3072, 2, False, False),
    (6272, 3072, 768, 0, True, True), (6272, 1536, 768, 0, False, False),
]
for M, K, N, act, res, f32 in SHAPES:
    out = []
    for flags in ((), (("igemm2_tile", 1),), (("igemm2_tile", 2),), (("no_igemm2", 1),), (("no_igemm2", 1), ("igemm_tile", 64))):
        us, k = run(M, N, K, act, res, f32, flags=flags)
        out.append(f"{k.replace('_bf16','').replace('_dense','')}:{us:6.1f}us/{2.0*M*N*K/us/1e6:4.0f}TF")
    print(f"M{M} K{K} N{N} act{act} res{int(res)} f32{int(f32)} | " + " | ".join(out), flush=True)
