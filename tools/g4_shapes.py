"""igemm4 (one wave per SIMD, 128x128 per wave) against the default dispatch on the GEMM shapes of the three models."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from gemm_sweep_lib import run
T = 197 * 256
shapes = [("vit qkv", T, 2304, 768, dict()), ("vit fc1 gelu", T, 3072, 768, dict(act=2)),
          ("vit fc2 res f32", T, 768, 3072, dict(res=True, f32=True)), ("vit proj res f32", T, 768, 768, dict(res=True, f32=True)),
          ("vit half fc1", T // 2, 3072, 768, dict(act=2)), ("vit half fc2", T // 2, 768, 3072, dict(res=True, f32=True)),
          ("rn 14x14 1x1 1024->256", 25088, 256, 1024, dict(act=1)), ("rn 14x14 1x1 256->1024 res", 25088, 1024, 256, dict(act=1, res=True)),
          ("rn 7x7 2048->512", 6272, 512, 2048, dict(act=1)), ("rn 7x7 512->2048 res", 6272, 2048, 512, dict(act=1, res=True)),
          ("rn 28x28 1x1 512->128", 100352, 128, 512, dict(act=1)), ("swin s2 fc1", 128 * 196, 1536, 384, dict(act=2)),
          ("swin s2 fc2", 128 * 196, 384, 1536, dict(res=True, f32=True)),
          ("square 8192", 8192, 8192, 8192, dict())]
for name, M, N, K, kw in shapes:
    u0, k0 = run(M, N, K, **kw)
    u4, k4 = run(M, N, K, flags=(("igemm4", 2),), **kw)
    u5, k5 = run(M, N, K, flags=(("igemm4", 3),), **kw)
    gf = 2.0 * M * N * K / 1e6
    print(f"{name:28s} M{M} N{N} K{K}: default {k0:28s} {u0:7.1f} us {gf/u0:6.0f} TF | g4 256x256 {u4:7.1f} us {gf/u4:6.0f} TF ({u0/u4:.2f}x) | g4 256x128 {u5:7.1f} us {gf/u5:6.0f} TF ({u0/u5:.2f}x)", flush=True)
