"""Where a tile's time goes (debug build: `EQV_PROF=1 python -m eqxvision_amd.build --force`): per-block wall-clock stamps
(prologue / main loop / epilogue, 100 MHz) of igemm8's 256 x 256 kernel and, for block 0, the shader clock at every barrier
exit of its 8 waves (LOAD / MFMA interval lengths).   usage: phase_prof.py [M N K act res f32]..."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
from gemm_sweep_lib import L


def one(M, N, K, act=0, res=False, f32=False, flags=(("igemm8", 2),)):
    x = torch.randn(M, K, device="cuda").to(torch.bfloat16)
    w = (torch.randn(N, K, device="cuda") / K ** 0.5).to(torch.bfloat16)
    b = torch.randn(N, device="cuda")
    odt = torch.float32 if f32 else torch.bfloat16
    r = torch.randn(M, N, device="cuda").to(odt) if res else None
    y = torch.empty(M, N, device="cuda", dtype=odt)
    s = torch.cuda.current_stream().cuda_stream
    prof = torch.zeros(1 << 17, dtype=torch.int64, device="cuda")
    for f, v in flags: L.set_flag(f, v)
    def go():
        L.call("mv_linear_fwd", x.data_ptr(), w.data_ptr(), None, b.data_ptr(), None if r is None else r.data_ptr(),
               y.data_ptr(), M, N, K, act, 1, 0 if f32 else 1, s)
    for _ in range(3): go()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
    e0.record(); go(); e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3
    p = prof.data_ptr()
    lo = p & 0xffffffff
    if lo >= 1 << 31: lo -= 1 << 32
    L.set_flag("prof_hi", p >> 32); L.set_flag("prof_lo", lo)
    go(); torch.cuda.synchronize()
    L.set_flag("prof_lo", 0); L.set_flag("prof_hi", 0)
    k = L.last_kernel()
    for f, v in flags: L.set_flag(f, 0)
    nb = ((M + 255) // 256) * ((N + 255) // 256)
    raw = prof.cpu().numpy()
    a = raw[:4 * nb].reshape(-1, 4)
    a = a[a[:, 0] > 0]
    if len(a) == 0:
        print(f"M{M} N{N} K{K}: {k} not instrumented ({us:.1f} us) -- build with EQV_PROF=1"); return
    t0 = a[:, 0].min()
    pro = (a[:, 1] - a[:, 0]) / 100.0; main = (a[:, 2] - a[:, 1]) / 100.0; epi = (a[:, 3] - a[:, 2]) / 100.0
    span = (a[:, 3].max() - t0) / 100.0
    busy = ((a[:, 3] - a[:, 0]) / 100.0).sum() / (256 * span)
    nk = K // 64
    print(f"M{M} N{N} K{K} act{act} res{int(res)} f32{int(f32)} {k}: {us:.1f} us {2.0*M*N*K/us/1e6:.0f} TF  blocks {len(a)} "
          f"span {span:.1f}  per tile: pro {pro.mean():.2f} main {main.mean():.2f} ({main.mean()/nk:.3f}/k-tile) epi {epi.mean():.2f} us  CU-busy {busy:.2f}")
    st = raw[4 * nb:4 * nb + 8 * 128].reshape(8, 128)
    for wv in (0, 4):
        v = st[wv]; v = v[v >= 0].astype(np.int64)
        if len(v) < 10: continue
        d = np.diff(v) & 0xffffffff
        # stamps: [after prologue] then per k-tile: after B1 (end LOAD1), after B2 (end MFMA1), after B3 (end LOAD2), after B4 (end MFMA2)
        d = d[: (len(d) // 4) * 4].reshape(-1, 4)[1:-1]
        if len(d):
            print(f"   wave {wv}: cycles per interval  LOAD1 {d[:,0].mean():6.0f}  MFMA1 {d[:,1].mean():6.0f}  LOAD2 {d[:,2].mean():6.0f}  MFMA2 {d[:,3].mean():6.0f}"
                  f"   k-tile {d.sum(1).mean():6.0f}  (first tiles: {d[:3].tolist()})")

def ablate():
    from gemm_sweep_lib import run
    for M, N, K in ((8192, 8192, 8192), (4096, 4096, 4096), (50432, 3072, 768)):
        row = []
        for name, v in (("all", 0), ("no-dma", 1), ("no-reads", 2), ("barriers+mfma", 3)):
            us, k = run(M, N, K, flags=(("igemm8", 2), ("i8_ablate", v)) if v else (("igemm8", 2),))
            row.append(f"{name} {us:7.1f}us")
        print(f"ablation M{M} N{N} K{K}: " + " | ".join(row), flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "ablate":
        ablate(); sys.exit(0)
    B = 256
    T = 197 * B
    one(8192, 8192, 8192)
    one(T, 2304, 768)
    one(T, 3072, 768, act=2)
    one(T, 768, 3072, res=True, f32=True)
    one(T, 768, 768, res=True, f32=True)
    one(T // 2, 3072, 768, act=2)
