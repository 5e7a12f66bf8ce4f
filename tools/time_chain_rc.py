"""ResNet-50 layer 1's three bottleneck boundaries, alone on the chip: the round-5 launches (dual chain with y, chain with residual,
chain -> 128 with the full y) against the round-6 plan (dual chain without y, recompute chain, chain -> 128 with y sub-sampled).
usage: time_chain_rc.py [images]   (default 128 = one lane of the bench)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from eqxvision_amd import _lib as L
from tests._cases import _rc_fragments

N = int(sys.argv[1]) if len(sys.argv) > 1 else 128
H = W = 56
M, C, K = N * H * W, 64, 256
s = torch.cuda.current_stream().cuda_stream
bf = lambda *sh: torch.randn(*sh, device="cuda").bfloat16()
x0, t20, t21, t22 = (bf(M, C) for _ in range(4))
y0, y1, y2 = (torch.empty(M, K, device="cuda", dtype=torch.bfloat16) for _ in range(3))
y2s = torch.empty(N * (H // 2) * (W // 2), K, device="cuda", dtype=torch.bfloat16)
t1a, t1b = (torch.empty(M, 64, device="cuda", dtype=torch.bfloat16) for _ in range(2))
t1c = torch.empty(M, 128, device="cuda", dtype=torch.bfloat16)
wcat = (torch.randn(K, 2 * C, device="cuda") / (2 * C) ** 0.5).bfloat16()
w3 = (torch.randn(K, C, device="cuda") / C ** 0.5).bfloat16()
w1 = (torch.randn(64, K, device="cuda") / K ** 0.5).bfloat16()
w1c = (torch.randn(128, K, device="cuda") / K ** 0.5).bfloat16()
sK, hK = torch.rand(K, device="cuda") + 0.5, torch.rand(K, device="cuda")
s64, h64, s128, h128 = (torch.rand(n, device="cuda") + 0.5 for n in (64, 64, 128, 128))
wf = torch.from_numpy(_rc_fragments(wcat.float().cpu().numpy(), w3.float().cpu().numpy(), w1.float().cpu().numpy())).bfloat16().cuda()
from tests._cases import _rc_shifts
tab = torch.from_numpy(_rc_shifts(hK.cpu().numpy(), hK.cpu().numpy(), h64.cpu().numpy()).view(np.int32)).cuda()
_f = _rc_fragments(wcat.float().cpu().numpy(), w3.float().cpu().numpy(), w1.float().cpu().numpy()).reshape(8, 16, 64, 8)
wf0 = torch.from_numpy(np.ascontiguousarray(np.concatenate([_f[:, :8], _f[:, 12:]], 1)).reshape(-1)).bfloat16().cuda()
tab0 = torch.from_numpy(_rc_shifts(hK.cpu().numpy(), h64.cpu().numpy()).view(np.int32)).cuda()
def _res_frags(w3s, w1s):
    K, T2 = w3s.shape[0], w1s.shape[0] // 32
    fr = np.zeros((K // 32, 4 + 2 * T2, 2, 32, 8), np.float32)
    e8 = np.arange(8)
    for c in range(K // 32):
        for fh in range(2):
            for kk in range(4):
                fr[c, kk, fh] = w3s[32 * c:32 * c + 32][:, 16 * kk + 8 * fh + e8]
            for s_ in range(2):
                cols = 32 * c + 16 * s_ + 4 * fh + np.array([0, 1, 2, 3, 8, 9, 10, 11])
                for a2 in range(T2):
                    fr[c, 4 + T2 * s_ + a2, fh] = w1s[32 * a2:32 * a2 + 32][:, cols]
    return fr.reshape(-1)
wfr = torch.from_numpy(_res_frags(w3.float().cpu().numpy(), w1c.float().cpu().numpy())).bfloat16().cuda()
shr = torch.from_numpy(_rc_shifts(hK.cpu().numpy(), h128.cpu().numpy()).view(np.int32)).cuda()
P = lambda t: t.data_ptr() if t is not None else None

launches = {
    "dual chain, y0 written (round 5)": lambda: L.call("mv_conv1x1_dual_chain_fwd", P(t20), P(x0), P(wcat), None, P(hK), P(y0), P(w1), P(s64), P(h64), P(t1a), M, C, C, K, 64, 1, s),
    "dual chain, y0 NOT written": lambda: L.call("mv_conv1x1_dual_chain_fwd", P(t20), P(x0), P(wcat), None, P(hK), None, P(w1), P(s64), P(h64), P(t1a), M, C, C, K, 64, 1, s),
    "chain_rc0, y0 NOT written": lambda: L.call("mv_conv1x1_chain_rc0_fwd", P(t20), P(x0), P(wf0), P(tab0), P(t1a), M, C, K, 64, 1, s),
    "chain, residual y0 read (round 5)": lambda: L.call("mv_conv1x1_chain_fwd", P(t21), P(w3), P(sK), P(hK), P(y0), P(y1), P(w1), P(s64), P(h64), P(t1b), M, C, K, 64, 1, s),
    "chain_rc, y0 recomputed": lambda: L.call("mv_conv1x1_chain_rc_fwd", P(t21), P(t20), P(x0), P(wf), P(tab), P(y1), P(t1b), M, C, K, 64, 1, s),
    "chain -> 128, y2 full (round 5)": lambda: L.call("mv_conv1x1_chain_fwd", P(t22), P(w3), P(sK), P(hK), P(y1), P(y2), P(w1c), P(s128), P(h128), P(t1c), M, C, K, 128, 1, s),
    "chain -> 128, y2 sub-sampled": lambda: L.call("mv_conv1x1_chain_sub_fwd", P(t22), P(w3), P(sK), P(hK), P(y1), P(y2s), P(w1c), P(s128), P(h128), P(t1c), N, H, W, C, K, 128, 1, s),
    "chain_res -> 128, y2 sub-sampled": lambda: L.call("mv_conv1x1_chain_res_fwd", P(t22), P(y1), P(wfr), P(shr), P(y2s), P(t1c), N, H, W, C, K, 128, 2, 1, s),
    "chain_res -> 128, y2 full": lambda: L.call("mv_conv1x1_chain_res_fwd", P(t22), P(y1), P(wfr), P(shr), P(y2), P(t1c), N, H, W, C, K, 128, 0, 1, s),
}
mb = {"chain_res -> 128, y2 sub-sampled": 2 * M * (C + K + K // 4 + 128), "chain_res -> 128, y2 full": 2 * M * (C + 2 * K + 128),
      "dual chain, y0 written (round 5)": 2 * M * (2 * C + K + 64), "dual chain, y0 NOT written": 2 * M * (2 * C + 64),
      "chain_rc0, y0 NOT written": 2 * M * (2 * C + 64),
      "chain, residual y0 read (round 5)": 2 * M * (C + 2 * K + 64), "chain_rc, y0 recomputed": 2 * M * (3 * C + K + 64),
      "chain -> 128, y2 full (round 5)": 2 * M * (C + 2 * K + 128), "chain -> 128, y2 sub-sampled": 2 * M * (C + K + K // 4 + 128)}


def t(fn, n=20):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


print(f"# ResNet-50 layer-1 boundaries, {N} images ({M} pixels), each launch alone on the chip (20 back-to-back launches)")
tot = {"round 5": 0.0, "round 6": 0.0}
for name, fn in launches.items():
    us = t(fn)
    if name not in ("dual chain, y0 NOT written", "chain -> 128, y2 sub-sampled", "chain_res -> 128, y2 full"):
        tot["round 5" if "round 5" in name else "round 6"] += us
    print(f"{name:36s} [{L.last_kernel():40s}] {us:7.1f} us  {mb[name] / 1e6:6.1f} MB algorithmic  {mb[name] / us / 1e6:5.2f} TB/s")
print(f"sum of the three boundaries: round 5 {tot['round 5']:.1f} us, round 6 {tot['round 6']:.1f} us")

if os.environ.get("EQV_LIB", "").endswith("_prof.so"):          # the ablation switch exists in the debug build only
    for dbg, what in ((1, "y stores dropped"), (2, "every tile reads the same 32 rows (cache hits)"), (4, "t1 stores dropped"),
                      (3, "no y stores + cached reads"), (7, "no stores at all + cached reads")):
        L.set_flag("rc_dbg", dbg)
        print(f"ablation rc_dbg={dbg} ({what}): chain_rc0 {t(launches['chain_rc0, y0 NOT written']):.1f} us, chain_rc1 {t(launches['chain_rc, y0 recomputed']):.1f} us")
    L.set_flag("rc_dbg", 0)
