"""GPU parity cases: each returns {"ok": bool, "err": ..., ...}.  HIP path (through the C ABI) vs the
CPU oracle (`oracle/np_ops.py`) on the same seeded inputs.

bf16 cases feed the oracle the SAME bf16-rounded inputs, so the only differences are fp32
accumulation order and the rounding of the stored output:
    bf16 output:  max|got-ref| <= 1e-2 * max(1, max|ref|)     (north-star bf16 tolerance)
    fp32 output:  max|got-ref| <= 1e-3 * max(1, max|ref|)     (north-star fp32 tolerance)
"""
from __future__ import annotations

import numpy as np
import torch

from oracle import np_ops as O

TOL_BF16 = 1e-2
TOL_F32 = 1e-3


def _lib():
    from eqxvision_amd import _lib as L
    return L


def _rng(seed):
    return np.random.Generator(np.random.PCG64(seed))


def bf(a):
    return O.bf16_round(np.asarray(a, np.float32))


def dev(a, dtype):
    t = torch.from_numpy(np.ascontiguousarray(a))
    if dtype == "bf16":
        t = t.to(torch.bfloat16)
    return t.cuda()


def host(t):
    return t.float().cpu().numpy()


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _cmp(got, ref, tol):
    got = np.asarray(got, np.float64)
    ref = np.asarray(ref, np.float64)
    if got.shape != ref.shape:
        return {"ok": False, "err": f"shape {got.shape} vs {ref.shape}"}
    if not np.isfinite(got).all():
        return {"ok": False, "err": "non-finite output", "nan": int((~np.isfinite(got)).sum())}
    d = np.abs(got - ref).max()
    lim = tol * max(1.0, np.abs(ref).max())
    return {"ok": bool(d <= lim), "err": float(d), "lim": float(lim), "refmax": float(np.abs(ref).max())}


DT = {"bf16": 1, "fp32": 0}


def _splitk_runs(L, launch, y, M, N, kred, info):
    """Split-K protocol of include/eqxvision_amd.h (mv_set_scratch): the launch must take the scratch (kernel name ..._splitk), give
    the same bits on every run (a + b == b + a; the arrival words are left zero), and stay within tolerance of the un-split result's
    oracle (checked by the caller on the last run)."""
    nb = int(L.load().mv_splitk_scratch_bytes(M, N, kred))
    if not nb:
        info.update(ok=False, err=f"mv_splitk_scratch_bytes({M}, {N}, {kred}) = 0: not a split shape")
        return None
    ws = torch.zeros(nb, dtype=torch.uint8, device="cuda")
    outs = []
    for _ in range(4):
        L.call("mv_set_scratch", ws.data_ptr(), nb, _stream())
        launch()
        kern = L.last_kernel()
        torch.cuda.synchronize()
        outs.append(y.clone())
    info["kernel"] = kern
    info["splitk"] = kern.endswith("_splitk")
    info["repeatable"] = all(torch.equal(outs[0], o) for o in outs[1:])
    info["sync_clean"] = bool((ws[:4096] == 0).all().item())
    return kern


def expect_kernel(case, name):
    """The case must pass AND have been served by the kernel called `name` (dispatch regressions show up as a different name)."""
    def run():
        info = case()
        info["expected_kernel"] = name
        info["ok"] = bool(info.get("ok")) and info.get("kernel") == name
        return info
    return run


def splitk_protocol_case(seed=0):
    """mv_set_scratch edge cases (include/eqxvision_amd.h): scratch that is too small, or handed over for ANOTHER stream, is ignored
    (un-split kernel, the bits of the no-scratch launch); a hand-over is consumed by one launch (the next launch without a new
    hand-over runs un-split); bytes = 0 withdraws it."""
    def run():
        L = _lib()
        rng = _rng(seed)
        M, K, N = 3136, 3072, 768
        x = dev(bf(rng.standard_normal((M, K))), "bf16")
        w = dev(bf(rng.standard_normal((N, K)) / np.sqrt(K)), "bf16")
        b = dev((0.1 * rng.standard_normal(N)).astype(np.float32), "fp32")
        y = torch.empty((M, N), dtype=torch.bfloat16, device="cuda")
        nb = int(L.load().mv_splitk_scratch_bytes(M, N, K))
        if not nb:
            return {"ok": False, "err": "not a split shape"}
        ws = torch.zeros(nb, dtype=torch.uint8, device="cuda")
        other = torch.cuda.Stream()

        def launch():
            L.call("mv_linear_fwd", x.data_ptr(), w.data_ptr(), None, b.data_ptr(), None, y.data_ptr(), M, N, K, 0, 1, 1, _stream())
            k = L.last_kernel()
            torch.cuda.synchronize()
            return k, y.clone()
        k0, y0 = launch()                                                    # no scratch at all
        L.call("mv_set_scratch", ws.data_ptr(), nb // 2, _stream())
        k1, y1 = launch()                                                    # too small
        L.call("mv_set_scratch", ws.data_ptr(), nb, other.cuda_stream)
        k2, y2 = launch()                                                    # for another stream
        L.call("mv_set_scratch", ws.data_ptr(), nb, _stream())
        k3, y3 = launch()                                                    # taken
        k4, y4 = launch()                                                    # consumed: not taken again
        L.call("mv_set_scratch", ws.data_ptr(), nb, _stream())
        L.call("mv_set_scratch", None, 0, _stream())
        k5, y5 = launch()                                                    # withdrawn
        unsplit = [k for k in (k0, k1, k2, k4, k5)]
        info = {"kernels": [k0, k1, k2, k3, k4, k5],
                "unsplit_ok": all(not k.endswith("_splitk") for k in unsplit), "split_ok": k3.endswith("_splitk"),
                "same_bits_unsplit": all(torch.equal(y0, t) for t in (y1, y2, y4, y5)),
                "split_close": float((y3.float() - y0.float()).abs().max().item())}
        info["ok"] = bool(info["unsplit_ok"] and info["split_ok"] and info["same_bits_unsplit"] and info["split_close"] <= 0.05)
        info["err"] = info["split_close"]
        return info
    return run


# --------------------------------------------------------------------------------------------
def conv_nhwc_case(N, H, W, C, K, R, S, stride=1, pad=0, dil=1, groups=1, act=0, res=False, scale=True,
                   dtype="bf16", out="same", generic=False, seed=0, tile=0, flags=(), splitk=False):
    def run():
        L = _lib()
        rng = _rng(seed)
        q = bf if dtype == "bf16" else (lambda a: np.asarray(a, np.float32))
        x = q(rng.standard_normal((N, C, H, W)))
        w = q(rng.standard_normal((K, C // groups, R, S)) / np.sqrt(C // groups * R * S))
        sc = rng.uniform(0.5, 1.5, K).astype(np.float32) if scale else None
        sf = (0.1 * rng.standard_normal(K)).astype(np.float32)
        Ho = (H + 2 * pad - dil * (R - 1) - 1) // stride + 1
        Wo = (W + 2 * pad - dil * (S - 1) - 1) // stride + 1
        odt = dtype if out == "same" else out
        qo = bf if odt == "bf16" else (lambda a: np.asarray(a, np.float32))
        r = qo(rng.standard_normal((N, K, Ho, Wo))) if res else None
        ref = np.stack([O.conv2d(x[i], w, None, stride, pad, dil, groups) for i in range(N)])
        if sc is not None:
            ref = ref * sc[None, :, None, None]
        ref = ref + sf[None, :, None, None]
        if r is not None:
            ref = ref + r
        if act == 1:
            ref = O.relu(ref)
        elif act == 2:
            ref = O.gelu_tanh(ref)
        elif act >= 3:
            ref = {3: O.hard_swish, 4: O.hard_sigmoid, 5: O.sigmoid, 6: O.silu}[act](ref)
        xd = dev(x.transpose(0, 2, 3, 1), dtype)
        wd = dev(w.transpose(0, 2, 3, 1), dtype)
        scd = None if sc is None else dev(sc, "fp32")
        sfd = dev(sf, "fp32")
        rd = None if r is None else dev(r.transpose(0, 2, 3, 1), odt)
        y = torch.empty((N, Ho, Wo, K), dtype=torch.bfloat16 if odt == "bf16" else torch.float32, device="cuda")
        L.set_flag("force_generic", 1 if generic else 0)
        L.set_flag("igemm_tile", tile)
        for f in flags:
            L.set_flag(f.split("=")[0], int(f.split("=")[1]) if "=" in f else 1)
        extra = {}
        try:
            def launch():
                L.call("mv_conv2d_nhwc_fwd", xd.data_ptr(), wd.data_ptr(), None if scd is None else scd.data_ptr(),
                       sfd.data_ptr(), None if rd is None else rd.data_ptr(), y.data_ptr(),
                       N, H, W, C, K, R, S, stride, stride, pad, pad, dil, dil, groups, act, DT[dtype], DT[odt], _stream())
            if splitk:
                kern = _splitk_runs(L, launch, y, N * Ho * Wo, K, R * S * C, extra)
                if kern is None:
                    return extra
            else:
                launch()
                kern = L.last_kernel()
        finally:
            L.set_flag("force_generic", 0)
            L.set_flag("igemm_tile", 0)
            for f in flags:
                L.set_flag(f.split("=")[0], 0)
        torch.cuda.synchronize()
        got = host(y).transpose(0, 3, 1, 2)
        info = _cmp(got, ref, TOL_BF16 if odt == "bf16" else TOL_F32)
        info["kernel"] = kern
        info.update(extra)
        if splitk:
            info["ok"] = info["ok"] and extra["splitk"] and extra["repeatable"] and extra["sync_clean"]
        return info
    return run


def chain_case(M, seed=0, N2=64, C=64, K=256):
    """mv_conv1x1_chain_fwd (bottleneck tail + next bottleneck head in one launch, resnet.py:144-162) vs the oracle,
    and bit-for-bit vs the library's own un-fused pair of 1x1 convolutions."""
    def run():
        L = _lib()
        rng = _rng(seed)
        x = bf(rng.standard_normal((M, C)))
        w3 = bf(rng.standard_normal((K, C)) / np.sqrt(C))
        s3 = rng.uniform(0.5, 1.5, K).astype(np.float32)
        s3[::5] *= -1.0
        h3 = (0.1 * rng.standard_normal(K)).astype(np.float32)
        r = bf(rng.standard_normal((M, K)))
        w1 = bf(rng.standard_normal((N2, K)) / np.sqrt(K))
        s1 = rng.uniform(0.5, 1.5, N2).astype(np.float32)
        h1 = (0.1 * rng.standard_normal(N2)).astype(np.float32)
        if not L.load().mv_conv1x1_chain_supported(M, C, K, N2, 1):
            return {"ok": False, "err": "mv_conv1x1_chain_supported says no"}
        yref = O.relu((x.astype(np.float64) @ w3.astype(np.float64).T) * s3 + h3 + r)
        t1ref = O.relu((bf(yref).astype(np.float64) @ w1.astype(np.float64).T) * s1 + h1)
        d = {k: dev(v, "bf16") for k, v in dict(x=x, w3=w3, r=r, w1=w1).items()}
        f = {k: dev(v, "fp32") for k, v in dict(s3=s3, h3=h3, s1=s1, h1=h1).items()}
        y = torch.full((M, K), -7.0, dtype=torch.bfloat16, device="cuda")
        t1 = torch.full((M, N2), -7.0, dtype=torch.bfloat16, device="cuda")
        L.call("mv_conv1x1_chain_fwd", d["x"].data_ptr(), d["w3"].data_ptr(), f["s3"].data_ptr(), f["h3"].data_ptr(),
               d["r"].data_ptr(), y.data_ptr(), d["w1"].data_ptr(), f["s1"].data_ptr(), f["h1"].data_ptr(), t1.data_ptr(),
               M, C, K, N2, 1, _stream())
        kern = L.last_kernel()
        y2 = torch.empty_like(y)
        t2 = torch.empty_like(t1)
        L.call("mv_linear_fwd", d["x"].data_ptr(), d["w3"].data_ptr(), f["s3"].data_ptr(), f["h3"].data_ptr(), d["r"].data_ptr(),
               y2.data_ptr(), M, K, C, 1, 1, 1, _stream())
        L.call("mv_linear_fwd", y2.data_ptr(), d["w1"].data_ptr(), f["s1"].data_ptr(), f["h1"].data_ptr(), None,
               t2.data_ptr(), M, N2, K, 1, 1, 1, _stream())
        torch.cuda.synchronize()
        a = _cmp(host(y), yref, TOL_BF16)
        b = _cmp(host(t1), t1ref, TOL_BF16)
        same_y = bool(torch.equal(y, y2))
        dt1 = float((t1.float() - t2.float()).abs().max())
        return {"ok": a["ok"] and b["ok"] and same_y and dt1 <= b["lim"], "err": max(a["err"], b["err"]), "lim": a["lim"],
                "y_bit_identical_to_unfused": same_y, "t1_bit_identical_to_unfused": dt1 == 0.0, "t1_vs_unfused": dt1,
                "kernel": kern}
    return run


def bneck_tail_case(B, seed=0, HW=14, WID=256, COUT=1024, big=False):
    """mv_bottleneck_tail_fwd (an identity bottleneck's conv2 3x3 + BN + ReLU -> conv3 1x1 + BN + identity + ReLU in one launch,
    one workgroup per image, resnet.py:144-162) vs the oracle; the intermediate is rounded to bf16 exactly where the un-fused
    pair of launches stores it.  Weights are handed over in the fragment order the header documents (ops.prep_bneck_tail)."""
    def run():
        L = _lib()
        rng = _rng(seed)
        amp = 4.0 if big else 1.0
        t1 = bf(np.maximum(rng.standard_normal((B, WID, HW, HW)), 0) * amp)
        w2 = bf(rng.standard_normal((WID, WID, 3, 3)) / np.sqrt(WID * 9))
        s2 = rng.uniform(0.5, 1.5, WID).astype(np.float32)
        s2[::7] *= -1.0
        h2 = (0.1 * rng.standard_normal(WID)).astype(np.float32)
        w3 = bf(rng.standard_normal((COUT, WID)) / np.sqrt(WID))
        s3 = rng.uniform(0.5, 1.5, COUT).astype(np.float32)
        s3[::5] *= -1.0
        h3 = (0.1 * rng.standard_normal(COUT)).astype(np.float32)
        r = bf(rng.standard_normal((B, COUT, HW, HW)) * amp)
        if not L.load().mv_bottleneck_tail_supported(HW, HW, WID, COUT, 1):
            return {"ok": False, "err": "mv_bottleneck_tail_supported says no"}
        t2 = np.stack([O.conv2d(t1[i], w2, None, 1, 1, 1, 1) for i in range(B)])
        t2 = bf(O.relu(t2 * s2[None, :, None, None] + h2[None, :, None, None]))
        yref = np.einsum("bchw,kc->bkhw", t2.astype(np.float64), w3.astype(np.float64))
        yref = O.relu(yref * s3[None, :, None, None] + h3[None, :, None, None] + r)
        w2k = np.ascontiguousarray(w2.transpose(0, 2, 3, 1))                                              # KRSC
        w2f = w2k.reshape(WID // 32, 32, 9, WID // 16, 2, 8).transpose(0, 2, 3, 4, 1, 5)
        w3f = w3.reshape(COUT // 256, 8, 32, WID // 16, 2, 8).transpose(0, 1, 3, 4, 2, 5)
        d = {k: dev(v, "bf16") for k, v in dict(t1=t1.transpose(0, 2, 3, 1), w2f=w2f, w3f=w3f, r=r.transpose(0, 2, 3, 1)).items()}
        f = {k: dev(v, "fp32") for k, v in dict(s2=s2, h2=h2, s3=s3, h3=h3).items()}
        y = torch.full((B, HW, HW, COUT), -7.0, dtype=torch.bfloat16, device="cuda")
        L.call("mv_bottleneck_tail_fwd", d["t1"].data_ptr(), d["w2f"].data_ptr(), f["s2"].data_ptr(), f["h2"].data_ptr(),
               d["w3f"].data_ptr(), f["s3"].data_ptr(), f["h3"].data_ptr(), d["r"].data_ptr(), y.data_ptr(), B, HW, HW, WID, COUT, 1,
               _stream())
        kern = L.last_kernel()
        torch.cuda.synchronize()
        info = _cmp(host(y).transpose(0, 3, 1, 2), yref, TOL_BF16)
        info["kernel"] = kern
        return info
    return run


# ---- backward kernels (csrc/train_bwd.hip) vs torch.autograd on the CPU (fp32 both sides)
def _ag(fn, *inputs):
    """torch.autograd of sum(fn(*inputs) * seed) for a fixed random seed tensor -> (output, seed, grads of the inputs)."""
    ts = [torch.from_numpy(np.ascontiguousarray(a)).clone().requires_grad_(True) for a in inputs]
    y = fn(*ts)
    g = torch.from_numpy(np.random.Generator(np.random.PCG64(99)).standard_normal(tuple(y.shape)).astype(np.float32))
    (y * g).sum().backward()
    return y.detach().numpy(), g.numpy(), [t.grad.numpy() for t in ts]


def conv_bwd_case(N, H, W, C, K, R, stride=1, pad=0, dil=1, seed=0, groups=1):
    def run():
        import torch.nn.functional as F
        L = _lib()
        rng = _rng(seed)
        x = rng.standard_normal((N, C, H, W)).astype(np.float32)
        w = (rng.standard_normal((K, C // groups, R, R)) / np.sqrt(C // groups * R * R)).astype(np.float32)
        y, g, (dx_ref, dw_ref) = _ag(lambda a, b: F.conv2d(a, b, None, stride, pad, dil, groups), x, w)
        d = {k: dev(v, "fp32") for k, v in dict(x=x.transpose(0, 2, 3, 1), w=w.transpose(0, 2, 3, 1), g=g.transpose(0, 2, 3, 1)).items()}
        dx = torch.full((N, H, W, C), -7.0, device="cuda")
        dw = torch.full((K, R, R, C // groups), -7.0, device="cuda")
        L.call("mv_conv2d_dgrad_nhwc_f32", d["g"].data_ptr(), d["w"].data_ptr(), dx.data_ptr(), N, H, W, C, K, R, R, stride, stride, pad, pad,
               dil, dil, groups, _stream())
        L.call("mv_conv2d_wgrad_nhwc_f32", d["x"].data_ptr(), d["g"].data_ptr(), dw.data_ptr(), N, H, W, C, K, R, R, stride, stride, pad, pad,
               dil, dil, groups, _stream())
        kern = L.last_kernel()
        torch.cuda.synchronize()
        a = _cmp(host(dx).transpose(0, 3, 1, 2), dx_ref, TOL_F32)
        b = _cmp(host(dw).transpose(0, 3, 1, 2), dw_ref, TOL_F32)
        # the same weight gradient with scratch on offer (mv_set_scratch): the positions are split over blocks and the parts added in a
        # fixed order -- same tolerance, and the same bits on a second run
        ws = torch.empty(32 << 20, dtype=torch.uint8, device="cuda")
        outs = []
        for _ in range(2):
            dw2 = torch.full((K, R, R, C // groups), -7.0, device="cuda")
            L.call("mv_set_scratch", ws.data_ptr(), ws.numel(), _stream())
            L.call("mv_conv2d_wgrad_nhwc_f32", d["x"].data_ptr(), d["g"].data_ptr(), dw2.data_ptr(), N, H, W, C, K, R, R, stride, stride, pad,
                   pad, dil, dil, groups, _stream())
            torch.cuda.synchronize()
            outs.append(dw2)
        L.call("mv_set_scratch", None, 0, _stream())
        c = _cmp(host(outs[0]).transpose(0, 3, 1, 2), dw_ref, TOL_F32)
        rep = bool(torch.equal(outs[0], outs[1]))
        return {"ok": a["ok"] and b["ok"] and c["ok"] and rep,
                "err": max(a["err"], b["err"], c["err"]) if all(isinstance(t["err"], float) for t in (a, b, c)) else (a["err"], b["err"], c["err"]),
                "dx": a, "dw": b, "dw_split": c, "repeatable": rep, "kernel": kern}
    return run


def mha_bwd_case(B, N, H, dh, seed=0):
    """mv_mha_bwd_f32 (two launches) vs torch.autograd of softmax(q k^T scale) v on the same qkv rows [q | k | v][head][dh]."""
    def run():
        L = _lib()
        rng = _rng(seed)
        D = H * dh
        qkv = rng.standard_normal((B, N, 3 * D)).astype(np.float32)
        g = rng.standard_normal((B, N, D)).astype(np.float32)
        scale = dh ** -0.5
        t = torch.from_numpy(qkv).requires_grad_(True)
        q, k, v = (t[:, :, i * D:(i + 1) * D].reshape(B, N, H, dh).permute(0, 2, 1, 3) for i in range(3))
        pr = torch.softmax(q @ k.transpose(-1, -2) * scale, -1)
        out = (pr @ v).permute(0, 2, 1, 3).reshape(B, N, D)
        out.backward(torch.from_numpy(g))
        ref = t.grad.numpy()
        qd, gd = dev(qkv, "fp32"), dev(g, "fp32")
        od = torch.empty((B, N, D), device="cuda")
        pd = torch.empty((B, H, N, N), device="cuda")
        L.call("mv_mha_fwd", qd.data_ptr(), od.data_ptr(), pd.data_ptr(), B, N, H, dh, float(scale), 0, _stream())
        ds = torch.empty((B, H, N, N), device="cuda")
        dq = torch.full((B, N, 3 * D), -7.0, device="cuda")
        L.call("mv_mha_bwd_f32", qd.data_ptr(), pd.data_ptr(), gd.data_ptr(), ds.data_ptr(), dq.data_ptr(), B, N, H, dh, float(scale), _stream())
        torch.cuda.synchronize()
        a = _cmp(host(od), out.detach().numpy(), TOL_F32)
        b = _cmp(host(dq), ref, TOL_F32)
        return {"ok": a["ok"] and b["ok"], "err": b["err"], "fwd": a, "kernel": L.last_kernel()}
    return run


def colsum_case(M, C, seed=0, product=False):
    """mv_colsum_f32 (bias / BatchNorm gradients): out[c] = sum_m a[m, c] (* b[m, c]) -- the single-block-per-64-columns kernel and, with
    scratch on offer, the row-split kernel + fixed-order finish: both vs float64, the split one repeatable bit for bit."""
    def run():
        L = _lib()
        rng = _rng(seed)
        a = rng.standard_normal((M, C)).astype(np.float32)
        b = rng.standard_normal((M, C)).astype(np.float32) if product else None
        ref = (a.astype(np.float64) * (b.astype(np.float64) if product else 1.0)).sum(0)
        ad, bd = dev(a, "fp32"), (dev(b, "fp32") if product else None)
        outs, kerns = [], []
        ws = torch.empty(8 << 20, dtype=torch.uint8, device="cuda")
        for use in (False, True, True):
            o = torch.full((C,), -7.0, device="cuda")
            if use:
                L.call("mv_set_scratch", ws.data_ptr(), ws.numel(), _stream())
            L.call("mv_colsum_f32", ad.data_ptr(), None if bd is None else bd.data_ptr(), o.data_ptr(), M, C, _stream())
            kerns.append(L.last_kernel())
            torch.cuda.synchronize()
            outs.append(o)
        L.call("mv_set_scratch", None, 0, _stream())
        tol = 2e-5 * np.sqrt(M)
        i0, i1 = _cmp(host(outs[0]), ref, tol), _cmp(host(outs[1]), ref, tol)
        split_taken = kerns[1] == "colsum_split_f32" or M < 512
        return {"ok": bool(i0["ok"] and i1["ok"] and torch.equal(outs[1], outs[2]) and split_taken and kerns[0] == "colsum_f32"),
                "err": max(i0["err"], i1["err"]), "kernels": kerns}
    return run


def maxpool_bwd_case(N, H, W, C, k, s, p, seed=0, relu=False):
    def run():
        import torch.nn.functional as F
        L = _lib()
        rng = _rng(seed)
        x = rng.standard_normal((N, C, H, W)).astype(np.float32)
        if relu:
            x = np.maximum(x, 0)                      # many exact ties at zero
        y, g, (dx_ref,) = _ag(lambda a: F.max_pool2d(a, k, s, p), x)
        xd, gd = dev(x.transpose(0, 2, 3, 1), "fp32"), dev(g.transpose(0, 2, 3, 1), "fp32")
        dx = torch.full((N, H, W, C), -7.0, device="cuda")
        L.call("mv_maxpool2d_bwd_nhwc_f32", xd.data_ptr(), gd.data_ptr(), dx.data_ptr(), N, H, W, C, k, k, s, s, p, p, _stream())
        torch.cuda.synchronize()
        got = host(dx).transpose(0, 3, 1, 2)
        info = _cmp(got, dx_ref, TOL_F32)
        info["sum_err"] = float(abs(got.sum() - dx_ref.sum()))           # with ties only the total is defined by the maths
        return info
    return run


def rowwise_bwd_case(kind, M, C, seed=0):
    """layernorm / softmax / act backward and the column sums vs torch.autograd."""
    def run():
        import torch.nn.functional as F
        L = _lib()
        rng = _rng(seed)
        x = rng.standard_normal((M, C)).astype(np.float32) * 1.5
        if kind == "layernorm":
            gam = rng.uniform(0.5, 1.5, C).astype(np.float32)
            bet = (0.1 * rng.standard_normal(C)).astype(np.float32)
            y, g, (dx_ref, dg_ref, db_ref) = _ag(lambda a, b, c: F.layer_norm(a, (C,), b, c, 1e-5), x, gam, bet)
            xd, gd, gm = dev(x, "fp32"), dev(g, "fp32"), dev(gam, "fp32")
            dx, gx = torch.empty(M, C, device="cuda"), torch.empty(M, C, device="cuda")
            dg, db = torch.empty(C, device="cuda"), torch.empty(C, device="cuda")
            L.call("mv_layernorm_bwd_f32", xd.data_ptr(), gm.data_ptr(), gd.data_ptr(), dx.data_ptr(), gx.data_ptr(), M, C, 1e-5, _stream())
            L.call("mv_colsum_f32", gx.data_ptr(), None, dg.data_ptr(), M, C, _stream())
            L.call("mv_colsum_f32", gd.data_ptr(), None, db.data_ptr(), M, C, _stream())
            torch.cuda.synchronize()
            parts = [_cmp(host(dx), dx_ref, TOL_F32), _cmp(host(dg), dg_ref, TOL_F32), _cmp(host(db), db_ref, TOL_F32)]
        elif kind == "softmax":
            scale = 0.37
            y, g, (ds_ref,) = _ag(lambda a: torch.softmax(a * scale, -1), x)
            pd, gd = dev(y, "fp32"), dev(g, "fp32")
            ds = torch.empty(M, C, device="cuda")
            L.call("mv_softmax_bwd_f32", pd.data_ptr(), gd.data_ptr(), ds.data_ptr(), M, C, scale, _stream())
            torch.cuda.synchronize()
            parts = [_cmp(host(ds), ds_ref, TOL_F32)]
        else:                                         # an activation
            act = {"relu": 1, "gelu": 2, "hard_swish": 3, "hard_sigmoid": 4, "sigmoid": 5, "silu": 6}[kind]
            f = {"relu": F.relu, "gelu": lambda a: F.gelu(a, approximate="tanh"), "hard_swish": F.hardswish, "hard_sigmoid": F.hardsigmoid,
                 "sigmoid": torch.sigmoid, "silu": F.silu}[kind]
            x = x * 2.5                                # reach both saturated ends of the hard_* functions
            y, g, (dx_ref,) = _ag(f, x)
            xd, gd = dev(x, "fp32"), dev(g, "fp32")
            dx = torch.empty(M, C, device="cuda")
            L.call("mv_act_bwd_f32", gd.data_ptr(), xd.data_ptr(), dx.data_ptr(), M * C, act, _stream())
            torch.cuda.synchronize()
            parts = [_cmp(host(dx), dx_ref, TOL_F32)]
        return {"ok": all(p["ok"] for p in parts), "err": max(p["err"] for p in parts), "parts": parts}
    return run


def swin_bwd_case(B, Hf, C, heads, shift, seed=0, ws=7):
    """mv_swin_window_attn_bwd_f32 (+ the bias-table gradient through mv_colsum_f32 / mv_scatter_rows_sum_f32) vs torch.autograd of
    the torchvision-style shifted-window attention core (roll, partition, bias, -100 mask, softmax, reverse)."""
    def run():
        L = _lib()
        rng = _rng(seed)
        n, dh, T = ws * ws, C // heads, (2 * ws - 1) ** 2
        qkv = rng.standard_normal((B, Hf, Hf, 3 * C)).astype(np.float32)
        table = (0.5 * rng.standard_normal((T, heads))).astype(np.float32)
        coords = np.stack(np.meshgrid(np.arange(ws), np.arange(ws), indexing="ij")).reshape(2, -1)
        rel = coords[:, :, None] - coords[:, None, :] + (ws - 1)
        index = (rel[0] * (2 * ws - 1) + rel[1]).reshape(-1).astype(np.int64)
        sh = [shift, shift] if Hf > ws else [0, 0]

        def core(qkv_t, table_t):
            x = qkv_t
            if sum(sh) > 0:
                x = torch.roll(x, (-sh[0], -sh[1]), (1, 2))
            nW = (Hf // ws) ** 2
            x = x.view(B, Hf // ws, ws, Hf // ws, ws, 3 * C).permute(0, 1, 3, 2, 4, 5).reshape(B * nW, n, 3, heads, dh).permute(2, 0, 3, 1, 4)
            q, k, v = x[0] * dh ** -0.5, x[1], x[2]
            attn = q @ k.transpose(-2, -1) + table_t[torch.from_numpy(index)].view(n, n, heads).permute(2, 0, 1)[None]
            if sum(sh) > 0:
                m = qkv_t.new_zeros((Hf, Hf))
                hs = ((0, -ws), (-ws, -sh[0]), (-sh[0], None))
                c = 0
                for hh in hs:
                    for ww in hs:
                        m[hh[0]:hh[1], ww[0]:ww[1]] = c
                        c += 1
                m = m.view(Hf // ws, ws, Hf // ws, ws).permute(0, 2, 1, 3).reshape(nW, n)
                m = m.unsqueeze(1) - m.unsqueeze(2)
                m = m.masked_fill(m != 0, -100.0).masked_fill(m == 0, 0.0)
                attn = (attn.view(B, nW, heads, n, n) + m[None, :, None]).view(-1, heads, n, n)
            y = (attn.softmax(-1) @ v).transpose(1, 2).reshape(B * nW, n, C)
            y = y.view(B, Hf // ws, Hf // ws, ws, ws, C).permute(0, 1, 3, 2, 4, 5).reshape(B, Hf, Hf, C)
            return torch.roll(y, (sh[0], sh[1]), (1, 2)) if sum(sh) > 0 else y
        y, g, (dq_ref, dt_ref) = _ag(core, qkv, table)
        bias = table[index].reshape(n, n, heads).transpose(2, 0, 1)
        qd, bd, gd = dev(qkv, "fp32"), dev(np.ascontiguousarray(bias), "fp32"), dev(g, "fp32")
        nW = (Hf // ws) ** 2
        out = torch.empty(B, Hf, Hf, C, device="cuda")
        L.call("mv_swin_window_attn_fwd", qd.data_ptr(), bd.data_ptr(), out.data_ptr(), B, Hf, Hf, C, heads, ws, ws, sh[0], sh[1], 0, _stream())
        dq, gw = torch.full((B, Hf, Hf, 3 * C), -7.0, device="cuda"), torch.empty(B * nW, heads * n * n, device="cuda")
        L.call("mv_swin_window_attn_bwd_f32", qd.data_ptr(), bd.data_ptr(), gd.data_ptr(), dq.data_ptr(), gw.data_ptr(), B, Hf, Hf, C, heads,
               ws, ws, sh[0], sh[1], _stream())
        db, dbt, dt = torch.empty(heads * n * n, device="cuda"), torch.empty(n * n, heads, device="cuda"), torch.empty(T, heads, device="cuda")
        L.call("mv_colsum_f32", gw.data_ptr(), None, db.data_ptr(), B * nW, heads * n * n, _stream())
        L.call("mv_transpose2d_f32", db.data_ptr(), dbt.data_ptr(), heads, n * n, 0, _stream())
        idx = torch.from_numpy(index.astype(np.int32)).cuda()
        L.call("mv_scatter_rows_sum_f32", dbt.data_ptr(), idx.data_ptr(), dt.data_ptr(), n * n, heads, T, _stream())
        torch.cuda.synchronize()
        parts = [_cmp(host(out), y, TOL_F32), _cmp(host(dq), dq_ref, TOL_F32), _cmp(host(dt), dt_ref, TOL_F32)]
        return {"ok": all(p_["ok"] for p_ in parts), "err": max(p_["err"] for p_ in parts), "parts": parts}
    return run


def patch_merge_bwd_case(B, H, C, seed=0):
    def run():
        L = _lib()
        rng = _rng(seed)
        x = rng.standard_normal((B, H, H, C)).astype(np.float32)
        f = lambda a: torch.cat([a[:, 0::2, 0::2], a[:, 1::2, 0::2], a[:, 0::2, 1::2], a[:, 1::2, 1::2]], -1)
        y, g, (dx_ref,) = _ag(f, x)
        gd = dev(g, "fp32")
        dx = torch.full((B, H, H, C), -7.0, device="cuda")
        L.call("mv_patch_merge_gather_bwd_f32", gd.data_ptr(), dx.data_ptr(), B, H, H, C, _stream())
        torch.cuda.synchronize()
        return _cmp(host(dx), dx_ref, TOL_F32)
    return run


def xent_adam_case(B, K, seed=0):
    def run():
        import torch.nn.functional as F
        L = _lib()
        rng = _rng(seed)
        logits = (3 * rng.standard_normal((B, K))).astype(np.float32)
        lab = rng.integers(0, K, B)
        tl = torch.from_numpy(logits).clone().requires_grad_(True)
        loss = F.cross_entropy(tl, torch.from_numpy(lab), reduction="mean")
        loss.backward()
        oh = np.zeros((B, K), np.float32)
        oh[np.arange(B), lab] = 1
        ld, od = dev(logits, "fp32"), dev(oh, "fp32")
        rows, mean, dl = torch.empty(B, device="cuda"), torch.empty(1, device="cuda"), torch.empty(B, K, device="cuda")
        L.call("mv_softmax_xent_f32", ld.data_ptr(), od.data_ptr(), rows.data_ptr(), mean.data_ptr(), dl.data_ptr(), B, K, _stream())
        # three Adam steps on the gradient tensor against torch.optim.Adam (optax.adam's update rule)
        p0 = rng.standard_normal((B, K)).astype(np.float32)
        tp = torch.from_numpy(p0).clone().requires_grad_(True)
        opt = torch.optim.Adam([tp], lr=0.01, betas=(0.9, 0.999), eps=1e-8)
        pd = dev(p0, "fp32")
        m, v, u = torch.zeros(B * K, device="cuda"), torch.zeros(B * K, device="cuda"), torch.empty(B * K, device="cuda")
        for step in range(1, 4):
            gstep = (rng.standard_normal((B, K)) * 0.3).astype(np.float32)
            tp.grad = torch.from_numpy(gstep)
            opt.step()
            gd = dev(gstep, "fp32")
            L.call("mv_adam_step_f32", gd.data_ptr(), m.data_ptr(), v.data_ptr(), u.data_ptr(), B * K, 0.01, 0.9, 0.999, 1e-8,
                   1 - 0.9 ** step, 1 - 0.999 ** step, _stream())
            new = torch.empty_like(pd)
            L.call("mv_add_fwd", pd.data_ptr(), u.data_ptr(), new.data_ptr(), B * K, 0, 0, _stream())
            pd = new
        torch.cuda.synchronize()
        parts = [_cmp(host(dl), tl.grad.numpy(), TOL_F32), _cmp(host(mean), np.asarray([float(loss.detach())]), TOL_F32),
                 _cmp(host(pd), tp.detach().numpy(), TOL_F32)]
        return {"ok": all(p["ok"] for p in parts), "err": max(p["err"] for p in parts), "parts": parts}
    return run


def dual_chain_case(M, seed=0):
    """mv_conv1x1_dual_chain_fwd: conv3 + BN and the downsample conv + BN as one GEMM over [t2 | x] (scales folded into
    the bf16 weight rows, as ops.conv1x1_dual_chain does), ReLU, then the next block's conv1 + BN + ReLU -- vs the oracle
    with the un-folded fp32 scales (resnet.py:144-162, 295-303)."""
    def run():
        L = _lib()
        rng = _rng(seed)
        C, K, N2 = 64, 256, 64
        x = bf(rng.standard_normal((M, C)))
        x2 = bf(rng.standard_normal((M, C)))
        w3 = bf(rng.standard_normal((K, C)) / np.sqrt(C))
        wd = bf(rng.standard_normal((K, C)) / np.sqrt(C))
        s3 = rng.uniform(0.5, 1.5, K).astype(np.float32)
        sd = rng.uniform(0.5, 1.5, K).astype(np.float32)
        sd[::7] *= -1.0
        h3 = (0.1 * rng.standard_normal(K)).astype(np.float32)
        hd = (0.1 * rng.standard_normal(K)).astype(np.float32)
        w1 = bf(rng.standard_normal((N2, K)) / np.sqrt(K))
        s1 = rng.uniform(0.5, 1.5, N2).astype(np.float32)
        h1 = (0.1 * rng.standard_normal(N2)).astype(np.float32)
        if not L.load().mv_conv1x1_dual_chain_supported(M, C, C, K, N2, 1):
            return {"ok": False, "err": "mv_conv1x1_dual_chain_supported says no"}
        f64 = np.float64
        yref = O.relu((x.astype(f64) @ w3.astype(f64).T) * s3 + h3 + (x2.astype(f64) @ wd.astype(f64).T) * sd + hd)
        t1ref = O.relu((bf(yref).astype(f64) @ w1.astype(f64).T) * s1 + h1)
        wcat = np.concatenate([w3.astype(np.float32) * s3[:, None], wd.astype(np.float32) * sd[:, None]], axis=1)
        d = {k: dev(v, "bf16") for k, v in dict(x=x, x2=x2, wcat=bf(wcat), w1=w1).items()}
        f = {k: dev(v, "fp32") for k, v in dict(h=(h3 + hd).astype(np.float32), s1=s1, h1=h1).items()}
        y = torch.full((M, K), -7.0, dtype=torch.bfloat16, device="cuda")
        t1 = torch.full((M, N2), -7.0, dtype=torch.bfloat16, device="cuda")
        L.call("mv_conv1x1_dual_chain_fwd", d["x"].data_ptr(), d["x2"].data_ptr(), d["wcat"].data_ptr(), None, f["h"].data_ptr(),
               y.data_ptr(), d["w1"].data_ptr(), f["s1"].data_ptr(), f["h1"].data_ptr(), t1.data_ptr(), M, C, C, K, N2, 1,
               _stream())
        kern = L.last_kernel()
        torch.cuda.synchronize()
        a = _cmp(host(y), yref, TOL_BF16)
        b = _cmp(host(t1), t1ref, TOL_BF16)
        return {"ok": a["ok"] and b["ok"], "err": max(a["err"], b["err"]), "lim": min(a["lim"], b["lim"]), "err_y": a["err"],
                "err_t1": b["err"], "kernel": kern}
    return run


def _rc_fragments(wcat, w31, w1n):
    """The fragment order mv_conv1x1_chain_rc_fwd documents (include/eqxvision_amd.h), written out independently of ops.py."""
    K, N2 = wcat.shape[0], w1n.shape[0]
    out = np.zeros((K // 32, 16, 64, 8), np.float32)
    for c in range(K // 32):
        for lane in range(64):
            fh, r = lane // 32, lane % 32
            for kk in range(8):
                out[c, kk, lane] = wcat[32 * c + r, 16 * kk + 8 * fh:16 * kk + 8 * fh + 8]
            for kk in range(4):
                out[c, 8 + kk, lane] = w31[32 * c + r, 16 * kk + 8 * fh:16 * kk + 8 * fh + 8]
            for s_ in range(2):
                for a2 in range(N2 // 32):
                    for i in range(8):
                        out[c, 12 + 2 * s_ + a2, lane, i] = w1n[32 * a2 + r, 32 * c + 8 * (2 * s_ + i // 4) + 4 * fh + i % 4]
    return out.reshape(-1)


def _rc_shifts(*vecs):
    """Shift rows as include/eqxvision_amd.h documents them (independent of ops.py): 64 words per 32 values, hi | lo << 16."""
    rows = []
    for v in vecs:
        v = np.asarray(v, np.float32)
        hi = bf(v)
        lo = bf(v - hi)
        w = (hi.view(np.uint32) >> 16) | (lo.view(np.uint32) & 0xffff0000)
        out = np.zeros((v.size // 32, 64), np.uint32)
        out[:, :32] = w.reshape(-1, 32)
        rows.append(out)
    return np.concatenate(rows, 0)


def chain_rc_case(M, seed=0):
    """mv_conv1x1_chain_rc_fwd (the second bottleneck boundary of a stage whose first block output is NOT in memory: y0 recomputed
    from [t2_0 | x0], then conv3 + BN + y0 + ReLU and the next conv1 + BN + ReLU; resnet.py:144-162, 295-303) and
    mv_conv1x1_chain_rc0_fwd (the first boundary without its output map) vs the oracle with fp32 scales, and within the same bound of
    the library's own pair mv_conv1x1_dual_chain_fwd (y0 written) -> mv_conv1x1_chain_fwd (y0 read back);
    mv_conv1x1_dual_chain_fwd with y = NULL must produce the same t1 as with y."""
    def run():
        L = _lib()
        rng = _rng(seed)
        C, K, N2 = 64, 256, 64
        f64 = np.float64
        x0, t20, t21 = (bf(np.maximum(rng.standard_normal((M, C)), -0.5)) for _ in range(3))
        w30 = bf(rng.standard_normal((K, C)) / np.sqrt(C))
        wd = bf(rng.standard_normal((K, C)) / np.sqrt(C))
        s30 = rng.uniform(0.5, 1.5, K).astype(np.float32)
        sd = rng.uniform(0.5, 1.5, K).astype(np.float32)
        sd[::7] *= -1.0
        h0 = (0.1 * rng.standard_normal(K)).astype(np.float32)
        w31 = bf(rng.standard_normal((K, C)) / np.sqrt(C))
        s31 = rng.uniform(0.5, 1.5, K).astype(np.float32)
        s31[::5] *= -1.0
        h31 = (0.1 * rng.standard_normal(K)).astype(np.float32)
        w1a = bf(rng.standard_normal((N2, K)) / np.sqrt(K))            # conv1 of block 1 (block 0's launch computes it)
        w1n = bf(rng.standard_normal((N2, K)) / np.sqrt(K))            # conv1 of block 2
        s1a, s1n = (rng.uniform(0.5, 1.5, N2).astype(np.float32) for _ in range(2))
        h1a, h1n = ((0.1 * rng.standard_normal(N2)).astype(np.float32) for _ in range(2))
        lib = L.load()
        if not (lib.mv_conv1x1_chain_rc_supported(M, C, K, N2, 1) and lib.mv_conv1x1_dual_chain_supported(M, C, C, K, N2, 1)):
            return {"ok": False, "err": "mv_conv1x1_chain_rc_supported / dual_chain_supported says no"}
        wcat = bf(np.concatenate([w30.astype(np.float32) * s30[:, None], wd.astype(np.float32) * sd[:, None]], axis=1))
        y0ref = O.relu(np.concatenate([t20, x0], 1).astype(f64) @ wcat.astype(f64).T + h0)
        y1ref = O.relu((t21.astype(f64) @ w31.astype(f64).T) * s31 + h31 + bf(y0ref))
        t1ref = O.relu((bf(y1ref).astype(f64) @ w1n.astype(f64).T) * s1n + h1n)
        # the recompute kernels take every BatchNorm scale folded into the bf16 weight rows and the shifts as two-term bf16 rows
        w31s, w1ns, w1as = (bf(w.astype(np.float32) * sc[:, None]) for w, sc in ((w31, s31), (w1n, s1n), (w1a, s1a)))
        d = {k: dev(v, "bf16") for k, v in dict(x0=x0, t20=t20, t21=t21, wcat=wcat, w31=w31, w1a=w1a, w1n=w1n,
                                               wf=bf(_rc_fragments(wcat.astype(np.float32), w31s.astype(np.float32), w1ns.astype(np.float32)))).items()}
        f = {k: dev(v, "fp32") for k, v in dict(h0=h0, s31=s31, h31=h31, s1a=s1a, h1a=h1a, s1n=s1n, h1n=h1n).items()}
        sh = torch.from_numpy(_rc_shifts(h0, h31, h1n).view(np.int32)).cuda()
        y1 = torch.full((M, K), -7.0, dtype=torch.bfloat16, device="cuda")
        t1 = torch.full((M, N2), -7.0, dtype=torch.bfloat16, device="cuda")
        L.call("mv_conv1x1_chain_rc_fwd", d["t21"].data_ptr(), d["t20"].data_ptr(), d["x0"].data_ptr(), d["wf"].data_ptr(),
               sh.data_ptr(), y1.data_ptr(), t1.data_ptr(), M, C, K, N2, 1, _stream())
        kern = L.last_kernel()
        # the pair it replaces: y0 written by the dual chain, read back as the residual of the plain chain
        y0 = torch.empty((M, K), dtype=torch.bfloat16, device="cuda")
        ta, tb = (torch.full((M, N2), -7.0, dtype=torch.bfloat16, device="cuda") for _ in range(2))
        L.call("mv_conv1x1_dual_chain_fwd", d["t20"].data_ptr(), d["x0"].data_ptr(), d["wcat"].data_ptr(), None, f["h0"].data_ptr(),
               y0.data_ptr(), d["w1a"].data_ptr(), f["s1a"].data_ptr(), f["h1a"].data_ptr(), ta.data_ptr(), M, C, C, K, N2, 1, _stream())
        L.call("mv_conv1x1_dual_chain_fwd", d["t20"].data_ptr(), d["x0"].data_ptr(), d["wcat"].data_ptr(), None, f["h0"].data_ptr(),
               None, d["w1a"].data_ptr(), f["s1a"].data_ptr(), f["h1a"].data_ptr(), tb.data_ptr(), M, C, C, K, N2, 1, _stream())
        kern_noy = L.last_kernel()
        # (1'): the first boundary without its output map in this file's style -- against the dual chain's t1
        wf0 = np.zeros((K // 32, 12, 64, 8), np.float32)
        full = _rc_fragments(wcat.astype(np.float32), w31s.astype(np.float32), w1as.astype(np.float32)).reshape(K // 32, 16, 64, 8)
        wf0[:, :8], wf0[:, 8:] = full[:, :8], full[:, 12:]
        wf0d, sh0 = dev(bf(wf0.reshape(-1)), "bf16"), torch.from_numpy(_rc_shifts(h0, h1a).view(np.int32)).cuda()
        tc = torch.full((M, N2), -7.0, dtype=torch.bfloat16, device="cuda")
        L.call("mv_conv1x1_chain_rc0_fwd", d["t20"].data_ptr(), d["x0"].data_ptr(), wf0d.data_ptr(), sh0.data_ptr(), tc.data_ptr(),
               M, C, K, N2, 1, _stream())
        kern_rc0 = L.last_kernel()
        y1p = torch.empty_like(y1)
        t1p = torch.empty_like(t1)
        L.call("mv_conv1x1_chain_fwd", d["t21"].data_ptr(), d["w31"].data_ptr(), f["s31"].data_ptr(), f["h31"].data_ptr(),
               y0.data_ptr(), y1p.data_ptr(), d["w1n"].data_ptr(), f["s1n"].data_ptr(), f["h1n"].data_ptr(), t1p.data_ptr(),
               M, C, K, N2, 1, _stream())
        torch.cuda.synchronize()
        a = _cmp(host(y1), y1ref, TOL_BF16)
        b = _cmp(host(t1), t1ref, TOL_BF16)
        dy1 = float((y1.float() - y1p.float()).abs().max())             # scales folded into bf16 rows: close to the pair, not identical
        same_y = dy1 <= a["lim"]
        dt1 = float((t1.float() - t1p.float()).abs().max())
        noy_same = bool(torch.equal(ta, tb))
        t1aref = O.relu((bf(y0ref).astype(f64) @ w1a.astype(f64).T) * s1a + h1a)
        c0 = _cmp(host(tc), t1aref, TOL_BF16)
        d0 = float((tc.float() - ta.float()).abs().max())
        return {"ok": a["ok"] and b["ok"] and same_y and dt1 <= b["lim"] and noy_same and c0["ok"] and d0 <= c0["lim"],
                "err": max(a["err"], b["err"], c0["err"]), "lim": a["lim"], "y1_vs_pair": dy1, "t1_vs_pair": dt1,
                "dual_chain_without_y_same_t1": noy_same, "rc0_t1_vs_dual_chain": d0, "kernel": kern, "kernel_noy": kern_noy,
                "kernel_rc0": kern_rc0}
    return run


def chain_sub_case(N, H, W, seed=0):
    """mv_conv1x1_chain_sub_fwd: as mv_conv1x1_chain_fwd (N2 = 128) but only the pixels with even (h, w) of y are written, compactly
    -- against the full launch: y_sub must equal y[:, ::2, ::2] bit for bit, t1 must be identical, and nothing beyond y_sub's
    extent may be touched."""
    def run():
        L = _lib()
        rng = _rng(seed)
        C, K, N2 = 64, 256, 128
        M = N * H * W
        x = bf(rng.standard_normal((M, C)))
        w3 = bf(rng.standard_normal((K, C)) / np.sqrt(C))
        s3 = rng.uniform(0.5, 1.5, K).astype(np.float32)
        h3 = (0.1 * rng.standard_normal(K)).astype(np.float32)
        r = bf(rng.standard_normal((M, K)))
        w1 = bf(rng.standard_normal((N2, K)) / np.sqrt(K))
        s1 = rng.uniform(0.5, 1.5, N2).astype(np.float32)
        h1 = (0.1 * rng.standard_normal(N2)).astype(np.float32)
        if not L.load().mv_conv1x1_chain_sub_supported(N, H, W, C, K, N2, 1):
            return {"ok": False, "err": "mv_conv1x1_chain_sub_supported says no"}
        d = {k: dev(v, "bf16") for k, v in dict(x=x, w3=w3, r=r, w1=w1).items()}
        f = {k: dev(v, "fp32") for k, v in dict(s3=s3, h3=h3, s1=s1, h1=h1).items()}
        y = torch.empty((N, H, W, K), dtype=torch.bfloat16, device="cuda")
        t1 = torch.empty((M, N2), dtype=torch.bfloat16, device="cuda")
        L.call("mv_conv1x1_chain_fwd", d["x"].data_ptr(), d["w3"].data_ptr(), f["s3"].data_ptr(), f["h3"].data_ptr(), d["r"].data_ptr(),
               y.data_ptr(), d["w1"].data_ptr(), f["s1"].data_ptr(), f["h1"].data_ptr(), t1.data_ptr(), M, C, K, N2, 1, _stream())
        nsub = N * (H // 2) * (W // 2) * K
        ybuf = torch.full((nsub + 4096,), -7.0, dtype=torch.bfloat16, device="cuda")        # guard band behind the compact map
        t1s = torch.full((M, N2), -7.0, dtype=torch.bfloat16, device="cuda")
        L.call("mv_conv1x1_chain_sub_fwd", d["x"].data_ptr(), d["w3"].data_ptr(), f["s3"].data_ptr(), f["h3"].data_ptr(), d["r"].data_ptr(),
               ybuf.data_ptr(), d["w1"].data_ptr(), f["s1"].data_ptr(), f["h1"].data_ptr(), t1s.data_ptr(), N, H, W, C, K, N2, 1, _stream())
        kern = L.last_kernel()
        torch.cuda.synchronize()
        ysub = ybuf[:nsub].view(N, H // 2, W // 2, K)
        same_y = bool(torch.equal(ysub, y[:, ::2, ::2].contiguous()))
        same_t = bool(torch.equal(t1, t1s))
        guard = bool((ybuf[nsub:] == -7.0).all())
        yref = O.relu((x.astype(np.float64) @ w3.astype(np.float64).T) * s3 + h3 + r).reshape(N, H, W, K)[:, ::2, ::2]
        a = _cmp(host(ysub), yref, TOL_BF16)
        return {"ok": a["ok"] and same_y and same_t and guard, "err": a["err"], "lim": a["lim"], "y_sub_equals_strided_full": same_y,
                "t1_identical": same_t, "guard_band_untouched": guard, "kernel": kern}
    return run


def chain_res_case(N, H, W, sub, seed=0):
    """mv_conv1x1_chain_res_fwd (a bottleneck boundary with the identity read from memory, in the accumulator-layout style of
    csrc/chain_rc.hip; N2 = 128; sub = 2: y written only at even (h, w)) vs the oracle with fp32 scales, and within the same bound of
    mv_conv1x1_chain_fwd on the same operands; nothing behind the (compact) y may be touched."""
    def run():
        L = _lib()
        rng = _rng(seed)
        C, K, N2 = 64, 256, 128
        M = N * H * W
        f64 = np.float64
        x = bf(rng.standard_normal((M, C)))
        w3 = bf(rng.standard_normal((K, C)) / np.sqrt(C))
        s3 = rng.uniform(0.5, 1.5, K).astype(np.float32)
        s3[::5] *= -1.0
        h3 = (0.1 * rng.standard_normal(K)).astype(np.float32)
        r = bf(rng.standard_normal((M, K)))
        w1 = bf(rng.standard_normal((N2, K)) / np.sqrt(K))
        s1 = rng.uniform(0.5, 1.5, N2).astype(np.float32)
        h1 = (0.1 * rng.standard_normal(N2)).astype(np.float32)
        if not L.load().mv_conv1x1_chain_res_supported(N, H, W, C, K, N2, sub, 1):
            return {"ok": False, "err": "mv_conv1x1_chain_res_supported says no"}
        yref = O.relu((x.astype(f64) @ w3.astype(f64).T) * s3 + h3 + r)
        t1ref = O.relu((bf(yref).astype(f64) @ w1.astype(f64).T) * s1 + h1)
        w3s, w1s = bf(w3.astype(np.float32) * s3[:, None]), bf(w1.astype(np.float32) * s1[:, None])
        T2 = N2 // 32
        fr = np.zeros((K // 32, 4 + 2 * T2, 64, 8), np.float32)
        for c in range(K // 32):
            for lane in range(64):
                fh, rr = lane // 32, lane % 32
                for kk in range(4):
                    fr[c, kk, lane] = w3s[32 * c + rr, 16 * kk + 8 * fh:16 * kk + 8 * fh + 8]
                for s_ in range(2):
                    for a2 in range(T2):
                        for i in range(8):
                            fr[c, 4 + T2 * s_ + a2, lane, i] = w1s[32 * a2 + rr, 32 * c + 8 * (2 * s_ + i // 4) + 4 * fh + i % 4]
        d = {k: dev(v, "bf16") for k, v in dict(x=x, w3=w3, r=r, w1=w1, wf=bf(fr.reshape(-1))).items()}
        f = {k: dev(v, "fp32") for k, v in dict(s3=s3, h3=h3, s1=s1, h1=h1).items()}
        sh = torch.from_numpy(_rc_shifts(h3, h1).view(np.int32)).cuda()
        ny = N * (H // 2) * (W // 2) * K if sub else M * K
        ybuf = torch.full((ny + 4096,), -7.0, dtype=torch.bfloat16, device="cuda")
        t1 = torch.full((M, N2), -7.0, dtype=torch.bfloat16, device="cuda")
        L.call("mv_conv1x1_chain_res_fwd", d["x"].data_ptr(), d["r"].data_ptr(), d["wf"].data_ptr(), sh.data_ptr(), ybuf.data_ptr(),
               t1.data_ptr(), N, H, W, C, K, N2, sub, 1, _stream())
        kern = L.last_kernel()
        y2 = torch.empty((M, K), dtype=torch.bfloat16, device="cuda")
        t2 = torch.empty((M, N2), dtype=torch.bfloat16, device="cuda")
        L.call("mv_conv1x1_chain_fwd", d["x"].data_ptr(), d["w3"].data_ptr(), f["s3"].data_ptr(), f["h3"].data_ptr(), d["r"].data_ptr(),
               y2.data_ptr(), d["w1"].data_ptr(), f["s1"].data_ptr(), f["h1"].data_ptr(), t2.data_ptr(), M, C, K, N2, 1, _stream())
        torch.cuda.synchronize()
        if sub:
            got_y = ybuf[:ny].view(N, H // 2, W // 2, K)
            ref_y = yref.reshape(N, H, W, K)[:, ::2, ::2]
            pair_y = y2.view(N, H, W, K)[:, ::2, ::2]
        else:
            got_y, ref_y, pair_y = ybuf[:ny].view(M, K), yref, y2
        a = _cmp(host(got_y), ref_y, TOL_BF16)
        b = _cmp(host(t1), t1ref, TOL_BF16)
        dy = float((got_y.float() - pair_y.float()).abs().max())
        dt = float((t1.float() - t2.float()).abs().max())
        guard = bool((ybuf[ny:] == -7.0).all())
        return {"ok": a["ok"] and b["ok"] and dy <= a["lim"] and dt <= b["lim"] and guard, "err": max(a["err"], b["err"]), "lim": a["lim"],
                "y_vs_chain1x1": dy, "t1_vs_chain1x1": dt, "guard_band_untouched": guard, "kernel": kern}
    return run


def dual_case(N, Ho, Wo, C1, C2, K, stride, act=1, seed=0, flags=(), splitk=False):
    """mv_conv1x1_dual_fwd: a dense pointwise layer on x and a strided pointwise layer on x2 accumulated in one GEMM
    (scales folded into the weight rows) vs the oracle's two convolutions with fp32 scales (resnet.py:144-162, 295-303)."""
    def run():
        L = _lib()
        rng = _rng(seed)
        H2, W2 = (Ho - 1) * stride + 1 + int(rng.integers(0, stride)), (Wo - 1) * stride + 1 + int(rng.integers(0, stride))
        x = bf(rng.standard_normal((N, Ho, Wo, C1)))
        x2 = bf(rng.standard_normal((N, H2, W2, C2)))
        w3 = bf(rng.standard_normal((K, C1)) / np.sqrt(C1))
        wd = bf(rng.standard_normal((K, C2)) / np.sqrt(C2))
        s3 = rng.uniform(0.5, 1.5, K).astype(np.float32)
        sd = rng.uniform(0.5, 1.5, K).astype(np.float32)
        h = (0.1 * rng.standard_normal(K)).astype(np.float32)
        M = N * Ho * Wo
        if not L.load().mv_conv1x1_dual_supported(M, C1, C2, K, 1):
            return {"ok": False, "err": "mv_conv1x1_dual_supported says no"}
        f64 = np.float64
        xs = x2[:, ::stride, ::stride][:, :Ho, :Wo]
        ref = (x.reshape(M, C1).astype(f64) @ w3.astype(f64).T) * s3 + (xs.reshape(M, C2).astype(f64) @ wd.astype(f64).T) * sd + h
        if act == 1:
            ref = O.relu(ref)
        wcat = bf(np.concatenate([w3.astype(np.float32) * s3[:, None], wd.astype(np.float32) * sd[:, None]], axis=1))
        xd, x2d, wd_, hd = dev(x, "bf16"), dev(x2, "bf16"), dev(wcat, "bf16"), dev(h, "fp32")
        y = torch.full((M, K), -7.0, dtype=torch.bfloat16, device="cuda")
        for f in flags:
            L.set_flag(f.split("=")[0], int(f.split("=")[1]) if "=" in f else 1)
        extra = {}
        try:
            def launch():
                L.call("mv_conv1x1_dual_fwd", xd.data_ptr(), x2d.data_ptr(), wd_.data_ptr(), None, hd.data_ptr(), y.data_ptr(), N, Ho,
                       Wo, C1, H2, W2, C2, stride, K, act, 1, _stream())
            if splitk:
                kern = _splitk_runs(L, launch, y, M, K, C1 + C2, extra)
                if kern is None:
                    return extra
            else:
                launch()
                kern = L.last_kernel()
        finally:
            for f in flags:
                L.set_flag(f.split("=")[0], 0)
        torch.cuda.synchronize()
        info = _cmp(host(y), ref, TOL_BF16)
        info["kernel"] = kern
        info.update(extra)
        if splitk:
            info["ok"] = info["ok"] and extra["splitk"] and extra["repeatable"] and extra["sync_clean"]
        return info
    return run


def conv_nchw_case(N, C, H, W, K, R, S, stride, pad, act=0, xdtype="fp32", tokens=False, generic=False, seed=0,
                   v0=False, f32out=False):
    """`f32out`: mv_conv2d_nchw_f32out_fwd -- bf16 operands, fp32 rows (the ViT tokens that start the fp32 residual stream): the
    result must match the exact product of the bf16 operands to fp32 accuracy, i.e. it was NOT rounded to bf16."""
    def run():
        L = _lib()
        rng = _rng(seed)
        x = rng.random((N, C, H, W), dtype=np.float32)
        if xdtype == "bf16":
            x = bf(x)
        w = bf(rng.standard_normal((K, C, R, S)) / np.sqrt(C * R * S))
        sc = rng.uniform(0.5, 1.5, K).astype(np.float32)
        sf = (0.1 * rng.standard_normal(K)).astype(np.float32)
        Ho = (H + 2 * pad - R) // stride + 1
        Wo = (W + 2 * pad - S) // stride + 1
        P = Ho * Wo
        xr = bf(x)   # the kernel rounds the image operand to bf16 for the MFMA
        ref = np.stack([O.conv2d(xr[i], w, None, stride, pad) for i in range(N)]) * sc[None, :, None, None] \
            + sf[None, :, None, None]
        if act == 1:
            ref = O.relu(ref)
        ref = ref.reshape(N, K, P).transpose(0, 2, 1)          # rows [N, P, K]
        xd = dev(x, xdtype)
        wd = dev(w, "bf16")
        scd, sfd = dev(sc, "fp32"), dev(sf, "fp32")
        if tokens:
            T = P + 1
            pos = rng.standard_normal((T, K)).astype(np.float32)
            posd = dev(pos, "fp32")
            y = torch.zeros((N, T, K), dtype=torch.float32 if f32out else torch.bfloat16, device="cuda")
            ref = np.concatenate([np.zeros((N, 1, K), np.float32), ref + pos[None, 1:]], 1)
            targs = (T, 1, posd.data_ptr())
        else:
            y = torch.empty((N, P, K), dtype=torch.float32 if f32out else torch.bfloat16, device="cuda")
            targs = (0, 0, None)
        L.set_flag("force_generic", 1 if generic else 0)
        L.set_flag("stem_v0", 1 if v0 else 0)
        try:
            if f32out:
                if not L.load().mv_conv2d_nchw_f32out_supported(C, H, W, K, R, S, stride, stride, pad, pad, DT[xdtype]):
                    return {"ok": False, "err": "mv_conv2d_nchw_f32out_supported says no"}
                L.call("mv_conv2d_nchw_f32out_fwd", xd.data_ptr(), wd.data_ptr(), scd.data_ptr(), sfd.data_ptr(), y.data_ptr(),
                       N, C, H, W, K, R, S, stride, stride, pad, pad, act, DT[xdtype], *targs, _stream())
                kern = L.last_kernel()
                torch.cuda.synchronize()
                info = _cmp(host(y), ref, TOL_F32)                  # bf16 operands, exact products, fp32 accumulation and store
                info["kernel"] = kern
                return info
            L.call("mv_conv2d_nchw_fwd", xd.data_ptr(), wd.data_ptr(), scd.data_ptr(), sfd.data_ptr(), y.data_ptr(),
                   N, C, H, W, K, R, S, stride, stride, pad, pad, act, DT[xdtype], 1, *targs, _stream())
            kern = L.last_kernel()
        finally:
            L.set_flag("force_generic", 0)
            L.set_flag("stem_v0", 0)
        torch.cuda.synchronize()
        info = _cmp(host(y), ref, TOL_BF16)
        info["kernel"] = kern
        return info
    return run


def linear_case(M, K, N, act=0, res=False, dtype="bf16", out="same", generic=False, seed=0, tile=0, flags=(), splitk=False):
    def run():
        L = _lib()
        rng = _rng(seed)
        q = bf if dtype == "bf16" else (lambda a: np.asarray(a, np.float32))
        x = q(rng.standard_normal((M, K)))
        w = q(rng.standard_normal((N, K)) / np.sqrt(K))
        b = (0.1 * rng.standard_normal(N)).astype(np.float32)
        odt = dtype if out == "same" else out
        qo = bf if odt == "bf16" else (lambda a: np.asarray(a, np.float32))
        r = qo(rng.standard_normal((M, N))) if res else None
        ref = x.astype(np.float64) @ w.astype(np.float64).T + b
        if r is not None:
            ref = ref + r
        if act == 1:
            ref = O.relu(ref)
        elif act == 2:
            ref = O.gelu_tanh(ref)
        xd, wd, bd = dev(x, dtype), dev(w, dtype), dev(b, "fp32")
        rd = None if r is None else dev(r, odt)
        y = torch.empty((M, N), dtype=torch.bfloat16 if odt == "bf16" else torch.float32, device="cuda")
        L.set_flag("force_generic", 1 if generic else 0)
        L.set_flag("igemm_tile", tile)
        for f in flags:
            L.set_flag(f.split("=")[0], int(f.split("=")[1]) if "=" in f else 1)
        extra = {}
        try:
            def launch():
                L.call("mv_linear_fwd", xd.data_ptr(), wd.data_ptr(), None, bd.data_ptr(),
                       None if rd is None else rd.data_ptr(), y.data_ptr(), M, N, K, act, DT[dtype], DT[odt], _stream())
            if splitk:
                kern = _splitk_runs(L, launch, y, M, N, K, extra)
                if kern is None:
                    return extra
            else:
                launch()
                kern = L.last_kernel()
        finally:
            L.set_flag("force_generic", 0)
            L.set_flag("igemm_tile", 0)
            for f in flags:
                L.set_flag(f.split("=")[0], 0)
        torch.cuda.synchronize()
        info = _cmp(host(y), ref, TOL_BF16 if odt == "bf16" else TOL_F32)
        info["kernel"] = kern
        info.update(extra)
        if splitk:
            info["ok"] = info["ok"] and extra["splitk"] and extra["repeatable"] and extra["sync_clean"]
        return info
    return run


def linear_f32_head_case(M, K, N, seed=0):
    """mv_linear_fwd with fp32 in / fp32 out, M <= 1024: the exact-fp32 MFMA head (skinny_f32) vs float64."""
    def run():
        L = _lib()
        rng = _rng(seed)
        x = rng.standard_normal((M, K)).astype(np.float32)
        w = (rng.standard_normal((N, K)) / np.sqrt(K)).astype(np.float32)
        b = (0.1 * rng.standard_normal(N)).astype(np.float32)
        ref = x.astype(np.float64) @ w.astype(np.float64).T + b
        xd, wd, bd = dev(x, "fp32"), dev(w, "fp32"), dev(b, "fp32")
        y = torch.empty((M, N), dtype=torch.float32, device="cuda")
        L.call("mv_linear_fwd", xd.data_ptr(), wd.data_ptr(), None, bd.data_ptr(), None, y.data_ptr(), M, N, K, 0, 0, 0, _stream())
        kern = L.last_kernel()
        torch.cuda.synchronize()
        info = _cmp(host(y), ref, 2e-5)
        info["kernel"] = kern
        info["ok"] = info["ok"] and "f32" in kern
        return info
    return run


def linear_split_case(M, K, N, out="fp32", seed=0, splitk=False):
    """mv_linear_split_fwd: x (bf16) . (w_hi + w_lo)^T with fp32 accumulation vs float64 on the UN-rounded fp32 weights:
    the error must be far below what bf16 weights alone would give."""
    def run():
        L = _lib()
        rng = _rng(seed)
        x = bf(rng.standard_normal((M, K)))
        w = (rng.standard_normal((N, K)) / np.sqrt(K)).astype(np.float32)
        b = (0.1 * rng.standard_normal(N)).astype(np.float32)
        if not L.load().mv_linear_split_supported(M, N, K, 1):
            return {"ok": False, "err": "mv_linear_split_supported says no"}
        ref = x.astype(np.float64) @ w.astype(np.float64).T + b
        ref_bf16w = x.astype(np.float64) @ bf(w).astype(np.float64).T + b
        hi = torch.from_numpy(w).to(torch.bfloat16)
        lo = (torch.from_numpy(w) - hi.float()).to(torch.bfloat16)
        wd = torch.cat([hi, lo], 1).contiguous().cuda()
        xd, bd = dev(x, "bf16"), dev(b, "fp32")
        odt = torch.float32 if out == "fp32" else torch.bfloat16
        y = torch.empty((M, N), dtype=odt, device="cuda")
        def launch():
            L.call("mv_linear_split_fwd", xd.data_ptr(), wd.data_ptr(), None, bd.data_ptr(), None, y.data_ptr(), M, N, K, 0, 1,
                   0 if out == "fp32" else 1, _stream())
        extra = {}
        if splitk:
            kern = _splitk_runs(L, launch, y, M, N, 2 * K, extra)
            if kern is None:
                return extra
        else:
            launch()
            kern = L.last_kernel()
        torch.cuda.synchronize()
        info = _cmp(host(y), ref, 2e-4 if out == "fp32" else TOL_BF16)
        info["kernel"] = kern
        info.update(extra)
        if splitk:
            info["ok"] = info["ok"] and extra["splitk"] and extra["repeatable"] and extra["sync_clean"]
        info["err_if_bf16_weights"] = float(np.abs(ref_bf16w - ref).max())
        return info
    return run


def dwconv_case(N, H, W, C, R=3, stride=1, pad=1, dil=1, act=1, scale=True, seed=0, flag=None):
    """mv_dwconv2d_nhwc_fwd (depthwise conv + folded BN + act) vs the oracle's conv2d with groups == channels."""
    def run():
        L = _lib()
        if flag:
            L.set_flag(flag, 1)
        try:
            return body(L)
        finally:
            if flag:
                L.set_flag(flag, 0)

    def body(L):
        rng = _rng(seed)
        x = bf(rng.standard_normal((N, C, H, W)))
        w = bf(rng.standard_normal((C, 1, R, R)) / np.sqrt(R * R))
        sc = rng.uniform(0.5, 1.5, C).astype(np.float32) if scale else None
        sf = (0.1 * rng.standard_normal(C)).astype(np.float32)
        if not L.load().mv_dwconv2d_supported(C, C, C, R, R, 1, 1):
            return {"ok": False, "err": "mv_dwconv2d_supported says no"}
        ref = np.stack([O.conv2d(x[i], w, None, stride, pad, dil, C) for i in range(N)]).astype(np.float64)
        if scale:
            ref = ref * sc[None, :, None, None]
        ref = ref + sf[None, :, None, None]
        if act == 1:
            ref = np.maximum(ref, 0)
        elif act == 3:
            ref = O.hard_swish(ref).astype(np.float64)
        elif act == 6:
            ref = O.silu(ref).astype(np.float64)
        elif act != 0:
            raise ValueError(f"dwconv_case: no reference for activation {act}")
        Ho = ref.shape[2]
        xd = dev(np.ascontiguousarray(x.transpose(0, 2, 3, 1)), "bf16")
        wd = dev(np.ascontiguousarray(w[:, 0].transpose(1, 2, 0)), "bf16")
        scd, sfd = (dev(sc, "fp32") if scale else None), dev(sf, "fp32")
        y = torch.empty((N, Ho, ref.shape[3], C), dtype=torch.bfloat16, device="cuda")
        L.call("mv_dwconv2d_nhwc_fwd", xd.data_ptr(), wd.data_ptr(), scd.data_ptr() if scale else None, sfd.data_ptr(), y.data_ptr(),
               N, H, W, C, R, R, stride, stride, pad, pad, dil, dil, act, 1, 1, _stream())
        kern = L.last_kernel()
        torch.cuda.synchronize()
        info = _cmp(host(y).transpose(0, 3, 1, 2), ref, TOL_BF16)
        info["kernel"] = kern
        return info
    return run


def conv_grouped64_case(N, H, W, C, groups, R=3, stride=1, pad=1, dil=1, act=1, res=False, seed=0):
    """mv_conv2d_nhwc_grouped64_fwd (grouped conv as block-diagonal 64-channel super-groups on the MFMA) vs the oracle's grouped
    conv2d (resnet.py:17-27 with groups > 1)."""
    def run():
        L = _lib()
        rng = _rng(seed)
        cg = C // groups
        x = bf(rng.standard_normal((N, C, H, W)))
        w = bf(rng.standard_normal((C, cg, R, R)) / np.sqrt(cg * R * R))
        sc = rng.uniform(0.5, 1.5, C).astype(np.float32)
        sf = (0.1 * rng.standard_normal(C)).astype(np.float32)
        if not L.load().mv_conv2d_grouped64_supported(C, C, R, R, groups, 1, 1):
            return {"ok": False, "err": "mv_conv2d_grouped64_supported says no"}
        Ho = (H + 2 * pad - dil * (R - 1) - 1) // stride + 1
        r = bf(rng.standard_normal((N, C, Ho, Ho))) if res else None
        ref = np.stack([O.conv2d(x[i], w, None, stride, pad, dil, groups) for i in range(N)]).astype(np.float64)
        ref = ref * sc[None, :, None, None] + sf[None, :, None, None]
        if res:
            ref = ref + r
        if act == 1:
            ref = np.maximum(ref, 0)
        win = int(L.load().mv_conv2d_grouped64_window(C, groups))
        w64 = np.zeros((C, R, R, win), np.float32)
        for k in range(C):
            g0 = (k // cg) * cg - ((k // 64 * 64) // cg) * cg
            w64[k, :, :, g0:g0 + cg] = w[k].transpose(1, 2, 0)
        xd = dev(np.ascontiguousarray(x.transpose(0, 2, 3, 1)), "bf16")
        wd, scd, sfd = dev(w64, "bf16"), dev(sc, "fp32"), dev(sf, "fp32")
        rd = dev(np.ascontiguousarray(r.transpose(0, 2, 3, 1)), "bf16") if res else None
        y = torch.empty((N, Ho, Ho, C), dtype=torch.bfloat16, device="cuda")
        L.call("mv_conv2d_nhwc_grouped64_fwd", xd.data_ptr(), wd.data_ptr(), scd.data_ptr(), sfd.data_ptr(),
               rd.data_ptr() if res else None, y.data_ptr(), N, H, W, C, C, R, R, stride, stride, pad, pad, dil, dil, groups, act, 1, 1,
               _stream())
        kern = L.last_kernel()
        torch.cuda.synchronize()
        info = _cmp(host(y).transpose(0, 3, 1, 2), ref, TOL_BF16)
        info["kernel"] = kern
        return info
    return run


def eltwise_act_case(name, code, dtype="bf16", seed=0):
    """mv_eltwise_fwd with the extended activations vs the oracle (jax.nn.hard_swish / hard_sigmoid / sigmoid / silu)."""
    def run():
        L = _lib()
        rng = _rng(seed)
        q = bf if dtype == "bf16" else (lambda a: np.asarray(a, np.float32))
        x = q(rng.uniform(-8, 8, 100003))
        ref = getattr(O, name)(x)
        xd = dev(x, dtype)
        y = torch.empty_like(xd)
        L.call("mv_eltwise_fwd", xd.data_ptr(), y.data_ptr(), x.size, code, DT[dtype], _stream())
        torch.cuda.synchronize()
        return _cmp(host(y), ref, TOL_BF16 if dtype == "bf16" else 2e-6)
    return run


def fused_act_refused_case():
    """the matrix-core entries implement none / relu / gelu only: anything else is MV_E_INVALID, not a silently un-activated result."""
    def run():
        L = _lib()
        x = torch.zeros((256, 64), dtype=torch.bfloat16, device="cuda")
        w = torch.zeros((64, 64), dtype=torch.bfloat16, device="cuda")
        y = torch.zeros((256, 64), dtype=torch.bfloat16, device="cuda")
        rc = L.load().mv_linear_fwd(x.data_ptr(), w.data_ptr(), None, None, None, y.data_ptr(), 256, 64, 64, 3, 1, 1, _stream())
        return {"ok": rc == -1, "err": float(rc)}
    return run


def channel_scale_case(N, HW, C, dtype="bf16", seed=0):
    def run():
        L = _lib()
        rng = _rng(seed)
        q = bf if dtype == "bf16" else (lambda a: np.asarray(a, np.float32))
        x, s = q(rng.standard_normal((N, HW, C))), q(rng.uniform(0, 1, (N, C)))
        xd, sd_ = dev(x, dtype), dev(s, dtype)
        y = torch.empty_like(xd)
        L.call("mv_channel_scale_nhwc_fwd", xd.data_ptr(), sd_.data_ptr(), y.data_ptr(), N, HW, C, DT[dtype], _stream())
        torch.cuda.synchronize()
        return _cmp(host(y), x.astype(np.float64) * s[:, None, :], TOL_BF16 if dtype == "bf16" else 1e-6)
    return run


def resize_case(N, h, w, C, H, W, dtype="bf16", nchw=True, seed=0):
    """mv_resize_bilinear_nhwc_fwd vs the restatement of jax.image.resize (oracle.np_ops.resize_bilinear)."""
    def run():
        L = _lib()
        rng = _rng(seed)
        q = bf if dtype == "bf16" else (lambda a: np.asarray(a, np.float32))
        x = q(rng.standard_normal((N, C, h, w)))
        ref = np.stack([O.resize_bilinear(x[i], (H, W)) for i in range(N)])
        xd = dev(np.ascontiguousarray(x.transpose(0, 2, 3, 1)), dtype)
        odt = torch.float32 if nchw else xd.dtype
        y = torch.empty((N, C, H, W) if nchw else (N, H, W, C), dtype=odt, device="cuda")
        L.call("mv_resize_bilinear_nhwc_fwd", xd.data_ptr(), y.data_ptr(), N, h, w, C, H, W, DT[dtype], 0 if odt == torch.float32 else 1,
               1 if nchw else 0, _stream())
        torch.cuda.synchronize()
        got = host(y) if nchw else host(y).transpose(0, 3, 1, 2)
        return _cmp(got, ref, 1e-5 if nchw else (TOL_BF16 if dtype == "bf16" else 1e-5))
    return run


def resize_down_refused_case():
    def run():
        L = _lib()
        x = torch.zeros((1, 8, 8, 4), dtype=torch.float32, device="cuda")
        y = torch.zeros((1, 4, 4, 4), dtype=torch.float32, device="cuda")
        rc = L.load().mv_resize_bilinear_nhwc_fwd(x.data_ptr(), y.data_ptr(), 1, 8, 8, 4, 4, 4, 0, 0, 0, _stream())
        return {"ok": rc == -2, "err": float(rc)}
    return run


def copy_rows_case(rows, parts, dtype="bf16", seed=0):
    """channel concatenation through mv_copy_rows (one call per source) vs np.concatenate."""
    def run():
        L = _lib()
        rng = _rng(seed)
        q = bf if dtype == "bf16" else (lambda a: np.asarray(a, np.float32))
        srcs = [q(rng.standard_normal((rows, c))) for c in parts]
        ref = np.concatenate(srcs, 1)
        es = 2 if dtype == "bf16" else 4
        y = torch.zeros((rows, sum(parts)), dtype=torch.bfloat16 if dtype == "bf16" else torch.float32, device="cuda")
        off, keep = 0, []
        for a in srcs:
            d = dev(a, dtype)
            keep.append(d)
            c = a.shape[1]
            L.call("mv_copy_rows", d.data_ptr(), y.data_ptr() + off * es, rows, c * es, c * es, sum(parts) * es, _stream())
            off += c
        torch.cuda.synchronize()
        return _cmp(host(y), ref, 0.0)
    return run


def ln_linear_case(M, K, N, stream="fp32", act=0, seed=0, flags=()):
    """mv_ln_linear_fwd (LayerNorm folded into the consuming Linear) vs float64 LayerNorm -> Linear (swin.py:572-578)."""
    def run():
        L = _lib()
        for k, v in flags:
            L.set_flag(k, v)
        try:
            return body(L)
        finally:
            for k, _ in flags:
                L.set_flag(k, 0)

    def body(L):
        rng = _rng(seed)
        x = (rng.standard_normal((M, K)) * rng.uniform(0.5, 2.0, (M, 1)) + rng.uniform(-1, 1, (M, 1))).astype(np.float32)
        if stream == "bf16":
            x = bf(x)
        g = rng.uniform(0.5, 1.5, K).astype(np.float32)
        be = (0.1 * rng.standard_normal(K)).astype(np.float32)
        w = (rng.standard_normal((N, K)) / np.sqrt(K)).astype(np.float32)
        b = (0.1 * rng.standard_normal(N)).astype(np.float32)
        xdt = 0 if stream == "fp32" else 1
        if not L.load().mv_ln_linear_supported(M, N, K, xdt, 1):
            return {"ok": False, "err": "mv_ln_linear_supported says no"}
        ref = O.layernorm_rows(x, g, be, 1e-5).astype(np.float64) @ w.astype(np.float64).T + b
        if act == 2:
            ref = O.gelu_tanh(ref).astype(np.float64)
        xd, wd, bd = dev(x, stream), dev(bf(w * g[None, :]), "bf16"), dev((b + w @ be).astype(np.float32), "fp32")
        y = torch.empty((M, N), dtype=torch.bfloat16, device="cuda")
        L.call("mv_ln_linear_fwd", xd.data_ptr(), wd.data_ptr(), bd.data_ptr(), y.data_ptr(), M, N, K, 1e-5, act, xdt, 1, _stream())
        kern = L.last_kernel()
        torch.cuda.synchronize()
        info = _cmp(host(y), ref, TOL_BF16)
        info["kernel"] = kern
        info["ok"] = info["ok"] and "_ln_" in kern
        return info
    return run


def ln_fold_case(M, D, Kp, N2, act=0, tokens=0, seed=0, row_mean=0.0, eps=1e-6):
    """The LayerNorm between two Linears folded into their epilogues (include/eqxvision_amd.h: mv_linear_lnout_fwd ->
    mv_linear_lnin_fwd; vit.py:139-157) vs float64, all three producer forms chained as a ViT does:
      1. fp32 residual rows -> planes:  y1 = res + x . w^T + b as hi = bf16(y1), lo = bf16(y1 - hi) (hi + lo within 3e-5 of float64, hi
         a correct bf16 rounding, |lo| below half an ulp of hi) + the statistics pieces (merged here with Chan's formula, against the
         float64 mean / variance of the rows);
      2. consumer: act(LayerNorm(y1) . w2^T + b2) as the oracle computes it (O.layernorm_rows on the rows hi + lo), row-major or
         head-major (tokens > 0);
      3. planes -> planes:  y2 = y1 + x . w^T + b, checked the same way;   4. planes -> fp32 rows:  y3 = y2 + x . w^T + b.
    `row_mean`: per-row offsets of that many row standard deviations in the residual rows -- the fold rounds y, not y - mean, to bf16."""
    def run():
        L = _lib()
        rng = _rng(seed)
        x = bf(rng.standard_normal((M, Kp)).astype(np.float32))
        w = bf((rng.standard_normal((D, Kp)) / np.sqrt(Kp)).astype(np.float32))
        b = (0.1 * rng.standard_normal(D)).astype(np.float32)
        res = (rng.standard_normal((M, D)) * rng.uniform(0.5, 2.0, (M, 1))).astype(np.float32)
        res += (row_mean * np.sqrt(2.0) * rng.uniform(-1, 1, (M, 1))).astype(np.float32) + rng.uniform(-0.3, 0.3, (M, 1)).astype(np.float32)
        g = rng.uniform(0.5, 1.5, D).astype(np.float32)
        be = (0.1 * rng.standard_normal(D)).astype(np.float32)
        w2 = (rng.standard_normal((N2, D)) / np.sqrt(D)).astype(np.float32)
        b2 = (0.1 * rng.standard_normal(N2)).astype(np.float32)
        dh = 64
        if not L.load().mv_linear_lnout_supported(M, D, Kp, 1):
            return {"ok": False, "err": "mv_linear_lnout_supported says no"}
        if not L.load().mv_linear_lnin_supported(M, N2, D, tokens, dh if tokens else 0, 1):
            return {"ok": False, "err": "mv_linear_lnin_supported says no"}
        delta = x.astype(np.float64) @ w.astype(np.float64).T + b
        xd, wd, bd, rd = dev(x, "bf16"), dev(w, "bf16"), dev(b, "fp32"), dev(res, "fp32")
        P = (D + 255) // 256
        npc = np.array([min(256, D - 256 * j) for j in range(P)], np.float64)[:, None]

        def planes_out(res_hi, res_lo):
            hi = torch.empty((M, D), dtype=torch.bfloat16, device="cuda")
            lo = torch.empty((M, D), dtype=torch.bfloat16, device="cuda")
            st = torch.full((P, M, 2), float("nan"), dtype=torch.float32, device="cuda")
            L.call("mv_linear_lnout_fwd", xd.data_ptr(), wd.data_ptr(), bd.data_ptr(), res_hi.data_ptr(), None if res_lo is None else res_lo.data_ptr(),
                   hi.data_ptr(), lo.data_ptr(), st.data_ptr(), M, D, Kp, 1, _stream())
            k = L.last_kernel()
            torch.cuda.synchronize()
            return hi, lo, st, k

        def check_planes(hi, lo, st, y_ref):
            hh, ll = host(hi).astype(np.float64), host(lo).astype(np.float64)
            i = _cmp(hh + ll, y_ref, 3e-5)
            slack = i["lim"]            # the fp32 accumulation may land on the other side of a rounding boundary; values near zero
            hi_ok = bool((np.abs(hh - y_ref) <= np.abs(y_ref) * 2.0 ** -8 + slack).all() and (np.abs(ll) <= np.abs(hh) * 2.0 ** -8 + slack).all())
            sth = st.cpu().numpy().astype(np.float64)                     # [P][M][2]
            mean = sth[:, :, 0].sum(0) / D
            m2 = (sth[:, :, 1] + npc * (sth[:, :, 0] / npc - mean[None, :]) ** 2).sum(0)
            e_mean = float(np.abs(mean - y_ref.mean(1)).max())
            e_var = float((np.abs(m2 / D - y_ref.var(1)) / y_ref.var(1)).max())
            ok = bool(i["ok"] and hi_ok and np.isfinite(sth).all() and e_mean < 1e-5 * max(1.0, np.abs(y_ref).max()) and e_var < 2e-5)
            return ok, {"err": i.get("err"), "lim": i.get("lim"), "hi_is_a_bf16_rounding": hi_ok, "stats_mean_err": e_mean, "stats_var_relerr": e_var}

        # ---- 1. fp32 rows -> planes
        y1_ref = res.astype(np.float64) + delta
        hi1, lo1, st1, k1 = planes_out(rd, None)
        ok1, info = check_planes(hi1, lo1, st1, y1_ref)
        info["ok"] = ok1 and "lnout_f32res" in k1
        # ---- 2. consumer on (hi1, st1)
        y1 = host(hi1).astype(np.float64) + host(lo1).astype(np.float64)
        ref = O.layernorm_rows(y1, g, be, eps).astype(np.float64) @ w2.astype(np.float64).T + b2
        if act == 2:
            ref = O.gelu_tanh(ref).astype(np.float64)
        elif act == 1:
            ref = np.maximum(ref, 0.0)
        wf = bf(w2 * g[None, :])
        cs = wf.astype(np.float64).sum(1).astype(np.float32)
        bfold = (b2.astype(np.float64) + w2.astype(np.float64) @ be.astype(np.float64)).astype(np.float32)
        wfd, csd, bfd = dev(wf, "bf16"), dev(cs, "fp32"), dev(bfold, "fp32")
        if tokens:
            B, H = M // tokens, N2 // 3 // dh
            out = torch.empty((B, 3 * H, tokens, dh), dtype=torch.bfloat16, device="cuda")
            ref = ref.reshape(B, tokens, 3 * H, dh).transpose(0, 2, 1, 3)
        else:
            out = torch.empty((M, N2), dtype=torch.bfloat16, device="cuda")
        L.call("mv_linear_lnin_fwd", hi1.data_ptr(), st1.data_ptr(), wfd.data_ptr(), csd.data_ptr(), bfd.data_ptr(), out.data_ptr(), M, N2, D,
               float(eps), act, tokens, dh if tokens else 0, 1, _stream())
        k2 = L.last_kernel()
        torch.cuda.synchronize()
        i2 = _cmp(host(out), ref, TOL_BF16)
        info["lnin_err"], info["lnin_lim"] = i2.get("err"), i2.get("lim")
        info["ok"] = bool(info["ok"] and i2["ok"] and "lnin" in k2)
        # ---- 3. planes -> planes
        hi2, lo2, st2, k3 = planes_out(hi1, lo1)
        ok3, i3 = check_planes(hi2, lo2, st2, y1 + delta)
        info["planes_to_planes"] = i3
        info["ok"] = bool(info["ok"] and ok3 and k3.endswith("lin_lnout"))
        # ---- 4. planes -> fp32 rows
        y2 = host(hi2).astype(np.float64) + host(lo2).astype(np.float64)
        y3 = torch.empty((M, D), dtype=torch.float32, device="cuda")
        L.call("mv_linear_lnout_fwd", xd.data_ptr(), wd.data_ptr(), bd.data_ptr(), hi2.data_ptr(), lo2.data_ptr(), y3.data_ptr(), None, None,
               M, D, Kp, 1, _stream())
        k4 = L.last_kernel()
        torch.cuda.synchronize()
        i4 = _cmp(y3.cpu().numpy(), y2 + delta, 2e-5)
        info["planes_to_f32_err"] = i4.get("err")
        info["kernel"] = "+".join((k1, k2, k3, k4))
        info["ok"] = bool(info["ok"] and i4["ok"] and "splitres" in k4)
        return info
    return run


def se_scale_case(N, H, W, C, S, act1=1, act2=5, seed=0):
    """mv_se_scale_fwd (SqueezeExcitation's scale vector in one launch, layers/squeeze.py:47-60) vs float64:
    act2(b2 + w2 . act1(b1 + w1 . mean_hw x))."""
    def run():
        L = _lib()
        rng = _rng(seed)
        acts = {0: lambda v: v, 1: O.relu, 3: O.hard_swish, 4: O.hard_sigmoid, 5: O.sigmoid, 6: O.silu}
        x = bf(rng.standard_normal((N, H * W, C)) + rng.uniform(-1, 1, (N, 1, C)))
        w1, w2 = bf(rng.standard_normal((S, C)) / np.sqrt(C)), bf(rng.standard_normal((C, S)) / np.sqrt(S))
        b1, b2 = (0.2 * rng.standard_normal(S)).astype(np.float32), (0.2 * rng.standard_normal(C)).astype(np.float32)
        L.set_flag("se_fused_always", 1)
        try:
            return body(L, rng, acts, x, w1, w2, b1, b2)
        finally:
            L.set_flag("se_fused_always", 0)

    def body(L, rng, acts, x, w1, w2, b1, b2):
        if not L.load().mv_se_scale_supported(C, S, 1):
            return {"ok": False, "err": "mv_se_scale_supported says no"}
        p = x.astype(np.float64).mean(1)
        h = np.asarray(acts[act1]((p @ w1.astype(np.float64).T + b1).astype(np.float32)), np.float64)
        ref = np.asarray(acts[act2]((h @ w2.astype(np.float64).T + b2).astype(np.float32)), np.float64)
        xd, w1d, w2d = dev(x, "bf16"), dev(w1, "bf16"), dev(np.ascontiguousarray(w2.T), "bf16")      # w2 handed over transposed
        b1d, b2d = dev(b1, "fp32"), dev(b2, "fp32")
        y = torch.full((N, C), -7.0, dtype=torch.bfloat16, device="cuda")
        L.call("mv_se_scale_fwd", xd.data_ptr(), w1d.data_ptr(), b1d.data_ptr(), w2d.data_ptr(), b2d.data_ptr(), y.data_ptr(), N, H * W, C, S,
               act1, act2, 1, _stream())
        kern = L.last_kernel()
        torch.cuda.synchronize()
        info = _cmp(host(y), ref, TOL_BF16)
        info["kernel"] = kern
        return info
    return run


def moments_case(rows, C, dtype="bf16", seed=0):
    """mv_channel_moments_fwd (per-channel batch sums of training-mode BatchNorm, two passes) vs float64 sums."""
    def run():
        L = _lib()
        rng = _rng(seed)
        x = (rng.standard_normal((rows, C)) * rng.uniform(0.5, 2.0, (1, C)) + rng.uniform(-2, 2, (1, C))).astype(np.float32)
        if dtype == "bf16":
            x = bf(x)
        dtc = 1 if dtype == "bf16" else 0
        if not L.load().mv_channel_moments_supported(rows, C, dtc):
            return {"ok": False, "err": "mv_channel_moments_supported says no"}
        xd = dev(x, dtype)
        ws = torch.empty((int(L.load().mv_channel_moments_ws(C)),), dtype=torch.float32, device="cuda")
        s1 = torch.full((C,), -7.0, dtype=torch.float32, device="cuda")
        s2 = torch.full((C,), -7.0, dtype=torch.float32, device="cuda")
        L.call("mv_channel_moments_fwd", xd.data_ptr(), None, s1.data_ptr(), ws.data_ptr(), rows, C, 0, dtc, _stream())
        mean = (host(s1).astype(np.float64) / rows).astype(np.float32)
        md = dev(mean, "fp32")
        L.call("mv_channel_moments_fwd", xd.data_ptr(), md.data_ptr(), s2.data_ptr(), ws.data_ptr(), rows, C, 1, dtc, _stream())
        s1b = torch.empty_like(s1)
        L.call("mv_channel_moments_fwd", xd.data_ptr(), None, s1b.data_ptr(), ws.data_ptr(), rows, C, 0, dtc, _stream())
        torch.cuda.synchronize()
        x64 = x.astype(np.float64)
        e1 = float(np.abs(host(s1) / rows - x64.mean(0)).max())
        e2 = float(np.abs(host(s2) / rows - ((x64 - mean.astype(np.float64)) ** 2).mean(0)).max())
        return {"ok": e1 < 1e-4 and e2 < 1e-3 and bool(torch.equal(s1, s1b)), "err": max(e1, e2), "lim": 1e-3, "mean_err": e1,
                "var_err": e2, "deterministic": bool(torch.equal(s1, s1b)), "kernel": L.last_kernel()}
    return run


def moments2_case(rows, C, dtype="bf16", seed=0, off=0.3):
    """mv_channel_moments2_fwd (both moments about a shift near the mean, one pass) + mv_bn_ema_fold1_fwd vs float64: the batch
    mean / variance they imply, the EMA of the running statistics and the folded scale / shift."""
    def run():
        L = _lib()
        rng = _rng(seed)
        mu = rng.uniform(-3, 3, (1, C))
        x = (rng.standard_normal((rows, C)) * rng.uniform(0.5, 2.0, (1, C)) + mu).astype(np.float32)
        if dtype == "bf16":
            x = bf(x)
        dtc = 1 if dtype == "bf16" else 0
        rm = (mu[0] + off * rng.standard_normal(C)).astype(np.float32)           # running mean: near the batch mean, not on it
        rv = rng.uniform(0.5, 2.0, C).astype(np.float32)
        w, b = rng.uniform(0.5, 1.5, C).astype(np.float32), (0.1 * rng.standard_normal(C)).astype(np.float32)
        xd = dev(x, dtype)
        ws = torch.empty((2 * int(L.load().mv_channel_moments_ws(C)),), dtype=torch.float32, device="cuda")
        sums = torch.empty((2 * C,), dtype=torch.float32, device="cuda")
        rmd, rvd, wd, bd = (dev(a, "fp32") for a in (rm, rv, w, b))
        sc, sh = torch.empty(C, device="cuda"), torch.empty(C, device="cuda")
        L.call("mv_channel_moments2_fwd", xd.data_ptr(), rmd.data_ptr(), sums.data_ptr(), ws.data_ptr(), rows, C, dtc, _stream())
        L.call("mv_bn_ema_fold1_fwd", sums.data_ptr(), None, float(rows), rmd.data_ptr(), rvd.data_ptr(), wd.data_ptr(), bd.data_ptr(),
               sc.data_ptr(), sh.data_ptr(), 0.99, 1e-5, C, _stream())
        torch.cuda.synchronize()
        x64 = x.astype(np.float64)
        bm, bvar = x64.mean(0), x64.var(0)
        erm, erv = 0.01 * bm + 0.99 * rm, 0.01 * bvar + 0.99 * rv
        esc = w / np.sqrt(erv + 1e-5)
        errs = {"run_mean": float(np.abs(host(rmd) - erm).max()), "run_var": float(np.abs(host(rvd) - erv).max()),
                "scale": float(np.abs(host(sc) - esc).max()), "shift": float(np.abs(host(sh) - (b - erm * esc)).max())}
        e = max(errs.values())
        return {"ok": e < 1e-4, "err": e, "lim": 1e-4, **errs, "kernel": L.last_kernel()}
    return run


def dropout_case(B, per, C, chw, p, dtype="bf16", seed=0, flag=None):
    """mv_dropout_fwd vs the oracle's eqx.nn.Dropout on JAX's bit stream (oracle.np_ops.dropout), bit for bit; x is NHWC
    [B][per / C][C], the mask indexed in the LOGICAL (C, per / C) order when `chw`."""
    def run():
        L = _lib()
        rng = _rng(seed)
        keys = np.stack([O.jax_split(np.array([seed, b], np.uint32), 2)[1] for b in range(B)])
        x = rng.standard_normal((B, per // C, C)).astype(np.float32)
        if dtype == "bf16":
            x = bf(x)
        xd = dev(x, dtype)
        kd = torch.from_numpy(keys.view(np.int32)).cuda()
        y = torch.empty_like(xd)
        if flag:
            L.set_flag(flag, 1)
        try:
            L.call("mv_dropout_fwd", xd.data_ptr(), kd.data_ptr(), y.data_ptr(), B, per, C, 1 if chw else 0, float(1.0 - p),
                   1 if dtype == "bf16" else 0, _stream())
            kern = L.last_kernel()
        finally:
            if flag:
                L.set_flag(flag, 0)
        torch.cuda.synchronize()
        want = []
        for b in range(B):
            logical = x[b].T if chw else x[b]                   # (C, HW) or the physical order
            r = O.dropout(logical, p, keys[b])
            want.append(r.T if chw else r)
        want = np.stack(want)
        if dtype == "bf16":
            want = bf(want)
        got = host(y)
        same = bool(np.array_equal(got, want))
        return {"ok": same, "err": 0.0 if same else float(np.abs(got - want).max()), "lim": 0.0, "kept": float((got != 0).mean()),
                "kernel": kern}
    return run


def ln_mlp_case(M, stream="fp32", seed=0, C=96, Hd=384):
    """mv_ln_mlp_fwd (LayerNorm -> fc1 -> GELU -> fc2 -> + x, one launch) vs the un-fused float64 restatement
    (swin.py:572-578 second line, mlps.py:54-66) with the affine LayerNorm applied the reference's way; the kernel gets the
    affine folded into fc1."""
    def run():
        L = _lib()
        rng = _rng(seed)
        x = (rng.standard_normal((M, C)) * rng.uniform(0.5, 2.0, (M, 1)) + rng.uniform(-1, 1, (M, 1))).astype(np.float32)
        if stream == "bf16":
            x = bf(x)
        g = rng.uniform(0.5, 1.5, C).astype(np.float32)
        be = (0.1 * rng.standard_normal(C)).astype(np.float32)
        w1 = (rng.standard_normal((Hd, C)) / np.sqrt(C)).astype(np.float32)
        b1 = (0.1 * rng.standard_normal(Hd)).astype(np.float32)
        w2 = (rng.standard_normal((C, Hd)) / np.sqrt(Hd)).astype(np.float32)
        b2 = (0.1 * rng.standard_normal(C)).astype(np.float32)
        xdt = 0 if stream == "fp32" else 1
        streamed = not L.load().mv_ln_mlp_supported(M, C, Hd, xdt)      # wide rows: the streamed-weights kernel (fragment order)
        if streamed and not L.load().mv_ln_mlp_stream_supported(M, C, Hd, xdt):
            return {"ok": False, "err": "neither mv_ln_mlp_supported nor mv_ln_mlp_stream_supported"}
        n = O.layernorm_rows(x, g, be, 1e-5).astype(np.float64)
        h = O.gelu_tanh(n @ w1.astype(np.float64).T + b1).astype(np.float64)
        ref = x.astype(np.float64) + h @ w2.astype(np.float64).T + b2
        w1f = bf(w1 * g[None, :])
        b1f = (b1 + w1 @ be).astype(np.float32)
        xd = dev(x, stream)
        w1d, b1d, w2d, b2d = dev(w1f, "bf16"), dev(b1f, "fp32"), dev(bf(w2), "bf16"), dev(b2, "fp32")
        y = torch.empty_like(xd)
        if streamed:
            w1fr = w1f.reshape(Hd // 256, 8, 32, C // 16, 2, 8).transpose(0, 1, 3, 4, 2, 5)
            w2fr = bf(w2).reshape(C // 32, 32, Hd // 256, 16, 2, 8).transpose(2, 0, 3, 4, 1, 5)
            w1s, w2s = dev(w1fr, "bf16"), dev(w2fr, "bf16")
            L.call("mv_ln_mlp_stream_fwd", xd.data_ptr(), w1s.data_ptr(), b1d.data_ptr(), w2s.data_ptr(), b2d.data_ptr(), y.data_ptr(),
                   M, C, Hd, 1e-5, xdt, _stream())
        else:
            L.call("mv_ln_mlp_fwd", xd.data_ptr(), w1d.data_ptr(), b1d.data_ptr(), w2d.data_ptr(), b2d.data_ptr(), y.data_ptr(),
                   M, C, Hd, 1e-5, xdt, _stream())
        kern = L.last_kernel()
        torch.cuda.synchronize()
        info = _cmp(host(y), ref, TOL_BF16)
        info["kernel"] = kern
        # the same numbers from the three separate launches (LayerNorm, fc1+GELU, fc2+residual): the fused kernel must not be
        # further from the float64 reference than they are (plus rounding noise)
        gd, bed = dev(g, "fp32"), dev(be, "fp32")
        nb = torch.empty((M, C), dtype=torch.bfloat16, device="cuda")
        L.call("mv_layernorm_fwd", xd.data_ptr(), gd.data_ptr(), bed.data_ptr(), nb.data_ptr(), M, C, 0, 1e-5, xdt, 1, _stream())
        hb = torch.empty((M, Hd), dtype=torch.bfloat16, device="cuda")
        w1u, b1u = dev(bf(w1), "bf16"), dev(b1, "fp32")
        L.call("mv_linear_fwd", nb.data_ptr(), w1u.data_ptr(), None, b1u.data_ptr(), None, hb.data_ptr(), M, Hd, C, 2, 1, 1, _stream())
        y3 = torch.empty_like(xd)
        L.call("mv_linear_fwd", hb.data_ptr(), w2d.data_ptr(), None, b2d.data_ptr(), xd.data_ptr(), y3.data_ptr(), M, C, Hd, 0, 1,
               xdt, _stream())
        torch.cuda.synchronize()
        e3 = float(np.abs(host(y3).astype(np.float64) - ref).max())
        info["err_unfused"] = e3
        info["ok"] = info["ok"] and "ln_mlp" in kern and info["err"] <= 1.5 * e3 + 1e-3
        return info
    return run


def conv_nchw_split_case(N, C, H, K, R, stride, seed=0):
    """mv_conv2d_nchw_split_fwd (entry conv with hi + lo weights) vs float64 on the un-rounded weights."""
    def run():
        L = _lib()
        rng = _rng(seed)
        x = rng.random((N, C, H, H)).astype(np.float32)
        w = (rng.standard_normal((K, C, R, R)) / np.sqrt(C * R * R)).astype(np.float32)
        b = (0.1 * rng.standard_normal(K)).astype(np.float32)
        ref = np.stack([O.conv2d(bf(x[i]).astype(np.float64), w.astype(np.float64), b.astype(np.float64), stride, 0) for i in range(N)])
        ref_bf16w = np.stack([O.conv2d(bf(x[i]).astype(np.float64), bf(w).astype(np.float64), b.astype(np.float64), stride, 0)
                              for i in range(N)])
        hi = torch.from_numpy(w).to(torch.bfloat16)
        lo = (torch.from_numpy(w) - hi.float()).to(torch.bfloat16)
        xd, bd, hid, lod = dev(x, "fp32"), dev(b, "fp32"), hi.cuda(), lo.cuda()
        Ho = (H - R) // stride + 1
        y = torch.empty((N, Ho, Ho, K), dtype=torch.bfloat16, device="cuda")
        L.call("mv_conv2d_nchw_split_fwd", xd.data_ptr(), hid.data_ptr(), lod.data_ptr(), None, bd.data_ptr(), y.data_ptr(),
               N, C, H, H, K, R, R, stride, stride, 0, 0, 0, 0, 1, _stream())
        kern = L.last_kernel()
        torch.cuda.synchronize()
        got = host(y).transpose(0, 3, 1, 2)
        info = _cmp(got, ref, TOL_BF16)
        info["kernel"] = kern
        info["err_if_bf16_weights"] = float(np.abs(ref_bf16w - ref).max())
        return info
    return run


def maxpool_case(N, H, W, C, k, s, p, dtype="bf16", seed=0):
    def run():
        L = _lib()
        rng = _rng(seed)
        x = rng.standard_normal((N, C, H, W)).astype(np.float32)
        if dtype == "bf16":
            x = bf(x)
        ref = np.stack([O.maxpool2d(x[i], k, s, p) for i in range(N)])
        Ho, Wo = ref.shape[2:]
        xd = dev(x.transpose(0, 2, 3, 1), dtype)
        y = torch.empty((N, Ho, Wo, C), dtype=xd.dtype, device="cuda")
        L.call("mv_maxpool2d_nhwc_fwd", xd.data_ptr(), y.data_ptr(), N, H, W, C, k, k, s, s, p, p, DT[dtype], _stream())
        torch.cuda.synchronize()
        info = _cmp(host(y).transpose(0, 3, 1, 2), ref, 1e-6)
        info["kernel"] = L.last_kernel()
        return info
    return run


def avgpool_case(N, H, W, C, oh, ow, dtype="bf16", seed=0):
    def run():
        L = _lib()
        rng = _rng(seed)
        x = rng.standard_normal((N, C, H, W)).astype(np.float32)
        if dtype == "bf16":
            x = bf(x)
        ref = np.stack([O.adaptive_avgpool2d(x[i], (oh, ow)) for i in range(N)])
        xd = dev(x.transpose(0, 2, 3, 1), dtype)
        y = torch.empty((N, oh, ow, C), dtype=torch.float32, device="cuda")
        L.call("mv_adaptive_avgpool2d_nhwc_fwd", xd.data_ptr(), y.data_ptr(), N, H, W, C, oh, ow, DT[dtype], 0, _stream())
        torch.cuda.synchronize()
        return _cmp(host(y).transpose(0, 3, 1, 2), ref, 1e-5)
    return run


def fc_stream_case(M, K, N, act=0, out="bf16", bias=True, seed=0):
    """mv_fc_stream_fwd (few rows x a big weight matrix: the AlexNet / VGG classifier Linears, alexnet.py:62-70): fragment-ordered
    weights, split-K + fixed-order reduction vs float64; two runs are bit-identical; also against mv_linear_fwd's kernel."""
    def run():
        L = _lib()
        rng = _rng(seed)
        x = bf(rng.standard_normal((M, K)))
        w = bf(rng.standard_normal((N, K)) / np.sqrt(K))
        b = (0.1 * rng.standard_normal(N)).astype(np.float32) if bias else None
        ref = x.astype(np.float64) @ w.astype(np.float64).T + (0 if b is None else b)
        if act == 1:
            ref = O.relu(ref)
        odc = 1 if out == "bf16" else 0
        if not L.load().mv_fc_stream_supported(M, N, K, 1, odc):
            return {"ok": False, "err": "mv_fc_stream_supported says no"}
        NT = (N + 31) // 32
        wp = np.zeros((NT * 32, K), np.float32)
        wp[:N] = w
        wf = np.ascontiguousarray(wp.reshape(NT, 32, K // 16, 2, 8).transpose(0, 2, 3, 1, 4))       # [tile][step][h][n][e]
        xd, wd = dev(x, "bf16"), dev(wf, "bf16")
        bd = None if b is None else dev(b, "fp32")
        nbytes = int(L.load().mv_fc_stream_workspace(M, N, K))
        ws = torch.empty((nbytes // 4,), dtype=torch.float32, device="cuda")
        ys = []
        for _ in range(2):
            y = torch.full((M, N), -7.0, dtype=torch.bfloat16 if out == "bf16" else torch.float32, device="cuda")
            L.call("mv_fc_stream_fwd", xd.data_ptr(), wd.data_ptr(), None if bd is None else bd.data_ptr(), y.data_ptr(), ws.data_ptr(), nbytes,
                   M, N, K, act, 1, odc, _stream())
            ys.append(y)
        kern = L.last_kernel()
        torch.cuda.synchronize()
        info = _cmp(host(ys[0]), ref, TOL_BF16 if out == "bf16" else TOL_F32 * 10)      # bf16 operands: the fp32 result carries their rounding
        info["kernel"] = kern
        info["reproducible"] = bool(torch.equal(ys[0], ys[1]))
        info["ok"] = info["ok"] and info["reproducible"]
        return info
    return run


def stem_pool_case(N, H, W, xdtype="fp32", seed=0, neg_scale=False, alexnet=False):
    """mv_stem_conv_pool_fwd (conv 7x7/2 + BN + ReLU + maxpool 3/2/1, one launch) vs the oracle chain
    conv2d -> scale/shift -> relu -> (bf16 rounding) -> maxpool2d (reference resnet.py:243-254); `alexnet`: the same entry for
    conv 11x11/4 pad 2 + bias + ReLU + maxpool 3/2 (reference alexnet.py:44-46; scale = NULL, shift = the bias)."""
    def run():
        L = _lib()
        rng = _rng(seed)
        C, K = 3, 64
        R, st, pad, pp = (11, 4, 2, 0) if alexnet else (7, 2, 3, 1)
        x = rng.random((N, C, H, W), dtype=np.float32) * 2 - 0.7
        if xdtype == "bf16":
            x = bf(x)
        w = bf(rng.standard_normal((K, C, R, R)) / np.sqrt(C * R * R))
        sc = rng.uniform(0.5, 1.5, K).astype(np.float32)
        if neg_scale:
            sc[::3] *= -1.0            # BN gamma may be negative: scale/shift must be applied BEFORE the max
        if alexnet:
            sc[:] = 1.0
        sf = (0.1 * rng.standard_normal(K)).astype(np.float32)
        if not L.load().mv_stem_conv_pool_supported(C, K, R, R, st, st, pad, pad, 3, 2, pp, 1, DT[xdtype], 1, N * C * H * W):
            return {"ok": False, "err": "mv_stem_conv_pool_supported says no"}
        xr = bf(x)
        conv = np.stack([O.conv2d(xr[i], w, None, st, pad) for i in range(N)]) * sc[None, :, None, None] + sf[None, :, None, None]
        conv = bf(O.relu(conv))
        ref = np.stack([O.maxpool2d(conv[i], 3, 2, pp) for i in range(N)])          # [N, K, Po, Qo]
        Po, Qo = ref.shape[2], ref.shape[3]
        ref = ref.transpose(0, 2, 3, 1)
        xd, wd = dev(x, xdtype), dev(w, "bf16")
        scd, sfd = dev(sc, "fp32"), dev(sf, "fp32")
        y = torch.full((N, Po, Qo, K), -7.0, dtype=torch.bfloat16, device="cuda")
        L.call("mv_stem_conv_pool_fwd", xd.data_ptr(), wd.data_ptr(), None if alexnet else scd.data_ptr(), sfd.data_ptr(), y.data_ptr(),
               N, C, H, W, K, R, R, st, st, pad, pad, 3, 2, pp, 1, DT[xdtype], 1, _stream())
        kern = L.last_kernel()
        torch.cuda.synchronize()
        info = _cmp(host(y), ref, TOL_BF16)
        info["kernel"] = kern
        if alexnet and not kern.startswith("stem_pool11"):
            info["ok"] = False
            info["err_msg"] = f"expected the fused 11x11 entry kernel, ran {kern}"
        return info
    return run


def layernorm_case(M, C, dtype="bf16", generic=False, stride=None, seed=0, out=None):
    def run():
        L = _lib()
        rng = _rng(seed)
        rs = C if stride is None else stride
        xfull = (rng.standard_normal((M, rs)) * 2 + 0.5).astype(np.float32)
        if dtype == "bf16":
            xfull = bf(xfull)
        x = xfull[:, :C]
        g = rng.uniform(0.5, 1.5, C).astype(np.float32)
        b = (0.1 * rng.standard_normal(C)).astype(np.float32)
        ref = O.layernorm_rows(x, g, b, 1e-5)
        xd = dev(xfull, dtype)
        odt = out or dtype
        y = torch.empty((M, C), dtype=torch.bfloat16 if odt == "bf16" else torch.float32, device="cuda")
        gd, bd = dev(g, "fp32"), dev(b, "fp32")
        L.set_flag("force_generic", 1 if generic else 0)
        try:
            L.call("mv_layernorm_fwd", xd.data_ptr(), gd.data_ptr(), bd.data_ptr(), y.data_ptr(), M, C, rs, 1e-5,
                   DT[dtype], DT[odt], _stream())
            kern = L.last_kernel()
        finally:
            L.set_flag("force_generic", 0)
        torch.cuda.synchronize()
        info = _cmp(host(y), ref, TOL_BF16 if "bf16" in (dtype, odt) else 1e-4)
        info["kernel"] = kern
        return info
    return run


def dropout_windows_case(B, Hf, Wf, C, ws, shift, p=0.25, dtype="bf16", seed=0):
    """mv_dropout_windows_fwd vs the reference's order of operations (swin.py:141-160, 233-250): roll, partition into windows,
    `_func_dropout` on (num_windows, n, C), reverse -- bit-exact mask, values x / keep."""
    def run():
        L = _lib()
        rng = _rng(seed)
        x = rng.standard_normal((B, Hf, Wf, C)).astype(np.float32)
        if dtype == "bf16":
            x = bf(x)
        keys = rng.integers(0, 2 ** 32, size=(B, 2), dtype=np.uint64).astype(np.uint32)
        sh = [0 if ws[0] >= Hf else shift[0], 0 if ws[1] >= Wf else shift[1]]
        ref = np.empty_like(x)
        for b in range(B):
            r = np.roll(x[b], (-sh[0], -sh[1]), axis=(0, 1))
            w = r.reshape(Hf // ws[0], ws[0], Wf // ws[1], ws[1], C).transpose(0, 2, 1, 3, 4).reshape(-1, ws[0] * ws[1], C)
            w = O.dropout_fn(w, p, keys[b])
            r = w.reshape(Hf // ws[0], Wf // ws[1], ws[0], ws[1], C).transpose(0, 2, 1, 3, 4).reshape(Hf, Wf, C)
            ref[b] = np.roll(r, (sh[0], sh[1]), axis=(0, 1))
        xd = dev(x, dtype)
        kd = torch.from_numpy(keys.view(np.int32)).cuda()
        y = torch.empty_like(xd)
        L.call("mv_dropout_windows_fwd", xd.data_ptr(), kd.data_ptr(), y.data_ptr(), B, Hf, Wf, C, ws[0], ws[1], shift[0], shift[1],
               float(np.float32(1.0 - p)), DT[dtype], _stream())
        torch.cuda.synchronize()
        got = host(y)
        info = _cmp(got, ref, TOL_BF16 if dtype == "bf16" else 1e-6)
        info["mask_mismatches"] = int(((got != 0) != (ref != 0)).sum())
        info["kernel"] = L.last_kernel()
        info["ok"] = info["ok"] and info["mask_mismatches"] == 0
        return info
    return run


def prng_split_case(R, num, child_major, seed=0):
    """mv_prng_split vs the oracle's jax.random.split of every key: bit-exact."""
    def run():
        L = _lib()
        rng = _rng(seed)
        keys = rng.integers(0, 2 ** 32, size=(R, 2), dtype=np.uint64).astype(np.uint32)
        ref = np.stack([O.jax_split(keys[r], num) for r in range(R)])              # [R, num, 2]
        if child_major:
            ref = ref.transpose(1, 0, 2)
        kd = torch.from_numpy(keys.view(np.int32)).cuda()
        out = torch.zeros(ref.shape, dtype=torch.int32, device="cuda")
        L.call("mv_prng_split", kd.data_ptr(), out.data_ptr(), R, num, child_major, _stream())
        torch.cuda.synchronize()
        got = out.cpu().numpy().view(np.uint32)
        bad = int((got != ref).sum())
        return {"ok": bad == 0, "mismatches": bad, "kernel": L.last_kernel()}
    return run


def mha_dropout_case(B, N, H, dh, p=0.2, head_major=False, probs=True, seed=0):
    """mv_mha_dropout_fwd: the attention core with vit.py:71's live dropout.  The returned (dropped) probabilities must be ZERO
    exactly where jax.random.bernoulli(key_b, 1 - p, (1, H, N, N)) is False (oracle bit stream) and the kernel's own un-dropped
    probabilities / (1 - p) elsewhere; the output must be that matrix times V."""
    def run():
        L = _lib()
        rng = _rng(seed)
        D = H * dh
        keep = np.float32(1.0 - p)
        qkv = bf(rng.standard_normal((B, N, 3 * D)).astype(np.float32))
        keys = rng.integers(0, 2 ** 32, size=(B, 2), dtype=np.uint64).astype(np.uint32)
        scale = dh ** -0.5
        t = qkv.reshape(B, N, 3, H, dh).transpose(2, 0, 3, 1, 4).astype(np.float64)       # [3, B, H, N, dh]
        q, k, v = t[0], t[1], t[2]
        a = O.softmax((q @ k.transpose(0, 1, 3, 2)) * scale, -1).astype(np.float64)
        mask = np.stack([O.jax_bernoulli(keys[b], keep, (1, H, N, N))[0] for b in range(B)])
        ad = np.where(mask, a / np.float64(keep), 0.0)
        ref = (ad @ v).transpose(0, 2, 1, 3).reshape(B, N, D)
        src = np.ascontiguousarray(qkv.reshape(B, N, 3 * H, dh).transpose(0, 2, 1, 3)) if head_major else qkv
        qd = dev(src, "bf16")
        kd = torch.from_numpy(keys.view(np.int32)).cuda()
        y = torch.empty((B, N, D), dtype=qd.dtype, device="cuda")
        pr = torch.zeros((B, H, N, N), dtype=torch.float32, device="cuda") if probs else None
        p0 = torch.empty((B, H, N, N), dtype=torch.float32, device="cuda")
        y0 = torch.empty_like(y)
        L.call("mv_mha_heads_fwd" if head_major else "mv_mha_fwd", qd.data_ptr(), y0.data_ptr(), p0.data_ptr(), B, N, H, dh,
               float(scale), DT["bf16"], _stream())
        L.call("mv_mha_dropout_fwd", qd.data_ptr(), 1 if head_major else 0, y.data_ptr(), None if pr is None else pr.data_ptr(),
               kd.data_ptr(), float(keep), B, N, H, dh, float(scale), DT["bf16"], _stream())
        kern = L.last_kernel()
        torch.cuda.synchronize()
        info = _cmp(host(y), ref, TOL_BF16)
        info["kernel"] = kern
        if probs:
            got, base = host(pr), host(p0)
            info["mask_mismatches"] = int(((got != 0) != (mask & (base != 0))).sum())
            pi = _cmp(got, np.where(mask, base / keep, np.float32(0)), 1e-5)
            info["probs_err"] = pi.get("err")
            info["kept_fraction"] = float(mask.mean())
            slack = max(0.02, 5.0 * float(np.sqrt(keep * (1 - keep) / mask.size)))      # a sanity check of the rate; the masks are compared bit for bit
            info["ok"] = info["ok"] and pi["ok"] and info["mask_mismatches"] == 0 and abs(info["kept_fraction"] - float(keep)) < slack
        return info
    return run


def mha_case(B, N, H, dh, dtype="bf16", probs=True, generic=False, seed=0, spike=False):
    def run():
        L = _lib()
        rng = _rng(seed)
        D = H * dh
        qkv = rng.standard_normal((B, N, 3 * D)).astype(np.float32)
        if spike:   # one huge logit per row: exercises the max-subtraction
            qkv[:, :, :D] *= 6.0
        if dtype == "bf16":
            qkv = bf(qkv)
        scale = dh ** -0.5
        t = qkv.reshape(B, N, 3, H, dh).transpose(2, 0, 3, 1, 4).astype(np.float64)
        q, k, v = t[0], t[1], t[2]
        a = O.softmax((q @ k.transpose(0, 1, 3, 2)) * scale, -1).astype(np.float64)
        ref = (a @ v).transpose(0, 2, 1, 3).reshape(B, N, D)
        qd = dev(qkv, dtype)
        y = torch.empty((B, N, D), dtype=qd.dtype, device="cuda")
        pr = torch.empty((B, H, N, N), dtype=torch.float32, device="cuda") if probs else None
        L.set_flag("force_generic", 1 if generic else 0)
        try:
            L.call("mv_mha_fwd", qd.data_ptr(), y.data_ptr(), None if pr is None else pr.data_ptr(), B, N, H, dh,
                   float(scale), DT[dtype], _stream())
            kern = L.last_kernel()
        finally:
            L.set_flag("force_generic", 0)
        torch.cuda.synchronize()
        info = _cmp(host(y), ref, TOL_BF16 if dtype == "bf16" else TOL_F32)
        info["kernel"] = kern
        if probs:
            pi = _cmp(host(pr), a, 2e-3)
            info["probs_err"] = pi.get("err")
            info["ok"] = info["ok"] and pi["ok"]
        return info
    return run


def qkv_heads_case(B, N, H, dh, seed=0, probs=False, flags=()):
    """mv_linear_heads_fwd (head-major qkv projection) + mv_mha_heads_fwd vs the oracle's
    Linear -> reshape/transpose -> attention (reference vit.py:64-73)."""
    def run():
        L = _lib()
        rng = _rng(seed)
        D = H * dh
        x = bf(rng.standard_normal((B, N, D)).astype(np.float32))
        w = bf((rng.standard_normal((3 * D, D)) / np.sqrt(D)).astype(np.float32))
        b = (0.1 * rng.standard_normal(3 * D)).astype(np.float32)
        if not L.load().mv_linear_heads_supported(B * N, 3 * D, D, N, dh, DT["bf16"]):
            return {"ok": False, "err": "mv_linear_heads_supported says no"}
        qkv_ref = bf(x.reshape(B * N, D).astype(np.float64) @ w.T.astype(np.float64) + b)          # [M, 3D] bf16-rounded
        t = qkv_ref.reshape(B, N, 3, H, dh).transpose(2, 0, 3, 1, 4).astype(np.float64)      # [3, B, H, N, dh]
        hm_ref = t.transpose(1, 0, 2, 3, 4).reshape(B, 3 * H, N, dh)
        scale = dh ** -0.5
        a = O.softmax((t[0] @ t[1].transpose(0, 1, 3, 2)) * scale, -1).astype(np.float64)
        ref = (a @ t[2]).transpose(0, 2, 1, 3).reshape(B, N, D)
        xd, wd, bd = dev(x, "bf16"), dev(w, "bf16"), dev(b, "fp32")
        qkv = torch.empty((B, 3 * H, N, dh), dtype=torch.bfloat16, device="cuda")
        y = torch.empty((B, N, D), dtype=torch.bfloat16, device="cuda")
        pr = torch.empty((B, H, N, N), dtype=torch.float32, device="cuda") if probs else None
        for f in flags:
            L.set_flag(f.split("=")[0], int(f.split("=")[1]) if "=" in f else 1)
        try:
            L.call("mv_linear_heads_fwd", xd.data_ptr(), wd.data_ptr(), None, bd.data_ptr(), qkv.data_ptr(), B * N, 3 * D, D,
                   N, dh, DT["bf16"], _stream())
            k1 = L.last_kernel()
        finally:
            for f in flags:
                L.set_flag(f.split("=")[0], 0)
        L.call("mv_mha_heads_fwd", qkv.data_ptr(), y.data_ptr(), None if pr is None else pr.data_ptr(), B, N, H, dh,
               float(scale), DT["bf16"], _stream())
        k2 = L.last_kernel()
        torch.cuda.synchronize()
        i1 = _cmp(host(qkv), hm_ref, TOL_BF16)
        info = _cmp(host(y), ref, 2 * TOL_BF16)      # attention on a bf16-rounded qkv that may differ by one ulp
        info["qkv_err"] = i1.get("err")
        info["ok"] = info["ok"] and i1["ok"]
        info["kernel"] = k1 + "+" + k2
        if probs:
            pi = _cmp(host(pr), a, 4e-3)
            info["probs_err"] = pi.get("err")
            info["ok"] = info["ok"] and pi["ok"]
        return info
    return run


def swin_attn_case(B, Hf, C, heads, ws, shift, dtype="bf16", seed=0, generic=False):
    def run():
        L = _lib()
        rng = _rng(seed)
        n = ws * ws
        qkv = rng.standard_normal((B, 3 * C, Hf, Hf)).astype(np.float32)
        if dtype == "bf16":
            qkv = bf(qkv)
        bias = (0.5 * rng.standard_normal((heads, n, n))).astype(np.float32)
        # oracle: shifted_window_attention with identity qkv / proj so that it consumes qkv directly
        eye3 = np.eye(3 * C, dtype=np.float32)
        refs = []
        for i in range(B):
            # feed x = qkv (3C channels) through a fake "qkv_weight" = identity on a 3C-dim input is not
            # shape-compatible with the C-dim API, so restate the core here from the oracle's pieces:
            refs.append(_swin_core_ref(qkv[i], bias, ws, heads, shift, C))
        ref = np.stack(refs)
        qd = dev(qkv.transpose(0, 2, 3, 1), dtype)
        bd = dev(bias, "fp32")
        y = torch.empty((B, Hf, Hf, C), dtype=qd.dtype, device="cuda")
        L.set_flag("force_generic", 1 if generic else 0)
        try:
            L.call("mv_swin_window_attn_fwd", qd.data_ptr(), bd.data_ptr(), y.data_ptr(), B, Hf, Hf, C, heads, ws, ws,
                   shift, shift, DT[dtype], _stream())
        finally:
            L.set_flag("force_generic", 0)
        torch.cuda.synchronize()
        info = _cmp(host(y).transpose(0, 3, 1, 2), ref, TOL_BF16 if dtype == "bf16" else TOL_F32)
        info["kernel"] = L.last_kernel()
        return info
    return run


def patch4_ln_case(B, H, K=96, split=True, seed=0):
    """mv_patch4_ln_fwd (Swin patch embedding conv 4x4 / 4 + LayerNorm2d in one launch, swin.py:705-711) vs the numpy restatement;
    fp32 output, split-precision weights (hi + lo bf16 terms): the input is the only bf16 rounding, so the fp32 tolerance applies
    to the image-rounded reference."""
    def run():
        L = _lib()
        rng = _rng(seed)
        x = rng.random((B, 3, H, H), dtype=np.float32)
        w = (rng.standard_normal((K, 3, 4, 4)) / np.sqrt(48)).astype(np.float32)
        b = (0.1 * rng.standard_normal(K)).astype(np.float32)
        g = rng.uniform(0.5, 1.5, K).astype(np.float32)
        be = (0.1 * rng.standard_normal(K)).astype(np.float32)
        if not L.load().mv_patch4_ln_supported(3, H, H, K, 0):
            return {"ok": False, "err": "mv_patch4_ln_supported says no"}
        hi = bf(w)
        lo = bf(w - hi)
        weff = (hi + lo) if split else hi
        conv = np.stack([O.conv2d(bf(x[i]), weff, b.reshape(-1, 1, 1), 4, 0) for i in range(B)])          # [B][K][H/4][W/4]
        rows = conv.transpose(0, 2, 3, 1).reshape(-1, K)
        ref = O.layernorm_rows(rows, g, be, 1e-5).reshape(B, H // 4, H // 4, K)
        xd = dev(x, "fp32")
        hid, lod = dev(hi.reshape(K, 48), "bf16"), dev(lo.reshape(K, 48), "bf16")
        bd, gd, bed = dev(b, "fp32"), dev(g, "fp32"), dev(be, "fp32")
        y = torch.full((B, H // 4, H // 4, K), -7.0, dtype=torch.float32, device="cuda")
        L.call("mv_patch4_ln_fwd", xd.data_ptr(), hid.data_ptr(), lod.data_ptr() if split else None, bd.data_ptr(), gd.data_ptr(),
               bed.data_ptr(), y.data_ptr(), B, 3, H, H, K, 1e-5, 0, _stream())
        kern = L.last_kernel()
        torch.cuda.synchronize()
        info = _cmp(host(y), ref, TOL_F32)
        info["kernel"] = kern
        return info
    return run


def patch_merge_ln_case(B, H, C, out="bf16", seed=0):
    """mv_patch_merge_ln_fwd (Swin patch merging's 2 x 2 gather + LayerNorm over the 4 C channels in one pass, swin.py:23-31, 61-65)
    vs the numpy restatement: concatenate [x[0::2,0::2], x[1::2,0::2], x[0::2,1::2], x[1::2,1::2]] along channels, then LayerNorm."""
    def run():
        L = _lib()
        rng = _rng(seed)
        x = (rng.standard_normal((B, H, H, C)) * rng.uniform(0.5, 2.0, (B, H, H, 1)) + rng.uniform(-1, 1, (B, H, H, 1))).astype(np.float32)
        g = rng.uniform(0.5, 1.5, 4 * C).astype(np.float32)
        be = (0.1 * rng.standard_normal(4 * C)).astype(np.float32)
        if not L.load().mv_patch_merge_ln_supported(H, H, C, 0):
            return {"ok": False, "err": "mv_patch_merge_ln_supported says no"}
        cat = np.concatenate([x[:, 0::2, 0::2], x[:, 1::2, 0::2], x[:, 0::2, 1::2], x[:, 1::2, 1::2]], axis=-1)
        ref = O.layernorm_rows(cat.reshape(-1, 4 * C), g, be, 1e-5).reshape(B, H // 2, H // 2, 4 * C)
        xd, gd, bd = dev(x, "fp32"), dev(g, "fp32"), dev(be, "fp32")
        y = torch.full((B, H // 2, H // 2, 4 * C), -7.0, dtype=torch.bfloat16 if out == "bf16" else torch.float32, device="cuda")
        L.call("mv_patch_merge_ln_fwd", xd.data_ptr(), gd.data_ptr(), bd.data_ptr(), y.data_ptr(), B, H, H, C, 1e-5, 0, DT[out], _stream())
        kern = L.last_kernel()
        torch.cuda.synchronize()
        info = _cmp(host(y), ref, TOL_BF16 if out == "bf16" else TOL_F32)
        info["kernel"] = kern
        return info
    return run


def swin_block_attn_case(B, Hf, shift, seed=0, C=384, heads=12, ws=7, flags=()):
    """mv_swin_block_attn_fwd (LayerNorm -> qkv -> shifted-window attention -> proj -> + x, one launch, one workgroup per window;
    swin.py:572-578 first line) vs the float64 restatement (LayerNorm, Linear, `_swin_core_ref`, Linear), and vs the library's own
    four-launch sequence (which must not be closer to the reference by more than rounding noise)."""
    def run():
        from eqxvision_amd.ops import swin_block_attn_fragments
        L = _lib()
        rng = _rng(seed)
        n = ws * ws
        x = (rng.standard_normal((B, Hf, Hf, C)) * rng.uniform(0.5, 2.0, (B, Hf, Hf, 1)) + rng.uniform(-1, 1, (B, Hf, Hf, 1))).astype(np.float32)
        g = rng.uniform(0.5, 1.5, C).astype(np.float32)
        be = (0.1 * rng.standard_normal(C)).astype(np.float32)
        wq = (rng.standard_normal((3 * C, C)) / np.sqrt(C)).astype(np.float32)
        bq = (0.1 * rng.standard_normal(3 * C)).astype(np.float32)
        wp = (rng.standard_normal((C, C)) / np.sqrt(C)).astype(np.float32)
        bp = (0.1 * rng.standard_normal(C)).astype(np.float32)
        bias = (0.5 * rng.standard_normal((heads, n, n))).astype(np.float32)
        if not L.load().mv_swin_block_attn_supported(Hf, Hf, C, heads, ws, ws, 0):
            return {"ok": False, "err": "mv_swin_block_attn_supported says no"}
        nx = O.layernorm_rows(x.reshape(-1, C), g, be, 1e-5).astype(np.float64)
        qkv = (nx @ wq.astype(np.float64).T + bq).reshape(B, Hf, Hf, 3 * C)
        ref = np.empty((B, Hf, Hf, C))
        for i in range(B):
            a = _swin_core_ref(qkv[i].transpose(2, 0, 1).astype(np.float32), bias, ws, heads, shift, C).transpose(1, 2, 0)
            ref[i] = x[i] + a.reshape(-1, C).astype(np.float64).dot(wp.astype(np.float64).T).reshape(Hf, Hf, C) + bp
        wf, bqf, wpf, b64 = swin_block_attn_fragments(wq * g[None, :], bq + wq @ be, wp, bias)
        xd = dev(x, "fp32")
        d = [dev(bf(wf), "bf16"), dev(bqf, "fp32"), dev(bf(wpf), "bf16"), dev(bp, "fp32"), dev(b64, "fp32")]
        y = torch.full_like(xd, -7.0)
        for f in flags:
            L.set_flag(f, 1)
        try:
            L.call("mv_swin_block_attn_fwd", xd.data_ptr(), d[0].data_ptr(), d[1].data_ptr(), d[2].data_ptr(), d[3].data_ptr(), d[4].data_ptr(),
                   y.data_ptr(), B, Hf, Hf, C, heads, ws, ws, shift, shift, 1e-5, 0, _stream())
            kern = L.last_kernel()
        finally:
            for f in flags:
                L.set_flag(f, 0)
        torch.cuda.synchronize()
        info = _cmp(host(y), ref, TOL_BF16)
        info["kernel"] = kern
        # the four launches it replaces
        M = B * Hf * Hf
        gd, bed = dev(g, "fp32"), dev(be, "fp32")
        nb = torch.empty((M, C), dtype=torch.bfloat16, device="cuda")
        L.call("mv_layernorm_fwd", xd.data_ptr(), gd.data_ptr(), bed.data_ptr(), nb.data_ptr(), M, C, 0, 1e-5, 0, 1, _stream())
        qd = torch.empty((M, 3 * C), dtype=torch.bfloat16, device="cuda")
        wqd, bqd = dev(bf(wq), "bf16"), dev(bq, "fp32")
        L.call("mv_linear_fwd", nb.data_ptr(), wqd.data_ptr(), None, bqd.data_ptr(), None, qd.data_ptr(), M, 3 * C, C, 0, 1, 1, _stream())
        ad = torch.empty((M, C), dtype=torch.bfloat16, device="cuda")
        biasd = dev(bias, "fp32")
        L.call("mv_swin_window_attn_fwd", qd.data_ptr(), biasd.data_ptr(), ad.data_ptr(), B, Hf, Hf, C, heads, ws, ws, shift, shift, 1,
               _stream())
        y4 = torch.empty_like(xd)
        wpd, bpd = dev(bf(wp), "bf16"), dev(bp, "fp32")
        L.call("mv_linear_fwd", ad.data_ptr(), wpd.data_ptr(), None, bpd.data_ptr(), xd.data_ptr(), y4.data_ptr(), M, C, C, 0, 1, 0, _stream())
        torch.cuda.synchronize()
        e4 = float(np.abs(host(y4).astype(np.float64) - ref).max())
        info["err_four_launches"] = e4
        info["ok"] = bool(info["ok"] and info["err"] <= 1.5 * e4 + 1e-3)
        return info
    return run


def _swin_core_ref(qkv_chw, bias, ws, heads, shift, C):
    """oracle.shifted_window_attention with the qkv projection already applied: call it with
    x = qkv viewed as a 3C-channel map and identity weights of matching size, then undo proj."""
    # Use the oracle function on a widened problem: x has 3C channels, qkv_weight maps 3C -> 3*(3C)?  Too
    # wasteful; instead call the oracle with C' = C by passing weights that SELECT q,k,v from a
    # concatenated input is impossible (input must be C-dim).  So: run the oracle three-in-one by
    # temporarily treating the projection as identity via linear algebra: x' = q part, and patch k,v
    # through the bias-free path below (same code as oracle lines, restated for pre-projected qkv).
    H = qkv_chw.shape[1]
    x = np.transpose(qkv_chw, (1, 2, 0))
    sh = [shift, shift]
    if ws >= H:
        sh = [0, 0]
    if sum(sh) > 0:
        x = np.roll(x, (-sh[0], -sh[1]), axis=(0, 1))
    nW = (H // ws) ** 2
    n = ws * ws
    x = x.reshape(H // ws, ws, H // ws, ws, 3 * C).transpose(0, 2, 1, 3, 4).reshape(nW, n, 3 * C)
    dh = C // heads
    t = x.reshape(nW, n, 3, heads, dh).transpose(2, 0, 3, 1, 4).astype(np.float64)
    q, k, v = t[0] * dh ** -0.5, t[1], t[2]
    attn = q @ k.transpose(0, 1, 3, 2) + bias[None]
    if sum(sh) > 0:
        m = np.zeros((H, H))
        sl = ((0, H - ws), (H - ws, H - sh[0]), (H - sh[0], H))
        c = 0
        for a in sl:
            for b in sl:
                m[a[0]:a[1], b[0]:b[1]] = c
                c += 1
        m = m.reshape(H // ws, ws, H // ws, ws).transpose(0, 2, 1, 3).reshape(nW, n)
        am = np.where((m[:, None, :] - m[:, :, None]) != 0, -100.0, 0.0)
        attn = attn + am[:, None]
    attn = O.softmax(attn, -1).astype(np.float64)
    y = (attn @ v).transpose(0, 2, 1, 3).reshape(nW, n, C)
    y = y.reshape(H // ws, H // ws, ws, ws, C).transpose(0, 2, 1, 3, 4).reshape(H, H, C)
    if sum(sh) > 0:
        y = np.roll(y, (sh[0], sh[1]), axis=(0, 1))
    return np.transpose(y, (2, 0, 1)).astype(np.float32)


def misc_case(kind, dtype="bf16", seed=0):
    def run():
        L = _lib()
        rng = _rng(seed)
        q = bf if dtype == "bf16" else (lambda a: np.asarray(a, np.float32))
        tdt = torch.bfloat16 if dtype == "bf16" else torch.float32
        if kind in ("patch_merge", "patch_merge_odd_c"):
            B, H, W, C = (2, 7, 6, 16) if kind == "patch_merge" else (3, 9, 11, 6)     # vector / scalar kernel
            x = q(rng.standard_normal((B, C, H, W)))
            xp = np.pad(x, ((0, 0), (0, 0), (0, H % 2), (0, W % 2)))
            ref = np.concatenate([xp[:, :, 0::2, 0::2], xp[:, :, 1::2, 0::2], xp[:, :, 0::2, 1::2], xp[:, :, 1::2, 1::2]], 1)
            xd = dev(x.transpose(0, 2, 3, 1), dtype)
            y = torch.empty((B, (H + 1) // 2, (W + 1) // 2, 4 * C), dtype=tdt, device="cuda")
            L.call("mv_patch_merge_gather_nhwc", xd.data_ptr(), y.data_ptr(), B, H, W, C, DT[dtype], _stream())
            torch.cuda.synchronize()
            return _cmp(host(y).transpose(0, 3, 1, 2), ref, 1e-6)
        if kind == "layout":
            N, C, H, W = 3, 37, 9, 11
            x = rng.standard_normal((N, C, H, W)).astype(np.float32)
            xd = dev(x, "fp32")
            y = torch.empty((N, H, W, C), dtype=tdt, device="cuda")
            L.call("mv_nchw_to_nhwc", xd.data_ptr(), y.data_ptr(), N, C, H, W, 0, DT[dtype], _stream())
            z = torch.empty((N, C, H, W), dtype=torch.float32, device="cuda")
            L.call("mv_nhwc_to_nchw", y.data_ptr(), z.data_ptr(), N, C, H, W, DT[dtype], 0, _stream())
            torch.cuda.synchronize()
            a = _cmp(host(y), q(x).transpose(0, 2, 3, 1), 1e-6)
            b = _cmp(host(z), q(x), 1e-6)
            a["ok"] = a["ok"] and b["ok"]
            return a
        if kind == "eltwise":
            n = 100003
            x = q(rng.standard_normal(n) * 3)
            y2 = q(rng.standard_normal(n))
            xd, yd = dev(x, dtype), dev(y2, dtype)
            o1 = torch.empty(n, dtype=tdt, device="cuda")
            o2 = torch.empty(n, dtype=tdt, device="cuda")
            o3 = torch.empty(n, dtype=tdt, device="cuda")
            L.call("mv_eltwise_fwd", xd.data_ptr(), o1.data_ptr(), n, 2, DT[dtype], _stream())
            L.call("mv_eltwise_fwd", xd.data_ptr(), o2.data_ptr(), n, 1, DT[dtype], _stream())
            L.call("mv_add_fwd", xd.data_ptr(), yd.data_ptr(), o3.data_ptr(), n, 1, DT[dtype], _stream())
            torch.cuda.synchronize()
            tol = TOL_BF16 if dtype == "bf16" else 1e-5
            a = _cmp(host(o1), O.gelu_tanh(x), tol)
            b = _cmp(host(o2), O.relu(x), tol)
            c = _cmp(host(o3), O.relu(x + y2), tol)
            a["ok"] = a["ok"] and b["ok"] and c["ok"]
            a["err"] = max(a.get("err", 9), b.get("err", 9), c.get("err", 9))
            return a
        if kind == "affine_cls":
            rows, C = 50, 24
            x = q(rng.standard_normal((rows, C)))
            sc = rng.uniform(0.5, 1.5, C).astype(np.float32)
            sf = rng.standard_normal(C).astype(np.float32)
            xd = dev(x, dtype)
            o = torch.empty((rows, C), dtype=tdt, device="cuda")
            scd, sfd = dev(sc, "fp32"), dev(sf, "fp32")     # keep the device tensors alive across the launch
            L.call("mv_channel_affine_fwd", xd.data_ptr(), scd.data_ptr(), sfd.data_ptr(),
                   o.data_ptr(), rows, C, 1, DT[dtype], _stream())
            B, T, D = 3, 5, 16
            cls = rng.standard_normal(D).astype(np.float32)
            pos = rng.standard_normal((T, D)).astype(np.float32)
            tok = torch.zeros((B, T, D), dtype=tdt, device="cuda")
            clsd, posd = dev(cls, "fp32"), dev(pos, "fp32")
            L.call("mv_vit_cls_pos_fwd", clsd.data_ptr(), posd.data_ptr(), tok.data_ptr(), B, T, D,
                   DT[dtype], _stream())
            torch.cuda.synchronize()
            a = _cmp(host(o), O.relu(x * sc + sf), TOL_BF16 if dtype == "bf16" else 1e-5)
            ref = np.zeros((B, T, D), np.float32)
            ref[:, 0] = cls + pos[0]
            b = _cmp(host(tok), ref, TOL_BF16 if dtype == "bf16" else 1e-6)
            a["ok"] = a["ok"] and b["ok"]
            return a
        raise ValueError(kind)
    return run


def fuzz_cases(n, seed=0, mfma_only=False, flags=()):
    """`n` random convolution / linear shapes for the DEFAULT dispatch (tools/fuzz_ops.py and the `fuzz/` pytest cases):
    any tap count / stride / padding / dilation / residual / activation / output dtype, sized so the oracle stays fast."""
    rng = np.random.default_rng(seed)
    ch = [64, 64, 128, 128, 192, 256, 256, 320, 384, 512, 768] if mfma_only else \
        [8, 16, 24, 32, 48, 64, 96, 128, 160, 192, 256, 320, 384, 512, 768]
    out = []
    i = 0
    while len(out) < n:
        i += 1
        if rng.random() < 0.66:
            R = int(rng.choice([1, 1, 3, 3, 3, 5, 7]))
            C, K = int(rng.choice(ch)), int(rng.choice(ch))
            H, W = int(rng.integers(5, 60)), int(rng.integers(5, 60))
            N = int(max(1, min(rng.integers(1, 300), 6e9 // (H * W * C * K * R * R))))
            stride = int(rng.choice([1, 1, 2]))
            dil = int(rng.choice([1, 1, 2])) if R > 1 else 1
            pad = int(rng.choice([0, (R - 1) // 2 * dil, 1]))
            if H + 2 * pad < dil * (R - 1) + 1 or W + 2 * pad < dil * (R - 1) + 1:
                continue
            act = int(rng.choice([0, 1, 2]))
            res = bool(rng.random() < 0.4)
            o = "fp32" if rng.random() < 0.25 else "same"
            name = f"fuzz/s{seed}_conv_N{N}_{H}x{W}_C{C}_K{K}_{R}x{R}_s{stride}_p{pad}_d{dil}_a{act}_r{int(res)}_{o}"
            out.append((name, conv_nhwc_case(N, H, W, C, K, R, R, stride=stride, pad=pad, dil=dil, act=act, res=res, out=o,
                                             seed=1000 + i, flags=flags)))
        else:
            K = int(rng.choice(ch + [1024, 2048, 3072]))
            Nn = int(rng.choice(ch + [1000, 1024, 2304]))
            M = int(rng.choice([1, 7, 33, 128, 200, 257, 1000, 3136, 6272, 12544, 25000]))
            M = int(max(1, min(M, 4e9 // (K * Nn))))
            act = int(rng.choice([0, 1, 2]))
            res = bool(rng.random() < 0.4)
            o = "fp32" if rng.random() < 0.3 else "same"
            name = f"fuzz/s{seed}_linear_M{M}_K{K}_N{Nn}_a{act}_r{int(res)}_{o}"
            out.append((name, linear_case(M, K, Nn, act=act, res=res, out=o, seed=2000 + i, flags=flags)))
    return out


def fuzz_misc_cases(n, seed=0):
    """Random shapes for the non-GEMM kernels: LayerNorm (row length / dtype / out dtype), attention (tokens, heads),
    Swin window attention (map size, shift), pools, the fused ResNet entry and the chained 1x1 pair."""
    rng = np.random.default_rng(seed)
    out = []
    i = 0
    while len(out) < n:
        i += 1
        k = int(rng.integers(0, 9))
        sd = 3000 + i
        if k == 0:
            C = int(rng.choice([32, 64, 96, 128, 192, 256, 384, 512, 768, 1024, 1536, 40, 100]))
            M = int(rng.choice([1, 5, 63, 197, 1000, 3136, 12544, 50000]))
            dt = str(rng.choice(["bf16", "fp32"]))
            o = str(rng.choice(["bf16", "fp32"])) if dt == "fp32" or rng.random() < 0.5 else None
            out.append((f"fuzzm/s{seed}_ln_M{M}_C{C}_{dt}_{o}", layernorm_case(M, C, dtype=dt, out=o, seed=sd)))
        elif k == 1:
            B, N = int(rng.integers(1, 40)), int(rng.choice([5, 17, 50, 64, 65, 128, 197, 200, 256]))
            H, dh = int(rng.choice([1, 2, 3, 4, 6, 12])), int(rng.choice([32, 64]))
            out.append((f"fuzzm/s{seed}_mha_B{B}_N{N}_H{H}_dh{dh}", mha_case(B, N, H, dh, probs=bool(rng.random() < 0.5), seed=sd)))
        elif k == 2:
            ws = 7
            Hf = ws * int(rng.choice([1, 2, 3, 4, 8]))
            heads = int(rng.choice([1, 2, 3, 6]))
            shift = 0 if Hf == ws or rng.random() < 0.4 else 3
            out.append((f"fuzzm/s{seed}_swin_B{3}_H{Hf}_h{heads}_sh{shift}",
                        swin_attn_case(int(rng.integers(1, 5)), Hf, 32 * heads, heads, ws, shift, seed=sd)))
        elif k == 3:
            kk, st = (3, 2) if rng.random() < 0.6 else (2, 2)
            pd = int(rng.choice([0, 1])) if kk == 3 else 0
            out.append((f"fuzzm/s{seed}_maxpool_{i}", maxpool_case(int(rng.integers(1, 6)), int(rng.integers(7, 70)),
                                                                  int(rng.integers(7, 70)), int(rng.choice([8, 64, 96, 192, 256])),
                                                                  kk, st, pd, seed=sd)))
        elif k == 4:
            hw = int(rng.choice([6, 7, 12, 14]))
            oh = int(rng.choice([1, hw]))
            out.append((f"fuzzm/s{seed}_avgpool_{i}", avgpool_case(int(rng.integers(1, 9)), hw, hw, int(rng.choice([64, 256, 768, 2048])),
                                                                  oh, oh, seed=sd)))
        elif k == 5:
            out.append((f"fuzzm/s{seed}_stem_pool_{i}", stem_pool_case(int(rng.integers(1, 5)), int(rng.integers(20, 130)),
                                                                      int(rng.integers(20, 130)),
                                                                      xdtype=str(rng.choice(["fp32", "bf16"])), seed=sd)))
        elif k == 6:
            out.append((f"fuzzm/s{seed}_chain_{i}", chain_case(8192 + int(rng.integers(0, 30000)), seed=sd,
                                                              N2=int(rng.choice([64, 128])))))
        elif k == 7:
            out.append((f"fuzzm/s{seed}_dual_chain_{i}", dual_chain_case(8192 + int(rng.integers(0, 30000)), seed=sd)))
        else:
            st = int(rng.choice([1, 2]))
            Ho, Wo = int(rng.integers(7, 40)), int(rng.integers(7, 40))
            N = int(max(1, -(-4096 // (Ho * Wo)) + rng.integers(0, 6)))
            out.append((f"fuzzm/s{seed}_dual_{i}", dual_case(N, Ho, Wo, int(rng.choice([64, 128, 192, 256])),
                                                            int(rng.choice([64, 128, 256, 512])),
                                                            int(rng.choice([64, 200, 256, 512, 1024])), st,
                                                            act=int(rng.choice([0, 1])), seed=sd)))
    return out


def fuzz_family_cases(n, seed=0):
    """Random shapes for the kernels added for the section-8 "next" rows: depthwise (all four fast paths + the general kernel),
    grouped convolutions through input windows (group widths that divide 64 and multiples of 8 that do not), bilinear
    up-sampling and the wide global average pool."""
    rng = np.random.default_rng(seed)
    out = []
    i = 0
    while len(out) < n:
        i += 1
        k = int(rng.integers(0, 6))
        sd = 4000 + i
        if k == 4:            # training-mode BatchNorm moments: widths of every family, few and many rows
            C = int(rng.choice([8, 16, 24, 40, 64, 96, 120, 256, 672, 1280, 2048]))
            out.append((f"fuzzf/s{seed}_moments_C{C}_{i}", moments_case(int(rng.integers(1, 40000)), C, str(rng.choice(["bf16", "fp32"])), seed=sd)))
            continue
        if k == 5:            # training-mode Dropout: even / odd element counts, logical (C,H,W) or physical order
            C = int(rng.choice([1, 3, 8, 21, 64, 100]))
            hw = int(rng.integers(1, 300))
            out.append((f"fuzzf/s{seed}_dropout_C{C}_hw{hw}_{i}", dropout_case(int(rng.integers(1, 5)), hw * C, C, bool(rng.random() < 0.5),
                                                                            float(rng.choice([0.1, 0.5, 0.9])),
                                                                            str(rng.choice(["bf16", "fp32"])), seed=sd)))
            continue
        if k == 0:
            C = int(rng.choice([8, 16, 24, 40, 72, 96, 120, 184, 240, 480, 672, 960]))
            R = int(rng.choice([3, 3, 5, 7]))
            dil = int(rng.choice([1, 1, 1, 2]))
            stride = int(rng.choice([1, 2]))
            H, W = int(rng.integers(5, 40)), int(rng.integers(5, 40))
            pad = (R - 1) // 2 * dil
            out.append((f"fuzzf/s{seed}_dw_C{C}_{H}x{W}_k{R}s{stride}d{dil}", dwconv_case(int(rng.integers(1, 4)), H, W, C, R=R, stride=stride,
                                                                                        pad=pad, dil=dil, act=int(rng.choice([0, 1, 3])),
                                                                                        scale=bool(rng.random() < 0.7), seed=sd)))
        elif k == 1:
            cg = int(rng.choice([1, 2, 4, 8, 16, 24, 32, 40, 48, 56, 64, 104, 120, 168]))
            groups = int(rng.integers(2, 9)) if cg >= 24 else int(rng.choice([16, 32, 64])) if cg <= 4 else int(rng.integers(2, 17))
            C = cg * groups
            if C % 8 or C > 1400:
                continue
            H = int(rng.integers(5, 20))
            R = int(rng.choice([1, 3]))
            out.append((f"fuzzf/s{seed}_grouped_C{C}_g{groups}_k{R}_{H}", conv_grouped64_case(int(rng.integers(1, 3)), H, H, C, groups, R=R,
                                                                                           stride=int(rng.choice([1, 2])), pad=R // 2,
                                                                                           act=int(rng.choice([0, 1])),
                                                                                           res=bool(rng.random() < 0.3), seed=sd)))
        elif k == 2:
            h, w = int(rng.integers(1, 20)), int(rng.integers(1, 20))
            out.append((f"fuzzf/s{seed}_resize_{h}x{w}_{i}", resize_case(int(rng.integers(1, 4)), h, w, int(rng.choice([1, 3, 21, 40, 128])),
                                                                        h * int(rng.integers(1, 9)) + int(rng.integers(0, 3)),
                                                                        w * int(rng.integers(1, 9)) + int(rng.integers(0, 3)),
                                                                        dtype=str(rng.choice(["bf16", "fp32"])), nchw=bool(rng.random() < 0.5),
                                                                        seed=sd)))
        else:
            hw = int(rng.choice([16, 17, 28, 56, 112]))
            out.append((f"fuzzf/s{seed}_avgpool_wide_{i}", avgpool_case(int(rng.integers(1, 5)), hw, hw, int(rng.choice([8, 16, 32, 96, 240, 672])),
                                                                       1, 1, seed=sd)))
    return out


def fuzz_ln_fold_cases(n, seed=0):
    """Random shapes for the LayerNorm-fold pair where the library offers it (the 256 x 256 tile is the dispatch's choice for both
    Linears): stream widths 256 ... 768 in steps of 64 (1 - 3 statistics pieces, last tile 1 - 4 waves wide), ragged row counts,
    all three activations, row-major and head-major consumers."""
    L = _lib().load()
    rng = _rng(seed)
    out = []
    tries = 0
    while len(out) < n and tries < 400:
        tries += 1
        D = int(rng.choice([256, 320, 384, 448, 512, 576, 640, 704, 768]))
        Kp = int(rng.choice([256, 384, 512, 768, 1024]))
        tokens = int(rng.choice([0, 0, 50, 197]))
        N2 = 3 * D if tokens else int(rng.choice([256, 512, 768, 1024, 1536]))
        if tokens and D % 64:
            continue
        M = int(rng.integers(150 * 256 // ((D + 255) // 256), 150 * 256 // ((D + 255) // 256) + 3000))
        if tokens:
            M = (M // tokens + 1) * tokens
        if not (L.mv_linear_lnout_supported(M, D, Kp, 1) and L.mv_linear_lnin_supported(M, N2, D, tokens, 64 if tokens else 0, 1)):
            continue
        act = int(rng.integers(0, 3)) if not tokens else 0
        out.append((f"fuzz/ln_fold_{len(out)}_M{M}_D{D}_K{Kp}_N{N2}_act{act}_tok{tokens}",
                    ln_fold_case(M, D, Kp, N2, act=act, tokens=tokens, seed=900 + len(out), eps=float(rng.choice([1e-5, 1e-6])))))
    return out


def fuzz_stochastic_cases(n, seed=0):
    """Random shapes for the in-kernel / window-layout / per-token draws of the training-mode transformers: attention dropout
    (both qkv layouts, with and without the probabilities), Dropout in window layout, key splits."""
    rng = np.random.default_rng(seed)
    out = []
    for i in range(n):
        k = int(rng.integers(0, 3))
        sd = 6000 + i
        if k == 0:
            N, H, dh = int(rng.integers(1, 257)), int(rng.integers(1, 7)), int(rng.choice([32, 64]))
            out.append((f"fuzzs/s{seed}_mha_dropout_N{N}_H{H}_dh{dh}_{i}",
                        mha_dropout_case(int(rng.integers(1, 4)), N, H, dh, p=float(rng.choice([0.05, 0.3, 0.7])),
                                         head_major=bool(rng.random() < 0.5), probs=bool(rng.random() < 0.7), seed=sd)))
        elif k == 1:
            ws = (int(rng.integers(1, 9)), int(rng.integers(1, 9)))
            g = (int(rng.integers(1, 5)), int(rng.integers(1, 5)))
            shift = (int(rng.integers(0, ws[0])), int(rng.integers(0, ws[1])))
            C = 8 * int(rng.integers(1, 25))
            out.append((f"fuzzs/s{seed}_dropout_windows_{g[0] * ws[0]}x{g[1] * ws[1]}_C{C}_{i}",
                        dropout_windows_case(int(rng.integers(1, 4)), g[0] * ws[0], g[1] * ws[1], C, ws, shift,
                                             p=float(rng.choice([0.1, 0.5, 0.9])), dtype=str(rng.choice(["bf16", "fp32"])), seed=sd)))
        else:
            out.append((f"fuzzs/s{seed}_prng_split_{i}", prng_split_case(int(rng.integers(1, 3000)), int(rng.integers(1, 12)),
                                                                        int(rng.integers(0, 2)), seed=sd)))
    return out


def all_cases():
    c = []
    # ---- implicit-GEMM conv (MFMA) : ResNet-50 shapes at reduced spatial size + edge cases
    c += [("igemm/1x1_64_64", conv_nhwc_case(2, 14, 14, 64, 64, 1, 1, act=1)),
          ("igemm/1x1_64_256_res", conv_nhwc_case(2, 14, 14, 64, 256, 1, 1, act=1, res=True)),
          ("igemm/3x3_64_64", conv_nhwc_case(2, 14, 14, 64, 64, 3, 3, pad=1, act=1)),
          ("igemm/3x3_128_s2", conv_nhwc_case(2, 14, 14, 128, 128, 3, 3, stride=2, pad=1, act=1)),
          ("igemm/1x1_256_512_s2", conv_nhwc_case(2, 14, 14, 256, 512, 1, 1, stride=2)),
          ("igemm/3x3_512_7x7", conv_nhwc_case(3, 7, 7, 512, 512, 3, 3, pad=1, act=1, res=True)),
          ("igemm/3x3_dil2", conv_nhwc_case(1, 13, 11, 64, 128, 3, 3, pad=2, dil=2)),
          ("igemm/5x5_192", conv_nhwc_case(1, 13, 13, 64, 192, 5, 5, pad=2, act=1, scale=False)),
          ("igemm/1x1_tile128_K64", conv_nhwc_case(2, 9, 9, 128, 64, 1, 1, tile=1)),
          ("igemm/3x3_tile64_K256", conv_nhwc_case(2, 9, 9, 64, 256, 3, 3, pad=1, tile=2)),
          ("igemm/1x1_M1", conv_nhwc_case(1, 1, 1, 2048, 1000, 1, 1, out="fp32")),
          ("igemm/gelu", conv_nhwc_case(1, 8, 8, 128, 128, 1, 1, act=2)),
          ("igemm/big_M", conv_nhwc_case(8, 56, 56, 64, 64, 3, 3, pad=1, act=1, seed=3))]
    c += [("c3x3c64/56", conv_nhwc_case(4, 56, 56, 64, 64, 3, 3, pad=1, act=1, seed=41)),
          ("c3x3c64/odd_37x29_noact", conv_nhwc_case(9, 37, 29, 64, 64, 3, 3, pad=1, scale=False, seed=42)),
          ("c3x3c64/tiny_rows", conv_nhwc_case(70, 12, 10, 64, 64, 3, 3, pad=1, act=1, seed=43)),
          ("c3x3c64/wide_112_two_col_tiles", conv_nhwc_case(3, 30, 112, 64, 64, 3, 3, pad=1, act=1, seed=44)),
          ("c3x3c64/width_100_H_not_mult4", conv_nhwc_case(5, 27, 100, 64, 64, 3, 3, pad=1, act=2, seed=45)),
          ("c3x3c64/many_tiles_persistent", conv_nhwc_case(40, 56, 56, 64, 64, 3, 3, pad=1, act=1, seed=46)),
          ("igemm2/3x3_128_28", conv_nhwc_case(8, 28, 28, 128, 128, 3, 3, pad=1, act=1, seed=11)),
          ("igemm2/3x3_64_56_bn64", conv_nhwc_case(2, 56, 56, 64, 64, 3, 3, pad=1, act=1, seed=12, flags=("no_stream",))),
          ("igemm2/3x3_256_s2", conv_nhwc_case(24, 28, 28, 256, 256, 3, 3, stride=2, pad=1, act=1, seed=13)),
          ("igemm2/1x1_512_128", conv_nhwc_case(8, 28, 28, 512, 128, 1, 1, act=1, seed=14)),
          ("igemm2/1x1_256_512_s2", conv_nhwc_case(8, 56, 56, 256, 512, 1, 1, stride=2, seed=15)),
          ("igemm2/1x1_1024_256_res", conv_nhwc_case(24, 14, 14, 1024, 256, 1, 1, act=1, res=True, seed=16)),
          ("igemm2/3x3_K192_tail", conv_nhwc_case(5, 29, 31, 64, 192, 3, 3, pad=1, act=1, seed=17)),
          ("igemm2/3x3_dil2_oddM", conv_nhwc_case(3, 37, 41, 128, 128, 3, 3, pad=2, dil=2, seed=18)),
          ("igemm2/5x5", conv_nhwc_case(6, 27, 27, 64, 192, 5, 5, pad=2, act=1, scale=False, seed=19)),
          ("igemm2/t256_3x3_256", conv_nhwc_case(40, 14, 14, 256, 256, 3, 3, pad=1, act=1, res=True, seed=31, flags=("igemm2_tile=3",))),
          ("igemm2/t256_1x1_768_512_tail", conv_nhwc_case(3, 41, 43, 768, 512, 1, 1, act=2, seed=32, flags=("igemm2_tile=3",))),
          ("igemm2/t256_3x3_s2_K320", conv_nhwc_case(9, 33, 35, 128, 320, 3, 3, stride=2, pad=1, seed=33, flags=("igemm2_tile=3",))),
          ("igemm2/t256_f32out_res", conv_nhwc_case(30, 14, 14, 512, 256, 1, 1, res=True, out="fp32", seed=34, flags=("igemm2_tile=3",))),
          ("igemm2/t64_forced_K128", conv_nhwc_case(6, 28, 28, 128, 128, 3, 3, pad=1, seed=35, flags=("igemm2_tile=1",))),
          ("igemm8/1x1_512_256", conv_nhwc_case(8, 28, 28, 512, 256, 1, 1, act=1, seed=261, flags=("igemm8=2",))),
          ("igemm8/3x3_128_256", conv_nhwc_case(8, 28, 28, 128, 256, 3, 3, pad=1, act=1, seed=262, flags=("igemm8=2",))),
          ("igemm8/3x3_s2_C128_K320", conv_nhwc_case(9, 33, 35, 128, 320, 3, 3, stride=2, pad=1, seed=263, flags=("igemm8=2",))),
          ("igemm8/3x3_dil2_oddM", conv_nhwc_case(3, 37, 41, 64, 192, 3, 3, pad=2, dil=2, seed=264, flags=("igemm8=2",))),
          ("igemm8/f32out_res", conv_nhwc_case(30, 14, 14, 512, 256, 1, 1, res=True, out="fp32", seed=265, flags=("igemm8=2",))),
          ("igemm8/bf16_res_relu", conv_nhwc_case(40, 14, 14, 256, 256, 3, 3, pad=1, act=1, res=True, seed=266, flags=("igemm8=2",))),
          ("igemm8/gelu_tail", conv_nhwc_case(3, 41, 43, 768, 512, 1, 1, act=2, seed=267, flags=("igemm8=2",))),
          ("igemm8/k64_one_tile", conv_nhwc_case(4, 20, 20, 64, 96, 1, 1, act=1, seed=268, flags=("igemm8=2",))),
          ("igemm8/k128_two_tiles", conv_nhwc_case(4, 20, 20, 128, 256, 1, 1, seed=269, flags=("igemm8=2",))),
          ("igemm8/k192_three_tiles", conv_nhwc_case(4, 20, 20, 192, 264, 1, 1, seed=270, flags=("igemm8=2",))),
          ("igemm8/5x5", conv_nhwc_case(6, 27, 27, 64, 192, 5, 5, pad=2, act=1, scale=False, seed=271, flags=("igemm8=2",))),
          ("igemm8/3x3_7x7_512", conv_nhwc_case(11, 7, 7, 512, 512, 3, 3, pad=1, act=1, seed=272, flags=("igemm8=2",))),
          ("igemm8/1x1_s2_256_512", conv_nhwc_case(5, 56, 56, 256, 512, 1, 1, stride=2, seed=273, flags=("igemm8=2",))),
          ("igemm8/linear_vit_fc2_f32res", linear_case(197 * 24, 3072, 768, res=True, out="fp32", seed=274, flags=("igemm8=2",))),
          ("igemm8/dual_layer2_entry_s2", dual_case(8, 28, 28, 128, 256, 512, 2, seed=275, flags=("igemm8=2",))),
          ("igemm8/dual_layer4_entry_ragged", dual_case(86, 7, 7, 512, 1024, 2048, 2, seed=276, flags=("igemm8=2",))),
          ("igemm8/dual_s1_64_64_K200_noact", dual_case(3, 37, 41, 64, 64, 200, 1, act=0, seed=277, flags=("igemm8=2",))),
          ("igemm8/qkv_heads_vit_base", qkv_heads_case(32, 197, 12, 64, seed=278, flags=("igemm8=2",))),
          # two-way split-K of the half-size tiles (mv_set_scratch): taken, repeatable bit for bit, arrival words left zero
          ("splitk/3x3_7x7_512_res", conv_nhwc_case(64, 7, 7, 512, 512, 3, 3, pad=1, act=1, res=True, seed=601, splitk=True)),
          ("splitk/1x1_2048_512_dense", conv_nhwc_case(64, 7, 7, 2048, 512, 1, 1, act=1, seed=602, splitk=True, flags=("splitk_min_nk=12",))),
          ("splitk/3x3_s2_14_512", conv_nhwc_case(64, 14, 14, 512, 512, 3, 3, stride=2, pad=1, act=1, seed=603, splitk=True)),
          ("splitk/256x128_3x3_256_128", conv_nhwc_case(8, 56, 56, 256, 128, 3, 3, pad=1, act=1, seed=604, splitk=True, flags=("splitk_min_nk=12",))),
          ("splitk/linear_swin_fc2_f32res", linear_case(3136, 3072, 768, res=True, out="fp32", seed=605, splitk=True)),
          ("splitk/linear_proj_nk12_ragged_M", linear_case(3000, 768, 768, res=True, out="fp32", seed=606, splitk=True, flags=("splitk_min_nk=12",))),
          ("splitk/linear_odd_nk13_gelu", linear_case(3136, 832, 512, act=2, seed=607, splitk=True, flags=("splitk_min_nk=12",))),
          ("splitk/dual_second_half_in_source2", dual_case(96, 7, 7, 512, 1024, 512, 2, seed=608, splitk=True, flags=("splitk_min_nk=12",))),
          ("splitk/dual_crossing_inside_half", dual_case(96, 7, 7, 1024, 512, 512, 2, seed=609, splitk=True, flags=("splitk_min_nk=12",))),
          ("splitk/linear_split_swin_merge", linear_split_case(3136, 1536, 768, seed=610, splitk=True)),
          ("splitk/scratch_protocol_edges", splitk_protocol_case(seed=611)),
          ("igemm8s/128x256_1x1_512_256", conv_nhwc_case(8, 28, 28, 512, 256, 1, 1, act=1, seed=361, flags=("igemm8=3",))),
          ("igemm8s/128x256_3x3_128_256", conv_nhwc_case(8, 28, 28, 128, 256, 3, 3, pad=1, act=1, seed=362, flags=("igemm8=3",))),
          ("igemm8s/128x256_3x3_s2_K320", conv_nhwc_case(9, 33, 35, 128, 320, 3, 3, stride=2, pad=1, seed=363, flags=("igemm8=3",))),
          ("igemm8s/128x256_dil2_oddM", conv_nhwc_case(3, 37, 41, 64, 192, 3, 3, pad=2, dil=2, seed=364, flags=("igemm8=3",))),
          ("igemm8s/128x256_f32out_res", conv_nhwc_case(30, 14, 14, 512, 256, 1, 1, res=True, out="fp32", seed=365, flags=("igemm8=3",))),
          ("igemm8s/128x256_k64_one_tile", conv_nhwc_case(4, 20, 20, 64, 96, 1, 1, act=1, seed=368, flags=("igemm8=3",))),
          ("igemm8s/128x256_k128_two", conv_nhwc_case(4, 20, 20, 128, 256, 1, 1, seed=369, flags=("igemm8=3",))),
          ("igemm8s/128x256_k256_four", conv_nhwc_case(4, 20, 20, 256, 264, 1, 1, seed=370, flags=("igemm8=3",))),
          ("igemm8s/128x256_3x3_7x7_512", conv_nhwc_case(11, 7, 7, 512, 512, 3, 3, pad=1, act=1, res=True, seed=372, flags=("igemm8=3",))),
          ("igemm8s/128x256_gelu_tail", conv_nhwc_case(3, 41, 43, 768, 512, 1, 1, act=2, seed=367, flags=("igemm8=3",))),
          ("igemm8s/128x256_dual_layer4", dual_case(86, 7, 7, 512, 1024, 2048, 2, seed=376, flags=("igemm8=3",))),
          ("igemm8s/128x256_qkv_heads", qkv_heads_case(32, 197, 12, 64, seed=378, flags=("igemm8=3",))),
          ("igemm8s/256x128_1x1_512_128", conv_nhwc_case(8, 28, 28, 512, 128, 1, 1, act=1, seed=461, flags=("igemm8=4",))),
          ("igemm8s/256x128_3x3_128_128", conv_nhwc_case(8, 28, 28, 128, 128, 3, 3, pad=1, act=1, seed=462, flags=("igemm8=4",))),
          ("igemm8s/256x128_3x3_s2_K320", conv_nhwc_case(9, 33, 35, 128, 320, 3, 3, stride=2, pad=1, seed=463, flags=("igemm8=4",))),
          ("igemm8s/256x128_dil2_oddM_K72", conv_nhwc_case(3, 37, 41, 64, 72, 3, 3, pad=2, dil=2, seed=464, flags=("igemm8=4",))),
          ("igemm8s/256x128_f32out_res", conv_nhwc_case(30, 14, 14, 512, 128, 1, 1, res=True, out="fp32", seed=465, flags=("igemm8=4",))),
          ("igemm8s/256x128_k64_one_tile", conv_nhwc_case(4, 20, 20, 64, 96, 1, 1, act=1, seed=468, flags=("igemm8=4",))),
          ("igemm8s/256x128_k192_three", conv_nhwc_case(4, 20, 20, 192, 264, 1, 1, seed=470, flags=("igemm8=4",))),
          ("igemm8s/256x128_5x5", conv_nhwc_case(6, 27, 27, 64, 192, 5, 5, pad=2, act=1, scale=False, seed=471, flags=("igemm8=4",))),
          ("igemm8s/256x128_dual_s1", dual_case(3, 37, 41, 64, 64, 200, 1, act=0, seed=477, flags=("igemm8=4",))),
          ("head/f32_resnet_fc_B256", linear_f32_head_case(256, 2048, 1000, seed=501)),
          ("head/f32_vit_head_B200", linear_f32_head_case(200, 768, 1000, seed=502)),
          ("head/f32_odd_rows_1", linear_f32_head_case(1, 768, 1000, seed=503)),
          ("head/f32_K72_N40_M33", linear_f32_head_case(33, 72, 40, seed=504)),
          ("split/linear_swin_merge_384_192", linear_split_case(128 * 28 * 28 // 4, 384, 192, seed=511)),
          ("split/linear_swin_merge_1536_768", linear_split_case(64 * 49, 1536, 768, seed=512)),
          ("split/linear_bf16out_ragged", linear_split_case(9000 + 37, 256, 200, out="bf16", seed=513)),
          ("dwconv/mbv2_96_112_s2", dwconv_case(2, 112, 112, 96, stride=2, seed=550)),
          ("dwconv/mbv2_144_56", dwconv_case(3, 56, 56, 144, seed=551)),
          ("dwconv/mbv2_960_7", dwconv_case(4, 7, 7, 960, seed=552)),
          ("dwconv/w_tail_s1_13x17", dwconv_case(2, 13, 17, 40, seed=560)),
          ("dwconv/w_tail_s2_15x21", dwconv_case(3, 15, 21, 16, stride=2, seed=561)),
          ("dwconv/k5_s1_120_28", dwconv_case(3, 28, 28, 120, R=5, pad=2, seed=562)),
          ("dwconv/k5_s2_72_w_tail", dwconv_case(2, 27, 31, 72, R=5, stride=2, pad=2, seed=563)),
          ("dwconv/k5_hard_swish_fused", dwconv_case(2, 14, 14, 672, R=5, pad=2, act=3, seed=564)),
          ("dwconv/k5_tile_ragged_9x21_c200", dwconv_case(3, 9, 21, 200, R=5, pad=2, act=6, seed=565)),
          ("dwconv/k5_tile_one_pixel_c8", dwconv_case(2, 1, 1, 8, R=5, pad=2, seed=566)),
          ("dwconv/k3_tile_flag_57x33_c72", dwconv_case(2, 57, 33, 72, seed=567, flag="dwconv_tile3")),
          ("dwconv/k5_register_window_flag", dwconv_case(3, 28, 28, 120, R=5, pad=2, seed=562, flag="dwconv_no_tile")),
          ("dwconv/odd_hw_k5_dil2_noscale_noact", dwconv_case(2, 13, 17, 24, R=5, pad=4, dil=2, act=0, scale=False, seed=553)),
          ("oddc/pw_24_144", conv_nhwc_case(4, 56, 56, 24, 144, 1, 1, act=1, seed=554)),
          ("oddc/pw_144_24_res", conv_nhwc_case(4, 56, 56, 144, 24, 1, 1, act=0, res=True, seed=555)),
          ("oddc/pw_16_96", conv_nhwc_case(2, 112, 112, 16, 96, 1, 1, act=1, seed=556)),
          ("stream_narrow/16_96_silu", conv_nhwc_case(2, 112, 112, 16, 96, 1, 1, act=6, seed=570)),
          ("stream_narrow/32_192_hswish_ragged", conv_nhwc_case(3, 57, 57, 32, 192, 1, 1, act=3, seed=571)),
          ("stream_narrow/48_288_relu", conv_nhwc_case(4, 56, 56, 48, 288, 1, 1, act=1, seed=572)),
          ("stream_narrow/80_480_hswish", conv_nhwc_case(12, 28, 28, 80, 480, 1, 1, act=3, seed=573)),
          ("stream_narrow/112_672_silu", conv_nhwc_case(12, 28, 28, 112, 672, 1, 1, act=6, seed=574)),
          ("stream_narrow/144_24_res_project", conv_nhwc_case(4, 56, 56, 144, 24, 1, 1, act=0, res=True, seed=575)),
          ("stream_narrow/160_960_hswish", conv_nhwc_case(48, 14, 14, 160, 960, 1, 1, act=3, seed=576)),
          ("stream_narrow/96_576_hsigmoid_f32out", conv_nhwc_case(4, 56, 56, 96, 576, 1, 1, act=4, out="fp32", seed=577)),
          ("stream_narrow/pad_24_144_relu", conv_nhwc_case(4, 56, 56, 24, 144, 1, 1, act=1, seed=578)),
          ("stream_narrow/pad_40_240_hswish_ragged", conv_nhwc_case(11, 28, 28, 40, 240, 1, 1, act=3, seed=579)),
          ("stream_narrow/pad_72_24_res", conv_nhwc_case(4, 56, 56, 72, 24, 1, 1, act=0, res=True, seed=580)),
          ("stream_narrow/pad_120_480_silu", conv_nhwc_case(12, 28, 28, 120, 480, 1, 1, act=6, seed=581)),
          ("stream_narrow/pad_184_80", conv_nhwc_case(48, 14, 14, 184, 80, 1, 1, act=0, seed=582)),
          ("stream_narrow/pad_200_80_f32out", conv_nhwc_case(48, 14, 14, 200, 80, 1, 1, act=0, out="fp32", seed=583)),
          ("stream_narrow/pad_232_464_relu", conv_nhwc_case(48, 14, 14, 232, 464, 1, 1, act=1, seed=584)),
          ("stream_narrow/exact_240_40_res", conv_nhwc_case(12, 28, 28, 240, 40, 1, 1, act=0, res=True, seed=585)),
          ("stream_narrow/off_16_96", conv_nhwc_case(2, 112, 112, 16, 96, 1, 1, act=6, seed=570, flags=("no_stream_narrow",))),
          ("oddc/pw_160_960", conv_nhwc_case(8, 7, 7, 160, 960, 1, 1, act=1, seed=557)),
          ("oddc/conv3x3_40_72_s2", conv_nhwc_case(2, 19, 19, 40, 72, 3, 3, stride=2, pad=1, act=1, seed=558)),
          ("oddc/linear_M77_K200_N136", conv_nhwc_case(1, 77, 1, 200, 136, 1, 1, act=0, seed=559)),
          ("grouped64/resnext_layer1_128_g32", conv_grouped64_case(3, 28, 28, 128, 32, seed=540)),
          ("grouped64/resnext_layer2_256_g32_s2", conv_grouped64_case(2, 28, 28, 256, 32, stride=2, seed=541)),
          ("grouped64/resnext101_256_g32_w8", conv_grouped64_case(2, 14, 14, 256, 32, seed=542)),
          ("grouped64/cg32_1024_res_noact", conv_grouped64_case(1, 7, 7, 1024, 32, act=0, res=True, seed=543)),
          ("grouped64/cg64_is_plain_grouping_dil2", conv_grouped64_case(2, 15, 15, 128, 2, pad=2, dil=2, seed=544)),
          ("grouped64/regnet_104_gw8", conv_grouped64_case(2, 28, 28, 104, 13, seed=546)),
          ("grouped64/regnet_208_gw24_s2", conv_grouped64_case(2, 28, 28, 216, 9, stride=2, seed=547)),
          ("grouped64/regnet_gw56_448", conv_grouped64_case(2, 14, 14, 448, 8, seed=548)),
          ("grouped64/regnet_gw264_528", conv_grouped64_case(1, 7, 7, 528, 2, seed=549)),
          ("grouped64/k1_cg16", conv_grouped64_case(2, 9, 9, 64, 4, R=1, pad=0, seed=545)),
          ("act/hard_swish", eltwise_act_case("hard_swish", 3, seed=570)),
          ("act/hard_sigmoid_f32", eltwise_act_case("hard_sigmoid", 4, dtype="fp32", seed=571)),
          ("act/sigmoid_f32", eltwise_act_case("sigmoid", 5, dtype="fp32", seed=572)),
          ("act/silu", eltwise_act_case("silu", 6, seed=573)),
          ("act/conv_fused_silu_pw_96_576", conv_nhwc_case(4, 28, 28, 96, 576, 1, 1, act=6, seed=576)),
          ("act/conv_fused_hard_swish_pw_640_128", conv_nhwc_case(2, 14, 14, 640, 128, 1, 1, act=3, seed=577)),
          ("act/conv_fused_silu_3x3_s2_64_48", conv_nhwc_case(2, 30, 30, 64, 48, 3, 3, stride=2, pad=1, act=6, seed=578)),
          ("act/conv_fused_sigmoid_k64_res", conv_nhwc_case(2, 9, 9, 128, 64, 1, 1, act=5, res=True, seed=579)),
          ("act/conv_generic_hard_sigmoid_f32", conv_nhwc_case(1, 7, 7, 12, 20, 1, 1, act=4, dtype="fp32", seed=580)),
          ("act/not_fused_into_gemm_entries", fused_act_refused_case()),
          ("act/se_channel_scale", channel_scale_case(3, 14 * 14, 120, seed=574)),
          ("act/se_channel_scale_f32", channel_scale_case(2, 49, 24, dtype="fp32", seed=575)),
          ("resize/logits_28_to_224_nchw", resize_case(2, 28, 28, 21, 224, 224)),
          ("resize/pooled_1_to_28_nhwc", resize_case(3, 1, 1, 256, 28, 28, nchw=False)),
          ("resize/odd_7x5_to_20x33_f32", resize_case(2, 7, 5, 3, 20, 33, dtype="fp32")),
          ("resize/identity_size_nhwc_f32", resize_case(1, 9, 9, 8, 9, 9, dtype="fp32", nchw=False)),
          ("resize/downsample_refused", resize_down_refused_case()),
          ("concat/aspp_5x256", copy_rows_case(2 * 28 * 28, (256, 256, 256, 256, 256))),
          ("concat/odd_widths_f32", copy_rows_case(37, (3, 5, 2), dtype="fp32")),
          ("concat/odd_widths_bf16", copy_rows_case(37, (3, 8, 1))),
          ("ln_linear/swin_qkv_96_288_f32stream", ln_linear_case(4 * 56 * 56, 96, 288, "fp32", seed=530)),
          ("ln_linear/swin_fc1_192_768_gelu", ln_linear_case(8 * 28 * 28 * 2, 192, 768, "fp32", act=2, seed=531, flags=(("ln_stream_192", 1),))),
          ("ln_linear/swin_qkv_192_576_bf16stream_ragged", ln_linear_case(8192 + 45, 192, 576, "bf16", seed=532, flags=(("ln_stream_192", 1),))),
          ("ln_linear/swin_fc1_96_384_gelu_bf16stream", ln_linear_case(8192 + 45, 96, 384, "bf16", act=2, seed=534)),
          ("ln_linear/k96_N200_tail", ln_linear_case(9000, 96, 200, "fp32", seed=533)),
          ("ln_fold/vit_base_proj_fc1_gelu", ln_fold_case(64 * 197, 768, 768, 3072, act=2, seed=540)),
          ("ln_fold/vit_base_fc2_qkv_heads_ragged", ln_fold_case(65 * 197, 768, 3072, 2304, tokens=197, seed=541)),
          ("ln_fold/vit_small_width_384", ln_fold_case(98 * 197, 384, 384, 1536, act=2, seed=542)),
          ("ln_fold/row_means_of_2_sigma", ln_fold_case(64 * 197 + 31, 768, 768, 768, seed=543, row_mean=2.0)),
          ("ln_fold/width_512_two_pieces_M_multiple_of_256", ln_fold_case(75 * 256, 512, 512, 1024, act=1, seed=544)),
          ("ln_fold/width_320_last_tile_one_wave_wide", ln_fold_case(75 * 256 + 1, 320, 256, 640, seed=545, eps=1e-5)),
          ("se_scale/effnet_32_8_silu_sigmoid_112", se_scale_case(3, 112, 112, 32, 8, act1=6, act2=5, seed=590)),
          ("se_scale/effnet_1152_48_7x7", se_scale_case(5, 7, 7, 1152, 48, act1=6, act2=5, seed=591)),
          ("se_scale/mbv3_72_24_relu_hsigmoid", se_scale_case(4, 28, 28, 72, 24, act1=1, act2=4, seed=592)),
          ("se_scale/regnet_104_26_odd_squeeze", se_scale_case(4, 14, 14, 104, 26, act1=1, act2=5, seed=593)),
          ("se_scale/squeeze_4_one_pixel", se_scale_case(2, 1, 1, 96, 4, act1=6, act2=5, seed=594)),
          ("se_scale/wide_2048_512", se_scale_case(2, 5, 5, 2048, 512, act1=1, act2=5, seed=595)),
          ("moments/c64_map_bf16", moments_case(8 * 112 * 112, 64, "bf16", seed=550)),
          ("moments/c2048_few_rows", moments_case(8 * 7 * 7, 2048, "bf16", seed=551)),
          ("moments/c96_fp32_ragged", moments_case(12345, 96, "fp32", seed=552)),
          ("moments/c8_one_row", moments_case(1, 8, "fp32", seed=553)),
          ("moments/single_pass_c64_map", moments2_case(8 * 56 * 56, 64, "bf16", seed=554)),
          ("moments/single_pass_c2048", moments2_case(300, 2048, "fp32", seed=555)),
          ("moments/single_pass_far_shift", moments2_case(5000, 96, "fp32", seed=556, off=3.0)),
          ("dropout/map_chw_bf16", dropout_case(3, 24 * 35, 24, True, 0.4, "bf16", seed=560)),
          ("dropout/rows_odd_count_fp32", dropout_case(2, 7 * 9, 9, False, 0.5, "fp32", seed=561)),
          ("dropout/one_element", dropout_case(4, 1, 1, False, 0.5, "fp32", seed=562)),
          ("dropout/map_chw_c32_pairs", dropout_case(3, 32 * 35, 32, True, 0.3, "bf16", seed=563)),
          ("dropout/map_chw_c32_x8", dropout_case(3, 32 * 35, 32, True, 0.3, "bf16", seed=563, flag="dropout_x8")),
          ("dropout/map_chw_c32_scalar", dropout_case(3, 32 * 35, 32, True, 0.3, "fp32", seed=563, flag="dropout_scalar")),
          ("dropout/rows_pairs_fp32", dropout_case(2, 48 * 16, 16, False, 0.5, "fp32", seed=564)),
          ("dropout/vec_9216_pairs", dropout_case(5, 9216, 9216, False, 0.5, "bf16", seed=565)),
          ("ln_mlp/swin_stage0_f32stream", ln_mlp_case(8 * 56 * 56, "fp32", seed=520)),
          ("ln_mlp/bf16stream_ragged", ln_mlp_case(4096 + 77, "bf16", seed=521)),
          ("ln_mlp/f32stream_many_tiles", ln_mlp_case(70001, "fp32", seed=522)),
          ("patch4_ln/k96_224_B2_split", patch4_ln_case(2, 224, seed=545)),
          ("patch4_ln/k96_56_B5_nosplit_ragged", patch4_ln_case(5, 56, split=False, seed=546)),
          ("patch4_ln/k128_112_B3_split", patch4_ln_case(3, 112, K=128, seed=547)),
          ("patch_merge_ln/c96_56_B2", patch_merge_ln_case(2, 56, 96, seed=540)),
          ("patch_merge_ln/c192_28_B3", patch_merge_ln_case(3, 28, 192, seed=541)),
          ("patch_merge_ln/c384_14_B5_f32out", patch_merge_ln_case(5, 14, 384, out="fp32", seed=542)),
          ("patch_merge_ln/c32_10_B1", patch_merge_ln_case(1, 10, 32, seed=543)),
          ("swin_block_attn/c384_14x14_B3_noshift", swin_block_attn_case(3, 14, 0, seed=530)),
          ("swin_block_attn/c384_14x14_B3_shift3", swin_block_attn_case(3, 14, 3, seed=531)),
          ("swin_block_attn/c384_28x28_B2_shift3", swin_block_attn_case(2, 28, 3, seed=532)),
          ("swin_block_attn/c192_28x28_B2_noshift", swin_block_attn_case(2, 28, 0, seed=533, C=192, heads=6)),
          ("swin_block_attn/c192_28x28_B3_shift3", swin_block_attn_case(3, 28, 3, seed=534, C=192, heads=6)),
          ("swin_block_attn/c192_14x14_B2_shift3", swin_block_attn_case(2, 14, 3, seed=535, C=192, heads=6)),
          ("swin_block_attn/c96_56x56_B1_noshift", swin_block_attn_case(1, 56, 0, seed=536, C=96, heads=3)),
          ("swin_block_attn/c96_56x56_B2_shift3", swin_block_attn_case(2, 56, 3, seed=537, C=96, heads=3)),
          ("swin_block_attn/c96_14x14_B3_shift3", swin_block_attn_case(3, 14, 3, seed=538, C=96, heads=3)),
          ("swin_block_attn/c96_56x56_B2_shift3_shared_kernel", swin_block_attn_case(2, 56, 3, seed=539, C=96, heads=3, flags=("swin_c96_shared",))),
          ("ln_mlp/stream_c384_swin_stage2_B8", ln_mlp_case(8 * 14 * 14, "fp32", seed=523, C=384, Hd=1536)),
          ("ln_mlp/stream_c384_ragged", ln_mlp_case(64 * 9 + 37, "fp32", seed=524, C=384, Hd=1536)),
          ("ln_mlp/stream_c384_many_tiles", ln_mlp_case(64 * 300 + 5, "fp32", seed=525, C=384, Hd=1536)),
          ("ln_mlp/stream_c192_swin_stage1_B4", ln_mlp_case(4 * 28 * 28, "fp32", seed=526, C=192, Hd=768)),
          ("ln_mlp/stream_c192_ragged", ln_mlp_case(128 * 5 + 77, "fp32", seed=527, C=192, Hd=768)),
          ("ln_mlp/stream_c192_many_tiles", ln_mlp_case(128 * 300 + 9, "fp32", seed=528, C=192, Hd=768)),
          ("split/conv_swin_patch4", conv_nchw_split_case(3, 3, 224, 96, 4, 4, seed=514)),
          ("split/conv_odd_k3s2", conv_nchw_split_case(2, 3, 65, 40, 3, 2, seed=515)),
          ("chain/56x56_B4", chain_case(4 * 56 * 56, seed=1)),
          ("chain/ragged_M", chain_case(8192 + 37, seed=2)),
          ("chain/many_tiles", chain_case(40 * 56 * 56 + 5, seed=3)),
          ("dual/layer2_entry_s2", dual_case(8, 28, 28, 128, 256, 512, 2, seed=1)),
          ("dual/layer3_entry_s2", dual_case(24, 14, 14, 256, 512, 1024, 2, seed=2)),
          ("dual/layer4_entry_s2_ragged", dual_case(86, 7, 7, 512, 1024, 2048, 2, seed=3)),
          ("dual/s1_64_64_K200_noact", dual_case(3, 37, 41, 64, 64, 200, 1, act=0, seed=4)),
          ("dual/s2_odd_input", dual_case(5, 31, 29, 192, 128, 320, 2, seed=5)),
          ("bneck_tail/14x14_B1", bneck_tail_case(1, seed=11)),
          ("bneck_tail/14x14_B5", bneck_tail_case(5, seed=12)),
          ("bneck_tail/14x14_B3_big", bneck_tail_case(3, seed=13, big=True)),
          ("bwd/conv3x3_s1_p1", conv_bwd_case(2, 14, 14, 64, 96, 3, 1, 1, seed=31)),
          ("bwd/conv3x3_s2_p1_odd", conv_bwd_case(2, 15, 13, 24, 40, 3, 2, 1, seed=32)),
          ("bwd/conv7x7_s2_p3_stem", conv_bwd_case(1, 32, 32, 3, 16, 7, 2, 3, seed=33)),
          ("bwd/conv11x11_s4_p2", conv_bwd_case(1, 67, 67, 3, 8, 11, 4, 2, seed=34)),
          ("bwd/conv1x1_s2_downsample", conv_bwd_case(2, 14, 14, 128, 256, 1, 2, 0, seed=35)),
          ("bwd/conv16x16_s16_patch", conv_bwd_case(2, 32, 32, 3, 48, 16, 16, 0, seed=36)),
          ("bwd/conv3x3_dil2", conv_bwd_case(1, 12, 12, 16, 16, 3, 1, 2, 2, seed=37)),
          ("bwd/conv3x3_c256_k512_28", conv_bwd_case(1, 28, 28, 256, 512, 3, 1, 1, seed=38)),
          ("bwd/conv3x3_depthwise_s2", conv_bwd_case(2, 14, 14, 48, 48, 3, 2, 1, seed=39, groups=48)),
          ("bwd/conv3x3_groups4", conv_bwd_case(2, 12, 12, 32, 64, 3, 1, 1, seed=40, groups=4)),
          ("bwd/conv5x5_depthwise", conv_bwd_case(1, 14, 14, 40, 40, 5, 1, 2, seed=51, groups=40)),
          ("bwd/conv3x3_c64_56_many_positions", conv_bwd_case(2, 56, 56, 64, 64, 3, 1, 1, seed=61)),
          ("bwd/colsum_100k_x_64", colsum_case(100352, 64, seed=62)),
          ("bwd/colsum_product_5000_x_200", colsum_case(5000, 200, seed=63, product=True)),
          ("bwd/colsum_small", colsum_case(300, 10, seed=64)),
          # fp32 compute mode / training-step forward: the fp32 matrix-core contractions and the LDS attention, at op level
          ("f32/conv3x3_c24_k40_direct", expect_kernel(conv_nhwc_case(2, 15, 13, 24, 40, 3, 3, pad=1, act=1, dtype="fp32", seed=701), "conv_f32_mfma")),
          ("f32/conv7x7_s2_c3_stem_direct", expect_kernel(conv_nhwc_case(2, 32, 32, 3, 16, 7, 7, stride=2, pad=3, act=1, dtype="fp32", seed=702), "conv_f32_mfma")),
          ("f32/conv3x3_s2_c128_k96_res_lds", expect_kernel(conv_nhwc_case(8, 28, 28, 128, 96, 3, 3, stride=2, pad=1, act=1, res=True, dtype="fp32", seed=703), "conv_f32_lds_mfma")),
          ("f32/conv1x1_c68_half_empty_chunk_lds", expect_kernel(conv_nhwc_case(6, 20, 20, 68, 72, 1, 1, dtype="fp32", seed=704), "conv_f32_lds_mfma")),
          ("f32/conv3x3_dil2_c64_lds", expect_kernel(conv_nhwc_case(3, 21, 19, 64, 64, 3, 3, pad=2, dil=2, act=2, scale=False, dtype="fp32", seed=705), "conv_f32_lds_mfma")),
          ("f32/linear_768_3072_gelu_lds", expect_kernel(linear_case(1576, 768, 3072, act=2, dtype="fp32", seed=706), "conv_f32_lds_mfma")),
          ("f32/linear_ragged_1500_200_k100_lds", expect_kernel(linear_case(1500, 200, 100, res=True, dtype="fp32", seed=707), "conv_f32_lds_mfma")),
          ("f32/linear_few_rows_skinny", expect_kernel(linear_case(300, 96, 288, dtype="fp32", seed=708), "skinny_linear_f32_mfma")),
          ("f32/linear_c48_direct", expect_kernel(linear_case(2000, 48, 96, act=1, dtype="fp32", seed=713), "conv_f32_mfma")),
          ("f32/mha_197_dh64_lds", expect_kernel(mha_case(16, 197, 12, 64, dtype="fp32", seed=709), "mha_f32_lds")),
          ("f32/mha_50_dh32_noprobs_lds", expect_kernel(mha_case(40, 50, 4, 32, dtype="fp32", probs=False, seed=710), "mha_f32_lds")),
          ("f32/swin_attn_shifted_lds", expect_kernel(swin_attn_case(2, 14, 96, 3, 7, 3, dtype="fp32", seed=711), "swin_attn_f32_lds")),
          ("f32/swin_attn_one_window_lds", expect_kernel(swin_attn_case(3, 7, 768, 24, 7, 3, dtype="fp32", seed=712), "swin_attn_f32_lds")),
          ("bwd/mha_197_dh64", mha_bwd_case(2, 197, 3, 64, seed=65)),
          ("bwd/mha_50_dh32", mha_bwd_case(3, 50, 4, 32, seed=66)),
          ("bwd/mha_17_dh96", mha_bwd_case(1, 17, 2, 96, seed=67)),
          ("bwd/maxpool_3x3_s2", maxpool_bwd_case(2, 27, 27, 64, 3, 2, 0, seed=41)),
          ("bwd/maxpool_3x3_s2_p1", maxpool_bwd_case(2, 28, 28, 64, 3, 2, 1, seed=42)),
          ("bwd/maxpool_2x2_s2", maxpool_bwd_case(1, 56, 56, 256, 2, 2, 0, seed=43)),
          ("bwd/maxpool_2x2_s2_relu_ties", maxpool_bwd_case(1, 56, 56, 256, 2, 2, 0, seed=44, relu=True)),
          ("bwd/layernorm_197x192", rowwise_bwd_case("layernorm", 394, 192, seed=45)),
          ("bwd/layernorm_odd_70", rowwise_bwd_case("layernorm", 33, 70, seed=46)),
          ("bwd/softmax_197", rowwise_bwd_case("softmax", 197, 197, seed=47)),
          ("bwd/relu", rowwise_bwd_case("relu", 100, 333, seed=48)),
          ("bwd/gelu_tanh", rowwise_bwd_case("gelu", 100, 333, seed=49)),
          ("bwd/hard_swish", rowwise_bwd_case("hard_swish", 64, 200, seed=52)),
          ("bwd/hard_sigmoid", rowwise_bwd_case("hard_sigmoid", 64, 200, seed=53)),
          ("bwd/sigmoid", rowwise_bwd_case("sigmoid", 64, 200, seed=54)),
          ("bwd/silu", rowwise_bwd_case("silu", 64, 200, seed=55)),
          ("bwd/softmax_xent_adam", xent_adam_case(7, 10, seed=50)),
          ("bwd/swin_attn_unshifted_56_c96", swin_bwd_case(2, 14, 96, 3, 0, seed=56)),
          ("bwd/swin_attn_shifted_c96", swin_bwd_case(2, 14, 96, 3, 3, seed=57)),
          ("bwd/swin_attn_shifted_c192_28", swin_bwd_case(1, 28, 192, 6, 3, seed=58)),
          ("bwd/swin_attn_one_window_c768", swin_bwd_case(2, 7, 768, 24, 3, seed=59)),
          ("bwd/patch_merge", patch_merge_bwd_case(2, 14, 96, seed=60)),
          ("chain/dual_56x56_B4", dual_chain_case(4 * 56 * 56, seed=6)),
          ("chain/dual_ragged_many", dual_chain_case(29 * 56 * 56 + 13, seed=7)),
          ("chain/rc_56x56_B4", chain_rc_case(4 * 56 * 56, seed=21)),
          ("chain/rc_ragged_many_tiles", chain_rc_case(31 * 56 * 56 + 19, seed=22)),
          ("chain/rc_min_rows", chain_rc_case(8192, seed=23)),
          ("chain/res_56x56_B4_ysub2", chain_res_case(4, 56, 56, 2, seed=26)),
          ("chain/res_56x56_B4_full_y", chain_res_case(4, 56, 56, 0, seed=27)),
          ("chain/res_ragged_B37_28x30_ysub2", chain_res_case(37, 28, 30, 2, seed=28)),
          ("chain/res_ragged_B33_full_y_many_tiles", chain_res_case(33, 57, 55, 0, seed=29)),
          ("chain/ysub2_56x56_B4", chain_sub_case(4, 56, 56, seed=24)),
          ("chain/ysub2_ragged_B37_28x30", chain_sub_case(37, 28, 30, seed=25)),
          ("chain/n128_56x56_B4", chain_case(4 * 56 * 56, seed=4, N2=128)),
          ("chain/n128_ragged_many", chain_case(33 * 56 * 56 + 21, seed=5, N2=128)),
          ("chain/stream_28x28_B32", chain_case(32 * 28 * 28, seed=8, N2=128, C=128, K=512)),
          ("chain/stream_ragged", chain_case(16384 + 37, seed=9, N2=128, C=128, K=512)),
          ("chain/stream_many_rounds", chain_case(150 * 28 * 28 + 5, seed=10, N2=128, C=128, K=512)),
          ("igemm/old_kernel_3x3_128_28", conv_nhwc_case(8, 28, 28, 128, 128, 3, 3, pad=1, act=1, seed=11, flags=("no_igemm2",))),
          ("igemm/old_kernel_1x1_64_256", conv_nhwc_case(4, 56, 56, 64, 256, 1, 1, act=1, res=True, flags=("no_stream",))),
          ("stream/64_256_res", conv_nhwc_case(4, 56, 56, 64, 256, 1, 1, act=1, res=True)),
          ("stream/256_64", conv_nhwc_case(4, 56, 56, 256, 64, 1, 1, act=1)),
          ("stream/128_512_res", conv_nhwc_case(16, 28, 28, 128, 512, 1, 1, act=1, res=True, seed=2)),
          ("stream/256_1024_oddM", conv_nhwc_case(47, 14, 14, 256, 1024, 1, 1, act=1, res=True, seed=3)),
          ("stream/64_64_noscale", conv_nhwc_case(3, 56, 56, 64, 64, 1, 1, scale=False)),
          ("stream/linear_f32out_res", linear_case(9000, 128, 200, res=True, out="fp32")),
          ("stream/gelu_N72", linear_case(8200, 64, 72, act=2)),
          ("stream/swin_qkv_96_288", linear_case(12544, 96, 288, seed=61)),
          ("stream/swin_fc1_96_384_gelu", linear_case(12544, 96, 384, act=2, seed=62)),
          ("stream/swin_proj_96_f32res", linear_case(9000, 96, 96, res=True, out="fp32", seed=63)),
          ("stream/swin_192_576", linear_case(8300, 192, 576, seed=64)),
          ("linear/vit_qkv", linear_case(197 * 2, 768, 2304)),
          ("linear/vit_fc1_gelu", linear_case(197 * 2, 768, 3072, act=2)),
          ("linear/vit_fc2_res", linear_case(197 * 2, 3072, 768, res=True)),
          ("linear/fc_f32out", linear_case(256, 2048, 1000, out="fp32")),
          ("linear/alex_fc", linear_case(4, 9216, 4096, act=1)),
          ("linear/odd_generic", linear_case(10, 20, 5, act=2)),
          ("linear/f32_generic", linear_case(33, 70, 18, dtype="fp32", res=True)),
          ("linear/generic_vs_oracle_768", linear_case(70, 768, 96, generic=True)),
          ("linear/skinny_fc_128", linear_case(128, 2048, 1000, out="fp32", seed=3)),
          ("linear/skinny_head_200x768", linear_case(200, 768, 1000, out="fp32", seed=4)),
          ("linear/skinny_odd_N_bf16", linear_case(33, 64, 42, act=1, seed=5)),
          ("linear/skinny_1_row", linear_case(1, 4096, 1000, act=2, seed=6)),
          ("linear/skinny_K_not_mult_of_D", linear_case(17, 144, 64, seed=7)),
          ("linear/skinny_one_step_per_wave", linear_case(33, 64, 40, act=1, seed=8)),
          ("linear/skinny_tail_3_steps", linear_case(70, 192, 104, out="fp32", seed=9))]
    c += [("conv_generic/groups", conv_nhwc_case(2, 9, 9, 32, 64, 3, 3, pad=1, groups=4, act=1)),
          ("conv_generic/f32", conv_nhwc_case(1, 10, 10, 12, 20, 3, 3, stride=2, pad=1, dtype="fp32", res=True)),
          ("conv_generic/same_as_igemm", conv_nhwc_case(1, 14, 14, 64, 64, 3, 3, pad=1, generic=True))]
    # ---- network-entry conv from NCHW images
    c += [("stem/resnet7x7", conv_nchw_case(2, 3, 64, 64, 64, 7, 7, 2, 3, act=1)),
          ("stem/resnet7x7_bf16in", conv_nchw_case(2, 3, 64, 64, 64, 7, 7, 2, 3, act=1, xdtype="bf16")),
          ("stem/alexnet11x11", conv_nchw_case(2, 3, 67, 67, 64, 11, 11, 4, 2, act=1)),
          ("stem/vit_patch16_tokens", conv_nchw_case(2, 3, 64, 64, 768, 16, 16, 16, 0, tokens=True)),
          ("stem/swin_patch4", conv_nchw_case(2, 3, 56, 56, 96, 4, 4, 4, 0)),
          ("stem/generic_tokens", conv_nchw_case(1, 3, 32, 32, 64, 8, 8, 8, 0, tokens=True, generic=True)),
          ("stem/full224", conv_nchw_case(2, 3, 224, 224, 64, 7, 7, 2, 3, act=1, seed=5)),
          ("stem/resnet7x7_tablekernel", conv_nchw_case(2, 3, 64, 64, 64, 7, 7, 2, 3, act=1, v0=True)),
          ("stem/vit_tokens_tablekernel", conv_nchw_case(1, 3, 64, 64, 768, 16, 16, 16, 0, tokens=True, v0=True)),
          ("stem/alexnet_224", conv_nchw_case(1, 3, 224, 224, 64, 11, 11, 4, 2, act=1, seed=2)),
          ("stem/pool_fused_224", stem_pool_case(3, 224, 224, seed=6)),
          ("stem/pool_fused_64_bf16in", stem_pool_case(2, 64, 64, xdtype="bf16", seed=7)),
          ("stem/pool_fused_odd_75x93", stem_pool_case(2, 75, 93, seed=8)),
          ("stem/pool_fused_negative_gamma", stem_pool_case(2, 96, 96, seed=9, neg_scale=True)),
          ("stem/pool_fused_many_tiles", stem_pool_case(40, 128, 128, seed=10)),
          ("stem/pool_fused_odd_bf16in_61x70", stem_pool_case(3, 61, 70, xdtype="bf16", seed=11)),
          ("stem/pool_fused_w228_h30", stem_pool_case(2, 30, 228, seed=12)),
          ("fc_stream/alexnet_fc1_M128", fc_stream_case(128, 9216, 4096, act=1, seed=1)),
          ("fc_stream/alexnet_fc2_M128", fc_stream_case(128, 4096, 4096, act=1, seed=2)),
          ("fc_stream/alexnet_fc3_M128_f32out_N1000", fc_stream_case(128, 4096, 1000, out="fp32", seed=3)),
          ("fc_stream/M3_small", fc_stream_case(3, 1024, 4096, act=1, seed=4)),
          ("fc_stream/M200_two_row_blocks_N512", fc_stream_case(200, 1152, 2048, seed=5)),
          ("fc_stream/M130_N260_K640_nobias", fc_stream_case(130, 8192, 260, bias=False, seed=6)),
          ("fc_stream/M256_vgg_fc1_K25088", fc_stream_case(256, 25088, 4096, act=1, seed=7)),
          ("stem/pool11_fused_224", stem_pool_case(3, 224, 224, seed=13, alexnet=True)),
          ("stem/pool11_fused_67_bf16in", stem_pool_case(2, 67, 67, xdtype="bf16", seed=14, alexnet=True)),
          ("stem/pool11_fused_odd_131x95", stem_pool_case(2, 131, 95, seed=15, alexnet=True)),
          ("stem/pool11_fused_many_tiles", stem_pool_case(20, 160, 160, seed=16, alexnet=True)),
          ("stem/pool11_fused_w228_h43", stem_pool_case(2, 43, 228, seed=17, alexnet=True)),
          ("stem/vit_224_bf16in", conv_nchw_case(2, 3, 224, 224, 768, 16, 16, 16, 0, tokens=True, xdtype="bf16")),
          ("stem/vit_patch16_tokens_f32out", conv_nchw_case(2, 3, 64, 64, 768, 16, 16, 16, 0, tokens=True, f32out=True)),
          ("stem/vit_224_f32out", conv_nchw_case(3, 3, 224, 224, 768, 16, 16, 16, 0, tokens=True, f32out=True, seed=3)),
          ("stem/vit_224_bf16in_f32out_no_tokens", conv_nchw_case(2, 3, 224, 224, 384, 16, 16, 16, 0, xdtype="bf16", f32out=True, seed=4)),
          ("stem/patch8_notokens", conv_nchw_case(3, 3, 40, 48, 192, 8, 8, 8, 0)),
          ("stem/odd_size_7x7", conv_nchw_case(1, 3, 61, 75, 32, 7, 7, 2, 3, act=1))]
    c += [("maxpool/3_2_1", maxpool_case(2, 112, 112, 64, 3, 2, 1)),
          ("maxpool/3_2_0", maxpool_case(2, 55, 55, 64, 3, 2, 0)),
          ("maxpool/f32_oddC", maxpool_case(1, 13, 13, 5, 3, 2, 0, dtype="fp32")),
          ("avgpool/global", avgpool_case(2, 7, 7, 2048, 1, 1)),
          ("avgpool/13to6", avgpool_case(1, 13, 13, 16, 6, 6)),
          ("avgpool/global_wide_se_112x112x32", avgpool_case(3, 112, 112, 32, 1, 1, seed=580)),
          ("avgpool/global_wide_28x28x96", avgpool_case(4, 28, 28, 96, 1, 1, seed=581)),
          ("avgpool/global_wide_16x16x520", avgpool_case(2, 16, 16, 520, 1, 1, seed=582)),
          ("avgpool/global_swin_7x7x768", avgpool_case(5, 7, 7, 768, 1, 1, seed=2)),
          ("avgpool/global_hw_not_mult4_c8", avgpool_case(3, 5, 3, 8, 1, 1, seed=3)),
          ("avgpool/global_odd_c_scalar_path", avgpool_case(3, 5, 3, 12, 1, 1, seed=4)),
          ("layernorm/768", layernorm_case(394, 768)),
          ("layernorm/96", layernorm_case(100, 96)),
          ("layernorm/768_generic", layernorm_case(50, 768, generic=True)),
          ("layernorm/strided", layernorm_case(8, 768, stride=768 * 5)),
          ("layernorm/odd_f32", layernorm_case(7, 20, dtype="fp32")),
          ("layernorm/768_f32", layernorm_case(64, 768, dtype="fp32")),
          ("layernorm/1536", layernorm_case(50, 1536)),
          ("layernorm/3072_f32", layernorm_case(9, 3072, dtype="fp32")),
          ("layernorm/96_f32_to_bf16_long", layernorm_case(40001, 96, dtype="fp32", out="bf16")),
          ("layernorm/96_bf16_long", layernorm_case(70001, 96)),
          ("layernorm/192_bf16_long", layernorm_case(33333, 192)),
          ("layernorm/384_f32_to_bf16", layernorm_case(9001, 384, dtype="fp32", out="bf16")),
          ("layernorm/768_f32_to_bf16_long", layernorm_case(20003, 768, dtype="fp32", out="bf16")),
          ("layernorm/768_bf16_to_f32", layernorm_case(515, 768, out="fp32")),
          ("layernorm/1_row", layernorm_case(1, 96)),
          ("layernorm/1536_f32_to_bf16", layernorm_case(6272, 1536, dtype="fp32", out="bf16")),
          ("layernorm/2048_f32", layernorm_case(33, 2048, dtype="fp32")),
          ("layernorm/4096_bf16", layernorm_case(17, 4096))]
    c += [("mha/vit_197_64", mha_case(2, 197, 12, 64)),
          ("mha/vit_197_64_noprobs", mha_case(1, 197, 3, 64, probs=False)),
          ("mha/spike", mha_case(1, 197, 2, 64, spike=True)),
          ("mha/17_32", mha_case(2, 17, 4, 32)),
          ("mha/256_32", mha_case(1, 256, 2, 32)),
          ("mha/generic_8_8", mha_case(1, 8, 4, 8)),
          ("mha/generic_f32", mha_case(1, 50, 2, 16, dtype="fp32")),
          ("mha/generic_vs_oracle_197", mha_case(1, 197, 2, 64, generic=True)),
          ("mha/qkv_heads_vit_base", qkv_heads_case(32, 197, 12, 64)),
          ("mha/qkv_heads_probs_small", qkv_heads_case(24, 197, 4, 64, probs=True, seed=1)),
          ("mha/qkv_heads_50tok", qkv_heads_case(100, 50, 4, 64, seed=2)),
          ("mha/many_pairs_persistent", mha_case(70, 197, 12, 64, probs=False, seed=3)),
          ("mha/many_pairs_225", mha_case(40, 225, 8, 32, probs=True, seed=4)),
          ("mha/one_pair", mha_case(1, 33, 1, 64, seed=5)),
          ("mha/dropout_vit_197_64", mha_dropout_case(2, 197, 12, 64)),
          ("mha/dropout_heads_197_64", mha_dropout_case(3, 197, 4, 64, p=0.5, head_major=True, seed=1)),
          ("mha/dropout_17_32", mha_dropout_case(4, 17, 2, 32, p=0.1, seed=2)),
          ("mha/dropout_256_32_noprobs", mha_dropout_case(1, 256, 2, 32, probs=False, seed=3)),
          ("mha/dropout_odd_count", mha_dropout_case(1, 33, 1, 64, p=0.3, seed=4)),
          ("dropout/windows_shifted", dropout_windows_case(2, 14, 14, 64, (7, 7), (3, 3))),
          ("dropout/windows_unshifted_f32", dropout_windows_case(2, 14, 21, 32, (7, 7), (0, 0), dtype="fp32", seed=1)),
          ("dropout/windows_one_window", dropout_windows_case(3, 7, 7, 128, (7, 7), (3, 3), seed=2)),
          ("dropout/windows_rect", dropout_windows_case(1, 8, 12, 40, (4, 6), (1, 2), p=0.5, seed=3)),
          ("prng/split_tokens", prng_split_case(37, 197, 0)),
          ("prng/split_pairs_child_major", prng_split_case(5000, 2, 1, seed=1)),
          ("prng/split_one", prng_split_case(3, 1, 0, seed=2)),
          ("prng/split_five_child_major", prng_split_case(64, 5, 1, seed=3))]
    c += [("swin/shift3", swin_attn_case(2, 14, 96, 3, 7, 3)),
          ("swin/noshift", swin_attn_case(2, 14, 96, 3, 7, 0)),
          ("swin/window_ge_map", swin_attn_case(1, 7, 192, 6, 7, 3)),
          ("swin/f32", swin_attn_case(1, 14, 32, 2, 7, 3, dtype="fp32")),
          ("swin/generic_shift3", swin_attn_case(2, 14, 96, 3, 7, 3, generic=True)),
          ("swin/stage0_56_shift", swin_attn_case(3, 56, 96, 3, 7, 3, seed=7)),
          ("swin/stage2_14_12heads", swin_attn_case(5, 14, 384, 12, 7, 3, seed=8)),
          ("swin/stage3_7_24heads", swin_attn_case(9, 7, 768, 24, 7, 3, seed=9)),
          ("swin/window8_64tokens", swin_attn_case(2, 16, 64, 2, 8, 4, seed=10)),
          ("swin/window4", swin_attn_case(2, 12, 64, 2, 4, 2, seed=11))]
    c += [("misc/patch_merge", misc_case("patch_merge")),
          ("misc/patch_merge_f32", misc_case("patch_merge", dtype="fp32")),
          ("misc/patch_merge_odd_c", misc_case("patch_merge_odd_c")),
          ("misc/layout", misc_case("layout")),
          ("misc/layout_f32", misc_case("layout", "fp32")),
          ("misc/eltwise", misc_case("eltwise")),
          ("misc/eltwise_f32", misc_case("eltwise", "fp32")),
          ("misc/affine_cls", misc_case("affine_cls"))]
    c += fuzz_cases(24, seed=3) + fuzz_cases(16, seed=4, mfma_only=True)      # default dispatch, random shapes
    c += fuzz_misc_cases(24, seed=5)
    c += fuzz_family_cases(42, seed=6)
    c += fuzz_stochastic_cases(18, seed=7)
    c += fuzz_ln_fold_cases(6, seed=8)
    return c
