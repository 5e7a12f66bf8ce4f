"""CPU error attribution for the un-fused Swin path (test infrastructure, not a test): the torch restatement of swin_b with bf16
rounding injected where the HIP path rounds, one class of sites at a time.  Answers "which rounding puts swin_b's logits at
1.15e-2" before a GPU minute is spent (round 6, review item 2b).

    python tests/attrib_swin_bf16.py [swin_b|swin_s|swin_t]
"""
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import state as S            # noqa: E402
from oracle import torch_ref as TR       # noqa: E402


def bf(t):
    return t.to(torch.bfloat16).to(torch.float32)


def forward(sd, x, depths, heads, rw, ra, window=(7, 7)):
    """rw: round the block Linears' weights; ra: round their input / output activations (LN out, qkv, attention out, GELU out).
    Stream-producing layers (patch embedding, merging reductions, head) stay fp32: split-precision / fp32 on the HIP path."""
    W = (lambda t: bf(t)) if rw else (lambda t: t)
    A = (lambda t: bf(t)) if ra else (lambda t: t)
    sd = TR._t(sd)
    x = torch.as_tensor(x)
    x = F.conv2d(x, sd["features.0.0.weight"], sd["features.0.0.bias"], (4, 4)).permute(0, 2, 3, 1)
    x = F.layer_norm(x, (x.shape[-1],), sd["features.0.2.weight"], sd["features.0.2.bias"], 1e-5)
    fi = 1
    for si, depth in enumerate(depths):
        for bi in range(depth):
            p = f"features.{fi}.{bi}"
            C = x.shape[-1]
            shift = [0 if bi % 2 == 0 else w // 2 for w in window]
            y = A(F.layer_norm(x, (C,), sd[p + ".norm1.weight"], sd[p + ".norm1.bias"], 1e-5))
            sd2 = dict(sd)
            sd2[p + ".attn.qkv.weight"] = W(sd[p + ".attn.qkv.weight"])
            sd2[p + ".attn.proj.weight"] = W(sd[p + ".attn.proj.weight"])
            if ra:      # qkv output and the attention output are bf16 tensors on the HIP path
                sd3 = dict(sd2)
                sd3[p + ".attn.proj.weight"] = torch.eye(C)
                sd3[p + ".attn.proj.bias"] = torch.zeros(C)
                a = A(TR._swin_attn(sd3, y, p, heads[si], list(window), shift))        # (qkv rounding itself not modelled)
                a = F.linear(a, sd2[p + ".attn.proj.weight"], sd[p + ".attn.proj.bias"])
            else:
                a = TR._swin_attn(sd2, y, p, heads[si], list(window), shift)
            x = x + a
            y = A(F.layer_norm(x, (C,), sd[p + ".norm2.weight"], sd[p + ".norm2.bias"], 1e-5))
            y = A(F.gelu(F.linear(y, W(sd[p + ".mlp.0.weight"]), sd[p + ".mlp.0.bias"]), approximate="tanh"))
            x = x + F.linear(y, W(sd[p + ".mlp.3.weight"]), sd[p + ".mlp.3.bias"])
        fi += 1
        if si < len(depths) - 1:
            p = f"features.{fi}"
            x = torch.cat([x[:, 0::2, 0::2], x[:, 1::2, 0::2], x[:, 0::2, 1::2], x[:, 1::2, 1::2]], -1)
            x = F.layer_norm(x, (x.shape[-1],), sd[p + ".norm.weight"], sd[p + ".norm.bias"], 1e-5)
            x = F.linear(x, sd[p + ".reduction.weight"])
            fi += 1
    x = F.layer_norm(x, (x.shape[-1],), sd["norm.weight"], sd["norm.bias"], 1e-5)
    return F.linear(x.mean((1, 2)), sd["head.weight"], sd["head.bias"])


if __name__ == "__main__":
    name = sys.argv[1] if len(sys.argv) > 1 else "swin_b"
    embed, depths, heads = {"swin_t": (96, (2, 2, 6, 2), (3, 6, 12, 24)), "swin_s": (96, (2, 2, 18, 2), (3, 6, 12, 24)),
                            "swin_b": (128, (2, 2, 18, 2), (4, 8, 16, 32))}[name]
    sd = S.swin_state(1, (4, 4), embed, depths, heads)
    x = S.synthetic_images(2, 224, seed=0)
    with torch.no_grad():
        ref = forward(sd, x, depths, heads, False, False)
        print(f"{name}: max|logit| {ref.abs().max():.3f}")
        for rw, ra in ((True, True), (True, False), (False, True)):
            got = forward(sd, x, depths, heads, rw, ra)
            print(f"  weights {'bf16' if rw else 'fp32'}  activations {'bf16' if ra else 'fp32'}:  max abs err {(got - ref).abs().max():.3e}")
