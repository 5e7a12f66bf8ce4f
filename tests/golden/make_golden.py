#!/usr/bin/env python
"""Generates tests/golden/*.npz : inputs (seeds) + expected outputs of the hot path at small sizes.

The reference itself cannot run here (jax / equinox are not installed; its tests/static/*.pred.pth
fixtures need torchvision checkpoints that cannot be downloaded), so the vectors are produced by the
independent torch.nn.functional restatement (`oracle/torch_ref.py` -- torchvision semantics are what
the reference's own tests treat as ground truth) and pin the numpy oracle and, on the GPU box, the HIP
path.  Run from the repo root:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import state as S  # noqa: E402
from oracle import torch_ref as TR  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))


def main():
    g = {}
    x = S.synthetic_images(2, 64, seed=0)
    sd = S.resnet_state(1, "bottleneck", (1, 1, 1, 1), 10)
    g["resnet_bottleneck_1111_64px"] = TR.resnet_forward(sd, x, "bottleneck", (1, 1, 1, 1)).numpy()
    sd = S.resnet_state(1, "basic", (2, 2, 2, 2), 10)
    g["resnet18_64px"] = TR.resnet_forward(sd, x, "basic", (2, 2, 2, 2)).numpy()
    x32 = S.synthetic_images(3, 32, seed=0)
    sd = S.vit_state(1, 32, 8, 64, 2, 2, 4, 10)
    g["vit_32px_p8_d64_h2_depth2"] = TR.vit_forward(sd, x32, 8, 2, 2).numpy()
    g["vit_32px_last_attn"] = TR.vit_last_self_attention(sd, x32[:2], 8, 2, 2).numpy()
    x56 = S.synthetic_images(2, 56, seed=0)
    sd = S.swin_state(1, (4, 4), 32, (2, 2), (2, 4), (7, 7), 4.0, 10)
    g["swin_56px_e32_d22"] = TR.swin_forward(sd, x56, (4, 4), (2, 2), (2, 4), (7, 7)).numpy()
    x224 = S.synthetic_images(2, 224, seed=0)
    sd = S.alexnet_state(1, 1000)
    lg = TR.alexnet_forward(sd, x224).numpy()
    g["alexnet_224_logits_head16"] = lg[:, :16]
    g["alexnet_224_logits_l2"] = np.linalg.norm(lg, axis=1)
    g["alexnet_224_features_l2"] = np.linalg.norm(TR.alexnet_features(sd, x224).numpy().reshape(2, -1), axis=1)
    sd = S.resnet_state(1)
    lg = TR.resnet_forward(sd, x224).numpy()
    g["resnet50_224_logits_head16"] = lg[:, :16]
    g["resnet50_224_logits_l2"] = np.linalg.norm(lg, axis=1)
    # full-size logits checksums (first 16 logits + L2 norm) for the other two hot models at B = 2
    sd = S.vit_state(1)
    lg = TR.vit_forward(sd, x224).numpy()
    g["vit_base_224_logits_head16"] = lg[:, :16]
    g["vit_base_224_logits_l2"] = np.linalg.norm(lg, axis=1)
    sd = S.swin_state(1)
    lg = TR.swin_forward(sd, x224).numpy()
    g["swin_t_224_logits_head16"] = lg[:, :16]
    g["swin_t_224_logits_l2"] = np.linalg.norm(lg, axis=1)
    np.savez_compressed(os.path.join(OUT, "hotpath_small.npz"), **{k: v.astype(np.float32) for k, v in g.items()})
    for k, v in g.items():
        print(k, v.shape, float(np.abs(v).max()))


if __name__ == "__main__":
    main()
