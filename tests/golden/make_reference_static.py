"""Converts the reference's own known-answer fixtures (data files of its test-suite) into torch-free arrays.

    /root/reference/tests/static/{alexnet,resnet18,swin_t,vgg11,vgg11_bn,mobilenet_v2,mobilenet_v3_small,efficientnet_b0,
                                  efficientnet_v2_s,regnet_x_400mf}.pred.pth  ->  tests/golden/reference_static/*.npy
    /root/reference/tests/static/img.png                             ->  tests/golden/reference_static/img.png

The .pred.pth files hold torchvision's outputs for `img.png` with the pretrained checkpoints
(reference tests/conftest.py:44-100; compared with atol 1e-4 in tests/test_models/test_{alexnet,resnet,swin}.py).
Run in the build container only (the reference tree does not exist on the GPU box):  python tests/golden/make_reference_static.py
"""
import os
import shutil

import numpy as np
import torch

SRC = "/root/reference/tests/static"
DST = os.path.join(os.path.dirname(os.path.abspath(__file__)), "reference_static")
os.makedirs(DST, exist_ok=True)
for name, key, out in (("alexnet", "output", "alexnet_features"), ("resnet18", "output", "resnet18_logits"),
                       ("swin_t", None, "swin_t_logits"),
                       # the families of SURVEY section 8 f1 (reference tests/test_models/test_{vgg,mobilenetv2,mobilenetv3,efficientnet,regnet}.py)
                       ("vgg11", "output", "vgg11_features"), ("vgg11_bn", "output", "vgg11_bn_features"),
                       ("mobilenet_v2", "output", "mobilenet_v2_logits"), ("mobilenet_v3_small", None, "mobilenet_v3_small_logits"),
                       ("efficientnet_b0", None, "efficientnet_b0_logits"), ("efficientnet_v2_s", None, "efficientnet_v2_s_logits"),
                       ("regnet_x_400mf", None, "regnet_x_400mf_logits")):
    t = torch.load(os.path.join(SRC, name + ".pred.pth"), map_location="cpu")
    a = (t[key] if key else t).detach().numpy()
    np.save(os.path.join(DST, out + ".npy"), a)
    print(out, a.shape, a.dtype, float(np.abs(a).max()))
shutil.copyfile(os.path.join(SRC, "img.png"), os.path.join(DST, "img.png"))
