"""The reference's OWN known-answer tests, armed wherever the torchvision checkpoints exist.

`tests/golden/reference_static/` holds the reference's fixtures (`/root/reference/tests/static/*.pred.pth` converted
to .npy by `tests/golden/make_reference_static.py`, and its `img.png`): torchvision outputs for that image with the
pretrained alexnet / resnet18 / swin_t checkpoints, which the reference compares against with atol 1e-4
(tests/test_models/test_alexnet.py:12-25, test_resnet.py:10-24, test_swin.py:34-40).  The checkpoints cannot be
downloaded here (no network), so the comparisons SKIP unless the `.pth` files are found in `$EQXVISION_WEIGHTS`
or `~/.eqxvision/models`; the preprocessing and the fixture plumbing are tested unconditionally.
"""
import os

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
STATIC = os.path.join(HERE, "golden", "reference_static")
CKPT = {"alexnet": "alexnet-owt-7be5be79.pth", "resnet18": "resnet18-f37072fd.pth", "swin_t": "swin_t-704ceda3.pth",
        # SURVEY section 8 f1 (reference eqxvision/utils.py:36-95)
        "vgg11": "vgg11-8a719046.pth", "vgg11_bn": "vgg11_bn-6002323d.pth", "mobilenet_v2": "mobilenet_v2-b0353104.pth",
        "mobilenet_v3_small": "mobilenet_v3_small-047dcff4.pth", "efficientnet_b0": "efficientnet_b0_rwightman-3dd342df.pth",
        "efficientnet_v2_s": "efficientnet_v2_s-dd5fe13b.pth", "regnet_x_400mf": "regnet_x_400mf-62229a5f.pth"}
# what the reference's own test asserts for each family: (fixture, the sub-module it runs, "close" = atol 1e-4 / "argmax")
F1 = {"vgg11": ("vgg11_features", "features", "close"), "vgg11_bn": ("vgg11_bn_features", "features", "close"),        # test_vgg.py:31-32
      "mobilenet_v2": ("mobilenet_v2_logits", None, "argmax"),                                                            # test_mobilenetv2.py:24
      "mobilenet_v3_small": ("mobilenet_v3_small_logits", None, "close"),                                                 # test_mobilenetv3.py:26
      "efficientnet_b0": ("efficientnet_b0_logits", None, "argmax"), "efficientnet_v2_s": ("efficientnet_v2_s_logits", None, "argmax"),
      "regnet_x_400mf": ("regnet_x_400mf_logits", None, "close")}                                                          # test_regnet.py:26


def demo_image(size=224):
    """reference tests/conftest.py:21-41: PIL RGB -> Resize(size) (shorter side, bilinear) -> ToTensor -> Normalize."""
    from PIL import Image
    img = Image.open(os.path.join(STATIC, "img.png")).convert("RGB")
    w, h = img.size
    if w <= h:
        nw, nh = size, int(size * h / w)
    else:
        nh, nw = size, int(size * w / h)
    img = img.resize((nw, nh), Image.BILINEAR)
    x = np.asarray(img, np.float32).transpose(2, 0, 1) / 255.0
    mean = np.array((0.485, 0.456, 0.406), np.float32)[:, None, None]
    std = np.array((0.229, 0.224, 0.225), np.float32)[:, None, None]
    return ((x - mean) / std)[None]


def _ckpt(name):
    for d in (os.environ.get("EQXVISION_WEIGHTS"), os.path.expanduser("~/.eqxvision/models")):
        if d and os.path.exists(os.path.join(d, CKPT[name])):
            return os.path.join(d, CKPT[name])
    pytest.skip(f"{CKPT[name]} not present (no network in this environment): reference known-answer test armed, not run")


def test_fixtures_and_preprocessing():
    x = demo_image(224)
    assert x.shape == (1, 3, 224, 224) and x.dtype == np.float32 and np.isfinite(x).all()
    assert -2.2 < x.min() < -1.0 and 1.5 < x.max() < 2.7          # normalised image statistics
    a = np.load(os.path.join(STATIC, "alexnet_features.npy"))
    r = np.load(os.path.join(STATIC, "resnet18_logits.npy"))
    s = np.load(os.path.join(STATIC, "swin_t_logits.npy"))
    assert a.shape == (1, 256, 6, 6) and r.shape == (1, 1000) and s.shape == (1, 1000)
    assert a.min() >= 0.0                                         # features end in ReLU + MaxPool
    assert int(r.argmax()) in np.argsort(-s[0])[:5] and int(s.argmax()) in np.argsort(-r[0])[:5]   # same bird, top-5
    top5 = set(np.argsort(-r[0])[:5].tolist())
    for name, (fix, sub, _) in F1.items():                        # the section-8 f1 fixtures: present, shaped, the same birds
        a = np.load(os.path.join(STATIC, fix + ".npy"))
        assert a.shape == ((1, 512, 7, 7) if sub == "features" else (1, 1000)) and np.isfinite(a).all(), name
        if sub is None:
            assert len(top5 & set(np.argsort(-a[0])[:5].tolist())) >= 2, name


def _oracle_state(path):
    import torch
    sd = torch.load(path, map_location="cpu")
    return {k: v.detach().numpy() for k, v in sd.items() if hasattr(v, "detach")}


def test_oracle_alexnet_features_vs_reference_golden():
    from oracle import torch_ref as TR
    sd = _oracle_state(_ckpt("alexnet"))
    got = TR.alexnet_features(sd, demo_image(224)).numpy()
    assert np.allclose(got, np.load(os.path.join(STATIC, "alexnet_features.npy")), atol=1e-4)


def test_oracle_resnet18_vs_reference_golden():
    from oracle import torch_ref as TR
    sd = _oracle_state(_ckpt("resnet18"))
    got = TR.resnet_forward(sd, demo_image(224), block="basic", layers=(2, 2, 2, 2)).numpy()
    assert np.allclose(got, np.load(os.path.join(STATIC, "resnet18_logits.npy")), atol=1e-4)


def test_oracle_swin_t_argmax_vs_reference_golden():
    from oracle import torch_ref as TR
    sd = _oracle_state(_ckpt("swin_t"))
    got = TR.swin_forward(sd, demo_image(224)).numpy()
    assert int(got.argmax()) == int(np.load(os.path.join(STATIC, "swin_t_logits.npy")).argmax())


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["alexnet", "resnet18", "swin_t"])
def test_hip_path_vs_reference_golden(name):
    """The HIP path in fp32 mode against the reference's goldens (same tolerance as the reference's tests)."""
    import torch
    import eqxvision_amd as eqv
    path = _ckpt(name)
    eqv.set_compute_dtype("fp32")
    try:
        net = eqv.tree_inference(getattr(eqv.models, name)(torch_weights=path), True)
        x = torch.from_numpy(demo_image(224)).cuda()
        keys = eqv.random.split(eqv.random.PRNGKey(0), 1)
        target = net.features if name == "alexnet" else net
        out = eqv.vmap(target, axis_name="batch")(x, key=keys)
        got = out.float().cpu().numpy()
    finally:
        eqv.set_compute_dtype("bf16")
    if name == "alexnet":
        assert np.allclose(got, np.load(os.path.join(STATIC, "alexnet_features.npy")), atol=1e-3)
    elif name == "resnet18":
        assert np.allclose(got, np.load(os.path.join(STATIC, "resnet18_logits.npy")), atol=1e-3)
    else:
        assert int(got.argmax()) == int(np.load(os.path.join(STATIC, "swin_t_logits.npy")).argmax())



@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(F1))
def test_hip_path_f1_families_vs_reference_golden(name):
    """SURVEY section 8 f1 families on the HIP path (fp32 mode) against the reference's goldens, each with the assertion the
    reference's own test makes (tests/test_models/test_{vgg,mobilenetv2,mobilenetv3,efficientnet,regnet}.py)."""
    import torch
    import eqxvision_amd as eqv
    path = _ckpt(name)
    fix, sub, how = F1[name]
    eqv.set_compute_dtype("fp32")
    try:
        net = eqv.tree_inference(getattr(eqv.models, name)(torch_weights=path), True)
        x = torch.from_numpy(demo_image(224)).cuda()
        keys = eqv.random.split(eqv.random.PRNGKey(0), 1)
        got = eqv.vmap(getattr(net, sub) if sub else net, axis_name="batch")(x, key=keys).float().cpu().numpy()
    finally:
        eqv.set_compute_dtype("bf16")
    want = np.load(os.path.join(STATIC, fix + ".npy"))
    if how == "close":
        assert np.allclose(got, want, atol=1e-3)
    else:
        assert int(got.argmax()) == int(want.argmax())
