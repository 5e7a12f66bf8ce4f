"""CPU: host-side mirror of the reference's module surface -- pytree order (the load_torch_weights
contract), constructors, mode switches and error behaviour.  No GPU needed."""
import os
import tempfile
import warnings
from functools import partial

import numpy as np
import pytest

import eqxvision_amd as eqv
from eqxvision_amd import nn
from oracle import state as S


def _keys(sd):
    return [k for k in sd.keys() if "num_batches" not in k]


@pytest.mark.parametrize("factory,state", [
    (lambda: eqv.models.alexnet(), lambda: S.alexnet_state(1)),
    (lambda: eqv.models.resnet50(), lambda: S.resnet_state(1)),
    (lambda: eqv.models.resnet18(), lambda: S.resnet_state(1, "basic", (2, 2, 2, 2))),
    (lambda: eqv.models.vit_base(num_classes=1000), lambda: S.vit_state(1)),
    (lambda: eqv.models.vit_small(), lambda: S.vit_state(1, embed_dim=384, num_heads=6, num_classes=0)),
    (lambda: eqv.models.mobilenet_v2(), lambda: S.mobilenet_v2_state(1)),
    (lambda: eqv.models.regnet_y_400mf(), lambda: S.regnet_state(1)),
    (lambda: eqv.models.regnet_x_400mf(), lambda: S.regnet_state(1, (32, 64, 160, 400), (1, 2, 7, 12), (16,) * 4, None)),
    (lambda: eqv.models.efficientnet_b0(), lambda: S.efficientnet_state(1)),
    (lambda: eqv.models.efficientnet_v2_s(), lambda: S.efficientnet_state(1, *S.efficientnet_stages("v2_s")[:2])),
    (lambda: eqv.models.mobilenet_v3_large(), lambda: S.mobilenet_v3_state(1)),
    (lambda: eqv.models.mobilenet_v3_small(), lambda: S.mobilenet_v3_state(1, *S.mobilenet_v3_conf("small"))),
    (lambda: eqv.models.resnext50_32x4d(), lambda: S.resnet_state(1, groups=32, width_per_group=4)),
    (lambda: eqv.models.vgg11(), lambda: S.vgg_state(1, "A", False)),
    (lambda: eqv.models.vgg16_bn(num_classes=10), lambda: S.vgg_state(1, "D", True, 10)),
])
def test_pytree_order_is_torchvision_registration_order(factory, state):
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        model = factory()
    sd = state()
    # a model exported with utils.state_dict lists its leaves in flatten order: must equal the checkpoint order
    model = eqv.utils.randomize_batchnorm(model)
    mine = eqv.utils.state_dict(model)
    # the contract is the ORDER (ordered zip, utils.py:120-219); names coincide with torchvision's except where the reference's
    # Sequential differs from torchvision's (VGG classifier: Linear at 0 / 2 / 5 instead of 0 / 3 / 6, vgg.py:96-105)
    strip = lambda k: k if not k.startswith("classifier.") or not isinstance(model, eqv.models.VGG) else "classifier." + k.split(".")[-1]
    assert [strip(k) for k in mine.keys()] == [strip(k) for k in _keys(sd)]
    for k, k_sd in zip(mine, _keys(sd)):
        assert mine[k].size == np.asarray(sd[k_sd]).size, k


def test_load_torch_weights_roundtrip_and_bn_state():
    sd = S.resnet_state(3, "bottleneck", (1, 1, 1, 1), 7)
    blk = eqv.models.classification.resnet._ResNetBottleneck
    with tempfile.TemporaryDirectory() as d:
        p = os.path.join(d, "w.pth")
        S.save_pth(sd, p)
        net = eqv.models.classification.resnet._resnet(blk, [1, 1, 1, 1], p, num_classes=7)
    out = eqv.utils.state_dict(net)
    for k in _keys(sd):
        np.testing.assert_array_equal(out[k].reshape(-1), np.asarray(sd[k]).reshape(-1))
    assert net.conv1.bias is None and net.bn1.first_time_index.value is False
    assert net.layer1.layers[0].downsample.layers[1].state_index.value[0].shape == (256,)
    with pytest.raises(ValueError):
        eqv.utils.load_torch_weights(net, None)
    # too few tensors in the checkpoint -> loud error, not silent truncation
    short = dict(list(sd.items())[:5])
    with tempfile.TemporaryDirectory() as d:
        p = os.path.join(d, "s.pth")
        S.save_pth(short, p)
        with pytest.raises(ValueError):
            eqv.utils.load_torch_weights(eqv.models.resnet18(), p)


def test_swin_checkpoint_overwrites_relative_position_index():
    sd = S.swin_state(1, (4, 4), 32, (2, 2), (2, 4), (7, 7), 4.0, 10)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        net = eqv.models.SwinTransformer(patch_size=[4, 4], embed_dim=32, depths=[2, 2], num_heads=[2, 4],
                                         window_size=[7, 7], num_classes=10)
    attn = net.features.layers[1].layers[0].attn
    # reference quirk (swin.py:314-335): random-init index is relative_coords.sum(-1) in [-12, 12]
    assert attn.relative_position_index.min() == -12 and attn.relative_position_index.max() == 12
    assert np.all(attn.relative_position_bias_table == 2.0)          # truncated_normal(2, 2) quirk
    with tempfile.TemporaryDirectory() as d:
        p = os.path.join(d, "w.pth")
        S.save_pth(sd, p)
        net2 = eqv.utils.load_torch_weights(net, p)
    idx = net2.features.layers[1].layers[0].attn.relative_position_index
    assert idx.min() == 0 and idx.max() == 168
    b = net2.features.layers[1].layers[0].attn.get_relative_position_bias()
    assert b.shape == (2, 49, 49)
    red = net2.features.layers[2]
    assert type(red).__name__ == "_PatchMerging" and red.reduction.weight.shape == (64, 128) and red.reduction.bias is None


def test_tree_inference_flips_every_inference_field_and_copies():
    v = eqv.models.vit_tiny(num_classes=10, drop_path_rate=0.1)
    assert v.inference is False and v.blocks[3].drop_path.inference is False
    vi = eqv.tree_inference(v, True)
    assert vi.inference and vi.blocks[3].drop_path.inference and vi.blocks[0].attn.attn_drop.inference
    assert v.inference is False                                        # original untouched (immutable pytrees)
    assert vi.blocks[0].attn.qkv.weight is v.blocks[0].attn.qkv.weight  # leaves shared
    r = eqv.tree_inference(eqv.models.resnet18(), True)
    assert r.bn1.inference and r.layer2.layers[0].downsample.layers[1].inference
    assert isinstance(v.blocks[0].drop_path, nn.Identity) and isinstance(v.blocks[5].drop_path.p, float)


def test_reference_error_behaviour_without_gpu():
    x = np.zeros((3, 64, 64), np.float32)
    with pytest.raises(RuntimeError, match="PRNGKey"):
        eqv.models.resnet18()(x, key=None)                             # resnet.py:341-342
    with pytest.raises(RuntimeError, match="PRNGKey"):
        eqv.models.alexnet()(x, key=None)                              # alexnet.py:78-79
    with pytest.raises(ValueError, match="doesn't match model"):
        eqv.layers.PatchEmbed()(x)                                     # patch_embed.py:74-77
    with pytest.raises(NotImplementedError):
        eqv.models.ResNet(eqv.models.classification.resnet._ResNetBottleneck, [1, 1, 1, 1], norm_layer=nn.LayerNorm)
    with pytest.raises(ValueError):
        eqv.models.ResNet(eqv.models.classification.resnet._ResNetBottleneck, [1, 1, 1, 1],
                          replace_stride_with_dilation=[True])
    with pytest.raises(ValueError):
        eqv.models.classification.resnet._ResNetBasicBlock(64, 64, groups=2, key=eqv.random.PRNGKey(0))
    with pytest.raises(RuntimeError, match="DropPath requires a key"):
        eqv.layers.DropPath(0.5)(x, key=None)
    with pytest.raises(ValueError):
        eqv.models.classification.swin._ShiftedWindowAttention(32, [7], [0, 0], 2)


def test_vgg_structure_and_reference_quirks():
    """reference vgg.py:96-148: features / avgpool / classifier, ONE relu in the classifier, key is mandatory."""
    net = eqv.models.vgg13_bn(num_classes=7)
    kinds = [type(l).__name__ for l in net.features.layers]
    assert kinds[:7] == ["Conv2d", "BatchNorm", "Lambda", "Conv2d", "BatchNorm", "Lambda", "MaxPool2d"]
    assert kinds.count("Conv2d") == 10 and kinds.count("MaxPool2d") == 5 and kinds.count("BatchNorm") == 10
    assert net.avgpool.target_shape == (7, 7)
    head = [type(l).__name__ for l in net.classifier.layers]
    assert head == ["Linear", "Dropout", "Linear", "Lambda", "Dropout", "Linear"]
    assert net.classifier.layers[0].in_features == 512 * 49 and net.classifier.layers[-1].out_features == 7
    plain = eqv.models.vgg19()
    assert sum(isinstance(l, nn.Conv2d) for l in plain.features.layers) == 16
    assert not any(isinstance(l, nn.BatchNorm) for l in plain.features.layers)
    with pytest.raises(RuntimeError, match="PRNGKey"):
        plain(np.zeros((3, 32, 32), np.float32), key=None)
    with pytest.raises(ValueError):
        eqv.models.VGG()
    assert eqv.utils.CLASSIFICATION_URLS["vgg16_bn"].endswith("vgg16_bn-6c64b313.pth")


def test_segmentation_constructors_and_checkpoint_order():
    """reference fcn.py:86-103 / deeplabv3.py:190-207 argument validation; pytree order == torchvision's state_dict order
    (backbone without fc, classifier, aux_classifier)."""
    from eqxvision_amd.models.classification import resnet as R
    small = lambda: R._resnet(R._ResNetBottleneck, [1, 1, 1, 1], None, replace_stride_with_dilation=[False, True, True])
    two = lambda m: [m.layer3, m.layer4]
    with pytest.raises(ValueError, match="exactly 2 layers"):
        eqv.models.fcn(backbone=small(), intermediate_layers=lambda m: [m.layer4], aux_in_channels=1024)
    with pytest.raises(ValueError, match="expected number of layers is 1"):
        eqv.models.fcn(backbone=small(), intermediate_layers=two)
    with pytest.raises(ValueError, match="exactly 2 layers"):
        eqv.models.deeplabv3(backbone=small(), intermediate_layers=lambda m: [m.layer4])      # aux_in_channels defaults to 1024
    for kind, build in (("fcn", eqv.models.fcn), ("deeplabv3", eqv.models.deeplabv3)):
        net = build(num_classes=5, backbone=small(), intermediate_layers=two, aux_in_channels=1024)
        assert isinstance(net.backbone.model.fc, nn.Identity)                                # silenced head holds no weights
        assert type(net.backbone.model.layer4).__name__ == "_Tap"
        mine = eqv.utils.state_dict(eqv.utils.randomize_batchnorm(net))
        sd = S.segmentation_state(1, kind, (1, 1, 1, 1), 5)
        want = [np.asarray(sd[k]).size for k in _keys(sd)]
        assert [v.size for v in mine.values()] == want
        with pytest.raises(RuntimeError, match="PRNGKey"):
            net(np.zeros((3, 64, 64), np.float32), key=None)
    assert eqv.utils.SEGMENTATION_URLS["fcn_resnet50"].endswith("fcn_resnet50_coco-1167a1af.pth")
    layer4 = eqv.models.deeplabv3(backbone=small(), intermediate_layers=two).backbone.model.layer4.inner
    assert layer4.layers[0].conv2.dilation == (2, 2) and layer4.layers[0].conv2.stride == (1, 1)


def test_mobilenet_v2_structure_and_errors():
    """reference mobilenetv2.py: relu (not relu6) blocks, use_res_connect only for stride 1 and equal widths, argument checks."""
    net = eqv.models.mobilenet_v2(num_classes=11)
    blocks = [m for m in net.features.layers if type(m).__name__ == "_InvertedResidual"]
    assert len(blocks) == 17 and len(net.features.layers) == 19
    assert [b.use_res_connect for b in blocks[:4]] == [False, False, True, False]
    assert len(blocks[0].conv.layers) == 3 and len(blocks[1].conv.layers) == 4                  # expand_ratio 1 has no expansion conv
    dw = blocks[1].conv.layers[1].layers[0]
    assert dw.groups == dw.in_channels == dw.out_channels == 96 and dw.stride == (2, 2)
    assert net.classifier.layers[-1].in_features == 1280 and net.classifier.layers[-1].out_features == 11
    assert eqv.utils._make_divisible(32 * 0.75, 8) == 24 and eqv.utils._make_divisible(10, 8) == 16
    with pytest.raises(ValueError, match="inverted_residual_setting"):
        eqv.models.MobileNetV2(inverted_residual_setting=[[1, 2, 3]])
    with pytest.raises(AssertionError):
        eqv.models.classification.mobilenetv2._InvertedResidual(8, 8, 3, 6, key=eqv.random.PRNGKey(0))
    with pytest.raises(RuntimeError, match="PRNGKey"):
        net(np.zeros((3, 32, 32), np.float32), key=None)


def test_mobilenet_v3_se_and_lraspp_structure():
    """reference mobilenetv3.py / layers/squeeze.py / lraspp.py: tables, SE defaults, dilated tail, taps [4, 16]."""
    from eqxvision_amd.models.classification import mobilenetv3 as M3
    large = eqv.models.mobilenet_v3_large()
    assert len(large.features.layers) == 17 and large.classifier.layers[0].out_features == 1280
    se = large.features.layers[4].block.layers[2]
    assert type(se).__name__ == "SqueezeExcitation" and se.fc1.out_channels == 24 and se.scale_activation.fn is nn.hard_sigmoid
    assert eqv.layers.SqueezeExcitation(16, 8, key=eqv.random.PRNGKey(0)).scale_activation.fn is nn.sigmoid
    bn = large.features.layers[0].layers[1]
    assert abs(bn.eps - 1e-3) < 1e-12                                         # partial(BatchNorm, eps=0.001, momentum=0.01)
    dil = eqv.models.mobilenet_v3_large(dilated=True).features.layers[13].block.layers[1].layers[0]
    assert dil.dilation == (2, 2) and dil.stride == (1, 1) and dil.padding == (4, 4)
    small = eqv.models.mobilenet_v3_small(num_classes=5)
    assert len(small.features.layers) == 13 and small.classifier.layers[-1].out_features == 5
    with pytest.raises(ValueError, match="Unsupported model type"):
        M3._mobilenet_v3_conf("mobilenet_v3_medium")
    with pytest.raises(TypeError):
        M3.MobileNetV3([1, 2], 10)
    with pytest.raises(ValueError, match="should not be empty"):
        M3.MobileNetV3([], 10)
    with pytest.raises(ValueError, match="illegal stride"):
        M3._InvertedResidual(M3._InvertedResidualConfig(16, 3, 16, 16, False, "RE", 3, 1, 1.0), nn.BatchNorm)
    net = eqv.models.lraspp_mobilenet_v3_large(num_classes=None)
    assert net.classifier.low_classifier.in_channels == 40 and net.classifier.cbr.layers[0].in_channels == 960
    assert net.classifier.high_classifier.out_channels == 21
    sd = S.lraspp_state(1)
    mine = eqv.utils.state_dict(eqv.utils.randomize_batchnorm(net))
    assert [v.size for v in mine.values()] == [np.asarray(sd[k]).size for k in _keys(sd)]
    for name, val in ((nn.hard_swish, "hard_swish"), (nn.hard_sigmoid, "hard_sigmoid"), (nn.sigmoid, "sigmoid"), (nn.silu, "silu")):
        assert nn.act_name(name) == val


def test_efficientnet_structure_and_errors():
    """reference efficientnet.py: width / depth scaling, SE squeeze width max(1, in // 4) with SiLU, BatchNorm eps overrides,
    stochastic-depth schedule, argument checks."""
    from eqxvision_amd.models.classification import efficientnet as E
    b0 = eqv.models.efficientnet_b0()
    assert len(b0.features.layers) == 9 and [len(s.layers) for s in b0.features.layers[1:8]] == [1, 2, 2, 3, 3, 4, 1]
    blk = b0.features.layers[2].layers[0]
    se = blk.block.layers[2]
    assert se.fc1.out_channels == 4 and se.activation.fn is nn.silu and se.scale_activation.fn is nn.sigmoid
    assert blk.stochastic_depth.mode == "per_channel" and abs(blk.stochastic_depth.p - 0.2 * 1 / 16) < 1e-12
    assert b0.classifier.layers[-1].in_features == 1280
    b3 = eqv.models.efficientnet_b3()
    assert b3.features.layers[0].layers[0].out_channels == 40 and [len(s.layers) for s in b3.features.layers[1:8]] == [2, 3, 3, 5, 5, 6, 2]
    assert abs(eqv.models.efficientnet_b5().features.layers[0].layers[1].eps - 1e-3) < 1e-12
    v2 = eqv.models.efficientnet_v2_s(num_classes=3)
    fused = v2.features.layers[1].layers[0]
    assert type(fused).__name__ == "_FusedMBConv" and len(fused.block.layers) == 1 and fused.use_res_connect
    assert type(v2.features.layers[4].layers[0]).__name__ == "_MBConv" and v2.classifier.layers[-1].out_features == 3
    with pytest.raises(ValueError, match="Unsupported model type"):
        E._efficientnet_conf("efficientnet_x")
    with pytest.raises(TypeError):
        E.EfficientNet([1], 0.2)
    with pytest.raises(ValueError, match="should not be empty"):
        E.EfficientNet([], 0.2)
    with pytest.raises(ValueError, match="illegal stride"):
        E._MBConv(E._MBConvConfig(1, 3, 3, 16, 16, 1), 0.0, nn.BatchNorm)
    with pytest.raises(RuntimeError, match="PRNGKey"):
        b0(np.zeros((3, 32, 32), np.float32), key=None)


def test_regnet_block_params_and_structure():
    """reference regnet.py:197-262: the quantised-linear width rule reproduces the published stage widths / depths (float32
    arithmetic, like jnp / torch); blocks: projection only when the shape changes, SE width relative to the block input."""
    from eqxvision_amd.models.classification import regnet as R
    published = {"regnet_y_400mf": ([48, 104, 208, 440], [1, 3, 6, 6]), "regnet_x_400mf": ([32, 64, 160, 400], [1, 2, 7, 12]),
                 "regnet_y_8gf": ([224, 448, 896, 2016], [2, 4, 10, 1]), "regnet_x_32gf": ([336, 672, 1344, 2520], [2, 7, 13, 1]),
                 "regnet_y_800mf": ([64, 144, 320, 784], [1, 3, 8, 2]), "regnet_x_8gf": ([80, 240, 720, 1920], [2, 5, 15, 1])}
    for name, (w, d) in published.items():
        bp = R.BlockParams.from_init_params(**R._CONFIGS[name])
        assert (bp.widths, bp.depths) == (w, d), name
    assert R.BlockParams.from_init_params(**R._CONFIGS["regnet_x_8gf"]).group_widths == [80, 120, 120, 120]
    with pytest.raises(ValueError, match="Invalid RegNet settings"):
        R.BlockParams.from_init_params(depth=4, w_0=30, w_a=10.0, w_m=2.0, group_width=8)
    net = eqv.models.regnet_y_400mf(num_classes=9)
    first, second = net.trunk_output.layers[1].layers[:2]
    assert type(first.proj).__name__ == "ConvNormActivation" and isinstance(second.proj, nn.Identity)
    assert first.f.layers[1].layers[0].groups == 13 and first.f.layers[2].fc1.out_channels == 12        # round(0.25 * 48)
    assert net.fc.in_features == 440 and net.fc.out_features == 9
    assert abs(net.stem.layers[1].eps - 1e-5) < 1e-12
    with pytest.raises(RuntimeError, match="PRNGKey"):
        net(np.zeros((3, 32, 32), np.float32), key=None)


def test_conv_norm_activation_structure():
    c = eqv.layers.ConvNormActivation(3, 4, key=eqv.random.PRNGKey(1))
    assert [type(l).__name__ for l in c.layers] == ["Conv2d", "BatchNorm", "Lambda"] and c.out_channels == 4
    assert c.layers[0].padding == (1, 1) and c.layers[0].bias is None and c.layers[1].axis_name == "batch"
    c = eqv.layers.ConvNormActivation(3, 4, kernel_size=5, dilation=2, norm_layer=None, activation_layer=None)
    assert len(c.layers) == 1 and c.layers[0].padding == (4, 4) and c.layers[0].bias.shape == (4, 1, 1)
    c = eqv.layers.ConvNormActivation(3, 4, norm_layer=partial(nn.BatchNorm, eps=1e-3))
    assert c.layers[1].eps == 1e-3 and c.layers[1].axis_name == "batch"
    c = eqv.layers.ConvNormActivation(3, 4, norm_layer=eqv.layers.LayerNorm2d, activation_layer=nn.gelu)
    assert type(c.layers[1]).__name__ == "LayerNorm2d" and c.layers[2].fn is nn.gelu


def test_resnet_variants_shapes_and_dilation():
    r = eqv.models.resnext50_32x4d()
    assert r.layer1.layers[0].conv2.groups == 32 and r.layer1.layers[0].conv2.weight.shape == (128, 4, 3, 3)
    w = eqv.models.wide_resnet50_2()
    assert w.layer1.layers[0].conv1.weight.shape == (128, 64, 1, 1)
    d = eqv.models.resnet50(replace_stride_with_dilation=[False, True, True])
    assert d.layer3.layers[1].conv2.dilation == (2, 2) and d.layer4.layers[1].conv2.dilation == (4, 4)
    assert d.layer4.layers[0].conv2.stride == (1, 1)
    assert eqv.models.vit_b_16 is eqv.models.vit_base


def test_urls_and_keys_mirror_the_reference():
    u = eqv.utils.CLASSIFICATION_URLS
    assert u["resnet50"].endswith("resnet50-19c8e357.pth") and u["alexnet"].endswith("alexnet-owt-7be5be79.pth")
    assert "sim_b" in u and u["vit_small_patch16_224_dino"].endswith("dino_deitsmall16_pretrain.pth")
    k = eqv.random.split(eqv.random.PRNGKey(3), 5)
    assert k.shape == (5, 2) and k.dtype == np.uint32 and len({tuple(r) for r in k}) == 5


def test_no_cpu_fallback_without_device():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from eqxvision_amd._lib import MVError
    with pytest.raises(MVError, match="no CPU fallback"):
        eqv.vmap(eqv.models.resnet18())(np.zeros((1, 3, 64, 64), np.float32), key=eqv.random.split(eqv.random.PRNGKey(0), 1))


def test_pth_reader_matches_torch_load(tmp_path):
    """eqxvision_amd/pth.py (zip + restricted pickle) == torch.load on a state_dict with the dtypes a checkpoint carries."""
    import torch
    from collections import OrderedDict
    from eqxvision_amd.pth import load_state_dict
    g = torch.Generator().manual_seed(0)
    sd = OrderedDict()
    sd["conv.weight"] = torch.randn(8, 3, 3, 3, generator=g)
    sd["bn.running_mean"] = torch.randn(8, generator=g)
    sd["bn.num_batches_tracked"] = torch.tensor(7)
    sd["idx"] = torch.arange(49).reshape(7, 7).t()            # int64, NON-contiguous view: strides must be honoured
    sd["half"] = torch.randn(5, 4, generator=g).half()
    sd["bf"] = torch.randn(6, generator=g).bfloat16()
    sd["slice"] = torch.randn(10, 10, generator=g)[2:5, 1::3]  # storage offset + strides
    p = tmp_path / "w.pth"
    torch.save(sd, str(p))
    got = load_state_dict(str(p))
    assert list(got) == list(sd)
    for k, v in sd.items():
        ref = v.float().numpy() if v.dtype == torch.bfloat16 else v.numpy()
        assert got[k].shape == tuple(v.shape), k
        np.testing.assert_array_equal(got[k], ref, err_msg=k)
    with pytest.raises(Exception):                             # arbitrary globals are refused, not executed
        torch.save({"f": os.system}, str(tmp_path / "evil.pth"))
        load_state_dict(str(tmp_path / "evil.pth"))


def test_load_torch_weights_without_torch_load(tmp_path, monkeypatch):
    import torch
    from oracle import state as S
    sd = S.resnet_state(1, "basic", (1, 1, 1, 1), 10)
    p = tmp_path / "r.pth"
    S.save_pth(sd, str(p))
    monkeypatch.setattr(torch, "load", None)
    blk = eqv.models.classification.resnet._ResNetBasicBlock
    net = eqv.models.classification.resnet._resnet(blk, [1, 1, 1, 1], str(p), num_classes=10)
    np.testing.assert_array_equal(net.conv1.weight, sd["conv1.weight"])
    np.testing.assert_array_equal(net.bn1.state_index.value[1], sd["bn1.running_var"])


def test_intermediate_layer_getter_structure():
    from eqxvision_amd.experimental import intermediate_layer_getter
    seq = nn.Sequential([nn.Conv2d(3, 4, 1), nn.Lambda(nn.relu), nn.Conv2d(4, 2, 1)])
    g = intermediate_layer_getter(seq, lambda m: [1])
    assert type(g.model.layers[1]).__name__ == "_Tap" and g.model.layers[0] is seq.layers[0]
    r = eqv.models.resnet18()
    g = intermediate_layer_getter(r, lambda m: [m.layer2, m.layer4])
    assert type(g.model.layer2).__name__ == "_Tap" and g.model.layer2.inner is r.layer2 and g.model.layer4.slot == 1
    g2 = eqv.tree_inference(g, True)                       # copies of the tree report into the same frame
    assert g2.model.layer2._frame is g.model.layer2._frame and g2._frame is g._frame
    assert g.model.layer1.layers[0].conv1.weight is r.layer1.layers[0].conv1.weight      # leaves are shared


def test_prng_is_jax_threefry():
    """eqxvision_amd.random: Threefry-2x32 known answers (Random123 kat_vectors), and the two values JAX's documentation prints
    for PRNGKey(0): split -> [[4146024105, 967050713], [2718843009, 1272950319]], uniform -> 0.41845703."""
    from eqxvision_amd import random as jr
    for k, c, e in (((0, 0), (0, 0), (0x6B200159, 0x99BA4EFE)),
                    ((0xFFFFFFFF, 0xFFFFFFFF), (0xFFFFFFFF, 0xFFFFFFFF), (0x1CB996FC, 0xBB002BE7)),
                    ((0x13198A2E, 0x03707344), (0x243F6A88, 0x85A308D3), (0xC4923A9C, 0x483DF7A0))):
        o = jr.threefry2x32(k[0], k[1], [c[0]], [c[1]])
        assert (int(o[0][0]), int(o[1][0])) == e
    assert jr.split(jr.PRNGKey(0)).tolist() == [[4146024105, 967050713], [2718843009, 1272950319]]
    assert float(jr.uniform01(jr.PRNGKey(0), 1)[0]) == np.float32(0.41845703)
    keys = jr.split(jr.PRNGKey(3), 6)                                   # batched keys: what vmap over the key axis computes
    kids = jr.split(keys, 4)
    assert kids.shape == (4, 6, 2)
    for b in range(6):
        np.testing.assert_array_equal(kids[:, b], jr.split(keys[b], 4))
        np.testing.assert_array_equal(jr.bernoulli(keys, 0.3, (5, 3))[b], jr.bernoulli(keys[b], 0.3, (5, 3)))
    assert jr.bernoulli(keys, 0.5).shape == (6,)
    assert abs(float(jr.bernoulli(jr.PRNGKey(1), 0.3, (20000,)).mean()) - 0.3) < 0.02
    many = jr.split(jr.PRNGKey(4), 301)                                 # many keys, few words each: the keys-innermost layout
    for n in (1, 2, 3, 4):
        got = jr.random_bits(many, n)
        assert got.shape == (301, n) and got.flags["C_CONTIGUOUS"]
        for b in (0, 150, 300):
            np.testing.assert_array_equal(got[b], jr.random_bits(many[b], n))


def test_stochastic_layers_need_keys_and_inference_is_identity():
    import eqxvision_amd as eqv
    from eqxvision_amd import layers, nn
    x = np.ones((4, 5), np.float32)
    for mod in (nn.Dropout(0.5), layers.DropPath(0.5), layers.DropPath(0.5, mode="local")):
        with pytest.raises(RuntimeError):
            mod(x, key=None)                                           # reference drop_path.py:46-49 / eqx.nn.Dropout
        assert eqv.tree_inference(mod, True)(x, key=None) is x
    assert nn.Dropout(0.0)(x, key=None) is x
    from eqxvision_amd.transforms import _needs_eager
    assert _needs_eager(nn.Sequential([nn.Linear(4, 4, key=eqv.random.PRNGKey(0)), nn.Dropout(0.5)]))
    assert not _needs_eager(eqv.tree_inference(nn.Sequential([nn.Dropout(0.5), nn.BatchNorm(8)]), True))
    assert _needs_eager(nn.Sequential([nn.BatchNorm(8)]))
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        # Swin's `_func_dropout` has no inference switch (swin.py:17-20): a model built with those rates is stochastic in EVERY mode
        kw = dict(patch_size=[4, 4], embed_dim=32, depths=[2], num_heads=[1], window_size=[7, 7], num_classes=3, stochastic_depth_prob=0.0)
        assert _needs_eager(eqv.tree_inference(eqv.models.SwinTransformer(attention_dropout=0.1, **kw), True))
        assert _needs_eager(eqv.tree_inference(eqv.models.SwinTransformer(dropout=0.1, **kw), True))
        assert not _needs_eager(eqv.tree_inference(eqv.models.SwinTransformer(**kw), True))
    vit = eqv.models.VisionTransformer(img_size=16, patch_size=8, embed_dim=32, depth=1, num_heads=1, num_classes=3, attn_drop_rate=0.1)
    assert _needs_eager(vit) and not _needs_eager(eqv.tree_inference(vit, True))



def test_pth_reader_rejects_views_outside_the_storage():
    """a checkpoint's (offset, size, stride) triple is untrusted: views that leave the storage must not reach as_strided"""
    import pickle
    from eqxvision_amd import pth
    st = pth._Storage("<f4", np.arange(12, dtype=np.float32).tobytes())
    ok = pth._rebuild_tensor_v2(st, 2, (2, 3), (3, 1))
    np.testing.assert_array_equal(ok, np.arange(2, 8, dtype=np.float32).reshape(2, 3))
    assert pth._rebuild_tensor_v2(st, 0, (0, 4), (4, 1)).shape == (0, 4)
    for off, size, stride in ((0, (4, 4), (4, 1)), (11, (2,), (1,)), (-1, (2,), (1,)), (0, (2, 2), (-1, 1)), (0, (3,), (1000,))):
        with pytest.raises(pickle.UnpicklingError):
            pth._rebuild_tensor_v2(st, off, size, stride)


def test_roctx_ranges_bracket_module_calls():
    """SURVEY section 5, tracing row: with tracing on, every Module.__call__ opens and closes a roctx range (nested per sub-module)."""
    from eqxvision_amd import _trace

    class Inner(eqv.Module):
        k: int

        def __init__(self):
            self.k = 1

        def __call__(self, x, *, key=None):
            return x + _trace.depth

    class Outer(eqv.Module):
        inner: Inner

        def __init__(self):
            self.inner = Inner()

        def __call__(self, x, *, key=None):
            return self.inner(x) * 10 + _trace.depth

    m = Outer()
    assert m(0) == 0                              # off: no ranges
    old, before = _trace.enabled, _trace.pushed
    _trace.enabled = True
    try:
        try:
            _trace._load()
        except RuntimeError:
            pytest.skip("libroctx64.so not loadable here")
        assert m(0) == 21                         # inner saw depth 2, outer depth 1
        assert _trace.pushed == before + 2 and _trace.depth == 0
    finally:
        _trace.enabled = old


def test_scratch_hand_over_travels_with_the_recorded_launch():
    """`mv_set_scratch` is not a launch: while a forward is recorded it is folded into the launch it precedes, and every replay hands the
    scratch over again on the stream it is replayed on (eqxvision_amd/_lib.py: call, _with_scratch)."""
    from eqxvision_amd import _lib
    rec = []
    old = _lib.set_recording(rec)
    try:
        _lib.call("mv_set_scratch", 0x1000, 8192, 7)
        assert rec == []                                              # nothing recorded yet
        _lib.call("mv_splitk_scratch_bytes", 1, 1, 1)                 # any recorded call that needs no GPU (returns 0 = MV_OK)
        _lib.call("mv_splitk_scratch_bytes", 1, 1, 1)                 # the hand-over belongs to ONE launch
    finally:
        _lib.set_recording(old)
    assert [r[2] for r in rec] == ["mv_splitk_scratch_bytes", "mv_splitk_scratch_bytes"]
    assert rec[0][0].__name__ == "launch" and rec[1][0].__name__ != "launch"
    seen = []
    wrapped = _lib._with_scratch(lambda p, n, s: seen.append(("set", p, n, s)), (0x1000, 8192),
                                 lambda *a: seen.append(("launch",) + a) or 0)
    assert wrapped(1, 2, 99) == 0
    assert seen == [("set", 0x1000, 8192, 99), ("launch", 1, 2, 99)]   # handed over first, on the replay's stream
    # the rule behind mv_splitk_scratch_bytes: few tiles AND a long reduction
    lib = _lib.load()
    assert lib.mv_splitk_scratch_bytes(3136, 512, 4608) == 4096 + 52 * 128 * 256 * 4      # ResNet layer 4 3x3 at 64 images: 50 / 52 tiles
    assert lib.mv_splitk_scratch_bytes(3136, 768, 768) == 0                                  # 12 k-tiles: not worth the hand-over
    assert lib.mv_splitk_scratch_bytes(100352, 512, 4608) == 0                               # fills the chip un-split


def test_devarray_leaf_behaves_like_an_array_leaf():
    """A `DevArray` (gradients, updates, the parameters after apply_updates: state that stays on the device) is an array leaf to every
    tree utility, converts through `np.asarray` ONCE (and read-only, like the reference's immutable arrays), and a model that carries
    such leaves still yields a numpy state_dict and a stable jit signature."""
    import torch
    import eqxvision_amd as eqv
    from eqxvision_amd._module import DevArray, tree_map
    from eqxvision_amd.transforms import _module_sig
    t = torch.arange(6, dtype=torch.float32).reshape(2, 3)
    d = DevArray(t)
    assert eqv.is_array(d) and d.shape == (2, 3) and d.ndim == 2 and d.size == 6 and d.dtype == np.float32 and len(d) == 2
    a = np.asarray(d)
    assert a is np.asarray(d) and not a.flags.writeable and np.array_equal(a, t.numpy())
    assert np.array_equal(np.asarray(d, np.float64), t.numpy().astype(np.float64))
    assert np.array_equal(d + 1, t.numpy() + 1) and np.array_equal(2 * d, 2 * t.numpy()) and np.array_equal(d.reshape(-1), t.numpy().reshape(-1))
    net = eqv.models.alexnet(num_classes=3, key=eqv.random.PRNGKey(0))
    dev = tree_map(lambda l: DevArray(torch.from_numpy(np.ascontiguousarray(l))) if isinstance(l, np.ndarray) and l.dtype.kind == "f" else l, net)
    leaves = [l for l in eqv.tree_leaves(dev) if eqv.is_array(l)]
    assert leaves and all(isinstance(l, DevArray) for l in leaves if l.dtype.kind == "f")
    sd, sd0 = eqv.utils.state_dict(dev), eqv.utils.state_dict(net)
    assert list(sd) == list(sd0) and all(isinstance(v, np.ndarray) and np.array_equal(v, sd0[k]) for k, v in sd.items())
    assert _module_sig(dev) == _module_sig(dev) and _module_sig(dev) != _module_sig(net)
    # numpy 2 protocol and the host-side arithmetic a training loop written against jax arrays does on its leaves (advisor, round 5):
    # a requested copy is the caller's own writable memory, copy=False refuses a conversion, ufuncs / reductions / array functions work
    c = np.array(d, copy=True)
    assert c is not np.asarray(d) and c.flags.writeable and np.array_equal(c, a)
    c *= 2.0
    assert np.array_equal(np.asarray(d), t.numpy())                       # the cached host copy is untouched
    assert np.array(d) is not np.asarray(d)
    with pytest.raises(ValueError):
        np.asarray(d, dtype=np.float64, copy=False)
    assert np.asarray(d, copy=False) is np.asarray(d)
    assert np.array_equal(d ** 2, t.numpy() ** 2) and np.array_equal(abs(-d), t.numpy()) and d.T.shape == (3, 2)
    assert float(d.sum()) == 15.0 and float(d.mean()) == 2.5 and float((d * d).sum()) == 55.0 and int((d > 2).sum()) == 3
    assert np.isclose(np.linalg.norm(d), np.sqrt(55.0)) and np.array_equal(np.sqrt(d), np.sqrt(t.numpy()))
    assert np.concatenate([d, d]).shape == (4, 3) and float(np.clip(d, 0, 1).max()) == 1.0 and hash(d) == hash(d)
    gnorm = np.sqrt(sum(float(np.sum(np.square(l))) for l in leaves if l.dtype.kind == "f"))        # global-norm clipping on device leaves
    assert np.isfinite(gnorm) and gnorm > 0


def test_filter_jit_structure_signature_ignores_leaf_identity():
    """The eager-fast-path key of `filter_jit` (transforms._struct_sig): equal for a model and a copy with NEW leaf objects of the same
    shapes (what apply_updates returns every step), different when a shape or a static field changes."""
    import eqxvision_amd as eqv
    from eqxvision_amd._module import tree_map
    from eqxvision_amd.transforms import _module_sig, _struct_sig
    net = eqv.models.alexnet(num_classes=3, key=eqv.random.PRNGKey(0))
    new = tree_map(lambda l: l.copy() if isinstance(l, np.ndarray) else l, net)
    assert _module_sig(new) != _module_sig(net) and _struct_sig(new) == _struct_sig(net)
    other = eqv.models.alexnet(num_classes=4, key=eqv.random.PRNGKey(0))
    assert _struct_sig(other) != _struct_sig(net)
    assert _struct_sig(eqv.tree_inference(net, True)) != _struct_sig(eqv.tree_inference(net, False))


def test_shift_rows_of_the_recompute_chain_kernels():
    """ops._rc_shift_rows (the host side of mv_conv1x1_chain_rc*_fwd / chain_res_fwd's `shifts` operand) against the layout the header
    documents, written out independently in tests/_cases.py: 64 words per 32 values, word r < 32 = bf16 hi | bf16 lo << 16 with
    hi + lo equal to the fp32 value to 16 mantissa bits, words 32 .. 63 zero."""
    from eqxvision_amd.ops import _rc_shift_rows
    from tests._cases import _rc_shifts
    rng = np.random.Generator(np.random.PCG64(3))
    a, b = rng.standard_normal(256).astype(np.float32), (10.0 * rng.standard_normal(64)).astype(np.float32)
    got, ref = _rc_shift_rows(a, b), _rc_shifts(a, b)
    assert got.shape == (10, 64) and got.dtype == np.uint32 and np.array_equal(got, ref)
    assert not got[:, 32:].any()
    hi = ((got[:, :32] & 0xffff) << 16).astype(np.uint32).view(np.float32).reshape(-1)
    lo = (got[:, :32] & 0xffff0000).astype(np.uint32).view(np.float32).reshape(-1)
    v = np.concatenate([a, b])
    assert np.abs(hi + lo - v).max() <= 2.0 ** -16 * np.abs(v).max()


def _launch_list(monkeypatch, factory, sd_fn, B, size=224, flags=()):
    """The C-ABI entries a forward would call, in order, WITHOUT a GPU: every launch is replaced by a recorder and every device
    allocation by a CPU tensor (the host-side dispatch -- which kernel serves which layer -- is pure Python + the library's
    `_supported` entries, which need no device)."""
    import torch
    import eqxvision_amd as eqv
    from eqxvision_amd import _act, _lib, ops
    names = []

    def fake_call(name, *a):
        names.append(name)
        return 0
    monkeypatch.setattr(_lib, "call", fake_call)
    monkeypatch.setattr(_act, "device", lambda: torch.device("cpu"))
    monkeypatch.setattr(ops, "device", lambda: torch.device("cpu"))
    monkeypatch.setattr(ops, "empty", lambda shape, dtype: torch.zeros(shape, dtype=dtype))
    monkeypatch.setattr(_act, "empty", lambda shape, dtype: torch.zeros(shape, dtype=dtype))
    monkeypatch.setattr(ops, "stream_ptr", lambda: 0)
    monkeypatch.setattr(ops, "_dev", lambda a, dt: torch.from_numpy(np.ascontiguousarray(a)).to(dt))
    monkeypatch.setattr(ops, "_splitk_scratch", lambda *a, **k: None)
    for f in flags:                                      # (the real entry: _lib.call is the recorder now)
        _lib.load().mv_set_flag(f.encode(), 1)
    try:
        with tempfile.TemporaryDirectory() as td:
            p = os.path.join(td, "w.pth")
            S.save_pth(sd_fn(), p)
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                net = eqv.tree_inference(factory(torch_weights=p), True)
        with eqv.precision("bf16"):
            eqv.vmap(net, axis_name="batch")(torch.zeros(B, 3, size, size), key=eqv.random.split(eqv.random.PRNGKey(0), B))
    finally:
        for f in flags:
            _lib.load().mv_set_flag(f.encode(), 0)
    return [n for n in names if n != "mv_set_scratch"]


def test_resnet50_layer1_plan_launch_list(monkeypatch, built_lib):
    """Round 6's layer-1 plan, host side: at B >= 3 (>= 8192 pixels at 56 x 56) the three block boundaries of ResNet-50's first stage
    are mv_conv1x1_chain_rc0_fwd (block 0's output never written) -> mv_conv1x1_chain_rc_fwd (it is recomputed) ->
    mv_conv1x1_chain_res_fwd; below that, and with the plan switched off, the round-5 entries; the launch COUNT is the same either way."""
    import eqxvision_amd as eqv
    new = ("mv_conv1x1_chain_rc0_fwd", "mv_conv1x1_chain_rc_fwd", "mv_conv1x1_chain_res_fwd")
    on = _launch_list(monkeypatch, eqv.models.resnet50, lambda: S.resnet_state(1), 4)
    assert [n for n in on if n in new] == list(new)
    assert on.index("mv_conv1x1_chain_rc0_fwd") < on.index("mv_conv1x1_chain_rc_fwd") < on.index("mv_conv1x1_chain_res_fwd")
    assert "mv_conv1x1_dual_chain_fwd" not in on and "mv_conv1x1_chain_fwd" not in on          # (layer 2 chains from 16 384 pixels up: not at B = 4)
    off = _launch_list(monkeypatch, eqv.models.resnet50, lambda: S.resnet_state(1), 4, flags=("no_chain_rc", "no_chain_res", "no_chain_sub"))
    assert not any(n in off for n in new) and "mv_conv1x1_dual_chain_fwd" in off and off.count("mv_conv1x1_chain_fwd") == 2
    assert len(on) == len(off)
    small = _launch_list(monkeypatch, eqv.models.resnet50, lambda: S.resnet_state(1), 2)           # 6272 pixels: below every chain kernel
    assert not any(n in small for n in new) and "mv_conv1x1_chain_fwd" not in small


def test_layer1_plan_only_where_the_stage_matches(monkeypatch, built_lib):
    """A stage that is not three 64 -> 256 bottlenecks never sees the plan: basic blocks (resnet34), width 128 (wide_resnet50_2)."""
    import eqxvision_amd as eqv
    new = ("mv_conv1x1_chain_rc0_fwd", "mv_conv1x1_chain_rc_fwd", "mv_conv1x1_chain_res_fwd")
    r34 = _launch_list(monkeypatch, eqv.models.resnet34, lambda: S.resnet_state(1, "basic", (3, 4, 6, 3), 1000), 4)
    wide = _launch_list(monkeypatch, eqv.models.wide_resnet50_2, lambda: S.resnet_state(1, "bottleneck", (3, 4, 6, 3), 1000, width_per_group=128), 4)
    assert not any(n in r34 for n in new) and not any(n in wide for n in new)
    r101 = _launch_list(monkeypatch, eqv.models.resnet101, lambda: S.resnet_state(1, "bottleneck", (3, 4, 23, 3), 1000), 4)
    assert [n for n in r101 if n in new] == list(new)


def test_vit_layernorm_fold_launch_list(monkeypatch, built_lib):
    """Round 6's ViT block without LayerNorm launches, host side: where the 256 x 256 GEMM tile is the dispatch's choice for the block's
    Linears (64 images of 197 tokens and up) proj / fc2 are mv_linear_lnout_fwd (the residual stream as two bf16 planes) and fc1 / every
    qkv but the first are mv_linear_lnin_fwd; the LayerNorms left are the first block's norm1 and the head's.  Below that size, and with
    the switch, the LayerNorm launches."""
    import eqxvision_amd as eqv
    depth = 3
    fac = lambda torch_weights=None: eqv.utils.load_torch_weights(
        eqv.models.VisionTransformer(img_size=224, patch_size=16, embed_dim=768, depth=depth, num_heads=12, num_classes=10), torch_weights)
    sd = lambda: S.vit_state(1, 224, 16, 768, depth, 12, 4, 10)
    on = _launch_list(monkeypatch, fac, sd, 64)
    assert on.count("mv_linear_lnout_fwd") == 2 * depth and on.count("mv_linear_lnin_fwd") == 2 * depth - 1
    assert on.count("mv_layernorm_fwd") == 2 and on.count("mv_linear_heads_fwd") == 1 and on.count("mv_mha_heads_fwd") == depth   # norm1 of block 0, the head's
    first_out = on.index("mv_linear_lnout_fwd")
    assert on.index("mv_layernorm_fwd") < on.index("mv_linear_heads_fwd") < first_out < on.index("mv_linear_lnin_fwd")
    off = _launch_list(monkeypatch, fac, sd, 64, flags=("no_ln_fold",))
    assert "mv_linear_lnout_fwd" not in off and "mv_linear_lnin_fwd" not in off
    assert off.count("mv_layernorm_fwd") == 2 * depth + 1 and off.count("mv_linear_heads_fwd") == depth
    assert len(off) == len(on) + 2 * depth - 1
    small = _launch_list(monkeypatch, fac, sd, 8)
    assert "mv_linear_lnout_fwd" not in small and small.count("mv_layernorm_fwd") == 2 * depth + 1
