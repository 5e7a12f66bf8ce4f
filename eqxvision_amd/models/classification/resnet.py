"""ResNet family (reference models/classification/resnet.py:37-511).

Field order mirrors the reference (= torchvision registration order, the `load_torch_weights`
contract).  Every residual block lowers to fused launches only:
  conv+BN+relu, conv+BN+relu, [downsample conv+BN], conv+BN + identity + relu
with the BatchNorm folded into the fp32 epilogue of the implicit-GEMM kernel."""
from __future__ import annotations

from typing import Any, Callable, List, Optional, Sequence, Type, Union

from ... import nn, ops
from ... import random as jr
from ..._act import head_fp32
from ..._module import Module
from ...nn import boundary
from ...utils import load_torch_weights


def _conv3x3(in_planes, out_planes, stride=1, groups=1, dilation=1, key=None):
    return nn.Conv2d(in_planes, out_planes, kernel_size=3, stride=stride, padding=dilation, groups=groups,
                     use_bias=False, dilation=dilation, key=key)


def _conv1x1(in_planes, out_planes, stride=1, key=None):
    return nn.Conv2d(in_planes, out_planes, kernel_size=1, stride=stride, use_bias=False, key=key)


def _shortcut(block, x):
    """identity = self.downsample(x): a [conv1x1(stride), BN] Sequential or nn.Identity."""
    return block.downsample(x)


class _ResNetBasicBlock(Module):
    expansion: int
    conv1: Module
    bn1: Module
    relu: Callable
    conv2: Module
    bn2: Module
    downsample: Module
    stride: int

    def __init__(self, inplanes, planes, stride=1, downsample=None, groups=1, base_width=64, dilation=1,
                 norm_layer=None, key=None):
        if norm_layer is None:
            norm_layer = nn.BatchNorm
        if groups != 1 or base_width != 64:
            raise ValueError("BasicBlock only supports groups=1 and base_width=64")
        if dilation > 1:
            raise NotImplementedError("Dilation > 1 not supported in BasicBlock")
        keys = jr.split(key, 2)
        self.expansion = 1
        self.conv1 = _conv3x3(inplanes, planes, stride, key=keys[0])
        self.bn1 = norm_layer(planes, axis_name="batch")
        self.relu = nn.relu
        self.conv2 = _conv3x3(planes, planes, key=keys[1])
        self.bn2 = norm_layer(planes, axis_name="batch")
        self.downsample = downsample if downsample else nn.Identity()
        self.stride = stride

    @boundary
    def __call__(self, x, *, key=None):                       # reference :80-92
        out = ops.conv2d(x, self.conv1, self.bn1, "relu")
        identity = _shortcut(self, x)
        return ops.conv2d(out, self.conv2, self.bn2, "relu", residual=identity)


class _ResNetBottleneck(Module):
    # stride sits on the 3x3 (ResNet v1.5), like torchvision / the reference (:96-100)
    expansion: int
    conv1: Module
    bn1: Module
    conv2: Module
    bn2: Module
    conv3: Module
    bn3: Module
    relu: Callable
    downsample: Module
    stride: int

    def __init__(self, inplanes, planes, stride=1, downsample=None, groups=1, base_width=64, dilation=1,
                 norm_layer=None, key=None):
        if norm_layer is None:
            norm_layer = nn.BatchNorm
        self.expansion = 4
        keys = jr.split(key, 3)
        width = int(planes * (base_width / 64.0)) * groups
        self.conv1 = _conv1x1(inplanes, width, key=keys[0])
        self.bn1 = norm_layer(width, axis_name="batch")
        self.conv2 = _conv3x3(width, width, stride, groups, dilation, key=keys[1])
        self.bn2 = norm_layer(width, axis_name="batch")
        self.conv3 = _conv1x1(width, planes * self.expansion, key=keys[2])
        self.bn3 = norm_layer(planes * self.expansion, axis_name="batch")
        self.relu = nn.relu
        self.downsample = downsample if downsample else nn.Identity()
        self.stride = stride

    @boundary
    def __call__(self, x, *, key=None):                       # reference :144-162
        return self.call_chained(x, None)

    def call_chained(self, x, nxt):
        """The block; `nxt` = the bottleneck that follows in the stage's nn.Sequential (or None).  When the library can,
        this block's last convolution and `nxt`'s first one are ONE launch (ops.conv1x1_chain) and `nxt` finds its
        conv1 output attached to its input."""
        pre = x.pre if ops.is_act(x) else None
        x = ops.as_map(x)
        ds = self.downsample
        conv_ds = isinstance(ds, nn.Sequential) and len(ds) == 2 and isinstance(ds[0], nn.Conv2d) and \
            isinstance(ds[1], nn.BatchNorm)
        if pre is not None and pre[0] is self.conv1:
            out = pre[1]
        else:
            out = ops.conv2d(x, self.conv1, self.bn1, "relu")
        if isinstance(ds, nn.Identity):
            # identity block on a map that fits a CU: conv2 + conv3 + identity in one launch, the `width`-channel intermediate
            # stays in LDS (ops.bottleneck_tail; None when the library has no such path for the shapes)
            y = ops.bottleneck_tail(out, self.conv2, self.bn2, self.conv3, self.bn3, x)
            if y is not None:
                return y
        out = ops.conv2d(out, self.conv2, self.bn2, "relu")
        if conv_ds:
            # identity = BN(conv1x1(x)) (resnet.py:295-303): it and conv3 add into one output -> one GEMM over the
            # concatenated reduction [out | x]; the identity map is never materialised
            if isinstance(nxt, _ResNetBottleneck):
                y = ops.conv1x1_dual_chain(out, self.conv3, self.bn3, x, ds[0], ds[1], nxt.conv1, nxt.bn1)
                if y is not None:
                    return y
            y = ops.conv1x1_dual(out, self.conv3, self.bn3, x, ds[0], ds[1], "relu")
            if y is not None:
                return y
        if x.sub is not None:
            raise RuntimeError("bottleneck: the input map was written sub-sampled for a downsample branch that did not run")
        identity = _shortcut(self, x)
        if isinstance(nxt, _ResNetBottleneck):
            # `nxt` opening the next stage with a stride-2 pointwise downsample branch is -- besides its conv1, which this launch
            # computes -- the only consumer of this block's output: then only the pixels that branch reads are written
            y = ops.conv1x1_chain(out, self.conv3, self.bn3, identity, nxt.conv1, nxt.bn1, sub=2 if _reads_strided_only(nxt, out) else 0)
            if y is not None:
                return y
        return ops.conv2d(out, self.conv3, self.bn3, "relu", residual=identity)


def _conv_downsample(block):
    ds = block.downsample
    if isinstance(ds, nn.Sequential) and len(ds) == 2 and isinstance(ds[0], nn.Conv2d) and isinstance(ds[1], nn.BatchNorm):
        return ds[0], ds[1]
    return None


def _reads_strided_only(nxt, out) -> bool:
    """True when bottleneck `nxt` reads its input map ONLY through a 1x1 stride-2 downsample convolution that ops.conv1x1_dual
    will run (its conv1 is pointwise and fused into the producer; ResNet v1.5 strides the 3x3, resnet.py:119-127, 295-303)."""
    cd = _conv_downsample(nxt)
    if cd is None or not ops._pointwise(nxt.conv1):
        return False
    conv, bn = cd
    if tuple(conv.kernel_size) != (1, 1) or tuple(conv.stride) != (2, 2) or tuple(conv.padding) != (0, 0) \
            or tuple(conv.dilation) != (1, 1) or conv.groups != 1:
        return False
    B, H, W, _ = out.t.shape
    if H % 2 or W % 2 or tuple(nxt.conv2.stride) != (2, 2):
        return False
    return ops.conv1x1_dual_available(B * (H // 2) * (W // 2), nxt.conv3, nxt.bn3, conv, bn)


def _stage_rc(stage, x, nxt):
    """A stage of exactly three bottlenecks (layer1 of ResNet-50 / 101 / 152) with the first block's output never written: block 0
    stores only the next conv1's result, block 1's boundary recomputes y0 from its two 64-channel sources (ops.conv1x1_chain_rc),
    block 2 runs as usual (and writes its output sub-sampled when `nxt` only reads it strided).  None when any piece has no path."""
    L = list(stage.layers) if isinstance(stage, nn.Sequential) else []
    if len(L) != 3 or not all(type(b) is _ResNetBottleneck for b in L) or not ops.is_act(x) or x.pre is not None:
        return None
    b0, b1, b2 = L
    cd = _conv_downsample(b0)
    if cd is None or not isinstance(b1.downsample, nn.Identity) or not isinstance(b2.downsample, nn.Identity):
        return None
    if not ops.chain_rc_available(x, b0, cd, b1, b2):
        return None
    x = ops.as_map(x)
    t1 = ops.conv2d(x, b0.conv1, b0.bn1, "relu")
    t2_0 = ops.conv2d(t1, b0.conv2, b0.bn2, "relu")
    t1 = ops.conv1x1_dual_chain(t2_0, b0.conv3, b0.bn3, x, cd[0], cd[1], b1.conv1, b1.bn1, store_y=False)
    if t1 is None:
        raise RuntimeError("stage_rc: the dual chain refused shapes ops.chain_rc_available accepted")
    t2_1 = ops.conv2d(t1, b1.conv2, b1.bn2, "relu")
    y1 = ops.conv1x1_chain_rc(t2_1, t2_0, x, b0.conv3, b0.bn3, cd[0], cd[1], b1.conv3, b1.bn3, b2.conv1, b2.bn1)
    if y1 is None:
        raise RuntimeError("stage_rc: the recompute chain refused shapes ops.chain_rc_available accepted")
    return b2.call_chained(y1, nxt)


EXPANSIONS = {_ResNetBasicBlock: 1, _ResNetBottleneck: 4}


class ResNet(Module):
    inplanes: int
    dilation: int
    groups: Sequence[int]
    base_width: int
    conv1: Module
    bn1: Module
    relu: Callable
    maxpool: Module
    layer1: Module
    layer2: Module
    layer3: Module
    layer4: Module
    avgpool: Module
    fc: Module

    def __init__(self, block: Type[Union[_ResNetBasicBlock, _ResNetBottleneck]], layers: List[int],
                 num_classes: int = 1000, groups: int = 1, width_per_group: int = 64,
                 replace_stride_with_dilation: List[bool] = None, norm_layer: Any = None, *, key=None):
        if not norm_layer:
            norm_layer = nn.BatchNorm
        if norm_layer is not nn.BatchNorm:                    # reference :222-225
            raise NotImplementedError(f"{type(norm_layer)} is not currently supported. Use `nn.BatchNorm` instead.")
        if key is None:
            key = jr.PRNGKey(0)
        keys = jr.split(key, 6)
        self.inplanes = 64
        self.dilation = 1
        if replace_stride_with_dilation is None:
            replace_stride_with_dilation = [False, False, False]
        if len(replace_stride_with_dilation) != 3:
            raise ValueError("replace_stride_with_dilation should be None or a 3-element tuple, got {}".format(
                replace_stride_with_dilation))
        self.groups = groups
        self.base_width = width_per_group
        self.conv1 = nn.Conv2d(3, self.inplanes, kernel_size=7, stride=2, padding=3, use_bias=False, key=keys[0])
        self.bn1 = norm_layer(input_size=self.inplanes, axis_name="batch")
        self.relu = nn.relu
        self.maxpool = nn.MaxPool2d(kernel_size=3, stride=2, padding=1)
        self.layer1 = self._make_layer(block, 64, layers[0], norm_layer, key=keys[1])
        self.layer2 = self._make_layer(block, 128, layers[1], norm_layer, stride=2,
                                       dilate=replace_stride_with_dilation[0], key=keys[2])
        self.layer3 = self._make_layer(block, 256, layers[2], norm_layer, stride=2,
                                       dilate=replace_stride_with_dilation[1], key=keys[3])
        self.layer4 = self._make_layer(block, 512, layers[3], norm_layer, stride=2,
                                       dilate=replace_stride_with_dilation[2], key=keys[4])
        self.avgpool = nn.AdaptiveAvgPool2d((1, 1))
        self.fc = nn.Linear(512 * EXPANSIONS[block], num_classes, key=keys[5])

    def _make_layer(self, block, planes, blocks, norm_layer, stride=1, dilate=False, key=None):
        keys = jr.split(key, blocks + 1)
        downsample = None
        previous_dilation = self.dilation
        if dilate:
            self.dilation *= stride
            stride = 1
        if stride != 1 or self.inplanes != planes * EXPANSIONS[block]:        # reference :295-303
            downsample = nn.Sequential([
                _conv1x1(self.inplanes, planes * EXPANSIONS[block], stride, key=keys[0]),
                norm_layer(planes * EXPANSIONS[block], axis_name="batch"),
            ])
        stack = [block(self.inplanes, planes, stride, downsample, self.groups, self.base_width, previous_dilation,
                       norm_layer, key=keys[1])]
        self.inplanes = planes * EXPANSIONS[block]
        for i in range(1, blocks):
            stack.append(block(self.inplanes, planes, groups=self.groups, base_width=self.base_width,
                               dilation=self.dilation, norm_layer=norm_layer, key=keys[i + 1]))
        return nn.Sequential(stack)

    def __call__(self, x, *, key):                            # reference :335-358
        if key is None:
            raise RuntimeError("The model requires a PRNGKey.")
        return self._forward(x)

    @boundary
    def _forward(self, x):
        if isinstance(self.maxpool, nn.MaxPool2d):
            x = ops.stem_conv_pool(x, self.conv1, self.bn1, "relu", self.maxpool)   # reference :243-254, one launch
        else:
            x = self.maxpool(ops.conv2d(x, self.conv1, self.bn1, "relu"))
        stages = [self.layer1, self.layer2, self.layer3, self.layer4]
        for i, stage in enumerate(stages):
            # the last block of a stage may fuse its tail with the head of the next stage's first block
            nxt = stages[i + 1][0] if i + 1 < len(stages) and isinstance(stages[i + 1], nn.Sequential) and \
                len(stages[i + 1]) > 0 else None
            y = _stage_rc(stage, x, nxt)
            if y is not None:
                x = y
                continue
            x = stage.call_chained(x, nxt) if isinstance(stage, nn.Sequential) else stage(x)
        if type(self.avgpool) is nn.AdaptiveAvgPool2d and head_fp32():     # reference :354-356, pooled features kept fp32
            x = ops.adaptive_avgpool2d(x, self.avgpool.target_shape, out_fp32=True)
        else:
            x = self.avgpool(x)
        x = ops.flatten(x)
        if not isinstance(self.fc, nn.Linear):                # e.g. silenced with nn.Identity by the segmentation models (fcn.py:106)
            return self.fc(x)
        return ops.linear_head(x, self.fc)


def _resnet(block, layers, torch_weights=None, **kwargs):
    model = ResNet(block, layers, **kwargs)
    if torch_weights:
        model = load_torch_weights(model, torch_weights=torch_weights)
    return model


# name -> (block, blocks per stage, constructor overrides); reference :361-511
_VARIANTS = {
    "resnet18": (_ResNetBasicBlock, (2, 2, 2, 2), {}),
    "resnet34": (_ResNetBasicBlock, (3, 4, 6, 3), {}),
    "resnet50": (_ResNetBottleneck, (3, 4, 6, 3), {}),      # v1.5: 53 convolutions, 4.09 GMAC per 224x224 image
    "resnet101": (_ResNetBottleneck, (3, 4, 23, 3), {}),
    "resnet152": (_ResNetBottleneck, (3, 8, 36, 3), {}),
    "resnext50_32x4d": (_ResNetBottleneck, (3, 4, 6, 3), {"groups": 32, "width_per_group": 4}),
    "resnext101_32x8d": (_ResNetBottleneck, (3, 4, 23, 3), {"groups": 32, "width_per_group": 8}),
    "wide_resnet50_2": (_ResNetBottleneck, (3, 4, 6, 3), {"width_per_group": 128}),
    "wide_resnet101_2": (_ResNetBottleneck, (3, 4, 23, 3), {"width_per_group": 128}),
}


def _variant(name):
    block, depths, fixed = _VARIANTS[name]

    def make(torch_weights=None, **kwargs) -> ResNet:
        kwargs.update(fixed)
        return _resnet(block, list(depths), torch_weights, **kwargs)

    make.__name__ = make.__qualname__ = name
    make.__doc__ = f"{name} (reference models/classification/resnet.py); `torch_weights`: torchvision checkpoint path / URL."
    return make


resnet18, resnet34, resnet50, resnet101, resnet152 = (_variant(n) for n in ("resnet18", "resnet34", "resnet50", "resnet101",
                                                                            "resnet152"))
resnext50_32x4d, resnext101_32x8d = _variant("resnext50_32x4d"), _variant("resnext101_32x8d")
wide_resnet50_2, wide_resnet101_2 = _variant("wide_resnet50_2"), _variant("wide_resnet101_2")
