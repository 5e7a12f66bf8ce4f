"""Flat model namespace (reference eqxvision/models/__init__.py:1-105) for the hot-path families."""
from .classification.alexnet import AlexNet, alexnet
from .classification.efficientnet import (
    EfficientNet,
    efficientnet_b0,
    efficientnet_b1,
    efficientnet_b2,
    efficientnet_b3,
    efficientnet_b4,
    efficientnet_b5,
    efficientnet_b6,
    efficientnet_b7,
    efficientnet_v2_l,
    efficientnet_v2_m,
    efficientnet_v2_s,
)
from .classification.mobilenetv2 import MobileNetV2, mobilenet_v2
from .classification.mobilenetv3 import MobileNetV3, mobilenet_v3_large, mobilenet_v3_small
from .classification.regnet import (
    RegNet,
    regnet_x_1_6gf,
    regnet_x_3_2gf,
    regnet_x_8gf,
    regnet_x_16gf,
    regnet_x_32gf,
    regnet_x_400mf,
    regnet_x_800mf,
    regnet_y_1_6gf,
    regnet_y_3_2gf,
    regnet_y_8gf,
    regnet_y_16gf,
    regnet_y_32gf,
    regnet_y_128gf,
    regnet_y_400mf,
    regnet_y_800mf,
)
from .classification.resnet import (
    ResNet,
    resnet18,
    resnet34,
    resnet50,
    resnet101,
    resnet152,
    resnext50_32x4d,
    resnext101_32x8d,
    wide_resnet50_2,
    wide_resnet101_2,
)
from .segmentation.deeplabv3 import ASPP, DeepLabHead, DeepLabV3, deeplabv3
from .segmentation.fcn import FCN, FCNHead, fcn
from .segmentation.lraspp import LRASPP, LRASPPHead, lraspp_mobilenet_v3_large
from .classification.swin import SwinTransformer, swin_b, swin_s, swin_t
from .classification.vgg import VGG, vgg11, vgg11_bn, vgg13, vgg13_bn, vgg16, vgg16_bn, vgg19, vgg19_bn
from .classification.vit import (
    _VitAttention,
    _VitBlock,
    VisionTransformer,
    vit_b_16,
    vit_base,
    vit_small,
    vit_tiny,
)
