"""YOLOv3-tiny head on a MobileNetV2 encoder — host mirror of reference
models/yolov3_tiny_mobilenet.py.

The reference takes the encoder from torchvision (``mobilenet_v2(pretrained=True).features``,
pinned torchvision==0.3.0, requirements.txt:8), which is not installed here and whose
pretrained weights need a download.  The encoder below restates the published MobileNetV2
architecture (Sandler et al. 2018; torchvision 0.3.0 layer order and ``state_dict`` key names:
``features.sequenceN.<i>.conv.<j>...``) with random initialisation; load real weights with
``load_state_dict``.  The split at feature index 14 and the head follow the reference
(yolov3_tiny_mobilenet.py:18-34,58-69).
"""
from __future__ import annotations

import torch
from torch import nn

from .. import engine
from ..utils.torch_utils import fold_conv_bn
from .yolo_base import ConvBlock, YOLOBase
from .yolo_layer import Concat, Upsample
from .yolov3_tiny import trace_tiny_heads

# (expand t, channels c, repeats n, stride s) — MobileNetV2 table 2
_MBV2_SETTING = ((1, 16, 1, 1), (6, 24, 2, 2), (6, 32, 3, 2), (6, 64, 4, 2), (6, 96, 3, 1), (6, 160, 3, 2), (6, 320, 1, 1))


def _folded(conv: nn.Conv2d, bn: nn.BatchNorm2d):
    return fold_conv_bn(conv.weight, conv.bias, bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.eps)


class ConvBNReLU(nn.Sequential):
    """conv(k, stride, groups) + BN + ReLU6; children '0','1','2' like torchvision 0.3.0."""

    def __init__(self, in_planes, out_planes, kernel_size=3, stride=1, groups=1):
        super().__init__(nn.Conv2d(in_planes, out_planes, kernel_size, stride, (kernel_size - 1) // 2,
                                   groups=groups, bias=False),
                         nn.BatchNorm2d(out_planes), nn.ReLU6(inplace=True))
        self.stride, self.depthwise = stride, groups > 1

    def _trace(self, g, x):
        w = _folded(self[0], self[1])
        if self.depthwise:
            return g.dwconv(x, w, stride=self.stride, act="relu6")
        return g.conv(x, w, stride=self.stride, act="relu6")


class InvertedResidual(nn.Module):
    def __init__(self, inp, oup, stride, expand_ratio):
        super().__init__()
        hidden = int(round(inp * expand_ratio))
        self.use_res_connect = stride == 1 and inp == oup
        layers = []
        if expand_ratio != 1:
            layers.append(ConvBNReLU(inp, hidden, kernel_size=1))
        layers += [ConvBNReLU(hidden, hidden, stride=stride, groups=hidden),
                   nn.Conv2d(hidden, oup, 1, 1, 0, bias=False), nn.BatchNorm2d(oup)]
        self.conv = nn.Sequential(*layers)

    def _trace(self, g, x):
        y = x
        for m in list(self.conv)[:-2]:
            y = m._trace(g, y)
        proj = _folded(self.conv[-2], self.conv[-1])                       # linear bottleneck: no activation
        return g.conv(y, proj, stride=1, act="none", residual=x if self.use_res_connect else None)


class MobileNetEncoder(nn.Module):
    """features[:14] -> 96ch @ /16 (route 1), features[14:] -> 1280ch @ /32 (route 2)."""

    route_index = 14

    def __init__(self, in_channels=3):
        super().__init__()
        feats = [ConvBNReLU(in_channels, 32, stride=2)]
        c_in = 32
        for t, c, n, s in _MBV2_SETTING:
            for i in range(n):
                feats.append(InvertedResidual(c_in, c, s if i == 0 else 1, t))
                c_in = c
        feats.append(ConvBNReLU(c_in, 1280, kernel_size=1))
        self.sequence1 = nn.Sequential(*feats[:self.route_index])
        self.sequence2 = nn.Sequential(*feats[self.route_index:])

    @property
    def out_channels(self):
        return 96, 1280

    def _trace(self, g, x):
        for m in self.sequence1:
            x = m._trace(g, x)
        route1 = x
        for m in self.sequence2:
            x = m._trace(g, x)
        return route1, x


class YOLOv3TinyMobile(YOLOBase):
    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        kd = self.kernels_divider
        wd = lambda c: max(8, c // kd)
        out = self.yolo_layer_input_size
        self.features = MobileNetEncoder(in_channels=self.in_channels)
        f1, f2 = self.features.out_channels

        b11 = nn.Sequential()
        b11.add_module("branch1_conv1", ConvBlock(f2, wd(128), size=1))
        b11.add_module("branch1_upsample", Upsample(2))
        self.sequence_branch1_1 = b11

        b12 = nn.Sequential()
        b12.add_module("branch1_concat", Concat(1))
        b12.add_module("branch1_conv2", ConvBlock(f1 + wd(128), wd(64)))
        b12.add_module("branch1_conv3", nn.Conv2d(wd(64), out, kernel_size=1))
        self.sequence_branch1_2 = b12

        b2 = nn.Sequential()
        b2.add_module("branch2_conv1", ConvBlock(f2, wd(64)))
        b2.add_module("branch2_conv2", nn.Conv2d(wd(64), out, kernel_size=1))
        self.sequence_branch2 = b2

        self.yolo1, self.yolo2 = self._create_yolo_layers()

    @property
    def yolo_layers(self):
        return self.yolo1, self.yolo2

    def _trace(self, g: engine.Recorder, x):
        route1, route2 = self.features._trace(g, x)
        trace_tiny_heads(self, g, route1, route2)
