"""YOLOv3-tiny head on a SqueezeNet 1.1 encoder — host mirror of reference models/yolov3_tiny_squeeze.py
(SURVEY.md 8f rank 4).

The reference takes the encoder from torchvision (``squeezenet1_1(True).features``, yolov3_tiny_squeeze.py:3,19-31),
which is not installed here and whose pretrained weights need a download.  The encoder below restates the published
SqueezeNet 1.1 (Iandola et al. 2016; torchvision layer order and ``state_dict`` key names
``features.sequenceN.<i>[.squeeze|.expand1x1|.expand3x3].{weight,bias}``) with random initialisation; load real
weights with ``load_state_dict``.  The split four modules from the end (``route_index = -4``, :22-31) and the head
(:51-65, 74-82: both routes have the SAME grid, so the concat takes no upsample) follow the reference.
"""
from __future__ import annotations

import torch
from torch import nn

from .. import engine
from .yolo_base import ConvBlock, YOLOBase
from .yolo_layer import Concat
from .yolov3_tiny import plain_head


def _wb(conv: nn.Conv2d):
    return conv.weight.detach().float().cpu(), conv.bias.detach().float().cpu()


class _ConvReLU0(nn.Conv2d):
    """features[0]: Conv2d(in, 64, kernel_size=3, stride=2), NO padding; the ReLU that follows is folded in."""

    def _trace(self, g, x):
        return g.conv(x, _wb(self), stride=2, act="relu", pad=0)


class _ReLU(nn.ReLU):
    def _trace(self, g, x):
        return x                                           # folded into the conv in front of it


class _PoolCeil(nn.MaxPool2d):
    """MaxPool2d(kernel_size=3, stride=2, ceil_mode=True)."""

    def _trace(self, g, x):
        return g.maxpool(x, 3, 2, pad=0, ceil_mode=True)


class Fire(nn.Module):
    """squeeze 1x1 + ReLU -> cat(expand 1x1 + ReLU, expand 3x3 (pad 1) + ReLU); child names as in torchvision."""

    def __init__(self, inplanes, squeeze_planes, expand1x1_planes, expand3x3_planes):
        super().__init__()
        self.squeeze = nn.Conv2d(inplanes, squeeze_planes, kernel_size=1)
        self.squeeze_activation = nn.ReLU(inplace=True)
        self.expand1x1 = nn.Conv2d(squeeze_planes, expand1x1_planes, kernel_size=1)
        self.expand1x1_activation = nn.ReLU(inplace=True)
        self.expand3x3 = nn.Conv2d(squeeze_planes, expand3x3_planes, kernel_size=3, padding=1)
        self.expand3x3_activation = nn.ReLU(inplace=True)

    def _trace(self, g, x):
        # the squeeze tensor (16 / 32 / 48 / 64 channels) is kept in a multiple of 32 physical channels (zero weight rows
        # and biases, zero input columns in the expand convs): the expand convs then take the LDS-DMA fast path
        sq, sp = self.squeeze.out_channels, -(-self.squeeze.out_channels // 32) * 32
        w, b = _wb(self.squeeze)
        s = g.conv(x, (torch.nn.functional.pad(w, (0, 0, 0, 0, 0, 0, 0, sp - sq)), torch.nn.functional.pad(b, (0, sp - sq))), act="relu")

        def widen(conv):
            w_, b_ = _wb(conv)
            return torch.nn.functional.pad(w_, (0, 0, 0, 0, 0, sp - sq)), b_
        return g.concat([g.conv(s, widen(self.expand1x1), act="relu"), g.conv(s, widen(self.expand3x3), act="relu")])


def _squeezenet1_1_features(in_channels):
    return [_ConvReLU0(in_channels, 64, kernel_size=3, stride=2), _ReLU(inplace=True), _PoolCeil(3, 2, ceil_mode=True),
            Fire(64, 16, 64, 64), Fire(128, 16, 64, 64), _PoolCeil(3, 2, ceil_mode=True),
            Fire(128, 32, 128, 128), Fire(256, 32, 128, 128), _PoolCeil(3, 2, ceil_mode=True),
            Fire(256, 48, 192, 192), Fire(384, 48, 192, 192), Fire(384, 64, 256, 256), Fire(512, 64, 256, 256)]


class SqueezeEncoder(nn.Module):
    """features[:-4] -> 256 channels, features[-4:] -> 512 channels on the same grid (yolov3_tiny_squeeze.py:15-44)."""

    route_index = -4

    def __init__(self, in_channels=3):
        super().__init__()
        feats = _squeezenet1_1_features(in_channels)
        self.sequence1 = nn.Sequential(*feats[:self.route_index])
        self.sequence2 = nn.Sequential(*feats[self.route_index:])

    @property
    def out_channels(self):
        return 256, 512

    def _trace(self, g, x):
        for m in self.sequence1:
            x = m._trace(g, x)
        b1 = x
        for m in self.sequence2:
            x = m._trace(g, x)
        return b1, x


class YOLOv3TinySqueeze(YOLOBase):
    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        kd = self.kernels_divider
        wd = lambda c: max(8, c // kd)
        out = self.yolo_layer_input_size
        self.features = SqueezeEncoder(in_channels=self.in_channels)
        f1, f2 = self.features.out_channels

        b11 = nn.Sequential()
        b11.add_module("branch1_conv1", ConvBlock(f2, wd(128), size=1))
        self.sequence_branch1_1 = b11

        b12 = nn.Sequential()
        b12.add_module("branch1_concat", Concat(1))
        b12.add_module("branch1_conv2", ConvBlock(f1 + wd(128), wd(128)))
        b12.add_module("branch1_conv3", nn.Conv2d(wd(128), out, kernel_size=1))
        self.sequence_branch1_2 = b12

        b2 = nn.Sequential()
        b2.add_module("branch2_conv1", ConvBlock(f2, wd(128)))
        b2.add_module("branch2_conv2", nn.Conv2d(wd(128), out, kernel_size=1))
        self.sequence_branch2 = b2

        self.yolo1, self.yolo2 = self._create_yolo_layers()

    @property
    def yolo_layers(self):
        return self.yolo1, self.yolo2

    def _trace(self, g: engine.Recorder, x):
        """Reference _forward_encoder + forward (yolov3_tiny_squeeze.py:71-104)."""
        route1, route2 = self.features._trace(g, x)
        b1 = self.sequence_branch1_1.branch1_conv1._trace(g, route2)
        b1 = g.concat([route1, b1])                                        # [x_route1, x_branch1] (:77), same grid
        b1 = self.sequence_branch1_2.branch1_conv2._trace(g, b1)
        g.head(plain_head(g, b1, self.sequence_branch1_2.branch1_conv3), self.yolo1)
        b2 = self.sequence_branch2.branch2_conv1._trace(g, route2)
        g.head(plain_head(g, b2, self.sequence_branch2.branch2_conv2), self.yolo2)
