"""YOLOv3-tiny head on a ShuffleNetV2 x1.0 encoder — host mirror of reference models/yolov3_tiny_shuffle.py
(SURVEY.md 8f rank 4).

The reference takes the encoder from torchvision (``shufflenet_v2_x1_0(True)``, yolov3_tiny_shuffle.py:3,13-47), which
is not installed here and whose pretrained weights need a download.  The encoder below restates the published
ShuffleNetV2 x1.0 (Ma et al. 2018; torchvision layer order and ``state_dict`` key names
``features.sequence1.{0,2,3}...`` / ``features.sequence2.{0,1}...``) with random initialisation; load real weights with
``load_state_dict``.  The split (conv1, maxpool, stage2, stage3 | stage4, conv5) and the tiny-style head follow the
reference (:28-35, :58-69).

On the device the stage widths 116 / 232 / 464 split into halves of 58 / 116 / 232 channels, and 58 and 116 are not
multiples of the 8-channel (16-byte) granule of the NHWC views.  The trace therefore runs in a padded physical space:
a unit's output keeps its two halves in two slots of 64 / 128 / 256 physical channels (multiples of 32, which also
keeps every conv on the fast LDS-DMA path); the weights recorded for every
conv are the logical ones scattered into that space (zero rows, columns and biases for the pad channels, which ReLU
and the linear depthwise stage keep at zero), the stride-1 split ``x.chunk(2)`` is two channel views, and
``channel_shuffle(cat(a, b), 2)`` is one copy kernel (``yolo_channel_shuffle2_fwd``).
"""
from __future__ import annotations

import torch
from torch import nn

from .. import engine
from ..kernels import roundup
from ..utils.torch_utils import fold_conv_bn
from .yolo_base import ConvBlock, YOLOBase
from .yolo_layer import Concat, Upsample
from .yolov3_tiny import plain_head

_STAGES = ((4, 116), (8, 232), (4, 464))
_SLOT = 32      # slot granule: 8 channels would do for the views; 32 keeps every 1x1 / 3x3 conv on the LDS-DMA fast path


def _folded(conv: nn.Conv2d, bn: nn.BatchNorm2d):
    return fold_conv_bn(conv.weight, conv.bias, bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.eps)


def _scatter(wb, out_map, out_phys, in_map=None, in_phys=None):
    """Logical folded (weight OIHW, bias) -> the same conv in physical channels: row o goes to out_map[o], input
    channel i (full convs only) to in_map[i]; everything else is zero."""
    w, b = wb
    cout, cin, k, _ = w.shape
    if in_map is None:                                      # depthwise: [c, 1, 3, 3]
        wp = torch.zeros((out_phys, 1, k, k), dtype=torch.float32)
        wp[out_map] = w
    else:
        wp = torch.zeros((out_phys, in_phys, k, k), dtype=torch.float32)
        wp[torch.as_tensor(out_map)[:, None], torch.as_tensor(in_map)[None, :]] = w
    bp = torch.zeros(out_phys, dtype=torch.float32)
    bp[out_map] = b
    return wp, bp


class _ConvBNReLU(nn.Sequential):
    def __init__(self, cin, cout, k, stride):
        super().__init__(nn.Conv2d(cin, cout, k, stride, (k - 1) // 2, bias=False), nn.BatchNorm2d(cout), nn.ReLU(inplace=True))


class _MaxPool321(nn.MaxPool2d):
    pass


class InvertedResidual(nn.Module):
    """torchvision.models.shufflenetv2.InvertedResidual: children ``branch1`` / ``branch2`` with the same indices."""

    def __init__(self, inp, oup, stride):
        super().__init__()
        self.stride, bf = stride, oup // 2
        if stride > 1:
            self.branch1 = nn.Sequential(nn.Conv2d(inp, inp, 3, stride, 1, groups=inp, bias=False), nn.BatchNorm2d(inp),
                                         nn.Conv2d(inp, bf, 1, 1, 0, bias=False), nn.BatchNorm2d(bf), nn.ReLU(inplace=True))
        else:
            self.branch1 = nn.Sequential()
        cin2 = inp if stride > 1 else bf
        self.branch2 = nn.Sequential(nn.Conv2d(cin2, bf, 1, 1, 0, bias=False), nn.BatchNorm2d(bf), nn.ReLU(inplace=True),
                                     nn.Conv2d(bf, bf, 3, stride, 1, groups=bf, bias=False), nn.BatchNorm2d(bf),
                                     nn.Conv2d(bf, bf, 1, 1, 0, bias=False), nn.BatchNorm2d(bf), nn.ReLU(inplace=True))
        self.bf = bf

    def _branch2(self, g, x, in_map, hp):
        h, b2 = self.bf, self.branch2
        rows = list(range(h))
        u = g.conv(x, _scatter(_folded(b2[0], b2[1]), rows, hp, in_map, x.c), act="relu")
        v = g.dwconv(u, _scatter(_folded(b2[3], b2[4]), rows, hp), stride=self.stride, act="none")
        return g.conv(v, _scatter(_folded(b2[5], b2[6]), rows, hp, rows, hp), act="relu")

    def _trace(self, g, x, in_map):
        """x: the physical tensor, in_map[i] = physical channel of logical input channel i.  Returns (y, map of y)."""
        h, hp = self.bf, roundup(self.bf, _SLOT)
        rows = list(range(h))
        if self.stride > 1:
            b1 = self.branch1
            t = g.dwconv(x, _scatter(_folded(b1[0], b1[1]), in_map, x.c), stride=self.stride, act="none")
            a = g.conv(t, _scatter(_folded(b1[2], b1[3]), rows, hp, in_map, x.c), act="relu")
            b = self._branch2(g, x, in_map, hp)
        else:
            a = g.slice(x, 0, hp)                                 # x.chunk(2): the two slots
            b = self._branch2(g, g.slice(x, hp, hp), rows, hp)
        return g.shuffle2(a, b, h), rows + [hp + i for i in rows]


class ShuffleEncoder(nn.Module):
    """sequence1 = (conv1, maxpool, stage2, stage3) -> 232 ch @/16, sequence2 = (stage4, conv5) -> 1024 ch @/32
    (yolov3_tiny_shuffle.py:13-47)."""

    def __init__(self, in_channels=3):
        super().__init__()
        stages, inp = [], 24
        for rep, oup in _STAGES:
            stages.append(nn.Sequential(*[InvertedResidual(inp if u == 0 else oup, oup, 2 if u == 0 else 1) for u in range(rep)]))
            inp = oup
        self.sequence1 = nn.Sequential(_ConvBNReLU(in_channels, 24, 3, 2), _MaxPool321(3, 2, 1), stages[0], stages[1])
        self.sequence2 = nn.Sequential(stages[2], _ConvBNReLU(464, 1024, 1, 1))

    @property
    def out_channels(self):
        return 232, 1024

    def _trace(self, g, x):
        """Returns (route1 physical tensor, its channel map, route2)."""
        c1 = self.sequence1[0]
        cmap = list(range(24))
        x = g.conv(x, _scatter(_folded(c1[0], c1[1]), cmap, roundup(24, _SLOT), list(range(c1[0].in_channels)), c1[0].in_channels),
                   stride=2, act="relu")                       # 24 real channels in a 32-channel tensor
        x = g.maxpool(x, 3, 2)
        for stage in (self.sequence1[2], self.sequence1[3]):
            for unit in stage:
                x, cmap = unit._trace(g, x, cmap)
        route1, map1 = x, cmap
        for unit in self.sequence2[0]:
            x, cmap = unit._trace(g, x, cmap)
        c5 = self.sequence2[1]
        route2 = g.conv(x, _scatter(_folded(c5[0], c5[1]), list(range(1024)), 1024, cmap, x.c), act="relu")
        return route1, map1, route2


class YOLOv3TinyShuffle(YOLOBase):
    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        kd = self.kernels_divider
        wd = lambda c: max(8, c // kd)
        out = self.yolo_layer_input_size
        self.features = ShuffleEncoder(in_channels=self.in_channels)
        f1, f2 = self.features.out_channels

        b11 = nn.Sequential()
        b11.add_module("branch1_conv1", ConvBlock(f2, wd(128), size=1))
        b11.add_module("branch1_upsample", Upsample(2))
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
        """Reference _forward_encoder + forward (yolov3_tiny_shuffle.py:79-110)."""
        route1, map1, route2 = self.features._trace(g, x)
        up = g.upsample2(self.sequence_branch1_1.branch1_conv1._trace(g, route2))
        cat = g.concat([route1, up])                                     # [x_route1, x_branch1] (:83)
        blk = self.sequence_branch1_2.branch1_conv2                      # its weights see 232 + 128 logical channels
        cmap = map1 + [route1.c + i for i in range(up.c)]
        b1 = g.conv(cat, _scatter(blk.folded(), list(range(blk.out_channels)), blk.out_channels, cmap, cat.c), act="leaky")
        g.head(plain_head(g, b1, self.sequence_branch1_2.branch1_conv3), self.yolo1)
        b2 = self.sequence_branch2.branch2_conv1._trace(g, route2)
        g.head(plain_head(g, b2, self.sequence_branch2.branch2_conv2), self.yolo2)
