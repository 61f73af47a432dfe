"""YOLOv3-tiny (Darknet-15, 2 heads) on the HIP path — host mirror of reference
models/yolov3_tiny.py (same ctor, module names / state_dict keys, return structure)."""
from __future__ import annotations

from torch import nn

from .. import engine
from .yolo_base import ConvBlock, ConvPoolBlock, MaxPool, YOLOBase
from .yolo_layer import Concat, Upsample


class _Pool22(nn.MaxPool2d):
    """``nn.MaxPool2d(2, 2)`` as used at reference yolov3_tiny.py:26 (parameter-free)."""

    def _trace(self, g, x):
        return g.maxpool(x, 2, 2)


def plain_head(g: engine.Recorder, x, conv: nn.Conv2d):
    """nn.Conv2d(C, 3*(5+nc), kernel_size=1) with bias, no BN / activation (yolov3_tiny.py:38,42)."""
    return g.conv(x, (conv.weight.detach().float().cpu(), conv.bias.detach().float().cpu()), stride=1,
                  act="none", f32_out=True, name=getattr(conv, "_trace_name", None))


class YOLOv3Tiny(YOLOBase):
    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        kd = self.kernels_divider
        wd = lambda c: max(8, c // kd)                                   # yolov3_tiny.py:19-28
        out = self.yolo_layer_input_size

        s1 = nn.Sequential()
        prev = self.in_channels
        for i, c in enumerate((16, 32, 64, 128), start=1):
            s1.add_module(f"conv{i}", ConvPoolBlock(prev, wd(c)))
            prev = wd(c)
        s1.add_module("conv5", ConvBlock(prev, wd(256)))
        self.sequence_1 = s1

        s2 = nn.Sequential()
        s2.add_module("max_pool5", _Pool22(2, 2))
        s2.add_module("conv6", ConvPoolBlock(wd(256), wd(512), pool_stride=1))
        s2.add_module("conv7", ConvBlock(wd(512), wd(1024)))
        s2.add_module("conv8", ConvBlock(wd(1024), wd(256), size=1))
        self.sequence_2 = s2

        b11 = nn.Sequential()
        b11.add_module("branch1_conv1", ConvBlock(wd(256), wd(128), size=1))
        b11.add_module("branch1_upsample", Upsample(2))
        self.sequence_branch1_1 = b11

        b12 = nn.Sequential()
        b12.add_module("branch1_concat", Concat(1))
        b12.add_module("branch1_conv2", ConvBlock(wd(256) + wd(128), wd(256)))
        b12.add_module("branch1_conv3", nn.Conv2d(wd(256), out, kernel_size=1))
        self.sequence_branch1_2 = b12

        b2 = nn.Sequential()
        b2.add_module("branch2_conv1", ConvBlock(wd(256), wd(512)))
        b2.add_module("branch2_conv2", nn.Conv2d(wd(512), out, kernel_size=1))
        self.sequence_branch2 = b2

        self.yolo1, self.yolo2 = self._create_yolo_layers()

    @property
    def yolo_layers(self):
        return self.yolo1, self.yolo2

    def _trace(self, g: engine.Recorder, x):
        """Reference _forward_encoder + forward (yolov3_tiny.py:67-100)."""
        for blk in self.sequence_1:
            x = blk._trace(g, x)
        route1 = x
        y = route1
        for blk in self.sequence_2:
            y = blk._trace(g, y)
        route2 = y
        trace_tiny_heads(self, g, route1, route2)


def trace_tiny_heads(model, g, route1, route2):
    """Tiny-style FPN head shared with the MobileNetV2 variant
    (yolov3_tiny.py:30-43,72-77; yolov3_tiny_mobilenet.py:58-69,81-86)."""
    up = g.upsample2(model.sequence_branch1_1.branch1_conv1._trace(g, route2))
    b1 = g.concat([route1, up])                                          # order [route1, upsampled] (:73)
    b1 = model.sequence_branch1_2.branch1_conv2._trace(g, b1)
    g.head(plain_head(g, b1, model.sequence_branch1_2.branch1_conv3), model.yolo1)
    b2 = model.sequence_branch2.branch2_conv1._trace(g, route2)
    g.head(plain_head(g, b2, model.sequence_branch2.branch2_conv2), model.yolo2)
