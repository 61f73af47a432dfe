"""LiteYOLOv3 on the HIP path — host mirror of reference models/lite_yolo.py (same constructor, module
names / state_dict keys, return structure).  Head 3 ends in a 3x3 ``ConvBlock`` like the reference (:75)."""
from __future__ import annotations

from torch import nn

from .. import engine
from .yolo_base import ConvBlock, ConvPoolBlock, YOLOBase
from .yolo_layer import Concat, Upsample
from .yolov3 import _seq
from .yolov3_tiny import plain_head


class Conv(nn.Module):
    """1x1 C -> out/2 then 3x3 -> out (reference lite_yolo.py:12-20)."""

    def __init__(self, in_shape, out_channels):
        super().__init__()
        self.conv1 = ConvBlock(in_shape, out_channels // 2, 1)
        self.conv2 = ConvBlock(out_channels // 2, out_channels, 3)
        self.out_channels = out_channels

    def _trace(self, g, x):
        return self.conv2._trace(g, self.conv1._trace(g, x))


class Down(nn.Module):
    """1x1 squeeze then 3x3 + 2/2 max-pool (reference lite_yolo.py:23-31)."""

    def __init__(self, in_channels, out_channels):
        super().__init__()
        self.conv = ConvBlock(in_channels, out_channels // 2, 1)
        self.conv_pool = ConvPoolBlock(out_channels // 2, out_channels)
        self.out_channels = out_channels

    def _trace(self, g, x):
        return self.conv_pool._trace(g, self.conv._trace(g, x))


class LiteYOLOv3(YOLOBase):
    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        kd = self.kernels_divider
        out = self.yolo_layer_input_size
        c = lambda v: v // kd
        self.down1 = ConvPoolBlock(self.in_channels, c(16))
        self.down2 = ConvPoolBlock(c(16), c(32))
        self.down3 = ConvPoolBlock(c(32), c(64))
        self.down4 = Down(c(64), c(128))
        self.down5 = Down(c(128), c(256))
        self.down = [self.down1, self.down2, self.down3, self.down4, self.down5]
        self.seq = _seq([("conv1", ConvBlock(c(256), c(512), 1))])
        self.seq_y1 = _seq([("conv1", ConvBlock(c(512), c(1024), 3)), ("conv2", nn.Conv2d(c(1024), out, 1, 1))])
        self.seqy2_1 = _seq([("conv", ConvBlock(c(512), c(256), 1)), ("up", Upsample(2))])
        self.seqy2_2 = _seq([("concat", Concat(1)), ("conv1", Conv(c(256) + c(128), c(512))),
                             ("conv2", ConvBlock(c(512), c(256), 1))])
        self.seqy2_3 = _seq([("conv6", ConvBlock(c(256), c(512))), ("conv7", nn.Conv2d(c(512), out, 1, 1))])
        self.seqy3_1 = _seq([("conv", ConvBlock(c(256), c(128), 1)), ("up", Upsample(2))])
        self.seqy3_2 = _seq([("concat", Concat(1)), ("conv1", Conv(c(128) + c(64), c(256))),
                             ("conv2", ConvBlock(c(256), out))])
        self.yolo1, self.yolo2, self.yolo3 = self._create_yolo_layers()

    @property
    def yolo_layers(self):
        return self.yolo1, self.yolo2, self.yolo3

    def _trace(self, g: engine.Recorder, x):
        """Reference _forward_encoder + forward (lite_yolo.py:79-119)."""
        downs = []
        for m in self.down:
            x = m._trace(g, x)
            downs.append(x)
        x = self.seq.conv1._trace(g, x)
        g.head(plain_head(g, self.seq_y1.conv1._trace(g, x), self.seq_y1.conv2), self.yolo1)
        y = g.concat([g.upsample2(self.seqy2_1.conv._trace(g, x)), downs[3]])
        y = self.seqy2_2.conv2._trace(g, self.seqy2_2.conv1._trace(g, y))
        g.head(plain_head(g, self.seqy2_3.conv6._trace(g, y), self.seqy2_3.conv7), self.yolo2)
        z = g.concat([g.upsample2(self.seqy3_1.conv._trace(g, y)), downs[2]])
        z = self.seqy3_2.conv1._trace(g, z)
        g.head(self.seqy3_2.conv2._trace(g, z, f32_out=True), self.yolo3)
