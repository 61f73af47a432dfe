"""YOLOv3 (Darknet-53, no SPP, 3 heads) on the HIP path — host mirror of reference models/yolov3.py:
same constructor, module names (state_dict keys) and forward return structure.  Note the reference's head
quirks, reproduced here: heads 1 and 2 end in a plain biased 1x1 ``nn.Conv2d`` (:38,:55), head 3 ends in a
3x3 ``ConvBlock`` (BN + LeakyReLU, default size=3, :70)."""
from __future__ import annotations

from torch import nn

from .. import engine
from .yolo_base import ConvBlock, YOLOBase
from .yolo_layer import Concat, Upsample
from .yolov3_spp import DownSample
from .yolov3_tiny import plain_head


def _seq(named):
    s = nn.Sequential()
    for name, m in named:
        s.add_module(name, m)
    return s


def _chain(g, x, modules, f32_last=False):
    blocks = [m for m in modules if isinstance(m, ConvBlock)]
    for i, b in enumerate(blocks):
        x = b._trace(g, x, f32_out=f32_last and i == len(blocks) - 1)
    return x


class YOLOv3(YOLOBase):
    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        kd = self.kernels_divider
        out = self.yolo_layer_input_size
        c = lambda v: v // kd
        self.conv1 = ConvBlock(self.in_channels, c(32))
        prev = c(32)
        for i, (wd, rep) in enumerate(zip((64, 128, 256, 512, 1024), (0, 1, 7, 7, 3)), start=1):
            setattr(self, f"down{i}", DownSample(prev, c(wd), repeat=rep))
            prev = c(wd)
        self.down = [getattr(self, f"down{i}") for i in range(1, 6)]

        self.seq = _seq([("conv1", ConvBlock(c(1024), c(512), 1)), ("conv2", ConvBlock(c(512), c(1024), 3)),
                         ("conv3", ConvBlock(c(1024), c(512), 1)), ("conv4", ConvBlock(c(512), c(1024), 3)),
                         ("conv5", ConvBlock(c(1024), c(512), 1))])
        self.seq_y1 = _seq([("conv1", ConvBlock(c(512), c(1024), 3)), ("conv2", nn.Conv2d(c(1024), out, 1, 1))])
        self.seqy2_1 = _seq([("conv", ConvBlock(c(512), c(256), 1)), ("up", Upsample(2))])
        self.seqy2_2 = _seq([("concat", Concat(1)), ("conv1", ConvBlock(c(256) + c(512), c(256), 1)),
                             ("conv2", ConvBlock(c(256), c(512))), ("conv3", ConvBlock(c(512), c(256), 1)),
                             ("conv4", ConvBlock(c(256), c(512))), ("conv5", ConvBlock(c(512), c(256), 1))])
        self.seqy2_3 = _seq([("conv6", ConvBlock(c(256), c(512))), ("conv7", nn.Conv2d(c(512), out, 1, 1))])
        self.seqy3_1 = _seq([("conv", ConvBlock(c(256), c(128), 1)), ("up", Upsample(2))])
        self.seqy3_2 = _seq([("concat", Concat(1)), ("conv1", ConvBlock(c(128) + c(256), c(128), 1)),
                             ("conv2", ConvBlock(c(128), c(256))), ("conv3", ConvBlock(c(256), c(128), 1)),
                             ("conv4", ConvBlock(c(128), c(256))), ("conv5", ConvBlock(c(256), c(128), 1)),
                             ("conv6", ConvBlock(c(128), c(256))), ("conv7", ConvBlock(c(256), out))])
        self.yolo1, self.yolo2, self.yolo3 = self._create_yolo_layers()

    @property
    def yolo_layers(self):
        return self.yolo1, self.yolo2, self.yolo3

    def _trace(self, g: engine.Recorder, x):
        """Reference _forward_encoder + forward (yolov3.py:74-116)."""
        x = self.conv1._trace(g, x)
        subs = []
        for i, stage in enumerate(self.down):
            x, sub = stage._trace(g, x, need_sub=i in (2, 3))
            subs.append(sub)
        x = _chain(g, x, self.seq)
        b1 = self.seq_y1.conv1._trace(g, x)
        g.head(plain_head(g, b1, self.seq_y1.conv2), self.yolo1)
        y = g.concat([g.upsample2(self.seqy2_1.conv._trace(g, x)), subs[3]])
        y = _chain(g, y, self.seqy2_2)
        b2 = self.seqy2_3.conv6._trace(g, y)
        g.head(plain_head(g, b2, self.seqy2_3.conv7), self.yolo2)
        z = g.concat([g.upsample2(self.seqy3_1.conv._trace(g, y)), subs[2]])
        g.head(_chain(g, z, self.seqy3_2, f32_last=True), self.yolo3)
