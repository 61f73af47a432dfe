"""YOLOv3-SPP (Darknet-53 + SPP + 3 heads) on the HIP path.

Host mirror of reference models/yolov3_spp.py: same constructor, module names
(hence ``state_dict`` keys) and ``forward`` return structure.  The wiring is
recorded once by ``_trace`` and executed by libyolo_hip.so; see
pytorch_yolo_amd/engine.py for how Add / Upsample / Concat / SPP are fused away.
"""
from __future__ import annotations

from torch import nn

from .. import engine
from .yolo_base import ConvBlock, MaxPool, YOLOBase
from .yolo_layer import Concat, Upsample


class Add(nn.Module):
    """Residual add marker (reference yolov3_spp.py:12-14); fused into the conv epilogue."""


class DownSample(nn.Module):
    """One Darknet-53 stage (reference yolov3_spp.py:17-50): stride-2 3x3, then ``repeat+1``
    residual units [1x1 C->C/2, 3x3 C/2->C].  Returns (x, sub) where ``sub`` is the LAST
    unit's branch output BEFORE its add — the tensor the reference routes to the FPN (:42-46)."""

    def __init__(self, in_channels, out_channels, repeat=0):
        super().__init__()
        self._out_channels = out_channels
        self.conv0 = ConvBlock(in_channels, out_channels, size=3, stride=2)
        self.n_units = repeat + 1
        for i in range(self.n_units):
            setattr(self, f"seq{i}", nn.Sequential(ConvBlock(out_channels, out_channels // 2, size=1, stride=1),
                                                   ConvBlock(out_channels // 2, out_channels, size=3, stride=1)))
            setattr(self, f"add{i}", Add())

    @property
    def out_channels(self):
        return self._out_channels

    def _trace(self, g: engine.Recorder, x, need_sub=True):
        x = self.conv0._trace(g, x)
        sub = x
        for i in range(self.n_units):
            squeeze, expand = getattr(self, f"seq{i}")
            last = need_sub and i == self.n_units - 1
            out = expand._trace(g, squeeze._trace(g, x), residual=x, want_preadd=last)
            x, sub = out if last else (out, None)
        return x, sub

    def forward(self, x):
        return engine.run_standalone(lambda g, s: self._trace(g, s, need_sub=True), x)


class YOLOv3SPP(YOLOBase):
    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        kd = self.kernels_divider
        out = self.yolo_layer_input_size

        self.conv1 = ConvBlock(self.in_channels, 32 // kd, stride=1, size=3)
        widths = (64, 128, 256, 512, 1024)
        repeats = (0, 1, 7, 7, 3)
        prev = self.conv1.out_channels
        for i, (wd, rep) in enumerate(zip(widths, repeats), start=1):
            setattr(self, f"down{i}", DownSample(prev, wd // kd, repeat=rep))
            prev = wd // kd
        self.down = [getattr(self, f"down{i}") for i in range(1, 6)]       # plain list, like the reference

        def seq(named_blocks):
            s = nn.Sequential()
            for name, blk in named_blocks:
                s.add_module(name, blk)
            return s

        c1024, c512, c256, c128 = 1024 // kd, 512 // kd, 256 // kd, 128 // kd
        self.sequence_spp = seq([("conv1", ConvBlock(c1024, c512, size=1)),
                                 ("conv2", ConvBlock(c512, c1024, size=3)),
                                 ("conv3", ConvBlock(c1024, c512, size=1))])
        self.spp1, self.spp2, self.spp3 = MaxPool(5, 1), MaxPool(9, 1), MaxPool(13, 1)

        self.branch1_1 = seq([("concat", Concat(1)),
                              ("conv1", ConvBlock(4 * c512, c512, size=1)),
                              ("conv2", ConvBlock(c512, c1024, size=3)),
                              ("conv3", ConvBlock(c1024, c512, size=1))])
        self.branch1_2 = seq([("conv1", ConvBlock(c512, c1024, size=3)),
                              ("conv2", ConvBlock(c1024, out, size=1))])

        self.branch2_1 = nn.Sequential(ConvBlock(c512, c256, size=1), Upsample(2))
        self.branch2_2 = seq([("concat", Concat(1)),
                              ("conv1", ConvBlock(c256 + self.down[3].out_channels, c256, size=1)),
                              ("conv2", ConvBlock(c256, c512, size=3)),
                              ("conv3", ConvBlock(c512, c256, size=1)),
                              ("conv4", ConvBlock(c256, c512, size=3)),
                              ("conv5", ConvBlock(c512, c256, size=1))])
        self.branch2_3 = seq([("conv6", ConvBlock(c256, c512, size=3)),
                              ("conv7", ConvBlock(c512, out, size=1))])

        self.branch3_1 = nn.Sequential(ConvBlock(c256, c128, size=1), Upsample(2))
        self.branch3_2 = seq([("concat", Concat(1)),
                              ("conv1", ConvBlock(c128 + self.down[2].out_channels, c128, size=1)),
                              ("conv2", ConvBlock(c128, c256, size=3)),
                              ("conv3", ConvBlock(c256, c128, size=1)),
                              ("conv4", ConvBlock(c128, c256, size=3)),
                              ("conv5", ConvBlock(c256, c128, size=1)),
                              ("conv6", ConvBlock(c128, c256, size=3)),
                              ("conv7", ConvBlock(c256, out, size=1))])

        self.yolo1, self.yolo2, self.yolo3 = self._create_yolo_layers()   # needs 3 anchor groups (:116)

    @property
    def yolo_layers(self):
        return self.yolo1, self.yolo2, self.yolo3

    @staticmethod
    def _chain(g, x, blocks, head_last=False):
        blocks = [b for b in blocks if isinstance(b, ConvBlock)]
        for i, b in enumerate(blocks):
            x = b._trace(g, x, f32_out=head_last and i == len(blocks) - 1)
        return x

    def _trace(self, g: engine.Recorder, x):
        """Reference _forward_encoder + forward (yolov3_spp.py:119-164)."""
        x = self.conv1._trace(g, x)
        subs = []
        for i, stage in enumerate(self.down):
            x, sub = stage._trace(g, x, need_sub=i in (2, 3))      # only downs[2], downs[3] are routed (:133,:137)
            subs.append(sub)
        x = self._chain(g, x, self.sequence_spp)
        x = g.spp_concat(x)                                        # cat([p5, p9, p13, x]) :129
        x = self._chain(g, x, self.branch1_1)
        g.head(self._chain(g, x, self.branch1_2, head_last=True), self.yolo1)

        y = g.concat([g.upsample2(self.branch2_1[0]._trace(g, x)), subs[3]])       # :132-133
        y = self._chain(g, y, self.branch2_2)
        g.head(self._chain(g, y, self.branch2_3, head_last=True), self.yolo2)

        z = g.concat([g.upsample2(self.branch3_1[0]._trace(g, y)), subs[2]])       # :136-137
        g.head(self._chain(g, z, self.branch3_2, head_last=True), self.yolo3)
