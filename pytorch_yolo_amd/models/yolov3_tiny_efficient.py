"""YOLOv3-tiny head on an EfficientNet-B0 encoder - host mirror of reference models/yolov3_tiny_efficient.py.

The reference takes the encoder from a third-party package (``EfficientNet.from_pretrained('efficientnet-b0')``,
efficientnet-pytorch==0.2.0, requirements.txt:1), which is not installed here and whose weights need a download.  The encoder
below restates the published EfficientNet-B0 (Tan & Le 2019) with that package's module layout and ``state_dict`` key names
(``_expand_conv / _bn0 / _depthwise_conv / _bn1 / _se_reduce / _se_expand / _project_conv / _bn2``; TensorFlow "same"
padding; BN eps 1e-3) under the reference encoder's attribute names (``features.stem.0/1``, ``features.sequence1.<i>``,
``features.sequence2.<i>``; yolov3_tiny_efficient.py:26-45), with random initialisation; load real weights with
``load_state_dict``.  The split after block 11 and the head follow the reference (:43-45,84-100).

On the device a block is: 1x1 conv + swish (implicit GEMM, swish in the epilogue) -> depthwise k x k + swish
(yolo_dwconv_fwd, TF "same" padding as a leading pad + zero reads beyond the image) -> squeeze-excite (yolo_se_fwd) -> linear
1x1 conv (+ the residual in its epilogue).
"""
from __future__ import annotations

from torch import nn

from .. import engine
from ..utils.torch_utils import fold_conv_bn
from .yolo_base import ConvBlock, YOLOBase
from .yolo_layer import Concat, Upsample
from .yolov3_tiny import trace_tiny_heads

# (repeats, kernel, stride, expand, cin, cout), squeeze-excite ratio 0.25 of the block's input width - EfficientNet-B0
_B0_BLOCKS = ((1, 3, 1, 1, 32, 16), (2, 3, 2, 6, 16, 24), (2, 5, 2, 6, 24, 40), (3, 3, 2, 6, 40, 80), (3, 5, 1, 6, 80, 112),
              (4, 5, 2, 6, 112, 192), (1, 3, 1, 6, 192, 320))
_BN_EPS, _BN_MOM = 1e-3, 0.01


def _folded(conv: nn.Conv2d, bn: nn.BatchNorm2d):
    return fold_conv_bn(conv.weight, conv.bias, bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.eps)


class SwissActivation(nn.Module):
    """x * sigmoid(x) (reference yolov3_tiny_efficient.py:13-19); a marker here: the activation runs in the conv epilogues."""

    def __init__(self):
        super().__init__()
        self.sigmoid = nn.Sigmoid()

    def forward(self, x):
        return x * self.sigmoid(x)


class MBConvBlock(nn.Module):
    """efficientnet_pytorch 0.2.0 MBConvBlock: same children names, TensorFlow "same" padding applied at trace time."""

    def __init__(self, kernel, stride, expand, cin, cout, se_ratio=0.25):
        super().__init__()
        hidden = cin * expand
        self.kernel, self.stride, self.expand, self.cin, self.cout = kernel, stride, expand, cin, cout
        if expand != 1:
            self._expand_conv = nn.Conv2d(cin, hidden, 1, bias=False)
            self._bn0 = nn.BatchNorm2d(hidden, momentum=_BN_MOM, eps=_BN_EPS)
        self._depthwise_conv = nn.Conv2d(hidden, hidden, kernel, stride, groups=hidden, bias=False)
        self._bn1 = nn.BatchNorm2d(hidden, momentum=_BN_MOM, eps=_BN_EPS)
        squeezed = max(1, int(cin * se_ratio))
        self._se_reduce = nn.Conv2d(hidden, squeezed, 1)
        self._se_expand = nn.Conv2d(squeezed, hidden, 1)
        self._project_conv = nn.Conv2d(hidden, cout, 1, bias=False)
        self._bn2 = nn.BatchNorm2d(cout, momentum=_BN_MOM, eps=_BN_EPS)

    def _trace(self, g: engine.Recorder, x):
        y = x
        if self.expand != 1:
            y = g.conv(y, _folded(self._expand_conv, self._bn0), act="swish")
        y = g.dwconv(y, _folded(self._depthwise_conv, self._bn1), stride=self.stride, act="swish", tf_same=True)
        y = g.se(y, self._se_reduce.weight, self._se_reduce.bias, self._se_expand.weight, self._se_expand.bias)
        skip = self.stride == 1 and self.cin == self.cout                  # drop_connect is the identity in eval mode
        return g.conv(y, _folded(self._project_conv, self._bn2), act="none", residual=x if skip else None)


class EfficientEncoder(nn.Module):
    """stem + blocks[:11] -> 112 ch @ /16 (route 1), blocks[11:] -> 320 ch @ /32 (route 2); reference :22-72."""

    route = 11

    def __init__(self, in_channels=3):
        super().__init__()
        self.stem = nn.Sequential(nn.Conv2d(in_channels, 32, 3, 2, bias=False), nn.BatchNorm2d(32, momentum=_BN_MOM, eps=_BN_EPS),
                                  SwissActivation())
        blocks = []
        for r, k, s, e, ci, co in _B0_BLOCKS:
            for i in range(r):
                blocks.append(MBConvBlock(k, s if i == 0 else 1, e, ci if i == 0 else co, co))
        self.sequence1 = nn.ModuleList(blocks[:self.route])
        self.sequence2 = nn.ModuleList(blocks[self.route:])

    @property
    def out_channels(self):
        return self.sequence1[-1].cout, self.sequence2[-1].cout

    def _trace(self, g: engine.Recorder, x):
        x = g.conv(x, _folded(self.stem[0], self.stem[1]), stride=2, act="swish", tf_same=True)
        for m in self.sequence1:
            x = m._trace(g, x)
        route1 = x
        for m in self.sequence2:
            x = m._trace(g, x)
        return route1, x


class YOLOv3TinyEfficient(YOLOBase):
    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        kd = self.kernels_divider
        wd = lambda c: max(8, c // kd)                                      # noqa: E731
        out = self.yolo_layer_input_size
        self.features = EfficientEncoder(in_channels=self.in_channels)
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
        route1, route2 = self.features._trace(g, x)
        trace_tiny_heads(self, g, route1, route2)
