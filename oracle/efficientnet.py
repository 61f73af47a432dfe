"""Oracle for the EfficientNet-B0 encoder of YOLOv3TinyEfficient.  TEST INFRASTRUCTURE.

**PARITY UNPINNED.**  The reference takes this encoder from a third-party package
(``efficientnet_pytorch.EfficientNet.from_pretrained('efficientnet-b0')``, pinned ``efficientnet-pytorch==0.2.0`` in
/root/reference/requirements.txt:1; call sites /root/reference/pytorch_yolo/models/yolov3_tiny_efficient.py:3,26-45: stem =
``_conv_stem`` + ``_bn0`` + swish, ``_blocks[:11]`` -> route 1, ``_blocks[11:]`` -> route 2).  The package is not installed in
the build image, its weights need a download, and none of the reference's own files pins this arithmetic, so this file
restates the *published* architecture (Tan & Le, ICML 2019, table 1) with efficientnet_pytorch 0.2.0's module layout,
``state_dict`` key names and conventions, and is checked only against itself on the GPU:

    stem        conv3x3 s2 3->32 (no bias), BN, swish
    16 MBConv   (repeats, kernel, stride, expand, cin, cout), all with squeeze-excite ratio 0.25 of the block's INPUT width:
                (1,3,1,1,32,16) (2,3,2,6,16,24) (2,5,2,6,24,40) (3,3,2,6,40,80) (3,5,1,6,80,112) (4,5,2,6,112,192) (1,3,1,6,192,320)
      block     [1x1 expand BN swish] -> depthwise k x k BN swish -> x * sigmoid(se_expand(swish(se_reduce(mean_hw x))))
                -> 1x1 project BN (+ input when stride 1 and cin == cout; drop_connect is the identity in eval mode)
    every conv  TensorFlow "same" padding (Conv2dSamePadding): out = ceil(in / stride), the odd pad row / column below / right
    every BN    eps = 1e-3 (global_params.batch_norm_epsilon)
Blocks 0..10 (through the 112-channel stage, /16) are ``features.sequence1``, blocks 11..15 (192 x 4, 320; /32) ``features.sequence2``.
"""
from __future__ import annotations

import math

import torch
import torch.nn.functional as F

# (repeats, kernel, stride, expand, cin, cout)
BLOCKS = ((1, 3, 1, 1, 32, 16), (2, 3, 2, 6, 16, 24), (2, 5, 2, 6, 24, 40), (3, 3, 2, 6, 40, 80), (3, 5, 1, 6, 80, 112),
          (4, 5, 2, 6, 112, 192), (1, 3, 1, 6, 192, 320))
ROUTE = 11
BN_EPS = 1e-3


def swish(x):
    return x * torch.sigmoid(x)


def conv_same(x, w, b=None, stride=1, groups=1):
    """efficientnet_pytorch 0.2.0 utils.Conv2dSamePadding.forward."""
    ih, iw = x.shape[-2:]
    kh, kw = w.shape[-2:]
    oh, ow = math.ceil(ih / stride), math.ceil(iw / stride)
    ph, pw = max((oh - 1) * stride + kh - ih, 0), max((ow - 1) * stride + kw - iw, 0)
    if ph > 0 or pw > 0:
        x = F.pad(x, [pw // 2, pw - pw // 2, ph // 2, ph - ph // 2])
    return F.conv2d(x, w, b, stride=stride, groups=groups)


def _bn(sd, p, x):
    return F.batch_norm(x, sd[p + ".running_mean"], sd[p + ".running_var"], sd[p + ".weight"], sd[p + ".bias"], training=False, eps=BN_EPS)


def mbconv(sd, p, x, k, stride, expand, cin, cout):
    """efficientnet_pytorch 0.2.0 model.MBConvBlock.forward (eval)."""
    y = x
    if expand != 1:
        y = swish(_bn(sd, p + "._bn0", conv_same(y, sd[p + "._expand_conv.weight"])))
    y = swish(_bn(sd, p + "._bn1", conv_same(y, sd[p + "._depthwise_conv.weight"], stride=stride, groups=cin * expand)))
    s = F.adaptive_avg_pool2d(y, 1)
    s = conv_same(swish(conv_same(s, sd[p + "._se_reduce.weight"], sd[p + "._se_reduce.bias"])), sd[p + "._se_expand.weight"], sd[p + "._se_expand.bias"])
    y = torch.sigmoid(s) * y
    y = _bn(sd, p + "._bn2", conv_same(y, sd[p + "._project_conv.weight"]))
    return y + x if (stride == 1 and cin == cout) else y


def block_list():
    """[(kernel, stride, expand, cin, cout)] of the 16 blocks in order."""
    out = []
    for r, k, s, e, ci, co in BLOCKS:
        for i in range(r):
            out.append((k, s if i == 0 else 1, e, ci if i == 0 else co, co))
    return out


def efficientnet_routes(sd, x, prefix="features"):
    """(route1 112 ch @/16, route2 320 ch @/32) from a state_dict with the reference encoder's key names
    (``features.stem.0/1``, ``features.sequence1.<i>._expand_conv...``, ``features.sequence2.<i>...``)."""
    x = swish(_bn(sd, f"{prefix}.stem.1", conv_same(x, sd[f"{prefix}.stem.0.weight"], stride=2)))
    route1 = None
    for i, (k, s, e, ci, co) in enumerate(block_list()):
        name = f"{prefix}.sequence1.{i}" if i < ROUTE else f"{prefix}.sequence2.{i - ROUTE}"
        x = mbconv(sd, name, x, k, s, e, ci, co)
        if i == ROUTE - 1:
            route1 = x
    return route1, x
