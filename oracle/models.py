"""Oracle model graphs (functional, fp32, CPU torch).  TEST INFRASTRUCTURE.

Restates the wiring of the three model families in BASELINE.json's configs.
All functions take a reference-keyed ``state_dict`` and return what the
reference's ``forward`` returns in eval mode: ``(io, (p_0, p_1, ...))``.
"""
from __future__ import annotations

import torch

from .blocks import (conv_bn_leaky, max_pool, plain_conv1x1, tap, upsample2,
                     yolo_decode)

SPP_STAGE_REPEATS = (1, 2, 8, 8, 4)   # DownSample(repeat=0,1,7,7,3)+1, yolov3_spp.py:63-67,22

# Anchor sets used throughout tests / bench.
SPP_ANCHORS = (((10., 13.), (16., 30.), (33., 23.)),
               ((30., 61.), (62., 45.), (59., 119.)),
               ((116., 90.), (156., 198.), (373., 326.)))       # order of yolov3_spp.py:196-198
TINY_ANCHORS = (((10., 14.), (23., 27.), (37., 58.)),
                ((81., 82.), (135., 169.), (344., 319.)))       # yolo_base.py:88-89


def darknet_stage(sd, prefix, x, repeats):
    """DownSample.forward — yolov3_spp.py:38-46.

    Returns (x_after_last_add, sub) where ``sub`` is the LAST residual branch
    output *before* its add (the reference's route tensors are pre-add).
    """
    x = conv_bn_leaky(sd, f"{prefix}.conv0", x, stride=2)
    sub = x
    for i in range(repeats):
        sub = conv_bn_leaky(sd, f"{prefix}.seq{i}.0", x)            # 1x1 C -> C/2
        sub = conv_bn_leaky(sd, f"{prefix}.seq{i}.1", sub)          # 3x3 C/2 -> C
        x = tap(f"{prefix}.add{i}", x + sub)                        # Add, :12-14
    return x, sub


def spp_encoder(sd, x):
    """YOLOv3SPP._forward_encoder — yolov3_spp.py:119-139."""
    x = conv_bn_leaky(sd, "conv1", x)
    subs = []
    for i, rep in enumerate(SPP_STAGE_REPEATS, start=1):
        x, sub = darknet_stage(sd, f"down{i}", x, rep)
        subs.append(sub)
    for n in ("conv1", "conv2", "conv3"):
        x = conv_bn_leaky(sd, f"sequence_spp.{n}", x)
    x = torch.cat([max_pool(x, 5, 1), max_pool(x, 9, 1), max_pool(x, 13, 1), x], 1)  # :129
    for n in ("conv1", "conv2", "conv3"):
        x = conv_bn_leaky(sd, f"branch1_1.{n}", x)
    b1 = conv_bn_leaky(sd, "branch1_2.conv1", x)
    b1 = conv_bn_leaky(sd, "branch1_2.conv2", b1)                   # head is a ConvBlock (:86)

    y = upsample2(conv_bn_leaky(sd, "branch2_1.0", x))
    y = torch.cat([y, subs[3]], 1)                                  # :133
    for n in ("conv1", "conv2", "conv3", "conv4", "conv5"):
        y = conv_bn_leaky(sd, f"branch2_2.{n}", y)
    b2 = conv_bn_leaky(sd, "branch2_3.conv6", y)
    b2 = conv_bn_leaky(sd, "branch2_3.conv7", b2)

    z = upsample2(conv_bn_leaky(sd, "branch3_1.0", y))
    z = torch.cat([z, subs[2]], 1)                                  # :137
    for n in ("conv1", "conv2", "conv3", "conv4", "conv5", "conv6", "conv7"):
        z = conv_bn_leaky(sd, f"branch3_2.{n}", z)
    return b1, b2, z


def spp_forward(sd, x, anchors=SPP_ANCHORS, n_class=80):
    """YOLOv3SPP.forward (eval) — yolov3_spp.py:141-164."""
    img_size = max(x.shape[-2:])
    heads = spp_encoder(sd, x)
    outs = [yolo_decode(h, a, n_class, img_size) for h, a in zip(heads, anchors)]
    io, p = zip(*outs)
    return torch.cat(io, 1), tuple(p)


def tiny_encoder(sd, x):
    """YOLOv3Tiny._forward_encoder — yolov3_tiny.py:67-77 (+ ctor :18-43)."""
    for i in (1, 2, 3, 4):                                          # ConvPoolBlock 2/2
        x = max_pool(conv_bn_leaky(sd, f"sequence_1.conv{i}", x), 2, 2)
    route1 = conv_bn_leaky(sd, "sequence_1.conv5", x)
    y = max_pool(route1, 2, 2)                                      # nn.MaxPool2d(2,2), :26
    y = max_pool(conv_bn_leaky(sd, "sequence_2.conv6", y), 2, 1)    # dilated special
    y = conv_bn_leaky(sd, "sequence_2.conv7", y)
    route2 = conv_bn_leaky(sd, "sequence_2.conv8", y)
    return _tiny_heads(sd, route1, route2)


def _tiny_heads(sd, route1, route2):
    """Shared tiny-style head — yolov3_tiny.py:30-43,72-77 and
    yolov3_tiny_mobilenet.py:58-69,81-86 (same wiring, different widths)."""
    up = upsample2(conv_bn_leaky(sd, "sequence_branch1_1.branch1_conv1", route2))
    b1 = torch.cat([route1, up], 1)                                 # order [route1, up], :73
    b1 = conv_bn_leaky(sd, "sequence_branch1_2.branch1_conv2", b1)
    b1 = plain_conv1x1(sd, "sequence_branch1_2.branch1_conv3", b1)
    b2 = conv_bn_leaky(sd, "sequence_branch2.branch2_conv1", route2)
    b2 = plain_conv1x1(sd, "sequence_branch2.branch2_conv2", b2)
    return b1, b2


def tiny_forward(sd, x, anchors=TINY_ANCHORS, n_class=80):
    """YOLOv3Tiny.forward (eval) — yolov3_tiny.py:79-100."""
    img_size = max(x.shape[-2:])
    heads = tiny_encoder(sd, x)
    outs = [yolo_decode(h, a, n_class, img_size) for h, a in zip(heads, anchors)]
    io, p = zip(*outs)
    return torch.cat(io, 1), tuple(p)


def tiny_mobile_head_forward(sd, route1, route2, img_size, anchors=TINY_ANCHORS, n_class=80):
    """The reference-owned part of YOLOv3TinyMobile — yolov3_tiny_mobilenet.py:78-109
    with the encoder outputs (96ch @/16, 1280ch @/32) supplied by the caller."""
    heads = _tiny_heads(sd, route1, route2)
    outs = [yolo_decode(h, a, n_class, img_size) for h, a in zip(heads, anchors)]
    io, p = zip(*outs)
    return torch.cat(io, 1), tuple(p)


def tiny_mobile_forward(sd, x, anchors=TINY_ANCHORS, n_class=80):
    """YOLOv3TinyMobile.forward (eval) — yolov3_tiny_mobilenet.py:88-109; encoder: oracle/mobilenet.py
    (parity unpinned, see its header)."""
    from .mobilenet import mobilenet_routes
    route1, route2 = mobilenet_routes(sd, x)
    return tiny_mobile_head_forward(sd, route1, route2, max(x.shape[-2:]), anchors, n_class)


def tiny_efficient_forward(sd, x, anchors=TINY_ANCHORS, n_class=80):
    """YOLOv3TinyEfficient.forward (eval) - yolov3_tiny_efficient.py:114-147 (the tiny-style head, :84-100: widths 128 / 128,
    the same wiring as yolov3_tiny.py); encoder: oracle/efficientnet.py (parity unpinned, see its header)."""
    from .efficientnet import efficientnet_routes
    route1, route2 = efficientnet_routes(sd, x)
    heads = _tiny_heads(sd, route1, route2)
    outs = [yolo_decode(h, a, n_class, max(x.shape[-2:])) for h, a in zip(heads, anchors)]
    io, p = zip(*outs)
    return torch.cat(io, 1), tuple(p)


def tiny_shuffle_forward(sd, x, anchors=TINY_ANCHORS, n_class=80):
    """YOLOv3TinyShuffle.forward (eval) — yolov3_tiny_shuffle.py:74-110 (the tiny-style head, :58-69); encoder:
    oracle/shufflenet.py (parity unpinned, see its header)."""
    from .shufflenet import shufflenet_routes
    route1, route2 = shufflenet_routes(sd, x)
    heads = _tiny_heads(sd, route1, route2)
    outs = [yolo_decode(h, a, n_class, max(x.shape[-2:])) for h, a in zip(heads, anchors)]
    io, p = zip(*outs)
    return torch.cat(io, 1), tuple(p)


def tiny_squeeze_forward(sd, x, anchors=TINY_ANCHORS, n_class=80):
    """YOLOv3TinySqueeze.forward (eval) — yolov3_tiny_squeeze.py:71-104; encoder: oracle/squeezenet.py (parity
    unpinned, see its header).  Both heads sit on the same grid: the concat takes no upsample (:77)."""
    from .squeezenet import squeezenet_routes
    img_size = max(x.shape[-2:])
    route1, route2 = squeezenet_routes(sd, x)
    b1 = conv_bn_leaky(sd, "sequence_branch1_1.branch1_conv1", route2)
    b1 = conv_bn_leaky(sd, "sequence_branch1_2.branch1_conv2", torch.cat([route1, b1], 1))
    b1 = plain_conv1x1(sd, "sequence_branch1_2.branch1_conv3", b1)
    b2 = plain_conv1x1(sd, "sequence_branch2.branch2_conv2", conv_bn_leaky(sd, "sequence_branch2.branch2_conv1", route2))
    outs = [yolo_decode(h, a, n_class, img_size) for h, a in zip((b1, b2), anchors)]
    io, p = zip(*outs)
    return torch.cat(io, 1), tuple(p)


def yolov3_forward(sd, x, anchors=SPP_ANCHORS, n_class=80):
    """YOLOv3.forward (eval) — /root/reference/pytorch_yolo/models/yolov3.py:74-116.  Heads 1/2 end in a plain
    biased 1x1 conv (:38,:55); head 3 ends in a 3x3 ConvBlock (:70)."""
    img_size = max(x.shape[-2:])
    x = conv_bn_leaky(sd, "conv1", x)
    subs = []
    for i, rep in enumerate(SPP_STAGE_REPEATS, start=1):
        x, sub = darknet_stage(sd, f"down{i}", x, rep)
        subs.append(sub)
    for n in ("conv1", "conv2", "conv3", "conv4", "conv5"):
        x = conv_bn_leaky(sd, f"seq.{n}", x)
    b1 = plain_conv1x1(sd, "seq_y1.conv2", conv_bn_leaky(sd, "seq_y1.conv1", x))
    y = torch.cat([upsample2(conv_bn_leaky(sd, "seqy2_1.conv", x)), subs[3]], 1)          # :87-88
    for n in ("conv1", "conv2", "conv3", "conv4", "conv5"):
        y = conv_bn_leaky(sd, f"seqy2_2.{n}", y)
    b2 = plain_conv1x1(sd, "seqy2_3.conv7", conv_bn_leaky(sd, "seqy2_3.conv6", y))
    z = torch.cat([upsample2(conv_bn_leaky(sd, "seqy3_1.conv", y)), subs[2]], 1)          # :91-92
    for n in ("conv1", "conv2", "conv3", "conv4", "conv5", "conv6", "conv7"):
        z = conv_bn_leaky(sd, f"seqy3_2.{n}", z)
    outs = [yolo_decode(h, a, n_class, img_size) for h, a in zip((b1, b2, z), anchors)]
    io, p = zip(*outs)
    return torch.cat(io, 1), tuple(p)


def lite_forward(sd, x, anchors=SPP_ANCHORS, n_class=80):
    """LiteYOLOv3.forward (eval) — /root/reference/pytorch_yolo/models/lite_yolo.py:79-119."""
    img_size = max(x.shape[-2:])
    downs = []
    for n in ("down1", "down2", "down3"):
        x = max_pool(conv_bn_leaky(sd, n, x), 2, 2)                                        # ConvPoolBlock
        downs.append(x)
    for n in ("down4", "down5"):                                                           # Down, :23-31
        x = max_pool(conv_bn_leaky(sd, f"{n}.conv_pool", conv_bn_leaky(sd, f"{n}.conv", x)), 2, 2)
        downs.append(x)
    x = conv_bn_leaky(sd, "seq.conv1", x)
    b1 = plain_conv1x1(sd, "seq_y1.conv2", conv_bn_leaky(sd, "seq_y1.conv1", x))
    y = torch.cat([upsample2(conv_bn_leaky(sd, "seqy2_1.conv", x)), downs[3]], 1)
    y = conv_bn_leaky(sd, "seqy2_2.conv1.conv2", conv_bn_leaky(sd, "seqy2_2.conv1.conv1", y))   # Conv, :12-20
    y = conv_bn_leaky(sd, "seqy2_2.conv2", y)
    b2 = plain_conv1x1(sd, "seqy2_3.conv7", conv_bn_leaky(sd, "seqy2_3.conv6", y))
    z = torch.cat([upsample2(conv_bn_leaky(sd, "seqy3_1.conv", y)), downs[2]], 1)
    z = conv_bn_leaky(sd, "seqy3_2.conv1.conv2", conv_bn_leaky(sd, "seqy3_2.conv1.conv1", z))
    z = conv_bn_leaky(sd, "seqy3_2.conv2", z)                                              # 3x3 ConvBlock head
    outs = [yolo_decode(h, a, n_class, img_size) for h, a in zip((b1, b2, z), anchors)]
    io, p = zip(*outs)
    return torch.cat(io, 1), tuple(p)
