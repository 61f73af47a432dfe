"""Oracle for the MobileNetV2 encoder of YOLOv3TinyMobile.  TEST INFRASTRUCTURE.

**PARITY UNPINNED.**  The reference takes this encoder from torchvision
(``torchvision.models.mobilenet.mobilenet_v2(pretrained=True).features``, pinned
``torchvision==0.3.0`` in /root/reference/requirements.txt:8, call sites
/root/reference/pytorch_yolo/models/yolov3_tiny_mobilenet.py:11,18-34).  torchvision is not
installed in the build image and none of the reference's files pins this arithmetic, so this
file restates the *published* MobileNetV2 (Sandler et al., CVPR 2018, table 2; torchvision 0.3.0
layer order and state_dict key names) and is checked only against itself on the GPU.

    features[0]      conv3x3 s2 3->32, BN, ReLU6
    features[1..17]  inverted residuals (t,c,n,s) = (1,16,1,1) (6,24,2,2) (6,32,3,2) (6,64,4,2)
                     (6,96,3,1) (6,160,3,2) (6,320,1,1):  [1x1 expand BN ReLU6] -> dw3x3 BN ReLU6 -> 1x1 BN
                     (+ identity when stride 1 and cin == cout)
    features[18]     conv1x1 320->1280, BN, ReLU6
The reference splits at index 14: sequence1 = features[:14] (96 ch, /16), sequence2 = features[14:] (1280 ch, /32).
"""
from __future__ import annotations

import torch.nn.functional as F

SETTING = ((1, 16, 1, 1), (6, 24, 2, 2), (6, 32, 3, 2), (6, 64, 4, 2), (6, 96, 3, 1), (6, 160, 3, 2), (6, 320, 1, 1))
ROUTE_INDEX = 14


def _bn(sd, p, x):
    return F.batch_norm(x, sd[p + ".running_mean"], sd[p + ".running_var"], sd[p + ".weight"], sd[p + ".bias"],
                        training=False, eps=1e-5)


def _conv_bn_relu6(sd, p, x, stride=1, groups=1):
    w = sd[p + ".0.weight"]
    x = F.conv2d(x, w, None, stride=stride, padding=(w.shape[-1] - 1) // 2, groups=groups)
    return F.relu6(_bn(sd, p + ".1", x))


def _inverted_residual(sd, p, x, cin, cout, stride, t):
    hidden = cin * t
    y, k = x, 0
    if t != 1:
        y = _conv_bn_relu6(sd, f"{p}.conv.{k}", y)
        k += 1
    y = _conv_bn_relu6(sd, f"{p}.conv.{k}", y, stride=stride, groups=hidden)
    k += 1
    y = _bn(sd, f"{p}.conv.{k + 1}", F.conv2d(y, sd[f"{p}.conv.{k}.weight"]))
    return x + y if (stride == 1 and cin == cout) else y


def mobilenet_routes(sd, x, prefix="features"):
    """Returns (route1 96ch @/16, route2 1280ch @/32) from a state_dict with the reference's key names
    (``features.sequence1.<i>...`` / ``features.sequence2.<i>...``)."""
    def name(i):
        return f"{prefix}.sequence1.{i}" if i < ROUTE_INDEX else f"{prefix}.sequence2.{i - ROUTE_INDEX}"
    x = _conv_bn_relu6(sd, name(0), x, stride=2)
    idx, cin, route1 = 1, 32, None
    for t, c, n, s in SETTING:
        for i in range(n):
            x = _inverted_residual(sd, name(idx), x, cin, c, s if i == 0 else 1, t)
            cin = c
            idx += 1
            if idx == ROUTE_INDEX:
                route1 = x
    x = _conv_bn_relu6(sd, name(idx), x)
    return route1, x
