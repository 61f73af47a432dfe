"""Oracle for the SqueezeNet 1.1 encoder of YOLOv3TinySqueeze.  TEST INFRASTRUCTURE.

**PARITY UNPINNED.**  The reference takes this encoder from torchvision
(``torchvision.models.squeezenet1_1(True).features``, /root/reference/pytorch_yolo/models/yolov3_tiny_squeeze.py:3,19-31).
torchvision is not installed in the build image and none of the reference's files pins this arithmetic, so this file
restates the *published* SqueezeNet 1.1 (Iandola et al. 2016; torchvision layer order and state_dict key names) and is
checked only against itself on the GPU.

    features[0..2]   Conv2d(3, 64, 3, stride 2, NO padding) + ReLU, MaxPool2d(3, 2, ceil_mode=True)
    features[3..4]   Fire(64,16,64,64), Fire(128,16,64,64);            features[5] MaxPool2d(3, 2, ceil_mode=True)
    features[6..7]   Fire(128,32,128,128), Fire(256,32,128,128);       features[8] MaxPool2d(3, 2, ceil_mode=True)
    features[9..12]  Fire(256,48,192,192), Fire(384,48,192,192), Fire(384,64,256,256), Fire(512,64,256,256)
    Fire = squeeze 1x1 + ReLU -> cat(expand 1x1 + ReLU, expand 3x3 pad 1 + ReLU); every conv has a bias, no BN.
The reference splits four modules from the end: sequence1 = features[:9] (256 ch), sequence2 = features[9:] (512 ch,
same grid as sequence1's output).
"""
from __future__ import annotations

import torch
import torch.nn.functional as F

FIRES = {3: (64, 16, 64, 64), 4: (128, 16, 64, 64), 6: (128, 32, 128, 128), 7: (256, 32, 128, 128),
         9: (256, 48, 192, 192), 10: (384, 48, 192, 192), 11: (384, 64, 256, 256), 12: (512, 64, 256, 256)}
POOLS = (2, 5, 8)
SPLIT = 9


def _fire(sd, p, x):
    s = F.relu(F.conv2d(x, sd[p + ".squeeze.weight"], sd[p + ".squeeze.bias"]))
    e1 = F.relu(F.conv2d(s, sd[p + ".expand1x1.weight"], sd[p + ".expand1x1.bias"]))
    e3 = F.relu(F.conv2d(s, sd[p + ".expand3x3.weight"], sd[p + ".expand3x3.bias"], padding=1))
    return torch.cat([e1, e3], 1)


def squeezenet_routes(sd, x, prefix="features"):
    """(route1 256 ch, route2 512 ch) from a state_dict with the reference's key names
    (``features.sequence1.<i>...`` / ``features.sequence2.<i>...``)."""
    def name(i):
        return f"{prefix}.sequence1.{i}" if i < SPLIT else f"{prefix}.sequence2.{i - SPLIT}"
    x = F.relu(F.conv2d(x, sd[name(0) + ".weight"], sd[name(0) + ".bias"], stride=2))
    route1 = None
    for i in range(2, 13):
        if i in POOLS:
            x = F.max_pool2d(x, 3, 2, ceil_mode=True)
        else:
            x = _fire(sd, name(i), x)
        if i == SPLIT - 1:
            route1 = x
    return route1, x
