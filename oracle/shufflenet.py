"""Oracle for the ShuffleNetV2 x1.0 encoder of YOLOv3TinyShuffle.  TEST INFRASTRUCTURE.

**PARITY UNPINNED.**  The reference takes this encoder from torchvision
(``torchvision.models.shufflenet_v2_x1_0(True)``, /root/reference/pytorch_yolo/models/yolov3_tiny_shuffle.py:3,13-47).
torchvision is not installed in the build image and none of the reference's files pins this arithmetic, so this file
restates the *published* ShuffleNetV2 x1.0 (Ma et al., ECCV 2018; torchvision layer order and state_dict key names) and
is checked only against itself on the GPU.

    conv1    Conv2d(3, 24, 3, stride 2, pad 1, no bias) + BN + ReLU;   maxpool MaxPool2d(3, 2, 1)
    stage2/3/4  4 / 8 / 4 units with 116 / 232 / 464 output channels; unit(inp, oup, stride), bf = oup // 2:
        stride 2:  out = cat(branch1(x), branch2(x));   stride 1:  x1, x2 = x.chunk(2); out = cat(x1, branch2(x2))
        branch1 = dw3x3(stride) + BN, 1x1 -> bf + BN + ReLU;   branch2 = 1x1 -> bf + BN + ReLU, dw3x3(stride) + BN,
        1x1 + BN + ReLU;   then channel_shuffle(out, groups = 2)
    conv5    Conv2d(464, 1024, 1, no bias) + BN + ReLU
The reference wraps them as sequence1 = (conv1, maxpool, stage2, stage3) -> 232 ch @/16 and sequence2 = (stage4, conv5)
-> 1024 ch @/32.
"""
from __future__ import annotations

import torch
import torch.nn.functional as F

STAGES = ((4, 116), (8, 232), (4, 464))


def _bn(sd, p, x):
    return F.batch_norm(x, sd[p + ".running_mean"], sd[p + ".running_var"], sd[p + ".weight"], sd[p + ".bias"], training=False,
                        eps=1e-5)


def channel_shuffle(x, groups=2):
    b, c, h, w = x.shape
    return x.view(b, groups, c // groups, h, w).transpose(1, 2).reshape(b, c, h, w)


def _branch1(sd, p, x, stride):
    x = _bn(sd, p + ".1", F.conv2d(x, sd[p + ".0.weight"], None, stride=stride, padding=1, groups=x.shape[1]))
    return F.relu(_bn(sd, p + ".3", F.conv2d(x, sd[p + ".2.weight"])))


def _branch2(sd, p, x, stride):
    x = F.relu(_bn(sd, p + ".1", F.conv2d(x, sd[p + ".0.weight"])))
    x = _bn(sd, p + ".4", F.conv2d(x, sd[p + ".3.weight"], None, stride=stride, padding=1, groups=x.shape[1]))
    return F.relu(_bn(sd, p + ".6", F.conv2d(x, sd[p + ".5.weight"])))


def _unit(sd, p, x, stride):
    if stride == 1:
        x1, x2 = x.chunk(2, 1)
        out = torch.cat([x1, _branch2(sd, p + ".branch2", x2, 1)], 1)
    else:
        out = torch.cat([_branch1(sd, p + ".branch1", x, stride), _branch2(sd, p + ".branch2", x, stride)], 1)
    return channel_shuffle(out, 2)


def shufflenet_routes(sd, x, prefix="features"):
    """(route1 232 ch @/16, route2 1024 ch @/32) from a state_dict with the reference's key names."""
    s1, s2 = prefix + ".sequence1", prefix + ".sequence2"
    x = F.relu(_bn(sd, s1 + ".0.1", F.conv2d(x, sd[s1 + ".0.0.weight"], None, stride=2, padding=1)))
    x = F.max_pool2d(x, 3, 2, 1)
    for name, (rep, _oup) in zip((s1 + ".2", s1 + ".3", s2 + ".0"), STAGES):
        for u in range(rep):
            x = _unit(sd, f"{name}.{u}", x, 2 if u == 0 else 1)
        if name.endswith("sequence1.3"):
            route1 = x
    x = F.relu(_bn(sd, s2 + ".1.1", F.conv2d(x, sd[s2 + ".1.0.weight"])))
    return route1, x
