"""Test helpers shared by CPU and GPU suites."""
import os

import numpy as np
import torch

import _cases as C
from pytorch_yolo_amd import LiteYOLOv3, YOLOv3, YOLOv3SPP, YOLOv3Tiny
from pytorch_yolo_amd.utils.synthetic import synth_images, synth_state_dict

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
FAMILY = {"spp": YOLOv3SPP, "tiny": YOLOv3Tiny, "yolov3": YOLOv3, "lite": LiteYOLOv3}


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"))


def build_case(case):
    """(product model in eval mode with seeded weights, state_dict, input x) for a MODEL/FULL case tuple."""
    family, kw, bs, h, w, wseed, xseed = case
    model = FAMILY[family](**kw).eval()
    sd = synth_state_dict(model.state_dict(), wseed, n_class=kw["n_class"])
    model.load_state_dict(sd)
    return model, sd, synth_images(bs, h, w, xseed)


def oracle_forward(case, sd, x):
    from oracle import models as om
    family, kw = case[0], case[1]
    fwd = {"spp": om.spp_forward, "tiny": om.tiny_forward, "yolov3": om.yolov3_forward, "lite": om.lite_forward}[family]
    with torch.no_grad():
        return fwd(sd, x, kw["anchors"], kw["n_class"])
