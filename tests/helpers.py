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


def build_separable_case():
    """(product YOLOv3-SPP in eval mode, state_dict, x) of tests/_cases.py::SEPARABLE: seeded weights with the head BN arrays
    the golden stores (calibrated by make_golden.py from the reference's own raw head outputs), the seeded patch image."""
    sep = C.SEPARABLE
    g = load_golden("full_spp_640_separable")
    model = YOLOv3SPP(n_class=80, kernels_divider=1, anchors=C.SPP_ANCHORS).eval()
    sd = synth_state_dict(model.state_dict(), sep["weight_seed"], n_class=80)
    for k, h in enumerate(C.SEPARABLE_HEADS):
        sd[h + ".sequence.batch_norm.weight"] = torch.from_numpy(g[f"head_bn_weight_{k}"].copy())
        sd[h + ".sequence.batch_norm.bias"] = torch.from_numpy(g[f"head_bn_bias_{k}"].copy())
    model.load_state_dict(sd)
    return model, sd, torch.from_numpy(C.patch_image(sep["image_seed"], sep["n_patches"])), g


def build_rule_case(seed: int):
    """(product YOLOv3-SPP in eval mode, state_dict, x, golden) of one rule-selected case (tests/_cases.py::RULE, RULE_SEEDS): seeded
    weights with the head BN arrays the golden stores (calibrated by make_golden.py from the reference's own raw head outputs)."""
    rule = C.RULE
    g = load_golden(f"full_spp_640_rule_{seed}")
    model = YOLOv3SPP(n_class=80, kernels_divider=1, anchors=C.SPP_ANCHORS).eval()
    sd = synth_state_dict(model.state_dict(), rule["weight_seed"], n_class=80)
    for k, h in enumerate(C.SEPARABLE_HEADS):
        sd[h + ".sequence.batch_norm.weight"] = torch.from_numpy(g[f"head_bn_weight_{k}"].copy())
        sd[h + ".sequence.batch_norm.bias"] = torch.from_numpy(g[f"head_bn_bias_{k}"].copy())
    model.load_state_dict(sd)
    return model, sd, torch.from_numpy(C.patch_image(seed, rule["n_patches"])), g


def box_iou(a, b):
    iw = max(0.0, min(a[2], b[2]) - max(a[0], b[0]))
    ih = max(0.0, min(a[3], b[3]) - max(a[1], b[1]))
    inter = iw * ih
    return inter / ((a[2] - a[0]) * (a[3] - a[1]) + (b[2] - b[0]) * (b[3] - b[1]) - inter + 1e-16)


def strict_share(da, db, iou_min=0.9, dconf=0.03):
    """Share of the detections [n, 7] in ``da`` that have a partner in ``db``: same class, IoU >= iou_min, |dconf| <= dconf."""
    if da is None or len(da) == 0:
        return 1.0
    if db is None or len(db) == 0:
        return 0.0
    return sum(any(int(r[6]) == int(q[6]) and abs(float(r[4]) - float(q[4])) <= dconf and box_iou(r[:4], q[:4]) >= iou_min
                   for q in db) for r in da) / len(da)
