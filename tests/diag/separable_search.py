#!/usr/bin/env python3
"""CPU search behind tests/_cases.py::SEPARABLE (no GPU, no reference import: the fp32 oracle and its bf16-policy re-run).

For patch-image seeds 1 .. 8 and 5 / 6 / 7 firing cells per anchor: calibrate the head BN (calibrate_separable_heads), run the oracle in
fp32 and under the product's bf16 rounding points (oracle/policy.py), MERGE-NMS both at conf 0.5 / IoU 0.5 and pair the detections
strictly (same class, IoU >= 0.9, |dconf| <= 0.03) in both directions.  The share ranges from 0.76 to 1.00 with the seed: how well
a synthetic-weight detection set survives bf16 rounding is a property of the data's conditioning (pile membership at IoU 0.5,
pivot order among saturated scores), which is why ONE well-conditioned seed is committed as a reference golden.

    PYTHONPATH=. python tests/diag/separable_search.py
"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import _cases as C                                       # noqa: E402
from oracle import models as om, nms as onms            # noqa: E402
from oracle.policy import run_policy                    # noqa: E402
from pytorch_yolo_amd import YOLOv3SPP                  # noqa: E402
from pytorch_yolo_amd.utils.synthetic import synth_state_dict   # noqa: E402


def iou(a, b):
    iw = max(0.0, min(a[2], b[2]) - max(a[0], b[0]))
    ih = max(0.0, min(a[3], b[3]) - max(a[1], b[1]))
    inter = iw * ih
    return inter / ((a[2] - a[0]) * (a[3] - a[1]) + (b[2] - b[0]) * (b[3] - b[1]) - inter + 1e-16)


def strict_share(da, db, iou_min=0.9, dconf=0.03):
    """share of the detections in ``da`` with a partner in ``db`` (same class, IoU >= iou_min, |dconf| <= dconf)"""
    if da is None or len(da) == 0:
        return 1.0
    if db is None:
        return 0.0
    return sum(any(int(r[6]) == int(q[6]) and abs(r[4] - q[4]) <= dconf and iou(r[:4], q[:4]) >= iou_min for q in db) for r in da) / len(da)


def patched_state_dict(sd, p, sep):
    wk = [h + ".sequence.batch_norm.weight" for h in C.SEPARABLE_HEADS]
    bk = [h + ".sequence.batch_norm.bias" for h in C.SEPARABLE_HEADS]
    new_w, new_b = C.calibrate_separable_heads([sd[k].numpy() for k in wk], [sd[k].numpy() for k in bk], [t[0].numpy() for t in p], 80,
                                               sep["per_anchor"], sep["span"], sep["gamma_obj"], sep["cls_gain"])
    out = dict(sd)
    for k_w, k_b, w_, b_ in zip(wk, bk, new_w, new_b):
        out[k_w], out[k_b] = torch.from_numpy(w_), torch.from_numpy(b_)
    return out


def main():
    torch.set_num_threads(os.cpu_count() or 1)
    sd = synth_state_dict(YOLOv3SPP(anchors=C.SPP_ANCHORS).state_dict(), C.SEPARABLE["weight_seed"], n_class=80)
    for seed in range(1, 9):
        x = torch.from_numpy(C.patch_image(seed, C.SEPARABLE["n_patches"]))
        with torch.no_grad():
            _, p = om.spp_forward(sd, x, C.SPP_ANCHORS, 80)
        for per_anchor in (5, 6, 7):
            sep = dict(C.SEPARABLE, per_anchor=per_anchor)
            sd2 = patched_state_dict(sd, p, sep)
            with torch.no_grad():
                io_f, _ = om.spp_forward(sd2, x, C.SPP_ANCHORS, 80)
            io_b, _ = run_policy(om.spp_forward, sd2, x, C.SPP_ANCHORS, 80, policy="bf16")
            df, _ = onms.non_max_suppression(io_f.numpy().copy(), sep["conf_thres"], sep["nms_thres"])
            db, _ = onms.non_max_suppression(io_b.numpy().copy(), sep["conf_thres"], sep["nms_thres"])
            score = io_f.numpy()[0, :, 4] * io_f.numpy()[0, :, 5:].max(1)
            print(f"patch seed {seed}, {per_anchor} cells per anchor: {0 if df[0] is None else len(df[0])} fp32 / "
                  f"{0 if db[0] is None else len(db[0])} bf16-policy detections, strict share {strict_share(df[0], db[0]):.3f} / "
                  f"{strict_share(db[0], df[0]):.3f}, candidates in (0.4, 0.6): {int(((score > 0.4) & (score < 0.6)).sum())}", flush=True)


if __name__ == "__main__":
    main()
