"""Exploration (CPU): which (head statistics, image seed, conf threshold) give YOLOv3-SPP 640 detections that the bf16 rounding
points can carry strictly - every fp32 detection has a bf16 partner (same class, IoU >= 0.9, |dconf| <= 0.03) and vice versa?
Uses the fp32 oracle and its bf16-policy re-run (oracle/policy.py); no GPU, no reference import."""
import sys, os, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import _cases as C
from oracle import models as om, nms as onms
from oracle.policy import run_policy
from pytorch_yolo_amd import YOLOv3SPP
from pytorch_yolo_amd.utils import synthetic as S


def iou(a, b):
    x1, y1, x2, y2 = np.maximum(a[0], b[0]), np.maximum(a[1], b[1]), np.minimum(a[2], b[2]), np.minimum(a[3], b[3])
    inter = max(0.0, x2 - x1) * max(0.0, y2 - y1)
    return inter / ((a[2] - a[0]) * (a[3] - a[1]) + (b[2] - b[0]) * (b[3] - b[1]) - inter + 1e-16)


def paired(da, db, iou_min=0.9, dconf=0.03):
    """share of detections in da with a partner in db"""
    if da is None:
        return 1.0, 0
    if db is None:
        return 0.0, len(da)
    ok = 0
    for r in da:
        ok += any(int(r[6]) == int(q[6]) and abs(r[4] - q[4]) <= dconf and iou(r[:4], q[:4]) >= iou_min for q in db)
    return ok / len(da), len(da)


def main():
    torch.set_num_threads(8)
    tmpl = YOLOv3SPP(anchors=C.SPP_ANCHORS).state_dict()
    variants = {"o55c20": ((1.5, 0.3, 25.0, 15.0), (0.0, 0.0, -55.0, -20.0)),
                "o65c20": ((1.5, 0.3, 25.0, 15.0), (0.0, 0.0, -65.0, -20.0)),
                "o60c10": ((1.5, 0.3, 25.0, 15.0), (0.0, 0.0, -60.0, -10.0))}
    for vname, hs in variants.items():
        old = S.HEAD_STATS["bn_leaky"]
        if hs is not None:
            S.HEAD_STATS["bn_leaky"] = hs
        sd = S.synth_state_dict(tmpl, 1234, n_class=80)
        S.HEAD_STATS["bn_leaky"] = old
        for xseed in (0, 3):
            x = S.synth_images(1, 640, 640, xseed)
            t0 = time.time()
            with torch.no_grad():
                io_f, _ = om.spp_forward(sd, x, C.SPP_ANCHORS, 80)
            io_b, _ = run_policy(om.spp_forward, sd, x, C.SPP_ANCHORS, 80, policy="bf16")
            io_f, io_b = io_f.numpy(), io_b.numpy()
            sc = io_f[0, :, 4] * io_f[0, :, 5:].max(1)
            srt = np.sort(sc)[::-1]
            print(f"[{vname} x{xseed}] {time.time()-t0:.1f}s  rows>0.1: {(sc>0.1).sum()}  >0.3: {(sc>0.3).sum()}  >0.5: {(sc>0.5).sum()}  >0.7: {(sc>0.7).sum()} top {srt[:3]}")
            for thr in (0.2, 0.3, 0.4, 0.5, 0.6, 0.7, 0.8):
                df, _ = onms.non_max_suppression(io_f.copy(), thr, 0.5)
                db, _ = onms.non_max_suppression(io_b.copy(), thr, 0.5)
                a, na = paired(df[0], db[0])
                b, nb = paired(db[0], df[0])
                print(f"    thr {thr}: ref dets {na}, bf16 dets {nb}, strict share ref->bf16 {a:.3f}, bf16->ref {b:.3f}")


if __name__ == "__main__":
    main()
