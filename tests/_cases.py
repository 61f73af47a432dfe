"""Shared case definitions for the golden-vector script and the parity tests.

Everything is derived from seeds (numpy PCG64), so the committed fixtures under
tests/golden/ only hold *outputs* of the reference; inputs and weights are
regenerated identically here, in the build container and on the GPU box.
"""
from __future__ import annotations

import numpy as np

SPP_ANCHORS = (((10., 13.), (16., 30.), (33., 23.)),
               ((30., 61.), (62., 45.), (59., 119.)),
               ((116., 90.), (156., 198.), (373., 326.)))
TINY_ANCHORS = (((10., 14.), (23., 27.), (37., 58.)),
                ((81., 82.), (135., 169.), (344., 319.)))

# name -> (family, ctor kwargs, batch, H, W, weight seed, image seed)
MODEL_CASES = {
    "tiny_small":     ("tiny", dict(n_class=3, kernels_divider=8, anchors=TINY_ANCHORS), 2, 64, 96, 11, 21),
    "tiny_nc1":       ("tiny", dict(n_class=1, kernels_divider=8, anchors=TINY_ANCHORS), 1, 64, 64, 12, 22),
    "tiny_kd2_nc80":  ("tiny", dict(n_class=80, kernels_divider=2, anchors=TINY_ANCHORS), 1, 96, 128, 13, 23),
    "spp_small":      ("spp", dict(n_class=3, kernels_divider=4, anchors=SPP_ANCHORS), 2, 64, 64, 14, 24),
    "spp_kd2_nc80":   ("spp", dict(n_class=80, kernels_divider=2, anchors=SPP_ANCHORS), 1, 96, 64, 15, 25),
    # the two model families that reuse the same blocks
    "yolov3_small":   ("yolov3", dict(n_class=3, kernels_divider=4, anchors=SPP_ANCHORS), 2, 64, 96, 16, 26),
    "lite_small":     ("lite", dict(n_class=5, kernels_divider=2, anchors=SPP_ANCHORS), 2, 64, 64, 17, 27),
    "lite_nc80":      ("lite", dict(n_class=80, kernels_divider=1, anchors=SPP_ANCHORS), 1, 128, 96, 18, 28),
}

# full-size configs of BASELINE.json: only sampled rows + column sums are stored
FULL_CASES = {
    "tiny_416": ("tiny", dict(n_class=80, kernels_divider=1, anchors=TINY_ANCHORS), 1, 416, 416, 1234, 0),
    "spp_640":  ("spp", dict(n_class=80, kernels_divider=1, anchors=SPP_ANCHORS), 1, 640, 640, 1234, 0),
}
FULL_SAMPLE_ROWS = 256
NMS_FULL = dict(conf_thres=0.1, nms_thres=0.5)


def sample_rows(n_rows: int, seed: int = 7) -> np.ndarray:
    rng = np.random.default_rng(seed)
    return np.sort(rng.choice(n_rows, size=min(FULL_SAMPLE_ROWS, n_rows), replace=False))


# name -> (seed, bs, rows, n_class, conf_thres, nms_thres)
NMS_CASES = {
    "nms_small_nc2":   (31, 1, 64, 2, 0.2, 0.5),
    "nms_mid_nc80":    (32, 2, 3000, 80, 0.1, 0.5),
    "nms_dense_nc3":   (33, 2, 2500, 3, 0.05, 0.45),     # >100 per class: exercises the cap (utils.py:247-250)
    "nms_nc1":         (34, 1, 800, 1, 0.3, 0.6),
    "nms_none_pass":   (35, 3, 200, 5, 0.3, 0.5),        # image 1 has no survivors -> None (utils.py:223-224)
    "nms_lowthres":    (36, 1, 1500, 20, 0.1, 0.1),      # test_model defaults (utils.py:359)
}

# SURVEY.md §4 known-answer test (captured from the reference during the survey)
NMS_KAT_ROWS = np.array([
    [50, 50, 20, 20, .9, .9, .1],
    [52, 50, 20, 20, .8, .8, .1],
    [100, 100, 30, 30, .7, .1, .9],
    [50, 52, 20, 20, .6, .7, .2],
    [200, 200, 1, 50, .99, .9, .1],
    [150, 150, 40, 40, .3, .5, .4]], dtype=np.float32)
NMS_KAT_ARGS = dict(conf_thres=0.2, nms_thres=0.5)
NMS_KAT_EXPECT = np.array([[40.6845, 40.4492, 60.6845, 60.4492, 0.81, 0.9, 0],
                           [85, 85, 115, 115, 0.63, 0.9, 1]], dtype=np.float32)
NMS_KAT_COL4 = np.array([.81, .64, .63, .42, .891, .15], dtype=np.float32)


def synth_predictions(seed: int, bs: int, rows: int, n_class: int) -> np.ndarray:
    """Decoded-head-like predictions [bs, rows, 5+nc] float32: boxes clustered
    around a few 'objects' (so that MERGE has groups to merge), some tiny boxes
    (w/h <= 2 filter), a few non-finite rows, scores with a long tail.  Scores
    are continuous, so conf ties do not occur."""
    rng = np.random.default_rng(seed)
    out = np.empty((bs, rows, 5 + n_class), dtype=np.float32)
    for b in range(bs):
        n_obj = int(rng.integers(3, 12))
        centres = rng.uniform(40, 600, (n_obj, 2))
        sizes = rng.uniform(15, 200, (n_obj, 2))
        owner = rng.integers(0, n_obj, rows)
        xy = centres[owner] + rng.normal(0, 0.12, (rows, 2)) * sizes[owner]
        wh = sizes[owner] * np.exp(rng.normal(0, 0.2, (rows, 2)))
        small = rng.random(rows) < 0.03
        wh[small] = rng.uniform(0.1, 2.5, (int(small.sum()), 2))
        obj = 1.0 / (1.0 + np.exp(-(rng.normal(-1.0, 2.0, rows))))
        obj_cls = rng.integers(0, n_class, n_obj)
        logits = rng.normal(-3.0, 1.5, (rows, n_class))
        logits[np.arange(rows), obj_cls[owner]] += rng.normal(4.0, 1.5, rows)
        cls = 1.0 / (1.0 + np.exp(-logits))
        out[b, :, 0:2] = xy
        out[b, :, 2:4] = wh
        out[b, :, 4] = obj
        out[b, :, 5:] = cls
        bad = rng.choice(rows, size=max(1, rows // 200), replace=False)
        out[b, bad[0::3], 2] = np.inf
        out[b, bad[1::3], 5 + int(rng.integers(0, n_class))] = np.nan
        out[b, bad[2::3], 0] = -np.inf
    return out


def nms_case_inputs(name: str):
    """(prediction [bs,rows,5+nc] float32, conf_thres, nms_thres) of an NMS case."""
    seed, bs, rows, n_class, conf, iou = NMS_CASES[name]
    pred = synth_predictions(seed, bs, rows, n_class)
    if name == "nms_none_pass":
        pred[1, :, 4] *= np.float32(1e-3)
    return pred, conf, iou


# scale_coords (utils.py:296-303): (network input (h, w), original image (h, w))
SCALE_CASES = [((416, 416), (480, 640)), ((640, 640), (1080, 1920)), ((320, 416), (375, 500)), ((640, 640), (333, 500))]


def scale_coords_boxes(n: int = 200, seed: int = 41) -> np.ndarray:
    rng = np.random.default_rng(seed)
    b = rng.uniform(-20, 660, (n, 4)).astype(np.float32)
    b[:, 2:] = b[:, :2] + rng.uniform(1, 300, (n, 2)).astype(np.float32)
    return b


def results_case(seed: int = 43):
    """Input of the reference's ``_dict_from_results`` (utils/utils.py:306-327): three images, the middle one without
    detections; rows (x1, y1, x2, y2, conf, cls_conf, cls) in a 416 x 416 network frame."""
    rng = np.random.default_rng(seed)
    dets = []
    for n in (7, 0, 12):
        if n == 0:
            dets.append(None)
            continue
        b = rng.uniform(-10, 400, (n, 4)).astype(np.float32)
        b[:, 2:] = b[:, :2] + rng.uniform(2, 200, (n, 2)).astype(np.float32)
        rest = np.stack([rng.uniform(0.1, 1, n), rng.uniform(0.1, 1, n), rng.integers(0, 80, n)], 1).astype(np.float32)
        dets.append(np.concatenate([b, rest], 1))
    return dets, ["a.jpg", "b.jpg", "a.jpg"], [(480, 640), (375, 500), (1080, 1920)], (416, 416)
