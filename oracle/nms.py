"""Oracle for the reference's post-process.  TEST INFRASTRUCTURE.

numpy float32 restatement of
/root/reference/pytorch_yolo/utils/utils.py:200-293 (``non_max_suppression``,
hard-coded 'MERGE' style, :240,:266-275) with ``xywh2xyxy`` (:46-60) and
``bbox_iou`` (:63-96).  Every arithmetic step is a single IEEE fp32 operation
in the reference's order, so a device kernel that avoids FMA contraction can
match it bit for bit.

Where the reference is under-specified the oracle fixes a rule and says so:

* ``argsort`` at :237 and :291 is unstable; the oracle sorts by
  (conf descending, earlier row first).  Golden inputs are tie-free.
* ``torch.max(1)`` (:212) tie -> lowest class index.
* The MERGE sums (:273-274) are accumulated sequentially in candidate order
  (torch's reduction order is an implementation detail; differences are
  <= a few ulp and the golden comparison allows for that on box coordinates
  only — kept-index sets, scores and classes are compared exactly).

Besides the reference's return value the oracle reports, for every output row,
the index of the input row (candidate) that was the pivot of its merge group —
the "kept-index set" of BASELINE.json's parity bar.
"""
from __future__ import annotations

import numpy as np

F32 = np.float32
MIN_WH = F32(2.0)           # utils.py:207
MAX_PER_CLASS = 100         # utils.py:247-250
AREA_EPS = F32(1e-16)       # utils.py:93


def _iou_1_to_n(b, boxes):
    """bbox_iou(box1, box2, x1y1x2y2=True) — utils.py:63-96, fp32 op-for-op."""
    iw = np.minimum(b[2], boxes[:, 2]) - np.maximum(b[0], boxes[:, 0])
    ih = np.minimum(b[3], boxes[:, 3]) - np.maximum(b[1], boxes[:, 1])
    inter = np.maximum(iw, F32(0)) * np.maximum(ih, F32(0))
    area1 = (b[2] - b[0]) * (b[3] - b[1]) + AREA_EPS
    area2 = (boxes[:, 2] - boxes[:, 0]) * (boxes[:, 3] - boxes[:, 1])
    union = (area1 + area2) - inter
    with np.errstate(divide="ignore", invalid="ignore"):
        return inter / union


def nms_image(pred, conf_thres, nms_thres, mutate=True):
    """One image: pred [N, 5+nc] float32 (xywh, obj, cls...).

    Returns (dets [n,7] float32, kept_idx [n] int64) or (None, None).
    With ``mutate`` the input's column 4 is overwritten with obj*max_cls like
    the reference does at utils.py:213.
    """
    assert pred.dtype == np.float32 and pred.ndim == 2
    conf_thres = F32(conf_thres)
    nms_thres = F32(nms_thres)
    cls = pred[:, 5:]
    class_pred = np.argmax(cls, axis=1)                      # first max on ties
    class_conf = cls[np.arange(len(cls)), class_pred]
    conf = pred[:, 4] * class_conf                           # :213
    if mutate:
        pred[:, 4] = conf
    row = pred if mutate else np.concatenate([pred[:, :4], conf[:, None], pred[:, 5:]], 1)
    keep = conf > conf_thres                                 # :216
    keep &= (pred[:, 2] > MIN_WH) & (pred[:, 3] > MIN_WH)    # :217
    keep &= np.isfinite(row).all(1)                          # :218
    idx = np.nonzero(keep)[0]
    if idx.size == 0:
        return None, None
    x, y, w, h = (pred[idx, k] for k in range(4))
    half = F32(2)
    boxes = np.stack([x - w / half, y - h / half, x + w / half, y + h / half], 1)  # :57-60
    conf = conf[idx]
    cconf = class_conf[idx]
    cpred = class_pred[idx]

    order = np.argsort(-conf, kind="stable")                 # :237 (stable rule)
    idx, boxes, conf, cconf, cpred = idx[order], boxes[order], conf[order], cconf[order], cpred[order]

    out_rows, out_idx = [], []
    for c in np.unique(cpred):                               # ascending, :241
        sel = np.nonzero(cpred == c)[0][:MAX_PER_CLASS]      # :242,:247-250
        b, s, cc, ii = boxes[sel].copy(), conf[sel], cconf[sel], idx[sel]
        alive = np.ones(len(sel), dtype=bool)
        while alive.any():                                   # :267
            live = np.nonzero(alive)[0]
            p = live[0]
            if live.size == 1:                               # :268-270 (kept unmerged)
                out_rows.append(np.array([*b[p], s[p], cc[p], F32(c)], dtype=F32))
                out_idx.append(ii[p])
                break
            hit = _iou_1_to_n(b[p], b[live]) > nms_thres     # :271 (includes the pivot itself)
            grp = live[hit]
            wsum = F32(0)
            acc = np.zeros(4, dtype=F32)
            for g in grp:                                    # :272-273, sequential fp32
                wsum = F32(wsum + s[g])
                acc = (acc + s[g] * b[g]).astype(F32)
            with np.errstate(divide="ignore", invalid="ignore"):
                merged = (acc / wsum).astype(F32)
            out_rows.append(np.array([*merged, s[p], cc[p], F32(c)], dtype=F32))  # :274
            out_idx.append(ii[p])
            if not hit.any():
                raise RuntimeError("pivot does not overlap itself (nms_thres >= 1 or degenerate box): "
                                   "the reference loops forever here")
            alive[grp] = False                               # :275
    dets = np.stack(out_rows).astype(F32)
    kept = np.asarray(out_idx, dtype=np.int64)
    final = np.argsort(-dets[:, 4], kind="stable")           # :291 (stable rule)
    return dets[final], kept[final]


def non_max_suppression(prediction, conf_thres=0.5, nms_thres=0.5, mutate=True):
    """Batch wrapper with the reference's signature/return (utils.py:200-206,293):
    list (len bs) of float32 [n,7] arrays or None, plus the kept-index lists."""
    dets, kept = [], []
    for pred in prediction:
        d, k = nms_image(pred, conf_thres, nms_thres, mutate=mutate)
        dets.append(d)
        kept.append(k)
    return dets, kept


def scale_coords(img1_shape, coords, img0_shape):
    """scale_coords — utils.py:296-303, fp32 op-for-op (python-float gain / pads cast to fp32 at the tensor op)."""
    gain = max(img1_shape) / max(img0_shape)
    out = coords.astype(F32).copy()
    out[:, [0, 2]] -= F32((img1_shape[1] - img0_shape[1] * gain) / 2)
    out[:, [1, 3]] -= F32((img1_shape[0] - img0_shape[0] * gain) / 2)
    out[:, :4] /= F32(gain)
    out[:, :4] = np.maximum(out[:, :4], F32(0))
    return out
