"""Host mirror of the post-process part of reference utils/utils.py.

``non_max_suppression`` keeps the reference signature and return value
(utils.py:200-206,293) and runs the batched MERGE-NMS kernels (csrc/nms.hip).
"""
from __future__ import annotations

import torch

from .. import kernels as K

MIN_WH = 2.0               # reference utils.py:207
MAX_PER_CLASS = 100        # reference utils.py:247-250

_ws_cache = {}
MAX_CACHED_WORKSPACES = 16


def _workspace(device, bs, rows, nc, slot=None):
    """NMS scratch (keys / counters / staging), cached per (device, shape, STREAM): launches on one stream are ordered, so
    they can share it; two same-shape calls on different streams get different buffers and cannot race.  Bounded: the
    least recently used entry is dropped (its memory returns to torch's allocator once queued work has finished)."""
    key = (device, bs, rows, nc, torch.cuda.current_stream(device).cuda_stream)
    ws = _ws_cache.pop(key, None)
    if ws is None:
        ws = torch.empty(K.nms_workspace_bytes(bs, rows, nc), dtype=torch.uint8, device=device)
    _ws_cache[key] = ws                                  # most recently used last
    while len(_ws_cache) > MAX_CACHED_WORKSPACES:
        old = _ws_cache.pop(next(iter(_ws_cache)))
        old.record_stream(torch.cuda.current_stream(device))
    return ws


def nms_capacity(rows: int, nc: int) -> int:
    """Upper bound on kept rows per image: every class keeps at most MAX_PER_CLASS."""
    return max(1, min(rows, nc * MAX_PER_CLASS))


def nms_raw(prediction: torch.Tensor, conf_thres: float, nms_thres: float, inplace_conf: bool = False,
            out=None):
    """Launch only (no host sync).  Returns (dets [bs,cap,7], idx [bs,cap], count [bs]) device tensors;
    rows beyond count[b] are unspecified.  ``out`` lets a caller pass static buffers (graph capture)."""
    if prediction.dim() != 3:
        raise RuntimeError("prediction must be [bs, rows, 5+nc]")
    if not prediction.is_cuda:
        raise RuntimeError("pytorch_yolo_amd.non_max_suppression runs on a ROCm device only (no CPU fallback)")
    if prediction.dtype != torch.float32 or not prediction.is_contiguous():
        if inplace_conf:
            raise RuntimeError("inplace_conf needs a contiguous float32 prediction tensor")
        prediction = prediction.float().contiguous()
    bs, rows, no = prediction.shape
    nc = no - 5
    if out is None:
        cap = nms_capacity(rows, nc)
        dev = prediction.device
        out = (torch.empty((bs, cap, 7), dtype=torch.float32, device=dev),
               torch.empty((bs, cap), dtype=torch.int32, device=dev),
               torch.empty((bs,), dtype=torch.int32, device=dev))
    with torch.cuda.device(prediction.device):          # the library launches on the current device's stream
        K.nms_merge(prediction, conf_thres, nms_thres, out[0], out[1], out[2], _workspace(prediction.device, bs, rows, nc),
                    min_wh=MIN_WH, max_per_class=MAX_PER_CLASS, mutate_conf=inplace_conf)
    return out


def nms_launch(prediction, conf_thres, nms_thres, out, slot=0, inplace_conf=False):
    """Launch the NMS kernels on the current stream into ``out`` = (dets, idx, count); the workspace is private to the
    current stream (``slot`` is kept for callers of the old signature and ignored)."""
    bs, rows, no = prediction.shape
    with torch.cuda.device(prediction.device):
        K.nms_merge(prediction, conf_thres, nms_thres, out[0], out[1], out[2],
                    _workspace(prediction.device, bs, rows, no - 5), min_wh=MIN_WH, max_per_class=MAX_PER_CLASS,
                    mutate_conf=inplace_conf)
    return out


def split_detections(dets, idx, count, with_indices=False):
    """Device buffers -> the reference's ``list[Tensor[n,7] | None]`` (one D2H copy of the counts).  The kept rows of all
    images are gathered into ONE packed tensor (an index built on the host, one index_select) and handed out as its per-image
    slices: two launches per call instead of one clone per image, and the big output buffers are not kept alive."""
    counts = count.cpu().tolist()
    cap = dets.shape[1]
    for b, n in enumerate(counts):
        if n > cap:
            raise RuntimeError(f"image {b}: {n} detections exceed the output capacity {cap}")
    total = sum(counts)
    if total == 0:
        out = [None] * len(counts)
        return (out, list(out)) if with_indices else out
    import numpy as np
    rows = np.concatenate([np.arange(b * cap, b * cap + n, dtype=np.int64) for b, n in enumerate(counts) if n])
    rows = torch.from_numpy(rows).to(dets.device, non_blocking=True)
    packed = dets.reshape(-1, dets.shape[2]).index_select(0, rows)
    parts = iter(torch.split(packed, [n for n in counts if n]))
    out = [next(parts) if n else None for n in counts]
    if not with_indices:
        return out
    packed_idx = idx.reshape(-1).index_select(0, rows).long()
    parts = iter(torch.split(packed_idx, [n for n in counts if n]))
    return out, [next(parts) if n else None for n in counts]


def non_max_suppression(prediction, conf_thres=0.5, nms_thres=0.5, inplace_conf=False, with_indices=False):
    """Drop-in for reference ``non_max_suppression`` (utils.py:200-293, 'MERGE' style).

    Returns a list (len bs) of ``Tensor[n,7]`` = (x1, y1, x2, y2, conf, class_conf, class) sorted by
    conf descending, or ``None`` for an image with no detections.

    Differences, both opt-in to the reference behaviour:
      * the reference overwrites ``prediction[..., 4]`` with obj*class_conf (:213); here the input
        is left untouched unless ``inplace_conf=True``;
      * ``with_indices=True`` additionally returns, per image, the input row of each kept box.
    The reference's unstable argsort (:237,:291) is replaced by a total order
    (conf desc, then class, then input row) — identical whenever conf values are distinct.
    """
    return split_detections(*nms_raw(prediction, conf_thres, nms_thres, inplace_conf), with_indices=with_indices)


def xywh2xyxy(x):
    """Reference utils.py:46-60 (kept for API compatibility; the NMS kernel does this itself)."""
    y = torch.zeros_like(x)
    y[:, 0] = x[:, 0] - x[:, 2] / 2
    y[:, 1] = x[:, 1] - x[:, 3] / 2
    y[:, 2] = x[:, 0] + x[:, 2] / 2
    y[:, 3] = x[:, 1] + x[:, 3] / 2
    return y


def _scale_params(img1_shape, img0_shape, n_rows):
    """(pad_x, pad_y, gain, n_rows) exactly as the reference computes them in python floats (utils.py:298-300)."""
    gain = max(img1_shape) / max(img0_shape)
    return [(img1_shape[1] - img0_shape[1] * gain) / 2, (img1_shape[0] - img0_shape[0] * gain) / 2, gain, float(n_rows)]


def scale_coords(img1_shape, coords, img0_shape, round_result=False):
    """Drop-in for reference ``scale_coords`` (utils.py:296-303): rescale xyxy boxes (columns 0..3 of ``coords``
    [n, >=4], modified IN PLACE like the reference) from the network-input frame ``img1_shape`` (h, w) to the
    original image frame ``img0_shape``.  Runs ``yolo_scale_coords`` on the device."""
    if not coords.is_cuda:
        raise RuntimeError("pytorch_yolo_amd.scale_coords runs on a ROCm device only (no CPU fallback)")
    if coords.dim() != 2 or coords.shape[1] < 4 or coords.dtype != torch.float32 or not coords.is_contiguous():
        raise RuntimeError("scale_coords: coords must be a contiguous float32 [n, >=4] tensor")
    n = coords.shape[0]
    if n == 0:
        return coords
    params = torch.tensor([_scale_params(img1_shape, img0_shape, n)], dtype=torch.float32, device=coords.device)
    from .._lib import check, load
    with torch.cuda.device(coords.device):
        check(load().yolo_scale_coords(coords.data_ptr(), 1, n, coords.shape[1], params.data_ptr(), int(round_result),
                                       K.stream_ptr()), "scale_coords")
    return coords


def scale_detections(dets, count, img1_shape, img0_shapes, round_result=True):
    """Batched form used after ``nms_raw``: dets [bs,cap,7] in place, one original (h, w) per image; counts are
    read on the host (they are needed there anyway to split the list)."""
    counts = count.cpu().tolist()
    params = torch.tensor([_scale_params(img1_shape, s0, n) for s0, n in zip(img0_shapes, counts)],
                          dtype=torch.float32, device=dets.device)
    from .._lib import check, load
    with torch.cuda.device(dets.device):
        check(load().yolo_scale_coords(dets.data_ptr(), dets.shape[0], dets.shape[1], dets.shape[2], params.data_ptr(),
                                       int(round_result), K.stream_ptr()), "scale_coords")
    return dets


def _dict_from_results(data, targets, imgs_path, orig_shapes, cur_shape):
    """Drop-in for the reference's ``_dict_from_results`` (utils.py:306-327): the detections of one batch (the list
    ``non_max_suppression`` returns, rows x1 y1 x2 y2 conf cls_conf cls in the network frame ``cur_shape``) are mapped
    back to each original image (``scale_coords(...).round()``, on the device, in place like the reference) and
    appended to ``data[img_path]`` as {'type','score','left','top','right','bottom'} dicts."""
    for i, pred in enumerate(targets):
        if pred is None:
            continue
        scale_coords(cur_shape, pred, orig_shapes[i], round_result=True)
        rows = data.setdefault(imgs_path[i], [])
        for x1, y1, x2, y2, conf, _cls_conf, cls in pred.detach().cpu().numpy():
            rows.append({"type": int(cls), "score": float(conf), "left": int(x1), "top": int(y1), "right": int(x2),
                         "bottom": int(y2)})
    return data


def predict_dataset(model, batches, conf_thresh=0.1, nms_thresh=0.1):
    """The loop of the reference's ``test_model`` (utils.py:357-378) up to its prediction dictionary: for every
    ``(imgs, targets, imgs_path, shapes)`` batch (the reference dataset's collate format; ``targets`` is ignored):
    forward, NMS, back-projection.  The COCO scoring that follows in the reference (``coco_helper`` + pycocotools,
    utils.py:380-393) is outside this path and not installed here."""
    was_training = model.training
    model.eval()
    data = {}
    try:
        for imgs, _targets, imgs_path, shapes in batches:
            imgs = imgs.to(next(model.parameters()).device)
            with torch.no_grad():
                det = model.detect(imgs, conf_thresh, nms_thresh)
            _dict_from_results(data, det, imgs_path, shapes, tuple(imgs.shape[-2:]))
    finally:
        if was_training:
            model.train()
    return data
