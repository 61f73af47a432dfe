"""Host mirror of the reference's inference pre-processing, run on the device:

* ``LetterBox`` — reference utils/augs.py:7-94 (``cv2.resize`` INTER_AREA by one ratio, replicate border to a
  /32 rectangle or a fixed shape); geometry = ``update_params`` (:24-63) line by line;
* ``convert_img_for_net`` / ``equalize_shapes`` — reference utils/dataset_csv.py:79-87,146-171;
* ``preprocess_batch`` — the three fused: one ``yolo_letterbox_u8_fwd`` launch per image writes its slice of the
  float32 NCHW batch directly (no resized / padded / float intermediates).

Inputs are decoded uint8 HWC images already on the device (decoding itself — ``cv.imread`` at
dataset_csv.py:67 — is out of scope).  No CPU fallback.  The resize arithmetic cannot be pinned against cv2 here
(not installed): see oracle/preprocess.py for what is restated from where.
"""
from __future__ import annotations

import ctypes as C
import math
from typing import List, Sequence

import torch

from .. import kernels as K
from .._lib import check, load


def _ceil32(v: float) -> int:
    return int(math.ceil(v / 32)) * 32


def letterbox_params(rows: int, cols: int, new_shape=416) -> dict:
    """``LetterBox.update_params`` (augs.py:24-63).  ``new_shape``: int (-> /32 rectangle) or (h, w)."""
    h, w = rows, cols
    if isinstance(new_shape, int):
        r = h / w
        shape = [1, 1]
        if r < 1:
            shape = [r, 1]
        elif r > 1:
            shape = [1, 1 / r]
        target_shape = [_ceil32(shape[0] * new_shape), _ceil32(shape[1] * new_shape)]       # :37-38
    else:
        target_shape = [int(new_shape[0]), int(new_shape[1])]
    ratio = max(target_shape) / max(h, w)                                                   # :42
    target_height, target_width = target_shape
    resize_height = int(round(h * ratio))                                                   # :45-46
    resize_width = int(round(w * ratio))
    if isinstance(new_shape, int):                                                          # :49-54
        pad_left = ((target_width - resize_width) % 32) / 2
        pad_top = ((target_height - resize_height) % 32) / 2
    else:
        pad_left = (target_width - resize_width) / 2
        pad_top = (target_height - resize_height) / 2
    pad_left, pad_top = int(pad_left), int(pad_top)                                         # :56-57
    return dict(pad_left=pad_left, pad_top=pad_top, pad_right=target_width - resize_width - pad_left,
                pad_bottom=target_height - resize_height - pad_top, resize_ratio=ratio,
                resize_height=resize_height, resize_width=resize_width,
                target_height=target_height, target_width=target_width)


def _check_image(img: torch.Tensor):
    if not img.is_cuda:
        raise RuntimeError("pytorch_yolo_amd pre-processing runs on a ROCm device only (no CPU fallback)")
    if img.dtype != torch.uint8 or img.dim() != 3 or img.shape[2] > 4 or img.stride(2) != 1 or img.stride(1) != img.shape[2]:
        raise RuntimeError("image must be a uint8 HWC tensor (<= 4 interleaved channels, dense rows)")


def _launch(img, p, dst_u8=None, dst_f32=None, dst_hw=(0, 0), off=(0, 0), fill=0.5):
    h, w, c = img.shape
    th, tw = p["target_height"], p["target_width"]
    # the reference pads pad_right / pad_bottom explicitly; when the /32 modulo leaves the resized image larger than
    # the rectangle minus pads the geometry is inconsistent in the reference too (copyMakeBorder would throw)
    if p["pad_right"] < 0 or p["pad_bottom"] < 0:
        raise RuntimeError("letterbox: negative padding (the reference's copyMakeBorder rejects this shape as well)")
    check(load().yolo_letterbox_u8_fwd(C.c_void_p(img.data_ptr()), h, w, c, img.stride(0), float(p["resize_ratio"]),
                                       p["resize_height"], p["resize_width"], p["pad_top"], p["pad_left"], th, tw,
                                       None if dst_u8 is None else C.c_void_p(dst_u8.data_ptr()),
                                       None if dst_f32 is None else C.c_void_p(dst_f32.data_ptr()),
                                       dst_hw[0], dst_hw[1], off[0], off[1], float(fill), K.stream_ptr()), "letterbox")


class LetterBox:
    """Drop-in for the image part of the reference transform: ``LetterBox(new_shape)(image=img)["image"]`` is the
    letterboxed uint8 HWC image (augs.py:64-74).  ``params`` of the last call are kept for ``scale_coords``-style
    back-projection (pad_left / pad_top / resize_ratio)."""

    def __init__(self, new_shape=416):
        self.new_shape = new_shape
        self.params = None

    def __call__(self, image: torch.Tensor, **kw):
        _check_image(image)
        p = letterbox_params(image.shape[0], image.shape[1], self.new_shape)
        out = torch.empty((p["target_height"], p["target_width"], image.shape[2]), dtype=torch.uint8, device=image.device)
        _launch(image, p, dst_u8=out)
        self.params = p
        return dict(kw, image=out)


def convert_img_for_net(img: torch.Tensor) -> torch.Tensor:
    """``_convert_img_for_net`` (dataset_csv.py:79-87): uint8 HWC -> float32 CHW in [0, 1] (division by 255 in fp32)."""
    _check_image(img)
    return (img.permute(2, 0, 1).to(torch.float32) / 255.0).contiguous()


def equalize_offsets(shapes: Sequence[Sequence[int]]):
    """Placement rule of ``equalize_shapes`` (dataset_csv.py:146-163): (new_h, new_w, [(top, left), ...])."""
    new_h, new_w = max(s[0] for s in shapes), max(s[1] for s in shapes)
    offs = []
    for h, w in shapes:
        if h == new_h and w == new_w:
            offs.append((0, 0))
        else:
            offs.append((int(round((new_h - h) / 2 - 0.1)), int(round((new_w - w) / 2 - 0.1))))
    return new_h, new_w, offs


def preprocess_batch(images: List[torch.Tensor], new_shape=416, out: torch.Tensor = None):
    """Decoded uint8 HWC device images -> (float32 NCHW batch, per-image geometry): LetterBox + /255 + CHW +
    equalize_shapes in one launch per image.  ``out`` lets a serving loop reuse its batch buffer."""
    if not images:
        raise RuntimeError("preprocess_batch: empty batch")
    for img in images:
        _check_image(img)
    c = images[0].shape[2]
    params = [letterbox_params(i.shape[0], i.shape[1], new_shape) for i in images]
    H, W, offs = equalize_offsets([(p["target_height"], p["target_width"]) for p in params])
    n = len(images)
    if out is None:
        out = torch.empty((n, c, H, W), dtype=torch.float32, device=images[0].device)
    elif tuple(out.shape) != (n, c, H, W) or out.dtype != torch.float32 or not out.is_contiguous():
        raise RuntimeError(f"preprocess_batch: out must be a contiguous float32 [{n},{c},{H},{W}] tensor")
    for i, (img, p, off) in enumerate(zip(images, params, offs)):
        if img.shape[2] != c:
            raise RuntimeError("preprocess_batch: images differ in channel count")
        _launch(img, p, dst_f32=out[i], dst_hw=(H, W), off=off, fill=0.5)
        p["off_y"], p["off_x"] = off
    return out, params
