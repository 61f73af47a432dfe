"""Oracle for the pre-processing in front of the hot path.  TEST INFRASTRUCTURE.

PARITY UNPINNED for the resize arithmetic: the reference calls ``cv2.resize(..., interpolation=cv2.INTER_AREA)``
and ``cv2.copyMakeBorder(..., cv2.BORDER_REPLICATE)`` through an albumentations transform
(/root/reference/pytorch_yolo/utils/augs.py:7-94); neither cv2 (opencv-python, unpinned in the reference's
requirements) nor albumentations is installed here, and the reference holds no fixture for this step.  What IS
restated from the reference's own source, line by line:

* the letterbox geometry — ``LetterBox.update_params``, augs.py:24-63;
* ``_convert_img_for_net`` — utils/dataset_csv.py:79-87 (float32, ``/= 255``, HWC -> CHW, channel order kept);
* ``equalize_shapes`` — utils/dataset_csv.py:146-171 (0.5 canvas, ``top = int(round(dh - 0.1))``).

The resize itself restates OpenCV's published INTER_AREA algorithm for down-scaling (``computeResizeAreaTab`` +
``ResizeArea_Invoker``, modules/imgproc/src/resize.cpp): per axis a table of (source index, weight) with the
1e-3 thresholds, x pass then y pass in float32, result rounded half-to-even to uint8 (``saturate_cast<uchar>``).
Documented deviations: OpenCV's integer-ratio fast path for 2x2 blocks rounds half up (differs by at most one
grey level on exact .5 sums); up-scaling (ratio > 1), where OpenCV switches to a bilinear variant, is restated as
that variant in float (OpenCV's uint8 path uses 11-bit fixed-point coefficients).
"""
from __future__ import annotations

import math

import numpy as np

F32 = np.float32


def letterbox_params(rows: int, cols: int, new_shape):
    """LetterBox.update_params (augs.py:24-63) -> dict(pad_left, pad_top, pad_right, pad_bottom, resize_ratio,
    resize_height, resize_width, target_height, target_width)."""
    h, w = rows, cols
    if isinstance(new_shape, int):                                   # rectangle (augs.py:29-38)
        r = h / w
        shape = [1, 1]
        if r < 1:
            shape = [r, 1]
        elif r > 1:
            shape = [1, 1 / r]
        target_shape = (np.ceil(np.array(shape) * new_shape / 32).astype(int) * 32).tolist()
    else:
        target_shape = list(new_shape)
    ratio = max(target_shape) / max(h, w)                            # :42
    target_height, target_width = target_shape
    resize_height = int(round(h * ratio))                            # :45-46 (python round: half to even)
    resize_width = int(round(w * ratio))
    if isinstance(new_shape, int):                                   # :49-54
        pad_left = np.mod(target_width - resize_width, 32) / 2
        pad_top = np.mod(target_height - resize_height, 32) / 2
    else:
        pad_left = (target_width - resize_width) / 2
        pad_top = (target_height - resize_height) / 2
    pad_left, pad_top = int(pad_left), int(pad_top)                  # :56-57
    return dict(pad_left=pad_left, pad_top=pad_top, pad_right=target_width - resize_width - pad_left,
                pad_bottom=target_height - resize_height - pad_top, resize_ratio=ratio,
                resize_height=resize_height, resize_width=resize_width,
                target_height=target_height, target_width=target_width)


def area_tab(ssize: int, dsize: int, scale: float):
    """computeResizeAreaTab: list per destination index of [(source index, weight float32), ...]."""
    tab = []
    for dx in range(dsize):
        fsx1 = dx * scale
        fsx2 = fsx1 + scale
        cell = min(scale, ssize - fsx1)
        sx1, sx2 = math.ceil(fsx1), math.floor(fsx2)
        sx2 = min(sx2, ssize - 1)
        sx1 = min(sx1, sx2)
        ent = []
        if sx1 - fsx1 > 1e-3:
            ent.append((sx1 - 1, F32((sx1 - fsx1) / cell)))
        for sx in range(sx1, sx2):
            ent.append((sx, F32(1.0 / cell)))
        if fsx2 - sx2 > 1e-3:
            ent.append((sx2, F32(min(min(fsx2 - sx2, 1.0), cell) / cell)))
        tab.append(ent)
    return tab


def linear_tab(ssize: int, dsize: int, scale: float):
    """INTER_AREA when enlarging (scale < 1): OpenCV's bilinear variant with area-style coefficients."""
    inv = 1.0 / scale
    tab = []
    for dx in range(dsize):
        sx = math.floor(dx * scale)
        fx = (dx + 1) - (sx + 1) * inv
        fx = 0.0 if fx <= 0 else fx - math.floor(fx)
        if sx < 0:
            sx, fx = 0, 0.0
        if sx >= ssize - 1:
            sx, fx = ssize - 1, 0.0
        s1 = min(sx + 1, ssize - 1)
        tab.append([(sx, F32(1.0 - fx)), (s1, F32(fx))])
    return tab


def resize_area(img: np.ndarray, ratio: float, out_h: int, out_w: int) -> np.ndarray:
    """cv2.resize(img, None, fx=ratio, fy=ratio, interpolation=INTER_AREA) for uint8 HWC (augs.py:66-67)."""
    assert img.dtype == np.uint8 and img.ndim == 3
    h, w, c = img.shape
    scale = 1.0 / ratio
    mk = area_tab if scale >= 1.0 else linear_tab
    xt, yt = mk(w, out_w, scale), mk(h, out_h, scale)
    out = np.zeros((out_h, out_w, c), dtype=np.uint8)
    src = img.astype(F32)
    for dy in range(out_h):
        for dx in range(out_w):
            tot = np.zeros(c, dtype=F32)
            for sy, beta in yt[dy]:
                row = np.zeros(c, dtype=F32)
                for sx, alpha in xt[dx]:
                    row = (row + src[sy, sx] * alpha).astype(F32)
                tot = (tot + row * beta).astype(F32)
            out[dy, dx] = np.clip(np.rint(tot), 0, 255).astype(np.uint8)       # saturate_cast<uchar>: half to even
    return out


def letterbox(img: np.ndarray, new_shape) -> tuple:
    """LetterBox.apply (augs.py:64-74): resized image inside a replicate border.  Returns (uint8 HWC, params)."""
    p = letterbox_params(img.shape[0], img.shape[1], new_shape)
    r = resize_area(img, p["resize_ratio"], p["resize_height"], p["resize_width"])
    out = np.pad(r, ((p["pad_top"], p["pad_bottom"]), (p["pad_left"], p["pad_right"]), (0, 0)), mode="edge")
    return out, p


def convert_img_for_net(img: np.ndarray) -> np.ndarray:
    """_convert_img_for_net (dataset_csv.py:79-87): float32 / 255, HWC -> CHW."""
    x = np.ascontiguousarray(img, dtype=F32)
    x /= F32(255)
    return np.transpose(x, (2, 0, 1))


def equalize_shapes(images):
    """equalize_shapes (dataset_csv.py:146-171) without labels: centre every CHW image on a 0.5 canvas of the
    batch's maximum height / width.  Returns (float32 [n,c,H,W], [(top, left), ...])."""
    new_h = max(i.shape[1] for i in images)
    new_w = max(i.shape[2] for i in images)
    out, offs = [], []
    for img in images:
        c, h, w = img.shape
        if h == new_h and w == new_w:
            out.append(img.astype(F32))
            offs.append((0, 0))
            continue
        dw, dh = (new_w - w) / 2, (new_h - h) / 2
        top, left = int(round(dh - 0.1)), int(round(dw - 0.1))
        canvas = np.full((c, new_h, new_w), 0.5, dtype=F32)
        canvas[:, top:top + h, left:left + w] = img
        out.append(canvas)
        offs.append((top, left))
    return np.stack(out, 0), offs


def preprocess_batch(images, new_shape):
    """uint8 HWC images -> the float32 NCHW batch the reference's loader hands to the model."""
    lb = [letterbox(i, new_shape) for i in images]
    x, offs = equalize_shapes([convert_img_for_net(i) for i, _ in lb])
    return x, [dict(p, off_y=o[0], off_x=o[1]) for (_, p), o in zip(lb, offs)]
