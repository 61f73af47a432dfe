"""Darknet ``.weights`` reader / writer for the mirrored models.

File format (what the reference intends at models/yolo_base.py:152-265; its own implementation indexes a
non-subscriptable ConvBlock and cannot run):

    int32[5] header  (header[3] = images seen)
    then, per convolutional layer in darknet cfg order, float32 little-endian:
      conv + batch-norm block :  bn.bias, bn.weight, bn.running_mean, bn.running_var, conv.weight (OIHW)
      plain biased conv       :  conv.bias, conv.weight

Layer order: module definition order, except YOLOv3-tiny where the reference writes the /32 branch before
the /16 branch "as in the original cfg" (models/yolov3_tiny.py:57-65).  A file may stop early (backbone-only
checkpoints such as ``yolov3-tiny.conv.15``): loading then fills the leading layers and reports how many.
"""
from __future__ import annotations

import numpy as np
import torch
from torch import nn

from ..models.yolo_base import ConvBlock


def darknet_layers(model):
    """Conv layers of ``model`` in darknet order: ConvBlock (incl. ConvPoolBlock) or biased nn.Conv2d."""
    from ..models.yolov3_tiny import YOLOv3Tiny
    if isinstance(model, YOLOv3Tiny):
        roots = [model.sequence_1, model.sequence_2, model.sequence_branch2, model.sequence_branch1_1,
                 model.sequence_branch1_2]
    else:
        roots = [model]
    out, seen = [], set()

    def walk(m):
        if isinstance(m, ConvBlock):
            if id(m) not in seen:
                seen.add(id(m))
                out.append(m)
            return
        if isinstance(m, nn.Conv2d):
            if id(m) not in seen and m.bias is not None:
                seen.add(id(m))
                out.append(m)
            return
        for child in m.children():
            walk(child)

    for r in roots:
        walk(r)
    return out


def _block_tensors(layer):
    if isinstance(layer, ConvBlock):
        if layer.is_fused:
            raise RuntimeError("darknet IO needs the un-fused model (call it before fuse())")
        bn, conv = layer.sequence.batch_norm, layer.sequence.conv
        return [bn.bias, bn.weight, bn.running_mean, bn.running_var, conv.weight]
    return [layer.bias, layer.weight]


def save_darknet_weights(model, path):
    header = np.asarray(model.header_info, dtype=np.int32).copy()
    header[3] = int(model.seen)
    with open(path, "wb") as f:
        header.tofile(f)
        for layer in darknet_layers(model):
            for t in _block_tensors(layer):
                t.detach().float().cpu().numpy().astype("<f4").tofile(f)


def load_darknet_weights(model, path) -> int:
    """Returns the number of conv layers filled (all of them unless the file is a backbone-only checkpoint)."""
    with open(path, "rb") as f:
        header = np.fromfile(f, dtype=np.int32, count=5)
        data = np.fromfile(f, dtype="<f4")
    if header.size != 5:
        raise RuntimeError(f"{path}: truncated darknet header")
    model.header_info = header
    model.seen = header[3]
    ptr = filled = 0
    for layer in darknet_layers(model):
        tensors = _block_tensors(layer)
        need = sum(t.numel() for t in tensors)
        if ptr + need > data.size:
            if ptr == data.size:
                break                                   # clean stop at a layer boundary
            raise RuntimeError(f"{path}: {data.size - ptr} trailing floats do not fill the next layer ({need})")
        with torch.no_grad():
            for t in tensors:
                n = t.numel()
                t.copy_(torch.from_numpy(data[ptr:ptr + n].copy()).view_as(t))
                ptr += n
        filled += 1
    else:
        if ptr != data.size:
            raise RuntimeError(f"{path}: {data.size - ptr} unused floats after the last layer")
    model.invalidate()
    return filled
