"""Host mirror of reference models/yolo_layer.py.

``Upsample`` and ``Concat`` exist so that code written against the reference
module tree still finds them, but on the HIP path they never run as separate
kernels: the planner (pytorch_yolo_amd/engine.py) folds them into conv stores.
``YOLOLayer`` keeps the reference's attributes (anchors, anchor_vec, stride,
n_grids ... — yolo_layer.py:30-45,101-111) in sync with the last forward; the
arithmetic itself is ``yolo_decode_fwd`` (csrc/pointwise.hip).
"""
from __future__ import annotations

import torch
from torch import nn


class Upsample(nn.Module):
    """Marker for nearest x2 upsampling (reference yolo_layer.py:6-13)."""

    def __init__(self, scale_factor=1, mode="nearest"):
        super().__init__()
        if scale_factor != 2 or mode != "nearest":
            raise NotImplementedError("the HIP path implements nearest x2 upsampling only")
        self.scale_factor, self.mode = scale_factor, mode


class Concat(nn.Module):
    """Marker for channel concatenation (reference yolo_layer.py:16-22)."""

    def __init__(self, dim=0):
        super().__init__()
        if dim != 1:
            raise NotImplementedError("the HIP path concatenates along channels only")
        self.dim = dim


class YOLOLayer(nn.Module):
    def __init__(self, anchors, nc, all_anchors, onnx=False, in_tensor=None, img_size=None):
        super().__init__()
        if onnx:
            raise NotImplementedError("the ONNX/OpenVINO export branch (yolo_layer.py:47-55,73-88) is out of scope")
        self.anchors = torch.tensor(anchors, dtype=torch.float32)
        self.anchors_px = [(float(a), float(b)) for a, b in anchors]
        self.n_anchors = len(anchors)
        self.n_classes = nc
        self.all_anchors = all_anchors
        self.onnx = False
        self.n_x_grids = self.n_y_grids = 0
        self.img_size = self.stride = 0
        self.n_grids = self.grid_xy = self.anchor_vec = self.anchor_wh = 0

    def _sync_grid_attrs(self, ny, nx, img_size, device):
        """create_grids (yolo_layer.py:101-111) — attribute bookkeeping only."""
        if (self.n_x_grids, self.n_y_grids) == (nx, ny) and self.img_size == img_size:
            return
        self.img_size, self.n_x_grids, self.n_y_grids = img_size, nx, ny
        self.stride = img_size / max(nx, ny)
        yv, xv = torch.meshgrid(torch.arange(ny), torch.arange(nx), indexing="ij")
        self.grid_xy = torch.stack((xv, yv), 2).float().view(1, 1, ny, nx, 2).to(device)
        self.anchor_vec = self.anchors.to(device) / self.stride
        self.anchor_wh = self.anchor_vec.view(1, self.n_anchors, 1, 1, 2)
        self.n_grids = torch.tensor((nx, ny), dtype=torch.float32, device=device)

    def forward(self, p, img_size):
        """p: raw head [bs, na*(5+nc), ny, nx] float32 (NCHW, like the reference).
        Returns (io, p_permuted) through the HIP decode kernel."""
        from .. import kernels as K
        bs, ch, ny, nx = p.shape
        no = self.n_classes + 5
        if ch != self.n_anchors * no:
            raise RuntimeError(f"head has {ch} channels, expected {self.n_anchors * no}")
        if self.training:
            raise NotImplementedError("training forward is outside the inference hot path")
        self._sync_grid_attrs(ny, nx, img_size, p.device)
        head = p.float().permute(0, 2, 3, 1).contiguous()
        io = torch.empty((bs, self.n_anchors * ny * nx, no), dtype=torch.float32, device=p.device)
        pp = torch.empty((bs, self.n_anchors, ny, nx, no), dtype=torch.float32, device=p.device)
        K.decode(head, self.anchors_px, self.n_classes, self.stride, io, 0, pp)
        return io, pp
