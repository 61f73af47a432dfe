"""Batch-sharded multi-GPU detect: one process per GPU, weights replicated, images independent
through conv / decode / NMS (reference NMS is per image, utils/utils.py:210), ONE exchange at the end.

The only collective is an all-gather (RCCL over xGMI when the backend is "nccl") of fixed-size
per-rank buffers: counts int32[bs_local] and dets f32[bs_local, gather_cap, 7].  The payload is
under 1 MB per rank, i.e. latency-bound — no bucketing, no overlap machinery.
"""
from __future__ import annotations

from typing import List, Optional

import torch
import torch.distributed as dist

from .utils.utils import nms_raw


def shard_bounds(total: int, world: int, rank: int):
    """Contiguous image range [lo, hi) of ``rank`` (remainder spread over the first ranks)."""
    base, rem = divmod(total, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def gather_detections(dets: torch.Tensor, count: torch.Tensor, group=None, gather_cap: int = 1024):
    """All-gather per-rank NMS outputs.

    dets [bs_local, cap, 7] float32, count [bs_local] int32 (device or CPU tensors, any backend).
    Every rank must hold the same bs_local.  Returns (all_dets [world*bs_local, gather_cap, 7],
    all_count [world*bs_local]) on every rank, images in global (rank-major) order."""
    world = dist.get_world_size(group)
    bs, cap, _ = dets.shape
    g = min(gather_cap, cap)
    send = dets[:, :g].contiguous()
    all_dets = torch.empty((world * bs, g, 7), dtype=dets.dtype, device=dets.device)
    all_count = torch.empty((world * bs,), dtype=count.dtype, device=count.device)
    dist.all_gather_into_tensor(all_count, count.contiguous(), group=group)
    dist.all_gather_into_tensor(all_dets, send, group=group)
    return all_dets, all_count


def split_gathered(all_dets, all_count) -> List[Optional[torch.Tensor]]:
    counts = all_count.cpu().tolist()
    g = all_dets.shape[1]
    if max(counts, default=0) > g:
        raise RuntimeError(f"an image produced {max(counts)} detections, more than gather_cap={g}; raise gather_cap")
    return [all_dets[i, :n].clone() if n else None for i, n in enumerate(counts)]


def detect_sharded(model, x_local: torch.Tensor, conf_thres=0.5, nms_thres=0.5, group=None, gather_cap: int = 1024):
    """``detect()`` over a batch sharded by rank: returns the reference-style list for ALL images
    (global order) on every rank.  ``x_local`` is this rank's contiguous slice of the batch."""
    io, _ = model(x_local)
    dets, _, count = nms_raw(io, conf_thres, nms_thres)
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        from .utils.utils import split_detections
        return split_detections(dets, _, count)
    return split_gathered(*gather_detections(dets, count, group, gather_cap))
