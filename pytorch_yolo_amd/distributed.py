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


def gather_detections(dets: torch.Tensor, count: torch.Tensor, group=None, gather_cap: int = 1024, equal_shards: bool = False,
                      sizes=None):
    """All-gather per-rank NMS outputs.

    dets [bs_local, cap, 7] float32, count [bs_local] int32 (device or CPU tensors, any backend).
    Ranks may hold different bs_local (``shard_bounds`` spreads a remainder over the first ranks); every shard is zero-padded to the
    largest one for the fixed-size all-gather.  The shard sizes come from ``sizes`` (a list of world ints, e.g. from
    ``shard_bounds``: nothing extra on the wire), or ``equal_shards=True`` (every rank holds this rank's bs_local); only when
    neither is given are they exchanged first - one more collective and a blocking device-to-host copy per call.
    Returns (all_dets [sum bs_local, gather_cap, 7], all_count [sum bs_local]) on every rank, images in global
    (rank-major) order."""
    world = dist.get_world_size(group)
    bs, cap, _ = dets.shape
    g = min(gather_cap, cap)
    if sizes is not None:
        sizes = [int(v) for v in sizes]
        if len(sizes) != world or sizes[dist.get_rank(group)] != bs:
            raise RuntimeError(f"gather_detections: sizes {sizes} do not describe {world} ranks with {bs} images on this one")
    elif equal_shards:
        sizes = [bs] * world
    else:
        mine = torch.tensor([bs], dtype=torch.int64, device=count.device)
        every = torch.empty((world,), dtype=torch.int64, device=count.device)
        dist.all_gather_into_tensor(every, mine, group=group)
        sizes = every.tolist()
    top = max(sizes)
    send = dets[:, :g].contiguous()
    send_count = count.contiguous()
    if bs < top:                                   # pad this shard: fixed-size buffers on the wire
        send = torch.cat([send, send.new_zeros((top - bs, g, 7))], 0)
        send_count = torch.cat([send_count, send_count.new_zeros((top - bs,))], 0)
    all_dets = torch.empty((world * top, g, 7), dtype=dets.dtype, device=dets.device)
    all_count = torch.empty((world * top,), dtype=count.dtype, device=count.device)
    dist.all_gather_into_tensor(all_count, send_count, group=group)
    dist.all_gather_into_tensor(all_dets, send, group=group)
    if any(sz != top for sz in sizes):             # drop the pad rows, keep rank-major image order
        keep = torch.cat([torch.arange(r * top, r * top + sz) for r, sz in enumerate(sizes)]).to(all_count.device)
        all_dets, all_count = all_dets.index_select(0, keep.to(all_dets.device)), all_count.index_select(0, keep)
    return all_dets, all_count


class PipelinedGather:
    """The same all-gather, taken off the critical path of a stream of batches (bench.py, serving loops).

    ``detect`` runs S free-running sub-batch pipelines per GPU (engine.StreamedPlan); joining them every batch
    just to exchange <1 MB would re-introduce the tile-quantisation tails the streams exist to hide.  Instead:
      * right after its NMS each sub-batch stream copies ITS rows into a staging slot (``stage``, pass it as
        ``after_nms=`` of ``launch_detect``) and records an event — no cross-stream dependency;
      * ``exchange()`` makes a dedicated side stream wait for those events and issues the two all-gathers there;
        the compute streams never wait for it.
    Two slots alternate, so batch i+1 stages while batch i is on the wire; a slot is rewritten only after its
    previous exchange completed (an event the staging copy waits for — two batches later, never a stall in
    practice).  ``exchange`` returns (all_dets, all_count, done_event); consume them after ``done_event``."""

    def __init__(self, bs_local: int, cap: int, n_streams: int, device, group=None, gather_cap: int = 1024):
        self.group, self.world = group, dist.get_world_size(group)
        self.g = min(gather_cap, cap)
        self.stream = torch.cuda.Stream(device=device)
        self.slots = []
        for _ in range(2):
            self.slots.append(dict(
                send=torch.zeros((bs_local, self.g, 7), dtype=torch.float32, device=device),
                send_count=torch.zeros((bs_local,), dtype=torch.int32, device=device),
                all_dets=torch.empty((self.world * bs_local, self.g, 7), dtype=torch.float32, device=device),
                all_count=torch.empty((self.world * bs_local,), dtype=torch.int32, device=device),
                staged=[torch.cuda.Event() for _ in range(n_streams)],
                done=torch.cuda.Event(), used=False))
        self._i = 0
        self._nms_out = None

    def begin(self, nms_out):
        """Select the slot of the coming batch; ``nms_out`` = the (dets, idx, count) buffers NMS writes."""
        self._slot = self.slots[self._i & 1]
        self._i += 1
        self._nms_out = nms_out
        return self.stage

    def stage(self, i: int, lo: int, hi: int):
        """Runs in sub-batch stream i (its current stream) right after its NMS launch."""
        sl = self._slot
        st = torch.cuda.current_stream()
        if sl["used"]:
            st.wait_event(sl["done"])                      # the slot's previous exchange has left the buffers
        sl["send"][lo:hi].copy_(self._nms_out[0][lo:hi, :self.g], non_blocking=True)
        sl["send_count"][lo:hi].copy_(self._nms_out[2][lo:hi], non_blocking=True)
        sl["staged"][i].record(st)

    def exchange(self):
        sl = self._slot
        for ev in sl["staged"]:
            self.stream.wait_event(ev)
        with torch.cuda.stream(self.stream):
            dist.all_gather_into_tensor(sl["all_count"], sl["send_count"], group=self.group)
            dist.all_gather_into_tensor(sl["all_dets"], sl["send"], group=self.group)
            sl["done"].record(self.stream)
        sl["used"] = True
        return sl["all_dets"], sl["all_count"], sl["done"]


def split_gathered(all_dets, all_count) -> List[Optional[torch.Tensor]]:
    counts = all_count.cpu().tolist()
    g = all_dets.shape[1]
    if max(counts, default=0) > g:
        raise RuntimeError(f"an image produced {max(counts)} detections, more than gather_cap={g}; raise gather_cap")
    return [all_dets[i, :n].clone() if n else None for i, n in enumerate(counts)]


def _local_detections(model, x_local, conf_thres, nms_thres):
    """(dets [bs, cap, 7], idx, count [bs]) of this rank's shard, by the path a lone ``detect()`` takes (round 5: the sharded API used to
    run ``model(x)`` + ``nms_raw`` - two joined half-batch lists with ``io`` materialised - and so missed what ``detect()`` gained in
    round 4): ONE whole-batch launch list + NMS behind one FFI call in the compact NMS form (engine.StreamedPlan.detect_step) where
    the plan allows it; the outputs are the step's own buffers, complete when this returns (the NMS runs on a side stream: the
    collectives that follow on the caller's stream must not start before it)."""
    x = x_local.float().contiguous()
    plan = model.plan_for(x)
    if hasattr(plan, "detect_step") and not model.training:
        with torch.cuda.device(x.device):
            fast = plan.detect_step(conf_thres, nms_thres)
            if fast is not None:
                fast.launch(x)
                fast.done.synchronize()
                return fast.out
    io, _ = model(x)
    return nms_raw(io, conf_thres, nms_thres)


def detect_sharded(model, x_local: torch.Tensor, conf_thres=0.5, nms_thres=0.5, group=None, gather_cap: int = 1024, total: int = None,
                   equal_shards: bool = False):
    """``detect()`` over a batch sharded by rank: returns the reference-style list for ALL images
    (global order) on every rank.  ``x_local`` is this rank's contiguous slice of the batch.
    ``total`` = the global batch size when the shards follow ``shard_bounds`` (then the shard sizes are known without an exchange),
    or ``equal_shards=True`` when every rank holds as many images as this one; with neither, the sizes are exchanged per call."""
    dets, _, count = _local_detections(model, x_local, conf_thres, nms_thres)
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        from .utils.utils import split_detections
        return split_detections(dets, _, count)
    sizes = None
    if total is not None:
        world = dist.get_world_size(group)
        sizes = [hi - lo for lo, hi in (shard_bounds(total, world, r) for r in range(world))]
    return split_gathered(*gather_detections(dets, count, group, gather_cap, equal_shards=equal_shards, sizes=sizes))
