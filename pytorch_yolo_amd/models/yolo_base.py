"""Host mirror of reference models/yolo_base.py for the HIP path.

The classes keep the reference's constructor arguments, attribute names and
``state_dict`` layout (``<block>.sequence.conv.weight``,
``<block>.sequence.batch_norm.*``; after ``fuse()``: ``<block>.sequence.0.*``)
so checkpoints move between the two unchanged.  The ``nn.Conv2d`` /
``nn.BatchNorm2d`` members are parameter containers only: the arithmetic runs
in libyolo_hip.so.  There is no eager / CPU path.
"""
from __future__ import annotations

import os
from collections import OrderedDict

import numpy as np
import torch
from torch import nn

from .. import engine
from ..utils.torch_utils import fold_conv_bn, fuse_conv_and_bn
from .yolo_layer import YOLOLayer

DEFAULT_ANCHORS = (((10.0, 14.0), (23.0, 27.0), (37.0, 58.0)),
                   ((81.0, 82.0), (135.0, 169.0), (344.0, 319.0)))   # reference yolo_base.py:88-89

# bumped whenever a block changes its module structure (ConvBlock.fuse on ANY block, also a sub-module's): cached plans
# hold packed copies of the weights and are rebuilt when this moves (YOLOBase._fingerprint)
_STRUCT_EPOCH = [0]
_DATA_PTR = torch.Tensor.data_ptr
_VERSION = __import__("operator").attrgetter("_version")
MAX_CACHED_PLANS = 8          # per model: (shape, device, streams, precision) combinations kept (least recently used out)


class ConvBlock(nn.Module):
    """conv(no bias) + BatchNorm + LeakyReLU(0.1) — reference yolo_base.py:19-44."""

    def __init__(self, in_channels, out_channels, size=3, stride=1, pad=True):
        super().__init__()
        if not pad:
            raise NotImplementedError("pad=False ConvBlocks are not used by the hot-path models")
        if size not in (1, 3) or stride not in (1, 2):
            raise NotImplementedError("HIP conv supports 1x1/3x3, stride 1/2")
        self.sequence = nn.Sequential(OrderedDict([
            ("conv", nn.Conv2d(in_channels, out_channels, size, stride, (size - 1) // 2, bias=False)),
            ("batch_norm", nn.BatchNorm2d(out_channels)),
            ("activation", nn.LeakyReLU(0.1, inplace=True)),
        ]))
        self.in_channels, self.out_channels = in_channels, out_channels
        self.size, self.stride = size, stride

    @property
    def is_fused(self) -> bool:
        return not hasattr(self.sequence, "batch_norm")

    def folded(self):
        """(weight OIHW f32, bias f32) with the BN folded in (torch_utils.py:33-60)."""
        if self.is_fused:
            conv = self.sequence[0]
            return conv.weight.detach().float().cpu(), conv.bias.detach().float().cpu()
        conv, bn = self.sequence.conv, self.sequence.batch_norm
        return fold_conv_bn(conv.weight, conv.bias, bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.eps)

    def fuse(self):
        """Reference ConvBlock.fuse (yolo_base.py:46-57): sequence becomes [Conv2d(bias), LeakyReLU, ...]."""
        if self.is_fused:
            return
        rest = [m for name, m in self.sequence.named_children() if name not in ("conv", "batch_norm")]
        self.sequence = nn.Sequential(fuse_conv_and_bn(self.sequence.conv, self.sequence.batch_norm), *rest)
        _STRUCT_EPOCH[0] += 1

    def _trace(self, g: engine.Recorder, x, **kw):
        return g.conv(x, self.folded(), stride=self.stride, act="leaky", name=getattr(self, "_trace_name", None), **kw)

    def forward(self, x):
        return engine.run_standalone(lambda g, s: self._trace(g, s), x)


class MaxPool(nn.Module):
    """Reference MaxPool (yolo_base.py:60-66): pad (size-1)//2, except (2,1) = pad 1 / dilation 2."""

    def __init__(self, size, stride):
        super().__init__()
        self.size, self.stride = size, stride

    def _trace(self, g, x):
        return g.maxpool(x, self.size, self.stride)

    def forward(self, x):
        return engine.run_standalone(lambda g, s: self._trace(g, s), x)


class ConvPoolBlock(ConvBlock):
    """ConvBlock followed by a MaxPool — reference yolo_base.py:69-80."""

    def __init__(self, in_channels, out_channels, conv_size=3, conv_stride=1, conv_pad=True,
                 pool_size=2, pool_stride=2):
        super().__init__(in_channels, out_channels, conv_size, conv_stride, conv_pad)
        self.sequence.add_module("max_pool", MaxPool(pool_size, pool_stride))

    @property
    def pool(self) -> MaxPool:
        return self.sequence[-1]

    def _trace(self, g, x, **kw):
        return self.pool._trace(g, super()._trace(g, x, **kw))


class YOLOBase(nn.Module):
    """Constructor contract of reference YOLOBase (yolo_base.py:84-110)."""

    def __init__(self, in_channels=3, n_class=80, kernels_divider=1, anchors=DEFAULT_ANCHORS,
                 onnx=False, in_shape=None, hyper_params=None):
        super().__init__()
        if onnx:
            raise NotImplementedError("onnx=True selects the export branch, which is out of scope (SURVEY §2 #13)")
        self.header_info = np.zeros(5, dtype=np.int32)
        self.seen = self.header_info[3]
        self.hyper_params = hyper_params
        self.n_class = n_class
        self.onnx = False
        self.in_channels = in_channels
        self.anchors = anchors
        self.kernels_divider = kernels_divider
        self.in_shape = in_shape
        self.yolo_layer_input_size = 15 + 3 * n_class           # yolo_base.py:105
        self.encoder = None
        self._plans = {}
        self._tensors = None
        self.use_hip_graph = False
        self.n_streams = 2          # sub-batches run concurrently on this many HIP streams (engine.StreamedPlan)
        # "bf16": bf16 activations / weights with fp32 accumulation (the fast path); "fp32": float32 end to end on the f32
        # MFMA — the reference's arithmetic up to summation order, ~20x slower, for parity checks (csrc/conv_f32.hip)
        self.precision = os.environ.get("YOLO_PRECISION", "bf16")

    def _create_yolo_layers(self, device="cpu"):
        """One YOLOLayer per anchor group, in order (yolo_base.py:117-136)."""
        return [YOLOLayer(a, self.n_class, self.anchors, False) for a in self.anchors]

    # ---- cache control: any change of parameters invalidates the packed weights -------------------
    def invalidate(self):
        self._plans = {}
        self._tensors = None

    def _fingerprint(self):
        """Cheap identity of the weights a cached plan was packed from: (structure epoch, storage address and in-place
        version counter of every parameter / buffer).  It moves on sub-module load_state_dict / fuse, in-place edits of a
        parameter (p.copy_, nn.init.*, optimizer steps), .to() and p.data = new — what would otherwise leave a plan with
        stale packed weights.  NOT seen: in-place writes through ``p.data`` (p.data.copy_ / p.data.add_ bypass torch's
        version counters by design); call ``model.invalidate()`` after such an edit."""
        if self.__dict__.get("_tensors") is None or self.__dict__.get("_epoch_seen") != _STRUCT_EPOCH[0]:
            self._epoch_seen = _STRUCT_EPOCH[0]
            for m in self.modules():                  # assign=True loads replace Parameter objects: drop the tensor list then
                if not m.__dict__.get("_yolo_hooked"):
                    m.register_load_state_dict_post_hook(lambda _mod, _keys, _self=self: _self.invalidate())
                    m.__dict__["_yolo_hooked"] = True
            self._tensors = list(self.parameters()) + list(self.buffers())
        # (C-level maps: ~45 us for YOLOv3-SPP's 457 tensors instead of ~100 us per call - 2 % of a lone detect())
        return hash((self._epoch_seen, tuple(map(_DATA_PTR, self._tensors)), tuple(map(_VERSION, self._tensors))))

    def _apply(self, fn, *a, **kw):
        self.invalidate()
        return super()._apply(fn, *a, **kw)

    def load_state_dict(self, *a, **kw):
        self.invalidate()
        return super().load_state_dict(*a, **kw)

    def fuse(self):
        """Reference YOLOBase.fuse (yolo_base.py:112-115)."""
        for m in self.modules():
            if isinstance(m, ConvBlock):
                m.fuse()
        self.invalidate()

    # ---- darknet .weights IO (reference yolo_base.py:152-265; see utils/darknet_io.py) -----------------------
    def load_darknet_weights(self, weights_path, warnings=True):
        from ..utils.darknet_io import load_darknet_weights
        return load_darknet_weights(self, weights_path)

    def save_darknet_weights(self, path, warnings=True):
        from ..utils.darknet_io import save_darknet_weights
        save_darknet_weights(self, path)

    # ---- the path -----------------------------------------------------------------------------------
    def _trace(self, g: engine.Recorder, x):
        raise NotImplementedError

    def plan_for(self, x: torch.Tensor) -> engine.Plan:
        if x.dim() != 4 or x.shape[1] != self.in_channels:
            raise RuntimeError(f"expected input [bs,{self.in_channels},H,W], got {tuple(x.shape)}")
        if not x.is_cuda:
            raise RuntimeError("pytorch_yolo_amd runs on a ROCm device only: move the input to cuda "
                               "(there is no CPU fallback)")
        bs, c, h, w = x.shape
        n_streams = self.n_streams if (self.n_streams > 1 and bs % self.n_streams == 0 and bs // self.n_streams >= 4) else 1
        key = (tuple(x.shape), x.device, n_streams, self.precision)
        fp = self._fingerprint()
        if self.__dict__.get("_plans_fp") != fp:           # the weights changed since the plans were packed
            self._plans = {}
            self._plans_fp = fp
        plan = self._plans.pop(key, None)
        if plan is not None:
            self._plans[key] = plan                          # most recently used last
        if plan is None:
            for name, m in self.named_modules():          # reference key prefix of every block (drift traces, diagnostics)
                m._trace_name = name

            def make(sub_bs):
                rec = engine.Recorder(sub_bs, c, h, w)
                self._trace(rec, rec.input)
                return engine.Plan(rec, x.device, self.n_class, max(h, w), self.precision)    # img_size, yolov3_spp.py:142
            with torch.cuda.device(x.device):
                plan = make(bs) if n_streams == 1 else engine.StreamedPlan(make, bs, n_streams, x.device)
            self._plans[key] = plan
            while len(self._plans) > MAX_CACHED_PLANS:
                self._plans.pop(next(iter(self._plans)))
        return plan

    def forward(self, x):
        """eval: (io [bs, sum(3*ny*nx), 5+nc], (p_k [bs,3,ny,nx,5+nc], ...)) like the reference
        (yolov3_spp.py:141-164, yolov3_tiny.py:79-100)."""
        if self.training:
            raise NotImplementedError(
                "training-mode forward (batch-statistics BN, raw p list) is outside the inference hot path; "
                "call .eval() first")
        if x.dtype != torch.float32:
            x = x.float()
        x = x.contiguous()
        plan = self.plan_for(x)
        for hd in plan.heads:
            hd["layer"]._sync_grid_attrs(hd["sym"].h, hd["sym"].w, plan.img_size, x.device)
        with torch.cuda.device(x.device):           # the library launches on the CURRENT device's stream: make it x's
            if self.use_hip_graph:
                return plan.run_graph(x)
            return plan.run(x)

    def detect_stream(self, batches, conf_thres=0.5, nms_thres=0.5, depth=None):
        """``detect()`` over a stream of equally shaped batches with the GPU kept busy: a generator that yields, in order, the
        reference-style ``list[Tensor[n,7] | None]`` of every batch - batch k's list after batch k+depth-1 has been launched.
        ``depth`` = batches in flight (their output buffers form a ring): default S = the plan's pipelines (2) for models whose
        batch keeps the GPU busy for milliseconds, 2 S for the small ones (< 1 TFLOP per batch: a YOLOv3-tiny batch takes 0.37 ms,
        and with only S in flight a pipeline idles while the host hands out one batch and launches the next).  Successive batches alternate between the pipelines (``launch_detect(whole_batch=True)``):
        no host sync per batch except the count read-back of the batch being handed out, which by then has left the GPU.  The
        pipelines share the chip (no CU partition: ``launch_detect(cu_partition=...)`` says why).
        SPP-640 x 32: ~6,000 images/s against ~4,700 for back-to-back ``detect()`` calls (bench.py)."""
        from ..utils.utils import nms_capacity, split_detections
        if self.training:
            raise NotImplementedError("detect_stream() is an inference call: .eval() first")
        ring, pending, plan, shape, dev = [], [], None, None, None
        # Every batch is launched on the pipeline / NMS side streams (join=False) into buffers of ``ring`` that this generator
        # owns; the caching allocator knows nothing about those streams.  So nothing may leave this frame - the consumer breaking
        # out (GeneratorExit), an exception in the consumer, the shape check below - while a launched batch is still running:
        # its io / out blocks would be handed to the next allocation and be overwritten by the kernels in flight.
        try:
            for x in batches:
                x = x.float().contiguous()
                if shape is None:
                    shape, dev = tuple(x.shape), x.device
                    plan = self.plan_for(x)
                    if depth is None:
                        depth = max(2, plan.n_streams) * (2 if plan.conv_flops() < 1e12 else 1)
                    depth = max(2, int(depth))
                    cap = nms_capacity(plan.rows_total, self.n_class)
                    # the ring (output buffers, pinned count buffers, events, prebuilt pipeline steps) of the last generator that
                    # finished on this plan is taken over: building one costs more than 40 YOLOv3-tiny batches take
                    pool = plan.__dict__.setdefault("_stream_rings", {})
                    ring = pool.pop(depth, None) or []
                    for item in ring:
                        if item[4] is not None:
                            item[4].step.conf_thres, item[4].step.nms_thres = float(conf_thres), float(nms_thres)
                    with torch.cuda.device(x.device):
                        for slot in range(0 if ring else depth):
                            out = (torch.empty((shape[0], cap, 7), dtype=torch.float32, device=x.device),
                                   torch.empty((shape[0], cap), dtype=torch.int32, device=x.device),
                                   torch.empty((shape[0],), dtype=torch.int32, device=x.device))
                            # one FFI call per batch where the plan allows it (engine.FastStep: yolo_pipeline_step), in the compact NMS
                            # form: the heads filter their own rows, io is never written (include/yolo_hip.h)
                            fast = plan.fast_pipeline(slot, None, out, conf_thres, nms_thres) if hasattr(plan, "fast_pipeline") else None
                            io, ps = (None, ()) if fast is not None else plan.new_outputs(want_p=False)
                            ring.append((io, ps, out, torch.cuda.Event(), fast))
                elif tuple(x.shape) != shape or x.device != dev:
                    raise RuntimeError(f"detect_stream: batch {tuple(x.shape)} on {x.device} differs from the first one "
                                       f"{shape} on {dev}")
                while len(pending) >= len(ring):               # the oldest batch's buffers are needed again: hand it out first
                    yield self._collect(pending.pop(0))
                k = self.__dict__.setdefault("_stream_calls", 0)
                self._stream_calls = k + 1
                io, ps, out, done, fast = ring[k % len(ring)]
                with torch.cuda.device(x.device):
                    pending.append((x, out, done, fast))         # (before the launch: a launch that fails half way is drained too)
                    if fast is not None:
                        fast.launch(x)
                        continue
                    ready = torch.cuda.Event()
                    ready.record()                               # x was produced on the caller's stream: the pipeline waits for it
                    plan.launch_detect(x, io, ps, out, conf_thres, nms_thres, join=False, whole_batch=True, wait_for=ready, compact=True,
                                       after_nms=lambda i, lo, hi, done=done: done.record(torch.cuda.current_stream()))
            while pending:
                yield self._collect(pending.pop(0))
        finally:
            if pending and dev is not None:                      # abnormal exit with batches in flight: drain before ring dies
                torch.cuda.synchronize(dev)
                pending.clear()
            if plan is not None and ring and len(ring) == depth:
                plan.__dict__.setdefault("_stream_rings", {})[depth] = ring      # drained: the next generator on this plan reuses it

    @staticmethod
    def _collect(item):
        from ..utils.utils import split_detections
        _, out, done, fast = item
        if fast is not None:
            return fast.collect()
        done.synchronize()
        return split_detections(*out)

    def detect(self, x, conf_thres=0.5, nms_thres=0.5):
        """The composition inside reference test_model (utils/utils.py:374-378):
        ``non_max_suppression(model(x)[0], conf_thres, nms_thres)``."""
        from ..utils.utils import nms_capacity, split_detections
        if self.training:
            raise NotImplementedError("detect() is an inference call: .eval() first")
        x = x.float().contiguous()
        plan = self.plan_for(x)
        with torch.cuda.device(x.device):
            fast = plan.detect_step(conf_thres, nms_thres) if hasattr(plan, "detect_step") else None
            if fast is not None:                             # one whole-batch launch list + NMS behind ONE FFI call (engine.FastStep)
                fast.launch(x)
                return fast.collect()
            io, ps = plan.new_outputs(want_p=False)          # NMS reads io only: the raw head tensors are not materialised
            bs, cap = x.shape[0], nms_capacity(plan.rows_total, self.n_class)
            out = (torch.empty((bs, cap, 7), dtype=torch.float32, device=x.device),
                   torch.empty((bs, cap), dtype=torch.int32, device=x.device),
                   torch.empty((bs,), dtype=torch.int32, device=x.device))
            plan.launch_detect(x, io, ps, out, conf_thres, nms_thres, compact=True)     # (compact NMS form where the plan allows it)
            return split_detections(*out)
