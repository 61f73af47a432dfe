"""Layer-graph recorder, placement planner and executor for the HIP path.

A model describes its forward once, symbolically, through ``Recorder`` (the
model's ``_trace`` method reads like the reference's ``_forward_encoder``).
``Plan`` then decides where every activation lives so that the reference's
element-wise glue disappears into the convolution epilogues:

* ``Concat`` (reference models/yolo_layer.py:16-22): every input is produced
  straight into its channel slice of one NHWC buffer;
* ``Upsample(2)`` (:6-13): folded into the producing conv's store (2x2 replicate);
* ``Add`` (models/yolov3_spp.py:12-14): residual read in the conv epilogue,
  written in place of the residual input; the pre-add branch output the
  reference routes to the FPN (yolov3_spp.py:42-46) is a second epilogue store;
* SPP ``cat([p5,p9,p13,x])`` (yolov3_spp.py:129): conv writes x into the last
  slice, one kernel fills the other three.

Execution is one ``yolo_run_ops`` FFI call for all layers + one decode launch
per head, optionally captured in a HIP graph.
"""
from __future__ import annotations

import atexit
import ctypes as C
import os
from dataclasses import dataclass, field
from typing import List, Optional

import torch

from . import kernels as K
from ._lib import (ACT_LEAKY01, ACT_NONE, ACT_RELU, ACT_RELU6, ACT_SWISH, DT_BF16, DT_F32, OP_SE, OP_CONV, OP_CONV1_NCHW, OP_CONV_F32, OP_DWCONV,
                   OP_CONV1_POOL, OP_CONV_POOL, OP_MAXPOOL_F32, OP_MBCONV, OP_SHUFFLE, OP_HEAD_DECODE, OP_MAXPOOL, OP_RESUNIT, OP_SPP,
                   OP_STEM, YoloOp)

# which residual-unit widths run as ONE launch (bit mask of C: 64 | 128 | 256); see DESIGN.md §3.3 for the
# measurements behind the default (round 3, three interleaved rounds on one box, images/s: 64 -> 6,180; 64 | 128 -> 6,222;
# 64 | 128 | 256 -> 6,022: profiles/r03_fuse_resunit_ab.txt).  YOLO_FUSE_RESUNIT overrides it (tuning only).
FUSE_RESUNIT_DEFAULT = 64 | 128

# launches [a, b) of the list as S depth-first passes over image sub-batches ("a-b:S,..."; "" = off): Plan._depth_first
DEPTH_FIRST_DEFAULT = ""


# ------------------------------------------------------------------------------------------------
# symbolic graph
@dataclass(eq=False)
class Sym:
    """A logical NHWC activation."""
    n: int
    h: int
    w: int
    c: int
    producer: Optional["Node"] = None
    slot: int = 0                      # 0 main output, 1 pre-add output
    f32: bool = False
    # placement (filled by the planner)
    buf: Optional["Buf"] = None
    c_offset: int = 0
    consumers: List["Node"] = field(default_factory=list)


@dataclass(eq=False)
class Buf:
    n: int
    h: int
    w: int
    c_total: int
    f32: bool = False
    tensor: Optional[torch.Tensor] = None


@dataclass(eq=False)
class Node:
    kind: str                          # input | conv | dwconv | pool | spp | up | cat | head
    srcs: List[Sym]
    outs: List[Sym]
    attrs: dict


class Recorder:
    def __init__(self, n, c_in, h, w):
        self.nodes: List[Node] = []
        self.c_in = c_in
        x = Sym(n, h, w, K.roundup(c_in, 8))
        self.input = x
        self._add("input", [], [x])
        self.heads = []

    def _add(self, kind, srcs, outs, **attrs):
        node = Node(kind, list(srcs), list(outs), attrs)
        for o in outs:
            o.producer = node
        for s in srcs:
            s.consumers.append(node)
        self.nodes.append(node)
        return node

    # weight = (w_oihw f32, bias f32) already BN-folded; act in {'leaky','relu6','none'}
    @staticmethod
    def tf_same(size: int, k: int, stride: int):
        """TensorFlow "same" padding as efficientnet_pytorch 0.2.0's Conv2dSamePadding computes it: output ceil(size / stride),
        total pad max((out - 1) stride + k - size, 0), the odd one BELOW / RIGHT.  Returns (out, leading pad)."""
        out = -(-size // stride)
        return out, max((out - 1) * stride + k - size, 0) // 2

    def conv(self, x: Sym, weight, stride=1, act="leaky", residual: Sym = None, want_preadd=False, f32_out=False, pad=None,
             name=None, tf_same=False):
        w, _ = weight
        cout, cin_w, k, _ = w.shape
        if cin_w > x.c:
            raise RuntimeError(f"conv expects {cin_w} input channels, tensor has {x.c}")
        same = (k - 1) // 2
        pad = same if pad is None else pad
        ho, wo = (x.h + 2 * pad - k) // stride + 1, (x.w + 2 * pad - k) // stride + 1
        if tf_same:
            (ho, pad), (wo, pad_w) = self.tf_same(x.h, k, stride), self.tf_same(x.w, k, stride)
            if pad != pad_w:
                raise RuntimeError("tf_same conv: the two axes need different leading pads (one odd, one even size): not supported")
        if not f32_out and cout % 8:
            raise RuntimeError(f"internal conv width {cout} is not a multiple of 8 (unsupported kernels_divider)")
        y = Sym(x.n, ho, wo, cout, f32=f32_out)
        outs = [y]
        if want_preadd:
            outs.append(Sym(x.n, ho, wo, cout, slot=1))
        srcs = [x] + ([residual] if residual is not None else [])
        self._add("conv", srcs, outs, weight=weight, stride=stride, act=act, has_res=residual is not None, name=name)
        if pad != same:
            self.nodes[-1].attrs["pad"] = pad            # (SqueezeNet's unpadded first conv; no fused form takes it)
        return (y, outs[1]) if want_preadd else y

    def dwconv(self, x: Sym, weight, stride=1, act="relu6", tf_same=False):
        """Depthwise k x k conv.  Default: 3x3 / pad 1 (MobileNetV2).  ``tf_same``: k = 3 or 5 with TensorFlow "same" padding
        (EfficientNet-B0's MBConvBlock._depthwise_conv) - the general kernel, which also takes the swish activation."""
        w, _ = weight                                   # [c,1,k,k]
        k = w.shape[2]
        if not tf_same:
            if k != 3:
                raise RuntimeError("dwconv: only 3x3 with torch-style pad 1; pass tf_same=True for k = 5")
            ho, wo = (x.h - 1) // stride + 1, (x.w - 1) // stride + 1
            y = Sym(x.n, ho, wo, x.c)
            self._add("dwconv", [x], [y], weight=weight, stride=stride, act=act)
            return y
        (ho, pad), (wo, pad_w) = self.tf_same(x.h, k, stride), self.tf_same(x.w, k, stride)
        if pad != pad_w:
            raise RuntimeError("tf_same dwconv: the two axes need different leading pads: not supported")
        y = Sym(x.n, ho, wo, x.c)
        self._add("dwconv", [x], [y], weight=weight, stride=stride, act=act, ksize=k, pad=pad)
        return y

    def se(self, x: Sym, w1, b1, w2, b2):
        """Squeeze-and-excitation: y = x * sigmoid(W2 swish(W1 mean_hw(x) + b1) + b2) (efficientnet_pytorch MBConvBlock).
        w1: [sq, c] (or [sq, c, 1, 1]), w2: [c, sq]."""
        y = Sym(x.n, x.h, x.w, x.c)
        self._add("se", [x], [y], w1=w1, b1=b1, w2=w2, b2=b2)
        return y

    def maxpool(self, x: Sym, size, stride, pad=None, ceil_mode=False):
        # reference MaxPool: (2,1) -> pad 1, dilation 2 (models/yolo_base.py:60-66)
        if size == 2 and stride == 1:
            pad, dil = 1, 2
        else:
            pad, dil = ((size - 1) // 2 if pad is None else pad), 1

        def out(n):                                       # torch.nn.MaxPool2d output size incl. ceil_mode
            span = n + 2 * pad - dil * (size - 1) - 1
            o = (-(-span // stride) if ceil_mode else span // stride) + 1
            return o - 1 if ceil_mode and (o - 1) * stride >= n + pad else o
        ho, wo = out(x.h), out(x.w)
        y = Sym(x.n, ho, wo, x.c)
        self._add("pool", [x], [y], size=size, stride=stride, pad=pad, dil=dil)
        return y

    def spp_concat(self, x: Sym):
        y = Sym(x.n, x.h, x.w, 4 * x.c)
        self._add("spp", [x], [y])
        return y

    def upsample2(self, x: Sym):
        y = Sym(x.n, 2 * x.h, 2 * x.w, x.c)
        self._add("up", [x], [y])
        return y

    def concat(self, xs: List[Sym]):
        y = Sym(xs[0].n, xs[0].h, xs[0].w, sum(x.c for x in xs))
        self._add("cat", xs, [y])
        return y

    def slice(self, x: Sym, offset: int, c: int):
        """Channel view [offset, offset + c) of ``x`` (8-channel aligned): no launch, the consumers read the view."""
        if offset % 8 or c % 8 or offset + c > x.c:
            raise RuntimeError("slice: views are 8-channel aligned")
        y = Sym(x.n, x.h, x.w, c)
        self._add("slice", [x], [y], offset=offset)
        return y

    def shuffle2(self, a: Sym, b: Sym, half: int):
        """ShuffleNetV2's ``channel_shuffle(cat(a, b), groups=2)`` for two tensors of ``half`` logical channels each,
        every one held in a slot of a.c == b.c >= half physical channels (zero beyond ``half``): logical channel j of the
        result is (a, b)[j % 2][j // 2]; the result keeps the two-slot layout (logical [0, half) in slot 0, [half, 2 half)
        in slot 1)."""
        if a.c != b.c or half > a.c or (a.h, a.w) != (b.h, b.w):
            raise RuntimeError("shuffle2: mismatched halves")
        y = Sym(a.n, a.h, a.w, 2 * a.c)
        self._add("shuffle", [a, b], [y], half=half)
        return y

    def head(self, x: Sym, yolo_layer):
        self.heads.append((x, yolo_layer))
        self._add("head", [x], [])


_ACT = {"leaky": ACT_LEAKY01, "relu6": ACT_RELU6, "relu": ACT_RELU, "none": ACT_NONE, "swish": ACT_SWISH}


# ------------------------------------------------------------------------------------------------
class Plan:
    """Buffers + packed weights + launch list for one (batch, H, W) on one device.

    ``precision``: "bf16" (default: bf16 activations / weights, fp32 accumulate, every fusion) or "fp32" (the
    reference-precision parity mode: float32 activations and weights on the f32 MFMA, one plain launch per layer,
    csrc/conv_f32.hip)."""

    def __init__(self, rec: Recorder, device, n_class: int, img_size: int, precision: str = "bf16"):
        if precision not in ("bf16", "fp32"):
            raise ValueError(f"precision must be 'bf16' or 'fp32', got {precision!r}")
        self.f32 = precision == "fp32"
        self.precision = precision
        self.device = device
        self.rec = rec
        self.n_class = n_class
        self.img_size = img_size
        self._bufs: List[Buf] = []
        self._keep = []                 # packed weights / biases kept alive
        self.depth_first, self._x_patch = [], [(0, 0)]
        self._place()
        self.fused_input = self._first_conv_reads_nchw()
        self._alloc()
        self._build_ops()
        self._graph = None
        self._static_out = None

    # -- placement ---------------------------------------------------------------------------------
    def _new_buf(self, s: Sym, c_total=None) -> Buf:
        b = Buf(s.n, s.h, s.w, c_total or s.c, f32=s.f32)
        self._bufs.append(b)
        return b

    def _place(self):
        nodes = self.rec.nodes
        order = {id(nd): i for i, nd in enumerate(nodes)}
        # 1. concat / spp outputs own a buffer; their inputs are placed into slices of it
        for nd in nodes:
            if nd.kind == "cat":
                y = nd.outs[0]
                if y.buf is None:
                    y.buf, y.c_offset = self._new_buf(y), 0
                off = y.c_offset
                for x in nd.srcs:
                    if x.buf is not None:
                        raise RuntimeError("a tensor feeds two concats: needs a copy (not required by any model here)")
                    x.buf, x.c_offset = y.buf, off
                    off += x.c
            elif nd.kind == "spp":
                y, x = nd.outs[0], nd.srcs[0]
                if y.buf is None:
                    y.buf, y.c_offset = self._new_buf(y), 0
                if y.c_offset != 0 or y.buf.c_total != y.c:
                    raise RuntimeError("spp output must own its buffer")
                if x.buf is not None:
                    raise RuntimeError("spp input already placed")
                x.buf, x.c_offset = y.buf, 3 * x.c
        # 2. upsample outputs: the producing conv stores the replicated pixels itself
        for nd in nodes:
            if nd.kind == "up":
                x, y = nd.srcs[0], nd.outs[0]
                if x.producer.kind != "conv" or len(x.consumers) != 1:
                    raise RuntimeError("upsample must directly follow a conv with no other consumer")
                if y.buf is None:
                    y.buf, y.c_offset = self._new_buf(y), 0
                x.producer.attrs["up_into"] = y
        # 2b. Darknet residual units (1x1 C->C/2, 3x3 C/2->C, add) on the large maps: one launch (yolo_resunit_fwd);
        #     the intermediate never leaves the chip and the output gets its own buffer (no in-place add there)
        fuse_mask = 0 if self.f32 else int(os.environ.get("YOLO_FUSE_RESUNIT", str(FUSE_RESUNIT_DEFAULT)))
        for nd in nodes:
            if nd.kind != "conv" or not nd.attrs["has_res"] or "up_into" in nd.attrs:
                continue
            mid, res = nd.srcs[0], nd.srcs[1]
            pa = mid.producer
            if pa is None or pa.kind != "conv" or pa.attrs["has_res"] or len(pa.outs) != 1 or len(mid.consumers) != 1:
                continue
            (w1, _), (w2, _) = pa.attrs["weight"], nd.attrs["weight"]
            c = res.c
            if (pa.srcs[0] is not res or w1.shape[2] != 1 or w2.shape[2] != 3 or nd.attrs["stride"] != 1
                    or pa.attrs["stride"] != 1 or w2.shape[0] != c or w1.shape[0] * 2 != c or w1.shape[1] != c
                    or pa.attrs["act"] != nd.attrs["act"] or "up_into" in pa.attrs or mid.buf is not None
                    or nd.outs[0].f32):
                continue
            # (above 64 channels only where the 20-pixel-wide tile kernel takes the unit: the generic fused kernel is slower than two launches)
            if (fuse_mask & c) and K.resunit_supported(c, res.h, res.w) and (c == 64 or K.resunit_form(c, res.n, res.h, res.w) == 3):
                nd.attrs["fuse_pre"] = pa
                pa.attrs["fused_away"] = True
        # 2c. MobileNetV2 inverted-residual blocks (1x1 expand + ReLU6, depthwise 3x3 + ReLU6, linear 1x1 [+ x]) with
        #     few channels, i.e. the large maps: one launch (yolo_mbconv_fwd), the 6x-expanded tensor and the
        #     depthwise output never exist in HBM; the output gets its own buffer (neighbouring tiles read x)
        #     YOLO_FUSE_MBCONV: "1" all covered blocks, "narrow" only those of csrc/conv_mbconv.hip (hidden <= 192), "0" none
        mb_mode = os.environ.get("YOLO_FUSE_MBCONV", "1")
        if mb_mode != "0" and not self.f32:
            for nd in nodes:
                if (nd.kind != "conv" or "up_into" in nd.attrs or len(nd.outs) != 1 or nd.outs[0].f32
                        or nd.attrs["act"] != "none" or nd.attrs["stride"] != 1 or nd.attrs["weight"][0].shape[2] != 1):
                    continue
                dsym = nd.srcs[0]
                dwn = dsym.producer
                if dwn is None or dwn.kind != "dwconv" or dwn.attrs["act"] != "relu6" or len(dsym.consumers) != 1 or dsym.buf is not None:
                    continue
                esym = dwn.srcs[0]
                ex = esym.producer
                has_exp = (ex is not None and ex.kind == "conv" and ex.attrs["weight"][0].shape[2] == 1 and ex.attrs["act"] == "relu6"
                           and ex.attrs["stride"] == 1 and not ex.attrs["has_res"] and len(ex.outs) == 1 and "up_into" not in ex.attrs
                           and len(esym.consumers) == 1 and esym.buf is None and not esym.f32)
                x = ex.srcs[0] if has_exp else esym
                if x is self.rec.input or x.f32 or (nd.attrs["has_res"] and nd.srcs[1] is not x):
                    continue
                form = K.mbconv_form(x.c, esym.c, nd.attrs["weight"][0].shape[0], dwn.attrs["stride"])
                if form == 1 or (form == 2 and has_exp and mb_mode != "narrow"):
                    nd.attrs["mb_pre"] = (ex if has_exp else None, dwn, x)
                    dwn.attrs["fused_away"] = True
                    if has_exp:
                        ex.attrs["fused_away"] = True
        # 2d. ConvPoolBlocks with few input channels (YOLOv3-tiny's second and third: 16 -> 32, 32 -> 64): conv + MaxPool2d(2, 2)
        #     in one launch (yolo_conv3x3_pool_fwd), the full-resolution conv output is never written
        if os.environ.get("YOLO_FUSE_POOL", "1") == "1" and not self.f32:
            for nd in nodes:
                if (nd.kind != "conv" or nd.attrs["has_res"] or len(nd.outs) != 1 or "up_into" in nd.attrs or nd.attrs["stride"] != 1
                        or nd.attrs.get("fused_away") or nd.srcs[0] is self.rec.input):
                    continue
                w, _ = nd.attrs["weight"]
                mid = nd.outs[0]
                if (w.shape[2] != 3 or mid.f32 or mid.buf is not None or len(mid.consumers) != 1 or mid.consumers[0].kind != "pool"
                        or not K.conv3x3_pool_supported(w.shape[1], w.shape[0]) or mid.h % 2 or mid.w % 2):
                    continue
                ndp = mid.consumers[0]
                if (ndp.attrs["size"], ndp.attrs["stride"], ndp.attrs["pad"], ndp.attrs["dil"]) != (2, 2, 0, 1):
                    continue
                nd.attrs["pool_into"] = ndp.outs[0]
                nd.attrs["small_pool"] = True
                ndp.attrs["fused_away"] = True
        # 3. residual adds are written in place of the residual input when it is dead afterwards
        for nd in nodes:
            if nd.kind == "conv" and nd.attrs["has_res"] and "fuse_pre" not in nd.attrs and "mb_pre" not in nd.attrs:
                res, y = nd.srcs[1], nd.outs[0]
                dead = all(order[id(cn)] <= order[id(nd)] for cn in res.consumers)
                if dead and y.buf is None and res.buf is not None and not y.f32:
                    y.buf, y.c_offset = res.buf, res.c_offset
                elif dead and y.buf is None and res.buf is None:
                    nd.attrs["alias_res"] = True
        # 3b. detection heads: the head conv decodes in its epilogue (yolo_head_decode_fwd); no head tensor
        if os.environ.get("YOLO_FUSE_HEAD", "1") == "1" and not self.f32:
            for nd in nodes:
                layer = self._head_layer(nd)
                if layer is None or nd.attrs["has_res"] or len(nd.outs) != 1 or "up_into" in nd.attrs or nd.attrs["stride"] != 1:
                    continue
                w, _ = nd.attrs["weight"]
                if K.head_decode_supported(w.shape[0], len(layer.anchors_px), self.n_class) and nd.outs[0].buf is None:
                    nd.attrs["head_fused"] = True
        # 4. everything else gets its own buffer
        for nd in nodes:
            if nd.attrs.get("fused_away") or nd.attrs.get("head_fused"):
                continue
            if nd.kind == "slice":
                continue                                   # a view: resolved below
            for o in nd.outs:
                if nd.kind == "conv" and nd.attrs.get("small_pool"):
                    pooled = nd.attrs["pool_into"]         # the full-resolution map is never materialised, the pooled one is
                    if pooled.buf is None:
                        pooled.buf, pooled.c_offset = self._new_buf(pooled), 0
                    continue
                if o.buf is None and not (nd.kind == "conv" and "up_into" in nd.attrs and o.slot == 0):
                    if nd.kind == "conv" and nd.attrs.get("alias_res") and o.slot == 0:
                        continue
                    o.buf, o.c_offset = self._new_buf(o), 0
        for nd in nodes:      # channel views (in node order: a view of a view resolves too)
            if nd.kind == "slice":
                src, y = nd.srcs[0], nd.outs[0]
                y.buf, y.c_offset = src.buf, src.c_offset + nd.attrs["offset"]
        for nd in nodes:      # resolve in-place aliases now that the residual inputs have buffers
            if nd.kind == "conv" and nd.attrs.get("alias_res"):
                res, y = nd.srcs[1], nd.outs[0]
                y.buf, y.c_offset = res.buf, res.c_offset

    def _head_layer(self, nd):
        """The YOLOLayer fed by conv node ``nd`` (its f32 output goes to a head and nowhere else), or None."""
        if nd.kind != "conv" or not nd.outs[0].f32:
            return None
        y = nd.outs[0]
        if len(y.consumers) != 1 or y.consumers[0].kind != "head":
            return None
        for x, layer in self.rec.heads:
            if x is y:
                return layer
        return None

    def _first_conv_reads_nchw(self) -> bool:
        """The first layer can read the caller's float32 NCHW batch itself (yolo_conv1_nchw_f32_fwd): then the
        NHWC bf16 copy of the input is never materialised."""
        x = self.rec.input
        if self.f32 or len(x.consumers) != 1 or x.consumers[0].kind != "conv" or self.rec.c_in > 8:
            return False
        nd = x.consumers[0]
        w, _ = nd.attrs["weight"]
        y = nd.outs[0]
        s1 = w.shape[0] in (16, 32) and nd.attrs["stride"] == 1
        s2 = w.shape[0] == 32 and nd.attrs["stride"] == 2 and os.environ.get("YOLO_FUSE_CONV1_S2", "1") == "1"   # MobileNetV2
        ok = ((s1 or s2) and w.shape[2] == 3 and "pad" not in nd.attrs and not nd.attrs["has_res"]
              and len(nd.outs) == 1 and "up_into" not in nd.attrs and not y.f32 and y.buf is not None
              and y.c_offset % 8 == 0)
        if ok and x.buf in self._bufs:
            self._bufs.remove(x.buf)          # no packed input buffer needed
        if ok and s1:
            self._try_fuse_stem(nd)
            if "fused_away" not in nd.attrs:
                self._try_fuse_pool(nd)
        return ok

    def _try_fuse_pool(self, nd1):
        """First ConvPoolBlock of YOLOv3-tiny: conv1 followed only by MaxPool2d(2, 2) becomes ONE launch
        (yolo_conv1_pool_nchw_f32_fwd); the full-resolution conv output is never written."""
        if os.environ.get("YOLO_FUSE_POOL", "1") != "1":
            return
        mid = nd1.outs[0]
        if len(mid.consumers) != 1 or mid.consumers[0].kind != "pool" or mid.c_offset != 0 or mid.buf.c_total != mid.c:
            return
        ndp = mid.consumers[0]
        if (ndp.attrs["size"], ndp.attrs["stride"], ndp.attrs["pad"], ndp.attrs["dil"]) != (2, 2, 0, 1) or mid.h % 2 or mid.w % 2:
            return
        nd1.attrs["pool_into"] = ndp.outs[0]
        ndp.attrs["fused_away"] = True
        if mid.buf in self._bufs:
            self._bufs.remove(mid.buf)

    def _try_fuse_stem(self, nd1):
        """Darknet stem: conv1 (3x3/s1 -> 32) followed only by a 3x3/s2 32 -> 64 conv becomes ONE launch
        (yolo_stem_fwd); the 32-channel full-resolution intermediate is never materialised."""
        if os.environ.get("YOLO_FUSE_STEM", "1") != "1":
            return
        mid = nd1.outs[0]
        if len(mid.consumers) != 1 or mid.consumers[0].kind != "conv" or mid.c_offset != 0 or mid.buf.c_total != mid.c:
            return
        nd2 = mid.consumers[0]
        w2, _ = nd2.attrs["weight"]
        y = nd2.outs[0]
        if (tuple(w2.shape) != (64, 32, 3, 3) or nd2.attrs["stride"] != 2 or nd2.attrs["has_res"] or len(nd2.outs) != 1
                or "up_into" in nd2.attrs or y.f32 or nd2.attrs["act"] != nd1.attrs["act"] or nd2.srcs[0] is not mid
                or y.c_offset % 8 or "fuse_pre" in nd2.attrs):
            return
        nd2.attrs["stem_pre"] = nd1
        nd1.attrs["fused_away"] = True
        if mid.buf in self._bufs:
            self._bufs.remove(mid.buf)

    def _live_ranges(self):
        """[first, last] launch-order index at which each buffer is touched.  A buffer is touched by the producer and by every
        consumer of each tensor placed in it (views, concat slices, in-place residual outputs and upsample targets share their
        buffer, so they widen ITS range); None for a buffer some tensor of which has no producer / is a model output."""
        nodes = self.rec.nodes
        order = {id(nd): i for i, nd in enumerate(nodes)}
        rng = {}
        pinned = set()

        def touch(buf, i):
            if buf is None:
                return
            lo, hi = rng.get(id(buf), (i, i))
            rng[id(buf)] = (min(lo, i), max(hi, i))

        for nd in nodes:
            i = order[id(nd)]
            for s_ in list(nd.srcs) + list(nd.outs):
                touch(s_.buf, i)
            for key in ("up_into", "pool_into"):                 # a conv that stores into another node's output buffer
                if key in nd.attrs:
                    touch(nd.attrs[key].buf, i)
            for key in ("fuse_pre", "stem_pre"):                 # fused pairs: the later node's launch reads the earlier node's inputs
                pre = nd.attrs.get(key)
                if pre is not None:
                    for s_ in pre.srcs:
                        touch(s_.buf, i)
            if "mb_pre" in nd.attrs:
                ex, dwn, x = nd.attrs["mb_pre"]
                touch(x.buf, i)
            if nd.kind in ("input", "head"):
                for s_ in list(nd.srcs) + list(nd.outs):
                    if s_.buf is not None:
                        pinned.add(id(s_.buf))
        return rng, pinned

    def _alloc(self):
        """One torch tensor per buffer; buffers of identical shape whose live ranges do not overlap share storage
        (YOLO_REUSE_BUFFERS=0 turns that off).  Sharing keeps a residual stage's working set - the stream x (in place) and ONE
        intermediate t instead of one per unit - inside the 256 MB Infinity Cache, and a dead intermediate is overwritten there
        instead of being written back to HBM.  Only buffers that their producers fill completely take part (a padded
        buffer relies on its zero fill)."""
        reuse = os.environ.get("YOLO_REUSE_BUFFERS", "1") == "1" and not self.f32
        rng, pinned = self._live_ranges() if reuse else ({}, set())
        filled = {}
        for nd in self.rec.nodes:
            for s_ in nd.outs:
                if s_.buf is not None:
                    filled[id(s_.buf)] = filled.get(id(s_.buf), 0) + s_.c
        pool = {}                                                   # shape key -> [(last use, tensor)]
        self.shared_buffers = 0
        for b in sorted(self._bufs, key=lambda b_: rng.get(id(b_), (0, 0))[0]):
            dt = torch.float32 if (b.f32 or self.f32) else torch.bfloat16
            ct = K.roundup(b.c_total, 8)
            exact = ct == b.c_total and filled.get(id(b), 0) == b.c_total
            b.c_total = ct
            key = (b.n, b.h, b.w, ct, dt)
            lo, hi = rng.get(id(b), (None, None))
            if reuse and exact and lo is not None and id(b) not in pinned:
                free = pool.setdefault(key, [])
                hit = next((e for e in free if e[0] < lo), None)
                if hit is not None:
                    free.remove(hit)
                    b.tensor = hit[1]
                    self.shared_buffers += 1
                else:
                    b.tensor = self._zeros((b.n, b.h, b.w, ct), dt)
                free.append((hi, b.tensor))
                continue
            # zero-filled once: padded channels (e.g. 255 -> 256 head rows) are never written
            b.tensor = self._zeros((b.n, b.h, b.w, ct), dt)

    # -- launch list ---------------------------------------------------------------------------------
    # -- red zones (diagnostic, tests/test_gpu_parity.py::test_launch_lists_stay_inside_their_buffers) ----------------------
    # YOLO_REDZONE=<bytes>: every activation buffer and every packed weight / bias of the plan sits in the middle of a larger
    # allocation whose margins are filled with REDZONE_BYTE (0x7f7f... = a large finite bf16 / f32 pattern that max pools, MFMAs
    # and decodes would carry into the results).  ``redzone_report()`` then tells whether a launch wrote outside its buffer;
    # results that change against a plain plan show reads outside.
    REDZONE_BYTE = 0x7F

    def _redzone(self) -> int:
        return int(os.environ.get("YOLO_REDZONE", "0")) // 256 * 256

    def _guarded(self, nbytes: int):
        """(raw uint8 allocation with poisoned margins, byte offset of the payload)."""
        rz = self._redzone()
        raw = torch.full((nbytes + 2 * rz,), self.REDZONE_BYTE, dtype=torch.uint8, device=self.device)
        self.__dict__.setdefault("_redzones", []).append((raw, rz, nbytes))
        return raw, rz

    def _zeros(self, shape, dtype):
        if not self._redzone():
            return torch.zeros(shape, dtype=dtype, device=self.device)
        numel = 1
        for v in shape:
            numel *= v
        nbytes = numel * torch.empty((), dtype=dtype).element_size()
        raw, rz = self._guarded(nbytes)
        t = raw[rz:rz + nbytes].view(dtype).view(shape)
        t.zero_()
        return t

    def redzone_report(self):
        """Number of margin bytes that no longer hold REDZONE_BYTE, per guarded allocation: [(index, payload bytes, below, above)]
        for the damaged ones (empty list: every launch stayed inside its buffers)."""
        bad = []
        for i, (raw, rz, nbytes) in enumerate(self.__dict__.get("_redzones", [])):
            lo = int((raw[:rz] != self.REDZONE_BYTE).sum())
            hi = int((raw[rz + nbytes:] != self.REDZONE_BYTE).sum())
            if lo or hi:
                bad.append((i, nbytes, lo, hi))
        return bad

    def _dev(self, t):
        if self._redzone() and t.numel():
            nbytes = t.numel() * t.element_size()
            raw, rz = self._guarded(nbytes)
            d = raw[rz:rz + nbytes].view(t.dtype).view(t.shape)
            d.copy_(t)
            t = d
        else:
            t = t.to(self.device)
        self._keep.append(t)
        return t

    def _build_ops(self):
        # heads: io row ranges in the order the model declares them (yolov3_spp.py:156-164)
        self.heads = []
        row = 0
        for x, layer in self.rec.heads:
            na = len(layer.anchors_px)
            stride = self.img_size / max(x.w, x.h)          # yolo_layer.py:102 (python float)
            self.heads.append(dict(sym=x, anchors=layer.anchors_px, stride=stride, row=row, na=na, layer=layer, op=None))
            row += na * x.h * x.w
        self.rows_total = row
        if self.f32:
            return self._build_ops_f32()
        ops = []
        op_nodes = []                                      # the graph node each launch comes from (tools/drift_trace.py)
        splitk_ops = []                                    # (op index, workspace bytes, counters) of the split-K launches
        for nd in self.rec.nodes:
            if nd.attrs.get("fused_away"):
                continue
            if nd.kind == "conv" and "stem_pre" in nd.attrs:
                nd1 = nd.attrs["stem_pre"]
                mid, y = nd.srcs[0], nd.outs[0]
                w1p, b1p, kpad1, _ = K.pack_conv_weight(*nd1.attrs["weight"], 8)
                w2p, b2p, kpad2, cout_pad2 = K.pack_conv_weight(*nd.attrs["weight"], 32)
                w1p, b1p, w2p, b2p = (self._dev(t) for t in (w1p, b1p, w2p, b2p))
                assert not ops and self.fused_input            # feed() patches op 0's x with the caller's batch
                op = YoloOp()
                op.kind = OP_STEM
                op.x, op.y = None, y.buf.tensor.data_ptr()
                op.w, op.bias, op.w_pre, op.bias_pre, op.kpad_pre = w2p.data_ptr(), b2p.data_ptr(), w1p.data_ptr(), b1p.data_ptr(), kpad1
                op.conv = K.conv_desc(n=mid.n, h=mid.h, w=mid.w, cin=32, in_c_total=32, in_c_offset=0, cout=64,
                                      out_c_total=y.buf.c_total, out_c_offset=y.c_offset, ksize=3, stride=2,
                                      act=_ACT[nd.attrs["act"]], kpad=kpad2, cout_pad=cout_pad2)
                op.conv.res_c_total = self.rec.c_in            # real input channels
                ops.append(op); op_nodes.append(nd)
            elif nd.kind == "conv" and nd.attrs.get("head_fused"):
                x, y = nd.srcs[0], nd.outs[0]
                hd = next(h for h in self.heads if h["sym"] is y)
                w, b = nd.attrs["weight"]
                wp, bp, kpad, cout_pad = K.pack_conv_weight(w, b, x.c)
                wp, bp = self._dev(wp), self._dev(bp)
                op = YoloOp()
                op.kind = OP_HEAD_DECODE
                op.x, op.w, op.bias = x.buf.tensor.data_ptr(), wp.data_ptr(), bp.data_ptr()
                op.y = op.y_aux = None                      # io / p of the call: bound in _bind_outputs
                op.conv = K.conv_desc(n=x.n, h=x.h, w=x.w, cin=x.c, in_c_total=x.buf.c_total, in_c_offset=x.c_offset,
                                      cout=w.shape[0], out_c_total=K.roundup(w.shape[0], 8), out_c_offset=0,
                                      ksize=w.shape[2], stride=1, act=_ACT[nd.attrs["act"]], kpad=kpad,
                                      cout_pad=cout_pad, out_dtype=DT_F32)
                for i, (aw, ah) in enumerate(hd["anchors"]):
                    op.head_anchors_px[2 * i], op.head_anchors_px[2 * i + 1] = float(aw), float(ah)
                op.head_stride_px, op.head_na, op.head_nc = float(hd["stride"]), hd["na"], self.n_class
                op.io_rows_total, op.io_row_offset = self.rows_total, hd["row"]
                hd["op"] = len(ops)
                ops.append(op); op_nodes.append(nd)
            elif nd.kind == "conv" and "mb_pre" in nd.attrs:
                ex, dwn, x = nd.attrs["mb_pre"]
                y = nd.outs[0]
                hidden = dwn.srcs[0].c
                we_b = ex.attrs["weight"] if ex is not None else (None, None)
                packed = K.pack_mbconv(we_b[0], we_b[1], *dwn.attrs["weight"], *nd.attrs["weight"], stride=dwn.attrs["stride"])
                we, be, wd, bd, wp, bp = (None if t is None else self._dev(t) for t in packed)
                op = YoloOp()
                op.kind = OP_MBCONV
                op.x, op.y = x.buf.tensor.data_ptr(), y.buf.tensor.data_ptr()
                op.w, op.bias, op.w_dw, op.bias_dw = wp.data_ptr(), bp.data_ptr(), wd.data_ptr(), bd.data_ptr()
                op.w_pre, op.bias_pre = (we.data_ptr(), be.data_ptr()) if we is not None else (None, None)
                op.kpad_pre = hidden
                d = op.conv
                d.n, d.h, d.w, d.cin, d.in_c_total, d.in_c_offset = x.n, x.h, x.w, x.c, x.buf.c_total, x.c_offset
                d.ho, d.wo, d.cout, d.out_c_total, d.out_c_offset = y.h, y.w, y.c, y.buf.c_total, y.c_offset
                d.ksize, d.stride, d.res_c_total = 3, dwn.attrs["stride"], 1 if nd.attrs["has_res"] else 0
                ops.append(op); op_nodes.append(nd)
            elif nd.kind == "conv" and "fuse_pre" in nd.attrs:
                pa = nd.attrs["fuse_pre"]
                x, y, mid = nd.srcs[1], nd.outs[0], nd.srcs[0]
                aux = nd.outs[1] if len(nd.outs) > 1 else None
                w1p, b1p, kpad1, cout_pad1 = K.pack_conv_weight(*pa.attrs["weight"], x.c)
                w2p, b2p, kpad2, cout_pad2 = K.pack_conv_weight(*nd.attrs["weight"], mid.c)
                w1p, b1p, w2p, b2p = (self._dev(t) for t in (w1p, b1p, w2p, b2p))
                op = YoloOp()
                op.kind = OP_RESUNIT
                op.x, op.y = x.buf.tensor.data_ptr(), y.buf.tensor.data_ptr()
                op.w, op.bias, op.w_pre, op.bias_pre = w2p.data_ptr(), b2p.data_ptr(), w1p.data_ptr(), b1p.data_ptr()
                op.kpad_pre, op.cout_pad_pre = kpad1, cout_pad1
                op.y_aux = aux.buf.tensor.data_ptr() if aux is not None else None
                op.conv = K.conv_desc(n=x.n, h=x.h, w=x.w, cin=mid.c, in_c_total=x.buf.c_total, in_c_offset=x.c_offset,
                                      cout=x.c, out_c_total=y.buf.c_total, out_c_offset=y.c_offset, ksize=3, stride=1,
                                      act=_ACT[nd.attrs["act"]], kpad=kpad2, cout_pad=cout_pad2,
                                      aux=(aux.buf.c_total, aux.c_offset) if aux is not None else (0, 0))
                ops.append(op); op_nodes.append(nd)
            elif nd.kind == "conv":
                x = nd.srcs[0]
                y = nd.outs[0]
                w, b = nd.attrs["weight"]
                wp, bp, kpad, cout_pad = K.pack_conv_weight(w, b, x.c)
                wp, bp = self._dev(wp), self._dev(bp)
                up = nd.attrs.get("up_into")
                pooled = nd.attrs.get("pool_into")         # conv + MaxPool2d(2, 2) in one launch: the op writes the pooled map
                dst = pooled if pooled is not None else (up if up is not None else y)
                res = nd.srcs[1] if nd.attrs["has_res"] else None
                aux = nd.outs[1] if len(nd.outs) > 1 else None
                d = K.conv_desc(n=x.n, h=x.h, w=x.w, cin=x.c, in_c_total=x.buf.c_total, in_c_offset=x.c_offset,
                                cout=w.shape[0], out_c_total=dst.buf.c_total, out_c_offset=dst.c_offset,
                                ksize=w.shape[2], stride=nd.attrs["stride"], act=_ACT[nd.attrs["act"]],
                                kpad=kpad, cout_pad=cout_pad, upsample2x=1 if up is not None else 0,
                                out_dtype=DT_F32 if y.f32 else DT_BF16, pad=nd.attrs.get("pad"),
                                res=(res.buf.c_total, res.c_offset) if res is not None else (0, 0),
                                aux=(aux.buf.c_total, aux.c_offset) if aux is not None else (0, 0))
                if up is None and pooled is None:
                    d.ho, d.wo = y.h, y.w                  # (tf_same convs: one more row / column than the symmetric-pad formula)
                op = YoloOp()
                fused_first = self.fused_input and x is self.rec.input
                if fused_first:
                    op.kind = OP_CONV1_POOL if pooled is not None else OP_CONV1_NCHW
                else:
                    op.kind = OP_CONV_POOL if nd.attrs.get("small_pool") else OP_CONV
                    assert pooled is None or op.kind == OP_CONV_POOL
                if fused_first:
                    d.res_c_total = self.rec.c_in          # real input channels (x pointer is patched per call)
                # split-K launches are OFF by default: correct and deterministic (tests), but on MI355X the cross-XCD exchange
                # of the fp32 partials (agent-scope accesses that bypass the per-XCD L2) costs more than the idle CUs it
                # fills: 0.041 -> 0.13 ms on YOLOv3-tiny's 3x3 256 -> 512 layer at 13x13 x 32
                if op.kind == OP_CONV and os.environ.get("YOLO_SPLITK", "0") == "1":
                    sp, wb, nc = K.conv2d_splitk_plan(d, res is not None, aux is not None)
                    if sp >= 2:                            # few pixels, long K: split-K launch (yolo_conv2d_splitk_fwd)
                        op.splits, op.ws_bytes = sp, wb
                        splitk_ops.append((len(ops), wb, nc))
                op.x = None if fused_first else x.buf.tensor.data_ptr()
                op.w, op.bias = wp.data_ptr(), bp.data_ptr()
                op.residual = res.buf.tensor.data_ptr() if res is not None else None
                op.y = dst.buf.tensor.data_ptr()
                op.y_aux = aux.buf.tensor.data_ptr() if aux is not None else None
                op.conv = d
                ops.append(op); op_nodes.append(nd)
            elif nd.kind == "dwconv":
                x, y = nd.srcs[0], nd.outs[0]
                w, b = nd.attrs["weight"]
                kk = w.shape[2] * w.shape[3]
                w9c = self._dev(w.detach().float().reshape(x.c, kk).t().contiguous())
                bb = self._dev(b.detach().float().contiguous())
                op = YoloOp()
                op.kind = OP_DWCONV
                op.x, op.w, op.bias, op.y = x.buf.tensor.data_ptr(), w9c.data_ptr(), bb.data_ptr(), y.buf.tensor.data_ptr()
                d = op.conv
                d.n, d.h, d.w, d.cin = x.n, x.h, x.w, x.c
                d.in_c_total, d.in_c_offset, d.ho, d.wo = x.buf.c_total, x.c_offset, y.h, y.w
                d.out_c_total, d.out_c_offset, d.stride, d.act = y.buf.c_total, y.c_offset, nd.attrs["stride"], _ACT[nd.attrs["act"]]
                d.ksize, d.pad = nd.attrs.get("ksize", 0), nd.attrs.get("pad", 0)      # ksize 0: the 3x3 / pad 1 strip kernel
                ops.append(op); op_nodes.append(nd)
            elif nd.kind == "se":
                x, y = nd.srcs[0], nd.outs[0]
                sq = nd.attrs["w1"].shape[0]
                w1 = self._dev(nd.attrs["w1"].detach().float().reshape(sq, x.c).contiguous())
                w2 = self._dev(nd.attrs["w2"].detach().float().reshape(x.c, sq).t().contiguous())    # [sq][c]
                b1, b2 = self._dev(nd.attrs["b1"].detach().float().contiguous()), self._dev(nd.attrs["b2"].detach().float().contiguous())
                ws = self._dev(torch.zeros(K.se_workspace_bytes(x.n, x.c) // 4, dtype=torch.float32))
                op = YoloOp()
                op.kind = OP_SE
                op.x, op.y, op.w, op.bias = x.buf.tensor.data_ptr(), y.buf.tensor.data_ptr(), w1.data_ptr(), b1.data_ptr()
                op.w_pre, op.bias_pre, op.kpad_pre = w2.data_ptr(), b2.data_ptr(), sq
                op.workspace, op.ws_bytes = ws.data_ptr(), ws.numel() * 4
                d = op.conv
                d.n, d.h, d.w, d.cin = x.n, x.h, x.w, x.c
                d.in_c_total, d.in_c_offset, d.ho, d.wo = x.buf.c_total, x.c_offset, y.h, y.w
                d.out_c_total, d.out_c_offset = y.buf.c_total, y.c_offset
                ops.append(op); op_nodes.append(nd)
            elif nd.kind == "shuffle":
                a_, b_, y = nd.srcs[0], nd.srcs[1], nd.outs[0]
                op = YoloOp()
                op.kind = OP_SHUFFLE
                op.x, op.residual, op.y = a_.buf.tensor.data_ptr(), b_.buf.tensor.data_ptr(), y.buf.tensor.data_ptr()
                d = op.conv
                d.n, d.h, d.w, d.cin = a_.n, a_.h, a_.w, a_.c            # cin = physical channels per slot
                d.in_c_total, d.in_c_offset = a_.buf.c_total, a_.c_offset
                d.res_c_total, d.res_c_offset = b_.buf.c_total, b_.c_offset
                d.out_c_total, d.out_c_offset, d.cout = y.buf.c_total, y.c_offset, nd.attrs["half"]   # cout = logical half
                ops.append(op); op_nodes.append(nd)
            elif nd.kind == "pool":
                x, y = nd.srcs[0], nd.outs[0]
                op = YoloOp()
                op.kind = OP_MAXPOOL
                op.x, op.y = x.buf.tensor.data_ptr(), y.buf.tensor.data_ptr()
                d = op.conv
                d.n, d.h, d.w, d.cin = x.n, x.h, x.w, x.c
                d.in_c_total, d.in_c_offset, d.ho, d.wo = x.buf.c_total, x.c_offset, y.h, y.w
                d.out_c_total, d.out_c_offset = y.buf.c_total, y.c_offset
                d.ksize, d.stride, d.pad, d.upsample2x = nd.attrs["size"], nd.attrs["stride"], nd.attrs["pad"], nd.attrs["dil"]
                ops.append(op); op_nodes.append(nd)
            elif nd.kind == "spp":
                x, y = nd.srcs[0], nd.outs[0]
                op = YoloOp()
                op.kind = OP_SPP
                op.y = y.buf.tensor.data_ptr()
                d = op.conv
                d.n, d.h, d.w, d.cin = x.n, x.h, x.w, x.c
                ops.append(op); op_nodes.append(nd)
        # Diagnostic only (timing; results wrong): YOLO_SHRINK_OPS="i,j,..." makes launches i, j, ... of the list process ONE image instead
        # of the batch - same list, same streams, the launch itself nearly free -, so that `bench.py` can measure what a launch family
        # costs INSIDE the pipelined step (tools/marginal_cost.py), as opposed to alone on an idle chip.
        shrink = os.environ.get("YOLO_SHRINK_OPS", "")
        if shrink and not self.f32:
            heads = {hd["op"] for hd in self.heads if hd["op"] is not None}
            for i in (int(v) for v in shrink.split(",") if v.strip()):
                op = ops[i]
                if i not in heads and op.splits < 2 and op.conv.n == self.rec.input.n and op.conv.n > 1:
                    sub = YoloOp.from_buffer_copy(op)      # image 0 only: every batch-major pointer already points at it
                    sub.conv.n = 1
                    ops[i] = sub
        ops, op_nodes, shift = self._depth_first(ops, op_nodes)
        if shift:
            splitk_ops = [(i + shift(i), wb, nc) for i, wb, nc in splitk_ops]
            for hd in self.heads:
                if hd["op"] is not None:
                    hd["op"] += shift(hd["op"])
        if splitk_ops:      # one fp32 workspace and one zeroed counter array per plan: the launches run in stream order
            self._splitk_ws = torch.empty((max(w for _, w, _ in splitk_ops) + 3) // 4, dtype=torch.float32, device=self.device)
            self._splitk_cnt = torch.zeros(max(c for _, _, c in splitk_ops), dtype=torch.int32, device=self.device)
            for i, _, _ in splitk_ops:
                ops[i].workspace, ops[i].counters = self._splitk_ws.data_ptr(), self._splitk_cnt.data_ptr()
        self.n_ops = len(ops)
        self.op_nodes = op_nodes
        self.op_array = (YoloOp * len(ops))(*ops)
    # -- depth-first sub-batches for the first stages ---------------------------------------------------
    # The first stages of YOLOv3-SPP at 640x640 move 105 MB per 8 images and tensor (320x320x64 bf16) and are bound by memory
    # latency, not by the matrix pipe.  A whole-batch launch list runs them breadth-first: every launch streams 0.4-0.8 GB, so by the
    # time a consumer launch starts, what its producer wrote has left the 256 MB Infinity Cache.  ``YOLO_DEPTH_FIRST="a-b:S,..."``
    # replaces launches [a, b) of the list by S passes over them, pass j on images [j n / S, (j + 1) n / S): same kernels, same tiles
    # (a tile never crosses an image), same values - bit for bit -, the producer's sub-batch output still cache-resident when the
    # consumer reads it.  Images are independent and the order inside a pass is the list's, so every dependency is kept; buffers
    # that share storage (``_alloc``) do so image slice by image slice.
    def _depth_first(self, ops, op_nodes):
        spec = os.environ.get("YOLO_DEPTH_FIRST", DEPTH_FIRST_DEFAULT if self.rec.input.n >= 16 else "")
        self.depth_first = []
        self._x_patch = [(0, 0)]                            # (op index, byte offset into the caller's NCHW batch) of the ops that read it
        if not spec or spec == "0" or self.f32:
            return ops, op_nodes, None
        n = self.rec.input.n
        segs = []
        for part in spec.split(","):
            rng, s_ = part.split(":")
            a, b = (int(v) for v in rng.split("-"))
            s_ = int(s_)
            b = min(b, len(ops))
            ok = s_ >= 2 and n % s_ == 0 and 0 <= a < b and (not segs or a >= segs[-1][1])
            ok = ok and all(op.kind in (OP_STEM, OP_RESUNIT, OP_CONV) and op.splits < 2 and op.conv.n == n for op in ops[a:b])
            if ok:
                segs.append((a, b, s_))
        if not segs:
            return ops, op_nodes, None
        new_ops, new_nodes, pos, growth = [], [], 0, []
        x_patch = []
        for a, b, s_ in segs:
            new_ops += ops[pos:a]; new_nodes += op_nodes[pos:a]
            for j in range(s_):
                for i in range(a, b):
                    sub, x_off = self._sub_op(ops[i], j, s_)
                    if i == 0 and self.fused_input:
                        x_patch.append((len(new_ops), x_off))
                    new_ops.append(sub); new_nodes.append(op_nodes[i])
            growth.append((b, (b - a) * (s_ - 1)))
            pos = b
            self.depth_first.append((a, b, s_))
        new_ops += ops[pos:]; new_nodes += op_nodes[pos:]
        if x_patch:
            self._x_patch = x_patch

        def shift(i):                                       # how far launch i of the plain list moved (launches behind a segment only)
            return sum(g for end, g in growth if i >= end)
        return new_ops, new_nodes, shift

    def _sub_op(self, op, j, s_):
        """Launch ``op`` restricted to images [j n / s, (j + 1) n / s): batch size and every batch-major pointer moved."""
        sub = YoloOp.from_buffer_copy(op)
        d = sub.conv
        n_sub = d.n // s_
        first = j * n_sub
        d.n = n_sub
        x_off = 0
        if op.kind == OP_STEM:
            x_off = first * self.rec.c_in * d.h * d.w * 4                      # the caller's float32 NCHW batch (patched per call)
        else:                                               # (a fused unit's x is also its residual: the same batch-major view)
            sub.x = op.x + first * op.conv.h * op.conv.w * op.conv.in_c_total * 2
        up = 4 if op.conv.upsample2x else 1
        m_out = first * op.conv.ho * op.conv.wo
        if op.y:
            sub.y = op.y + m_out * up * op.conv.out_c_total * 2
        if op.residual:
            sub.residual = op.residual + m_out * op.conv.res_c_total * 2
        if op.y_aux:
            sub.y_aux = op.y_aux + m_out * op.conv.aux_c_total * 2
        return sub, x_off

    def set_input_ptr(self, ptr: int):
        """The first layer reads the caller's float32 NCHW batch itself: point its launch(es) at this call's batch."""
        for idx, off in self._x_patch:
            self.op_array[idx].x = ptr + off

    def _build_ops_f32(self):
        """fp32 mode: one plain launch per recorded layer (yolo_conv2d_f32_fwd / yolo_maxpool_f32_fwd); concat, upsample and
        residual placement are the same epilogue options as in the bf16 list, nothing else is fused."""
        ops, op_nodes = [], []
        for nd in self.rec.nodes:
            if nd.kind == "conv":
                x, y = nd.srcs[0], nd.outs[0]
                w, b = nd.attrs["weight"]
                wp, bp, kpad, cout_pad = K.pack_conv_weight_f32(w, b, x.c)
                wp, bp = self._dev(wp), self._dev(bp)
                up = nd.attrs.get("up_into")
                dst = up if up is not None else y
                res = nd.srcs[1] if nd.attrs["has_res"] else None
                aux = nd.outs[1] if len(nd.outs) > 1 else None
                op = YoloOp()
                op.kind = OP_CONV_F32
                op.conv = K.conv_desc(n=x.n, h=x.h, w=x.w, cin=x.c, in_c_total=x.buf.c_total, in_c_offset=x.c_offset,
                                      cout=w.shape[0], out_c_total=dst.buf.c_total, out_c_offset=dst.c_offset,
                                      ksize=w.shape[2], stride=nd.attrs["stride"], act=_ACT[nd.attrs["act"]], kpad=kpad,
                                      cout_pad=cout_pad, upsample2x=1 if up is not None else 0, out_dtype=DT_F32,
                                      pad=nd.attrs.get("pad"),
                                      res=(res.buf.c_total, res.c_offset) if res is not None else (0, 0),
                                      aux=(aux.buf.c_total, aux.c_offset) if aux is not None else (0, 0))
                op.x, op.w, op.bias = x.buf.tensor.data_ptr(), wp.data_ptr(), bp.data_ptr()
                op.residual = res.buf.tensor.data_ptr() if res is not None else None
                op.y = dst.buf.tensor.data_ptr()
                op.y_aux = aux.buf.tensor.data_ptr() if aux is not None else None
                ops.append(op); op_nodes.append(nd)
            elif nd.kind in ("pool", "spp"):
                x, y = nd.srcs[0], nd.outs[0]
                if nd.kind == "pool":
                    jobs = [(nd.attrs["size"], nd.attrs["stride"], nd.attrs["pad"], nd.attrs["dil"], y.c_offset)]
                else:            # cat([p5, p9, p13, x]) (yolov3_spp.py:129): x already sits in slice [3c, 4c)
                    jobs = [(k, 1, k // 2, 1, y.c_offset + lvl * x.c) for lvl, k in enumerate((5, 9, 13))]
                for k, st, pad, dil, out_off in jobs:
                    op = YoloOp()
                    op.kind = OP_MAXPOOL_F32
                    op.x, op.y = x.buf.tensor.data_ptr(), y.buf.tensor.data_ptr()
                    d = op.conv
                    d.n, d.h, d.w, d.cin = x.n, x.h, x.w, x.c
                    d.in_c_total, d.in_c_offset, d.ho, d.wo = x.buf.c_total, x.c_offset, y.h, y.w
                    d.out_c_total, d.out_c_offset = y.buf.c_total, out_off
                    d.ksize, d.stride, d.pad, d.upsample2x = k, st, pad, dil
                    ops.append(op); op_nodes.append(nd)
            elif nd.kind in ("dwconv", "shuffle", "se"):
                raise NotImplementedError("precision='fp32' covers the Darknet families (YOLOv3-SPP / -tiny / YOLOv3 / Lite); "
                                          f"no fp32 kernel for '{nd.kind}' layers")
        self.n_ops = len(ops)
        self.op_nodes = op_nodes
        self.op_array = (YoloOp * len(ops))(*ops)

    # -- execution -----------------------------------------------------------------------------------
    @property
    def input_buffer(self) -> torch.Tensor:
        return self.rec.input.buf.tensor

    def feed(self, x: torch.Tensor):
        """Hand the float32 NCHW batch to the layer list: either the first conv reads it directly
        (pointer patched into its op) or it is packed to NHWC bf16 first."""
        if self.fused_input:
            if x.dtype != torch.float32 or not x.is_contiguous() or tuple(x.shape[1:]) != (self.rec.c_in, self.rec.input.h, self.rec.input.w):
                raise RuntimeError("input must be contiguous float32 NCHW of the planned shape")
            self.set_input_ptr(x.data_ptr())
        elif self.f32:
            K.pack_input_f32(x, self.input_buffer)
        else:
            K.pack_input(x, self.input_buffer)

    def _launch(self, x: torch.Tensor, io: torch.Tensor, ps, timing=None, before_io=None, compact=None):
        """pack -> layer list -> decodes on the current stream.  ``timing`` = (start, end) torch events
        recorded around the layer list (bench.py's roofline measurement).  ``before_io()`` is called right before the first
        launch that writes ``io`` (a pipelined caller makes the stream wait there for the previous batch's NMS, which reads it).
        ``compact`` = (workspace, conf_thres, min_wh): the heads filter their rows into the compact NMS workspace instead of
        storing ``io`` (which may be None then); ``before_io`` then guards the workspace the same way."""
        self.feed(x)
        self._bind_outputs(io, ps, compact)
        if timing is not None:
            timing[0].record()
        fused = [hd["op"] for hd in self.heads if hd["op"] is not None]
        k = min(fused) if fused else self.n_ops
        if before_io is None or k == 0:
            if before_io is not None:
                before_io()
            K.run_ops(self.op_array, self.n_ops)
        else:
            K.run_ops(self.op_array, k)
            before_io()
            if k < self.n_ops:
                K.run_ops(C.cast(C.byref(self.op_array, k * C.sizeof(YoloOp)), C.POINTER(YoloOp)), self.n_ops - k)
        if timing is not None:
            timing[1].record()
        self._decode_unfused(io, ps)

    @property
    def compact_ok(self) -> bool:
        """Every head decodes in its conv epilogue (bf16 mode): detect() can take the compact NMS form (no io)."""
        return not self.f32 and all(hd["op"] is not None for hd in self.heads)

    def compact_workspace(self) -> torch.Tensor:
        """This plan's compact NMS workspace (survivor counts, keys, staging, records: yolo_nms_compact_workspace_bytes), made once."""
        ws = self.__dict__.get("_compact_ws")
        if ws is None:
            ws = self._compact_ws = torch.empty(K.nms_compact_workspace_bytes(self.rec.input.n, self.rows_total, self.n_class),
                                                dtype=torch.uint8, device=self.device)
        return ws

    def _bind_outputs(self, io, ps, compact=None):
        """Heads that decode in their conv epilogue write io / p themselves: patch this call's buffers into their ops.
        ``compact`` = (workspace, conf_thres, min_wh): they filter into the compact NMS workspace instead and store no io."""
        if compact is not None:
            if not self.compact_ok:
                raise RuntimeError("the compact NMS form needs every head to decode in its conv epilogue")
            ws, conf, min_wh = compact
            for hd, p in zip(self.heads, ps):
                op = self.op_array[hd["op"]]
                op.y, op.y_aux = None, (p.data_ptr() if p is not None else None)
                op.workspace, op.ws_bytes = ws.data_ptr(), ws.numel()
                op.head_filter_conf, op.head_filter_min_wh = float(conf), float(min_wh)
            self._bound = (ws.data_ptr(), float(conf), tuple(None if p is None else p.data_ptr() for p in ps))
            return
        if io.shape[1] != self.rows_total or not io.is_contiguous():
            raise RuntimeError("io must be a contiguous [bs, rows_total, 5+nc] tensor")
        for hd, p in zip(self.heads, ps):
            if hd["op"] is not None:
                op = self.op_array[hd["op"]]
                op.y, op.y_aux = io.data_ptr(), (p.data_ptr() if p is not None else None)
                op.workspace, op.ws_bytes = None, 0
        self._bound = (io.data_ptr(), None, tuple(None if p is None else p.data_ptr() for p in ps))

    def _decode_unfused(self, io, ps):
        for hd, p in zip(self.heads, ps):
            if hd["op"] is None:
                K.decode(hd["sym"].buf.tensor, hd["anchors"], self.n_class, hd["stride"], io, hd["row"], p)

    def launch_detect(self, x, io, ps, nms_out, conf_thres, nms_thres, timing=None, join=True, after_nms=None, whole_batch=False,
                      wait_for=None, compact=False, cu_partition=False):
        """forward + decode + MERGE-NMS into caller-provided buffers; no host sync (``cu_partition`` / ``whole_batch`` / ``join`` are
        the StreamedPlan's: one list on the current stream has nothing to partition).  ``compact=True``: the compact NMS form
        (``io`` is not written and may be None; include/yolo_hip.h yolo_head_decode_filter_fwd) where the plan allows it.
        nms_out = (dets [bs,cap,7] f32, idx [bs,cap] i32, count [bs] i32).  ``timing``: one (start, end)
        event pair per stream, recorded around the conv launch list.  ``after_nms(i, lo, hi)`` is called in
        the context of stream i right after the NMS launch of images [lo, hi) (see distributed.PipelinedGather)."""
        from .utils.utils import MAX_PER_CLASS, MIN_WH, nms_launch
        if compact and self.compact_ok:
            ws = self.compact_workspace()
            self._launch(x, None, ps, timing=timing[0] if timing else None, compact=(ws, conf_thres, MIN_WH))
            K.nms_merge_compact(ws, x.shape[0], self.rows_total, self.n_class, nms_thres, *nms_out, max_per_class=MAX_PER_CLASS)
        else:
            self._launch(x, io, ps, timing=timing[0] if timing else None)
            nms_launch(io, conf_thres, nms_thres, nms_out, slot=0)
        if after_nms is not None:
            after_nms(0, 0, x.shape[0])

    n_streams = 1

    def conv_flops(self) -> float:
        """Exact algorithmic FLOPs of the recorded conv launches (2*M*Cout*K with logical sizes)."""
        total = 0.0
        first = True
        for i in range(self.n_ops):
            op = self.op_array[i]
            d = op.conv
            if op.kind in (OP_CONV, OP_CONV1_NCHW, OP_CONV1_POOL, OP_CONV_POOL, OP_HEAD_DECODE, OP_CONV_F32):
                cin = self.rec.c_in if first else d.cin      # the first layer's 3 -> 8 channel pad is not work
                first = False
                total += 2.0 * d.n * d.ho * d.wo * d.cout * d.ksize * d.ksize * cin
            elif op.kind == OP_STEM:                         # conv1 (real input channels) + the stride-2 conv
                first = False
                total += 2.0 * d.n * d.h * d.w * 32 * 9 * self.rec.c_in + 2.0 * d.n * d.ho * d.wo * 64 * 9 * 32
            elif op.kind == OP_RESUNIT:                      # 1x1 C->C/2 plus 3x3 C/2->C (the halo recompute is not work)
                total += 2.0 * d.n * d.h * d.w * (d.cout * d.cin) * 10
            elif op.kind == OP_DWCONV:
                total += 2.0 * d.n * d.ho * d.wo * d.cin * 9
            elif op.kind == OP_MBCONV:                       # expand at the input size, depthwise + projection at the output size
                hid = op.kpad_pre
                total += (2.0 * d.n * d.h * d.w * d.cin * hid if op.w_pre else 0.0) + 2.0 * d.n * d.ho * d.wo * hid * (9 + d.cout)
        return total

    def algorithmic_bytes(self, detect: bool = False) -> float:
        """HBM bytes one pass must move if every tensor that exists in HBM is read once and written once:
        per launch its input view, its output (x4 for a 2x2-replicated store, fp32 head rows + decoded rows for a fused head),
        the residual and the pre-add copy, and its weights.  Fused launches count only what crosses the chip boundary.
        ``detect=True``: the pass of ``detect()`` in the compact NMS form - a head writes one 8-byte key per row instead of p and io."""
        total = 0.0
        first = True
        for i in range(self.n_ops):
            op = self.op_array[i]
            d = op.conv
            m_in, m_out = d.n * d.h * d.w, d.n * d.ho * d.wo
            if op.kind in (OP_CONV, OP_CONV1_NCHW, OP_CONV1_POOL, OP_CONV_POOL, OP_HEAD_DECODE, OP_CONV_F32):
                x_b = m_in * (self.rec.c_in * 4 if (first and op.kind in (OP_CONV1_NCHW, OP_CONV1_POOL)) else d.cin * (4 if op.kind == OP_CONV_F32 else 2))
                first = False
                pooled = 4 if op.kind in (OP_CONV1_POOL, OP_CONV_POOL) else 1       # only the 2x2-pooled map is written
                y_b = m_out * d.cout * (4 if d.out_dtype else 2) * (4 if d.upsample2x else 1) / pooled
                if op.kind == OP_HEAD_DECODE:
                    y_b = 2.0 * m_out * d.cout * 4                                   # p (raw) + io (decoded), fp32
                    if detect:
                        y_b = m_out * op.head_na * 8.0                               # one sort key per (pixel, anchor) row
                total += x_b + y_b + d.cout * d.ksize * d.ksize * d.cin * 2
                total += (m_out * d.cout * 2 if op.residual else 0) + (m_out * d.cout * 2 if op.y_aux else 0)
            elif op.kind == OP_STEM:
                first = False
                total += m_in * self.rec.c_in * 4 + m_out * 64 * 2
            elif op.kind == OP_RESUNIT:
                total += m_in * d.cout * 2 * (3 if op.y_aux else 2)
            elif op.kind == OP_MBCONV:
                total += m_in * d.cin * 2 + m_out * d.cout * 2
            elif op.kind in (OP_MAXPOOL, OP_DWCONV, OP_MAXPOOL_F32):
                total += (m_in + m_out) * d.cin * (4 if op.kind == OP_MAXPOOL_F32 else 2)
            elif op.kind == OP_SE:
                total += 3.0 * m_in * d.cin * 2                                      # pooled once, read again for the rescale, written
            elif op.kind == OP_SPP:
                total += m_in * d.cin * 2 * 4                                        # reads c, writes the three pooled copies
            elif op.kind == OP_SHUFFLE:
                total += 2.0 * m_in * d.cin * 2
        return total

    def new_outputs(self, want_p: bool = True):
        """io and the raw head tensors p.  ``want_p=False`` (detect(): NMS reads io only) leaves p out: the head kernels then
        skip its store (0.87 GB per 32 SPP-640 images) - p is ``(None, ...)``."""
        n = self.rec.input.n
        no = self.n_class + 5
        io = torch.empty((n, self.rows_total, no), dtype=torch.float32, device=self.device)
        ps = tuple(torch.empty((n, hd["na"], hd["sym"].h, hd["sym"].w, no), dtype=torch.float32, device=self.device) if want_p else None
                   for hd in self.heads)
        return io, ps

    def run(self, x: torch.Tensor):
        """Eager launch on the current stream; returns fresh (io, p) tensors."""
        io, ps = self.new_outputs()
        self._launch(x, io, ps)
        return io, ps

    def activation_bytes(self) -> int:
        return sum(t.numel() * t.element_size() for t in {b.tensor.data_ptr(): b.tensor for b in self._bufs}.values())

    # -- HIP-graph replay ------------------------------------------------------------------------------
    def run_graph(self, x: torch.Tensor):
        """Replay the whole forward (pack + layers + decodes) as one captured HIP graph.
        Returns STATIC output tensors: they are overwritten by the next call."""
        if self._graph is None:
            self._static_x = torch.empty_like(x)
            self._static_out = self.new_outputs()
            side = torch.cuda.Stream(device=self.device)
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):                    # warm-up outside capture
                self._static_x.copy_(x)
                self._launch(self._static_x, *self._static_out)
            torch.cuda.current_stream().wait_stream(side)
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                self._launch(self._static_x, *self._static_out)
            self._graph = graph
        self._static_x.copy_(x)
        self._graph.replay()
        return self._static_out


_masked_streams = {}
_streamed_plans = None      # weakref.WeakSet of the StreamedPlans alive (they may hold references to the masked streams)


def destroy_masked_streams():
    """Drain and destroy every CU-masked HIP stream this process created (hipExtStreamCreateWithCUMask has no owner in torch:
    an ExternalStream never destroys its handle).  Registered with ``atexit``: streams still alive when the HIP runtime and a
    profiler's tool library unwind their own state crashed ``rocprofv3`` runs of the partitioned bench inside ``__cxa_finalize``
    (VERDICT r3 item 6).  May also be called between batches (nothing in flight on the partitioned streams - they are drained
    here): every live ``StreamedPlan`` forgets its references to the destroyed handles, so its next partitioned call creates
    new streams instead of enqueueing on destroyed ones."""
    streams = [st for group in _masked_streams.values() for st in group]
    _masked_streams.clear()
    for sp in list(_streamed_plans or ()):
        for attr in ("_pipe_streams", "_full_streams"):
            held = getattr(sp, attr, None)
            if held is not None and any(h is st for h in held for st in streams):
                setattr(sp, attr, None)
                if attr == "_full_streams":
                    sp._full_mode = None
    for st in streams:
        try:
            st.synchronize()
            K.check(K.load().yolo_stream_destroy(C.c_void_p(st.cuda_stream)), "stream_destroy")
        except Exception:                                  # noqa: BLE001 (interpreter teardown: nothing left to report to)
            pass


atexit.register(destroy_masked_streams)


class _launch_cus:
    """with _launch_cus(n): the launches of this thread size their grids for n compute units (None: leave as is)."""

    def __init__(self, n):
        self.n = n

    def __enter__(self):
        self.old = K.set_launch_cus(self.n) if self.n else None

    def __exit__(self, *exc):
        if self.old is not None:
            K.set_launch_cus(self.old)
        return False


class StreamedPlan:
    """The batch split into S contiguous sub-batches, each with its own Plan, run on S HIP streams.

    Images are independent, so results are identical; what changes is occupancy: the heavy layers launch
    200..800 tiles of 256x256 on 256 CUs, i.e. 1.56 or 3.1 "rounds", and the partial last round leaves a
    quarter of the chip idle.  Two kernels from two streams fill each other's tails (measured: -9.5 % on
    the SPP-640 bs=32 layer list; four streams are no better than two).  Each stream owns half of every XCD's
    compute units (_make_streams)."""

    def __init__(self, make_plan, bs: int, n_streams: int, device):
        assert bs % n_streams == 0
        self.sub = bs // n_streams
        self.subs = [make_plan(self.sub) for _ in range(n_streams)]
        # joined calls (forward(), detect()): ordinary streams - both sub-batches start together and stay in step, layer for layer,
        # so a CU partition has nothing complementary to overlap and only costs (-6 % on detect()).  Free-running pipelines
        # (launch_detect(join=False): bench.py, a serving loop) drift apart: there each stream owns half of every XCD.
        # The partitioned streams are created on the first pipelined call only: they are BLOCKING streams
        # (hipExtStreamCreateWithCUMask takes no flags), and while any exists every operation on the NULL stream - torch's default
        # current stream - pays for the implicit synchronisation with them (host-bound detect() loop: -5..9 %).
        self.streams = [torch.cuda.Stream(device=device) for _ in range(n_streams)]
        self._pipe_streams = None
        # pipelined calls: the NMS of a sub-batch runs on a stream of its own (all CUs), so the pipeline's stream goes straight on
        # to the next batch's layer list; it waits for that NMS only in front of its first head launch (the next writer of io)
        self._make_plan, self._full, self._full_streams, self._batches = make_plan, None, None, 0
        self._full_mode = None          # cu_partition flag of the last whole-batch call (launch_detect)
        self._fast_events = None
        self._nms_stream = None
        self._nms_streams = None
        self._heads_done = [torch.cuda.Event() for _ in range(n_streams)]
        self._nms_done = [None] * n_streams
        self._marks = [torch.cuda.Event() for _ in range(n_streams)]
        p0 = self.subs[0]
        self.device, self.n_class, self.img_size = device, p0.n_class, p0.img_size
        self.heads, self.rows_total = p0.heads, p0.rows_total
        self.bs = bs
        self._graph = None
        global _streamed_plans
        if _streamed_plans is None:
            import weakref
            _streamed_plans = weakref.WeakSet()
        _streamed_plans.add(self)

    @staticmethod
    def _make_streams(n_streams, device, flops_per_launch=0.0):
        """Each sub-batch stream owns its own compute units: stream i gets CUs [i, i+1) * per_xcd / S of every XCD (CU-mask bit b
        lies on XCD b % 8), so two layer lists really run side by side - one stream's HBM-bound phases (1x1 layers, epilogues)
        under the other's MFMA-dense ones - instead of time-slicing the workgroup slots of the whole chip (a 3x3 launch takes
        every register of every CU it lands on).  It pays where the launches are MFMA-dense and about one round of workgroups
        long - the chip then runs against its socket power limit and half the CUs hold a higher clock.
        Measured images/s, split vs shared: SPP-640 bs=32 +2.6 %, bs=16 +1.5 %, bs=64 -0.7 %; SPP-416 bs=32 -9 %; tiny-416 -2.5 %,
        MobileNetV2-tiny -4.6 % (small launches).  "auto" therefore splits between 16 and 96 GFLOP per launch of a pipeline
        (SPP-640 at 8..32 images per stream); whole XCDs per stream ("xcd") lose 1 %.
        YOLO_CU_PARTITION = auto (default) | split | xcd | off."""
        mode = os.environ.get("YOLO_CU_PARTITION", "auto")
        if mode == "auto":      # (round 3: the whole-batch pipelines of SPP-640 x 32 - 69 GFLOP per launch - gain 1.0 % too: profiles/r03_cu_partition_ab.txt)
            mode = "split" if 16e9 <= flops_per_launch < 96e9 else "off"
        n_cu = torch.cuda.get_device_properties(device).multi_processor_count
        n_xcd = 8
        per = n_cu // n_xcd
        if mode in ("off", "0", "") or n_streams < 2 or n_cu % n_xcd or per % n_streams:
            return [torch.cuda.Stream(device=device) for _ in range(n_streams)]
        if mode == "split":
            sets = [[b for b in range(n_cu) if (b // n_xcd) * n_streams // per == i] for i in range(n_streams)]
        elif mode == "xcd" and n_xcd % n_streams == 0:
            sets = [[b for b in range(n_cu) if (b % n_xcd) * n_streams // n_xcd == i] for i in range(n_streams)]
        else:
            raise RuntimeError(f"YOLO_CU_PARTITION={mode!r}: expected off, split or xcd")
        idx = torch.device(device).index
        key = (torch.cuda.current_device() if idx is None else idx, n_streams, mode)
        if key not in _masked_streams:                      # HIP streams with a CU mask are created once per device and shared by the plans
            _masked_streams[key] = [K.cu_masked_stream(s, device) for s in sets]
        return _masked_streams[key]

    def new_outputs(self, want_p: bool = True):
        no = self.n_class + 5
        io = torch.empty((self.bs, self.rows_total, no), dtype=torch.float32, device=self.device)
        ps = tuple(torch.empty((self.bs, hd["na"], hd["sym"].h, hd["sym"].w, no), dtype=torch.float32, device=self.device) if want_p else None
                   for hd in self.heads)
        return io, ps

    def _launch(self, x, io, ps, timing=None):
        cur = torch.cuda.current_stream()
        if timing is not None:
            timing[0].record(cur)
        self._fork(cur)
        for i, (pl, st) in enumerate(zip(self.subs, self.streams)):
            lo, hi = i * self.sub, (i + 1) * self.sub
            with torch.cuda.stream(st):
                sub_ps = tuple(None if p is None else p[lo:hi] for p in ps)
                pl.feed(x[lo:hi])
                pl._bind_outputs(io[lo:hi], sub_ps)
                K.run_ops(pl.op_array, pl.n_ops)
                self._marks[i].record(st)
                pl._decode_unfused(io[lo:hi], sub_ps)
        if timing is not None:
            for m in self._marks:
                cur.wait_event(m)
            timing[1].record(cur)          # every stream's layer list is done (decodes may still run)
        for st in self.streams:
            cur.wait_stream(st)

    @property
    def pipe_streams(self):
        if self._pipe_streams is None:
            made = self._make_streams(len(self.streams), self.device, self.subs[0].conv_flops() / max(1, self.subs[0].n_ops))
            # no partition: the SAME streams as the joined calls (more streams than hardware queues - 4 by default - end up sharing
            # a queue, and two sub-batch pipelines on one queue run one after the other)
            self._pipe_streams = made if type(made[0]).__name__ == "ExternalStream" else self.streams
        return self._pipe_streams

    def _fork(self, cur):
        """Every sub-batch stream waits for the calling stream - ONE event, recorded before anything is launched.  The CU-masked
        streams are blocking streams (hipExtStreamCreateWithCUMask takes no flags): an event recorded on the NULL stream between
        the launches of two of them would wait for the first one's whole pass, and the sub-batches would run one after the other."""
        ev = torch.cuda.Event()
        ev.record(cur)
        for st in self.streams:
            st.wait_event(ev)

    def run(self, x):
        io, ps = self.new_outputs()
        self._launch(x, io, ps)
        return io, ps

    @property
    def n_streams(self):
        return len(self.streams)

    def launch_detect(self, x, io, ps, nms_out, conf_thres, nms_thres, timing=None, join=True, after_nms=None, whole_batch=False,
                      wait_for=None, cu_partition=False, compact=False):
        """``compact=True``: the compact NMS form - the heads filter their own rows, ``io`` is neither written nor read and may be
        None (include/yolo_hip.h yolo_head_decode_filter_fwd; +1.7 % on SPP-640 x 32 from the io store alone) - where the plans allow it.

        ``wait_for``: an event every pipeline of this call waits for before its first launch (pipelined calls do not wait for
        the calling stream: pass the event that says ``x`` is ready when another stream produced it).

        ``cu_partition=True`` (whole-batch pipelines only): each pipeline launches on a stream that owns half of every XCD's CUs
        when ``_make_streams`` finds the launches in the range where that pays (+0.4..1 % on SPP-640 x 32).  Opt-in, for loops that
        only queue launches (bench.py): HIP creates CU-masked streams as BLOCKING streams, so every operation on the default stream
        between two calls - the event that says ``x`` is ready, an H2D copy of the next batch - is a barrier between the
        pipelines, and each of them then has half the chip: ``detect_stream()`` fell from 6,190 to 3,850 images/s with it.

        ``whole_batch=True`` (pipelined calls only): this call's WHOLE batch goes down ONE pipeline and successive calls
        alternate between the pipelines - two batches in flight instead of two halves of one.  The launches are twice as large
        (SPP-640: 32 images instead of 16: +3.4 % images/s), a batch takes twice as long to come out, and the caller must hand
        consecutive calls DIFFERENT io / nms_out buffers (a buffer set may be reused every ``n_streams`` calls: the head launches
        of the later batch wait for the NMS of the earlier one).  ``timing`` then records its first event pair only;
        ``after_nms(k, 0, bs)`` gets the pipeline index.

        Otherwise:
        Each stream runs the WHOLE pipeline (pack/conv1 -> layers -> decode -> NMS) of its sub-batch into
        slices of the shared buffers.  With ``join=False`` the calling stream neither waits for the previous
        work nor for this one: successive calls then form S free-running pipelines (in-order per stream, so
        buffer reuse is safe) — the caller synchronises before reading results, and before a joined call on the same plan
        (the pipelines run on their own, CU-partitioned streams: ``pipe_streams``)."""
        from .utils.utils import MAX_PER_CLASS, MIN_WH, nms_launch
        cur = torch.cuda.current_stream()
        whole = whole_batch and not join
        if whole:
            if self._full is None:                           # one whole-batch plan per pipeline, built on first use
                with torch.cuda.device(self.device):
                    self._full = [self._make_plan(self.bs) for _ in self.streams]
            if cu_partition and self._full_streams is None:
                made = self._make_streams(len(self.streams), self.device, self._full[0].conv_flops() / max(1, self._full[0].n_ops))
                self._full_streams = made if type(made[0]).__name__ == "ExternalStream" else self.streams
            streams = self._full_streams if cu_partition else self.streams
            # the whole-batch plans (activations, events, batch counter) are shared by the two stream sets, which are not ordered
            # with each other: a call that flips ``cu_partition`` drains the device first (a rare transition - bench.py
            # makes it once, between its detect_stream() timing and its step loop)
            if self._full_mode is not None and self._full_mode != bool(cu_partition):
                torch.cuda.synchronize(self.device)
            self._full_mode = bool(cu_partition)
            k = self._batches % len(streams)
            self._batches += 1
            work = [(k, self._full[k], streams[k], 0, self.bs)]
        else:
            streams = self.streams if join else self.pipe_streams
            work = [(i, pl, st, i * self.sub, (i + 1) * self.sub) for i, (pl, st) in enumerate(zip(self.subs, streams))]
        side_nms = not join and os.environ.get("YOLO_NMS_STREAM", "1") != "0"
        if side_nms:
            self._make_nms_streams()
        if join:
            self._fork(cur)
        # a partitioned pipeline launches on its share of the CUs: the tile rules size their grids against it
        share = None
        if not join and type(streams[0]).__name__ == "ExternalStream" and os.environ.get("YOLO_RULES_FOR_SHARE", "1") != "0":
            share = torch.cuda.get_device_properties(self.device).multi_processor_count // len(streams)
        for i, pl, st, lo, hi in work:
            sub_ps = tuple(None if p is None else p[lo:hi] for p in ps)
            tm = (timing[0 if whole else i] if timing else None)
            if wait_for is not None:
                st.wait_event(wait_for)
            cmp_ = (pl.compact_workspace(), conf_thres, MIN_WH) if (compact and pl.compact_ok) else None
            sub_io = None if cmp_ is not None else io[lo:hi]
            with torch.cuda.stream(st), _launch_cus(share):
                if not side_nms:
                    pl._launch(x[lo:hi], sub_io, sub_ps, timing=tm, compact=cmp_)
                else:
                    prev = self._nms_done[i]
                    pl._launch(x[lo:hi], sub_io, sub_ps, timing=tm, compact=cmp_,
                               before_io=(lambda prev=prev, st=st: st.wait_event(prev)) if prev is not None else None)
                    self._heads_done[i].record(st)
            nst = self._nms_streams[i % len(self._nms_streams)] if side_nms else st
            with torch.cuda.stream(nst):
                if side_nms:
                    nst.wait_event(self._heads_done[i])
                if cmp_ is not None:
                    K.nms_merge_compact(cmp_[0], hi - lo, pl.rows_total, pl.n_class, nms_thres, *(t[lo:hi] for t in nms_out),
                                        max_per_class=MAX_PER_CLASS)
                else:
                    nms_launch(io[lo:hi], conf_thres, nms_thres, tuple(t[lo:hi] for t in nms_out), slot=i)
                if after_nms is not None:
                    after_nms(i, lo, hi)
                if side_nms:
                    if self._nms_done[i] is None:
                        self._nms_done[i] = torch.cuda.Event()
                    self._nms_done[i].record(nst)
        if join:
            for st in self.streams:
                cur.wait_stream(st)

    def _make_nms_streams(self):
        """The NMS side stream, created at HIGH priority and shared by the pipelines: the NMS of a batch (one 1024-thread workgroup
        per image, latency-bound) is the tail of that batch; at equal priority it waited for CUs behind the other pipeline's queued
        conv workgroups.  ONE stream, not one per pipeline (YOLO_NMS_STREAMS=per-pipeline: no faster on any workload, 88.3 k vs
        87.1 k images/s on YOLOv3-tiny): every HIP stream that has been used owns a hardware queue, and with the default stream,
        the two pipeline streams, two NMS streams and - created last - the two CU-masked streams of ``launch_detect(cu_partition=True)``
        the process held more queues than the scheduler runs side by side: the partitioned loop dropped from 6,600 to 5,720 images/s
        whenever ``detect_stream()`` had run before it (tools/dbg/stream_then_loop.py; GPU_MAX_HW_QUEUES = 12 / 16 changed nothing,
        one queue fewer restored it)."""
        if self._nms_streams is None:
            prio = int(os.environ.get("YOLO_NMS_PRIORITY", "-1"))
            if os.environ.get("YOLO_NMS_STREAMS", "1") == "1":
                one = torch.cuda.Stream(device=self.device, priority=prio)
                self._nms_streams = [one for _ in self.streams]
            else:
                self._nms_streams = [torch.cuda.Stream(device=self.device, priority=prio) for _ in self.streams]
            self._nms_stream = self._nms_streams[0]

    # -- the one-call pipeline step (yolo_pipeline_step): what detect_stream() launches per batch ------------------------------
    def fast_pipeline(self, slot: int, io, out, conf_thres, nms_thres):
        """A ``FastStep`` for ring slot ``slot`` (``out`` = (dets, idx, count); ``io`` = None: the compact NMS form, no io at all;
        a tensor: the heads store io there and the plain NMS reads it): the whole-batch launch list of pipeline ``slot % S`` + NMS
        as ONE FFI call per batch, or None when this plan cannot take it (a head that decodes in a launch of its own, an input the
        first layer does not read itself: then ``launch_detect`` does the same work)."""
        k = slot % len(self.streams)
        if self._full is None:
            with torch.cuda.device(self.device):
                self._full = [self._make_plan(self.bs) for _ in self.streams]
        pl = self._full[k]
        if not pl.fused_input or any(hd["op"] is None for hd in pl.heads) or pl.f32:
            return None
        self._make_nms_streams()
        self._fast_events = True                                  # (tests: the fast path was taken)
        return FastStep(self, k, pl, io, out, conf_thres, nms_thres)

    def detect_step(self, conf_thres, nms_thres):
        """The ``FastStep`` a lone ``detect()`` call uses (its own io / output buffers, made once per plan), or None.  One
        whole-batch list + NMS through one FFI call beats two joined half-batch lists for a host-synchronous call: SPP-640 x 32
        5.45 vs 5.87 ms, YOLOv3-tiny x 32 0.71 vs 0.90 ms (tools/detect_modes.py, three interleaved rounds on one box)."""
        fs = self.__dict__.get("_detect_fast")
        if fs is None:
            from .utils.utils import nms_capacity
            with torch.cuda.device(self.device):
                cap = nms_capacity(self.rows_total, self.n_class)
                out = (torch.empty((self.bs, cap, 7), dtype=torch.float32, device=self.device),
                       torch.empty((self.bs, cap), dtype=torch.int32, device=self.device),
                       torch.empty((self.bs,), dtype=torch.int32, device=self.device))
                fs = self.fast_pipeline(0, None, out, conf_thres, nms_thres) or False      # (compact NMS form: no io)
            self._detect_fast = fs
        if fs is False:
            return None
        fs.step.conf_thres, fs.step.nms_thres = float(conf_thres), float(nms_thres)
        return fs

    run_graph = None   # bound below (shares Plan.run_graph's capture logic)

    def conv_flops(self) -> float:
        return sum(p.conv_flops() for p in self.subs)

    def activation_bytes(self) -> int:
        return sum(p.activation_bytes() for p in self.subs)

    def algorithmic_bytes(self, detect: bool = False) -> float:
        return sum(p.algorithmic_bytes(detect) for p in self.subs)


StreamedPlan.run_graph = Plan.run_graph


class FastStep:
    """One ring slot of ``detect_stream()``: a prebuilt ``YoloPipeStep`` (include/yolo_hip.h) - per batch the host patches the input
    pointer, records "x is ready" and makes ONE call into the library, which enqueues the layer list on the pipeline's stream, the
    NMS on the side stream, the count read-back into pinned memory and the events between them.  Slot and pipeline are tied
    (slot % S), so a slot's buffers are written by one pipeline only and its head launches wait for that pipeline's previous NMS."""

    def __init__(self, sp: "StreamedPlan", k: int, pl: Plan, io, out, conf_thres, nms_thres):
        from ._lib import YoloPipeStep
        from .utils.utils import MAX_PER_CLASS, MIN_WH
        self.sp, self.k, self.pl, self.io, self.out = sp, k, pl, io, out
        bs, rows, nc = sp.bs, pl.rows_total, pl.n_class
        self.compact = (torch.empty(K.nms_compact_workspace_bytes(bs, rows, nc), dtype=torch.uint8, device=sp.device), conf_thres,
                        MIN_WH) if io is None else None            # (a workspace per slot: two slots of a pipeline may be in flight)
        self.ws = self.compact[0] if io is None else torch.empty(K.nms_workspace_bytes(bs, rows, nc), dtype=torch.uint8, device=sp.device)
        self.count_host = torch.empty(bs, dtype=torch.int32).pin_memory()
        self.count_np = self.count_host.numpy()
        self.ready, self.done = K.Event(), K.Event()
        heads_done, nms_done = K.Event(), K.Event()              # per slot: the slot's head launches wait for the slot's previous NMS
        self._events = (heads_done, nms_done)
        self._launched = False
        fused = [hd["op"] for hd in pl.heads]
        st = YoloPipeStep()
        st.ops, st.n_ops, st.k_io = pl.op_array, pl.n_ops, min(fused)
        st.stream, st.nms_stream = sp.streams[k].cuda_stream, sp._nms_streams[k].cuda_stream
        st.wait_x, st.wait_io, st.heads_done, st.nms_done, st.done = self.ready.handle, None, heads_done.handle, nms_done.handle, self.done.handle
        st.io, st.bs, st.rows, st.nc, st.max_per_class = (None if io is None else io.data_ptr()), bs, rows, nc, MAX_PER_CLASS
        st.conf_thres, st.nms_thres, st.min_wh, st.cap = float(conf_thres), float(nms_thres), MIN_WH, out[0].shape[1]
        st.out_dets, st.out_idx, st.out_count = out[0].data_ptr(), out[1].data_ptr(), out[2].data_ptr()
        st.workspace, st.workspace_bytes, st.count_host = self.ws.data_ptr(), self.ws.numel(), self.count_host.data_ptr()
        self.step = st
        self._nms_done = nms_done
        self._ps = tuple(None for _ in pl.heads)
        self._shape = tuple(pl.rec.input.buf.tensor.shape) if pl.rec.input.buf is not None and pl.rec.input.buf.tensor is not None else None
        self._x_shape = (pl.rec.c_in, pl.rec.input.h, pl.rec.input.w)

    def launch(self, x: torch.Tensor):
        """Enqueue one batch (no host sync).  ``x``: contiguous float32 NCHW of the planned shape on the plan's device."""
        pl, sp = self.pl, self.sp
        if x.dtype != torch.float32 or not x.is_contiguous() or tuple(x.shape[1:]) != self._x_shape or x.shape[0] != sp.bs:
            raise RuntimeError("input must be contiguous float32 NCHW of the planned shape")
        pl.set_input_ptr(x.data_ptr())
        # (the plan may have served another caller's buffers in between, and detect() may change the threshold per call: Plan._bound
        # says what the head ops currently point at)
        if self.compact is not None:
            want = (self.compact[0].data_ptr(), float(self.step.conf_thres), self._ps)
            if pl.__dict__.get("_bound") != want:
                self.compact = (self.compact[0], float(self.step.conf_thres), self.compact[2])
                pl._bind_outputs(None, self._ps, self.compact)
        elif pl.__dict__.get("_bound") != (self.io.data_ptr(), None, self._ps):
            pl._bind_outputs(self.io, self._ps)
        # the whole-batch plans are shared with launch_detect's stream sets: same ordering rule as there
        if sp._full_mode:
            torch.cuda.synchronize(sp.device)
        sp._full_mode = False
        self.ready.record()                                     # on the caller's current stream: x was produced there
        # this slot's io was last read by the NMS of the slot's previous batch: the head launches wait for it
        self.step.wait_io = self._nms_done.handle if self._launched else None
        self._launched = True
        K.pipeline_step(self.step)

    def collect(self):
        """Wait for this slot's batch and hand out the reference's ``list[Tensor[n, 7] | None]``."""
        self.done.synchronize()
        counts = self.count_np.tolist()
        cap = self.out[0].shape[1]
        total = 0
        for b, n in enumerate(counts):
            if n > cap:
                raise RuntimeError(f"image {b}: {n} detections exceed the output capacity {cap}")
            total += n
        if total == 0:
            return [None] * len(counts)
        with torch.cuda.device(self.sp.device):                  # (the library launches on the CURRENT device's stream)
            packed = torch.empty((total, 7), dtype=torch.float32, device=self.sp.device)
            K.pack_detections(self.out[0], None, self.out[2], packed)
        parts = iter(torch.split(packed, [n for n in counts if n]))
        return [next(parts) if n else None for n in counts]


def _sym_to_nchw(s: Sym) -> torch.Tensor:
    t = s.buf.tensor[..., s.c_offset:s.c_offset + s.c]
    return t.permute(0, 3, 1, 2).float().contiguous()


def run_standalone(trace_fn, x: torch.Tensor):
    """Run a recorded sub-graph (a block, a stage) on an NCHW float32 CUDA tensor and return its
    output(s) as NCHW float32 — used by the block-level ``forward`` methods and their tests."""
    if not x.is_cuda:
        raise RuntimeError("pytorch_yolo_amd runs on a ROCm device only (no CPU fallback)")
    x = x.float().contiguous()
    bs, c, h, w = x.shape
    rec = Recorder(bs, c, h, w)
    out = trace_fn(rec, rec.input)
    with torch.cuda.device(x.device):               # the library launches on the current device's stream: make it x's
        plan = Plan(rec, x.device, n_class=0, img_size=max(h, w))
        plan.feed(x)
        K.run_ops(plan.op_array, plan.n_ops)
        if isinstance(out, (tuple, list)):
            return tuple(_sym_to_nchw(s) for s in out)
        return _sym_to_nchw(out)
