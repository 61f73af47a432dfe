"""CPU-only checks: the C-ABI library loads and exports everything include/yolo_hip.h declares, the host
mirror keeps the reference's state_dict layout, the planner fuses what DESIGN.md says it fuses, the host-side
weight preparation matches the oracle, and the product refuses to run without a GPU."""
import ctypes
import json
import os
import re

import numpy as np
import pytest
import torch

import _cases as C
from pytorch_yolo_amd import YOLOv3SPP, YOLOv3Tiny, YOLOv3TinyMobile, _lib, engine
from pytorch_yolo_amd import kernels as K
from pytorch_yolo_amd._lib import OP_MBCONV, OP_CONV_POOL, OP_CONV, OP_CONV1_NCHW, OP_CONV1_POOL, OP_HEAD_DECODE, OP_RESUNIT, OP_STEM, OP_DWCONV, OP_MAXPOOL, OP_SPP

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_functions():
    text = open(os.path.join(ROOT, "include", "yolo_hip.h")).read()
    return re.findall(r"YOLO_API\s+[\w\s\*]+?\b(yolo_\w+)\s*\(", text)


def test_library_exports_every_declared_symbol():
    names = _header_functions()
    assert len(names) >= 12 and len(set(names)) == len(names)
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/yolo_hip.h but not exported"
    assert set(names) == set(_lib.SIGNATURES), "ctypes table and header disagree"
    assert _lib.load().yolo_abi_version() == _lib.ABI_VERSION == 2


def test_struct_layout_matches_header():
    # 23 int32 fields in YoloConvDesc; YoloOp = 2 int32 + 6 pointers + desc (+4 pad) + 2 pointers + 2 int32 + 9 float + 5 int32
    # + 2 pointers
    assert ctypes.sizeof(_lib.YoloConvDesc) == 23 * 4
    assert ctypes.sizeof(_lib.YoloMbconvDesc) == 14 * 4
    assert ctypes.sizeof(_lib.YoloOp) == 8 + 6 * 8 + 23 * 4 + 4 + 2 * 8 + 2 * 4 + 9 * 4 + 5 * 4 + 2 * 8 + 3 * 8 + 2 * 4   # + split-K fields
    assert _lib.YoloOp.w_pre.offset == 152 and _lib.YoloOp.kpad_pre.offset == 168 and _lib.YoloOp.w_dw.offset == 232 and _lib.YoloOp.workspace.offset == 248 and _lib.YoloOp.splits.offset == 272
    mb = text_mb = open(os.path.join(ROOT, "include", "yolo_hip.h")).read()
    mb = mb[mb.index("typedef struct YoloMbconvDesc {"):mb.index("} YoloMbconvDesc;")]
    assert re.findall(r"\b([a-z_0-9]+)\s*[,;]", mb.split("{", 1)[1]) == [f for f, _ in _lib.YoloMbconvDesc._fields_]
    text = open(os.path.join(ROOT, "include", "yolo_hip.h")).read()
    body = text[text.index("typedef struct YoloConvDesc {"):text.index("} YoloConvDesc;")]
    fields = re.findall(r"\b([a-z_0-9]+)\s*[,;]", re.sub(r"/\*.*?\*/", "", body, flags=re.S))
    assert fields == [f for f, _ in _lib.YoloConvDesc._fields_]
    # the pipeline step (round 4): field order as declared, size as the library compiled it
    body = text[text.index("typedef struct YoloPipeStep {"):text.index("} YoloPipeStep;")]
    fields = re.findall(r"\b([a-z_0-9]+)\s*[,;]", re.sub(r"/\*.*?\*/", "", body, flags=re.S))
    assert fields == [f for f, _ in _lib.YoloPipeStep._fields_]
    for which, st in enumerate((_lib.YoloConvDesc, _lib.YoloOp, _lib.YoloMbconvDesc, _lib.YoloPipeStep)):
        assert _lib.load().yolo_abi_sizeof(which) == ctypes.sizeof(st)


def test_argument_errors_are_reported_without_a_gpu():
    lib = _lib.load()
    d = _lib.YoloConvDesc()
    rc = lib.yolo_conv2d_fwd(None, None, None, None, None, None, ctypes.byref(d), None)
    assert rc == -1 and b"null pointer" in lib.yolo_last_error()
    assert lib.yolo_nms_workspace_bytes(32, 25200, 80) > 32 * 25200 * 8
    assert lib.yolo_nms_workspace_bytes(0, 1, 1) == 0
    out = ctypes.c_void_p()
    assert lib.yolo_stream_create_cu_mask(None, 8, ctypes.byref(out)) == -1
    zeros = (ctypes.c_uint32 * 8)()
    assert lib.yolo_stream_create_cu_mask(zeros, 8, ctypes.byref(out)) == -1 and b"empty mask" in lib.yolo_last_error()
    assert lib.yolo_stream_destroy(None) == -1


def test_c_weight_packer_matches_torch_packer():
    lib = _lib.load()
    w = torch.randn(5, 3, 3, 3)
    wp, bp, kpad, cout_pad = K.pack_conv_weight(w, torch.arange(5.), 8)
    out = np.zeros((cout_pad, kpad), dtype=np.uint16)
    wc = w.contiguous().numpy()
    rc = lib.yolo_pack_conv_weight_f32(wc.ctypes.data_as(ctypes.c_void_p), 5, 3, 3, 8, cout_pad, kpad,
                                       out.ctypes.data_as(ctypes.c_void_p))
    assert rc == 0
    assert np.array_equal(out, wp.view(torch.int16).numpy().view(np.uint16))
    assert (kpad, cout_pad) == (128, 128) and bp[:5].tolist() == [0, 1, 2, 3, 4] and float(bp[5:].abs().sum()) == 0
    # layout: k = (kh*3 + kw)*cin + c
    assert float(wp[2, (1 * 3 + 2) * 8 + 1]) == float(w[2, 1, 1, 2].to(torch.bfloat16))


def test_state_dict_layout_matches_reference():
    keys = json.load(open(os.path.join(ROOT, "tests", "golden", "state_keys.json")))
    from pytorch_yolo_amd import LiteYOLOv3, YOLOv3
    for fam, model in (("tiny", YOLOv3Tiny(kernels_divider=2)),
                       ("spp", YOLOv3SPP(kernels_divider=4, anchors=C.SPP_ANCHORS)),
                       ("yolov3", YOLOv3(kernels_divider=4, anchors=C.SPP_ANCHORS)),
                       ("lite", LiteYOLOv3(kernels_divider=2, anchors=C.SPP_ANCHORS))):
        sd = model.state_dict()
        assert {k: list(v.shape) for k, v in sd.items()} == keys[fam]
        assert list(sd) == list(keys[fam])                       # same order too
        model.fuse()
        assert {k: list(v.shape) for k, v in model.state_dict().items()} == keys[fam + "_fused"]


def test_fold_bn_matches_oracle():
    from oracle.blocks import fold_bn
    from pytorch_yolo_amd.utils.torch_utils import fold_conv_bn
    w, g, b, m, v = torch.randn(6, 4, 3, 3), torch.rand(6) + 0.5, torch.randn(6), torch.randn(6), torch.rand(6) + 0.5
    w1, b1 = fold_conv_bn(w, None, g, b, m, v, 1e-5)
    w2, b2 = fold_bn(w, g, b, m, v)
    torch.testing.assert_close(w1, w2, rtol=1e-6, atol=1e-7)
    torch.testing.assert_close(b1, b2, rtol=1e-6, atol=1e-7)


def _dry_plan(model, hw, bs=1):
    rec = engine.Recorder(bs, 3, hw, hw)
    model._trace(rec, rec.input)
    return engine.Plan(rec, torch.device("cpu"), model.n_class, hw, model.precision)


def _ops(plan):
    return [plan.op_array[i] for i in range(plan.n_ops)]


def test_planner_spp_fusions():
    # (8 images: the smallest batch at which the 128-channel units at 160x160 give the fused kernel its 512 tiles; the CPU
    # allocation of the activations is why this is not the bench's 32)
    plan = _dry_plan(YOLOv3SPP(anchors=C.SPP_ANCHORS).eval(), 640, bs=8)
    ops = _ops(plan)
    kinds = [o.kind for o in ops]
    # the stem (conv1 + the stride-2 conv) reads the NCHW f32 batch itself and is ONE launch (yolo_stem_fwd)
    assert ops[0].kind == OP_STEM and plan.fused_input and kinds.count(OP_STEM) == 1
    assert (ops[0].conv.h, ops[0].conv.ho, ops[0].conv.cout, ops[0].conv.res_c_total) == (640, 320, 64, 3)
    # no add / cat / upsample / pack launches; the 64-channel residual unit at 320^2 and the two 128-channel units at 160^2 are
    # ONE launch each (yolo_resunit_fwd; round 3: engine.FUSE_RESUNIT_DEFAULT = 64 | 128)
    assert kinds.count(OP_CONV) == 65 and kinds.count(OP_RESUNIT) == 3 and kinds.count(OP_SPP) == 1 and len(ops) == 73
    units = [o for o in ops if o.kind == OP_RESUNIT]
    assert [(u.conv.cout, u.conv.cin, u.conv.h) for u in units] == [(64, 32, 320), (128, 64, 160), (128, 64, 160)]
    assert all(u.y != u.x and u.w_pre and u.bias_pre for u in units)
    # a single image leaves those units as two launches each: only the generic fused kernel would take them, and it is slower
    kinds1 = [o.kind for o in _ops(_dry_plan(YOLOv3SPP(anchors=C.SPP_ANCHORS).eval(), 640, bs=1))]
    assert kinds1.count(OP_RESUNIT) == 1 and kinds1.count(OP_CONV) == 69 and K.resunit_form(128, 1, 160, 160) == 1
    assert K.resunit_form(128, 8, 160, 160) == 3 and K.resunit_form(256, 32, 80, 80) == 3 and K.resunit_form(64, 1, 320, 320) == 3
    convs = [o for o in ops if o.kind == OP_CONV]
    assert sum(1 for o in convs if o.residual) == 20                                        # every other Add is an epilogue
    assert all(o.residual == o.y for o in convs if o.residual)                              # ... written in place
    aux = [o for o in convs if o.y_aux]
    assert [(o.conv.aux_c_total, o.conv.aux_c_offset) for o in aux] == [(384, 128), (768, 256)]   # pre-add routes -> concat slices
    ups = [o for o in convs if o.conv.upsample2x]
    assert [(o.conv.out_c_total, o.conv.out_c_offset) for o in ups] == [(768, 0), (384, 0)]
    spp_in = [o for o in convs if o.conv.out_c_total == 2048]
    assert len(spp_in) == 1 and spp_in[0].conv.out_c_offset == 1536                         # x lands in the last SPP slice
    # the three detection heads decode in their conv epilogue (yolo_head_decode_fwd): no fp32 head tensor, no decode launch
    heads = [o for o in ops if o.kind == OP_HEAD_DECODE]
    assert [(o.conv.cout, o.conv.h, o.io_row_offset, o.io_rows_total, o.head_na, o.head_nc) for o in heads] == \
        [(255, 20, 0, 25200, 3, 80), (255, 40, 1200, 25200, 3, 80), (255, 80, 6000, 25200, 3, 80)]
    assert [round(o.head_stride_px) for o in heads] == [32, 16, 8] and [h["op"] is not None for h in plan.heads] == [True] * 3
    assert list(heads[0].head_anchors_px[:6]) == [10.0, 13.0, 16.0, 30.0, 33.0, 23.0]      # anchors[i] -> head i, as the reference wires them
    assert plan.rows_total == 25200 and [h["row"] for h in plan.heads] == [0, 1200, 6000]
    assert [h["stride"] for h in plan.heads] == [32.0, 16.0, 8.0]


def test_planner_tiny_and_mobile():
    plan = _dry_plan(YOLOv3Tiny().eval(), 416)
    kinds = [o.kind for o in _ops(plan)]
    # the first ConvPoolBlock (3 -> 16, MaxPool2d(2, 2)) reads the NCHW f32 batch and writes the pooled map in ONE launch
    ops = _ops(plan)
    assert ops[0].kind == OP_CONV1_POOL and plan.fused_input and (ops[0].conv.cout, ops[0].conv.h, ops[0].conv.out_c_total) == (16, 416, 16)
    # the second and third ConvPoolBlock (16 -> 32 @208, 32 -> 64 @104) are one launch each too (yolo_conv3x3_pool_fwd)
    cp = [o for o in ops if o.kind == OP_CONV_POOL]
    assert [(o.conv.cin, o.conv.cout, o.conv.h, o.conv.out_c_total) for o in cp] == [(16, 32, 208, 32), (32, 64, 104, 64)]
    assert kinds.count(OP_CONV) == 8 and kinds.count(OP_HEAD_DECODE) == 2 and kinds.count(OP_MAXPOOL) == 3
    pools = [o.conv for o in _ops(plan) if o.kind == OP_MAXPOOL]
    assert (pools[-1].ksize, pools[-1].stride, pools[-1].pad, pools[-1].upsample2x) == (2, 1, 1, 2)   # dilated special
    # route1 is produced straight into the concat buffer [route1(256) | upsampled(128)]
    cat_writers = [(o.conv.out_c_offset, o.conv.upsample2x) for o in _ops(plan) if o.kind == OP_CONV and o.conv.out_c_total == 384]
    assert sorted(cat_writers) == [(0, 0), (256, 1)]
    assert plan.rows_total == 2535 and [h["stride"] for h in plan.heads] == [16.0, 32.0]
    plan = _dry_plan(YOLOv3TinyMobile().eval(), 416)
    kinds = [o.kind for o in _ops(plan)]
    # all seventeen inverted-residual blocks are one launch each (yolo_mbconv_fwd): the seven on the 208..52 maps (hidden <= 192)
    # with their hidden tile in LDS, the ten wide ones on the 26 / 13 maps with the hidden dimension streamed (round 4)
    mb = [o for o in _ops(plan) if o.kind == OP_MBCONV]
    assert [(o.conv.cin, o.kpad_pre, o.conv.cout, o.conv.stride, o.conv.h, bool(o.w_pre), o.conv.res_c_total) for o in mb] == [
        (32, 32, 16, 1, 208, False, 0), (16, 96, 24, 2, 208, True, 0), (24, 144, 24, 1, 104, True, 1), (24, 144, 32, 2, 104, True, 0),
        (32, 192, 32, 1, 52, True, 1), (32, 192, 32, 1, 52, True, 1), (32, 192, 64, 2, 52, True, 0),
        (64, 384, 64, 1, 26, True, 1), (64, 384, 64, 1, 26, True, 1), (64, 384, 64, 1, 26, True, 1), (64, 384, 96, 1, 26, True, 0),
        (96, 576, 96, 1, 26, True, 1), (96, 576, 96, 1, 26, True, 1), (96, 576, 160, 2, 26, True, 0),
        (160, 960, 160, 1, 13, True, 1), (160, 960, 160, 1, 13, True, 1), (160, 960, 320, 1, 13, True, 0)]
    assert [K.mbconv_form(o.conv.cin, o.kpad_pre, o.conv.cout, o.conv.stride) for o in mb] == [1] * 7 + [2] * 10
    # the wide form says "supported" only for shapes its launcher takes (stride 2, 96 -> 576 -> 320 needs 173 KB of LDS
    # for the 7x7 tile form - the planner must leave such a block as three launches instead of fusing it into a run-time error)
    assert K.mbconv_form(96, 576, 320, 2) == 0 and K.mbconv_form(96, 576, 256, 2) == 2 and K.mbconv_form(160, 960, 320, 1) == 2
    assert all(o.y != o.x and o.w_dw and o.bias_dw for o in mb)
    # the stride-2 first layer reads the NCHW f32 batch itself (yolo_conv1_nchw_f32_fwd, stride-2 form)
    first = _ops(plan)[0]
    assert plan.fused_input and first.kind == OP_CONV1_NCHW and (first.conv.stride, first.conv.cout, first.conv.ho) == (2, 32, 208)
    assert kinds.count(OP_DWCONV) == 0 and kinds.count(OP_CONV) == 4 and kinds.count(OP_HEAD_DECODE) == 2
    assert sum(1 for o in _ops(plan) if o.kind == OP_CONV and o.residual) == 0           # the identity shortcuts are inside the fused blocks


def test_depth_first_sub_batches_of_the_first_stages(monkeypatch):
    """YOLO_DEPTH_FIRST="a-b:S,...": launches [a, b) of the list become S passes over image sub-batches (engine.Plan._depth_first) -
    same ops, batch size n / S, every batch-major pointer moved by the sub-batch's first image; everything behind keeps its order,
    the head ops' indices follow, and the input pointer is patched into every sub-launch that reads the caller's batch."""
    model = YOLOv3SPP(anchors=C.SPP_ANCHORS).eval()
    plain = _dry_plan(model, 640, bs=8)
    monkeypatch.setenv("YOLO_DEPTH_FIRST", "0-2:4,2-5:2")
    plan = _dry_plan(model, 640, bs=8)
    assert plan.depth_first == [(0, 2, 4), (2, 5, 2)] and plain.depth_first == []
    po, do = _ops(plain), _ops(plan)
    assert len(do) == len(po) + 2 * 3 + 3 * 1
    assert [o.kind for o in do[:8]] == [OP_STEM, OP_RESUNIT] * 4 and [o.conv.n for o in do[:8]] == [2] * 8
    assert [o.kind for o in do[8:14]] == [OP_CONV, OP_RESUNIT, OP_RESUNIT] * 2 and [o.conv.n for o in do[8:14]] == [4] * 6
    assert [o.kind for o in do[14:]] == [o.kind for o in po[5:]] and all(o.conv.n == 8 for o in do[14:] if o.kind == OP_CONV)
    img = lambda o, hw, ct: o.conv.n * hw * hw * ct * 2                      # bytes of a sub-batch of an NHWC bf16 tensor
    for j in range(4):                                                        # stem -> unit, pass j on images [2 j, 2 j + 2)
        st, ru = do[2 * j], do[2 * j + 1]
        assert st.y == do[0].y + j * img(st, 320, 64) and ru.x == do[1].x + j * img(ru, 320, 64) and ru.y == do[1].y + j * img(ru, 320, 64)
        assert ru.x == st.y and st.x is None and plan._x_patch[j] == (2 * j, j * 2 * 3 * 640 * 640 * 4)
        assert (st.w, ru.w, ru.w_pre) == (do[0].w, do[1].w, do[1].w_pre)
    for j in range(2):
        cv, r1, r2 = do[8 + 3 * j: 11 + 3 * j]
        assert cv.x == do[1].y + j * img(cv, 320, 64) and cv.y == do[8].y + j * img(cv, 160, 128)
        assert r1.x == cv.y and r2.x == r1.y and r2.y == do[10].y + j * img(r2, 160, 128)
    assert do[14].x == do[10].y                                               # the first whole-batch launch reads all of the last unit's output
    assert [hd["op"] for hd in plan.heads] == [hd["op"] + 9 for hd in plain.heads]
    assert all(do[hd["op"]].kind == OP_HEAD_DECODE for hd in plan.heads)
    assert plan.conv_flops() == plain.conv_flops()
    plan.set_input_ptr(1 << 20)
    assert [do[i].x for i, _ in plan._x_patch] == [(1 << 20) + off for _, off in plan._x_patch]
    # a spec the list cannot take (a head op inside the range, a batch the sub-batch count does not divide) leaves the list alone
    monkeypatch.setenv("YOLO_DEPTH_FIRST", "0-5:3")
    assert _dry_plan(model, 640, bs=8).depth_first == []


@pytest.mark.parametrize("family,bs,hw,precision", [("tiny", 32, 416, "bf16"), ("tiny", 4, 416, "fp32"), ("mobile", 16, 416, "bf16"),
                                                     ("spp", 8, 640, "bf16"), ("spp", 1, 640, "fp32")])
def test_every_launch_stays_inside_the_plans_allocations(family, bs, hw, precision):
    """Static memory audit of the BASELINE launch lists.  For every launch of a plan built on the CPU, the byte
    range each pointer may be dereferenced over BY THE C ABI'S CONTRACT (include/yolo_hip.h: a view is [n, h, w, c_total] elements
    from its base, a packed weight matrix is cout_pad x kpad, a bias is cout_pad floats) must lie inside ONE allocation the plan
    owns (an activation buffer or a packed weight it keeps alive), reads and writes alike; buffers that share storage
    (engine.Plan._alloc) must have identical extents.  Catches planner errors: a view wider than its buffer, a shared buffer of
    another size, a weight packed for fewer input channels than the conv reads, a pooled / upsampled output sized for the wrong map.
    (The kernels' own indexing is audited dynamically on the GPU: test_launch_lists_stay_inside_their_buffers.)"""
    from pytorch_yolo_amd._lib import OP_CONV_F32, OP_MAXPOOL_F32
    model = {"tiny": YOLOv3Tiny, "mobile": YOLOv3TinyMobile, "spp": lambda: YOLOv3SPP(anchors=C.SPP_ANCHORS)}[family]().eval()
    model.precision = precision
    plan = _dry_plan(model, hw, bs=bs)
    allocs = {}
    for b in plan._bufs:
        t = b.tensor
        size = t.numel() * t.element_size()
        assert allocs.setdefault(t.data_ptr(), size) == size, "two buffers share storage but not their size"
    for t in plan._keep:
        allocs[t.data_ptr()] = t.numel() * t.element_size()
    starts = sorted(allocs)

    def inside(ptr, nbytes, what):
        import bisect
        assert nbytes > 0, what
        i = bisect.bisect_right(starts, ptr) - 1
        assert i >= 0 and ptr + nbytes <= starts[i] + allocs[starts[i]], \
            f"{what}: [{ptr:#x}, +{nbytes}) is not inside one allocation of the plan"

    f32 = precision == "fp32"
    es = 4 if f32 else 2
    checked = 0
    for i, op in enumerate(_ops(plan)):
        d, tag = op.conv, f"op {i} kind {op.kind}"
        m_in, m_out = d.n * d.h * d.w, d.n * d.ho * d.wo
        if op.kind in (OP_CONV, OP_CONV_F32, OP_CONV_POOL, OP_CONV1_NCHW, OP_CONV1_POOL, OP_HEAD_DECODE, OP_RESUNIT, OP_STEM):
            first = op.kind in (OP_CONV1_NCHW, OP_CONV1_POOL, OP_STEM)
            if first:
                assert not op.x                                      # the caller's NCHW batch: bound per call (Plan.feed checks its shape)
            else:
                inside(op.x, m_in * d.in_c_total * es, tag + " x")
                assert d.in_c_offset + d.cin <= d.in_c_total
            inside(op.w, d.cout_pad * d.kpad * es, tag + " w")
            inside(op.bias, d.cout_pad * 4, tag + " bias")
            # a conv's k index runs to k*k*cin of the view it reads (the stem's second conv: 9 * 32)
            assert d.kpad >= d.ksize * d.ksize * (32 if op.kind == OP_STEM else d.cin) and d.cout_pad >= d.cout, tag
            if op.kind == OP_HEAD_DECODE:
                assert not op.y and not op.y_aux                     # io / p of the call: bound per call (Plan._bind_outputs checks io)
            else:
                scale = 4 if d.upsample2x else 1
                if op.kind in (OP_CONV_POOL, OP_CONV1_POOL):
                    m_y = d.n * (d.ho // 2) * (d.wo // 2)
                else:
                    m_y = m_out * scale
                inside(op.y, m_y * d.out_c_total * (4 if (d.out_dtype or f32) else 2), tag + " y")
                assert d.out_c_offset + d.cout <= d.out_c_total + (7 if d.out_dtype else 0)
            if op.residual:
                inside(op.residual, m_out * d.res_c_total * es, tag + " residual")
                assert d.res_c_offset + d.cout <= d.res_c_total
            if op.y_aux and op.kind != OP_HEAD_DECODE:
                inside(op.y_aux, m_out * d.aux_c_total * es, tag + " pre-add copy")
                assert d.aux_c_offset + d.cout <= d.aux_c_total
            if op.kind == OP_RESUNIT:
                inside(op.w_pre, op.cout_pad_pre * op.kpad_pre * 2, tag + " w1")
                inside(op.bias_pre, op.cout_pad_pre * 4, tag + " b1")
                assert op.kpad_pre >= d.cout and op.cout_pad_pre >= d.cin
            if op.kind == OP_STEM:
                inside(op.w_pre, 128 * op.kpad_pre * 2, tag + " w1")
                inside(op.bias_pre, 128 * 4, tag + " b1")
        elif op.kind in (OP_MAXPOOL, OP_MAXPOOL_F32, OP_DWCONV):
            inside(op.x, m_in * d.in_c_total * es, tag + " x")
            inside(op.y, m_out * d.out_c_total * es, tag + " y")
            assert d.in_c_offset + d.cin <= d.in_c_total and d.out_c_offset + d.cin <= d.out_c_total
            if op.kind == OP_DWCONV:
                inside(op.w, 9 * d.cin * 4, tag + " w")
                inside(op.bias, d.cin * 4, tag + " bias")
        elif op.kind == OP_SPP:
            inside(op.y, m_in * 4 * d.cin * 2, tag + " concat buffer")
        elif op.kind == OP_MBCONV:
            inside(op.x, m_in * d.in_c_total * 2, tag + " x")
            inside(op.y, m_out * d.out_c_total * 2, tag + " y")
            ce = K.roundup(op.kpad_pre, 32)
            wide = K.mbconv_form(d.cin, op.kpad_pre, d.cout, d.stride) == 2      # plain matrices, read through buffer descriptors
            inside(op.w, K.roundup(d.cout, 16) * (ce if wide else _lib.load().yolo_mbconv_dstride(ce) // 2) * 2, tag + " w proj")
            inside(op.bias, K.roundup(d.cout, 16) * 4, tag + " b proj")
            inside(op.w_dw, 9 * ce * 4, tag + " w dw")
            inside(op.bias_dw, ce * 4, tag + " b dw")
            if op.w_pre:
                inside(op.w_pre, ce * (d.cin if wide else 48) * 2, tag + " w expand")
                inside(op.bias_pre, ce * 4, tag + " b expand")
            assert not wide or (op.w_pre and op.kpad_pre % 64 == 0)
        else:
            raise AssertionError(f"{tag}: not covered by the audit")
        checked += 1
    assert checked == plan.n_ops


def test_asm_mfma_kernels_keep_their_accumulator_distance(tmp_path):
    """the 20x20-tile kernels (conv3x3_t20v2 / conv3x3s2_t20 / resunit64_t20 / resunit_t20w - 60 % of the headline's
    launch time) issue their MFMAs as opaque ``asm volatile``, so LLVM's hazard recognizer does not pad around them and their
    correctness rests on conventions in the source (s_nop runs behind the loops, accumulators that start as MFMA results).  This
    test inspects the ISA hipcc actually emits (tools/isa_hazards.py): behind EVERY asm MFMA of every instantiation no non-MFMA
    instruction reads or writes its accumulator registers within 11 wait states (LLVM's own padding for an 8-pass MFMA; the need
    measured on MI355X is 7 for a read, 4 for a write: tools/micro/mfma_war.hip, profiles/r04_mfma_hazard_probe.txt), and the
    same accumulator is not reused by another MFMA within 4.  A compiler update or an edit that moves a VALU instruction next to
    the MFMAs fails here, on the CPU, instead of as 1 % wrong values under load.  The analyser is first shown to see a planted hazard."""
    import importlib.util
    from concurrent.futures import ThreadPoolExecutor
    spec = importlib.util.spec_from_file_location("isa_hazards", os.path.join(ROOT, "tools", "isa_hazards.py"))
    H = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(H)
    planted = tmp_path / "planted.s"
    planted.write_text("_Z4testv:\n\tv_mfma_f32_16x16x32_bf16 v[0:3], v[4:7], v[8:11], v[0:3]\n\ts_nop 2\n\tv_add_f32_e32 v20, v1, v21\n"
                       "\tv_mfma_f32_16x16x32_bf16 v[0:3], v[4:7], v[8:11], v[0:3]\n\ts_nop 15\n\ts_nop 15\n\tv_mov_b32_e32 v9, 0\n\ts_endpgm\n.Lfunc_end0:\n")
    n, best = H.analyse(H.parse(str(planted))["_Z4testv"])
    assert n == 2 and best["d_touch"][0] == 3 and best["d_reuse"][0] == 4 and best["war_valu"][0] == 32
    csrc = os.path.join(ROOT, "pytorch_yolo_amd", "csrc")
    files = ["conv3x3_t20.hip", "conv_resunit_t20.hip"]
    with ThreadPoolExecutor(2) as pool:
        outs = list(pool.map(lambda f: H.device_asm(os.path.join(csrc, f), str(tmp_path / (f + ".s"))), files))
    seen = 0
    for path in outs:
        for name, insts in H.parse(path).items():
            n, best = H.analyse(insts)
            if not n:
                continue
            seen += 1
            assert n >= 400, f"{name}: {n} MFMAs - not the fully unrolled tile loop?"
            assert best["d_touch"] is None or best["d_touch"][0] >= H.D_WINDOW, \
                f"{name}: a non-MFMA instruction touches an accumulator {best['d_touch'][0]} wait states behind its MFMA (asm line {best['d_touch'][1]} -> {best['d_touch'][2]}: {best['d_touch'][3]})"
            assert best["d_reuse"] is None or best["d_reuse"][0] >= 4, f"{name}: accumulator reused {best['d_reuse'][0]} wait states behind its MFMA"
    assert seen == 12            # t20v2 x 2, t20s2 x 4 (chunk-by-chunk / chunk-pair order), resunit_t20w x 4, resunit64_t20 x 2 (x LeakyReLU fast path / generic)


def test_planner_squeezenet_variant():
    """YOLOv3TinySqueeze (reference models/yolov3_tiny_squeeze.py): torchvision's SqueezeNet 1.1 key names, the
    unpadded first conv, ceil-mode pools and Fire modules that write their two expand convs into one concat buffer."""
    from pytorch_yolo_amd import YOLOv3TinySqueeze
    m = YOLOv3TinySqueeze().eval()
    keys = list(m.state_dict().keys())
    assert keys[:2] == ["features.sequence1.0.weight", "features.sequence1.0.bias"]
    assert "features.sequence1.7.expand3x3.weight" in keys and "features.sequence2.3.squeeze.bias" in keys
    assert m.state_dict()["features.sequence2.3.expand3x3.weight"].shape == (256, 64, 3, 3)
    assert m.state_dict()["sequence_branch1_2.branch1_conv2.sequence.conv.weight"].shape == (128, 256 + 128, 3, 3)
    plan = _dry_plan(m, 416)
    ops = _ops(plan)
    first = ops[0].conv
    assert (first.ksize, first.stride, first.pad, first.ho, first.act) == (3, 2, 0, 207, _lib.ACT_RELU)
    pools = [o.conv for o in ops if o.kind == OP_MAXPOOL]
    assert [(p.h, p.ho, p.ksize, p.stride, p.pad) for p in pools] == [(207, 103, 3, 2, 0), (103, 51, 3, 2, 0), (51, 25, 3, 2, 0)]
    convs = [o for o in ops if o.kind == OP_CONV]
    # 1 stem + 8 fires x 3 convs + 3 ConvBlocks of the head; the two heads decode in their conv epilogue
    assert len(convs) == 1 + 24 + 3 and sum(1 for o in ops if o.kind == OP_HEAD_DECODE) == 2
    fire_out = [(o.conv.out_c_total, o.conv.out_c_offset) for o in convs if o.conv.cin == 32 and o.conv.h == 103][:2]   # squeeze 16 -> 32 physical
    assert fire_out == [(128, 0), (128, 64)]                       # expand1x1 | expand3x3 share the concat buffer
    assert plan.rows_total == 2 * 3 * 25 * 25 and [h["stride"] for h in plan.heads] == [416 / 25, 416 / 25]


def test_planner_shufflenet_variant():
    """YOLOv3TinyShuffle (reference models/yolov3_tiny_shuffle.py): torchvision's ShuffleNetV2 x1.0 key names; on the
    device the halves of 58 / 116 / 232 channels sit in 64- / 128- / 256-channel slots, x.chunk(2) is two views and every unit ends
    in one channel-shuffle copy."""
    from pytorch_yolo_amd import YOLOv3TinyShuffle
    from pytorch_yolo_amd._lib import OP_SHUFFLE
    m = YOLOv3TinyShuffle().eval()
    sd = m.state_dict()
    assert sd["features.sequence1.0.0.weight"].shape == (24, 3, 3, 3) and sd["features.sequence1.2.0.branch1.2.weight"].shape == (58, 24, 1, 1)
    assert sd["features.sequence1.3.5.branch2.3.weight"].shape == (116, 1, 3, 3) and sd["features.sequence2.1.0.weight"].shape == (1024, 464, 1, 1)
    assert sd["sequence_branch1_2.branch1_conv2.sequence.conv.weight"].shape == (128, 232 + 128, 3, 3)
    plan = _dry_plan(m, 416)
    ops = _ops(plan)
    kinds = [o.kind for o in ops]
    assert kinds.count(OP_SHUFFLE) == 16 and kinds.count(OP_DWCONV) == 19 and kinds.count(OP_HEAD_DECODE) == 2
    sh = [o.conv for o in ops if o.kind == OP_SHUFFLE]
    assert [(c.cin, c.cout) for c in sh] == [(64, 58)] * 4 + [(128, 116)] * 8 + [(256, 232)] * 4        # (slot, logical half)
    assert (sh[1].in_c_total, sh[1].in_c_offset, sh[1].res_c_total) == (128, 0, 64)      # x1 = slot 0 of the previous unit, b = branch2
    assert sh[11].out_c_total == 256 + 128                           # route1 is shuffled straight into the head's concat buffer
    assert plan.rows_total == 3 * (26 * 26 + 13 * 13)


def test_unsupported_widths_fail_loudly():
    m = YOLOv3SPP(anchors=C.SPP_ANCHORS, kernels_divider=8).eval()
    with pytest.raises(RuntimeError, match="multiple of 8"):
        _dry_plan(m, 64)
    with pytest.raises(NotImplementedError):
        YOLOv3Tiny(onnx=True, in_shape=(1, 3, 64, 64))


def test_package_asks_for_enough_hardware_queues():
    """pytorch_yolo_amd/__init__.py: GPU_MAX_HW_QUEUES defaults to 8 (two pipelines + joined-call streams + all-gather side
    stream + RCCL's own do not fit HIP's default 4 queues: two sub-batch pipelines on one queue serialise)."""
    assert int(os.environ["GPU_MAX_HW_QUEUES"]) >= 8


def test_no_cpu_fallback():
    m = YOLOv3Tiny(kernels_divider=8, n_class=3).eval()
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        m(torch.rand(1, 3, 64, 64))
    from pytorch_yolo_amd import non_max_suppression
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        non_max_suppression(torch.rand(1, 10, 8))
    import pytorch_yolo_amd
    src = "".join(open(os.path.join(os.path.dirname(pytorch_yolo_amd.__file__), f)).read()
                  for f in ("engine.py", "kernels.py", "_lib.py", "distributed.py"))
    assert "oracle" not in src, "the product must never import the oracle"


def test_synthetic_generator_is_deterministic_and_order_free():
    from pytorch_yolo_amd.utils.synthetic import synth_images, synth_state_dict
    m = YOLOv3Tiny(kernels_divider=8, n_class=3)
    a = synth_state_dict(m.state_dict(), 11, n_class=3)
    b = synth_state_dict(dict(reversed(list(m.state_dict().items()))), 11, n_class=3)
    assert all(torch.equal(a[k], b[k]) for k in a)
    x = synth_images(2, 32, 48, 21)
    assert x.shape == (2, 3, 32, 48) and 0 <= float(x.min()) and float(x.max()) < 1
    assert torch.equal(x, synth_images(2, 32, 48, 21))


def test_darknet_weights_roundtrip_and_layout(tmp_path):
    """Darknet .weights IO: byte layout and round trip, incl. a backbone-only file."""
    from pytorch_yolo_amd.utils.darknet_io import darknet_layers
    from pytorch_yolo_amd.utils.synthetic import synth_state_dict
    m = YOLOv3Tiny(kernels_divider=8, n_class=3).eval()
    m.load_state_dict(synth_state_dict(m.state_dict(), 3, n_class=3))
    m.seen = 12345
    path = str(tmp_path / "tiny.weights")
    m.save_darknet_weights(path)
    raw = np.fromfile(path, dtype=np.int32, count=5)
    assert raw[3] == 12345
    data = np.fromfile(path, dtype="<f4")[5:]
    first = m.sequence_1.conv1.sequence
    n = first.batch_norm.bias.numel()
    # per block: bn.bias, bn.weight, running_mean, running_var, conv.weight
    assert np.array_equal(data[:n], first.batch_norm.bias.detach().numpy())
    assert np.array_equal(data[n:2 * n], first.batch_norm.weight.detach().numpy())
    assert np.array_equal(data[2 * n:3 * n], first.batch_norm.running_mean.numpy())
    assert np.array_equal(data[3 * n:4 * n], first.batch_norm.running_var.numpy())
    assert np.array_equal(data[4 * n:4 * n + first.conv.weight.numel()], first.conv.weight.detach().numpy().ravel())
    layers = darknet_layers(m)
    assert len(layers) == 13 and layers[8] is m.sequence_branch2.branch2_conv1        # /32 branch before /16 branch
    assert data.size == sum(p.numel() for p in m.parameters()) + sum(b.numel() for k, b in m.named_buffers() if "running" in k)
    m2 = YOLOv3Tiny(kernels_divider=8, n_class=3).eval()
    assert m2.load_darknet_weights(path) == 13 and int(m2.seen) == 12345
    for (k, a), (_, b) in zip(m.state_dict().items(), m2.state_dict().items()):
        if "num_batches_tracked" not in k:
            assert torch.equal(a, b), k
    # backbone-only checkpoint: first 7 blocks
    upto = sum(sum(t.numel() for t in ([l.sequence.batch_norm.bias] * 4 + [l.sequence.conv.weight])) for l in layers[:7])
    np.concatenate([raw.view(np.float32), data[:upto]]).astype("<f4").tofile(str(tmp_path / "backbone.weights"))
    m3 = YOLOv3Tiny(kernels_divider=8, n_class=3).eval()
    assert m3.load_darknet_weights(str(tmp_path / "backbone.weights")) == 7
    spp = YOLOv3SPP(kernels_divider=4, n_class=3, anchors=C.SPP_ANCHORS)
    assert len(darknet_layers(spp)) == 76
    spp.save_darknet_weights(str(tmp_path / "spp.weights"))
    spp2 = YOLOv3SPP(kernels_divider=4, n_class=3, anchors=C.SPP_ANCHORS)
    assert spp2.load_darknet_weights(str(tmp_path / "spp.weights")) == 76
    assert all(torch.equal(a, b) for (k, a), (_, b) in zip(spp.state_dict().items(), spp2.state_dict().items()) if "tracked" not in k)


def test_efficientnet_mirror_structure_and_padding():
    """YOLOv3TinyEfficient: state_dict keys follow efficientnet_pytorch 0.2.0's MBConvBlock names under the reference encoder's
    attribute names (models/yolov3_tiny_efficient.py:26-45), the route widths are the reference's (112, 320), and the
    TensorFlow "same" padding helper reproduces Conv2dSamePadding's arithmetic (hand-worked cases)."""
    import torch
    import torch.nn.functional as F
    from oracle.efficientnet import block_list, conv_same
    from pytorch_yolo_amd import YOLOv3TinyEfficient
    from pytorch_yolo_amd.engine import Recorder
    m = YOLOv3TinyEfficient(n_class=80)
    keys = list(m.state_dict())
    assert m.features.out_channels == (112, 320)
    assert len(m.features.sequence1) == 11 and len(m.features.sequence2) == 5 and len(block_list()) == 16
    assert "features.stem.0.weight" in keys and "features.stem.1.running_var" in keys
    assert "features.sequence1.0._expand_conv.weight" not in keys            # the first block has expand ratio 1
    for name in ("_expand_conv.weight", "_bn0.weight", "_depthwise_conv.weight", "_bn1.running_mean", "_se_reduce.weight", "_se_reduce.bias",
                 "_se_expand.weight", "_se_expand.bias", "_project_conv.weight", "_bn2.bias"):
        assert f"features.sequence1.1.{name}" in keys and f"features.sequence2.4.{name}" in keys
    sd = m.state_dict()
    assert sd["features.sequence1.3._depthwise_conv.weight"].shape == (144, 1, 5, 5)       # block 4: k5, 24 * 6
    assert sd["features.sequence1.3._se_reduce.weight"].shape == (6, 144, 1, 1)            # squeeze = int(24 * 0.25)
    assert sd["sequence_branch1_2.branch1_conv2.sequence.conv.weight"].shape == (128, 112 + 128, 3, 3)
    assert m.features.stem[1].eps == 1e-3
    # (size, k, stride) -> (out, leading pad): 416 / k3 / s2: total pad 1 -> 0 above; 13 / k3 / s2: out 7, total 2 -> 1;
    # 208 / k5 / s2: out 104, total 3 -> 1; 26 / k5 / s1: total 4 -> 2
    assert Recorder.tf_same(416, 3, 2) == (208, 0) and Recorder.tf_same(13, 3, 2) == (7, 1)
    assert Recorder.tf_same(208, 5, 2) == (104, 1) and Recorder.tf_same(26, 5, 1) == (26, 2)
    x, w = torch.randn(1, 4, 8, 8), torch.randn(4, 1, 5, 5)
    ref = F.conv2d(F.pad(x, [1, 2, 1, 2]), w, stride=2, groups=4)           # 8 / k5 / s2: out 4, total pad 3 = 1 + 2
    assert torch.equal(conv_same(x, w, stride=2, groups=4), ref)


def _pick(n, h, w, cin, cout, k, s, res=False):
    kpad = ((k * k * cin + 63) // 64) * 64
    cout_pad = ((cout + 127) // 128) * 128
    d = K.conv_desc(n=n, h=h, w=w, cin=cin, in_c_total=cin, in_c_offset=0, cout=cout, out_c_total=cout, out_c_offset=0,
                    ksize=k, stride=s, act=_lib.ACT_LEAKY01, kpad=kpad, cout_pad=cout_pad, res=(cout, 0) if res else (0, 0))
    return K.conv2d_pick(d, res)


def test_tile_rules_pick_the_intended_kernels():
    """Regression guard for the selection rules (conv_igemm.hip::conv2d_launch_ex): yolo_conv2d_pick runs the
    dispatcher without launching, so the kernel family each BASELINE SPP-640 layer (16-image sub-batch) lands on
    is pinned here without a GPU.  A rule edit that silently moves a headline layer shows up as a diff."""
    t20 = "t20v2<400px x 128 couts, 4 waves> grid %d"
    expect = {
        # 3x3 stride-1 residual convs of the four late Darknet stages: the 20x20-tile kernel, one block per (tile, 128 couts)
        (16, 160, 160, 64, 128, 3, 1, True): t20 % 1024,
        (16, 80, 80, 128, 256, 3, 1, True): t20 % 512,
        (16, 40, 40, 256, 512, 3, 1, True): t20 % 256,
        (16, 20, 20, 512, 1024, 3, 1, True): t20 % 128,
        # stride-2 downsamples: the parity-plane form of the 20x20-tile kernel where it gets two workgroups per CU, else implicit GEMM
        (16, 160, 160, 128, 256, 3, 2): "t20s2<400px x 128 couts, 4 waves, parity planes> grid 512",
        (16, 80, 80, 256, 512, 3, 2): "igemm<256x256,4x4 waves,BK64,2 stages,16x16x32> grid 200",
        (16, 40, 40, 512, 1024, 3, 2): "igemm<128x256,2x4 waves+4 loaders,BK64,3 stages,16x16x32> grid 200",
        # memory-bound 1x1 bottlenecks at the two big maps: weight-stationary streaming kernel
        (16, 80, 80, 256, 128, 1, 1): "stream1x1p<128 couts,K 256,80 px> grid 512",
        (16, 160, 160, 128, 64, 1, 1): "stream1x1<64 couts,K 128> grid 512",
        # 1x1 at 40^2 / 20^2: implicit GEMM
        (16, 40, 40, 512, 256, 1, 1): "igemm<128x256,2x8 waves,BK64,3 stages,16x16x32> grid 200",
        (16, 20, 20, 1024, 512, 1, 1): "igemm<128x128,2x4 waves,BK64,3 stages,16x16x32> grid 200",
        # too few tiles to fill the chip with 20x20 tiles: single-image 3x3 stays on the halo kernel
        (1, 80, 80, 128, 256, 3, 1, True): "halo<16x16,128 couts,CK64,1 halo buffers,16x16x32> grid 50",
    }
    got = {k: _pick(*k) for k in expect}
    assert got == expect
    # the whole-batch launch lists of bench.py (32 images per pipeline): what the headline number actually runs
    s2 = "t20s2<400px x 128 couts, 4 waves, parity planes> grid %d"
    expect32 = {
        (32, 160, 160, 64, 128, 3, 1, True): t20 % 2048,
        (32, 80, 80, 128, 256, 3, 1, True): t20 % 1024,
        (32, 80, 80, 256, 256, 3, 1): t20 % 1024,
        (32, 40, 40, 256, 512, 3, 1, True): t20 % 512,
        (32, 20, 20, 512, 1024, 3, 1, True): t20 % 256,
        (32, 320, 320, 64, 128, 3, 2): s2 % 2048,
        (32, 160, 160, 128, 256, 3, 2): s2 % 1024,
        (32, 80, 80, 256, 512, 3, 2): s2 % 512,
        (32, 40, 40, 512, 1024, 3, 2): "igemm<256x256,4x4 waves,BK64,2 stages,16x16x32> grid 200",     # 256 workgroups of 20x20: one per CU
        (32, 160, 160, 128, 64, 1, 1): "stream1x1<64 couts,K 128> grid 512",
        (32, 80, 80, 256, 128, 1, 1): "stream1x1p<128 couts,K 256,80 px> grid 512",
        (32, 80, 80, 384, 128, 1, 1): "stream1x1p<128 couts,K 384,48 px> grid 512",      # the /8 FPN's first conv (round 5: was 128x256 tiles)
        # (M = 51,200 >= 40,000 puts this layer under the "short-K 1x1 on a big map" rule at 32 images; measured there: 0.0256 ms,
        # 128x256 / 3 stages 0.0261, 256x256 0.0234 - profiles/r03_1x1_variants_n32.txt)
        (32, 40, 40, 512, 256, 1, 1): "igemm<256x128,4x2 waves,BK32,2 stages,16x16x32> grid 400",
        (32, 20, 20, 1024, 512, 1, 1): "igemm<128x256,2x8 waves,BK64,3 stages,16x16x32> grid 200",
    }
    got32 = {k: _pick(*k) for k in expect32}
    assert got32 == expect32, {k: v for k, v in got32.items() if expect32[k] != v}
    # the detection heads (head conv + decode + row filter in one launch): 64-pixel x 256-cout DECODE tiles
    def head(n, hw, cin, filt=True):
        d = K.conv_desc(n=n, h=hw, w=hw, cin=cin, in_c_total=cin, in_c_offset=0, cout=255, out_c_total=256, out_c_offset=0, ksize=1,
                        stride=1, act=_lib.ACT_LEAKY01, kpad=cin, cout_pad=256, out_dtype=_lib.DT_F32)
        return K.head_decode_pick(d, 3, 80, filt)
    assert head(32, 80, 256) == head(32, 80, 256, False) == "igemm<64x256,1x8 waves,BK64,2 stages,32x32x16,decode> grid 3200"
    assert head(32, 20, 1024).endswith(",decode> grid 200")
    # 13x13 maps (416 input) do not tile by 20: never the t20 kernel
    assert _pick(32, 13, 13, 512, 1024, 3, 1).startswith("igemm<")
    # a bad descriptor is still rejected in pick mode
    d = K.conv_desc(n=1, h=8, w=8, cin=32, in_c_total=32, in_c_offset=0, cout=32, out_c_total=32, out_c_offset=0,
                    ksize=3, stride=1, act=_lib.ACT_LEAKY01, kpad=320, cout_pad=128)
    d.cin = 0
    with pytest.raises(Exception):
        K.conv2d_pick(d)


def test_tile_rules_follow_the_cu_share_of_a_partitioned_stream():
    """yolo_set_launch_cus: a pipelined sub-batch stream owns 128 of the 256 CUs and the rules size grids for that - the long-K
    1x1 layers of the 20x20 maps leave the 3-stage 128x128 form for 128x256 tiles (100 workgroups = one round of 128 CUs), the
    40x40 1x1 and the last stride-2 layer take 256x256 tiles (100 tiles), the stride-2 128->256 layer takes 128x128 tiles
    (6.25 rounds at 0.89 instead of 3.1 at 0.78).  The 3x3 / stride-1 layers stay on the 20x20-tile kernel."""
    old = K.set_launch_cus(128)
    try:
        assert K.set_launch_cus(128) == 128
        assert _pick(16, 20, 20, 1024, 512, 1, 1) == "igemm<128x256,2x8 waves,BK64,3 stages,16x16x32> grid 100"
        assert _pick(16, 40, 40, 512, 256, 1, 1) == "igemm<256x256,4x4 waves,BK64,2 stages,16x16x32> grid 100"
        assert _pick(16, 40, 40, 512, 1024, 3, 2) == "igemm<256x256,4x4 waves,BK64,2 stages,16x16x32> grid 100"
        assert _pick(16, 160, 160, 128, 256, 3, 2) == "t20s2<400px x 128 couts, 4 waves, parity planes> grid 512"     # 4 per CU of the share
        assert _pick(16, 20, 20, 512, 1024, 3, 1, True) == "t20v2<400px x 128 couts, 4 waves> grid 128"
        assert _pick(16, 80, 80, 256, 128, 1, 1) == "stream1x1p<128 couts,K 256,80 px> grid 512"
    finally:
        K.set_launch_cus(old)
    assert _pick(16, 20, 20, 1024, 512, 1, 1) == "igemm<128x128,2x4 waves,BK64,3 stages,16x16x32> grid 200"
    with pytest.raises(RuntimeError):
        K.set_launch_cus(3)
