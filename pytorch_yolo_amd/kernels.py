"""Python-side wrappers of the C-ABI entry points (shape / dtype validation
happens here, before the FFI crossing — include/yolo_hip.h "Errors").

All tensors are torch CUDA (ROCm) tensors; launches go to the current stream.
"""
from __future__ import annotations

import ctypes as C

import torch

from . import _lib
from ._lib import (ACT_LEAKY01, ACT_NONE, ACT_RELU6, DT_BF16, DT_F32, YoloConvDesc, YoloMbconvDesc,
                   check, load)

__all__ = ["stream_ptr", "pack_input", "conv2d", "conv2d_pick", "head_decode_pick", "stem", "resunit", "resunit_supported", "resunit_form", "maxpool", "spp", "dwconv3x3", "dwconv", "se", "se_workspace_bytes", "mbconv", "mbconv_supported", "mbconv_form", "pack_mbconv", "conv3x3_pool", "conv3x3_pool_supported", "conv2d_splitk", "conv2d_splitk_plan", "decode", "head_decode", "head_decode_supported",
           "nms_merge", "pack_conv_weight", "roundup", "run_ops"]


def roundup(v: int, m: int) -> int:
    return (v + m - 1) // m * m


def stream_ptr() -> int:
    return torch.cuda.current_stream().cuda_stream


def _ptr(t):
    return None if t is None else C.c_void_p(t.data_ptr())


def _need_cuda(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise RuntimeError("pytorch_yolo_amd kernels need tensors on a ROCm device (no CPU fallback)")


def pack_conv_weight(w_oihw: torch.Tensor, bias: torch.Tensor | None, cin: int):
    """OIHW f32 weights (+bias) -> (bf16 [cout_pad, kpad], f32 [cout_pad]) on the host.
    k = (kh*ks + kw)*cin + c ; cin >= w.shape[1] pads extra input channels with zeros."""
    w = w_oihw.detach().float().cpu()
    cout, cin_w, k, _ = w.shape
    kpad, cout_pad = roundup(k * k * cin, 64), roundup(cout, 128)
    packed = torch.zeros((cout_pad, k * k, cin), dtype=torch.float32)
    packed[:cout, :, :cin_w] = w.permute(0, 2, 3, 1).reshape(cout, k * k, cin_w)
    packed = torch.nn.functional.pad(packed.reshape(cout_pad, k * k * cin), (0, kpad - k * k * cin))
    b = torch.zeros(cout_pad, dtype=torch.float32)
    if bias is not None:
        b[:cout] = bias.detach().float().cpu()
    return packed.to(torch.bfloat16).contiguous(), b, kpad, cout_pad


def pack_conv_weight_f32(w_oihw: torch.Tensor, bias: torch.Tensor | None, cin: int):
    """fp32 mode: OIHW f32 weights (+bias) -> (f32 [cout_pad, kpad], f32 [cout_pad]) on the host, same k order as the bf16 form
    (kpad = roundup(k*k*cin, 32), cout_pad = roundup(cout, 128))."""
    w = w_oihw.detach().float().cpu()
    cout, cin_w, k, _ = w.shape
    kpad, cout_pad = roundup(k * k * cin, 32), roundup(cout, 128)
    packed = torch.zeros((cout_pad, k * k, cin), dtype=torch.float32)
    packed[:cout, :, :cin_w] = w.permute(0, 2, 3, 1).reshape(cout, k * k, cin_w)
    packed = torch.nn.functional.pad(packed.reshape(cout_pad, k * k * cin), (0, kpad - k * k * cin))
    b = torch.zeros(cout_pad, dtype=torch.float32)
    if bias is not None:
        b[:cout] = bias.detach().float().cpu()
    return packed.contiguous(), b, kpad, cout_pad


def pack_input_f32(x: torch.Tensor, out: torch.Tensor) -> torch.Tensor:
    """fp32 mode: f32 NCHW -> f32 NHWC (channels zero-padded to out.shape[-1])."""
    _need_cuda(x, out)
    if x.dtype != torch.float32 or not x.is_contiguous():
        raise RuntimeError("pack_input_f32: x must be contiguous float32 NCHW")
    n, c, h, w = x.shape
    if out.dtype != torch.float32 or tuple(out.shape[:3]) != (n, h, w) or not out.is_contiguous():
        raise RuntimeError("pack_input_f32: out must be contiguous f32 [n,h,w,c_pad]")
    check(load().yolo_pack_input_nchw_f32_nhwc(_ptr(x), _ptr(out), n, c, h, w, out.shape[3], stream_ptr()), "pack_input_f32")
    return out


def conv2d_f32(x, w_packed, bias, y, desc: YoloConvDesc, residual=None, y_preadd=None):
    """fp32 mode conv (yolo_conv2d_f32_fwd): all tensors float32, ``desc`` as for conv2d with the fp32 packing sizes."""
    _need_cuda(x, w_packed, bias, y, residual, y_preadd)
    for t in (x, w_packed, bias, y, residual, y_preadd):
        if t is not None and t.dtype != torch.float32:
            raise RuntimeError("conv2d_f32: every tensor must be float32")
    check(load().yolo_conv2d_f32_fwd(_ptr(x), _ptr(w_packed), _ptr(bias), _ptr(residual), _ptr(y), _ptr(y_preadd),
                                     C.byref(desc), stream_ptr()), "conv2d_f32")
    return y


def maxpool_f32(x, y, *, n, h, w, c, in_view, out_view, ksize, stride, pad, dilation=1):
    _need_cuda(x, y)
    ho = (h + 2 * pad - dilation * (ksize - 1) - 1) // stride + 1
    wo = (w + 2 * pad - dilation * (ksize - 1) - 1) // stride + 1
    check(load().yolo_maxpool_f32_fwd(_ptr(x), _ptr(y), n, h, w, c, in_view[0], in_view[1], ho, wo, out_view[0],
                                      out_view[1], ksize, stride, pad, dilation, stream_ptr()), "maxpool_f32")
    return y


def pack_input(x: torch.Tensor, out: torch.Tensor) -> torch.Tensor:
    """f32 NCHW -> bf16 NHWC (channels zero-padded to out.shape[-1])."""
    _need_cuda(x, out)
    if x.dtype != torch.float32 or not x.is_contiguous():
        raise RuntimeError("pack_input: x must be contiguous float32 NCHW")
    n, c, h, w = x.shape
    if out.dtype != torch.bfloat16 or tuple(out.shape[:3]) != (n, h, w) or not out.is_contiguous():
        raise RuntimeError("pack_input: out must be contiguous bf16 [n,h,w,c_pad]")
    check(load().yolo_pack_input_nchw_f32(_ptr(x), _ptr(out), n, c, h, w, out.shape[3], stream_ptr()), "pack_input")
    return out


def conv_desc(*, n, h, w, cin, in_c_total, in_c_offset, cout, out_c_total, out_c_offset, ksize, stride,
              act, kpad, cout_pad, upsample2x=0, out_dtype=DT_BF16, res=(0, 0), aux=(0, 0), pad=None) -> YoloConvDesc:
    pad = (ksize - 1) // 2 if pad is None else pad
    d = YoloConvDesc()
    d.n, d.h, d.w, d.cin, d.in_c_total, d.in_c_offset = n, h, w, cin, in_c_total, in_c_offset
    d.ho, d.wo = (h + 2 * pad - ksize) // stride + 1, (w + 2 * pad - ksize) // stride + 1
    d.cout, d.out_c_total, d.out_c_offset = cout, out_c_total, out_c_offset
    d.ksize, d.stride, d.pad, d.act, d.upsample2x, d.out_dtype = ksize, stride, pad, act, upsample2x, out_dtype
    d.kpad, d.cout_pad = kpad, cout_pad
    d.res_c_total, d.res_c_offset = res
    d.aux_c_total, d.aux_c_offset = aux
    return d


def conv2d_pick(desc: YoloConvDesc, has_residual=False, has_preadd=False) -> str:
    """Name + grid of the kernel instance yolo_conv2d_fwd would launch for ``desc`` (no launch, works without a GPU)."""
    buf = C.create_string_buffer(256)
    check(load().yolo_conv2d_pick(C.byref(desc), int(has_residual), int(has_preadd), buf, 256), "conv2d_pick")
    return buf.value.decode()


def head_decode_pick(desc: YoloConvDesc, na: int, nc: int, filter: bool = False) -> str:
    """Name + grid of the kernel instance a head op would launch (yolo_head_decode_fwd, or with ``filter`` yolo_head_decode_filter_fwd)
    for ``desc`` (no launch, works without a GPU)."""
    buf = C.create_string_buffer(256)
    check(load().yolo_head_decode_pick(C.byref(desc), na, nc, int(filter), buf, 256), "head_decode_pick")
    return buf.value.decode()


def conv2d(x, w_packed, bias, y, desc: YoloConvDesc, residual=None, y_preadd=None):
    _need_cuda(x, w_packed, bias, y, residual, y_preadd)
    check(load().yolo_conv2d_fwd(_ptr(x), _ptr(w_packed), _ptr(bias), _ptr(residual), _ptr(y), _ptr(y_preadd),
                                 C.byref(desc), stream_ptr()), "conv2d")
    return y


def conv2d_splitk_plan(desc: YoloConvDesc, has_residual=False, has_preadd=False):
    """(splits, workspace bytes, int32 counters) a layer wants for yolo_conv2d_splitk_fwd; splits == 1: it does not take it."""
    sp, wb, nc = C.c_int(1), C.c_size_t(0), C.c_int(0)
    check(load().yolo_conv2d_splitk_plan(C.byref(desc), int(has_residual), int(has_preadd), C.byref(sp), C.byref(wb), C.byref(nc)),
          "conv2d_splitk_plan")
    return sp.value, wb.value, nc.value


def conv2d_splitk(x, w_packed, bias, y, desc: YoloConvDesc, splits, workspace, counters, residual=None, y_preadd=None):
    """yolo_conv2d_splitk_fwd: ``workspace`` a byte / float tensor of at least the planned size, ``counters`` int32 zeros."""
    _need_cuda(x, w_packed, bias, y, residual, y_preadd, workspace, counters)
    check(load().yolo_conv2d_splitk_fwd(_ptr(x), _ptr(w_packed), _ptr(bias), _ptr(residual), _ptr(y), _ptr(y_preadd), C.byref(desc),
                                        splits, _ptr(workspace), workspace.numel() * workspace.element_size(), _ptr(counters),
                                        stream_ptr()), "conv2d_splitk")
    return y


def stem(x_nchw, cin_real, w1_packed, b1, kpad1, w2_packed, b2, y, desc: YoloConvDesc):
    """conv3x3/s1 (cin_real -> 32) + conv3x3/s2 (32 -> 64) straight from the float32 NCHW batch (yolo_stem_fwd)."""
    _need_cuda(x_nchw, w1_packed, b1, w2_packed, b2, y)
    check(load().yolo_stem_fwd(_ptr(x_nchw), cin_real, _ptr(w1_packed), _ptr(b1), kpad1, _ptr(w2_packed), _ptr(b2), _ptr(y),
                               C.byref(desc), stream_ptr()), "stem")
    return y


def resunit_supported(c: int, h: int, w: int) -> bool:
    return bool(load().yolo_resunit_supported(c, h, w))


def resunit_form(c: int, n: int, h: int, w: int) -> int:
    """0 not supported, 1 generic 16x16-tile kernel, 2 persistent 64-channel kernel, 3 the 20-pixel-wide tile kernels (yolo_resunit_form)."""
    return int(load().yolo_resunit_form(c, n, h, w))


def resunit(x, w1_packed, b1, w2_packed, b2, y, desc: YoloConvDesc, kpad1: int, cout_pad1: int, y_preadd=None):
    """One Darknet residual unit: y = x + act(conv3x3(act(conv1x1(x)))) (yolo_resunit_fwd); ``desc`` is the 3x3's."""
    _need_cuda(x, w1_packed, b1, w2_packed, b2, y, y_preadd)
    check(load().yolo_resunit_fwd(_ptr(x), _ptr(w1_packed), _ptr(b1), _ptr(w2_packed), _ptr(b2), _ptr(y), _ptr(y_preadd),
                                  C.byref(desc), kpad1, cout_pad1, stream_ptr()), "resunit")
    return y


def maxpool(x, y, *, n, h, w, c, in_view, out_view, ksize, stride, pad, dilation=1):
    _need_cuda(x, y)
    ho = (h + 2 * pad - dilation * (ksize - 1) - 1) // stride + 1
    wo = (w + 2 * pad - dilation * (ksize - 1) - 1) // stride + 1
    check(load().yolo_maxpool_fwd(_ptr(x), _ptr(y), n, h, w, c, in_view[0], in_view[1], ho, wo, out_view[0],
                                  out_view[1], ksize, stride, pad, dilation, stream_ptr()), "maxpool")
    return y


def spp(buf, *, n, h, w, c):
    _need_cuda(buf)
    check(load().yolo_spp_fwd(_ptr(buf), n, h, w, c, stream_ptr()), "spp")
    return buf


def se_workspace_bytes(n, c) -> int:
    return int(load().yolo_se_workspace_bytes(n, c))


def dwconv(x, wkc, bias, y, *, n, h, w, c, in_view, out_view, ho, wo, ksize, stride, pad, act):
    """Depthwise k x k (3 or 5) conv with an explicit leading pad (yolo_dwconv_fwd); wkc: f32 [k*k][c]."""
    _need_cuda(x, wkc, bias, y)
    check(load().yolo_dwconv_fwd(_ptr(x), _ptr(wkc), _ptr(bias), _ptr(y), n, h, w, c, in_view[0], in_view[1], ho, wo, out_view[0],
                                 out_view[1], ksize, stride, pad, act, stream_ptr()), "dwconv")
    return y


def se(x, y, w1, b1, w2, b2, workspace, *, n, h, w, c, in_view, out_view):
    """Squeeze-and-excitation (yolo_se_fwd): w1 f32 [sq][c], w2 f32 [sq][c] (the expand weight transposed)."""
    _need_cuda(x, y, w1, b1, w2, b2, workspace)
    check(load().yolo_se_fwd(_ptr(x), _ptr(y), n, h, w, c, in_view[0], in_view[1], out_view[0], out_view[1], _ptr(w1), _ptr(b1),
                             _ptr(w2), _ptr(b2), w1.shape[0], _ptr(workspace), workspace.numel() * workspace.element_size(),
                             stream_ptr()), "se")
    return y


def dwconv3x3(x, w9c, bias, y, *, n, h, w, c, in_view, out_view, stride, act):
    _need_cuda(x, w9c, bias, y)
    ho, wo = (h + 2 - 3) // stride + 1, (w + 2 - 3) // stride + 1
    check(load().yolo_dwconv3x3_fwd(_ptr(x), _ptr(w9c), _ptr(bias), _ptr(y), n, h, w, c, in_view[0], in_view[1],
                                    ho, wo, out_view[0], out_view[1], stride, act, stream_ptr()), "dwconv3x3")
    return y


def conv3x3_pool_supported(cin: int, cout: int) -> bool:
    return bool(load().yolo_conv3x3_pool_supported(cin, cout))


def conv3x3_pool(x, w_packed, bias, y, desc: YoloConvDesc, pool: bool = True):
    """3x3 / s1 ConvBlock with 16 | 32 input and 32 | 64 output channels (+ MaxPool2d(2, 2) when ``pool``): y is the
    pooled map then (yolo_conv3x3_pool_fwd)."""
    _need_cuda(x, w_packed, bias, y)
    check(load().yolo_conv3x3_pool_fwd(_ptr(x), _ptr(w_packed), _ptr(bias), _ptr(y), C.byref(desc), 1 if pool else 0,
                                       stream_ptr()), "conv3x3_pool")
    return y


def mbconv_supported(cin: int, hidden: int, cout: int, stride: int) -> bool:
    return bool(load().yolo_mbconv_supported(cin, hidden, cout, stride))


def mbconv_form(cin: int, hidden: int, cout: int, stride: int) -> int:
    """0: not covered; 1: whole hidden tile in LDS (csrc/conv_mbconv.hip); 2: hidden dimension streamed in chunks of 64
    (csrc/conv_mbwide.hip, the blocks with 64..160 input channels).  The two forms read different weight images."""
    return int(load().yolo_mbconv_supported(cin, hidden, cout, stride))


def pack_mbconv(w_exp, b_exp, w_dw, b_dw, w_proj, b_proj, stride: int = 1):
    """Folded weights of one inverted-residual block -> the images yolo_mbconv_fwd reads (host tensors).
    w_exp [hidden,cin,1,1] or None, w_dw [hidden,1,3,3], w_proj [cout,hidden,1,1]; biases f32.
    Returns (we bf16 [ce,48] | None, be f32 [ce] | None, wd f32 [9,ce], bd f32 [ce], wp bf16 [cout_pad,dstride/2],
    bp f32 [cout_pad]); for the wide form (mbconv_form() == 2, hidden a multiple of 64) the plain matrices:
    (we bf16 [hidden,cin], be f32 [hidden], wd f32 [9,hidden], bd f32 [hidden], wp bf16 [cout_pad,hidden], bp f32 [cout_pad])."""
    hidden, cout = w_dw.shape[0], w_proj.shape[0]
    if w_exp is not None and mbconv_form(w_exp.shape[1], hidden, cout, stride) == 2:
        cop = roundup(cout, 16)
        wp = torch.zeros((cop, hidden), dtype=torch.float32)
        wp[:cout] = w_proj.detach().float().cpu().reshape(cout, hidden)
        bp = torch.zeros(cop, dtype=torch.float32)
        bp[:cout] = b_proj.detach().float().cpu()
        return (w_exp.detach().float().cpu().reshape(hidden, -1).to(torch.bfloat16).contiguous(), b_exp.detach().float().cpu().contiguous(),
                w_dw.detach().float().cpu().reshape(hidden, 9).t().contiguous(), b_dw.detach().float().cpu().contiguous(),
                wp.to(torch.bfloat16).contiguous(), bp)
    ce, cop = roundup(hidden, 32), roundup(cout, 16)
    dstride = load().yolo_mbconv_dstride(ce)
    we = be = None
    if w_exp is not None:
        cin = w_exp.shape[1]
        we = torch.zeros((ce, 48), dtype=torch.float32)
        we[:hidden, :cin] = w_exp.detach().float().cpu().reshape(hidden, cin)
        we = we.to(torch.bfloat16).contiguous()
        be = torch.zeros(ce, dtype=torch.float32)
        be[:hidden] = b_exp.detach().float().cpu()
    wd = torch.zeros((9, ce), dtype=torch.float32)
    wd[:, :hidden] = w_dw.detach().float().cpu().reshape(hidden, 9).t()
    bd = torch.zeros(ce, dtype=torch.float32)
    bd[:hidden] = b_dw.detach().float().cpu()
    wp = torch.zeros((cop, dstride // 2), dtype=torch.float32)
    wp[:cout, :hidden] = w_proj.detach().float().cpu().reshape(cout, hidden)
    bp = torch.zeros(cop, dtype=torch.float32)
    bp[:cout] = b_proj.detach().float().cpu()
    return we, be, wd.contiguous(), bd, wp.to(torch.bfloat16).contiguous(), bp


def mbconv(x, packed, y, *, n, h, w, cin, hidden, cout, in_view, out_view, stride, has_res):
    """One MobileNetV2 inverted-residual block (yolo_mbconv_fwd); ``packed`` = pack_mbconv(...) moved to the device."""
    we, be, wd, bd, wp, bp = packed
    _need_cuda(x, we, be, wd, bd, wp, bp, y)
    d = YoloMbconvDesc(n=n, h=h, w=w, cin=cin, in_c_total=in_view[0], in_c_offset=in_view[1], hidden=hidden, cout=cout,
                       out_c_total=out_view[0], out_c_offset=out_view[1], stride=stride,
                       has_expand=0 if we is None else 1, has_res=1 if has_res else 0)
    check(load().yolo_mbconv_fwd(_ptr(x), _ptr(we), _ptr(be), _ptr(wd), _ptr(bd), _ptr(wp), _ptr(bp), _ptr(y),
                                 C.byref(d), stream_ptr()), "mbconv")
    return y


def decode(head, anchors_px, nc, stride_px, io, io_row_offset, p=None):
    """head f32 [bs,ny,nx,ct] -> rows of io f32 [bs,rows_total,5+nc] (+ p f32 [bs,na,ny,nx,5+nc])."""
    _need_cuda(head, io, p)
    bs, ny, nx, ct = head.shape
    na = len(anchors_px)
    if head.dtype != torch.float32 or io.dtype != torch.float32 or not head.is_contiguous() or not io.is_contiguous():
        raise RuntimeError("decode: head and io must be contiguous float32")
    if io.shape[0] != bs or io.shape[2] != nc + 5:
        raise RuntimeError("decode: io shape mismatch")
    if p is not None and (tuple(p.shape) != (bs, na, ny, nx, nc + 5) or p.dtype != torch.float32 or not p.is_contiguous()):
        raise RuntimeError("decode: p shape mismatch")
    flat = (C.c_float * (2 * na))(*[float(v) for a in anchors_px for v in a])
    check(load().yolo_decode_fwd(_ptr(head), ct, flat, na, nc, bs, ny, nx, float(stride_px), _ptr(io), io.shape[1],
                                 io_row_offset, _ptr(p), stream_ptr()), "decode")
    return io


def head_decode_supported(cout: int, na: int, nc: int) -> bool:
    return bool(load().yolo_head_decode_supported(cout, na, nc))


def head_decode(x, w_packed, bias, desc: YoloConvDesc, anchors_px, nc, stride_px, io, io_row_offset, p=None):
    """Head conv + YOLOLayer decode in one launch (yolo_head_decode_fwd): x bf16 NHWC view described by ``desc``."""
    _need_cuda(x, w_packed, bias, io, p)
    na = len(anchors_px)
    bs, ny, nx = desc.n, desc.ho, desc.wo
    if io.dtype != torch.float32 or not io.is_contiguous() or io.shape[0] != bs or io.shape[2] != nc + 5:
        raise RuntimeError("head_decode: io must be contiguous float32 [bs, rows, 5+nc]")
    if p is not None and (tuple(p.shape) != (bs, na, ny, nx, nc + 5) or p.dtype != torch.float32 or not p.is_contiguous()):
        raise RuntimeError("head_decode: p shape mismatch")
    flat = (C.c_float * (2 * na))(*[float(v) for a in anchors_px for v in a])
    check(load().yolo_head_decode_fwd(_ptr(x), _ptr(w_packed), _ptr(bias), C.byref(desc), flat, na, nc, float(stride_px),
                                      _ptr(io), io.shape[1], io_row_offset, _ptr(p), stream_ptr()), "head_decode")
    return io


def nms_workspace_bytes(bs, rows, nc) -> int:
    return int(load().yolo_nms_workspace_bytes(bs, rows, nc))


def nms_merge(pred, conf_thres, nms_thres, out_dets, out_idx, out_count, workspace, *, min_wh=2.0,
              max_per_class=100, mutate_conf=False):
    _need_cuda(pred, out_dets, out_idx, out_count, workspace)
    if pred.dtype != torch.float32 or pred.dim() != 3 or not pred.is_contiguous():
        raise RuntimeError("nms: prediction must be contiguous float32 [bs, rows, 5+nc]")
    bs, rows, no = pred.shape
    cap = out_dets.shape[1]
    if tuple(out_dets.shape) != (bs, cap, 7) or tuple(out_idx.shape) != (bs, cap) or out_count.numel() != bs:
        raise RuntimeError("nms: output shape mismatch")
    if out_dets.dtype != torch.float32 or out_idx.dtype != torch.int32 or out_count.dtype != torch.int32:
        raise RuntimeError("nms: output dtype mismatch")
    check(load().yolo_nms_merge(_ptr(pred), bs, rows, no - 5, float(conf_thres), float(nms_thres), float(min_wh),
                                int(max_per_class), int(bool(mutate_conf)), _ptr(out_dets), _ptr(out_idx),
                                _ptr(out_count), cap, _ptr(workspace), workspace.numel() * workspace.element_size(),
                                stream_ptr()), "nms_merge")


def nms_compact_workspace_bytes(bs, rows, nc) -> int:
    return int(load().yolo_nms_compact_workspace_bytes(bs, rows, nc))


def head_decode_filter(x, w_packed, bias, desc: YoloConvDesc, anchors_px, nc, stride_px, rows_total, io_row_offset, conf_thres,
                       workspace, *, min_wh=2.0, p=None):
    """Head conv + decode + NMS row filter in one launch (yolo_head_decode_filter_fwd): survivors go to ``workspace``."""
    _need_cuda(x, w_packed, bias, workspace, p)
    na = len(anchors_px)
    flat = (C.c_float * (2 * na))(*[float(v) for a in anchors_px for v in a])
    check(load().yolo_head_decode_filter_fwd(_ptr(x), _ptr(w_packed), _ptr(bias), C.byref(desc), flat, na, nc, float(stride_px),
                                             rows_total, io_row_offset, float(conf_thres), float(min_wh), _ptr(workspace),
                                             workspace.numel() * workspace.element_size(), _ptr(p), stream_ptr()), "head_decode_filter")


def nms_merge_compact(workspace, bs, rows, nc, nms_thres, out_dets, out_idx, out_count, *, max_per_class=100):
    """Sort / MERGE / final order over the survivors the head epilogues left in ``workspace`` (yolo_nms_merge_compact)."""
    _need_cuda(workspace, out_dets, out_idx, out_count)
    cap = out_dets.shape[1]
    if tuple(out_dets.shape) != (bs, cap, 7) or tuple(out_idx.shape) != (bs, cap) or out_count.numel() != bs:
        raise RuntimeError("nms: output shape mismatch")
    if out_dets.dtype != torch.float32 or out_idx.dtype != torch.int32 or out_count.dtype != torch.int32:
        raise RuntimeError("nms: output dtype mismatch")
    check(load().yolo_nms_merge_compact(_ptr(workspace), workspace.numel() * workspace.element_size(), bs, rows, nc, float(nms_thres),
                                        int(max_per_class), _ptr(out_dets), _ptr(out_idx), _ptr(out_count), cap, stream_ptr()),
          "nms_merge_compact")


def cu_masked_stream(cu_bits, device):
    """A torch stream (ExternalStream over hipExtStreamCreateWithCUMask) whose kernels run only on the listed CU
    indices (CU i sits on XCD i % 8).  The HIP stream lives as long as the process: it is never destroyed behind torch."""
    cu_bits = list(cu_bits)
    if not cu_bits or min(cu_bits) < 0:
        raise RuntimeError("cu_masked_stream: empty or negative CU list")
    n_words = (max(cu_bits) + 32) // 32
    words = (C.c_uint32 * n_words)()
    for b in cu_bits:
        words[b // 32] |= 1 << (b % 32)
    out = C.c_void_p()
    with torch.cuda.device(device):
        check(load().yolo_stream_create_cu_mask(words, n_words, C.byref(out)), "stream_create_cu_mask")
    return torch.cuda.ExternalStream(out.value, device=device)


def set_launch_cus(n_cu: int) -> int:
    """CUs the coming launches of this thread may use (tile rules size grids against it); returns the previous value."""
    old = load().yolo_set_launch_cus(int(n_cu))
    if old < 0:
        check(old, "set_launch_cus")
    return old


class Event:
    """A HIP event owned by this library's C side (yolo_event_*): no timing, recorded / waited for by yolo_pipeline_step."""

    def __init__(self):
        h = C.c_void_p()
        check(load().yolo_event_create(C.byref(h)), "event_create")
        self.handle = h.value

    def record(self, stream: int = None):
        check(load().yolo_event_record(self.handle, stream_ptr() if stream is None else stream), "event_record")

    def synchronize(self):
        check(load().yolo_event_synchronize(self.handle), "event_synchronize")      # (ctypes releases the GIL while it waits)

    def __del__(self):
        try:
            load().yolo_event_destroy(self.handle)
        except Exception:                                   # noqa: BLE001 (interpreter teardown)
            pass


def pipeline_step(step):
    check(load().yolo_pipeline_step(C.byref(step)), "pipeline_step")


def pack_detections(dets, idx, count, packed, packed_idx=None):
    """Kept rows of all images back to back (yolo_pack_detections): dets [bs, cap, 7], count int32 [bs] on the device -> packed
    [total, 7] (+ packed_idx int64 [total] from idx int32 [bs, cap])."""
    _need_cuda(dets, count, packed, idx, packed_idx)
    bs, cap = dets.shape[0], dets.shape[1]
    check(load().yolo_pack_detections(_ptr(dets), _ptr(idx), _ptr(count), bs, cap, _ptr(packed), _ptr(packed_idx), stream_ptr()),
          "pack_detections")
    return packed


def run_ops(op_array, n_ops: int):
    check(load().yolo_run_ops(op_array, n_ops, stream_ptr()), "run_ops")
