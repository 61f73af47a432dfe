"""GPU parity tests: every call goes through the C-ABI of libyolo_hip.so and is compared with the
CPU oracle (oracle/) on the same seeded inputs, and with the reference's golden vectors.

Tolerances (stated once, used below):
  * pools / SPP / pack: exact (max and rounding are order-free).
  * conv (bf16 in, fp32 accumulate): against an fp32 conv of the SAME bf16-rounded operands,
    |err| <= 2e-3 * sqrt(K)-scaled bound -> we use rtol 1e-2 on the bf16-rounded output (1 bf16 ulp = 2^-8).
  * decode: fp32 vs torch.  Standalone kernel (yolo_decode_fwd: expf + IEEE division, the decode of precision = "fp32"):
    rtol 2e-6 / atol 1e-5 for |logit| up to 30, <= 2 ulp on values straddling the NMS thresholds.  Head-conv epilogue (the bf16
    path: v_exp_f32 on r * log2 e + v_rcp_f32): w / h relative error <= (3 + |r|) * 2^-23, sigmoid absolute error <= 2^-22
    (test_fused_head_decode_wide_logits).
  * NMS: kept-index sets, conf, class_conf, class BIT-EXACT vs the oracle; merged boxes bit-exact vs the
    oracle (same sequential fp32 order) and within 2e-4 px of the reference golden.
  * whole model in bf16 vs the fp32 oracle: boxes within max(1.5 px, 2 %), scores within 2e-2 (4e-2 for YOLOv3, see there)
    (bf16 activations through up to 75 conv layers); measured values are printed.
"""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

import _cases as C
from helpers import build_case, build_rule_case, build_separable_case, load_golden, oracle_forward, strict_share

pytestmark = pytest.mark.gpu

DEV = "cuda:0"


def _bf16r(t):
    return t.to(torch.bfloat16).float()


def _nhwc(t):      # NCHW f32 -> NHWC bf16 on device
    return t.permute(0, 2, 3, 1).contiguous().to(torch.bfloat16).to(DEV)


def _nchw(t):      # NHWC (any dtype) on device -> NCHW f32 on host
    return t.float().permute(0, 3, 1, 2).contiguous().cpu()


# ------------------------------------------------------------------------------------------------
def test_library_loaded():
    from pytorch_yolo_amd import _lib
    assert _lib.load().yolo_abi_version() == _lib.ABI_VERSION == 2


def test_pack_input_exact():
    from pytorch_yolo_amd import kernels as K
    x = torch.rand(3, 3, 17, 23)
    out = torch.full((3, 17, 23, 8), 7.0, dtype=torch.bfloat16, device=DEV)
    K.pack_input(x.to(DEV), out)
    got = _nchw(out)
    assert torch.equal(got[:, :3], _bf16r(x)) and torch.count_nonzero(got[:, 3:]) == 0


CONV_CASES = [
    # n, h, w, cin, cout, k, stride, act, residual, aux, upsample, f32
    (2, 16, 16, 8, 32, 3, 1, "leaky", False, False, False, False),     # first-layer shape: K=72 -> taps straddle K steps
    (2, 20, 20, 64, 128, 3, 1, "leaky", True, True, False, False),     # residual + pre-add copy
    (1, 33, 29, 32, 64, 3, 2, "leaky", False, False, False, False),    # odd size, stride 2
    (3, 13, 13, 256, 255, 1, 1, "none", False, False, False, True),    # detection head: 255 couts, fp32 store
    (2, 10, 10, 128, 64, 1, 1, "leaky", False, False, True, False),    # 2x nearest upsample on store
    (1, 40, 40, 16, 16, 3, 1, "relu6", False, False, False, False),    # cin not a multiple of 32
    (2, 7, 9, 96, 24, 1, 1, "none", True, False, False, False),        # linear bottleneck + residual
    (1, 26, 26, 384, 256, 3, 1, "leaky", False, False, False, False),  # tiny-yolo concat conv
    (1, 9, 9, 1024, 512, 1, 1, "leaky", False, False, False, False),   # long K
    # large 3x3/s1 maps take the halo-staged kernel (csrc/conv3x3_halo.hip)
    (1, 80, 96, 64, 128, 3, 1, "leaky", True, True, False, False),     # one cin chunk (CK 64), BN 128, residual + pre-add
    (1, 96, 80, 128, 256, 3, 1, "leaky", True, False, False, False),   # two cin chunks, double-buffered halo, BN 256
    (1, 112, 80, 32, 64, 3, 1, "leaky", False, False, False, False),   # CK 32, BN 64
    (2, 94, 100, 96, 64, 3, 1, "none", True, False, False, False),     # partial edge tiles, 3 chunks of 32, batch 2
    (1, 80, 80, 192, 128, 3, 1, "relu6", False, False, False, False),  # 3 chunks of 64
]


@pytest.mark.parametrize("case", CONV_CASES, ids=lambda c: "n%d_%dx%d_c%d-%d_k%d_s%d_%s_r%d_a%d_u%d_f%d" % tuple(int(v) if not isinstance(v, str) else v for v in c))
def test_conv_kernel(case):
    from pytorch_yolo_amd import kernels as K
    n, h, w, cin, cout, k, stride, act, use_res, use_aux, up, f32 = case
    g = torch.Generator().manual_seed(hash(case) & 0xffff)
    x = torch.randn(n, cin, h, w, generator=g)
    wt = torch.randn(cout, cin, k, k, generator=g) * (2.0 / (cin * k * k)) ** 0.5
    bias = torch.randn(cout, generator=g) * 0.1
    pad = (k - 1) // 2
    ho, wo = (h + 2 * pad - k) // stride + 1, (w + 2 * pad - k) // stride + 1
    res = torch.randn(n, cout, ho, wo, generator=g) if use_res else None
    # views with channel offsets on every tensor to exercise the concat plumbing
    in_ct, in_co = cin + 16, 8
    xin = torch.zeros(n, h, w, in_ct, dtype=torch.bfloat16, device=DEV)
    xin[..., in_co:in_co + cin] = _nhwc(x)
    oh, ow = (2 * ho, 2 * wo) if up else (ho, wo)
    out_ct = (K.roundup(cout, 8) + 8)
    out_co = 8
    y = torch.full((n, oh, ow, out_ct), -77.0, dtype=torch.float32 if f32 else torch.bfloat16, device=DEV)
    aux = torch.full((n, ho, wo, cout + 8), -77.0, dtype=torch.bfloat16, device=DEV) if use_aux else None
    rin = _nhwc(res) if use_res else None
    wp, bp, kpad, cout_pad = K.pack_conv_weight(wt, bias, cin)
    from pytorch_yolo_amd._lib import ACT_LEAKY01, ACT_NONE, ACT_RELU6, DT_BF16, DT_F32
    d = K.conv_desc(n=n, h=h, w=w, cin=cin, in_c_total=in_ct, in_c_offset=in_co, cout=cout, out_c_total=out_ct,
                    out_c_offset=out_co, ksize=k, stride=stride,
                    act={"leaky": ACT_LEAKY01, "none": ACT_NONE, "relu6": ACT_RELU6}[act], kpad=kpad, cout_pad=cout_pad,
                    upsample2x=int(up), out_dtype=DT_F32 if f32 else DT_BF16,
                    res=(cout, 0) if use_res else (0, 0), aux=(cout + 8, 8) if use_aux else (0, 0))
    K.conv2d(xin, wp.to(DEV), bp.to(DEV), y, d, residual=rin, y_preadd=aux)
    torch.cuda.synchronize()
    # fp32 reference on the same bf16-rounded operands
    ref = F.conv2d(_bf16r(x), _bf16r(wt), bias, stride=stride, padding=pad)
    ref = {"leaky": lambda t: F.leaky_relu(t, 0.1), "none": lambda t: t, "relu6": F.relu6}[act](ref)
    pre = ref
    if use_res:
        ref = ref + _bf16r(res)
    if up:
        ref = F.interpolate(ref, scale_factor=2, mode="nearest")
    got = _nchw(y[..., out_co:out_co + cout])
    tol = dict(rtol=1e-5, atol=2e-4) if f32 else dict(rtol=1e-2, atol=1e-2)
    torch.testing.assert_close(got, ref, **tol)
    # nothing outside the view is touched
    assert torch.all(y[..., :out_co] == -77.0) and torch.all(y[..., out_co + cout:] == -77.0)
    if use_aux:
        torch.testing.assert_close(_nchw(aux[..., 8:]), pre, rtol=1e-2, atol=1e-2)
        assert torch.all(aux[..., :8] == -77.0)


@pytest.mark.parametrize("shape", [(1, 13, 13, 256, 255, 1, "none", True),      # plain fp32 head (direct per-lane epilogue)
                                   (1, 40, 40, 64, 128, 3, "leaky", False),      # conv3x3_t20v2 epilogue
                                   (1, 20, 20, 128, 64, 1, "leaky", False),      # LDS-staged gather epilogue
                                   (2, 80, 80, 256, 128, 1, "none", False)])     # conv1x1_stream epilogue
def test_conv_epilogue_keeps_nan_and_inf(shape):
    """A NaN / +-inf pre-activation leaves the epilogue as NaN / +-inf under `none` and LeakyReLU, as in the reference's
    Conv2d -> BatchNorm2d -> LeakyReLU (models/yolo_base.py:31-38): the data-independent min(max(v, lo), hi) form once turned a NaN
    into +inf, which a plain head then decoded to a confidence-1 detection.  Injected through the bias of three couts."""
    from pytorch_yolo_amd import kernels as K
    from pytorch_yolo_amd._lib import ACT_LEAKY01, ACT_NONE, DT_BF16, DT_F32
    n, h, w, cin, cout, k, act, f32 = shape
    g = torch.Generator().manual_seed(11)
    x = torch.randn(n, cin, h, w, generator=g)
    wt = torch.randn(cout, cin, k, k, generator=g) * (2.0 / (cin * k * k)) ** 0.5
    bias = torch.randn(cout, generator=g) * 0.1
    bias[3], bias[17], bias[cout - 2] = float("nan"), float("inf"), float("-inf")
    y = torch.zeros(n, h, w, K.roundup(cout, 8), dtype=torch.float32 if f32 else torch.bfloat16, device=DEV)
    wp, bp, kpad, cout_pad = K.pack_conv_weight(wt, bias, cin)
    d = K.conv_desc(n=n, h=h, w=w, cin=cin, in_c_total=cin, in_c_offset=0, cout=cout, out_c_total=y.shape[-1], out_c_offset=0,
                    ksize=k, stride=1, act={"leaky": ACT_LEAKY01, "none": ACT_NONE}[act], kpad=kpad, cout_pad=cout_pad,
                    out_dtype=DT_F32 if f32 else DT_BF16)
    K.conv2d(_nhwc(x), wp.to(DEV), bp.to(DEV), y, d)
    torch.cuda.synchronize()
    got = y.float().cpu()
    assert torch.isnan(got[..., 3]).all()
    assert (got[..., 17] == float("inf")).all() and (got[..., cout - 2] == float("-inf")).all()
    keep = [c for c in range(cout) if c not in (3, 17, cout - 2)]
    assert torch.isfinite(got[..., keep]).all()


BIG_CONV_CASES = [
    # shapes of the SPP-640 layer list at 16 images: each one selects a different tile configuration in yolo_conv2d_launch
    (16, 20, 20, 1024, 512, 1, True),     # 8-wave 128x128 tiles, three-stage ring (1x1 on the 20x20 maps)
    (16, 40, 40, 512, 256, 1, False),     # 16-wave 128x256 tiles, three stages (1x1 on the 40x40 maps)
    (16, 20, 20, 256, 512, 3, True),      # 128x256 tiles, 8 MFMA waves + 4 loader waves, three stages
    (8, 40, 40, 128, 512, 3, True),       # 16-wave 256x256 tiles
    (4, 80, 80, 256, 128, 1, False),      # 256x128 tiles, 32-deep stages (short-K 1x1 on the 80x80 maps)
    (2, 160, 160, 64, 128, 3, True),      # halo kernel, one 64-channel chunk, 16x16x32 MFMA, two blocks per CU
    (2, 161, 160, 32, 128, 3, False),     # 8-wave 128x128 tiles through the gather path (map does not tile by 16 well enough)
]


@pytest.mark.parametrize("case", BIG_CONV_CASES, ids=lambda c: "n%d_%dx%d_c%d-%d_k%d_r%d" % tuple(int(v) for v in c))
def test_conv_tile_configurations(case):
    """The tile rules of yolo_conv2d_launch are exercised at the sizes that trigger them (the small CONV_CASES all land
    in two or three configurations): every configuration against fp32 torch on the same bf16-rounded operands."""
    from pytorch_yolo_amd import kernels as K
    from pytorch_yolo_amd._lib import ACT_LEAKY01
    n, h, w, cin, cout, k, use_res = case
    g = torch.Generator().manual_seed(cin * 7 + cout)
    x = torch.randn(n, cin, h, w, generator=g)
    wt = torch.randn(cout, cin, k, k, generator=g) * (2.0 / (cin * k * k)) ** 0.5
    bias = torch.randn(cout, generator=g) * 0.1
    res = torch.randn(n, cout, h, w, generator=g) if use_res else None
    xin, rin = _nhwc(x), (_nhwc(res) if use_res else None)
    y = torch.empty(n, h, w, cout, dtype=torch.bfloat16, device=DEV)
    wp, bp, kpad, cout_pad = K.pack_conv_weight(wt, bias, cin)
    d = K.conv_desc(n=n, h=h, w=w, cin=cin, in_c_total=cin, in_c_offset=0, cout=cout, out_c_total=cout, out_c_offset=0,
                    ksize=k, stride=1, act=ACT_LEAKY01, kpad=kpad, cout_pad=cout_pad, res=(cout, 0) if use_res else (0, 0))
    K.conv2d(xin, wp.to(DEV), bp.to(DEV), y, d, residual=rin)
    torch.cuda.synchronize()
    ref = F.leaky_relu(F.conv2d(_bf16r(x), _bf16r(wt), bias, padding=(k - 1) // 2), 0.1)
    if use_res:
        ref = ref + _bf16r(res)
    torch.testing.assert_close(_nchw(y), ref, rtol=1e-2, atol=2e-2)


@pytest.mark.parametrize("n,h,w,c,use_aux", [(1, 80, 80, 256, False), (2, 96, 80, 128, True), (1, 94, 100, 64, False),
                                              (2, 80, 112, 256, True), (3, 96, 160, 64, True), (16, 320, 320, 64, False),
                                              # >= 256 tiles of 20x20 covering >= 90 % of the map: conv_resunit_t20.hip (whole tiles
                                              # with the pre-add copy; partial edge tiles in both directions)
                                              (3, 200, 200, 64, True), (3, 190, 230, 64, True), (4, 163, 178, 64, False),
                                              # round 3, resunit_t20w_kernel: C = 128 on >= 512 tiles of 20x20, C = 256 on >= 512
                                              # tiles of 20x8 (whole tiles; partial edge tiles in both directions + pre-add copy)
                                              (2, 320, 320, 128, False), (3, 290, 310, 128, True),
                                              (4, 160, 160, 256, True), (5, 150, 170, 256, False)])
def test_fused_residual_unit(n, h, w, c, use_aux):
    _check_fused_residual_unit(n, h, w, c, use_aux)


def _check_fused_residual_unit(n, h, w, c, use_aux):
    """yolo_resunit_fwd (1x1 -> 3x3 -> add in one launch) against fp32 torch on the same bf16-rounded operands
    (the 1x1 output rounded to bf16 like the stored intermediate of the two-kernel path), and against the
    two-kernel path itself; partial edge tiles, channel-offset views and the pre-add copy are exercised."""
    from pytorch_yolo_amd import kernels as K
    from pytorch_yolo_amd._lib import ACT_LEAKY01
    assert K.resunit_supported(c, h, w)
    g = torch.Generator().manual_seed(c + h)
    x = torch.randn(n, c, h, w, generator=g)
    w1 = torch.randn(c // 2, c, 1, 1, generator=g) * (2.0 / c) ** 0.5
    b1 = torch.randn(c // 2, generator=g) * 0.1
    w2 = torch.randn(c, c // 2, 3, 3, generator=g) * (2.0 / (c // 2 * 9)) ** 0.5
    b2 = torch.randn(c, generator=g) * 0.1
    in_ct, in_co, out_ct, out_co = c + 16, 8, c + 8, 8
    xin = torch.zeros(n, h, w, in_ct, dtype=torch.bfloat16, device=DEV)
    xin[..., in_co:in_co + c] = _nhwc(x)
    y = torch.full((n, h, w, out_ct), -77.0, dtype=torch.bfloat16, device=DEV)
    aux = torch.full((n, h, w, c + 8), -77.0, dtype=torch.bfloat16, device=DEV) if use_aux else None
    w1p, b1p, kpad1, cpad1 = K.pack_conv_weight(w1, b1, c)
    w2p, b2p, kpad2, cpad2 = K.pack_conv_weight(w2, b2, c // 2)
    d = K.conv_desc(n=n, h=h, w=w, cin=c // 2, in_c_total=in_ct, in_c_offset=in_co, cout=c, out_c_total=out_ct,
                    out_c_offset=out_co, ksize=3, stride=1, act=ACT_LEAKY01, kpad=kpad2, cout_pad=cpad2,
                    aux=(c + 8, 8) if use_aux else (0, 0))
    K.resunit(xin, w1p.to(DEV), b1p.to(DEV), w2p.to(DEV), b2p.to(DEV), y, d, kpad1, cpad1, y_preadd=aux)
    torch.cuda.synchronize()
    mid = _bf16r(F.leaky_relu(F.conv2d(_bf16r(x), _bf16r(w1), b1), 0.1))
    pre = F.leaky_relu(F.conv2d(mid, _bf16r(w2), b2, padding=1), 0.1)
    ref = pre + _bf16r(x)
    got = _nchw(y[..., out_co:])
    torch.testing.assert_close(got, ref, rtol=1e-2, atol=2e-2)
    assert torch.all(y[..., :out_co] == -77.0)
    if use_aux:
        torch.testing.assert_close(_nchw(aux[..., 8:]), pre, rtol=1e-2, atol=2e-2)
        assert torch.all(aux[..., :8] == -77.0)
    # two-kernel path on the same operands: differences only from fp32 summation order (<= 1 bf16 ulp of a few values)
    midb = torch.empty(n, h, w, c // 2, dtype=torch.bfloat16, device=DEV)
    d1 = K.conv_desc(n=n, h=h, w=w, cin=c, in_c_total=in_ct, in_c_offset=in_co, cout=c // 2, out_c_total=c // 2,
                     out_c_offset=0, ksize=1, stride=1, act=ACT_LEAKY01, kpad=kpad1, cout_pad=cpad1)
    K.conv2d(xin, w1p.to(DEV), b1p.to(DEV), midb, d1)
    y2 = torch.empty(n, h, w, c, dtype=torch.bfloat16, device=DEV)
    d2 = K.conv_desc(n=n, h=h, w=w, cin=c // 2, in_c_total=c // 2, in_c_offset=0, cout=c, out_c_total=c, out_c_offset=0,
                     ksize=3, stride=1, act=ACT_LEAKY01, kpad=kpad2, cout_pad=cpad2, res=(in_ct, in_co))
    K.conv2d(midb, w2p.to(DEV), b2p.to(DEV), y2, d2, residual=xin)
    torch.cuda.synchronize()
    diff = (y[..., out_co:].float() - y2.float()).abs()
    assert float(diff.max()) <= 0.07 and float((diff > 0).float().mean()) < 0.25, (float(diff.max()), float((diff > 0).float().mean()))
    # run to run identical: the kernels hand LDS buffers between waves and phases and issue their MFMAs as asm the compiler
    # cannot see through - a missed wait or a register reused under a running MFMA shows up as a timing-dependent result
    for _ in range(2):
        y3 = torch.full_like(y, -77.0)
        K.resunit(xin, w1p.to(DEV), b1p.to(DEV), w2p.to(DEV), b2p.to(DEV), y3, d, kpad1, cpad1, y_preadd=aux)
        torch.cuda.synchronize()
        assert torch.equal(y, y3)


@pytest.mark.parametrize("n,h,w,c", [(3, 200, 200, 64), (2, 320, 320, 128)])
def test_fused_residual_unit_relu6(n, h, w, c):
    """The fused-unit kernels' non-LeakyReLU instantiation (round 3: they are templated on the activation): ReLU6 on both convs
    against fp32 torch."""
    from pytorch_yolo_amd import kernels as K
    from pytorch_yolo_amd._lib import ACT_RELU6
    g = torch.Generator().manual_seed(c + 3)
    x = torch.randn(n, c, h, w, generator=g)
    w1 = torch.randn(c // 2, c, 1, 1, generator=g) * (2.0 / c) ** 0.5
    b1 = torch.randn(c // 2, generator=g) * 0.1 + 0.3
    w2 = torch.randn(c, c // 2, 3, 3, generator=g) * (2.0 / (c // 2 * 9)) ** 0.5 * 3.0          # (some outputs above 6)
    b2 = torch.randn(c, generator=g) * 0.1
    y = torch.empty(n, h, w, c, dtype=torch.bfloat16, device=DEV)
    w1p, b1p, kpad1, cpad1 = K.pack_conv_weight(w1, b1, c)
    w2p, b2p, kpad2, cpad2 = K.pack_conv_weight(w2, b2, c // 2)
    d = K.conv_desc(n=n, h=h, w=w, cin=c // 2, in_c_total=c, in_c_offset=0, cout=c, out_c_total=c, out_c_offset=0, ksize=3, stride=1,
                    act=ACT_RELU6, kpad=kpad2, cout_pad=cpad2)
    assert K.resunit_form(c, n, h, w) == 3
    K.resunit(_nhwc(x), w1p.to(DEV), b1p.to(DEV), w2p.to(DEV), b2p.to(DEV), y, d, kpad1, cpad1)
    torch.cuda.synchronize()
    mid = _bf16r(F.relu6(F.conv2d(_bf16r(x), _bf16r(w1), b1)))
    pre = F.relu6(F.conv2d(mid, _bf16r(w2), b2, padding=1))
    assert float((pre == 6.0).float().mean()) > 0.01 and float((pre == 0.0).float().mean()) > 0.1
    torch.testing.assert_close(_nchw(y), pre + _bf16r(x), rtol=1e-2, atol=3e-2)


@pytest.mark.parametrize("n,h,w,cin,hidden,cout,stride", [
    (2, 26, 30, 32, 32, 16, 1),       # first MobileNetV2 block: no expand conv
    (2, 40, 36, 16, 96, 24, 2),       # expand + stride 2, odd tile remainders
    (1, 23, 19, 24, 144, 24, 1),      # residual, hidden not a multiple of 32, odd sizes
    (3, 33, 41, 24, 144, 32, 2),
    (2, 16, 24, 32, 192, 32, 1),      # residual
    (1, 52, 52, 32, 192, 64, 2),
    (4, 104, 104, 24, 144, 24, 1),    # more tiles than workgroups: the persistent loop and its prefetch
    # the wide form (csrc/conv_mbwide.hip: hidden dimension streamed in chunks of 64)
    (12, 52, 52, 64, 384, 64, 1),     # 13x13 tiles (192 of them), residual
    (13, 50, 45, 96, 192, 96, 1),     # 13x13 tiles with partial ones at both edges, three K steps
    (70, 26, 26, 64, 128, 64, 1),     # 13x13: more tiles than workgroups
    (2, 26, 26, 64, 128, 96, 1),      # 7x7 tiles, no residual
    (3, 13, 13, 160, 320, 160, 1),    # 7x7, five K steps, residual
    (2, 13, 13, 160, 192, 320, 1),    # 20 cout tiles
    (2, 26, 26, 96, 192, 160, 2),     # stride 2
    (1, 27, 23, 64, 192, 24, 2),      # stride 2, odd sizes, cout not a multiple of 16
    (20, 28, 28, 96, 128, 96, 1),     # 7x7: more tiles than workgroups
    # round 5, the row-strip form of the narrow blocks at sizes with several column segments and bands (partial last segment / band)
    (3, 104, 104, 16, 96, 24, 2), (2, 61, 83, 32, 32, 16, 1), (2, 57, 70, 32, 192, 64, 2), (5, 52, 52, 32, 192, 32, 1),
    # round 5, late: residual blocks in the tile form's 256- / 512-thread instances (the residual is read into registers before the
    # next tile's halo overwrites the x tile; a wave of the 256-thread instance owns TWO projection tiles here)
    (3, 30, 26, 32, 32, 32, 1), (2, 24, 40, 16, 64, 16, 1)])
@pytest.mark.parametrize("form", ["strip", "tile"])
def test_fused_inverted_residual(n, h, w, cin, hidden, cout, stride, form):
    from pytorch_yolo_amd import kernels as K
    from pytorch_yolo_amd._lib import load
    if form == "strip" and K.mbconv_form(cin, hidden, cout, stride) != 1:
        pytest.skip("the wide blocks have one form")
    old = load().yolo_set_tuning(4, 128 if form == "strip" else 0)    # YOLO_MBCONV_DEBUG bit 128: the row-strip form (opt-in, round 5)
    try:
        _check_fused_inverted_residual(n, h, w, cin, hidden, cout, stride)
    finally:
        load().yolo_set_tuning(4, old)


def _check_fused_inverted_residual(n, h, w, cin, hidden, cout, stride):
    """yolo_mbconv_fwd (expand 1x1 -> depthwise 3x3 -> projection 1x1 [-> add] in one launch) against fp32 torch on
    the same bf16-rounded operands (both intermediates rounded to bf16 like the stored tensors of the three-launch
    path) and against the three-launch path itself; channel-offset views, image borders inside a tile and the
    zero padding of the EXPANDED map are exercised."""
    from pytorch_yolo_amd import kernels as K
    from pytorch_yolo_amd._lib import ACT_NONE, ACT_RELU6
    assert K.mbconv_supported(cin, hidden, cout, stride)
    has_exp, has_res = hidden != cin, stride == 1 and cin == cout
    g = torch.Generator().manual_seed(cin * 7 + hidden + h)
    x = torch.randn(n, cin, h, w, generator=g).clamp_(-3, 3)
    we = torch.randn(hidden, cin, 1, 1, generator=g) * (2.0 / cin) ** 0.5 if has_exp else None
    be = torch.randn(hidden, generator=g) * 0.5 if has_exp else None
    wd = torch.randn(hidden, 1, 3, 3, generator=g) * (2.0 / 9) ** 0.5
    bd = torch.randn(hidden, generator=g) * 0.5
    wp = torch.randn(cout, hidden, 1, 1, generator=g) * (1.0 / hidden) ** 0.5
    bp = torch.randn(cout, generator=g) * 0.1
    in_ct, in_co, out_ct, out_co = cin + 16, 8, cout + 8, 4
    xin = torch.zeros(n, h, w, in_ct, dtype=torch.bfloat16, device=DEV)
    xin[..., in_co:in_co + cin] = _nhwc(x)
    ho, wo = (h - 1) // stride + 1, (w - 1) // stride + 1
    y = torch.full((n, ho, wo, out_ct), -77.0, dtype=torch.bfloat16, device=DEV)
    packed = tuple(None if t is None else t.to(DEV) for t in K.pack_mbconv(we, be, wd, bd, wp, bp, stride=stride))
    K.mbconv(xin, packed, y, n=n, h=h, w=w, cin=cin, hidden=hidden, cout=cout, in_view=(in_ct, in_co),
             out_view=(out_ct, out_co), stride=stride, has_res=has_res)
    torch.cuda.synchronize()
    xr = _bf16r(x)
    e = _bf16r(F.conv2d(xr, _bf16r(we), be).clamp(0, 6)) if has_exp else xr
    dd = _bf16r(F.conv2d(e, wd, bd, stride=stride, padding=1, groups=hidden).clamp(0, 6))
    ref = F.conv2d(dd, _bf16r(wp), bp)
    if has_res:
        ref = ref + xr
    got = _nchw(y[..., out_co:out_co + cout])
    torch.testing.assert_close(got, ref, rtol=1e-2, atol=3e-2)
    assert torch.all(y[..., :out_co] == -77.0) and torch.all(y[..., out_co + cout:] == -77.0)
    # the three-launch path on the same operands
    cur, ct, co = xin, in_ct, in_co
    if has_exp:
        w1p, b1p, kpad1, cpad1 = K.pack_conv_weight(we, be, cin)
        eb = torch.empty(n, h, w, hidden, dtype=torch.bfloat16, device=DEV)
        d1 = K.conv_desc(n=n, h=h, w=w, cin=cin, in_c_total=in_ct, in_c_offset=in_co, cout=hidden, out_c_total=hidden,
                         out_c_offset=0, ksize=1, stride=1, act=ACT_RELU6, kpad=kpad1, cout_pad=cpad1)
        K.conv2d(xin, w1p.to(DEV), b1p.to(DEV), eb, d1)
        cur, ct, co = eb, hidden, 0
    db = torch.empty(n, ho, wo, hidden, dtype=torch.bfloat16, device=DEV)
    K.dwconv3x3(cur, wd.reshape(hidden, 9).t().contiguous().to(DEV), bd.to(DEV), db, n=n, h=h, w=w, c=hidden,
                in_view=(ct, co), out_view=(hidden, 0), stride=stride, act=ACT_RELU6)
    w3p, b3p, kpad3, cpad3 = K.pack_conv_weight(wp, bp, hidden)
    y3 = torch.empty(n, ho, wo, cout, dtype=torch.bfloat16, device=DEV)
    d3 = K.conv_desc(n=n, h=ho, w=wo, cin=hidden, in_c_total=hidden, in_c_offset=0, cout=cout, out_c_total=cout,
                     out_c_offset=0, ksize=1, stride=1, act=ACT_NONE, kpad=kpad3, cout_pad=cpad3,
                     res=(in_ct, in_co) if has_res else (0, 0))
    K.conv2d(db, w3p.to(DEV), b3p.to(DEV), y3, d3, residual=xin if has_res else None)
    torch.cuda.synchronize()
    diff = (y[..., out_co:out_co + cout].float() - y3.float()).abs()
    assert float(diff.max()) <= 0.07 and float((diff > 0).float().mean()) < 0.25, (float(diff.max()), float((diff > 0).float().mean()))


@pytest.mark.parametrize("n,h,w,cin,k,nc,act", [(2, 20, 20, 256, 1, 80, "leaky"), (3, 13, 13, 512, 1, 80, "none"),
                                                  (1, 10, 12, 64, 3, 80, "leaky"), (2, 8, 8, 72, 1, 3, "none"),
                                                  (1, 26, 26, 128, 1, 20, "none"), (2, 14, 14, 96, 1, 80, "leaky"),
                                                  (9, 20, 20, 256, 1, 80, "leaky"), (1, 37, 23, 64, 1, 3, "none")])
def test_fused_head_decode(n, h, w, cin, k, nc, act):
    """yolo_head_decode_fwd (head conv with the YOLOLayer decode as its epilogue) against the two-launch path
    (yolo_conv2d_fwd to an fp32 NHWC head + yolo_decode_fwd): p and io agree to fp32 summation-order noise, image
    boundaries inside a pixel tile, 3x3 heads (YOLOv3 / Lite head 3) and small class counts are exercised.
    (The epilogue is csrc/head_epilogue.h.)"""
    _check_fused_head_decode(n, h, w, cin, k, nc, act)


def _check_fused_head_decode(n, h, w, cin, k, nc, act):
    from pytorch_yolo_amd import kernels as K
    from pytorch_yolo_amd._lib import ACT_LEAKY01, ACT_NONE, DT_F32
    na, no = 3, nc + 5
    cout = na * no
    anchors = [(10., 13.), (33., 23.), (59., 119.)]
    stride = 16.0
    g = torch.Generator().manual_seed(h * 7 + cin)
    x = torch.randn(n, cin, h, w, generator=g)
    wt = torch.randn(cout, cin, k, k, generator=g) * (1.0 / (cin * k * k)) ** 0.5
    bias = torch.randn(cout, generator=g)
    xin = torch.zeros(n, h, w, cin + 8, dtype=torch.bfloat16, device=DEV)
    xin[..., 8:] = _nhwc(x)
    wp, bp, kpad, cout_pad = K.pack_conv_weight(wt, bias, cin)
    a = {"leaky": ACT_LEAKY01, "none": ACT_NONE}[act]
    rows_total, row_off = na * h * w + 7, 5
    assert K.head_decode_supported(cout, na, nc)
    d = K.conv_desc(n=n, h=h, w=w, cin=cin, in_c_total=cin + 8, in_c_offset=8, cout=cout, out_c_total=K.roundup(cout, 8),
                    out_c_offset=0, ksize=k, stride=1, act=a, kpad=kpad, cout_pad=cout_pad, out_dtype=DT_F32)
    io = torch.full((n, rows_total, no), -7.0, device=DEV)
    p = torch.full((n, na, h, w, no), -7.0, device=DEV)
    assert K.head_decode_pick(d, na, nc).startswith("igemm<") and ",decode>" in K.head_decode_pick(d, na, nc)
    K.head_decode(xin, wp.to(DEV), bp.to(DEV), d, anchors, nc, stride, io, row_off, p)
    head = torch.zeros(n, h, w, K.roundup(cout, 8), device=DEV)
    K.conv2d(xin, wp.to(DEV), bp.to(DEV), head, d)
    io2 = torch.full((n, rows_total, no), -7.0, device=DEV)
    p2 = torch.full((n, na, h, w, no), -7.0, device=DEV)
    K.decode(head, anchors, nc, stride, io2, row_off, p2)
    torch.cuda.synchronize()
    torch.testing.assert_close(p, p2, rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(io, io2, rtol=2e-5, atol=1e-4)
    assert torch.all(io[:, :row_off] == -7.0) and torch.all(io[:, row_off + na * h * w:] == -7.0)
    # and against fp32 torch on the bf16-rounded operands (reference formulas, yolo_layer.py:90-96)
    ref = F.conv2d(_bf16r(x), _bf16r(wt), bias, padding=(k - 1) // 2)
    if act == "leaky":
        ref = F.leaky_relu(ref, 0.1)
    ref = ref.view(n, na, no, h, w).permute(0, 1, 3, 4, 2).contiguous()
    torch.testing.assert_close(p.cpu(), ref, rtol=1e-4, atol=2e-4)


@pytest.mark.parametrize("n,h,w,cin,k,nc,act,conf", [(2, 20, 20, 256, 1, 80, "leaky", 0.05), (3, 13, 13, 512, 1, 80, "none", 0.1),
                                                      (1, 10, 12, 64, 3, 80, "leaky", 0.02), (2, 8, 8, 72, 1, 3, "none", 0.2),
                                                      (1, 26, 26, 128, 1, 20, "none", 0.1), (2, 14, 14, 96, 1, 80, "leaky", 0.001),
                                                      (5, 3, 3, 64, 1, 1, "none", 0.3), (32, 2, 2, 128, 1, 80, "none", 0.01),
                                                      (9, 20, 20, 256, 1, 80, "leaky", 0.05), (3, 37, 23, 512, 1, 20, "none", 0.05)])
def test_head_decode_filter_is_the_plain_head_plus_nms(n, h, w, cin, k, nc, act, conf):
    """The compact NMS form (round 4: yolo_head_decode_filter_fwd -> yolo_nms_merge_compact; detect()
    never writes io) against the plain one (yolo_head_decode_fwd stores io, yolo_nms_merge filters and merges it) on the same
    operands: counts, kept rows and all 7 columns BIT-EQUAL - the epilogue's row filter repeats nms_filter's arithmetic on the same
    decoded values.  Cases: image boundaries inside a 64-pixel tile and several images per wave (3x3 and 2x2 maps), 8- and 4-wave
    head tiles (cin 72 / 96), nc = 1 / 3 / 20 / 80, thresholds from 0.001 (nearly every row survives) to 0.3, a row offset and
    foreign rows in io, and non-finite logits (a few +inf input values: rows with inf / NaN scores or boxes are dropped alike)."""
    from pytorch_yolo_amd import kernels as K
    from pytorch_yolo_amd._lib import ACT_LEAKY01, ACT_NONE, DT_F32
    from pytorch_yolo_amd.utils.utils import MAX_PER_CLASS, MIN_WH, nms_capacity
    na, no = 3, nc + 5
    cout = na * no
    anchors = [(10., 13.), (33., 23.), (59., 119.)]
    stride = 16.0
    g = torch.Generator().manual_seed(h * 7 + cin + n)
    x = torch.randn(n, cin, h, w, generator=g)
    flat = x.view(-1)
    flat[torch.randint(0, flat.numel(), (max(1, flat.numel() // 5000),), generator=g)] = float("inf")
    wt = torch.randn(cout, cin, k, k, generator=g) * (2.0 / (cin * k * k)) ** 0.5
    bias = torch.randn(cout, generator=g)
    xin = torch.zeros(n, h, w, cin + 8, dtype=torch.bfloat16, device=DEV)
    xin[..., 8:] = _nhwc(x)
    wp, bp, kpad, cout_pad = K.pack_conv_weight(wt, bias, cin)
    wp, bp = wp.to(DEV), bp.to(DEV)
    a = {"leaky": ACT_LEAKY01, "none": ACT_NONE}[act]
    rows_total, row_off = na * h * w + 7, 5
    d = K.conv_desc(n=n, h=h, w=w, cin=cin, in_c_total=cin + 8, in_c_offset=8, cout=cout, out_c_total=K.roundup(cout, 8),
                    out_c_offset=0, ksize=k, stride=1, act=a, kpad=kpad, cout_pad=cout_pad, out_dtype=DT_F32)
    cap = nms_capacity(rows_total, nc)
    mk = lambda: (torch.full((n, cap, 7), -3.0, device=DEV), torch.full((n, cap), -3, dtype=torch.int32, device=DEV),
                  torch.full((n,), -3, dtype=torch.int32, device=DEV))
    # plain: io (foreign rows zero: they never survive) -> yolo_nms_merge
    io = torch.zeros((n, rows_total, no), device=DEV)
    K.head_decode(xin, wp, bp, d, anchors, nc, stride, io, row_off, None)
    out_a = mk()
    ws_a = torch.empty(K.nms_workspace_bytes(n, rows_total, nc), dtype=torch.uint8, device=DEV)
    K.nms_merge(io, conf, 0.5, *out_a, ws_a, min_wh=MIN_WH, max_per_class=MAX_PER_CLASS)
    # compact: the head filters its own rows, no io
    ws_b = torch.full((K.nms_compact_workspace_bytes(n, rows_total, nc),), 0xCD, dtype=torch.uint8, device=DEV)
    out_b = mk()
    K.head_decode_filter(xin, wp, bp, d, anchors, nc, stride, rows_total, row_off, conf, ws_b, min_wh=MIN_WH)
    K.nms_merge_compact(ws_b, n, rows_total, nc, 0.5, *out_b, max_per_class=MAX_PER_CLASS)
    torch.cuda.synchronize()
    cnt = out_a[2].cpu()
    assert torch.equal(out_b[2].cpu(), cnt), (out_b[2].cpu().tolist(), cnt.tolist())
    assert int(cnt.sum()) > 0, "the case is vacuous: nothing survives"
    assert not bool(torch.isfinite(io).all()), "the case has no non-finite rows"
    for b in range(n):
        m = int(cnt[b])
        assert torch.equal(out_a[1][b, :m], out_b[1][b, :m]), f"image {b}: kept rows differ"
        assert torch.equal(out_a[0][b, :m], out_b[0][b, :m]), f"image {b}: detections differ"
    print(f"[compact NMS {n}x{h}x{w} nc {nc} conf {conf}] detections per image {cnt.tolist()[:8]} - bit-equal to the plain form")


@pytest.mark.parametrize("nc", [80, 20, 1])
def test_head_filter_class_choice_on_crafted_logits(nc):
    """The filter epilogue (csrc/head_epilogue.h, round 5) decodes only the classes whose LOGIT lies in a window below the row's
    maximum and takes the reference's "first maximum of the DECODED scores" among them.  Crafted logits - an identity head conv feeds
    bf16-exact values straight through - hold it to the plain form (head -> io -> nms_filter, which decodes and scans everything) on
    exactly the cases that rule has to get right, bit for bit (counts, kept rows, all 7 columns):
      saturated ties (logits 17.5 / 20 / 30 all decode to 1.0f: the FIRST wins, reference utils/utils.py:212 torch.max), exact ties,
      near ties inside and just outside the 0.25 window, a maximum above 12 with runners-up at 11.2 / 10.9, all classes equal, huge
      logits, an objectness of ~0 and objectness values around the threshold, w / h below min_wh.  (Non-finite logits cannot be
      crafted this way - 0 x inf poisons the whole pixel - and are covered by test_head_decode_filter_is_the_plain_head_plus_nms.)"""
    from pytorch_yolo_amd import kernels as K
    from pytorch_yolo_amd._lib import ACT_NONE, DT_F32
    from pytorch_yolo_amd.utils.utils import MAX_PER_CLASS, MIN_WH, nms_capacity
    na, no = 3, nc + 5
    cout = na * no
    cin = K.roundup(cout, 32)
    anchors = [(10., 13.), (33., 23.), (59., 119.)]
    stride, conf = 16.0, 0.3
    n, h, w = 2, 6, 8
    g = torch.Generator().manual_seed(nc)
    # logits per (image, anchor, y, x, 5 + nc), bf16-exact
    lg = (torch.randn(n, na, h, w, no, generator=g) * 2.0 - 1.0).to(torch.bfloat16).float()
    lg[..., 4] = 3.0                                                # objectness passes by default
    lg[..., 2:4] = 0.5                                              # boxes wider than min_wh
    rows = lg.view(n, na * h * w, no)
    def put(i, cls_vals, obj=None, wh=None):
        r = rows[0, i]
        r[5:] = -6.0
        for c, v in cls_vals.items():
            if c < nc:
                r[5 + c] = v
        if obj is not None:
            r[4] = obj
        if wh is not None:
            r[2:4] = wh
    put(0, {3: 20.0, 7: 30.0, 11: 17.5})                           # saturated ties: class 3
    put(1, {9: 30.0, 2: 17.5})                                     # ... class 2 (the smaller logit comes first)
    put(2, {4: 5.0, 6: 5.0, 8: 4.96875})                           # exact tie: class 4
    put(3, {12: 2.0, 5: 1.875, 1: 1.6875})                         # inside / outside the window: class 12
    put(4, {15: 12.5, 0: 11.25, 17: 10.875})                       # maximum above 12
    put(5, {c: -30.0 for c in range(nc)})                          # every score ~1e-13 and equal: class 0 (conf far below the threshold)
    put(6, {3: 88.0, 1: 88.0})                                     # huge logits, both 1.0f: class 1
    put(9, {2: 25.0}, obj=-88.0)                                   # objectness ~0: dropped
    put(10, {2: 25.0}, obj=-0.84375)                               # s(obj) just below / at the threshold region
    put(11, {2: 25.0}, obj=-0.8515625)
    put(12, {2: 25.0}, wh=-3.0)                                    # w, h below min_wh for the small anchors
    put(13, {0: 0.0, 1: 0.0})                                      # tie at 0.5: class 0
    put(14, {nc - 1: 9.0})                                         # the last class (second slot of a lane when nc > 64)
    assert torch.equal(lg.to(torch.bfloat16).float(), lg), "the crafted logits must be bf16-exact"
    x = torch.zeros(n, cin, h, w)
    x[:, :cout] = lg.permute(0, 1, 4, 2, 3).reshape(n, cout, h, w)
    wt = torch.zeros(cout, cin, 1, 1)
    wt[torch.arange(cout), torch.arange(cout), 0, 0] = 1.0
    bias = torch.zeros(cout)
    xin = _nhwc(x)
    wp, bp, kpad, cout_pad = K.pack_conv_weight(wt, bias, cin)
    wp, bp = wp.to(DEV), bp.to(DEV)
    rows_total, row_off = na * h * w, 0
    d = K.conv_desc(n=n, h=h, w=w, cin=cin, in_c_total=cin, in_c_offset=0, cout=cout, out_c_total=K.roundup(cout, 8),
                    out_c_offset=0, ksize=1, stride=1, act=ACT_NONE, kpad=kpad, cout_pad=cout_pad, out_dtype=DT_F32)
    cap = nms_capacity(rows_total, nc)
    mk = lambda: (torch.full((n, cap, 7), -3.0, device=DEV), torch.full((n, cap), -3, dtype=torch.int32, device=DEV),
                  torch.full((n,), -3, dtype=torch.int32, device=DEV))
    io = torch.zeros((n, rows_total, no), device=DEV)
    p = torch.zeros((n, na, h, w, no), device=DEV)
    K.head_decode(xin, wp, bp, d, anchors, nc, stride, io, row_off, p)
    torch.cuda.synchronize()
    assert torch.equal(p.cpu(), lg), "the identity head does not pass the crafted logits through"
    out_a = mk()
    ws_a = torch.empty(K.nms_workspace_bytes(n, rows_total, nc), dtype=torch.uint8, device=DEV)
    K.nms_merge(io, conf, 0.5, *out_a, ws_a, min_wh=MIN_WH, max_per_class=MAX_PER_CLASS)
    ws_b = torch.full((K.nms_compact_workspace_bytes(n, rows_total, nc),), 0xCD, dtype=torch.uint8, device=DEV)
    out_b = mk()
    K.head_decode_filter(xin, wp, bp, d, anchors, nc, stride, rows_total, row_off, conf, ws_b, min_wh=MIN_WH)
    K.nms_merge_compact(ws_b, n, rows_total, nc, 0.5, *out_b, max_per_class=MAX_PER_CLASS)
    torch.cuda.synchronize()
    cnt = out_a[2].cpu()
    assert torch.equal(out_b[2].cpu(), cnt), (out_b[2].cpu().tolist(), cnt.tolist())
    for b in range(n):
        m = int(cnt[b])
        assert torch.equal(out_a[1][b, :m], out_b[1][b, :m]), f"image {b}: kept rows differ"
        assert torch.equal(out_a[0][b, :m], out_b[0][b, :m]), f"image {b}: detections differ"
    # the crafted rows of image 0 really exercise the cases: which class a kept row got (NMS may merge rows of one class, so look
    # at the rows through a merge-free pass: at nms_thres = 0.9999 a pivot suppresses itself only)
    out_c = mk()
    K.head_decode_filter(xin, wp, bp, d, anchors, nc, stride, rows_total, row_off, conf, ws_b, min_wh=MIN_WH)
    K.nms_merge_compact(ws_b, n, rows_total, nc, 0.9999, *out_c, max_per_class=MAX_PER_CLASS)
    torch.cuda.synchronize()
    m0 = int(out_c[2][0])
    kept = {int(r): int(c) for r, c in zip(out_c[1][0, :m0].cpu().tolist(), out_c[0][0, :m0, 6].cpu().tolist())}
    if nc == 80:
        want = {0: 3, 1: 2, 2: 4, 3: 12, 4: 15, 6: 1, 13: 0, 14: 79}
        for r, c in want.items():
            assert kept.get(r) == c, f"row {r}: class {kept.get(r)} instead of {c}"
    for r in ((5, 9) if nc > 1 else (9,)):               # (nc == 1: the class score is the reference's constant 1, row 5 survives)
        assert r not in kept, f"row {r} must be dropped"


@pytest.mark.parametrize("n,cin,h,w", [(2, 3, 64, 64), (1, 3, 70, 106), (1, 1, 37, 50), (2, 3, 192, 352), (1, 3, 2, 2), (1, 3, 33, 17),
                                        (3, 3, 416, 416), (40, 3, 96, 96)])
def test_fused_stem(n, cin, h, w):
    """yolo_stem_fwd (conv3x3/s1 cin->32 + conv3x3/s2 32->64 from the float32 NCHW batch in one launch) against fp32
    torch on the same bf16-rounded operands (intermediate rounded to bf16 like the two-kernel path stores it);
    odd sizes exercise partial tiles, image borders of both convs and the zero padding of the intermediate."""
    from pytorch_yolo_amd import kernels as K
    from pytorch_yolo_amd._lib import ACT_LEAKY01
    g = torch.Generator().manual_seed(h * w + cin)
    x = torch.rand(n, cin, h, w, generator=g)
    w1 = torch.randn(32, cin, 3, 3, generator=g) * (2.0 / (cin * 9)) ** 0.5
    b1 = torch.randn(32, generator=g) * 0.1
    w2 = torch.randn(64, 32, 3, 3, generator=g) * (2.0 / (32 * 9)) ** 0.5
    b2 = torch.randn(64, generator=g) * 0.1
    ho, wo = (h - 1) // 2 + 1, (w - 1) // 2 + 1
    w1p, b1p, kpad1, _ = K.pack_conv_weight(w1, b1, 8)
    w2p, b2p, kpad2, cpad2 = K.pack_conv_weight(w2, b2, 32)
    assert kpad1 >= 80
    out_ct, out_co = 80, 8
    y = torch.full((n, ho, wo, out_ct), -77.0, dtype=torch.bfloat16, device=DEV)
    d = K.conv_desc(n=n, h=h, w=w, cin=32, in_c_total=32, in_c_offset=0, cout=64, out_c_total=out_ct, out_c_offset=out_co,
                    ksize=3, stride=2, act=ACT_LEAKY01, kpad=kpad2, cout_pad=cpad2)
    K.stem(x.to(DEV), cin, w1p.to(DEV), b1p.to(DEV), kpad1, w2p.to(DEV), b2p.to(DEV), y, d)
    torch.cuda.synchronize()
    mid = _bf16r(F.leaky_relu(F.conv2d(_bf16r(x), _bf16r(w1), b1, padding=1), 0.1))
    ref = F.leaky_relu(F.conv2d(mid, _bf16r(w2), b2, stride=2, padding=1), 0.1)
    torch.testing.assert_close(_nchw(y[..., out_co:out_co + 64]), ref, rtol=1e-2, atol=2e-2)
    assert torch.all(y[..., :out_co] == -77.0) and torch.all(y[..., out_co + 64:] == -77.0)
    # and what the two launches it replaces give (first-layer kernel -> bf16 intermediate -> stride-2 conv).  Not bit for bit:
    # the fused kernel sums conv1's 27 products in another order (K = 48 instead of K = 80 inside the MFMA) and the stride-2 conv's
    # 288 in the order of its own tile shape, so an intermediate value can land on the neighbouring bf16 and an output likewise
    import ctypes as C
    from pytorch_yolo_amd._lib import check, load
    mid_dev = torch.empty(n, h, w, 32, dtype=torch.bfloat16, device=DEV)
    y2 = torch.full_like(y, -77.0)
    xd, w1d, b1d = x.to(DEV), w1p.to(DEV), b1p.to(DEV)
    d1 = K.conv_desc(n=n, h=h, w=w, cin=8, in_c_total=8, in_c_offset=0, cout=32, out_c_total=32, out_c_offset=0, ksize=3, stride=1,
                     act=ACT_LEAKY01, kpad=kpad1, cout_pad=32)
    check(load().yolo_conv1_nchw_f32_fwd(xd.data_ptr(), cin, w1d.data_ptr(), b1d.data_ptr(), mid_dev.data_ptr(), C.byref(d1),
                                         K.stream_ptr()), "conv1")
    K.conv2d(mid_dev, w2p.to(DEV), b2p.to(DEV), y2, d)
    torch.cuda.synchronize()
    ya, yb = y.float(), y2.float()
    torch.testing.assert_close(ya, yb, rtol=2 ** -6, atol=2e-2)
    assert float((ya != yb).float().mean()) < 0.05


def test_fused_stem_relu6_and_repeatable():
    """stem2_kernel's non-LeakyReLU instantiation (ReLU6) against fp32 torch, partial tiles on both edges, and two launches give
    the same bits (producer and consumer waves hand `mid` over through LDS with one barrier per tile)."""
    from pytorch_yolo_amd import kernels as K
    from pytorch_yolo_amd._lib import ACT_RELU6
    n, h, w = 3, 150, 214
    g = torch.Generator().manual_seed(9)
    x = torch.rand(n, 3, h, w, generator=g)
    w1 = torch.randn(32, 3, 3, 3, generator=g) * (2.0 / 27) ** 0.5 * 4.0
    b1 = torch.randn(32, generator=g) * 0.1
    w2 = torch.randn(64, 32, 3, 3, generator=g) * (2.0 / 288) ** 0.5 * 2.0
    b2 = torch.randn(64, generator=g) * 0.1
    ho, wo = (h - 1) // 2 + 1, (w - 1) // 2 + 1
    w1p, b1p, kpad1, _ = K.pack_conv_weight(w1, b1, 8)
    w2p, b2p, kpad2, cpad2 = K.pack_conv_weight(w2, b2, 32)
    d = K.conv_desc(n=n, h=h, w=w, cin=32, in_c_total=32, in_c_offset=0, cout=64, out_c_total=64, out_c_offset=0, ksize=3, stride=2,
                    act=ACT_RELU6, kpad=kpad2, cout_pad=cpad2)
    ys = []
    for _ in range(2):
        y = torch.full((n, ho, wo, 64), -77.0, dtype=torch.bfloat16, device=DEV)
        K.stem(x.to(DEV), 3, w1p.to(DEV), b1p.to(DEV), kpad1, w2p.to(DEV), b2p.to(DEV), y, d)
        torch.cuda.synchronize()
        ys.append(y)
    assert torch.equal(ys[0], ys[1])
    mid = _bf16r(F.relu6(F.conv2d(_bf16r(x), _bf16r(w1), b1, padding=1)))
    ref = F.relu6(F.conv2d(mid, _bf16r(w2), b2, stride=2, padding=1))
    assert float((mid == 6.0).float().mean()) > 0.001 and float((ref == 6.0).float().mean()) > 0.001
    torch.testing.assert_close(_nchw(ys[0]), ref, rtol=1e-2, atol=3e-2)


@pytest.mark.parametrize("cin,cout,h,w", [(3, 16, 64, 96), (3, 32, 32, 32), (1, 16, 38, 50)])
def test_first_layer_fused_with_maxpool(cin, cout, h, w):
    """yolo_conv1_pool_nchw_f32_fwd: first ConvBlock + MaxPool2d(2, 2) of YOLOv3-tiny in one launch (the planner picks it
    for ConvPoolBlock): equals conv -> bf16 rounding -> pool of the two-launch path (max is exact on bf16 values)."""
    from pytorch_yolo_amd import engine
    from pytorch_yolo_amd._lib import OP_CONV1_POOL
    from pytorch_yolo_amd.models.yolo_base import ConvPoolBlock
    from pytorch_yolo_amd.utils.synthetic import synth_images, synth_state_dict
    blk = ConvPoolBlock(cin, cout, pool_size=2, pool_stride=2).eval()
    blk.load_state_dict(synth_state_dict(blk.state_dict(), 5))
    x = synth_images(2, h, w, 7, channels=cin)
    rec = engine.Recorder(2, cin, h, w)
    blk._trace(rec, rec.input)
    plan = engine.Plan(rec, torch.device(DEV), 0, max(h, w))
    assert plan.fused_input and plan.op_array[0].kind == OP_CONV1_POOL and plan.n_ops == 1
    got = blk(x.to(DEV)).cpu()
    import os
    os.environ["YOLO_FUSE_POOL"] = "0"
    try:
        blk2 = ConvPoolBlock(cin, cout, pool_size=2, pool_stride=2).eval()
        blk2.load_state_dict(blk.state_dict())
        want = blk2(x.to(DEV)).cpu()
    finally:
        del os.environ["YOLO_FUSE_POOL"]
    assert got.shape == (2, cout, h // 2, w // 2) and torch.equal(got, want)


@pytest.mark.parametrize("cin,h,w", [(3, 37, 50), (3, 64, 64), (1, 20, 33), (8, 16, 16)])
def test_first_layer_fused_with_input_packing(cin, h, w):
    """yolo_conv1_nchw_f32_fwd: the first ConvBlock reads the float32 NCHW batch directly (no packed copy)."""
    from oracle.blocks import conv_bn_leaky
    from pytorch_yolo_amd import engine
    from pytorch_yolo_amd.models.yolo_base import ConvBlock
    from pytorch_yolo_amd.utils.synthetic import synth_images, synth_state_dict
    blk = ConvBlock(cin, 32, size=3, stride=1).eval()
    blk.load_state_dict(synth_state_dict(blk.state_dict(), 9))
    x = synth_images(2, h, w, 4, channels=cin)
    rec = engine.Recorder(2, cin, h, w)
    blk._trace(rec, rec.input)
    assert engine.Plan(rec, torch.device(DEV), 0, max(h, w)).fused_input
    got = blk(x.to(DEV)).cpu()
    sd = {"b." + k: v for k, v in blk.state_dict().items()}
    want = conv_bn_leaky({k: v for k, v in sd.items()}, "b", _bf16r(x))
    # weights are bf16-rounded after BN folding in the product: compare at bf16 resolution
    torch.testing.assert_close(got, want, rtol=2e-2, atol=2e-2)


@pytest.mark.parametrize("k,s", [(2, 2), (2, 1), (5, 1), (9, 1), (13, 1), (3, 2)])
def test_maxpool_exact(k, s):
    from pytorch_yolo_amd.models.yolo_base import MaxPool
    from oracle.blocks import max_pool
    x = _bf16r(torch.randn(2, 16, 13, 20))
    got = MaxPool(k, s)(x.to(DEV)).cpu()
    assert torch.equal(got, max_pool(x, k, s))


@pytest.mark.parametrize("h,w", [(10, 13), (64, 80), (7, 7), (103, 51)])
def test_maxpool_ceil_mode_exact(h, w):
    """MaxPool2d(3, 2, ceil_mode=True) of the SqueezeNet encoder: when (size - 3) is odd torch adds a row / column whose
    window hangs over the border (those taps are -inf).  Exact, through engine.Recorder.maxpool and yolo_maxpool_fwd."""
    from pytorch_yolo_amd import engine
    x = _bf16r(torch.randn(2, 16, h, w))
    want = F.max_pool2d(x, 3, 2, ceil_mode=True)
    got = engine.run_standalone(lambda g, s: g.maxpool(s, 3, 2, pad=0, ceil_mode=True), x.to(DEV)).cpu()
    assert got.shape == want.shape and torch.equal(got, want)


def test_maxpool21_kat():
    from pytorch_yolo_amd.models.yolo_base import MaxPool
    x = torch.arange(16.).view(1, 1, 4, 4).repeat(1, 8, 1, 1)
    got = MaxPool(2, 1)(x.to(DEV)).cpu()
    assert torch.equal(got[0, 0], torch.tensor([[5., 6, 7, 6], [9, 10, 11, 10], [13, 14, 15, 14], [9, 10, 11, 10]]))
    assert np.array_equal(got[0, 3].numpy(), load_golden("kat")["maxpool21"][0, 0])


@pytest.mark.parametrize("h,w,c", [(20, 20, 24), (13, 17, 24), (4, 3, 24), (72, 72, 24),
                                   # c % 64 == 0: spp_lines_kernel (whole 128-byte lines, radii grown from one another)
                                   (20, 20, 128), (13, 17, 64), (4, 3, 64), (1, 1, 64), (22, 23, 192), (7, 1, 64), (1, 9, 64)])
def test_spp_exact(h, w, c):
    from pytorch_yolo_amd import kernels as K
    from oracle.blocks import max_pool
    x = _bf16r(torch.randn(3, c, h, w, generator=torch.Generator().manual_seed(h * 100 + w)))
    x[0, 0, 0, 0] = float("-inf")                  # -inf inputs survive (padding is -inf too: nn.MaxPool2d)
    buf = torch.zeros(3, h, w, 4 * c, dtype=torch.bfloat16, device=DEV)
    buf[..., 3 * c:] = _nhwc(x)
    K.spp(buf, n=3, h=h, w=w, c=c)
    got = _nchw(buf)
    want = torch.cat([max_pool(x, 5, 1), max_pool(x, 9, 1), max_pool(x, 13, 1), x], 1)
    assert torch.equal(got, want)


@pytest.mark.parametrize("n,h,w,cin,cout,k,use_res", [(32, 13, 13, 256, 512, 3, False), (8, 13, 13, 1280, 64, 3, False),
                                                      (16, 13, 13, 256, 512, 3, True), (3, 13, 13, 1280, 64, 3, False),
                                                      (4, 26, 26, 128, 256, 3, False)])
def test_split_k_conv(n, h, w, cin, cout, k, use_res):
    """yolo_conv2d_splitk_fwd (layers with few pixels and a long K: several workgroups share a tile's K range, the last
    to arrive sums the fp32 partials in split order) against fp32 torch and against the plain launch of the same layer;
    two launches in a row must agree bit for bit (fixed summation order, self-resetting counters)."""
    from pytorch_yolo_amd import kernels as K
    from pytorch_yolo_amd._lib import ACT_LEAKY01
    g = torch.Generator().manual_seed(cin + cout + n)
    x = torch.randn(n, cin, h, w, generator=g)
    wt = torch.randn(cout, cin, k, k, generator=g) * (2.0 / (cin * k * k)) ** 0.5
    b = torch.randn(cout, generator=g) * 0.1
    res = torch.randn(n, cout, h, w, generator=g) if use_res else None
    wp, bp, kpad, cpad = K.pack_conv_weight(wt, b, cin)
    d = K.conv_desc(n=n, h=h, w=w, cin=cin, in_c_total=cin, in_c_offset=0, cout=cout, out_c_total=cout, out_c_offset=0,
                    ksize=k, stride=1, act=ACT_LEAKY01, kpad=kpad, cout_pad=cpad, res=(cout, 0) if use_res else (0, 0))
    splits, ws_bytes, n_cnt = K.conv2d_splitk_plan(d, has_residual=use_res)
    assert splits >= 2 and ws_bytes == splits * n * h * w * cout * 4 and n_cnt > 0
    xin, rin = _nhwc(x), (_nhwc(res) if use_res else None)
    ws = torch.empty(ws_bytes // 4, dtype=torch.float32, device=DEV)
    cnt = torch.zeros(n_cnt, dtype=torch.int32, device=DEV)
    ys = [torch.empty(n, h, w, cout, dtype=torch.bfloat16, device=DEV) for _ in range(3)]
    K.conv2d_splitk(xin, wp.to(DEV), bp.to(DEV), ys[0], d, splits, ws, cnt, residual=rin)
    K.conv2d_splitk(xin, wp.to(DEV), bp.to(DEV), ys[1], d, splits, ws, cnt, residual=rin)
    K.conv2d(xin, wp.to(DEV), bp.to(DEV), ys[2], d, residual=rin)
    torch.cuda.synchronize()
    assert int(cnt.abs().sum()) == 0                                   # counters are back at zero
    assert torch.equal(ys[0], ys[1])                                   # deterministic
    ref = F.leaky_relu(F.conv2d(_bf16r(x), _bf16r(wt), b, padding=k // 2), 0.1)
    if use_res:
        ref = ref + _bf16r(res)
    torch.testing.assert_close(_nchw(ys[0]), ref, rtol=1e-2, atol=2e-2)
    diff = (ys[0].float() - ys[2].float()).abs()
    assert float(diff.max()) <= 0.07 and float((diff > 0).float().mean()) < 0.25, (float(diff.max()), float((diff > 0).float().mean()))


@pytest.mark.parametrize("n,h,w,cin,cout,pool", [(2, 52, 52, 16, 32, True), (1, 40, 72, 32, 64, True), (2, 26, 38, 16, 64, True),
                                                 (1, 208, 208, 16, 32, True), (2, 30, 22, 32, 32, False), (1, 17, 33, 16, 32, True)])
def test_small_cin_conv_with_maxpool(n, h, w, cin, cout, pool):
    """yolo_conv3x3_pool_fwd (3x3 ConvBlock with 16 / 32 input channels + MaxPool2d(2, 2) in one launch) against fp32
    torch on the same bf16 operands, and against the two-launch path (yolo_conv2d_fwd + yolo_maxpool_fwd): the pooled
    values are maxima of the same bf16-rounded conv outputs, so they agree up to fp32 summation order.  Partial tiles,
    odd sizes (floor pooling), several tiles per workgroup and channel-offset views are exercised."""
    from pytorch_yolo_amd import kernels as K
    from pytorch_yolo_amd._lib import ACT_LEAKY01
    assert K.conv3x3_pool_supported(cin, cout)
    g = torch.Generator().manual_seed(cin + cout + h)
    x = torch.randn(n, cin, h, w, generator=g)
    wt = torch.randn(cout, cin, 3, 3, generator=g) * (2.0 / (cin * 9)) ** 0.5
    b = torch.randn(cout, generator=g) * 0.1
    in_ct, in_co, out_ct, out_co = cin + 16, 8, cout + 8, 8
    xin = torch.zeros(n, h, w, in_ct, dtype=torch.bfloat16, device=DEV)
    xin[..., in_co:in_co + cin] = _nhwc(x)
    ho, wo = (h // 2, w // 2) if pool else (h, w)
    y = torch.full((n, ho, wo, out_ct), -77.0, dtype=torch.bfloat16, device=DEV)
    wp, bp, kpad, cpad = K.pack_conv_weight(wt, b, cin)
    d = K.conv_desc(n=n, h=h, w=w, cin=cin, in_c_total=in_ct, in_c_offset=in_co, cout=cout, out_c_total=out_ct,
                    out_c_offset=out_co, ksize=3, stride=1, act=ACT_LEAKY01, kpad=kpad, cout_pad=cpad)
    K.conv3x3_pool(xin, wp.to(DEV), bp.to(DEV), y, d, pool=pool)
    torch.cuda.synchronize()
    conv = _bf16r(F.leaky_relu(F.conv2d(_bf16r(x), _bf16r(wt), b, padding=1), 0.1))
    ref = F.max_pool2d(conv, 2, 2) if pool else conv
    torch.testing.assert_close(_nchw(y[..., out_co:]), ref, rtol=1e-2, atol=2e-2)
    assert torch.all(y[..., :out_co] == -77.0)
    # two-launch path
    full = torch.empty(n, h, w, cout, dtype=torch.bfloat16, device=DEV)
    d2 = K.conv_desc(n=n, h=h, w=w, cin=cin, in_c_total=in_ct, in_c_offset=in_co, cout=cout, out_c_total=cout, out_c_offset=0,
                     ksize=3, stride=1, act=ACT_LEAKY01, kpad=kpad, cout_pad=cpad)
    K.conv2d(xin, wp.to(DEV), bp.to(DEV), full, d2)
    if pool:
        y2 = torch.empty(n, ho, wo, cout, dtype=torch.bfloat16, device=DEV)
        K.maxpool(full, y2, n=n, h=h, w=w, c=cout, in_view=(cout, 0), out_view=(cout, 0), ksize=2, stride=2, pad=0, dilation=1)
    else:
        y2 = full
    torch.cuda.synchronize()
    diff = (y[..., out_co:].float() - y2.float()).abs()
    assert float(diff.max()) <= 0.07 and float((diff > 0).float().mean()) < 0.25, (float(diff.max()), float((diff > 0).float().mean()))


@pytest.mark.parametrize("n,c,h,w", [(2, 32, 15, 18), (1, 384, 26, 26), (2, 960, 13, 13), (1, 48, 7, 33), (1, 8, 1, 1)])
def test_dwconv(n, c, h, w):
    """yolo_dwconv3x3_fwd (strip kernel: 8 output rows per thread, sliding input rows) against fp32 torch on the same
    bf16 input, strides 1 and 2, channel-offset views on both sides, heights that are not a multiple of the strip."""
    from pytorch_yolo_amd import kernels as K
    from pytorch_yolo_amd._lib import ACT_RELU6
    g = torch.Generator().manual_seed(c + h)
    for stride in (1, 2):
        x = torch.randn(n, c, h, w, generator=g)
        wt = torch.randn(c, 1, 3, 3, generator=g) * 0.3
        b = torch.randn(c, generator=g) * 0.1
        ho, wo = (h - 1) // stride + 1, (w - 1) // stride + 1
        xin = torch.zeros(n, h, w, c + 16, dtype=torch.bfloat16, device=DEV)
        xin[..., 8:8 + c] = _nhwc(x)
        y = torch.full((n, ho, wo, c + 8), -77.0, dtype=torch.bfloat16, device=DEV)
        K.dwconv3x3(xin, wt.reshape(c, 9).t().contiguous().to(DEV), b.to(DEV), y, n=n, h=h, w=w, c=c,
                    in_view=(c + 16, 8), out_view=(c + 8, 8), stride=stride, act=ACT_RELU6)
        ref = F.relu6(F.conv2d(_bf16r(x), wt, b, stride=stride, padding=1, groups=c))
        torch.testing.assert_close(_nchw(y[..., 8:]), ref, rtol=1e-2, atol=1e-2)
        assert torch.all(y[..., :8] == -77.0)


@pytest.mark.parametrize("nc,ny,nx,img", [(80, 13, 13, 416), (3, 4, 6, 96), (1, 5, 5, 160), (80, 80, 80, 640)])
def test_decode_vs_oracle(nc, ny, nx, img):
    from pytorch_yolo_amd.models.yolo_layer import YOLOLayer
    from oracle.blocks import yolo_decode
    anchors = C.TINY_ANCHORS[1]
    p_raw = torch.randn(2, 3 * (5 + nc), ny, nx) * 2.0
    layer = YOLOLayer(anchors, nc, C.TINY_ANCHORS).eval()
    io, p = layer(p_raw.to(DEV), img)
    io_ref, p_ref = yolo_decode(p_raw, anchors, nc, img)
    assert torch.equal(p.cpu(), p_ref)
    torch.testing.assert_close(io.cpu(), io_ref, rtol=2e-6, atol=1e-5)
    assert layer.stride == img / max(nx, ny)
    if nc == 1:
        assert torch.all(io[..., 5] == 1)


def test_decode_wide_logits_and_threshold_straddlers():
    """the decode beyond randn * 2.  (a) logits uniform in [-30, 30] (w / h up to e^30 anchors) hold the same
    rtol 2e-6; (b) logits placed 0, +-1, +-2, +-8 fp32 steps around logit(t) for the thresholds an NMS call uses (0.001 ... 0.9):
    the decoded sigmoid is within 2 ulp of torch's, so an objectness that the reference puts a few ulp on one side of a threshold
    cannot land far on the other side (the kept-index comparison of the fp32 mode rests on this kernel)."""
    from pytorch_yolo_amd.models.yolo_layer import YOLOLayer
    from oracle.blocks import yolo_decode
    nc, ny, nx, img = 80, 13, 13, 416
    anchors = C.TINY_ANCHORS[1]
    g = torch.Generator().manual_seed(77)
    p_raw = (torch.rand(2, 3 * (5 + nc), ny, nx, generator=g) - 0.5) * 60.0
    layer = YOLOLayer(anchors, nc, C.TINY_ANCHORS).eval()
    io, p = layer(p_raw.to(DEV), img)
    io_ref, p_ref = yolo_decode(p_raw, anchors, nc, img)
    assert torch.equal(p.cpu(), p_ref)
    torch.testing.assert_close(io.cpu(), io_ref, rtol=2e-6, atol=1e-5)
    thr = torch.tensor([0.001, 0.01, 0.1, 0.25, 0.5, 0.75, 0.9], dtype=torch.float64)
    base = torch.log(thr / (1 - thr)).float()
    steps = torch.tensor([-8, -2, -1, 0, 1, 2, 8], dtype=torch.int32)
    vals = (base.view(-1, 1).view(torch.int32) + steps.view(1, -1)).view(torch.float32).reshape(-1)     # neighbouring floats
    p_raw = vals[torch.randint(0, vals.numel(), (2, 3 * (5 + nc), ny, nx), generator=g)]
    io, _ = layer(p_raw.to(DEV), img)
    io_ref, _ = yolo_decode(p_raw, anchors, nc, img)
    sc, sc_ref = io.cpu()[..., 4:], io_ref[..., 4:]
    ulps = ((sc.view(torch.int32) - sc_ref.view(torch.int32)).abs()).max().item()
    print(f"[decode] sigmoid on threshold straddlers: max distance to torch {ulps} ulp")
    assert ulps <= 2


def test_fused_head_decode_wide_logits():
    """The head conv's decode epilogue (hardware exp / reciprocal, common.h yolo_decode_elem<false>) on logits up to |r| ~ 30,
    against the reference formulas evaluated in float64 ON ITS OWN raw logits p (so only the decode is measured): w / h relative
    error <= (3 + |r|) * 2^-23, sigmoid / xy absolute error <= 2^-22 (* stride for xy)."""
    from pytorch_yolo_amd import kernels as K
    from pytorch_yolo_amd._lib import ACT_NONE, DT_F32
    n, h, w, cin, nc = 2, 13, 13, 64, 80
    na, no = 3, nc + 5
    cout = na * no
    anchors = [(10., 13.), (33., 23.), (59., 119.)]
    stride = 32.0
    g = torch.Generator().manual_seed(5)
    x = torch.randn(n, cin, h, w, generator=g)
    wt = torch.randn(cout, cin, 1, 1, generator=g) * (6.0 / cin ** 0.5)           # logit std ~6 ...
    bias = (torch.rand(cout, generator=g) - 0.5) * 30.0                           # ... around a bias in [-15, 15]
    xin = _nhwc(x).to(torch.bfloat16).to(DEV).contiguous()
    wp, bp, kpad, cout_pad = K.pack_conv_weight(wt, bias, cin)
    d = K.conv_desc(n=n, h=h, w=w, cin=cin, in_c_total=cin, in_c_offset=0, cout=cout, out_c_total=K.roundup(cout, 8),
                    out_c_offset=0, ksize=1, stride=1, act=ACT_NONE, kpad=kpad, cout_pad=cout_pad, out_dtype=DT_F32)
    io = torch.empty((n, na * h * w, no), device=DEV)
    p = torch.empty((n, na, h, w, no), device=DEV)
    K.head_decode(xin, wp.to(DEV), bp.to(DEV), d, anchors, nc, stride, io, 0, p)
    torch.cuda.synchronize()
    r = p.cpu().double()
    assert float(r.abs().max()) > 25.0, "the case does not reach wide logits"
    got = io.cpu().double().view(n, na, h, w, no)
    yv, xv = torch.meshgrid(torch.arange(h), torch.arange(w), indexing="ij")
    grid = torch.stack((xv, yv), 2).double().view(1, 1, h, w, 2)
    anc = (torch.tensor(anchors, dtype=torch.float32) / stride).double().view(1, na, 1, 1, 2)
    xy = (torch.sigmoid(r[..., :2]) + grid) * stride
    wh = torch.exp(r[..., 2:4]) * anc * stride
    sg = torch.sigmoid(r[..., 4:])
    assert float((got[..., :2] - xy).abs().max()) <= stride * 2.0 ** -20          # (sigmoid + cell index <= 13: half an ulp of the sum is 2^-21)
    assert float((got[..., 4:] - sg).abs().max()) <= 2.0 ** -22
    rel = (got[..., 2:4] - wh).abs() / wh
    bound = (3.0 + r[..., 2:4].abs()) * 2.0 ** -23
    print(f"[head decode] wide logits: |r| max {float(r.abs().max()):.1f}, w/h rel err max {float(rel.max()):.2e} "
          f"(bound at that r {float(bound[rel == rel.max()].max()):.2e})")
    assert bool((rel <= bound).all())


# ------------------------------------------------------------------------------------------------
def _run_nms(pred_np, conf, iou, inplace=False):
    from pytorch_yolo_amd.utils.utils import non_max_suppression
    pred = torch.from_numpy(pred_np.copy()).to(DEV)
    dets, idx = non_max_suppression(pred, conf, iou, inplace_conf=inplace, with_indices=True)
    to_np = lambda t: None if t is None else t.cpu().numpy()
    return [to_np(d) for d in dets], [to_np(i) for i in idx], pred.cpu().numpy()


@pytest.mark.parametrize("name", list(C.NMS_CASES))
def test_nms_vs_oracle_and_golden(name):
    from oracle import nms as onms
    pred, conf, iou = C.nms_case_inputs(name)
    dets, idx, after = _run_nms(pred, conf, iou, inplace=True)
    work = pred.copy()
    odets, okept = onms.non_max_suppression(work, conf, iou)
    g = load_golden(name)
    for b in range(pred.shape[0]):
        assert np.array_equal(after[b, :, 4], work[b, :, 4], equal_nan=True)      # in-place conf like utils.py:213
        if odets[b] is None:
            assert dets[b] is None
            continue
        assert np.array_equal(idx[b], okept[b]), "kept-index set differs from the oracle"
        assert np.array_equal(dets[b], odets[b]), "detections differ from the oracle (bit-exact expected)"
        assert np.array_equal(idx[b], g[f"kept_{b}"]) and np.array_equal(dets[b][:, 4:], g[f"dets_{b}"][:, 4:])
        np.testing.assert_allclose(dets[b][:, :4], g[f"dets_{b}"][:, :4], rtol=2e-6, atol=2e-4)


@pytest.mark.parametrize("seed", range(12))
def test_nms_random_sweep_vs_oracle(seed):
    """Random batch sizes, row counts, class counts (incl. 1, 2, 3, 5, 7: the class scan splits them over four lanes
    unevenly) and thresholds: kept indices and all seven columns bit-equal to the oracle; ties in the class maximum are
    forced on some rows (first maximum must win, torch.max semantics of utils.py:212)."""
    from oracle import nms as onms
    rng = np.random.default_rng(1000 + seed)
    bs = int(rng.integers(1, 4))
    rows = int(rng.choice([17, 64, 65, 300, 1000]))
    nc = int(rng.choice([1, 2, 3, 5, 7, 20, 80]))
    conf, iou = float(rng.choice([0.05, 0.1, 0.3])), float(rng.choice([0.3, 0.5, 0.7]))
    pred = C.synth_predictions(2000 + seed, bs, rows, nc)
    if nc > 1:                                   # exact ties between two classes on a tenth of the rows
        tie = rng.random((bs, rows)) < 0.1
        a, b2 = rng.integers(0, nc, 2)
        top = pred[..., 5:].max(-1)
        for cls in (int(a), int(b2)):
            pred[..., 5 + cls] = np.where(tie, top, pred[..., 5 + cls])
    dets, idx, _ = _run_nms(pred, conf, iou)
    odets, okept = onms.non_max_suppression(pred.copy(), conf, iou)
    for b in range(bs):
        if odets[b] is None:
            assert dets[b] is None
            continue
        assert np.array_equal(idx[b], okept[b]) and np.array_equal(dets[b], odets[b])


def test_nms_kat_and_non_mutating_default():
    kat = (C.NMS_KAT_ARGS['conf_thres'], C.NMS_KAT_ARGS['nms_thres'])
    dets, idx, after = _run_nms(C.NMS_KAT_ROWS[None], *kat)
    assert np.allclose(dets[0], C.NMS_KAT_EXPECT, atol=1e-4) and idx[0].tolist() == [0, 2]
    assert np.array_equal(after[0], C.NMS_KAT_ROWS)                                # default leaves the input alone
    _, _, after = _run_nms(C.NMS_KAT_ROWS[None], *kat, inplace=True)
    assert np.allclose(after[0, :, 4], C.NMS_KAT_COL4, atol=1e-6)


def test_scale_coords_bit_exact():
    """yolo_scale_coords vs the reference golden (utils.py:296-303) and the batched form with rounding."""
    from oracle import nms as onms
    from pytorch_yolo_amd.utils.utils import scale_coords, scale_detections
    g = load_golden("kat")
    boxes = C.scale_coords_boxes()
    for i, (s1, s0) in enumerate(C.SCALE_CASES):
        t = torch.from_numpy(np.concatenate([boxes, np.ones((len(boxes), 3), np.float32)], 1)).to(DEV)
        out = scale_coords(s1, t, s0).cpu().numpy()
        assert np.array_equal(out[:, :4], g[f"scale_{i}"]) and np.all(out[:, 4:] == 1)
    dets = torch.from_numpy(np.stack([np.concatenate([boxes, np.zeros((len(boxes), 3), np.float32)], 1)] * 2)).to(DEV)
    cnt = torch.tensor([50, 200], dtype=torch.int32, device=DEV)
    scale_detections(dets, cnt, (640, 640), [(1080, 1920), (333, 500)], round_result=True)
    d = dets.cpu().numpy()
    assert np.array_equal(d[0, :50, :4], np.round(onms.scale_coords((640, 640), boxes[:50], (1080, 1920))))
    assert np.array_equal(d[0, 50:, :4], boxes[50:])                               # rows beyond the count untouched
    assert np.array_equal(d[1, :, :4], np.round(onms.scale_coords((640, 640), boxes, (333, 500))))


def test_nms_many_survivors_global_sort_path():
    """> 8192 survivors in one image: the keys are sorted in the global workspace instead of LDS."""
    from oracle import nms as onms
    pred = C.synth_predictions(77, 1, 12000, 4)
    pred[0, :, 4] = np.maximum(pred[0, :, 4], np.float32(0.5))
    dets, idx, _ = _run_nms(pred, 0.001, 0.5)
    odets, okept = onms.non_max_suppression(pred.copy(), 0.001, 0.5)
    assert np.array_equal(idx[0], okept[0]) and np.array_equal(dets[0], odets[0])


# ------------------------------------------------------------------------------------------------
def _model_errors(io, io_ref):
    io, io_ref = io.double(), io_ref.double()
    box = (io[..., :4] - io_ref[..., :4]).abs()
    box_rel = box / io_ref[..., :4].abs().clamp_min(1.0)
    score = (io[..., 4:] - io_ref[..., 4:]).abs()
    return box.max().item(), box_rel.max().item(), score.max().item()


def _assert_model_close(io, io_ref, tag, score_max=2e-2, score_rms=2e-3, box_rel_tol=0.02):
    box_abs, box_rel, score = _model_errors(io, io_ref)
    rms = (io[..., 4:].double() - io_ref[..., 4:].double()).pow(2).mean().sqrt().item()
    print(f"[{tag}] bf16-vs-fp32: max box abs {box_abs:.4f} px, max box rel {box_rel:.4f}, "
          f"max score abs {score:.5f}, rms score {rms:.6f}")
    ok_box = ((io[..., :4] - io_ref[..., :4]).abs() <= torch.maximum(torch.tensor(1.5), box_rel_tol * io_ref[..., :4].abs())).all()
    assert ok_box, f"{tag}: boxes outside max(1.5 px, {100 * box_rel_tol:.0f} %)"
    assert score <= score_max, f"{tag}: scores differ by {score}"
    assert rms <= score_rms, f"{tag}: rms score error {rms}"


@pytest.mark.parametrize("name", list(C.MODEL_CASES))
def test_model_small_vs_oracle_and_golden(name):
    case = C.MODEL_CASES[name]
    model, sd, x = build_case(case)
    io_ref, p_ref = oracle_forward(case, sd, x)
    model = model.to(DEV)
    with torch.no_grad():
        io, p = model(x.to(DEV))
    assert io.shape == io_ref.shape and [t.shape for t in p] == [t.shape for t in p_ref]
    # Darknet-53 depth (23 bf16 residual units) in front of PLAIN conv heads: the ~0.7 % relative drift of the
    # logits (|logit| up to 9) moves a sigmoid by up to 0.025 — measured; the RMS bound stays at 2e-3
    tol = dict(score_max=4e-2) if case[0] == "yolov3" else {}
    _assert_model_close(io.cpu(), io_ref, name, **tol)
    g = load_golden("model_" + name)
    _assert_model_close(io.cpu(), torch.from_numpy(g["io"]), name + "/golden", **tol)
    # fused weights give the same function
    model.fuse()
    with torch.no_grad():
        io_f, _ = model(x.to(DEV))
    _assert_model_close(io_f.cpu(), torch.from_numpy(g["io_fused"]), name + "/fused", **tol)


def test_mobilenet_variant_vs_oracle():
    """YOLOv3TinyMobile: depthwise kernels + linear bottlenecks.  The encoder oracle is a restatement of the
    published MobileNetV2 (torchvision absent: parity unpinned, oracle/mobilenet.py)."""
    from oracle import models as om
    from pytorch_yolo_amd import YOLOv3TinyMobile
    from pytorch_yolo_amd.utils.synthetic import synth_images, synth_state_dict
    model = YOLOv3TinyMobile(n_class=3).eval()
    sd = synth_state_dict(model.state_dict(), 5, n_class=3)
    model.load_state_dict(sd)
    x = synth_images(2, 96, 128, 3)
    with torch.no_grad():
        io_ref, p_ref = om.tiny_mobile_forward(sd, x, om.TINY_ANCHORS, 3)
        io, p = model.to(DEV)(x.to(DEV))
    assert io.shape == io_ref.shape == (2, 3 * (6 * 8 + 3 * 4), 8)
    _assert_model_close(io.cpu(), io_ref, "mobile_96x128", score_max=3e-2, score_rms=4e-3)


@pytest.mark.parametrize("n,h,w", [(2, 127, 159), (1, 416, 416), (1, 130, 162)])
def test_squeezenet_variant_vs_oracle(n, h, w):
    """YOLOv3TinySqueeze: unpadded stride-2 first conv, ReLU, ceil-mode 3x3/2 max pools (windows
    that hang over the border), Fire modules written straight into their concat slices, two heads on one grid.  The
    encoder oracle is a restatement of the published SqueezeNet 1.1 (torchvision absent: parity unpinned,
    oracle/squeezenet.py)."""
    from oracle import models as om
    from pytorch_yolo_amd import YOLOv3TinySqueeze
    from pytorch_yolo_amd.utils.synthetic import synth_images, synth_state_dict
    model = YOLOv3TinySqueeze(n_class=3).eval()
    sd = synth_state_dict(model.state_dict(), 5, n_class=3)
    model.load_state_dict(sd)
    x = synth_images(n, h, w, 3)
    with torch.no_grad():
        io_ref, p_ref = om.tiny_squeeze_forward(sd, x, om.TINY_ANCHORS, 3)
        io, p = model.to(DEV)(x.to(DEV))
    assert io.shape == io_ref.shape and [tuple(q.shape) for q in p] == [tuple(q.shape) for q in p_ref]
    _assert_model_close(io.cpu(), io_ref, f"squeeze_{h}x{w}", score_max=3e-2, score_rms=4e-3)


def test_channel_shuffle_kernel_exact():
    """yolo_channel_shuffle2_fwd against torch's view / transpose / reshape channel shuffle of cat(a, b), for halves
    that do not fill their 8-channel-aligned slots (58 in 64, 116 in 120) and for views into wider buffers."""
    from pytorch_yolo_amd import engine
    for half, slot in ((58, 64), (116, 120), (232, 232), (8, 8)):
        g = torch.Generator().manual_seed(half)
        x = _bf16r(torch.randn(2, 2 * slot, 9, 7, generator=g))
        x[:, half:slot] = 0
        x[:, slot + half:] = 0                                  # the pad channels of both slots are zero by contract

        def trace(rec, s):
            return rec.shuffle2(rec.slice(s, 0, slot), rec.slice(s, slot, slot), half)
        got = engine.run_standalone(trace, x.to(DEV)).cpu()
        logical = torch.cat([x[:, :half], x[:, slot:slot + half]], 1)
        want = logical.view(2, 2, half, 9, 7).transpose(1, 2).reshape(2, 2 * half, 9, 7)
        assert torch.equal(got[:, :half], want[:, :half]) and torch.equal(got[:, slot:slot + half], want[:, half:])
        assert float(got[:, half:slot].abs().sum()) == 0 and float(got[:, slot + half:].abs().sum()) == 0


@pytest.mark.parametrize("n,h,w", [(2, 96, 128), (1, 416, 416)])
def test_shufflenet_variant_vs_oracle(n, h, w):
    """YOLOv3TinyShuffle: ShuffleNetV2 x1.0 traced in a padded physical channel space (halves of 58 /
    116 / 232 channels in 64- / 128- / 256-channel slots), channel split as views, channel shuffle as one copy kernel, linear
    depthwise convs.  The encoder oracle restates the published network (torchvision absent: parity unpinned,
    oracle/shufflenet.py)."""
    from oracle import models as om
    from pytorch_yolo_amd import YOLOv3TinyShuffle
    from pytorch_yolo_amd.utils.synthetic import synth_images, synth_state_dict
    model = YOLOv3TinyShuffle(n_class=3).eval()
    sd = synth_state_dict(model.state_dict(), 5, n_class=3)
    model.load_state_dict(sd)
    x = synth_images(n, h, w, 3)
    with torch.no_grad():
        io_ref, p_ref = om.tiny_shuffle_forward(sd, x, om.TINY_ANCHORS, 3)
        io, p = model.to(DEV)(x.to(DEV))
    assert io.shape == io_ref.shape and [tuple(q.shape) for q in p] == [tuple(q.shape) for q in p_ref]
    # 16 units without any normalising residual: with the synthetic BN gains the activations grow to |x| ~ 60 and the bf16
    # drift to ~1.2 % rms by the last unit (test_shufflenet_units_track_the_oracle shows it grow smoothly, unit by unit);
    # the heads turn that into up to 0.09 on a sigmoid and up to 16 % on exp(tw) - hence the wider bounds than for the other models
    _assert_model_close(io.cpu(), io_ref, f"shuffle_{h}x{w}", score_max=0.12, score_rms=1.2e-2, box_rel_tol=0.25)


def test_shufflenet_units_track_the_oracle():
    """Every one of the 16 ShuffleNetV2 units against the oracle's unit on the same input image: the physical (padded,
    two-slot) tensor mapped back to logical channels stays within 3 % of the oracle's dynamic range and 2 % relative rms,
    drifting smoothly (a mis-wired channel map or shuffle would be an O(1) error from that unit on)."""
    import torch.nn.functional as F2
    from oracle.shufflenet import _bn, _unit
    from pytorch_yolo_amd import YOLOv3TinyShuffle, engine
    from pytorch_yolo_amd.models.yolov3_tiny_shuffle import _folded
    from pytorch_yolo_amd.utils.synthetic import synth_images, synth_state_dict
    model = YOLOv3TinyShuffle(n_class=3).eval()
    sd = synth_state_dict(model.state_dict(), 5, n_class=3)
    model.load_state_dict(sd)
    enc, x, maps = model.features, synth_images(2, 96, 128, 3), []

    def trace(g, s):
        c1 = enc.sequence1[0]
        t = g.maxpool(g.conv(s, _folded(c1[0], c1[1]), stride=2, act="relu"), 3, 2)
        res, cmap = [t], list(range(24))
        maps.append(cmap)
        for stage in (enc.sequence1[2], enc.sequence1[3], enc.sequence2[0]):
            for unit in stage:
                t, cmap = unit._trace(g, t, cmap)
                res.append(t)
                maps.append(cmap)
        return res
    got = engine.run_standalone(trace, x.to(DEV))
    p = "features.sequence1"
    r = F2.max_pool2d(F2.relu(_bn(sd, p + ".0.1", F2.conv2d(x, sd[p + ".0.0.weight"], None, stride=2, padding=1))), 3, 2, 1)
    refs = [r]
    for name, rep in ((p + ".2", 4), (p + ".3", 8), ("features.sequence2.0", 4)):
        for u in range(rep):
            r = _unit(sd, f"{name}.{u}", r, 2 if u == 0 else 1)
            refs.append(r)
    assert len(got) == len(refs) == 17
    for i, (g_, r_, mp) in enumerate(zip(got, refs, maps)):
        gl = g_.cpu()
        pad = [c for c in range(gl.shape[1]) if c not in set(mp)]
        assert float(gl[:, pad].abs().sum()) == 0, f"unit {i}: pad channels are not zero"
        err = (gl[:, mp] - r_).abs()
        assert float(err.max()) <= 0.03 * float(r_.abs().max()), (i, float(err.max()), float(r_.abs().max()))
        assert float(err.pow(2).mean().sqrt() / r_.pow(2).mean().sqrt()) <= 0.02, i


def test_mobilenet_fused_blocks_match_three_launch_path(monkeypatch):
    """Whole YOLOv3TinyMobile at 416x416, 8 images: the plan with all seventeen inverted-residual blocks fused
    (yolo_mbconv_fwd, thousands of tiles per launch: persistent loops, 4 / 2 / 1 workgroups per CU; the ten wide ones in the
    chunk-streaming form) against the plan
    that runs every block as expand conv + depthwise conv + projection conv.  Same rounding points, so the decoded
    boxes and scores agree to bf16 summation-order noise."""
    from pytorch_yolo_amd import YOLOv3TinyMobile
    from pytorch_yolo_amd._lib import OP_MBCONV
    from pytorch_yolo_amd.utils.synthetic import synth_images, synth_state_dict
    x = synth_images(8, 416, 416, 11).to(DEV)
    outs, n_fused = [], []
    for fuse in ("1", "0"):
        monkeypatch.setenv("YOLO_FUSE_MBCONV", fuse)
        model = YOLOv3TinyMobile(n_class=80).eval()
        model.load_state_dict(synth_state_dict(model.state_dict(), 1234, n_class=80))
        model = model.to(DEV)
        with torch.no_grad():
            io, _ = model(x)
        plan = model.plan_for(x)
        plan = plan.subs[0] if hasattr(plan, "subs") else plan          # (sub-batch streams: every sub-plan has the same list)
        n_fused.append(sum(1 for i in range(plan.n_ops) if plan.op_array[i].kind == OP_MBCONV))
        outs.append(io.float().cpu())
    assert n_fused == [17, 0]
    a, b = outs
    assert torch.isfinite(a).all() and a.shape == (8, 3 * (26 * 26 + 13 * 13), 85)
    xy_err = (a[..., :2] - b[..., :2]).abs().max().item()                       # pixels
    wh_err = ((a[..., 2:4] - b[..., 2:4]).abs() / b[..., 2:4].clamp_min(1.0)).max().item()   # relative: w, h = exp(t) * anchor
    score_err = (a[..., 4:] - b[..., 4:]).abs().max().item()
    assert xy_err < 0.5 and wh_err < 2e-2 and score_err < 3e-2, (xy_err, wh_err, score_err)
    assert (a[..., 4:] - b[..., 4:]).pow(2).mean().sqrt().item() < 2e-3


def test_dict_from_results_matches_reference_golden():
    """``_dict_from_results`` (reference utils.py:306-327: scale_coords(...).round() + per-image dict rows) against the
    output of the reference itself on the same detections (tests/golden/dict_from_results.json), and ``predict_dataset``
    (the loop of the reference's test_model) producing the same structure from a model."""
    import json
    import os
    from pytorch_yolo_amd import YOLOv3Tiny
    from pytorch_yolo_amd.utils.synthetic import synth_images, synth_state_dict
    from pytorch_yolo_amd.utils.utils import _dict_from_results, predict_dataset
    dets, paths, shapes, cur = C.results_case()
    targets = [None if d is None else torch.from_numpy(d.copy()).to(DEV) for d in dets]
    got = _dict_from_results({}, targets, paths, shapes, cur)
    want = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "dict_from_results.json")))
    assert set(got) == set(want) == {"a.jpg"} and len(got["a.jpg"]) == len(want["a.jpg"]) == 19
    for g, w in zip(got["a.jpg"], want["a.jpg"]):
        assert {k: g[k] for k in ("type", "left", "top", "right", "bottom")} == {k: w[k] for k in ("type", "left", "top", "right", "bottom")}
        assert abs(g["score"] - w["score"]) < 1e-7
    model = YOLOv3Tiny(n_class=3, kernels_divider=4).eval()
    model.load_state_dict(synth_state_dict(model.state_dict(), 5, n_class=3))
    model = model.to(DEV)
    x = synth_images(3, 96, 128, 3)
    batches = [(x[:2], None, ["p.jpg", "q.jpg"], [(240, 320), (480, 640)]), (x[2:], None, ["p.jpg"], [(96, 128)])]
    data = predict_dataset(model, batches, conf_thresh=1e-4, nms_thresh=0.5)
    direct = model.detect(x.to(DEV), 1e-4, 0.5)
    assert sum(len(v) for v in data.values()) == sum(0 if d is None else len(d) for d in direct) > 0
    assert all(set(r) == {"type", "score", "left", "top", "right", "bottom"} for v in data.values() for r in v)


def test_downsample_sub_is_pre_add():
    from pytorch_yolo_amd.models.yolov3_spp import DownSample
    from pytorch_yolo_amd.utils.synthetic import synth_images, synth_state_dict
    ds = DownSample(8, 16, repeat=1).eval()
    ds.load_state_dict(synth_state_dict(ds.state_dict(), 5))
    from oracle.models import darknet_stage
    x = synth_images(1, 16, 16, 6, channels=8)
    sd = {"s." + k: v for k, v in ds.state_dict().items()}
    x_ref, sub_ref = darknet_stage(sd, "s", x, 2)
    x_out, sub = ds(x.to(DEV))
    torch.testing.assert_close(x_out.cpu(), x_ref, rtol=3e-2, atol=3e-2)
    torch.testing.assert_close(sub.cpu(), sub_ref, rtol=3e-2, atol=3e-2)
    assert not torch.allclose(x_out, sub)


@pytest.mark.parametrize("name", list(C.FULL_CASES))
def test_model_full_size(name):
    """BASELINE.json configs at full size: golden samples + checksums + end-to-end detect()."""
    case = C.FULL_CASES[name]
    model, sd, x = build_case(case)
    g = load_golden("full_" + name)
    model = model.to(DEV)
    with torch.no_grad():
        io, p = model(x.to(DEV))
    assert list(io.shape) == g["io_shape"].tolist()
    ref_rows = torch.from_numpy(g["io_rows"])
    # YOLOv3-SPP's heads are ConvBlocks (BN + LeakyReLU after the head conv, yolov3_spp.py:86,99,111): to
    # score low they need pre-activation logits around -40 with a BN gain of 10..25, which multiplies the
    # ~1 % bf16 drift of a 23-unit bf16 residual stream.  Full-size SPP therefore gets a wider max-score
    # bound (still tight in RMS); the plain-conv tiny heads keep the 2e-2 bound.
    # (round 3: 0.25 -> measured + 25 %: image 0 of SPP-640 shows 0.098 on the sampled rows and 0.138 over all rows)
    wide = dict(score_max=0.18, score_rms=1e-2) if name == "spp_640" else {}
    _assert_model_close(io.cpu()[:, g["rows"]], ref_rows, name + "/rows", **wide)
    colsum = io.double().sum(1).cpu().numpy()
    rel = np.abs(colsum - g["io_colsum"]) / np.maximum(np.abs(g["io_colsum"]), 1.0)
    print(f"[{name}] column checksum: max rel diff {rel.max():.5f}")
    assert rel.max() < 2e-2
    for k, t in enumerate(p):
        assert list(t.shape) == g[f"p{k}_shape"].tolist()
        psum = t.double().sum().item()
        assert abs(psum - float(g[f"p{k}_sum"])) <= 2e-2 * abs(float(g[f"p{k}_sum"])) + 1.0
    # NMS on the HIP forward's own output equals the oracle NMS on the same tensor, bit for bit
    from oracle import nms as onms
    from pytorch_yolo_amd.utils.utils import non_max_suppression
    dets, idx = non_max_suppression(io, with_indices=True, **C.NMS_FULL)
    odets, okept = onms.non_max_suppression(io.cpu().numpy().copy(), **C.NMS_FULL)
    for b in range(io.shape[0]):
        if odets[b] is None:
            assert dets[b] is None
            continue
        assert np.array_equal(idx[b].cpu().numpy(), okept[b]) and np.array_equal(dets[b].cpu().numpy(), odets[b])
    n_ref = int(g["nms_count_0"])
    n_got = 0 if dets[0] is None else len(dets[0])
    print(f"[{name}] detections: HIP bf16 path {n_got}, fp32 reference {n_ref}")
    # detect() == the two-line composition (utils.py:374-378)
    with torch.no_grad():
        d2 = model.detect(x.to(DEV), **C.NMS_FULL)
    assert (d2[0] is None) == (dets[0] is None) and (d2[0] is None or torch.equal(d2[0], dets[0]))


@pytest.mark.parametrize("family,bs,h,w", [("spp", 2, 416, 416), ("spp", 1, 320, 512), ("tiny", 3, 320, 416), ("spp", 16, 160, 160),
                                           ("spp", 2, 608, 608), ("spp", 1, 608, 320)])
def test_full_width_models_at_other_input_sizes(family, bs, h, w):
    """Full-width models (kernels_divider 1, nc 80) at input sizes other than the BASELINE ones — other map sizes pick
    other tile configurations / fusions (e.g. 52x52 and 26x26 maps, rectangular inputs, the stem with partial tiles,
    sub-batches of 8 on two streams; 608: maps of 152 / 76 / 38 / 19 pixels, which the 20x20-tile kernel takes with partial
    tiles on two edges) — against the fp32 oracle on the same seeded weights and images."""
    kw = dict(n_class=80, kernels_divider=1, anchors=C.SPP_ANCHORS if family == "spp" else C.TINY_ANCHORS)
    case = (family, kw, bs, h, w, 41, 42)
    model, sd, x = build_case(case)
    io_ref, p_ref = oracle_forward(case, sd, x)
    model = model.to(DEV)
    with torch.no_grad():
        io, p = model(x.to(DEV))
    assert io.shape == io_ref.shape and [t.shape for t in p] == [t.shape for t in p_ref]
    wide = dict(score_max=0.25, score_rms=1e-2) if family == "spp" else {}      # (measured: 0.228 over the 16 images at 160x160)
    _assert_model_close(io.cpu(), io_ref, f"{family}_{bs}x{h}x{w}", **wide)


def test_forward_idempotent_and_graph_replay():
    case = C.MODEL_CASES["tiny_small"]
    model, sd, x = build_case(case)
    model = model.to(DEV)
    with torch.no_grad():
        io1, p1 = model(x.to(DEV))
        io2, _ = model(x.to(DEV))
        model.use_hip_graph = True
        io3, p3 = model(x.to(DEV))
        io3 = io3.clone()
        io4, _ = model(x.to(DEV))
    assert torch.equal(io1, io2) and torch.equal(io1, io3) and torch.equal(io1, io4)
    assert all(torch.equal(a, b) for a, b in zip(p1, p3))


def test_graph_replay_at_the_tiny416_bench_shape():
    """``use_hip_graph`` on the configuration whose per-pipeline graph experiment ended in a GPU memory access fault in round 3
    (YOLOv3-tiny 416x416, 32 images, two sub-batch streams captured into one HIP graph): replay == eager, bit for bit, twice, and
    with another input.  (The launch lists of this shape passed the red-zone audit - test_launch_lists_stay_inside_their_buffers -
    and the static one; the experiment's own capture code was never committed.)"""
    case = C.FULL_CASES["tiny_416"]
    model, sd, _ = build_case(case)
    model = model.to(DEV)
    xa, xb = _seeded_batch(32, 416).to(DEV), _seeded_batch(32, 416, first_seed=200).to(DEV)
    with torch.no_grad():
        assert type(model.plan_for(xa)).__name__ == "StreamedPlan"
        io_a, p_a = model(xa)
        io_b, _ = model(xb)
        model.use_hip_graph = True
        g1 = model(xa)[0].clone()
        g2 = model(xb)[0].clone()
        g3 = model(xa)
        torch.cuda.synchronize()
    assert torch.equal(g1, io_a) and torch.equal(g2, io_b) and torch.equal(g3[0], io_a)
    assert all(torch.equal(a, b) for a, b in zip(g3[1], p_a))


def test_concurrent_streams_give_identical_results():
    """engine.StreamedPlan: sub-batches on 2 HIP streams == one launch list, bit for bit."""
    from pytorch_yolo_amd.utils.synthetic import synth_images
    case = C.MODEL_CASES["spp_small"]
    model, sd, _ = build_case(case)
    x = synth_images(8, 64, 64, 3).to(DEV)
    model = model.to(DEV)
    with torch.no_grad():
        model.n_streams = 1
        io1, p1 = model(x)
        model.n_streams = 2
        io2, p2 = model(x)
        assert type(model.plan_for(x)).__name__ == "StreamedPlan"
        model.use_hip_graph = True
        io3, _ = model(x)
        io3 = io3.clone()
        io4, _ = model(x)
    assert torch.equal(io1, io2) and all(torch.equal(a, b) for a, b in zip(p1, p2))
    assert torch.equal(io1, io3) and torch.equal(io1, io4)


def test_cpu_input_fails_loudly():
    case = C.MODEL_CASES["tiny_small"]
    model, sd, x = build_case(case)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        model(x)
    model.train()
    with pytest.raises(NotImplementedError):
        model(x.to(DEV))


def test_pipelined_gather_matches_joined_gather():
    """distributed.PipelinedGather (side-stream all-gather of free-running sub-batch pipelines) returns what the
    joined gather_detections returns, batch after batch, and detect results stay those of a joined run.
    One rank over RCCL here (the GPU box has one card); the world_size-2 exchange itself is covered on gloo."""
    import torch.distributed as dist
    from pytorch_yolo_amd.distributed import PipelinedGather, gather_detections
    from pytorch_yolo_amd.utils.synthetic import synth_images
    from pytorch_yolo_amd.utils.utils import nms_capacity
    created = not dist.is_initialized()
    if created:
        dist.init_process_group("nccl", init_method="tcp://127.0.0.1:29533", rank=0, world_size=1,
                                device_id=torch.device(DEV))
    try:
        case = C.MODEL_CASES["spp_small"]
        model, sd, _ = build_case(case)
        model = model.to(DEV)
        model.n_streams = 2
        bs = 8
        xs = [synth_images(bs, 64, 64, 10 + k).to(DEV) for k in range(4)]
        plan = model.plan_for(xs[0])
        assert type(plan).__name__ == "StreamedPlan"
        cap = nms_capacity(plan.rows_total, model.n_class)
        mk = lambda: (torch.zeros((bs, cap, 7), device=DEV), torch.zeros((bs, cap), dtype=torch.int32, device=DEV),
                      torch.zeros((bs,), dtype=torch.int32, device=DEV))
        io, ps = plan.new_outputs()
        # joined reference, one batch at a time
        want = []
        for x in xs:
            out = mk()
            plan.launch_detect(x, io, ps, out, 1e-4, 0.5, join=True)
            torch.cuda.synchronize()
            d, c = gather_detections(out[0], out[2])
            torch.cuda.synchronize()
            want.append((d.clone(), c.clone()))
        # free-running pipelines + side-stream exchange, no sync until the end
        g = PipelinedGather(bs, cap, plan.n_streams, torch.device(DEV))
        out = mk()
        got = []
        for x in xs:
            plan.launch_detect(x, io, ps, out, 1e-4, 0.5, join=False, after_nms=g.begin(out))
            d, c, done = g.exchange()
            done.synchronize()                # results of this batch are complete here (slots alternate)
            got.append((d.clone(), c.clone()))
        torch.cuda.synchronize()
        assert sum(int(c.sum()) for _, c in want) > 0
        for (dw, cw), (dg, cg) in zip(want, got):
            assert torch.equal(cw, cg)
            for b, n in enumerate(cw.tolist()):
                assert torch.equal(dw[b, :n], dg[b, :n])
    finally:
        if created:
            dist.destroy_process_group()


@pytest.mark.parametrize("h,w,c,new_shape", [(60, 100, 3, 64), (100, 60, 3, 64), (97, 131, 3, 96), (48, 48, 1, 96),
                                             (90, 120, 3, (64, 64)), (33, 200, 4, 128)])
def test_letterbox_vs_oracle(h, w, c, new_shape):
    """yolo_letterbox_u8_fwd (LetterBox, reference utils/augs.py:7-94): the uint8 letterboxed image equals the
    oracle's bit for bit (same float32 operation order), for non-integer / integer down-scaling, up-scaling,
    1..4 channels and fixed-shape targets."""
    from oracle import preprocess as O
    from pytorch_yolo_amd.utils.augs import LetterBox
    rng = np.random.default_rng(h * 1000 + w)
    img = rng.integers(0, 256, (h, w, c), dtype=np.uint8)
    want, p = O.letterbox(img, new_shape)
    t = LetterBox(new_shape)
    got = t(image=torch.from_numpy(img).to(DEV))["image"].cpu().numpy()
    assert t.params == p and got.shape == want.shape
    assert np.array_equal(got, want)


def test_preprocess_batch_vs_oracle():
    """LetterBox + /255 + HWC->CHW + equalize_shapes fused (one launch per image) == the oracle's composition of
    the reference steps, exactly; then the batch goes through detect() like any other input."""
    from oracle import preprocess as O
    from pytorch_yolo_amd.utils.augs import preprocess_batch
    rng = np.random.default_rng(11)
    imgs = [rng.integers(0, 256, s, dtype=np.uint8) for s in ((60, 100, 3), (100, 60, 3), (64, 64, 3), (50, 111, 3))]
    want, wmeta = O.preprocess_batch(imgs, 64)
    got, meta = preprocess_batch([torch.from_numpy(i).to(DEV) for i in imgs], 64)
    assert tuple(got.shape) == want.shape and meta == wmeta
    assert np.array_equal(got.cpu().numpy(), want)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        preprocess_batch([torch.from_numpy(imgs[0])], 64)
    case = C.MODEL_CASES["tiny_small"]
    model, sd, _ = build_case(case)
    model = model.to(DEV)
    with torch.no_grad():
        io, _ = model(got)
    assert io.shape[0] == len(imgs) and bool(torch.isfinite(io).all())


# ------------------------------------------------------------------------------------------------
# fp32 reference-precision mode (csrc/conv_f32.hip) and the end-to-end detection-set checks
F32_CONV_CASES = [
    # n, h, w, cin, cout, k, stride, act, residual, aux, upsample
    (2, 16, 16, 8, 32, 3, 1, "leaky", False, False, False),       # first-layer shape (3 -> 8 padded channels)
    (2, 20, 20, 64, 128, 3, 1, "leaky", True, True, False),       # residual + pre-add copy
    (1, 33, 29, 32, 64, 3, 2, "leaky", False, False, False),      # odd size, stride 2
    (3, 13, 13, 256, 255, 1, 1, "none", False, False, False),     # detection head: 255 couts
    (2, 10, 10, 128, 64, 1, 1, "leaky", False, False, True),      # 2x nearest upsample on store
    (1, 9, 9, 1024, 512, 1, 1, "leaky", False, False, False),     # long K
    (1, 40, 24, 20, 36, 3, 1, "relu6", False, False, False),      # channel counts that are only multiples of 4
]


@pytest.mark.parametrize("case", F32_CONV_CASES, ids=lambda c: "n%d_%dx%d_c%d-%d_k%d_s%d_%s_r%d_a%d_u%d" % tuple(int(v) if not isinstance(v, str) else v for v in c))
def test_conv_f32_kernel(case):
    """yolo_conv2d_f32_fwd against torch's fp32 conv on the SAME fp32 operands: only the summation order differs."""
    from pytorch_yolo_amd import kernels as K
    from pytorch_yolo_amd._lib import ACT_LEAKY01, ACT_NONE, ACT_RELU6, DT_F32
    n, h, w, cin, cout, k, stride, act, use_res, use_aux, up = case
    g = torch.Generator().manual_seed(hash(case) & 0xffff)
    x = torch.randn(n, cin, h, w, generator=g)
    wt = torch.randn(cout, cin, k, k, generator=g) * (2.0 / (cin * k * k)) ** 0.5
    bias = torch.randn(cout, generator=g) * 0.1
    pad = (k - 1) // 2
    ho, wo = (h + 2 * pad - k) // stride + 1, (w + 2 * pad - k) // stride + 1
    res = torch.randn(n, cout, ho, wo, generator=g) if use_res else None
    in_ct, in_co = cin + 8, 4
    xin = torch.zeros(n, h, w, in_ct, dtype=torch.float32, device=DEV)
    xin[..., in_co:in_co + cin] = x.permute(0, 2, 3, 1).to(DEV)
    oh, ow = (2 * ho, 2 * wo) if up else (ho, wo)
    out_ct, out_co = K.roundup(cout, 4) + 8, 4
    y = torch.full((n, oh, ow, out_ct), -77.0, dtype=torch.float32, device=DEV)
    aux = torch.full((n, ho, wo, K.roundup(cout, 4) + 4), -77.0, dtype=torch.float32, device=DEV) if use_aux else None
    rin = res.permute(0, 2, 3, 1).contiguous().to(DEV) if use_res else None
    wp, bp, kpad, cout_pad = K.pack_conv_weight_f32(wt, bias, cin)
    d = K.conv_desc(n=n, h=h, w=w, cin=cin, in_c_total=in_ct, in_c_offset=in_co, cout=cout, out_c_total=out_ct,
                    out_c_offset=out_co, ksize=k, stride=stride,
                    act={"leaky": ACT_LEAKY01, "none": ACT_NONE, "relu6": ACT_RELU6}[act], kpad=kpad, cout_pad=cout_pad,
                    upsample2x=int(up), out_dtype=DT_F32, res=(cout, 0) if use_res else (0, 0),
                    aux=(aux.shape[-1], 4) if use_aux else (0, 0))
    K.conv2d_f32(xin, wp.to(DEV), bp.to(DEV), y, d, residual=rin, y_preadd=aux)
    torch.cuda.synchronize()
    ref = F.conv2d(x, wt, bias, stride=stride, padding=pad)
    ref = {"leaky": lambda t: F.leaky_relu(t, 0.1), "none": lambda t: t, "relu6": F.relu6}[act](ref)
    pre = ref
    if use_res:
        ref = ref + res
    if up:
        ref = F.interpolate(ref, scale_factor=2, mode="nearest")
    got = _nchw(y[..., out_co:out_co + cout])
    torch.testing.assert_close(got, ref, rtol=2e-5, atol=2e-5)
    assert torch.all(y[..., :out_co] == -77.0) and torch.all(y[..., out_co + cout:] == -77.0)
    if use_aux:
        torch.testing.assert_close(_nchw(aux[..., 4:4 + cout]), pre, rtol=2e-5, atol=2e-5)
        assert torch.all(aux[..., :4] == -77.0)


def _assert_fp32_close(io, io_ref, tag, box_atol=2e-3, box_rtol=2e-5, score_atol=2e-5):
    """fp32 mode vs the fp32 reference: only BN folding and summation order differ."""
    io, io_ref = io.double(), io_ref.double()
    box = (io[..., :4] - io_ref[..., :4]).abs()
    score = (io[..., 4:] - io_ref[..., 4:]).abs()
    print(f"[{tag}] fp32 mode vs fp32 reference: max box abs {box.max().item():.6f} px, max score abs {score.max().item():.2e}")
    assert bool((box <= box_atol + box_rtol * io_ref[..., :4].abs()).all()), f"{tag}: boxes differ by {box.max().item()} px"
    assert score.max().item() <= score_atol, f"{tag}: scores differ by {score.max().item()}"


@pytest.mark.parametrize("name", list(C.MODEL_CASES))
def test_fp32_mode_small_models_vs_reference_golden(name):
    """model.precision = 'fp32': io / p against the REFERENCE's golden tensors at fp32 accuracy."""
    case = C.MODEL_CASES[name]
    model, sd, x = build_case(case)
    model.precision = "fp32"
    model = model.to(DEV)
    with torch.no_grad():
        io, p = model(x.to(DEV))
    g = load_golden("model_" + name)
    _assert_fp32_close(io.cpu(), torch.from_numpy(g["io"]), name + "/fp32")
    for k, t in enumerate(p):
        torch.testing.assert_close(t.cpu(), torch.from_numpy(g[f"p{k}"]), rtol=2e-4, atol=2e-4)
    # and the bf16 plan of the same model is a different plan (the cache is keyed by precision)
    model.precision = "bf16"
    with torch.no_grad():
        io_b, _ = model(x.to(DEV))
    assert not torch.equal(io_b, io)


def _iou_matrix(a, b):
    """pairwise IoU of xyxy boxes a [n,4], b [m,4] (numpy float64)."""
    a, b = a.astype(np.float64), b.astype(np.float64)
    x1 = np.maximum(a[:, None, 0], b[None, :, 0]); y1 = np.maximum(a[:, None, 1], b[None, :, 1])
    x2 = np.minimum(a[:, None, 2], b[None, :, 2]); y2 = np.minimum(a[:, None, 3], b[None, :, 3])
    inter = np.clip(x2 - x1, 0, None) * np.clip(y2 - y1, 0, None)
    aa = (a[:, 2] - a[:, 0]) * (a[:, 3] - a[:, 1]); bb = (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])
    return inter / np.maximum(aa[:, None] + bb[None, :] - inter, 1e-12)


def _unmatched(a, b, iou_min, conf_tol):
    """rows of a [n,7] without a partner in b [m,7] of the same class with IoU >= iou_min and |dconf| <= conf_tol."""
    if len(a) == 0:
        return []
    if len(b) == 0:
        return list(range(len(a)))
    iou = _iou_matrix(a[:, :4], b[:, :4])
    ok = (iou >= iou_min) & (a[:, None, 6] == b[None, :, 6]) & (np.abs(a[:, None, 4] - b[None, :, 4]) <= conf_tol)
    return [i for i in range(len(a)) if not ok[i].any()]


def _stable_detections(pred_np, conf, iou, eps_conf, eps_iou, eps_cls, sigma_box=0.0, sigma_score=0.0, n_noise=0):
    """SURVEY 7's guard band, made operational.  Returns (all detections, those outside the guard band) of ``pred`` under
    the oracle MERGE-NMS.  A detection is INSIDE the band (excluded from the comparison) when
      * its pivot row does not survive every perturbation of the two thresholds by +-eps (candidates within eps of
        conf_thres; merge groups whose membership hinges on an IoU within eps of nms_thres), or
      * its pivot row's two best class scores are within eps_cls of each other (the arg-max class can flip), or
      * it does not come out (same class, IoU >= 0.9, |dconf| <= 0.03) of every one of ``n_noise`` re-runs on the prediction
        with gaussian noise of the bf16 drift's own size added (sigma_box px on xywh, sigma_score on objectness and class
        scores): MERGE replaces a kept box by the confidence-weighted mean of the whole pile it suppresses
        (utils.py:266-275), and with synthetic weights the piles are 20..100 overlapping candidates whose membership - hence
        the mean, hence which pile survives - moves with perturbations far below the stated tolerance."""
    from oracle import nms as onms
    dets, kept = onms.nms_image(pred_np.copy(), conf, iou)
    if dets is None:
        return np.zeros((0, 7), np.float32), np.zeros((0, 7), np.float32)
    stable = set(kept.tolist())
    for dc in (-eps_conf, eps_conf):
        for di in (-eps_iou, eps_iou):
            d2, k2 = onms.nms_image(pred_np.copy(), conf + dc, iou + di)
            stable &= set(() if d2 is None else k2.tolist())
    cls = np.sort(pred_np[kept, 5:], axis=1)
    gap_ok = (cls[:, -1] - cls[:, -2] > eps_cls) if cls.shape[1] > 1 else np.ones(len(kept), bool)
    mask = np.array([k in stable for k in kept.tolist()], bool) & gap_ok
    rng = np.random.default_rng(20261004)
    for _ in range(n_noise):
        noisy = pred_np.copy()
        noisy[:, :4] += rng.normal(0.0, sigma_box, noisy[:, :4].shape).astype(np.float32)
        noisy[:, 4:] = np.clip(noisy[:, 4:] + rng.normal(0.0, sigma_score, noisy[:, 4:].shape).astype(np.float32), 0.0, 1.0)
        d2, _ = onms.nms_image(noisy, conf, iou)
        d2 = np.zeros((0, 7), np.float32) if d2 is None else d2
        lost = set(_unmatched(dets, d2, iou_min=0.9, conf_tol=0.03))
        mask &= np.array([i not in lost for i in range(len(dets))], bool)
    return dets, dets[mask]


GUARD = dict(eps_conf=0.03, eps_iou=0.05, eps_cls=0.03)   # guard band around conf_thres / nms_thres / class arg-max
# a partner: same class, IoU >= 0.7, |dconf| <= 0.06.  (0.7, not 0.9: the synthetic weights produce lattices of near-identical
# boxes one grid cell apart - e.g. class 6, 58 px wide, every 8 px - and which cell's pile survives MERGE is decided by score
# differences far below the bf16 drift; one cell's shift of such a box is IoU 0.76.  COCO matching itself uses IoU >= 0.5.)
MATCH = dict(iou_min=0.7, conf_tol=0.06)
N_NOISE = 6                                               # noise re-runs of the guard band (see _stable_detections)
CONFIDENT = 0.5


@pytest.mark.parametrize("name", list(C.FULL_CASES))
def test_full_size_detection_sets_vs_reference(name):
    """End to end on the BASELINE configs at full size, against the reference's own NMS output stored in the golden
    (nms_dets_0 / nms_kept_0, tests/golden/make_golden.py):
      * fp32 mode: the kept-index set IS the reference's, class equal, conf / class_conf within 2e-5, boxes within 1e-2 px;
      * bf16 mode: the detections outside the guard band (_stable_detections: thresholds, class arg-max, and robustness
        to noise of twice the measured bf16 drift) have a partner on the other side (same class, IoU >= 0.7,
        |dconf| <= 0.06) in both directions, at most 5 % excepted; the share of ALL detections with a partner is printed."""
    from oracle import nms as onms
    from pytorch_yolo_amd.utils.utils import non_max_suppression
    case = C.FULL_CASES[name]
    model, sd, x = build_case(case)
    g = load_golden("full_" + name)
    ref_dets, ref_kept = g["nms_dets_0"], g["nms_kept_0"]
    io_ref, _ = oracle_forward(case, sd, x)                     # the reference's arithmetic (bit-equal in the build container; another
    odets, okept = onms.non_max_suppression(io_ref.numpy().copy(), **C.NMS_FULL)   # host CPU sums in another order: 1e-6 relative)
    # ties this oracle run to the golden: same kept set, conf / class to 1e-5; MERGE's weighted box means move visibly when a
    # borderline member flips on the 1e-6 differences between two hosts' fp32 convolutions, so boxes get 98 % / 1 px
    assert np.array_equal(okept[0], ref_kept) and np.allclose(odets[0][:, 4:], ref_dets[:, 4:], rtol=0, atol=1e-5)
    dbox = np.abs(odets[0][:, :4] - ref_dets[:, :4]).max(1)
    # (a flipped member moves a merged box by whole pixels — 3.8 px seen on one host —, so the outliers get a loose bound)
    assert (dbox <= 1e-3).mean() >= 0.97 and dbox.max() <= 8.0, f"oracle-vs-golden boxes: max {dbox.max()}"
    model = model.to(DEV)
    # ---- fp32 mode
    model.precision = "fp32"
    with torch.no_grad():
        io32, _ = model(x.to(DEV))
    # full-size SPP logits have std 25..40: fp32 summation-order noise of 1e-6 relative is up to 4e-5 of a score (3.3e-5 seen)
    _assert_fp32_close(io32.cpu()[:, g["rows"]], torch.from_numpy(g["io_rows"]), name + "/fp32 rows", score_atol=1e-4)
    _assert_fp32_close(io32.cpu(), io_ref, name + "/fp32 all rows", score_atol=1e-4)
    dets32, idx32 = non_max_suppression(io32, with_indices=True, **C.NMS_FULL)
    d32, k32 = dets32[0].cpu().numpy(), idx32[0].cpu().numpy()
    print(f"[{name}] fp32 mode: {len(d32)} detections, reference {len(ref_dets)}")
    assert np.array_equal(k32, ref_kept), "fp32 mode: kept-index set differs from the reference's"
    assert np.array_equal(d32[:, 6], ref_dets[:, 6])
    np.testing.assert_allclose(d32[:, 4:6], ref_dets[:, 4:6], rtol=0, atol=1e-4)
    dbox = np.abs(d32[:, :4] - ref_dets[:, :4]).max(1)
    print(f"[{name}] fp32 mode: merged boxes vs the reference's: max {dbox.max():.5f} px, {100 * (dbox <= 1e-2).mean():.1f} % within 0.01 px")
    assert (dbox <= 1e-2).mean() >= 0.97 and dbox.max() <= 8.0        # (a borderline merge member may flip: see above)
    # ---- bf16 mode with the guard band
    model.precision = "bf16"
    with torch.no_grad():
        io16, _ = model(x.to(DEV))
    dets16, _ = non_max_suppression(io16, with_indices=True, **C.NMS_FULL)
    d16 = dets16[0].cpu().numpy()
    io16_0 = io16[0].cpu().numpy()
    live = io_ref[0].numpy()[:, 4] > 0.5 * C.NMS_FULL["conf_thres"]            # rows that can matter to NMS
    drift = io16_0[live] - io_ref[0].numpy()[live]
    noise = dict(sigma_box=2 * float(np.sqrt((drift[:, :4] ** 2).mean())), sigma_score=2 * float(np.sqrt((drift[:, 4:] ** 2).mean())), n_noise=N_NOISE)
    print(f"[{name}] bf16 drift on live rows: rms box {noise['sigma_box'] / 2:.3f} px, rms score {noise['sigma_score'] / 2:.4f} (the guard band's noise is twice that)")
    ref_all, ref_stable = _stable_detections(io_ref[0].numpy(), C.NMS_FULL["conf_thres"], C.NMS_FULL["nms_thres"], **GUARD, **noise)
    hip_all, hip_stable = _stable_detections(io16_0, C.NMS_FULL["conf_thres"], C.NMS_FULL["nms_thres"], **GUARD, **noise)
    assert np.array_equal(hip_all, d16)                            # (the device NMS is the oracle NMS, bit for bit)
    miss_ref = _unmatched(ref_stable, d16, **MATCH)
    miss_hip = _unmatched(hip_stable, ref_all, **MATCH)
    print(f"[{name}] bf16 mode: {len(d16)} detections ({len(hip_stable)} outside the guard band), reference {len(ref_all)} "
          f"({len(ref_stable)} outside); unmatched: reference {len(miss_ref)}, bf16 {len(miss_hip)}")
    for tag, rows, pool in (("reference", ref_stable[miss_ref], d16), ("bf16", hip_stable[miss_hip], ref_all)):
        for r in rows:
            same = pool[pool[:, 6] == r[6]]
            best = _iou_matrix(r[None, :4], same[:, :4]).max() if len(same) else 0.0
            print(f"   {tag} detection without a partner: cls {int(r[6])} conf {r[4]:.3f} box {np.round(r[:4], 1)} best same-class IoU {best:.2f}")
    share = 1.0 - len(_unmatched(ref_all, d16, **MATCH)) / max(1, len(ref_all))
    print(f"[{name}] share of ALL reference detections with a bf16 partner: {share:.2f}")
    assert len(ref_stable) >= 10 and len(hip_stable) >= 10, "the comparison is vacuous"
    # outside the guard band at most 5 % (at least one allowed) may lack a partner: measured 0..1 of 26..49
    assert len(miss_ref) <= max(1, 0.05 * len(ref_stable)) and len(miss_hip) <= max(1, 0.05 * len(hip_stable)), \
        "detections outside the guard band have no partner"
    assert not any(r[4] >= CONFIDENT for r in ref_all[_unmatched(ref_all, d16, **MATCH)])    # confident ones always pair up
    assert share >= 0.6
    assert abs(len(d16) - len(ref_all)) <= 0.1 * len(ref_all)


def test_separable_detections_strictly_vs_reference():
    """Detection-level bf16 parity on data that can carry it (tests/_cases.py::SEPARABLE, golden
    full_spp_640_separable.npz generated from the reference by tests/golden/make_golden.py): YOLOv3-SPP 640x640 whose head BN was
    calibrated so that the reference's 22 detections are isolated and saturated (conf >= 0.83, none between 0.4 and 0.6, no ties).
    WITHOUT guard band or noise re-runs:
      * bf16 (the benched path): reference and bf16 detections pair up in both directions - same class, IoU >= 0.9,
        |dconf| <= 0.03 - for at least 90 % (measured: 21 of 22 reference detections, 21 of 23 bf16 ones), and at the lattice
        test's criteria (IoU >= 0.7, |dconf| <= 0.06) for at least 95 %; the counts differ by at most one;
      * the same bars against the CPU rounding model (the oracle re-run under the product's bf16 rounding points,
        oracle/policy.py), which on this seed pairs with the reference 22 of 22;
      * fp32 mode: the kept-index set IS the reference's, conf / class to 2e-4, boxes to 0.05 px.
    What the two strict misses are (tools/dbg/sep_gpu.py): one cell next to the objectness cut (fp32 0.35, CPU rounding model
    0.998, HIP 0.998: the calibration's gamma of 1000 turns a 0.007 drift of the normalised conv output into 6.8 logits) and one
    MERGE pile whose pivot changes between two saturated candidates (0.985 vs 1.000), which moves the merged 120-pixel box by
    5 pixels (IoU 0.88).  In this amplified domain the HIP path and the CPU rounding model are two realisations of the same
    rounding noise (median objectness-logit distance of the candidates: rounding model - fp32 0.30, HIP - fp32 0.53, HIP -
    rounding model 0.39).  The lattice fixture's 0.79 (test_full_size_detection_sets_vs_reference) is the data's conditioning:
    over 39 patch seeds the two CPU runs themselves pair between 0.68 and 1.00 (profiles/r03_separable_search.txt); this seed is
    the one at 1.00 for the CPU pair."""
    from oracle import models as om, nms as onms
    from oracle.policy import run_policy
    from pytorch_yolo_amd.utils.utils import non_max_suppression
    sep = C.SEPARABLE
    model, sd, x, g = build_separable_case()
    ref = g["nms_dets_0"]
    model = model.to(DEV)
    with torch.no_grad():
        io16, _ = model(x.to(DEV))
        dets16, _ = non_max_suppression(io16, sep["conf_thres"], sep["nms_thres"], with_indices=True)
    d16 = dets16[0].cpu().numpy()
    a, b = strict_share(ref, d16), strict_share(d16, ref)
    loose = strict_share(ref, d16, 0.7, 0.06), strict_share(d16, ref, 0.7, 0.06)
    print(f"[separable] reference {len(ref)} detections, bf16 {len(d16)}; strict share (IoU 0.9, dconf 0.03) {a:.3f} / {b:.3f}; "
          f"at the lattice test's criteria (IoU 0.7, dconf 0.06) {loose[0]:.3f} / {loose[1]:.3f}")
    assert min(a, b) >= 0.90 and min(loose) >= 0.95 and abs(len(d16) - len(ref)) <= 1
    io_b, _ = run_policy(om.spp_forward, sd, x, C.SPP_ANCHORS, 80, policy="bf16")
    db, _ = onms.non_max_suppression(io_b.numpy().copy(), sep["conf_thres"], sep["nms_thres"])
    am, bm = strict_share(db[0], d16), strict_share(d16, db[0])
    print(f"[separable] against the CPU rounding model ({len(db[0])} detections): {am:.3f} / {bm:.3f}")
    assert min(am, bm) >= 0.90 and min(strict_share(db[0], d16, 0.7, 0.06), strict_share(d16, db[0], 0.7, 0.06)) >= 0.95
    model.precision = "fp32"
    with torch.no_grad():
        io32, _ = model(x.to(DEV))
        dets32, idx32 = non_max_suppression(io32, sep["conf_thres"], sep["nms_thres"], with_indices=True)
    d32, k32 = dets32[0].cpu().numpy(), idx32[0].cpu().numpy()
    assert np.array_equal(k32, g["nms_kept_0"]), "fp32 mode: kept-index set differs from the reference's"
    assert np.array_equal(d32[:, 6], ref[:, 6])
    np.testing.assert_allclose(d32[:, 4:6], ref[:, 4:6], rtol=0, atol=2e-4)
    dbox = np.abs(d32[:, :4] - ref[:, :4]).max(1)
    print(f"[separable] fp32 mode: {len(d32)} detections, kept-index set equal, boxes within {dbox.max():.4f} px of the reference's")
    assert dbox.max() <= 0.05


@pytest.mark.parametrize("seed", C.RULE_SEEDS)
def test_rule_selected_detections_strictly_vs_reference(seed):
    """Detection-level bf16 parity on cases chosen by a RULE that looks at the reference's fp32 outputs only (VERDICT r3 item 6;
    tests/_cases.py::RULE / rule_verdict, goldens full_spp_640_rule_<seed>.npz generated from the reference by
    tests/golden/make_golden.py --only rule; tests/test_oracle_golden.py proves the committed seeds are the FIRST ones that
    qualify).  YOLOv3-SPP 640x640, 120 colour patches, head BN calibrated from the reference's raw heads so that every objectness
    cut lies in a gap of >= 8 bf16 drifts of the normalised conv output.  WITHOUT guard band or noise re-runs:
      * bf16 (the benched path): every reference detection has a bf16 partner and vice versa - same class, IoU >= 0.9,
        |dconf| <= 0.03 (strict share >= 0.95 in both directions), same count;
      * fp32 mode: the kept-index set IS the reference's, conf / class to 2e-4, boxes to 0.05 px."""
    from pytorch_yolo_amd.utils.utils import non_max_suppression
    rule = C.RULE
    model, sd, x, g = build_rule_case(seed)
    ref = g["nms_dets_0"]
    model = model.to(DEV)
    with torch.no_grad():
        io16, _ = model(x.to(DEV))
        dets16, _ = non_max_suppression(io16, rule["conf_thres"], rule["nms_thres"], with_indices=True)
    d16 = dets16[0].cpu().numpy()
    a, b = strict_share(ref, d16), strict_share(d16, ref)
    dconf = max(min(abs(float(r[4]) - float(q[4])) for q in d16 if int(q[6]) == int(r[6])) if any(int(q[6]) == int(r[6]) for q in d16) else 1.0 for r in ref)
    print(f"[rule case {seed}] reference {len(ref)} detections, bf16 {len(d16)}; strict share (IoU 0.9, dconf 0.03) {a:.3f} / {b:.3f}; "
          f"largest conf distance of a reference detection to its class mate {dconf:.4f}")
    assert min(a, b) >= 0.95 and len(d16) == len(ref)
    model.precision = "fp32"
    with torch.no_grad():
        io32, _ = model(x.to(DEV))
        dets32, idx32 = non_max_suppression(io32, rule["conf_thres"], rule["nms_thres"], with_indices=True)
    d32, k32 = dets32[0].cpu().numpy(), idx32[0].cpu().numpy()
    assert np.array_equal(k32, g["nms_kept_0"]), "fp32 mode: kept-index set differs from the reference's"
    assert np.array_equal(d32[:, 6], ref[:, 6])
    np.testing.assert_allclose(d32[:, 4:6], ref[:, 4:6], rtol=0, atol=2e-4)
    assert np.abs(d32[:, :4] - ref[:, :4]).max() <= 0.05


def test_bf16_path_within_its_rounding_budget():
    """SPP-640 at full size: the distance of the HIP bf16 forward to the fp32 reference must be the distance that bf16
    operand rounding alone produces — the oracle re-run under the fast path's rounding policy (oracle/policy.py: BN folded
    in fp32, bf16 conv operands, fp32 accumulate, bf16 residual stream) is the budget, per head, with 25 % head-room.
    profiles/r02_drift_trace_spp640.md holds the same comparison launch by launch.  A mis-wired layer, a wrong pad or a
    dropped tap is an error the budget does not contain."""
    from oracle import models as om
    from oracle.policy import run_policy
    case = C.FULL_CASES["spp_640"]
    model, sd, x = build_case(case)
    io_pol, p_pol = run_policy(om.spp_forward, sd, x, C.SPP_ANCHORS, 80, policy="bf16")
    io_ref, p_ref = oracle_forward(case, sd, x)
    model = model.to(DEV)
    with torch.no_grad():
        io, p = model(x.to(DEV))
    rel = lambda a, b: float((a.double() - b.double()).norm() / b.double().norm())
    for k in range(3):
        got, budget = rel(p[k].cpu(), p_ref[k]), rel(p_pol[k], p_ref[k])
        mx, mx_b = float((p[k].cpu() - p_ref[k]).abs().max()), float((p_pol[k] - p_ref[k]).abs().max())
        print(f"[spp_640] head {k} raw logits vs fp32 reference: rel rms {got:.5f} (budget {budget:.5f}), max abs {mx:.3f} (budget {mx_b:.3f})")
        assert got <= 1.25 * budget + 2e-4, f"head {k}: error {got} beyond the bf16 rounding budget {budget}"
        assert mx <= 2.0 * mx_b + 0.05
    rms = lambda a, b: float((a[..., 4:].double() - b[..., 4:].double()).pow(2).mean().sqrt())
    got, budget = rms(io.cpu(), io_ref), rms(io_pol, io_ref)
    print(f"[spp_640] scores vs fp32 reference: rms {got:.6f} (budget {budget:.6f})")
    assert got <= 1.25 * budget + 1e-4
    # the stated end-to-end bound of the bf16 mode on this configuration: raw head logits within 1.0 of the reference
    # (measured 0.71; logit std is 25..40 here), hence scores within 0.25 (the sigmoid's slope is <= 1/4)
    _assert_model_close(io.cpu(), io_ref, "spp_640 HIP bf16 vs fp32 reference", score_max=0.18, score_rms=1e-2, box_rel_tol=0.02)


def _seeded_batch(n, hw, first_seed=0):
    """n images, image i = synth_images(1, hw, hw, first_seed + i): image 0 is the golden's image."""
    from pytorch_yolo_amd.utils.synthetic import synth_images
    return torch.cat([synth_images(1, hw, hw, first_seed + i) for i in range(n)], 0)


def test_headline_config_bs32_two_streams():
    """The configuration bench.py times — YOLOv3-SPP 640x640, 32 images, two sub-batch streams of 16 — end to end:
    image 0 against the reference golden (same bounds as the bs=1 test), the two-stream run bit-equal to one-stream runs
    of its halves, sampled images against the fp32 oracle, detect() against the oracle NMS."""
    from oracle import models as om
    from oracle import nms as onms
    case = C.FULL_CASES["spp_640"]
    model, sd, _ = build_case(case)
    g = load_golden("full_spp_640")
    x = _seeded_batch(32, 640)
    model = model.to(DEV)
    xd = x.to(DEV)
    with torch.no_grad():
        model.n_streams = 2
        assert type(model.plan_for(xd)).__name__ == "StreamedPlan"
        io, p = model(xd)
        io_c = io.cpu()
        _assert_model_close(io_c[:1][:, g["rows"]], torch.from_numpy(g["io_rows"]), "spp_640x32 image 0 / golden rows", score_max=0.18, score_rms=1e-2)
        model.n_streams = 1
        for half in (0, 1):
            io_h, _ = model(xd[16 * half:16 * half + 16])
            assert torch.equal(io_h, io[16 * half:16 * half + 16]), "two-stream run differs from the one-stream run of its half"
        for i in (0, 13, 31):
            io_ref, _ = om.spp_forward(sd, x[i:i + 1], C.SPP_ANCHORS, 80)
            _assert_model_close(io_c[i:i + 1], io_ref, f"spp_640x32 image {i} vs fp32 oracle", score_max=0.225, score_rms=1e-2)       # measured over images 0 / 13 / 31: 0.140 / 0.178 / 0.180 (+ 25 %)
        model.n_streams = 2
        dets = model.detect(xd, **C.NMS_FULL)
        # (round 4) a lone detect() runs ONE whole-batch list (engine.StreamedPlan.detect_step): the oracle NMS gets the io that
        # call produced (test_benched_launch_list_bs32_whole_batch holds that list against the reference)
        plan = model.plan_for(xd)
        fast = plan.detect_step(**C.NMS_FULL)
        assert fast is not None and fast._launched and fast.io is None          # compact NMS form: that call wrote no io
        io_w, _ = plan.new_outputs(want_p=False)                                # the same list once more, io materialised
        from pytorch_yolo_amd.utils.utils import nms_capacity
        cap = nms_capacity(plan.rows_total, model.n_class)
        scratch = (torch.empty((32, cap, 7), device=DEV), torch.empty((32, cap), dtype=torch.int32, device=DEV),
                   torch.empty((32,), dtype=torch.int32, device=DEV))
        plan.launch_detect(xd, io_w, (None, None, None), scratch, join=False, whole_batch=True, **C.NMS_FULL)
        torch.cuda.synchronize()
        io_c = io_w.cpu()
    odets, _ = onms.non_max_suppression(io_c.numpy().copy(), **C.NMS_FULL)
    for b in range(32):
        assert (dets[b] is None) == (odets[b] is None)
        if odets[b] is not None:
            assert np.array_equal(dets[b].cpu().numpy(), odets[b])
    n = [0 if d is None else len(d) for d in dets]
    print(f"[spp_640x32] detections per image: min {min(n)}, mean {sum(n) / 32:.1f}, max {max(n)}")
    assert min(n) > 0


@pytest.mark.parametrize("cu_partition", [True, False])
def test_benched_launch_list_bs32_whole_batch(cu_partition):
    """The launch list ``bench.py`` times - YOLOv3-SPP 640x640, ONE 32-image list per pipeline
    (``launch_detect(join=False, whole_batch=True)``), with ``cu_partition=True`` on CU-masked streams and tile rules sized for
    128 CUs, and without it (what ``detect_stream()`` runs) - under the oracle, at full size
    (the composition of reference utils/utils.py:374-378 on the configuration BASELINE.json quotes):
      (a) the partitioned call really launches on ExternalStreams (hipExtStreamCreateWithCUMask), the plain one never creates them;
      (b) image 0's ``io`` rows against the reference golden at the bounds of the bs = 1 test, images 13 / 31 against the fp32 oracle;
      (c) the detections of all 32 images == the oracle NMS on that ``io``, bit for bit;
      (d) raw head logits against the n = 16 / 256-CU lists that ``model(x)`` runs.  The two lists pick kernels with another K order
          on two layers only (the 80 -> 40 and 40 -> 20 stride-2 convs: parity planes vs the gather form); launches 0 - 21 are
          bit-equal, launch 22 differs by ONE bf16 ulp on 0.03 % of its outputs, and from there the share of differing values grows
          launch by launch (0.2 %, 1 %, 2.5 %, ... 80 %: profiles/r04_list_diff_spp640.txt, tools/list_diff.py) - a single flipped
          rounding re-draws the roundings downstream, so the two lists end as two realisations of the SAME bf16 rounding noise:
          their mutual distance per head must stay below the distance the rounding model puts between one realisation and fp32
          (test_bf16_path_within_its_rounding_budget: 3.6e-3 / 3.8e-3 / 6.4e-3; measured 2.7e-3 / 3.1e-3 / 5.6e-3).  "Within 2 bf16
          ulp on the head inputs"  is not reachable by any two lists that differ anywhere: see the growth table;
      (e) four successive calls (both pipelines, twice each) give bit-identical ``io`` and detections."""
    from oracle import models as om
    from oracle import nms as onms
    from pytorch_yolo_amd.utils.utils import nms_capacity
    case = C.FULL_CASES["spp_640"]
    model, sd, _ = build_case(case)
    g = load_golden("full_spp_640")
    x = _seeded_batch(32, 640)
    model = model.to(DEV)
    xd = x.to(DEV)
    model.n_streams = 2
    plan = model.plan_for(xd)
    assert type(plan).__name__ == "StreamedPlan"
    cap = nms_capacity(plan.rows_total, model.n_class)
    runs = []
    with torch.no_grad():
        io16, p16 = model(xd)                                   # two 16-image lists, ordinary streams, 256-CU rules
        torch.cuda.synchronize()
        for call in range(4):
            io, ps = plan.new_outputs(want_p=(call == 0))       # bench.py runs without p; call 0 keeps it for (d)
            out = (torch.empty((32, cap, 7), device=DEV), torch.empty((32, cap), dtype=torch.int32, device=DEV),
                   torch.empty((32,), dtype=torch.int32, device=DEV))
            plan.launch_detect(xd, io, ps, out, C.NMS_FULL["conf_thres"], C.NMS_FULL["nms_thres"], join=False, whole_batch=True,
                               cu_partition=cu_partition)
            runs.append((io, ps, out))
        torch.cuda.synchronize()
    # (a)
    assert plan._full is not None and len(plan._full) == 2 and plan._full[0].rec.input.n == 32
    if cu_partition:
        assert plan._full_streams is not None and all(type(st).__name__ == "ExternalStream" for st in plan._full_streams), \
            "cu_partition=True did not launch on CU-masked streams"
    else:
        assert plan._full_streams is None
    # (e)
    io0, ps0, out0 = runs[0]
    n0 = out0[2].cpu()
    for io, _, out in runs[1:]:
        assert torch.equal(io, io0), "whole-batch lists are not run-to-run / pipeline-to-pipeline identical"
        assert torch.equal(out[2].cpu(), n0)
        for b in range(32):
            assert torch.equal(out[0][b, :n0[b]], out0[0][b, :n0[b]])
    # (f) what bench.py / detect() / detect_stream() actually run is the COMPACT form of these lists (the heads filter their own rows,
    # io is never written): its detections are bit-equal to the ones the plain NMS made from the materialised io above
    with torch.no_grad():
        for call in range(2):
            outc = (torch.full((32, cap, 7), -1.0, device=DEV), torch.zeros((32, cap), dtype=torch.int32, device=DEV),
                    torch.zeros((32,), dtype=torch.int32, device=DEV))
            plan.launch_detect(xd, None, (None, None, None), outc, C.NMS_FULL["conf_thres"], C.NMS_FULL["nms_thres"], join=False,
                               whole_batch=True, cu_partition=cu_partition, compact=True)
            torch.cuda.synchronize()
            assert torch.equal(outc[2].cpu(), n0), "compact NMS form: counts differ from the plain form"
            for b in range(32):
                assert torch.equal(outc[0][b, :n0[b]], out0[0][b, :n0[b]]) and torch.equal(outc[1][b, :n0[b]], out0[1][b, :n0[b]])
    # (b)
    io_c = io0.cpu()
    _assert_model_close(io_c[:1][:, g["rows"]], torch.from_numpy(g["io_rows"]), "benched list, image 0 / golden rows", score_max=0.18, score_rms=1e-2)
    for i in (13, 31):
        io_ref, _ = om.spp_forward(sd, x[i:i + 1], C.SPP_ANCHORS, 80)
        _assert_model_close(io_c[i:i + 1], io_ref, f"benched list, image {i} vs fp32 oracle", score_max=0.225, score_rms=1e-2)
    # (c)
    odets, _ = onms.non_max_suppression(io_c.numpy().copy(), **C.NMS_FULL)
    dets = out0[0].cpu().numpy()
    for b in range(32):
        assert (odets[b] is None) == (int(n0[b]) == 0)
        if odets[b] is not None:
            assert np.array_equal(dets[b, :int(n0[b])], odets[b]), f"image {b}: detections differ from the oracle NMS"
    assert int(n0.min()) > 0
    # (d)
    rel = lambda a, b: float((a.double() - b.double()).norm() / b.double().norm())
    for k, budget in enumerate((3.6e-3, 3.8e-3, 6.4e-3)):
        r = rel(ps0[k], p16[k])
        print(f"[benched list, cu_partition={cu_partition}] head {k} raw logits vs the 16-image / 256-CU lists: rel rms {r:.2e} (bf16 budget {budget:.1e})")
        assert r <= budget, f"head {k}: the two lists are further apart than one of them may be from fp32"
    d_sc = float((io0[..., 4:] - io16[..., 4:]).abs().max())
    print(f"[benched list, cu_partition={cu_partition}] scores vs the 16-image lists: max abs diff {d_sc:.4f}; "
          f"detections per image: min {int(n0.min())}, mean {float(n0.float().mean()):.1f}")
    assert d_sc <= 0.25


def test_benched_list_diverges_only_where_the_summation_order_does():
    """Launch-by-launch companion of (d) above (tools/list_diff.py as a test): the 16-image / 256-CU list and the 32-image /
    128-CU list on the same images, one launch at a time.  While both lists have picked the same kernel family for every launch so
    far, outputs are BIT-EQUAL (batch size, grid size and tile shape do not change a value: K is walked in the same order); the
    first launch where the families differ (gather form vs parity planes on the 80 -> 40 stride-2 conv) may differ by one bf16
    ulp on a small share of its outputs - nothing else."""
    import ctypes as CT
    from pytorch_yolo_amd import engine, kernels as K
    from pytorch_yolo_amd._lib import OP_CONV, OP_HEAD_DECODE, YoloOp
    case = C.FULL_CASES["spp_640"]
    model, sd, _ = build_case(case)
    model = model.to(DEV)
    x = _seeded_batch(32, 640).to(DEV)

    def make(bs):
        rec = engine.Recorder(bs, 3, 640, 640)
        model._trace(rec, rec.input)
        return engine.Plan(rec, torch.device(DEV), 80, 640)
    pa, pb = make(16), make(32)
    pa.feed(x[:16].contiguous()), pb.feed(x)
    pa._bind_outputs(*pa.new_outputs()), pb._bind_outputs(*pb.new_outputs())
    fam = lambda s_: s_.split("<")[0]
    diverged, checked_equal = False, 0
    for i in range(pa.n_ops):
        if pa.op_array[i].kind == OP_HEAD_DECODE:
            break
        names = []
        for plan, cus in ((pa, 256), (pb, 128)):
            old = K.set_launch_cus(cus)
            try:
                op = plan.op_array[i]
                names.append(fam(K.conv2d_pick(op.conv, bool(op.residual), bool(op.y_aux))) if op.kind == OP_CONV else f"kind {op.kind}")
                K.run_ops(CT.cast(CT.byref(plan.op_array, i * CT.sizeof(YoloOp)), CT.POINTER(YoloOp)), 1)
            finally:
                K.set_launch_cus(old)
        outs = []
        for plan in (pa, pb):
            nd = plan.op_nodes[i]
            sy = (nd.attrs.get("up_into") or nd.attrs.get("pool_into") or nd.outs[0]) if nd.kind == "conv" else nd.outs[0]
            outs.append(sy.buf.tensor[:16, ..., sy.c_offset:sy.c_offset + sy.c])
        a, b = outs
        if names[0] == names[1] and not diverged:
            assert torch.equal(a, b), f"launch {i} ({names[0]}): same kernel family, same inputs, different outputs"
            checked_equal += 1
            continue
        if not diverged:
            diverged = True
            d = (a.float() - b.float()).abs()
            share = float((d != 0).float().mean())
            # one bf16 rounding step of the larger value; values next to zero are sums that cancel, whose fp32 order noise
            # (~1e-6 of the terms' magnitude) is many of THEIR ulps: those get an absolute allowance of 1e-4 of the tensor's maximum
            excess = float((d - (2.0 ** -7 * torch.maximum(a.float().abs(), b.float().abs()) + 1e-4 * float(a.float().abs().max()))).max())
            print(f"[list diff] launches 0..{i - 1} bit-equal; launch {i} ({names[0]} vs {names[1]}): {100 * share:.3f} % of the outputs differ, max abs {float(d.max()):.4g}")
            assert share <= 2e-3 and excess <= 0.0, "the first diverging launch differs by more than single bf16 roundings"
            break
    assert diverged and checked_equal >= 20


@pytest.mark.parametrize("spec", ["0-5:2", "0-2:4,2-5:2"])
def test_depth_first_sub_batch_lists_are_bit_equal(spec, monkeypatch):
    """engine.Plan._depth_first (YOLO_DEPTH_FIRST): the first launches of the YOLOv3-SPP list run as S passes over image sub-batches
    (stem -> 64-channel unit -> stride-2 conv -> the two 128-channel units, reference models/yolov3_spp.py:98-125) so that a
    producer's output is still in the Infinity Cache when its consumer reads it.  Same kernels, same tiles (a tile never crosses an
    image): every output - decoded rows and raw head tensors - must be BIT-EQUAL to the plain list's, twice in a row.  (16 images:
    the sub-batches must still satisfy the tile rules the plan's kernels were chosen by - 8 images in two passes put the 128-channel
    units below the 512 tiles their fused kernel wants, the generic form sums in another order, and the results differ by roundings.)"""
    from pytorch_yolo_amd import engine
    case = C.FULL_CASES["spp_640"]
    model, sd, _ = build_case(case)
    model = model.to(DEV)
    x = _seeded_batch(16, 640, first_seed=40).to(DEV)

    def make():
        rec = engine.Recorder(16, 3, 640, 640)
        model._trace(rec, rec.input)
        return engine.Plan(rec, torch.device(DEV), 80, 640)
    with torch.no_grad():
        plain = make()
        monkeypatch.setenv("YOLO_DEPTH_FIRST", spec)
        deep = make()
        assert plain.depth_first == [] and len(deep.depth_first) == len(spec.split(",")) and deep.n_ops > plain.n_ops
        io0, ps0 = plain.run(x)
        for _ in range(2):
            io1, ps1 = deep.run(x)
            torch.cuda.synchronize()
            assert torch.equal(io0, io1), "depth-first list: decoded rows differ from the plain list's"
            for a, b in zip(ps0, ps1):
                assert torch.equal(a, b)
        assert torch.isfinite(io0).all() and float(io0[..., 4].max()) > 0.1


def _small_bench_model(name):
    """(model on the device with the heads calibrated as bench.py calibrates them, CPU state_dict after that calibration, CPU batch,
    oracle forward, sampled images, score bounds) of the two small BASELINE workloads at their bench batch sizes."""
    from oracle import models as om
    from pytorch_yolo_amd import YOLOv3TinyMobile
    from pytorch_yolo_amd.utils.synthetic import calibrate_plain_heads, synth_state_dict
    if name == "tiny_416x32":
        model, _, _ = build_case(C.FULL_CASES["tiny_416"])
        x, fwd, anchors, sample, bounds = _seeded_batch(32, 416), om.tiny_forward, C.TINY_ANCHORS, (0, 7, 31), {}
    else:
        model = YOLOv3TinyMobile(n_class=80).eval()
        model.load_state_dict(synth_state_dict(model.state_dict(), 1234, n_class=80))
        # (bounds: the calibration multiplies the plain heads' weights by ~30, and with them the encoder's bf16 drift: measured on image 0
        # max score error 0.036, rms 3.3e-3, boxes within 3.9 % - the uncalibrated model's bounds are 3e-2 / 4e-3 / 2 %)
        x, fwd, anchors, sample, bounds = (_seeded_batch(64, 416, first_seed=100), om.tiny_mobile_forward, om.TINY_ANCHORS, (0, 40, 63),
                                           dict(score_max=6e-2, score_rms=6e-3, box_rel_tol=0.06))
    model = model.to(DEV)
    model.n_streams = 2
    with torch.no_grad():
        calibrate_plain_heads(model, x.to(DEV))      # (bench.py does: synthetic plain heads otherwise leave the NMS leg empty)
    sd = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    return model, sd, x, (lambda xi: fwd(sd, xi, anchors, 80)), sample, bounds


@pytest.mark.parametrize("name", ["tiny_416x32", "mobile_416x64"])
def test_benched_whole_batch_lists_of_the_small_models(name):
    """BASELINE configs 2 and 4 as ``bench.py --workload tiny | mobile`` times them: ONE whole-batch launch list
    per pipeline - 32 / 64 images, ``launch_detect(join=False, whole_batch=True)``, other tile rules than the two half-batch lists
    of ``model(x)`` - in the compact NMS form, heads calibrated as the bench calibrates them.  Reference: yolov3_tiny.py:67-100,
    yolov3_tiny_mobilenet.py:78-109, utils/utils.py:374-378.
      (b) sampled images of the materialised ``io`` against the fp32 oracle at the small-model bounds;
      (c) the detections of ALL images == the oracle NMS on that ``io``, bit for bit, and there are detections;
      (e) four calls over both pipelines bit-identical;
      (f) the compact form (what the bench runs: the heads filter their rows, no ``io``) bit-equal to the materialised one."""
    from oracle import nms as onms
    from pytorch_yolo_amd.utils.utils import nms_capacity
    model, sd, x, oracle, sample, bounds = _small_bench_model(name)
    xd = x.to(DEV)
    bs = x.shape[0]
    plan = model.plan_for(xd)
    assert type(plan).__name__ == "StreamedPlan"
    cap = nms_capacity(plan.rows_total, model.n_class)
    mk = lambda: (torch.full((bs, cap, 7), -1.0, device=DEV), torch.zeros((bs, cap), dtype=torch.int32, device=DEV),
                  torch.zeros((bs,), dtype=torch.int32, device=DEV))
    runs = []
    with torch.no_grad():
        for call in range(4):
            io, ps = plan.new_outputs(want_p=False)
            out = mk()
            plan.launch_detect(xd, io, ps, out, C.NMS_FULL["conf_thres"], C.NMS_FULL["nms_thres"], join=False, whole_batch=True)
            runs.append((io, out))
        torch.cuda.synchronize()
    assert plan._full is not None and len(plan._full) == 2 and plan._full[0].rec.input.n == bs and plan._full_streams is None
    io0, out0 = runs[0]
    n0 = out0[2].cpu()
    for io, out in runs[1:]:                                   # (e)
        assert torch.equal(io, io0) and torch.equal(out[2].cpu(), n0)
        for b in range(bs):
            assert torch.equal(out[0][b, :n0[b]], out0[0][b, :n0[b]])
    with torch.no_grad():                                      # (f)
        for call in range(2):
            outc = mk()
            plan.launch_detect(xd, None, tuple(None for _ in plan.heads), outc, C.NMS_FULL["conf_thres"], C.NMS_FULL["nms_thres"], join=False,
                               whole_batch=True, compact=True)
            torch.cuda.synchronize()
            assert torch.equal(outc[2].cpu(), n0), "compact NMS form: counts differ from the plain form"
            for b in range(bs):
                assert torch.equal(outc[0][b, :n0[b]], out0[0][b, :n0[b]]) and torch.equal(outc[1][b, :n0[b]], out0[1][b, :n0[b]])
    io_c = io0.cpu()
    for i in sample:                                           # (b)
        io_ref, _ = oracle(x[i:i + 1])
        _assert_model_close(io_c[i:i + 1], io_ref, f"{name} benched list, image {i} vs fp32 oracle", **bounds)
    odets, _ = onms.non_max_suppression(io_c.numpy().copy(), **C.NMS_FULL)       # (c)
    dets = out0[0].cpu().numpy()
    for b in range(bs):
        assert (odets[b] is None) == (int(n0[b]) == 0)
        if odets[b] is not None:
            assert np.array_equal(dets[b, :int(n0[b])], odets[b]), f"image {b}: detections differ from the oracle NMS"
    print(f"[{name} benched list] detections per image: min {int(n0.min())}, mean {float(n0.float().mean()):.1f}, max {int(n0.max())}")
    assert float(n0.float().mean()) >= 5.0, "the NMS leg is vacuous"


def test_detect_stream_ring_of_four_on_tiny416_x32():
    """``detect_stream()`` as the small models run it: ring depth 4 (two batches per pipeline in flight,
    models under 1 TFLOP per batch), one ``yolo_pipeline_step`` FFI call per batch, compact NMS form - YOLOv3-tiny 416x416 x 32,
    eleven DIFFERENT batches (more than two turns of the ring; every slot is reused while its neighbours are in flight): every batch's
    lists, in order, equal a lone ``detect()`` on that batch (the same whole-batch launch list, host-synchronous)."""
    model, sd, x, _, _, _ = _small_bench_model("tiny_416x32")
    batches = [_seeded_batch(32, 416, first_seed=200 + 32 * k) for k in range(11)]
    with torch.no_grad():
        want = [model.detect(b.to(DEV), **C.NMS_FULL) for b in batches]
        got = list(model.detect_stream((b.to(DEV, non_blocking=True) for b in batches), **C.NMS_FULL))
    plan = model.plan_for(batches[0].to(DEV))
    rings = plan.__dict__.get("_stream_rings", {})
    assert list(rings) == [4] and len(rings[4]) == 4 and all(item[4] is not None for item in rings[4]), "not the depth-4 ring of fast steps"
    assert len(got) == len(want) == 11
    n_det = 0
    for k, (w, g) in enumerate(zip(want, got)):
        assert len(w) == len(g) == 32
        for a, b in zip(w, g):
            assert (a is None) == (b is None), f"batch {k}"
            if a is not None:
                assert torch.equal(a, b), f"batch {k}: detect_stream() differs from detect()"
                n_det += len(a)
    assert n_det >= 11 * 32 * 5
    # (the batches really differ: a ring that handed out a stale slot would still pass a test on identical batches)
    assert any(want[0][i] is None or want[1][i] is None or want[0][i].shape != want[1][i].shape or not torch.equal(want[0][i], want[1][i]) for i in range(32))


def test_bf16_strict_pairing_rate_on_generic_weights():
    """A number for north_star's "bit-exact kept-index sets after NMS" in the bf16 mode on GENERIC weights: the
    strict pairing rate - same class, IoU >= 0.9, |dconf| <= 0.03 - between the detections of the benched bf16 path and
      * the reference's own detections on the golden image (full_spp_640.npz, reference utils/utils.py:200-293 on its own forward),
      * the fp32 oracle's detections over the whole 32-image bench batch,
    without guard band, noise re-runs or constructed weights.  For scale, the same rate for the CPU rounding model (the oracle re-run
    under the product's rounding points, oracle/policy.py) on the golden image: the HIP path may not pair worse than 0.9 x that.
    The synthetic weights give lattices of near-identical boxes one grid cell apart whose MERGE piles flip on score differences far
    below the bf16 drift, so the rate is a property of data + precision, not of the kernels; profiles/r05_pairing_rate.txt
    holds the printed table, profiles/r05_drift_attribution.md says which rounding points carry the drift (no cheap mixed mode)."""
    from oracle import models as om, nms as onms
    from oracle.policy import run_policy
    case = C.FULL_CASES["spp_640"]
    model, sd, x0 = build_case(case)
    g = load_golden("full_spp_640")
    ref = g["nms_dets_0"]
    x = _seeded_batch(32, 640)
    assert torch.equal(x[:1], x0)
    model = model.to(DEV)
    model.n_streams = 2
    with torch.no_grad():
        dets = model.detect(x.to(DEV), **C.NMS_FULL)            # one whole-batch list per call, compact NMS form: the benched path
    d = [np.zeros((0, 7), np.float32) if t is None else t.cpu().numpy() for t in dets]
    lines = []
    a, b = strict_share(ref, d[0]), strict_share(d[0], ref)
    la, lb = strict_share(ref, d[0], 0.7, 0.06), strict_share(d[0], ref, 0.7, 0.06)
    lines.append(f"golden image vs the REFERENCE's detections: reference {len(ref)}, bf16 {len(d[0])}; strict (IoU 0.9, dconf 0.03) {a:.3f} / {b:.3f}; loose (IoU 0.7, dconf 0.06) {la:.3f} / {lb:.3f}")
    io_pol, _ = run_policy(om.spp_forward, sd, x[:1], C.SPP_ANCHORS, 80, policy="bf16")
    dp, _ = onms.non_max_suppression(io_pol.numpy().copy(), **C.NMS_FULL)
    pa, pb = strict_share(ref, dp[0]), strict_share(dp[0], ref)
    lines.append(f"golden image, CPU rounding model vs the reference: {len(dp[0])} detections; strict {pa:.3f} / {pb:.3f}")
    tot = [0, 0, 0, 0, 0, 0]                                     # oracle dets, paired strictly, bf16 dets, paired strictly, loose pairs both ways
    per_image = []
    with torch.no_grad():
        for lo in range(0, 32, 4):
            io_ref, _ = om.spp_forward(sd, x[lo:lo + 4], C.SPP_ANCHORS, 80)
            od, _ = onms.non_max_suppression(io_ref.numpy().copy(), **C.NMS_FULL)
            for i in range(4):
                o = np.zeros((0, 7), np.float32) if od[i] is None else od[i]
                h = d[lo + i]
                sa, sb = strict_share(o, h), strict_share(h, o)
                per_image.append(min(sa, sb))
                tot[0] += len(o); tot[1] += round(sa * len(o)); tot[2] += len(h); tot[3] += round(sb * len(h))
                tot[4] += round(strict_share(o, h, 0.7, 0.06) * len(o)); tot[5] += round(strict_share(h, o, 0.7, 0.06) * len(h))
    ra, rb = tot[1] / max(1, tot[0]), tot[3] / max(1, tot[2])
    lines.append(f"32-image batch vs the fp32 ORACLE's detections: oracle {tot[0]}, bf16 {tot[2]}; strict {ra:.3f} / {rb:.3f}; loose {tot[4] / max(1, tot[0]):.3f} / {tot[5] / max(1, tot[2]):.3f}; "
                 f"per image (min of both directions): min {min(per_image):.3f}, median {sorted(per_image)[16]:.3f}, max {max(per_image):.3f}")
    for ln in lines:
        print("[pairing rate] " + ln)
    out_dir = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    if os.path.isdir(out_dir):
        open(os.path.join(out_dir, "r5_pairing_rate.txt"), "w").write("\n".join(lines) + "\n")
    assert min(a, b) >= 0.9 * min(pa, pb) - 0.02, "the HIP bf16 path pairs worse with the reference than the CPU rounding model does"
    assert min(ra, rb) >= PAIRING_FLOOR and abs(tot[0] - tot[2]) <= 0.1 * tot[0]


PAIRING_FLOOR = 0.75      # strict pairing rate over the 32-image batch: measured 0.838 / 0.827 (profiles/r05_pairing_rate.txt); the CPU rounding model pairs 0.84 / 0.80 on the golden image


def test_secondary_configs_at_bench_batch_sizes():
    """YOLOv3-tiny 416x416 bs=32 (two streams of 16) and YOLOv3-tiny/MobileNetV2 416x416 bs=64 (two streams of 32): sampled
    images against the fp32 oracle with the small-model bounds."""
    from oracle import models as om
    from pytorch_yolo_amd import YOLOv3TinyMobile
    from pytorch_yolo_amd.utils.synthetic import synth_state_dict
    case = C.FULL_CASES["tiny_416"]
    model, sd, _ = build_case(case)
    x = _seeded_batch(32, 416)
    model = model.to(DEV)
    with torch.no_grad():
        io, _ = model(x.to(DEV))
        assert type(model.plan_for(x.to(DEV))).__name__ == "StreamedPlan"
        g = load_golden("full_tiny_416")
        _assert_model_close(io.cpu()[:1][:, g["rows"]], torch.from_numpy(g["io_rows"]), "tiny_416x32 image 0 / golden rows")
        for i in (7, 31):
            io_ref, _ = om.tiny_forward(sd, x[i:i + 1], C.TINY_ANCHORS, 80)
            _assert_model_close(io.cpu()[i:i + 1], io_ref, f"tiny_416x32 image {i}")
    mob = YOLOv3TinyMobile(n_class=80).eval()
    sdm = synth_state_dict(mob.state_dict(), 1234, n_class=80)
    mob.load_state_dict(sdm)
    mob = mob.to(DEV)
    xm = _seeded_batch(64, 416, first_seed=100)
    with torch.no_grad():
        io, _ = mob(xm.to(DEV))
        for i in (0, 40, 63):
            io_ref, _ = om.tiny_mobile_forward(sdm, xm[i:i + 1], om.TINY_ANCHORS, 80)
            _assert_model_close(io.cpu()[i:i + 1], io_ref, f"mobile_416x64 image {i}", score_max=3e-2, score_rms=4e-3)


def test_pack_detections_and_events():
    """yolo_pack_detections (the kept rows of all images back to back, with and without their pivot rows; images without
    detections; counts beyond the capacity are clipped to it) and the library's events (record on a side stream, wait on the host)."""
    from pytorch_yolo_amd import kernels as K
    g = torch.Generator().manual_seed(3)
    bs, cap = 7, 20
    dets = torch.rand(bs, cap, 7, generator=g).to(DEV)
    idx = torch.randint(0, 1000, (bs, cap), generator=g, dtype=torch.int32).to(DEV)
    counts = [3, 0, 20, 1, 0, 25, 7]                                      # (25 > cap: the NMS reports the true count, cap rows exist)
    count = torch.tensor(counts, dtype=torch.int32, device=DEV)
    kept = [min(n, cap) for n in counts]
    packed = torch.full((sum(kept), 7), -1.0, device=DEV)
    pidx = torch.full((sum(kept),), -1, dtype=torch.int64, device=DEV)
    side = torch.cuda.Stream()
    ev = K.Event()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        K.pack_detections(dets, idx, count, packed, pidx)
        ev.record()
    ev.synchronize()
    want = torch.cat([dets[b, :n] for b, n in enumerate(kept) if n])
    want_idx = torch.cat([idx[b, :n] for b, n in enumerate(kept) if n]).long()
    assert torch.equal(packed, want) and torch.equal(pidx, want_idx)
    packed2 = torch.empty_like(packed)
    K.pack_detections(dets, None, count, packed2)
    torch.cuda.synchronize()
    assert torch.equal(packed2, want)


# ------------------------------------------------------------------------------------------------
# memory-safety audit of the launch lists
def _guarded(shape, dtype, rz=65536, fill=0x7F):
    """A tensor in the middle of a larger allocation whose margins hold ``fill``; returns (tensor, raw bytes, rz)."""
    numel = int(np.prod(shape))
    nbytes = numel * torch.empty((), dtype=dtype).element_size()
    raw = torch.full((nbytes + 2 * rz,), fill, dtype=torch.uint8, device=DEV)
    return raw[rz:rz + nbytes].view(dtype).view(shape), raw, rz


def _margins_intact(raw, rz, fill=0x7F):
    return bool((raw[:rz] == fill).all()) and bool((raw[-rz:] == fill).all())


@pytest.mark.parametrize("name", ["tiny_416x32", "mobile_416x64", "spp_640x32", "tiny_416x32_fp32"])
def test_launch_lists_stay_inside_their_buffers(name, monkeypatch):
    """Dynamic bounds audit of the BASELINE launch lists at the bench shapes (the tiny-416 x 32 list is the one whose HIP-graph
    replay ended in a GPU memory access fault in round 3).  Every activation buffer, packed weight and bias of the plan
    (engine.Plan, YOLO_REDZONE), the input batch, io, the NMS outputs and the NMS workspace sit inside larger allocations whose
    64 KB margins are poisoned with 0x7f bytes (3.4e38 as bf16 / f32).  After forward + decode + NMS:
      * writes: every margin still holds the poison;
      * reads: io and the detections are bit-equal to those of a plain plan - a kernel that picked up margin bytes (a max pool
        window, a halo pixel, a weight row past the matrix) would carry 3.4e38 / inf / NaN into them.
    The static side of the audit (what every launch may touch by the C ABI's contract against the planner's allocations) is
    tests/test_host_cpu.py::test_every_launch_stays_inside_the_plans_allocations."""
    from pytorch_yolo_amd import YOLOv3TinyMobile, engine, kernels as K
    from pytorch_yolo_amd.utils.synthetic import synth_state_dict
    from pytorch_yolo_amd.utils.utils import MAX_PER_CLASS, MIN_WH, nms_capacity
    fp32 = name.endswith("_fp32")
    if name.startswith("tiny"):
        model, sd, _ = build_case(C.FULL_CASES["tiny_416"])
        bs, hw = 32, 416
    elif name.startswith("mobile"):
        model = YOLOv3TinyMobile(n_class=80).eval()
        model.load_state_dict(synth_state_dict(model.state_dict(), 1234, n_class=80))
        bs, hw = 64, 416
    else:
        model, sd, _ = build_case(C.FULL_CASES["spp_640"])
        bs, hw = 32, 640
    model = model.to(DEV)
    x_plain = _seeded_batch(bs, hw).to(DEV)

    def run(guard):
        if guard:
            monkeypatch.setenv("YOLO_REDZONE", "65536")
        else:
            monkeypatch.delenv("YOLO_REDZONE", raising=False)
        rec = engine.Recorder(bs, 3, hw, hw)
        model._trace(rec, rec.input)
        plan = engine.Plan(rec, torch.device(DEV), model.n_class, hw, "fp32" if fp32 else "bf16")
        no, cap = model.n_class + 5, nms_capacity(plan.rows_total, model.n_class)
        shapes = [((bs, 3, hw, hw), torch.float32), ((bs, plan.rows_total, no), torch.float32), ((bs, cap, 7), torch.float32),
                  ((bs, cap), torch.int32), ((bs,), torch.int32), ((K.nms_workspace_bytes(bs, plan.rows_total, model.n_class),), torch.uint8)]
        shapes += [((bs, hd["na"], hd["sym"].h, hd["sym"].w, no), torch.float32) for hd in plan.heads]
        ts = [_guarded(sh, dt) if guard else (torch.empty(sh, dtype=dt, device=DEV), None, 0) for sh, dt in shapes]
        x, io, dets, idx, cnt, ws = (t[0] for t in ts[:6])
        ps = tuple(t[0] for t in ts[6:])
        x.copy_(x_plain)
        with torch.no_grad():
            plan._launch(x, io, ps)
            # (the MobileNetV2 model's uncalibrated synthetic heads score low: a threshold its survivors clear, so that the NMS kernels run)
            K.nms_merge(io, 0.002 if name.startswith("mobile") else C.NMS_FULL["conf_thres"], C.NMS_FULL["nms_thres"], dets, idx, cnt, ws,
                        min_wh=MIN_WH, max_per_class=MAX_PER_CLASS)
        torch.cuda.synchronize()
        return plan, ts, io, ps, dets, cnt

    plan_g, ts_g, io_g, ps_g, dets_g, cnt_g = run(True)
    assert len(plan_g._redzones) >= plan_g.n_ops, "the plan's buffers were not guarded"
    assert plan_g.redzone_report() == [], f"a launch wrote outside a plan buffer: {plan_g.redzone_report()}"
    for i, (_, raw, rz) in enumerate(ts_g):
        assert _margins_intact(raw, rz), f"a launch wrote outside call buffer {i} (x, io, dets, idx, count, nms workspace, p...)"
    plan_p, _, io_p, ps_p, dets_p, cnt_p = run(False)
    assert torch.equal(io_g, io_p), "io differs between the guarded and the plain plan: a kernel reads outside its buffers"
    assert all(torch.equal(a, b) for a, b in zip(ps_g, ps_p))
    assert torch.equal(cnt_g, cnt_p) and int(cnt_p.sum()) > 0
    for b in range(bs):
        assert torch.equal(dets_g[b, :cnt_p[b]], dets_p[b, :cnt_p[b]])
    assert bool(torch.isfinite(io_p).all())


# ------------------------------------------------------------------------------------------------
# boundary hygiene
def test_integration_md_stub_b_runs_verbatim():
    """INTEGRATION.md section B (the hand-written ctypes binding a reference maintainer would add) is executed exactly as
    printed and its two functions are checked against the oracle: decode (io rtol 2e-6, p exact) and NMS (bit-exact)."""
    import os
    import re
    import types
    from oracle import blocks as ob
    from oracle import nms as onms
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    text = open(os.path.join(root, "INTEGRATION.md")).read()
    section = text[text.index("## B."):text.index("## Contract reminders")]
    code = re.search(r"```python\n(.*?)```", section, re.S).group(1)
    ns = {}
    cwd = os.getcwd()
    os.chdir(root)                                             # the stub opens the library by its in-tree relative path
    try:
        exec(compile(code, "INTEGRATION.md#B", "exec"), ns)
    finally:
        os.chdir(cwd)
    nc, ny, nx, img = 80, 13, 17, 544
    anchors = torch.tensor(C.TINY_ANCHORS[1])
    layer = types.SimpleNamespace(n_classes=nc, anchors=anchors)
    g = torch.Generator().manual_seed(9)
    p_raw = torch.randn(2, 3 * (5 + nc), ny, nx, generator=g)
    io, pp = ns["yolo_layer_forward"](layer, p_raw.to(DEV), img)
    torch.cuda.synchronize()
    io_ref, p_ref = ob.yolo_decode(p_raw, C.TINY_ANCHORS[1], nc, img)
    assert torch.equal(pp.cpu(), p_ref)
    torch.testing.assert_close(io.cpu(), io_ref, rtol=2e-6, atol=1e-5)
    pred, conf, iou = C.nms_case_inputs("nms_mid_nc80")
    work = torch.from_numpy(pred.copy()).to(DEV)
    out = ns["non_max_suppression"](work, conf, iou)
    torch.cuda.synchronize()
    ref = pred.copy()
    odets, _ = onms.non_max_suppression(ref, conf, iou)
    for b in range(pred.shape[0]):
        assert (out[b] is None) == (odets[b] is None)
        if odets[b] is not None:
            assert np.array_equal(out[b].cpu().numpy(), odets[b])
    assert np.array_equal(work.cpu().numpy()[..., 4], ref[..., 4], equal_nan=True)     # mutate_conf=1: the reference's side effect


def test_plan_cache_follows_weight_changes():
    """The cached plan holds packed copies of the weights: any change of a parameter — through a SUB-module's
    load_state_dict, an in-place copy, a block-level fuse() — must rebuild it (stale weights were used silently)."""
    case = C.MODEL_CASES["tiny_small"]
    model, sd, x = build_case(case)
    model = model.to(DEV)
    xd = x.to(DEV)
    with torch.no_grad():
        io0, _ = model(xd)
        plan0 = model.plan_for(xd)
        assert model.plan_for(xd) is plan0                                   # unchanged weights: cache hit
        blk = model.sequence_1.conv1
        new = {k: v * 1.5 for k, v in blk.state_dict().items() if k.endswith("conv.weight")}
        blk.load_state_dict({**blk.state_dict(), **new})                     # sub-module load: the top-level hook never sees it
        io1, _ = model(xd)
        assert model.plan_for(xd) is not plan0 and not torch.equal(io0, io1)
        fresh, _, _ = build_case(case)
        fresh.sequence_1.conv1.load_state_dict({**fresh.sequence_1.conv1.state_dict(), **{k: v.cpu() for k, v in new.items()}})
        io1_ref, _ = fresh.to(DEV)(xd)
        assert torch.equal(io1, io1_ref)
        plan1 = model.plan_for(xd)
        model.sequence_2.conv7.sequence.batch_norm.bias.add_(0.25)           # in-place edit (what nn.init.* and optimizers do)
        io2, _ = model(xd)
        assert model.plan_for(xd) is not plan1 and not torch.equal(io1, io2)
        plan2 = model.plan_for(xd)
        model.sequence_2.conv8.fuse()                                        # block-level fuse: same function, new parameters
        io3, _ = model(xd)
        assert model.plan_for(xd) is not plan2
        _assert_model_close(io3.cpu(), io2.cpu(), "after block-level fuse", score_max=2e-3, score_rms=2e-4)
    for i in range(12):                                                      # the per-model cache is bounded
        with torch.no_grad():
            model(torch.zeros(1, 3, 32 + 32 * i, 64, device=DEV))
    from pytorch_yolo_amd.models.yolo_base import MAX_CACHED_PLANS
    assert len(model._plans) <= MAX_CACHED_PLANS


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs a second GPU")
def test_model_on_a_non_current_device():
    """Model and input on cuda:1 while the current device is cuda:0 (launches landed on GPU 0)."""
    case = C.MODEL_CASES["tiny_small"]
    model, sd, x = build_case(case)
    io_ref, _ = oracle_forward(case, sd, x)
    torch.cuda.set_device(0)
    model = model.to("cuda:1")
    with torch.no_grad():
        io, _ = model(x.to("cuda:1"))
        dets = model.detect(x.to("cuda:1"), 0.01, 0.5)
    assert io.device.index == 1 and torch.cuda.current_device() == 0
    _assert_model_close(io.cpu(), io_ref, "tiny_small on cuda:1")
    assert all(d is None or d.device.index == 1 for d in dets)


T20_CASES = [
    # n, h, w, cin, cout, use_res, use_aux: conv3x3_t20v2_kernel (4 waves x 32 couts, weights straight to registers, block-wide staging)
    # forced onto every shape class it takes (yolo_set_tuning(2, 16))
    (2, 40, 40, 64, 256, True, True, 16),
    (1, 80, 80, 32, 128, False, False, 16),        # one channel chunk: prologue only, no halo double-buffering
    (3, 37, 41, 96, 384, True, True, 16),          # partial tiles on both edges, three chunks (odd count), three cout tiles
    (1, 20, 20, 256, 512, False, True, 16),        # one tile per image, eight chunks
]


@pytest.mark.parametrize("case", T20_CASES, ids=lambda c: "n%d_%dx%d_c%d-%d_r%d_a%d_k%d" % tuple(int(v) for v in c))
def test_t20_conv_kernel(case):
    """conv3x3_t20.hip (20x20 output tiles = 25 patches of 4x4 pixels, halo staged once per 32-channel chunk) forced onto
    layers of every shape class it takes: against fp32 torch on the same bf16-rounded operands, the untouched channels of
    the output / pre-add views intact, and within fp32 summation-order noise of the shipped kernels' result."""
    from pytorch_yolo_amd import kernels as K
    from pytorch_yolo_amd._lib import ACT_LEAKY01, load
    n, h, w, cin, cout, use_res, use_aux, knob = case
    g = torch.Generator().manual_seed(hash(case) & 0xffff)
    x = torch.randn(n, cin, h, w, generator=g)
    wt = torch.randn(cout, cin, 3, 3, generator=g) * (2.0 / (cin * 9)) ** 0.5
    bias = torch.randn(cout, generator=g) * 0.1
    res = torch.randn(n, cout, h, w, generator=g) if use_res else None
    in_ct, in_co = cin + 16, 8
    xin = torch.zeros(n, h, w, in_ct, dtype=torch.bfloat16, device=DEV)
    xin[..., in_co:in_co + cin] = _nhwc(x)
    out_ct, out_co = cout + 8, 8
    rin = _nhwc(res) if use_res else None
    wp, bp, kpad, cout_pad = K.pack_conv_weight(wt, bias, cin)
    d = K.conv_desc(n=n, h=h, w=w, cin=cin, in_c_total=in_ct, in_c_offset=in_co, cout=cout, out_c_total=out_ct,
                    out_c_offset=out_co, ksize=3, stride=1, act=ACT_LEAKY01, kpad=kpad, cout_pad=cout_pad,
                    res=(cout, 0) if use_res else (0, 0), aux=(cout + 8, 8) if use_aux else (0, 0))
    outs = {}
    lib = load()
    old = lib.yolo_set_tuning(2, 0)
    try:
        for arm in (64, knob):                      # 64: the 20x20-tile kernel never (the other kernels' result)
            lib.yolo_set_tuning(2, arm)
            y = torch.full((n, h, w, out_ct), -77.0, dtype=torch.bfloat16, device=DEV)
            aux = torch.full((n, h, w, cout + 8), -77.0, dtype=torch.bfloat16, device=DEV) if use_aux else None
            K.conv2d(xin, wp.to(DEV), bp.to(DEV), y, d, residual=rin, y_preadd=aux)
            torch.cuda.synchronize()
            outs[arm] = (y, aux)
    finally:
        lib.yolo_set_tuning(2, old)
    ref = F.leaky_relu(F.conv2d(_bf16r(x), _bf16r(wt), bias, padding=1), 0.1)
    pre = ref
    if use_res:
        ref = ref + _bf16r(res)
    y, aux = outs[knob]
    torch.testing.assert_close(_nchw(y[..., out_co:out_co + cout]), ref, rtol=1e-2, atol=1e-2)
    assert torch.all(y[..., :out_co] == -77.0)
    if use_aux:
        torch.testing.assert_close(_nchw(aux[..., 8:8 + cout]), pre, rtol=1e-2, atol=1e-2)
        assert torch.all(aux[..., :8] == -77.0)
    y0 = outs[64][0]
    diff = (y.float() - y0.float()).abs()
    assert float(diff.max()) <= 2 ** -6 * max(1.0, float(y0.float().abs().max())), "differs from the shipped kernels by more than 2 bf16 ulp"
    assert float((diff > 0).float().mean()) < 0.25   # same operands, another fp32 summation order: a minority of last-bit flips


@pytest.mark.parametrize("stride,act", [(1, "leaky"), (1, "relu6"), (1, "none"), (2, "leaky"), (2, "relu6"), (2, "none")])
def test_t20_epilogue_activations(stride, act):
    """The 20x20-tile kernels are templated on LeakyReLU(0.1) (round 3: `leaky4`, 2 v_pk_mul + 4 v_max per four values) and keep the
    data-independent min(max(v, lo), hi) form for the other activations: both instantiations of the stride-1 and the stride-2 kernel
    (forced with yolo_set_tuning(2, 16)) against fp32 torch, and a NaN / +inf / -inf pre-activation (through the bias) comes out as
    the reference's activation would give it (LeakyReLU / none: NaN stays NaN, both infinities stay; ReLU6 clamps the infinities)."""
    from pytorch_yolo_amd import kernels as K
    from pytorch_yolo_amd._lib import ACT_LEAKY01, ACT_NONE, ACT_RELU6, load
    n, hw, cin, cout = 2, 40 * stride, 64, 128
    g = torch.Generator().manual_seed(17 + stride)
    x = torch.randn(n, cin, hw, hw, generator=g)
    wt = torch.randn(cout, cin, 3, 3, generator=g) * (2.0 / (cin * 9)) ** 0.5
    bias = torch.randn(cout, generator=g) * 0.1
    bias[3], bias[17], bias[cout - 2] = float("nan"), float("inf"), float("-inf")
    wp, bp, kpad, cout_pad = K.pack_conv_weight(wt, bias, cin)
    ho = (hw - 1) // stride + 1
    y = torch.zeros(n, ho, ho, cout, dtype=torch.bfloat16, device=DEV)
    d = K.conv_desc(n=n, h=hw, w=hw, cin=cin, in_c_total=cin, in_c_offset=0, cout=cout, out_c_total=cout, out_c_offset=0, ksize=3,
                    stride=stride, act={"leaky": ACT_LEAKY01, "none": ACT_NONE, "relu6": ACT_RELU6}[act], kpad=kpad, cout_pad=cout_pad)
    lib = load()
    old = lib.yolo_set_tuning(2, 16)
    try:
        assert "t20" in K.conv2d_pick(d)                 # (the forced rule: the kernel under test really is the one launched)
        K.conv2d(_nhwc(x), wp.to(DEV), bp.to(DEV), y, d)
        torch.cuda.synchronize()
    finally:
        lib.yolo_set_tuning(2, old)
    f = {"leaky": lambda t: F.leaky_relu(t, 0.1), "none": lambda t: t, "relu6": F.relu6}[act]
    ref = f(F.conv2d(_bf16r(x), _bf16r(wt), bias, stride=stride, padding=1))
    got = _nchw(y)
    keep = [c for c in range(cout) if c not in (3, 17, cout - 2)]
    torch.testing.assert_close(got[:, keep], ref[:, keep], rtol=1e-2, atol=1e-2)
    if act == "relu6":        # (the clamp form maps a NaN to the lower bound; pinned for LeakyReLU / none, which SPP's layers use)
        assert (got[:, 17] == 6.0).all() and (got[:, cout - 2] == 0.0).all()
    else:
        assert torch.isnan(got[:, 3]).all()
        assert (got[:, 17] == float("inf")).all() and (got[:, cout - 2] == float("-inf")).all()


T20S2_CASES = [
    # n, h, w (input), cin, cout, use_aux: stride-2 3x3 on 20x20 output tiles, four parity planes per 32-channel chunk (round 3)
    (2, 80, 80, 64, 128, False),          # whole tiles, two chunks (the 320 -> 160 layer's shape class)
    (1, 40, 40, 256, 512, True),          # one tile per image, eight chunks, four cout tiles, pre-add copy
    (3, 75, 83, 96, 256, True),           # odd input sizes: outputs 38 x 42, partial tiles on both edges, three chunks
    (2, 41, 40, 32, 128, False),          # one chunk: prologue and the dummy plane only; odd height (the last input row is used)
    (1, 160, 160, 128, 256, False),       # 16 tiles per image
    # (round 4: layers with an even number of chunks run the chunk-PAIR step order - cases 1, 2, 5 above -, odd ones - 3 and 1 chunks -
    # the chunk-by-chunk order)
    (1, 150, 146, 512, 128, True),        # 16 chunks = 8 pairs, partial tiles
]


@pytest.mark.parametrize("case", T20S2_CASES, ids=lambda c: "n%d_%dx%d_c%d-%d_a%d" % tuple(int(v) for v in c))
def test_t20_stride2_conv_kernel(case):
    """conv3x3s2_t20_kernel (reference DownSample.conv0, models/yolov3_spp.py:26-27: 3x3 / stride 2 / pad 1 + BN + LeakyReLU) forced
    onto every shape class: against fp32 torch on the same bf16-rounded operands, untouched channels of the output views intact,
    and within fp32 summation-order noise of the gather kernel's result."""
    from pytorch_yolo_amd import kernels as K
    from pytorch_yolo_amd._lib import ACT_LEAKY01, load
    n, h, w, cin, cout, use_aux = case
    g = torch.Generator().manual_seed(hash(case) & 0xffff)
    x = torch.randn(n, cin, h, w, generator=g)
    wt = torch.randn(cout, cin, 3, 3, generator=g) * (2.0 / (cin * 9)) ** 0.5
    bias = torch.randn(cout, generator=g) * 0.1
    ho, wo = (h - 1) // 2 + 1, (w - 1) // 2 + 1
    in_ct, in_co = cin + 16, 8
    xin = torch.zeros(n, h, w, in_ct, dtype=torch.bfloat16, device=DEV)
    xin[..., in_co:in_co + cin] = _nhwc(x)
    out_ct, out_co = cout + 8, 8
    wp, bp, kpad, cout_pad = K.pack_conv_weight(wt, bias, cin)
    d = K.conv_desc(n=n, h=h, w=w, cin=cin, in_c_total=in_ct, in_c_offset=in_co, cout=cout, out_c_total=out_ct,
                    out_c_offset=out_co, ksize=3, stride=2, act=ACT_LEAKY01, kpad=kpad, cout_pad=cout_pad,
                    aux=(cout + 8, 8) if use_aux else (0, 0))
    assert (d.ho, d.wo) == (ho, wo)
    outs = {}
    lib = load()
    old = lib.yolo_set_tuning(2, 0)
    try:
        for arm in (64, 16):                        # 64: the 20x20-tile kernels never (the gather kernel's result); 16: always
            lib.yolo_set_tuning(2, arm)
            assert ("t20s2" in K.conv2d_pick(d, False, use_aux)) == (arm == 16)
            y = torch.full((n, ho, wo, out_ct), -77.0, dtype=torch.bfloat16, device=DEV)
            aux = torch.full((n, ho, wo, cout + 8), -77.0, dtype=torch.bfloat16, device=DEV) if use_aux else None
            K.conv2d(xin, wp.to(DEV), bp.to(DEV), y, d, y_preadd=aux)
            torch.cuda.synchronize()
            outs[arm] = (y, aux)
    finally:
        lib.yolo_set_tuning(2, old)
    ref = F.leaky_relu(F.conv2d(_bf16r(x), _bf16r(wt), bias, stride=2, padding=1), 0.1)
    y, aux = outs[16]
    torch.testing.assert_close(_nchw(y[..., out_co:out_co + cout]), ref, rtol=1e-2, atol=1e-2)
    assert torch.all(y[..., :out_co] == -77.0)
    if use_aux:
        torch.testing.assert_close(_nchw(aux[..., 8:8 + cout]), ref, rtol=1e-2, atol=1e-2)
        assert torch.all(aux[..., :8] == -77.0)
    y0 = outs[64][0]
    diff = (y.float() - y0.float()).abs()
    assert float(diff.max()) <= 2 ** -6 * max(1.0, float(y0.float().abs().max())), "differs from the gather kernel by more than 2 bf16 ulp"
    assert float((diff > 0).float().mean()) < 0.25
    y2 = torch.full_like(y, -77.0)                   # run to run identical (a race between the plane buffers would show here)
    lib.yolo_set_tuning(2, 16)
    try:
        K.conv2d(xin, wp.to(DEV), bp.to(DEV), y2, d, y_preadd=aux)
        torch.cuda.synchronize()
    finally:
        lib.yolo_set_tuning(2, old)
    assert torch.equal(y, y2)


# ------------------------------------------------------------------------------------------------
# EfficientNet-B0 encoder (third of three): swish, depthwise 3x3 / 5x5 with TensorFlow "same" padding,
# squeeze-and-excitation.  torch is the reference for the kernels; the encoder oracle restates the published architecture
# (efficientnet_pytorch absent: parity unpinned, oracle/efficientnet.py).
@pytest.mark.parametrize("k,stride,h,w,c,act", [(3, 1, 20, 26, 32, "swish"), (3, 2, 26, 26, 96, "swish"), (5, 2, 52, 52, 144, "swish"),
                                                (5, 1, 13, 13, 672, "swish"), (5, 2, 13, 13, 40, "relu6"), (3, 2, 27, 27, 24, "none")])
def test_dwconv_general(k, stride, h, w, c, act):
    """yolo_dwconv_fwd against fp32 torch on the same bf16 input: both kernel sizes and strides, even sizes (the odd pad row /
    column below / right) and odd ones (symmetric), channel-offset views on both sides."""
    from oracle.efficientnet import conv_same, swish
    from pytorch_yolo_amd import kernels as K
    from pytorch_yolo_amd._lib import ACT_NONE, ACT_RELU6, ACT_SWISH
    from pytorch_yolo_amd.engine import Recorder
    n = 2
    g = torch.Generator().manual_seed(k * 100 + h + c)
    x = torch.randn(n, c, h, w, generator=g)
    wt = torch.randn(c, 1, k, k, generator=g) * 0.3
    b = torch.randn(c, generator=g) * 0.1
    (ho, pad), (wo, _) = Recorder.tf_same(h, k, stride), Recorder.tf_same(w, k, stride)
    xin = torch.zeros(n, h, w, c + 16, dtype=torch.bfloat16, device=DEV)
    xin[..., 8:8 + c] = _nhwc(x)
    y = torch.full((n, ho, wo, c + 8), -77.0, dtype=torch.bfloat16, device=DEV)
    K.dwconv(xin, wt.reshape(c, k * k).t().contiguous().to(DEV), b.to(DEV), y, n=n, h=h, w=w, c=c, in_view=(c + 16, 8),
             out_view=(c + 8, 8), ho=ho, wo=wo, ksize=k, stride=stride, pad=pad, act={"swish": ACT_SWISH, "relu6": ACT_RELU6, "none": ACT_NONE}[act])
    ref = conv_same(_bf16r(x), wt, b, stride=stride, groups=c)
    ref = {"swish": swish, "relu6": F.relu6, "none": lambda t: t}[act](ref)
    assert ref.shape[-2:] == (ho, wo)
    torch.testing.assert_close(_nchw(y[..., 8:]), ref, rtol=1e-2, atol=1e-2)
    assert torch.all(y[..., :8] == -77.0)


@pytest.mark.parametrize("n,h,w,c,sq", [(2, 13, 13, 672, 28), (3, 52, 52, 96, 4), (1, 104, 104, 32, 8), (2, 7, 9, 1152, 48)])
def test_squeeze_excite(n, h, w, c, sq):
    """yolo_se_fwd against torch: y = x * sigmoid(W2 swish(W1 mean(x) + b1) + b2), in place of nothing (separate output view)."""
    from oracle.efficientnet import swish
    from pytorch_yolo_amd import kernels as K
    g = torch.Generator().manual_seed(c + sq)
    x = torch.randn(n, c, h, w, generator=g)
    w1, b1 = torch.randn(sq, c, generator=g) * (1.0 / c) ** 0.5, torch.randn(sq, generator=g) * 0.1
    w2, b2 = torch.randn(c, sq, generator=g) * (1.0 / sq) ** 0.5, torch.randn(c, generator=g) * 0.1
    xin = torch.zeros(n, h, w, c + 8, dtype=torch.bfloat16, device=DEV)
    xin[..., 8:] = _nhwc(x)
    y = torch.full((n, h, w, c + 16), -77.0, dtype=torch.bfloat16, device=DEV)
    ws = torch.zeros(K.se_workspace_bytes(n, c) // 4, dtype=torch.float32, device=DEV)
    K.se(xin, y, w1.to(DEV), b1.to(DEV), w2.t().contiguous().to(DEV), b2.to(DEV), ws, n=n, h=h, w=w, c=c, in_view=(c + 8, 8), out_view=(c + 16, 16))
    xr = _bf16r(x)
    s_ = F.conv2d(swish(F.conv2d(F.adaptive_avg_pool2d(xr, 1), w1[:, :, None, None], b1)), w2[:, :, None, None], b2)
    ref = torch.sigmoid(s_) * xr
    torch.testing.assert_close(_nchw(y[..., 16:]), ref, rtol=1e-2, atol=1e-2)
    assert torch.all(y[..., :16] == -77.0)
    torch.testing.assert_close(ws[:n * c].cpu().reshape(n, c), xr.mean((2, 3)), rtol=1e-4, atol=1e-5)   # the pooled means themselves


@pytest.mark.parametrize("h,w,cin,cout,k,stride", [(32, 48, 8, 32, 3, 2), (26, 26, 96, 24, 1, 1), (40, 40, 64, 96, 3, 2)])
def test_conv_tf_same_and_swish(h, w, cin, cout, k, stride):
    """yolo_conv2d_fwd with the output one row / column beyond the symmetric-pad size (the window of the last row hangs over
    the edge by one more zero: TensorFlow "same" at stride 2 on an even map) and the swish epilogue."""
    from oracle.efficientnet import conv_same, swish
    from pytorch_yolo_amd import kernels as K
    from pytorch_yolo_amd._lib import ACT_SWISH
    from pytorch_yolo_amd.engine import Recorder
    n = 2
    g = torch.Generator().manual_seed(h + cin)
    x = torch.randn(n, cin, h, w, generator=g)
    wt = torch.randn(cout, cin, k, k, generator=g) * (2.0 / (cin * k * k)) ** 0.5
    bias = torch.randn(cout, generator=g) * 0.1
    (ho, pad), (wo, _) = Recorder.tf_same(h, k, stride), Recorder.tf_same(w, k, stride)
    wp, bp, kpad, cout_pad = K.pack_conv_weight(wt, bias, cin)
    d = K.conv_desc(n=n, h=h, w=w, cin=cin, in_c_total=cin, in_c_offset=0, cout=cout, out_c_total=cout, out_c_offset=0, ksize=k,
                    stride=stride, act=ACT_SWISH, kpad=kpad, cout_pad=cout_pad, pad=pad)
    d.ho, d.wo = ho, wo
    y = torch.zeros(n, ho, wo, cout, dtype=torch.bfloat16, device=DEV)
    K.conv2d(_nhwc(x), wp.to(DEV), bp.to(DEV), y, d)
    ref = swish(conv_same(_bf16r(x), _bf16r(wt), bias, stride=stride))
    assert ref.shape[-2:] == (ho, wo)
    torch.testing.assert_close(_nchw(y), ref, rtol=1e-2, atol=1e-2)


@pytest.mark.parametrize("n,h,w", [(2, 96, 128), (1, 416, 416), (1, 160, 224)])
def test_efficientnet_variant_vs_oracle(n, h, w):
    """YOLOv3TinyEfficient (reference models/yolov3_tiny_efficient.py): 16 MBConv blocks with swish, 3x3 / 5x5 depthwise convs under
    TensorFlow "same" padding and squeeze-excite, routes after blocks 11 and 16, the tiny-style head.  (Sizes are multiples of
    32: with ceil-mode "same" convs anything else breaks the reference's own upsample + concat; odd maps are covered at kernel
    level by test_dwconv_general / test_conv_tf_same_and_swish.)"""
    from oracle import models as om
    from pytorch_yolo_amd import YOLOv3TinyEfficient
    from pytorch_yolo_amd.utils.synthetic import synth_images, synth_state_dict
    model = YOLOv3TinyEfficient(n_class=3).eval()
    sd = synth_state_dict(model.state_dict(), 5, n_class=3)
    model.load_state_dict(sd)
    x = synth_images(n, h, w, 3)
    with torch.no_grad():
        io_ref, p_ref = om.tiny_efficient_forward(sd, x, om.TINY_ANCHORS, 3)
        io, p = model.to(DEV)(x.to(DEV))
    assert io.shape == io_ref.shape and [tuple(q.shape) for q in p] == [tuple(q.shape) for q in p_ref]
    for k_, (a, b) in enumerate(zip(p, p_ref)):
        rel = float((a.cpu().double() - b.double()).norm() / b.double().norm())
        print(f"[efficient_{h}x{w}] head {k_} raw logits: relative error {rel:.4f}")
        assert rel < 0.03
    _assert_model_close(io.cpu(), io_ref, f"efficient_{h}x{w}", score_max=4e-2, score_rms=5e-3, box_rel_tol=0.04)


@pytest.mark.parametrize("n,h,w,cin,cout,act", [(2, 80, 80, 256, 128, "leaky"), (1, 160, 160, 128, 64, "leaky"), (3, 37, 41, 128, 128, "none"),
                                              (1, 20, 20, 256, 64, "relu6"), (2, 13, 13, 256, 128, "swish"),
                                              # round 5, the pipelined form (conv1x1_stream2_kernel): many tiles per workgroup (the tile two
                                              # steps ahead lands while a tile is multiplied and stored), a last tile partly beyond the
                                              # tensor, workgroups without a tile, K = 384 on 48-pixel tiles
                                              (9, 80, 80, 256, 128, "leaky"), (2, 83, 79, 384, 128, "leaky"), (1, 9, 7, 128, 128, "relu6")])
def test_conv1x1_stream_kernel(n, h, w, cin, cout, act):
    """conv1x1_stream.hip (weights stationary in registers, persistent workgroups, whole-K pixel tiles by LDS-DMA) forced onto
    every layer it takes (YOLO_CONV_PP bit 2048): against fp32 torch on the same bf16-rounded operands, channel-offset views on
    both sides, a last tile that is partly beyond the tensor, and the tiled kernel's result within one bf16 ulp."""
    from oracle.efficientnet import swish
    from pytorch_yolo_amd import kernels as K
    from pytorch_yolo_amd._lib import ACT_LEAKY01, ACT_NONE, ACT_RELU6, ACT_SWISH, load
    g = torch.Generator().manual_seed(h * 7 + cin)
    x = torch.randn(n, cin, h, w, generator=g)
    wt = torch.randn(cout, cin, 1, 1, generator=g) * (2.0 / cin) ** 0.5
    bias = torch.randn(cout, generator=g) * 0.1
    xin = torch.zeros(n, h, w, cin + 16, dtype=torch.bfloat16, device=DEV)
    xin[..., 8:8 + cin] = _nhwc(x)
    wp, bp, kpad, cout_pad = K.pack_conv_weight(wt, bias, cin)
    code = {"leaky": ACT_LEAKY01, "none": ACT_NONE, "relu6": ACT_RELU6, "swish": ACT_SWISH}[act]
    d = K.conv_desc(n=n, h=h, w=w, cin=cin, in_c_total=cin + 16, in_c_offset=8, cout=cout, out_c_total=cout + 8, out_c_offset=8,
                    ksize=1, stride=1, act=code, kpad=kpad, cout_pad=cout_pad)
    lib = load()
    outs = {}
    old = lib.yolo_set_tuning(2, 0)
    try:
        for arm in (1024, 2048):
            lib.yolo_set_tuning(2, arm)
            y = torch.full((n, h, w, cout + 8), -77.0, dtype=torch.bfloat16, device=DEV)
            K.conv2d(xin, wp.to(DEV), bp.to(DEV), y, d)
            torch.cuda.synchronize()
            outs[arm] = y
    finally:
        lib.yolo_set_tuning(2, old)
    ref = F.conv2d(_bf16r(x), _bf16r(wt), bias)
    ref = {"leaky": lambda t: F.leaky_relu(t, 0.1), "none": lambda t: t, "relu6": F.relu6, "swish": swish}[act](ref)
    y = outs[2048]
    torch.testing.assert_close(_nchw(y[..., 8:]), ref, rtol=1e-2, atol=1e-2)
    assert torch.all(y[..., :8] == -77.0)
    diff = (y.float() - outs[1024].float()).abs()
    assert float(diff.max()) <= 2 ** -6 * max(1.0, float(y.float().abs().max()))
    # the pipelined form (cout 128) against the first form (YOLO_CONV_DEBUG bit 33554432): same fragments, same MFMA order - bit-equal;
    # and run to run identical (a tile buffer refilled before every wave has read its staged rows would show here)
    old = lib.yolo_set_tuning(2, 2048)
    old_dbg = lib.yolo_set_tuning(1, 0)
    try:
        if cout == 128 and cin in (128, 256, 384):
            assert "stream1x1p<" in K.conv2d_pick(d)
        lib.yolo_set_tuning(1, 33554432)
        if cin != 384:                                       # (the first form has no K = 384 instance)
            assert "stream1x1<" in K.conv2d_pick(d)
            y1 = torch.full_like(y, -77.0)
            K.conv2d(xin, wp.to(DEV), bp.to(DEV), y1, d)
            torch.cuda.synchronize()
            assert torch.equal(y1, y), "pipelined streaming 1x1 differs from the first form"
        lib.yolo_set_tuning(1, 0)
        for _ in range(2):
            y2 = torch.full_like(y, -77.0)
            K.conv2d(xin, wp.to(DEV), bp.to(DEV), y2, d)
            torch.cuda.synchronize()
            assert torch.equal(y2, y)
    finally:
        lib.yolo_set_tuning(1, old_dbg)
        lib.yolo_set_tuning(2, old)


def test_cu_masked_streams():
    """yolo_stream_create_cu_mask: a conv launched on a stream that owns half of every XCD's CUs gives the bits of the unmasked
    launch (work -> workgroup mapping never depends on where a workgroup lands), and the BASELINE plan (SPP-640, 16 images per
    stream) gets such streams by default while the small-launch models keep ordinary ones."""
    from pytorch_yolo_amd import kernels as K
    from pytorch_yolo_amd import engine
    from pytorch_yolo_amd._lib import ACT_LEAKY01
    g = torch.Generator().manual_seed(5)
    cin, cout, n, h, w = 128, 256, 2, 40, 40
    x = torch.randn(n, h, w, cin, generator=g).to(torch.bfloat16).to(DEV)
    wt = torch.randn(cout, cin, 3, 3, generator=g) * (2.0 / (9 * cin)) ** 0.5
    wp, bp, kpad, cout_pad = K.pack_conv_weight(wt, torch.randn(cout, generator=g) * 0.1, cin)
    wp, bp = wp.to(DEV), bp.to(DEV)
    d = K.conv_desc(n=n, h=h, w=w, cin=cin, in_c_total=cin, in_c_offset=0, cout=cout, out_c_total=cout, out_c_offset=0,
                    ksize=3, stride=1, act=ACT_LEAKY01, kpad=kpad, cout_pad=cout_pad)
    y0 = torch.zeros(n, h, w, cout, dtype=torch.bfloat16, device=DEV)
    y1 = torch.zeros_like(y0)
    K.conv2d(x, wp, bp, y0, d)
    n_cu = torch.cuda.get_device_properties(DEV).multi_processor_count
    st = K.cu_masked_stream([b for b in range(n_cu) if (b // 8) % 2 == 1], DEV)
    torch.cuda.synchronize()
    with torch.cuda.stream(st):
        K.conv2d(x, wp, bp, y1, d)
    st.synchronize()
    assert torch.equal(y0, y1) and float(y0.float().abs().max()) > 0
    with pytest.raises(RuntimeError):
        K.cu_masked_stream([], DEV)
    a = engine.StreamedPlan._make_streams(2, DEV, 32e9)
    b = engine.StreamedPlan._make_streams(2, DEV, 32e9)
    assert [type(s).__name__ for s in a] == ["ExternalStream"] * 2 and a[0].cuda_stream != a[1].cuda_stream
    assert [s.cuda_stream for s in a] == [s.cuda_stream for s in b]          # one pair per device, shared by the plans
    assert all(type(s).__name__ == "Stream" for s in engine.StreamedPlan._make_streams(2, DEV, 5e9))


@pytest.mark.parametrize("name", ["spp_kd2_nc80", "tiny_small"])
def test_detect_without_raw_head_tensors(name):
    """detect() does not materialise p (engine.Plan.new_outputs(want_p=False)): io - and so the detections - are the bits of the
    run that also stores p, for the head+decode epilogue (SPP) and the standalone decode kernel (narrow tiny heads) alike."""
    from oracle import nms as onms
    case = C.MODEL_CASES[name]
    model, sd, x = build_case(case)
    model = model.to(DEV)
    xd = x.to(DEV)
    with torch.no_grad():
        io_ref, p_ref = model(xd)
        plan = model.plan_for(xd)
        io, ps = plan.new_outputs(want_p=False)
        assert all(p is None for p in ps)
        io.fill_(-1.0)
        plan._launch(xd, io, ps)
        torch.cuda.synchronize()
        assert torch.equal(io, io_ref)
        dets = model.detect(xd, conf_thres=0.05, nms_thres=0.5)
    odets, _ = onms.non_max_suppression(io_ref.cpu().numpy().copy(), conf_thres=0.05, nms_thres=0.5)
    for b in range(x.shape[0]):
        assert (dets[b] is None) == (odets[b] is None)
        if odets[b] is not None:
            assert np.array_equal(dets[b].cpu().numpy(), odets[b])


def test_pipelined_detect_matches_joined_calls():
    """launch_detect(join=False): sub-batch pipelines on their own streams, NMS on a side stream, io shared by successive batches
    and no host sync in between - the detections of every batch equal those of joined calls (the pipeline's only cross-batch
    hazard is io: the next batch's head launches wait for the previous batch's NMS)."""
    from pytorch_yolo_amd.utils.synthetic import synth_images
    from pytorch_yolo_amd.utils.utils import nms_capacity
    case = C.MODEL_CASES["spp_kd2_nc80"]
    model, sd, _ = build_case(case)
    model = model.to(DEV)
    model.n_streams = 2
    bs = 8
    xs = [synth_images(bs, 96, 64, 30 + k).to(DEV) for k in range(6)]
    plan = model.plan_for(xs[0])
    assert type(plan).__name__ == "StreamedPlan"
    cap = nms_capacity(plan.rows_total, model.n_class)
    mk = lambda: (torch.zeros((bs, cap, 7), device=DEV), torch.zeros((bs, cap), dtype=torch.int32, device=DEV),
                  torch.zeros((bs,), dtype=torch.int32, device=DEV))
    io, ps = plan.new_outputs(want_p=False)
    with torch.no_grad():
        want = []
        for x in xs:
            out = mk()
            plan.launch_detect(x, io, ps, out, 1e-4, 0.5, join=True)
            torch.cuda.synchronize()
            want.append(out)
        got = [mk() for _ in xs]
        for _ in range(3):                                   # the same six batches three times over: 18 back-to-back steps
            for x, out in zip(xs, got):
                plan.launch_detect(x, io, ps, out, 1e-4, 0.5, join=False)
        torch.cuda.synchronize()
    assert plan._nms_stream is not None
    assert sum(int(w[2].sum()) for w in want) > 0
    for w, g in zip(want, got):
        assert torch.equal(w[2], g[2])
        for b, n in enumerate(w[2].tolist()):
            assert torch.equal(w[0][b, :n], g[0][b, :n]) and torch.equal(w[1][b, :n], g[1][b, :n])
    # whole batches alternating between the pipelines: two batches in flight, io alternates between two buffers.  Reference: the
    # one-stream plan of the whole batch (the launch list a pipeline runs; the 4-image sub-batch lists above may pick other tiles)
    model.n_streams = 1
    plan1 = model.plan_for(xs[0])
    assert type(plan1).__name__ == "Plan"
    want = []
    with torch.no_grad():
        for x in xs:
            out = mk()
            plan1.launch_detect(x, io, ps, out, 1e-4, 0.5)
            torch.cuda.synchronize()
            want.append(out)
    ios = [plan.new_outputs(want_p=False) for _ in range(2)]
    got2 = [mk() for _ in xs]
    with torch.no_grad():
        for _ in range(3):
            for k, (x, out) in enumerate(zip(xs, got2)):     # six batches per round: call k -> pipeline k % 2 -> buffer set k % 2
                plan.launch_detect(x, ios[k % 2][0], ios[k % 2][1], out, 1e-4, 0.5, join=False, whole_batch=True)
        torch.cuda.synchronize()
    assert plan._full is not None and len(plan._full) == 2
    for w, g in zip(want, got2):
        assert torch.equal(w[2], g[2])
        for b, n in enumerate(w[2].tolist()):
            assert torch.equal(w[0][b, :n], g[0][b, :n]) and torch.equal(w[1][b, :n], g[1][b, :n])


def test_detect_stream_yields_every_batch_in_order():
    """model.detect_stream(): the pipelined serving loop behind a generator - every batch comes out, in order, with the detections
    of the whole-batch launch list (= the one-stream model's detect()), whatever the number of batches (fewer than the ring,
    odd, many); inputs produced on the caller's stream right before the call are waited for."""
    from pytorch_yolo_amd.utils.synthetic import synth_images
    case = C.MODEL_CASES["spp_kd2_nc80"]
    model, sd, _ = build_case(case)
    model = model.to(DEV)
    ref, _, _ = build_case(case)
    ref = ref.to(DEV)
    ref.n_streams = 1
    host = [synth_images(8, 96, 64, 50 + k) for k in range(7)]
    with torch.no_grad():
        want = [ref.detect(h.to(DEV), 1e-4, 0.5) for h in host]
        for n in (1, 2, 7):
            got = list(model.detect_stream((h.to(DEV, non_blocking=True) for h in host[:n]), 1e-4, 0.5))
            assert len(got) == n
            for w, g in zip(want, got):
                assert len(w) == len(g) == 8
                for a, b in zip(w, g):
                    assert (a is None) == (b is None)
                    if a is not None:
                        assert torch.equal(a, b)
    assert sum(d is not None for w in want for d in w) > 0
    # the API pipelines must not run on CU-masked streams: HIP creates those as blocking streams, and the default-stream operations of
    # a serving loop (the "x is ready" event, the H2D copies above) would then serialise the two pipelines on half the chip each
    assert model.plan_for(host[0].to(DEV))._full_streams is None
    # ... and they took the one-call pipeline step (yolo_pipeline_step through engine.FastStep), not the Python launch sequence
    assert model.plan_for(host[0].to(DEV))._fast_events is not None
    def idle():      # nothing of a dropped generator may still be running: its output ring goes back to the allocator
        plan = model.plan_for(host[0].to(DEV))
        sts = list(plan.streams) + list(plan._full_streams or []) + list(plan._nms_streams or [])
        return all(st is None or st.query() for st in sts)
    with pytest.raises(RuntimeError):
        list(model.detect_stream([host[0].to(DEV), host[0][:4].to(DEV)], 1e-4, 0.5))
    assert idle()                                    # the shape error was raised with batch 0 in flight
    gen = model.detect_stream((h.to(DEV) for h in host), 1e-4, 0.5)
    first = next(gen)                                # batches 0..2 launched, 0 handed out; the consumer walks away
    gen.close()
    assert idle() and len(first) == 8
    # a batch too small to split (one plain Plan, launches on the caller's stream): the same generator, the same lists
    small = [h[:2] for h in host[:3]]
    with torch.no_grad():
        assert type(model.plan_for(small[0].to(DEV))).__name__ == "Plan"
        want_s = [ref.detect(h.to(DEV), 1e-4, 0.5) for h in small]
        got_s = list(model.detect_stream((h.to(DEV) for h in small), 1e-4, 0.5))
    for w, g in zip(want_s, got_s):
        for a, b in zip(w, g):
            assert (a is None) == (b is None) and (a is None or torch.equal(a, b))


@pytest.mark.parametrize("extra", [(), ("--pipeline", "halves", "--no-api"), ("--streams", "1", "--no-api")])
def test_bench_line_contract(extra):
    """bench.py as the driver runs it (a child process, one JSON line on stdout): the contract keys, the roofline object and the
    echo of --steps / --warmup, on the small workload in both pipeline modes."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--workload", "tiny", "--steps", "6", "--warmup", "2",
                        "--no-cpu-baseline", *extra], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
              "data", "config", "roofline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 6 and d["warmup"] == 2 and d["higher_is_better"] is True and d["scaling"] == "weak"
    assert d["value"] > 0 and abs(d["value"] * d["ms_per_step"] * 1e-3 / d["config"]["images_per_gpu"] - 1.0) < 1e-3
    rf = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in rf, k
    assert rf["bound"] in ("hbm", "mfma") and 0 < rf["frac"] < 1 and abs(rf["achieved"] / rf["peak"] - rf["frac"]) < 1e-3
    assert "workload" in d["config"] and "model" not in d["config"]
    if "--streams" not in extra:
        assert d["config"]["pipelines"].startswith("2 x " + ("sub-batches" if extra else "whole batches"))
    assert d["config"]["mean_detections_per_image"] > 0
    assert rf["ms_per_step_layers" if rf["bound"] == "hbm" else "ms_per_step_conv"] <= d["ms_per_step"] * 1.02      # a measured span, not an assumption
    assert d["config"]["sustained_images_per_s"] > 0
    if not extra:
        assert d["config"]["detect_api_images_per_s"] > 0 and d["config"]["detect_stream_api_images_per_s"] > 0
        assert 0 < d["config"]["fp32_mode_images_per_s"] < d["value"]
