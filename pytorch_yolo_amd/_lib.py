"""ctypes binding of libyolo_hip.so (include/yolo_hip.h).

There is NO fallback: if the shared library is missing or a symbol does not
resolve, importing the kernels raises.  The product path never runs on the CPU.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("YOLO_HIP_LIB") or os.path.join(_HERE, "csrc", "libyolo_hip.so")   # override: A/B runs of two builds

ABI_VERSION = 2          # include/yolo_hip.h, yolo_abi_version
ACT_NONE, ACT_LEAKY01, ACT_RELU6, ACT_RELU, ACT_SWISH = 0, 1, 2, 3, 4
DT_BF16, DT_F32 = 0, 1
OP_CONV, OP_MAXPOOL, OP_SPP, OP_DWCONV, OP_CONV1_NCHW, OP_RESUNIT, OP_STEM, OP_HEAD_DECODE, OP_CONV1_POOL = 1, 2, 3, 4, 5, 6, 7, 8, 9
OP_MBCONV, OP_CONV_POOL, OP_SHUFFLE, OP_CONV_F32, OP_MAXPOOL_F32, OP_SE = 10, 11, 12, 13, 14, 15


class YoloConvDesc(C.Structure):
    _fields_ = [(n, C.c_int32) for n in (
        "n", "h", "w", "cin", "in_c_total", "in_c_offset", "ho", "wo", "cout",
        "out_c_total", "out_c_offset", "ksize", "stride", "pad", "act", "upsample2x",
        "out_dtype", "kpad", "cout_pad", "res_c_total", "res_c_offset",
        "aux_c_total", "aux_c_offset")]


class YoloOp(C.Structure):
    _fields_ = [("kind", C.c_int32), ("_pad", C.c_int32),
                ("x", C.c_void_p), ("w", C.c_void_p), ("bias", C.c_void_p),
                ("residual", C.c_void_p), ("y", C.c_void_p), ("y_aux", C.c_void_p),
                ("conv", YoloConvDesc),
                ("w_pre", C.c_void_p), ("bias_pre", C.c_void_p), ("kpad_pre", C.c_int32), ("cout_pad_pre", C.c_int32),
                ("head_anchors_px", C.c_float * 8), ("head_stride_px", C.c_float),
                ("head_na", C.c_int32), ("head_nc", C.c_int32), ("io_rows_total", C.c_int32),
                ("io_row_offset", C.c_int32), ("head_filter_conf", C.c_float),
                ("w_dw", C.c_void_p), ("bias_dw", C.c_void_p),
                ("workspace", C.c_void_p), ("counters", C.c_void_p), ("ws_bytes", C.c_size_t),
                ("splits", C.c_int32), ("head_filter_min_wh", C.c_float)]


class YoloMbconvDesc(C.Structure):
    _fields_ = [(n, C.c_int32) for n in (
        "n", "h", "w", "cin", "in_c_total", "in_c_offset", "hidden", "cout", "out_c_total", "out_c_offset",
        "stride", "has_expand", "has_res", "_pad")]


class YoloPipeStep(C.Structure):
    _fields_ = [("ops", C.POINTER(YoloOp)), ("n_ops", C.c_int32), ("k_io", C.c_int32),
                ("stream", C.c_void_p), ("nms_stream", C.c_void_p),
                ("wait_x", C.c_void_p), ("wait_io", C.c_void_p), ("heads_done", C.c_void_p), ("nms_done", C.c_void_p), ("done", C.c_void_p),
                ("io", C.c_void_p), ("bs", C.c_int32), ("rows", C.c_int32), ("nc", C.c_int32), ("max_per_class", C.c_int32),
                ("conf_thres", C.c_float), ("nms_thres", C.c_float), ("min_wh", C.c_float), ("cap", C.c_int32),
                ("out_dets", C.c_void_p), ("out_idx", C.c_void_p), ("out_count", C.c_void_p), ("workspace", C.c_void_p),
                ("workspace_bytes", C.c_size_t), ("count_host", C.c_void_p)]


# symbol -> (restype, argtypes); kept in one table so tests can check it against the header
SIGNATURES = {
    "yolo_last_error": (C.c_char_p, []),
    "yolo_abi_version": (C.c_int, []),
    "yolo_abi_sizeof": (C.c_int, [C.c_int]),
    "yolo_set_tuning": (C.c_int, [C.c_int, C.c_int]),
    "yolo_pack_input_nchw_f32": (C.c_int, [C.c_void_p, C.c_void_p] + [C.c_int] * 5 + [C.c_void_p]),
    "yolo_conv2d_fwd": (C.c_int, [C.c_void_p] * 6 + [C.POINTER(YoloConvDesc), C.c_void_p]),
    "yolo_conv2d_splitk_plan": (C.c_int, [C.POINTER(YoloConvDesc), C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_size_t),
                                          C.POINTER(C.c_int)]),
    "yolo_conv2d_splitk_fwd": (C.c_int, [C.c_void_p] * 6 + [C.POINTER(YoloConvDesc), C.c_int, C.c_void_p, C.c_size_t, C.c_void_p,
                                         C.c_void_p]),
    "yolo_conv1_nchw_f32_fwd": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                          C.POINTER(YoloConvDesc), C.c_void_p]),
    "yolo_conv1_pool_nchw_f32_fwd": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                               C.POINTER(YoloConvDesc), C.c_void_p]),
    "yolo_conv2d_pick": (C.c_int, [C.POINTER(YoloConvDesc), C.c_int, C.c_int, C.c_char_p, C.c_int]),
    "yolo_pack_conv_weight_f32": (C.c_int, [C.c_void_p] + [C.c_int] * 6 + [C.c_void_p]),
    "yolo_stem_supported": (C.c_int, [C.c_int] * 5),
    "yolo_stem_fwd": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                C.POINTER(YoloConvDesc), C.c_void_p]),
    "yolo_resunit_supported": (C.c_int, [C.c_int] * 3),
    "yolo_resunit_form": (C.c_int, [C.c_int] * 4),
    "yolo_resunit_fwd": (C.c_int, [C.c_void_p] * 7 + [C.POINTER(YoloConvDesc), C.c_int, C.c_int, C.c_void_p]),
    "yolo_conv3x3_pool_supported": (C.c_int, [C.c_int] * 2),
    "yolo_conv3x3_pool_fwd": (C.c_int, [C.c_void_p] * 4 + [C.POINTER(YoloConvDesc), C.c_int, C.c_void_p]),
    "yolo_mbconv_dstride": (C.c_int, [C.c_int]),
    "yolo_mbconv_supported": (C.c_int, [C.c_int] * 4),
    "yolo_mbconv_fwd": (C.c_int, [C.c_void_p] * 8 + [C.POINTER(YoloMbconvDesc), C.c_void_p]),
    "yolo_dwconv3x3_fwd": (C.c_int, [C.c_void_p] * 4 + [C.c_int] * 12 + [C.c_void_p]),
    "yolo_dwconv_fwd": (C.c_int, [C.c_void_p] * 4 + [C.c_int] * 14 + [C.c_void_p]),
    "yolo_se_workspace_bytes": (C.c_size_t, [C.c_int] * 2),
    "yolo_se_fwd": (C.c_int, [C.c_void_p] * 2 + [C.c_int] * 8 + [C.c_void_p] * 4 + [C.c_int, C.c_void_p, C.c_size_t, C.c_void_p]),
    "yolo_channel_shuffle2_fwd": (C.c_int, [C.c_void_p] * 3 + [C.c_int] * 11 + [C.c_void_p]),
    "yolo_maxpool_fwd": (C.c_int, [C.c_void_p] * 2 + [C.c_int] * 14 + [C.c_void_p]),
    "yolo_spp_fwd": (C.c_int, [C.c_void_p] + [C.c_int] * 4 + [C.c_void_p]),
    "yolo_decode_fwd": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p] + [C.c_int] * 5 + [C.c_float, C.c_void_p,
                                  C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "yolo_head_decode_supported": (C.c_int, [C.c_int] * 3),
    "yolo_head_decode_pick": (C.c_int, [C.POINTER(YoloConvDesc), C.c_int, C.c_int, C.c_int, C.c_char_p, C.c_int]),
    "yolo_head_decode_fwd": (C.c_int, [C.c_void_p] * 3 + [C.POINTER(YoloConvDesc), C.c_void_p, C.c_int, C.c_int, C.c_float,
                                       C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "yolo_nms_workspace_bytes": (C.c_size_t, [C.c_int] * 3),
    "yolo_nms_merge": (C.c_int, [C.c_void_p] + [C.c_int] * 3 + [C.c_float] * 3 + [C.c_int] * 2 +
                       [C.c_void_p] * 3 + [C.c_int, C.c_void_p, C.c_size_t, C.c_void_p]),
    "yolo_nms_compact_workspace_bytes": (C.c_size_t, [C.c_int] * 3),
    "yolo_head_decode_filter_fwd": (C.c_int, [C.c_void_p] * 3 + [C.POINTER(YoloConvDesc), C.c_void_p, C.c_int, C.c_int, C.c_float,
                                              C.c_int, C.c_int, C.c_float, C.c_float, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]),
    "yolo_nms_merge_compact": (C.c_int, [C.c_void_p, C.c_size_t] + [C.c_int] * 3 + [C.c_float, C.c_int] + [C.c_void_p] * 3 +
                               [C.c_int, C.c_void_p]),
    "yolo_scale_coords": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p]),
    "yolo_letterbox_u8_fwd": (C.c_int, [C.c_void_p] + [C.c_int] * 4 + [C.c_double] + [C.c_int] * 6 + [C.c_void_p, C.c_void_p] +
                              [C.c_int] * 4 + [C.c_float, C.c_void_p]),
    "yolo_conv2d_f32_fwd": (C.c_int, [C.c_void_p] * 6 + [C.POINTER(YoloConvDesc), C.c_void_p]),
    "yolo_maxpool_f32_fwd": (C.c_int, [C.c_void_p] * 2 + [C.c_int] * 14 + [C.c_void_p]),
    "yolo_pack_input_nchw_f32_nhwc": (C.c_int, [C.c_void_p, C.c_void_p] + [C.c_int] * 5 + [C.c_void_p]),
    "yolo_pack_conv_weight_f32_f32": (C.c_int, [C.c_void_p] + [C.c_int] * 6 + [C.c_void_p]),
    "yolo_run_ops": (C.c_int, [C.POINTER(YoloOp), C.c_int, C.c_void_p]),
    "yolo_stream_create_cu_mask": (C.c_int, [C.POINTER(C.c_uint32), C.c_int, C.POINTER(C.c_void_p)]),
    "yolo_stream_destroy": (C.c_int, [C.c_void_p]),
    "yolo_set_launch_cus": (C.c_int, [C.c_int]),
    "yolo_event_create": (C.c_int, [C.POINTER(C.c_void_p)]),
    "yolo_event_destroy": (C.c_int, [C.c_void_p]),
    "yolo_event_record": (C.c_int, [C.c_void_p, C.c_void_p]),
    "yolo_event_synchronize": (C.c_int, [C.c_void_p]),
    "yolo_pipeline_step": (C.c_int, [C.POINTER(YoloPipeStep)]),
    "yolo_pack_detections": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
}

_lib = None


def load():
    """Load (once) and type the library.  Raises if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} is missing: build it with `python -m pytorch_yolo_amd.build` "
            "(hipcc --offload-arch=gfx950). pytorch_yolo_amd has no CPU / eager fallback.")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    if lib.yolo_abi_version() != ABI_VERSION:
        raise RuntimeError(f"libyolo_hip.so ABI version {lib.yolo_abi_version()} != {ABI_VERSION} of this binding: rebuild with "
                           "`python -m pytorch_yolo_amd.build --force`")
    for which, st in enumerate((YoloConvDesc, YoloOp, YoloMbconvDesc, YoloPipeStep)):
        if lib.yolo_abi_sizeof(which) != C.sizeof(st):
            raise RuntimeError(f"libyolo_hip.so: struct {st.__name__} is {lib.yolo_abi_sizeof(which)} bytes in the library, {C.sizeof(st)} in the binding")
    _lib = lib
    return lib


def check(rc: int, what: str = "") -> None:
    if rc != 0:
        msg = load().yolo_last_error().decode("utf-8", "replace")
        raise RuntimeError(f"libyolo_hip {what} failed (code {rc}): {msg}")
