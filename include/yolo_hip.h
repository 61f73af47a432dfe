/*
 * yolo_hip.h — C ABI of libyolo_hip.so: the MI355X (gfx950) YOLOv3 inference hot path.
 *
 * The reference (Dipet/pytorch_yolo) has no FFI: its seam is the nn.Module API.  Each entry
 * point below replaces the ATen dispatches of one reference function (file:line relative to
 * /root/reference/pytorch_yolo) and is what a ctypes stub on the reference side would bind
 * (INTEGRATION.md shows that stub).
 *
 * Conventions
 *  - plain pointers + sizes only; every device buffer is owned by the caller (torch tensors);
 *    the library never allocates, frees or retains device memory.  State it does keep (none of it numerics): the process-wide
 *    tuning knobs of yolo_set_tuning / yolo_set_launch_cus, per-device "LDS limit raised" flags, and the events / CU-masked streams
 *    a caller creates through yolo_event_create / yolo_stream_create_cu_mask and owns until it destroys them.
 *  - asynchronous launch on the caller's hipStream_t; no internal synchronisation; graph-capturable.
 *  - return 0 on success; >0 = hipError_t from the launch; <0 = YOLO_E_* argument error.
 *    yolo_last_error() returns a thread-local message for the last non-zero return.
 *  - activations are NHWC; "view" = (base pointer, c_offset, c_total): channel c of pixel p lives at
 *    base[p * c_total + c_offset + c].  bf16 activation channel counts/offsets are multiples of 8.
 */
#ifndef YOLO_HIP_H
#define YOLO_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* yolo_stream_t; /* hipStream_t */

#if defined(__GNUC__)
#define YOLO_API __attribute__((visibility("default")))
#else
#define YOLO_API
#endif

enum { YOLO_E_ARG = -1, YOLO_E_UNSUPPORTED = -2, YOLO_E_WORKSPACE = -3 };
enum { YOLO_ACT_NONE = 0, YOLO_ACT_LEAKY01 = 1, YOLO_ACT_RELU6 = 2, YOLO_ACT_RELU = 3,
       YOLO_ACT_SWISH = 4 /* x * sigmoid(x): SwissActivation, models/yolov3_tiny_efficient.py:13-19 */ };
enum { YOLO_DT_BF16 = 0, YOLO_DT_F32 = 1 };

YOLO_API const char* yolo_last_error(void);
/* ABI version of this header: 2.  History: 1 -> 2 (round 5; the change itself dates from round 4): YoloOp's two former padding words
 * are head_filter_conf / head_filter_min_wh, and a HEAD_DECODE op with y == NULL and workspace != NULL now means the FILTER form
 * (yolo_head_decode_filter_fwd) - a caller built against version 1 that left y NULL or garbage in `workspace` on a head op would
 * silently change behaviour, so the number moved although no struct size did (yolo_abi_sizeof cannot see such a change);
 * YoloPipeStep / yolo_pipeline_step / yolo_event_* / yolo_pack_detections / yolo_nms_merge_compact were added. */
YOLO_API int yolo_abi_version(void);
/* Tuning / A-B hook (process-wide, not part of the numerics contract): overrides what the environment variables
 * YOLO_CONV_VARIANT (knob 0), YOLO_CONV_DEBUG (knob 1), YOLO_CONV_PP (knob 2), YOLO_RESUNIT_DEBUG (knob 3) and YOLO_MBCONV_DEBUG (knob 4) set at load time.
 * Returns the old value. */
YOLO_API int yolo_set_tuning(int knob, int value);

/* ---- input packing: the `imgs.to(device)` + first-layer layout step (utils/utils.py:374) -----
 * x: f32 NCHW [n,c,h,w]  ->  y: bf16 NHWC [n,h,w,c_pad] (channels c..c_pad-1 zero). */
YOLO_API int yolo_pack_input_nchw_f32(const float* x, void* y, int n, int c, int h, int w, int c_pad,
                             yolo_stream_t s);

/* ---- ConvBlock.forward (models/yolo_base.py:19-44) with the BN already folded
 *      (utils/torch_utils.py:33-60), plain nn.Conv2d heads (models/yolov3_tiny.py:38,42),
 *      and the element-wise neighbours fused into the epilogue:
 *      Add (models/yolov3_spp.py:12-14), Upsample x2 (models/yolo_layer.py:6-13),
 *      Concat placement (models/yolo_layer.py:16-22).
 *
 *  y = act(conv(x, W) + bias);  if y_preadd: y_preadd = y;  if residual: y += residual;  store.
 *  Implicit GEMM on MFMA 32x32x16 bf16, fp32 accumulate.  groups == 1.
 */
typedef struct YoloConvDesc {
  int32_t n, h, w;                 /* input batch / spatial size                           */
  int32_t cin;                     /* logical input channels, multiple of 8                */
  int32_t in_c_total, in_c_offset; /* input view                                           */
  int32_t ho, wo;                  /* conv output spatial size (before optional upsample):
                                      (h + 2 pad - ksize) / stride + 1, or ONE more - the window of the last row /
                                      column then hangs over the bottom / right edge by one more zero (TensorFlow
                                      "same" padding of efficientnet_pytorch's Conv2dSamePadding at stride 2)       */
  int32_t cout;                    /* logical output channels                              */
  int32_t out_c_total, out_c_offset;
  int32_t ksize, stride, pad;      /* square kernel 1 or 3; zero padding                   */
  int32_t act;                     /* YOLO_ACT_*                                           */
  int32_t upsample2x;              /* 1: each output pixel is written to a 2x2 block of a
                                      [n,2ho,2wo,out_c_total] tensor                       */
  int32_t out_dtype;               /* YOLO_DT_BF16 | YOLO_DT_F32 (detection heads)         */
  int32_t kpad;                    /* packed K = roundup(ksize*ksize*cin, 64)              */
  int32_t cout_pad;                /* packed rows = roundup(cout, 128) (zero rows)         */
  int32_t res_c_total, res_c_offset; /* residual view (bf16), same spatial size as output  */
  int32_t aux_c_total, aux_c_offset; /* pre-add copy view (bf16)                           */
} YoloConvDesc;

/* w_packed: bf16 [cout_pad][kpad], k = (kh*ksize + kw)*cin + c  (yolo_pack_conv_weight_f32).
 * bias: f32 [cout_pad].  residual / y_preadd may be NULL. */
YOLO_API int yolo_conv2d_fwd(const void* x, const void* w_packed, const float* bias, const void* residual,
                    void* y, void* y_preadd, const YoloConvDesc* d, yolo_stream_t s);

/* The kernel instance + grid yolo_conv2d_fwd would launch for d ("t20v2<...> grid 512"): nothing is launched and no GPU is
 * needed - the regression guard of the tile rules (tests pin the BASELINE shapes). */
YOLO_API int yolo_conv2d_pick(const YoloConvDesc* d, int has_residual, int has_preadd, char* out, int out_len);

/* Split-K form for layers with few pixels and a long K (3x3 256 -> 512 on 13x13, 3x3 1280 -> 64 on 13x13: fewer tiles than
 * half the CUs): `splits` workgroups share a tile's K range, write fp32 partials to `workspace` and the last one to
 * arrive (counters, zero-initialised once by the caller, self-resetting) sums them in split order and runs the normal
 * epilogue - deterministic.  yolo_conv2d_splitk_plan() says whether a layer takes it (splits >= 2) and how much
 * workspace / how many int32 counters it needs; layers it declines go through yolo_conv2d_fwd.
 * EXPERIMENTAL: exact, but slower than the plain launch on MI355X today (the partial exchange crosses XCD L2s); the
 * engine only uses it with YOLO_SPLITK=1. */
YOLO_API int yolo_conv2d_splitk_plan(const YoloConvDesc* d, int has_residual, int has_preadd, int* splits, size_t* ws_bytes,
                                     int* n_counters);
YOLO_API int yolo_conv2d_splitk_fwd(const void* x, const void* w_packed, const float* bias, const void* residual, void* y,
                                    void* y_preadd, const YoloConvDesc* d, int splits, void* workspace, size_t ws_bytes,
                                    int32_t* counters, yolo_stream_t s);

/* First layer fused with the input packing: x is the caller's f32 NCHW batch [n,cin_real,h,w] (cin_real <= 8),
 * w_packed / bias as for yolo_conv2d_fwd with d->cin = 8; 3x3 / pad 1, bf16 NHWC output: stride 1 with cout 16 or 32
 * (Darknet), or stride 2 with cout 32 (MobileNetV2's first layer). */
YOLO_API int yolo_conv1_nchw_f32_fwd(const float* x_nchw, int cin_real, const void* w_packed, const float* bias,
                                     void* y, const YoloConvDesc* d, yolo_stream_t s);
/* same, followed by MaxPool2d(2, 2) (the first ConvPoolBlock of YOLOv3-tiny, models/yolo_base.py:69-80 with
 * yolov3_tiny.py:26): y_pooled is the bf16 NHWC view of the POOLED map [n, h/2, w/2, ...]; d still describes the
 * conv (ho = h, wo = w).  cout 16 or 32. */
YOLO_API int yolo_conv1_pool_nchw_f32_fwd(const float* x_nchw, int cin_real, const void* w_packed, const float* bias,
                                          void* y_pooled, const YoloConvDesc* d, yolo_stream_t s);

/* host-side helper (CPU): OIHW f32 [cout,cin_w,k,k] -> packed bf16 (round-to-nearest-even);
 * cin_w <= cin (extra input channels, e.g. the RGB->8 pad, get zero weights). */
YOLO_API int yolo_pack_conv_weight_f32(const float* w_oihw, int cout, int cin_w, int ksize, int cin,
                              int cout_pad, int kpad, uint16_t* out);

/* ---- 3x3 / stride 1 / pad 1 ConvBlock with 16 or 32 input channels and 32 or 64 output channels, optionally followed by
 *  MaxPool2d(2, 2): the second and third ConvPoolBlock of YOLOv3-tiny (models/yolo_base.py:69-80, yolov3_tiny.py:26-29)
 *  in ONE launch; with pool != 0 y is the bf16 NHWC view of the POOLED map [n, h/2, w/2, ...] and the full-resolution
 *  conv output is never written.  x: bf16 NHWC view, w_packed / bias as for yolo_conv2d_fwd, d describes the conv. */
YOLO_API int yolo_conv3x3_pool_supported(int cin, int cout);
YOLO_API int yolo_conv3x3_pool_fwd(const void* x, const void* w_packed, const float* bias, void* y, const YoloConvDesc* d,
                                   int pool, yolo_stream_t s);

/* ---- depthwise 3x3 conv + bias + act (MobileNetV2 inverted residual; torchvision, see
 *      models/yolov3_tiny_mobilenet.py:11-34).  w: f32 [9][c] tap-major, bias f32 [c]. */
YOLO_API int yolo_dwconv3x3_fwd(const void* x, const float* w, const float* bias, void* y, int n, int h, int w_,
                       int c, int in_c_total, int in_c_offset, int ho, int wo, int out_c_total,
                       int out_c_offset, int stride, int act, yolo_stream_t s);

/* ---- depthwise k x k conv (k = 3 or 5) + bias + act with an explicit leading pad (rows above / columns left of the image);
 *      windows that hang over the bottom / right edge read zeros, so TensorFlow "same" padding (efficientnet_pytorch 0.2.0
 *      Conv2dSamePadding: stride 2 on an even map pads 0 + 1 for k = 3, 1 + 2 for k = 5) is pad = pad_total / 2 with the caller's
 *      ho x wo = ceil(h / stride) x ceil(w / stride).  EfficientNet-B0's MBConvBlock._depthwise_conv behind
 *      models/yolov3_tiny_efficient.py:22-45.  w: f32 [k*k][c] tap-major, bias f32 [c]. */
YOLO_API int yolo_dwconv_fwd(const void* x, const float* w, const float* bias, void* y, int n, int h, int w_, int c,
                             int in_c_total, int in_c_offset, int ho, int wo, int out_c_total, int out_c_offset, int ksize,
                             int stride, int pad, int act, yolo_stream_t s);

/* ---- squeeze-and-excitation of an MBConvBlock (efficientnet_pytorch 0.2.0 model.py, used through
 *      models/yolov3_tiny_efficient.py:47-56): y = x * sigmoid(W2 swish(W1 mean_hw(x) + b1) + b2), per image and channel.
 *      x, y: bf16 NHWC views of c channels (y may be x); w1: f32 [squeeze][c] (= _se_reduce.weight), b1 f32 [squeeze],
 *      w2: f32 [squeeze][c] (= _se_expand.weight TRANSPOSED: lanes read consecutive channels), b2 f32 [c];
 *      workspace: yolo_se_workspace_bytes(n, c) bytes (pooled means, scales and the pooling pass's partial sums, fp32;
 *      the means stay in its first n*c floats). */
YOLO_API size_t yolo_se_workspace_bytes(int n, int c);
YOLO_API int yolo_se_fwd(const void* x, void* y, int n, int h, int w, int c, int in_c_total, int in_c_offset, int out_c_total,
                         int out_c_offset, const float* w1, const float* b1, const float* w2, const float* b2, int squeeze,
                         void* workspace, size_t ws_bytes, yolo_stream_t s);

/* ---- ShuffleNetV2's channel_shuffle(cat(a, b), groups = 2) (torchvision shufflenetv2, used by
 *  models/yolov3_tiny_shuffle.py:13-47): a and b are bf16 NHWC views holding `half` logical channels each in slots of
 *  c_slot physical channels (zero beyond `half`); y gets logical channel j = (a, b)[j % 2][j / 2] in the same two-slot
 *  layout: logical [0, half) at [0, ...), logical [half, 2 half) at [c_slot, ...); its pad channels are left alone. */
YOLO_API int yolo_channel_shuffle2_fwd(const void* a, const void* b, void* y, int n, int h, int w, int half, int c_slot,
                                       int a_c_total, int a_c_offset, int b_c_total, int b_c_offset, int y_c_total,
                                       int y_c_offset, yolo_stream_t s);

/* ---- MaxPool (models/yolo_base.py:60-66): -inf padding; the (2,1) special is pad 1, dilation 2. */
YOLO_API int yolo_maxpool_fwd(const void* x, void* y, int n, int h, int w, int c, int in_c_total, int in_c_offset,
                     int ho, int wo, int out_c_total, int out_c_offset, int ksize, int stride, int pad,
                     int dilation, yolo_stream_t s);

/* ---- SPP pyramid (models/yolov3_spp.py:75-77,129): buf is the [n,h,w,4c] concat buffer whose
 *      slice [3c,4c) already holds x; writes pool5 -> [0,c), pool9 -> [c,2c), pool13 -> [2c,3c). */
YOLO_API int yolo_spp_fwd(void* buf, int n, int h, int w, int c, yolo_stream_t s);

/* ---- one Darknet residual unit in one launch (models/yolov3_spp.py:17-32: ConvBlock 1x1 C->C/2, ConvBlock 3x3
 *  C/2->C, Add):  y = x + act(conv3x3(act(conv1x1(x,W1)+b1), W2) + b2);  y_preadd (optional) = the 3x3 output
 *  before the add (the reference returns it from the last unit of a DownSample stage).
 *  d describes the 3x3/s1/p1 conv: cin = C/2, cout = C, kpad/cout_pad = packing of W2; its INPUT view fields
 *  (in_c_total/in_c_offset) describe x (C channels); res_* fields are ignored.  W1 is packed like any 1x1
 *  ([cout_pad1][kpad1]).  y must not alias x.  yolo_resunit_supported(): C in {64,128,256} on maps >= 80x80 that
 *  tile well by 16x16; other shapes use two yolo_conv2d_fwd calls. */
YOLO_API int yolo_resunit_supported(int c, int h, int w);
/** Which kernel yolo_resunit_fwd launches for a C-channel unit on n maps of h x w: 0 not supported, 1 the generic 16x16-tile kernel
 *  (slower than the two-launch path above C = 64: a planner should not fuse there), 2 the persistent 64-channel kernel, 3 the
 *  20-pixel-wide tile kernels (conv_resunit_t20.hip).  No launch, works without a GPU. */
YOLO_API int yolo_resunit_form(int c, int n, int h, int w);
YOLO_API int yolo_resunit_fwd(const void* x, const void* w1_packed, const float* b1, const void* w2_packed,
                              const float* b2, void* y, void* y_preadd, const YoloConvDesc* d, int kpad1,
                              int cout_pad1, yolo_stream_t s);

/* ---- the Darknet stem in one launch (models/yolov3_spp.py:98 ConvBlock 3x3/s1 cin->32, then DownSample's
 *  ConvBlock 3x3/s2 32->64, :26-27):  y = act(conv_s2(act(conv_s1(x,W1)+b1), W2)+b2), x = float32 NCHW
 *  [n,cin_real,h,w] (1..8 channels), y = bf16 NHWC view at h/2 x w/2.  The 32-channel intermediate (the
 *  largest activation of the network) stays in LDS.  d describes the stride-2 conv (h,w = input size,
 *  cin 32, cout 64, kpad = packing of W2); W1 is packed as for yolo_conv1_nchw_f32_fwd (cin = 8, kpad1 >= 80). */
YOLO_API int yolo_stem_supported(int cin_real, int c1, int c2, int h, int w);
YOLO_API int yolo_stem_fwd(const float* x_nchw, int cin_real, const void* w1_packed, const float* b1, int kpad1,
                           const void* w2_packed, const float* b2, void* y, const YoloConvDesc* d, yolo_stream_t s);

/* ---- one MobileNetV2 inverted-residual block in one launch (torchvision InvertedResidual as the reference uses it,
 *  models/yolov3_tiny_mobilenet.py:14-46):  y = [x +] proj1x1(relu6(dw3x3_stride(relu6(expand1x1(x))))), BN folded.
 *  The expanded tensor (6x the input) and the depthwise output stay on the CU.  x, y: bf16 NHWC views.
 *  ce = hidden rounded up to 32, cout_pad = cout rounded up to 16, dstride = yolo_mbconv_dstride(ce) bytes.
 *    w_exp  bf16 [ce][48]: row = hidden channel, the first cin entries real, the rest zero (NULL: the block has
 *           no expand conv, hidden == cin);  b_exp f32 [ce]
 *    w_dw   f32 [9][ce] tap-major, b_dw f32 [ce]  (zero beyond hidden)
 *    w_proj bf16 [cout_pad][dstride/2]: row = output channel, the first hidden entries real;  b_proj f32 [cout_pad]
 *  has_res: y = x + ... (stride 1, cin == cout).
 *  yolo_mbconv_supported() returns the FORM that covers a block, which decides the weight images:
 *    1  cin <= 32, hidden <= 192, cout <= 64 (the blocks on the 208..52 maps of a 416x416 input): the images above;
 *    2  the wide blocks (csrc/conv_mbwide.hip: the hidden dimension streamed in chunks of 64 channels): cin a multiple of 32
 *       in 64..160 (stride 2: ..96), hidden a multiple of 64, cout <= 320, expand conv present.  Plain row-major matrices,
 *       read through buffer descriptors:  w_exp bf16 [hidden][cin], b_exp f32 [hidden], w_dw f32 [9][hidden], b_dw f32 [hidden],
 *       w_proj bf16 [cout_pad][hidden], b_proj f32 [cout_pad];
 *    0  not covered: the block runs as yolo_conv2d_fwd + yolo_dwconv3x3_fwd + yolo_conv2d_fwd. */
typedef struct YoloMbconvDesc {
  int32_t n, h, w, cin, in_c_total, in_c_offset, hidden, cout, out_c_total, out_c_offset, stride, has_expand, has_res, _pad;
} YoloMbconvDesc;
YOLO_API int yolo_mbconv_dstride(int ce);
YOLO_API int yolo_mbconv_supported(int cin, int hidden, int cout, int stride);
YOLO_API int yolo_mbconv_fwd(const void* x, const void* w_exp, const float* b_exp, const float* w_dw, const float* b_dw,
                             const void* w_proj, const float* b_proj, void* y, const YoloMbconvDesc* d, yolo_stream_t s);

/* ---- YOLOLayer.forward eval branch (models/yolo_layer.py:57-69,90-111).
 *  head: f32 NHWC [bs,ny,nx,head_c_total], channel a*(5+nc)+k.
 *  io:   f32 [bs, io_rows_total, 5+nc]; this head fills rows [io_row_offset, +na*ny*nx).
 *  p:    f32 [bs,na,ny,nx,5+nc] or NULL.
 *  anchors_px: host array of na*2 floats (pixels).  stride_px = img_size / max(nx,ny). */
YOLO_API int yolo_decode_fwd(const float* head, int head_c_total, const float* anchors_px, int na, int nc, int bs,
                    int ny, int nx, float stride_px, float* io, int io_rows_total, int io_row_offset,
                    float* p, yolo_stream_t s);

/* ---- detection head in one launch: head conv (1x1 or 3x3, stride 1; bias; act none for the plain heads of
 *  yolov3_tiny.py:38,42, leaky for the ConvBlock heads of yolov3_spp.py:104) with YOLOLayer.forward's eval branch
 *  as its epilogue: p[bs,na,ny,nx,5+nc] = the raw head values, io rows = the decoded ones (see yolo_decode_fwd).
 *  The NHWC head tensor is never materialised.  d = the head conv (cout = na*(5+nc) <= 256; its output view is
 *  ignored).  yolo_head_decode_supported(): na <= 4, 5+nc <= 128; other heads use yolo_conv2d_fwd + yolo_decode_fwd. */
YOLO_API int yolo_head_decode_supported(int cout, int na, int nc);
/* The kernel instance + grid a head op would launch (yolo_conv2d_pick for yolo_head_decode_fwd / - filter != 0 - yolo_head_decode_filter_fwd):
 * "igemm<64x256,1x8 waves,BK64,2 stages,32x32x16,decode> grid 3200"; no launch, no GPU. */
YOLO_API int yolo_head_decode_pick(const YoloConvDesc* d, int na, int nc, int filter, char* out, int out_len);
YOLO_API int yolo_head_decode_fwd(const void* x, const void* w_packed, const float* bias, const YoloConvDesc* d,
                                  const float* anchors_px, int na, int nc, float stride_px, float* io,
                                  int io_rows_total, int io_row_offset, float* p, yolo_stream_t s);

/* ---- non_max_suppression, 'MERGE' style (utils/utils.py:200-293; xywh2xyxy :46-60, bbox_iou :63-96).
 *  pred: f32 [bs,rows,5+nc].  If mutate_conf != 0 column 4 is overwritten with obj*max_cls like the
 *  reference (:213); otherwise pred is read-only.
 *  out_dets [bs,cap,7] (x1,y1,x2,y2,conf,cls_conf,cls), out_idx [bs,cap] = pivot row of each output,
 *  out_count [bs].  Rows are ordered by conf descending (ties: class, then pivot order).
 *  cap >= min(rows, nc*max_per_class) guarantees nothing is dropped; if an image would exceed cap,
 *  out_count holds the true count and only the first cap rows are written. */
YOLO_API size_t yolo_nms_workspace_bytes(int bs, int rows, int nc);
YOLO_API int yolo_nms_merge(float* pred, int bs, int rows, int nc, float conf_thres, float nms_thres, float min_wh,
                   int max_per_class, int mutate_conf, float* out_dets, int32_t* out_idx,
                   int32_t* out_count, int cap, void* workspace, size_t workspace_bytes, yolo_stream_t s);

/* ---- the compact form of the same post-process (round 4): detect() = non_max_suppression(forward(x)[0]) without ever writing io.
 *  The head convs filter their own decoded rows in their epilogue (yolo_head_decode_filter_fwd: the row filter of utils.py:212-218,
 *  operation for operation what yolo_nms_merge's first kernel does on a materialised io) and leave, in the workspace, ONE sort key
 *  per io row (~0 for a row that does not survive) plus a 32-byte record (x, y, w, h, class score) for the survivors; no atomics and
 *  nothing to initialise - every io row belongs to exactly one head.  yolo_nms_merge_compact gathers the keys and runs the same
 *  sort / MERGE / final order.  Saves the io store and its read-back (2 x 274 MB per 32 SPP-640 images) and one launch; the
 *  detections are bit-equal to yolo_nms_merge on the io the plain heads would have written.  Order per batch, on one stream or on
 *  ordered streams:  every head's yolo_head_decode_filter_fwd (together they must cover all `rows`) -> yolo_nms_merge_compact. */
YOLO_API size_t yolo_nms_compact_workspace_bytes(int bs, int rows, int nc);
YOLO_API int yolo_head_decode_filter_fwd(const void* x, const void* w_packed, const float* bias, const YoloConvDesc* d,
                                         const float* anchors_px, int na, int nc, float stride_px, int io_rows_total,
                                         int io_row_offset, float conf_thres, float min_wh, void* workspace, size_t workspace_bytes,
                                         float* p, yolo_stream_t s);
YOLO_API int yolo_nms_merge_compact(void* workspace, size_t workspace_bytes, int bs, int rows, int nc, float nms_thres,
                                    int max_per_class, float* out_dets, int32_t* out_idx, int32_t* out_count, int cap,
                                    yolo_stream_t s);

/* ---- scale_coords (utils/utils.py:296-303): map kept boxes from the network-input frame back to each original
 *  image: dets [bs,cap,row_floats] (columns 0..3 = x1,y1,x2,y2) in place; params_dev: device f32 [bs][4] =
 *  {pad_x, pad_y, gain, n_rows}; do_round = the `.round()` of the caller at utils.py:313. */
YOLO_API int yolo_scale_coords(float* dets, int bs, int cap, int row_floats, const float* params_dev, int do_round,
                               yolo_stream_t s);

/* ---- pre-processing of one decoded image: LetterBox (utils/augs.py:7-94: cv2.resize
 *  INTER_AREA by resize_ratio to rh x rw, placed at (top,left) of a th x tw rectangle with BORDER_REPLICATE) and,
 *  when dst_f32 is given, _convert_img_for_net + equalize_shapes as well (utils/dataset_csv.py:79-87,146-171:
 *  float32 /255, HWC -> CHW, the rectangle at (off_y,off_x) of a dst_h x dst_w canvas filled with `fill` = 0.5).
 *  src: uint8 [h,w,c] interleaved, c <= 4, row pitch src_pitch bytes.  Exactly one destination:
 *  dst_u8 [th,tw,c] (= LetterBox.apply's image) or dst_f32 [c,dst_h,dst_w] (one image of the NCHW batch).
 *  The geometry (rh, rw, pads) is computed by the caller exactly as LetterBox.update_params does (augs.py:24-63;
 *  pytorch_yolo_amd.utils.augs.letterbox_params).  cv2 is not available to pin the resize: see oracle/preprocess.py. */
YOLO_API int yolo_letterbox_u8_fwd(const uint8_t* src, int h, int w, int c, int src_pitch, double resize_ratio, int rh,
                                   int rw, int top, int left, int th, int tw, uint8_t* dst_u8, float* dst_f32,
                                   int dst_h, int dst_w, int off_y, int off_x, float fill, yolo_stream_t s);

/* ---- fp32 "reference-precision" mode: the same operators with float32 NHWC activations and float32 weights, the
 *  contraction on v_mfma_f32_32x32x2_f32 (exact f32 products, f32 accumulation) - the arithmetic of the reference's fp32
 *  ATen ops up to summation order.  ~1/16 of the bf16 MFMA rate: a parity mode (model.precision = "fp32"), not the bench.
 *  yolo_conv2d_f32_fwd: ConvBlock.forward / plain heads / Add / Upsample / Concat placement exactly as yolo_conv2d_fwd
 *  (models/yolo_base.py:19-44, yolov3_tiny.py:38,42, yolov3_spp.py:12-14, yolo_layer.py:6-22); views are multiples of 4
 *  channels, w_packed f32 [cout_pad][kpad] (yolo_pack_conv_weight_f32_f32; kpad multiple of 32, cout_pad of 128),
 *  d->out_dtype is ignored (always f32), any odd ksize <= 7.
 *  yolo_maxpool_f32_fwd: MaxPool (models/yolo_base.py:60-66); the SPP pyramid (yolov3_spp.py:75-77) is three calls
 *  (5 / 9 / 13, stride 1) from slice [3c,4c) into slices [0,c) [c,2c) [2c,3c) of the concat buffer.
 *  yolo_pack_input_nchw_f32_nhwc: the layout step in front of the first conv, channels zero-padded to c_pad. */
YOLO_API int yolo_conv2d_f32_fwd(const float* x, const float* w_packed, const float* bias, const float* residual, float* y,
                                 float* y_preadd, const YoloConvDesc* d, yolo_stream_t s);
YOLO_API int yolo_maxpool_f32_fwd(const float* x, float* y, int n, int h, int w, int c, int in_c_total, int in_c_offset, int ho,
                                  int wo, int out_c_total, int out_c_offset, int ksize, int stride, int pad, int dilation,
                                  yolo_stream_t s);
YOLO_API int yolo_pack_input_nchw_f32_nhwc(const float* x, float* y, int n, int c, int h, int w, int c_pad, yolo_stream_t s);
YOLO_API int yolo_pack_conv_weight_f32_f32(const float* w_oihw, int cout, int cin_w, int ksize, int cin, int cout_pad, int kpad,
                                           float* out);

/* ---- batched launcher: run a recorded list of ops with one FFI crossing (host overhead only). */
enum { YOLO_OP_CONV = 1, YOLO_OP_MAXPOOL = 2, YOLO_OP_SPP = 3, YOLO_OP_DWCONV = 4, YOLO_OP_CONV1_NCHW = 5,
       YOLO_OP_RESUNIT = 6, YOLO_OP_STEM = 7, YOLO_OP_HEAD_DECODE = 8, YOLO_OP_CONV1_POOL = 9, YOLO_OP_MBCONV = 10,
       YOLO_OP_CONV_POOL = 11 /* yolo_conv3x3_pool_fwd with pool = 1: x = bf16 NHWC, y = the pooled map */,
       YOLO_OP_SHUFFLE = 12 /* yolo_channel_shuffle2_fwd: x = a, residual = b, conv.cin = c_slot, conv.cout = half, res_* = view of b */,
       YOLO_OP_CONV_F32 = 13 /* yolo_conv2d_f32_fwd */, YOLO_OP_MAXPOOL_F32 = 14 /* yolo_maxpool_f32_fwd, fields as MAXPOOL */,
       YOLO_OP_SE = 15 /* yolo_se_fwd: x / y views in conv (n, h, w, cin, in_*, out_*), w / bias = W1 / b1, w_pre / bias_pre = W2 / b2,
                          kpad_pre = squeezed channels, workspace / ws_bytes */ };
typedef struct YoloOp {
  int32_t kind, _pad;
  const void* x; const void* w; const float* bias; const void* residual; void* y; void* y_aux;
  YoloConvDesc conv;             /* kind CONV; DWCONV/MAXPOOL/SPP reuse the geometry fields (DWCONV with ksize 0: the 3x3 / pad 1 form)
                                    (ksize/stride/pad, act, views); MAXPOOL dilation = upsample2x field;
                                    CONV1_NCHW: x = f32 NCHW input, res_c_total = real input channels;
                                    RESUNIT: the unit's 3x3 (w/bias = W2/b2), see yolo_resunit_fwd;
                                    STEM: the stride-2 conv (w/bias = W2/b2), x = f32 NCHW input,
                                    res_c_total = real input channels, see yolo_stem_fwd */
  const void* w_pre; const float* bias_pre;   /* RESUNIT / STEM: packed W1 / b1 of the leading conv */
  int32_t kpad_pre, cout_pad_pre;
  /* HEAD_DECODE (yolo_head_decode_fwd): y = io, y_aux = p (nullable), conv = the head conv.  y == NULL with workspace set:
     yolo_head_decode_filter_fwd into the compact NMS workspace (workspace / ws_bytes), thresholds head_filter_conf / head_filter_min_wh */
  float head_anchors_px[8]; float head_stride_px;
  int32_t head_na, head_nc, io_rows_total, io_row_offset; float head_filter_conf;
  /* MBCONV (yolo_mbconv_fwd): w/bias = W_proj/b_proj, w_pre/bias_pre = W_expand/b_expand (NULL: no expand conv),
     w_dw/bias_dw = the depthwise conv; geometry from conv (n,h,w,cin,views,cout,stride), hidden = kpad_pre,
     has_res = conv.res_c_total != 0 */
  const float* w_dw; const float* bias_dw;
  /* CONV with splits >= 2: yolo_conv2d_splitk_fwd */
  void* workspace; int32_t* counters; size_t ws_bytes; int32_t splits; float head_filter_min_wh;
} YoloOp;
YOLO_API int yolo_run_ops(const YoloOp* ops, int n_ops, yolo_stream_t s);

/* A HIP stream restricted to the compute units of cu_mask (bit i of the n_words x 32-bit vector = CU i; on MI355X CU i sits
 * on XCD i % 8).  The host mirror runs the sub-batches of one detect() call on disjoint CU sets (engine.StreamedPlan);
 * the reference has no counterpart (one torch stream, models/yolo_base.py forward).  Destroy with yolo_stream_destroy. */
YOLO_API int yolo_stream_create_cu_mask(const uint32_t* cu_mask, int n_words, yolo_stream_t* out);
YOLO_API int yolo_stream_destroy(yolo_stream_t s);
/* Tell the tile rules how many compute units the coming launches of THIS thread may use (default 256; a CU-masked stream's share
 * while launching on it): grids are sized against it.  Returns the previous value (<0: argument error). */
YOLO_API int yolo_set_launch_cus(int n_cu);

/* ---- one pipelined detect() step with ONE FFI crossing (round 4): what pytorch_yolo_amd.engine.StreamedPlan.launch_detect does
 *  for a whole-batch pipeline - the composition of reference utils/utils.py:374-378 (forward -> non_max_suppression) on a stream of
 *  batches - as a single call, because at YOLOv3-tiny rates (0.37 ms per 32 images) the Python between the launches was the
 *  bottleneck of detect_stream().  Order of what it enqueues:
 *    stream:      [wait wait_x] ops[0 .. k_io) [wait wait_io] ops[k_io .. n_ops) ; record heads_done
 *    nms_stream:  wait heads_done ; yolo_nms_merge(io ...) ; [count -> count_host, async] ; record nms_done ; [record done]
 *  wait_x: "the input batch is ready" (recorded by the caller on the stream that produced x); wait_io: the nms_done of the step that
 *  used this io buffer before (its NMS still reads io when the head launches - ops from k_io on - would overwrite it).
 *  Events are yolo_event_t (hipEvent_t, timing disabled) from yolo_event_create; NULL = skip.  nms_stream may equal stream.
 *  io == NULL selects the compact form: `workspace` is a yolo_nms_compact_workspace_bytes one, the head ops (already bound to it,
 *  YoloOp) filter into it behind wait_io and the NMS is yolo_nms_merge_compact.
 *  count_host: PINNED host memory for bs int32 (NULL: no copy): valid once `done` (or nms_done) has completed.
 *  Nothing is synchronised on the host; buffers stay owned by the caller as everywhere else. */
typedef void* yolo_event_t; /* hipEvent_t */
YOLO_API int yolo_event_create(yolo_event_t* out);
YOLO_API int yolo_event_destroy(yolo_event_t e);
YOLO_API int yolo_event_record(yolo_event_t e, yolo_stream_t s);
YOLO_API int yolo_event_synchronize(yolo_event_t e);
typedef struct YoloPipeStep {
  const YoloOp* ops; int32_t n_ops, k_io;
  yolo_stream_t stream, nms_stream;
  yolo_event_t wait_x, wait_io, heads_done, nms_done, done;
  float* io; int32_t bs, rows, nc, max_per_class; float conf_thres, nms_thres, min_wh; int32_t cap;
  float* out_dets; int32_t* out_idx; int32_t* out_count; void* workspace; size_t workspace_bytes;
  int32_t* count_host;
} YoloPipeStep;
YOLO_API int yolo_pipeline_step(const YoloPipeStep* st);
/* sizeof of the ABI's structs as the library was compiled (0 YoloConvDesc, 1 YoloOp, 2 YoloMbconvDesc, 3 YoloPipeStep; else -1):
 * a binding checks its own layout against it. */
YOLO_API int yolo_abi_sizeof(int which);
/* The kept rows of all images packed back to back (the reference returns one [n_i, 7] tensor per image, utils.py:293): image b's
 * min(count[b], cap) rows of dets [bs, cap, 7] go to packed rows [sum_{i<b} count'[i], ...), their pivot rows to packed_idx (nullable).
 * One launch, offsets computed on the device from `count` (the host knows the total from its copy of the counts). */
YOLO_API int yolo_pack_detections(const float* dets, const int32_t* idx, const int32_t* count, int bs, int cap, float* packed,
                                  int64_t* packed_idx, yolo_stream_t s);

#ifdef __cplusplus
}
#endif
#endif /* YOLO_HIP_H */
