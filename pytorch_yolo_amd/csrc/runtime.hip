// Error reporting, ABI version and the batched launcher of libyolo_hip.so.
#include "common.h"

static thread_local char g_err[512] = "";

int yolo_set_error(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return code;
}

extern "C" const char* yolo_last_error(void) { return g_err; }
extern "C" int yolo_abi_version(void) { return 2; }   // 2 (round 5): YoloOp's former pad fields carry head_filter_conf / head_filter_min_wh, YoloPipeStep added
extern "C" int yolo_abi_sizeof(int which) {
  switch (which) {
    case 0: return (int)sizeof(YoloConvDesc);
    case 1: return (int)sizeof(YoloOp);
    case 2: return (int)sizeof(YoloMbconvDesc);
    case 3: return (int)sizeof(YoloPipeStep);
    default: return -1;
  }
}

// A stream whose kernels run only on the compute units named in cu_mask (bit i = CU i; on MI355X bit i lies on XCD i % 8).
// engine.StreamedPlan gives each sub-batch stream its own half of every XCD so two layer lists really run side by side.
extern "C" int yolo_stream_create_cu_mask(const uint32_t* cu_mask, int n_words, yolo_stream_t* out) {
  YOLO_REQUIRE(cu_mask && out && n_words > 0 && n_words <= 64, "stream_create_cu_mask: bad arguments");
  bool any = false;
  for (int i = 0; i < n_words; ++i) any |= cu_mask[i] != 0;
  YOLO_REQUIRE(any, "stream_create_cu_mask: empty mask");
  hipStream_t st = nullptr;
  hipError_t e = hipExtStreamCreateWithCUMask(&st, (uint32_t)n_words, cu_mask);
  if (e != hipSuccess) return yolo_set_error((int)e, "hipExtStreamCreateWithCUMask: %s", hipGetErrorString(e));
  *out = (yolo_stream_t)st;
  return 0;
}
extern "C" int yolo_stream_destroy(yolo_stream_t s) {
  YOLO_REQUIRE(s, "stream_destroy: null stream");
  hipError_t e = hipStreamDestroy((hipStream_t)s);
  if (e != hipSuccess) return yolo_set_error((int)e, "hipStreamDestroy: %s", hipGetErrorString(e));
  return 0;
}

// One FFI crossing for a whole recorded layer list: the forward pass is ~80 launches, and at
// ~3 us of Python/ctypes per call the host would otherwise be a visible fraction of a 5 ms batch.
extern "C" int yolo_run_ops(const YoloOp* ops, int n_ops, yolo_stream_t s) {
  YOLO_REQUIRE(ops || n_ops == 0, "run_ops: null op list");
  for (int i = 0; i < n_ops; ++i) {
    const YoloOp& o = ops[i];
    const YoloConvDesc& d = o.conv;
    int rc;
    switch (o.kind) {
      case YOLO_OP_CONV:
        rc = o.splits >= 2 ? yolo_conv2d_splitk_fwd(o.x, o.w, o.bias, o.residual, o.y, o.y_aux, &d, o.splits, o.workspace, o.ws_bytes,
                                                    o.counters, s)
                           : yolo_conv2d_launch(o.x, o.w, o.bias, o.residual, o.y, o.y_aux, &d, (hipStream_t)s);
        break;
      case YOLO_OP_CONV1_NCHW:
        rc = yolo_conv1_nchw_f32_fwd((const float*)o.x, d.res_c_total, o.w, o.bias, o.y, &d, s);
        break;
      case YOLO_OP_CONV1_POOL:
        rc = yolo_conv1_pool_nchw_f32_fwd((const float*)o.x, d.res_c_total, o.w, o.bias, o.y, &d, s);
        break;
      case YOLO_OP_MAXPOOL:
        rc = yolo_maxpool_fwd(o.x, o.y, d.n, d.h, d.w, d.cin, d.in_c_total, d.in_c_offset, d.ho, d.wo, d.out_c_total,
                              d.out_c_offset, d.ksize, d.stride, d.pad, d.upsample2x /* dilation */, s);
        break;
      case YOLO_OP_SPP:
        rc = yolo_spp_fwd(o.y, d.n, d.h, d.w, d.cin, s);
        break;
      case YOLO_OP_DWCONV:
        rc = d.ksize == 0 ? yolo_dwconv3x3_fwd(o.x, (const float*)o.w, o.bias, o.y, d.n, d.h, d.w, d.cin, d.in_c_total, d.in_c_offset, d.ho,
                                               d.wo, d.out_c_total, d.out_c_offset, d.stride, d.act, s)
                          : yolo_dwconv_fwd(o.x, (const float*)o.w, o.bias, o.y, d.n, d.h, d.w, d.cin, d.in_c_total, d.in_c_offset, d.ho,
                                            d.wo, d.out_c_total, d.out_c_offset, d.ksize, d.stride, d.pad, d.act, s);
        break;
      case YOLO_OP_SE:
        rc = yolo_se_fwd(o.x, o.y, d.n, d.h, d.w, d.cin, d.in_c_total, d.in_c_offset, d.out_c_total, d.out_c_offset, (const float*)o.w,
                         o.bias, (const float*)o.w_pre, o.bias_pre, o.kpad_pre, o.workspace, o.ws_bytes, s);
        break;
      case YOLO_OP_RESUNIT:
        rc = yolo_resunit_fwd(o.x, o.w_pre, o.bias_pre, o.w, o.bias, o.y, o.y_aux, &d, o.kpad_pre, o.cout_pad_pre, s);
        break;
      case YOLO_OP_STEM:
        rc = yolo_stem_fwd((const float*)o.x, d.res_c_total, o.w_pre, o.bias_pre, o.kpad_pre, o.w, o.bias, o.y, &d, s);
        break;
      case YOLO_OP_HEAD_DECODE:
        if (!o.y && o.workspace) {
          rc = yolo_head_decode_filter_fwd(o.x, o.w, o.bias, &d, o.head_anchors_px, o.head_na, o.head_nc, o.head_stride_px, o.io_rows_total,
                                           o.io_row_offset, o.head_filter_conf, o.head_filter_min_wh, o.workspace, o.ws_bytes,
                                           (float*)o.y_aux, s);
          break;
        }
        rc = yolo_head_decode_fwd(o.x, o.w, o.bias, &d, o.head_anchors_px, o.head_na, o.head_nc, o.head_stride_px, (float*)o.y,
                                  o.io_rows_total, o.io_row_offset, (float*)o.y_aux, s);
        break;
      case YOLO_OP_SHUFFLE:
        rc = yolo_channel_shuffle2_fwd(o.x, o.residual, o.y, d.n, d.h, d.w, d.cout, d.cin, d.in_c_total, d.in_c_offset, d.res_c_total,
                                       d.res_c_offset, d.out_c_total, d.out_c_offset, s);
        break;
      case YOLO_OP_CONV_F32:
        rc = yolo_conv2d_f32_fwd((const float*)o.x, (const float*)o.w, o.bias, (const float*)o.residual, (float*)o.y, (float*)o.y_aux, &d, s);
        break;
      case YOLO_OP_MAXPOOL_F32:
        rc = yolo_maxpool_f32_fwd((const float*)o.x, (float*)o.y, d.n, d.h, d.w, d.cin, d.in_c_total, d.in_c_offset, d.ho, d.wo,
                                  d.out_c_total, d.out_c_offset, d.ksize, d.stride, d.pad, d.upsample2x /* dilation */, s);
        break;
      case YOLO_OP_CONV_POOL:
        rc = yolo_conv3x3_pool_fwd(o.x, o.w, o.bias, o.y, &d, 1, s);
        break;
      case YOLO_OP_MBCONV: {
        YoloMbconvDesc m;
        m.n = d.n, m.h = d.h, m.w = d.w, m.cin = d.cin, m.in_c_total = d.in_c_total, m.in_c_offset = d.in_c_offset;
        m.hidden = o.kpad_pre, m.cout = d.cout, m.out_c_total = d.out_c_total, m.out_c_offset = d.out_c_offset;
        m.stride = d.stride, m.has_expand = o.w_pre != nullptr, m.has_res = d.res_c_total != 0, m._pad = 0;
        rc = yolo_mbconv_fwd(o.x, o.w_pre, o.bias_pre, o.w_dw, o.bias_dw, o.w, o.bias, o.y, &m, s);
        break;
      }
      default:
        return yolo_set_error(YOLO_E_ARG, "run_ops: op %d has unknown kind %d", i, o.kind);
    }
    if (rc) return rc;
  }
  return 0;
}

// ---- events and the one-call pipeline step (include/yolo_hip.h: yolo_pipeline_step) ------------------------------------------------
#define YOLO_HIP_TRY(expr, what)                                                                   \
  do {                                                                                             \
    hipError_t e_ = (expr);                                                                        \
    if (e_ != hipSuccess) return yolo_set_error((int)e_, "%s: %s", what, hipGetErrorString(e_));   \
  } while (0)

extern "C" int yolo_event_create(yolo_event_t* out) {
  YOLO_REQUIRE(out, "event_create: null pointer");
  hipEvent_t e = nullptr;
  YOLO_HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming), "hipEventCreateWithFlags");
  *out = (yolo_event_t)e;
  return 0;
}
extern "C" int yolo_event_destroy(yolo_event_t e) {
  YOLO_REQUIRE(e, "event_destroy: null event");
  YOLO_HIP_TRY(hipEventDestroy((hipEvent_t)e), "hipEventDestroy");
  return 0;
}
extern "C" int yolo_event_record(yolo_event_t e, yolo_stream_t s) {
  YOLO_REQUIRE(e, "event_record: null event");
  YOLO_HIP_TRY(hipEventRecord((hipEvent_t)e, (hipStream_t)s), "hipEventRecord");
  return 0;
}
extern "C" int yolo_event_synchronize(yolo_event_t e) {
  YOLO_REQUIRE(e, "event_synchronize: null event");
  YOLO_HIP_TRY(hipEventSynchronize((hipEvent_t)e), "hipEventSynchronize");
  return 0;
}

extern "C" int yolo_pipeline_step(const YoloPipeStep* st) {
  YOLO_REQUIRE(st && st->ops && st->n_ops > 0 && st->k_io >= 0 && st->k_io <= st->n_ops, "pipeline_step: bad op list");
  YOLO_REQUIRE(st->heads_done && st->nms_done, "pipeline_step: heads_done / nms_done events are required");
  hipStream_t s = (hipStream_t)st->stream, ns = (hipStream_t)st->nms_stream;
  if (st->wait_x) YOLO_HIP_TRY(hipStreamWaitEvent(s, (hipEvent_t)st->wait_x, 0), "hipStreamWaitEvent(x)");
  int rc = yolo_run_ops(st->ops, st->k_io, st->stream);
  if (rc) return rc;
  if (st->wait_io) YOLO_HIP_TRY(hipStreamWaitEvent(s, (hipEvent_t)st->wait_io, 0), "hipStreamWaitEvent(io)");
  const bool compact = st->io == nullptr;
  rc = yolo_run_ops(st->ops + st->k_io, st->n_ops - st->k_io, st->stream);
  if (rc) return rc;
  if (ns != s) {
    YOLO_HIP_TRY(hipEventRecord((hipEvent_t)st->heads_done, s), "hipEventRecord(heads)");
    YOLO_HIP_TRY(hipStreamWaitEvent(ns, (hipEvent_t)st->heads_done, 0), "hipStreamWaitEvent(heads)");
  }
  rc = compact ? yolo_nms_merge_compact(st->workspace, st->workspace_bytes, st->bs, st->rows, st->nc, st->nms_thres, st->max_per_class,
                                        st->out_dets, st->out_idx, st->out_count, st->cap, st->nms_stream)
               : yolo_nms_merge(st->io, st->bs, st->rows, st->nc, st->conf_thres, st->nms_thres, st->min_wh, st->max_per_class, 0,
                                st->out_dets, st->out_idx, st->out_count, st->cap, st->workspace, st->workspace_bytes, st->nms_stream);
  if (rc) return rc;
  if (st->count_host)
    YOLO_HIP_TRY(hipMemcpyAsync(st->count_host, st->out_count, (size_t)st->bs * 4, hipMemcpyDeviceToHost, ns), "hipMemcpyAsync(count)");
  YOLO_HIP_TRY(hipEventRecord((hipEvent_t)st->nms_done, ns), "hipEventRecord(nms)");
  if (st->done) YOLO_HIP_TRY(hipEventRecord((hipEvent_t)st->done, ns), "hipEventRecord(done)");
  return 0;
}
