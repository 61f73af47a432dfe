// Shared device/host helpers for libyolo_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <atomic>
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include "../../include/yolo_hip.h"

typedef __bf16 bf16_t;
typedef bf16_t bf16x4 __attribute__((ext_vector_type(4)));
typedef bf16_t bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

#define YOLO_WAVE 64

int yolo_set_error(int code, const char* fmt, ...);

#define YOLO_REQUIRE(cond, ...)                                   \
  do {                                                            \
    if (!(cond)) return yolo_set_error(YOLO_E_ARG, __VA_ARGS__);  \
  } while (0)

static inline int yolo_check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return yolo_set_error((int)e, "%s: %s", what, hipGetErrorString(e));
  return 0;
}

// hipFuncAttributeMaxDynamicSharedMemorySize belongs to the CURRENT device's code object: a launcher raises it once per
// (kernel, device), not once per process (one process may drive several GPUs: engine.Plan carries a device).  `done` is the
// launcher's own bit mask of devices that have it (function-local static per kernel instantiation); setting the attribute twice
// from two threads is harmless.
static inline int yolo_max_dyn_lds(const void* fn, int bytes, std::atomic<uint64_t>& done, const char* what) {
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return yolo_set_error((int)e, "%s: hipGetDevice: %s", what, hipGetErrorString(e));
  const uint64_t bit = 1ull << (dev & 63);
  if (done.load(std::memory_order_acquire) & bit) return 0;
  e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
  if (e != hipSuccess) return yolo_set_error((int)e, "%s: hipFuncSetAttribute: %s", what, hipGetErrorString(e));
  done.fetch_or(bit, std::memory_order_release);
  return 0;
}

// bf16 <-> f32 bit helpers (device).  Widening is exact; narrowing uses the hardware RNE cast.
__device__ __forceinline__ float bf16_bits_to_f32(uint32_t hi16) { return __uint_as_float(hi16 << 16); }

// One element of the YOLOLayer decode (reference models/yolo_layer.py:90-96): raw head value r of role k
// (0,1 = x,y; 2,3 = w,h; 4 = objectness; 5.. = classes) at grid cell (gx, gy); anchor = anchor_px / stride for
// this role.  One exponential and one reciprocal per element whatever the role (the roles differ per lane inside a wave).
//   PRECISE = true : expf() + an IEEE division (each <= 1 ulp).  The standalone decode kernel (yolo_decode_fwd), i.e. the
//                    decode of model.precision = "fp32" - the parity mode whose kept-index sets are compared with the reference's.
//   PRECISE = false: v_exp_f32 on r * log2(e) and v_rcp_f32 (~6 instead of ~25 VALU instructions for each of the 68.5 M head
//                    elements of 32 SPP-640 images): the head conv's epilogue (yolo_head_decode_fwd), i.e. the bf16 path.  The
//                    product r * log2(e) is rounded once, so exp's relative error is about (1.5 + |r|) * 2^-24: <= 1e-6
//                    for |r| <= 15 and 2e-6 at |r| = 30 (w / h = exp(r) * anchor; where the sigmoid is used the ABSOLUTE error stays
//                    below 2^-23 for every r).  tests: test_decode_vs_oracle (PRECISE), test_fused_head_decode_wide_logits (fast).
template <bool PRECISE>
__device__ __forceinline__ float yolo_decode_elem(float r, int k, int gx, int gy, float anchor, float stride, int nc) {
  const bool wh = (k & ~1) == 2;
  const float e = PRECISE ? expf(wh ? r : -r) : __expf(wh ? r : -r);
  const float sg = PRECISE ? 1.f / (1.f + e) : __builtin_amdgcn_rcpf(1.f + e);   // sigmoid(r) where it is used
  float v = sg;                                                       // :93
  if (k < 2) v = (sg + (float)(k == 0 ? gx : gy)) * stride;           // :91,:94
  if (wh) v = (e * anchor) * stride;                                  // :92,:94
  if (nc == 1 && k == 5) v = 1.f;                                     // :95-96
  return v;
}

// per-op launch entry points shared with the batched launcher
int yolo_conv2d_launch(const void* x, const void* w, const float* bias, const void* res, void* y, void* y_aux,
                       const YoloConvDesc* d, hipStream_t s);
