// Shared device/host helpers for libyolo_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
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

// bf16 <-> f32 bit helpers (device).  Widening is exact; narrowing uses the hardware RNE cast.
__device__ __forceinline__ float bf16_bits_to_f32(uint32_t hi16) { return __uint_as_float(hi16 << 16); }

// One element of the YOLOLayer decode (reference models/yolo_layer.py:90-96): raw head value r of role k
// (0,1 = x,y; 2,3 = w,h; 4 = objectness; 5.. = classes) at grid cell (gx, gy); anchor = anchor_px / stride for
// this role.  Shared by the standalone decode kernel and the head-conv epilogue so that both give the same bits.
__device__ __forceinline__ float yolo_decode_elem(float r, int k, int gx, int gy, float anchor, float stride, int nc) {
  // one exponential and one reciprocal per element whatever the role (the roles differ per lane inside a wave).  Round 3: the
  // hardware forms (v_exp_f32 on r * log2(e), v_rcp_f32: about 2 ulp together, far inside the 2e-6 the decode is checked to) instead
  // of expf() + an IEEE division, ~6 instead of ~25 VALU instructions per element of 68.5 M per 32 SPP-640 images
  const bool wh = (k & ~1) == 2;
  const float e = __expf(wh ? r : -r);
  const float sg = __builtin_amdgcn_rcpf(1.f + e);                    // sigmoid(r) where it is used
  float v = sg;                                                       // :93
  if (k < 2) v = (sg + (float)(k == 0 ? gx : gy)) * stride;           // :91,:94
  if (wh) v = (e * anchor) * stride;                                  // :92,:94
  if (nc == 1 && k == 5) v = 1.f;                                     // :95-96
  return v;
}

// per-op launch entry points shared with the batched launcher
int yolo_conv2d_launch(const void* x, const void* w, const float* bias, const void* res, void* y, void* y_aux,
                       const YoloConvDesc* d, hipStream_t s);
