// Shared device/host helpers for libyolo_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>

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

// per-op launch entry points shared with the batched launcher
int yolo_conv2d_launch(const void* x, const void* w, const float* bias, const void* res, void* y, void* y_aux,
                       const YoloConvDesc* d, hipStream_t s);
