// Pieces of the MERGE-NMS shared by csrc/nms.hip and the head conv's filter epilogue (csrc/conv_igemm.hip): sort keys, the finite test,
// and the layout of the "compact" workspace in which a detect() step never materialises io (include/yolo_hip.h, round 4).
#pragma once
#include "common.h"

namespace yolo_nms {

typedef unsigned long long u64;

constexpr int kLdsKeys = 8192;        // keys sorted in LDS; more survivors -> sort in the global workspace
constexpr int kMaxClasses = 1024;
constexpr int kMaxPerClassCap = 128;  // 2 candidates per lane
constexpr int kRecFloats = 8;         // compact survivor record: x, y, w, h, class_conf, 3 x pad (32 bytes, indexed by io row)

// monotone float -> uint map (ascending), valid for every non-NaN float
__device__ __forceinline__ uint32_t f32_sortable(float f) {
  const uint32_t u = __float_as_uint(f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float f32_unsortable(uint32_t s) {
  return __uint_as_float((s & 0x80000000u) ? (s & 0x7fffffffu) : ~s);
}
// key = class:12 | ~sortable(conf):32 | row:20   -> ascending key == (class asc, conf desc, row asc)
__device__ __forceinline__ u64 make_key(int cls, float conf, int row) {
  return ((u64)cls << 52) | ((u64)(~f32_sortable(conf)) << 20) | (u64)row;
}
__device__ __forceinline__ int key_class(u64 k) { return (int)(k >> 52); }
__device__ __forceinline__ float key_conf(u64 k) { return f32_unsortable(~(uint32_t)(k >> 20)); }
__device__ __forceinline__ int key_row(u64 k) { return (int)(k & 0xfffffu); }

__device__ __forceinline__ bool finite_f(float v) { return (__float_as_uint(v) & 0x7f800000u) != 0x7f800000u; }

inline size_t align256(size_t v) { return (v + 255) & ~(size_t)255; }
inline size_t pow2_at_least(size_t v) {
  size_t p = 1;
  while (p < v) p <<= 1;
  return p;
}
inline int stage_cap_for(int rows, int nc, int max_per_class) {
  long c = (long)nc * max_per_class;
  if (c > rows) c = rows;
  if (c > kLdsKeys) c = kLdsKeys;
  return (int)c;
}

// Workspace of one NMS call: counts int32 [bs] | keys u64 [bs][key_pitch] | stage f32 [bs][stage_cap][8]
//   ( | row_keys u64 [bs][rows] | rec f32 [bs][rows][8]  in the compact form: the head epilogues write ONE key per io row - the sort key
//   of a surviving row, ~0 for the others - and the survivors' box + class score at the row's place; no atomics, nothing to zero:
//   every io row belongs to exactly one head.  nms_merge compacts the row keys into `keys` before it sorts.)
struct Workspace {
  int* counts;
  u64* keys;
  float* stage;
  u64* row_keys;       // compact form only
  float* rec;          // compact form only
  long key_pitch;
  int stage_cap;
  size_t bytes;
};
inline Workspace carve(void* base, int bs, int rows, int nc, bool compact) {
  Workspace w;
  char* p = (char*)base;
  w.counts = (int*)p;
  p += align256((size_t)bs * 4);
  w.key_pitch = (long)pow2_at_least((size_t)rows);
  w.keys = (u64*)p;
  p += align256((size_t)bs * w.key_pitch * 8);
  w.stage_cap = stage_cap_for(rows, nc, kMaxPerClassCap);
  w.stage = (float*)p;
  p += align256((size_t)bs * w.stage_cap * 8 * 4);
  w.row_keys = nullptr;
  w.rec = nullptr;
  if (compact) {
    w.row_keys = (u64*)p;
    p += align256((size_t)bs * rows * 8);
    w.rec = (float*)p;
    p += align256((size_t)bs * rows * kRecFloats * 4);
  }
  w.bytes = (size_t)(p - (char*)base);
  return w;
}

}  // namespace yolo_nms
