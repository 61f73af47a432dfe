// HBM-bound element kernels of the path (gfx950): input packing, max pools / SPP pyramid,
// depthwise 3x3, YOLO-layer decode.  All are coalesced 16-byte-per-lane streams over NHWC data.
#include "common.h"

namespace {

// ------------------------------------------------------------------------------------------------
// NCHW f32 -> NHWC bf16, channels padded with zeros (first-layer layout).
__global__ __launch_bounds__(256) void pack_input_kernel(const float* __restrict__ x, bf16_t* __restrict__ y, int c,
                                                         long hw, long total_pix, int c_pad) {
  const long p = (long)blockIdx.x * 256 + threadIdx.x;
  if (p >= total_pix) return;
  const long b = p / hw, sp = p - b * hw;
  const float* src = x + b * c * hw + sp;
  bf16_t* dst = y + p * c_pad;
  for (int c0 = 0; c0 < c_pad; c0 += 8) {
    bf16x8 o;
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = (bf16_t)((c0 + e < c) ? src[(long)(c0 + e) * hw] : 0.f);
    *reinterpret_cast<bf16x8*>(dst + c0) = o;
  }
}

// ------------------------------------------------------------------------------------------------
// bf16x8 lane-wise max (bf16 -> f32 widening is exact, so comparing as f32 and keeping the bits is exact)
__device__ __forceinline__ void max8(float (&m)[8], const u32x4 v) {
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    m[2 * e] = fmaxf(m[2 * e], __uint_as_float(v[e] << 16));
    m[2 * e + 1] = fmaxf(m[2 * e + 1], __uint_as_float(v[e] & 0xffff0000u));
  }
}
__device__ __forceinline__ u32x4 pack8(const float (&m)[8]) {
  u32x4 o;
#pragma unroll
  for (int e = 0; e < 4; ++e) o[e] = (__float_as_uint(m[2 * e]) >> 16) | (__float_as_uint(m[2 * e + 1]) & 0xffff0000u);
  return o;
}

// Generic max pool, -inf padding (nn.MaxPool2d semantics), one thread = 8 channels of one output pixel.
__global__ __launch_bounds__(256) void maxpool_kernel(const bf16_t* __restrict__ x, bf16_t* __restrict__ y, int h, int w,
                                                      int cg, int in_ct, int in_co, int ho, int wo, int out_ct, int out_co,
                                                      int k, int stride, int pad, int dil, long total) {
  const long t = (long)blockIdx.x * 256 + threadIdx.x;
  if (t >= total) return;
  const int g = (int)(t % cg);
  long p = t / cg;
  const int ow = (int)(p % wo);
  p /= wo;
  const int oh = (int)(p % ho);
  const long b = p / ho;
  float m[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) m[e] = -INFINITY;
  for (int i = 0; i < k; ++i) {
    const int hi = oh * stride - pad + i * dil;
    if ((unsigned)hi >= (unsigned)h) continue;
    for (int j = 0; j < k; ++j) {
      const int wi = ow * stride - pad + j * dil;
      if ((unsigned)wi >= (unsigned)w) continue;
      max8(m, *reinterpret_cast<const u32x4*>(x + ((b * h + hi) * w + wi) * in_ct + in_co + g * 8));
    }
  }
  *reinterpret_cast<u32x4*>(y + ((b * ho + oh) * wo + ow) * out_ct + out_co + g * 8) = pack8(m);
}

// SPP pyramid: one block = one image x 8 channels; the hxw plane lives in LDS; separable running max
// (rows then columns) gives 5/9/13 windows in one pass.  Reads slice [3c,4c) of the concat buffer,
// writes slices [0,c) [c,2c) [2c,3c).
// bf16 pairs as order-preserving int16 pairs (x ^ 0x7fff for negative halves; an involution): the 5/9/13 window maxima
// then cost ONE v_pk_max_i16 per 32-bit word and tap instead of unpack + two fmax
typedef short s16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ u32x4 bf16_sortable(u32x4 v) {
  return v ^ (((v >> 15) & 0x00010001u) * 0x7fffu);
}
__device__ __forceinline__ u32x4 pk_max(u32x4 a, u32x4 b) {
  return __builtin_bit_cast(u32x4, __builtin_elementwise_max(__builtin_bit_cast(s16x8, a), __builtin_bit_cast(s16x8, b)));
}

__global__ __launch_bounds__(256) void spp_kernel(bf16_t* __restrict__ buf, int h, int w, int c) {
  extern __shared__ __attribute__((aligned(16))) char lds_raw[];
  u32x4* plane = reinterpret_cast<u32x4*>(lds_raw);  // [h*w] input, then 3 x [h*w] row-maxima (all in sortable form)
  const int hw = h * w, ct = 4 * c;
  const int cg = c / 8;
  const long b = blockIdx.x / cg;
  const int g = blockIdx.x % cg;
  bf16_t* base = buf + b * hw * ct + g * 8;
  for (int p = threadIdx.x; p < hw; p += 256) plane[p] = bf16_sortable(*reinterpret_cast<const u32x4*>(base + (long)p * ct + 3 * c));
  __syncthreads();
  for (int p = threadIdx.x; p < hw; p += 256) {
    const int yy = p / w, xx = p - yy * w;
    u32x4 m = plane[p];
    for (int r = 1; r <= 6; ++r) {
      if (xx - r >= 0) m = pk_max(m, plane[p - r]);
      if (xx + r < w) m = pk_max(m, plane[p + r]);
      if (r == 2) plane[hw + p] = m;
      if (r == 4) plane[2 * hw + p] = m;
    }
    plane[3 * hw + p] = m;
  }
  __syncthreads();
  for (int p = threadIdx.x; p < hw; p += 256) {
    const int yy = p / w;
#pragma unroll
    for (int lvl = 0; lvl < 3; ++lvl) {
      const int rad = 2 + 2 * lvl;
      const u32x4* rows = plane + (lvl + 1) * hw;
      u32x4 m = rows[p];
      for (int r = 1; r <= rad; ++r) {
        if (yy - r >= 0) m = pk_max(m, rows[p - r * w]);
        if (yy + r < h) m = pk_max(m, rows[p + r * w]);
      }
      *reinterpret_cast<u32x4*>(base + (long)p * ct + lvl * c) = bf16_sortable(m);
    }
  }
}

// Line-wide form for c % 64 == 0 (the shipped SPP: 512 channels on 20x20): one block = one image x 64 channels, i.e. whole
// 128-byte lines of the NHWC buffer per pixel both ways (the 8-channel form above touches 16 of a line's 128 bytes per access:
// rocprofv3 counted 100 MB fetched / 75 MB written for 13 + 39 MB).  Two LDS planes of hw x 128 B only; the radii grow from one
// another, r4(x) = max(r2(x-2), r2(x+2)) and r6 from r4 likewise (positions clamped to the row: a clamped window stays inside the
// true one and the pair still covers it), so a level's row maxima overwrite the plane the previous level no longer reads:
//   B = r2(A) | out5 = V2(B), A = r4(B) | out9 = V4(A), B = r6(A) | out13 = V6(B)           (| = barrier)
__global__ __launch_bounds__(512) void spp_lines_kernel(bf16_t* __restrict__ buf, int h, int w, int c) {
  extern __shared__ __attribute__((aligned(16))) char lds_raw[];
  const int hw = h * w, ct = 4 * c, cg = c / 64;
  u32x4* pa = reinterpret_cast<u32x4*>(lds_raw);     // [hw][8 chunks of 8 channels]
  u32x4* pb = pa + hw * 8;
  const long b = blockIdx.x / cg;
  const int g = blockIdx.x % cg;
  bf16_t* base = buf + b * hw * ct + g * 64;
  const int items = hw * 8;
  for (int i = threadIdx.x; i < items; i += 512)
    pa[i] = bf16_sortable(*reinterpret_cast<const u32x4*>(base + (long)(i >> 3) * ct + 3 * c + (i & 7) * 8));
  __syncthreads();
  for (int i = threadIdx.x; i < items; i += 512) {
    const int p = i >> 3, xx = p % w;
    u32x4 m = pa[i];
#pragma unroll
    for (int r = 1; r <= 2; ++r) {
      if (xx - r >= 0) m = pk_max(m, pa[i - r * 8]);
      if (xx + r < w) m = pk_max(m, pa[i + r * 8]);
    }
    pb[i] = m;
  }
  __syncthreads();
  auto level = [&](const u32x4* rows, u32x4* next, int rad, int slot) {
    for (int i = threadIdx.x; i < items; i += 512) {
      const int p = i >> 3, yy = p / w, xx = p - yy * w;
      u32x4 m = rows[i];
      for (int r = 1; r <= rad; ++r) {
        if (yy - r >= 0) m = pk_max(m, rows[i - r * w * 8]);
        if (yy + r < h) m = pk_max(m, rows[i + r * w * 8]);
      }
      *reinterpret_cast<u32x4*>(base + (long)p * ct + slot * c + (i & 7) * 8) = bf16_sortable(m);
      if (next) {
        const int lo = xx - 2 < 0 ? -xx : -2, hi = xx + 2 >= w ? w - 1 - xx : 2;
        next[i] = pk_max(rows[i + lo * 8], rows[i + hi * 8]);
      }
    }
  };
  level(pb, pa, 2, 0);
  __syncthreads();
  level(pa, pb, 4, 1);
  __syncthreads();
  level(pb, nullptr, 6, 2);
}

// ------------------------------------------------------------------------------------------------
// Depthwise 3x3 (pad 1) + bias + activation, one thread = 8 channels of one output pixel, fp32 math.  (Reference form:
// yolo_dwconv3x3_fwd launches the strip kernel below; YOLO_DWCONV_DEBUG=1 selects this one.)
__global__ __launch_bounds__(256) void dwconv3x3_kernel(const bf16_t* __restrict__ x, const float* __restrict__ wt,
                                                        const float* __restrict__ bias, bf16_t* __restrict__ y, int h, int w,
                                                        int c, int in_ct, int in_co, int ho, int wo, int out_ct, int out_co,
                                                        int stride, int act, long total) {
  const long t = (long)blockIdx.x * 256 + threadIdx.x;
  if (t >= total) return;
  const int cg = c / 8;
  const int g = (int)(t % cg);
  long p = t / cg;
  const int ow = (int)(p % wo);
  p /= wo;
  const int oh = (int)(p % ho);
  const long b = p / ho;
  float acc[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) acc[e] = bias[g * 8 + e];
  for (int i = 0; i < 3; ++i) {
    const int hi = oh * stride - 1 + i;
    if ((unsigned)hi >= (unsigned)h) continue;
    for (int j = 0; j < 3; ++j) {
      const int wi = ow * stride - 1 + j;
      if ((unsigned)wi >= (unsigned)w) continue;
      const u32x4 v = *reinterpret_cast<const u32x4*>(x + ((b * h + hi) * w + wi) * in_ct + in_co + g * 8);
      const float* wp = wt + (i * 3 + j) * c + g * 8;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        acc[2 * e] = fmaf(__uint_as_float(v[e] << 16), wp[2 * e], acc[2 * e]);
        acc[2 * e + 1] = fmaf(__uint_as_float(v[e] & 0xffff0000u), wp[2 * e + 1], acc[2 * e + 1]);
      }
    }
  }
  bf16x8 o;
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    float v = acc[e];
    if (act == YOLO_ACT_LEAKY01) v = v > 0.f ? v : 0.1f * v;
    if (act == YOLO_ACT_RELU6) v = fminf(fmaxf(v, 0.f), 6.f);
    o[e] = (bf16_t)v;
  }
  *reinterpret_cast<bf16x8*>(y + ((b * ho + oh) * wo + ow) * out_ct + out_co + g * 8) = o;
}

// ------------------------------------------------------------------------------------------------
// YOLOLayer decode (reference models/yolo_layer.py:57-69,90-111).  Thread t <-> element t of
// p[bs,na,ny,nx,5+nc] (so the p and io stores are fully contiguous); the head read is contiguous
// within each (5+nc)-run because the head is NHWC with channel a*(5+nc)+k.
struct DecodeArgs {
  const float* head;
  float* io;
  float* p;
  int head_ct, na, nc, ny, nx, io_rows_total, io_row_offset;
  float stride;
  float anchor_w[8], anchor_h[8];  // anchors / stride (yolo_layer.py:109)
  long total;
};

// Thread = one head channel c = anchor*(5+nc) + k (constant over the block's pixels, so the anchor / role
// split costs nothing per element); a block walks kDecodePix consecutive pixels.  Reads are 1 KiB runs per
// pixel, the p and io stores 340-byte runs per (pixel, anchor) that continue with the next pixel.
constexpr int kDecodePix = 16;

__global__ __launch_bounds__(256) void decode_kernel(const DecodeArgs a) {
  const int no = a.nc + 5;
  for (int c = threadIdx.x; c < a.na * no; c += 256) {   // one pass for na*(5+nc) <= 256 (nc <= 80)
  const int an = c / no, k = c - an * no;
  const float anchor = k == 2 ? a.anchor_w[an] : a.anchor_h[an];
  const int cells = a.ny * a.nx;
  const long total_pix = (long)a.total;                 // bs * ny * nx
  const long pix0 = (long)blockIdx.x * kDecodePix;
  float raw[kDecodePix];
#pragma unroll
  for (int i = 0; i < kDecodePix; ++i)
    raw[i] = pix0 + i < total_pix ? a.head[(pix0 + i) * a.head_ct + c] : 0.f;
#pragma unroll
  for (int i = 0; i < kDecodePix; ++i) {
    const long pix = pix0 + i;
    if (pix >= total_pix) break;
    const int b = (int)(pix / cells), cell = (int)(pix - (long)b * cells);
    const int gy = cell / a.nx, gx = cell - gy * a.nx;
    const float r = raw[i];
    const float v = yolo_decode_elem<true>(r, k, gx, gy, anchor, a.stride, a.nc);
    const long plane_elem = (long)cell * no + k;
    if (a.p) a.p[((long)b * a.na + an) * cells * no + plane_elem] = r;
    a.io[((long)b * a.io_rows_total + a.io_row_offset + (long)an * cells) * no + plane_elem] = v;
  }
  }
}

inline unsigned blocks_for(long total) { return (unsigned)((total + 255) / 256); }

}  // namespace

extern "C" int yolo_pack_input_nchw_f32(const float* x, void* y, int n, int c, int h, int w, int c_pad, yolo_stream_t s) {
  YOLO_REQUIRE(x && y && n > 0 && c > 0 && h > 0 && w > 0, "pack_input: bad arguments");
  YOLO_REQUIRE(c_pad % 8 == 0 && c_pad >= c, "pack_input: c_pad %d must be a multiple of 8 and >= c %d", c_pad, c);
  const long total = (long)n * h * w;
  hipLaunchKernelGGL(pack_input_kernel, dim3(blocks_for(total)), dim3(256), 0, (hipStream_t)s, x, (bf16_t*)y, c,
                     (long)h * w, total, c_pad);
  return yolo_check_launch("yolo_pack_input_nchw_f32");
}

extern "C" int yolo_maxpool_fwd(const void* x, void* y, int n, int h, int w, int c, int in_c_total, int in_c_offset, int ho,
                                int wo, int out_c_total, int out_c_offset, int ksize, int stride, int pad, int dilation,
                                yolo_stream_t s) {
  YOLO_REQUIRE(x && y && n > 0 && c > 0 && c % 8 == 0, "maxpool: bad arguments (c %d must be a multiple of 8)", c);
  YOLO_REQUIRE(in_c_total % 8 == 0 && in_c_offset % 8 == 0 && out_c_total % 8 == 0 && out_c_offset % 8 == 0,
               "maxpool: views must be 8-channel aligned");
  YOLO_REQUIRE(ksize >= 1 && stride >= 1 && dilation >= 1 && pad >= 0, "maxpool: bad geometry");
  // floor mode, or torch's ceil_mode (one more row / column whose window still starts inside the input or its left
  // padding; the taps beyond the input count as -inf like the padding does)
  const int hf = (h + 2 * pad - dilation * (ksize - 1) - 1) / stride + 1, wf = (w + 2 * pad - dilation * (ksize - 1) - 1) / stride + 1;
  YOLO_REQUIRE((ho == hf || (ho == hf + 1 && (ho - 1) * stride < h + pad)) && (wo == wf || (wo == wf + 1 && (wo - 1) * stride < w + pad)),
               "maxpool: output %dx%d inconsistent with %dx%d k%d s%d p%d d%d", ho, wo, h, w, ksize, stride, pad, dilation);
  const long total = (long)n * ho * wo * (c / 8);
  hipLaunchKernelGGL(maxpool_kernel, dim3(blocks_for(total)), dim3(256), 0, (hipStream_t)s, (const bf16_t*)x, (bf16_t*)y, h,
                     w, c / 8, in_c_total, in_c_offset, ho, wo, out_c_total, out_c_offset, ksize, stride, pad, dilation, total);
  return yolo_check_launch("yolo_maxpool_fwd");
}

// channel_shuffle(cat(a, b), 2) in the two-slot layout (see include/yolo_hip.h): thread = 8 physical output channels of a pixel
__global__ __launch_bounds__(256) void shuffle2_kernel(const bf16_t* __restrict__ a, const bf16_t* __restrict__ b,
                                                       bf16_t* __restrict__ y, int half, int c_slot, int a_ct, int a_co, int b_ct,
                                                       int b_co, int y_ct, int y_co, long total) {
  const long t = (long)blockIdx.x * 256 + threadIdx.x;
  if (t >= total) return;
  const int groups = 2 * c_slot / 8;
  const int g = (int)(t % groups);
  const long pix = t / groups;
  const bf16_t* const pa = a + pix * a_ct + a_co;
  const bf16_t* const pb = b + pix * b_ct + b_co;
  bf16x8 o;
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int pc = g * 8 + e, slot = pc / c_slot, r = pc - slot * c_slot;
    const int j = slot * half + r;                    // logical output channel
    o[e] = r < half ? ((j & 1) ? pb[j >> 1] : pa[j >> 1]) : (bf16_t)0.f;
  }
  *reinterpret_cast<bf16x8*>(y + pix * y_ct + y_co + g * 8) = o;
}

extern "C" int yolo_channel_shuffle2_fwd(const void* a, const void* b, void* y, int n, int h, int w, int half, int c_slot,
                                         int a_c_total, int a_c_offset, int b_c_total, int b_c_offset, int y_c_total,
                                         int y_c_offset, yolo_stream_t s) {
  YOLO_REQUIRE(a && b && y && n > 0 && h > 0 && w > 0, "shuffle2: bad arguments");
  YOLO_REQUIRE(half >= 1 && half <= c_slot && c_slot % 8 == 0, "shuffle2: %d logical channels per slot of %d", half, c_slot);
  YOLO_REQUIRE(a_c_offset + c_slot <= a_c_total && b_c_offset + c_slot <= b_c_total && y_c_offset + 2 * c_slot <= y_c_total &&
                   y_c_offset % 8 == 0 && y_c_total % 8 == 0,
               "shuffle2: bad views");
  const long total = (long)n * h * w * (2 * c_slot / 8);
  hipLaunchKernelGGL(shuffle2_kernel, dim3(blocks_for(total)), dim3(256), 0, (hipStream_t)s, (const bf16_t*)a, (const bf16_t*)b,
                     (bf16_t*)y, half, c_slot, a_c_total, a_c_offset, b_c_total, b_c_offset, y_c_total, y_c_offset, total);
  return yolo_check_launch("yolo_channel_shuffle2_fwd");
}

extern "C" int yolo_spp_fwd(void* buf, int n, int h, int w, int c, yolo_stream_t s) {
  YOLO_REQUIRE(buf && n > 0 && h > 0 && w > 0 && c > 0 && c % 8 == 0, "spp: bad arguments");
  const size_t lds_lines = (size_t)2 * h * w * 128;
  static const bool no_lines = getenv("YOLO_SPP_NO_LINES") != nullptr;     // A/B knob: the 8-channel form for every shape
  if (c % 64 == 0 && lds_lines <= 128 * 1024 && !no_lines) {
    static std::atomic<uint64_t> lds_set{0};               // per device (common.h)
    if (const int rc = yolo_max_dyn_lds((const void*)spp_lines_kernel, 128 * 1024, lds_set, "spp")) return rc;
    hipLaunchKernelGGL(spp_lines_kernel, dim3((unsigned)(n * (c / 64))), dim3(512), lds_lines, (hipStream_t)s, (bf16_t*)buf, h, w, c);
    return yolo_check_launch("yolo_spp_fwd");
  }
  const size_t lds = (size_t)4 * h * w * 16;
  if (lds <= 64 * 1024) {
    hipLaunchKernelGGL(spp_kernel, dim3((unsigned)(n * (c / 8))), dim3(256), lds, (hipStream_t)s, (bf16_t*)buf, h, w, c);
    return yolo_check_launch("yolo_spp_fwd");
  }
  // large feature maps: three direct pools (same results, more L2 traffic)
  for (int lvl = 0; lvl < 3; ++lvl) {
    const int k = 5 + 4 * lvl;
    int rc = yolo_maxpool_fwd(buf, buf, n, h, w, c, 4 * c, 3 * c, h, w, 4 * c, lvl * c, k, 1, k / 2, 1, s);
    if (rc) return rc;
  }
  return 0;
}

// Strip form: one thread = 8 channels of R vertically adjacent output pixels.  Its 72 weights are read once and stay in
// registers (the one-pixel form re-reads them for every output: 18 of its 27 loads), and the three input rows of an
// output slide down the strip, so an output costs 3 (stride 1) or 6 (stride 2) new 16-byte loads instead of 9.
// Same fp32 operation order as the one-pixel form (bias, then taps row by row), so the results are identical.
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int S, int R>
__global__ __launch_bounds__(256) void dwconv3x3_strip_kernel(const bf16_t* __restrict__ x, const float* __restrict__ wt,
                                                              const float* __restrict__ bias, bf16_t* __restrict__ y, int h,
                                                              int w, int c, int in_ct, int in_co, int ho, int wo, int out_ct,
                                                              int out_co, int act, int strips, long total) {
  const long t = (long)blockIdx.x * 256 + threadIdx.x;
  if (t >= total) return;
  const int cg = c / 8;
  const int g = (int)(t % cg);
  long p = t / cg;
  const int ow = (int)(p % wo);
  p /= wo;
  const int oh0 = (int)(p % strips) * R;
  const long b = p / strips;
  f32x2 wv[9][4], bv[4];
#pragma unroll
  for (int k = 0; k < 9; ++k) {
    const f32x4 lo = *reinterpret_cast<const f32x4*>(wt + k * c + g * 8), hi = *reinterpret_cast<const f32x4*>(wt + k * c + g * 8 + 4);
    wv[k][0] = f32x2{lo[0], lo[1]}, wv[k][1] = f32x2{lo[2], lo[3]}, wv[k][2] = f32x2{hi[0], hi[1]}, wv[k][3] = f32x2{hi[2], hi[3]};
  }
  {
    const f32x4 lo = *reinterpret_cast<const f32x4*>(bias + g * 8), hi = *reinterpret_cast<const f32x4*>(bias + g * 8 + 4);
    bv[0] = f32x2{lo[0], lo[1]}, bv[1] = f32x2{lo[2], lo[3]}, bv[2] = f32x2{hi[0], hi[1]}, bv[3] = f32x2{hi[2], hi[3]};
  }
  const bf16_t* const xb = x + (b * h) * (long)w * in_ct + in_co + g * 8;
  const int wi0 = ow * S - 1;
  auto load_row = [&](int hi, u32x4 (&row)[3]) {
    const bool rok = (unsigned)hi < (unsigned)h;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const int wi = wi0 + j;
      row[j] = (rok && (unsigned)wi < (unsigned)w) ? *reinterpret_cast<const u32x4*>(xb + ((long)hi * w + wi) * in_ct) : u32x4{0, 0, 0, 0};
    }
  };
  // input rows of output row oh0 + k: (oh0 + k) * S - 1 + {0, 1, 2}
  u32x4 rows[3][3];
  load_row(oh0 * S - 1, rows[0]);
  if (S == 1) load_row(oh0, rows[1]);
#pragma unroll
  for (int k = 0; k < R; ++k) {
    const int oh = oh0 + k;
    if (oh >= ho) break;
    // slots: at stride 1 the window slides by one row, at stride 2 by two (the last row becomes the first)
    constexpr int NS = 3;
    const int s0 = (S == 1 ? k : 2 * k) % NS, s1 = (s0 + 1) % NS, s2 = (s0 + 2) % NS;
    if (S == 1) {
      load_row(oh + 1, rows[s2]);
    } else {
      load_row(oh * 2, rows[s1]);
      load_row(oh * 2 + 1, rows[s2]);
    }
    f32x2 acc[4] = {bv[0], bv[1], bv[2], bv[3]};
    const int sl[3] = {s0, s1, s2};
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        const u32x4 v = rows[sl[i]][j];
#pragma unroll
        for (int e = 0; e < 4; ++e)
          acc[e] = __builtin_elementwise_fma(f32x2{__uint_as_float(v[e] << 16), __uint_as_float(v[e] & 0xffff0000u)}, wv[i * 3 + j][e], acc[e]);
      }
    bf16x8 o;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      float v = acc[e >> 1][e & 1];
      if (act == YOLO_ACT_LEAKY01) v = v > 0.f ? v : 0.1f * v;
      if (act == YOLO_ACT_RELU6) v = fminf(fmaxf(v, 0.f), 6.f);
      o[e] = (bf16_t)v;
    }
    *reinterpret_cast<bf16x8*>(y + ((b * ho + oh) * wo + ow) * out_ct + out_co + g * 8) = o;
  }
}

template <int S, int R>
static int launch_dw_strip(const bf16_t* x, const float* w, const float* bias, bf16_t* y, int n, int h, int w_, int c, int in_ct,
                           int in_co, int ho, int wo, int out_ct, int out_co, int act, hipStream_t s) {
  const int strips = (ho + R - 1) / R;
  const long total = (long)n * strips * wo * (c / 8);
  hipLaunchKernelGGL((dwconv3x3_strip_kernel<S, R>), dim3(blocks_for(total)), dim3(256), 0, s, x, w, bias, y, h, w_, c, in_ct, in_co,
                     ho, wo, out_ct, out_co, act, strips, total);
  return yolo_check_launch("yolo_dwconv3x3_fwd");
}

static const int dw_debug = [] {      // YOLO_DWCONV_DEBUG: 1 = one-pixel form, 2 = strips of 4 rows instead of 8 (tuning only)
  const char* e = getenv("YOLO_DWCONV_DEBUG");
  return e ? atoi(e) : 0;
}();

extern "C" int yolo_dwconv3x3_fwd(const void* x, const float* w, const float* bias, void* y, int n, int h, int w_, int c,
                                  int in_c_total, int in_c_offset, int ho, int wo, int out_c_total, int out_c_offset,
                                  int stride, int act, yolo_stream_t s) {
  YOLO_REQUIRE(x && w && bias && y && n > 0 && c > 0 && c % 8 == 0, "dwconv: bad arguments");
  YOLO_REQUIRE(stride == 1 || stride == 2, "dwconv: stride %d", stride);
  YOLO_REQUIRE(ho == (h + 2 - 3) / stride + 1 && wo == (w_ + 2 - 3) / stride + 1, "dwconv: bad output size");
  YOLO_REQUIRE(in_c_total % 8 == 0 && in_c_offset % 8 == 0 && out_c_total % 8 == 0 && out_c_offset % 8 == 0,
               "dwconv: views must be 8-channel aligned");
  if (!(dw_debug & 1)) {
    const bf16_t* xb = (const bf16_t*)x;
    bf16_t* yb = (bf16_t*)y;
    hipStream_t st = (hipStream_t)s;
    if (dw_debug & 2)
      return stride == 1 ? launch_dw_strip<1, 4>(xb, w, bias, yb, n, h, w_, c, in_c_total, in_c_offset, ho, wo, out_c_total, out_c_offset, act, st)
                         : launch_dw_strip<2, 4>(xb, w, bias, yb, n, h, w_, c, in_c_total, in_c_offset, ho, wo, out_c_total, out_c_offset, act, st);
    // 8-row strips: 0.040 -> 0.028 ms on the 26x26x384 maps of 64 images (4-row strips 0.032), also on 13x13 (13 -> 16 rows)
    return stride == 1 ? launch_dw_strip<1, 8>(xb, w, bias, yb, n, h, w_, c, in_c_total, in_c_offset, ho, wo, out_c_total, out_c_offset, act, st)
                       : launch_dw_strip<2, 8>(xb, w, bias, yb, n, h, w_, c, in_c_total, in_c_offset, ho, wo, out_c_total, out_c_offset, act, st);
  }
  const long total = (long)n * ho * wo * (c / 8);
  hipLaunchKernelGGL(dwconv3x3_kernel, dim3(blocks_for(total)), dim3(256), 0, (hipStream_t)s, (const bf16_t*)x, w, bias,
                     (bf16_t*)y, h, w_, c, in_c_total, in_c_offset, ho, wo, out_c_total, out_c_offset, stride, act, total);
  return yolo_check_launch("yolo_dwconv3x3_fwd");
}

extern "C" int yolo_decode_fwd(const float* head, int head_c_total, const float* anchors_px, int na, int nc, int bs, int ny,
                               int nx, float stride_px, float* io, int io_rows_total, int io_row_offset, float* p,
                               yolo_stream_t s) {
  YOLO_REQUIRE(head && anchors_px && io, "decode: null pointer");
  YOLO_REQUIRE(na >= 1 && na <= 8 && nc >= 1 && bs > 0 && ny > 0 && nx > 0, "decode: bad sizes");
  YOLO_REQUIRE(head_c_total >= na * (5 + nc), "decode: head_c_total %d < %d", head_c_total, na * (5 + nc));
  YOLO_REQUIRE(io_row_offset >= 0 && io_row_offset + na * ny * nx <= io_rows_total, "decode: rows out of range");
  DecodeArgs a;
  a.head = head;
  a.io = io;
  a.p = p;
  a.head_ct = head_c_total;
  a.na = na;
  a.nc = nc;
  a.ny = ny;
  a.nx = nx;
  a.io_rows_total = io_rows_total;
  a.io_row_offset = io_row_offset;
  a.stride = stride_px;
  for (int i = 0; i < na; ++i) {
    a.anchor_w[i] = anchors_px[2 * i] / stride_px;
    a.anchor_h[i] = anchors_px[2 * i + 1] / stride_px;
  }
  a.total = (long)bs * ny * nx;                          // pixels
  const long blocks = (a.total + kDecodePix - 1) / kDecodePix;
  YOLO_REQUIRE(blocks < 0x7fffffffL, "decode: head too large");
  hipLaunchKernelGGL(decode_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)s, a);
  return yolo_check_launch("yolo_decode_fwd");
}
