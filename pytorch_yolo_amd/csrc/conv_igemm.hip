// Implicit-GEMM convolution for gfx950 (MI355X): bf16 NHWC activations, fp32 accumulate on
// v_mfma_f32_32x32x16_bf16, fused bias / LeakyReLU(0.1) / ReLU6 / residual add / pre-add copy /
// channel-offset (concat) store / 2x nearest-upsample store / fp32 head store.
//
// Replaces the ATen dispatches of ConvBlock.forward (reference models/yolo_base.py:19-44, BN folded
// per utils/torch_utils.py:33-60), Add (models/yolov3_spp.py:12-14), Upsample (models/yolo_layer.py:6-13)
// and the Concat placement (models/yolo_layer.py:16-22).
//
// GEMM view:  D[cout][pixel] = sum_k W[cout][k] * X[k][pixel],   k = (kh*ks + kw)*cin + c.
// The weights are the MFMA "A" operand and the pixels the "B" operand, so an accumulator lane owns
// one pixel and 4 consecutive output channels per register group -> 8-byte NHWC stores.
//
// Tiling: 256 threads = 4 waves; block tile BM pixels x BN couts, K step 32 (four 16-byte chunks per row).
// Both operand tiles are staged global -> LDS with `buffer_load_dwordx4 ... lds` (LDS-DMA): the LDS image
// is lane-linear (64-byte rows), conflict-free ds_read_b128 comes from an XOR swizzle applied on the
// per-lane SOURCE address (chunk ^= (row>>2)&3) and again on the read.  Zero padding (image border,
// K tail, M tail) costs nothing: those lanes get a voffset beyond the descriptor's num_records and the
// hardware writes zeros.  Double-buffered, one barrier per K step.
#include "common.h"

namespace {

constexpr uint32_t kOobOffset = 0xF0000000u;  // > any buffer we accept (host checks < 0xF0000000 bytes)

struct ConvArgs {
  const bf16_t* x;
  const bf16_t* w;
  const float* bias;
  const bf16_t* res;
  void* y;
  bf16_t* aux;
  YoloConvDesc d;
  int M;        // n*ho*wo
  int n_tiles;  // cout tiles
  int steps;    // kpad / 32
  uint32_t x_bytes, w_bytes;
};

__device__ __forceinline__ void lds_dma16(__amdgpu_buffer_rsrc_t rsrc, char* lds_wave_base, uint32_t voffset) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)lds_wave_base, 16,
                                           voffset, 0, 0, 0);
}

__device__ __forceinline__ float apply_act(float v, int act) {
  if (act == YOLO_ACT_LEAKY01) return v > 0.f ? v : 0.1f * v;
  if (act == YOLO_ACT_RELU6) return fminf(fmaxf(v, 0.f), 6.f);
  return v;
}

template <int BM, int BN, int WAVES_M, int WAVES_N>
__global__ __launch_bounds__(256) void conv_igemm_bf16_kernel(const ConvArgs a) {
  static_assert(WAVES_M * WAVES_N == 4, "4 waves");
  constexpr int TM = BM / WAVES_M, TN = BN / WAVES_N;  // per-wave tile (pixels x couts)
  constexpr int NI = TM / 32, MI = TN / 32;
  constexpr int PIT = BM / 64, WIT = (BN + 63) / 64;   // LDS-DMA instructions per thread per step
  constexpr int ROWB = 64;                             // bytes per LDS row (32 bf16)
  static_assert(TM % 32 == 0 && TN % 32 == 0, "tile");

  __shared__ __attribute__((aligned(16))) char smem[2 * (BM + BN) * ROWB];
  char* const sW = smem;                     // [2][BN][64B]
  char* const sX = smem + 2 * BN * ROWB;     // [2][BM][64B]

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WAVES_N, wn = wave % WAVES_N;
  const YoloConvDesc& d = a.d;

  // XCD-aware tile order: blocks with equal blockIdx%8 share an L2; give each such group a contiguous
  // run of tiles (cout tile fastest) so the pixel tile is re-read from that L2 (bijective for any grid).
  int m0, n0;
  {
    const int nblk = gridDim.x, bid = blockIdx.x;
    const int q = nblk >> 3, r = nblk & 7, xcd = bid & 7;
    const int swz = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    const int mt = swz / a.n_tiles;
    m0 = mt * BM;
    n0 = (swz - mt * a.n_tiles) * BN;
  }

  const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)a.x, 0, a.x_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void*)a.w, 0, a.w_bytes, 0x00020000);

  // ---- per-thread staging state -----------------------------------------------------------------
  const int frow = lane >> 2;                          // row inside a 16-row LDS-DMA piece
  const int chunk = (lane & 3) ^ ((lane >> 4) & 3);    // logical 8-channel chunk this lane fetches
  int px_base[PIT], px_hi0[PIT], px_wi0[PIT];
  bool px_ok[PIT];
  const int hw_out = d.ho * d.wo;
#pragma unroll
  for (int it = 0; it < PIT; ++it) {
    const int m = m0 + it * 64 + wave * 16 + frow;
    px_ok[it] = m < a.M;
    const int mm = px_ok[it] ? m : 0;
    const int b = mm / hw_out, rem = mm - b * hw_out;
    const int oh = rem / d.wo, ow = rem - oh * d.wo;
    px_hi0[it] = oh * d.stride - d.pad;
    px_wi0[it] = ow * d.stride - d.pad;
    px_base[it] = ((b * d.h + px_hi0[it]) * d.w + px_wi0[it]) * d.in_c_total + d.in_c_offset;
  }
  uint32_t w_off[WIT];
#pragma unroll
  for (int it = 0; it < WIT; ++it)
    w_off[it] = (uint32_t)(((n0 + it * 64 + wave * 16 + frow) * d.kpad + chunk * 8) * 2);
  const int ntaps = d.ksize * d.ksize;
  int tap = (chunk * 8) / d.cin;
  int kc = chunk * 8 - tap * d.cin;

  auto stage = [&](int buf, int step) {
    int dh = 0, dw = 0;
    if (d.ksize == 3) {
      dh = (tap * 11) >> 5;  // tap / 3 for tap < 12
      dw = tap - 3 * dh;
    }
    const bool tap_ok = tap < ntaps;
    const int tap_off = (dh * d.w + dw) * d.in_c_total + kc;
    char* const xb = sX + buf * (BM * ROWB) + wave * 1024;
#pragma unroll
    for (int it = 0; it < PIT; ++it) {
      const int hi = px_hi0[it] + dh, wi = px_wi0[it] + dw;
      const bool ok = px_ok[it] && tap_ok && (unsigned)hi < (unsigned)d.h && (unsigned)wi < (unsigned)d.w;
      const uint32_t voff = ok ? (uint32_t)(px_base[it] + tap_off) * 2u : kOobOffset;
      lds_dma16(rx, xb + it * 4096, voff);
    }
    char* const wb = sW + buf * (BN * ROWB) + wave * 1024;
#pragma unroll
    for (int it = 0; it < WIT; ++it) {
      if (BN >= 64 || wave * 16 < BN)   // BN == 32: only waves 0,1 carry weight rows
        lds_dma16(rw, wb + it * 4096, w_off[it] + (uint32_t)step * 64u);
    }
    kc += 32;
    while (kc >= d.cin) {
      kc -= d.cin;
      ++tap;
    }
  };

  f32x16 acc[MI][NI];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NI; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  const int r32 = lane & 31, khalf = lane >> 5;

  stage(0, 0);
  for (int s = 0; s < a.steps; ++s) {
    __syncthreads();  // hipcc drains vmcnt(0) here: step s has landed, step s-1's reads are done
    if (s + 1 < a.steps) stage((s + 1) & 1, s + 1);
    const char* wbuf = sW + (s & 1) * (BN * ROWB);
    const char* xbuf = sX + (s & 1) * (BM * ROWB);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const int g = ks * 2 + khalf;
      bf16x8 wf[MI], xf[NI];
#pragma unroll
      for (int i = 0; i < MI; ++i) {
        const int R = wn * TN + i * 32 + r32;
        wf[i] = *reinterpret_cast<const bf16x8*>(wbuf + R * ROWB + ((g ^ ((R >> 2) & 3)) << 4));
      }
#pragma unroll
      for (int j = 0; j < NI; ++j) {
        const int R = wm * TM + j * 32 + r32;
        xf[j] = *reinterpret_cast<const bf16x8*>(xbuf + R * ROWB + ((g ^ ((R >> 2) & 3)) << 4));
      }
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[i], xf[j], acc[i][j], 0, 0, 0);
    }
  }

  // ---- epilogue: lane = pixel (col), registers = couts (row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)) ----
  const bool f32_out = d.out_dtype == YOLO_DT_F32;
#pragma unroll
  for (int j = 0; j < NI; ++j) {
    const int pix = m0 + wm * TM + j * 32 + r32;
    if (pix >= a.M) continue;
    long out_pix = pix;
    int out_row_pitch = 0;  // pixels per output row when upsampling
    if (d.upsample2x) {
      const int b = pix / hw_out, rem = pix - b * hw_out;
      const int oh = rem / d.wo, ow = rem - oh * d.wo;
      out_row_pitch = 2 * d.wo;
      out_pix = ((long)(b * 2 * d.ho + 2 * oh)) * out_row_pitch + 2 * ow;
    }
#pragma unroll
    for (int i = 0; i < MI; ++i) {
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) {
        const int c0 = n0 + wn * TN + i * 32 + g4 * 8 + khalf * 4;
        if (c0 >= d.cout) continue;
        const f32x4 bv = *reinterpret_cast<const f32x4*>(a.bias + c0);
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = apply_act(acc[i][j][g4 * 4 + e] + bv[e], d.act);
        const bool full = c0 + 3 < d.cout;
        if (a.aux) {
          bf16_t* ap = a.aux + (long)pix * d.aux_c_total + d.aux_c_offset + c0;
          if (full) {
            bf16x4 o = {(bf16_t)v[0], (bf16_t)v[1], (bf16_t)v[2], (bf16_t)v[3]};
            *reinterpret_cast<bf16x4*>(ap) = o;
          } else {
            for (int e = 0; e < 4 && c0 + e < d.cout; ++e) ap[e] = (bf16_t)v[e];
          }
        }
        if (a.res) {
          const bf16_t* rp = a.res + (long)pix * d.res_c_total + d.res_c_offset + c0;
          if (full) {
            const bf16x4 rv = *reinterpret_cast<const bf16x4*>(rp);
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] += (float)rv[e];
          } else {
            for (int e = 0; e < 4 && c0 + e < d.cout; ++e) v[e] += (float)rp[e];
          }
        }
        const int reps = d.upsample2x ? 4 : 1;
        for (int rep = 0; rep < reps; ++rep) {
          const long op = out_pix + (rep >> 1) * out_row_pitch + (rep & 1);
          const long eo = op * d.out_c_total + d.out_c_offset + c0;
          if (f32_out) {
            float* yp = reinterpret_cast<float*>(a.y) + eo;
            if (full) {
              f32x4 o = {v[0], v[1], v[2], v[3]};
              *reinterpret_cast<f32x4*>(yp) = o;
            } else {
              for (int e = 0; e < 4 && c0 + e < d.cout; ++e) yp[e] = v[e];
            }
          } else {
            bf16_t* yp = reinterpret_cast<bf16_t*>(a.y) + eo;
            if (full) {
              bf16x4 o = {(bf16_t)v[0], (bf16_t)v[1], (bf16_t)v[2], (bf16_t)v[3]};
              *reinterpret_cast<bf16x4*>(yp) = o;
            } else {
              for (int e = 0; e < 4 && c0 + e < d.cout; ++e) yp[e] = (bf16_t)v[e];
            }
          }
        }
      }
    }
  }
}

template <int BM, int BN, int WAVES_M, int WAVES_N>
int launch_cfg(const ConvArgs& a, hipStream_t s) {
  const int m_tiles = (a.M + BM - 1) / BM;
  ConvArgs b = a;
  b.n_tiles = (a.d.cout + BN - 1) / BN;
  const long grid = (long)m_tiles * b.n_tiles;
  if (grid > 0x7fffffffL) return yolo_set_error(YOLO_E_UNSUPPORTED, "conv grid too large");
  hipLaunchKernelGGL((conv_igemm_bf16_kernel<BM, BN, WAVES_M, WAVES_N>), dim3((unsigned)grid), dim3(256), 0, s, b);
  return yolo_check_launch("yolo_conv2d_fwd");
}

}  // namespace

int yolo_conv2d_launch(const void* x, const void* w, const float* bias, const void* res, void* y, void* y_aux,
                       const YoloConvDesc* dp, hipStream_t s) {
  YOLO_REQUIRE(x && w && bias && y && dp, "conv: null pointer");
  const YoloConvDesc& d = *dp;
  YOLO_REQUIRE(d.ksize == 1 || d.ksize == 3, "conv: ksize %d unsupported (1 or 3)", d.ksize);
  YOLO_REQUIRE(d.stride == 1 || d.stride == 2, "conv: stride %d unsupported", d.stride);
  YOLO_REQUIRE(d.cin > 0 && d.cin % 8 == 0, "conv: cin %d must be a positive multiple of 8", d.cin);
  YOLO_REQUIRE(d.in_c_offset % 8 == 0 && d.in_c_total % 8 == 0 && d.in_c_offset + d.cin <= d.in_c_total,
               "conv: bad input view (cin %d, offset %d, total %d)", d.cin, d.in_c_offset, d.in_c_total);
  YOLO_REQUIRE(d.out_c_offset % 4 == 0 && d.out_c_total % 4 == 0 && d.out_c_offset + d.cout <= d.out_c_total,
               "conv: bad output view (cout %d, offset %d, total %d)", d.cout, d.out_c_offset, d.out_c_total);
  YOLO_REQUIRE(d.kpad % 32 == 0 && d.kpad >= d.ksize * d.ksize * d.cin, "conv: kpad %d", d.kpad);
  YOLO_REQUIRE(d.cout_pad % 128 == 0 && d.cout_pad >= d.cout, "conv: cout_pad %d", d.cout_pad);
  YOLO_REQUIRE(d.ho == (d.h + 2 * d.pad - d.ksize) / d.stride + 1 && d.wo == (d.w + 2 * d.pad - d.ksize) / d.stride + 1,
               "conv: output size %dx%d inconsistent with input %dx%d k%d s%d p%d", d.ho, d.wo, d.h, d.w, d.ksize,
               d.stride, d.pad);
  if (res) YOLO_REQUIRE(d.res_c_total % 4 == 0 && d.res_c_offset % 4 == 0 && !d.upsample2x, "conv: bad residual view");
  if (y_aux) YOLO_REQUIRE(d.aux_c_total % 4 == 0 && d.aux_c_offset % 4 == 0, "conv: bad aux view");
  const size_t x_bytes = (size_t)d.n * d.h * d.w * d.in_c_total * 2;
  const size_t w_bytes = (size_t)d.cout_pad * d.kpad * 2;
  YOLO_REQUIRE(x_bytes < kOobOffset && w_bytes < kOobOffset, "conv: tensor larger than 3.75 GiB not supported");
  const long M = (long)d.n * d.ho * d.wo;
  YOLO_REQUIRE(M > 0 && M < 0x7fffffffL / 4, "conv: M out of range");

  ConvArgs a;
  a.x = (const bf16_t*)x;
  a.w = (const bf16_t*)w;
  a.bias = bias;
  a.res = (const bf16_t*)res;
  a.y = y;
  a.aux = (bf16_t*)y_aux;
  a.d = d;
  a.M = (int)M;
  a.n_tiles = 0;
  a.steps = d.kpad / 32;
  a.x_bytes = (uint32_t)x_bytes;
  a.w_bytes = (uint32_t)w_bytes;
  if (d.cout <= 32) return launch_cfg<256, 32, 4, 1>(a, s);
  if (d.cout <= 64) return launch_cfg<256, 64, 4, 1>(a, s);
  return launch_cfg<128, 128, 2, 2>(a, s);
}

extern "C" int yolo_conv2d_fwd(const void* x, const void* w_packed, const float* bias, const void* residual, void* y,
                               void* y_preadd, const YoloConvDesc* d, yolo_stream_t s) {
  return yolo_conv2d_launch(x, w_packed, bias, residual, y, y_preadd, d, (hipStream_t)s);
}

// Host-side weight packer: OIHW f32 -> [cout_pad][kpad] bf16, k = (kh*ks+kw)*cin + c, zero padded.
static inline uint16_t f32_to_bf16_rne(float f) {
  uint32_t u;
  __builtin_memcpy(&u, &f, 4);
  if ((u & 0x7f800000u) == 0x7f800000u && (u & 0x007fffffu)) return (uint16_t)((u >> 16) | 0x0040u);  // NaN stays NaN
  return (uint16_t)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
}

extern "C" int yolo_pack_conv_weight_f32(const float* w, int cout, int cin_w, int ksize, int cin, int cout_pad, int kpad,
                                         uint16_t* out) {
  YOLO_REQUIRE(w && out, "pack: null pointer");
  YOLO_REQUIRE(cin_w <= cin && cout <= cout_pad && ksize * ksize * cin <= kpad, "pack: bad sizes");
  for (size_t i = 0; i < (size_t)cout_pad * kpad; ++i) out[i] = 0;
  for (int o = 0; o < cout; ++o)
    for (int c = 0; c < cin_w; ++c)
      for (int t = 0; t < ksize * ksize; ++t)
        out[(size_t)o * kpad + (size_t)t * cin + c] = f32_to_bf16_rne(w[((size_t)o * cin_w + c) * ksize * ksize + t]);
  return 0;
}
