// fp32 "reference-precision" execution mode of the path for gfx950 (MI355X): the same graph as the bf16 kernels, but
// every activation and weight stays float32 and the contraction runs on v_mfma_f32_32x32x2_f32 — exact f32 products and
// f32 accumulation (a k-ordered fmaf chain, cdna_hip_programming.md 3 "FP32-input MFMA"), i.e. the arithmetic of the
// reference's ATen fp32 convolution up to summation order.  It exists for parity, not for throughput: with it the
// boxes / scores of forward() agree with the reference to ~1e-5 and the kept-index sets after NMS are the reference's
// (tests/test_gpu_parity.py, fp32 cases), which bf16 operands cannot deliver (profiles/r02_drift_model.md).
//
// Replaces, like conv_igemm.hip: ConvBlock.forward (reference models/yolo_base.py:19-44, BN folded per
// utils/torch_utils.py:33-60), plain nn.Conv2d heads (models/yolov3_tiny.py:38,42), Add (models/yolov3_spp.py:12-14),
// Upsample (models/yolo_layer.py:6-13), the Concat placement (models/yolo_layer.py:16-22) and MaxPool
// (models/yolo_base.py:60-66; the SPP pyramid models/yolov3_spp.py:75-77 is three pool launches into the concat buffer).
//
// GEMM view: D[cout][pixel] = sum_k W[cout][k] X[k][pixel], k = (kh*ks + kw)*cin + c.  Block tile 128 pixels x 128 couts,
// K step 32 floats, 4 waves of 64 x 64 (2 x 2 accumulators of 32 x 32).  Operands are staged global -> registers -> LDS
// (the gather needs a per-lane tap / border decision; bandwidth is a non-issue at 1/16 of the bf16 MFMA rate), two LDS
// buffers, one barrier per K step.  LDS rows are 128 B with the 16-byte chunk XOR-swizzled by (row >> 1) & 7, so the
// ds_read_b128 fragment reads are conflict-free.  A lane reads FOUR consecutive k of its row at once and feeds them to four
// MFMAs: MFMA t of a chunk pair multiplies k = {4*chunk_lo + t, 4*chunk_hi + t} — a permutation of the K order that is the
// same for both operands, so the sum is unchanged.
#include "conv_common.h"

using namespace yolo_conv;

namespace {

struct ConvF32Args {
  const float* x;
  const float* w;
  const float* bias;
  const float* res;
  float* y;
  float* aux;
  YoloConvDesc d;
  int M, n_tiles, steps;
};

constexpr int F_BM = 128, F_BN = 128, F_BK = 32;
constexpr int F_ROWB = F_BK * 4;                       // 128 bytes per LDS row
constexpr int F_STAGE = (F_BM + F_BN) * F_ROWB;        // 32 KB

__global__ __launch_bounds__(256) void conv_f32_kernel(const ConvF32Args a) {
  __shared__ __attribute__((aligned(16))) char smem[2 * F_STAGE];   // per stage: [128 weight rows][128 pixel rows]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const YoloConvDesc& d = a.d;
  int m0, n0;
  {
    const int swz = xcd_swizzle(blockIdx.x, gridDim.x);
    const int mt = swz / a.n_tiles;
    m0 = mt * F_BM;
    n0 = (swz - mt * a.n_tiles) * F_BN;
  }
  // ---- staging: thread -> 16-byte chunk column cc of rows r0 + 32 i
  const int cc = tid & 7, r0 = tid >> 3;
  const int hw_out = d.ho * d.wo, ntaps = d.ksize * d.ksize;
  long px_base[4];
  int px_hi0[4], px_wi0[4];
  bool px_ok[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int m = m0 + r0 + 32 * i;
    px_ok[i] = m < a.M;
    const int mm = px_ok[i] ? m : 0;
    const int b = mm / hw_out, rem = mm - b * hw_out;
    const int oh = rem / d.wo, ow = rem - oh * d.wo;
    px_hi0[i] = oh * d.stride - d.pad;
    px_wi0[i] = ow * d.stride - d.pad;
    px_base[i] = ((long)(b * d.h + px_hi0[i]) * d.w + px_wi0[i]) * d.in_c_total + d.in_c_offset;
  }
  const float* wrow[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) wrow[i] = a.w + (long)(n0 + r0 + 32 * i) * d.kpad + cc * 4;

  f32x4 xr[4], wr[4];
  auto fetch = [&](int s) {
    const int k = s * F_BK + cc * 4;
    const int tap = k / d.cin, c = k - tap * d.cin;
    const int dh = tap / d.ksize, dw = tap - dh * d.ksize;
    const bool tap_ok = tap < ntaps;
    const long toff = ((long)dh * d.w + dw) * d.in_c_total + c;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const bool ok = tap_ok && px_ok[i] && (unsigned)(px_hi0[i] + dh) < (unsigned)d.h && (unsigned)(px_wi0[i] + dw) < (unsigned)d.w;
      const f32x4 z = {0.f, 0.f, 0.f, 0.f};
      xr[i] = ok ? *reinterpret_cast<const f32x4*>(a.x + px_base[i] + toff) : z;
      wr[i] = *reinterpret_cast<const f32x4*>(wrow[i] + s * F_BK);
    }
  };
  auto commit = [&](int buf) {
    char* const wb = smem + buf * F_STAGE;
    char* const xb = wb + F_BN * F_ROWB;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int r = r0 + 32 * i;
      const int slot = (cc ^ ((r >> 1) & 7)) << 4;
      *reinterpret_cast<f32x4*>(wb + r * F_ROWB + slot) = wr[i];
      *reinterpret_cast<f32x4*>(xb + r * F_ROWB + slot) = xr[i];
    }
  };

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  const int r32 = lane & 31, khalf = lane >> 5;
  fetch(0);
  commit(0);
  __syncthreads();
  for (int s = 0; s < a.steps; ++s) {
    const bool more = s + 1 < a.steps;
    if (more) fetch(s + 1);                           // global loads fly under the MFMAs of this step
    const char* const wb = smem + (s & 1) * F_STAGE;
    const char* const xb = wb + F_BN * F_ROWB;
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      const int g = kk * 2 + khalf;
      f32x4 wf[2], xf[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int R = wn * 64 + i * 32 + r32;
        wf[i] = *reinterpret_cast<const f32x4*>(wb + R * F_ROWB + ((g ^ ((R >> 1) & 7)) << 4));
      }
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int R = wm * 64 + j * 32 + r32;
        xf[j] = *reinterpret_cast<const f32x4*>(xb + R * F_ROWB + ((g ^ ((R >> 1) & 7)) << 4));
      }
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(wf[i][t], xf[j][t], acc[i][j], 0, 0, 0);
    }
    if (more) commit((s + 1) & 1);                    // the other buffer: its readers finished before the last barrier
    __syncthreads();
  }

  // ---- epilogue: lane = pixel (col), registers = couts (row = (reg&3) + 8*(reg>>2) + 4*(lane>>5))
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int pix = m0 + wm * 64 + j * 32 + r32;
    if (pix >= a.M) continue;
    long out_pix = pix;
    int out_row_pitch = 0;
    if (d.upsample2x) {
      const int b = pix / hw_out, rem = pix - b * hw_out;
      const int oh = rem / d.wo, ow = rem - oh * d.wo;
      out_row_pitch = 2 * d.wo;
      out_pix = ((long)(b * 2 * d.ho + 2 * oh)) * out_row_pitch + 2 * ow;
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) {
        const int c0 = n0 + wn * 64 + i * 32 + g4 * 8 + khalf * 4;
        if (c0 >= d.cout) continue;
        const f32x4 bv = *reinterpret_cast<const f32x4*>(a.bias + c0);
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = apply_act(acc[i][j][g4 * 4 + e] + bv[e], d.act);
        const bool full = c0 + 3 < d.cout;
        if (a.aux) {
          float* ap = a.aux + (long)pix * d.aux_c_total + d.aux_c_offset + c0;
          if (full) *reinterpret_cast<f32x4*>(ap) = f32x4{v[0], v[1], v[2], v[3]};
          else
            for (int e = 0; e < 4 && c0 + e < d.cout; ++e) ap[e] = v[e];
        }
        if (a.res) {
          const float* rp = a.res + (long)pix * d.res_c_total + d.res_c_offset + c0;
          if (full) {
            const f32x4 rv = *reinterpret_cast<const f32x4*>(rp);
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] += rv[e];
          } else {
            for (int e = 0; e < 4 && c0 + e < d.cout; ++e) v[e] += rp[e];
          }
        }
        const int reps = d.upsample2x ? 4 : 1;
        for (int rep = 0; rep < reps; ++rep) {
          const long op = out_pix + (rep >> 1) * out_row_pitch + (rep & 1);
          float* yp = a.y + op * d.out_c_total + d.out_c_offset + c0;
          if (full) *reinterpret_cast<f32x4*>(yp) = f32x4{v[0], v[1], v[2], v[3]};
          else
            for (int e = 0; e < 4 && c0 + e < d.cout; ++e) yp[e] = v[e];
        }
      }
  }
}

// Generic max pool on fp32 NHWC views, -inf padding (nn.MaxPool2d semantics); one thread = 4 channels of one output pixel.
__global__ __launch_bounds__(256) void maxpool_f32_kernel(const float* __restrict__ x, float* __restrict__ y, int h, int w, int cg,
                                                          int in_ct, int in_co, int ho, int wo, int out_ct, int out_co, int k,
                                                          int stride, int pad, int dil, long total) {
  const long t = (long)blockIdx.x * 256 + threadIdx.x;
  if (t >= total) return;
  const int g = (int)(t % cg);
  long p = t / cg;
  const int ow = (int)(p % wo);
  p /= wo;
  const int oh = (int)(p % ho);
  const long b = p / ho;
  f32x4 m = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
  for (int i = 0; i < k; ++i) {
    const int hi = oh * stride - pad + i * dil;
    if ((unsigned)hi >= (unsigned)h) continue;
    for (int j = 0; j < k; ++j) {
      const int wi = ow * stride - pad + j * dil;
      if ((unsigned)wi >= (unsigned)w) continue;
      const f32x4 v = *reinterpret_cast<const f32x4*>(x + ((b * h + hi) * w + wi) * in_ct + in_co + g * 4);
#pragma unroll
      for (int e = 0; e < 4; ++e) m[e] = fmaxf(m[e], v[e]);
    }
  }
  *reinterpret_cast<f32x4*>(y + ((b * ho + oh) * wo + ow) * out_ct + out_co + g * 4) = m;
}

// NCHW f32 -> NHWC f32, channels padded with zeros.
__global__ __launch_bounds__(256) void pack_input_f32_kernel(const float* __restrict__ x, float* __restrict__ y, int c, long hw,
                                                             long total_pix, int c_pad) {
  const long p = (long)blockIdx.x * 256 + threadIdx.x;
  if (p >= total_pix) return;
  const long b = p / hw, sp = p - b * hw;
  const float* src = x + b * c * hw + sp;
  float* dst = y + p * c_pad;
  for (int c0 = 0; c0 < c_pad; c0 += 4) {
    f32x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = (c0 + e < c) ? src[(long)(c0 + e) * hw] : 0.f;
    *reinterpret_cast<f32x4*>(dst + c0) = o;
  }
}

}  // namespace

extern "C" int yolo_conv2d_f32_fwd(const float* x, const float* w_packed, const float* bias, const float* residual, float* y,
                                   float* y_preadd, const YoloConvDesc* dp, yolo_stream_t s) {
  YOLO_REQUIRE(x && w_packed && bias && y && dp, "conv_f32: null pointer");
  const YoloConvDesc& d = *dp;
  YOLO_REQUIRE(d.ksize >= 1 && d.ksize <= 7 && (d.stride == 1 || d.stride == 2), "conv_f32: ksize %d / stride %d unsupported", d.ksize,
               d.stride);
  YOLO_REQUIRE(d.cin > 0 && d.cin % 4 == 0, "conv_f32: cin %d must be a positive multiple of 4", d.cin);
  YOLO_REQUIRE(d.in_c_offset % 4 == 0 && d.in_c_total % 4 == 0 && d.in_c_offset + d.cin <= d.in_c_total,
               "conv_f32: bad input view (cin %d, offset %d, total %d)", d.cin, d.in_c_offset, d.in_c_total);
  YOLO_REQUIRE(d.out_c_offset % 4 == 0 && d.out_c_total % 4 == 0 && d.out_c_offset + d.cout <= d.out_c_total,
               "conv_f32: bad output view (cout %d, offset %d, total %d)", d.cout, d.out_c_offset, d.out_c_total);
  YOLO_REQUIRE(d.kpad % F_BK == 0 && d.kpad >= d.ksize * d.ksize * d.cin, "conv_f32: kpad %d must be a multiple of 32 covering K", d.kpad);
  YOLO_REQUIRE(d.cout_pad % F_BN == 0 && d.cout_pad >= d.cout, "conv_f32: cout_pad %d", d.cout_pad);
  YOLO_REQUIRE(d.ho == (d.h + 2 * d.pad - d.ksize) / d.stride + 1 && d.wo == (d.w + 2 * d.pad - d.ksize) / d.stride + 1,
               "conv_f32: output size %dx%d inconsistent with input %dx%d k%d s%d p%d", d.ho, d.wo, d.h, d.w, d.ksize, d.stride, d.pad);
  if (residual) YOLO_REQUIRE(d.res_c_total % 4 == 0 && d.res_c_offset % 4 == 0 && !d.upsample2x, "conv_f32: bad residual view");
  if (y_preadd) YOLO_REQUIRE(d.aux_c_total % 4 == 0 && d.aux_c_offset % 4 == 0, "conv_f32: bad aux view");
  const long M = (long)d.n * d.ho * d.wo;
  YOLO_REQUIRE(M > 0 && M < 0x7fffffffL / 4, "conv_f32: M out of range");
  ConvF32Args a;
  a.x = x, a.w = w_packed, a.bias = bias, a.res = residual, a.y = y, a.aux = y_preadd, a.d = d;
  a.M = (int)M;
  a.n_tiles = (d.cout + F_BN - 1) / F_BN;
  a.steps = (d.ksize * d.ksize * d.cin + F_BK - 1) / F_BK;
  const long grid = ((M + F_BM - 1) / F_BM) * a.n_tiles;
  YOLO_REQUIRE(grid <= 0x7fffffffL, "conv_f32: grid too large");
  hipLaunchKernelGGL(conv_f32_kernel, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)s, a);
  return yolo_check_launch("yolo_conv2d_f32_fwd");
}

extern "C" int yolo_maxpool_f32_fwd(const float* x, float* y, int n, int h, int w, int c, int in_c_total, int in_c_offset, int ho,
                                    int wo, int out_c_total, int out_c_offset, int ksize, int stride, int pad, int dilation,
                                    yolo_stream_t s) {
  YOLO_REQUIRE(x && y, "maxpool_f32: null pointer");
  YOLO_REQUIRE(c > 0 && c % 4 == 0 && in_c_total % 4 == 0 && in_c_offset % 4 == 0 && out_c_total % 4 == 0 && out_c_offset % 4 == 0,
               "maxpool_f32: channels / views must be multiples of 4");
  YOLO_REQUIRE(in_c_offset + c <= in_c_total && out_c_offset + c <= out_c_total, "maxpool_f32: view out of range");
  YOLO_REQUIRE(ksize >= 1 && stride >= 1 && dilation >= 1 && pad >= 0 && ho > 0 && wo > 0, "maxpool_f32: bad geometry");
  YOLO_REQUIRE((ho - 1) * stride - pad < h && (wo - 1) * stride - pad < w, "maxpool_f32: an output window starts outside the input");
  const long total = (long)n * ho * wo * (c / 4);
  hipLaunchKernelGGL(maxpool_f32_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)s, x, y, h, w, c / 4,
                     in_c_total, in_c_offset, ho, wo, out_c_total, out_c_offset, ksize, stride, pad, dilation, total);
  return yolo_check_launch("yolo_maxpool_f32_fwd");
}

extern "C" int yolo_pack_input_nchw_f32_nhwc(const float* x, float* y, int n, int c, int h, int w, int c_pad, yolo_stream_t s) {
  YOLO_REQUIRE(x && y && n > 0 && c > 0 && h > 0 && w > 0, "pack_input_f32: bad arguments");
  YOLO_REQUIRE(c_pad >= c && c_pad % 4 == 0, "pack_input_f32: c_pad %d must be a multiple of 4 >= c", c_pad);
  const long total = (long)n * h * w;
  hipLaunchKernelGGL(pack_input_f32_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)s, x, y, c, (long)h * w,
                     total, c_pad);
  return yolo_check_launch("yolo_pack_input_nchw_f32_nhwc");
}

// Host-side weight packer of the fp32 mode: OIHW f32 -> [cout_pad][kpad] f32, k = (kh*ks + kw)*cin + c, zero padded.
extern "C" int yolo_pack_conv_weight_f32_f32(const float* w, int cout, int cin_w, int ksize, int cin, int cout_pad, int kpad,
                                             float* out) {
  YOLO_REQUIRE(w && out, "pack_f32: null pointer");
  YOLO_REQUIRE(cin_w <= cin && cout <= cout_pad && ksize * ksize * cin <= kpad, "pack_f32: bad sizes");
  for (size_t i = 0; i < (size_t)cout_pad * kpad; ++i) out[i] = 0.f;
  for (int o = 0; o < cout; ++o)
    for (int c = 0; c < cin_w; ++c)
      for (int t = 0; t < ksize * ksize; ++t)
        out[(size_t)o * kpad + (size_t)t * cin + c] = w[((size_t)o * cin_w + c) * ksize * ksize + t];
  return 0;
}
