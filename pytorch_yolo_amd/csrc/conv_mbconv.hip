// One MobileNetV2 inverted-residual block per launch (torchvision InvertedResidual as used by
// /root/reference/pytorch_yolo/models/yolov3_tiny_mobilenet.py:14-46):
//     y = [x +] proj1x1( relu6( dw3x3_stride( relu6( expand1x1(x) ) ) ) ),   BN folded into every conv.
// Unfused, the 6x-expanded tensor is written by the expand conv, read and written by the depthwise conv and read
// again by the projection: 13-25 x the block's input + output bytes, and the early blocks (208x208 .. 52x52 maps)
// are pure HBM traffic.  Here a persistent workgroup owns a TH x TW output tile at a time and keeps everything on
// the CU:
//   A  x halo tile ((TH-1)*S+3) x ((TW-1)*S+3) pixels -> LDS (bf16, K padded to 32), next tile's loads already in flight
//   B  expand GEMM on the halo (v_mfma_f32_16x16x32_bf16, K = cin <= 32: one MFMA per 16 pixels x 16 channels),
//      + bias, ReLU6, zero outside the image (the depthwise conv pads the EXPANDED map), bf16 -> LDS E[pixel][ce]
//   C  depthwise 3x3 on E: a thread owns 4 channels (its 36 weights live in registers) and walks pixels, -> LDS D[pixel][ce]
//   D  projection GEMM D x Wp (K = ce), + bias (+ x from the LDS tile), bf16 -> y
// The rounding points are those of the three-launch path (E and D are bf16 there too), so both paths agree to the
// last bit of the bf16 results up to fp32 summation order.
#include "common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef bf16_t bf16x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

struct MbArgs {
  const bf16_t* x;
  bf16_t* y;
  const bf16_t* we;     // [ce][48] bf16 (LDS image), nullptr: no expand conv
  const float* be;      // [ce]
  const float* wd;      // [9][ce]
  const float* bd;      // [ce]
  const bf16_t* wp;     // [cop][dstride/2] bf16 (LDS image)
  const float* bp;      // [cop]
  int n, h, w, ho, wo, cin, in_ct, in_co, ce, cout, cop, out_ct, out_co, has_res;
  int tiles_x, tiles_y, n_tiles, dstride;
  int debug;            // YOLO_MBCONV_DEBUG (timing only, results wrong): 2 no expand stage, 4 no depthwise stage, 8 no projection stage, 16 no x loads
};

constexpr int kXStride = 96;      // bytes per pixel row of the x tile / per row of W_expand: 32 bf16 + pad; rows 24 banks
                                  // apart make the 16x16x32 fragment reads (ds_read_b128) conflict-free

__device__ __forceinline__ float relu6(float v) { return __builtin_amdgcn_fmed3f(v, 0.f, 6.f); }   // one instruction

// two bf16 packed in a dword -> two f32 (exact: a bf16 is the high half of its f32)
__device__ __forceinline__ f32x2 bf16pair_to_f32(uint32_t w) {
  return f32x2{__builtin_bit_cast(float, w << 16), __builtin_bit_cast(float, w & 0xffff0000u)};
}

template <int S, int TH, int TW, bool EXPAND, int NT>
__global__ __launch_bounds__(NT) __attribute__((amdgpu_waves_per_eu(4))) void mbconv_kernel(const MbArgs a) {   // 16 waves per CU
  constexpr int IH = (TH - 1) * S + 3, IW = (TW - 1) * S + 3, HP = IH * IW, HPP = (HP + 15) / 16 * 16, P = TH * TW;
  static_assert(P % 16 == 0, "tile");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  constexpr int nt = NT, nw = NT / 64;                   // 256 / 512 threads (four / two workgroups per CU) or 1024
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // E rows: +8 bytes when the expand epilogue writes them (16 lanes = 16 pixels per ds_write_b64: an odd multiple of
  // 8 bytes apart spreads them over the banks); without an expand conv the rows take 16-byte tile loads
  const int ce = a.ce, estride = ce * 2 + (EXPAND ? 8 : 0), dstride = a.dstride;
  // LDS map: [X tile HPP x 96 B (EXPAND only)] [E: HPP x estride] [D: P x dstride] [W_expand: ce x 96 B] [W_proj: cop x dstride]
  // [b_expand f32 ce] [b_proj f32 cop] [-1e30 f32 ce]
  char* const lx = smem;
  char* const le = lx + (EXPAND ? HPP * kXStride : 0);
  char* const ld = le + HPP * estride;
  char* const lwe = ld + P * dstride;
  char* const lwp = lwe + (EXPAND ? ce * kXStride : 0);
  float* const lbe = reinterpret_cast<float*>(lwp + a.cop * dstride);
  float* const lbp = lbe + ce;
  float* const lbad = lbp + a.cop;                       // [ce] x -1e30 (EXPAND only)

  // ---- once per workgroup: weights -> LDS, zero the K padding of the x tile
  if (EXPAND) {
    for (int i = tid; i < ce * (kXStride / 16); i += nt)
      reinterpret_cast<uint4*>(lwe)[i] = reinterpret_cast<const uint4*>(a.we)[i];
    for (int i = tid; i < HPP * (kXStride / 16); i += nt) reinterpret_cast<uint4*>(lx)[i] = make_uint4(0, 0, 0, 0);
    for (int i = tid; i < ce; i += nt) lbe[i] = a.be[i], lbad[i] = -1e30f;
  } else {
    for (int i = tid; i < HPP * estride / 16; i += nt) reinterpret_cast<uint4*>(le)[i] = make_uint4(0, 0, 0, 0);
  }
  for (int i = tid; i < a.cop * dstride / 16; i += nt)
    reinterpret_cast<uint4*>(lwp)[i] = reinterpret_cast<const uint4*>(a.wp)[i];
  for (int i = tid; i < a.cop; i += nt) lbp[i] = a.bp[i];

  // depthwise: this thread's 4 channels
  const int qn = ce / 4, groups = nt / qn;
  const int qd = tid % qn, grp = tid / qn;
  const bool dw_on = grp < groups;
  f32x2 wdr2[9][2];
  float bdr[4];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int e = 0; e < 4; ++e) wdr2[t][e >> 1][e & 1] = dw_on ? a.wd[t * ce + qd * 4 + e] : 0.f;
#pragma unroll
  for (int e = 0; e < 4; ++e) bdr[e] = dw_on ? a.bd[qd * 4 + e] : 0.f;

  // ---- x tile loads: piece = (halo pixel, 16-byte channel chunk); at most 4 * HP <= NPRE * NT pieces
  constexpr int NPRE = (4 * HP + NT - 1) / NT;
  const int chunks = a.cin / 8, pieces = HP * chunks;
  const int xrow = EXPAND ? kXStride : estride;          // without an expand conv the tile IS E (hidden == cin)
  char* const xdst = EXPAND ? lx : le;
  uint4 pre[NPRE];
  auto fetch = [&](int tile) {
    const int tx = tile % a.tiles_x, r = tile / a.tiles_x, ty = r % a.tiles_y, b = r / a.tiles_y;
    const int iy0 = ty * TH * S - 1, ix0 = tx * TW * S - 1;
#pragma unroll
    for (int k = 0; k < NPRE; ++k) {
      const int pc = tid + k * nt;
      pre[k] = make_uint4(0, 0, 0, 0);
      if (pc < pieces) {
        const int pix = pc / chunks, ch = pc - pix * chunks;
        const int iy = iy0 + pix / IW, ix = ix0 + pix % IW;
        if ((unsigned)iy < (unsigned)a.h && (unsigned)ix < (unsigned)a.w)
          pre[k] = *reinterpret_cast<const uint4*>(a.x + ((long)(b * a.h + iy) * a.w + ix) * a.in_ct + a.in_co + ch * 8);
      }
    }
  };
  auto stash = [&]() {
#pragma unroll
    for (int k = 0; k < NPRE; ++k) {
      const int pc = tid + k * nt;
      if (pc < pieces) {
        const int pix = pc / chunks, ch = pc - pix * chunks;
        *reinterpret_cast<uint4*>(xdst + pix * xrow + ch * 16) = pre[k];
      }
    }
  };

  const int c16 = lane & 15, q = lane >> 4;
  int tile = blockIdx.x;
  if (tile < a.n_tiles) fetch(tile);
  __syncthreads();
  for (; tile < a.n_tiles; tile += gridDim.x) {
    const int tx = tile % a.tiles_x, r = tile / a.tiles_x, ty = r % a.tiles_y, b = r / a.tiles_y;
    const int oy0 = ty * TH, ox0 = tx * TW;
    const int iy0 = oy0 * S - 1, ix0 = ox0 * S - 1;
    // ---- A: this tile's x halo (fetched during the previous tile) -> LDS; next tile's loads go out
    stash();
    if (tile + (int)gridDim.x < a.n_tiles && !(a.debug & 16)) fetch(tile + gridDim.x);
    __syncthreads();
    // ---- B: E = relu6(X We^T + be), 0 outside the image
    if (EXPAND && !(a.debug & 2)) {
      // A wave takes a contiguous run of 16x16 tiles in CHANNEL-tile-major order: the weight fragment and the bias (the accumulator's
      // start value) stay in registers over the run's pixel-row tiles, so a tile costs one 1 KB LDS read (its pixel fragment) instead
      // of three (round 4: the phase was bound by LDS traffic and by a read -> MFMA -> pack -> write chain per tile with a scalar
      // division in front of it).  Out-of-image pixels are zeroed after packing: the depthwise conv pads the EXPANDED map.
      const int nct = ce / 16, ntl = (HPP / 16) * nct, per = (ntl + nw - 1) / nw;
      int t = wave * per;
      const int t_hi = min(ntl, t + per);
      if (t < t_hi) {
        int ct = t / (HPP / 16), rt = t - ct * (HPP / 16);
        bf16x8 wf = *reinterpret_cast<const bf16x8*>(lwe + (ct * 16 + c16) * kXStride + q * 16);
        f32x4 bias = *reinterpret_cast<const f32x4*>(lbe + ct * 16 + q * 4);
        bf16x8 xf = *reinterpret_cast<const bf16x8*>(lx + (rt * 16 + c16) * kXStride + q * 16);
        for (; t < t_hi; ++t) {
          const f32x4 acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf, xf, bias, 0, 0, 0);
          const int pix = rt * 16 + c16, ct_now = ct;
          // the next tile's operands are on their way while this one is packed
          if (++rt == HPP / 16) rt = 0, ++ct;
          if (t + 1 < t_hi) {
            xf = *reinterpret_cast<const bf16x8*>(lx + (rt * 16 + c16) * kXStride + q * 16);
            if (rt == 0) {
              wf = *reinterpret_cast<const bf16x8*>(lwe + (ct * 16 + c16) * kXStride + q * 16);
              bias = *reinterpret_cast<const f32x4*>(lbe + ct * 16 + q * 4);
            }
          }
          const int py = pix / IW, px = pix - py * IW;
          const int iy = iy0 + py, ix = ix0 + px;
          const bool in = (unsigned)iy < (unsigned)a.h && (unsigned)ix < (unsigned)a.w;
          bf16x4 o;                                                          // lane: pixel c16, channels ct*16 + q*4 ..+3
#pragma unroll
          for (int e = 0; e < 4; ++e) o[e] = (bf16_t)relu6(acc[e]);
          u32x2 ow = __builtin_bit_cast(u32x2, o);
          ow[0] = in ? ow[0] : 0u;
          ow[1] = in ? ow[1] : 0u;
          if (pix < HP) *reinterpret_cast<u32x2*>(le + pix * estride + ct_now * 32 + q * 8) = ow;
        }
      }
    }
    if (EXPAND) __syncthreads();
    // ---- C: D = relu6(dw3x3(E) + bd)
    if (dw_on && !(a.debug & 4)) {
      for (int p = grp; p < P; p += 2 * groups) {       // two pixels per iteration
        const bool two = p + groups < P;
        const int pp[2] = {p, two ? p + groups : p};
        const char* e0[2];
        f32x2 s[2][2];                                   // v_pk_fma_f32: two channels per instruction
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          const int oy = pp[u] / TW, ox = pp[u] - oy * TW;
          e0[u] = le + ((oy * S) * IW + ox * S) * estride + qd * 8;
          s[u][0] = f32x2{bdr[0], bdr[1]};
          s[u][1] = f32x2{bdr[2], bdr[3]};
        }
#pragma unroll
        for (int dy = 0; dy < 3; ++dy)
#pragma unroll
          for (int dx = 0; dx < 3; ++dx)
#pragma unroll
            for (int u = 0; u < 2; ++u) {
              const u32x2 v = *reinterpret_cast<const u32x2*>(e0[u] + (dy * IW + dx) * estride);
              s[u][0] = __builtin_elementwise_fma(bf16pair_to_f32(v[0]), wdr2[dy * 3 + dx][0], s[u][0]);
              s[u][1] = __builtin_elementwise_fma(bf16pair_to_f32(v[1]), wdr2[dy * 3 + dx][1], s[u][1]);
            }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          bf16x4 o;
#pragma unroll
          for (int e = 0; e < 4; ++e) o[e] = (bf16_t)relu6(s[u][e >> 1][e & 1]);
          if (u == 0 || two) *reinterpret_cast<bf16x4*>(ld + pp[u] * dstride + qd * 8) = o;
        }
      }
    }
    __syncthreads();
    // ---- D: y = D Wp^T + bp (+ x)
    if (!(a.debug & 8)) {
      const int nct = a.cop / 16, ntl = (P / 16) * nct, ksteps = ce / 32;
      for (int t = wave; t < ntl; t += nw) {
        const int rt = t / nct, ct = t - rt * nct;
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        for (int k = 0; k < ksteps; ++k) {
          const bf16x8 wf = *reinterpret_cast<const bf16x8*>(lwp + (ct * 16 + c16) * dstride + k * 64 + q * 16);
          const bf16x8 df = *reinterpret_cast<const bf16x8*>(ld + (rt * 16 + c16) * dstride + k * 64 + q * 16);
          acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf, df, acc, 0, 0, 0);
        }
        const int p = rt * 16 + c16, c0 = ct * 16 + q * 4;
        const int oy = oy0 + p / TW, ox = ox0 + p % TW;
        if (oy < a.ho && ox < a.wo && c0 < a.cout) {
          const f32x4 bv = *reinterpret_cast<const f32x4*>(lbp + c0);
          float v[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = acc[e] + bv[e];
          if (a.has_res) {        // stride 1, cin == cout: x at the tile's centre pixel, from the LDS tile
            const bf16x4 xv = *reinterpret_cast<const bf16x4*>(xdst + ((p / TW + 1) * IW + p % TW + 1) * xrow + c0 * 2);
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] += (float)xv[e];
          }
          bf16x4 o;
#pragma unroll
          for (int e = 0; e < 4; ++e) o[e] = (bf16_t)v[e];
          bf16_t* const yp = a.y + ((long)(b * a.ho + oy) * a.wo + ox) * a.out_ct + a.out_co + c0;
          if (c0 + 4 <= a.cout) *reinterpret_cast<bf16x4*>(yp) = o;
          else
            for (int e = 0; e < 4 && c0 + e < a.cout; ++e) yp[e] = o[e];
        }
      }
    }
    __syncthreads();     // x tile, E and D are free for the next tile
  }
}

template <int S, int TH, int TW, bool EXPAND, int NT>
int launch_nt(const MbArgs& a, size_t lds, int slots, hipStream_t s) {
  static std::atomic<uint64_t> lds_set{0};                 // per device (common.h)
  if (const int rc = yolo_max_dyn_lds(reinterpret_cast<const void*>(&mbconv_kernel<S, TH, TW, EXPAND, NT>), 160 * 1024, lds_set, "mbconv")) return rc;
  const int grid = a.n_tiles < slots ? a.n_tiles : slots;
  hipLaunchKernelGGL((mbconv_kernel<S, TH, TW, EXPAND, NT>), dim3((unsigned)grid), dim3(NT), lds, s, a);
  return yolo_check_launch("yolo_mbconv_fwd");
}

template <int S, int TH, int TW, bool EXPAND>
size_t lds_bytes(const MbArgs& a) {
  constexpr int IH = (TH - 1) * S + 3, IW = (TW - 1) * S + 3, HP = IH * IW, HPP = (HP + 15) / 16 * 16, P = TH * TW;
  return (size_t)(EXPAND ? HPP * kXStride : 0) + (size_t)HPP * (a.ce * 2 + (EXPAND ? 8 : 0)) + (size_t)P * a.dstride +
         (size_t)(EXPAND ? a.ce * kXStride : 0) + (size_t)a.cop * a.dstride + (size_t)(a.ce + a.cop + (EXPAND ? a.ce : 0)) * 4;
}

template <int S, int TH, int TW, bool EXPAND>
int launch(const MbArgs& a0, hipStream_t s) {
  MbArgs a = a0;
  a.tiles_x = (a.wo + TW - 1) / TW;
  a.tiles_y = (a.ho + TH - 1) / TH;
  a.n_tiles = a.n * a.tiles_x * a.tiles_y;
  const size_t lds = lds_bytes<S, TH, TW, EXPAND>(a);
  YOLO_REQUIRE(lds <= 160 * 1024, "mbconv: %zu bytes of LDS needed", lds);
  // persistent workgroups; the phases of one tile (load, expand, depthwise, project) are latency chains, so a CU
  // needs more than 8 waves in flight: two 512-thread workgroups per CU when the LDS allows, else one of 1024 threads
  if (lds <= 40 * 1024) return launch_nt<S, TH, TW, EXPAND, 256>(a, lds, 1024, s);   // four per CU: phases of four tiles overlap
  if (lds <= 80 * 1024) return launch_nt<S, TH, TW, EXPAND, 512>(a, lds, 512, s);
  return launch_nt<S, TH, TW, EXPAND, 1024>(a, lds, 256, s);
}

const int conv_mb_debug = [] {      // YOLO_MBCONV_DEBUG bit 1: never halve the tile (tuning only)
  const char* e = getenv("YOLO_MBCONV_DEBUG");
  return e ? atoi(e) : 0;
}();

}  // namespace

// bytes per row of the depthwise-output tile and of W_proj: >= 2*ce, a multiple of 16 and == 96 or 160 (mod 256),
// which spreads the 16 rows of an MFMA fragment read over all banks
extern "C" int yolo_mbconv_dstride(int ce) {
  int s = ce * 2;
  while (s % 256 != 96 && s % 256 != 160) s += 16;
  return s;
}

// the wide blocks (hidden dimension streamed in chunks): conv_mbwide.hip
int yolo_mbwide_supported(int cin, int hidden, int cout, int stride);
int yolo_mbwide_launch(const void* x, const void* w_exp, const float* b_exp, const float* w_dw, const float* b_dw, const void* w_proj,
                       const float* b_proj, void* y, const YoloMbconvDesc& d, hipStream_t st);

// 0: not covered; 1: the whole hidden tile in LDS (this file); 2: the wide form (conv_mbwide.hip) - the two take different weight images
extern "C" int yolo_mbconv_supported(int cin, int hidden, int cout, int stride) {
  const int ce = (hidden + 31) / 32 * 32;
  if (cin >= 8 && cin <= 32 && cin % 8 == 0 && hidden % 4 == 0 && hidden >= cin && ce <= 192 && cout >= 4 && cout <= 64 &&
      cout % 4 == 0 && (stride == 1 || stride == 2))
    return 1;
  return yolo_mbwide_supported(cin, hidden, cout, stride) ? 2 : 0;
}

extern "C" int yolo_mbconv_fwd(const void* x, const void* w_exp, const float* b_exp, const float* w_dw, const float* b_dw,
                               const void* w_proj, const float* b_proj, void* y, const YoloMbconvDesc* dp, yolo_stream_t s) {
  YOLO_REQUIRE(x && w_dw && b_dw && w_proj && b_proj && y && dp, "mbconv: null argument");
  const YoloMbconvDesc& d = *dp;
  YOLO_REQUIRE(yolo_mbconv_supported(d.cin, d.hidden, d.cout, d.stride), "mbconv: cin %d hidden %d cout %d stride %d not covered",
               d.cin, d.hidden, d.cout, d.stride);
  YOLO_REQUIRE(d.has_expand ? (w_exp && b_exp) : d.hidden == d.cin, "mbconv: a block without expand conv has hidden == cin");
  YOLO_REQUIRE(!d.has_res || (d.stride == 1 && d.cin == d.cout), "mbconv: residual needs stride 1 and cin == cout");
  YOLO_REQUIRE(d.in_c_offset % 8 == 0 && d.in_c_total % 8 == 0 && d.in_c_offset + d.cin <= d.in_c_total, "mbconv: bad input view");
  YOLO_REQUIRE(d.out_c_offset % 4 == 0 && d.out_c_total % 4 == 0 && d.out_c_offset + d.cout <= d.out_c_total, "mbconv: bad output view");
  YOLO_REQUIRE(d.n > 0 && d.h > 0 && d.w > 0, "mbconv: empty input");
  if (yolo_mbconv_supported(d.cin, d.hidden, d.cout, d.stride) == 2) {
    YOLO_REQUIRE(d.has_expand, "mbconv: the wide form needs the expand conv");
    return yolo_mbwide_launch(x, w_exp, b_exp, w_dw, b_dw, w_proj, b_proj, y, d, (hipStream_t)s);
  }
  MbArgs a;
  a.x = (const bf16_t*)x;
  a.y = (bf16_t*)y;
  a.we = (const bf16_t*)w_exp;
  a.be = b_exp;
  a.wd = w_dw;
  a.bd = b_dw;
  a.wp = (const bf16_t*)w_proj;
  a.bp = b_proj;
  a.n = d.n;
  a.h = d.h;
  a.w = d.w;
  a.ho = (d.h + 2 - 3) / d.stride + 1;
  a.wo = (d.w + 2 - 3) / d.stride + 1;
  a.cin = d.cin;
  a.in_ct = d.in_c_total;
  a.in_co = d.in_c_offset;
  a.ce = (d.hidden + 31) / 32 * 32;
  a.cout = d.cout;
  a.cop = (d.cout + 15) / 16 * 16;
  a.out_ct = d.out_c_total;
  a.out_co = d.out_c_offset;
  a.has_res = d.has_res;
  a.dstride = yolo_mbconv_dstride(a.ce);
  a.tiles_x = a.tiles_y = a.n_tiles = 0;
  a.debug = conv_mb_debug;
  hipStream_t st = (hipStream_t)s;
  // tile: 8x8 outputs (4x8 at stride 2).  192 hidden channels at stride 1: 4x8, which lets two 512-thread workgroups
  // share a CU instead of one of 1024 threads (-10 %; with 144 hidden channels and at stride 2 the larger halo
  // share of a half tile costs more than the overlap gains: measured, YOLO_MBCONV_DEBUG bit 1 = always the full tile)
  if (d.stride == 1) {
    if (!d.has_expand) return launch<1, 8, 8, false>(a, st);
    if ((a.ce >= 192 || (conv_mb_debug & 32)) && !(conv_mb_debug & 1) && lds_bytes<1, 4, 8, true>(a) <= 80 * 1024) return launch<1, 4, 8, true>(a, st);
    return launch<1, 8, 8, true>(a, st);
  }
  return d.has_expand ? launch<2, 4, 8, true>(a, st) : launch<2, 4, 8, false>(a, st);
}
