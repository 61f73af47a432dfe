// 3x3 / stride 1 / pad 1 ConvBlock with few input channels (16 or 32) fused with the MaxPool2d(2, 2) that follows it:
// the second and third ConvPoolBlock of YOLOv3-tiny (reference models/yolo_base.py:69-80, yolov3_tiny.py:26-29:
// 16 -> 32 on 208x208, 32 -> 64 on 104x104 at 416 input).  These layers are HBM traffic, not MFMA work (K = 144 / 288):
// as conv + pool launches the full-resolution conv output is written and read back (4 x the pooled map), and with 16
// input channels the implicit-GEMM kernel falls to its per-lane gather path.  Here a workgroup walks 16x16 output
// tiles: the 18x18 halo goes to LDS once (next tile's global loads already in flight), every wave multiplies 64
// pixels x 32 output channels with its weight fragments held in registers (v_mfma_f32_32x32x16_bf16, one k slice =
// 8 channels of one tap), and the 2x2 maximum is taken out of the wave's own LDS staging: only the pooled map is
// written.  Same structure as conv1_nchw_kernel (conv3x3_halo.hip), which reads the float32 NCHW batch instead.
#include "conv_common.h"

using namespace yolo_conv;

#ifndef YOLO_SMALL_WGS
#define YOLO_SMALL_WGS 2         // resident workgroups per CU the register budget is cut for (and the persistent grid is sized for)
#endif
#ifndef YOLO_SMALL_PAD32
#define YOLO_SMALL_PAD32 16      // extra bytes per 64-byte halo pixel in LDS
#endif

namespace {

// CIN 16 or 32 (NHWC bf16), COUT 32 or 64: 4 waves per 32 output channels (wave = (64-pixel group, cout half)).
template <int CIN, int COUT, bool POOL>
__global__ __launch_bounds__(COUT * 8, (COUT == 32 ? YOLO_SMALL_WGS : (YOLO_SMALL_WGS + 1) / 2)) void conv3x3_small_kernel(const ConvArgs a) {
  constexpr int NWV = COUT / 8, NT = NWV * 64;
  constexpr int HW2 = 18, HP = 18 * 18, TM = 64, KS = 9 * CIN / 16, CPP = CIN / 8;
  constexpr int PS = CIN * 2 + (CIN == 32 ? YOLO_SMALL_PAD32 : 0);  // halo pixel pitch (measured: + 16 B helps 64-byte pixels only)
  constexpr int SP = 32 * 2 + 16;                                  // staging pitch per pixel: 32 couts bf16 + 16 B pad
  constexpr int HALO_B = ((HP * PS + 1023) / 1024) * 1024;
  constexpr int NPF = (HP * CPP + NT - 1) / NT;                    // 16-byte halo pieces per thread
  extern __shared__ __attribute__((aligned(16))) char smem[];     // HALO_B + NWV * TM * SP (up to 66 KB)
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int pg = wave & 3, ch = wave >> 2;                         // pixel group (tile rows 4pg..4pg+3), cout half
  const YoloConvDesc& d = a.d;
  // persistent workgroup: tiles blockIdx.x, + gridDim.x, ... of the n * tiles_y * tiles_x tiles (x fastest).  The weights
  // are read once, and the halo of tile i + 2 is fetched while tile i is multiplied: with 8 waves per CU nothing else
  // hides an HBM round trip, and a short per-workgroup tile run would expose it on its first tiles every time.
  const int tiles_x = (d.w + 15) / 16, tiles_y = (d.h + 15) / 16;
  const int n_tiles = d.n * tiles_y * tiles_x;

  // weight fragments (A operand): lane = cout r32 of this wave's half, k = ks*16 + khalf*8 + 0..7 of [cout_pad][kpad]
  const int r32 = lane & 31, khalf = lane >> 5;
  bf16x8 wf[KS];
#pragma unroll
  for (int ks = 0; ks < KS; ++ks)
    wf[ks] = *reinterpret_cast<const bf16x8*>(a.w + (long)(ch * 32 + r32) * d.kpad + ks * 16 + khalf * 8);
  f32x4 bias4[4];
#pragma unroll
  for (int g4 = 0; g4 < 4; ++g4) bias4[g4] = *reinterpret_cast<const f32x4*>(a.bias + ch * 32 + g4 * 8 + khalf * 4);

  // halo pieces owned by this thread, prefetched two tiles ahead into registers (two sets, used alternately)
  u32x4 pre[2][NPF];
  auto fetch = [&](int tile, u32x4 (&dst)[NPF]) {
    const int tx = tile % tiles_x, r = tile / tiles_x, ty = r % tiles_y, bb = r / tiles_y;
    const int x0 = tx * 16, y0 = ty * 16;
#pragma unroll
    for (int u = 0; u < NPF; ++u) {
      const int pc = tid + u * NT, hp = pc / CPP, c = pc - hp * CPP;
      const int hy = hp / HW2, hx = hp - hy * HW2;
      const int yy = y0 - 1 + hy, xx = x0 - 1 + hx;
      const bool ok = tile < n_tiles && pc < HP * CPP && (unsigned)yy < (unsigned)d.h && (unsigned)xx < (unsigned)d.w;
      dst[u] = ok ? *reinterpret_cast<const u32x4*>(a.x + ((long)(bb * d.h + yy) * d.w + xx) * d.in_c_total + d.in_c_offset + c * 8)
                  : u32x4{0, 0, 0, 0};
    }
  };
  auto commit = [&](const u32x4 (&src)[NPF]) {
#pragma unroll
    for (int u = 0; u < NPF; ++u) {
      const int pc = tid + u * NT;
      if (pc < HP * CPP) *reinterpret_cast<u32x4*>(smem + (pc / CPP) * PS + (pc % CPP) * 16) = src[u];      // [pixel][CIN] bf16
    }
  };

  char* const stg = smem + HALO_B + wave * (TM * SP);
  const int stride = gridDim.x;
  fetch(blockIdx.x, pre[0]);
  fetch(blockIdx.x + stride, pre[1]);
  int it = 0;
  for (int tile = blockIdx.x; tile < n_tiles; tile += stride, ++it) {
    if (it) __syncthreads();                 // everyone is past the MFMA reads of the previous halo
    // (the two register sets alternate; the branch is uniform, the sets stay in registers)
    if (it & 1) commit(pre[1]); else commit(pre[0]);
    __syncthreads();
    if (it & 1) fetch(tile + 2 * stride, pre[1]); else fetch(tile + 2 * stride, pre[0]);   // flies during two tiles of MFMAs and stores
    const int tx = tile % tiles_x, r_ = tile / tiles_x, ty = r_ % tiles_y, b = r_ / tiles_y;
    const int y0 = ty * 16;
    const int x0 = tx * 16;
    f32x16 acc[2];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;
    // k slices in groups of three (one filter row for 16 input channels), the fragments of the next group are read
    // while the current group multiplies: without that every MFMA waits for its own LDS round trip
    constexpr int G = 3, NG = KS / G;
    static_assert(KS % G == 0, "k slices");
    auto read_group = [&](int g, bf16x8 (&xf)[G][2]) {
#pragma unroll
      for (int i = 0; i < G; ++i) {
        const int k = (g * G + i) * 16 + khalf * 8;     // this lane's k slice: 8 channels c0.. of tap
        const int tap = k / CIN, c0 = k - tap * CIN;
        const int dh = (tap * 11) >> 5, dw = tap - 3 * dh;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const int q = pg * TM + j * 32 + r32;
          const int hr = ((q >> 4) + dh) * HW2 + (q & 15) + dw;
          xf[i][j] = *reinterpret_cast<const bf16x8*>(smem + hr * PS + c0 * 2);
        }
      }
    };
    bf16x8 xa[G][2], xb[G][2];
    read_group(0, xa);
#pragma unroll
    for (int g = 0; g < NG; ++g) {
      bf16x8 (&cur)[G][2] = (g & 1) ? xb : xa;
      bf16x8 (&nxt)[G][2] = (g & 1) ? xa : xb;
      if (g + 1 < NG) read_group(g + 1, nxt);
#pragma unroll
      for (int i = 0; i < G; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[g * G + i], cur[i][j], acc[j], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
    // epilogue: lane = pixel, registers = couts -> the wave's LDS staging [pixel][32 couts] bf16 (single rounding)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) {
        bf16x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = (bf16_t)apply_act(acc[j][g4 * 4 + e] + bias4[g4][e], d.act);
        *reinterpret_cast<bf16x4*>(stg + (j * 32 + r32) * SP + (g4 * 8 + khalf * 4) * 2) = o;
      }
    __builtin_amdgcn_wave_barrier();
    const int pp = lane >> 2, chunk = lane & 3;                    // (pixel, 8-channel chunk)
    bf16_t* const ybase = reinterpret_cast<bf16_t*>(a.y) + d.out_c_offset + ch * 32 + chunk * 8;
    if constexpr (POOL) {
      // the wave's 64 pixels are 4 tile rows x 16 columns -> 2 x 8 pooled pixels
      const int hp = d.h >> 1, wp = d.w >> 1;                      // MaxPool2d(2, 2): floor
      const int pr = pp >> 3, pc = pp & 7;
      const int py = (y0 >> 1) + pg * 2 + pr, px = (x0 >> 1) + pc;
      if (py < hp && px < wp) {
        float m[8];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const int q = (2 * pr + (k >> 1)) * 16 + 2 * pc + (k & 1);
          const bf16x8 v = *reinterpret_cast<const bf16x8*>(stg + q * SP + chunk * 16);
#pragma unroll
          for (int e = 0; e < 8; ++e) m[e] = k == 0 ? (float)v[e] : fmaxf(m[e], (float)v[e]);
        }
        bf16x8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = (bf16_t)m[e];           // exact: the maximum is one of the bf16 inputs
        *reinterpret_cast<bf16x8*>(ybase + ((long)(b * hp + py) * wp + px) * d.out_c_total) = o;
      }
    } else {
#pragma unroll
      for (int pass = 0; pass < 4; ++pass) {
        const int row = pass * 16 + pp;
        const int q = pg * TM + row;
        const int yy = y0 + (q >> 4), xx = x0 + (q & 15);
        if (yy < d.h && xx < d.w)
          *reinterpret_cast<u32x4*>(ybase + ((long)(b * d.h + yy) * d.w + xx) * d.out_c_total) =
              *reinterpret_cast<const u32x4*>(stg + row * SP + chunk * 16);
      }
    }
    __builtin_amdgcn_wave_barrier();
  }
}

// First layer at stride 2 (MobileNetV2's ConvBNReLU(3, 32, stride=2), yolov3_tiny_mobilenet.py:14-46 via torchvision):
// float32 NCHW batch -> 3x3 / stride 2 / pad 1 conv -> 32 channels bf16 NHWC, without the packed NHWC copy of the input
// (conv1_nchw_kernel of conv3x3_halo.hip is the stride-1 form).  16x16 output tiles, 33x33 input halo.
template <int CINR>
__global__ __launch_bounds__(256) void conv1_s2_nchw_kernel(const ConvArgs a, const float* __restrict__ x_nchw, int cin_real_arg) {
  const int cin_real = CINR ? CINR : cin_real_arg;
  constexpr int HW2 = 33, HP = 33 * 33, TM = 64, NPX = (HP + 255) / 256, TPB = 5;
  constexpr int SP = 32 * 2 + 16;
  constexpr int HALO_B = ((HP * 16 + 1023) / 1024) * 1024;    // 17 KB
  __shared__ __attribute__((aligned(16))) char smem[2 * HALO_B + 4 * TM * SP];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const YoloConvDesc& d = a.d;
  const int tiles_x = (d.wo + 15) / 16, tiles_y = (d.ho + 15) / 16;
  const int groups_x = (tiles_x + TPB - 1) / TPB;
  int swz = xcd_swizzle(blockIdx.x, gridDim.x);
  const int tx0 = (swz % groups_x) * TPB;
  swz /= groups_x;
  const int oy0 = (swz % tiles_y) * 16;
  const int b = swz / tiles_y;
  const int n_t = min(TPB, tiles_x - tx0);

  const int r32 = lane & 31, khalf = lane >> 5;
  bf16x8 wf[5];
#pragma unroll
  for (int ks = 0; ks < 5; ++ks) wf[ks] = *reinterpret_cast<const bf16x8*>(a.w + (long)r32 * d.kpad + ks * 16 + khalf * 8);
  f32x4 bias4[4];
#pragma unroll
  for (int g4 = 0; g4 < 4; ++g4) bias4[g4] = *reinterpret_cast<const f32x4*>(a.bias + g4 * 8 + khalf * 4);

  // halo pixels owned by this thread (5 of the 1089), prefetched one tile ahead into registers
  const long plane = (long)d.h * d.w;
  int hy[NPX], hx[NPX];
#pragma unroll
  for (int u = 0; u < NPX; ++u) {
    const int hp = tid + u * 256;
    hy[u] = hp / HW2;
    hx[u] = hp - hy[u] * HW2;
  }
  float pre[NPX][CINR ? CINR : 8];
  auto fetch = [&](int t) {
    const int ix0 = (tx0 + t) * 32 - 1, iy0 = oy0 * 2 - 1;
#pragma unroll
    for (int u = 0; u < NPX; ++u) {
      const int yy = iy0 + hy[u], xx = ix0 + hx[u];
      const bool ok = tid + u * 256 < HP && (unsigned)yy < (unsigned)d.h && (unsigned)xx < (unsigned)d.w;
      const float* src = x_nchw + ((long)b * cin_real) * plane + (long)yy * d.w + xx;
#pragma unroll
      for (int e = 0; e < (CINR ? CINR : 8); ++e) pre[u][e] = (ok && e < cin_real) ? src[e * plane] : 0.f;
    }
  };
  auto commit = [&](int hb) {                                  // registers -> LDS halo buffer as NHWC8 bf16
    char* const hbuf = smem + hb * HALO_B;
#pragma unroll
    for (int u = 0; u < NPX; ++u) {
      if (tid + u * 256 >= HP) continue;
      bf16x8 v;
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = e < (CINR ? CINR : 8) ? (bf16_t)pre[u][e < (CINR ? CINR : 8) ? e : 0] : (bf16_t)0.f;
      *reinterpret_cast<bf16x8*>(hbuf + (tid + u * 256) * 16) = v;
    }
  };

  char* const stg = smem + 2 * HALO_B + wave * (TM * SP);
  fetch(0);
  for (int t = 0; t < n_t; ++t) {
    commit(t & 1);
    __syncthreads();                       // halo t visible; everyone is past the reads of halo t-2's buffer
    if (t + 1 < n_t) fetch(t + 1);
    const char* const hbuf = smem + (t & 1) * HALO_B;
    const int ox0 = (tx0 + t) * 16;
    f32x16 acc[2];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;
#pragma unroll
    for (int ks = 0; ks < 5; ++ks) {
      int tap = ks * 2 + khalf;
      if (tap > 8) tap = 8;                                   // tap 9 has zero weights: read any valid pixel
      const int dh = (tap * 11) >> 5, dw = tap - 3 * dh;
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int q = wave * TM + j * 32 + r32;
        const int hr = ((q >> 4) * 2 + dh) * HW2 + (q & 15) * 2 + dw;
        const bf16x8 xf = *reinterpret_cast<const bf16x8*>(hbuf + hr * 16);
        acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[ks], xf, acc[j], 0, 0, 0);
      }
    }
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) {
        bf16x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = (bf16_t)apply_act(acc[j][g4 * 4 + e] + bias4[g4][e], d.act);
        *reinterpret_cast<bf16x4*>(stg + (j * 32 + r32) * SP + (g4 * 8 + khalf * 4) * 2) = o;
      }
    __builtin_amdgcn_wave_barrier();
    bf16_t* const ybase = reinterpret_cast<bf16_t*>(a.y) + d.out_c_offset + (lane & 3) * 8;
#pragma unroll
    for (int pass = 0; pass < 4; ++pass) {
      const int row = pass * 16 + (lane >> 2);
      const int q = wave * TM + row;
      const int yy = oy0 + (q >> 4), xx = ox0 + (q & 15);
      if (yy < d.ho && xx < d.wo)
        *reinterpret_cast<u32x4*>(ybase + ((long)(b * d.ho + yy) * d.wo + xx) * d.out_c_total) =
            *reinterpret_cast<const u32x4*>(stg + row * SP + (lane & 3) * 16);
    }
    __builtin_amdgcn_wave_barrier();
  }
}

template <int CIN, int COUT, bool POOL>
int launch_small_p(const ConvArgs& a, hipStream_t s) {
  constexpr int lds = ((18 * 18 * (CIN * 2 + (CIN == 32 ? YOLO_SMALL_PAD32 : 0)) + 1023) / 1024) * 1024 + (COUT / 8) * 64 * (32 * 2 + 16);
  static std::atomic<uint64_t> lds_set{0};                 // per device (common.h)
  if (const int rc = yolo_max_dyn_lds(reinterpret_cast<const void*>(&conv3x3_small_kernel<CIN, COUT, POOL>), lds, lds_set, "conv3x3_pool")) return rc;
  const YoloConvDesc& d = a.d;
  const long n_tiles = (long)d.n * ((d.h + 15) / 16) * ((d.w + 15) / 16);
  if (n_tiles > 0x7fffffffL) return yolo_set_error(YOLO_E_UNSUPPORTED, "conv grid too large");
  constexpr long kGrid = 256L * (COUT == 32 ? YOLO_SMALL_WGS : (YOLO_SMALL_WGS + 1) / 2 * 1);
  const long grid = n_tiles < kGrid ? n_tiles : kGrid;   // persistent: YOLO_SMALL_WGS workgroups of 4 waves (half as many of 8) per CU
  hipLaunchKernelGGL((conv3x3_small_kernel<CIN, COUT, POOL>), dim3((unsigned)grid), dim3(COUT * 8), lds, s, a);
  return yolo_check_launch("yolo_conv3x3_pool_fwd");
}

template <int CIN, int COUT>
int launch_small(const ConvArgs& a, bool pool, hipStream_t s) {
  return pool ? launch_small_p<CIN, COUT, true>(a, s) : launch_small_p<CIN, COUT, false>(a, s);
}

}  // namespace

// float32 NCHW input -> first conv layer at stride 2 (cout 32, pad 1).  Returns 1 if the shape is not covered.
int yolo_conv::launch_conv1_s2_nchw(const ConvArgs& a, const float* x_nchw, int cin_real, hipStream_t s) {
  const YoloConvDesc& d = a.d;
  if (d.ksize != 3 || d.stride != 2 || d.pad != 1 || d.cin != 8 || cin_real > 8 || d.cout != 32 || d.upsample2x ||
      d.out_dtype != YOLO_DT_BF16 || d.kpad < 80 || a.res || a.aux)
    return 1;
  const long grid = (long)d.n * ((d.ho + 15) / 16) * (((d.wo + 15) / 16 + 4) / 5);
  if (grid > 0x7fffffffL) return yolo_set_error(YOLO_E_UNSUPPORTED, "conv1 grid too large");
  if (cin_real == 3) hipLaunchKernelGGL((conv1_s2_nchw_kernel<3>), dim3((unsigned)grid), dim3(256), 0, s, a, x_nchw, cin_real);
  else hipLaunchKernelGGL((conv1_s2_nchw_kernel<0>), dim3((unsigned)grid), dim3(256), 0, s, a, x_nchw, cin_real);
  return yolo_check_launch("yolo_conv1_nchw_f32_fwd(stride 2)");
}

extern "C" int yolo_conv3x3_pool_supported(int cin, int cout) { return (cin == 16 || cin == 32) && (cout == 32 || cout == 64); }

extern "C" int yolo_conv3x3_pool_fwd(const void* x, const void* w_packed, const float* bias, void* y, const YoloConvDesc* dp,
                                     int pool, yolo_stream_t s) {
  YOLO_REQUIRE(x && w_packed && bias && y && dp, "conv3x3_pool: null pointer");
  const YoloConvDesc& d = *dp;
  YOLO_REQUIRE(yolo_conv3x3_pool_supported(d.cin, d.cout), "conv3x3_pool: cin %d / cout %d not covered (16 | 32 -> 32 | 64)", d.cin, d.cout);
  YOLO_REQUIRE(d.ksize == 3 && d.stride == 1 && d.pad == 1 && !d.upsample2x && d.out_dtype == YOLO_DT_BF16 && d.ho == d.h && d.wo == d.w,
               "conv3x3_pool: 3x3 / stride 1 / pad 1, bf16 output only");
  YOLO_REQUIRE(d.kpad >= 9 * d.cin && d.kpad % 8 == 0 && d.cout_pad >= d.cout, "conv3x3_pool: weights not packed for cin %d", d.cin);
  YOLO_REQUIRE(d.in_c_offset % 8 == 0 && d.in_c_total % 8 == 0 && d.in_c_offset + d.cin <= d.in_c_total, "conv3x3_pool: bad input view");
  YOLO_REQUIRE(d.out_c_offset % 8 == 0 && d.out_c_total % 8 == 0 && d.out_c_offset + d.cout <= d.out_c_total, "conv3x3_pool: bad output view");
  YOLO_REQUIRE(d.n > 0 && d.h > 0 && d.w > 0 && (!pool || (d.h >= 2 && d.w >= 2)), "conv3x3_pool: empty input");
  ConvArgs a;
  a.x = (const bf16_t*)x;
  a.w = (const bf16_t*)w_packed;
  a.bias = bias;
  a.res = nullptr;
  a.y = y;
  a.aux = nullptr;
  a.d = d;
  a.M = d.n * d.h * d.w;
  a.n_tiles = 1;
  a.steps = 0;
  a.x_bytes = 0;
  a.w_bytes = 0;
  a.debug = 0;
  YOLO_SET_STAMPS(a);
  hipStream_t st = (hipStream_t)s;
  if (d.cin == 16) return d.cout == 32 ? launch_small<16, 32>(a, pool != 0, st) : launch_small<16, 64>(a, pool != 0, st);
  return d.cout == 32 ? launch_small<32, 32>(a, pool != 0, st) : launch_small<32, 64>(a, pool != 0, st);
}
