// Halo-staged 3x3 / stride-1 / pad-1 convolution for gfx950 (MI355X), same contract and numerics as
// conv_igemm.hip (bf16 NHWC in, fp32 MFMA accumulate, fused epilogue) — used for the large feature maps.
//
// Why a second kernel: the gather implicit-GEMM stages the pixel operand once PER FILTER TAP (9x the input
// bytes through the L2 -> LDS path).  On the big maps (>= 80x80) that path, not the matrix pipe, is the
// limiter (profiles/r02_DESIGN_lab_notebook.md §3.1a: loads-only 0.095 ms vs MFMA-only 0.10 ms vs 0.31 ms total for 64->128 @160^2).
// Here a block owns a TH x TW output tile of ONE image and stages the (TH+2) x (TW+2) input halo tile once per
// cin chunk; the nine taps then read it at shifted LDS rows.  Only the weights are streamed per tap.
//
//   LDS:  halo [HB][(TH+2)(TW+2) rows][CK ch]   weights ring [2][BN rows][CK ch]     (rows XOR-swizzled)
//   step s = chunk*9 + tap:   s_waitcnt vmcnt(0) ; s_barrier ; issue weights(s+1) + a slice of halo(chunk+1)
//                             ; CK/16 x { ds_read_b128 fragments, MFMA 32x32x16 } .
// Image borders need no masks: halo pixels outside the image are fetched with an out-of-range buffer offset,
// which makes the LDS-DMA write zeros.
#include "conv_common.h"

using namespace yolo_conv;

namespace {

template <int TH, int TW, int BN, int WAVES_M, int WAVES_N, int CK, int HB, bool M16 = false>
__global__ __launch_bounds__(64 * WAVES_M * WAVES_N) void conv3x3_halo_kernel(const ConvArgs a) {
  YOLO_BLOCK_STAMP(a);
  constexpr int NW = WAVES_M * WAVES_N;
  constexpr int BM = TH * TW;
  constexpr int TM = BM / WAVES_M, TN = BN / WAVES_N;
  constexpr int NI = TM / 32, MI = TN / 32;
  constexpr int ROWB = CK * 2, CPR = CK / 8, RPP = 1024 / ROWB;     // LDS row bytes, chunks per row, rows per 1-KiB piece
  constexpr int HW2 = TW + 2;
  constexpr int HP = (TH + 2) * HW2;                                 // halo pixels
  constexpr int HPIECES = (HP + RPP - 1) / RPP;
  constexpr int HPT = (HPIECES + NW - 1) / NW;                       // halo pieces per wave
  constexpr int HALO_B = HPT * NW * 1024;                            // bytes per halo buffer (padded to whole pieces)
  constexpr int WPIECES = BN / RPP;
  constexpr int WIT = WPIECES / NW;                                  // weight pieces per wave per tap
  constexpr int WBUF_B = BN * ROWB;
  constexpr int RING_B = HB * HALO_B + 2 * WBUF_B;
  constexpr int LDS_B = RING_B > NW * TM * kEpiPitch ? RING_B : NW * TM * kEpiPitch;   // the epilogue slab reuses it
  constexpr int KS = CK / 16;
  static_assert(CK == 32 || CK == 64, "CK");
  static_assert(TM % 32 == 0 && TN % 32 == 0 && WPIECES % NW == 0 && WIT >= 1, "tile");
  static_assert(LDS_B <= 160 * 1024 && NW * TM * kEpiPitch <= LDS_B, "LDS");

  __shared__ __attribute__((aligned(16))) char smem[LDS_B];
  char* const s_halo = smem;
  char* const s_w = smem + HB * HALO_B;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WAVES_N, wn = wave % WAVES_N;
  const YoloConvDesc& d = a.d;

  // work item -> (image, tile row, tile col, cout tile); cout tile fastest so an XCD re-reads its halo from L2
  const int tiles_x = (d.w + TW - 1) / TW, tiles_y = (d.h + TH - 1) / TH;
  int b, y0, x0, n0;
  {
    int swz = xcd_swizzle(blockIdx.x, gridDim.x);
    n0 = (swz % a.n_tiles) * BN;
    swz /= a.n_tiles;
    x0 = (swz % tiles_x) * TW;
    swz /= tiles_x;
    y0 = (swz % tiles_y) * TH;
    b = swz / tiles_y;
  }

  const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)a.x, 0, a.x_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void*)a.w, 0, a.w_bytes, 0x00020000);

  // ---- LDS-DMA source offsets (per lane, constant over the whole K loop; chunk / tap go in the scalar offset)
  const int frow = lane / CPR;
  const int fsw = (CK == 32) ? swz32(lane >> 4) : (((lane >> 4) + 4 * (wave & 1)) & 7);   // f(row) of this lane's rows
  const int chunk = (lane & (CPR - 1)) ^ fsw;
  uint32_t h_off[HPT];
#pragma unroll
  for (int it = 0; it < HPT; ++it) {
    const int hr = (it * NW + wave) * RPP + frow;      // halo row (pixel) this lane fills
    const int hy = hr / HW2, hx = hr - hy * HW2;
    const int yy = y0 - 1 + hy, xx = x0 - 1 + hx;
    const bool ok = hr < HP && (unsigned)yy < (unsigned)d.h && (unsigned)xx < (unsigned)d.w;
    h_off[it] = ok ? (uint32_t)((((b * d.h + yy) * d.w + xx) * d.in_c_total + d.in_c_offset + chunk * 8) * 2) : kOobOffset;
  }
  uint32_t w_off[WIT];
#pragma unroll
  for (int it = 0; it < WIT; ++it)
    w_off[it] = (uint32_t)(((n0 + (it * NW + wave) * RPP + frow) * d.kpad + chunk * 8) * 2);

  auto issue_halo = [&](int hb, int c, int lo, int hi) {
    char* const base = s_halo + hb * HALO_B + wave * 1024;
#pragma unroll
    for (int it = 0; it < HPT; ++it)
      if (it >= lo && it < hi && !(a.debug & 1)) lds_dma16s(rx, base + it * (NW * 1024), h_off[it], (uint32_t)c * (CK * 2u));
  };
  auto issue_w = [&](int wb, int c, int tap) {
    char* const base = s_w + wb * WBUF_B + wave * 1024;
#pragma unroll
    for (int it = 0; it < WIT; ++it)
      if (!(a.debug & 2)) lds_dma16s(rw, base + it * (NW * 1024), w_off[it], (uint32_t)((tap * d.cin + c * CK) * 2));
  };

  // ---- fragment row bases: tile pixel q = wm*TM + j*32 + r32 -> halo row of tap (0,0)
  const int r32 = lane & 31, khalf = lane >> 5;
  int hrow0[NI];
#pragma unroll
  for (int j = 0; j < NI; ++j) {
    const int q = wm * TM + j * 32 + r32;
    hrow0[j] = (q / TW) * HW2 + (q % TW);
  }

  static_assert(!M16 || (CK == 64 && TW == 16), "16x16x32 form: 64-channel chunks, 16-wide tiles");
  constexpr int MI16 = M16 ? TN / 16 : 1, NI16 = M16 ? TM / 16 : 1;
  f32x16 acc[M16 ? 1 : MI][M16 ? 1 : NI];
  f32x4 acc16[MI16][NI16];
#pragma unroll
  for (int i = 0; i < (M16 ? 1 : MI); ++i)
#pragma unroll
    for (int j = 0; j < (M16 ? 1 : NI); ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
#pragma unroll
  for (int i = 0; i < MI16; ++i)
#pragma unroll
    for (int j = 0; j < NI16; ++j)
#pragma unroll
      for (int e = 0; e < 4; ++e) acc16[i][j][e] = 0.f;
  // 16x16x32 form: lane = (row c16 of a 16-block, 8-channel chunk q of a 32-channel K slice); a 16-pixel block is one
  // row of the 16-wide tile
  const int c16 = lane & 15, q16 = lane >> 4;
  const int hrow16 = (wm * (TM / 16)) * HW2 + c16;

  const int n_chunks = d.cin / CK;
  issue_halo(0, 0, 0, HPT);
  issue_w(0, 0, 0);

  auto pix_of = [&](int row) -> long {
    const int q = wm * TM + row;
    const int yy = y0 + q / TW, xx = x0 + q % TW;
    return (yy < d.h && xx < d.w) ? ((long)(b * d.h + yy) * d.w + xx) : -1L;
  };

  int wb = 0;
  for (int c = 0; c < n_chunks; ++c) {
    const char* const hbuf = s_halo + (HB == 2 ? (c & 1) : 0) * HALO_B;
    const bool next_chunk = c + 1 < n_chunks;
#pragma unroll 1
    for (int tap = 0; tap < 9; ++tap) {
      wait_vmcnt<0>();
      __builtin_amdgcn_s_barrier();       // stage (c,tap) is in LDS; every wave finished reading stage (c,tap)-1
      // prefetch: weights of the next step, and 1/9 of the next chunk's halo (HB == 2)
      if (tap < 8) {
        issue_w(wb ^ 1, c, tap + 1);
      } else if (next_chunk) {
        issue_w(wb ^ 1, c + 1, 0);
      }
      if (HB == 2 && next_chunk) issue_halo((c + 1) & 1, c + 1, tap * HPT / 9, (tap + 1) * HPT / 9);
      const int dh = (tap * 11) >> 5, dw = tap - 3 * dh;
      const int toff = dh * HW2 + dw;
      const char* const wbuf = s_w + wb * WBUF_B;
      if constexpr (M16) {
#pragma unroll
        for (int kk = 0; kk < CK / 32; ++kk) {
          const int g = kk * 4 + q16;
          bf16x8 wf[MI16], xf[NI16];
#pragma unroll
          for (int i = 0; i < MI16; ++i) {
            const int R = wn * TN + i * 16 + c16;
            wf[i] = *reinterpret_cast<const bf16x8*>(wbuf + R * ROWB + ((g ^ ((R >> 1) & 7)) << 4));
          }
#pragma unroll
          for (int j = 0; j < NI16; ++j) {
            const int R = hrow16 + j * HW2 + toff;
            xf[j] = *reinterpret_cast<const bf16x8*>(hbuf + R * ROWB + ((g ^ ((R >> 1) & 7)) << 4));
          }
#pragma unroll
          for (int i = 0; i < MI16; ++i)
#pragma unroll
            for (int j = 0; j < NI16; ++j)
              acc16[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[i], xf[j], acc16[i][j], 0, 0, 0);
        }
      } else if (!(a.debug & 4)) {
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
          const int g = ks * 2 + khalf;
          bf16x8 wf[MI], xf[NI];
#pragma unroll
          for (int i = 0; i < MI; ++i) {
            const int R = wn * TN + i * 32 + r32;
            const int sw = (CK == 32) ? swz32(R >> 2) : ((R >> 1) & 7);
            wf[i] = *reinterpret_cast<const bf16x8*>(wbuf + R * ROWB + ((g ^ sw) << 4));
          }
#pragma unroll
          for (int j = 0; j < NI; ++j) {
            const int R = hrow0[j] + toff;
            const int sw = (CK == 32) ? swz32(R >> 2) : ((R >> 1) & 7);
            xf[j] = *reinterpret_cast<const bf16x8*>(hbuf + R * ROWB + ((g ^ sw) << 4));
          }
#pragma unroll
          for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < NI; ++j)
              acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[i], xf[j], acc[i][j], 0, 0, 0);
        }
      }
      wb ^= 1;
    }
    if (HB == 1 && next_chunk) {          // single halo buffer: reload between chunks (only used when it must fit)
      __builtin_amdgcn_s_barrier();
      issue_halo(0, c + 1, 0, HPT);
    }
  }

  if (a.debug & 8) {
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int j = 0; j < NI; ++j) {
#if defined(__HIP_DEVICE_COMPILE__)
        asm volatile("" ::"v"(acc[i][j]));
#endif
      }
    return;
  }
  __syncthreads();
  if constexpr (M16) epilogue_lds16<MI, NI16, TM>(a, acc16, smem + wave * (TM * kEpiPitch), lane, n0 + wn * TN, pix_of);
  else epilogue_lds<MI, NI, TM>(a, acc, smem + wave * (TM * kEpiPitch), lane, n0 + wn * TN, pix_of);
}

// ---------------------------------------------------------------------------------------------------
// First layer, fused with the input packing: x is the caller's float32 NCHW image batch (c <= 8 channels,
// reference input contract utils/dataset_csv.py:79-87), y = act(conv3x3/s1(x) + bias) in bf16 NHWC, cout = 32.
// One block = one 16x16 output tile: the 18x18 halo is read from the c planes, rounded to bf16 and laid out
// NHWC8 in LDS (16 B per pixel); K = 10 taps x 8 channels (tap 9 and channels >= c carry zero weights), i.e.
// five 32x32x16 MFMAs per 32 pixels with the weight fragments held in registers.  The layer is HBM-bound
// (write 64 B per pixel); the implicit-GEMM kernel spent its time issuing nine 16-byte gathers per pixel.
// Tiles per workgroup, swept on YOLOv3-tiny's first layer at 32 images of 416x416 (round 5, profiles/r05_conv1_tiles_per_workgroup.txt):
// 2: 0.088 ms, 3: 0.071, 4: 0.071, 5 (rounds 3-4): 0.077, 6: 0.087, 7: 0.094 - more, shorter workgroups hide one another's start-up
// better than longer runs amortise it; end to end 3: 90.5 k, 4: 89.1 k, 5: 88.2 k images/s.
#ifndef YOLO_CONV1_TPB
#define YOLO_CONV1_TPB 3
#endif
constexpr int kConv1TilesPerBlock = YOLO_CONV1_TPB;

// COUT = 32 or 16 output channels (rows COUT..31 of the MFMA tile carry zero weights); POOL fuses the MaxPool2d(2, 2)
// that follows the first ConvBlock of YOLOv3-tiny (reference models/yolo_base.py:69-80, yolov3_tiny.py:26): the 16 x 16
// tile is pooled out of the LDS staging and only the pooled map is written.
// CINR: the real channel count when known at compile time (3: RGB, every model here), 0: taken from cin_real; with it the
// halo fetch is 3 loads and 3 conversions per pixel instead of 8 predicated ones.
template <int COUT, bool POOL, int CINR = 0>
__global__ __launch_bounds__(256) void conv1_nchw_kernel(const ConvArgs a, const float* __restrict__ x_nchw, int cin_real_arg) {
  const int cin_real = CINR ? CINR : cin_real_arg;
  constexpr int HW2 = 18, HP = 18 * 18, TM = 64;
  constexpr int SP = COUT * 2 + 16;                           // bf16 staging pitch: COUT couts + 16 B pad
  constexpr int LPP = COUT / 8;                               // 16-byte lanes per pixel
  constexpr int HALO_B = ((HP * 16 + 1023) / 1024) * 1024;    // 6 KB
  constexpr int LDS_B = 2 * HALO_B + 4 * TM * SP;             // 32.5 KB (COUT 32)
  constexpr int TPB = kConv1TilesPerBlock;
  __shared__ __attribute__((aligned(16))) char smem[LDS_B];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const YoloConvDesc& d = a.d;
  const int tiles_x = (d.w + 15) / 16, tiles_y = (d.h + 15) / 16;
  const int groups_x = (tiles_x + TPB - 1) / TPB;
  int swz = xcd_swizzle(blockIdx.x, gridDim.x);
  const int tx0 = (swz % groups_x) * TPB;
  swz /= groups_x;
  const int y0 = (swz % tiles_y) * 16;
  const int b = swz / tiles_y;
  const int n_t = min(TPB, tiles_x - tx0);

  // weight fragments (A operand): lane = cout r32, k = ks*16 + khalf*8 + 0..7 of the packed [cout_pad][kpad] matrix
  const int r32 = lane & 31, khalf = lane >> 5;
  bf16x8 wf[5];
#pragma unroll
  for (int ks = 0; ks < 5; ++ks) wf[ks] = *reinterpret_cast<const bf16x8*>(a.w + (long)r32 * d.kpad + ks * 16 + khalf * 8);
  f32x4 bias4[4];
#pragma unroll
  for (int g4 = 0; g4 < 4; ++g4) bias4[g4] = *reinterpret_cast<const f32x4*>(a.bias + g4 * 8 + khalf * 4);

  // halo pixels owned by this thread (2 of the 324).  Round 4: the halos of ALL the block's tiles are requested up front (TPB x 2
  // pixels x cin_real floats in registers; CINR == 3: 30 registers) instead of one tile ahead: with one tile in flight a workgroup
  // had 3.9 KB outstanding - 16 KB per CU at four workgroups -, and the layer (read 12 B, write 32 B per pooled pixel: HBM traffic,
  // not MFMA work) ran at 1.1 TB/s, a latency chain of one HBM round trip per tile.  Vector-memory operations return in issue order,
  // so tile t's commit waits for tile t's loads only.
  const long plane = (long)d.h * d.w;
  const int hp0 = tid, hp1 = tid + 256;
  const int hy0 = hp0 / HW2, hx0 = hp0 - hy0 * HW2, hy1 = hp1 / HW2, hx1 = hp1 - hy1 * HW2;
  constexpr int NCH = CINR ? CINR : 8;
  float pre[TPB][2][NCH];
  auto fetch = [&](int t, float (&dst)[2][NCH]) {
    const int x0 = (tx0 + t) * 16;
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int hy = u ? hy1 : hy0, hx = u ? hx1 : hx0;
      const int yy = y0 - 1 + hy, xx = x0 - 1 + hx;
      const bool ok = (u == 0 || hp1 < HP) && (unsigned)yy < (unsigned)d.h && (unsigned)xx < (unsigned)d.w;
      const float* src = x_nchw + ((long)b * cin_real) * plane + (long)yy * d.w + xx;
#pragma unroll
      for (int e = 0; e < NCH; ++e) dst[u][e] = (ok && e < cin_real) ? src[e * plane] : 0.f;
    }
  };
  auto commit = [&](int hb, const float (&src)[2][NCH]) {       // registers -> LDS halo buffer as NHWC8 bf16
    char* const hbuf = smem + hb * HALO_B;
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      if (u == 1 && hp1 >= HP) continue;
      bf16x8 v;
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = e < NCH ? (bf16_t)src[u][e < NCH ? e : 0] : (bf16_t)0.f;
      *reinterpret_cast<bf16x8*>(hbuf + (u ? hp1 : hp0) * 16) = v;
    }
  };

  char* const stg = smem + 2 * HALO_B + wave * (TM * SP);
#pragma unroll
  for (int t = 0; t < TPB; ++t)
    if (t < n_t) fetch(t, pre[t]);
#pragma unroll
  for (int t = 0; t < TPB; ++t) {
    if (t >= n_t) break;                   // (uniform)
    commit(t & 1, pre[t]);
    wait_lds();                            // LDS-only barrier: the previous tile's stores stay in flight
    __builtin_amdgcn_s_barrier();          // halo t visible; (also: everyone is past the reads of halo t-2's buffer)
    const char* const hbuf = smem + (t & 1) * HALO_B;
    const int x0 = (tx0 + t) * 16;
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
        const int hr = ((q >> 4) + dh) * HW2 + (q & 15) + dw;
        const bf16x8 xf = *reinterpret_cast<const bf16x8*>(hbuf + hr * 16);
        acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[ks], xf, acc[j], 0, 0, 0);
      }
    }
    // epilogue: no residual / pre-add copy here, so the tile is rounded to bf16 before staging (single rounding);
    // lane = pixel, registers = couts -> LDS [pixel][32 couts] -> 16 B per lane, 1 KiB contiguous per instruction
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int g4 = 0; g4 < COUT / 8; ++g4) {
        bf16x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = (bf16_t)apply_act(acc[j][g4 * 4 + e] + bias4[g4][e], d.act);
        *reinterpret_cast<bf16x4*>(stg + (j * 32 + r32) * SP + (g4 * 8 + khalf * 4) * 2) = o;
      }
    __builtin_amdgcn_wave_barrier();
    if constexpr (POOL) {
      // the wave's 64 pixels are 4 tile rows x 16 columns -> 2 x 8 pooled pixels; lane = (pooled pixel, 8-channel chunk)
      const int pp = lane / LPP, chunk = lane % LPP;
      const int hp = d.h >> 1, wp = d.w >> 1;                 // MaxPool2d(2, 2): floor
      if (pp < 16) {
        const int pr = pp >> 3, pc = pp & 7;
        const int py = (y0 >> 1) + wave * 2 + pr, px = (x0 >> 1) + pc;
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
          for (int e = 0; e < 8; ++e) o[e] = (bf16_t)m[e];    // exact: the maximum is one of the bf16 inputs
          *reinterpret_cast<bf16x8*>(reinterpret_cast<bf16_t*>(a.y) + ((long)(b * hp + py) * wp + px) * d.out_c_total +
                                     d.out_c_offset + chunk * 8) = o;
        }
      }
    } else {
      bf16_t* const ybase = reinterpret_cast<bf16_t*>(a.y) + d.out_c_offset + (lane % LPP) * 8;
#pragma unroll
      for (int pass = 0; pass < TM / (64 / LPP); ++pass) {
        const int row = pass * (64 / LPP) + lane / LPP;
        const int q = wave * TM + row;
        const int yy = y0 + (q >> 4), xx = x0 + (q & 15);
        if (yy < d.h && xx < d.w) {
          const u32x4 val = *reinterpret_cast<const u32x4*>(stg + row * SP + (lane % LPP) * 16);
          *reinterpret_cast<u32x4*>(ybase + ((long)(b * d.h + yy) * d.w + xx) * d.out_c_total) = val;
        }
      }
    }
    __builtin_amdgcn_wave_barrier();
  }
}

template <int TH, int TW, int BN, int WAVES_M, int WAVES_N, int CK, int HB, bool M16 = false>
int launch(const ConvArgs& a, hipStream_t s) {
  ConvArgs b = a;
  b.n_tiles = (a.d.cout + BN - 1) / BN;
  const long grid = (long)a.d.n * ((a.d.h + TH - 1) / TH) * ((a.d.w + TW - 1) / TW) * b.n_tiles;
  if (grid > 0x7fffffffL) return yolo_set_error(YOLO_E_UNSUPPORTED, "conv grid too large");
  if (pick_only("halo<%dx%d,%d couts,CK%d,%d halo buffers%s> grid %ld", TH, TW, BN, CK, HB, M16 ? ",16x16x32" : "", grid)) return 0;
  hipLaunchKernelGGL((conv3x3_halo_kernel<TH, TW, BN, WAVES_M, WAVES_N, CK, HB, M16>), dim3((unsigned)grid),
                     dim3(64 * WAVES_M * WAVES_N), 0, s, b);
  return yolo_check_launch("yolo_conv2d_fwd(halo)");
}

}  // namespace

// Returns 1 when the halo kernel does not apply (caller falls back to the gather kernel), else the launch status.
int yolo_conv::launch_halo3x3(const ConvArgs& a, hipStream_t s) {
  const YoloConvDesc& d = a.d;
  if (d.ksize != 3 || d.stride != 1 || d.pad != 1 || d.upsample2x || d.out_dtype != YOLO_DT_BF16) return 1;
  if (d.cin % 32 != 0 || d.cout % 64 != 0) return 1;
  // 16x16 output tiles: only worth it when the map tiles well (partial tiles idle lanes) and is large
  const long tiles = (long)((d.h + 15) / 16) * ((d.w + 15) / 16);
  if ((double)d.h * d.w < 0.85 * 256.0 * tiles || (long)d.h * d.w < 80 * 80) return 1;
  if (d.cin % 64 == 0) {
    const bool one = d.cin == 64;
    // 128 couts per block with ONE halo buffer = 73 KB of LDS, i.e. two blocks per CU: they drift apart, so one
    // block's epilogue (a third of a layer's time on the 80x80 maps) runs under the other's MFMA loop.  Measured
    // -12..-14 % per layer against 256 couts per block (146 KB, one block per CU); YOLO_CONV_DEBUG bit 1024 selects
    // the old form.  (128 couts with two halo buffers, 114 KB, is the worst of the three.)
    if (d.cout % 256 == 0 && (a.debug & 1024)) return one ? launch<16, 16, 256, 4, 2, 64, 1>(a, s) : launch<16, 16, 256, 4, 2, 64, 2>(a, s);
    // v_mfma_f32_16x16x32_bf16 form (the chip holds a higher clock on it): -6 % per layer; bit 65536 selects 32x32x16
    if (d.cout % 128 == 0 && !(a.debug & 65536)) return launch<16, 16, 128, 4, 2, 64, 1, true>(a, s);
    if (d.cout % 128 == 0) return launch<16, 16, 128, 4, 2, 64, 1>(a, s);
    return one ? launch<16, 16, 64, 4, 1, 64, 1>(a, s) : launch<16, 16, 64, 4, 1, 64, 2>(a, s);
  }
  const bool one = d.cin == 32;
  if (d.cout % 128 == 0) return one ? launch<16, 16, 128, 4, 2, 32, 1>(a, s) : launch<16, 16, 128, 4, 2, 32, 2>(a, s);
  return one ? launch<16, 16, 64, 4, 1, 32, 1>(a, s) : launch<16, 16, 64, 4, 1, 32, 2>(a, s);
}


// float32 NCHW input -> first conv layer (see conv1_nchw_kernel).  Returns 1 if the shape is not covered.
int yolo_conv::launch_conv1_nchw(const ConvArgs& a, const float* x_nchw, int cin_real, bool pool, hipStream_t s) {
  const YoloConvDesc& d = a.d;
  if (d.ksize != 3 || d.stride != 1 || d.pad != 1 || d.cin != 8 || cin_real > 8 || (d.cout != 32 && d.cout != 16) || d.upsample2x ||
      d.out_dtype != YOLO_DT_BF16 || d.kpad < 80 || a.res || a.aux)
    return 1;
  if (pool && (d.h < 2 || d.w < 2)) return 1;
  const int tiles_x = (d.w + 15) / 16;
  const long grid = (long)d.n * ((d.h + 15) / 16) * ((tiles_x + kConv1TilesPerBlock - 1) / kConv1TilesPerBlock);
  if (grid > 0x7fffffffL) return yolo_set_error(YOLO_E_UNSUPPORTED, "conv1 grid too large");
  const dim3 g((unsigned)grid), blk(256);
#define YOLO_CONV1(CO, PL)                                                                            \
  do {                                                                                                  \
    if (cin_real == 3) hipLaunchKernelGGL((conv1_nchw_kernel<CO, PL, 3>), g, blk, 0, s, a, x_nchw, cin_real); \
    else hipLaunchKernelGGL((conv1_nchw_kernel<CO, PL, 0>), g, blk, 0, s, a, x_nchw, cin_real);              \
  } while (0)
  if (d.cout == 32) {
    if (pool) YOLO_CONV1(32, true);
    else YOLO_CONV1(32, false);
  } else {
    if (pool) YOLO_CONV1(16, true);
    else YOLO_CONV1(16, false);
  }
#undef YOLO_CONV1
  return yolo_check_launch("yolo_conv1_nchw_f32_fwd");
}
