// Fused Darknet residual unit for gfx950 (MI355X):
//     y = x + act(conv3x3(act(conv1x1(x, W1) + b1), W2) + b2)           (reference models/yolov3_spp.py:17-32,
//                                                                        the `Add` of a DownSample unit)
// Same numerics as running the two layers through conv_igemm.hip / conv3x3_halo.hip: bf16 NHWC in, fp32 MFMA
// accumulation, the 1x1 output rounded to bf16 once, the sum formed in fp32 and rounded once.
//
// Why fuse: on the large maps the unit is bound by what the two kernels move, not by the matrix pipe: the
// C/2-channel intermediate is written to and re-read from HBM/L2, and each kernel pays its own prologue,
// epilogue and tail.  Here a block owns a 16x16 output tile of one image and all C output channels:
//   phase A  mid[18x18 halo][C/2] = act(W1 . x_halo + b1) by MFMA straight into LDS, in the halo layout of
//            conv3x3_halo.hip (pixels outside the image are ZERO there: the 3x3 zero-pads `mid`, not x);
//   phase B  the nine taps read `mid` at shifted LDS rows, only W2 is streamed (one 2-stage ring);
//   epilogue +b2, act, optional pre-add copy, + x (the tile's own pixels, L2-hot), one coalesced store.
// The halo costs (18/16)^2 = 1.27x recompute of the 1x1 (~11% of the unit's FLOPs) and removes the
// intermediate's round trip.  y must not alias x (neighbouring blocks read x's halo).
#include "conv_common.h"

using namespace yolo_conv;

namespace {

struct ResUnitArgs {
  ConvArgs c;           // the 3x3: w = W2, bias = b2, res = x view, y, aux; d.cin = C/2 (mid), d.cout = C
  const bf16_t* w1;     // packed [cout_pad1][kpad1], K = C
  const float* b1;
  int kpad1;
  uint32_t w1_bytes;
};

template <int C>
__global__ __launch_bounds__(512) void resunit_kernel(const ResUnitArgs ra) {
  constexpr int NW = 8;
  constexpr int CM = C / 2;                          // mid channels
  constexpr int CK = CM >= 64 ? 64 : 32;             // phase B cin chunk
  constexpr int NCH = CM / CK;
  constexpr int WAVES_N = C >= 128 ? 2 : 1, WAVES_M = NW / WAVES_N;
  constexpr int BN = C, BM = 256, TW = 16, HW2 = 18, HP = 18 * 18;
  constexpr int TM = BM / WAVES_M, TN = BN / WAVES_N;
  constexpr int NI = TM / 32, MI = TN / 32;
  constexpr int ROWB = CK * 2, CPR = CK / 8, RPP = 1024 / ROWB;
  constexpr int HALO_B = ((HP * ROWB + 1023) / 1024) * 1024;
  constexpr int WPIECES = BN / RPP, WIT = WPIECES >= NW ? WPIECES / NW : 1;   // C = 64: only waves 0..3 carry W2 pieces
  constexpr int WBUF_B = BN * ROWB;
  constexpr int KS = CK / 16;
  // phase A: K chunks of 32 channels (64-byte rows, 16 rows per 1-KiB piece); 352 halo rows padded to 384
  constexpr int A_XPT = 3;                           // x pieces per wave per step (24 pieces = 384 rows)
  constexpr int A_XB = A_XPT * NW * 1024;            // 24 KiB
  constexpr int A_WPIECES = CM / 16;                 // <= 8
  constexpr int A_WB = A_WPIECES * 1024;
  constexpr int A_STAGE = A_XB + A_WB;
  constexpr int MI1 = CM / 32;
  constexpr int A_STEPS = C / 32;
  constexpr int RING_B = 2 * WBUF_B > 2 * A_STAGE ? 2 * WBUF_B : 2 * A_STAGE;
  constexpr int MAIN_B = NCH * HALO_B + RING_B;
  constexpr int LDS_B = MAIN_B > NW * TM * kEpiPitch ? MAIN_B : NW * TM * kEpiPitch;
  static_assert(C == 64 || C == 128 || C == 256, "C");
  static_assert((WPIECES % NW == 0 || WPIECES < NW) && A_WPIECES <= NW, "pieces");
  static_assert(LDS_B <= 160 * 1024, "LDS");
  // when all 9 * NCH tap slices of W2 fit the (now idle) phase-A staging region they are fetched in one go: one
  // load latency for phase B instead of one per tap, and no barriers inside the tap loop
  constexpr bool ALLW = 9 * NCH * WBUF_B <= RING_B;

  __shared__ __attribute__((aligned(16))) char smem[LDS_B];
  char* const s_mid = smem;
  char* const s_ring = smem + NCH * HALO_B;

  const ConvArgs& a = ra.c;
  const YoloConvDesc& d = a.d;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WAVES_N, wn = wave % WAVES_N;
  const int r32 = lane & 31, khalf = lane >> 5;

  const int tiles_x = (d.w + 15) / 16, tiles_y = (d.h + 15) / 16;
  int b, y0, x0;
  {
    int swz = xcd_swizzle(blockIdx.x, gridDim.x);
    x0 = (swz % tiles_x) * 16;
    swz /= tiles_x;
    y0 = (swz % tiles_y) * 16;
    b = swz / tiles_y;
  }

  const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)a.x, 0, a.x_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rw1 = __builtin_amdgcn_make_buffer_rsrc((void*)ra.w1, 0, ra.w1_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rw2 = __builtin_amdgcn_make_buffer_rsrc((void*)a.w, 0, a.w_bytes, 0x00020000);

  // C = 64: the residual (= the tile's own x pixels) is picked out of the phase-A staging while it is still in LDS —
  // both K steps of x are resident then — instead of being re-read from global in the epilogue, where four
  // dependent load -> add -> store rounds per wave were half of this kernel's time (ablation: profiles/r02_DESIGN_lab_notebook.md 3.1d)
  ResPrefetch<MI, TM> rp;
  constexpr bool RES_FROM_LDS = C == 64;

  // ================================ phase A: mid = act(W1 . x_halo + b1) ================================
  {
    const int frow = lane >> 2;                                    // 16 rows x 4 sixteen-byte chunks per piece
    const int chunk = (lane & 3) ^ ((lane >> 4) & 3);              // source chunk of this lane's LDS slot (64-B rows)
    uint32_t xa_off[A_XPT];
#pragma unroll
    for (int it = 0; it < A_XPT; ++it) {
      const int hr = (it * NW + wave) * 16 + frow;
      const int hy = hr / HW2, hx = hr - hy * HW2;
      const int yy = y0 - 1 + hy, xx = x0 - 1 + hx;
      const bool ok = hr < HP && (unsigned)yy < (unsigned)d.h && (unsigned)xx < (unsigned)d.w;
      xa_off[it] = ok ? (uint32_t)((((b * d.h + yy) * d.w + xx) * d.in_c_total + d.in_c_offset + chunk * 8) * 2) : kOobOffset;
    }
    const uint32_t wa_off = wave < A_WPIECES ? (uint32_t)(((wave * 16 + frow) * ra.kpad1 + chunk * 8) * 2) : kOobOffset;
    auto issue_a = [&](int st, int step) {
      char* const base = s_ring + st * A_STAGE;
#pragma unroll
      for (int it = 0; it < A_XPT; ++it)
        if (!(a.debug & 8)) lds_dma16s(rx, base + (it * NW + wave) * 1024, xa_off[it], (uint32_t)step * 64u);
      if (wave < A_WPIECES) lds_dma16s(rw1, base + A_XB + wave * 1024, wa_off, (uint32_t)step * 64u);
    };

    // wave -> pixel blocks {wave, wave + 8} of the 11 (x 32 halo rows)
    const bool two = wave + 8 < 11;
    f32x16 acc1[2][MI1];
#pragma unroll
    for (int p = 0; p < 2; ++p)
#pragma unroll
      for (int i = 0; i < MI1; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc1[p][i][e] = 0.f;

    issue_a(0, 0);
    issue_a(1, 1);                            // both ring slots are empty: two K steps go out together (A_STEPS >= 2)
#pragma unroll 1
    for (int step = 0; step < A_STEPS; ++step) {
      if (step == 0) {                        // stage 1 may stay in flight (3 x pieces, +1 W1 piece on the first waves)
        if (wave < A_WPIECES) wait_vmcnt<A_XPT + 1>();
        else wait_vmcnt<A_XPT>();
      } else {
        wait_vmcnt<0>();
      }
      __builtin_amdgcn_s_barrier();
      if (step >= 1 && step + 1 < A_STEPS) issue_a((step + 1) & 1, step + 1);
      const char* const xb = s_ring + (step & 1) * A_STAGE;
      const char* const wbuf = xb + A_XB;
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        const int g = ks * 2 + khalf;
        bf16x8 wf[MI1], xf[2];
#pragma unroll
        for (int i = 0; i < MI1; ++i) {
          const int R = i * 32 + r32;
          wf[i] = *reinterpret_cast<const bf16x8*>(wbuf + R * 64 + ((g ^ ((R >> 2) & 3)) << 4));
        }
#pragma unroll
        for (int p = 0; p < 2; ++p) {
          const int R = (wave + 8 * p) * 32 + r32;                 // p == 1 of waves 3..7 reads the zero padding rows
          xf[p] = *reinterpret_cast<const bf16x8*>(xb + (R < 384 ? R : 0) * 64 + ((g ^ ((R >> 2) & 3)) << 4));
        }
#pragma unroll
        for (int i = 0; i < MI1 && !(a.debug & 1); ++i) {
          acc1[0][i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[i], xf[0], acc1[0][i], 0, 0, 0);
          if (two) acc1[1][i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[i], xf[1], acc1[1][i], 0, 0, 0);
        }
      }
    }
    if constexpr (RES_FROM_LDS) {
      // lane mapping of the (paired) epilogue: per half of the wave's 32 pixel rows, 8 rows x 8 chunks of 8 channels
      // per store instruction; chunk c = channels [8c, 8c+8) = K step c/4 of phase A, 16-byte chunk c%4 of its row
      static_assert(MI == 2 && TM == 32, "C = 64 layout");
      const int crow = lane >> 3, cchunk = lane & 7;
      int idx = 0;
#pragma unroll
      for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int pass = 0; pass < 2; ++pass, ++idx) {
          const int q = wm * TM + h * 16 + pass * 8 + crow;
          const int hr = (q / TW + 1) * HW2 + (q % TW) + 1;
          rp.v[idx] = *reinterpret_cast<const bf16x8*>(s_ring + (cchunk >> 2) * A_STAGE + hr * 64 +
                                                       (((cchunk & 3) ^ ((hr >> 2) & 3)) << 4));
        }
    }
    __builtin_amdgcn_s_barrier();            // every wave is done with the phase-A stages: the ring may be reused
    // first W2 stage flies while `mid` is written (issued below, after the offsets are set up)

    // mid rows: lane = halo pixel, registers = mid channels (e&3) + 8*(e>>2) + 4*khalf
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      if ((p == 1 && !two) || (a.debug & 16)) break;
      const int hr = (wave + 8 * p) * 32 + r32;
      const int hy = hr / HW2, hx = hr - hy * HW2;
      const int yy = y0 - 1 + hy, xx = x0 - 1 + hx;
      const bool inside = (unsigned)yy < (unsigned)d.h && (unsigned)xx < (unsigned)d.w;
      if (hr < HP) {
        const int sw = (CK == 32) ? ((hr >> 2) & 3) : ((hr >> 1) & 7);
#pragma unroll
        for (int i = 0; i < MI1; ++i)
#pragma unroll
          for (int g4 = 0; g4 < 4; ++g4) {
            const int cm = i * 32 + g4 * 8 + khalf * 4;            // first of 4 consecutive mid channels
            const f32x4 bv = *reinterpret_cast<const f32x4*>(ra.b1 + cm);
            bf16x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = (bf16_t)(inside ? apply_act(acc1[p][i][g4 * 4 + e] + bv[e], d.act) : 0.f);
            const int slot = (cm % CK) >> 3;
            *reinterpret_cast<bf16x4*>(s_mid + (cm / CK) * HALO_B + hr * ROWB + ((slot ^ sw) << 4) + khalf * 8) = o;
          }
      }
    }
  }

  // ================================ phase B: 3x3 over `mid`, W2 streamed ================================
  const int frow = lane / CPR;
  const int fsw = (CK == 32) ? ((lane >> 4) & 3) : (((lane >> 4) + 4 * (wave & 1)) & 7);
  const int chunk = (lane & (CPR - 1)) ^ fsw;
  uint32_t w_off[WIT];
#pragma unroll
  for (int it = 0; it < WIT; ++it) w_off[it] = (uint32_t)((((it * NW + wave) * RPP + frow) * d.kpad + chunk * 8) * 2);
  auto issue_w = [&](int wb, int c, int tap) {
    char* const base = s_ring + wb * WBUF_B + wave * 1024;
    if (WPIECES < NW && wave >= WPIECES) return;
#pragma unroll
    for (int it = 0; it < WIT; ++it) lds_dma16s(rw2, base + it * (NW * 1024), w_off[it], (uint32_t)((tap * CM + c * CK) * 2));
  };
  if constexpr (ALLW) {
#pragma unroll
    for (int c = 0; c < NCH; ++c)
#pragma unroll
      for (int tap = 0; tap < 9; ++tap) issue_w(c * 9 + tap, c, tap);
  } else {
    issue_w(0, 0, 0);
  }
  __syncthreads();                        // `mid` (plain LDS stores of every wave) visible to all

  int hrow0[NI];
#pragma unroll
  for (int j = 0; j < NI; ++j) {
    const int q = wm * TM + j * 32 + r32;
    hrow0[j] = (q / TW) * HW2 + (q % TW);
  }
  f32x16 acc[MI][NI];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NI; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  int wb = 0;
#pragma unroll 1
  for (int c = 0; c < NCH; ++c) {
    const char* const hbuf = s_mid + c * HALO_B;
#pragma unroll 1
    for (int tap = 0; tap < 9; ++tap) {
      if (!ALLW || (c == 0 && tap == 0)) {
        wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();     // W2 stage (c,tap) landed — with ALLW: every slice of every wave
      }
      if constexpr (!ALLW) {
        if (tap < 8) {
          issue_w(wb ^ 1, c, tap + 1);
        } else if (c + 1 < NCH) {
          issue_w(wb ^ 1, c + 1, 0);
        }
      }
      const int dh = (tap * 11) >> 5, dw = tap - 3 * dh;
      const int toff = dh * HW2 + dw;
      const char* const wbuf = s_ring + (ALLW ? c * 9 + tap : wb) * WBUF_B;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        const int g = ks * 2 + khalf;
        bf16x8 wf[MI], xf[NI];
#pragma unroll
        for (int i = 0; i < MI; ++i) {
          const int R = wn * TN + i * 32 + r32;
          const int sw = (CK == 32) ? ((R >> 2) & 3) : ((R >> 1) & 7);
          wf[i] = *reinterpret_cast<const bf16x8*>(wbuf + R * ROWB + ((g ^ sw) << 4));
        }
#pragma unroll
        for (int j = 0; j < NI; ++j) {
          const int R = hrow0[j] + toff;
          const int sw = (CK == 32) ? ((R >> 2) & 3) : ((R >> 1) & 7);
          xf[j] = *reinterpret_cast<const bf16x8*>(hbuf + R * ROWB + ((g ^ sw) << 4));
        }
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
          for (int j = 0; j < NI; ++j)
            if (!(a.debug & 2)) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[i], xf[j], acc[i][j], 0, 0, 0);
      }
      wb ^= 1;
    }
  }

  auto pix_of = [&](int row) -> long {
    const int q = wm * TM + row;
    const int yy = y0 + q / TW, xx = x0 + q % TW;
    return (yy < d.h && xx < d.w) ? ((long)(b * d.h + yy) * d.w + xx) : -1L;
  };
  __syncthreads();
  if (a.debug & 4) {
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
  epilogue_lds<MI, NI, TM>(a, acc, smem + wave * (TM * kEpiPitch), lane, wn * TN, pix_of, RES_FROM_LDS ? &rp : nullptr);
}

// ---------------------------------------------------------------------------------------------------------------
// C = 64 (the unit on the 320 x 320 map): persistent, software-pipelined form.  The generic kernel above is a chain of
// dependent waits per tile (x halo from HBM -> phase A -> W2 -> phase B -> epilogue) with only two blocks per CU to
// hide them; ablation: 0.22 ms against an HBM floor of 0.095.  Here ONE block per CU walks its share of the tile list
// with W1 / W2 / biases resident for the whole block (40 KB, loaded once) and the x halo of tile t+1 streaming into
// LDS (LDS-DMA, one 128-byte row per halo pixel) while tile t runs phase B and its epilogue — no wait inside a tile
// is longer than an LDS round trip.  The residual is taken from the x halo in LDS before it is overwritten.
//   LDS: mid 21 KB | x halo 48 KB | W1 4 KB | W2 36 KB | per-wave fp32 store staging 34 KB   (144 KB)
__global__ __launch_bounds__(512) void resunit64_kernel(const ResUnitArgs ra) {
  constexpr int NW = 8, C = 64, CM = 32, TW = 16, HW2 = 18, HP = 18 * 18;
  constexpr int TM = 32, MI = 2;                       // wave tile of phase B: 32 pixels x 64 couts
  constexpr int MID_B = ((HP * 64 + 1023) / 1024) * 1024;            // 64-byte rows (32 mid channels), swizzle (R>>2)&3
  constexpr int X_ROWS = 384, X_B = X_ROWS * 128;                    // 128-byte rows (64 x channels), swizzle (R>>1)&7
  constexpr int W1_B = CM * 128;                                     // [32 mid couts][64 k]
  constexpr int W2_B = 9 * C * 64;                                   // 9 taps x [64 couts][32 k]
  constexpr int STG_B = NW * (TM / 2) * kEpiPitch2;
  constexpr int LDS_B = MID_B + X_B + W1_B + W2_B + STG_B;
  static_assert(LDS_B <= 160 * 1024, "LDS");
  __shared__ __attribute__((aligned(16))) char smem[LDS_B];
  char* const s_mid = smem;
  char* const s_x = smem + MID_B;
  char* const s_w1 = s_x + X_B;
  char* const s_w2 = s_w1 + W1_B;
  char* const s_stg = s_w2 + W2_B;

  const ConvArgs& a = ra.c;
  const YoloConvDesc& d = a.d;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r32 = lane & 31, khalf = lane >> 5;

  const int tiles_x = (d.w + 15) / 16, tiles_y = (d.h + 15) / 16;
  const long total = (long)d.n * tiles_y * tiles_x;
  const int lb = xcd_swizzle(blockIdx.x, gridDim.x);
  const int t_lo = (int)(lb * total / gridDim.x), t_hi = (int)((lb + 1) * total / gridDim.x);
  if (t_lo >= t_hi) return;

  const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)a.x, 0, a.x_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rw1 = __builtin_amdgcn_make_buffer_rsrc((void*)ra.w1, 0, ra.w1_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rw2 = __builtin_amdgcn_make_buffer_rsrc((void*)a.w, 0, a.w_bytes, 0x00020000);

  // ---- LDS-DMA lane roles.  128-byte rows: 8 rows per 1-KiB piece, lane -> (row lane/8, slot lane%8), source chunk =
  // slot ^ f(row) with f = (row >> 1) & 7.  x: 48 pieces = 6 per wave; W1: 4 pieces (waves 0..3).
  const int frow8 = lane >> 3;
  const int chunk8 = (lane & 7) ^ (((lane >> 4) + 4 * (wave & 1)) & 7);
  int x_rel[6];          // element offset of this lane's source chunk relative to the tile's halo origin, or -1
  int x_yx[6];
#pragma unroll
  for (int it = 0; it < 6; ++it) {
    const int hr = (it * NW + wave) * 8 + frow8;
    const int hy = hr / HW2, hx = hr - hy * HW2;
    x_rel[it] = hr < HP ? (hy * d.w + hx) * d.in_c_total + chunk8 * 8 : -1;
    x_yx[it] = (hy << 8) | hx;
  }
  auto issue_x = [&](int b, int ty, int tx) {
    const int y0 = ty * 16 - 1, x0 = tx * 16 - 1;                    // halo origin
    const long base = ((long)(b * d.h + y0) * d.w + x0) * d.in_c_total + d.in_c_offset;
    const bool interior = y0 >= 0 && y0 + HW2 <= d.h && x0 >= 0 && x0 + HW2 <= d.w;
#pragma unroll
    for (int it = 0; it < 6; ++it) {
      bool ok = x_rel[it] >= 0;
      if (!interior) ok = ok && (unsigned)(y0 + (x_yx[it] >> 8)) < (unsigned)d.h && (unsigned)(x0 + (x_yx[it] & 255)) < (unsigned)d.w;
      const uint32_t voff = ok ? (uint32_t)((base + x_rel[it]) * 2) : kOobOffset;
      lds_dma16(rx, s_x + (it * NW + wave) * 1024, voff);
    }
  };
  // one-time operands: W1 (32 rows x 128 B) and the nine W2 tap slices (64 rows x 64 B each: 16 rows per piece)
  if (wave < 4) lds_dma16(rw1, s_w1 + wave * 1024, (uint32_t)((((wave * 8 + frow8) * ra.kpad1) + chunk8 * 8) * 2));
  {
    const int frow16 = lane >> 2, chunk4 = (lane & 3) ^ ((lane >> 4) & 3);
    // 36 pieces (tap-major: 4 pieces per tap): wave w takes pieces w, w+8, ... (< 36)
#pragma unroll
    for (int it = 0; it < 5; ++it) {
      const int piece = it * NW + wave;
      if (piece < 36) {
        const int tap = piece >> 2, row = (piece & 3) * 16 + frow16;
        lds_dma16(rw2, s_w2 + piece * 1024, (uint32_t)(((row * d.kpad) + tap * CM + chunk4 * 8) * 2));
      }
    }
  }

  // ---- per-lane constants
  f32x16 bias1;                                          // mid channel (e&3) + 8*(e>>2) + 4*khalf
  f32x16 bias2[MI];
#pragma unroll
  for (int g4 = 0; g4 < 4; ++g4)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      bias1[g4 * 4 + e] = ra.b1[g4 * 8 + khalf * 4 + e];
#pragma unroll
      for (int i = 0; i < MI; ++i) bias2[i][g4 * 4 + e] = a.bias[i * 32 + g4 * 8 + khalf * 4 + e];
    }
  // phase A: halo pixel blocks {wave, wave + 8} of the 11 (x 32 rows); row offsets are tile-invariant
  const bool two = wave + 8 < 11;
  int a_rd[2], a_wr[2], a_pos[2];
#pragma unroll
  for (int p = 0; p < 2; ++p) {
    const int hr = (wave + 8 * p) * 32 + r32;
    const int hy = hr / HW2, hx = hr - hy * HW2;
    a_rd[p] = hr < X_ROWS ? hr * 128 : 0;
    a_wr[p] = hr < HP ? ((hr * 64 + khalf * 8) ^ (((hr >> 2) & 3) << 4)) : -1;
    a_pos[p] = (hy << 8) | hx;
  }
  const int a_sw[2] = {(((wave) * 32 + r32) >> 1) & 7, (((wave + 8) * 32 + r32) >> 1) & 7};
  // phase B: output pixel q = wave*32 + r32 -> halo row of tap (0,0)
  const int hrow0 = ((wave * 32 + r32) >> 4) * HW2 + (r32 & 15);
  // residual chunks (epilogue lane mapping: 8 rows x 8 chunks per pass, two halves x two passes)
  int res_off[4];
  {
    const int crow = lane >> 3, cchunk = lane & 7;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int q = wave * TM + (k >> 1) * 16 + (k & 1) * 8 + crow;
      const int hr = (q / TW + 1) * HW2 + (q % TW) + 1;
      res_off[k] = hr * 128 + ((cchunk ^ ((hr >> 1) & 7)) << 4);
    }
  }
  char* const stg = s_stg + wave * ((TM / 2) * kEpiPitch2);

  int tx = t_lo % tiles_x, ty = (t_lo / tiles_x) % tiles_y, b = t_lo / (tiles_x * tiles_y);
  int ntx = tx, nty = ty, nb = b;
  auto advance = [&](int& x_, int& y_, int& b_) {
    if (++x_ == tiles_x) {
      x_ = 0;
      if (++y_ == tiles_y) {
        y_ = 0;
        ++b_;
      }
    }
  };
  issue_x(b, ty, tx);
  bool prev_full = false;                 // first tile: wait for everything (W1 / W2 / halo)

  for (int t = t_lo; t < t_hi; ++t, advance(tx, ty, b)) {
    const int x0 = tx * 16, y0 = ty * 16;
    // the halo of tile t must have landed; the previous tile's stores, issued after it, may still be in flight:
    // a full tile issues exactly 4 (8 with the pre-add copy) store instructions per wave after its halo prefetch
    if (prev_full && a.aux) wait_vmcnt<8>();
    else if (prev_full) wait_vmcnt<4>();
    else wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();         // x halo of tile t (and, first time, W1 / W2) landed; mid / s_x free of readers

    // ================= phase A: mid = act(W1 . x_halo + b1), K = 64 in one LDS-resident stage =================
    f32x16 acc1[2] = {bias1, bias1};
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const int g = ks * 2 + khalf;
      const bf16x8 wf = *reinterpret_cast<const bf16x8*>(s_w1 + r32 * 128 + ((g ^ ((r32 >> 1) & 7)) << 4));
      const bf16x8 xf0 = *reinterpret_cast<const bf16x8*>(s_x + a_rd[0] + ((g ^ a_sw[0]) << 4));
      acc1[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf, xf0, acc1[0], 0, 0, 0);
      if (two) {
        const bf16x8 xf1 = *reinterpret_cast<const bf16x8*>(s_x + a_rd[1] + ((g ^ a_sw[1]) << 4));
        acc1[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf, xf1, acc1[1], 0, 0, 0);
      }
    }
    bf16x8 res[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) res[k] = *reinterpret_cast<const bf16x8*>(s_x + res_off[k]);
    wait_lds();
    __builtin_amdgcn_s_barrier();         // every wave has what it needs from s_x (its LDS reads have completed)
    if (t + 1 < t_hi) {                   // the next tile's halo streams in under phase B and the epilogue
      advance(ntx, nty, nb);
      issue_x(nb, nty, ntx);
    }
    const bool interior = y0 >= 1 && y0 + 17 <= d.h && x0 >= 1 && x0 + 17 <= d.w;
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      if ((p == 1 && !two) || a_wr[p] < 0) continue;
      bool inside = true;
      if (!interior) {
        const int yy = y0 - 1 + (a_pos[p] >> 8), xx = x0 - 1 + (a_pos[p] & 255);
        inside = (unsigned)yy < (unsigned)d.h && (unsigned)xx < (unsigned)d.w;
      }
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) {
        bf16x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = (bf16_t)apply_act(acc1[p][g4 * 4 + e], d.act);
        u32x2 bits = __builtin_bit_cast(u32x2, o);
        if (!inside) bits = u32x2{0u, 0u};
        *reinterpret_cast<u32x2*>(s_mid + (a_wr[p] ^ (g4 << 4))) = bits;
      }
    }
    wait_lds();
    __builtin_amdgcn_s_barrier();         // mid complete

    // ================= phase B: 3x3 over mid, W2 resident =================
    f32x16 acc[MI] = {bias2[0], bias2[1]};
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      const int dh = tap / 3, dw = tap - 3 * dh;
      const int R = hrow0 + dh * HW2 + dw;
      const char* const xrow = s_mid + R * 64;
      const int swx = (R >> 2) & 3;
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        const int g = ks * 2 + khalf;
        const bf16x8 xf = *reinterpret_cast<const bf16x8*>(xrow + ((g ^ swx) << 4));
#pragma unroll
        for (int i = 0; i < MI; ++i) {
          const int Rw = i * 32 + r32;
          const bf16x8 wf = *reinterpret_cast<const bf16x8*>(s_w2 + tap * 4096 + Rw * 64 + ((g ^ ((Rw >> 2) & 3)) << 4));
          acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf, xf, acc[i], 0, 0, 0);
        }
      }
    }

    // ================= epilogue: act, (+ pre-add copy), + x, 128-byte lines =================
    {
      const int crow = lane >> 3, cchunk = lane & 7;
      const bool full = y0 + 16 <= d.h && x0 + 16 <= d.w;
      prev_full = full;
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        if ((r32 >> 4) == h) {
#pragma unroll
          for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
              f32x4 v;
#pragma unroll
              for (int e = 0; e < 4; ++e) v[e] = apply_act(acc[i][g4 * 4 + e], d.act);
              *reinterpret_cast<f32x4*>(stg + (r32 & 15) * kEpiPitch2 + i * 128 + (g4 * 8 + khalf * 4) * 4) = v;
            }
        }
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int pass = 0; pass < 2; ++pass) {
          const int lrow = pass * 8 + crow;
          const int q = wave * TM + h * 16 + lrow;
          const int yy = y0 + (q >> 4), xx = x0 + (q & 15);
          if (full || (yy < d.h && xx < d.w)) {
            const long pix = (long)(b * d.h + yy) * d.w + xx;
            const f32x4 lo = *reinterpret_cast<const f32x4*>(stg + lrow * kEpiPitch2 + cchunk * 32);
            const f32x4 hi = *reinterpret_cast<const f32x4*>(stg + lrow * kEpiPitch2 + cchunk * 32 + 16);
            float v[8] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
            if (a.aux) {
              bf16x8 o;
#pragma unroll
              for (int e = 0; e < 8; ++e) o[e] = (bf16_t)v[e];
              *reinterpret_cast<bf16x8*>(a.aux + pix * d.aux_c_total + d.aux_c_offset + cchunk * 8) = o;
            }
            const bf16x8 rv = res[h * 2 + pass];
            bf16x8 o;
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = (bf16_t)(v[e] + (float)rv[e]);
            *reinterpret_cast<bf16x8*>(reinterpret_cast<bf16_t*>(a.y) + pix * d.out_c_total + d.out_c_offset + cchunk * 8) = o;
          }
        }
        __builtin_amdgcn_wave_barrier();
      }
    }
  }
}

template <int C>
int launch(const ResUnitArgs& ra, hipStream_t s) {
  const YoloConvDesc& d = ra.c.d;
  const long grid = (long)d.n * ((d.h + 15) / 16) * ((d.w + 15) / 16);
  if (grid > 0x7fffffffL) return yolo_set_error(YOLO_E_UNSUPPORTED, "resunit grid too large");
  hipLaunchKernelGGL((resunit_kernel<C>), dim3((unsigned)grid), dim3(512), 0, s, ra);
  return yolo_check_launch("yolo_resunit_fwd");
}

}  // namespace

int& yolo_conv::resunit_debug() {
  static int v = getenv("YOLO_RESUNIT_DEBUG") ? atoi(getenv("YOLO_RESUNIT_DEBUG")) : 0;
  return v;
}

extern "C" int yolo_resunit_supported(int c, int h, int w) {
  if (c != 64 && c != 128 && c != 256) return 0;
  const long tiles = (long)((h + 15) / 16) * ((w + 15) / 16);
  return (double)h * w >= 0.85 * 256.0 * tiles && (long)h * w >= 80 * 80;
}

extern "C" int yolo_resunit_form(int c, int n, int h, int w) {
  if (!yolo_resunit_supported(c, h, w)) return 0;
  if (resunit_t20_applies(c, n, h, w)) return 3;
  return c == 64 ? 2 : 1;
}

extern "C" int yolo_resunit_fwd(const void* x, const void* w1_packed, const float* b1, const void* w2_packed, const float* b2,
                                void* y, void* y_preadd, const YoloConvDesc* dp, int kpad1, int cout_pad1,
                                yolo_stream_t s) {
  YOLO_REQUIRE(x && w1_packed && b1 && w2_packed && b2 && y && dp, "resunit: null pointer");
  const YoloConvDesc& d = *dp;     // describes the 3x3: cin = C/2, cout = C; the input view describes x (C channels)
  const int C = d.cout;
  YOLO_REQUIRE(yolo_resunit_supported(C, d.h, d.w), "resunit: C=%d on %dx%d is not covered (use the two-kernel path)", C, d.h, d.w);
  YOLO_REQUIRE(d.cin * 2 == C && d.ksize == 3 && d.stride == 1 && d.pad == 1 && d.ho == d.h && d.wo == d.w && !d.upsample2x &&
                   d.out_dtype == YOLO_DT_BF16,
               "resunit: descriptor must be the unit's 3x3/s1/p1 conv with cin = cout/2");
  YOLO_REQUIRE(d.in_c_offset % 8 == 0 && d.in_c_total % 8 == 0 && d.in_c_offset + C <= d.in_c_total, "resunit: bad input view");
  YOLO_REQUIRE(d.out_c_offset % 8 == 0 && d.out_c_total % 8 == 0 && d.out_c_offset + C <= d.out_c_total, "resunit: bad output view");
  if (y_preadd) YOLO_REQUIRE(d.aux_c_total % 8 == 0 && d.aux_c_offset % 8 == 0, "resunit: bad aux view");
  YOLO_REQUIRE(d.kpad % 64 == 0 && d.kpad >= 9 * d.cin && d.cout_pad % 128 == 0 && d.cout_pad >= C, "resunit: bad W2 packing");
  YOLO_REQUIRE(kpad1 % 64 == 0 && kpad1 >= C && cout_pad1 % 128 == 0 && cout_pad1 >= d.cin, "resunit: bad W1 packing");
  YOLO_REQUIRE(x != y, "resunit: y must not alias x (blocks read their neighbours' x halo)");
  const size_t x_bytes = (size_t)d.n * d.h * d.w * d.in_c_total * 2;
  const size_t w_bytes = (size_t)d.cout_pad * d.kpad * 2, w1_bytes = (size_t)cout_pad1 * kpad1 * 2;
  YOLO_REQUIRE(x_bytes < kOobOffset && w_bytes < kOobOffset, "resunit: tensor larger than 3.75 GiB not supported");
  ResUnitArgs ra;
  ra.c.x = (const bf16_t*)x;
  ra.c.w = (const bf16_t*)w2_packed;
  ra.c.bias = b2;
  ra.c.res = (const bf16_t*)x;
  ra.c.y = y;
  ra.c.aux = (bf16_t*)y_preadd;
  ra.c.d = d;
  ra.c.d.res_c_total = d.in_c_total;
  ra.c.d.res_c_offset = d.in_c_offset;
  ra.c.M = d.n * d.h * d.w;
  ra.c.n_tiles = 1;
  ra.c.steps = 0;
  ra.c.x_bytes = (uint32_t)x_bytes;
  ra.c.w_bytes = (uint32_t)w_bytes;
  ra.c.debug = resunit_debug();      // YOLO_RESUNIT_DEBUG / yolo_set_tuning(3, .): timing ablations and A/B forms only
  ra.w1 = (const bf16_t*)w1_packed;
  ra.b1 = b1;
  ra.kpad1 = kpad1;
  ra.w1_bytes = (uint32_t)w1_bytes;
  if (!(ra.c.debug & 128)) {       // the 20-pixel-wide tile kernels (conv_resunit_t20.hip) where their tiles cover the map; bit 128: never, 64: always
    const int rc = launch_resunit_t20(ra.c, ra.w1, ra.b1, ra.kpad1, ra.w1_bytes, (ra.c.debug & 64) != 0, (hipStream_t)s);
    if (rc != 1) return rc;
  }
  if (C == 64 && !(ra.c.debug & 32)) {        // persistent, software-pipelined form (YOLO_RESUNIT_DEBUG bit 32: generic kernel)
    static const int n_cu = [] {
      int dev = 0, v = 0;
      if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0)
        v = 256;
      return v;
    }();
    const long tiles = (long)d.n * ((d.h + 15) / 16) * ((d.w + 15) / 16);
    const long grid = tiles < n_cu ? tiles : n_cu;
    hipLaunchKernelGGL(resunit64_kernel, dim3((unsigned)grid), dim3(512), 0, (hipStream_t)s, ra);
    return yolo_check_launch("yolo_resunit_fwd");
  }
  if (C == 64) return launch<64>(ra, (hipStream_t)s);
  if (C == 128) return launch<128>(ra, (hipStream_t)s);
  return launch<256>(ra, (hipStream_t)s);
}
