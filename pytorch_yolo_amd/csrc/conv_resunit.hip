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
  // dependent load -> add -> store rounds per wave were half of this kernel's time (ablation in DESIGN.md 3.1d)
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

template <int C>
int launch(const ResUnitArgs& ra, hipStream_t s) {
  const YoloConvDesc& d = ra.c.d;
  const long grid = (long)d.n * ((d.h + 15) / 16) * ((d.w + 15) / 16);
  if (grid > 0x7fffffffL) return yolo_set_error(YOLO_E_UNSUPPORTED, "resunit grid too large");
  hipLaunchKernelGGL((resunit_kernel<C>), dim3((unsigned)grid), dim3(512), 0, s, ra);
  return yolo_check_launch("yolo_resunit_fwd");
}

}  // namespace

extern "C" int yolo_resunit_supported(int c, int h, int w) {
  if (c != 64 && c != 128 && c != 256) return 0;
  const long tiles = (long)((h + 15) / 16) * ((w + 15) / 16);
  return (double)h * w >= 0.85 * 256.0 * tiles && (long)h * w >= 80 * 80;
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
  static const int dbg = getenv("YOLO_RESUNIT_DEBUG") ? atoi(getenv("YOLO_RESUNIT_DEBUG")) : 0;   // timing ablations only
  ra.c.debug = dbg;
  ra.w1 = (const bf16_t*)w1_packed;
  ra.b1 = b1;
  ra.kpad1 = kpad1;
  ra.w1_bytes = (uint32_t)w1_bytes;
  if (C == 64) return launch<64>(ra, (hipStream_t)s);
  if (C == 128) return launch<128>(ra, (hipStream_t)s);
  return launch<256>(ra, (hipStream_t)s);
}
