// The 64-channel residual unit of the Darknet stem stage (reference models/yolov3_spp.py:17-32: y = x + act(conv3x3(act(conv1x1(x)))),
// 64 -> 32 -> 64 channels on the 320x320 maps of a 640x640 input) on 20x20 output tiles, with the machinery of conv3x3_t20.hip:
// same contract and rounding points as yolo_resunit_fwd's other kernels (the 32-channel intermediate is rounded to bf16 once,
// the sum with x is formed in fp32 and rounded once).
//
//   workgroup   4 waves, one 20x20 tile, all 64 couts; two workgroups per CU (64 KB of LDS, <= 256 registers)
//   phase A     the 22x22 halo of x arrives as two 32-channel chunk images [484 pixel rows of 64 B] by LDS-DMA (16 per wave, all
//               in flight at once).  A wave multiplies exactly the rows it fetched itself - 8 pieces of 16 pixels - with W1
//               (32 couts x 64 k: four fragments, in registers): 32 MFMAs, no barrier in front.  bias + act, zero outside the image
//               (the 3x3 zero-pads the INTERMEDIATE), bf16, written IN PLACE over its own rows of chunk image 0: that image now
//               is the 3x3's halo in conv3x3_t20's layout (slot = 8-channel group ^ 2 * (halo row & 1)).
//   phase B     one barrier, then the nine taps with all of W2 (64 couts x 288 k) in registers: wave = (cout half, patch parity),
//               32 couts x 13 (12) of the 25 4x4-pixel patches, fragment addresses = per-lane base + compile-time constants.
//               234 (216) MFMAs per wave, no barrier, no global traffic.
//   epilogue    patch pairs (even patch from waves 0/1, odd from 2/3) staged as [32 pixels][64 couts] fp32 in the idle second
//               chunk image, one raw barrier per pair; every lane then owns 8 couts of one pixel: residual (x, L2-hot) added in
//               fp32, 16-byte buffer stores over whole 128-byte pixel rows, residual rows requested three pairs ahead.
// Per tile the CU moves 62 KB in + 51 KB out (+ 51 KB of residual from L2) for 1,024 MFMAs: 0.93 GB per 32 images of 320x320, a
// floor of ~0.17 ms at 5.5 TB/s.  Measured 0.318 ms (the persistent 16x16-tile kernel of conv_resunit.hip: 0.365 ms): a workgroup
// is a load -> 1x1 -> 3x3 -> store chain and only two fit a CU.  Round 3 timeline (tools/ru_timeline.py 64 320, medians per tile):
// halo DMA 3.0 us, 1x1 + intermediate 2.7, nine taps 2.5, epilogue 5.1 - 13.6 us for 2 us of MFMA work.  Shortening single links
// of that chain moved the kernel by < 1.5 % each (same-box A/B): an epilogue per wave without workgroup barriers, residual rows 6 or
// 9 patches ahead instead of 3, persistent workgroups that keep W1 / W2 in registers across tiles (needs an opaque per-trip lane
// id, or the hoisted address constants spill).  Tried and dropped: a persistent form that fetches the next
// tile's halo straight into MFMA operand registers during the epilogue (no LDS for x, 48 KB per workgroup) - 0.370 ms: in the
// operand layout the four lanes that share a pixel row are 16 lanes apart, so a load instruction makes 64 separate 16-byte
// requests where the LDS-DMA layout makes 16 of 64 bytes, and hoisted per-lane address constants pushed it into spills.
#include "conv_common.h"

using namespace yolo_conv;

namespace {

constexpr int kT = 20, kHW = 22, kHPix = kHW * kHW;     // tile edge, halo edge, halo pixels
constexpr int kBuf = 32 * 1024;                          // one chunk image: 32 pieces of 16 rows of 64 B (31 used)
constexpr int kPitch = 272;                              // fp32 staging row: 64 couts + 16 B
constexpr int kSlab = 32 * kPitch;
static_assert(2 * kSlab <= kBuf, "the epilogue slabs live in the second chunk image");

struct RU20Args {
  ConvArgs c;           // the 3x3: w = W2, bias = b2, res = x view, y, aux; d.cin = 32, d.cout = 64
  const bf16_t* w1;     // packed [cout_pad1][kpad1], K = 64
  const float* b1;
  int kpad1;
  uint32_t w1_bytes;
  int stagger;          // resunit_t20w_kernel: start-up delay of the first round's second workgroups, in ~2k-cycle sleeps
};

// Diagnostic build only (-DYOLO_STAMPS, tools/ru_timeline.py): wave 0 of every workgroup records when it reached each phase
// boundary (s_memrealtime, 100 MHz) - 8 words per workgroup in a buffer nothing else reads.
#ifdef YOLO_STAMPS
#define RU_STAMP(k)                                                                                                  \
  do {                                                                                                               \
    if (a.stamps && tid == 0) a.stamps[8 * (size_t)blockIdx.x + (k)] = __builtin_amdgcn_s_memrealtime();             \
  } while (0)
#else
#define RU_STAMP(k)
#endif

template <bool LEAKY>
__global__ __launch_bounds__(256, 2) void resunit64_t20_kernel(const RU20Args ra) {
  constexpr int RD = 3;                                  // residual rows in flight in the epilogue (patch pairs ahead)
  __shared__ __attribute__((aligned(16))) char smem[2 * kBuf];
  const ConvArgs& a = ra.c;
  const YoloConvDesc& d = a.d;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int tiles_x = (d.w + kT - 1) / kT, tiles_y = (d.h + kT - 1) / kT;
  int b, y0, x0;
  {
    int swz = xcd_swizzle(blockIdx.x, gridDim.x);
    x0 = (swz % tiles_x) * kT;
    swz /= tiles_x;
    y0 = (swz % tiles_y) * kT;
    b = swz / tiles_y;
  }
  const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)a.x, 0, a.x_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void*)a.w, 0, a.w_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rw1 = __builtin_amdgcn_make_buffer_rsrc((void*)ra.w1, 0, ra.w1_bytes, 0x00020000);

  RU_STAMP(0);
  // ---- the x halo: piece (it * 4 + wave) = LDS rows [16 piece, +16); lane -> (row lane >> 2, physical slot lane & 3)
#pragma unroll
  for (int it = 0; it < 8; ++it) {
    const int piece = it * 4 + wave;
    const int hp = piece * 16 + (lane >> 2);
    const int hy = hp / kHW, hx = hp - hy * kHW;
    const int yy = y0 - 1 + hy, xx = x0 - 1 + hx;
    const int chunk = (lane & 3) ^ ((hy & 1) << 1);
    const bool ok = hp < kHPix && (unsigned)yy < (unsigned)d.h && (unsigned)xx < (unsigned)d.w;
    const uint32_t ho = ok ? (uint32_t)((((b * d.h + yy) * d.w + xx) * d.in_c_total + d.in_c_offset + chunk * 8) * 2) : kOobOffset;
    lds_dma16s(rx, smem + piece * 1024, ho, 0u);             // channels 0..31
    lds_dma16s(rx, smem + kBuf + piece * 1024, ho, 64u);     // channels 32..63
  }

  const int c16 = lane & 15, q = lane >> 4;
  auto mfma = [&](f32x4& t, const bf16x8& wa, const bf16x8& xb) {
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(t) : "v"(wa), "v"(xb));
#endif
  };
  const bool floor0 = d.act == YOLO_ACT_RELU || d.act == YOLO_ACT_RELU6;
  const float slope = d.act == YOLO_ACT_LEAKY01 ? 0.1f : 1.f;
  const float hi_clamp = act_hi(d.act);
  auto act4 = [&](f32x4 v) -> f32x4 {
    if (LEAKY) return leaky4(v);                           // (the launcher's choice for LeakyReLU(0.1): conv_common.h)
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = fminf(fmaxf(v[e], floor0 ? 0.f : slope * v[e]), hi_clamp);
    return v;
  };

  // ---- phase A: intermediate[32 couts][my 8 x 16 halo pixels] = W1 x (both chunk images)
  bf16x8 w1f[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int kc = 0; kc < 2; ++kc)
      w1f[i][kc] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(
                                                  rw1, (uint32_t)(((i * 16 + c16) * ra.kpad1 + kc * 32 + q * 8) * 2), 0, 0));
  f32x4 b1v[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) b1v[i] = *reinterpret_cast<const f32x4*>(ra.b1 + i * 16 + q * 4);
  uint32_t rowA[8];                                        // this lane's 16 bytes of piece `it`: row (piece * 16 + c16), k group q
#pragma unroll
  for (int it = 0; it < 8; ++it) {
    const int row = (it * 4 + wave) * 16 + c16;
    const int hy = row / kHW;
    rowA[it] = (uint32_t)(row * 64 + ((q ^ ((hy & 1) << 1)) << 4));
  }
  f32x4 acc1[8][2];
#pragma unroll
  for (int it = 0; it < 8; ++it)
#pragma unroll
    for (int i = 0; i < 2; ++i) acc1[it][i] = f32x4{0.f, 0.f, 0.f, 0.f};
  wait_vmcnt<0>();                                         // my own DMA pieces (and W1) have landed: nobody else's rows are read here
  RU_STAMP(1);
  static_for<2>([&](auto kcc) {
    constexpr int kc = decltype(kcc)::value;
    static_for<8>([&](auto itc) {
      constexpr int it = decltype(itc)::value;
      const bf16x8 xb = *reinterpret_cast<const bf16x8*>(smem + kc * kBuf + rowA[it]);
      static_for<2>([&](auto ic) { mfma(acc1[it][decltype(ic)::value], w1f[decltype(ic)::value][kc], xb); });
    });
  });

  // all of W2 for this wave: 32 couts (cout half cg) x 9 taps x 32 k; they land under phase A's epilogue
  const int cg = wave & 1, pg = wave >> 1;
  const uint32_t wv = (uint32_t)(((cg * 32 + c16) * d.kpad + q * 8) * 2);
  const uint32_t wfrag = (uint32_t)(16 * d.kpad * 2);
  bf16x8 wf[9][2];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int i = 0; i < 2; ++i)
      wf[t][i] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rw, wv + (uint32_t)(t * 64) + i * wfrag, 0, 0));

#if defined(__HIP_DEVICE_COMPILE__)
  asm volatile("s_nop 15\n\ts_nop 15");                  // the asm MFMAs' D registers: wait states before any other reader
#endif
#pragma unroll
  for (int it = 0; it < 8; ++it) {
    const int row = (it * 4 + wave) * 16 + c16;
    const int hy = row / kHW, hx = row - hy * kHW;
    const int yy = y0 - 1 + hy, xx = x0 - 1 + hx;
    const bool inimg = row < kHPix && (unsigned)yy < (unsigned)d.h && (unsigned)xx < (unsigned)d.w;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      f32x4 v = act4(acc1[it][i] + b1v[i]);
      if (!inimg) v = f32x4{0.f, 0.f, 0.f, 0.f};
      bf16x4 o;
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = (bf16_t)v[e];
      // channels i * 16 + q * 4 + e: 8-channel group i * 2 + (q >> 1), second half of the slot for odd q
      *reinterpret_cast<bf16x4*>(smem + row * 64 + (((i * 2 + (q >> 1)) ^ ((hy & 1) << 1)) << 4) + (q & 1) * 8) = o;
    }
  }
  wait_lds();
  __builtin_amdgcn_s_barrier();                            // chunk image 0 now holds the whole intermediate halo
  RU_STAMP(2);

  // ---- phase B: the nine taps
  const int dy = c16 >> 2, dx = c16 & 3;
  uint32_t A[2];                                           // halo bases by parity of (dy + tap row)
#pragma unroll
  for (int par = 0; par < 2; ++par) A[par] = (uint32_t)((dy * kHW + dx) * 64 + ((q ^ (((dy + par) & 1) << 1)) << 4));
#if defined(__HIP_DEVICE_COMPILE__)
  asm volatile("" : "+v"(A[0]), "+v"(A[1]));
#endif
  auto xread = [&](int jj, int tap, uint32_t am) -> bf16x8 {
    const char* const p = smem + am;
    return *reinterpret_cast<const bf16x8*>(p + ((4 * (jj / 5)) * kHW + 4 * (jj % 5)) * 64 + ((tap / 3) * kHW + tap % 3) * 64);
  };

  const uint32_t y_pitch = (uint32_t)d.out_c_total * 2u, r_pitch = (uint32_t)d.res_c_total * 2u, x_pitch = (uint32_t)d.aux_c_total * 2u;
  const uint32_t npix = (uint32_t)d.n * d.h * d.w;
  const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc(a.y, 0, npix * y_pitch, 0x00020000);
  const __amdgpu_buffer_rsrc_t rr = __builtin_amdgcn_make_buffer_rsrc((void*)a.res, 0, npix * r_pitch, 0x00020000);
  const __amdgpu_buffer_rsrc_t rax = __builtin_amdgcn_make_buffer_rsrc((void*)a.aux, 0, a.aux ? npix * x_pitch : 0u, 0x00020000);

  auto run = [&](auto pgc) {
    constexpr int PG = decltype(pgc)::value;               // patches 2 k + PG
    constexpr int NPW = PG ? 12 : 13, NSTEP = 9 * NPW, XD = 3;
    f32x4 acc[2][NPW];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int k = 0; k < NPW; ++k) acc[i][k] = f32x4{0.f, 0.f, 0.f, 0.f};
    // ---- epilogue addressing, and the first residual rows: requested before the nine taps, they land under them
    f32x4 b2v[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) b2v[i] = *reinterpret_cast<const f32x4*>(a.bias + cg * 32 + i * 16 + q * 4);
    // coalesced phase: this wave stores half of ITS OWN patch of the pair: pixel rows 8 (wave & 1) + (lane >> 3), 8 couts per lane
    const int prow = 8 * (wave & 1) + (lane >> 3), cch = lane & 7;
    const int dyr = prow >> 2, dxr = prow & 3;
    const uint32_t lpix = (uint32_t)((b * d.h + y0 + dyr) * d.w + x0 + dxr);
    const uint32_t ccol = (uint32_t)(cch * 8) * 2u;
    const uint32_t yo = lpix * y_pitch + (uint32_t)d.out_c_offset * 2u + ccol;
    const uint32_t ro = lpix * r_pitch + (uint32_t)d.res_c_offset * 2u + ccol;
    const uint32_t ao = lpix * x_pitch + (uint32_t)d.aux_c_offset * 2u + ccol;
    const int ylim = d.h - y0 - dyr, xlim = d.w - x0 - dxr;
    auto voff = [&](uint32_t base, uint32_t pitch, int jj) -> uint32_t {
      const int pr = jj / 5, pc = jj % 5;
      const bool ok = 4 * pr < ylim && 4 * pc < xlim;
      return ok ? base + (uint32_t)(4 * pr * d.w + 4 * pc) * pitch : kOobOffset;
    };
    u32x4 rv[RD + 1];
    auto fetch_res = [&](auto rc) {
      constexpr int r = decltype(rc)::value;
      rv[r % (RD + 1)] = __builtin_amdgcn_raw_buffer_load_b128(rr, voff(ro, r_pitch, 2 * r + PG), 0, 0);
    };
    static_for<RD>([&](auto rc) { fetch_res(rc); });
    bf16x8 xf[XD];
#pragma unroll
    for (int s = 0; s < XD - 1; ++s) xf[s] = xread(2 * (s % NPW) + PG, s / NPW, A[((s / NPW) / 3) & 1]);
    static_for<NSTEP>([&](auto sc) {
      constexpr int s = decltype(sc)::value, tap = s / NPW, k = s % NPW;
      if constexpr (s + XD - 1 < NSTEP) {
        constexpr int s2 = s + XD - 1, tap2 = s2 / NPW, k2 = s2 % NPW;
        xf[s2 % XD] = xread(2 * k2 + PG, tap2, A[(tap2 / 3) & 1]);
      }
      static_for<2>([&](auto ic) { mfma(acc[decltype(ic)::value][k], wf[tap][decltype(ic)::value], xf[s % XD]); });
    });
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("s_nop 15\n\ts_nop 15");
#endif
    RU_STAMP(3);
    if (a.debug & 8) return;

    // ---- epilogue: pair r = patches 2 r (waves 0, 1) and 2 r + 1 (waves 2, 3)
    static_for<13>([&](auto rc) {
      constexpr int r = decltype(rc)::value;
      char* const slab = smem + kBuf + (r & 1) * kSlab;
      if constexpr (r < NPW) {
        static_for<2>([&](auto ic) {
          constexpr int i = decltype(ic)::value;
          *reinterpret_cast<f32x4*>(slab + (PG * 16 + c16) * kPitch + (cg * 32 + i * 16 + q * 4) * 4) = act4(acc[i][r] + b2v[i]);
        });
      }
      if constexpr (r + RD < NPW) fetch_res(std::integral_constant<int, r + RD>{});
      wait_lds();
      __builtin_amdgcn_s_barrier();                        // the pair is staged by all four waves (and pair r - 1 has been read by all)
      if constexpr (r < NPW) {
        const int row = PG * 16 + prow;
        const f32x4 lo = *reinterpret_cast<const f32x4*>(slab + row * kPitch + cch * 32);
        const f32x4 hi = *reinterpret_cast<const f32x4*>(slab + row * kPitch + cch * 32 + 16);
        float v[8] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        if (a.aux) {
          bf16x8 o;
#pragma unroll
          for (int e = 0; e < 8; ++e) o[e] = (bf16_t)v[e];
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o), rax, voff(ao, x_pitch, 2 * r + PG), 0, 0);
        }
        const bf16x8 r8 = __builtin_bit_cast(bf16x8, rv[r % (RD + 1)]);
        bf16x8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = (bf16_t)(v[e] + (float)r8[e]);
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o), ry, voff(yo, y_pitch, 2 * r + PG), 0, 0);
      }
    });
  };
  if (pg == 0) run(std::integral_constant<int, 0>{});
  else run(std::integral_constant<int, 1>{});
  RU_STAMP(4);
}


// ---------------------------------------------------------------------------------------------------------------------
// The wider units (round 3): C = 128 on 160x160 maps and C = 256 on 80x80 maps of a 640x640 input (reference
// models/yolov3_spp.py:17-46).  Same idea as the 64-channel kernel - the C/2-channel intermediate t lives in LDS only, the nine
// taps run against it with conv3x3_t20v2's register-resident weight stream - generalised so that the WHOLE intermediate halo of a
// tile fits 64 KB and TWO workgroups share a CU (one's x fetch and store phases run under the other's MFMA stream):
//
//   <CMID = 64, TPH = 5>    tile 20 x 20 (25 patches of 4 x 4), halo 22 x 22, t = 2 chunk images of 32 KB,
//                           wave = 32 couts x 25 patches (200 accumulator registers) - conv3x3_t20v2's shape
//   <CMID = 128, TPH = 2>   tile 20 x 8 (10 patches), halo 22 x 10, t = 4 chunk images of 16 KB,
//                           wave = 64 couts x 10 patches (160 accumulator registers): all 256 couts of a pixel in one workgroup,
//                           so the intermediate is computed once per tile (a 20 x 20 tile would need 124 KB for t: one
//                           workgroup per CU, every phase of it exposed)
//
//   phase A   x arrives as C / 32 chunk images [halo pixel rows of 64 B] by LDS-DMA through the CMID / 32 image buffers that
//             later hold t.  A wave multiplies exactly the pixel rows it fetched itself (PPW pieces of 16) with W1 - no barrier,
//             only its own counted vmcnt: chunk c is multiplied while chunks c + 1 .. are in flight, and buffer c % NB is
//             refilled with chunk c + NB as soon as the wave's own fragment reads of it have returned.  W1's fragments of a chunk
//             (CMID / 16 of them) come straight from L2 into registers, one chunk ahead.  bias + act, zero outside the image
//             (the 3x3 zero-pads t, not x), bf16, written over the wave's own rows: the buffers now hold t in conv3x3_t20's
//             layout (slot = 8-channel group ^ 2 * (halo row & 1)).
//   phase B   one barrier, then (CMID / 32) x 9 steps of NFW x NP MFMAs per wave against the resident halo: W2 fragments by
//             buffer_load two steps ahead in three register sets, pixel fragments by ds_read with compile-time offsets; no
//             barrier, no LDS-DMA.
//   epilogue  patch pairs (CT 128) / single patches (CT 256) staged as fp32 rows of all CT couts, x (L2 / Infinity-Cache hot)
//             added in fp32, whole 256- / 512-byte pixel rows stored with 16-byte buffer stores, residual rows RD units ahead.
// Rounding points = the two-launch path's: t rounded to bf16 once, conv + residual summed in fp32 and rounded once.
template <int CMID, int TPH>
struct RUW {
  static constexpr int C = 2 * CMID;
  static constexpr int HY = 4 * TPH + 2, HP = HY * kHW;            // halo rows, halo pixels
  static constexpr int PIECES = (HP + 15) / 16, PPW = (PIECES + 3) / 4;
  static constexpr int IMG_B = PPW * 4 * 1024;                      // one chunk image (incl. padding pieces)
  static constexpr int NIMG = CMID / 32, NXC = C / 32, NF = CMID / 16;
  static constexpr int NFW = C / 64, NP = 5 * TPH;                  // cout fragments per wave, patches
  static constexpr int CT = C, U = CT == 128 ? 2 : 1;               // patches per staging unit
  static constexpr int PITCH = CT * 4 + 16, SLAB = 16 * U * PITCH;
  static constexpr int LDS_B = NIMG * IMG_B;
  static_assert(2 * SLAB <= LDS_B, "the epilogue slabs live in the image buffers");
  static_assert(LDS_B <= 80 * 1024, "two workgroups per CU");
};

template <int CMID, int TPH, bool LEAKY>
__global__ __launch_bounds__(256, 2) void resunit_t20w_kernel(const RU20Args ra) {
  using G = RUW<CMID, TPH>;
  constexpr int PPW = G::PPW, IMG_B = G::IMG_B, NIMG = G::NIMG, NXC = G::NXC, NFW = G::NFW, NP = G::NP;
  constexpr int NB = NIMG;                               // x staging buffers = the t images
  constexpr int RD = CMID == 64 ? 2 : 3, XD = 3;         // (25 patches x 2 fragments: 200 accumulator registers leave room for two units of residual rows in flight)
  __shared__ __attribute__((aligned(16))) char smem[G::LDS_B];
  const ConvArgs& a = ra.c;
  const YoloConvDesc& d = a.d;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int tiles_x = (d.w + kT - 1) / kT, tiles_y = (d.h + 4 * TPH - 1) / (4 * TPH);
  int b, y0, x0;
  {
    int swz = xcd_swizzle(blockIdx.x, gridDim.x);
    x0 = (swz % tiles_x) * kT;
    swz /= tiles_x;
    y0 = (swz % tiles_y) * (4 * TPH);
    b = swz / tiles_y;
  }
  const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)a.x, 0, a.x_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void*)a.w, 0, a.w_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rw1 = __builtin_amdgcn_make_buffer_rsrc((void*)ra.w1, 0, ra.w1_bytes, 0x00020000);
  const int c16 = lane & 15, q = lane >> 4;

  // ---- phase A addressing.  DMA: piece (it * 4 + wave) = LDS rows [16 piece, +16); lane -> (row lane >> 2, physical slot lane & 3)
  uint32_t h_off[PPW];
#pragma unroll
  for (int it = 0; it < PPW; ++it) {
    const int hp = (it * 4 + wave) * 16 + (lane >> 2);
    const int hy = hp / kHW, hx = hp - hy * kHW;
    const int yy = y0 - 1 + hy, xx = x0 - 1 + hx;
    const int chunk = (lane & 3) ^ ((hy & 1) << 1);
    const bool ok = hp < G::HP && (unsigned)yy < (unsigned)d.h && (unsigned)xx < (unsigned)d.w;
    h_off[it] = ok ? (uint32_t)((((b * d.h + yy) * d.w + xx) * d.in_c_total + d.in_c_offset + chunk * 8) * 2) : kOobOffset;
  }
  auto issue_x = [&](auto cc) {                            // chunk c of x -> buffer c % NB, this wave's pieces
    constexpr int c = decltype(cc)::value;
#pragma unroll
    for (int it = 0; it < PPW; ++it) lds_dma16s(rx, smem + (c % NB) * IMG_B + (it * 4 + wave) * 1024, h_off[it], (uint32_t)(c * 64));
  };
  auto mfma = [&](f32x4& t, const bf16x8& wa, const bf16x8& xb) {
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(t) : "v"(wa), "v"(xb));
#endif
  };
  const bool floor0 = d.act == YOLO_ACT_RELU || d.act == YOLO_ACT_RELU6;
  const float slope = d.act == YOLO_ACT_LEAKY01 ? 0.1f : 1.f;
  const float hi_clamp = act_hi(d.act);
  auto act4 = [&](f32x4 v) -> f32x4 {
    if (LEAKY) return leaky4(v);                           // (the launcher's choice for LeakyReLU(0.1): conv_common.h)
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = fminf(fmaxf(v[e], floor0 ? 0.f : slope * v[e]), hi_clamp);
    return v;
  };

  // Start-up stagger: the second workgroup of every CU in the first round of the grid (blocks 256 .. 511 under the round-robin
  // dispatch; a guess that only ever costs time) starts late, so that from then on one workgroup of a CU fetches / stores while
  // the other multiplies.  Without it the two run in lockstep - both wait for x, both multiply, both store - and a launch takes
  // the SUM of its memory and matrix phases (measured: 0.114 ms without the epilogue, 0.179 ms with it on the 160x160 unit).
  if (ra.stagger > 0 && blockIdx.x >= 256 && blockIdx.x < 512) {
    for (int i = 0; i < ra.stagger; ++i) __builtin_amdgcn_s_sleep(32);        // ~2k cycles each
  }

  // ---- phase A: W1-stationary, the couts of t split over the waves.  Wave w owns t channels [CMID / 4 * w, + CMID / 4): its NFA
  // cout fragments of W1 for the WHOLE K sit in registers (loaded once, before any x is requested: vector-memory operations retire
  // in issue order, and a W1 load issued behind an x DMA that is still on its way from HBM waits for it - the first version, which
  // streamed W1 beside x chunk by chunk, paid an HBM round trip per step, and an x-stationary version that streamed W1 from L2
  // one fragment ahead spent 10 us per tile on L2 round trips for 2 us of MFMAs).  x streams through the image buffers as a ring
  // of 32-channel chunks (every wave fetches its pieces of a chunk; one barrier per chunk says "chunk c has landed for everybody
  // and everybody is done with chunk c - 1", behind which the freed buffer is refilled), and each wave multiplies ALL pixel rows
  // of the chunk: PIECES fragment reads and PIECES * NFA MFMAs per chunk.
  constexpr int NFA = CMID / 64, PIECES = G::PIECES;
  RU_STAMP(0);
  bf16x8 w1f[NFA][NXC];
#pragma unroll
  for (int fa = 0; fa < NFA; ++fa)
#pragma unroll
    for (int kc = 0; kc < NXC; ++kc)
      w1f[fa][kc] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(
                                                   rw1, (uint32_t)((((wave * NFA + fa) * 16 + c16) * ra.kpad1 + q * 8) * 2), (uint32_t)(kc * 64), 0));
  f32x4 b1v[NFA];
#pragma unroll
  for (int fa = 0; fa < NFA; ++fa) b1v[fa] = *reinterpret_cast<const f32x4*>(ra.b1 + (wave * NFA + fa) * 16 + q * 4);
#if defined(__HIP_DEVICE_COMPILE__)
  asm volatile("" ::: "memory");                           // W1 / bias are requested BEFORE the x DMAs (and the counted waits below rely on
#endif                                                     // the order: hipcc moves plain loads across the DMA builtins otherwise)
  static_for<NB>([&](auto kc) { issue_x(kc); });
#if defined(__HIP_DEVICE_COMPILE__)
  asm volatile("" ::: "memory");
#endif
  // Fragment address of piece p in a buffer: pixel row 16 p + c16, k group q, slot = q ^ 2 * (halo row & 1) - computed HERE, once,
  // and kept in registers: inside the chunk loop nothing but ds_read, MFMA and waits may run.  hipcc does not see through the asm
  // MFMAs, and a VALU result placed in a register that the MFMA issued just before still reads as its B operand (the fragment of
  // the piece before, dead by then for the allocator) corrupts that MFMA - seen as run-to-run differing outputs with the address
  // arithmetic inside the loop.  For the same reason a fragment register stays live (empty asm use) past the next piece's MFMAs.
  uint32_t xaddr[PIECES];
#pragma unroll
  for (int pcs = 0; pcs < PIECES; ++pcs) {
    const int hy = (pcs * 16 + c16) / kHW;
    xaddr[pcs] = (uint32_t)(pcs * 1024 + c16 * 64 + ((q ^ ((hy & 1) << 1)) << 4));
  }
#if defined(__HIP_DEVICE_COMPILE__)
  static_for<PIECES>([&](auto pc) {
    uint32_t& t = xaddr[decltype(pc)::value];
    asm volatile("" : "+v"(t));                            // opaque: never rematerialised inside the loop
  });
#endif
  f32x4 acc1[PIECES][NFA];
  static_for<NXC>([&](auto cc) {
    constexpr int c = decltype(cc)::value;
    // DMAs of this wave younger than chunk c's: chunks c + 1 .. min(c - 2 + NB, NXC - 1) (step k >= 1 issues chunk k - 1 + NB)
    constexpr int last = c == 0 ? NB - 1 : (c - 2 + NB < NXC - 1 ? c - 2 + NB : NXC - 1);
    constexpr int younger = (last > c ? last - c : 0) * PPW;
    wait_vmcnt<younger>();
    __builtin_amdgcn_s_barrier();
    if constexpr (c == 1) RU_STAMP(1);
    if constexpr (c >= 1 && c - 1 + NB < NXC) {
      issue_x(std::integral_constant<int, c - 1 + NB>{});
#if defined(__HIP_DEVICE_COMPILE__)
      asm volatile("" ::: "memory");
#endif
    }
    bf16x8 xb[3];                                          // fragments in flight: piece p + 2 is requested before piece p is multiplied
    auto xld = [&](auto pc) {
      constexpr int pcs = decltype(pc)::value;
      const char* const ptr = smem + xaddr[pcs];
      xb[pcs % 3] = *reinterpret_cast<const bf16x8*>(ptr + (c % NB) * IMG_B);
    };
    xld(std::integral_constant<int, 0>{});
    if constexpr (PIECES > 1) xld(std::integral_constant<int, 1>{});
    static_for<PIECES>([&](auto pc) {
      constexpr int pcs = decltype(pc)::value;
      if constexpr (pcs + 2 < PIECES) xld(std::integral_constant<int, pcs + 2>{});
      static_for<NFA>([&](auto fc) {
        constexpr int fa = decltype(fc)::value;
        f32x4& t = acc1[pcs][fa];
        const bf16x8 &wa = w1f[fa][c], &xbr = xb[pcs % 3];      // (bound here: asm operands alone do not capture in a lambda)
        (void)t, (void)wa, (void)xbr;
#if defined(__HIP_DEVICE_COMPILE__)
        // (c == 0: the accumulator starts as this MFMA's result, C operand = the inline constant 0 - never a v_mov next to the asm MFMAs)
        if constexpr (c == 0) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, 0" : "=&v"(t) : "v"(wa), "v"(xbr));
        else asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(t) : "v"(wa), "v"(xbr));
#endif
      });
#if defined(__HIP_DEVICE_COMPILE__)
      if constexpr (pcs >= 1) {
        const bf16x8& prev = xb[(pcs - 1) % 3];
        asm volatile("" ::"v"(prev));
      }
#endif
    });
#if defined(__HIP_DEVICE_COMPILE__)
    {
      const bf16x8& lastf = xb[(PIECES - 1) % 3];
      asm volatile("s_nop 7" ::"v"(lastf));                // the chunk's last MFMAs have read their operands before anything else runs
    }
#endif
  });
  RU_STAMP(2);
#if defined(__HIP_DEVICE_COMPILE__)
  asm volatile("s_nop 15\n\ts_nop 15");                  // the asm MFMAs' D registers: wait states before any other reader
  static_for<NFA>([&](auto fc) {                           // (W1 stays live until here: nothing may reuse its registers next to the MFMAs)
    static_for<NXC>([&](auto kc) {
      const bf16x8& wa = w1f[decltype(fc)::value][decltype(kc)::value];
      asm volatile("" ::"v"(wa));
    });
  });
#endif
  wait_lds();
  __builtin_amdgcn_s_barrier();                            // everybody is done with the last x chunks: the buffers take t now
  static_for<PIECES>([&](auto pc) {
    constexpr int pcs = decltype(pc)::value;
    const int row = pcs * 16 + c16;
    const int hy = row / kHW, hx = row - hy * kHW;
    const int yy = y0 - 1 + hy, xx = x0 - 1 + hx;
    const bool inimg = row < G::HP && (unsigned)yy < (unsigned)d.h && (unsigned)xx < (unsigned)d.w;
    static_for<NFA>([&](auto fc) {
      constexpr int fa = decltype(fc)::value;
      f32x4 v = act4(acc1[pcs][fa] + b1v[fa]);
      if (!inimg) v = f32x4{0.f, 0.f, 0.f, 0.f};           // the 3x3 zero-pads t, not x
      bf16x4 o;
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = (bf16_t)v[e];
      // t channels f * 16 + q * 4 + e with f = wave * NFA + fa: image f / 2, 8-channel group (f & 1) * 2 + (q >> 1), second
      // half of the slot for odd q
      const int f = wave * NFA + fa;
      *reinterpret_cast<bf16x4*>(smem + (f >> 1) * IMG_B + row * 64 + ((((f & 1) * 2 + (q >> 1)) ^ ((hy & 1) << 1)) << 4) + (q & 1) * 8) = o;
    });
  });

  // W2 fragments of the first two steps: they land under phase A's epilogue
  const uint32_t wv = (uint32_t)(((wave * (NFW * 16) + c16) * d.kpad + q * 8) * 2);
  const uint32_t wfrag = (uint32_t)(16 * d.kpad * 2);
  auto wload = [&](int c, int tap, int i) -> bf16x8 {
    return __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rw, wv + i * wfrag, (uint32_t)((tap * CMID + c * 32) * 2), 0));
  };
  bf16x8 wf[3][NFW];
#pragma unroll
  for (int i = 0; i < NFW; ++i) {
    wf[0][i] = wload(0, 0, i);
    wf[1][i] = wload(0, 1, i);
  }

  wait_lds();
  __builtin_amdgcn_s_barrier();                            // the image buffers now hold the whole intermediate halo
  RU_STAMP(3);

  if (a.debug & 1024) {                                    // diagnosis: block 0 dumps its t images into y, nobody computes
    if (blockIdx.x == 0)
      for (int i = tid; i < G::LDS_B / 16; i += 256) reinterpret_cast<u32x4*>(a.y)[i] = reinterpret_cast<const u32x4*>(smem)[i];
    return;
  }

  // ---- phase B
  const int dy = c16 >> 2, dx = c16 & 3;
  uint32_t A[2];                                           // halo bases by parity of (dy + tap row)
#pragma unroll
  for (int par = 0; par < 2; ++par) A[par] = (uint32_t)((dy * kHW + dx) * 64 + ((q ^ (((dy + par) & 1) << 1)) << 4));
#if defined(__HIP_DEVICE_COMPILE__)
  asm volatile("" : "+v"(A[0]), "+v"(A[1]));
#endif
  auto xread = [&](int jj, int tap, uint32_t am) -> bf16x8 {
    const char* const p = smem + am;
    return *reinterpret_cast<const bf16x8*>(p + ((4 * (jj / 5)) * kHW + 4 * (jj % 5)) * 64 + ((tap / 3) * kHW + tap % 3) * 64);
  };
  f32x4 acc[NFW][NP];
#pragma unroll
  for (int i = 0; i < NFW; ++i)
#pragma unroll
    for (int j = 0; j < NP; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  bf16x8 xf[XD];
  for (int c = 0; c < NIMG; ++c) {
    static_for<9>([&](auto tc) {
      constexpr int tap = decltype(tc)::value;
      constexpr int par = (tap / 3) & 1, par_n = (((tap + 1) % 9) / 3) & 1;
      {
        constexpr int t2 = (tap + 2) % 9;
        const int c2 = c + (tap + 2) / 9;                  // (beyond the last chunk: in-bounds rows of W2's k padding or zeros, never used)
#pragma unroll
        for (int i = 0; i < NFW; ++i) wf[(tap + 2) % 3][i] = wload(c2, t2, i);
      }
      const uint32_t am = A[par];
      const uint32_t am_n = tap == 8 ? A[par_n] + (uint32_t)IMG_B : A[par_n];
      if constexpr (tap == 0) {
        if (c == 0) {
#pragma unroll
          for (int j = 0; j < XD - 1; ++j) xf[j] = xread(j, 0, am);
        }
      }
      static_for<NP>([&](auto jc) {
        constexpr int jj = decltype(jc)::value;
        // rotation: patch jj of tap t lives in xf[(jj + t * NP) % XD]; NP % XD == 1 (25, 10), so the slot of the next tap's patch j is
        // the slot "patch NP + j of this tap" would take
        constexpr int R = (tap * NP) % XD;
        static_assert(NP % XD == 1, "rotation");
        if constexpr (jj + XD - 1 < NP) xf[(jj + XD - 1 + R) % XD] = xread(jj + XD - 1, tap, am);
        else if constexpr (tap < 8) xf[(jj + XD - 1 + R) % XD] = xread(jj + XD - 1 - NP, tap + 1, am_n);   // the next step's first ones
        else if (c + 1 < NIMG) xf[(jj + XD - 1 + R) % XD] = xread(jj + XD - 1 - NP, 0, am_n);               // ... in the next image
        static_for<NFW>([&](auto ic) { mfma(acc[decltype(ic)::value][jj], wf[tap % 3][decltype(ic)::value], xf[(jj + R) % XD]); });
      });
    });
    A[0] += (uint32_t)IMG_B;
    A[1] += (uint32_t)IMG_B;
  }
#if defined(__HIP_DEVICE_COMPILE__)
  asm volatile("s_nop 15\n\ts_nop 15");
#endif
  RU_STAMP(4);
#ifdef YOLO_STAMPS
  if (a.stamps && tid == 0) a.stamps[8 * (size_t)blockIdx.x + 6] = __builtin_amdgcn_s_getreg((31 << 11) | 4) | ((unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 20) << 32);
#endif
  if (a.debug & 8) return;
  __syncthreads();                                         // every wave is done with t: the staging slabs reuse the image buffers

  // ---- epilogue
  constexpr int CT = G::CT, U = G::U, PITCH = G::PITCH, SLAB = G::SLAB;
  constexpr int LPR = CT / 8, RPW = 64 / LPR;              // lanes per pixel row (8 couts each), rows per wave and pass
  constexpr int NUNIT = (NP + U - 1) / U;
  f32x4 b2v[NFW];
#pragma unroll
  for (int i = 0; i < NFW; ++i) b2v[i] = *reinterpret_cast<const f32x4*>(a.bias + wave * (NFW * 16) + i * 16 + q * 4);
  const uint32_t y_pitch = (uint32_t)d.out_c_total * 2u, r_pitch = (uint32_t)d.res_c_total * 2u, x_pitch = (uint32_t)d.aux_c_total * 2u;
  const uint32_t npix = (uint32_t)d.n * d.h * d.w;
  const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc(a.y, 0, npix * y_pitch, 0x00020000);
  const __amdgpu_buffer_rsrc_t rr = __builtin_amdgcn_make_buffer_rsrc((void*)a.res, 0, npix * r_pitch, 0x00020000);
  const __amdgpu_buffer_rsrc_t rax = __builtin_amdgcn_make_buffer_rsrc((void*)a.aux, 0, a.aux ? npix * x_pitch : 0u, 0x00020000);
  // pass ps of a unit covers slab rows ps * 8 U + wave * RPW + lane / LPR: patch u = row / 16, patch pixel row % 16
  const int cch = lane % LPR;
  uint32_t yo[2], ro[2], ao[2];
  int ylim[2], xlim[2], srow[2];
#pragma unroll
  for (int ps = 0; ps < 2; ++ps) {
    srow[ps] = ps * 8 * U + wave * RPW + lane / LPR;
    const int r16 = srow[ps] & 15;
    const int dyr = r16 >> 2, dxr = r16 & 3;
    const uint32_t lpix = (uint32_t)((b * d.h + y0 + dyr) * d.w + x0 + dxr);
    const uint32_t ccol = (uint32_t)(cch * 8) * 2u;
    yo[ps] = lpix * y_pitch + (uint32_t)d.out_c_offset * 2u + ccol;
    ro[ps] = lpix * r_pitch + (uint32_t)d.res_c_offset * 2u + ccol;
    ao[ps] = lpix * x_pitch + (uint32_t)d.aux_c_offset * 2u + ccol;
    ylim[ps] = d.h - y0 - dyr;
    xlim[ps] = d.w - x0 - dxr;
  }
  auto voff = [&](uint32_t base, uint32_t pitch, int ps, int jj) -> uint32_t {
    const int pr = jj / 5, pc = jj % 5;
    const bool ok = jj < NP && 4 * pr < ylim[ps] && 4 * pc < xlim[ps];
    return ok ? base + (uint32_t)(4 * pr * d.w + 4 * pc) * pitch : kOobOffset;
  };
  u32x4 rv[RD + 1][2];
  auto fetch_res = [&](auto pc_) {
    constexpr int pi = decltype(pc_)::value;
#pragma unroll
    for (int ps = 0; ps < 2; ++ps)                         // (U == 2: pass ps is patch U pi + ps; U == 1: both passes are patch pi)
      rv[pi % (RD + 1)][ps] = __builtin_amdgcn_raw_buffer_load_b128(rr, voff(ro[ps], r_pitch, ps, U * pi + (U == 2 ? ps : 0)), 0, 0);
  };
  static_for<(RD < NUNIT ? RD : NUNIT)>([&](auto kc) { fetch_res(kc); });
  static_for<NUNIT>([&](auto pc_) {
    constexpr int pi = decltype(pc_)::value;
    char* const slab = smem + (pi & 1) * SLAB;
    static_for<U>([&](auto uc) {
      constexpr int u = decltype(uc)::value;
      if constexpr (U * pi + u < NP) {
        static_for<NFW>([&](auto ic) {
          constexpr int i = decltype(ic)::value;
          *reinterpret_cast<f32x4*>(slab + (u * 16 + c16) * PITCH + (wave * (NFW * 16) + i * 16 + q * 4) * 4) = act4(acc[i][U * pi + u] + b2v[i]);
        });
      }
    });
    if constexpr (pi + RD < NUNIT) fetch_res(std::integral_constant<int, pi + RD>{});
    wait_lds();
    __builtin_amdgcn_s_barrier();                          // the unit is staged by all four waves (and unit pi - 1 has been read by all)
#pragma unroll
    for (int ps = 0; ps < 2; ++ps) {
      const int jj = U * pi + (U == 2 ? ps : 0);
      const f32x4 lo = *reinterpret_cast<const f32x4*>(slab + srow[ps] * PITCH + cch * 32);
      const f32x4 hi = *reinterpret_cast<const f32x4*>(slab + srow[ps] * PITCH + cch * 32 + 16);
      float v[8] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
      if (a.aux) {
        bf16x8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = (bf16_t)v[e];
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o), rax, voff(ao[ps], x_pitch, ps, jj), 0, 0);
      }
      const bf16x8 r8 = __builtin_bit_cast(bf16x8, rv[pi % (RD + 1)][ps]);
      bf16x8 o;
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] = (bf16_t)(v[e] + (float)r8[e]);
      __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o), ry, voff(yo[ps], y_pitch, ps, jj), 0, 0);
    }
  });
#ifdef YOLO_STAMPS
  wait_vmcnt<0>();                                         // (the stores have left: the diagnostic build times the whole epilogue)
#endif
  RU_STAMP(5);
}

template <int CMID, int TPH>
int launch_ruw(const RU20Args& ra, bool force, hipStream_t s) {
  const YoloConvDesc& d = ra.c.d;
  const long tiles = (long)d.n * ((d.h + 4 * TPH - 1) / (4 * TPH)) * ((d.w + kT - 1) / kT);
  if (!force && !resunit_t20_applies(2 * CMID, d.n, d.h, d.w)) return 1;
  if (tiles > 0x7fffffffL) return 1;
  if (pick_only("resunit_t20w<C %d, %dpx x %d couts, 4 waves> grid %ld", 2 * CMID, 80 * TPH, 2 * CMID, tiles)) return 0;
  const size_t dyn = (ra.c.debug & 512) ? 40 * 1024 : 0;                    // bit 512: one workgroup per CU (diagnosis)
  if (ra.c.d.act == YOLO_ACT_LEAKY01) hipLaunchKernelGGL((resunit_t20w_kernel<CMID, TPH, true>), dim3((unsigned)tiles), dim3(256), dyn, s, ra);
  else hipLaunchKernelGGL((resunit_t20w_kernel<CMID, TPH, false>), dim3((unsigned)tiles), dim3(256), dyn, s, ra);
  return yolo_check_launch("yolo_resunit_fwd(t20w)");
}

}  // namespace

namespace yolo_conv {

// The shipped rule: do the 20-pixel-wide tile kernels take a C-channel unit on n maps of h x w?  Partial tiles idle lanes (and
// MFMAs): tiles must cover >= 90 % of the map; few tiles leave CUs without their two workgroups: >= 256 tiles of 20x20 for C = 64,
// >= 512 of 20x20 (C = 128) or 20x8 (C = 256).
bool resunit_t20_applies(int c, int n, int h, int w) {
  const int th = c == 256 ? 8 : 20;
  if (c != 64 && c != 128 && c != 256) return false;
  const long tiles = (long)n * ((h + th - 1) / th) * ((w + kT - 1) / kT);
  return (double)n * h * w >= 0.9 * (20.0 * th) * tiles && tiles >= (c == 64 ? 256 : 512);
}

// 1: not this kernel's case (the caller falls back to the 16x16-tile kernels)
int launch_resunit_t20(const ConvArgs& c, const bf16_t* w1, const float* b1, int kpad1, uint32_t w1_bytes, bool force, hipStream_t s) {
  const YoloConvDesc& d = c.d;
  if (d.cin * 2 != d.cout || d.act == YOLO_ACT_SWISH || !c.res) return 1;
  if (d.cout != 64 && d.cout != 128 && d.cout != 256) return 1;
  const size_t npix = (size_t)d.n * d.h * d.w;
  if (npix * d.out_c_total * 2 >= kOobOffset || npix * d.res_c_total * 2 >= kOobOffset || (c.aux && npix * d.aux_c_total * 2 >= kOobOffset))
    return 1;
  RU20Args ra;
  ra.c = c;
  ra.w1 = w1;
  ra.b1 = b1;
  ra.kpad1 = kpad1;
  ra.w1_bytes = w1_bytes;
  YOLO_SET_STAMPS(ra.c);
  static const int stagger_env = getenv("YOLO_RESUNIT_STAGGER") ? atoi(getenv("YOLO_RESUNIT_STAGGER")) : -1;   // tuning only
  ra.stagger = stagger_env >= 0 ? stagger_env : 0;
  if (d.cout == 128) return launch_ruw<64, 5>(ra, force, s);
  if (d.cout == 256) return launch_ruw<128, 2>(ra, force, s);
  const long tiles = (long)d.n * ((d.h + kT - 1) / kT) * ((d.w + kT - 1) / kT);
  // otherwise the persistent 16x16-tile kernel (YOLO_RESUNIT_DEBUG bit 64 forces this one)
  if (!force && !resunit_t20_applies(64, d.n, d.h, d.w)) return 1;
  if (tiles > 0x7fffffffL) return 1;
  if (pick_only("resunit64_t20<400px x 64 couts, 4 waves> grid %ld", tiles)) return 0;
  if (d.act == YOLO_ACT_LEAKY01) hipLaunchKernelGGL(resunit64_t20_kernel<true>, dim3((unsigned)tiles), dim3(256), 0, s, ra);
  else hipLaunchKernelGGL(resunit64_t20_kernel<false>, dim3((unsigned)tiles), dim3(256), 0, s, ra);
  return yolo_check_launch("yolo_resunit_fwd(t20)");
}

}  // namespace yolo_conv
