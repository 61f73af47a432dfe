// Fused Darknet stem for gfx950 (MI355X): the first two ConvBlocks of the YOLOv3 encoders in one launch
//     mid = act(conv3x3/s1(x, W1) + b1)          3 -> 32 channels at H x W        (models/yolov3_spp.py:98)
//     y   = act(conv3x3/s2(mid, W2) + b2)        32 -> 64 channels at H/2 x W/2   (DownSample conv0, :26-27)
// x is the caller's float32 NCHW batch (reference input contract utils/dataset_csv.py:79-87).
//
// Why: both layers are HBM-bound, and between them sits the largest activation of the network
// (H*W*32 bf16 = 26 MB per 640^2 image): written once, read 2.25x (stride-2 taps) through L2.  Run
// separately they cost 0.14 + 0.24 ms per 16 images; their algorithmic traffic WITHOUT the intermediate
// (read x, write y) is a quarter of what they move.  Here a block owns a 16 x 16 tile of y:
//   stage   the 35 x 35 input halo: NCHW f32 -> registers (prefetched one tile ahead) -> LDS as NHWC8 bf16;
//   phase A mid on the 33 x 33 halo (6 % recompute) by MFMA, K = 10 taps x 8 channels with the weight
//           fragments in registers (as conv1_nchw_kernel), the bias as the accumulators' initial value, act,
//           rounded to bf16 once (as the two-kernel path rounds what it stores) into LDS.  Pixels outside the image are ZERO (the second
//           conv zero-pads `mid`).  Columns are split by parity into two planes so that the stride-2 taps
//           of phase B read consecutive 64-byte rows (XOR-swizzled, conflict-free);
//   phase B the nine taps of the stride-2 conv read `mid` from LDS; W2 (64 x 288, 37 KB) sits in LDS for
//           the whole block;
//   store   bias + act -> bf16 -> LDS staging -> 16 B per lane, 128 contiguous bytes per pixel.
// One persistent block per CU walks its share of the tile list, so W2 / W1 / biases are loaded once per CU.
#include "conv_common.h"

using namespace yolo_conv;

namespace {

struct StemArgs {
  const float* x;       // [n, cin_real, h, w] f32
  const bf16_t* w1;     // packed [>=32][kpad1 >= 80], k = tap*8 + c
  const float* b1;
  const bf16_t* w2;     // packed [>=64][kpad2 >= 288], k = tap*32 + c
  const float* b2;
  bf16_t* y;            // NHWC view
  int n, h, w, cin_real, ho, wo, out_c_total, out_c_offset, kpad1, kpad2, act;
  int debug;            // timing ablations (YOLO_STEM_DEBUG): 1 no input loads, 2 no phase A, 4 no phase B (stem_kernel only), 8 no
                        // stores, 64 producer waves alone (stem2_kernel only)
  unsigned long long* stamps;   // diagnostic build only (-DYOLO_STAMPS, tools/stem_timeline.py)
};

// Diagnostic build only: the first lane of wave 0 (producer) and wave 4 (consumer) of every workgroup records s_memrealtime
// (100 MHz) at up to twelve points of steps 8..23: [workgroup][role][step - 8][point]
#ifdef YOLO_STAMPS
#define STEM_STAMP(role, t, k)                                                                                                 \
  do {                                                                                                                         \
    if (a.stamps && lane == 0 && (wave & 3) == 0 && (t) >= 8 && (t) < 24)                                                      \
    {                                                                                                                        \
      a.stamps[(((size_t)blockIdx.x * 2 + (role)) * 16 + ((t)-8)) * 12 + (k)] = __builtin_amdgcn_s_memrealtime();            \
      if ((k) == 0) a.stamps[(((size_t)blockIdx.x * 2 + (role)) * 16 + ((t)-8)) * 12 + 10] = __builtin_amdgcn_s_memtime();    \
    }                                                                                                                        \
  } while (0)
#else
#define STEM_STAMP(role, t, k)
#endif

typedef u32x4 u32x4_a8 __attribute__((aligned(8)));

template <bool LEAKY, int CIN>
__global__ __launch_bounds__(512) void stem_kernel(const StemArgs a) {
  constexpr int NW = 8;
  constexpr int IW = 35, IP = IW * IW;                 // input halo (pixels)
  constexpr int MW = 33, MP = MW * MW;                 // mid halo
  constexpr int EV_COLS = 17, OD_COLS = 16;            // parity planes of mid: even / odd columns
  constexpr int OD_BASE = MW * EV_COLS;                // first row of the odd plane
  constexpr int MID_B = ((MP * 64 + 1023) / 1024) * 1024;
  constexpr int W2_PITCH = 592;                        // 288 k * 2 B + 16 B: the 16 B-per-lane fragment reads spread over all banks
  constexpr int W2_B = 64 * W2_PITCH;
  constexpr int IN_B = ((IP * 16 + 1023) / 1024) * 1024;
  constexpr int SP = 144;                              // output staging pitch: 64 couts bf16 + 16 B
  constexpr int OUT_B = NW * 16 * SP;                  // staging: 16 pixels per wave at a time
  constexpr int LDS_B = IN_B + OUT_B + MID_B + W2_B;
  constexpr int A_BLOCKS = (MP + 31) / 32;             // 35 blocks of 32 mid pixels
  static_assert(LDS_B <= 160 * 1024, "LDS");

  __shared__ __attribute__((aligned(16))) char smem[LDS_B];
  char* const s_in = smem;
  char* const s_out = smem + IN_B;
  char* const s_mid = smem + IN_B + OUT_B;
  char* const s_w2 = smem + IN_B + OUT_B + MID_B;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r32 = lane & 31, khalf = lane >> 5;

  // persistent blocks: block i owns the i-th of gridDim.x equal runs of the tile list (image, tile row, tile col),
  // so the weights / biases / per-lane tables are set up once per CU and neighbouring halos come from L2
  const int tiles_x = (a.wo + 15) / 16, tiles_y = (a.ho + 15) / 16;
  const long total = (long)a.n * tiles_y * tiles_x;
  const int lb = xcd_swizzle(blockIdx.x, gridDim.x);
  const int t_lo = (int)(lb * total / gridDim.x), t_hi = (int)((lb + 1) * total / gridDim.x);
  if (t_lo >= t_hi) return;

  // ---- block-constant operands: W2 -> LDS, W1 fragments and biases -> registers
  u32x4 w2tmp[5];                                      // W2: 64 rows x 36 sixteen-byte pieces (288 k); stored to LDS below
#pragma unroll
  for (int j = 0; j < 5; ++j) {
    const int i = tid + j * 512, row = i / 36, piece = i - row * 36;
    if (i < 64 * 36) w2tmp[j] = *reinterpret_cast<const u32x4*>(a.w2 + (long)row * a.kpad2 + piece * 8);
  }
  bf16x8 wf1[5];
#pragma unroll
  for (int ks = 0; ks < 5; ++ks) wf1[ks] = *reinterpret_cast<const bf16x8*>(a.w1 + (long)r32 * a.kpad1 + ks * 16 + khalf * 8);
  // biases enter as the accumulators' initial value (register e <-> cout (e&3) + 8*(e>>2) + 4*khalf)
  f32x16 bias1, bias2[2];
#pragma unroll
  for (int g4 = 0; g4 < 4; ++g4)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      bias1[g4 * 4 + e] = a.b1[g4 * 8 + khalf * 4 + e];
#pragma unroll
      for (int i = 0; i < 2; ++i) bias2[i][g4 * 4 + e] = a.b2[i * 32 + g4 * 8 + khalf * 4 + e];
    }
  auto act = [&](float v) -> float { return LEAKY ? fmaxf(v, 0.1f * v) : apply_act(v, a.act); };

  // phase A, per lane and per block slot s (mid pixel q = (wave + 8 s)*32 + r32): LDS offsets are tile-invariant
  constexpr int A_SLOTS = (A_BLOCKS + NW - 1) / NW;    // 5
  int a_rd[A_SLOTS], a_wr[A_SLOTS], a_pos[A_SLOTS];
#pragma unroll
  for (int sl = 0; sl < A_SLOTS; ++sl) {
    const int q = (wave + NW * sl) * 32 + r32;
    const bool valid = q < MP;
    const int qq = valid ? q : 0;
    const int my = qq / MW, mx = qq - my * MW;
    const int R = (mx & 1) ? OD_BASE + my * OD_COLS + (mx >> 1) : my * EV_COLS + (mx >> 1);
    a_rd[sl] = (my * IW + mx) * 16;
    // row base has bits 4,5 clear, so the XOR swizzle of the 16-byte slot can be applied to the address itself
    a_wr[sl] = valid ? ((R * 64 + khalf * 8) ^ (((R >> 2) & 3) << 4)) : -1;
    a_pos[sl] = (my << 8) | mx;
  }
  int tapoff[5];
#pragma unroll
  for (int ks = 0; ks < 5; ++ks) {
    int tap = ks * 2 + khalf;
    if (tap > 8) tap = 8;                                 // tap 9 carries zero weights: read any valid pixel
    const int dh = (tap * 11) >> 5, dw = tap - 3 * dh;
    tapoff[ks] = (dh * IW + dw) * 16;
  }

  // ---- input halo pixels owned by this thread (3 of the 1225), prefetched one tile ahead into registers.
  // Everything per-lane is tile-invariant (offset inside the halo); a tile contributes one scalar base, and only
  // border tiles test pixels against the image.
  constexpr int NCH = CIN ? CIN : 8;                     // channels held in registers
  const long plane = (long)a.h * a.w;
  int h_rel[3], h_yx[3];
#pragma unroll
  for (int u = 0; u < 3; ++u) {
    const int hp = tid + u * 512;
    const int hy = hp / IW, hx = hp - hy * IW;
    h_rel[u] = hp < IP ? hy * a.w + hx : -1;
    h_yx[u] = (hy << 8) | hx;
  }
  float pre[3][NCH];
  auto fetch = [&](int b, int ty, int tx) {
    const int iy0 = 2 * ty * 16 - 2, ix0 = 2 * tx * 16 - 2;     // halo origin in input coordinates
    const float* const base = a.x + ((long)b * a.cin_real) * plane + (long)iy0 * a.w + ix0;
    const bool interior = iy0 >= 0 && iy0 + IW <= a.h && ix0 >= 0 && ix0 + IW <= a.w && !(a.debug & 1);
#pragma unroll
    for (int u = 0; u < 3; ++u) {
      bool ok = h_rel[u] >= 0;
      if (!interior)
        ok = ok && (unsigned)(iy0 + (h_yx[u] >> 8)) < (unsigned)a.h && (unsigned)(ix0 + (h_yx[u] & 255)) < (unsigned)a.w &&
             !(a.debug & 1);
      const float* src = base + h_rel[u];
#pragma unroll
      for (int e = 0; e < NCH; ++e) pre[u][e] = (ok && (CIN || e < a.cin_real)) ? src[e * plane] : 0.f;
    }
  };
  auto commit = [&]() {
#pragma unroll
    for (int u = 0; u < 3; ++u) {
      if (h_rel[u] < 0) continue;
      bf16x8 v;
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = e < NCH ? (bf16_t)pre[u][e] : (bf16_t)0.f;
      *reinterpret_cast<bf16x8*>(s_in + (tid + u * 512) * 16) = v;
    }
  };

  // output rows of this lane in the store passes (tile-invariant part of the address)
  int o_rel[4], o_yx[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int q = wave * 32 + k * 8 + (lane >> 3);          // k = hh*2 + pass
    o_rel[k] = ((q >> 4) * a.wo + (q & 15)) * a.out_c_total + a.out_c_offset + (lane & 7) * 8;
    o_yx[k] = ((q >> 4) << 8) | (q & 15);
  }

  // phase B fragment rows: output pixel q = wave*32 + r32 of the 16 x 16 tile
  const int qy = (wave * 32 + r32) >> 4, qx = r32 & 15;

  // tile cursor (image, tile row, tile col) advanced without divisions; `n*` = the tile being prefetched
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
  fetch(b, ty, tx);
#pragma unroll
  for (int j = 0; j < 5; ++j) {                        // issued before fetch(): landed by the time fetch's data is
    const int i = tid + j * 512, row = i / 36, piece = i - row * 36;
    if (i < 64 * 36) *reinterpret_cast<u32x4*>(s_w2 + row * W2_PITCH + piece * 16) = w2tmp[j];
  }
  for (int t = t_lo; t < t_hi; ++t, advance(tx, ty, b)) {
    const int ox0 = tx * 16, oy0 = ty * 16;
    commit();
    wait_lds();                            // (an LDS-only barrier: __syncthreads() would also drain the previous tile's
    __builtin_amdgcn_s_barrier();          //  stores and, below, the prefetch loads it is supposed to overlap)
                                           // input halo t (and, first time, W2) in LDS; every wave has left phase B of t-1
    if (t + 1 < t_hi) {                    // global loads fly during both MFMA phases
      advance(ntx, nty, nb);
      fetch(nb, nty, ntx);
    }

    // ================= phase A: mid = act(conv3x3/s1(x) + b1) on the 33 x 33 halo -> LDS =================
    // tiles whose whole 33 x 33 halo lies inside the image (all but the border tiles) need no per-pixel test
    const bool interior = 2 * oy0 - 1 >= 0 && 2 * oy0 + 31 < a.h && 2 * ox0 - 1 >= 0 && 2 * ox0 + 31 < a.w;
#pragma unroll
    for (int sl = 0; sl < A_SLOTS; ++sl) {
      if (wave + NW * sl >= A_BLOCKS || (a.debug & 2)) break;
      f32x16 acc = bias1;
#pragma unroll
      for (int ks = 0; ks < 5; ++ks) {
        const bf16x8 xf = *reinterpret_cast<const bf16x8*>(s_in + a_rd[sl] + tapoff[ks]);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf1[ks], xf, acc, 0, 0, 0);
      }
      if (a_wr[sl] >= 0) {
        bool inside = true;
        if (!interior) {
          const int gy = 2 * oy0 - 1 + (a_pos[sl] >> 8), gx = 2 * ox0 - 1 + (a_pos[sl] & 255);   // position in the H x W map
          inside = (unsigned)gy < (unsigned)a.h && (unsigned)gx < (unsigned)a.w;
        }
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
          bf16x4 o;
#pragma unroll
          for (int e = 0; e < 4; ++e) o[e] = (bf16_t)act(acc[g4 * 4 + e]);
          u32x2 bits = __builtin_bit_cast(u32x2, o);
          if (!inside) bits = u32x2{0u, 0u};
          *reinterpret_cast<u32x2*>(s_mid + (a_wr[sl] ^ (g4 << 4))) = bits;
        }
      }
    }
    wait_lds();
    __builtin_amdgcn_s_barrier();          // mid complete

    // ================= phase B: y = act(conv3x3/s2(mid) + b2), 32 pixels x 64 couts per wave =================
    f32x16 acc2[2] = {bias2[0], bias2[1]};
#pragma unroll
    for (int tap = 0; tap < ((a.debug & 4) ? 0 : 9); ++tap) {
      const int dh = tap / 3, dw = tap - 3 * dh;
      const int my = 2 * qy + dh, mxh = qx + (dw >> 1);
      const int R = (dw & 1) ? OD_BASE + my * OD_COLS + mxh : my * EV_COLS + mxh;
      const char* const rowp = s_mid + R * 64;
      const int sw = (R >> 2) & 3;
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        const int g = ks * 2 + khalf;
        const bf16x8 xf = *reinterpret_cast<const bf16x8*>(rowp + ((g ^ sw) << 4));
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const bf16x8 wf = *reinterpret_cast<const bf16x8*>(s_w2 + (i * 32 + r32) * W2_PITCH + (tap * 32 + g * 8) * 2);
          acc2[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf, xf, acc2[i], 0, 0, 0);
        }
      }
    }
    // epilogue: lane = pixel, registers = couts -> staging [16 pixels][64 couts] -> 16 B per lane, 128 B per pixel
    char* const stg = s_out + wave * (16 * SP);
    bf16_t* const ytile = a.y + ((long)(b * a.ho + oy0) * a.wo + ox0) * a.out_c_total;
    const bool full = oy0 + 16 <= a.ho && ox0 + 16 <= a.wo;
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) {
      if ((r32 >> 4) == hh) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int g4 = 0; g4 < 4; ++g4) {
            bf16x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = (bf16_t)act(acc2[i][g4 * 4 + e]);
            *reinterpret_cast<bf16x4*>(stg + (r32 & 15) * SP + (i * 32 + g4 * 8 + khalf * 4) * 2) = o;
          }
      }
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int pass = 0; pass < 2; ++pass) {
        const int k = hh * 2 + pass;
        const bool ok = (full || (oy0 + (o_yx[k] >> 8) < a.ho && ox0 + (o_yx[k] & 255) < a.wo)) && !(a.debug & 8);
        if (ok) {
          const u32x4 val = *reinterpret_cast<const u32x4*>(stg + (pass * 8 + (lane >> 3)) * SP + (lane & 7) * 16);
          *reinterpret_cast<u32x4*>(ytile + o_rel[k]) = val;
        }
      }
      __builtin_amdgcn_wave_barrier();
    }
  }
}

// ------------------------------------------------------------------------------------------------------------------------------
// Round 3: the same computation with the two phases on DIFFERENT waves (3 input channels; other counts keep stem_kernel).
// Ablations of stem_kernel (YOLO_STEM_DEBUG, 32 images, 0.294 ms): without both MFMA phases it still takes 0.169 ms and with nothing
// but its skeleton 0.084 ms - eight waves that all convert, multiply, activate, stage and store one after the other leave every pipe
// idle most of the time (3,700 of 11,700 clocks per tile are MFMA), and W2 read from LDS for every tile makes phase B LDS-bound
// (432 KB per tile).  Here
//   * waves 0-3 (one per SIMD) PRODUCE: input halo NCHW f32 -> registers (two tiles ahead) -> LDS as 8-byte pixels (c0 c1 c2 0),
//     conv1 by MFMA with K = 3 rows x (2 + 2) pixels x 4 channels = 48 (three 32x32x16 steps instead of the five of K = 10 taps x 8
//     channels), act, bf16 -> `mid` of tile t + 1 (double-buffered in LDS);
//   * waves 4-7 (the other wave of each SIMD) CONSUME: the nine stride-2 taps of tile t against W2 held in REGISTERS for the whole
//     kernel (36 fragments = 144 VGPRs; a wave owns 32 pixels x 64 couts) while they activate, stage and store tile t - 1 from a
//     second set of accumulators;
//   * one LDS-only barrier per tile hands `mid` over.  A SIMD's matrix pipe takes the consumer's MFMAs while its producer wave is in
//     VALU / LDS work and vice versa.
// Tiles are 16 x 8 outputs (mid halo 33 x 17, input halo 35 x 19).  `mid` rows are 80 bytes apart (64 + 16: sixteen consecutive rows
// hit sixteen different 16-byte bank groups), so every LDS address is a per-lane base plus a compile-time offset.
// conv1 sums its 27 products in another order than the K = 80 kernels (fp32 accumulation inside the MFMA): `mid` can differ from
// theirs by one bf16 ulp on a few elements.
// Measured (32 images 640x640): 0.285 -> 0.209 ms.  tools/stem_timeline.py (stamps build): a step takes ~2.1 us, the producers'
// conv1 1.6 us of it (1.2 us with the consumers parked, 0.2 us per block of 3 MFMAs + 36 VALU + 7 LDS instructions), the consumers'
// 36 MFMAs + epilogue 1.2-1.3 us: both roles run at about twice their issue cost and the producers are the longer pole.
template <bool LEAKY>
__global__ __launch_bounds__(512) void stem2_kernel(const StemArgs a) {
  constexpr int TW = 16, TH = 8;                                     // output tile
  constexpr int IW = 2 * TW + 3, IH = 2 * TH + 3, IP = IW * IH;      // input halo 35 x 19
  constexpr int MW = 2 * TW + 1, MH = 2 * TH + 1, MP = MW * MH;      // mid halo 33 x 17
  constexpr int EV_COLS = TW + 1, OD_COLS = TW, OD_BASE = MH * EV_COLS;
  constexpr int MPITCH = 80;
  constexpr int MID_B = (MP + 15) * MPITCH;                          // + 15 rows that lanes without a pixel write to
  constexpr int IN_B = ((IP * 8 + 1023) / 1024) * 1024;
  constexpr int SP = 144, OUT_B = 4 * 32 * SP;                       // staging: 32 pixels per consumer wave
  constexpr int BIAS_B = 4 * 64;                                     // b2 in accumulator order: [cout half][k half][16] f32
  constexpr int LDS_B = 2 * IN_B + 2 * MID_B + OUT_B + BIAS_B;
  constexpr int A_BLOCKS = (MP + 31) / 32, A_SLOTS = (A_BLOCKS + 3) / 4;   // 18 blocks of 32 mid pixels, 5 per producer wave
  static_assert(LDS_B <= 160 * 1024 && MID_B % 16 == 0 && MID_B + (2 * EV_COLS + 1) * MPITCH + 64 < 65536, "LDS");

  __shared__ __attribute__((aligned(16))) char smem[LDS_B];
  char* const s_mid = smem;                            // first: the consumers' offsets into it are instruction immediates
  char* const s_in = smem + 2 * MID_B;
  char* const s_out = smem + 2 * MID_B + 2 * IN_B;
  char* const s_b2 = smem + 2 * MID_B + 2 * IN_B + OUT_B;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r32 = lane & 31, khalf = lane >> 5;

  const int tiles_x = (a.wo + TW - 1) / TW, tiles_y = (a.ho + TH - 1) / TH;
  const long total = (long)a.n * tiles_y * tiles_x;
  const int lb = xcd_swizzle(blockIdx.x, gridDim.x);
  const int t_lo = (int)(lb * total / gridDim.x), t_hi = (int)((lb + 1) * total / gridDim.x);
  const int count = t_hi - t_lo;
  if (count <= 0) return;
  auto act = [&](float v) -> float { return LEAKY ? fmaxf(v, 0.1f * v) : apply_act(v, a.act); };
  // four activated values as bf16.  LeakyReLU: fmaxf() costs a canonicalising v_max v, v, v per operand on top of mul + max (MFMA
  // results are not known to be quiet); v_pk_mul_f32 + the bare v_max_f32 give the same bits in 2 instead of 3.5 VALU per value
  auto act4 = [&](float v0, float v1, float v2, float v3) -> u32x2 {
    bf16x4 o;
    if (LEAKY) {
      const f32x2 lo = f32x2{v0, v1} * 0.1f, hi = f32x2{v2, v3} * 0.1f;
      o[0] = (bf16_t)raw_max(v0, lo[0]);
      o[1] = (bf16_t)raw_max(v1, lo[1]);
      o[2] = (bf16_t)raw_max(v2, hi[0]);
      o[3] = (bf16_t)raw_max(v3, hi[1]);
    } else {
      o[0] = (bf16_t)act(v0);
      o[1] = (bf16_t)act(v1);
      o[2] = (bf16_t)act(v2);
      o[3] = (bf16_t)act(v3);
    }
    return __builtin_bit_cast(u32x2, o);
  };
  auto advance = [&](int& x_, int& y_, int& b_) {
    if (++x_ == tiles_x) {
      x_ = 0;
      if (++y_ == tiles_y) {
        y_ = 0;
        ++b_;
      }
    }
  };
  const int tx0 = t_lo % tiles_x, ty0 = (t_lo / tiles_x) % tiles_y, b0 = t_lo / (tiles_x * tiles_y);

  if (wave < 4) {
    // =========================================== producer waves ===========================================
    // W1 fragment of kernel row dh: lane (cout r32, k half): k element e = pixel khalf + (e >> 2) of the row, channel e & 3, i.e.
    // pixels (0, 1) for k half 0 and (1, 2) for k half 1, whose pixel 1 carries zero weights like every fourth channel
    bf16x8 wf1[3];
#pragma unroll
    for (int dh = 0; dh < 3; ++dh)
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int dw = khalf + (e >> 2), c = e & 3;
        wf1[dh][e] = (c < 3 && !(khalf && dw == 1)) ? a.w1[(long)r32 * a.kpad1 + (dh * 3 + dw) * 8 + c] : (bf16_t)0.f;
      }
    f32x16 bias1;
#pragma unroll
    for (int g4 = 0; g4 < 4; ++g4)
#pragma unroll
      for (int e = 0; e < 4; ++e) bias1[g4 * 4 + e] = a.b1[g4 * 8 + khalf * 4 + e];
    // per block slot sl (mid pixel q = (wave + 4 sl) * 32 + r32): the lane's 16-byte read per kernel row (two pixels, 8-byte
    // aligned: ds_read2_b64; the zero-weight pixel of k half 1 lies inside the 3 x 3 window, so a non-finite input pixel touches
    // exactly the outputs it touches in the reference) and its write row in `mid`
    int a_rd[A_SLOTS], a_wr[A_SLOTS], a_pos[A_SLOTS];
#pragma unroll
    for (int sl = 0; sl < A_SLOTS; ++sl) {
      const int q = (wave + 4 * sl) * 32 + r32;
      const bool valid = q < MP;
      const int qq = valid ? q : 0;
      const int my = qq / MW, mx = qq - my * MW;
      const int R = (mx & 1) ? OD_BASE + my * OD_COLS + (mx >> 1) : my * EV_COLS + (mx >> 1);
      a_rd[sl] = (my * IW + mx + khalf) * 8;
      // (odd-plane rows keep the two 8-byte halves of every 16-byte slot swapped: a 16-lane write group holds 8 even and 8 odd
      //  columns, and rows a multiple of 16 bytes apart all start on banks = 0 mod 4 - the swap moves the odd columns' 8 bytes to
      //  banks = 2 mod 4; the consumers' W2 fragments of the dw = 1 taps carry the same swap in k)
      a_wr[sl] = (valid ? R : MP + r32 % 15) * MPITCH + (khalf ^ (mx & 1)) * 8;
      a_pos[sl] = (my << 8) | mx;
    }
    // input halo: 3 of its 665 pixels per lane, by buffer loads without a branch (a pixel outside the image, or a lane without a
    // third pixel, gets the out-of-range offset and reads 0).  With `ok ? load : 0` under branches the compiler had to put an
    // s_waitcnt vmcnt(0) between the pixels (the zeroing moves write registers that loads in flight may target): one HBM round
    // trip, 0.8 us, on the producers' critical path in every step.
    const int plane = a.h * a.w;
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)a.x, 0, (uint32_t)a.n * 3u * plane * 4u, 0x00020000);
    int h_rel[3], h_yx[3];
#pragma unroll
    for (int u = 0; u < 3; ++u) {
      const int hp = tid + u * 256;
      const int hy = hp / IW, hx = hp - hy * IW;
      h_rel[u] = hp < IP ? hy * a.w + hx : -1;
      h_yx[u] = (hy << 8) | hx;
    }
    float pre[3][3];
    auto fetch = [&](int b, int ty, int tx) {
      const int iy0 = 2 * ty * TH - 2, ix0 = 2 * tx * TW - 2;
      const int base = b * 3 * plane + iy0 * a.w + ix0;            // (may be negative: only in-image pixels use it)
#pragma unroll
      for (int u = 0; u < 3; ++u) {
        const bool ok = (h_rel[u] >= 0) & ((unsigned)(iy0 + (h_yx[u] >> 8)) < (unsigned)a.h) &
                        ((unsigned)(ix0 + (h_yx[u] & 255)) < (unsigned)a.w) & !(a.debug & 1);
        const uint32_t vo = ok ? (uint32_t)(base + h_rel[u]) * 4u : kOobOffset;
#pragma unroll
        for (int e = 0; e < 3; ++e)
          pre[u][e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rx, vo, (uint32_t)(e * plane) * 4u, 0));
      }
    };
    auto commit = [&](char* dst) {
#pragma unroll
      for (int u = 0; u < 3; ++u) {
        if (h_rel[u] < 0) continue;
        bf16x4 v;
#pragma unroll
        for (int e = 0; e < 3; ++e) v[e] = (bf16_t)pre[u][e];
        v[3] = (bf16_t)0.f;
        *reinterpret_cast<bf16x4*>(dst + (tid + u * 256) * 8) = v;
      }
    };
    int ptx = tx0, pty = ty0, pb = b0;        // tile of the coming conv1 phase
    int ftx = tx0, fty = ty0, fb = b0;        // tile last fetched
    fetch(fb, fty, ftx);
    commit(s_in);
    if (count > 1) {
      advance(ftx, fty, fb);
      fetch(fb, fty, ftx);
    }
    wait_lds();
    __builtin_amdgcn_s_barrier();
    // step t: the halo of tile t + 2 (in registers since the start of step t - 1) goes to input buffer t&1, which conv1 of tile t
    // read one step ago, and the loads of tile t + 3 are issued into the freed registers; then conv1 of tile t + 1 (input buffer
    // (t+1)&1 -> mid buffer (t+1)&1) while the consumers work on tile t.  (Two register sets, loads two steps ahead, ran slower:
    // the compiler's wait before the older set's first use - vmcnt(6) with 18 loads in flight - also waits for three of the loads
    // just issued.)
    for (int t = -1; t < count; ++t) {
      STEM_STAMP(0, t, 0);
      if (t + 2 < count) commit(s_in + (t & 1) * IN_B);
      if (t + 3 < count) {
        advance(ftx, fty, fb);
        fetch(fb, fty, ftx);
      }
      STEM_STAMP(0, t, 2);
      if (t + 1 < count) {
        const char* const in = s_in + ((t + 1) & 1) * IN_B;
        char* const mid = s_mid + ((t + 1) & 1) * MID_B;
        const int ox0 = ptx * TW, oy0 = pty * TH;
        const bool interior = 2 * oy0 - 1 >= 0 && 2 * oy0 + MH - 2 < a.h && 2 * ox0 - 1 >= 0 && 2 * ox0 + MW - 2 < a.w;
        // One straight-line body for all five blocks (no per-block branches: lanes past the halo's last pixel, and the fifth block
        // of waves 2 and 3, compute on pixel 0 and write to the spare rows), block j + 1's reads and MFMAs issued before block j is
        // activated and written: measured alone, a block-at-a-time loop took 890 clocks per block (read latency -> dependent
        // MFMAs -> 80 VALU -> writes, nothing overlapping) and WAS the kernel's critical path.
        // Three blocks in flight, order pinned in the source (sched_barrier): the LDS reads of block j + 2, the three dependent
        // MFMAs of block j + 1 and the activation / bf16 / write of block j's 16 values, one MFMA per third of the VALU work.
        auto conv1 = [&](auto border) {
          auto reads = [&](int sl, u32x4 (&xf)[3]) {
#pragma unroll
            for (int dh = 0; dh < 3; ++dh) xf[dh] = *reinterpret_cast<const u32x4_a8*>(in + a_rd[sl] + dh * IW * 8);
          };
          auto put = [&](const f32x16& acc, int sl, int g4, bool inside) {
            u32x2 bits = act4(acc[g4 * 4], acc[g4 * 4 + 1], acc[g4 * 4 + 2], acc[g4 * 4 + 3]);
            if (!inside) bits = u32x2{0u, 0u};
            *reinterpret_cast<u32x2*>(mid + a_wr[sl] + g4 * 16) = bits;
          };
          u32x4 xf[3][3];                                  // [block % 3][kernel row]
          f32x16 accs[2];
          reads(0, xf[0]);
          reads(1, xf[1]);
          accs[0] = bias1;
#pragma unroll
          for (int dh = 0; dh < 3; ++dh)
            accs[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf1[dh], __builtin_bit_cast(bf16x8, xf[0][dh]), accs[0], 0, 0, 0);
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int sl = 0; sl < A_SLOTS; ++sl) {
            const bool more = sl + 1 < A_SLOTS;
            bool inside = true;
            if (decltype(border)::value) {
              const int gy = 2 * oy0 - 1 + (a_pos[sl] >> 8), gx = 2 * ox0 - 1 + (a_pos[sl] & 255);
              inside = (unsigned)gy < (unsigned)a.h && (unsigned)gx < (unsigned)a.w;
            }
            f32x16& nxt = accs[(sl + 1) & 1];
            const f32x16& cur = accs[sl & 1];
            STEM_STAMP(0, t, 4 + sl);
            if (sl + 2 < A_SLOTS) reads(sl + 2, xf[(sl + 2) % 3]);
            if (more) nxt = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf1[0], __builtin_bit_cast(bf16x8, xf[(sl + 1) % 3][0]), bias1, 0, 0, 0);
            put(cur, sl, 0, inside);
            __builtin_amdgcn_sched_barrier(0);
            if (more) nxt = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf1[1], __builtin_bit_cast(bf16x8, xf[(sl + 1) % 3][1]), nxt, 0, 0, 0);
            put(cur, sl, 1, inside);
            __builtin_amdgcn_sched_barrier(0);
            if (more) nxt = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf1[2], __builtin_bit_cast(bf16x8, xf[(sl + 1) % 3][2]), nxt, 0, 0, 0);
            put(cur, sl, 2, inside);
            put(cur, sl, 3, inside);
            __builtin_amdgcn_sched_barrier(0);
          }
        };
        if (!(a.debug & 2)) {
          if (interior) conv1(std::false_type{});
          else conv1(std::true_type{});
        }
        advance(ptx, pty, pb);
      }
      STEM_STAMP(0, t, 1);
      wait_lds();
      STEM_STAMP(0, t, 3);
      __builtin_amdgcn_s_barrier();
    }
  } else {
    // =========================================== consumer waves ===========================================
    const int bw = wave - 4;
    bf16x8 wreg[9][2][2];                                    // [tap][k step][cout half]: couts i*32 + r32, k = tap*32 + (ks*2+khalf)*8
#pragma unroll
    for (int tap = 0; tap < 9; ++tap)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int i = 0; i < 2; ++i)
        {
          const bf16x8 w = *reinterpret_cast<const bf16x8*>(a.w2 + (long)(i * 32 + r32) * a.kpad2 + tap * 32 + (ks * 2 + khalf) * 8);
          wreg[tap][ks][i] = (tap % 3 == 1) ? __builtin_shufflevector(w, w, 4, 5, 6, 7, 0, 1, 2, 3) : w;   // (odd plane: halves swapped)
        }
    if (tid - 256 < 64) {                                    // b2 in accumulator order
      const int idx = tid - 256, i = idx >> 5, kh = (idx >> 4) & 1, r = idx & 15;
      reinterpret_cast<float*>(s_b2)[idx] = a.b2[i * 32 + (r >> 2) * 8 + kh * 4 + (r & 3)];
    }
    // This lane's output pixel: column qx = r32 & 15 of tile row 2 bw + (parity of r32 >> 2), so that each of ds_read_b128's lane
    // groups ({0-3, 12-15, 20-27}, {4-11, 16-19, 28-31}: MI355X_MICROARCH.md, LDS) reads 16 consecutive 80-byte rows of ONE tile row
    // (with r32 >> 4 as the row, a group's lanes 20-27 sat 34 rows further and hit two of its banks a second time).
    // Tap (dh, dw) reads mid row 2 qy + dh, column 2 qx + dw: even columns in the even plane at qx + (dw >> 1), odd ones in the
    // odd plane at qx.
    const int qrow = __builtin_popcount((r32 >> 2) & 7) & 1;
    const int qy = bw * 2 + qrow, qx = r32 & 15;
    const char* const ev = s_mid + (2 * qy * EV_COLS + qx) * MPITCH + khalf * 16;
    const char* const od = s_mid + (OD_BASE + 2 * qy * OD_COLS + qx) * MPITCH + khalf * 16;
    // store pass k (0..3): pixel row bw * 2 + (k >> 1), column (k & 1) * 8 + (lane >> 3) of the tile, 16 bytes (lane & 7) of its 128
    const int o_x = lane >> 3;
    const uint32_t o_rel = (uint32_t)((bw * 2 * a.wo + o_x) * a.out_c_total + a.out_c_offset + (lane & 7) * 8) * 2u;
    const __amdgpu_buffer_rsrc_t ry =
        __builtin_amdgcn_make_buffer_rsrc(a.y, 0, (uint32_t)a.n * a.ho * a.wo * a.out_c_total * 2u, 0x00020000);
    char* const stg = s_out + bw * (32 * SP);
    f32x16 acc2[2][2];                                       // [tile parity][cout half]
    int tx = tx0, ty = ty0, b = b0;                          // tile of the next epilogue
    // One step: the 36 MFMAs of a tile from mid buffer P (TAPS), and activation -> staging -> stores of the PREVIOUS tile from the
    // other accumulator set (EPI).  The order is pinned in the source - per tap: the next tap's two LDS reads, four MFMAs, one eighth
    // of the activation work, sched_barrier - because left alone the scheduler issues all MFMAs first and the epilogue behind them.
    auto step = [&](auto parity, auto do_taps, auto do_epi, int tstamp) {
      constexpr int P = decltype(parity)::value;
      (void)tstamp;
      constexpr bool TAPS = decltype(do_taps)::value, EPI = decltype(do_epi)::value;
      auto row = [&](int tap) {
        const int dh = tap / 3, dw = tap - 3 * dh;
        return ((dw & 1) ? od + dh * OD_COLS * MPITCH : ev + (dh * EV_COLS + (dw >> 1)) * MPITCH) + P * MID_B;
      };
      bf16x8 xf[2][2];
      if (TAPS) {
#pragma unroll
        for (int i = 0; i < 2; ++i) acc2[P][i] = *reinterpret_cast<const f32x16*>(s_b2 + (i * 2 + khalf) * 64);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) xf[0][ks] = *reinterpret_cast<const bf16x8*>(row(0) + ks * 32);
      }
#pragma unroll
      for (int tap = 0; tap < 9; ++tap) {
        if (TAPS && tap % 3 == 0) STEM_STAMP(1, tstamp, 4 + tap / 3);
        if (TAPS) {
          if (tap < 8)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) xf[(tap + 1) & 1][ks] = *reinterpret_cast<const bf16x8*>(row(tap + 1) + ks * 32);
#pragma unroll
          for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int i = 0; i < 2; ++i)
              acc2[P][i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wreg[tap][ks][i], xf[tap & 1][ks], acc2[P][i], 0, 0, 0);
        }
        if (EPI && tap < 8) {
          const int i = tap >> 2, g4 = tap & 3;
          *reinterpret_cast<u32x2*>(stg + (qrow * 16 + qx) * SP + (i * 32 + g4 * 8 + khalf * 4) * 2) =
              act4(acc2[P ^ 1][i][g4 * 4], acc2[P ^ 1][i][g4 * 4 + 1], acc2[P ^ 1][i][g4 * 4 + 2], acc2[P ^ 1][i][g4 * 4 + 3]);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      if (TAPS) STEM_STAMP(1, tstamp, 7);
      if (EPI) {                                             // staging [32 pixels][64 couts] -> 16 B per lane, 128 B per pixel
        const int ox0 = tx * TW, oy0 = ty * TH;
        const uint32_t tile_off = (uint32_t)(((b * a.ho + oy0) * a.wo + ox0) * a.out_c_total) * 2u;
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int k = 0; k < 4; ++k) {                        // (no branches: pixels past the map get the out-of-range offset)
          const bool ok = (oy0 + bw * 2 + (k >> 1) < a.ho) & (ox0 + (k & 1) * 8 + o_x < a.wo) & !(a.debug & 8);
          const u32x4 val = *reinterpret_cast<const u32x4*>(stg + (k * 8 + (lane >> 3)) * SP + (lane & 7) * 16);
          const uint32_t off = tile_off + (uint32_t)(((k >> 1) * a.wo + (k & 1) * 8) * a.out_c_total) * 2u + o_rel;
          __builtin_amdgcn_raw_buffer_store_b128(val, ry, ok ? off : kOobOffset, 0, 0);
        }
        __builtin_amdgcn_wave_barrier();
        advance(tx, ty, b);
      }
    };
    using P0 = std::integral_constant<int, 0>;
    using P1 = std::integral_constant<int, 1>;
    wait_lds();
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_s_barrier();                            // step -1: the producers' first conv1
    // step t: the taps of tile t with the epilogue of tile t - 1
    step(P0{}, std::true_type{}, std::false_type{}, 0);
    wait_lds();
    __builtin_amdgcn_s_barrier();
    if (a.debug & 64) {                                      // timing only: the producers alone
      for (int t = 1; t < count; ++t) __builtin_amdgcn_s_barrier();
      return;
    }
    for (int t = 1; t < count; t += 2) {
      STEM_STAMP(1, t, 0);
      step(P1{}, std::true_type{}, std::true_type{}, t);
      STEM_STAMP(1, t, 1);
      wait_lds();
      STEM_STAMP(1, t, 3);
      __builtin_amdgcn_s_barrier();
      if (t + 1 < count) {
        STEM_STAMP(1, t + 1, 0);
        step(P0{}, std::true_type{}, std::true_type{}, t + 1);
        STEM_STAMP(1, t + 1, 1);
        wait_lds();
        STEM_STAMP(1, t + 1, 3);
        __builtin_amdgcn_s_barrier();
      }
    }
    if ((count - 1) & 1) step(P0{}, std::false_type{}, std::true_type{}, 0);      // epilogue of accumulator set 1
    else step(P1{}, std::false_type{}, std::true_type{}, 0);
  }
}

}  // namespace

extern "C" int yolo_stem_supported(int cin_real, int c1, int c2, int h, int w) {
  return cin_real >= 1 && cin_real <= 8 && c1 == 32 && c2 == 64 && h >= 2 && w >= 2;
}

extern "C" int yolo_stem_fwd(const float* x_nchw, int cin_real, const void* w1_packed, const float* b1, int kpad1,
                             const void* w2_packed, const float* b2, void* y, const YoloConvDesc* dp, yolo_stream_t s) {
  YOLO_REQUIRE(x_nchw && w1_packed && b1 && w2_packed && b2 && y && dp, "stem: null pointer");
  const YoloConvDesc& d = *dp;     // the stride-2 conv: h x w = size of x and of the intermediate, cin 32, cout 64
  YOLO_REQUIRE(yolo_stem_supported(cin_real, d.cin, d.cout, d.h, d.w), "stem: needs 1..8 input channels, 32 -> 64 channels");
  YOLO_REQUIRE(d.ksize == 3 && d.stride == 2 && d.pad == 1 && d.ho == (d.h - 1) / 2 + 1 && d.wo == (d.w - 1) / 2 + 1 &&
                   !d.upsample2x && d.out_dtype == YOLO_DT_BF16,
               "stem: descriptor must be the 3x3 / stride 2 / pad 1 conv");
  YOLO_REQUIRE(d.out_c_offset % 8 == 0 && d.out_c_total % 8 == 0 && d.out_c_offset + d.cout <= d.out_c_total, "stem: bad output view");
  YOLO_REQUIRE(kpad1 >= 80 && kpad1 % 8 == 0 && d.kpad >= 288 && d.kpad % 8 == 0, "stem: bad weight packing");
  StemArgs a;
  a.x = x_nchw;
  a.w1 = (const bf16_t*)w1_packed;
  a.b1 = b1;
  a.w2 = (const bf16_t*)w2_packed;
  a.b2 = b2;
  a.y = (bf16_t*)y;
  a.n = d.n;
  a.h = d.h;
  a.w = d.w;
  a.cin_real = cin_real;
  a.ho = d.ho;
  a.wo = d.wo;
  a.out_c_total = d.out_c_total;
  a.out_c_offset = d.out_c_offset;
  a.kpad1 = kpad1;
  a.kpad2 = d.kpad;
  a.act = d.act;
  static const int dbg = getenv("YOLO_STEM_DEBUG") ? atoi(getenv("YOLO_STEM_DEBUG")) : 0;
  a.debug = dbg;
  a.stamps = nullptr;
  YOLO_SET_STAMPS(a);
  static const int n_cu = [] {
    int dev = 0, v = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0)
      v = 256;
    return v;
  }();
  hipStream_t st = (hipStream_t)s;
  const bool leaky = d.act == YOLO_ACT_LEAKY01;
  static const bool one_role = getenv("YOLO_STEM_ONE_ROLE") != nullptr;      // A/B knob: stem_kernel (round 2)
  // stem2_kernel addresses x and y through buffer descriptors with 32-bit byte offsets (and keeps pixel indices in int):
  // batches whose input or output view reaches the out-of-range marker (~4 GB: 873 images of 640x640) take stem_kernel
  const bool fits32 = (size_t)d.n * 3 * d.h * d.w * 4 < kOobOffset && (size_t)d.n * d.ho * d.wo * d.out_c_total * 2 < kOobOffset;
  if (!one_role && cin_real == 3 && fits32) {
    const long tiles = (long)d.n * ((d.ho + 7) / 8) * ((d.wo + 15) / 16);
    if (tiles > 0x7fffffffL) return yolo_set_error(YOLO_E_UNSUPPORTED, "stem: too many tiles");
    const dim3 g((unsigned)(tiles < n_cu ? tiles : n_cu)), blk(512);     // one persistent block per CU
    if (leaky) hipLaunchKernelGGL((stem2_kernel<true>), g, blk, 0, st, a);
    else hipLaunchKernelGGL((stem2_kernel<false>), g, blk, 0, st, a);
    return yolo_check_launch("yolo_stem_fwd");
  }
  const long tiles = (long)d.n * ((d.ho + 15) / 16) * ((d.wo + 15) / 16);
  if (tiles > 0x7fffffffL) return yolo_set_error(YOLO_E_UNSUPPORTED, "stem: too many tiles");
  const long grid = tiles < n_cu ? tiles : n_cu;       // one persistent block per CU (144 KB LDS each)
  const dim3 g((unsigned)grid), blk(512);
  if (cin_real == 3) {
    if (leaky) hipLaunchKernelGGL((stem_kernel<true, 3>), g, blk, 0, st, a);
    else hipLaunchKernelGGL((stem_kernel<false, 3>), g, blk, 0, st, a);
  } else {
    if (leaky) hipLaunchKernelGGL((stem_kernel<true, 0>), g, blk, 0, st, a);
    else hipLaunchKernelGGL((stem_kernel<false, 0>), g, blk, 0, st, a);
  }
  return yolo_check_launch("yolo_stem_fwd");
}
