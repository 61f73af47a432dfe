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
// hardware writes zeros.  NS-stage LDS ring, counted vmcnt, one raw s_barrier per K step.
#include "conv_common.h"
#include "nms_common.h"
#include "head_epilogue.h"
#include <stdlib.h>

using namespace yolo_conv;

// fragment read from LDS; the timing-only ablation build YOLO_ABL_NOLDS (tools/block_timeline.py) feeds the MFMAs
// whatever the registers hold instead
#ifdef YOLO_ABL_NOLDS
__device__ __forceinline__ bf16x8 lds_frag_stub() {
  bf16x8 v;
  asm volatile("" : "=v"(v));
  return v;
}
#define YOLO_LDS_FRAG(p) lds_frag_stub()
#else
#define YOLO_LDS_FRAG(p) (*reinterpret_cast<const bf16x8*>(p))
#endif

namespace {

// BM pixels x BN couts block tile, WAVES_M x WAVES_N waves (4 or 8), K step BK (32 or 64), NS-stage LDS ring.
// Stage s+NS-1 is in flight (LDS-DMA) while stage s is multiplied: the waits are COUNTED
// (s_waitcnt vmcnt(N), never 0 in steady state) and the barrier is a raw s_barrier, so the DMA
// stays in flight across it (a __syncthreads() would drain vmcnt to 0).
// FAST (cin % BK == 0): a K step never straddles a filter tap, so tap / channel offset are wave-uniform
// scalars, the image-border test is one precomputed bit per tap, and the weight address advances through
// the instruction's scalar offset: ~3 VALU per LDS-DMA instead of ~15.
// LDS_EPI (bf16 output, cout % 32 == 0): the finished tile goes registers -> LDS (fp32) -> global so that
// residual read, pre-add copy and store are 16-byte-per-lane accesses over whole channel runs.
template <int BM, int BN, int WAVES_M, int WAVES_N, int BK, int NS, bool FAST, bool LDS_EPI, bool MFMA16 = false, bool DECODE = false,
          int NP = 0, bool SPLITK = false>
__global__ __launch_bounds__(64 * (WAVES_M * WAVES_N + NP)) void conv_igemm_bf16_kernel(const ConvArgs a) {
  static_assert(!SPLITK || (FAST && LDS_EPI && MFMA16 && !DECODE), "split-K: 16x16x32 path with the LDS epilogue");
  constexpr int NW = WAVES_M * WAVES_N;
  // NP > 0: NP extra LOADER waves issue every LDS-DMA of the block; the NW MFMA waves only read LDS and multiply
  // (an LDS-DMA costs its issuing wave 60-180 cycles: ablation: profiles/r02_DESIGN_lab_notebook.md 3.1c).  NL = number of loader waves.
  constexpr int NL = NP > 0 ? NP : NW;
  static_assert(NP == 0 || (NS <= 3 && FAST && NP % 2 == 0), "loader waves: 2- or 3-stage ring, FAST path");
  static_assert(NW == 4 || NW == 8 || NW == 16, "4, 8 or 16 waves");
  static_assert(BK == 32 || BK == 64, "BK");
  static_assert(NS >= 2 && NS <= 4, "ring depth");
  constexpr int TM = BM / WAVES_M, TN = BN / WAVES_N;  // per-wave tile (pixels x couts)
  constexpr int NI = TM / 32, MI = TN / 32;
  constexpr int ROWB = BK * 2;                         // bytes per LDS row
  constexpr int CPR = BK / 8;                          // 16-byte chunks per row
  constexpr int RPP = 1024 / ROWB;                     // rows per 1-KiB LDS-DMA piece (16 or 8)
  constexpr int BNL = BN < NL * RPP ? NL * RPP : BN;   // weight rows staged (every loader issues the same count)
  constexpr int PIT = BM / (NL * RPP), WIT = BNL / (NL * RPP);   // LDS-DMA instructions per loader thread per stage
  constexpr int LPS = PIT + WIT;
  constexpr int STAGE_B = (BM + BNL) * ROWB;
  static_assert(TM % 32 == 0 && TN % 32 == 0 && PIT >= 1 && WIT >= 1, "tile");
  static_assert(NS * STAGE_B <= 160 * 1024, "LDS");

  constexpr int RING_B = NS * STAGE_B;
  constexpr int DP = BN + 4;                           // DECODE: fp32 staging pitch (floats) of the [BM][BN] tile
  constexpr int EPI_B = DECODE ? BM * DP * 4 : (LDS_EPI ? NW * TM * kEpiPitch : 0);
  constexpr int LDS_B = EPI_B > RING_B ? EPI_B : RING_B;   // the epilogue staging reuses the ring
  static_assert(!DECODE || (!LDS_EPI && !MFMA16 && BM % NW == 0), "DECODE instances use the plain 32x32 accumulators");
  __shared__ __attribute__((aligned(16))) char smem[LDS_B];   // per stage: [BNL weight rows][BM pixel rows]

  YOLO_BLOCK_STAMP(a);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WAVES_N, wn = wave % WAVES_N;
  const int lw = NP == 0 ? wave : wave - NW;            // its index among the loaders
  const YoloConvDesc& d = a.d;

  // XCD-aware tile order: blocks with equal blockIdx%8 share an L2; give each such group a contiguous
  // run of tiles (cout tile fastest) so the pixel tile is re-read from that L2 (bijective for any grid).
  int m0, n0;
  {
    const int swz = xcd_swizzle(blockIdx.x, gridDim.x);
    const int mt = swz / a.n_tiles;
    m0 = mt * BM;
    n0 = (swz - mt * a.n_tiles) * BN;
  }

  const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)a.x, 0, a.x_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void*)a.w, 0, a.w_bytes, 0x00020000);

  // ---- per-thread staging state -----------------------------------------------------------------
  // piece p = it*NW + wave covers LDS rows [p*RPP, (p+1)*RPP); lane -> (row in piece, physical chunk slot).
  // swizzle: physical slot = logical chunk ^ f(row);  f = (row>>2)&3 for 64-B rows, (row>>1)&7 for 128-B rows.
  const int frow = lane / CPR;
  const int fsw = (BK == 32) ? swz32(lane >> 4) : (((lane >> 4) + 4 * (lw & 1)) & 7);
  const int chunk = (lane & (CPR - 1)) ^ fsw;          // logical 8-channel chunk this lane fetches
  const int ntaps = d.ksize * d.ksize;
  int px_base[PIT], px_hi0[PIT], px_wi0[PIT];          // generic path
  uint32_t px_mask[PIT];                               // bit t: tap t of this pixel is inside the image
  const int hw_out = d.ho * d.wo;
#pragma unroll
  for (int it = 0; it < PIT; ++it) {
    const int m = m0 + (it * NL + lw) * RPP + frow;
    const bool ok = m < a.M;
    const int mm = ok ? m : 0;
    const int b = mm / hw_out, rem = mm - b * hw_out;
    const int oh = rem / d.wo, ow = rem - oh * d.wo;
    px_hi0[it] = oh * d.stride - d.pad;
    px_wi0[it] = ow * d.stride - d.pad;
    px_base[it] = ((b * d.h + px_hi0[it]) * d.w + px_wi0[it]) * d.in_c_total + d.in_c_offset;
    // bit t = kh*3+kw set iff that tap lies inside the image: outer product of 3 row bits and 3 column bits
    uint32_t mask;
    if (d.ksize == 3) {
      uint32_t cb = 0, rb = 0;
#pragma unroll
      for (int t = 0; t < 3; ++t) {
        cb |= (uint32_t)((unsigned)(px_wi0[it] + t) < (unsigned)d.w) << t;
        rb |= (uint32_t)((unsigned)(px_hi0[it] + t) < (unsigned)d.h) << t;
      }
      mask = ((rb & 1) ? cb : 0) | ((rb & 2) ? cb << 3 : 0) | ((rb & 4) ? cb << 6 : 0);
    } else {
      mask = ((unsigned)px_hi0[it] < (unsigned)d.h && (unsigned)px_wi0[it] < (unsigned)d.w) ? 1u : 0u;
    }
    if (!ok) mask = 0;
    px_mask[it] = mask;
    if (FAST) px_base[it] = (px_base[it] + chunk * 8) * 2;   // byte offset of this lane's chunk at tap (0,0)
  }
  uint32_t w_off[WIT];
#pragma unroll
  for (int it = 0; it < WIT; ++it) {
    const int row = (it * NL + lw) * RPP + frow;
    w_off[it] = row < BN ? (uint32_t)(((n0 + row) * d.kpad + chunk * 8) * 2) : kOobOffset;
  }
  // generic: this lane's chunk has its own (tap, channel); FAST: both are wave-uniform scalars
  int tap = FAST ? 0 : (chunk * 8) / d.cin;
  int kc = FAST ? 0 : chunk * 8 - tap * d.cin;
  if constexpr (SPLITK) {      // this workgroup multiplies K steps [blockIdx.y * steps, + steps): start tap / channel, weight columns
    const int k0 = blockIdx.y * a.steps * BK;
    tap = k0 / d.cin;
    kc = k0 - tap * d.cin;
#pragma unroll
    for (int it = 0; it < WIT; ++it)
      if (w_off[it] != kOobOffset) w_off[it] += (uint32_t)k0 * 2u;
  }

  // One stage = LPS LDS-DMA instructions per thread (PIT pixel pieces, then WIT weight pieces).
  // issue(buf, step, lo, hi) launches pieces [lo, hi) so that the main loop can spread them between
  // its MFMA groups instead of bursting them right after the barrier (an LDS-DMA costs the issuing
  // wave ~60-180 cycles; bursting 8 of them idles the matrix pipe of every wave at once).
  int st_dh = 0, st_dw = 0;
  auto stage_begin = [&]() {
    st_dh = st_dw = 0;
    if (d.ksize == 3) {
      st_dh = (tap * 11) >> 5;  // tap / 3 for tap < 12
      st_dw = tap - 3 * st_dh;
    }
  };
  auto issue = [&](int buf, int step, int lo, int hi) {
    char* const wb = smem + buf * STAGE_B + lw * 1024;
    char* const xb = wb + BNL * ROWB;
    if (FAST) {
      const uint32_t tap_off = (uint32_t)(((st_dh * d.w + st_dw) * d.in_c_total + kc) * 2);
      // (1x1 convs take this path with any cin % 8 == 0 - round 5: a chunk beyond the last channel reads zeros; with one tap a K
      // step cannot straddle taps whatever cin is.  The generic path cost EfficientNet's 16 / 24 / 40-channel expand convs a
      // division per lane and stage.)
      const uint32_t bit = (kc + chunk * 8 < d.cin) ? 1u << tap : 0u;
#pragma unroll
      for (int it = 0; it < PIT; ++it) {
        if (it < lo || it >= hi) continue;
        const uint32_t voff = (px_mask[it] & bit) ? (uint32_t)px_base[it] + tap_off : kOobOffset;
        if (!(a.debug & 1)) lds_dma16(rx, xb + it * (NL * 1024), voff);
      }
#pragma unroll
      for (int it = 0; it < WIT; ++it) {
        if (PIT + it < lo || PIT + it >= hi) continue;
        if (!(a.debug & 2)) lds_dma16s(rw, wb + it * (NL * 1024), w_off[it], (uint32_t)step * (BK * 2u));
      }
    } else {
      const bool tap_ok = tap < ntaps;
      const int tap_off = (st_dh * d.w + st_dw) * d.in_c_total + kc;
#pragma unroll
      for (int it = 0; it < PIT; ++it) {
        if (it < lo || it >= hi) continue;
        const bool ok = tap_ok && ((px_mask[it] >> (tap_ok ? tap : 0)) & 1u);
        const uint32_t voff = ok ? (uint32_t)(px_base[it] + tap_off) * 2u : kOobOffset;
        lds_dma16(rx, xb + it * (NL * 1024), voff);
      }
#pragma unroll
      for (int it = 0; it < WIT; ++it) {
        if (PIT + it < lo || PIT + it >= hi) continue;
        const uint32_t voff = w_off[it] == kOobOffset ? kOobOffset : w_off[it] + (uint32_t)step * (BK * 2u);
        lds_dma16(rw, wb + it * (NL * 1024), voff);
      }
    }
  };
  auto stage_end = [&]() {
    kc += BK;
    if (FAST) {
      if (kc >= d.cin) {
        kc = 0;
        ++tap;
      }
    } else {
      while (kc >= d.cin) {
        kc -= d.cin;
        ++tap;
      }
    }
  };

  if constexpr (NP > 0) {
    if (wave >= NW) {
      // ---- loader wave: stage s+1 goes out right after the barrier that frees its ring slot; the wave then sleeps
      // in s_waitcnt / s_barrier and takes no issue slots from the MFMA waves of its SIMD
#pragma unroll
      for (int p = 0; p < NS - 1; ++p)        // prologue: NS-1 stages in flight
        if (p < a.steps) {
          stage_begin();
          issue(p, p, 0, LPS);
          stage_end();
        }
      int slot = NS - 1;                      // ring slot of the next stage to issue
      for (int s = 0; s < a.steps; ++s) {
        if (NS == 3 && s + 1 < a.steps) wait_vmcnt<LPS>();     // stage s landed; stage s+1 may stay in flight
        else wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();       // stage s landed and visible; every MFMA wave finished stage s-1
        if (s + NS - 1 < a.steps) {
          stage_begin();
          issue(slot, s + NS - 1, 0, LPS);  // into the slot stage s-1 was read from
          stage_end();
        }
        if (++slot == NS) slot = 0;
      }
      return;                               // a finished wave no longer counts in the block's barriers
    }
  }

  static_assert(!MFMA16 || LDS_EPI, "the 16x16x32 path only has the LDS epilogue");
  constexpr int MI16 = MFMA16 ? TN / 16 : 1, NI16 = MFMA16 ? TM / 16 : 1;
  f32x16 acc[MFMA16 ? 1 : MI][MFMA16 ? 1 : NI];
  f32x4 acc16[MI16][NI16];
#pragma unroll
  for (int i = 0; i < (MFMA16 ? 1 : MI); ++i)
#pragma unroll
    for (int j = 0; j < (MFMA16 ? 1 : NI); ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
#pragma unroll
  for (int i = 0; i < MI16; ++i)
#pragma unroll
    for (int j = 0; j < NI16; ++j)
#pragma unroll
      for (int e = 0; e < 4; ++e) acc16[i][j][e] = 0.f;

  const int r32 = lane & 31, khalf = lane >> 5;
  const int steps = a.steps;
  constexpr int KS = BK / 16;

  // prologue: NS-1 stages in flight
  if constexpr (NP == 0) {
#pragma unroll
  for (int p = 0; p < NS - 1; ++p)
    if (p < steps) {
      stage_begin();
      issue(p, p, 0, LPS);
      stage_end();
    }
  }

  auto pix_of = [&](int row) -> long {
    const int pix = m0 + wm * TM + row;
    return pix < a.M ? (long)pix : -1L;
  };

  int buf = 0;
  for (int s = 0; s < steps; ++s) {
    // stage s must have landed: allow the younger stages (up to NS-2 of them) to stay in flight
    if constexpr (NP == 0) {
      const int younger = min(NS - 2, steps - 1 - s);
      if (younger >= 2) wait_vmcnt<2 * LPS>();
      else if (younger == 1) wait_vmcnt<1 * LPS>();
      else wait_vmcnt<0>();
    }
#ifndef YOLO_ABL_NOBARRIER          // (timing-only ablation builds of tools/block_timeline.py)
    __builtin_amdgcn_s_barrier();   // everyone's stage s is in LDS; everyone finished reading stage s-1
#endif
    const bool more = NP == 0 && s + NS - 1 < steps;
    int nb = buf + NS - 1;          // ring slot read in iteration s-1: free again after the barrier
    if (nb >= NS) nb -= NS;
    if (more) stage_begin();
    const char* wbuf = smem + buf * STAGE_B;
    const char* xbuf = wbuf + BNL * ROWB;
    if constexpr (MFMA16) {
      // v_mfma_f32_16x16x32_bf16: lane = (row l&15, k-chunk l>>4); one instruction eats 32 k of a 16x16 tile.
      // Rows 16 apart share their swizzle term, so every fragment address is one per-lane base (two for BK 64:
      // the second K half flips bit 6) plus a compile-time offset that goes into the ds_read immediate.
      const int c16 = lane & 15, q = lane >> 4;
      const int rw0 = wn * TN + c16, rx0 = wm * TM + c16;
      const int sww = (BK == 32) ? swz32(rw0 >> 2) : ((rw0 >> 1) & 7), swx = (BK == 32) ? swz32(rx0 >> 2) : ((rx0 >> 1) & 7);
      const int wo = rw0 * ROWB + ((q ^ sww) << 4), xo = rx0 * ROWB + ((q ^ swx) << 4);
#pragma unroll
      for (int kk = 0; kk < BK / 32; ++kk) {
        if (more) issue(nb, s + NS - 1, kk * LPS / (BK / 32), (kk + 1) * LPS / (BK / 32));
        if (!(a.debug & 4)) {
          const char* const wk = wbuf + (kk ? (wo ^ 64) : wo);
          const char* const xk = xbuf + (kk ? (xo ^ 64) : xo);
          bf16x8 xf[NI16];
#pragma unroll
          for (int j = 0; j < NI16; ++j) xf[j] = YOLO_LDS_FRAG(xk + j * 16 * ROWB);
          if constexpr (NP > 0) {
            // register-lean order (3 waves per SIMD leave 168 registers): weight fragments one ahead of their MFMAs
            bf16x8 wcur = *reinterpret_cast<const bf16x8*>(wk);
#pragma unroll
            for (int i = 0; i < MI16; ++i) {
              const bf16x8 wnext = *reinterpret_cast<const bf16x8*>(wk + (i + 1 < MI16 ? i + 1 : i) * 16 * ROWB);
#pragma unroll
              for (int j = 0; j < NI16; ++j)
                acc16[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wcur, xf[j], acc16[i][j], 0, 0, 0);
              wcur = wnext;
            }
          } else {
            bf16x8 wf[MI16];
#pragma unroll
            for (int i = 0; i < MI16; ++i) wf[i] = YOLO_LDS_FRAG(wk + i * 16 * ROWB);
#pragma unroll
            for (int i = 0; i < MI16; ++i)
#pragma unroll
              for (int j = 0; j < NI16; ++j)
                acc16[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[i], xf[j], acc16[i][j], 0, 0, 0);
          }
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    } else
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      if (more) issue(nb, s + NS - 1, ks * LPS / KS, (ks + 1) * LPS / KS);
      if (!(a.debug & 4)) {
        const int g = ks * 2 + khalf;
        bf16x8 wf[MI], xf[NI];
#pragma unroll
        for (int i = 0; i < MI; ++i) {
          const int R = wn * TN + i * 32 + r32;
          const int sw = (BK == 32) ? swz32(R >> 2) : ((R >> 1) & 7);
          wf[i] = *reinterpret_cast<const bf16x8*>(wbuf + R * ROWB + ((g ^ sw) << 4));
        }
#pragma unroll
        for (int j = 0; j < NI; ++j) {
          const int R = wm * TM + j * 32 + r32;
          const int sw = (BK == 32) ? swz32(R >> 2) : ((R >> 1) & 7);
          xf[j] = *reinterpret_cast<const bf16x8*>(xbuf + R * ROWB + ((g ^ sw) << 4));
        }
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
          for (int j = 0; j < NI; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[i], xf[j], acc[i][j], 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);   // keep each DMA slice in front of its MFMA group
    }
    if (more) stage_end();
    if (++buf == NS) buf = 0;
  }

  if (a.debug & 8) {
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int j = 0; j < NI; ++j) {
#if defined(__HIP_DEVICE_COMPILE__)
        if (!MFMA16) asm volatile("" ::"v"(acc[i][j]));   // keep the accumulators live in the timing-only build path
#endif
      }
#pragma unroll
    for (int i = 0; i < MI16; ++i)
#pragma unroll
      for (int j = 0; j < NI16; ++j) {
#if defined(__HIP_DEVICE_COMPILE__)
        if (MFMA16) asm volatile("" ::"v"(acc16[i][j]));
#endif
      }
    return;
  }
  // ---- epilogue: lane = pixel (col), registers = couts (row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)) ----
  const bool f32_out = d.out_dtype == YOLO_DT_F32;
  if constexpr (DECODE) {
    // Head conv + YOLOLayer decode (yolo_layer.py:57-69,90-96): the tile [BM pixels][na*(5+nc) couts] is staged in
    // LDS as fp32, then each wave walks its pixels; per (pixel, anchor) the lanes take k = lane, lane + 64 of the
    // (5+nc)-run, so the raw logits go to p and the decoded values to io as contiguous 340-byte runs that the
    // next pixel continues.  The head tensor itself is never written.
    __syncthreads();                                 // every wave is done with the last stage: LDS is free
    float* const stg = reinterpret_cast<float*>(smem);
#pragma unroll
    for (int j = 0; j < NI; ++j)
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
          const int c0 = wn * TN + i * 32 + g4 * 8 + khalf * 4;
          f32x4 v = {0.f, 0.f, 0.f, 0.f};
          if (c0 < d.cout_pad) {
            const f32x4 bv = *reinterpret_cast<const f32x4*>(a.bias + c0);
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = apply_act(acc[i][j][g4 * 4 + e] + bv[e], d.act);
          }
          *reinterpret_cast<f32x4*>(stg + (wm * TM + j * 32 + r32) * DP + c0) = v;
        }
    __syncthreads();
    const HeadLanes<BN> hl = head_lanes<BN>(a.hd, lane);
    head_decode_rows<BM / NW, BN>(a, hl, stg, DP, m0, wave, lane);      // head_epilogue.h
  } else if constexpr (LDS_EPI && MFMA16) {
    __syncthreads();
    if constexpr (SPLITK) {
      // partial tile -> ws[split]; lane = pixel (lane & 15), 4 consecutive couts 4 * (lane >> 4) + e of 16x16 tile (i, j)
      const int c16 = lane & 15, q4 = lane >> 4;
      const long slice = (long)a.M * d.cout;
      auto at = [&](int i, int j) -> long {
        const int pix = m0 + wm * TM + j * 16 + c16, c = n0 + wn * TN + i * 16 + q4 * 4;
        return (pix < a.M && c < d.cout) ? (long)pix * d.cout + c : -1L;
      };
      // The exchange goes through agent-scope (write-through / L2-bypassing) accesses: the eight XCD L2s are not coherent
      // with each other for plain accesses, and a __threadfence() here writes back and invalidates the whole L2 of
      // the XCD for every workgroup (measured: 3 x slower than not splitting at all).
      float* const mine = a.ws + blockIdx.y * slice;
#pragma unroll
      for (int i = 0; i < MI16; ++i)
#pragma unroll
        for (int j = 0; j < NI16; ++j) {
          const long o = at(i, j);
          if (o >= 0)
#pragma unroll
            for (int e = 0; e < 4; ++e) __hip_atomic_store(mine + o + e, acc16[i][j][e], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
      wait_vmcnt<0>();                       // this wave's partial stores have been acknowledged
      __syncthreads();                       // ... and every wave's
      __shared__ int s_last;
      if (tid == 0) s_last = __hip_atomic_fetch_add(a.cnt + blockIdx.x, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == a.splits - 1;
      __syncthreads();
      if (!s_last) return;
#pragma unroll
      for (int i = 0; i < MI16; ++i)
#pragma unroll
        for (int j = 0; j < NI16; ++j) {
          const long o = at(i, j);
          f32x4 sum = {0.f, 0.f, 0.f, 0.f};
          if (o >= 0)
            for (int z = 0; z < a.splits; ++z)        // fixed order: the result does not depend on who arrives last
#pragma unroll
              for (int e = 0; e < 4; ++e) sum[e] += __hip_atomic_load(a.ws + z * slice + o + e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          acc16[i][j] = sum;
        }
      if (tid == 0) __hip_atomic_store(a.cnt + blockIdx.x, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ready for the next launch
    }
    epilogue_lds16<MI, NI16, TM>(a, acc16, smem + wave * (TM * kEpiPitch), lane, n0 + wn * TN, pix_of);
  } else if constexpr (LDS_EPI) {
    __syncthreads();                                 // every wave is done with the last stage: LDS is free
    epilogue_lds<MI, NI, TM>(a, acc, smem + wave * (TM * kEpiPitch), lane, n0 + wn * TN, pix_of);
  } else {
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
}

template <int BM, int BN, int WAVES_M, int WAVES_N, int BK, int NS, bool FAST, bool LDS_EPI, bool MFMA16 = false, bool DECODE = false,
          int NP = 0, bool SPLITK = false>
int launch_cfg(const ConvArgs& a, hipStream_t s) {
  const int m_tiles = (a.M + BM - 1) / BM;
  ConvArgs b = a;
  b.n_tiles = (a.d.cout + BN - 1) / BN;
  b.steps = (a.d.ksize * a.d.ksize * a.d.cin + BK - 1) / BK;   // kpad >= steps*BK: the K tail is zero-padded
  const long grid = (long)m_tiles * b.n_tiles;
  if (grid > 0x7fffffffL) return yolo_set_error(YOLO_E_UNSUPPORTED, "conv grid too large");
  if (pick_only("igemm<%dx%d,%dx%d waves%s,BK%d,%d stages%s%s%s%s> grid %ld", BM, BN, WAVES_M, WAVES_N, NP ? "+4 loaders" : "", BK, NS,
                FAST ? "" : ",generic", MFMA16 ? ",16x16x32" : ",32x32x16", DECODE ? ",decode" : "", SPLITK ? ",splitK" : "", grid))
    return 0;
  unsigned gy = 1;
  if constexpr (SPLITK) {
    YOLO_REQUIRE(a.splits >= 2 && b.steps % a.splits == 0 && a.ws && a.cnt, "split-K: %d K steps do not split %d ways", b.steps, a.splits);
    b.steps /= a.splits;
    gy = (unsigned)a.splits;
  }
  hipLaunchKernelGGL((conv_igemm_bf16_kernel<BM, BN, WAVES_M, WAVES_N, BK, NS, FAST, LDS_EPI, MFMA16, DECODE, NP, SPLITK>),
                     dim3((unsigned)grid, gy), dim3(64 * (WAVES_M * WAVES_N + NP)), 0, s, b);
  return yolo_check_launch("yolo_conv2d_fwd");
}

}  // namespace

namespace yolo_conv {
static thread_local char* g_pick = nullptr;
static thread_local int g_pick_len = 0;
char* pick_buffer() { return g_pick; }
bool pick_only(const char* fmt, ...) {
  if (!g_pick) return false;
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_pick, g_pick_len, fmt, ap);
  va_end(ap);
  return true;
}
}  // namespace yolo_conv

int& yolo_conv_mb_debug();   // conv_mbconv.hip: YOLO_MBCONV_DEBUG / yolo_set_tuning(4, .)

namespace {
int conv_variant_override = -1;
int conv_debug_flags = 0;   // tuning hook (YOLO_CONV_VARIANT), see yolo_conv2d_launch
int conv_pp_mask = 0;       // YOLO_CONV_PP / yolo_set_tuning(2, .): kernel-family selection for tests and A/B runs - 8 no halo kernel,
                            // 16 the 20x20-tile kernels on every layer they can compute, 64 ... on none, 1024 no streaming 1x1, 2048 ... on every layer it can compute

// split-K request of the current yolo_conv2d_splitk_fwd call (consumed by conv2d_launch_ex)
struct SplitK {
  int splits = 1;
  float* ws = nullptr;
  int* cnt = nullptr;
};
thread_local SplitK g_splitk;
thread_local int g_launch_cus = 256;   // compute units the coming launches may use (a CU-masked stream: yolo_set_launch_cus)

// The two shapes split-K serves (few pixels, long K, so few tiles that most CUs idle): tiles and K steps of the tile
// configuration the dispatch below picks, or 0 tiles when the layer is not one of them.
void splitk_shape(const YoloConvDesc& d, bool has_res_or_aux_views_ok, long* tiles, int* steps) {
  *tiles = 0;
  *steps = 0;
  const long M = (long)d.n * d.ho * d.wo;
  const bool epi = d.out_dtype == YOLO_DT_BF16 && d.cout % 32 == 0 && d.out_c_offset % 8 == 0 && d.out_c_total % 8 == 0 && has_res_or_aux_views_ok;
  if (!epi || d.cin % 64 != 0 || d.upsample2x || (long)d.h * d.w >= 80 * 80) return;
  if (d.cout == 64 && (M + 255) / 256 < 128) {                       // 64x64 tiles, BK 64
    *tiles = (M + 63) / 64;
    *steps = d.ksize * d.ksize * d.cin / 64;
  } else if (d.ksize == 3 && d.stride == 1 && d.cout % 256 == 0 && ((M + 255) / 256) * (d.cout / 256) < 160 &&
             ((M + 127) / 128) * (d.cout / 256) <= 256) {            // 128x256 loader-wave tiles, BK 64
    *tiles = ((M + 127) / 128) * (d.cout / 256);
    *steps = 9 * d.cin / 64;
  }
}

}  // namespace

static int conv2d_launch_ex(const void* x, const void* w, const float* bias, const void* res, void* y, void* y_aux,
                            const YoloConvDesc* dp, const HeadDecodeArgs* hd, hipStream_t s);

static bool read_conv_env() {
  static const bool done = [] {
    if (const char* e = getenv("YOLO_CONV_VARIANT")) conv_variant_override = atoi(e);
    if (const char* e = getenv("YOLO_CONV_DEBUG")) conv_debug_flags = atoi(e);
    if (const char* e = getenv("YOLO_CONV_PP")) conv_pp_mask = atoi(e);
    return true;
  }();
  return done;
}

namespace yolo_conv {
int launch_cus() { return g_launch_cus; }
}  // namespace yolo_conv

// The tile rules below size grids against the compute units a launch can use: 256, or what the CU mask of the stream leaves
// (engine.StreamedPlan gives each sub-batch pipeline half of every XCD).  Thread-local; returns the previous value.
extern "C" int yolo_set_launch_cus(int n_cu) {
  YOLO_REQUIRE(n_cu >= 8 && n_cu <= 1024, "set_launch_cus: %d out of range", n_cu);
  const int old = g_launch_cus;
  g_launch_cus = n_cu;
  return old;
}

extern "C" int yolo_set_tuning(int knob, int value) {
  read_conv_env();
  int* const slot = knob == 0 ? &conv_variant_override : knob == 1 ? &conv_debug_flags : knob == 2 ? &conv_pp_mask : knob == 3 ? &resunit_debug()
                    : knob == 4 ? &yolo_conv_mb_debug() : nullptr;
  YOLO_REQUIRE(slot, "set_tuning: unknown knob %d", knob);
  const int old = *slot;
  *slot = value;
  return old;
}

int yolo_conv2d_launch(const void* x, const void* w, const float* bias, const void* res, void* y, void* y_aux,
                       const YoloConvDesc* dp, hipStream_t s) {
  return conv2d_launch_ex(x, w, bias, res, y, y_aux, dp, nullptr, s);
}

static int conv2d_launch_ex(const void* x, const void* w, const float* bias, const void* res, void* y, void* y_aux,
                            const YoloConvDesc* dp, const HeadDecodeArgs* hd, hipStream_t s) {
  YOLO_REQUIRE(x && w && bias && (y || hd) && dp, "conv: null pointer");
  read_conv_env();
  const YoloConvDesc& d = *dp;
  YOLO_REQUIRE(d.ksize == 1 || d.ksize == 3, "conv: ksize %d unsupported (1 or 3)", d.ksize);
  YOLO_REQUIRE(d.stride == 1 || d.stride == 2, "conv: stride %d unsupported", d.stride);
  YOLO_REQUIRE(d.cin > 0 && d.cin % 8 == 0, "conv: cin %d must be a positive multiple of 8", d.cin);
  YOLO_REQUIRE(d.in_c_offset % 8 == 0 && d.in_c_total % 8 == 0 && d.in_c_offset + d.cin <= d.in_c_total,
               "conv: bad input view (cin %d, offset %d, total %d)", d.cin, d.in_c_offset, d.in_c_total);
  YOLO_REQUIRE(d.out_c_offset % 4 == 0 && d.out_c_total % 4 == 0 && d.out_c_offset + d.cout <= d.out_c_total,
               "conv: bad output view (cout %d, offset %d, total %d)", d.cout, d.out_c_offset, d.out_c_total);
  YOLO_REQUIRE(d.kpad % 64 == 0 && d.kpad >= d.ksize * d.ksize * d.cin, "conv: kpad %d must be a multiple of 64", d.kpad);
  YOLO_REQUIRE(d.cout_pad % 128 == 0 && d.cout_pad >= d.cout, "conv: cout_pad %d", d.cout_pad);
  const int ho_std = (d.h + 2 * d.pad - d.ksize) / d.stride + 1, wo_std = (d.w + 2 * d.pad - d.ksize) / d.stride + 1;
  // one more row / column than the symmetric-pad size = one more zero below / right of the image (TensorFlow "same" padding)
  YOLO_REQUIRE((d.ho == ho_std || (d.ho == ho_std + 1 && (d.ho - 1) * d.stride - d.pad < d.h)) &&
                   (d.wo == wo_std || (d.wo == wo_std + 1 && (d.wo - 1) * d.stride - d.pad < d.w)),
               "conv: output size %dx%d inconsistent with input %dx%d k%d s%d p%d", d.ho, d.wo, d.h, d.w, d.ksize,
               d.stride, d.pad);
  const bool std_out = d.ho == ho_std && d.wo == wo_std;
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
  a.steps = 0;
  a.x_bytes = (uint32_t)x_bytes;
  a.w_bytes = (uint32_t)w_bytes;
  a.debug = conv_debug_flags;
  a.splits = g_splitk.splits;
  a.ws = g_splitk.ws;
  a.cnt = g_splitk.cnt;
  YOLO_SET_STAMPS(a);
  if (hd) {   // head conv with the YOLOLayer decode as its epilogue: one 64-pixel x 256-cout tile per block
    a.hd = *hd;
    // 8 waves (2 blocks per CU -> 4 per SIMD) when cin allows 64-deep stages; bit 8192 selects the 4-wave forms
    if (d.cin % 64 == 0 && !(conv_debug_flags & 8192)) return launch_cfg<64, 256, 1, 8, 64, 2, true, false, false, true>(a, s);
    return d.cin % 32 == 0 ? launch_cfg<64, 256, 1, 4, 32, 2, true, false, false, true>(a, s)
                           : launch_cfg<64, 256, 1, 4, 32, 2, false, false, false, true>(a, s);
  }
  const int variant = conv_variant_override >= 0 ? conv_variant_override : 0;
  const bool one_tap = d.ksize == 1 && d.cin % 8 == 0 && !(conv_debug_flags & 268435456);      // (bit 268435456: the generic path for them, A/B)
  const bool fast64 = d.cin % 64 == 0, fast32 = d.cin % 32 == 0 || one_tap;
  const bool epi = d.out_dtype == YOLO_DT_BF16 && d.cout % 32 == 0 && d.out_c_offset % 8 == 0 && d.out_c_total % 8 == 0 &&
                   (!res || (d.res_c_offset % 8 == 0 && d.res_c_total % 8 == 0)) &&
                   (!y_aux || (d.aux_c_offset % 8 == 0 && d.aux_c_total % 8 == 0)) && !(conv_debug_flags & 16);
  // 3x3 layers whose maps 20x20 tiles cover and fill the chip with (conv3x3_t20.hip: stride 1 and, round 3, stride 2).
  // YOLO_CONV_PP bit 16: every layer the kernels can compute, bit 64: never.
  if (epi && std_out && conv_variant_override < 0 && !(conv_pp_mask & 64) && a.splits <= 1) {
    const int rc = launch_t20_3x3(a, conv_pp_mask & 16 ? 1 : 0, s);
    if (rc != 1) return rc;
  }
  // short-K 1x1 layers on the large maps: weight-stationary streaming kernel (conv1x1_stream.hip).  YOLO_CONV_PP bit 1024: never,
  // bit 2048: every layer it can compute.
  if (epi && std_out && d.ksize == 1 && conv_variant_override < 0 && !(conv_pp_mask & 1024) && a.splits <= 1) {
    const int rc = launch_stream1x1(a, conv_pp_mask & 2048 ? 1 : 0, s);
    if (rc != 1) return rc;
  }
  if (epi && std_out && !(conv_debug_flags & 32) && conv_variant_override < 0 && !(conv_pp_mask & 8)) {   // large 3x3/s1 maps: halo-staged kernel
    a.n_tiles = 0;
    const int rc = launch_halo3x3(a, s);
    if (rc != 1) return rc;
  }
#define YOLO_CFG(BM, BN, WM, WN, BK, NS, FASTV)                                                           \
  (epi ? launch_cfg<BM, BN, WM, WN, BK, NS, FASTV, true>(a, s) : launch_cfg<BM, BN, WM, WN, BK, NS, FASTV, false>(a, s))
  if (d.cout <= 32) return fast32 ? YOLO_CFG(256, 32, 4, 1, 32, 2, true) : YOLO_CFG(256, 32, 4, 1, 32, 2, false);
  // few pixels, few output channels, long K (MobileNetV2-tiny head: 3x3 1280 -> 64 on 13x13): 256-row tiles leave most
  // CUs idle, 64x64 tiles quadruple the workgroup count
  // ... and with one small workgroup per CU nothing hides the LDS-DMA latency of a two-stage ring: four stages
  if (d.cout == 64 && fast64 && epi && (M + 255) / 256 < g_launch_cus / 2 && conv_variant_override < 0 && !(conv_debug_flags & 2048)) {
    if (a.splits > 1) return launch_cfg<64, 64, 2, 2, 64, 4, true, true, true, false, 0, true>(a, s);
    if (conv_debug_flags & 4194304) return launch_cfg<64, 64, 2, 2, 64, 2, true, true, true>(a, s);
    return launch_cfg<64, 64, 2, 2, 64, 4, true, true, true>(a, s);
  }
  if (d.cout <= 64) return fast32 ? YOLO_CFG(256, 64, 4, 1, 32, 2, true) : YOLO_CFG(256, 64, 4, 1, 32, 2, false);
  if (!fast64) return fast32 ? YOLO_CFG(128, 128, 2, 2, 32, 3, true) : YOLO_CFG(128, 128, 2, 2, 32, 3, false);
  // 256x256 (8 waves, one block per CU) halves the operand traffic per FLOP but needs enough tiles to fill
  // the 256 CUs; otherwise 128x128 (4 waves, two blocks per CU).  Measured on MI355X, see DESIGN.md.
  int pick = variant;
  if (conv_variant_override < 0) {
    const int n_cu = g_launch_cus;
    const long tiles256 = ((M + 255) / 256) * (d.cout / 256);
    pick = (d.cout % 256 == 0 && tiles256 >= n_cu * 5 / 8) ? 5 : 0;
    // ... unless 128x128 tiles (two workgroups per CU) fill their last round much better: on 128 CUs the stride-2 128 -> 256 layer
    // is 400 tiles of 256x256 = 3.1 rounds (0.78) against 6.25 rounds of 128x128 (0.89): -8 %
    // (measured on a 128-CU share only; on the whole chip the same round counts at 32 images favour the 256x256 tiles by 5 %)
    if (pick == 5 && d.ksize == 3 && n_cu <= 128) {
      const long t128 = ((M + 127) / 128) * (d.cout / 128);
      const double e256 = (double)tiles256 / (double)(((tiles256 + n_cu - 1) / n_cu) * n_cu);
      const double e128 = (double)t128 / (double)(((t128 + 2 * n_cu - 1) / (2 * n_cu)) * 2 * n_cu);
      if (e128 > 1.1 * e256) pick = 0;
    }
    // short-K 1x1 layers on big maps are latency/HBM-bound: 256x128 tiles with 32-deep stages keep
    // 16 waves per CU resident (two 8-wave blocks), which hides the per-tile prologue/epilogue
    if (d.ksize == 1 && d.cin <= 512 && M >= 40000) pick = 9;
    // 128x256 tiles (8 waves) when they cover the layer in one round of the 256 CUs: 25 % less operand traffic per
    // FLOP than 128x128 (-7 % on the 20x20 3x3 and the 40x40 1x1 layers)
    if (pick == 0 && d.cout % 256 == 0 && ((M + 127) / 128) * (d.cout / 256) <= n_cu && !(conv_debug_flags & 128)) pick = 12;
    // tiny grids (1x1 layers on the 20x20 maps): 64x64 tiles quadruple the block count so the chip fills
    if (d.ksize == 1 && ((M + 127) / 128) * ((d.cout + 127) / 128) < n_cu) pick = 11;
  }
  // 16x16x32 MFMA mainloop (same LDS traffic and cycles per FLOP as 32x32x16; the chip holds a higher clock on
  // it: +2..3 % measured on every shape).  YOLO_CONV_DEBUG bit 2048 falls back to 32x32x16.
  if (epi && !(conv_debug_flags & 2048)) {
    switch (pick) {
      case 5:
        // 16 waves (64x64 each) instead of 8 (64x128): four waves per SIMD hide the LDS-DMA issue stalls of one another
        // (-7..8 % per layer; loader waves would spill here: 128 accumulators + 3 waves per SIMD).  Bit 512: old form.
        if (conv_debug_flags & 512) return launch_cfg<256, 256, 4, 2, 64, 2, true, true, true>(a, s);
        return launch_cfg<256, 256, 4, 4, 64, 2, true, true, true>(a, s);
      case 3: return launch_cfg<256, 128, 4, 2, 64, 2, true, true, true>(a, s);
      case 13: return launch_cfg<256, 256, 4, 4, 64, 2, true, true, true>(a, s);
      case 12:
        // 3x3 layers: four extra loader waves issue all LDS-DMA, the eight MFMA waves only read LDS and multiply
        // (-7 % on the 20x20 layers; neutral on 1x1, so those keep the symmetric form).  Bit 256 disables it.
        // ... and a THREE-stage ring: a K step of this tile is ~0.9 us of MFMA, less than an HBM round trip, and in the
        // model the weights of these layers (9.4 MB each) come from HBM: 0.094 -> 0.075 ms per layer inside the network
        // (nothing in a back-to-back micro-benchmark, where they sit in the Infinity Cache).  Bit 16384: two stages.
        if (d.ksize == 3 && a.splits > 1) return launch_cfg<128, 256, 2, 4, 64, 3, true, true, true, false, 4, true>(a, s);
        if (d.ksize == 3 && !(conv_debug_flags & (256 | 16384))) return launch_cfg<128, 256, 2, 4, 64, 3, true, true, true, false, 4>(a, s);
        if (d.ksize == 3 && !(conv_debug_flags & 256)) return launch_cfg<128, 256, 2, 4, 64, 2, true, true, true, false, 4>(a, s);
        if (d.ksize == 1) return launch_cfg<128, 256, 2, 8, 64, 3, true, true, true>(a, s);   // 16 waves (-7..12 % on the 40x40 1x1 layers), 3 stages
        return launch_cfg<128, 256, 2, 4, 64, 2, true, true, true>(a, s);
      case 9: return launch_cfg<256, 128, 4, 2, 32, 2, true, true, true>(a, s);   // (a third stage costs a resident block: slower)
      case 11:
        // inside the network (weights from HBM) an 8-wave 128x128 tile with a three-stage ring edges out the 64x64
        // tiles that win a back-to-back micro-benchmark; layers with fewer than 64 such tiles keep the small ones
        if (((M + 127) / 128) * ((d.cout + 127) / 128) >= 64 && !(conv_debug_flags & 32768))
          return launch_cfg<128, 128, 2, 4, 64, 3, true, true, true>(a, s);
        return launch_cfg<64, 64, 2, 2, 64, 2, true, true, true>(a, s);
      default:
        if (conv_debug_flags & 8192) return launch_cfg<128, 128, 2, 2, 64, 2, true, true, true>(a, s);
        return launch_cfg<128, 128, 2, 4, 64, 2, true, true, true>(a, s);   // 8 waves of 64x32: -13 % on the stride-2 64->128 layer
    }
  }
  switch (pick) {
    case 3: return YOLO_CFG(256, 128, 4, 2, 64, 2, true);
    case 7: return YOLO_CFG(128, 128, 2, 2, 32, 2, true);
    case 8: return YOLO_CFG(128, 64, 2, 2, 64, 2, true);
    case 10: return YOLO_CFG(64, 128, 2, 2, 64, 2, true);
    case 11: return YOLO_CFG(64, 64, 2, 2, 64, 2, true);
    case 9: return YOLO_CFG(256, 128, 4, 2, 32, 2, true);
    case 5: return YOLO_CFG(256, 256, 4, 2, 64, 2, true);
    default: return YOLO_CFG(128, 128, 2, 2, 64, 2, true);
  }
#undef YOLO_CFG
}

// First layer straight from the caller's float32 NCHW batch (fuses yolo_pack_input_nchw_f32 + conv), optionally with
// the MaxPool2d(2, 2) that follows it in YOLOv3-tiny.
static int conv1_nchw(const float* x_nchw, int cin_real, const void* w_packed, const float* bias, void* y, const YoloConvDesc* dp,
                      bool pool, yolo_stream_t s) {
  YOLO_REQUIRE(x_nchw && w_packed && bias && y && dp, "conv1: null pointer");
  const YoloConvDesc& d = *dp;
  YOLO_REQUIRE(cin_real >= 1 && cin_real <= 8 && d.cin == 8, "conv1: 1..8 input channels (packed K uses cin = 8)");
  YOLO_REQUIRE(d.out_c_offset % 8 == 0 && d.out_c_total % 8 == 0 && d.out_c_offset + d.cout <= d.out_c_total, "conv1: bad output view");
  YOLO_REQUIRE((d.stride == 1 && d.ho == d.h && d.wo == d.w) ||
                   (d.stride == 2 && !pool && d.ho == (d.h - 1) / 2 + 1 && d.wo == (d.w - 1) / 2 + 1),
               "conv1: 3x3 / pad 1 at stride 1, or at stride 2 (cout 32, no pool) only");
  ConvArgs a;
  a.x = nullptr;
  a.w = (const bf16_t*)w_packed;
  a.bias = bias;
  a.res = nullptr;
  a.y = y;
  a.aux = nullptr;
  a.d = d;
  a.M = d.n * d.ho * d.wo;
  a.n_tiles = 1;
  a.steps = 0;
  a.x_bytes = 0;
  a.w_bytes = 0;
  a.debug = 0;
  a.splits = 1;
  a.ws = nullptr;
  a.cnt = nullptr;
  YOLO_SET_STAMPS(a);
  const int rc = d.stride == 2 ? launch_conv1_s2_nchw(a, x_nchw, cin_real, (hipStream_t)s)
                               : launch_conv1_nchw(a, x_nchw, cin_real, pool, (hipStream_t)s);
  if (rc == 1) return yolo_set_error(YOLO_E_UNSUPPORTED, "conv1: shape not covered (3x3 s1 with cout 16 or 32, 3x3 s2 with cout 32; bf16 out)");
  return rc;
}

extern "C" int yolo_conv1_nchw_f32_fwd(const float* x_nchw, int cin_real, const void* w_packed, const float* bias, void* y,
                                       const YoloConvDesc* dp, yolo_stream_t s) {
  return conv1_nchw(x_nchw, cin_real, w_packed, bias, y, dp, false, s);
}

extern "C" int yolo_conv1_pool_nchw_f32_fwd(const float* x_nchw, int cin_real, const void* w_packed, const float* bias,
                                            void* y_pooled, const YoloConvDesc* dp, yolo_stream_t s) {
  return conv1_nchw(x_nchw, cin_real, w_packed, bias, y_pooled, dp, true, s);
}

// Which kernel instance and grid yolo_conv2d_fwd would launch for this layer (no launch, no GPU): the regression guard of the
// tile rules (tests/test_host_cpu.py pins the BASELINE shapes).
extern "C" int yolo_conv2d_pick(const YoloConvDesc* d, int has_residual, int has_preadd, char* out, int out_len) {
  YOLO_REQUIRE(d && out && out_len > 0, "conv2d_pick: bad arguments");
  out[0] = 0;
  yolo_conv::g_pick = out;
  yolo_conv::g_pick_len = out_len;
  static const char dummy[16] = {0};
  const int rc = yolo_conv2d_launch(dummy, dummy, (const float*)dummy, has_residual ? dummy : nullptr, (void*)dummy,
                                    has_preadd ? (void*)dummy : nullptr, d, nullptr);
  yolo_conv::g_pick = nullptr;
  return rc;
}

extern "C" int yolo_conv2d_fwd(const void* x, const void* w_packed, const float* bias, const void* residual, void* y,
                               void* y_preadd, const YoloConvDesc* d, yolo_stream_t s) {
  return yolo_conv2d_launch(x, w_packed, bias, residual, y, y_preadd, d, (hipStream_t)s);
}

// Split-K form of yolo_conv2d_fwd for layers with few pixels and a long K (see splitk_shape).
extern "C" int yolo_conv2d_splitk_plan(const YoloConvDesc* dp, int has_residual, int has_preadd, int* splits, size_t* ws_bytes,
                                       int* n_counters) {
  YOLO_REQUIRE(dp && splits && ws_bytes && n_counters, "splitk_plan: null pointer");
  const YoloConvDesc& d = *dp;
  *splits = 1;
  *ws_bytes = 0;
  *n_counters = 0;
  const bool views_ok = (!has_residual || (d.res_c_offset % 8 == 0 && d.res_c_total % 8 == 0)) &&
                        (!has_preadd || (d.aux_c_offset % 8 == 0 && d.aux_c_total % 8 == 0));
  long tiles;
  int steps;
  splitk_shape(d, views_ok, &tiles, &steps);
  if (tiles == 0 || tiles >= 200) return 0;
  int best = 1;
  for (int sp = 2; sp <= 8; ++sp)
    if (steps % sp == 0 && steps / sp >= 6 && tiles * sp <= 256) best = sp;   // one workgroup per CU: more would queue
  if (best > 1) {
    *splits = best;
    *ws_bytes = (size_t)best * d.n * d.ho * d.wo * d.cout * sizeof(float);
    *n_counters = (int)tiles;
  }
  return 0;
}

extern "C" int yolo_conv2d_splitk_fwd(const void* x, const void* w_packed, const float* bias, const void* residual, void* y,
                                      void* y_preadd, const YoloConvDesc* d, int splits, void* workspace, size_t ws_bytes,
                                      int32_t* counters, yolo_stream_t s) {
  YOLO_REQUIRE(d && splits >= 2 && workspace && counters, "splitk: bad arguments");
  int want;
  size_t need;
  int ncnt;
  const int rc0 = yolo_conv2d_splitk_plan(d, residual != nullptr, y_preadd != nullptr, &want, &need, &ncnt);
  if (rc0) return rc0;
  YOLO_REQUIRE(want >= 2, "splitk: this layer does not take a split-K launch");
  YOLO_REQUIRE((size_t)splits * d->n * d->ho * d->wo * d->cout * sizeof(float) <= ws_bytes, "splitk: workspace too small");
  g_splitk.splits = splits;
  g_splitk.ws = (float*)workspace;
  g_splitk.cnt = counters;
  const int rc = yolo_conv2d_launch(x, w_packed, bias, residual, y, y_preadd, d, (hipStream_t)s);
  g_splitk = SplitK();
  return rc;
}

// Head conv + decode in one launch (see the DECODE epilogue of conv_igemm_bf16_kernel).
extern "C" int yolo_head_decode_supported(int cout, int na, int nc) {
  return na >= 1 && na <= 4 && nc >= 1 && cout == na * (5 + nc) && cout <= 256 && 5 + nc <= 128;
}

extern "C" int yolo_head_decode_fwd(const void* x, const void* w_packed, const float* bias, const YoloConvDesc* dp,
                                    const float* anchors_px, int na, int nc, float stride_px, float* io, int io_rows_total,
                                    int io_row_offset, float* p, yolo_stream_t s) {
  YOLO_REQUIRE(dp && anchors_px && io, "head_decode: null pointer");
  const YoloConvDesc& d = *dp;
  YOLO_REQUIRE(yolo_head_decode_supported(d.cout, na, nc), "head_decode: cout %d != na*(5+nc) = %d*%d or out of range", d.cout, na,
               5 + nc);
  YOLO_REQUIRE(d.stride == 1 && !d.upsample2x, "head_decode: stride-1 head convs only");
  YOLO_REQUIRE(io_row_offset >= 0 && io_row_offset + na * d.ho * d.wo <= io_rows_total, "head_decode: rows out of range");
  YOLO_REQUIRE(stride_px > 0.f, "head_decode: bad stride");
  HeadDecodeArgs h;
  h.io = io;
  h.p = p;
  h.na = na;
  h.no = nc + 5;
  h.io_rows_total = io_rows_total;
  h.io_row_offset = io_row_offset;
  h.stride = stride_px;
  for (int i = 0; i < 4; ++i) {
    h.anchor_w[i] = i < na ? anchors_px[2 * i] / stride_px : 0.f;
    h.anchor_h[i] = i < na ? anchors_px[2 * i + 1] / stride_px : 0.f;
  }
  h.rec = nullptr, h.row_keys = nullptr, h.conf_thres = 0.f, h.min_wh = 0.f;
  YoloConvDesc dd = d;              // the output view is unused: give the shared checks a consistent one
  dd.out_dtype = YOLO_DT_F32;
  dd.out_c_total = (d.cout + 3) & ~3;
  dd.out_c_offset = 0;
  return conv2d_launch_ex(x, w_packed, bias, nullptr, nullptr, nullptr, &dd, &h, (hipStream_t)s);
}

// Which kernel instance and grid a head op would launch (no launch, no GPU): yolo_conv2d_pick for yolo_head_decode_fwd (filter == 0,
// with io) / yolo_head_decode_filter_fwd (filter != 0: the compact NMS form without p).
extern "C" int yolo_head_decode_pick(const YoloConvDesc* dp, int na, int nc, int filter, char* out, int out_len) {
  YOLO_REQUIRE(dp && out && out_len > 0, "head_decode_pick: bad arguments");
  YOLO_REQUIRE(yolo_head_decode_supported(dp->cout, na, nc), "head_decode_pick: cout %d != na*(5+nc)", dp->cout);
  out[0] = 0;
  static float dummy[16] = {0};
  HeadDecodeArgs h;
  h.io = filter ? nullptr : dummy;
  h.p = nullptr;
  h.na = na;
  h.no = nc + 5;
  h.io_rows_total = na * dp->ho * dp->wo;
  h.io_row_offset = 0;
  h.stride = 1.f;
  for (int i = 0; i < 4; ++i) h.anchor_w[i] = h.anchor_h[i] = 1.f;
  h.rec = filter ? dummy : nullptr;
  h.row_keys = filter ? (unsigned long long*)dummy : nullptr;
  h.conf_thres = 0.f, h.min_wh = 0.f;
  YoloConvDesc dd = *dp;
  dd.out_dtype = YOLO_DT_F32;
  dd.out_c_total = (dp->cout + 3) & ~3;
  dd.out_c_offset = 0;
  yolo_conv::g_pick = out;
  yolo_conv::g_pick_len = out_len;
  const int rc = conv2d_launch_ex(dummy, dummy, dummy, nullptr, nullptr, nullptr, &dd, &h, nullptr);
  yolo_conv::g_pick = nullptr;
  return rc;
}

// Head conv + decode + the NMS row filter in one launch: io is never written (include/yolo_hip.h, the compact NMS form).
extern "C" int yolo_head_decode_filter_fwd(const void* x, const void* w_packed, const float* bias, const YoloConvDesc* dp,
                                           const float* anchors_px, int na, int nc, float stride_px, int io_rows_total,
                                           int io_row_offset, float conf_thres, float min_wh, void* workspace,
                                           size_t workspace_bytes, float* p, yolo_stream_t s) {
  YOLO_REQUIRE(dp && anchors_px && workspace, "head_decode_filter: null pointer");
  const YoloConvDesc& d = *dp;
  YOLO_REQUIRE(yolo_head_decode_supported(d.cout, na, nc), "head_decode_filter: cout %d != na*(5+nc) = %d*%d or out of range", d.cout,
               na, 5 + nc);
  YOLO_REQUIRE(d.stride == 1 && !d.upsample2x, "head_decode_filter: stride-1 head convs only");
  YOLO_REQUIRE(io_row_offset >= 0 && io_row_offset + na * d.ho * d.wo <= io_rows_total, "head_decode_filter: rows out of range");
  YOLO_REQUIRE(io_rows_total < (1 << 20), "head_decode_filter: rows %d >= 2^20 unsupported", io_rows_total);
  YOLO_REQUIRE(stride_px > 0.f, "head_decode_filter: bad stride");
  if (workspace_bytes < yolo_nms_compact_workspace_bytes(d.n, io_rows_total, nc))
    return yolo_set_error(YOLO_E_WORKSPACE, "head_decode_filter: workspace %zu < %zu bytes", workspace_bytes,
                          yolo_nms_compact_workspace_bytes(d.n, io_rows_total, nc));
  YOLO_REQUIRE((size_t)d.n * io_rows_total * yolo_nms::kRecFloats * 4 < kOobOffset, "head_decode_filter: batch too large for 32-bit record offsets");
  const yolo_nms::Workspace w = yolo_nms::carve(workspace, d.n, io_rows_total, nc, true);
  HeadDecodeArgs h;
  h.io = nullptr;
  h.p = p;
  h.na = na;
  h.no = nc + 5;
  h.io_rows_total = io_rows_total;
  h.io_row_offset = io_row_offset;
  h.stride = stride_px;
  for (int i = 0; i < 4; ++i) {
    h.anchor_w[i] = i < na ? anchors_px[2 * i] / stride_px : 0.f;
    h.anchor_h[i] = i < na ? anchors_px[2 * i + 1] / stride_px : 0.f;
  }
  h.rec = w.rec;
  h.row_keys = w.row_keys;
  h.conf_thres = conf_thres;
  h.min_wh = min_wh;
  YoloConvDesc dd = d;
  dd.out_dtype = YOLO_DT_F32;
  dd.out_c_total = (d.cout + 3) & ~3;
  dd.out_c_offset = 0;
  return conv2d_launch_ex(x, w_packed, bias, nullptr, nullptr, nullptr, &dd, &h, (hipStream_t)s);
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
