// Ping-pong implicit-GEMM convolution for gfx950 (MI355X): same contract, packing and numerics as conv_igemm.hip
// (bf16 NHWC in, fp32 accumulate on v_mfma_f32_16x16x32_bf16, fused LDS-staged epilogue), another main loop.
//
// conv_igemm's loop is "wait, barrier, every wave reads its fragments, every wave multiplies": after each barrier all
// waves of a SIMD want LDS data at once and the matrix pipe idles until it arrives (profiles/r01_block_timeline.md: 410 of
// ~2,900 cycles per K step, plus the LDS-DMA issue stalls).  Here the 8 waves form two groups of four - one wave of each
// group per SIMD - that run the SAME program half a phase apart (cdna_hip_programming.md 5, "the 256^2 8-phase template";
// MI355X_MICROARCH.md "Two waves per SIMD"): between two barriers one group only issues MFMAs from registers (16 per wave,
// raised priority) while the other group reads its next fragments from LDS, issues the LDS-DMA of a future K tile and waits
// for its own memory; at the barrier they swap roles.  A SIMD's matrix pipe therefore always has one wave with operands
// ready, and every LDS / DMA latency hides under the partner's MFMA section.
//
//   LDS ring: 4 stages x [BN weight rows | BM pixel rows] x 64 B (K tile = 32 channels of one filter tap; rows XOR-swizzled
//   on the LDS-DMA source address and on the ds_read, as in conv_igemm).  K tile t lives in slot t % 4.
//   Wave tile TM x TN = (BM/2) x (BN/4); a phase = 64 pixels x TN couts x 32 k = 16 MFMAs; PH = TM/64 phases per K tile.
//   Section order per phase (both groups; group 1 enters one barrier late, group 0 leaves one barrier late):
//     load section : ds_read this phase's fragments of K tile t | issue this phase's share of the LDS-DMA of K tile t+3
//                    | last phase only: s_waitcnt vmcnt(2 K tiles) = K tile t+1 has landed | s_waitcnt lgkmcnt(0) | s_barrier
//     MFMA section : 16 x v_mfma_f32_16x16x32_bf16 from registers | s_barrier
//   Hazards: RAW - K tile t+1 is read only after BOTH groups passed the barrier that follows their vmcnt wait for it;
//   WAR - slot (t+3) % 4 = slot of K tile t-1, whose last reads (lgkmcnt(0) before a barrier) precede the barrier that opens
//   the first load section of K tile t.  Beyond the last K tile the DMA is still issued, with out-of-range offsets (zeros into
//   slots nobody reads): the vmcnt arithmetic stays uniform.
#include "conv_common.h"

using namespace yolo_conv;

namespace {

template <int BM, int BN>
__global__ __launch_bounds__(512) void conv_pp_kernel(const ConvArgs a) {
  constexpr int BK = 32, NS = 4, ROWB = 64, RPP = 16;
  constexpr int TM = BM / 2, TN = BN / 4;
  constexpr int NI16 = TM / 16, MI16 = TN / 16;
  constexpr int PH = TM / 64;                          // phases per K tile
  constexpr int PIT = BM / (8 * RPP), WIT = BN / (8 * RPP);   // LDS-DMA instructions per wave per K tile (pixels, weights)
  constexpr int LPS = PIT + WIT;
  constexpr int STAGE_B = (BM + BN) * ROWB;
  constexpr int RING_B = NS * STAGE_B;
  constexpr int EPI_B = 8 * TM * kEpiPitch;
  constexpr int LDS_B = EPI_B > RING_B ? EPI_B : RING_B;
  static_assert(TM % 64 == 0 && TN == 64 && PIT >= 1 && WIT >= 1 && LDS_B <= 160 * 1024, "tile");
  __shared__ __attribute__((aligned(16))) char smem[LDS_B];   // per stage: [BN weight rows][BM pixel rows]

  YOLO_BLOCK_STAMP(a);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = wave >> 2;                           // = the wave's pixel half: waves 0-3 | 4-7, one of each per SIMD
  const int wm = grp, wn = wave & 3;
  const YoloConvDesc& d = a.d;

  int m0, n0;
  {
    const int swz = xcd_swizzle(blockIdx.x, gridDim.x);
    const int mt = swz / a.n_tiles;
    m0 = mt * BM;
    n0 = (swz - mt * a.n_tiles) * BN;
  }
  const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)a.x, 0, a.x_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void*)a.w, 0, a.w_bytes, 0x00020000);

  // ---- LDS-DMA source offsets: piece p = it * 8 + wave covers LDS rows [16 p, 16 p + 16); lane -> (row, physical chunk)
  const int frow = lane >> 2;
  const int chunk = (lane & 3) ^ swz32(lane >> 4);    // logical 8-channel chunk behind this lane's physical slot
  const int hw_out = d.ho * d.wo;
  int px_base[PIT];
  uint32_t px_mask[PIT];                               // bit t: tap t of this pixel lies inside the image
#pragma unroll
  for (int it = 0; it < PIT; ++it) {
    const int m = m0 + (it * 8 + wave) * RPP + frow;
    const bool ok = m < a.M;
    const int mm = ok ? m : 0;
    const int b = mm / hw_out, rem = mm - b * hw_out;
    const int oh = rem / d.wo, ow = rem - oh * d.wo;
    const int hi0 = oh * d.stride - d.pad, wi0 = ow * d.stride - d.pad;
    uint32_t mask;
    if (d.ksize == 3) {
      uint32_t cb = 0, rb = 0;
#pragma unroll
      for (int t = 0; t < 3; ++t) {
        cb |= (uint32_t)((unsigned)(wi0 + t) < (unsigned)d.w) << t;
        rb |= (uint32_t)((unsigned)(hi0 + t) < (unsigned)d.h) << t;
      }
      mask = ((rb & 1) ? cb : 0) | ((rb & 2) ? cb << 3 : 0) | ((rb & 4) ? cb << 6 : 0);
    } else {
      mask = ((unsigned)hi0 < (unsigned)d.h && (unsigned)wi0 < (unsigned)d.w) ? 1u : 0u;
    }
    px_mask[it] = ok ? mask : 0u;
    px_base[it] = ((((b * d.h + hi0) * d.w + wi0) * d.in_c_total + d.in_c_offset) + chunk * 8) * 2;
  }
  uint32_t w_off[WIT];
#pragma unroll
  for (int it = 0; it < WIT; ++it) w_off[it] = (uint32_t)(((n0 + (it * 8 + wave) * RPP + frow) * d.kpad + chunk * 8) * 2);

  // K tile counter of the DMA front: (tap, channel offset) advance by 32 channels per K tile
  int dma_kt = 0, dma_tap = 0, dma_kc = 0;
  const int nk = a.steps;
  // part 0 of a K tile = its pixel pieces, part 1 = its weight pieces (PH == 1: both in the one phase)
  auto issue_x = [&](int slot) {
    char* const xb = smem + slot * STAGE_B + BN * ROWB + wave * 1024;
    const int dh = d.ksize == 3 ? (dma_tap * 11) >> 5 : 0, dw = d.ksize == 3 ? dma_tap - 3 * dh : 0;
    const uint32_t tap_off = (uint32_t)(((dh * d.w + dw) * d.in_c_total + dma_kc) * 2);
    const uint32_t bit = dma_kt < nk ? 1u << dma_tap : 0u;           // beyond the last K tile: zeros
#pragma unroll
    for (int it = 0; it < PIT; ++it) {
      const uint32_t voff = (px_mask[it] & bit) ? (uint32_t)px_base[it] + tap_off : kOobOffset;
      lds_dma16(rx, xb + it * (8 * 1024), voff);
    }
  };
  auto issue_w = [&](int slot) {
    char* const wb = smem + slot * STAGE_B + wave * 1024;
    const bool live = dma_kt < nk;
#pragma unroll
    for (int it = 0; it < WIT; ++it)
      lds_dma16s(rw, wb + it * (8 * 1024), live ? w_off[it] : kOobOffset, (uint32_t)dma_kt * (BK * 2u));
  };
  auto advance = [&]() {
    ++dma_kt;
    dma_kc += BK;
    if (dma_kc >= d.cin) {
      dma_kc = 0;
      ++dma_tap;
    }
  };

  f32x4 acc[MI16][NI16];
#pragma unroll
  for (int i = 0; i < MI16; ++i)
#pragma unroll
    for (int j = 0; j < NI16; ++j)
#pragma unroll
      for (int e = 0; e < 4; ++e) acc[i][j][e] = 0.f;

  // fragment addresses: lane = (row c16 of a 16-row block, 8-channel chunk q); rows 16 apart share their swizzle term
  const int c16 = lane & 15, q = lane >> 4;
  const int frag = ((q ^ swz32(c16 >> 2)) << 4) + c16 * ROWB;
  const int w_frag = (wn * TN) * ROWB + frag;
  const int x_frag = BN * ROWB + (wm * TM) * ROWB + frag;

  // ---- prologue: K tiles 0, 1, 2 in flight; K tile 0 landed and visible
#pragma unroll
  for (int p = 0; p < NS - 1; ++p) {
    issue_x(p);
    issue_w(p);
    advance();
  }
  wait_vmcnt<2 * LPS>();
  __builtin_amdgcn_s_barrier();
  if (grp == 1) __builtin_amdgcn_s_barrier();          // stagger: group 1 runs one section behind group 0
  __builtin_amdgcn_sched_barrier(0);

  int slot = 0, dslot = NS - 1;                        // slot of K tile t / of K tile t + 3
  bf16x8 wf[MI16], xf[4];
  for (int t = 0; t < nk; ++t) {
    const char* const sb = smem + slot * STAGE_B;
#pragma unroll
    for (int ph = 0; ph < PH; ++ph) {
      // ---------------- load section (the partner group multiplies meanwhile)
      if (ph == 0) {
#pragma unroll
        for (int i = 0; i < MI16; ++i) wf[i] = *reinterpret_cast<const bf16x8*>(sb + w_frag + i * 16 * ROWB);
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) xf[j] = *reinterpret_cast<const bf16x8*>(sb + x_frag + (ph * 4 + j) * 16 * ROWB);
      if (PH == 1) {
        issue_x(dslot);
        issue_w(dslot);
      } else if (ph == 0) {
        issue_x(dslot);
      } else {
        issue_w(dslot);
      }
      if (ph == PH - 1) {
        advance();
        wait_vmcnt<2 * LPS>();                         // K tile t + 1 has landed (this wave's share); t + 2, t + 3 keep flying
      }
      wait_lds();                                      // fragments are in registers: nobody reads slot t after the barrier of the last phase
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
      // ---------------- MFMA section (the partner group loads meanwhile)
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int i = 0; i < MI16; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[i][ph * 4 + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[i], xf[j], acc[i][ph * 4 + j], 0, 0, 0);
      __builtin_amdgcn_s_setprio(0);
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
    }
    if (++slot == NS) slot = 0;
    if (++dslot == NS) dslot = 0;
  }
  if (grp == 0) __builtin_amdgcn_s_barrier();          // pairs with group 1's extra barrier at the start
  wait_vmcnt<0>();                                     // the zero-fill DMAs of the tail have landed: LDS is about to be reused
  __syncthreads();

  auto pix_of = [&](int row) -> long {
    const int pix = m0 + wm * TM + row;
    return pix < a.M ? (long)pix : -1L;
  };
  epilogue_lds16<TN / 32, NI16, TM>(a, acc, smem + wave * (TM * kEpiPitch), lane, n0 + wn * TN, pix_of);
}

template <int BM, int BN>
int launch_pp_cfg(const ConvArgs& a, hipStream_t s) {
  ConvArgs b = a;
  b.n_tiles = (a.d.cout + BN - 1) / BN;
  b.steps = a.d.ksize * a.d.ksize * a.d.cin / 32;
  const long grid = (long)((a.M + BM - 1) / BM) * b.n_tiles;
  if (grid > 0x7fffffffL) return yolo_set_error(YOLO_E_UNSUPPORTED, "conv grid too large");
  if (pick_only("pingpong<%dx%d> grid %ld", BM, BN, grid)) return 0;
  hipLaunchKernelGGL((conv_pp_kernel<BM, BN>), dim3((unsigned)grid), dim3(512), 0, s, b);
  return yolo_check_launch("yolo_conv2d_fwd(pp)");
}

}  // namespace

// Returns 1 when the ping-pong kernel does not take the layer (the caller goes on to the other kernels).
// which: 1 = 256 x 256 tiles, 2 = 128 x 256 tiles.
int yolo_conv::launch_pingpong(const ConvArgs& a, int which, hipStream_t s) {
  const YoloConvDesc& d = a.d;
  if (d.cin % 32 != 0 || d.cout % 256 != 0 || d.out_dtype != YOLO_DT_BF16 || d.cout_pad < ((d.cout + 255) / 256) * 256) return 1;
  if (d.ksize != 1 && d.ksize != 3) return 1;
  if (which == 1) return launch_pp_cfg<256, 256>(a, s);
  if (which == 2) return launch_pp_cfg<128, 256>(a, s);
  return 1;
}
