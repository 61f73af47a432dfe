// 3x3 / stride-1 / pad-1 convolution on 20x20 output tiles for gfx950 (MI355X): same contract and numerics as
// conv_igemm.hip / conv3x3_halo.hip (bf16 NHWC in, fp32 accumulate on v_mfma_f32_16x16x32_bf16, fused epilogue).
//
// Why a third 3x3 kernel.  The feature maps of a 640x640 (or 320, 416 -> 20-multiples) YOLOv3 are 20 * 2^k pixels wide, and one
// sub-batch of 16 images gives every heavy 3x3 layer the same work: 16 images x (h/20)(w/20) tiles x cout / CT workgroups with
// (h, w, cout, CT) = (80, 80, 256, 256), (40, 40, 512, 128): exactly 256 workgroups of 400 pixels x CT couts x K each - one per CU,
// no partial last round (the 256x256 gather tiles and the 16x16 halo tiles leave 22 % of the chip idle there,
// profiles/r01_block_timeline.md).  Like the halo kernel it stages each input pixel once per channel chunk instead of once per
// filter tap, so the L2 -> LDS traffic per FLOP is 2.2x below the gather kernel's, and a wave issues 1-3 LDS-DMA per 50 MFMAs.
//
//   tile      20 x 20 output pixels of one image = 25 patches of 4 x 4 pixels; a patch is one 16-pixel MFMA fragment
//             (lane c16 -> pixel (c16 >> 2, c16 & 3) of the patch), so a fragment's LDS address is
//             per-lane base + compile-time constant for every patch and every filter tap (ds_read immediates, no VALU)
//   LDS       halo [2][22*22 pixel rows of 64 B] (32 channels per chunk, double-buffered across chunks)
//             weights ring [3][CT rows of 64 B] (one filter tap of one chunk per slot)
//             halo rows are swizzled by physical 16-byte slot = chunk ^ 2*(halo row y & 1), weight rows by swz32 (conv_common.h):
//             both conflict-free for ds_read_b128's lane groups at every tap offset (tools/micro/lds_groups.hip)
//   waves     8 = NWN cout groups of 64 x NWM pixel groups.  24 patches are dealt evenly to the pixel groups (RPG patch rows each
//             + RPG patches of the last patch row); the 25th patch is shared: pixel group g multiplies it with cout fragments
//             [g*NL, g*NL+NL) of its 64 - every wave issues the same number of MFMAs (50 or 25 per 32-deep K step)
//   step      (chunk, tap): s_waitcnt vmcnt(0); s_barrier; issue weights of step+2 and a piece of the next chunk's halo;
//             4 weight + 13 (7) pixel fragment reads; 50 (25) MFMAs
//   epilogue  32 pixels x 64 couts at a time through a per-wave fp32 LDS slab -> 16-byte accesses over whole 128-byte lines
//             (residual read, pre-add copy, store); the shared patch goes out with 8-byte accesses straight from registers
#include "conv_common.h"

using namespace yolo_conv;

namespace {

constexpr int kT20 = 20, kHW2 = 22, kHP = kHW2 * kHW2;       // tile edge, halo edge, halo pixels
constexpr int kHaloB = 32 * 1024;                             // 31 pieces of 16 rows, padded to 32 (4 per wave)

// NWM = 1: 4 waves (one per SIMD, up to 512 registers), each 64 couts x all 25 patches (400 accumulator registers): the form for
//          CT = 256 - two waves per SIMD would need 200 accumulators + fragments in 256 registers, which hipcc only reaches by
//          spilling accumulators inside the loop
// NWM = 2, 4: NWN * NWM waves, the patches dealt to NWM pixel groups as described above
template <int CT, int NWM>
__global__ __launch_bounds__(64 * (CT / 64) * NWM) void conv3x3_t20_kernel(const ConvArgs a) {
  constexpr int NWN = CT / 64, NW = NWN * NWM;
  static_assert((NW == 4 || NW == 8) && (NWM == 1 || NWM == 2 || NWM == 4), "4 or 8 waves");
  constexpr int RPG = NWM == 1 ? 5 : 4 / NWM;       // full patch rows per pixel group
  constexpr int NFULL = 5 * RPG, NMAIN = NWM == 1 ? 25 : NFULL + RPG;   // patches of the full rows, + RPG patches of patch row 4
  constexpr int NL = NWM == 1 ? 0 : 4 / NWM;        // cout fragments of the shared patch per wave
  constexpr int WBUF_B = CT * 64, NS = 3;
  constexpr int HPT = 32 / NW;                      // halo pieces per wave and chunk
  constexpr int WPIECES = CT / 16, WIT = WPIECES / NW;
  constexpr int RING_B = 2 * kHaloB + NS * WBUF_B;
  constexpr int EPI_ROWS = 32, EPI_SLAB = EPI_ROWS * kEpiPitch2;
  constexpr int LDS_B = RING_B > NW * EPI_SLAB ? RING_B : NW * EPI_SLAB;
  static_assert(WIT >= 1 && LDS_B <= 160 * 1024, "tile");
  __shared__ __attribute__((aligned(16))) char smem[LDS_B];
  char* const s_w = smem + 2 * kHaloB;

  YOLO_BLOCK_STAMP(a);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int pg = wave / NWN, cg = wave % NWN;       // waves w and w + 4 share a SIMD: with NWN = 4 they hold both pixel groups
  const YoloConvDesc& d = a.d;

  const int tiles_x = (d.w + kT20 - 1) / kT20, tiles_y = (d.h + kT20 - 1) / kT20;
  int b, y0, x0, n0;
  {
    int swz = xcd_swizzle(blockIdx.x, gridDim.x);
    n0 = (swz % a.n_tiles) * CT;
    swz /= a.n_tiles;
    x0 = (swz % tiles_x) * kT20;
    swz /= tiles_x;
    y0 = (swz % tiles_y) * kT20;
    b = swz / tiles_y;
  }
  const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)a.x, 0, a.x_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void*)a.w, 0, a.w_bytes, 0x00020000);

  // ---- LDS-DMA sources.  A piece = 16 LDS rows x 64 B; lane -> (row lane >> 2, physical slot lane & 3)
  uint32_t h_off[HPT];
#pragma unroll
  for (int it = 0; it < HPT; ++it) {
    const int hp = (it * NW + wave) * 16 + (lane >> 2);
    const int hy = hp / kHW2, hx = hp - hy * kHW2;
    const int yy = y0 - 1 + hy, xx = x0 - 1 + hx;
    const int chunk = (lane & 3) ^ ((hy & 1) << 1);
    const bool ok = hp < kHP && (unsigned)yy < (unsigned)d.h && (unsigned)xx < (unsigned)d.w;
    h_off[it] = ok ? (uint32_t)((((b * d.h + yy) * d.w + xx) * d.in_c_total + d.in_c_offset + chunk * 8) * 2) : kOobOffset;
  }
  uint32_t w_off[WIT];
#pragma unroll
  for (int it = 0; it < WIT; ++it) {
    const int r = (it * NW + wave) * 16 + (lane >> 2);
    const int chunk = (lane & 3) ^ swz32(lane >> 4);
    w_off[it] = (uint32_t)(((n0 + r) * d.kpad + chunk * 8) * 2);
  }
  auto issue_halo = [&](int hb, int c, int it) {
    lds_dma16s(rx, smem + hb * kHaloB + (it * NW + wave) * 1024, h_off[it], (uint32_t)c * 64u);
  };
  auto issue_w = [&](int slot, int c, int tap) {
#pragma unroll
    for (int it = 0; it < WIT; ++it)
      lds_dma16s(rw, s_w + slot * WBUF_B + (it * NW + wave) * 1024, w_off[it], (uint32_t)((tap * d.cin + c * 32) * 2));
  };

  // ---- fragment bases.  lane = (patch pixel c16 = (dy, dx), 8-channel chunk q)
  const int c16 = lane & 15, q = lane >> 4;
  const int dy = c16 >> 2, dx = c16 & 3;
  constexpr uint32_t lds0 = 0;                      // fragment bases are byte offsets into smem
  // halo row parity of (dy + dh): L[0] for dh = 0, 2; L[1] for dh = 1
  uint32_t L[2], A[2], B[2];
#pragma unroll
  for (int par = 0; par < 2; ++par) {
    L[par] = lds0 + (uint32_t)((dy * kHW2 + dx) * 64 + ((q ^ (((dy + par) & 1) << 1)) << 4));
    A[par] = L[par] + (uint32_t)(RPG * pg * 4 * kHW2 * 64);     // patches of the full rows: (RPG*pg + jj/5, jj%5)
    B[par] = L[par] + (uint32_t)(RPG * pg * 4 * 64);            // patches of patch row 4: (4, RPG*pg + jj - NFULL)
  }
  uint32_t W = lds0 + (uint32_t)(2 * kHaloB + (cg * 64 + c16) * 64 + ((q ^ swz32(c16 >> 2)) << 4));
  uint32_t WL = W + (uint32_t)(NL * pg * 1024);                  // the shared patch's cout fragments of this wave
  // The bases stay opaque registers: only the small per-fragment constants below may go into the ds_read immediates (folded
  // together with 2 * kHaloB they exceed 16 bits and the compiler materialises one address register per fragment).
#if defined(__HIP_DEVICE_COMPILE__)
  asm volatile("" : "+v"(L[0]), "+v"(L[1]), "+v"(A[0]), "+v"(A[1]), "+v"(B[0]), "+v"(B[1]), "+v"(W), "+v"(WL));
#endif

  // NWM = 1 holds 100 accumulator tiles: patches [0, NACC_A) live in the accumulator file (AGPRs), the rest in VGPRs, and the MFMAs
  // are written as asm with the accumulator tied to itself: given the builtin, hipcc renames every accumulator per MFMA and
  // shuffles tiles between the two files (v_accvgpr_*) and scratch inside the loop.  The asm statements are volatile, i.e. issued
  // in program order; every tile sees one MFMA per K step, so no MFMA reads the D of one still in flight.
  constexpr bool ASM_MFMA = true;
  constexpr int NACC_A = NWM == 1 ? 16 : 0;
  f32x4 acc[4][NMAIN - NACC_A], acca[4][NACC_A ? NACC_A : 1], accl[NL ? NL : 1];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
#pragma unroll
    for (int j = 0; j < NMAIN - NACC_A; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < NACC_A; ++j) acca[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
#pragma unroll
  for (int i = 0; i < NL; ++i) accl[i] = f32x4{0.f, 0.f, 0.f, 0.f};

  auto ldsr = [&](uint32_t off, int imm) -> bf16x8 {   // per-lane offset + compile-time constant (the ds_read immediate)
    const char* const p = smem + off;
    return *reinterpret_cast<const bf16x8*>(p + imm);
  };

  const int nch = d.cin / 32;
  auto mfma = [&](auto ic, auto jc, const bf16x8& wa, const bf16x8& xb) {
    constexpr int i = decltype(ic)::value, jj = decltype(jc)::value;
#if defined(__HIP_DEVICE_COMPILE__)
    if constexpr (jj < NACC_A) {
      f32x4& t = acca[i][jj];
      asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(t) : "v"(wa), "v"(xb));
    } else {
      f32x4& t = acc[i][jj - NACC_A];
      asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(t) : "v"(wa), "v"(xb));
    }
#endif
  };
  // fragment reads of filter tap `tap` (compile-time after unrolling) against the halo bases am / bm / lm
  auto xread = [&](int jj, int tap, uint32_t am, uint32_t bm) -> bf16x8 {
    const int toff = ((tap / 3) * kHW2 + tap % 3) * 64;
    if (jj < NFULL) return ldsr(am, ((4 * (jj / 5)) * kHW2 + 4 * (jj % 5)) * 64 + toff);
    return ldsr(bm, (16 * kHW2 + 4 * (jj - NFULL)) * 64 + toff);
  };

  // ---- prologue: halo of chunk 0, weights of steps 0 and 1; then the fragments the first step starts with
#pragma unroll
  for (int it = 0; it < HPT; ++it) issue_halo(0, 0, it);
  issue_w(0, 0, 0);
  issue_w(1, 0, 1);
  wait_vmcnt<0>();
  __builtin_amdgcn_s_barrier();
  bf16x8 wf[4], xf[3];
#pragma unroll
  for (int i = 0; i < 4; ++i) wf[i] = ldsr(W, i * 1024);
  xf[0] = xread(0, 0, A[0], B[0]);
  xf[1] = xread(1, 0, A[0], B[0]);

  // Step s = (chunk c, tap): the step's weight fragments and its first two pixel fragments are already in registers (fetched in
  // the tail of step s-1, from a ring slot / halo buffer made visible by the barrier before it), so the MFMA stream starts right
  // behind the barrier:
  //   issue LDS-DMA: weights of step s+2 (ring slot (s+2) % 3, last read in the tail of step s-2), a piece of the next chunk's halo
  //   patches 0 .. N-3: 4 MFMAs each, the pixel fragment of patch jj+2 fetched meanwhile (three fragment registers in rotation)
  //   patches N-2, N-1 by cout fragment i: 2 MFMAs, then weight fragment i of step s+1 into the same registers; pixel fragments
  //   0, 1 of step s+1 around them
  //   s_waitcnt vmcnt(0); s_barrier: the DMA issued above has landed (step s+1's tail reads it)
  for (int c = 0; c < nch; ++c) {
    const bool next_chunk = c + 1 < nch;
    static_for<9>([&](auto tc) {
      constexpr int tap = decltype(tc)::value;
      constexpr int slot_n = (tap + 1) % 3, tap_n = (tap + 1) % 9;
      constexpr int par = (tap / 3) & 1, par_n = (tap_n / 3) & 1;
      constexpr int r = (tap * NMAIN) % 3;          // rotation of the three pixel-fragment registers in this step
      {
        constexpr int t2 = (tap + 2) % 9;
        const int c2 = c + (tap + 2) / 9;
        if (c2 < nch) issue_w((tap + 2) % 3, c2, t2);
        if (tap < HPT && next_chunk) issue_halo((c + 1) & 1, c + 1, tap);
      }
      const uint32_t am = A[par], bm = B[par];
      // the next step's halo bases: the other buffer after tap 8
      const uint32_t flip = tap == 8 ? (uint32_t)kHaloB : 0u;
      const uint32_t am_n = A[par_n] ^ flip, bm_n = B[par_n] ^ flip;
      bf16x8 xl, wl[NL ? NL : 1];
      if constexpr (NL > 0) {                       // shared patch (4, 4): multiplied behind patch 1
        xl = ldsr(L[par], (16 * kHW2 + 16) * 64 + ((tap / 3) * kHW2 + tap % 3) * 64);
#pragma unroll
        for (int i = 0; i < NL; ++i) wl[i] = ldsr(WL, (tap % 3) * WBUF_B + i * 1024);
      }
      static_for<NMAIN - 2>([&](auto jc) {
        constexpr int jj = decltype(jc)::value;
        xf[(jj + 2 + r) % 3] = xread(jj + 2, tap, am, bm);
        static_for<4>([&](auto ic) { mfma(ic, jc, wf[decltype(ic)::value], xf[(jj + r) % 3]); });
        if constexpr (NL > 0 && jj == 1) {
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
          for (int i = 0; i < NL; ++i) {
            f32x4& t = accl[i];                       // (bound outside the asm: operands alone do not capture in a lambda)
            const bf16x8 &wa = wl[i], &xb = xl;
            asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(t) : "v"(wa), "v"(xb));
          }
#endif
        }
      });
      xf[(NMAIN + r) % 3] = xread(0, tap_n, am_n, bm_n);
      static_for<4>([&](auto ic) {
        constexpr int i = decltype(ic)::value;
        mfma(ic, std::integral_constant<int, NMAIN - 2>{}, wf[i], xf[(NMAIN - 2 + r) % 3]);
        mfma(ic, std::integral_constant<int, NMAIN - 1>{}, wf[i], xf[(NMAIN - 1 + r) % 3]);
        wf[i] = ldsr(W, slot_n * WBUF_B + i * 1024);
      });
      xf[(NMAIN + 1 + r) % 3] = xread(1, tap_n, am_n, bm_n);
      wait_vmcnt<0>();
      __builtin_amdgcn_s_barrier();
    });
    // the other halo buffer (every base is < 32 KiB: the XOR toggles bit 15)
#pragma unroll
    for (int par = 0; par < 2; ++par) {
      L[par] ^= (uint32_t)kHaloB;
      A[par] ^= (uint32_t)kHaloB;
      B[par] ^= (uint32_t)kHaloB;
    }
  }
#if defined(__HIP_DEVICE_COMPILE__)
  if constexpr (ASM_MFMA) asm volatile("s_nop 15\n\ts_nop 15");   // the last asm MFMAs' D registers: 12 wait states before any other reader
#endif
  if (a.debug & 8) return;                          // timing-only ablation (YOLO_CONV_DEBUG bit 8): no epilogue, y is not written
  __syncthreads();                                  // every wave is done with the ring: the epilogue slabs reuse it

  // ---- epilogue
  const int cbase = n0 + cg * 64;                   // this wave's 64 couts
  char* const stg = smem + wave * EPI_SLAB;
  const int crow = lane >> 3, cchunk = lane & 7;    // coalesced phase: 8 lanes per pixel, 8 couts each
  f32x4 bv[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) bv[i] = *reinterpret_cast<const f32x4*>(a.bias + cbase + i * 16 + q * 4);
  // act(v) = min(max(v, lo), hi), lo = 0 (ReLU, ReLU6) or slope * v (LeakyReLU 0.1: slope 0.1; none: slope 1): the same values as
  // apply_act (up to the sign of a zero) without a per-element switch on d.act
  const bool floor0 = d.act == YOLO_ACT_RELU || d.act == YOLO_ACT_RELU6;
  const float slope = d.act == YOLO_ACT_LEAKY01 ? 0.1f : 1.f;
  const float hi_clamp = act_hi(d.act);
  auto act4 = [&](f32x4 v) -> f32x4 {
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = fminf(fmaxf(v[e], floor0 ? 0.f : slope * v[e]), hi_clamp);
    return v;
  };
  // Global accesses of the coalesced phase go through buffer descriptors: address = per-lane byte offset (row of the pass,
  // 8-cout chunk) + a wave-uniform offset per patch, and a lane whose pixel lies outside the image gets an out-of-range offset
  // (the store is dropped, the load returns 0) - no 64-bit address arithmetic and no branches per pass.
  // The patch offset is ADDED INTO THE VGPR offset, not passed in the instruction's SGPR soffset field: hipcc pads the
  // "VALU overwrites the data registers of a > 64-bit store" hazard only for buffer stores without a register soffset, and with
  // one it placed a v_or_b32 of a data register right behind buffer_store_dwordx4 - on gfx950 the store then wrote the new
  // value in its second dword for some lanes (a few wrong output pairs per launch, lanes 12-15 / 28-31).
  const uint32_t y_pitch = (uint32_t)d.out_c_total * 2u, r_pitch = (uint32_t)d.res_c_total * 2u, x_pitch = (uint32_t)d.aux_c_total * 2u;
  const uint32_t npix = (uint32_t)d.n * d.h * d.w;
  const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc(a.y, 0, npix * y_pitch, 0x00020000);
  const __amdgpu_buffer_rsrc_t rr = __builtin_amdgcn_make_buffer_rsrc((void*)a.res, 0, a.res ? npix * r_pitch : 0u, 0x00020000);
  const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc((void*)a.aux, 0, a.aux ? npix * x_pitch : 0u, 0x00020000);
  const bool cout_ok = cbase < d.cout;              // cout % 64 == 0 (launcher): a wave's group is all-or-nothing
  // pass p of a patch pair covers patch pixels (p & 1) * 8 + crow: (dy, dx) = ((p & 1) * 2 + (crow >> 2), crow & 3)
  uint32_t yo[2], ro[2], ao[2];                     // byte offsets of this lane's pixel of patch (0, 0) in y / residual / pre-add copy
  int ylim[2];                                      // patch row pr holds the pixel iff 4 * pr < ylim[half]
  const uint32_t ccol = (uint32_t)(cbase + cchunk * 8) * 2u;
#pragma unroll
  for (int hf = 0; hf < 2; ++hf) {
    const int dyh = hf * 2 + (crow >> 2);
    const uint32_t lpix = (uint32_t)((b * d.h + y0 + dyh) * d.w + x0 + (crow & 3));
    yo[hf] = lpix * y_pitch + (uint32_t)d.out_c_offset * 2u + ccol;
    ro[hf] = lpix * r_pitch + (uint32_t)d.res_c_offset * 2u + ccol;
    ao[hf] = lpix * x_pitch + (uint32_t)d.aux_c_offset * 2u + ccol;
    ylim[hf] = cout_ok ? d.h - y0 - dyh : 0;
  }
  const int xlim = d.w - x0 - (crow & 3);           // patch column pc holds the pixel iff 4 * pc < xlim
  auto patch_of = [&](int jj, int& pr, int& pc) {
    if (jj < NFULL) {
      pr = RPG * pg + jj / 5;
      pc = jj % 5;
    } else {
      pr = 4;
      pc = RPG * pg + (jj - NFULL);
    }
  };
  // byte offset of pass `pass` of pair jp (base = yo / ro / ao): out of range when the pixel lies outside the image
  auto voff = [&](const uint32_t (&base)[2], uint32_t pitch, int jp, int pass) -> uint32_t {
    int pr, pc;
    patch_of(jp + (pass >> 1), pr, pc);
    const int hf = pass & 1;
    const bool ok = 4 * pr < ylim[hf] && 4 * pc < xlim;
    return ok ? base[hf] + (uint32_t)(4 * pr * d.w + 4 * pc) * pitch : kOobOffset;
  };
  // The residual of pair p + 1 is fetched while pair p is staged and stored: with one or two waves per SIMD nothing else hides
  // the latency of a load that sits between the LDS read and the store of the same pass.
  constexpr int NPAIR = (NMAIN + 1) / 2;
  u32x4 rv[2][4];
  auto fetch_res = [&](auto jpc) {
    constexpr int pi = decltype(jpc)::value, jp = 2 * pi;
#pragma unroll
    for (int pass = 0; pass < (jp + 1 < NMAIN ? 4 : 2); ++pass)
      rv[pi & 1][pass] = __builtin_amdgcn_raw_buffer_load_b128(rr, voff(ro, r_pitch, jp, pass), 0, 0);
  };
  if (a.res) fetch_res(std::integral_constant<int, 0>{});
  static_for<NPAIR>([&](auto jpc) {
    constexpr int pi = decltype(jpc)::value, jp = 2 * pi;
    static_for<(jp + 1 < NMAIN ? 2 : 1)>([&](auto uc) {
      constexpr int u = decltype(uc)::value;
      static_for<4>([&](auto ic) {
        constexpr int i = decltype(ic)::value;
        f32x4 av;
        if constexpr (jp + u < NACC_A) av = acca[i][jp + u];
        else av = acc[i][jp + u - NACC_A];
        *reinterpret_cast<f32x4*>(stg + (u * 16 + c16) * kEpiPitch2 + (i * 16 + q * 4) * 4) = act4(av + bv[i]);
      });
    });
    if constexpr (pi + 1 < NPAIR) {
      if (a.res) fetch_res(std::integral_constant<int, pi + 1>{});
    }
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int pass = 0; pass < (jp + 1 < NMAIN ? 4 : 2); ++pass) {
      const int row = pass * 8 + crow;              // 0..31: patch jp + (row >> 4), patch pixel row & 15
      const f32x4 lo = *reinterpret_cast<const f32x4*>(stg + row * kEpiPitch2 + cchunk * 32);
      const f32x4 hi = *reinterpret_cast<const f32x4*>(stg + row * kEpiPitch2 + cchunk * 32 + 16);
      float v[8] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
      if (a.aux) {
        bf16x8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = (bf16_t)v[e];
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o), ra, voff(ao, x_pitch, jp, pass), 0, 0);
      }
      if (a.res) {
        const bf16x8 r8 = __builtin_bit_cast(bf16x8, rv[pi & 1][pass]);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] += (float)r8[e];
      }
      bf16x8 o;
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] = (bf16_t)v[e];
      __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o), ry, voff(yo, y_pitch, jp, pass), 0, 0);
    }
    __builtin_amdgcn_wave_barrier();
  });
  // shared patch (4, 4): lane = (pixel c16, couts 4q..4q+3 of fragment NL*pg + i)
  if constexpr (NL > 0) {
    const int yy = y0 + 16 + dy, xx = x0 + 16 + dx;
    if (yy < d.h && xx < d.w && cout_ok) {
      const long pix = (long)(b * d.h + yy) * d.w + xx;
#pragma unroll
      for (int i = 0; i < NL; ++i) {
        const int cl = (NL * pg + i) * 16 + q * 4;
        const f32x4 bl = *reinterpret_cast<const f32x4*>(a.bias + cbase + cl);
        const long cofs = cbase + cl;
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = apply_act(accl[i][e] + bl[e], d.act);
        if (a.aux) {
          bf16x4 o;
#pragma unroll
          for (int e = 0; e < 4; ++e) o[e] = (bf16_t)v[e];
          *reinterpret_cast<bf16x4*>(a.aux + pix * d.aux_c_total + d.aux_c_offset + cofs) = o;
        }
        if (a.res) {
          const bf16x4 rv = *reinterpret_cast<const bf16x4*>(a.res + pix * d.res_c_total + d.res_c_offset + cofs);
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] += (float)rv[e];
        }
        bf16x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = (bf16_t)v[e];
        *reinterpret_cast<bf16x4*>(reinterpret_cast<bf16_t*>(a.y) + pix * d.out_c_total + d.out_c_offset + cofs) = o;
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// Second form: 4 waves, 400 pixels x 128 couts per workgroup, TWO workgroups per CU (72 KB of LDS, 256 registers).
//   * a wave owns 32 couts (two fragments) x all 25 patches: its weight fragments are nobody else's, so they never touch LDS -
//     each lane fetches its 16 bytes of W[cout row][k] straight into registers (buffer_load_dwordx4, three register sets in
//     rotation, two steps ahead).  No weight ring, no LDS-DMA for weights, and the only barrier left in the main loop is the one
//     that hands over a halo buffer: one per 9 steps.
//   * two resident workgroups drift apart, so one's epilogue (HBM-bound: 100 KB written + 100 KB of residual read) runs under the
//     other's MFMA stream; with two sub-batch streams the second workgroup of a CU can come from the other stream's launch.
//   * epilogue: the four waves stage a patch pair as [32 pixels][128 couts] fp32 in LDS (two buffers in turn, one barrier per
//     pair) and store whole 256-byte pixel rows.
constexpr int kV2Hofs = 8 * 256 * 4;                 // per-thread halo source offsets (8 pieces per wave): kept in LDS, not in registers
constexpr int kV2Pitch = 528;                        // fp32 staging row: 128 couts + 16 B
constexpr int kV2Lds = 2 * kHaloB + kV2Hofs;
static_assert(2 * 32 * kV2Pitch <= 2 * kHaloB, "epilogue buffers reuse the halo region");

// Epilogue shared by the 4-wave forms (conv3x3_t20v2_kernel, conv3x3s2_t20_kernel): acc[i][jj] = couts (32 wave + 16 i + 4 q ..+3) x
// the 16 pixels of patch jj of the 20x20 output tile at (b, y0, x0); `smem` = 2 x 32 x kV2Pitch bytes the main loop no longer needs.
template <int RD>
__device__ __forceinline__ void t20v2_epilogue(const ConvArgs& a, char* smem, f32x4 (&acc)[2][25], int wave, int lane, int b, int y0, int x0, int n0) {
  constexpr int NP = 25;
  const YoloConvDesc& d = a.d;
  const int c16 = lane & 15, q = lane >> 4;
  const int ho = d.ho, wo = d.wo;                      // (stride 1: = h, w)
  // ---- epilogue (the final barrier of the loop has passed: the halo buffers are free; the weight prefetches of the two
  // steps beyond the end land in registers nobody reads)
  const int lrow = lane >> 4, cch = lane & 15;          // coalesced phase: 16 lanes per pixel row, 8 couts each
  f32x4 bv[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) bv[i] = *reinterpret_cast<const f32x4*>(a.bias + n0 + wave * 32 + i * 16 + q * 4);
  const bool floor0 = d.act == YOLO_ACT_RELU || d.act == YOLO_ACT_RELU6;
  const float slope = d.act == YOLO_ACT_LEAKY01 ? 0.1f : 1.f;
  const float hi_clamp = act_hi(d.act);
  auto act4 = [&](f32x4 v) -> f32x4 {
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = fminf(fmaxf(v[e], floor0 ? 0.f : slope * v[e]), hi_clamp);
    return v;
  };
  const uint32_t y_pitch = (uint32_t)d.out_c_total * 2u, r_pitch = (uint32_t)d.res_c_total * 2u, x_pitch = (uint32_t)d.aux_c_total * 2u;
  const uint32_t npix = (uint32_t)d.n * ho * wo;
  const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc(a.y, 0, npix * y_pitch, 0x00020000);
  const __amdgpu_buffer_rsrc_t rr = __builtin_amdgcn_make_buffer_rsrc((void*)a.res, 0, a.res ? npix * r_pitch : 0u, 0x00020000);
  const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc((void*)a.aux, 0, a.aux ? npix * x_pitch : 0u, 0x00020000);
  // pass u of a pair covers patch jp + u; this lane's pixel of a patch is (dy, dx) = (wave, lrow)
  const uint32_t lpix = (uint32_t)((b * ho + y0 + wave) * wo + x0 + lrow);
  const uint32_t ccol = (uint32_t)(n0 + cch * 8) * 2u;
  const uint32_t yo = lpix * y_pitch + (uint32_t)d.out_c_offset * 2u + ccol;
  const uint32_t ro = lpix * r_pitch + (uint32_t)d.res_c_offset * 2u + ccol;
  const uint32_t ao = lpix * x_pitch + (uint32_t)d.aux_c_offset * 2u + ccol;
  const int ylim = ho - y0 - wave, xlim = wo - x0 - lrow;       // patch (pr, pc) holds this lane's pixel iff 4 pr < ylim && 4 pc < xlim
  // (no SGPR soffset on the 16-byte stores: see the first form's epilogue)
  auto voff = [&](uint32_t base, uint32_t pitch, int jj) -> uint32_t {
    const int pr = jj / 5, pc = jj % 5;
    const bool ok = 4 * pr < ylim && 4 * pc < xlim;
    return ok ? base + (uint32_t)(4 * pr * wo + 4 * pc) * pitch : kOobOffset;
  };
  constexpr int NPAIR = (NP + 1) / 2;
  // The residual rows of pair pi are requested RD pairs before they are added, and nothing in the loop drains the vector-memory
  // counter: the barrier is the raw one behind an LDS-only wait (a __syncthreads() would wait for every load and store in flight
  // - one HBM round trip per pair, 13 in a row), the compiler's own counted vmcnt before the first use of a row does the rest.
  u32x4 rv[RD + 1][2];
  auto fetch_res = [&](auto pc_) {
    constexpr int pi = decltype(pc_)::value, jp = 2 * pi;
#pragma unroll
    for (int u = 0; u < (jp + 1 < NP ? 2 : 1); ++u)
      rv[pi % (RD + 1)][u] = __builtin_amdgcn_raw_buffer_load_b128(rr, voff(ro, r_pitch, jp + u), 0, 0);
  };
  if (a.res) static_for<RD>([&](auto kc) { fetch_res(kc); });
  static_for<NPAIR>([&](auto pc_) {
    constexpr int pi = decltype(pc_)::value, jp = 2 * pi;
    constexpr int NU = jp + 1 < NP ? 2 : 1;
    char* const slab = smem + (pi & 1) * (32 * kV2Pitch);
    static_for<NU>([&](auto uc) {
      constexpr int u = decltype(uc)::value;
      static_for<2>([&](auto ic) {
        constexpr int i = decltype(ic)::value;
        *reinterpret_cast<f32x4*>(slab + (u * 16 + c16) * kV2Pitch + (wave * 32 + i * 16 + q * 4) * 4) = act4(acc[i][jp + u] + bv[i]);
      });
    });
    if constexpr (pi + RD < NPAIR) {
      if (a.res) fetch_res(std::integral_constant<int, pi + RD>{});
    }
    wait_lds();
    __builtin_amdgcn_s_barrier();                       // the pair is staged by all four waves (and pair pi - 1 has been read by all)
#pragma unroll
    for (int u = 0; u < NU; ++u) {
      const int row = u * 16 + wave * 4 + lrow;
      const f32x4 lo = *reinterpret_cast<const f32x4*>(slab + row * kV2Pitch + cch * 32);
      const f32x4 hi = *reinterpret_cast<const f32x4*>(slab + row * kV2Pitch + cch * 32 + 16);
      float v[8] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
      if (a.aux) {
        bf16x8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = (bf16_t)v[e];
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o), ra, voff(ao, x_pitch, jp + u), 0, 0);
      }
      if (a.res) {
        const bf16x8 r8 = __builtin_bit_cast(bf16x8, rv[pi % (RD + 1)][u]);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] += (float)r8[e];
      }
      bf16x8 o;
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] = (bf16_t)v[e];
      __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o), ry, voff(yo, y_pitch, jp + u), 0, 0);
    }
  });
}

// XD: pixel fragments in flight (XD register sets in rotation, XD - 1 patches ahead of the MFMAs); RD: residual rows in flight
// in the epilogue (RD patch pairs ahead of the pair being stored)
template <int XD, int RD>
__global__ __launch_bounds__(256, 2) void conv3x3_t20v2_kernel(const ConvArgs a) {
  constexpr int CT = 128, NW = 4, NP = 25, HPT = 8;
  __shared__ __attribute__((aligned(16))) char smem[kV2Lds];
  uint32_t* const hofs = reinterpret_cast<uint32_t*>(smem + 2 * kHaloB);

  YOLO_BLOCK_STAMP(a);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const YoloConvDesc& d = a.d;
  const int tiles_x = (d.w + kT20 - 1) / kT20, tiles_y = (d.h + kT20 - 1) / kT20;
  int b, y0, x0, n0;
  {
    int swz = a.blk_total ? xcd_swizzle(blockIdx.x + a.blk_off, a.blk_total) : xcd_swizzle(blockIdx.x, gridDim.x);
    n0 = (swz % a.n_tiles) * CT;
    swz /= a.n_tiles;
    x0 = (swz % tiles_x) * kT20;
    swz /= tiles_x;
    y0 = (swz % tiles_y) * kT20;
    b = swz / tiles_y;
  }
  const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)a.x, 0, a.x_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void*)a.w, 0, a.w_bytes, 0x00020000);

  // ---- halo pieces of this wave: piece (it * 4 + wave) = LDS rows [16 piece, +16); lane -> (row lane >> 2, physical slot lane & 3)
#pragma unroll
  for (int it = 0; it < HPT; ++it) {
    const int hp = (it * NW + wave) * 16 + (lane >> 2);
    const int hy = hp / kHW2, hx = hp - hy * kHW2;
    const int yy = y0 - 1 + hy, xx = x0 - 1 + hx;
    const int chunk = (lane & 3) ^ ((hy & 1) << 1);
    const bool ok = hp < kHP && (unsigned)yy < (unsigned)d.h && (unsigned)xx < (unsigned)d.w;
    const uint32_t ho = ok ? (uint32_t)((((b * d.h + yy) * d.w + xx) * d.in_c_total + d.in_c_offset + chunk * 8) * 2) : kOobOffset;
    hofs[it * 256 + tid] = ho;
    lds_dma16s(rx, smem + (it * NW + wave) * 1024, ho, 0u);          // chunk 0 -> halo buffer 0
  }
  auto issue_halo = [&](int hb, int c, int it) {
    lds_dma16s(rx, smem + hb * kHaloB + (it * NW + wave) * 1024, hofs[it * 256 + tid], (uint32_t)c * 64u);
  };

  // ---- fragment addressing
  const int c16 = lane & 15, q = lane >> 4;
  const int dy = c16 >> 2, dx = c16 & 3;
  uint32_t A[2];                                       // halo bases by parity of (dy + dh), see the first form
#pragma unroll
  for (int par = 0; par < 2; ++par) A[par] = (uint32_t)((dy * kHW2 + dx) * 64 + ((q ^ (((dy + par) & 1) << 1)) << 4));
#if defined(__HIP_DEVICE_COMPILE__)
  asm volatile("" : "+v"(A[0]), "+v"(A[1]));
#endif
  const uint32_t wv = (uint32_t)(((n0 + wave * 32 + c16) * d.kpad + q * 8) * 2);   // this lane's 16 bytes of fragment 0, k = 0
  const uint32_t wfrag = (uint32_t)(16 * d.kpad * 2);                             // fragment 1: 16 cout rows further
  auto wload = [&](int c, int tap, int i) -> bf16x8 {
    return __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rw, wv, (uint32_t)((tap * d.cin + c * 32) * 2) + i * wfrag, 0));
  };
  auto ldsr = [&](uint32_t off, int imm) -> bf16x8 {
    const char* const p = smem + off;
    return *reinterpret_cast<const bf16x8*>(p + imm);
  };
  auto xread = [&](int jj, int tap, uint32_t am) -> bf16x8 {
    return ldsr(am, ((4 * (jj / 5)) * kHW2 + 4 * (jj % 5)) * 64 + ((tap / 3) * kHW2 + tap % 3) * 64);
  };

  f32x4 acc[2][NP];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < NP; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  auto mfma = [&](auto ic, auto jc, const bf16x8& wa, const bf16x8& xb) {
    constexpr int i = decltype(ic)::value, jj = decltype(jc)::value;
    f32x4& t = acc[i][jj];
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(t) : "v"(wa), "v"(xb));
#endif
  };

  const int nch = d.cin / 32;
  bf16x8 wf[3][2], xf[XD];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    wf[0][i] = wload(0, 0, i);
    wf[1][i] = wload(0, 1, i);
  }
  wait_vmcnt<4>();                                     // the halo DMAs (issued before the four weight loads) have landed
  __builtin_amdgcn_s_barrier();

  for (int c = 0; c < nch; ++c) {
    static_for<9>([&](auto tc) {
      constexpr int tap = decltype(tc)::value;
      constexpr int par = (tap / 3) & 1, par_n = (((tap + 1) % 9) / 3) & 1;
      // next chunk's halo (the last chunk fetches a dummy one: the vmcnt arithmetic stays uniform), then the weights of step + 2
      if constexpr (tap < HPT) issue_halo((c + 1) & 1, c + 1, tap);
      {
        constexpr int t2 = (tap + 2) % 9;
        const int c2 = c + (tap + 2) / 9;
#pragma unroll
        for (int i = 0; i < 2; ++i) wf[(tap + 2) % 3][i] = wload(c2, t2, i);
      }
      const uint32_t am = A[par], am_n = A[par_n];
      if constexpr (tap == 0) {                        // a new halo buffer: nothing of it could be fetched before the barrier
#pragma unroll
        for (int j = 0; j < XD - 1; ++j) xf[j] = xread(j, 0, am);
      }
      static_for<NP>([&](auto jc) {
        constexpr int jj = decltype(jc)::value;
        // rotation: patch jj of tap t lives in xf[(jj + t * NP) % XD]; NP % XD == 1 for XD = 3, 4 (25 = 24 + 1), so the slot of
        // the next tap's patch j is (NP + j + t * NP) % XD = the slot "patch NP + j of this tap" would take
        constexpr int R = (tap * NP) % XD;
        if constexpr (jj + XD - 1 < NP) xf[(jj + XD - 1 + R) % XD] = xread(jj + XD - 1, tap, am);
        else if constexpr (tap < 8) xf[(jj + XD - 1 + R) % XD] = xread(jj + XD - 1 - NP, tap + 1, am_n);   // the next step's first ones
        static_for<2>([&](auto ic) { mfma(ic, jc, wf[tap % 3][decltype(ic)::value], xf[(jj + R) % XD]); });
      });
      if constexpr (tap == 8) {
        wait_vmcnt<2>();                               // everything but the two youngest weight loads: the next halo has landed
        __builtin_amdgcn_s_barrier();
      }
    });
    A[0] ^= (uint32_t)kHaloB;
    A[1] ^= (uint32_t)kHaloB;
  }
#if defined(__HIP_DEVICE_COMPILE__)
  asm volatile("s_nop 15\n\ts_nop 15");              // the last asm MFMAs' D registers: 12 wait states before any other reader
#endif
  if (a.debug & 8) return;

  // ---- epilogue (the final barrier of the loop has passed: the halo buffers are free; the weight prefetches of the two
  // steps beyond the end land in registers nobody reads)
  t20v2_epilogue<RD>(a, smem, acc, wave, lane, b, y0, x0, n0);
}

// ---------------------------------------------------------------------------------------------------------------------
// Stride-2 form (round 3): the four DownSample.conv0 layers of YOLOv3-SPP (reference models/yolov3_spp.py:26-27; 320 -> 160 ..
// 40 -> 20 at a 640 input) on the same 20x20 OUTPUT tiles, 128 couts, 4 waves x 32 couts x 25 patches, register-resident weight
// stream and epilogue as conv3x3_t20v2_kernel.  What differs is the pixel operand: the 41x41 input window of a tile is staged as
// four PARITY PLANES of 21x21 pixels (input row 2 oy + kh - 1 is even for kh = 1, odd for kh = 0 / 2; same for columns), on which
// every filter tap is a plain stride-1 window shifted by (kh == 2, kw == 2) - so a fragment's LDS address is again per-lane base +
// compile-time constant (plane pitch 21 pixels; the slot swizzle chunk ^ 2 * (plane row & 1) is conflict-free for any pitch):
//   plane 0  odd rows  x odd cols   taps (0,0) (0,2) (2,0) (2,2)      plane 1  even rows x odd cols   taps (1,0) (1,2)
//   plane 2  odd rows  x even cols  taps (0,1) (2,1)                  plane 3  even rows x even cols  tap  (1,1)
// A 32-channel chunk of a plane is 441 pixel rows of 64 B (28 KB); two plane buffers alternate (63 KB of LDS with the source
// table: two workgroups per CU): the next plane's LDS-DMA (7 pieces per wave, per-lane gather addresses at pixel stride 2) is
// issued behind the barrier that ends the plane before, and lands while the current plane's 1 - 4 taps are multiplied.  The
// one-tap plane leaves 50 MFMAs per wave to cover a 28 KB fetch - the second workgroup of the CU covers the rest.
constexpr int kPW = 21, kPlanePix = kPW * kPW, kPlaneB = 28 * 1024;
constexpr int kS2Tab = 7 * 256 * 4;                  // per-thread packed plane coordinates of its 7 pieces
static_assert(2 * 32 * kV2Pitch <= 2 * kPlaneB, "epilogue buffers reuse the plane buffers");

// NBUF = 2 (shipped): two plane buffers, the next plane in flight (63 KB: two workgroups per CU).  NBUF = 4 (a buffer per plane, up to
// three planes in flight, 119 KB, one workgroup per CU) was measured for the grids that give a CU one workgroup anyway and is NOT
// instantiated: 0.133 vs 0.125 ms on the 40 -> 20 layer at 32 images - with one wave per SIMD the MFMA stream itself, not the plane
// fetch, is what stalls, and the gather kernel's 16-wave tiles do as well there; such grids stay with it (launch rule below).
template <int XD, int RD, int NBUF>
__global__ __launch_bounds__(256, 2) void conv3x3s2_t20_kernel(const ConvArgs a) {
  constexpr int NP = 25, PPW = 7, AHEAD = NBUF - 1;         // planes requested ahead of the one being multiplied
  static_assert(NBUF == 2 || NBUF == 4, "plane buffers");
  // the nine steps of a chunk: plane, filter tap (kh * 3 + kw), window shift
  constexpr int ST_PLANE[9] = {0, 0, 0, 0, 1, 1, 2, 2, 3};
  constexpr int ST_TAP[9] = {0, 2, 6, 8, 3, 5, 1, 7, 4};
  constexpr int ST_SY[9] = {0, 0, 1, 1, 0, 0, 0, 1, 0};
  constexpr int ST_SX[9] = {0, 1, 0, 1, 0, 1, 0, 0, 0};
  constexpr int PL_RY[4] = {1, 0, 1, 0}, PL_RX[4] = {1, 1, 0, 0};     // odd rows / odd cols
  __shared__ __attribute__((aligned(16))) char smem[NBUF * kPlaneB + kS2Tab];
  uint32_t* const tab = reinterpret_cast<uint32_t*>(smem + NBUF * kPlaneB);

  YOLO_BLOCK_STAMP(a);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const YoloConvDesc& d = a.d;
  const int tiles_x = (d.wo + kT20 - 1) / kT20, tiles_y = (d.ho + kT20 - 1) / kT20;
  int b, y0, x0, n0;
  {
    int swz = xcd_swizzle(blockIdx.x, gridDim.x);
    n0 = (swz % a.n_tiles) * 128;
    swz /= a.n_tiles;
    x0 = (swz % tiles_x) * kT20;
    swz /= tiles_x;
    y0 = (swz % tiles_y) * kT20;
    b = swz / tiles_y;
  }
  const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)a.x, 0, a.x_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void*)a.w, 0, a.w_bytes, 0x00020000);

  // ---- plane pieces of this wave: piece (it * 4 + wave) = LDS rows [16 piece, +16); lane -> (row lane >> 2, physical slot lane & 3).
  // Table entry: input position of plane pixel (r, c) in the even / even plane, (2 (y0 + r)) << 16 | (r & 1) << 15 | 2 (x0 + c);
  // the other planes lie one input row / column before it.  0xffffffff: beyond the plane.
#pragma unroll
  for (int it = 0; it < PPW; ++it) {
    const int hp = (it * 4 + wave) * 16 + (lane >> 2);
    const int r = hp / kPW, c = hp - r * kPW;
    tab[it * 256 + tid] = hp < kPlanePix ? ((uint32_t)(2 * (y0 + r)) << 16) | ((uint32_t)(r & 1) << 15) | (uint32_t)(2 * (x0 + c)) : 0xffffffffu;
  }
  const uint32_t img_base = (uint32_t)(b * d.h * d.w);
  auto issue_plane = [&](int buf, int c, int ry, int rx_) {
#pragma unroll
    for (int it = 0; it < PPW; ++it) {
      const uint32_t pk = tab[it * 256 + tid];
      const int iy = (int)(pk >> 16) - ry, ix = (int)(pk & 0x7fffu) - rx_;
      const int chunk = (lane & 3) ^ (int)((pk >> 14) & 2u);
      const bool ok = pk != 0xffffffffu && (unsigned)iy < (unsigned)d.h && (unsigned)ix < (unsigned)d.w;
      const uint32_t off = ok ? ((img_base + (uint32_t)(iy * d.w + ix)) * (uint32_t)d.in_c_total + (uint32_t)(d.in_c_offset + chunk * 8)) * 2u : kOobOffset;
      lds_dma16s(rx, smem + buf * kPlaneB + (it * 4 + wave) * 1024, off, (uint32_t)c * 64u);
    }
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("" ::: "memory");                     // the weight loads that follow stay behind these DMAs (counted waits below)
#endif
  };

  // ---- fragment addressing: lane = (patch pixel c16 = (dy, dx), 8-channel group q); base by parity of the plane row (dy + shift)
  const int c16 = lane & 15, q = lane >> 4;
  const int dy = c16 >> 2, dx = c16 & 3;
  uint32_t A[2];
#pragma unroll
  for (int par = 0; par < 2; ++par) A[par] = (uint32_t)((dy * kPW + dx) * 64 + ((q ^ (((dy + par) & 1) << 1)) << 4));
#if defined(__HIP_DEVICE_COMPILE__)
  asm volatile("" : "+v"(A[0]), "+v"(A[1]));
#endif
  const uint32_t wv = (uint32_t)(((n0 + wave * 32 + c16) * d.kpad + q * 8) * 2);
  const uint32_t wfrag = (uint32_t)(16 * d.kpad * 2);
  auto wload = [&](int c, int tap, int i) -> bf16x8 {
    return __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rw, wv, (uint32_t)((tap * d.cin + c * 32) * 2) + i * wfrag, 0));
  };
  auto xread = [&](int jj, int st, uint32_t am) -> bf16x8 {     // patch jj under step st's plane buffer and window shift
    const char* const p = smem + am;
    return *reinterpret_cast<const bf16x8*>(p + (ST_PLANE[st] % NBUF) * kPlaneB + ((4 * (jj / 5) + ST_SY[st]) * kPW + 4 * (jj % 5) + ST_SX[st]) * 64);
  };

  f32x4 acc[2][NP];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < NP; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  auto mfma = [&](auto ic, auto jc, const bf16x8& wa, const bf16x8& xb) {
    constexpr int i = decltype(ic)::value, jj = decltype(jc)::value;
    f32x4& t = acc[i][jj];
    (void)t;
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(t) : "v"(wa), "v"(xb));
#endif
  };

  const int nch = d.cin / 32;
  wait_lds();                                          // (the table entries are this thread's own)
  static_for<AHEAD>([&](auto kc) {                     // planes 0 .. AHEAD - 1 of chunk 0
    constexpr int k = decltype(kc)::value;
    issue_plane(k % NBUF, 0, PL_RY[k], PL_RX[k]);
  });
  bf16x8 wf[3][2], xf[XD];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    wf[0][i] = wload(0, ST_TAP[0], i);
    wf[1][i] = wload(0, ST_TAP[1], i);
  }
  wait_vmcnt<4 + 7 * (AHEAD - 1)>();                   // plane 0 of chunk 0 has landed (later planes and the four weight loads may still fly)
  __builtin_amdgcn_s_barrier();

  for (int c = 0; c < nch; ++c) {
    static_for<9>([&](auto sc) {
      constexpr int st = decltype(sc)::value, pl = ST_PLANE[st];
      constexpr bool first = st == 0 || ST_PLANE[st > 0 ? st - 1 : 0] != pl, last = st == 8 || ST_PLANE[st < 8 ? st + 1 : 8] != pl;
      if constexpr (first) {
        // behind the barrier that ended the plane before: the buffer it used takes the plane AHEAD planes on (beyond the last
        // chunk dummy ones, so that the counted waits stay uniform; they read inside the input buffer or return zeros)
        constexpr int npl = (pl + AHEAD) & 3;
        issue_plane(npl % NBUF, pl + AHEAD > 3 ? c + 1 : c, PL_RY[npl], PL_RX[npl]);
      }
      {
        constexpr int s2 = (st + 2) % 9;
        const int c2 = c + (st + 2) / 9;
#pragma unroll
        for (int i = 0; i < 2; ++i) wf[(st + 2) % 3][i] = wload(c2, ST_TAP[s2], i);
      }
      const uint32_t am = A[ST_SY[st]];
      if constexpr (first) {                           // a new plane buffer: nothing of it could be fetched before the barrier
#pragma unroll
        for (int j = 0; j < XD - 1; ++j) xf[(j + (st * NP) % XD) % XD] = xread(j, st, am);
      }
      static_for<NP>([&](auto jc) {
        constexpr int jj = decltype(jc)::value;
        constexpr int R = (st * NP) % XD;              // patch jj of step st lives in xf[(jj + st * NP) % XD] (NP % XD == 1)
        if constexpr (jj + XD - 1 < NP) xf[(jj + XD - 1 + R) % XD] = xread(jj + XD - 1, st, am);
        else if constexpr (!last) xf[(jj + XD - 1 + R) % XD] = xread(jj + XD - 1 - NP, st < 8 ? st + 1 : 8, A[ST_SY[st < 8 ? st + 1 : 8]]);   // the next step's first ones (same plane)
        static_for<2>([&](auto ic) { mfma(ic, jc, wf[st % 3][decltype(ic)::value], xf[(jj + R) % XD]); });
      });
      if constexpr (last) {
        // The next plane has landed when at most the youngest AHEAD - 1 plane requests and the weight loads of the coming steps are
        // still out (behind its DMAs this plane issued two weight loads per step; two steps' worth - one step's for the one-tap
        // plane - belong to steps not yet multiplied).  Vector-memory operations retire in issue order.
        constexpr int nsteps = pl == 0 ? 4 : (pl == 3 ? 1 : 2);
        wait_vmcnt<(nsteps >= 2 ? 4 : 2) + 7 * (AHEAD - 1)>();
        __builtin_amdgcn_s_barrier();
      }
    });
  }
#if defined(__HIP_DEVICE_COMPILE__)
  asm volatile("s_nop 15\n\ts_nop 15");              // the last asm MFMAs' D registers: 12 wait states before any other reader
#endif
  if (a.debug & 8) return;
  wait_vmcnt<0>();                                     // (the dummy plane behind the last chunk is still landing in buffer 0)
  __builtin_amdgcn_s_barrier();
  t20v2_epilogue<RD>(a, smem, acc, wave, lane, b, y0, x0, n0);
}

int launch_t20s2(const ConvArgs& a, hipStream_t s) {
  ConvArgs b = a;
  b.n_tiles = a.d.cout / 128;
  const long grid = (long)a.d.n * ((a.d.ho + kT20 - 1) / kT20) * ((a.d.wo + kT20 - 1) / kT20) * b.n_tiles;
  if (grid > 0x7fffffffL) return yolo_set_error(YOLO_E_UNSUPPORTED, "conv grid too large");
  if (pick_only("t20s2<400px x 128 couts, 4 waves, parity planes> grid %ld", grid)) return 0;
  hipLaunchKernelGGL((conv3x3s2_t20_kernel<3, 3, 2>), dim3((unsigned)grid), dim3(256), 0, s, b);
  return yolo_check_launch("yolo_conv2d_fwd(t20s2)");
}

int launch_t20v2(const ConvArgs& a, hipStream_t s) {
  ConvArgs b = a;
  b.n_tiles = a.d.cout / 128;
  const long grid = (long)a.d.n * ((a.d.h + kT20 - 1) / kT20) * ((a.d.w + kT20 - 1) / kT20) * b.n_tiles;
  if (grid > 0x7fffffffL) return yolo_set_error(YOLO_E_UNSUPPORTED, "conv grid too large");
  if (pick_only("t20v2<400px x 128 couts, 4 waves> grid %ld", grid)) return 0;
  if (a.debug & 131072) {     // A/B: pieces of <= 256 workgroups, one after the other (one workgroup of this launch list per CU)
    for (long off = 0; off < grid; off += 256) {
      b.blk_off = (int)off;
      b.blk_total = (int)grid;
      hipLaunchKernelGGL((conv3x3_t20v2_kernel<3, 3>), dim3((unsigned)(grid - off < 256 ? grid - off : 256)), dim3(256), 0, s, b);
    }
    return yolo_check_launch("yolo_conv2d_fwd(t20v2 pieces)");
  }
  if (a.debug & 64) hipLaunchKernelGGL((conv3x3_t20v2_kernel<3, 1>), dim3((unsigned)grid), dim3(256), 0, s, b);   // A/B: residual one pair ahead
  else hipLaunchKernelGGL((conv3x3_t20v2_kernel<3, 3>), dim3((unsigned)grid), dim3(256), 0, s, b);
  return yolo_check_launch("yolo_conv2d_fwd(t20v2)");
}

template <int CT, int NWM>
int launch_t20(const ConvArgs& a, hipStream_t s) {
  ConvArgs b = a;
  b.n_tiles = a.d.cout / CT;
  const long grid = (long)a.d.n * ((a.d.h + kT20 - 1) / kT20) * ((a.d.w + kT20 - 1) / kT20) * b.n_tiles;
  if (grid > 0x7fffffffL) return yolo_set_error(YOLO_E_UNSUPPORTED, "conv grid too large");
  if (pick_only("t20<%d couts,%d pixel groups> grid %ld", CT, NWM, grid)) return 0;
  hipLaunchKernelGGL((conv3x3_t20_kernel<CT, NWM>), dim3((unsigned)grid), dim3(64 * (CT / 64) * NWM), 0, s, b);
  return yolo_check_launch("yolo_conv2d_fwd(t20)");
}

}  // namespace

// Returns 1 when this kernel does not take the layer (the caller goes on to the other kernels).
// force: 0 = the shipped rule (maps that 20x20 tiles cover >= 90 % and whose workgroup count fills the chip),
//        1 = every layer the kernel can compute (tests, A/B runs).
int yolo_conv::launch_t20_3x3(const ConvArgs& a, int force, hipStream_t s) {
  const YoloConvDesc& d = a.d;
  if (d.ksize != 3 || (d.stride != 1 && d.stride != 2) || d.pad != 1 || d.upsample2x || d.out_dtype != YOLO_DT_BF16) return 1;
  if (d.cin % 32 != 0 || d.cout % 128 != 0 || d.act == YOLO_ACT_SWISH) return 1;   // (the epilogue's min / max form has no swish)
  const size_t npix = (size_t)d.n * d.ho * d.wo;    // the epilogue addresses y / residual / pre-add copy with 32-bit byte offsets
  if (npix * d.out_c_total * 2 >= kOobOffset || (a.res && npix * d.res_c_total * 2 >= kOobOffset) ||
      (a.aux && npix * d.aux_c_total * 2 >= kOobOffset))
    return 1;
  if (d.out_c_offset % 8 || d.out_c_total % 8 || (a.res && (d.res_c_offset % 8 || d.res_c_total % 8)) ||
      (a.aux && (d.aux_c_offset % 8 || d.aux_c_total % 8)))
    return 1;
  const long tiles = (long)d.n * ((d.ho + kT20 - 1) / kT20) * ((d.wo + kT20 - 1) / kT20);
  if (d.stride == 2) {        // parity-plane form; the table packs input coordinates into 15 / 16 bits
    if (2 * (d.wo + kT20) >= 0x8000 || 2 * (d.ho + kT20) >= 0x10000 || (force & 8)) return 1;
    // two workgroups per CU are what covers the plane fetches: grids below that stay with the gather kernel (measured at 16 / 32
    // images: 80 -> 40 with 256 workgroups 0.067 vs 0.064 ms, 40 -> 20 with 256: 0.125 vs 0.126, with 128: 0.108 vs 0.066)
    if (!force && ((double)d.n * d.ho * d.wo < 0.9 * 400.0 * tiles || tiles * (d.cout / 128) < 2 * launch_cus())) return 1;
    return launch_t20s2(a, s);
  }
  if (!force) {
    if ((double)d.n * d.h * d.w < 0.9 * 400.0 * tiles) return 1;          // partial tiles idle lanes
    // The second form is the shipped one: 400 pixels x 128 couts per workgroup, two workgroups per CU.  It needs enough workgroups
    // for half the chip (the other sub-batch stream's launch fills the rest): 16 images of the 20x20 maps give 128.
    if (tiles * (d.cout / 128) < launch_cus() / 2) return 1;
    return launch_t20v2(a, s);
  }
  if (force & 4) return launch_t20v2(a, s);
  // first form (kept for A/B runs): 256 couts per workgroup halve the halo traffic; 128 double the workgroup count
  if (d.cout % 256 == 0 && (tiles * (d.cout / 256) >= 224 || (force & 2))) return launch_t20<256, 1>(a, s);
  return launch_t20<128, 4>(a, s);
}
