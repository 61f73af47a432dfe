// 3x3 / pad-1 convolution on 20x20 OUTPUT tiles for gfx950 (MI355X), stride 1 (conv3x3_t20v2_kernel) and stride 2
// (conv3x3s2_t20_kernel): same contract and numerics as conv_igemm.hip / conv3x3_halo.hip (bf16 NHWC in, fp32 accumulate on
// v_mfma_f32_16x16x32_bf16, fused epilogue).
//
// Why.  The feature maps of a 640x640 (or 320, 416 -> 20-multiples) YOLOv3 are 20 * 2^k pixels wide, so with 20x20-pixel tiles
// every heavy 3x3 layer of a batch gives images x (h/20)(w/20) x cout/128 workgroups of EQUAL work (400 pixels x 128 couts x K):
// whole multiples of the chip at two workgroups per CU, no partial last round (the 256x256 gather tiles and the 16x16 halo
// tiles left 22 % of the chip idle, profiles/r01_block_timeline.md).  Each input pixel is staged once per 32-channel chunk
// instead of once per filter tap (L2 -> LDS traffic per FLOP 2.2x below the gather kernel's).
//
//   tile      20 x 20 output pixels of one image = 25 patches of 4 x 4 pixels; a patch is one 16-pixel MFMA fragment
//             (lane c16 -> pixel (c16 >> 2, c16 & 3) of the patch), so a fragment's LDS address is
//             per-lane base + compile-time constant for every patch and every filter tap (ds_read immediates, no VALU)
//   LDS       stride 1: halo [2][22*22 pixel rows of 64 B] (32 channels per chunk, double-buffered across chunks);
//             stride 2: two parity-plane buffers of 21*21 pixel rows; rows are swizzled by physical 16-byte slot =
//             chunk ^ 2*(row y & 1): conflict-free for ds_read_b128's lane groups at every tap offset and any row pitch
//             (tools/micro/lds_groups.hip)
//   waves     4, each 32 couts (two fragments) x all 25 patches = 50 MFMAs per (chunk, tap) step, 200 accumulator registers;
//             its weight fragments are nobody else's: straight from L2 into registers, two steps ahead, three register sets
//   epilogue  patch pairs staged as [32 pixels][128 couts] fp32 in LDS -> 16-byte buffer stores over whole 256-byte pixel rows
//             (residual read, pre-add copy, store)
#include "conv_common.h"

using namespace yolo_conv;

namespace {

constexpr int kT20 = 20, kHW2 = 22, kHP = kHW2 * kHW2;       // tile edge, halo edge, halo pixels
constexpr int kHaloB = 32 * 1024;                             // 31 pieces of 16 rows, padded to 32 (8 per wave)

// ---------------------------------------------------------------------------------------------------------------------
// The shipped form: 4 waves, 400 pixels x 128 couts per workgroup, TWO workgroups per CU (72 KB of LDS, 256 registers).
// (Round 2's first form - weights through a 3-slot LDS ring, 256 couts on 4 waves of 512 registers or 128 couts on 8 waves, one
// workgroup per CU, 1.07-1.27 PFLOP/s where this one reaches 1.11-1.39 - was deleted in round 3; DESIGN.md appendix.)
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
template <int RD, bool LEAKY>
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
  auto act4 = [&](f32x4 v) -> f32x4 {                   // (LEAKY: the launcher's choice for LeakyReLU(0.1) layers - all of SPP's but the heads)
    if (LEAKY) return leaky4(v);
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
template <int XD, int RD, bool LEAKY>
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
    (void)t;                                             // (the host pass sees no asm)
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(t) : "v"(wa), "v"(xb));
#endif
  };

  const int nch = d.cin / 32;
  __builtin_assume(nch >= 1);                          // (launcher: cin % 32 == 0, cin > 0) - without it the loop guard makes the compiler zero the 200 accumulators twice
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
  t20v2_epilogue<RD, LEAKY>(a, smem, acc, wave, lane, b, y0, x0, n0);
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
// PAIR (round 4, layers with an even number of 32-channel chunks): the steps of TWO chunks are interleaved plane by plane - plane 0 of
// chunk 2 p, plane 0 of chunk 2 p + 1, plane 1 of chunk 2 p, ... - because the chunks 2 p and 2 p + 1 of a pixel are the two 64-byte
// halves of ONE 128-byte line: fetched a whole chunk (four plane blocks) apart, the second half came from HBM again (the counters read
// 1.8 x the layer's input on the 320 -> 160 layer, whose 0.19 ms were exactly that traffic at 5.5 TB/s); fetched one plane block apart
// it is an L2 hit.  Same planes, taps, accumulators and weight stream: only the order of the 18 (chunk, tap) steps of a pair changes.
template <int XD, int RD, int NBUF, bool LEAKY, bool PAIR = false>
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
  __builtin_assume(nch >= 1);                          // (launcher: cin % 32 == 0, cin > 0) - without it the loop guard makes the compiler zero the 200 accumulators twice
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

  if constexpr (!PAIR) {
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
  } else {
    static_assert(NBUF == 2, "pair order: two plane buffers");
    // the 18 steps of a chunk pair: blocks (plane, chunk of the pair) = (0,0) (0,1) (1,0) (1,1) (2,0) (2,1) (3,0) (3,1); block k uses
    // plane buffer k % 2 and requests block k + 1's plane behind the barrier that ended block k - 1
    constexpr int PB_PLANE[8] = {0, 0, 1, 1, 2, 2, 3, 3}, PB_CH[8] = {0, 1, 0, 1, 0, 1, 0, 1}, PB_FIRST[9] = {0, 4, 8, 10, 12, 14, 16, 17, 18};
    constexpr int PS_BLOCK[18] = {0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 7};
    constexpr int PS_TAP[18] = {0, 2, 6, 8, 0, 2, 6, 8, 3, 5, 3, 5, 1, 7, 1, 7, 4, 4};
    constexpr int PS_SY[18] = {0, 0, 1, 1, 0, 0, 1, 1, 0, 0, 0, 0, 0, 1, 0, 1, 0, 0};
    constexpr int PS_SX[18] = {0, 1, 0, 1, 0, 1, 0, 1, 0, 1, 0, 1, 0, 0, 0, 0, 0, 0};
    auto xread2 = [&](int jj, int s_, uint32_t am) -> bf16x8 {
      const char* const p = smem + am;
      return *reinterpret_cast<const bf16x8*>(p + (PS_BLOCK[s_] & 1) * kPlaneB + ((4 * (jj / 5) + PS_SY[s_]) * kPW + 4 * (jj % 5) + PS_SX[s_]) * 64);
    };
    const int npairs = nch >> 1;
    for (int cp = 0; cp < npairs; ++cp) {
      static_for<18>([&](auto sc) {
        constexpr int st = decltype(sc)::value, blk = PS_BLOCK[st];
        constexpr bool first = st == PB_FIRST[blk], last = st + 1 == PB_FIRST[blk + 1];
        if constexpr (first) {
          // the buffer of the block before takes the NEXT block's plane (behind the last pair a dummy one: the counted waits stay uniform)
          constexpr int nb = (blk + 1) & 7;
          issue_plane(nb & 1, 2 * (blk == 7 ? cp + 1 : cp) + PB_CH[nb], PL_RY[PB_PLANE[nb]], PL_RX[PB_PLANE[nb]]);
        }
        {
          constexpr int s2 = (st + 2) % 18;
          const int c2 = 2 * (cp + (st + 2) / 18) + PB_CH[PS_BLOCK[s2]];
#pragma unroll
          for (int i = 0; i < 2; ++i) wf[(st + 2) % 3][i] = wload(c2, PS_TAP[s2], i);
        }
        const uint32_t am = A[PS_SY[st]];
        if constexpr (first) {                           // a new plane buffer: nothing of it could be fetched before the barrier
#pragma unroll
          for (int j = 0; j < XD - 1; ++j) xf[(j + (st * NP) % XD) % XD] = xread2(j, st, am);
        }
        static_for<NP>([&](auto jc) {
          constexpr int jj = decltype(jc)::value;
          constexpr int R = (st * NP) % XD;              // patch jj of step st lives in xf[(jj + st * NP) % XD] (NP % XD == 1)
          if constexpr (jj + XD - 1 < NP) xf[(jj + XD - 1 + R) % XD] = xread2(jj + XD - 1, st, am);
          else if constexpr (!last) xf[(jj + XD - 1 + R) % XD] = xread2(jj + XD - 1 - NP, st < 17 ? st + 1 : 17, A[PS_SY[st < 17 ? st + 1 : 17]]);   // the next step's first ones (same block)
          static_for<2>([&](auto ic) { mfma(ic, jc, wf[st % 3][decltype(ic)::value], xf[(jj + R) % XD]); });
        });
        if constexpr (last) {
          constexpr int nsteps = PB_FIRST[blk + 1] - PB_FIRST[blk];
          wait_vmcnt<(nsteps >= 2 ? 4 : 2)>();           // the next block's plane has landed; the weight loads of the coming steps may still fly
          __builtin_amdgcn_s_barrier();
        }
      });
    }
  }
#if defined(__HIP_DEVICE_COMPILE__)
  asm volatile("s_nop 15\n\ts_nop 15");              // the last asm MFMAs' D registers: 12 wait states before any other reader
#endif
  if (a.debug & 8) return;
  wait_vmcnt<0>();                                     // (the dummy plane behind the last chunk is still landing in buffer 0)
  __builtin_amdgcn_s_barrier();
  t20v2_epilogue<RD, LEAKY>(a, smem, acc, wave, lane, b, y0, x0, n0);
}

int launch_t20s2(const ConvArgs& a, hipStream_t s) {
  ConvArgs b = a;
  b.n_tiles = a.d.cout / 128;
  const long grid = (long)a.d.n * ((a.d.ho + kT20 - 1) / kT20) * ((a.d.wo + kT20 - 1) / kT20) * b.n_tiles;
  if (grid > 0x7fffffffL) return yolo_set_error(YOLO_E_UNSUPPORTED, "conv grid too large");
  if (pick_only("t20s2<400px x 128 couts, 4 waves, parity planes> grid %ld", grid)) return 0;
  // (chunk pairs where the layer has an even number of 32-channel chunks: every stride-2 layer of YOLOv3-SPP; YOLO_CONV_DEBUG bit
  // 16777216 keeps the chunk-by-chunk order for A/Bs)
  const bool pair = (a.d.cin / 32) % 2 == 0 && !(a.debug & 16777216);
  if (pair) {
    if (a.d.act == YOLO_ACT_LEAKY01) hipLaunchKernelGGL((conv3x3s2_t20_kernel<3, 3, 2, true, true>), dim3((unsigned)grid), dim3(256), 0, s, b);
    else hipLaunchKernelGGL((conv3x3s2_t20_kernel<3, 3, 2, false, true>), dim3((unsigned)grid), dim3(256), 0, s, b);
  } else if (a.d.act == YOLO_ACT_LEAKY01) hipLaunchKernelGGL((conv3x3s2_t20_kernel<3, 3, 2, true>), dim3((unsigned)grid), dim3(256), 0, s, b);
  else hipLaunchKernelGGL((conv3x3s2_t20_kernel<3, 3, 2, false>), dim3((unsigned)grid), dim3(256), 0, s, b);
  return yolo_check_launch("yolo_conv2d_fwd(t20s2)");
}

int launch_t20v2(const ConvArgs& a, hipStream_t s) {
  ConvArgs b = a;
  b.n_tiles = a.d.cout / 128;
  const long grid = (long)a.d.n * ((a.d.h + kT20 - 1) / kT20) * ((a.d.w + kT20 - 1) / kT20) * b.n_tiles;
  if (grid > 0x7fffffffL) return yolo_set_error(YOLO_E_UNSUPPORTED, "conv grid too large");
  if (pick_only("t20v2<400px x 128 couts, 4 waves> grid %ld", grid)) return 0;
  if (a.d.act == YOLO_ACT_LEAKY01) hipLaunchKernelGGL((conv3x3_t20v2_kernel<3, 3, true>), dim3((unsigned)grid), dim3(256), 0, s, b);
  else hipLaunchKernelGGL((conv3x3_t20v2_kernel<3, 3, false>), dim3((unsigned)grid), dim3(256), 0, s, b);
  return yolo_check_launch("yolo_conv2d_fwd(t20v2)");
}

}  // namespace

// Returns 1 when this kernel does not take the layer (the caller goes on to the other kernels).
// force: 0 = the shipped rules (maps that 20x20 tiles cover >= 90 % and whose workgroup count fills the chip),
//        1 = every layer the kernels can compute (tests, A/B runs).
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
    if (2 * (d.wo + kT20) >= 0x8000 || 2 * (d.ho + kT20) >= 0x10000) return 1;
    // two workgroups per CU are what covers the plane fetches: grids below that stay with the gather kernel (measured at 16 / 32
    // images: 80 -> 40 with 256 workgroups 0.067 vs 0.064 ms, 40 -> 20 with 256: 0.125 vs 0.126, with 128: 0.108 vs 0.066)
    if (!force && ((double)d.n * d.ho * d.wo < 0.9 * 400.0 * tiles || tiles * (d.cout / 128) < 2 * launch_cus())) return 1;
    return launch_t20s2(a, s);
  }
  if (!force) {
    if ((double)d.n * d.h * d.w < 0.9 * 400.0 * tiles) return 1;          // partial tiles idle lanes
    // 400 pixels x 128 couts per workgroup, two workgroups per CU: it needs enough workgroups for half the chip (the other
    // pipeline's launch fills the rest): 16 images of the 20x20 maps give 128.
    if (tiles * (d.cout / 128) < launch_cus() / 2) return 1;
    return launch_t20v2(a, s);
  }
  return launch_t20v2(a, s);
}
