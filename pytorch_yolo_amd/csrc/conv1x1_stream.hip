// Weight-stationary 1x1 convolution for the large feature maps of gfx950 (MI355X): same contract and numerics as
// conv_igemm.hip (bf16 NHWC in, fp32 accumulate on v_mfma_f32_16x16x32_bf16, bias + activation, bf16 out).
//
// The residual units' 1x1 convs on the 160x160 / 80x80 maps (128 -> 64, 256 -> 128) are not MFMA work: K is 128 / 256, the layer
// reads 2 K and writes N bytes per pixel and is bound by how fast pixels stream through a CU (HBM roofline).  The
// tiled implicit GEMM pays a prologue, a K loop of 4-8 barriers and an epilogue per 256 pixels.  Here the weights never move:
//   * a wave owns N/4 couts (one or two 16-row fragments) and keeps ITS slice of W for the whole K in registers (K/32 fragments
//     each), loaded once per workgroup; workgroups are persistent (two per CU) and walk 128-pixel tiles;
//   * a tile's pixels come in by LDS-DMA, all K channels at once ([K/32 chunks][128 pixels][64 B], swz32 rows), one wait + one
//     barrier per tile; the two workgroups of a CU alternate between loading and multiplying;
//   * the result goes registers -> LDS (bf16, [128 pixels][N couts]) -> whole 2N-byte pixel rows by buffer stores.
#include "conv_common.h"

using namespace yolo_conv;

namespace {

// act(v) without a per-element switch on the runtime activation code: min(max(v, lo), hi) with lo = 0 (ReLU, ReLU6) or slope * v
// (LeakyReLU: 0.1, none: 1) covers everything but swish, which takes its own copy of the staging loop under one uniform branch
// (apply_act inside the unrolled staging loops compiled to a branch tree per element: twice the MFMA time of a tile).
struct ActParams {
  bool floor0, swish;
  float slope, hi;
  __device__ explicit ActParams(int act)
      : floor0(act == YOLO_ACT_RELU || act == YOLO_ACT_RELU6), swish(act == YOLO_ACT_SWISH), slope(act == YOLO_ACT_LEAKY01 ? 0.1f : 1.f),
        hi(act_hi(act)) {}
  __device__ __forceinline__ float plain(float v) const { return fminf(fmaxf(v, floor0 ? 0.f : slope * v), hi); }
};

template <int N, int K>   // couts (64 or 128), input channels (128 or 256)
__global__ __launch_bounds__(256, 2) void conv1x1_stream_kernel(const ConvArgs a, int n_tiles_px) {
  constexpr int P = 128, NF = N / 64, KC = K / 32;            // pixels per tile, cout fragments per wave, 32-channel chunks
  constexpr int XB = KC * P * 64;                              // bytes of a pixel tile
  constexpr int SP = N * 2 + 16;                               // staging pitch (bytes): a pixel's couts + 16 B
  static_assert((N == 64 || N == 128) && (K == 128 || K == 256), "shape");
  __shared__ __attribute__((aligned(16))) char smem[XB > P * SP ? XB : P * SP];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const YoloConvDesc& d = a.d;
  const ActParams ap(d.act);
  const int c16 = lane & 15, q = lane >> 4;
  const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)a.x, 0, a.x_bytes, 0x00020000);
  const uint32_t y_pitch = (uint32_t)d.out_c_total * 2u;
  const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc(a.y, 0, (uint32_t)a.M * y_pitch, 0x00020000);

  // ---- this wave's weights: fragment f = couts [wave * 16 NF + 16 f, +16), k-step kc: lane (row c16, 8-channel chunk q)
  bf16x8 wreg[NF][KC];
  f32x4 bv[NF];
#pragma unroll
  for (int f = 0; f < NF; ++f) {
    const int row = wave * 16 * NF + f * 16 + c16;
#pragma unroll
    for (int kc = 0; kc < KC; ++kc) wreg[f][kc] = *reinterpret_cast<const bf16x8*>(a.w + (long)row * d.kpad + kc * 32 + q * 8);
    bv[f] = *reinterpret_cast<const f32x4*>(a.bias + wave * 16 * NF + f * 16 + q * 4);
  }
  // ---- LDS-DMA geometry: piece = 16 pixel rows x 64 B of one chunk; this wave's pieces: (kc, pixel block pb) with
  // (kc * 8 + pb) % 4 == wave
  const int drow = lane >> 2, dchunk = (lane & 3) ^ swz32(lane >> 4);
  const uint32_t x_pitch = (uint32_t)d.in_c_total * 2u;
  const uint32_t lane_src = (uint32_t)drow * x_pitch + (uint32_t)(d.in_c_offset + dchunk * 8) * 2u;
  // fragment read: pixel row 16 j + c16 of chunk kc, slot q ^ swz32(c16 >> 2)
  const uint32_t xfrag = (uint32_t)(c16 * 64 + ((q ^ swz32(c16 >> 2)) << 4));
  // coalesced output phase: 2 N bytes per pixel = N / 8 lanes of 16 B; rows per wave-instruction
  constexpr int LPR = N / 8, RPI = 64 / LPR;
  const int orow = lane / LPR, ocol = lane % LPR;

  for (int t = blockIdx.x; t < n_tiles_px; t += gridDim.x) {
    const int p0 = t * P;
    // ---- pixels in: KC * 8 pieces, 2 KC per wave
#pragma unroll
    for (int i = 0; i < KC * 2; ++i) {
      const int piece = i * 4 + wave, kc = piece >> 3, pb = piece & 7;
      const int px = p0 + pb * 16 + drow;
      const uint32_t vo = px < a.M ? (uint32_t)(p0 + pb * 16) * x_pitch + lane_src : kOobOffset;
      if (!(a.debug & 1)) lds_dma16s(rx, smem + piece * 1024, vo, (uint32_t)kc * 64u);
    }
    wait_vmcnt<0>();
    __syncthreads();
    // ---- multiply: 8 patches x NF fragments x KC steps
    f32x4 acc[NF][8];
#pragma unroll
    for (int f = 0; f < NF; ++f)
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[f][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (!(a.debug & 4))
#pragma unroll
    for (int kc = 0; kc < KC; ++kc)
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const bf16x8 xf = *reinterpret_cast<const bf16x8*>(smem + (kc * 8 + j) * 1024 + xfrag);
#pragma unroll
        for (int f = 0; f < NF; ++f) acc[f][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wreg[f][kc], xf, acc[f][j], 0, 0, 0);
      }
    __syncthreads();                                  // every wave is done with the pixel tile: stage the result over it
    if (a.debug & 8) continue;                        // (timing-only ablations: bit 1 no DMA, 4 no MFMA, 8 no staging / stores)
    auto stage = [&](auto act) {
#pragma unroll
      for (int f = 0; f < NF; ++f)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          bf16x4 o;
#pragma unroll
          for (int e = 0; e < 4; ++e) o[e] = (bf16_t)act(acc[f][j][e] + bv[f][e]);
          *reinterpret_cast<bf16x4*>(smem + (j * 16 + c16) * SP + (wave * 16 * NF + f * 16 + q * 4) * 2) = o;
        }
    };
    if (ap.swish) stage([](float v) { return v / (1.f + expf(-v)); });
    else stage([&](float v) { return ap.plain(v); });
    __syncthreads();
#pragma unroll
    for (int i = 0; i < P / (4 * RPI); ++i) {
      const int row = (i * 4 + wave) * RPI + orow;
      const u32x4 v = *reinterpret_cast<const u32x4*>(smem + row * SP + ocol * 16);
      const uint32_t vo = p0 + row < a.M ? (uint32_t)(p0 + row) * y_pitch + (uint32_t)(d.out_c_offset + ocol * 8) * 2u : kOobOffset;
      __builtin_amdgcn_raw_buffer_store_b128(v, ry, vo, 0, 0);       // (no SGPR soffset: conv3x3_t20.hip, epilogue note)
    }
    // the staging area becomes the next pixel tile once every wave has READ its rows (LDS only: a __syncthreads() here would
    // also wait for the stores' acknowledgements - 2-3 us per tile, measured as half of the kernel)
    wait_lds();
    __builtin_amdgcn_s_barrier();
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// Round 5: the same kernel software-pipelined inside the workgroup, with tiles that divide the layer evenly.
// What the form above leaves on the table on the 80x80 maps of SPP-640 (256 -> 128, 204,800 pixels per 32 images: 0.035-0.037 ms =
// 4.2-4.4 TB/s of in + out traffic against ~5.5 reachable):
//   * 1,600 tiles of 128 pixels over 512 persistent workgroups = 3.125 each: 64 workgroups run a fourth tile while 448 idle (78 %);
//   * a workgroup has nothing in flight while it multiplies, stages and stores: per tile it pays an HBM round trip (DMA -> wait ->
//     barrier) that only the CU's second workgroup can cover.
// Here P = 80 pixels per tile (2,560 tiles = 5 per workgroup, exactly), two tile buffers per workgroup (2 x 40 KB: still two
// workgroups per CU), and the tile two steps ahead is requested as soon as its buffer is free - in the MIDDLE of a step, behind the
// barrier that says every wave has read its staged rows and BEFORE this step's stores are issued: vector-memory operations retire in
// issue order, so the wait for tile j + 1 then never waits for the acknowledgement of tile j's stores (2-3 us: header note above):
//     issue order   DMA(0) DMA(1) | DMA(2) st(0) | DMA(3) st(1) | DMA(4) st(2) ...
//     wait for DMA(j): all but the youngest NPW (j = 0), NPW + NST (j = 1), NPW + 2 NST (j >= 2) operations of this wave
// (NPW DMA pieces and NST stores per wave and tile; tiles beyond the last one are requested with out-of-range offsets - zeros written
// to a buffer nobody reads - so the counts stay uniform).  Same fragment layout, MFMA order and rounding points as the form above:
// outputs are bit-identical to it.
template <int N, int K, int P>   // couts (128), input channels (128 / 256 / 384), pixels per tile (a multiple of 16)
__global__ __launch_bounds__(256, 2) void conv1x1_stream2_kernel(const ConvArgs a, int n_tiles_px) {
  constexpr int NF = N / 64, KC = K / 32, PB = P / 16;        // cout fragments per wave, 32-channel chunks, 16-pixel blocks per tile
  constexpr int XB = KC * P * 64;                              // bytes of a pixel tile
  constexpr int SP = N * 2 + 16;                               // staging pitch (bytes): a pixel's couts + 16 B
  constexpr int NPW = KC * PB / 4;                             // DMA pieces per wave and tile
  constexpr int LPR = N / 8, RPI = 64 / LPR, NST = P / (4 * RPI);   // lanes per pixel row, rows per store instruction, stores per wave and tile
  constexpr int BUF = ((XB > P * SP ? XB : P * SP) + 1023) / 1024 * 1024;   // a tile buffer also takes the staged result of its tile
  static_assert(N == 128 && (KC * PB) % 4 == 0 && P % (4 * RPI) == 0 && 2 * BUF <= 80 * 1024, "shape");
  static_assert(NPW + 2 * NST <= 60, "vmcnt is a 6-bit counter");
  __shared__ __attribute__((aligned(16))) char smem[2 * BUF];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const YoloConvDesc& d = a.d;
  const ActParams ap(d.act);
  const int c16 = lane & 15, q = lane >> 4;
  const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)a.x, 0, a.x_bytes, 0x00020000);
  const uint32_t y_pitch = (uint32_t)d.out_c_total * 2u;
  const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc(a.y, 0, (uint32_t)a.M * y_pitch, 0x00020000);

  bf16x8 wreg[NF][KC];
  f32x4 bv[NF];
#pragma unroll
  for (int f = 0; f < NF; ++f) {
    const int row = wave * 16 * NF + f * 16 + c16;
#pragma unroll
    for (int kc = 0; kc < KC; ++kc) wreg[f][kc] = *reinterpret_cast<const bf16x8*>(a.w + (long)row * d.kpad + kc * 32 + q * 8);
    bv[f] = *reinterpret_cast<const f32x4*>(a.bias + wave * 16 * NF + f * 16 + q * 4);
  }
  const int drow = lane >> 2, dchunk = (lane & 3) ^ swz32(lane >> 4);
  const uint32_t x_pitch = (uint32_t)d.in_c_total * 2u;
  const uint32_t lane_src = (uint32_t)drow * x_pitch + (uint32_t)(d.in_c_offset + dchunk * 8) * 2u;
  const uint32_t xfrag = (uint32_t)(c16 * 64 + ((q ^ swz32(c16 >> 2)) << 4));
  const int orow = lane / LPR, ocol = lane % LPR;

  // tile j of this workgroup = blockIdx.x + j * gridDim.x; beyond the layer: every piece out of range
  auto issue_tile = [&](int j) {
    const long t = (long)blockIdx.x + (long)j * gridDim.x;
    const long p0 = t * P;
    char* const buf = smem + (j & 1) * BUF;
#pragma unroll
    for (int i = 0; i < NPW; ++i) {
      const int piece = i * 4 + wave, kc = piece / PB, pb = piece - kc * PB;
      const long px = p0 + pb * 16 + drow;
      const uint32_t vo = (t < n_tiles_px && px < a.M) ? (uint32_t)(p0 + pb * 16) * x_pitch + lane_src : kOobOffset;
      lds_dma16s(rx, buf + piece * 1024, vo, (uint32_t)kc * 64u);
    }
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("" ::: "memory");                       // the stores / DMAs that follow stay behind these (counted waits)
#endif
  };

  const int nt = ((int)blockIdx.x < n_tiles_px) ? (n_tiles_px - 1 - (int)blockIdx.x) / (int)gridDim.x + 1 : 0;
  issue_tile(0);
  issue_tile(1);
  for (int j = 0; j < nt; ++j) {
    const long p0 = ((long)blockIdx.x + (long)j * gridDim.x) * P;
    char* const buf = smem + (j & 1) * BUF;
    if (j == 0) wait_vmcnt<NPW>();
    else if (j == 1) wait_vmcnt<NPW + NST>();
    else wait_vmcnt<NPW + 2 * NST>();
    __builtin_amdgcn_s_barrier();                       // tile j has landed for every wave
    f32x4 acc[NF][PB];
#pragma unroll
    for (int f = 0; f < NF; ++f)
#pragma unroll
      for (int b = 0; b < PB; ++b) acc[f][b] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kc = 0; kc < KC; ++kc)
#pragma unroll
      for (int b = 0; b < PB; ++b) {
        const bf16x8 xf = *reinterpret_cast<const bf16x8*>(buf + (kc * PB + b) * 1024 + xfrag);
#pragma unroll
        for (int f = 0; f < NF; ++f) acc[f][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wreg[f][kc], xf, acc[f][b], 0, 0, 0);
      }
    wait_lds();
    __builtin_amdgcn_s_barrier();                       // every wave is done with the pixel tile: stage the result over it
    auto stage = [&](auto act) {
#pragma unroll
      for (int f = 0; f < NF; ++f)
#pragma unroll
        for (int b = 0; b < PB; ++b) {
          bf16x4 o;
#pragma unroll
          for (int e = 0; e < 4; ++e) o[e] = (bf16_t)act(acc[f][b][e] + bv[f][e]);
          *reinterpret_cast<bf16x4*>(buf + (b * 16 + c16) * SP + (wave * 16 * NF + f * 16 + q * 4) * 2) = o;
        }
    };
    if (ap.swish) stage([](float v) { return v / (1.f + expf(-v)); });
    else stage([&](float v) { return ap.plain(v); });
    wait_lds();
    __builtin_amdgcn_s_barrier();
    u32x4 v[NST];
#pragma unroll
    for (int i = 0; i < NST; ++i) v[i] = *reinterpret_cast<const u32x4*>(buf + ((i * 4 + wave) * RPI + orow) * SP + ocol * 16);
    wait_lds();
    __builtin_amdgcn_s_barrier();                       // every wave has read its rows: the buffer takes the tile two steps ahead
    issue_tile(j + 2);
#pragma unroll
    for (int i = 0; i < NST; ++i) {
      const int row = (i * 4 + wave) * RPI + orow;
      const uint32_t vo = p0 + row < a.M ? (uint32_t)(p0 + row) * y_pitch + (uint32_t)(d.out_c_offset + ocol * 8) * 2u : kOobOffset;
      __builtin_amdgcn_raw_buffer_store_b128(v[i], ry, vo, 0, 0);       // (no SGPR soffset: conv3x3_t20.hip, epilogue note)
    }
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("" ::: "memory");
#endif
  }
  wait_vmcnt<0>();                                       // (the two dummy tiles behind the last one are still landing in LDS)
}

template <int N, int K, int P>
int launch_stream2(const ConvArgs& a, hipStream_t s) {
  const int tiles = (a.M + P - 1) / P;
  const int grid = tiles < 512 ? tiles : 512;        // two persistent workgroups per CU
  if (pick_only("stream1x1p<%d couts,K %d,%d px> grid %d", N, K, P, grid)) return 0;
  hipLaunchKernelGGL((conv1x1_stream2_kernel<N, K, P>), dim3((unsigned)grid), dim3(256), 0, s, a, tiles);
  return yolo_check_launch("yolo_conv2d_fwd(1x1 stream, pipelined)");
}

template <int N, int K>
int launch_stream(const ConvArgs& a, hipStream_t s) {
  const int tiles = (a.M + 127) / 128;
  const int grid = tiles < 512 ? tiles : 512;        // two persistent workgroups per CU
  if (pick_only("stream1x1<%d couts,K %d> grid %d", N, K, grid)) return 0;
  hipLaunchKernelGGL((conv1x1_stream_kernel<N, K>), dim3((unsigned)grid), dim3(256), 0, s, a, tiles);
  return yolo_check_launch("yolo_conv2d_fwd(1x1 stream)");
}

}  // namespace

// Returns 1 when the kernel does not take the layer.  force: also layers below the size the shipped rule asks for.
int yolo_conv::launch_stream1x1(const ConvArgs& a, int force, hipStream_t s) {
  const YoloConvDesc& d = a.d;
  if (d.ksize != 1 || d.stride != 1 || d.pad != 0 || d.upsample2x || d.out_dtype != YOLO_DT_BF16 || a.res || a.aux) return 1;
  if (d.in_c_offset % 8 || d.in_c_total % 8 || d.out_c_offset % 8 || d.out_c_total % 8) return 1;
  if ((size_t)a.M * d.out_c_total * 2 >= kOobOffset) return 1;
  // Shipped rule: the short-K layers of the large maps (M >= 40000 pixels: the 160x160 and 80x80 maps at 16 images).  Per 16
  // images against the tiled kernel: 128 -> 64 @160x160 0.0446 -> 0.0268 ms, 256 -> 128 @80x80 0.0211 -> 0.0191 (32 images:
  // 0.0412 -> 0.0322).  With apply_act's former branch chain in the staging loop the same kernel LOST to the tiled one
  // (0.0255 vs 0.0207): four waves per workgroup have nobody to hide a serial epilogue behind.
  if (!force && a.M < 40000) return 1;
  // round 5: the pipelined form for the 128-cout layers (YOLO_CONV_DEBUG bit 33554432: the first form, for A/Bs)
  if (!(a.debug & 33554432)) {
    if (d.cout == 128 && d.cin == 256) return launch_stream2<128, 256, 80>(a, s);
    if (d.cout == 128 && d.cin == 128) return launch_stream2<128, 128, 80>(a, s);
    if (d.cout == 128 && d.cin == 384) return launch_stream2<128, 384, 48>(a, s);
  }
  if (d.cout == 128 && d.cin == 256) return launch_stream<128, 256>(a, s);
  if (d.cout == 128 && d.cin == 128) return launch_stream<128, 128>(a, s);
  if (d.cout == 64 && d.cin == 128) return launch_stream<64, 128>(a, s);
  if (d.cout == 64 && d.cin == 256) return launch_stream<64, 256>(a, s);
  return 1;
}
