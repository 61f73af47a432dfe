// Detection-head conv + YOLOLayer decode (+ NMS row filter) as a weight-stationary, software-pipelined kernel for gfx950 (MI355X):
// same contract and numerics as the DECODE instances of conv_igemm.hip (reference models/yolov3_spp.py:86,99,111 - the head
// ConvBlocks -, models/yolo_layer.py:57-69,90-96, utils/utils.py:212-218), same epilogue code (head_epilogue.h).
//
// Why (round 5).  The head of the 80x80 map of YOLOv3-SPP 640 (256 -> 255 channels, 204,800 pixels per 32 images) took 0.115-0.119 ms
// as 3,200 workgroups of 64 pixels x 256 couts: every workgroup pulled the whole 128 KB weight matrix through its CU's L2 port
// (410 MB per launch for a layer that reads 105 MB of activations: floor 0.02 ms), ran a K loop of barriers and then a serial
// decode / filter epilogue with nothing in flight.  Here
//   * the weights never move: a wave owns 256 / NW couts and keeps its slice of W for the WHOLE K in registers (128 registers:
//     K = 256 on 4 waves, K = 512 on 8), loaded once per persistent workgroup;
//   * pixels stream through two tile buffers of P = 32 pixels x K channels by LDS-DMA ([K / 32 chunks][32 pixels][64 B], swz32
//     rows as in conv1x1_stream.hip); the tile TWO steps ahead is requested as soon as its buffer is free (behind the barrier that
//     ends the MFMA phase), so it lands while this tile is decoded and filtered;
//   * act(conv + bias) is staged as fp32 [32 pixels][256] in a region of its own, and every wave decodes / filters its own
//     32 / NW pixels straight from there (head_decode_rows): two barriers per tile.
// In the FILTER form (detect(): io is never written) a wave issues a fixed number of vector-memory operations per tile (NPW DMA
// pieces + 3 buffer stores: head_epilogue.h), so the wait for a tile is a counted s_waitcnt that never waits for store
// acknowledgements; with io / p stores (forward()) the counts depend on the data layout and the wait is vmcnt(0).
#include "conv_common.h"
#include "head_epilogue.h"

using namespace yolo_conv;

namespace {

template <int K, int NW>   // input channels (64 .. 512, a multiple of 32), waves (4 or 8)
__global__ __launch_bounds__(64 * NW, NW == 4 ? 2 : 1) void head_stream_kernel(const ConvArgs a, int n_tiles_px) {
  constexpr int P = 32, PB = P / 16, KC = K / 32, CW = 256 / NW, NF = CW / 16, PPW = P / NW;
  constexpr int XB = KC * P * 64;                              // bytes of a pixel tile
  constexpr int DP = 256 + 4;                                  // fp32 staging pitch (floats)
  constexpr int NPW = (KC * PB + NW - 1) / NW;                 // DMA pieces per wave and tile (the last round may be partly empty)
  constexpr int NST = 3;                                       // filter-mode stores per wave and tile (head_epilogue.h: one pass of <= 32 rows)
  static_assert(NF * KC * 4 <= 128 && NPW + 2 * NST <= 60, "register / vmcnt budget");
  __shared__ __attribute__((aligned(16))) char smem[2 * XB + P * DP * 4];
  float* const stg = reinterpret_cast<float*>(smem + 2 * XB);

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const YoloConvDesc& d = a.d;
  const int c16 = lane & 15, q = lane >> 4;
  const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)a.x, 0, a.x_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void*)a.w, 0, a.w_bytes, 0x00020000);

  // ---- this wave's weights: fragment f = couts [wave * CW + 16 f, +16), k-step kc: lane (row c16, 8-channel chunk q); rows beyond
  // cout_pad (heads with few classes: cout_pad = 128) read zeros through the buffer descriptor, their bias is 0
  bf16x8 wreg[NF][KC];
  f32x4 bv[NF];
#pragma unroll
  for (int f = 0; f < NF; ++f) {
    const int row = wave * CW + f * 16 + c16;
#pragma unroll
    for (int kc = 0; kc < KC; ++kc)
      wreg[f][kc] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rw, (uint32_t)((row * d.kpad + q * 8) * 2), (uint32_t)(kc * 64), 0));
    const int c0 = wave * CW + f * 16 + q * 4;
    bv[f] = c0 < d.cout_pad ? *reinterpret_cast<const f32x4*>(a.bias + c0) : f32x4{0.f, 0.f, 0.f, 0.f};
  }
#if defined(__HIP_DEVICE_COMPILE__)
  asm volatile("" ::: "memory");                           // W is requested BEFORE the pixel DMAs (in-order retirement: the counted waits)
#endif
  const int drow = lane >> 2, dchunk = (lane & 3) ^ swz32(lane >> 4);
  const uint32_t x_pitch = (uint32_t)d.in_c_total * 2u;
  const uint32_t lane_src = (uint32_t)drow * x_pitch + (uint32_t)(d.in_c_offset + dchunk * 8) * 2u;
  const uint32_t xfrag = (uint32_t)(c16 * 64 + ((q ^ swz32(c16 >> 2)) << 4));
  const HeadLanes<256> hl = head_lanes<256>(a.hd, lane);
  const bool counted = !a.hd.io && !a.hd.p;                // filter form without p: every tile costs a wave NPW + NST operations

  // tile j of this workgroup = blockIdx.x + j * gridDim.x; beyond the layer (and pieces beyond KC * PB): out-of-range offsets
  auto issue_tile = [&](int j) {
    const long t = (long)blockIdx.x + (long)j * gridDim.x;
    const long p0 = t * P;
    char* const buf = smem + (j & 1) * XB;
#pragma unroll
    for (int i = 0; i < NPW; ++i) {
      const int piece = i * NW + wave, kc = piece / PB, pb = piece - kc * PB;
      const long px = p0 + pb * 16 + drow;
      const bool ok = piece < KC * PB && t < n_tiles_px && px < a.M;
      const uint32_t vo = ok ? (uint32_t)(p0 + pb * 16) * x_pitch + lane_src : kOobOffset;
      lds_dma16s(rx, buf + (piece < KC * PB ? piece : 0) * 1024, vo, (uint32_t)kc * 64u);
    }
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("" ::: "memory");
#endif
  };
  // NOTE on the pieces beyond KC * PB (only when KC * PB is not a multiple of NW): they would write zeros over piece 0 of the buffer
  // - which some other wave's real piece 0 also fills; the launcher only instantiates shapes where KC * PB % NW == 0.
  static_assert((KC * PB) % NW == 0, "every wave issues the same number of real pieces");

  const int nt = ((int)blockIdx.x < n_tiles_px) ? (n_tiles_px - 1 - (int)blockIdx.x) / (int)gridDim.x + 1 : 0;
  issue_tile(0);
  issue_tile(1);
#if defined(__HIP_DEVICE_COMPILE__)
  // W is in registers before the loop starts (an empty use: the compiler's own wait for these loads - all but the 2 NPW tile pieces
  // issued behind them - lands here once, instead of as "vmcnt(8)"-style waits in front of the first MFMAs of EVERY tile)
  static_for<NF>([&](auto fc) {
    static_for<KC>([&](auto kc) {
      const bf16x8& wv = wreg[decltype(fc)::value][decltype(kc)::value];
      asm volatile("" ::"v"(wv));
    });
  });
#endif
  for (int j = 0; j < nt; ++j) {
    const long p0 = ((long)blockIdx.x + (long)j * gridDim.x) * P;
    char* const buf = smem + (j & 1) * XB;
    if (!counted) wait_vmcnt<0>();
    else if (j == 0) wait_vmcnt<NPW>();
    else if (j == 1) wait_vmcnt<NPW + NST>();
    else wait_vmcnt<NPW + 2 * NST>();
    __builtin_amdgcn_s_barrier();                       // tile j has landed for every wave; every wave is done with tile j - 1's staged rows
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
    // act(conv + bias) -> fp32 staging: lane = pixel c16 of block b, 4 consecutive couts
#pragma unroll
    for (int f = 0; f < NF; ++f)
#pragma unroll
      for (int b = 0; b < PB; ++b) {
        f32x4 v;
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = apply_act(acc[f][b][e] + bv[f][e], d.act);
        *reinterpret_cast<f32x4*>(stg + (b * 16 + c16) * DP + wave * CW + f * 16 + q * 4) = v;
      }
    wait_lds();
    __builtin_amdgcn_s_barrier();                       // the tile is staged; every wave is done with the pixel buffer
    issue_tile(j + 2);
    head_decode_rows<PPW, 256>(a, hl, stg, DP, (int)p0, wave, lane);
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("" ::: "memory");
#endif
  }
  wait_vmcnt<0>();                                       // (the dummy tiles behind the last one are still landing in LDS)
}

template <int K, int NW>
int launch_hs(const ConvArgs& a, hipStream_t s) {
  const int tiles = (a.M + 31) / 32;
  const int slots = launch_cus() * (NW == 4 ? 2 : 1);    // persistent: two 4-wave workgroups per CU, or one of 8 waves
  const int grid = tiles < slots ? tiles : slots;
  if (pick_only("head_stream<K %d, %d waves, 32 px> grid %d", K, NW, grid)) return 0;
  hipLaunchKernelGGL((head_stream_kernel<K, NW>), dim3((unsigned)grid), dim3(64 * NW), 0, s, a, tiles);
  return yolo_check_launch("yolo_head_decode_fwd(stream)");
}

}  // namespace

// 1: this kernel does not take the head (the caller goes on to the tiled DECODE instances).
// force: 0 = the shipped rule (enough 32-pixel tiles to give every persistent workgroup several), 1 = every shape it can compute.
int yolo_conv::launch_head_stream(const ConvArgs& a, int force, hipStream_t s) {
  const YoloConvDesc& d = a.d;
  if (d.ksize != 1 || d.stride != 1 || d.pad != 0 || d.cout > 256 || d.cout_pad % 4 != 0 || d.cin % 32 != 0) return 1;
  if (d.in_c_offset % 8 || d.in_c_total % 8 || a.hd.no * a.hd.na > 256 || a.hd.na > 4) return 1;
  if ((size_t)a.M * d.in_c_total * 2 >= kOobOffset || d.act == YOLO_ACT_SWISH) return 1;
  // one filter pass per wave and tile (the counted waits assume it): <= 32 rows = 8 (4 waves) / 4 (8 waves) pixels x na anchors
  const int tiles = (a.M + 31) / 32;
  if (!force && tiles < 4 * launch_cus()) return 1;       // (13x13 / 20x20 maps at small batches: a round or two of tiles, nothing to pipeline)
  switch (d.cin) {
    case 64: return launch_hs<64, 4>(a, s);
    case 128: return launch_hs<128, 4>(a, s);
    case 256: return launch_hs<256, 4>(a, s);
    case 512: return launch_hs<512, 8>(a, s);
    default: return 1;
  }
}
