// The WIDE MobileNetV2 inverted-residual blocks in one launch each (torchvision InvertedResidual
// as used by /root/reference/pytorch_yolo/models/yolov3_tiny_mobilenet.py:14-46; VERDICT r3 missing item 4):
//     y = [x +] proj1x1( relu6( dw3x3_stride( relu6( expand1x1(x) ) ) ) ),   BN folded into every conv,
// with 64..160 input channels and 384..960 hidden ones, i.e. the ten blocks on the 26x26 / 13x13 maps of a 416x416
// input.  conv_mbconv.hip keeps a whole tile of the hidden tensor in LDS, which stops at 192 hidden channels; here
// the hidden dimension is STREAMED in chunks of 64 channels through a three-phase pipeline while the projection's
// accumulators stay in registers over all chunks:
//   per tile   x halo tile -> LDS (all cin channels), stays for the whole tile (and supplies the residual)
//   per chunk  B  E[halo pixel][64] = relu6(x We_chunk^T + be)  (v_mfma_f32_16x16x32_bf16; zero outside the image: the
//                 depthwise conv pads the EXPANDED map)
//              C  D[pixel][64] = relu6(dw3x3(E) + bd)            (packed fp32 FMA, weights from LDS into registers)
//              D  acc[pixel][cout] += D Wp_chunk^T               (MFMA, fp32 accumulators live across the chunks)
//   per tile   y = acc + bp (+ x)
// One workgroup of 8 waves per CU (~210 registers per lane; with 16 waves of 128 registers the kernel spills and is 1.5 x slower):
// a 64-image batch has 256 tiles of 13x13 outputs on the 26x26 maps and 256 tiles of
// 7x7 on the 13x13 maps, one round of the chip either way.  The x tile and the chunk's weights (We 64 x cin, Wp cout x 64,
// the 9 x 64 depthwise taps, biases) go global -> LDS by DMA (buffer_load ... lds; zeros for padding come from out-of-range
// offsets), each set issued as soon as the phase that read its region has ended: We / be / the depthwise set of chunk
// hc + 1 after phase B of chunk hc, Wp of chunk hc before its phase B - a chunk costs three workgroup barriers and no
// staging registers.
// Rounding points = those of the three-launch path (E and D rounded to bf16 where that path stores them), so the two
// agree up to fp32 summation order.
#include "conv_common.h"

namespace {
using namespace yolo_conv;

struct MbwArgs {
  const bf16_t* x;
  bf16_t* y;
  const bf16_t* we;     // [ce][cin]
  const float* be;      // [ce]
  const float* wd;      // [9][ce]
  const float* bd;      // [ce]
  const bf16_t* wp;     // [cop][ce]
  const float* bp;      // [cop]
  int n, h, w, ho, wo, cin, in_ct, in_co, ce, cout, cop, out_ct, out_co, has_res;
  int tiles_x, tiles_y, n_tiles;
  uint32_t x_bytes;
  int debug;            // YOLO_MBWIDE_DEBUG (timing only, results wrong): 2 no expand phase, 4 no depthwise phase, 8 no projection phase,
                        // 16 no weight DMAs after chunk 0
};

constexpr int kCH = 64;            // hidden channels per chunk
constexpr int kES = kCH * 2 + 32;  // bytes per row of E, D and Wp_chunk.  Row strides are 16 * (2 mod 4) bytes: ds_read_b128 serves the
                                   // lane groups {0-3, 12-15, 20-27} / {4-11, 16-19, 28-31} (MI355X_MICROARCH.md, LDS), i.e. fragment
                                   // rows {0-3, 12-15} at 16-byte chunk q and rows {4-11} at chunk q + 1 together - with such a stride
                                   // the first eight land on the even 16-byte slots of a 256-byte window, the others on the odd ones
constexpr uint32_t kOob = 0x80000000u;   // a buffer offset beyond every num_records: the DMA writes zeros

__device__ __forceinline__ float relu6(float v) { return __builtin_amdgcn_fmed3f(v, 0.f, 6.f); }
__device__ __forceinline__ f32x2 bf16pair_to_f32(uint32_t w) {
  return f32x2{__builtin_bit_cast(float, w << 16), __builtin_bit_cast(float, w & 0xffff0000u)};
}
__host__ __device__ constexpr int kb(int bytes) { return (bytes + 1023) / 1024 * 1024; }    // a wave's LDS-DMA writes 1 KB

// LDS map (bytes).  Rows of x and We are cin * 2 + 32 bytes apart (cin is a multiple of 32: the same slot rule).  Every region a
// DMA fills ends on a 1 KB boundary of its own (the tail lanes of its last DMA write zeros there).
struct Lds {
  int x, e, d, we, wp, wd, be, bp, total;
};
template <int S, int TH, int TW>
__host__ __device__ inline Lds lds_map(int cin, int cop) {
  constexpr int IH = (TH - 1) * S + 3, IW = (TW - 1) * S + 3, HP = IH * IW, P = TH * TW, RTP = (P + 15) / 16;
  const int xs = cin * 2 + 32;
  Lds m;
  m.x = 0;
  m.e = m.x + HP * xs;                  // (the x DMA's tail and fragment rows >= HP spill into E: harmless at those points)
  m.d = m.e + kb(HP * kES);
  m.we = m.d + kb(RTP * 16 * kES);
  m.wp = m.we + kCH * xs;               // 64 * xs is a multiple of 1 KB
  m.wd = m.wp + kb(cop * kES);          // two sets of [wd 9 x 64 f32 (3 KB) | bd 64 f32 (1 KB)]
  m.be = m.wd + 2 * 4096;
  m.bp = m.be + 1024;
  m.total = m.bp + cop * 4;
  return m;
}

// S: stride of the depthwise conv; TH x TW: output tile; NT threads; MC: cout tiles (16 wide) per wave in phase D - the waves form
// 4 row groups x NT/256 column groups, cout <= 16 * MC * NT/256
template <int S, int TH, int TW, int NT, int MC>
__global__ __launch_bounds__(NT) void mbwide_kernel(const MbwArgs a) {
  constexpr int IH = (TH - 1) * S + 3, IW = (TW - 1) * S + 3, HP = IH * IW, RTE = (HP + 15) / 16, P = TH * TW, RTP = (P + 15) / 16,
                MR = (RTP + 3) / 4, NW = NT / 64, NCG = NW / 4;
  extern __shared__ __attribute__((aligned(1024))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, c16 = lane & 15, q = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int cin = a.cin, xs = cin * 2 + 32, ks = cin >> 5, cop = a.cop, nct = cop >> 4, nch = a.ce / kCH, ce = a.ce;
  const Lds m = lds_map<S, TH, TW>(cin, cop);
  char* const lx = smem + m.x;
  char* const le = smem + m.e;
  char* const ld = smem + m.d;
  char* const lwe = smem + m.we;
  char* const lwp = smem + m.wp;
  char* const lwd = smem + m.wd;
  float* const lbe = reinterpret_cast<float*>(smem + m.be);
  float* const lbp = reinterpret_cast<float*>(smem + m.bp);
  const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)a.x, 0, a.x_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rwe = __builtin_amdgcn_make_buffer_rsrc((void*)a.we, 0, (uint32_t)(ce * cin * 2), 0x00020000);
  const __amdgpu_buffer_rsrc_t rwp = __builtin_amdgcn_make_buffer_rsrc((void*)a.wp, 0, (uint32_t)(cop * ce * 2), 0x00020000);
  const __amdgpu_buffer_rsrc_t rwd = __builtin_amdgcn_make_buffer_rsrc((void*)a.wd, 0, (uint32_t)(9 * ce * 4), 0x00020000);
  const __amdgpu_buffer_rsrc_t rbd = __builtin_amdgcn_make_buffer_rsrc((void*)a.bd, 0, (uint32_t)(ce * 4), 0x00020000);
  const __amdgpu_buffer_rsrc_t rbe = __builtin_amdgcn_make_buffer_rsrc((void*)a.be, 0, (uint32_t)(ce * 4), 0x00020000);

  for (int i = tid; i < cop; i += NT) lbp[i] = a.bp[i];

  // ---- weight chunks: global -> LDS DMAs (1 KB per wave and instruction), DMA number d on wave d % NW.
  // set E of chunk hc: We rows [hc*64, +64) -> lwe (padded rows), be -> lbe, wd / bd -> lwd[hc & 1]
  const int n_we = kCH * xs / 1024;                       // 10 / 14 / 22
  const int n_e = n_we + 5;                               // + be, 3 x wd, bd
  const int my_e = (n_e - wave + NW - 1) / NW;               // how many of them this wave issues (for the counted wait)
  auto issue_e = [&](int hc) {
    for (int d = wave; d < n_e; d += NW) {
      if (d < n_we) {
        const int o = d * 1024 + lane * 16, row = o / xs, col = o - row * xs;
        lds_dma16(rwe, lwe + d * 1024, col < cin * 2 ? (uint32_t)(((hc * kCH + row) * cin) * 2 + col) : kOob);
      } else if (d == n_we) {
        lds_dma16(rbe, reinterpret_cast<char*>(lbe), lane < 16 ? (uint32_t)((hc * kCH) * 4 + lane * 16) : kOob);
      } else if (d < n_we + 4) {
        const int o = (d - n_we - 1) * 1024 + lane * 16, t = o >> 8, c = o & 255;    // [9][64] f32 rows of 256 bytes
        lds_dma16(rwd, lwd + (hc & 1) * 4096 + (d - n_we - 1) * 1024, t < 9 ? (uint32_t)((t * ce + hc * kCH) * 4 + c) : kOob);
      } else {
        lds_dma16(rbd, lwd + (hc & 1) * 4096 + 3072, lane < 16 ? (uint32_t)((hc * kCH) * 4 + lane * 16) : kOob);
      }
    }
  };
  const int n_p = kb(cop * kES) / 1024;
  auto issue_p = [&](int hc) {                            // Wp[:, hc*64 .. +64) -> lwp rows of kES bytes
    for (int d = wave; d < n_p; d += NW) {
      const int o = d * 1024 + lane * 16, row = o / kES, col = o - row * kES;
      lds_dma16(rwp, lwp + d * 1024, (row < cop && col < kCH * 2) ? (uint32_t)((row * ce + hc * kCH) * 2 + col) : kOob);
    }
  };

  for (int tile = blockIdx.x; tile < a.n_tiles; tile += gridDim.x) {
    const int tx = tile % a.tiles_x, r = tile / a.tiles_x, ty = r % a.tiles_y, b = r / a.tiles_y;
    const int oy0 = ty * TH, ox0 = tx * TW;
    const int iy0 = oy0 * S - 1, ix0 = ox0 * S - 1;
    // ---- x halo tile (zeros outside the image and in the row padding), chunk 0's set E
    for (int d = wave; d < (HP * xs + 1023) / 1024; d += NW) {
      const int o = d * 1024 + lane * 16, pix = o / xs, col = o - pix * xs;
      const int iy = iy0 + pix / IW, ix = ix0 + pix % IW;
      const bool ok = pix < HP && col < cin * 2 && (unsigned)iy < (unsigned)a.h && (unsigned)ix < (unsigned)a.w;
      lds_dma16(rx, lx + d * 1024, ok ? (uint32_t)((((b * a.h + iy) * a.w + ix) * a.in_ct + a.in_co) * 2 + col) : kOob);
    }
    issue_e(0);
    f32x4 acc[MR][MC];
#pragma unroll
    for (int i = 0; i < MR; ++i)
#pragma unroll
      for (int j = 0; j < MC; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    wait_vmcnt<0>();
    __syncthreads();

    for (int hc = 0; hc < nch; ++hc) {
      if (!(a.debug & 16) || hc == 0) issue_p(hc);       // lwp is free: phase D of the previous chunk is behind the loop's last barrier
      // ---- B: E = relu6(X We^T + be), 0 outside the image.  wave: two of the chunk's four channel tiles, every 8th row tile
      if (!(a.debug & 2)) {
        const int cg = wave & 1;
        bf16x8 wf[2][5];
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
          for (int k = 0; k < 5; ++k)
            if (k < ks) wf[c][k] = *reinterpret_cast<const bf16x8*>(lwe + ((cg * 2 + c) * 16 + c16) * xs + k * 64 + q * 16);
        const f32x4 b0 = *reinterpret_cast<const f32x4*>(lbe + (cg * 2) * 16 + q * 4);
        const f32x4 b1 = *reinterpret_cast<const f32x4*>(lbe + (cg * 2 + 1) * 16 + q * 4);
        for (int rt = wave >> 1; rt < RTE; rt += NW / 2) {
          const int pix = rt * 16 + c16;
          const int iy = iy0 + pix / IW, ix = ix0 + pix % IW;
          const bool in = pix < HP && (unsigned)iy < (unsigned)a.h && (unsigned)ix < (unsigned)a.w;
          f32x4 a0 = b0, a1 = b1;
#pragma unroll
          for (int k = 0; k < 5; ++k)
            if (k < ks) {
              const bf16x8 xf = *reinterpret_cast<const bf16x8*>(lx + pix * xs + k * 64 + q * 16);
              a0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[0][k], xf, a0, 0, 0, 0);
              a1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[1][k], xf, a1, 0, 0, 0);
            }
          bf16x4 o0, o1;                                 // lane: halo pixel c16, channels (ct * 16 + q * 4) ..+3
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            o0[e] = (bf16_t)(in ? relu6(a0[e]) : 0.f);
            o1[e] = (bf16_t)(in ? relu6(a1[e]) : 0.f);
          }
          if (pix < HP) {
            *reinterpret_cast<bf16x4*>(le + pix * kES + (cg * 2) * 32 + q * 8) = o0;
            *reinterpret_cast<bf16x4*>(le + pix * kES + (cg * 2 + 1) * 32 + q * 8) = o1;
          }
        }
      }
      __syncthreads();
      // the next chunk's set E: lwe / lbe are free once every wave has left phase B (wd / bd go to the other set)
      if (hc + 1 < nch && !(a.debug & 16)) issue_e(hc + 1);
      // ---- C: D = relu6(dw3x3(E) + bd).  thread: 4 channels (its 36 taps in registers), every 64th pixel
      if (!(a.debug & 4)) {
        const int qd = tid & 15;
        const float* const wdc = reinterpret_cast<const float*>(lwd + (hc & 1) * 4096);
        f32x2 wr[9][2];
#pragma unroll
        for (int t = 0; t < 9; ++t) {
          const f32x4 v = *reinterpret_cast<const f32x4*>(wdc + t * kCH + qd * 4);
          wr[t][0] = f32x2{v[0], v[1]};
          wr[t][1] = f32x2{v[2], v[3]};
        }
        const f32x4 bv = *reinterpret_cast<const f32x4*>(wdc + 768 + qd * 4);
        if constexpr (S == 1) {
          // Stride 1 (round 5, late): a task is a run of FOUR adjacent output pixels of a row - its 3 x 6 window of E is read once
          // (18 eight-byte reads for 16 outputs instead of 9 per 4): the phase is bound by LDS reads.  The last run of a row starts at
          // TW - 4 and recomputes up to three pixels of its neighbour (identical values, stored twice) instead of masking.  Every
          // output still sums its bias and nine taps in (dy, dx) order.
          constexpr int RUN = 4, RPR = (TW + RUN - 1) / RUN, NRUN = TH * RPR;
          for (int run = tid >> 4; run < NRUN; run += NT / 16) {
            const int oy = run / RPR, r_ = run - oy * RPR;
            const int ox0 = r_ * RUN < TW - RUN ? r_ * RUN : TW - RUN;
            const char* e0 = le + (oy * IW + ox0) * kES + qd * 8;
            f32x2 s0[RUN], s1[RUN];
#pragma unroll
            for (int u = 0; u < RUN; ++u) s0[u] = f32x2{bv[0], bv[1]}, s1[u] = f32x2{bv[2], bv[3]};
#pragma unroll
            for (int dy = 0; dy < 3; ++dy) {
              f32x2 lo[RUN + 2], hi[RUN + 2];
#pragma unroll
              for (int c = 0; c < RUN + 2; ++c) {
                const u32x2 v = *reinterpret_cast<const u32x2*>(e0 + (dy * IW + c) * kES);
                lo[c] = bf16pair_to_f32(v[0]), hi[c] = bf16pair_to_f32(v[1]);
              }
#pragma unroll
              for (int dx = 0; dx < 3; ++dx)
#pragma unroll
                for (int u = 0; u < RUN; ++u) {
                  s0[u] = __builtin_elementwise_fma(lo[u + dx], wr[dy * 3 + dx][0], s0[u]);
                  s1[u] = __builtin_elementwise_fma(hi[u + dx], wr[dy * 3 + dx][1], s1[u]);
                }
            }
#pragma unroll
            for (int u = 0; u < RUN; ++u) {
              bf16x4 o;
              o[0] = (bf16_t)relu6(s0[u][0]);
              o[1] = (bf16_t)relu6(s0[u][1]);
              o[2] = (bf16_t)relu6(s1[u][0]);
              o[3] = (bf16_t)relu6(s1[u][1]);
              *reinterpret_cast<bf16x4*>(ld + (oy * TW + ox0 + u) * kES + qd * 8) = o;
            }
          }
        } else {
        for (int p = tid >> 4; p < P; p += NT / 16) {
          const int oy = p / TW, ox = p - oy * TW;
          const char* e0 = le + ((oy * S) * IW + ox * S) * kES + qd * 8;
          f32x2 s0 = f32x2{bv[0], bv[1]}, s1 = f32x2{bv[2], bv[3]};
#pragma unroll
          for (int dy = 0; dy < 3; ++dy)
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) {
              const u32x2 v = *reinterpret_cast<const u32x2*>(e0 + (dy * IW + dx) * kES);
              s0 = __builtin_elementwise_fma(bf16pair_to_f32(v[0]), wr[dy * 3 + dx][0], s0);
              s1 = __builtin_elementwise_fma(bf16pair_to_f32(v[1]), wr[dy * 3 + dx][1], s1);
            }
          bf16x4 o;
          o[0] = (bf16_t)relu6(s0[0]);
          o[1] = (bf16_t)relu6(s0[1]);
          o[2] = (bf16_t)relu6(s1[0]);
          o[3] = (bf16_t)relu6(s1[1]);
          *reinterpret_cast<bf16x4*>(ld + p * kES + qd * 8) = o;
        }
        }
      }
      // Wp of this chunk has landed: it is older than the set-E DMAs this wave issued after phase B
      if (hc + 1 >= nch || my_e == 0) wait_vmcnt<0>();
      else if (my_e == 1) wait_vmcnt<1>();
      else if (my_e == 2) wait_vmcnt<2>();
      else if (my_e == 3) wait_vmcnt<3>();
      else wait_vmcnt<4>();
      __syncthreads();
      // ---- D: acc += D Wp^T.  wave: row tiles rg + 4 i, cout tiles cq + NCG j
      if (!(a.debug & 8)) {
        const int rg = wave & 3, cq = wave >> 2;
#pragma unroll
        for (int k = 0; k < kCH / 32; ++k) {
          bf16x8 df[MR];
#pragma unroll
          for (int i = 0; i < MR; ++i)
            if (rg + 4 * i < RTP) df[i] = *reinterpret_cast<const bf16x8*>(ld + ((rg + 4 * i) * 16 + c16) * kES + k * 64 + q * 16);
#pragma unroll
          for (int j = 0; j < MC; ++j) {
            const int ct = cq + NCG * j;
            if (ct < nct) {
              const bf16x8 wf = *reinterpret_cast<const bf16x8*>(lwp + (ct * 16 + c16) * kES + k * 64 + q * 16);
#pragma unroll
              for (int i = 0; i < MR; ++i)
                if (rg + 4 * i < RTP) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf, df[i], acc[i][j], 0, 0, 0);
            }
          }
        }
      }
      wait_vmcnt<0>();   // the next chunk's set E
      __syncthreads();
    }
    // ---- y = acc + bp (+ x at the tile's centre pixel, from the LDS tile)
    {
      const int rg = wave & 3, cq = wave >> 2;
#pragma unroll
      for (int i = 0; i < MR; ++i) {
        const int p = (rg + 4 * i) * 16 + c16;
        const int ty_ = p / TW, tx_ = p - ty_ * TW;
        const int oy = oy0 + ty_, ox = ox0 + tx_;
        if (rg + 4 * i >= RTP || p >= P || oy >= a.ho || ox >= a.wo) continue;
#pragma unroll
        for (int j = 0; j < MC; ++j) {
          const int ct = cq + NCG * j, c0 = ct * 16 + q * 4;
          if (ct >= nct || c0 >= a.cout) continue;
          const f32x4 bv = *reinterpret_cast<const f32x4*>(lbp + c0);
          float v[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = acc[i][j][e] + bv[e];
          if (a.has_res) {
            const bf16x4 xv = *reinterpret_cast<const bf16x4*>(lx + ((ty_ + 1) * IW + tx_ + 1) * xs + c0 * 2);
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] += (float)xv[e];
          }
          bf16x4 o;
#pragma unroll
          for (int e = 0; e < 4; ++e) o[e] = (bf16_t)v[e];
          *reinterpret_cast<bf16x4*>(a.y + ((long)(b * a.ho + oy) * a.wo + ox) * a.out_ct + a.out_co + c0) = o;
        }
      }
    }
    __syncthreads();     // the x tile is free for the next tile
  }
}

template <int S, int TH, int TW, int NT, int MC>
int launch(MbwArgs a, hipStream_t s) {
  a.tiles_x = (a.wo + TW - 1) / TW;
  a.tiles_y = (a.ho + TH - 1) / TH;
  a.n_tiles = a.n * a.tiles_x * a.tiles_y;
  const Lds m = lds_map<S, TH, TW>(a.cin, a.cop);
  const long x_bytes = (long)a.n * a.h * a.w * a.in_ct * 2;
  YOLO_REQUIRE(m.total <= 160 * 1024 && a.cop <= 16 * MC * (NT / 256) && a.cin <= 160 && x_bytes < (1l << 31) && (long)a.cop * a.ce * 2 < (1l << 31),
               "mbconv (wide): cin %d cout %d does not fit the %dx%d tile form (%d bytes of LDS), or a tensor of 2 GB", a.cin, a.cout,
               TH, TW, m.total);
  a.x_bytes = (uint32_t)x_bytes;
  static std::atomic<uint64_t> lds_set{0};                 // per device (common.h)
  if (const int rc = yolo_max_dyn_lds(reinterpret_cast<const void*>(&mbwide_kernel<S, TH, TW, NT, MC>), 160 * 1024, lds_set, "mbconv (wide)")) return rc;
  const int grid = a.n_tiles < 256 ? a.n_tiles : 256;
  hipLaunchKernelGGL((mbwide_kernel<S, TH, TW, NT, MC>), dim3((unsigned)grid), dim3(NT), (size_t)m.total, s, a);
  return yolo_check_launch("yolo_mbconv_fwd (wide)");
}

const int mbw_form = [] {      // YOLO_MBWIDE_FORM (tests / tuning): 1 = 13x13 tiles whenever they fit, 2 = 7x7 tiles always
  const char* e = getenv("YOLO_MBWIDE_FORM");
  return e ? atoi(e) : 0;
}();
const int mbw_debug = [] {
  const char* e = getenv("YOLO_MBWIDE_DEBUG");
  return e ? atoi(e) : 0;
}();

}  // namespace

// The wide form covers: expand conv present, cin a multiple of 32 in 64..160 (stride 2: ..96), hidden a multiple of 64,
// cout a multiple of 4 up to 320 - AND the 7x7-tile form (the one yolo_mbwide_launch falls back to for every map size) must fit the
// CU: its LDS map within 160 KB and cout within its 16 * MC * 2 accumulator columns.  (stride 2, cin 96, cout in
// (256, 320] passed the channel tests, the planner fused the block away, and the launch then refused it at run time.)
int yolo_mbwide_supported(int cin, int hidden, int cout, int stride) {
  if (!(cin >= 64 && cin <= (stride == 1 ? 160 : 96) && cin % 32 == 0 && hidden % kCH == 0 && hidden >= kCH && hidden > cin && cout >= 4 &&
        cout <= 320 && cout % 4 == 0 && (stride == 1 || stride == 2)))
    return 0;
  const int cop = (cout + 15) / 16 * 16;
  const int total = stride == 2 ? lds_map<2, 7, 7>(cin, cop).total : lds_map<1, 7, 7>(cin, cop).total;
  return total <= 160 * 1024 && cop <= 16 * 10 * 2;
}

int yolo_mbwide_launch(const void* x, const void* w_exp, const float* b_exp, const float* w_dw, const float* b_dw, const void* w_proj,
                       const float* b_proj, void* y, const YoloMbconvDesc& d, hipStream_t st) {
  MbwArgs a;
  a.x = (const bf16_t*)x;
  a.y = (bf16_t*)y;
  a.we = (const bf16_t*)w_exp;
  a.be = b_exp;
  a.wd = w_dw;
  a.bd = b_dw;
  a.wp = (const bf16_t*)w_proj;
  a.bp = b_proj;
  a.n = d.n;
  a.h = d.h;
  a.w = d.w;
  a.ho = (d.h + 2 - 3) / d.stride + 1;
  a.wo = (d.w + 2 - 3) / d.stride + 1;
  a.cin = d.cin;
  a.in_ct = d.in_c_total;
  a.in_co = d.in_c_offset;
  a.ce = d.hidden;
  a.cout = d.cout;
  a.cop = (d.cout + 15) / 16 * 16;
  a.out_ct = d.out_c_total;
  a.out_co = d.out_c_offset;
  a.has_res = d.has_res;
  a.tiles_x = a.tiles_y = a.n_tiles = 0;
  a.debug = mbw_debug;
  if (d.stride == 2) return launch<2, 7, 7, 512, 10>(a, st);
  // 13x13 tiles (a quarter of a 26x26 map: 42 % halo instead of 127 %, a quarter of the weight passes) while they give every
  // CU a tile and the block is narrow enough for their LDS / accumulator budget
  const long t13 = (long)d.n * ((a.ho + 12) / 13) * ((a.wo + 12) / 13);
  if (d.cin <= 96 && a.cop <= 128 && (t13 >= 192 || mbw_form == 1) && mbw_form != 2 && lds_map<1, 13, 13>(a.cin, a.cop).total <= 160 * 1024)
    return launch<1, 13, 13, 512, 4>(a, st);
  return launch<1, 7, 7, 512, 10>(a, st);
}
