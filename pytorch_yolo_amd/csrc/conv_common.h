// Pieces shared by the convolution kernels (gather implicit-GEMM and halo-staged 3x3).
#pragma once
#include "common.h"

#include <utility>

namespace yolo_conv {

// static_for<N>(f): f(std::integral_constant<int, 0>{}) ... f(std::integral_constant<int, N-1>{}) - loop indices that are
// constant expressions inside the body (if constexpr, asm "i" operands, register-array subscripts the optimiser never sees as
// run-time values)
template <typename F, int... Is>
__device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, Is...>) {
  (f(std::integral_constant<int, Is>{}), ...);
}
template <int N, typename F>
__device__ __forceinline__ void static_for(F&& f) {
  static_for_impl(f, std::make_integer_sequence<int, N>{});
}

constexpr uint32_t kOobOffset = 0xF0000000u;  // > any buffer we accept (host checks < 0xF0000000 bytes)

// YOLOLayer decode fused into a head conv's epilogue (conv_igemm.hip, DECODE instances)
struct HeadDecodeArgs {
  float* io;            // [bs, io_rows_total, no]
  float* p;             // [bs, na, ny, nx, no] or null
  int na, no, io_rows_total, io_row_offset;
  float stride;
  float anchor_w[4], anchor_h[4];   // anchors_px / stride (yolo_layer.py:109)
  // filter mode (yolo_head_decode_filter_fwd; io == nullptr): the epilogue runs the NMS row filter on its own decoded rows and
  // leaves a key per row + the survivors' records in the compact NMS workspace (nms_common.h) instead of storing io
  float* rec;                        // [bs][io_rows_total][8]: x, y, w, h, class score of a surviving row, at its io row
  unsigned long long* row_keys;      // [bs][io_rows_total]: the row's sort key if it survives, ~0 otherwise
  float conf_thres, min_wh;
};

struct ConvArgs {
  const bf16_t* x;
  const bf16_t* w;
  const float* bias;
  const bf16_t* res;
  void* y;
  bf16_t* aux;
  YoloConvDesc d;
  int M;        // n*ho*wo
  int n_tiles;  // cout tiles
  int steps;    // kpad / 32
  uint32_t x_bytes, w_bytes;
  HeadDecodeArgs hd;   // DECODE instances only
  // split-K launches (conv_igemm_bf16_kernel<..., SPLITK>): grid.y = splits, `steps` = K steps PER split; every split
  // writes its fp32 partial tile to ws[split][M][cout], the last one to arrive at a tile (cnt[tile], self-resetting)
  // adds the partials in split order and runs the normal epilogue
  int splits;
  float* ws;
  int* cnt;
  int debug;    // YOLO_CONV_DEBUG / yolo_set_tuning(1, .), timing ablations and A/B rules only (results are wrong with bits 1..8 set).
                // Ablation bits every conv kernel understands:   1 no pixel DMA   2 no weight DMA   4 no MFMA   8 no epilogue
                // Rule bits of the dispatcher (conv_igemm.hip):  16 no LDS-staged epilogue   32 no halo kernel   128 no 128x256 tiles
                //   256 no loader waves   512 8-wave 256x256 tiles   2048 32x32x16 MFMA in the gather kernel   8192 4-wave 128x128 / head
                //   tiles   16384 two-stage ring for the loader-wave tiles   32768 64x64 tiles for every tiny-grid 1x1 layer
                //   4194304 two-stage 64x64 tiles;   halo kernel: 1024 blocks of 256 couts   65536 32x32x16 MFMA
                // Kernel FAMILIES are selected by the other knob, YOLO_CONV_PP / yolo_set_tuning(2, .): 8 no halo kernel, 16 / 64 the
                //   20x20-tile kernels always / never, 1024 / 2048 the streaming 1x1 never / always.  YOLO_RESUNIT_DEBUG (fused units):
                //   8 no epilogue, 32 generic 16x16 kernel for C = 64, 64 / 128 the 20-pixel-wide tile kernels always / never,
                //   512 one workgroup per CU, 1024 dump tile 0's intermediate (tools/dbg/ruw_tdump.py).
#ifdef YOLO_STAMPS
  unsigned long long* stamps;   // diagnostic build only (tools/block_timeline.py): 4 words per workgroup
#endif
};

// Diagnostic build (-DYOLO_STAMPS, never the shipped library): the first lane of every workgroup records when it
// started and ended (s_memrealtime, 100 MHz), how many shader cycles that took (s_memtime) and where it ran
// (HW_ID, XCC_ID).  The words go to a buffer nothing else reads (MI355X_MICROARCH.md, DVFS give-back item 6).
#ifdef YOLO_STAMPS
struct BlockStamp {
  unsigned long long* base_;
  unsigned long long* p;
  unsigned long long t0, c0;
  __device__ __forceinline__ unsigned long long* slot_() const {
    return base_ + 4 * ((size_t)blockIdx.x + (size_t)gridDim.x * (blockIdx.y + (size_t)gridDim.y * blockIdx.z)) + 1;
  }
  __device__ __forceinline__ explicit BlockStamp(unsigned long long* base) : base_(base), p(nullptr), t0(0), c0(0) {
    if (base && threadIdx.x == 0) {
      p = slot_() - 1;
      t0 = __builtin_amdgcn_s_memrealtime();
      c0 = __builtin_amdgcn_s_memtime();
    }
  }
  __device__ __forceinline__ ~BlockStamp() {
    // the workgroup ends when its last wave does: every wave's first lane folds its end time in
    if (p) {
      const unsigned long long c1 = __builtin_amdgcn_s_memtime();
      const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | 4), xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20);
      p[0] = t0;
      p[2] = c1 - c0;
      p[3] = (unsigned long long)hw | ((unsigned long long)xcc << 32);
    }
    if ((threadIdx.x & 63) == 0 && base_) atomicMax(slot_(), (unsigned long long)__builtin_amdgcn_s_memrealtime());
  }
};
inline unsigned long long* stamp_buffer_from_env() {
  static long launch_idx = 0;
  const char* e = getenv("YOLO_STAMP_PTR");
  if (!e) return nullptr;
  const char* st = getenv("YOLO_STAMP_STRIDE");   // words per launch (0 / unset: every launch writes at the base)
  const long stride = st ? atol(st) : 0;
  return (unsigned long long*)strtoull(e, nullptr, 0) + stride * launch_idx++;
}
#define YOLO_BLOCK_STAMP(args) BlockStamp block_stamp_((args).stamps)
#define YOLO_SET_STAMPS(args) (args).stamps = stamp_buffer_from_env()
#else
#define YOLO_BLOCK_STAMP(args)
#define YOLO_SET_STAMPS(args)
#endif

__device__ __forceinline__ void lds_dma16(__amdgpu_buffer_rsrc_t rsrc, char* lds_wave_base, uint32_t voffset) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)lds_wave_base, 16,
                                           voffset, 0, 0, 0);
}

// Swizzle term of a 64-byte LDS row (K tile of 32 bf16): physical 16-byte slot = logical chunk ^ swz32(row >> 2).
// ds_read_b128 is serviced in the lane groups {0-3,12-15,20-27}, {4-11,16-19,28-31} (+32) (MI355X_MICROARCH.md, LDS; confirmed
// with tools/micro/lds_groups.hip), not in runs of 16 lanes: under that grouping the identity map (row >> 2) & 3 makes every
// 16x16x32 fragment read 2-way bank-conflicted, while the permutation (0,2,3,1) is conflict-free for the 16x16x32 AND the
// 32x32x16 fragment shapes.
__device__ __forceinline__ int swz32(int v) { return (0x78 >> (2 * (v & 3))) & 3; }

// Upper operand of the activation formula min(max(v, lo), hi): 6 for ReLU6, otherwise a quiet NaN.  v_min_f32 / fminf return the
// OTHER operand when one of the two is a NaN, so min(r, NaN) = r for every r and a NaN pre-activation stays a NaN.  (With +inf as
// the "no clamp" value fminf(NaN, inf) = inf: a NaN logit of a plain head decoded to confidence 1, where the reference keeps the
// NaN and its `conf > thres` drops the row.)
__device__ __forceinline__ float act_hi(int act) { return act == YOLO_ACT_RELU6 ? 6.f : __builtin_nanf(""); }

// act(v + bias) of the conv epilogues.  ONE data-independent formula covers none / LeakyReLU(0.1) / ReLU / ReLU6:
// min(max(v, lo), hi) with lo = 0 or slope * v - the selects are wave-uniform and loop-invariant, so an unrolled epilogue pays
// 4 VALU per value.  (As a chain of `if (act == ...) return ...` every one of a tile's 64-256 values carried its own tree of
// scalar compares and branches: ~10 s_cbranch per value, more cycles than the MFMAs of a short-K layer.)  Same bits as the chain
// for every finite input and for +-inf; a NaN stays a NaN under none / LeakyReLU (max(NaN, slope * NaN) = NaN) and becomes 0 under
// ReLU / ReLU6 (max(NaN, 0) = 0, as `v > 0 ? v : 0` gave).  swish (x * sigmoid(x)) keeps a uniform branch.
__device__ __forceinline__ float apply_act(float v, int act) {
  if (act == YOLO_ACT_SWISH) return v / (1.f + expf(-v));
  const bool floor0 = act == YOLO_ACT_RELU || act == YOLO_ACT_RELU6;
  const float slope = act == YOLO_ACT_LEAKY01 ? 0.1f : 1.f;
  return fminf(fmaxf(v, floor0 ? 0.f : slope * v), act_hi(act));
}

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
#if defined(__HIP_DEVICE_COMPILE__)
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
#endif
}

// LDS operations of this wave have completed (data in registers / visible in LDS); vector-memory traffic keeps flying
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float raw_max(float x, float y) {      // v_max_f32 without fmaxf()'s operand canonicalisation
  float r;
  asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(x), "v"(y));
  return r;
}
// LeakyReLU(0.1) of four values in 6 VALU instructions (2 v_pk_mul_f32 + 4 v_max_f32): the epilogues' data-independent
// min(max(v, lo), hi) form costs a multiply, a select, a max and a min per value.  max(v, 0.1 v) keeps NaN (both operands NaN) and
// +-inf like the generic form.
__device__ __forceinline__ f32x4 leaky4(f32x4 v) {
  const f32x2 lo = f32x2{v[0], v[1]} * 0.1f, hi = f32x2{v[2], v[3]} * 0.1f;
  return f32x4{raw_max(v[0], lo[0]), raw_max(v[1], lo[1]), raw_max(v[2], hi[0]), raw_max(v[3], hi[1])};
}
// activation of four values under a wave-uniform branch: LeakyReLU(0.1) takes leaky4 (6 VALU), the others apply_act (16)
__device__ __forceinline__ f32x4 act4(f32x4 v, int act) {
  if (act == YOLO_ACT_LEAKY01) return leaky4(v);
#pragma unroll
  for (int e = 0; e < 4; ++e) v[e] = apply_act(v[e], act);
  return v;
}
__device__ __forceinline__ void wait_lds() {
#if defined(__HIP_DEVICE_COMPILE__)
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#endif
}

__device__ __forceinline__ void lds_dma16s(__amdgpu_buffer_rsrc_t rsrc, char* lds_wave_base, uint32_t voffset,
                                           uint32_t soffset) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)lds_wave_base, 16,
                                           voffset, soffset, 0, 0);
}


// LDS-staged epilogue shared by the conv kernels.  acc[MI][NI] are 32x32 MFMA accumulators (lane = pixel
// column r32 of pixel block j, registers = couts); the wave's TM x (MI*32) tile goes registers -> LDS (fp32,
// one 32-cout slab per pass) -> coalesced phase in which every lane owns 8 consecutive couts of one pixel,
// so the residual read, the pre-add copy and the store are 16-byte accesses over whole 64-byte runs per
// pixel; the sum is formed in fp32 and rounded once.  pix_of(row) maps a row of the wave's pixel tile to
// the output pixel index (n*ho*wo order) or -1.  `stg` = private per-wave LDS slab of TM*kEpiPitch bytes.
constexpr int kEpiPitch = 144;   // 32 f32 + 16 B pad

// Residual values handed to the epilogue in registers instead of being read from global there (conv_resunit.hip
// takes them from its LDS staging).  Flat order = the order the coalesced phase consumes them in.
template <int MI, int TM>
struct ResPrefetch {
  bf16x8 v[MI * (TM / 16)];
};

// Core of the LDS-staged epilogue.  write_rows(i, cbase, row_lo, nrows, pitch, col_off) puts act(acc + bias) of
// 32-cout slab i, pixel rows [row_lo, row_lo + nrows) of the wave's tile, as fp32 at
// stg + (row - row_lo) * pitch + col_off (+ 4 * cout within the slab).
// PAIRED form (MI even): two slabs = 64 couts are staged side by side for HALF the rows at a time (same LDS
// footprint), so that in the coalesced phase eight lanes cover one pixel's 64 couts: every store instruction
// writes whole 128-byte lines.  With one 32-cout slab at a time a line was written as two 64-byte halves in two
// different passes, and the partially written lines cost the memory system about twice the traffic of the store.
constexpr int kEpiPitch2 = 272;   // 64 f32 + 16 B pad
template <int MI, int TM, typename WriteRows, typename PixOf>
__device__ __forceinline__ void epilogue_lds_core(const ConvArgs& a, char* stg, int lane, int cout0, WriteRows write_rows,
                                                  PixOf pix_of, const ResPrefetch<MI, TM>* rpre) {
  const YoloConvDesc& d = a.d;
  const int hw_out = d.ho * d.wo;
  constexpr bool PAIR = MI % 2 == 0;
  static_assert(!PAIR || (TM / 2) * kEpiPitch2 <= TM * kEpiPitch, "paired staging must fit the slab");
  constexpr int LPP = PAIR ? 8 : 4;                  // lanes per pixel in the coalesced phase
  constexpr int RPI = 64 / LPP;                      // pixel rows per store instruction
  constexpr int PITCH = PAIR ? kEpiPitch2 : kEpiPitch;
  constexpr int HALVES = PAIR ? 2 : 1, HROWS = TM / HALVES;
  const int crow = lane / LPP, cchunk = lane % LPP;
  int rp_idx = 0;
#pragma unroll
  for (int ig = 0; ig < (PAIR ? MI / 2 : MI); ++ig) {
    const int cbase = cout0 + ig * (PAIR ? 64 : 32);   // first cout of this group
    if (cbase >= d.cout) continue;                     // cout % 32 == 0: a slab is all-or-nothing
    const bool second = PAIR && cbase + 32 < d.cout;   // the group's second slab exists
#pragma unroll
    for (int h = 0; h < HALVES; ++h) {
      write_rows(PAIR ? 2 * ig : ig, cbase, h * HROWS, HROWS, PITCH, 0);
      if (second) write_rows(2 * ig + 1, cbase + 32, h * HROWS, HROWS, PITCH, 128);
      __builtin_amdgcn_wave_barrier();               // LDS ops of one wave execute in order
      const long cofs = cbase + cchunk * 8;
      const bool chunk_ok = cbase + cchunk * 8 < d.cout;
#pragma unroll
      for (int pass = 0; pass < HROWS / RPI; ++pass, ++rp_idx) {
        const int lrow = pass * RPI + crow;
        const long pix = pix_of(h * HROWS + lrow);
        if (pix >= 0 && chunk_ok) {
          const f32x4 lo = *reinterpret_cast<const f32x4*>(stg + lrow * PITCH + cchunk * 32);
          const f32x4 hi = *reinterpret_cast<const f32x4*>(stg + lrow * PITCH + cchunk * 32 + 16);
          float v[8] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
          if (a.aux) {
            bf16x8 o;
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = (bf16_t)v[e];
            *reinterpret_cast<bf16x8*>(a.aux + pix * d.aux_c_total + d.aux_c_offset + cofs) = o;
          }
          if (a.res) {
            const bf16x8 rv = rpre ? rpre->v[rp_idx]
                                   : *reinterpret_cast<const bf16x8*>(a.res + pix * d.res_c_total + d.res_c_offset + cofs);
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] += (float)rv[e];
          }
          bf16x8 o;
#pragma unroll
          for (int e = 0; e < 8; ++e) o[e] = (bf16_t)v[e];
          bf16_t* const ybase = reinterpret_cast<bf16_t*>(a.y) + d.out_c_offset + cofs;
          if (d.upsample2x) {
            const int b = (int)(pix / hw_out), rem = (int)(pix - (long)b * hw_out);
            const int oh = rem / d.wo, ow = rem - oh * d.wo;
            const long op = ((long)(b * 2 * d.ho + 2 * oh)) * (2 * d.wo) + 2 * ow;
#pragma unroll
            for (int rep = 0; rep < 4; ++rep)
              *reinterpret_cast<bf16x8*>(ybase + (op + (rep >> 1) * 2 * d.wo + (rep & 1)) * d.out_c_total) = o;
          } else {
            *reinterpret_cast<bf16x8*>(ybase + pix * d.out_c_total) = o;
          }
        }
      }
      __builtin_amdgcn_wave_barrier();
    }
  }
}

// 32x32x16 accumulators: acc[i][j] = couts [32i, 32i+32) x pixels [32j, 32j+32); lane = pixel column r32,
// register e -> cout (e&3) + 8*(e>>2) + 4*(lane>>5).
template <int MI, int NI, int TM, typename PixOf>
__device__ __forceinline__ void epilogue_lds(const ConvArgs& a, const f32x16 (&acc)[MI][NI], char* stg, int lane,
                                             int cout0, PixOf pix_of, const ResPrefetch<MI, TM>* rpre = nullptr) {
  const int r32 = lane & 31, khalf = lane >> 5;
  epilogue_lds_core<MI, TM>(a, stg, lane, cout0, [&](int i, int cbase, int row_lo, int nrows, int pitch, int col_off) {
#pragma unroll
    for (int j = 0; j < NI; ++j) {
      const int row = j * 32 + r32 - row_lo;
      if (row < 0 || row >= nrows) continue;
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) {
        const int cl = g4 * 8 + khalf * 4;
        const f32x4 bv = *reinterpret_cast<const f32x4*>(a.bias + cbase + cl);
        f32x4 v;
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = acc[i][j][g4 * 4 + e] + bv[e];
        v = act4(v, a.d.act);
        *reinterpret_cast<f32x4*>(stg + row * pitch + col_off + cl * 4) = v;
      }
    }
  }, pix_of, rpre);
}

// 16x16x32 accumulators: acc[i][j] = couts [16i, 16i+16) x pixels [16j, 16j+16); lane = pixel column lane&15,
// register e -> cout 4*(lane>>4) + e.  MI16 = 2*MI cout tiles, NI16 pixel tiles.
template <int MI, int NI16, int TM, typename PixOf>
__device__ __forceinline__ void epilogue_lds16(const ConvArgs& a, const f32x4 (&acc)[2 * MI][NI16], char* stg, int lane,
                                               int cout0, PixOf pix_of) {
  const int c16 = lane & 15, q = lane >> 4;
  epilogue_lds_core<MI, TM>(a, stg, lane, cout0, [&](int i, int cbase, int row_lo, int nrows, int pitch, int col_off) {
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const int cl = t * 16 + q * 4;
      const f32x4 bv = *reinterpret_cast<const f32x4*>(a.bias + cbase + cl);
#pragma unroll
      for (int j = 0; j < NI16; ++j) {
        const int row = j * 16 + c16 - row_lo;
        if (row < 0 || row >= nrows) continue;
        const f32x4 v = act4(acc[2 * i + t][j] + bv, a.d.act);
        *reinterpret_cast<f32x4*>(stg + row * pitch + col_off + cl * 4) = v;
      }
    }
  }, pix_of, (const ResPrefetch<MI, TM>*)nullptr);
}

// XCD-aware block order (bijective for any grid): blocks with equal blockIdx%8 share an XCD / L2 and get a
// contiguous run of work items.
__device__ __forceinline__ int xcd_swizzle(int bid, int nblk) {
  const int q = nblk >> 3, r = nblk & 7, xcd = bid & 7;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
}

// yolo_conv2d_pick (the tile-rule regression guard): while pick_buffer() is non-null the launch functions write the name of the
// kernel instance they would launch there and return 0 WITHOUT launching (no GPU needed).
bool resunit_t20_applies(int c, int n, int h, int w);   // conv_resunit_t20.hip: the shipped rule of the 20-pixel-wide tile kernels
int launch_resunit_t20(const ConvArgs& c, const bf16_t* w1, const float* b1, int kpad1, uint32_t w1_bytes, bool force, hipStream_t s);
int& resunit_debug();              // conv_resunit.hip: YOLO_RESUNIT_DEBUG / yolo_set_tuning(3, .)
int launch_cus();                  // compute units the coming launches may use (256, or a CU-masked stream's share)
char* pick_buffer();
bool pick_only(const char* fmt, ...);   // true (and the name recorded) in pick mode

int launch_halo3x3(const ConvArgs& a, hipStream_t s);   // conv3x3_halo.hip; returns 1 if it does not apply
int launch_t20_3x3(const ConvArgs& a, int force, hipStream_t s);   // conv3x3_t20.hip (20x20 output tiles); 1 if it does not apply
int launch_stream1x1(const ConvArgs& a, int force, hipStream_t s);   // conv1x1_stream.hip (weight-stationary 1x1 on large maps); 1 if it does not apply
int launch_conv1_nchw(const ConvArgs& a, const float* x_nchw, int cin_real, bool pool, hipStream_t s);
int launch_conv1_s2_nchw(const ConvArgs& a, const float* x_nchw, int cin_real, hipStream_t s);   // conv_small.hip; 1 if it does not apply

}  // namespace yolo_conv
