// One MobileNetV2 inverted-residual block per launch (torchvision InvertedResidual as used by
// /root/reference/pytorch_yolo/models/yolov3_tiny_mobilenet.py:14-46):
//     y = [x +] proj1x1( relu6( dw3x3_stride( relu6( expand1x1(x) ) ) ) ),   BN folded into every conv.
// Unfused, the 6x-expanded tensor is written by the expand conv, read and written by the depthwise conv and read
// again by the projection: 13-25 x the block's input + output bytes, and the early blocks (208x208 .. 52x52 maps)
// are pure HBM traffic.  Here a persistent workgroup owns a TH x TW output tile at a time and keeps everything on
// the CU:
//   A  x halo tile ((TH-1)*S+3) x ((TW-1)*S+3) pixels -> LDS (bf16, K padded to 32), next tile's loads already in flight
//   B  expand GEMM on the halo (v_mfma_f32_16x16x32_bf16, K = cin <= 32: one MFMA per 16 pixels x 16 channels),
//      + bias, ReLU6, zero outside the image (the depthwise conv pads the EXPANDED map), bf16 -> LDS E[pixel][ce]
//   C  depthwise 3x3 on E: a thread owns 4 channels (its 36 weights live in registers) and walks pixels, -> LDS D[pixel][ce]
//   D  projection GEMM D x Wp (K = ce), + bias (+ x from the LDS tile), bf16 -> y
// The rounding points are those of the three-launch path (E and D are bf16 there too), so both paths agree to the
// last bit of the bf16 results up to fp32 summation order.
#include "common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef bf16_t bf16x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

struct MbArgs {
  const bf16_t* x;
  bf16_t* y;
  const bf16_t* we;     // [ce][48] bf16 (LDS image), nullptr: no expand conv
  const float* be;      // [ce]
  const float* wd;      // [9][ce]
  const float* bd;      // [ce]
  const bf16_t* wp;     // [cop][dstride/2] bf16 (LDS image)
  const float* bp;      // [cop]
  int n, h, w, ho, wo, cin, in_ct, in_co, ce, cout, cop, out_ct, out_co, has_res;
  int tiles_x, tiles_y, n_tiles, dstride;
  int th;               // strip form: output rows per band
  int debug;            // YOLO_MBCONV_DEBUG (timing only, results wrong): 2 no expand stage, 4 no depthwise stage, 8 no projection stage, 16 no x loads;
                        // form selection: 64 never the strip form (round 5), 1 / 32 tile-shape knobs of the tile form
#ifdef YOLO_STAMPS
  unsigned long long* stamps;   // diagnostic build only (tools/mbstrip_timeline.py)
  int stamp_lds;                // byte offset of the 768-byte stamp area in LDS
#endif
};

// Diagnostic build only (-DYOLO_STAMPS): the lead wave of every role of the strip kernel records the shader clock (s_memtime) at the
// top of intervals 4 .. 11 (0), after the expand role's stash + fetch (1) and when the interval's work is done, in front of the
// barrier (2): [workgroup][role 0..2][interval - 4][4 words].
#ifdef YOLO_STAMPS
// (into LDS, behind the kernel's own map, and copied out when the role has finished: a global store per stamp would sit in the expand
// role's vmcnt queue between its counted loads)
#define MB_STAMP(role, k, which)                                                                                            \
  do {                                                                                                                      \
    if (a.stamps && stamp_lead && (k) >= 4 && (k) < 12)                                                                     \
      reinterpret_cast<unsigned long long*>(smem + a.stamp_lds)[((role)*8 + ((k)-4)) * 4 + (which)] = __builtin_amdgcn_s_memtime(); \
  } while (0)
#define MB_STAMP_FLUSH(role)                                                                                                \
  do {                                                                                                                      \
    if (a.stamps && stamp_lead)                                                                                             \
      for (int i_ = 0; i_ < 32; ++i_)                                                                                       \
        a.stamps[((size_t)blockIdx.x * 3 + (role)) * 32 + i_] = reinterpret_cast<unsigned long long*>(smem + a.stamp_lds)[(role)*32 + i_]; \
  } while (0)
#else
#define MB_STAMP(role, k, which)
#define MB_STAMP_FLUSH(role)
#endif

constexpr int kXStride = 96;      // bytes per pixel row of the x tile / per row of W_expand: 32 bf16 + pad; rows 24 banks
                                  // apart make the 16x16x32 fragment reads (ds_read_b128) conflict-free

__device__ __forceinline__ float relu6(float v) { return __builtin_amdgcn_fmed3f(v, 0.f, 6.f); }   // one instruction

// two bf16 packed in a dword -> two f32 (exact: a bf16 is the high half of its f32)
__device__ __forceinline__ f32x2 bf16pair_to_f32(uint32_t w) {
  return f32x2{__builtin_bit_cast(float, w << 16), __builtin_bit_cast(float, w & 0xffff0000u)};
}

__device__ __forceinline__ void lds_barrier() {          // waits for this wave's LDS operations only: x prefetches and y stores keep flying
#if defined(__HIP_DEVICE_COMPILE__)
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
#endif
}

template <int S, int TH, int TW, bool EXPAND, int NT>
__global__ __launch_bounds__(NT) __attribute__((amdgpu_waves_per_eu(4))) void mbconv_kernel(const MbArgs a) {   // 16 waves per CU
  constexpr int IH = (TH - 1) * S + 3, IW = (TW - 1) * S + 3, HP = IH * IW, HPP = (HP + 15) / 16 * 16, P = TH * TW;
  static_assert(P % 16 == 0, "tile");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  constexpr int nt = NT, nw = NT / 64;                   // 256 / 512 threads (four / two workgroups per CU) or 1024
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // E rows: +8 bytes when the expand epilogue writes them (16 lanes = 16 pixels per ds_write_b64: an odd multiple of
  // 8 bytes apart spreads them over the banks); without an expand conv the rows take 16-byte tile loads
  const int ce = a.ce, estride = ce * 2 + (EXPAND ? 8 : 0), dstride = a.dstride;
  // LDS map: [X tile HPP x 96 B (EXPAND only)] [E: HPP x estride] [D: P x dstride] [W_expand: ce x 96 B] [W_proj: cop x dstride]
  // [b_expand f32 ce] [b_proj f32 cop] [-1e30 f32 ce]
  char* const lx = smem;
  char* const le = lx + (EXPAND ? HPP * kXStride : 0);
  char* const ld = le + HPP * estride;
  char* const lwe = ld + P * dstride;
  char* const lwp = lwe + (EXPAND ? ce * kXStride : 0);
  float* const lbe = reinterpret_cast<float*>(lwp + a.cop * dstride);
  float* const lbp = lbe + ce;
  float* const lbad = lbp + a.cop;                       // [ce] x -1e30 (EXPAND only)

  // ---- once per workgroup: weights -> LDS, zero the K padding of the x tile
  if (EXPAND) {
    for (int i = tid; i < ce * (kXStride / 16); i += nt)
      reinterpret_cast<uint4*>(lwe)[i] = reinterpret_cast<const uint4*>(a.we)[i];
    for (int i = tid; i < HPP * (kXStride / 16); i += nt) reinterpret_cast<uint4*>(lx)[i] = make_uint4(0, 0, 0, 0);
    for (int i = tid; i < ce; i += nt) lbe[i] = a.be[i], lbad[i] = -1e30f;
  } else {
    for (int i = tid; i < HPP * estride / 16; i += nt) reinterpret_cast<uint4*>(le)[i] = make_uint4(0, 0, 0, 0);
  }
  for (int i = tid; i < a.cop * dstride / 16; i += nt)
    reinterpret_cast<uint4*>(lwp)[i] = reinterpret_cast<const uint4*>(a.wp)[i];
  for (int i = tid; i < a.cop; i += nt) lbp[i] = a.bp[i];

  // depthwise: this thread's 4 channels
  const int qn = ce / 4, groups = nt / qn;
  const int qd = tid % qn, grp = tid / qn;
  const bool dw_on = grp < groups;
  f32x2 wdr2[9][2];
  float bdr[4];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int e = 0; e < 4; ++e) wdr2[t][e >> 1][e & 1] = dw_on ? a.wd[t * ce + qd * 4 + e] : 0.f;
#pragma unroll
  for (int e = 0; e < 4; ++e) bdr[e] = dw_on ? a.bd[qd * 4 + e] : 0.f;

  // ---- x tile loads: piece = (halo pixel, 16-byte channel chunk); at most 4 * HP <= NPRE * NT pieces
  constexpr int NPRE = (4 * HP + NT - 1) / NT;
  const int chunks = a.cin / 8, pieces = HP * chunks;
  const int xrow = EXPAND ? kXStride : estride;          // without an expand conv the tile IS E (hidden == cin)
  char* const xdst = EXPAND ? lx : le;
  uint4 pre[NPRE];
  // This thread's pieces, decomposed ONCE (round 5, late: the tile loop divided by the runtime chunk count twice per piece in fetch
  // AND stash, and by the runtime tile counts three times per tile - ~350 of the ~600 instructions a tile costs a wave with all three
  // phases' arithmetic off): LDS byte offset (0xffff: no such piece) | halo row << 16 | halo column << 24, and the element offset
  // from the halo's origin pixel.
  int pc_pos[NPRE], pc_goff[NPRE];
#pragma unroll
  for (int k = 0; k < NPRE; ++k) {
    const int pc = tid + k * nt;
    const int pix = pc / chunks, ch = pc - pix * chunks;
    const int py = pix / IW, px = pix - py * IW;
    pc_pos[k] = (pc < pieces ? pix * xrow + ch * 16 : 0xffff) | py << 16 | px << 24;      // (the x tile is < 64 KB, the halo < 256 wide)
    pc_goff[k] = (py * a.w + px) * a.in_ct + ch * 8;
  }
  // tile cursors: (tx, ty, b) of a tile index advanced by gridDim.x without divisions
  struct Cursor { int tile, tx, ty, b; };
  const int G = (int)gridDim.x;
  const int c_dtx = G % a.tiles_x, c_dr = G / a.tiles_x, c_dty = c_dr % a.tiles_y, c_db = c_dr / a.tiles_y;
  auto cursor_at = [&](int tile) {
    const int tx = tile % a.tiles_x, r = tile / a.tiles_x;
    return Cursor{tile, tx, r % a.tiles_y, r / a.tiles_y};
  };
  auto advance = [&](Cursor& c) {
    c.tile += G;
    c.tx += c_dtx;
    const int w0 = c.tx >= a.tiles_x;
    c.tx -= w0 ? a.tiles_x : 0;
    c.ty += c_dty + w0;
    const int w1 = c.ty >= a.tiles_y;
    c.ty -= w1 ? a.tiles_y : 0;
    c.b += c_db + w1;
  };
  auto fetch = [&](const Cursor& c) {
    const int iy0 = c.ty * TH * S - 1, ix0 = c.tx * TW * S - 1;
    const bf16_t* const base = a.x + ((long)(c.b * a.h + iy0) * a.w + ix0) * a.in_ct + a.in_co;      // (formed, not read, for iy0 / ix0 = -1)
#pragma unroll
    for (int k = 0; k < NPRE; ++k) {
      const int iy = iy0 + ((pc_pos[k] >> 16) & 0xff), ix = ix0 + ((unsigned)pc_pos[k] >> 24);
      pre[k] = make_uint4(0, 0, 0, 0);
      if ((pc_pos[k] & 0xffff) != 0xffff && (unsigned)iy < (unsigned)a.h && (unsigned)ix < (unsigned)a.w) pre[k] = *reinterpret_cast<const uint4*>(base + pc_goff[k]);
    }
  };
  auto stash = [&]() {
#pragma unroll
    for (int k = 0; k < NPRE; ++k)
      if ((pc_pos[k] & 0xffff) != 0xffff) *reinterpret_cast<uint4*>(xdst + (pc_pos[k] & 0xffff)) = pre[k];
  };

  const int c16 = lane & 15, q = lane >> 4;
  Cursor cur = cursor_at((int)blockIdx.x), nxt = cur;     // the tile being computed; the tile whose halo is being fetched
  if (cur.tile < a.n_tiles) fetch(nxt);
#if defined(__HIP_DEVICE_COMPILE__)
  // The depthwise weights are in registers from here on, and hipcc is told so: it cannot count vector-memory operations across the
  // tile loop, and without this a "s_waitcnt vmcnt(0)" stood in front of the first FMA of the depthwise phase of EVERY tile (the
  // registers' loads "may" be outstanding) - draining the next tile's x loads and the y stores there.
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
  for (int t = 0; t < 9; ++t) asm volatile("" : "+v"(wdr2[t][0]), "+v"(wdr2[t][1]));
  asm volatile("" : "+v"(bdr[0]), "+v"(bdr[1]), "+v"(bdr[2]), "+v"(bdr[3]));
#endif
  __syncthreads();
  // Where the x tile goes to LDS (round 5, late): the first tile's here, tile i + 1's inside tile i BETWEEN the depthwise phase
  // and the projection's stores.  It used to be the first thing of a tile, right behind the previous tile's y stores: the wait for
  // the prefetched registers is "vmcnt(0)" (the number of younger stores is not a constant), so every tile began by waiting out the
  // acknowledgement of those stores.  Now the youngest vector-memory operations in front of the wait are
  // the loads themselves, one tile old.  The residual (x at the tile's centre pixels) is read into registers before the x tile is
  // overwritten.
  constexpr int kMaxPT = (16 + nw - 1) / nw;             // projection tiles per wave: P / 16 x cop / 16 <= 16 (launch_nt checks)
  // projection tile t of a wave -> (pixel-row tile, cout tile): cop / 16 is 1, 2 or 4 on every shipped shape (3: cout 33-48)
  const int nct_p = a.cop / 16, nct_sh = nct_p == 4 ? 2 : nct_p == 2 ? 1 : 0;
  auto proj_rt = [&](int t) { return nct_p == 3 ? t / 3 : t >> nct_sh; };
  if (cur.tile < a.n_tiles) {
    stash();
    advance(nxt);
    if (nxt.tile < a.n_tiles && !(a.debug & 16)) fetch(nxt);
    lds_barrier();
  }
  for (; cur.tile < a.n_tiles; advance(cur)) {
    const int b = cur.b;
    const int oy0 = cur.ty * TH, ox0 = cur.tx * TW;
    const int iy0 = oy0 * S - 1, ix0 = ox0 * S - 1;
    // the residual values of this wave's projection tiles
    bf16x4 xres[kMaxPT];
    if (S == 1 && a.has_res) {                             // (a stride-2 block has no residual)
      const int ntl_p = (P / 16) * nct_p;
#pragma unroll
      for (int i = 0; i < kMaxPT; ++i) {
        const int t = wave + i * nw;
        const int rt = proj_rt(t), ct = t - rt * nct_p;
        const int p = rt * 16 + c16, c0 = ct * 16 + q * 4;
        xres[i] = bf16x4{(bf16_t)0.f, (bf16_t)0.f, (bf16_t)0.f, (bf16_t)0.f};
        if (t < ntl_p) xres[i] = *reinterpret_cast<const bf16x4*>(xdst + ((p / TW + 1) * IW + p % TW + 1) * xrow + c0 * 2);
      }
    }
    // ---- B: E = relu6(X We^T + be), 0 outside the image
    if (EXPAND && !(a.debug & 2)) {
      // A wave takes a contiguous run of 16x16 tiles in CHANNEL-tile-major order: the weight fragment and the bias (the accumulator's
      // start value) stay in registers over the run's pixel-row tiles, so a tile costs one 1 KB LDS read (its pixel fragment) instead
      // of three (round 4: the phase was bound by LDS traffic and by a read -> MFMA -> pack -> write chain per tile with a scalar
      // division in front of it).  Out-of-image pixels are zeroed after packing: the depthwise conv pads the EXPANDED map.
      const int nct = ce / 16, ntl = (HPP / 16) * nct, per = (ntl + nw - 1) / nw;
      int t = wave * per;
      const int t_hi = min(ntl, t + per);
      if (t < t_hi) {
        int ct = t / (HPP / 16), rt = t - ct * (HPP / 16);
        bf16x8 wf = *reinterpret_cast<const bf16x8*>(lwe + (ct * 16 + c16) * kXStride + q * 16);
        f32x4 bias = *reinterpret_cast<const f32x4*>(lbe + ct * 16 + q * 4);
        bf16x8 xf = *reinterpret_cast<const bf16x8*>(lx + (rt * 16 + c16) * kXStride + q * 16);
        for (; t < t_hi; ++t) {
          const f32x4 acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf, xf, bias, 0, 0, 0);
          const int pix = rt * 16 + c16, ct_now = ct;
          // the next tile's operands are on their way while this one is packed
          if (++rt == HPP / 16) rt = 0, ++ct;
          if (t + 1 < t_hi) {
            xf = *reinterpret_cast<const bf16x8*>(lx + (rt * 16 + c16) * kXStride + q * 16);
            if (rt == 0) {
              wf = *reinterpret_cast<const bf16x8*>(lwe + (ct * 16 + c16) * kXStride + q * 16);
              bias = *reinterpret_cast<const f32x4*>(lbe + ct * 16 + q * 4);
            }
          }
          const int py = pix / IW, px = pix - py * IW;
          const int iy = iy0 + py, ix = ix0 + px;
          const bool in = (unsigned)iy < (unsigned)a.h && (unsigned)ix < (unsigned)a.w;
          bf16x4 o;                                                          // lane: pixel c16, channels ct*16 + q*4 ..+3
#pragma unroll
          for (int e = 0; e < 4; ++e) o[e] = (bf16_t)relu6(acc[e]);
          u32x2 ow = __builtin_bit_cast(u32x2, o);
          ow[0] = in ? ow[0] : 0u;
          ow[1] = in ? ow[1] : 0u;
          if (pix < HP) *reinterpret_cast<u32x2*>(le + pix * estride + ct_now * 32 + q * 8) = ow;
        }
      }
    }
    if (EXPAND) { if (a.debug & 256) __syncthreads(); else lds_barrier(); }
    // ---- C: D = relu6(dw3x3(E) + bd)
    if constexpr (S == 1 && NT >= 512) {
      if (dw_on && !(a.debug & 4)) {
      // Stride 1 (round 5, late): a task is a run of FOUR adjacent output pixels of a row x 4 channels - its 3 x 6 window of E is read
      // once (18 eight-byte reads, the same as the two-pixel iteration below) and feeds 16 outputs instead of 8: half the LDS bytes and
      // 13 instead of 22 instructions per output.  The phase is bound by LDS reads (the float32-E experiment: half the instructions,
      // twice the bytes, same time), and the tile has fewer such tasks (TH x TW / 4 x ce / 4 = 576 at 144 hidden channels) than the
      // workgroup has threads: one pass.  Every output sums bias, then its nine taps in (dy, dx) order as before: the same numbers.
      constexpr int RUN = 4, RPR = TW / RUN, NRUN = TH * RPR;
      static_assert(TW % RUN == 0, "runs");
      for (int run = grp; run < NRUN; run += groups) {
        const int oy = run / RPR, ox0 = (run - oy * RPR) * RUN;
        const char* const e0 = le + (oy * IW + ox0) * estride + qd * 8;
        f32x2 s[RUN][2];
#pragma unroll
        for (int u = 0; u < RUN; ++u) {
          s[u][0] = f32x2{bdr[0], bdr[1]};
          s[u][1] = f32x2{bdr[2], bdr[3]};
        }
        // one window row in registers at a time (the whole 3 x 6 window at once - 36 registers beside the 40 of the weights and the 16
        // accumulators - spills under the 128-register cap; so does the 256-thread instance with its two sets of x pieces: it keeps
        // the two-pixel form below)
        u32x2 v[2][RUN + 2];
#pragma unroll
        for (int c = 0; c < RUN + 2; ++c) v[0][c] = *reinterpret_cast<const u32x2*>(e0 + c * estride);
#pragma unroll
        for (int dy = 0; dy < 3; ++dy) {
          if (dy < 2) {                                    // the next window row is on its way while this one is multiplied
#pragma unroll
            for (int c = 0; c < RUN + 2; ++c) v[(dy + 1) & 1][c] = *reinterpret_cast<const u32x2*>(e0 + ((dy + 1) * IW + c) * estride);
          }
          f32x2 lo[RUN + 2], hi[RUN + 2];
#pragma unroll
          for (int c = 0; c < RUN + 2; ++c) lo[c] = bf16pair_to_f32(v[dy & 1][c][0]), hi[c] = bf16pair_to_f32(v[dy & 1][c][1]);
#pragma unroll
          for (int dx = 0; dx < 3; ++dx)
#pragma unroll
            for (int u = 0; u < RUN; ++u) {
              s[u][0] = __builtin_elementwise_fma(lo[u + dx], wdr2[dy * 3 + dx][0], s[u][0]);
              s[u][1] = __builtin_elementwise_fma(hi[u + dx], wdr2[dy * 3 + dx][1], s[u][1]);
            }
        }
#pragma unroll
        for (int u = 0; u < RUN; ++u) {
          bf16x4 o;
#pragma unroll
          for (int e = 0; e < 4; ++e) o[e] = (bf16_t)relu6(s[u][e >> 1][e & 1]);
          *reinterpret_cast<bf16x4*>(ld + (oy * TW + ox0 + u) * dstride + qd * 8) = o;
        }
      }
      }
    } else if (dw_on && !(a.debug & 4)) {
      for (int p = grp; p < P; p += 2 * groups) {       // two pixels per iteration
        const bool two = p + groups < P;
        const int pp[2] = {p, two ? p + groups : p};
        const char* e0[2];
        f32x2 s[2][2];                                   // v_pk_fma_f32: two channels per instruction
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          const int oy = pp[u] / TW, ox = pp[u] - oy * TW;
          e0[u] = le + ((oy * S) * IW + ox * S) * estride + qd * 8;
          s[u][0] = f32x2{bdr[0], bdr[1]};
          s[u][1] = f32x2{bdr[2], bdr[3]};
        }
#pragma unroll
        for (int dy = 0; dy < 3; ++dy)
#pragma unroll
          for (int dx = 0; dx < 3; ++dx)
#pragma unroll
            for (int u = 0; u < 2; ++u) {
              const u32x2 v = *reinterpret_cast<const u32x2*>(e0[u] + (dy * IW + dx) * estride);
              s[u][0] = __builtin_elementwise_fma(bf16pair_to_f32(v[0]), wdr2[dy * 3 + dx][0], s[u][0]);
              s[u][1] = __builtin_elementwise_fma(bf16pair_to_f32(v[1]), wdr2[dy * 3 + dx][1], s[u][1]);
            }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          bf16x4 o;
#pragma unroll
          for (int e = 0; e < 4; ++e) o[e] = (bf16_t)relu6(s[u][e >> 1][e & 1]);
          if (u == 0 || two) *reinterpret_cast<bf16x4*>(ld + pp[u] * dstride + qd * 8) = o;
        }
      }
    }
    if (a.debug & 256) __syncthreads(); else lds_barrier();
    // ---- the NEXT tile's x halo (fetched during this tile's first phases) -> LDS; the loads of the tile after it go out
    if (cur.tile + G < a.n_tiles) {
      stash();
      advance(nxt);
      if (nxt.tile < a.n_tiles && !(a.debug & 16)) fetch(nxt);
    }
    // ---- D: y = D Wp^T + bp (+ x)
    if (!(a.debug & 8)) {
      const int nct = nct_p, ntl = (P / 16) * nct, ksteps = ce / 32;
      int ti = 0;
      for (int t = wave; t < ntl; t += nw, ++ti) {
        const int rt = proj_rt(t), ct = t - rt * nct;
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        for (int k = 0; k < ksteps; ++k) {
          const bf16x8 wf = *reinterpret_cast<const bf16x8*>(lwp + (ct * 16 + c16) * dstride + k * 64 + q * 16);
          const bf16x8 df = *reinterpret_cast<const bf16x8*>(ld + (rt * 16 + c16) * dstride + k * 64 + q * 16);
          acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf, df, acc, 0, 0, 0);
        }
        const int p = rt * 16 + c16, c0 = ct * 16 + q * 4;
        const int oy = oy0 + p / TW, ox = ox0 + p % TW;
        if (oy < a.ho && ox < a.wo && c0 < a.cout) {
          const f32x4 bv = *reinterpret_cast<const f32x4*>(lbp + c0);
          float v[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = acc[e] + bv[e];
          if (S == 1 && a.has_res) {        // stride 1, cin == cout: x at the tile's centre pixel (read from the LDS tile at the top of the tile)
            bf16x4 xv = xres[0];
#pragma unroll
            for (int i = 1; i < kMaxPT; ++i) xv = ti == i ? xres[i] : xv;
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] += (float)xv[e];
          }
          bf16x4 o;
#pragma unroll
          for (int e = 0; e < 4; ++e) o[e] = (bf16_t)v[e];
          bf16_t* const yp = a.y + ((long)(b * a.ho + oy) * a.wo + ox) * a.out_ct + a.out_co + c0;
          if (c0 + 4 <= a.cout) *reinterpret_cast<bf16x4*>(yp) = o;
          else
            for (int e = 0; e < 4 && c0 + e < a.cout; ++e) yp[e] = o[e];
        }
      }
    }
    if (a.debug & 256) __syncthreads(); else lds_barrier();     // E and D are free for the next tile (whose x tile is in LDS already)
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// Round 5: the same block as a ROW-STRIP kernel with the three stages on DIFFERENT waves (reference
// yolov3_tiny_mobilenet.py:14-46, blocks 1-6 on the 208x208 .. 52x52 maps at a 416 input).  What the 8x8-tile form above pays on these
// maps: the expand conv runs on a 10x10 halo for 8x8 outputs (1.56 x the work, 1.75 x the x bytes by the counters), and a
// workgroup walks four barrier-separated phases per 64 outputs - no unit saturated, the waves parked 40-55 % of their cycles.
// Here a workgroup (16 waves, one per CU) owns a band of output rows of one column segment (TW = 26 columns: 208 / 104 / 52 = 8 / 4 / 2
// segments) and marches down; every input row is expanded ONCE per segment into a ring of LDS rows (the halo is two columns of 28
// and one or two rows per band).  Between two barriers ("interval" k) the roles work on three different rows:
//   waves 0-5    stash the x rows fetched during the last interval (registers -> LDS, bf16 [pixel][32 channels, K padded]), request the
//                next ones, and EXPAND the rows output row oy0 + k will need: E = relu6(x We^T + be), 0 outside the image (the
//                depthwise conv pads the EXPANDED map); MFMA 16x16x32, two tiles' operands in flight
//   waves 6-13   DEPTHWISE row oy0 + k - 1: D = relu6(dw3x3(E rows) + bd), weights in registers; at stride 1 a thread owns two channels
//                x four pixels and keeps its 3 x 6 window of E values, as fp32, across intervals
//   waves 14-15  PROJECT row oy0 + k - 2: y = D Wp^T + bp (+ x, from the LDS ring), bf16 stores
// ONE LDS-only barrier per output row (the first, single-role version ran three __syncthreads-style phases per row on 26 pixels:
// 3.5 us per row, slower than the tile form).  Rings: E rows 4 (stride 1: three being read, one being written) / 5 (stride 2: three
// read, two written), x rows 5 / 4 (the residual reads the x row three intervals after its expansion), D rows 2.
// Same packed weight images and rounding points (E and D bf16, projection + residual summed in fp32) as the tile form: results agree
// to fp32 summation order (tests/test_gpu_parity.py::test_fused_inverted_residual runs both forms).
template <int S>
struct MbStrip {
  static constexpr int TW = 26, OPS = 32, IW = (TW - 1) * S + 3, IWF = (IW + 15) / 16, IWS = IWF * 16;
  // pixel slots of an E row (the expand writes pixels < IW only; the stride-1 depthwise windows of the last run read up to column 33)
  static constexpr int IWE = S == 1 ? 40 : (IW + 7) / 8 * 8;
  static constexpr int RX = 5, RE = S == 1 ? 4 : 5;         // x rows: read by the expand (interval 0: three), written by the stash, read by the residual
  static constexpr int RNE = S == 1 ? 6 : 8;                // rows of the single ring of a block without expand conv
  // waves of the expand / depthwise / projection roles.  (First split: 3 / 11 / 2 - the role ablation showed the expand role, 6 tiles
  // per wave and interval, bounding block 3 at 0.2 ms with the depthwise role switched off; the register-window depthwise needs few
  // threads.)
  static constexpr int WE = 6, WD = 8, WP = 2, NE = 64 * WE, NDW = 64 * WD;
  static constexpr int NPX = (3 * IW * 4 + NE - 1) / NE;    // x pieces per E-role thread: up to three rows of IW pixels x 4 chunks
};

template <int S, bool EXPAND>
__global__ __launch_bounds__(1024) void mbstrip_kernel(const MbArgs a) {
  using G = MbStrip<S>;
  constexpr int TW = G::TW, OPS = G::OPS, IW = G::IW, IWF = G::IWF, IWS = G::IWS, IWE = G::IWE, RX = G::RX, RE = G::RE, NPX = G::NPX;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ce = a.ce, estride = ce * 2 + (EXPAND ? 8 : 0), dstride = a.dstride;
  // LDS map: [X ring RX x IWS x 96 B (EXPAND)] [E ring RE x IWE x estride] [D: 2 x OPS x dstride] [W_expand ce x 96 B] [W_proj cop x dstride]
  // [b_expand f32 ce] [b_proj f32 cop].  Without an expand conv the x rows ARE the E rows (hidden == cin): one ring of RE + 1 rows.
  constexpr int RXE = EXPAND ? RE : G::RNE;              // (no expand: rows being read (3), written one interval ahead (S) and the residual's row)
  char* const lx = smem;
  char* const le = lx + (EXPAND ? RX * IWS * kXStride : 0);
  char* const ld = le + RXE * IWE * estride;
  char* const lwe = ld + 2 * OPS * dstride;
  char* const lwp = lwe + (EXPAND ? ce * kXStride : 0);
  float* const lbe = reinterpret_cast<float*>(lwp + a.cop * dstride);
  float* const lbp = lbe + ce;
  constexpr int NT = 1024;

  // ---- once per workgroup: weights -> LDS, zero what the intervals never write (K padding of the x rows, D rows beyond TW)
  if (EXPAND) {
    for (int i = tid; i < ce * (kXStride / 16); i += NT) reinterpret_cast<uint4*>(lwe)[i] = reinterpret_cast<const uint4*>(a.we)[i];
    for (int i = tid; i < RX * IWS * (kXStride / 16); i += NT) reinterpret_cast<uint4*>(lx)[i] = make_uint4(0, 0, 0, 0);
    for (int i = tid; i < ce; i += NT) lbe[i] = a.be[i];
  } else {
    for (int i = tid; i < RXE * IWE * estride / 16; i += NT) reinterpret_cast<uint4*>(le)[i] = make_uint4(0, 0, 0, 0);
  }
  for (int i = tid; i < 2 * OPS * dstride / 16; i += NT) reinterpret_cast<uint4*>(ld)[i] = make_uint4(0, 0, 0, 0);
  for (int i = tid; i < a.cop * dstride / 16; i += NT) reinterpret_cast<uint4*>(lwp)[i] = reinterpret_cast<const uint4*>(a.wp)[i];
  for (int i = tid; i < a.cop; i += NT) lbp[i] = a.bp[i];
  if (EXPAND) {      // the expand role's tile table (three rows at most): no division by the runtime channel-tile count in its loop
    const int nct = ce / 16;
    int* const ttab = reinterpret_cast<int*>(lbp + a.cop);
    for (int t = tid; t < 3 * IWF * nct; t += NT) {
      const int ct = t % nct, v = t / nct;
      ttab[t] = ct | (v % IWF) << 8 | (v / IWF) << 16;
    }
  }

  // ---- this workgroup's strip: image b, output columns [ox0, ox0 + TW), output rows [oy0, oy1)
  const int seg = blockIdx.x % a.tiles_x, r_ = blockIdx.x / a.tiles_x, band = r_ % a.tiles_y, b = r_ / a.tiles_y;
  const int ox0 = seg * TW, oy0 = band * a.th, oy1 = min(a.ho, oy0 + a.th), K = oy1 - oy0;
  const int ix0 = ox0 * S - 1;
  const int chunks = a.cin / 8;
  const int xrow = EXPAND ? kXStride : estride;
  const int xpix = EXPAND ? IWS : IWE;                   // pixel slots per x-ring row
  char* const xring = EXPAND ? lx : le;
  constexpr int RXR = EXPAND ? RX : RXE;
  auto xslot = [](int iy) { return ((iy % RXR) + RXR) % RXR; };
  auto eslot = [](int iy) { return ((iy % RXE) + RXE) % RXE; };
  // rows the E role produces for output row q: the prologue rows come with q == oy0
  auto first_new = [&](int q) { return q == oy0 ? oy0 * S - 1 : (S == 1 ? q + 1 : 2 * q); };
  auto count_new = [&](int q) { return q == oy0 ? S + (S == 1 ? 2 : 1) : S; };      // (stride 1: rows oy0-1, oy0, oy0+1; stride 2: 2 oy0 - 1 .. + 1)
  const int c16 = lane & 15, q16 = lane >> 4;
#ifdef YOLO_STAMPS
  const bool stamp_lead = lane == 0 && (wave == 0 || wave == G::WE || wave == G::WE + G::WD);
#endif

  if (wave < G::WE) {
    // ================= E role =================
    const int et = tid;
    // x pieces, THREE register sets in rotation: the rows of output row q are requested three intervals before they are stashed (one
    // interval is 0.3-0.5 us of work, an HBM round trip 1.5-2: with one set - requested in interval k, stashed in k + 1 - the
    // expand role waited out a round trip per row, 0.19 ms for block 1 with every phase switched off)
    uint4 pre[3][NPX];
    // this thread's pieces: (row r of the request, pixel, 16-byte channel chunk) of piece et + u * NE - decomposed ONCE (the first
    // version divided by the runtime chunk count in every fetch and stash: ~400 instructions per interval on the three expand waves,
    // which alone made an interval 1.3 us)
    int pc_r[NPX], pc_goff[NPX], pc_loff[NPX];
    bool pc_col[NPX];
    {
      const int per = IW * chunks;
#pragma unroll
      for (int u = 0; u < NPX; ++u) {
        const int pc = et + u * G::NE;
        const int r = pc / per, rem = pc - r * per, pix = rem / chunks, ch = rem - pix * chunks;
        pc_r[u] = r;                                     // (r >= cnt: not part of this request)
        pc_col[u] = (unsigned)(ix0 + pix) < (unsigned)a.w;
        pc_goff[u] = (ix0 + pix) * a.in_ct + a.in_co + ch * 8;
        pc_loff[u] = pix * xrow + ch * 16;
      }
    }
    const bf16_t* const ximg = a.x + (long)b * a.h * a.w * a.in_ct;
    const int grow = a.w * a.in_ct;
    // The requests are asm loads the compiler does not track, waited for by a COUNTED s_waitcnt in front of the stash: every fetch
    // issues exactly NPX loads per thread (pieces that are not part of the request or lie outside the image read a clamped, valid
    // address and are zeroed at the stash), so "all but the youngest 2 NPX" = this set has landed, the two younger sets keep flying.
    // (With plain loads hipcc placed s_waitcnt vmcnt(0) in front of every stash - it cannot count across the rotating sets - and an
    // interval cost an HBM round trip, 0.6 us, whatever the three roles did.)
    auto fetch = [&](auto sc, int iy, int cnt) {         // x pieces of rows [iy, iy + cnt)
      constexpr int st = decltype(sc)::value;
#pragma unroll
      for (int u = 0; u < NPX; ++u) {
        const int yy = min(max(iy + pc_r[u], 0), a.h - 1);
        const bf16_t* const src = ximg + (long)yy * grow + (pc_col[u] ? pc_goff[u] : a.in_co);
        uint4& dst = pre[st][u];                          // (bound here: asm operands alone do not capture in a lambda)
        (void)src, (void)dst;
#if defined(__HIP_DEVICE_COMPILE__)
        asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(dst) : "v"(src) : "memory");
#endif
      }
    };
    auto stash = [&](auto sc, int iy, int cnt) {
      constexpr int st = decltype(sc)::value;
#if defined(__HIP_DEVICE_COMPILE__)
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * NPX) : "memory");
#endif
#pragma unroll
      for (int u = 0; u < NPX; ++u) {
        const int yy = iy + pc_r[u];
        const bool ok = pc_col[u] && (unsigned)yy < (unsigned)a.h;
        uint4 v = pre[st][u];
        v.x = ok ? v.x : 0u, v.y = ok ? v.y : 0u, v.z = ok ? v.z : 0u, v.w = ok ? v.w : 0u;
        if (pc_r[u] < cnt) *reinterpret_cast<uint4*>(xring + xslot(yy) * xpix * xrow + pc_loff[u]) = v;
      }
    };
    // E[rows iy .. iy + cnt) = relu6(X We^T + be), 0 outside the image; tiles (row, 16 pixels, 16 channels) round-robin over the role's waves
    auto expand = [&](int iy, int cnt) {
      const int nct = ce / 16, ntl = cnt * IWF * nct;
      const int* const ttab = reinterpret_cast<const int*>(lbp + a.cop);      // tile -> channel tile | pixel tile << 8 | row << 16
      constexpr int NB = 3;                              // operands of NB tiles are requested before the first MFMA
      for (int t0 = wave; t0 < ntl; t0 += NB * G::WE) {
        bf16x8 wf[NB], xf[NB];
        f32x4 acc[NB];
        int pixv[NB], ctv[NB], rv[NB];
#pragma unroll
        for (int u = 0; u < NB; ++u) {
          const int t = min(t0 + u * G::WE, ntl - 1);    // (a trip's surplus tile repeats the last one and is not written)
          const int te = __builtin_amdgcn_readfirstlane(ttab[t]);
          const int ct = te & 255, rt = (te >> 8) & 255, r = te >> 16;
          ctv[u] = ct, rv[u] = r, pixv[u] = rt * 16 + c16;
          wf[u] = *reinterpret_cast<const bf16x8*>(lwe + (ct * 16 + c16) * kXStride + q16 * 16);
          xf[u] = *reinterpret_cast<const bf16x8*>(lx + (xslot(iy + r) * IWS + rt * 16 + c16) * kXStride + q16 * 16);
          acc[u] = *reinterpret_cast<const f32x4*>(lbe + ct * 16 + q16 * 4);
        }
#pragma unroll
        for (int u = 0; u < NB; ++u) acc[u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[u], xf[u], acc[u], 0, 0, 0);
#pragma unroll
        for (int u = 0; u < NB; ++u) {
          const int yy = iy + rv[u], xx = ix0 + pixv[u];
          const bool in = (unsigned)yy < (unsigned)a.h && (unsigned)xx < (unsigned)a.w;
          bf16x4 o;                                                          // lane: pixel c16, channels ct*16 + q*4 ..+3
#pragma unroll
          for (int e = 0; e < 4; ++e) o[e] = (bf16_t)relu6(acc[u][e]);
          u32x2 ow = __builtin_bit_cast(u32x2, o);
          ow[0] = in ? ow[0] : 0u;
          ow[1] = in ? ow[1] : 0u;
          if (t0 + u * G::WE < ntl && pixv[u] < IW) *reinterpret_cast<u32x2*>(le + (eslot(yy) * IWE + pixv[u]) * estride + ctv[u] * 32 + q16 * 8) = ow;
        }
      }
    };
    // interval -1: the x rows of output row oy0 go straight to LDS, those of oy0 + 1 .. + 3 into the three register sets
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    using I2 = std::integral_constant<int, 2>;
    // (every fetch is issued whether its rows exist or not - beyond the band they read clamped rows that are never stashed -: the
    // counted wait needs the same number of requests in flight at every stash)
    fetch(I0{}, first_new(oy0), count_new(oy0));
    fetch(I1{}, first_new(oy0), 0);                      // two dummy sets in front, so that the first stash counts like the others
    fetch(I2{}, first_new(oy0), 0);
    __syncthreads();                                     // (the weights and the zero fill are in LDS; every role passes this barrier)
    {
#if defined(__HIP_DEVICE_COMPILE__)
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
      stash(I0{}, first_new(oy0), count_new(oy0));
    }
    fetch(I1{}, first_new(oy0 + 1), count_new(oy0 + 1));                 // set (q - oy0) % 3 holds the rows of output row q
    fetch(I2{}, first_new(oy0 + 2), count_new(oy0 + 2));
    fetch(I0{}, first_new(oy0 + 3), count_new(oy0 + 3));
    lds_barrier();
    // (three intervals per trip: the register set of every stash / fetch is a constant)
    auto interval = [&](auto sc, int k) {
      const int qn = oy0 + k;                            // expand for output row qn; stash the rows of qn + 1; request those of qn + 4
      MB_STAMP(0, k, 0);
      if (k + 1 < K) stash(sc, first_new(qn + 1), count_new(qn + 1));
      else {
#if defined(__HIP_DEVICE_COMPILE__)
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * NPX) : "memory");      // (the set is overwritten by the fetch below: its old loads must have landed)
#endif
      }
      fetch(sc, first_new(qn + 4), count_new(qn + 4));
      MB_STAMP(0, k, 1);
      if (EXPAND && k < K && !(a.debug & 2)) expand(first_new(qn), count_new(qn));
      MB_STAMP(0, k, 2);
      lds_barrier();
    };
    for (int k = 0; k <= K + 1; k += 3) {                // set of interval k: (k + 1) % 3
      interval(I1{}, k);
      if (k + 1 <= K + 1) interval(I2{}, k + 1);
      if (k + 2 <= K + 1) interval(I0{}, k + 2);
    }
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // (the dummy requests behind the band)
#endif
    MB_STAMP_FLUSH(0);
  } else if (wave < G::WE + G::WD) {
    // ================= depthwise role =================
    const int dt = tid - G::NE;
    if constexpr (S == 1) {
      // Stride 1: a thread owns a channel PAIR x a run of FOUR output pixels and keeps the 3 x 6 window of E values it needs as fp32
      // in registers ACROSS intervals: per output row it reads and unpacks ONE new row of 6 pixels (6 LDS reads, 12 unpack operations)
      // and issues 36 packed FMAs for 4 pixels x 2 channels - the tap-by-tap form (stride 2 below, and the tile kernel) re-reads and
      // re-unpacks every E value for each of the 9 outputs that use it: ~63 instead of ~15 vector instructions per pixel and four
      // channels, and the block is bound by exactly that instruction count (VALU 0.6 of the issue cycles: profiles/
      // r04_mobile_sq_counters.md).  Window row slots rotate with (oy - oy0) mod 3: the row body is instantiated three times so
      // that every register index is a constant.
      constexpr int RUN = 4, NCOL = RUN + 2;             // output pixels per thread and interval; window columns (6 or 8 pixels: the window spills)
      const int ncp = ce / 2;
      const int cp = dt % ncp, run = dt / ncp, p0 = RUN * run;
      const bool on = run < (TW + RUN - 1) / RUN;
      f32x2 w2[9], b2 = {0.f, 0.f};
#pragma unroll
      for (int t = 0; t < 9; ++t) w2[t] = on ? f32x2{a.wd[t * ce + 2 * cp], a.wd[t * ce + 2 * cp + 1]} : f32x2{0.f, 0.f};
      if (on) b2 = f32x2{a.bd[2 * cp], a.bd[2 * cp + 1]};
      f32x2 win[3][NCOL];                                // [row slot][column p0 + c]
      auto load_row = [&](auto slc, int iy) {
        constexpr int sl = decltype(slc)::value;
        const char* const er = le + eslot(iy) * IWE * estride + p0 * estride + cp * 4;
#pragma unroll
        for (int c = 0; c < NCOL; ++c) win[sl][c] = bf16pair_to_f32(*reinterpret_cast<const uint32_t*>(er + c * estride));
      };
      auto dw_row = [&](auto mc, int k) {
        constexpr int M = decltype(mc)::value;           // (oy - oy0) mod 3: E row oy - 1 + dy sits in slot (M + dy) % 3
        const int oy = oy0 + k - 1;
        if (k == 1) {
          load_row(std::integral_constant<int, 0>{}, oy - 1);
          load_row(std::integral_constant<int, 1>{}, oy);
        }
        load_row(std::integral_constant<int, (M + 2) % 3>{}, oy + 1);
        char* const dbuf = ld + (k & 1) * OPS * dstride + p0 * dstride + cp * 4;
#pragma unroll
        for (int u = 0; u < RUN; ++u) {
          f32x2 acc = b2;
#pragma unroll
          for (int dy = 0; dy < 3; ++dy)
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) acc = __builtin_elementwise_fma(win[(M + dy) % 3][u + dx], w2[dy * 3 + dx], acc);
          typedef bf16_t bf16x2_t __attribute__((ext_vector_type(2)));
          bf16x2_t o;
          o[0] = (bf16_t)relu6(acc[0]), o[1] = (bf16_t)relu6(acc[1]);
          *reinterpret_cast<bf16x2_t*>(dbuf + u * dstride) = o;
        }
      };
      __syncthreads();
      lds_barrier();
      for (int k = 0; k <= K + 1; ++k) {
        MB_STAMP(1, k, 0);
        if (k >= 1 && k <= K && on && !(a.debug & 4)) {
          const int m = (k - 1) % 3;
          if (m == 0) dw_row(std::integral_constant<int, 0>{}, k);
          else if (m == 1) dw_row(std::integral_constant<int, 1>{}, k);
          else dw_row(std::integral_constant<int, 2>{}, k);
        }
        MB_STAMP(1, k, 2);
        lds_barrier();
      }
      MB_STAMP_FLUSH(1);
    } else {
      const int qn = ce / 4, groups = G::NDW / qn;
      const int qd = dt % qn, grp = dt / qn;
      const bool dw_on = grp < groups;
      f32x2 wdr2[9][2];
      float bdr[4];
#pragma unroll
      for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int e = 0; e < 4; ++e) wdr2[t][e >> 1][e & 1] = dw_on ? a.wd[t * ce + qd * 4 + e] : 0.f;
#pragma unroll
      for (int e = 0; e < 4; ++e) bdr[e] = dw_on ? a.bd[qd * 4 + e] : 0.f;
      __syncthreads();
      lds_barrier();
      for (int k = 0; k <= K + 1; ++k) {
        const int oy = oy0 + k - 1;                      // D[k & 1] = relu6(dw3x3(E) + bd) for the TW pixels of output row oy
        MB_STAMP(1, k, 0);
        if (k >= 1 && k <= K && dw_on && !(a.debug & 4)) {
          const int iy0 = oy * S - 1;
          char* const dbuf = ld + (k & 1) * OPS * dstride;
          const char* const er[3] = {le + eslot(iy0) * IWE * estride, le + eslot(iy0 + 1) * IWE * estride, le + eslot(iy0 + 2) * IWE * estride};
          for (int p = grp; p < TW; p += 2 * groups) {  // two pixels per iteration
            const bool two = p + groups < TW;
            const int pp[2] = {p, two ? p + groups : p};
            f32x2 s2[2][2];                              // v_pk_fma_f32: two channels per instruction
#pragma unroll
            for (int u = 0; u < 2; ++u) {
              s2[u][0] = f32x2{bdr[0], bdr[1]};
              s2[u][1] = f32x2{bdr[2], bdr[3]};
            }
#pragma unroll
            for (int dy = 0; dy < 3; ++dy)
#pragma unroll
              for (int dx = 0; dx < 3; ++dx)
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                  const u32x2 v = *reinterpret_cast<const u32x2*>(er[dy] + (pp[u] * S + dx) * estride + qd * 8);
                  s2[u][0] = __builtin_elementwise_fma(bf16pair_to_f32(v[0]), wdr2[dy * 3 + dx][0], s2[u][0]);
                  s2[u][1] = __builtin_elementwise_fma(bf16pair_to_f32(v[1]), wdr2[dy * 3 + dx][1], s2[u][1]);
                }
#pragma unroll
            for (int u = 0; u < 2; ++u) {
              bf16x4 o;
#pragma unroll
              for (int e = 0; e < 4; ++e) o[e] = (bf16_t)relu6(s2[u][e >> 1][e & 1]);
              if (u == 0 || two) *reinterpret_cast<bf16x4*>(dbuf + pp[u] * dstride + qd * 8) = o;
            }
          }
        }
        MB_STAMP(1, k, 2);
        lds_barrier();
      }
      MB_STAMP_FLUSH(1);
    }
  } else {
    // ================= projection role =================
    const int pw = wave - (G::WE + G::WD);
    __syncthreads();
    lds_barrier();
    for (int k = 0; k <= K + 1; ++k) {
      const int oy = oy0 + k - 2;                        // y[oy] = D[(k - 1) & 1] Wp^T + bp (+ x)
      MB_STAMP(2, k, 0);
      if (k >= 2 && !(a.debug & 8)) {
        const char* const dbuf = ld + ((k - 1) & 1) * OPS * dstride;
        const int nct = a.cop / 16, ntl = (OPS / 16) * nct, ksteps = ce / 32;
        for (int t = pw; t < ntl; t += G::WP) {
          const int rt = t / nct, ct = t - rt * nct;
          // all of the tile's operands are requested before the first MFMA (K = ce <= 192: six steps at most); read one step at a time
          // the chain was an LDS round trip + a dependent MFMA per step, ~1,000 cycles per tile on a role of two waves
          f32x4 acc = {0.f, 0.f, 0.f, 0.f};
          bf16x8 wfv[6], dfv[6];
#pragma unroll
          for (int kk = 0; kk < 6; ++kk) {
            const int kq = kk < ksteps ? kk : 0;
            wfv[kk] = *reinterpret_cast<const bf16x8*>(lwp + (ct * 16 + c16) * dstride + kq * 64 + q16 * 16);
            dfv[kk] = *reinterpret_cast<const bf16x8*>(dbuf + (rt * 16 + c16) * dstride + kq * 64 + q16 * 16);
          }
#pragma unroll
          for (int kk = 0; kk < 6; ++kk)
            if (kk < ksteps) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wfv[kk], dfv[kk], acc, 0, 0, 0);
          const int p = rt * 16 + c16, c0 = ct * 16 + q16 * 4;
          const int ox = ox0 + p;
          if (p < TW && ox < a.wo && c0 < a.cout) {
            const f32x4 bv = *reinterpret_cast<const f32x4*>(lbp + c0);
            float v[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = acc[e] + bv[e];
            if (a.has_res) {        // stride 1, cin == cout: x at the output pixel = ring row oy, column p + 1
              const bf16x4 xv = *reinterpret_cast<const bf16x4*>(xring + (xslot(oy) * xpix + p + 1) * xrow + c0 * 2);
#pragma unroll
              for (int e = 0; e < 4; ++e) v[e] += (float)xv[e];
            }
            bf16x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = (bf16_t)v[e];
            bf16_t* const yp = a.y + ((long)(b * a.ho + oy) * a.wo + ox) * a.out_ct + a.out_co + c0;
            if (c0 + 4 <= a.cout) *reinterpret_cast<bf16x4*>(yp) = o;
            else
              for (int e = 0; e < 4 && c0 + e < a.cout; ++e) yp[e] = o[e];
          }
        }
      }
      MB_STAMP(2, k, 2);
      lds_barrier();
    }
    MB_STAMP_FLUSH(2);
  }
}

template <int S, bool EXPAND>
size_t strip_lds_bytes(const MbArgs& a) {
  using G = MbStrip<S>;
  const size_t estride = a.ce * 2 + (EXPAND ? 8 : 0);
  return (size_t)(EXPAND ? G::RX * G::IWS * kXStride : 0) + (size_t)(EXPAND ? G::RE : G::RNE) * G::IWE * estride + (size_t)2 * G::OPS * a.dstride +
         (size_t)(EXPAND ? a.ce * kXStride : 0) + (size_t)a.cop * a.dstride + (size_t)(a.ce + a.cop) * 4 +
         (size_t)(EXPAND ? 3 * G::IWF * (a.ce / 16) * 4 : 0);
}

// 1: the strip form does not take the block (the caller goes on to the tile form)
template <int S, bool EXPAND>
int launch_strip(const MbArgs& a0, hipStream_t s) {
  MbArgs a = a0;
  using G = MbStrip<S>;
  // (depthwise role: 512 threads; the stride-1 form needs seven four-pixel runs per channel pair: hidden <= 146)
  if (a.cin % 8 || a.cin > 32 || 3 * G::IW * (a.cin / 8) > G::NPX * G::NE || a.ce % 32 || a.cop > 64 || (S == 1 ? (a.ce / 2) * 7 : a.ce / 4) > G::NDW) return 1;
  const size_t lds = strip_lds_bytes<S, EXPAND>(a);
  if (lds > 160 * 1024) return 1;
  a.tiles_x = (a.wo + G::TW - 1) / G::TW;                    // column segments
  // one 16-wave workgroup per CU: bands so that the grid is about four rounds of the chip, at least 8 rows each (a band re-expands
  // S + 1 input rows); narrow maps (few strips of few rows) stay with the tile form
  const long strips = (long)a.n * a.tiles_x;
  int bands = (int)((4 * 256 + strips - 1) / strips);
  if (bands < 1) bands = 1;
  a.th = (a.ho + bands - 1) / bands;
  if (a.th < 8) a.th = a.ho < 8 ? a.ho : 8;
  a.tiles_y = (a.ho + a.th - 1) / a.th;
  a.n_tiles = a.n * a.tiles_x * a.tiles_y;
#ifdef YOLO_STAMPS
  a.stamps = getenv("YOLO_STAMP_PTR") ? (unsigned long long*)strtoull(getenv("YOLO_STAMP_PTR"), nullptr, 0) : nullptr;
  a.stamp_lds = (int)((lds + 15) / 16 * 16);
  if (a.stamp_lds + 768 > 160 * 1024) a.stamps = nullptr;
  const size_t lds_launch = a.stamps ? (size_t)a.stamp_lds + 768 : lds;
#else
  const size_t lds_launch = lds;
#endif
  static std::atomic<uint64_t> lds_set{0};                 // per device (common.h)
  if (const int rc = yolo_max_dyn_lds(reinterpret_cast<const void*>(&mbstrip_kernel<S, EXPAND>), 160 * 1024, lds_set, "mbconv (strip)")) return rc;
  hipLaunchKernelGGL((mbstrip_kernel<S, EXPAND>), dim3((unsigned)a.n_tiles), dim3(1024), lds_launch, s, a);
  return yolo_check_launch("yolo_mbconv_fwd (strip)");
}

template <int S, int TH, int TW, bool EXPAND, int NT>
int launch_nt(const MbArgs& a, size_t lds, int slots, hipStream_t s) {
  static std::atomic<uint64_t> lds_set{0};                 // per device (common.h)
  if (const int rc = yolo_max_dyn_lds(reinterpret_cast<const void*>(&mbconv_kernel<S, TH, TW, EXPAND, NT>), 160 * 1024, lds_set, "mbconv")) return rc;
  const int grid = a.n_tiles < slots ? a.n_tiles : slots;
  hipLaunchKernelGGL((mbconv_kernel<S, TH, TW, EXPAND, NT>), dim3((unsigned)grid), dim3(NT), lds, s, a);
  return yolo_check_launch("yolo_mbconv_fwd");
}

template <int S, int TH, int TW, bool EXPAND>
size_t lds_bytes(const MbArgs& a) {
  constexpr int IH = (TH - 1) * S + 3, IW = (TW - 1) * S + 3, HP = IH * IW, HPP = (HP + 15) / 16 * 16, P = TH * TW;
  return (size_t)(EXPAND ? HPP * kXStride : 0) + (size_t)HPP * (a.ce * 2 + (EXPAND ? 8 : 0)) + (size_t)P * a.dstride +
         (size_t)(EXPAND ? a.ce * kXStride : 0) + (size_t)a.cop * a.dstride + (size_t)(a.ce + a.cop + (EXPAND ? a.ce : 0)) * 4;
}

template <int S, int TH, int TW, bool EXPAND>
int launch(const MbArgs& a0, hipStream_t s) {
  MbArgs a = a0;
  a.tiles_x = (a.wo + TW - 1) / TW;
  a.tiles_y = (a.ho + TH - 1) / TH;
  a.n_tiles = a.n * a.tiles_x * a.tiles_y;
  const size_t lds = lds_bytes<S, TH, TW, EXPAND>(a);
  YOLO_REQUIRE(lds <= 160 * 1024, "mbconv: %zu bytes of LDS needed", lds);
  YOLO_REQUIRE((TH * TW / 16) * (a.cop / 16) <= 16, "mbconv: %d projection tiles per output tile (the kernel keeps <= 16 residual fragments)", (TH * TW / 16) * (a.cop / 16));
  // persistent workgroups; the phases of one tile (load, expand, depthwise, project) are latency chains, so a CU
  // needs more than 8 waves in flight: two 512-thread workgroups per CU when the LDS allows, else one of 1024 threads
  if (lds <= 40 * 1024) return launch_nt<S, TH, TW, EXPAND, 256>(a, lds, 1024, s);   // four per CU: phases of four tiles overlap
  if (lds <= 80 * 1024) return launch_nt<S, TH, TW, EXPAND, 512>(a, lds, 512, s);
  return launch_nt<S, TH, TW, EXPAND, 1024>(a, lds, 256, s);
}

}  // namespace

// YOLO_MBCONV_DEBUG / yolo_set_tuning(4, .): bit 1 never halve the tile, 64 never the strip form (tests run both forms), 2 / 4 / 8 / 16
// timing ablations (results wrong)
int& yolo_conv_mb_debug() {
  static int v = getenv("YOLO_MBCONV_DEBUG") ? atoi(getenv("YOLO_MBCONV_DEBUG")) : 0;
  return v;
}
#define conv_mb_debug yolo_conv_mb_debug()

// bytes per row of the depthwise-output tile and of W_proj: >= 2*ce, a multiple of 16 and == 96 or 160 (mod 256),
// which spreads the 16 rows of an MFMA fragment read over all banks
extern "C" int yolo_mbconv_dstride(int ce) {
  int s = ce * 2;
  while (s % 256 != 96 && s % 256 != 160) s += 16;
  return s;
}

// the wide blocks (hidden dimension streamed in chunks): conv_mbwide.hip
int yolo_mbwide_supported(int cin, int hidden, int cout, int stride);
int yolo_mbwide_launch(const void* x, const void* w_exp, const float* b_exp, const float* w_dw, const float* b_dw, const void* w_proj,
                       const float* b_proj, void* y, const YoloMbconvDesc& d, hipStream_t st);

// 0: not covered; 1: the whole hidden tile in LDS (this file); 2: the wide form (conv_mbwide.hip) - the two take different weight images
extern "C" int yolo_mbconv_supported(int cin, int hidden, int cout, int stride) {
  const int ce = (hidden + 31) / 32 * 32;
  if (cin >= 8 && cin <= 32 && cin % 8 == 0 && hidden % 4 == 0 && hidden >= cin && ce <= 192 && cout >= 4 && cout <= 64 &&
      cout % 4 == 0 && (stride == 1 || stride == 2))
    return 1;
  return yolo_mbwide_supported(cin, hidden, cout, stride) ? 2 : 0;
}

extern "C" int yolo_mbconv_fwd(const void* x, const void* w_exp, const float* b_exp, const float* w_dw, const float* b_dw,
                               const void* w_proj, const float* b_proj, void* y, const YoloMbconvDesc* dp, yolo_stream_t s) {
  YOLO_REQUIRE(x && w_dw && b_dw && w_proj && b_proj && y && dp, "mbconv: null argument");
  const YoloMbconvDesc& d = *dp;
  YOLO_REQUIRE(yolo_mbconv_supported(d.cin, d.hidden, d.cout, d.stride), "mbconv: cin %d hidden %d cout %d stride %d not covered",
               d.cin, d.hidden, d.cout, d.stride);
  YOLO_REQUIRE(d.has_expand ? (w_exp && b_exp) : d.hidden == d.cin, "mbconv: a block without expand conv has hidden == cin");
  YOLO_REQUIRE(!d.has_res || (d.stride == 1 && d.cin == d.cout), "mbconv: residual needs stride 1 and cin == cout");
  YOLO_REQUIRE(d.in_c_offset % 8 == 0 && d.in_c_total % 8 == 0 && d.in_c_offset + d.cin <= d.in_c_total, "mbconv: bad input view");
  YOLO_REQUIRE(d.out_c_offset % 4 == 0 && d.out_c_total % 4 == 0 && d.out_c_offset + d.cout <= d.out_c_total, "mbconv: bad output view");
  YOLO_REQUIRE(d.n > 0 && d.h > 0 && d.w > 0, "mbconv: empty input");
  if (yolo_mbconv_supported(d.cin, d.hidden, d.cout, d.stride) == 2) {
    YOLO_REQUIRE(d.has_expand, "mbconv: the wide form needs the expand conv");
    return yolo_mbwide_launch(x, w_exp, b_exp, w_dw, b_dw, w_proj, b_proj, y, d, (hipStream_t)s);
  }
  MbArgs a;
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
  a.ce = (d.hidden + 31) / 32 * 32;
  a.cout = d.cout;
  a.cop = (d.cout + 15) / 16 * 16;
  a.out_ct = d.out_c_total;
  a.out_co = d.out_c_offset;
  a.has_res = d.has_res;
  a.dstride = yolo_mbconv_dstride(a.ce);
  a.tiles_x = a.tiles_y = a.n_tiles = a.th = 0;
  a.debug = conv_mb_debug;
  hipStream_t st = (hipStream_t)s;
  if (conv_mb_debug & 128) {        // round 5: the row-strip form, OPT-IN (measured slower than the tile form on every block: DESIGN.md
                                    // Appendix A; tests run both); 1: it does not take the block
    const int rc = d.stride == 1 ? (d.has_expand ? launch_strip<1, true>(a, st) : launch_strip<1, false>(a, st))
                                 : (d.has_expand ? launch_strip<2, true>(a, st) : launch_strip<2, false>(a, st));
    if (rc != 1) return rc;
  }
  // tile: 8x8 outputs (4x8 at stride 2).  192 hidden channels at stride 1: 4x8, which lets two 512-thread workgroups
  // share a CU instead of one of 1024 threads (-10 %; with 144 hidden channels and at stride 2 the larger halo
  // share of a half tile costs more than the overlap gains: measured, YOLO_MBCONV_DEBUG bit 1 = always the full tile)
  if (d.stride == 1) {
    if (!d.has_expand) return launch<1, 8, 8, false>(a, st);
    if ((a.ce >= 192 || (conv_mb_debug & 32)) && !(conv_mb_debug & 1) && lds_bytes<1, 4, 8, true>(a) <= 80 * 1024) return launch<1, 4, 8, true>(a, st);
    return launch<1, 8, 8, true>(a, st);
  }
  return d.has_expand ? launch<2, 4, 8, true>(a, st) : launch<2, 4, 8, false>(a, st);
}
