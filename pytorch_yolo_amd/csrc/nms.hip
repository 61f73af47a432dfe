// Batched confidence-weighted "MERGE" NMS for gfx950 (wave64 primitives, LDS sort).
//
// Replaces the Python loops of non_max_suppression (reference utils/utils.py:200-293, style 'MERGE'
// :240,:266-275) with xywh2xyxy (:46-60) and bbox_iou (:63-96).
//
// Bit-exactness contract: every floating-point step that decides WHICH rows survive (conf product,
// thresholds, box corners, IoU) is a single IEEE fp32 operation in the reference's order — this file is
// compiled with -ffp-contract=off and uses correctly rounded division — so kept-index sets equal the
// CPU oracle's exactly.  The merged box is accumulated sequentially in candidate order (oracle/nms.py).
//
// Two kernels:
//   nms_filter : all rows, HBM-bound.  Row tiles are staged in LDS with coalesced loads, one lane then
//                owns one row: class max/argmax, conf = obj*cls, thresholds, finite check; survivors are
//                appended (wave-aggregated atomic) as 64-bit sort keys  class | ~conf | row.
//   nms_merge  : one 1024-thread workgroup per image: bitonic sort of the keys (LDS up to 8192 keys,
//                global workspace beyond), class segments, one wave per class runs the sequential
//                MERGE over its first max_per_class rows with ballot masks, final sort by conf.
#include "nms_common.h"

namespace {

using namespace yolo_nms;

constexpr int kMergeThreads = 1024;
constexpr int kFilterRows = 64;

__device__ __forceinline__ float readlane_f(float v, int l) {
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l));
}

// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void nms_filter_kernel(float* __restrict__ pred, int rows, int no, float conf_thres,
                                                         float min_wh, int mutate, u64* __restrict__ keys,
                                                         long key_pitch, int* __restrict__ counts) {
  extern __shared__ __attribute__((aligned(16))) char lds_raw[];
  float* tile = reinterpret_cast<float*>(lds_raw);
  const int b = blockIdx.y;
  const int r0 = blockIdx.x * kFilterRows;
  const int nrows = min(kFilterRows, rows - r0);
  float* src = pred + ((long)b * rows + r0) * no;
  const int nflt = nrows * no;
  if ((reinterpret_cast<uintptr_t>(src) & 15) == 0) {       // whole tiles of 16-byte aligned rows: 16 B per lane
    const int n4 = nflt >> 2;
    const float4* src4 = reinterpret_cast<const float4*>(src);
    float4* tile4 = reinterpret_cast<float4*>(tile);
    for (int i = threadIdx.x; i < n4; i += 256) tile4[i] = src4[i];
    for (int i = (n4 << 2) + threadIdx.x; i < nflt; i += 256) tile[i] = src[i];
  } else {
    for (int i = threadIdx.x; i < nflt; i += 256) tile[i] = src[i];
  }
  __syncthreads();
  // four lanes per row, each scanning a quarter of the classes in order; the quarters are then merged in order,
  // which reproduces the sequential scan exactly: first maximum wins, the first NaN poisons the result
  // (torch.max semantics, utils.py:212)
  const int rloc = threadIdx.x >> 2, seg = threadIdx.x & 3;
  const int nc = no - 5, cps = (nc + 3) >> 2;
  const int k0 = seg * cps, k1 = min(nc, k0 + cps);
  const bool live = rloc < nrows;
  const float* row = tile + (live ? rloc : 0) * no;
  bool have = k0 < k1;
  float best = have ? row[5 + k0] : 0.f;
  int arg = k0;
  bool all_finite = !have || finite_f(best);
  for (int k = k0 + 1; k < k1; ++k) {
    const float v = row[5 + k];
    all_finite = all_finite && finite_f(v);
    if (v > best || (v != v && best == best)) {
      best = v;
      arg = k;
    }
  }
#pragma unroll
  for (int m = 1; m <= 2; m <<= 1) {
    const float ob = __shfl_xor(best, m);
    const int oa = __shfl_xor(arg, m);
    const bool oh = __shfl_xor((int)have, m) != 0;
    const bool of = __shfl_xor((int)all_finite, m) != 0;
    const bool other_is_right = (seg & m) == 0;
    // (lb, la) = the earlier quarter(s), (rb, ra) = the later ones
    const float lb = other_is_right ? best : ob, rb = other_is_right ? ob : best;
    const int la = other_is_right ? arg : oa, ra = other_is_right ? oa : arg;
    const bool lh = other_is_right ? have : oh, rh = other_is_right ? oh : have;
    const bool take_r = rh && (!lh || rb > lb || (rb != rb && lb == lb));
    best = take_r ? rb : lb;
    arg = take_r ? ra : la;
    have = lh || rh;
    all_finite = all_finite && of;
  }
  // survivors are appended with ONE global atomic per block: thousands of same-address atomics per image (one per
  // surviving row) serialise in L2 and were most of this kernel's time.  The slot order inside an image is free —
  // nms_merge sorts the keys, and a key carries its row.
  __shared__ int s_cnt, s_base;
  if (threadIdx.x == 0) s_cnt = 0;
  __syncthreads();
  bool keep = false;
  float conf = 0.f;
  int pos = 0;
  if (live && seg == 0) {
    conf = row[4] * best;                                                        // utils.py:213
    if (mutate) src[rloc * no + 4] = conf;
    const float bw = row[2], bh = row[3];
    all_finite = all_finite && finite_f(row[0]) && finite_f(row[1]) && finite_f(bw) && finite_f(bh) && finite_f(conf);
    keep = (conf > conf_thres) && (bw > min_wh) && (bh > min_wh) && all_finite;  // :216-218
    if (keep) pos = atomicAdd(&s_cnt, 1);                                        // LDS atomic
  }
  __syncthreads();
  if (threadIdx.x == 0 && s_cnt > 0) s_base = atomicAdd(&counts[b], s_cnt);
  __syncthreads();
  if (keep) keys[(long)b * key_pitch + s_base + pos] = make_key(arg, conf, r0 + rloc);
}

// ---------------------------------------------------------------------------------------------------
// Bitonic sort of n_pad (power of two) keys by the whole block.  Thread t owns pair slots t, t + 1024, ...; a wave's 64
// consecutive slots touch one private 128-key chunk in every stage whose stride is <= 64, so in LDS those stages need
// no block barrier (a wave's LDS instructions execute in order): of the 78 stages of a 4096-key sort only 15 keep
// their __syncthreads().  `wave_local_ok` is false for keys in global memory (no such ordering there).
__device__ void block_bitonic_sort(u64* keys, int n_pad, bool wave_local_ok) {
  const int pairs = n_pad >> 1;
  bool prev_local = false;
  for (int k = 2; k <= n_pad; k <<= 1) {
    for (int j = k >> 1, lj = 31 - __builtin_clz(j); j > 0; j >>= 1, --lj) {
      const bool local = wave_local_ok && j <= 64;
      if (!local && prev_local) __syncthreads();          // the other waves' private stages must be complete
      for (int t = threadIdx.x; t < pairs; t += kMergeThreads) {
        const int i = ((t >> lj) << (lj + 1)) + (t & (j - 1));
        const int l = i + j;
        const bool up = (i & k) == 0;
        const u64 a = keys[i], c = keys[l];
        if ((a > c) == up) {
          keys[i] = c;
          keys[l] = a;
        }
      }
      if (local) {
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      } else {
        __syncthreads();
      }
      prev_local = local;
    }
  }
  if (prev_local) __syncthreads();
}

// Sort of 1024 * KPT keys held in LDS, with the keys in REGISTERS (thread t owns keys [t*KPT, (t+1)*KPT)): bitonic
// strides below KPT are compare-exchanges inside a thread, strides below 64*KPT are wave shuffles, and only the
// remaining ones (10 of the 78 stages of a 4096-key sort) go through LDS with block barriers.  The LDS round trips of
// the plain network were 54 % of nms_merge (s_memtime stamps: 146 k of 268 k cycles per image).
template <int KPT>
__device__ __forceinline__ void cmpx(u64 (&r)[KPT], int lo, int hi, bool up) {
  const u64 a = r[lo], c = r[hi];
  const bool sw = (a > c) == up;
  r[lo] = sw ? c : a;
  r[hi] = sw ? a : c;
}
template <int KPT>
__device__ void block_sort_regs(u64* keys) {
  constexpr int N = kMergeThreads * KPT;
  const int tid = threadIdx.x;
  u64 r[KPT];
#pragma unroll
  for (int e = 0; e < KPT; ++e) r[e] = keys[tid * KPT + e];
  for (int k = 2; k <= N; k <<= 1) {
    for (int j = k >> 1; j > 0; j >>= 1) {
      if (j < KPT) {                                   // partner key lives in this thread
#pragma unroll
        for (int jj = KPT >> 1; jj > 0; jj >>= 1) {
          if (jj != j) continue;
#pragma unroll
          for (int e = 0; e < KPT; ++e)
            if ((e & jj) == 0) cmpx<KPT>(r, e, e | jj, ((tid * KPT + e) & k) == 0);
        }
      } else if (j < 64 * KPT) {                       // partner key lives in another lane of this wave
        const int lx = j / KPT;
#pragma unroll
        for (int e = 0; e < KPT; ++e) {
          const int i = tid * KPT + e;
          const u64 o = __shfl_xor(r[e], lx);
          const bool take_min = ((i & j) == 0) == ((i & k) == 0);
          r[e] = take_min ? (o < r[e] ? o : r[e]) : (o > r[e] ? o : r[e]);
        }
      } else {                                         // partner key lives in another wave: through LDS
#pragma unroll
        for (int e = 0; e < KPT; ++e) keys[tid * KPT + e] = r[e];
        __syncthreads();
        u64 o[KPT];
#pragma unroll
        for (int e = 0; e < KPT; ++e) o[e] = keys[(tid * KPT + e) ^ j];
        __syncthreads();
#pragma unroll
        for (int e = 0; e < KPT; ++e) {
          const int i = tid * KPT + e;
          const bool take_min = ((i & j) == 0) == ((i & k) == 0);
          r[e] = take_min ? (o[e] < r[e] ? o[e] : r[e]) : (o[e] > r[e] ? o[e] : r[e]);
        }
      }
    }
  }
#pragma unroll
  for (int e = 0; e < KPT; ++e) keys[tid * KPT + e] = r[e];
  __syncthreads();
}
// n_pad: power of two, kMergeThreads <= n_pad <= kLdsKeys, keys in LDS
__device__ void block_sort_lds(u64* keys, int n_pad) {
  switch (n_pad / kMergeThreads) {
    case 1: block_sort_regs<1>(keys); break;
    case 2: block_sort_regs<2>(keys); break;
    case 4: block_sort_regs<4>(keys); break;
    default: block_sort_regs<8>(keys); break;
  }
}

struct MergeArgs {
  const float* pred;
  u64* keys;          // [bs][rows] survivors from the filter (sorted in place when they exceed LDS)
  const int* counts;  // [bs]
  float* stage;       // [bs][stage_cap][8] unsorted kept rows (x1,y1,x2,y2,conf,cls_conf,cls,row bits)
  float* out_dets;    // [bs][cap][7]
  int* out_idx;       // [bs][cap]
  int* out_count;     // [bs]
  int rows, no, nc, max_per_class, cap, stage_cap;
  int cc_index;       // index of the class score inside a row: -1 = 5 + class (an io row), >= 0 = fixed (compact records: 4)
  const u64* row_keys;  // compact form: [bs][rows] one key per io row (~0: not a survivor), compacted into `keys` here; else null
  long key_pitch;     // keys per image row of the workspace (power of two >= rows)
  float nms_thres;
};

__global__ __launch_bounds__(kMergeThreads) void nms_merge_kernel(const MergeArgs a) {
  __shared__ __attribute__((aligned(16))) u64 s_keys[kLdsKeys];
  __shared__ int s_seg_start[kMaxClasses];
  __shared__ int s_seg_len[kMaxClasses];
  __shared__ int s_nout, s_nlist, s_next;
  __shared__ int s_list[kMaxClasses];

  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
  u64* const gkeys0 = a.keys + (long)b * a.key_pitch;
  if (a.row_keys) {
    // compact form: gather the survivors' keys (order is free: they are sorted below) - one LDS atomic per wave and pass
    if (tid == 0) s_nout = 0;
    __syncthreads();
    const u64* const rk = a.row_keys + (long)b * a.rows;
    for (int i0 = 0; i0 < a.rows; i0 += kMergeThreads) {
      const int i = i0 + tid;
      const u64 k = i < a.rows ? rk[i] : ~0ull;
      const bool sv = k != ~0ull;
      const u64 mask = __ballot(sv);
      if (mask) {
        const int leader = __builtin_ctzll(mask);
        int base = 0;
        if (lane == leader) base = atomicAdd(&s_nout, __builtin_popcountll(mask));
        base = __builtin_amdgcn_readlane(base, leader);
        if (sv) gkeys0[base + __builtin_popcountll(mask & ((1ull << lane) - 1ull))] = k;
      }
    }
    __syncthreads();                                   // (block-wide: the gathered keys are visible to every thread)
  }
  const int n = a.row_keys ? s_nout : min(a.counts[b], a.rows);
  __syncthreads();                                     // (s_nout is reset below)
  if (n == 0) {
    if (tid == 0) a.out_count[b] = 0;
    return;
  }
  int n_pad = kMergeThreads;                        // the register sort handles 1024 * {1, 2, 4, 8} keys
  while (n_pad < n) n_pad <<= 1;
  u64* gkeys = a.keys + (long)b * a.key_pitch;
  u64* keys;
  if (n_pad <= kLdsKeys) {
    keys = s_keys;
    for (int i = tid; i < n_pad; i += kMergeThreads) keys[i] = i < n ? gkeys[i] : ~0ull;
  } else {
    keys = gkeys;  // the workspace row is sized to a power of two >= rows
    for (int i = n + tid; i < n_pad; i += kMergeThreads) keys[i] = ~0ull;
  }
  for (int c = tid; c < a.nc; c += kMergeThreads) s_seg_len[c] = 0;
  if (tid == 0) s_nout = s_nlist = s_next = 0;
  __syncthreads();
  if (keys == s_keys) block_sort_lds(keys, n_pad);
  else block_bitonic_sort(keys, n_pad, false);

  // class segments of the sorted list
  for (int i = tid; i < n; i += kMergeThreads) {
    const int c = key_class(keys[i]);
    if (i == 0 || key_class(keys[i - 1]) != c) s_seg_start[c] = i;
    if (i == n - 1 || key_class(keys[i + 1]) != c) s_seg_len[c] = i + 1;  // end for now
  }
  __syncthreads();
  for (int c = tid; c < a.nc; c += kMergeThreads)
    if (s_seg_len[c] > 0) s_list[atomicAdd(&s_nlist, 1)] = c;
  __syncthreads();

  float* stage = a.stage + (long)b * a.stage_cap * 8;
  const float* pred = a.pred + (long)b * a.rows * a.no;

  // classes are handed to the 16 waves through a work queue: survivors usually crowd into a few classes, and a static
  // class -> wave map left most waves idle while one or two walked several long segments
  for (;;) {
    int qi = 0;
    if (lane == 0) qi = atomicAdd(&s_next, 1);
    qi = __builtin_amdgcn_readfirstlane(qi);
    if (qi >= s_nlist) break;
    const int c = s_list[qi];
    const int end = s_seg_len[c];
    const int start = s_seg_start[c];
    const int seg_n = end - start;
    const int m = min(seg_n, a.max_per_class);                          // utils.py:247-250
    // gather my (up to two) candidates: xywh -> xyxy (utils.py:57-60).  Everything the merge loop needs stays in
    // registers: lane q (and q + 64) owns candidate q's corners, area, score and score*corner products; the pivot's
    // values are pulled with v_readlane, so the loop body has no LDS or memory round trip.
    int my_row[2] = {0, 0};
    float my_cconf[2] = {0.f, 0.f};
    float bx[2][4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}}, area[2] = {0.f, 0.f};
    float ws[2] = {0.f, 0.f}, wp[2][4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int q = lane + 64 * h;
      if (q < m) {
        const u64 k = keys[start + q];
        const int r = key_row(k);
        const float* pr = pred + (long)r * a.no;
        const float x = pr[0], y = pr[1], w = pr[2], hh = pr[3];
        bx[h][0] = x - w / 2.f;
        bx[h][1] = y - hh / 2.f;
        bx[h][2] = x + w / 2.f;
        bx[h][3] = y + hh / 2.f;
        area[h] = (bx[h][2] - bx[h][0]) * (bx[h][3] - bx[h][1]);          // area2 of bbox_iou (:92)
        ws[h] = key_conf(k);
#pragma unroll
        for (int e = 0; e < 4; ++e) wp[h][e] = ws[h] * bx[h][e];
        my_row[h] = r;
        my_cconf[h] = pr[a.cc_index >= 0 ? a.cc_index : 5 + c];
      }
    }

    u64 alive0 = (m >= 64) ? ~0ull : ((1ull << m) - 1ull);
    u64 alive1 = (m > 64) ? ((m - 64 >= 64) ? ~0ull : ((1ull << (m - 64)) - 1ull)) : 0ull;
    while (alive0 | alive1) {
      const int p = alive0 ? __builtin_ctzll(alive0) : 64 + __builtin_ctzll(alive1);
      const int pl = p & 63;
      const bool phi = p >= 64;
      const bool last = (__builtin_popcountll(alive0) + __builtin_popcountll(alive1)) == 1;
      const float px1 = readlane_f(phi ? bx[1][0] : bx[0][0], pl), py1 = readlane_f(phi ? bx[1][1] : bx[0][1], pl);
      const float px2 = readlane_f(phi ? bx[1][2] : bx[0][2], pl), py2 = readlane_f(phi ? bx[1][3] : bx[0][3], pl);
      const float pconf = readlane_f(phi ? ws[1] : ws[0], pl);
      float mx1 = px1, my1 = py1, mx2 = px2, my2 = py2;
      u64 h0 = 0, h1 = 0;
      if (last) {                                                        // :268-270 kept as is
        if (!phi) h0 = 1ull << pl; else h1 = 1ull << pl;
      } else {
        const float area1 = (px2 - px1) * (py2 - py1) + 1e-16f;          // :91,:93
        bool hit[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const bool live = ((h ? alive1 : alive0) >> lane) & 1ull;
          hit[h] = false;
          if (live) {                                                    // bbox_iou, utils.py:85-96
            const float iw = fminf(px2, bx[h][2]) - fmaxf(px1, bx[h][0]);
            const float ih = fminf(py2, bx[h][3]) - fmaxf(py1, bx[h][1]);
            const float inter = fmaxf(iw, 0.f) * fmaxf(ih, 0.f);
            const float uni = (area1 + area[h]) - inter;
            hit[h] = __fdiv_rn(inter, uni) > a.nms_thres;                // :271
          }
        }
        h0 = __ballot(hit[0]);
        h1 = __ballot(hit[1]);
        // weighted mixture box (:272-274), sequential fp32 in candidate order (products rounded once, as the
        // reference's `weights * boxes` does before the sum)
        float wsum = 0.f, ax1 = 0.f, ay1 = 0.f, ax2 = 0.f, ay2 = 0.f;
        u64 w0 = h0, w1 = h1;
        while (w0) {
          const int g = __builtin_ctzll(w0);
          w0 &= w0 - 1;
          wsum = wsum + readlane_f(ws[0], g);
          ax1 = ax1 + readlane_f(wp[0][0], g);
          ay1 = ay1 + readlane_f(wp[0][1], g);
          ax2 = ax2 + readlane_f(wp[0][2], g);
          ay2 = ay2 + readlane_f(wp[0][3], g);
        }
        while (w1) {
          const int g = __builtin_ctzll(w1);
          w1 &= w1 - 1;
          wsum = wsum + readlane_f(ws[1], g);
          ax1 = ax1 + readlane_f(wp[1][0], g);
          ay1 = ay1 + readlane_f(wp[1][1], g);
          ax2 = ax2 + readlane_f(wp[1][2], g);
          ay2 = ay2 + readlane_f(wp[1][3], g);
        }
        mx1 = __fdiv_rn(ax1, wsum);
        my1 = __fdiv_rn(ay1, wsum);
        mx2 = __fdiv_rn(ax2, wsum);
        my2 = __fdiv_rn(ay2, wsum);
        if ((h0 | h1) == 0) {  // pivot does not overlap itself (the reference would spin forever): drop it
          if (!phi) h0 = 1ull << pl; else h1 = 1ull << pl;
        }
      }
      // the lane that owns the pivot emits the row
      if (lane == pl) {
        const int slot = atomicAdd(&s_nout, 1);
        if (slot < a.stage_cap) {
          float* o = stage + (long)slot * 8;
          o[0] = mx1; o[1] = my1; o[2] = mx2; o[3] = my2;
          o[4] = pconf;
          o[5] = phi ? my_cconf[1] : my_cconf[0];
          o[6] = (float)c;
          o[7] = __int_as_float(phi ? my_row[1] : my_row[0]);
        }
      }
      alive0 &= ~h0;
      alive1 &= ~h1;
    }
    __builtin_amdgcn_wave_barrier();
  }
  __syncthreads();

  // ---- final order: conf descending, ties by class then pivot order (utils.py:289-291) -------------
  const int n_out = min(s_nout, a.stage_cap);
  __threadfence_block();
  int f_pad = 1;
  while (f_pad < n_out) f_pad <<= 1;
  const bool small_final = f_pad <= 512;            // few kept rows: the plain LDS network (mostly wave-local) is cheaper
  if (!small_final && f_pad < kMergeThreads) f_pad = kMergeThreads;
  for (int i = tid; i < f_pad; i += kMergeThreads) {
    u64 k = ~0ull;
    if (i < n_out) {
      const float* o = stage + (long)i * 8;
      const uint32_t dconf = ~f32_sortable(o[4]);
      // conf:32 | class:12 | slot:13 (slot order within a class follows pivot order because each
      // class is emitted by one wave in sequence)  -> compare on conf, class first; slot breaks ties
      k = ((u64)dconf << 32) | ((u64)(uint32_t)o[6] << 20) | (u64)i;
    }
    s_keys[i] = k;
  }
  __syncthreads();
  if (small_final) block_bitonic_sort(s_keys, f_pad, true);
  else block_sort_lds(s_keys, f_pad);
  for (int i = tid; i < n_out && i < a.cap; i += kMergeThreads) {
    const int slot = (int)(s_keys[i] & 0xfffffu);
    const float* o = stage + (long)slot * 8;
    float* d = a.out_dets + ((long)b * a.cap + i) * 7;
#pragma unroll
    for (int e = 0; e < 7; ++e) d[e] = o[e];
    a.out_idx[(long)b * a.cap + i] = __float_as_int(o[7]);
  }
  if (tid == 0) a.out_count[b] = s_nout;
}

// scale_coords (reference utils/utils.py:296-303), batched: boxes of image b are mapped from the network input
// frame back to the original image frame: (x - pad_x) / gain, (y - pad_y) / gain, clamped at 0; optional
// round-half-even like the caller at utils.py:313.  params[b] = {pad_x, pad_y, gain, n_rows}.
__global__ __launch_bounds__(256) void scale_coords_kernel(float* __restrict__ dets, int row_floats, int cap,
                                                           const float4* __restrict__ params, int do_round, long total) {
  const long t = (long)blockIdx.x * 256 + threadIdx.x;
  if (t >= total) return;
  const int c = (int)(t & 3);
  const long r = t >> 2;
  const int b = (int)(r / cap), i = (int)(r - (long)b * cap);
  const float4 p = params[b];
  if (i >= (int)p.w) return;
  float* v = dets + ((long)b * cap + i) * row_floats + c;
  float x = __fdiv_rn(*v - ((c & 1) ? p.y : p.x), p.z);
  x = fmaxf(x, 0.f);
  if (do_round) x = rintf(x);
  *v = x;
}

// kept rows of all images back to back: block b copies image b's rows behind the rows of the images before it
__global__ __launch_bounds__(256) void pack_detections_kernel(const float* __restrict__ dets, const int* __restrict__ idx,
                                                              const int* __restrict__ count, int cap, float* __restrict__ packed,
                                                              long long* __restrict__ packed_idx) {
  const int b = blockIdx.x;
  int off = 0;
  for (int i = 0; i < b; ++i) off += min(count[i], cap);
  const int n = min(count[b], cap);
  const float* src = dets + (long)b * cap * 7;
  float* dst = packed + (long)off * 7;
  for (int i = threadIdx.x; i < n * 7; i += 256) dst[i] = src[i];
  if (packed_idx)
    for (int i = threadIdx.x; i < n; i += 256) packed_idx[off + i] = idx[(long)b * cap + i];
}

}  // namespace

extern "C" size_t yolo_nms_workspace_bytes(int bs, int rows, int nc) {
  if (bs <= 0 || rows <= 0 || nc <= 0) return 0;
  return carve(nullptr, bs, rows, nc, false).bytes;
}
extern "C" size_t yolo_nms_compact_workspace_bytes(int bs, int rows, int nc) {
  if (bs <= 0 || rows <= 0 || nc <= 0) return 0;
  return carve(nullptr, bs, rows, nc, true).bytes;
}

static int nms_check(int bs, int rows, int nc, float nms_thres, int max_per_class, int cap) {
  YOLO_REQUIRE(bs > 0 && rows > 0 && nc > 0 && cap > 0, "nms: bad sizes");
  YOLO_REQUIRE(nc <= kMaxClasses, "nms: n_class %d > %d unsupported", nc, kMaxClasses);
  YOLO_REQUIRE(rows < (1 << 20), "nms: rows %d >= 2^20 unsupported", rows);
  YOLO_REQUIRE(max_per_class >= 1 && max_per_class <= kMaxPerClassCap, "nms: max_per_class %d not in [1,%d]", max_per_class,
               kMaxPerClassCap);
  YOLO_REQUIRE(nms_thres < 1.f, "nms: nms_thres must be < 1 (the reference never terminates otherwise)");
  return 0;
}

static int launch_merge(const float* rows_base, int row_floats, int cc_index, const Workspace& w, int bs, int rows, int nc,
                        float nms_thres, int max_per_class, float* out_dets, int32_t* out_idx, int32_t* out_count, int cap,
                        hipStream_t st) {
  MergeArgs a;
  a.pred = rows_base;
  a.keys = w.keys;
  a.counts = w.counts;
  a.stage = w.stage;
  a.out_dets = out_dets;
  a.out_idx = out_idx;
  a.out_count = out_count;
  a.rows = rows;
  a.key_pitch = w.key_pitch;
  a.no = row_floats;
  a.nc = nc;
  a.max_per_class = max_per_class;
  a.cap = cap;
  a.stage_cap = w.stage_cap;
  a.cc_index = cc_index;
  a.row_keys = w.row_keys;
  a.nms_thres = nms_thres;
  hipLaunchKernelGGL(nms_merge_kernel, dim3((unsigned)bs), dim3(kMergeThreads), 0, st, a);
  return yolo_check_launch("yolo_nms_merge(merge)");
}

extern "C" int yolo_nms_merge(float* pred, int bs, int rows, int nc, float conf_thres, float nms_thres, float min_wh,
                              int max_per_class, int mutate_conf, float* out_dets, int32_t* out_idx, int32_t* out_count,
                              int cap, void* workspace, size_t workspace_bytes, yolo_stream_t s) {
  YOLO_REQUIRE(pred && out_dets && out_idx && out_count && workspace, "nms: null pointer");
  if (int rc = nms_check(bs, rows, nc, nms_thres, max_per_class, cap)) return rc;
  const int no = nc + 5;
  const size_t tile_bytes = (size_t)kFilterRows * no * 4;
  YOLO_REQUIRE(tile_bytes <= 64 * 1024, "nms: row of %d floats too wide", no);
  if (workspace_bytes < yolo_nms_workspace_bytes(bs, rows, nc))
    return yolo_set_error(YOLO_E_WORKSPACE, "nms: workspace %zu < %zu bytes", workspace_bytes,
                          yolo_nms_workspace_bytes(bs, rows, nc));
  hipStream_t st = (hipStream_t)s;
  const Workspace w = carve(workspace, bs, rows, nc, false);
  hipError_t e = hipMemsetAsync(w.counts, 0, align256((size_t)bs * 4), st);
  if (e != hipSuccess) return yolo_set_error((int)e, "nms: memset: %s", hipGetErrorString(e));
  dim3 fgrid((unsigned)((rows + kFilterRows - 1) / kFilterRows), (unsigned)bs);
  hipLaunchKernelGGL(nms_filter_kernel, fgrid, dim3(256), tile_bytes, st, pred, rows, no, conf_thres, min_wh, mutate_conf,
                     w.keys, w.key_pitch, w.counts);
  if (int rc = yolo_check_launch("yolo_nms_merge(filter)")) return rc;
  return launch_merge(pred, no, -1, w, bs, rows, nc, nms_thres, max_per_class, out_dets, out_idx, out_count, cap, st);
}

// ---- the compact form (include/yolo_hip.h): the head convs' epilogues filter their own rows (yolo_head_decode_filter_fwd appends keys
// and writes the survivors' records), so io is never written and never read back: head launches -> merge
extern "C" int yolo_nms_merge_compact(void* workspace, size_t workspace_bytes, int bs, int rows, int nc, float nms_thres,
                                      int max_per_class, float* out_dets, int32_t* out_idx, int32_t* out_count, int cap,
                                      yolo_stream_t s) {
  YOLO_REQUIRE(workspace && out_dets && out_idx && out_count, "nms_merge_compact: null pointer");
  if (int rc = nms_check(bs, rows, nc, nms_thres, max_per_class, cap)) return rc;
  if (workspace_bytes < yolo_nms_compact_workspace_bytes(bs, rows, nc))
    return yolo_set_error(YOLO_E_WORKSPACE, "nms_compact: workspace %zu < %zu bytes", workspace_bytes,
                          yolo_nms_compact_workspace_bytes(bs, rows, nc));
  const Workspace w = carve(workspace, bs, rows, nc, true);
  return launch_merge(w.rec, kRecFloats, 4, w, bs, rows, nc, nms_thres, max_per_class, out_dets, out_idx, out_count, cap, (hipStream_t)s);
}


extern "C" int yolo_scale_coords(float* dets, int bs, int cap, int row_floats, const float* params_dev, int do_round,
                                 yolo_stream_t s) {
  YOLO_REQUIRE(dets && params_dev && bs > 0 && cap > 0 && row_floats >= 4, "scale_coords: bad arguments");
  const long total = (long)bs * cap * 4;
  hipLaunchKernelGGL(scale_coords_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)s, dets,
                     row_floats, cap, (const float4*)params_dev, do_round, total);
  return yolo_check_launch("yolo_scale_coords");
}

extern "C" int yolo_pack_detections(const float* dets, const int32_t* idx, const int32_t* count, int bs, int cap, float* packed,
                                    int64_t* packed_idx, yolo_stream_t s) {
  YOLO_REQUIRE(dets && count && packed && bs > 0 && cap > 0 && (idx || !packed_idx), "pack_detections: bad arguments");
  hipLaunchKernelGGL(pack_detections_kernel, dim3((unsigned)bs), dim3(256), 0, (hipStream_t)s, dets, idx, count, cap, packed,
                     (long long*)packed_idx);
  return yolo_check_launch("yolo_pack_detections");
}
