// The YOLOLayer decode + NMS row filter that a head conv runs as its epilogue (reference models/yolo_layer.py:57-69,90-96 and
// utils/utils.py:212-218), shared by the tiled head kernel (conv_igemm.hip, DECODE instances) and the pipelined weight-stationary
// one (conv_head_stream.hip, round 5).
#pragma once
#include "conv_common.h"
#include "nms_common.h"

namespace yolo_conv {

// The workgroup's tile [pixels][na * (5 + nc) head channels, pitch DP floats] of act(conv + bias) sits in LDS as fp32 (`stg`); the
// calling wave walks ITS pixels m0 + wave * PPW .. + PPW - 1 (flattened n * ho * wo order): per (pixel, anchor) the lanes take
// k = lane, lane + 64 .. of the (5 + nc)-run, so the raw logits go to p and the decoded values to io as contiguous 340-byte runs that
// the next pixel continues.  The head tensor itself is never written.  h.io == nullptr: FILTER mode (the compact NMS form) - the
// decoded rows stay in LDS and the wave runs the row filter of non_max_suppression on them (below).  BN = padded head channels (256).
// Per-lane constants of the decode: head channel c = lane + 64 j -> (anchor, role k), and that (anchor, role)'s anchor size.  Computed
// ONCE per workgroup (the pipelined head kernel calls head_decode_rows per 32-pixel tile: two integer divisions per channel and the
// anchor look-up do not belong into its tile loop).
template <int BN>
struct HeadLanes {
  int ak[BN / 64];        // anchor << 8 | role k, -1 beyond the head's channels
  float anchor[BN / 64];  // anchor_w (k == 2) / anchor_h of the lane's anchor
};
template <int BN>
__device__ __forceinline__ HeadLanes<BN> head_lanes(const HeadDecodeArgs& h, const int lane) {
  HeadLanes<BN> hl;
  // The four anchors as opaque scalars + selects: indexed by the per-lane anchor number (or written as a select chain over h.anchor_w[i],
  // which LLVM folds back into an indexed load) the table goes to scratch memory, i.e. a vector-memory load and an s_waitcnt vmcnt(0)
  // per use - behind the tile DMAs of the pipelined kernel that wait is an HBM round trip.
  float aw[4], ah[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    aw[i] = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, h.anchor_w[i])));
    ah[i] = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, h.anchor_h[i])));
  }
#pragma unroll
  for (int j = 0; j < BN / 64; ++j) {
    const int c = lane + 64 * j;
    const int an = c / h.no, k = c - an * h.no;
    const bool live = c < h.na * h.no;
    hl.ak[j] = live ? (an << 8 | k) : -1;
    const float w_ = an == 0 ? aw[0] : an == 1 ? aw[1] : an == 2 ? aw[2] : aw[3];
    const float h_ = an == 0 ? ah[0] : an == 1 ? ah[1] : an == 2 ? ah[2] : ah[3];
    hl.anchor[j] = live ? (k == 2 ? w_ : h_) : 0.f;
  }
  return hl;
}

template <int PPW, int BN>
__device__ __forceinline__ void head_decode_rows(const ConvArgs& a, const HeadLanes<BN>& hl, float* const stg, const int DP, const int m0, const int wave,
                                                 const int lane) {
  const HeadDecodeArgs& h = a.hd;
  const YoloConvDesc& d = a.d;
  const int hw_out = d.ho * d.wo;
  constexpr int CJ = BN / 64;
  int c_k[CJ];
  long c_poff[CJ], c_ioff[CJ];                      // element offsets of (anchor, k) inside one image's p / io block
  float c_anchor[CJ];
#pragma unroll
  for (int j = 0; j < CJ; ++j) {
    const int an = hl.ak[j] >> 8, k = hl.ak[j] & 255;
    c_k[j] = hl.ak[j] < 0 ? -1 : k;
    c_poff[j] = (long)an * hw_out * h.no + k;
    c_ioff[j] = ((long)h.io_row_offset + (long)an * hw_out) * h.no + k;
    c_anchor[j] = hl.anchor[j];
  }
  int m = m0 + wave * PPW;
  if (m < a.M) {
    int b = m / hw_out, cell = m - b * hw_out;
    int gy = cell / d.wo, gx = cell - gy * d.wo;
    for (int i = 0; i < PPW && m < a.M; ++i, ++m) {
      const float* const srow = stg + (wave * PPW + i) * DP;
      float* const pimg = h.p ? h.p + ((long)b * h.na * hw_out + cell) * h.no : nullptr;
      float* const iimg = h.io ? h.io + ((long)b * h.io_rows_total + cell) * h.no : nullptr;
#pragma unroll
      for (int j = 0; j < CJ; ++j) {
        if (c_k[j] < 0) continue;
        const float r = srow[lane + 64 * j];
        const float v = yolo_decode_elem<false>(r, c_k[j], gx, gy, c_anchor[j], h.stride, h.no - 5);
        if (pimg) pimg[c_poff[j]] = r;
        if (iimg) iimg[c_ioff[j]] = v;
        else const_cast<float*>(srow)[lane + 64 * j] = v;   // filter mode: the decoded row stays in LDS for the scan below
      }
      ++cell;
      if (++gx == d.wo) {
        gx = 0;
        if (++gy == d.ho) {
          gy = 0;
          cell = 0;
          ++b;
        }
      }
    }
  }
  if (!h.io) {
    // ---- filter mode: the row filter of non_max_suppression (reference utils/utils.py:212-218; csrc/nms.hip nms_filter_kernel, whose
    // arithmetic this repeats operation for operation on the SAME decoded values) over the wave's own PPW x na rows, straight from
    // LDS.  Two lanes per row scan half of the classes each in order and are merged in order: first maximum wins, the first NaN
    // poisons (torch.max semantics).  Every row leaves its key (class | ~conf | io row; ~0 if it does not survive) at its place in
    // `row_keys`, a survivor also its record (x, y, w, h, class score) in `rec`.  io itself is never stored.
    using namespace yolo_nms;
    const uint32_t n_slots = (uint32_t)d.n * (uint32_t)h.io_rows_total;
    const __amdgpu_buffer_rsrc_t rkeys = __builtin_amdgcn_make_buffer_rsrc((void*)h.row_keys, 0, n_slots * 8u, 0x00020000);
    const __amdgpu_buffer_rsrc_t rrec = __builtin_amdgcn_make_buffer_rsrc((void*)h.rec, 0, n_slots * (uint32_t)(kRecFloats * 4), 0x00020000);
    __builtin_amdgcn_wave_barrier();
    wait_lds();                                       // this wave's decoded rows are in LDS (nobody else touches them)
    const int nc = h.no - 5, half = (nc + 1) >> 1;
    const int m_w0 = m0 + wave * PPW, rows_w = PPW * h.na;
    for (int base = 0; base < rows_w; base += 32) {
      const int rl = base + (lane >> 1), seg = lane & 1;
      const int pi = rl / h.na, an = rl - pi * h.na;
      const int mm = m_w0 + pi;
      const bool live = rl < rows_w && mm < a.M;
      const float* const row = stg + (wave * PPW + (live ? pi : 0)) * DP + (live ? an : 0) * h.no;
      const int k0 = seg * half, k1 = min(nc, k0 + half);
      // A row with a non-finite class score is dropped whatever its maximum is (utils.py:218), so the scan needs the NaN rules of
      // torch.max only where they cannot matter: it keeps the first maximum (strict >) and the largest |bits| of the classes seen -
      // all of them finite <=> that is below the exponent mask.  3 + 2 VALU instructions per class, no branches.
      bool have = k0 < k1;
      float best = have ? row[5 + k0] : 0.f;
      int arg = k0;
      uint32_t amax = have ? (__float_as_uint(best) & 0x7fffffffu) : 0u;
      auto take = [&](float v, int k) {
        amax = max(amax, __float_as_uint(v) & 0x7fffffffu);
        const bool t = v > best;
        best = t ? v : best;
        arg = t ? k : arg;
      };
      int k = k0 + 1;
      for (; k + 8 <= k1; k += 8) {                    // eight LDS reads in flight, then the compares in class order
        float v8[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) v8[e] = row[5 + k + e];
#pragma unroll
        for (int e = 0; e < 8; ++e) take(v8[e], k + e);
      }
      for (; k < k1; ++k) take(row[5 + k], k);
      {                                               // the two halves in class order: the right one wins only with a larger maximum
        const float ob = __shfl_xor(best, 1);
        const int oa = __shfl_xor(arg, 1);
        const bool oh = __shfl_xor((int)have, 1) != 0;
        amax = max(amax, (uint32_t)__shfl_xor((int)amax, 1));
        const bool other_is_right = seg == 0;
        const float lb = other_is_right ? best : ob, rb = other_is_right ? ob : best;
        const int la = other_is_right ? arg : oa, ra = other_is_right ? oa : arg;
        const bool lh = other_is_right ? have : oh, rh = other_is_right ? oh : have;
        const bool take_r = rh & (!lh | (rb > lb));
        best = take_r ? rb : lb;
        arg = take_r ? ra : la;
      }
      bool all_finite = amax < 0x7f800000u;
      bool keep = false;
      float conf = 0.f;
      const int bimg = live ? mm / hw_out : 0;
      const int iorow = h.io_row_offset + an * hw_out + (mm - bimg * hw_out);
      if (live && seg == 0) {
        conf = row[4] * best;                                                        // utils.py:213
        const float bw = row[2], bh = row[3];
        all_finite = all_finite && finite_f(row[0]) && finite_f(row[1]) && finite_f(bw) && finite_f(bh) && finite_f(conf);
        keep = (conf > h.conf_thres) && (bw > h.min_wh) && (bh > h.min_wh) && all_finite;  // :216-218
      }
      // One key per row (no atomics: nms_merge compacts them), the record of a survivor at the row's place.  Buffer stores, issued
      // by every lane of every pass - a lane with nothing to store gets an out-of-range offset, which the hardware drops -, so that a
      // wave issues the SAME number of vector-memory operations for every tile: the pipelined head kernel (conv_head_stream.hip)
      // counts them in its s_waitcnt vmcnt arithmetic.
      {
        const bool st_key = live && seg == 0, st_rec = st_key && keep;
        const uint32_t slot = (uint32_t)bimg * (uint32_t)h.io_rows_total + (uint32_t)iorow;
        const u64 key = keep ? make_key(arg, conf, iorow) : ~0ull;
        __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, key), rkeys, st_key ? slot * 8u : yolo_conv::kOobOffset, 0, 0);
        const uint32_t ro = st_rec ? slot * (uint32_t)(kRecFloats * 4) : yolo_conv::kOobOffset;
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, f32x4{row[0], row[1], row[2], row[3]}), rrec, ro, 0, 0);
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(uint32_t, best), rrec, ro, 16, 0);
      }
    }
  }
}

}  // namespace yolo_conv
