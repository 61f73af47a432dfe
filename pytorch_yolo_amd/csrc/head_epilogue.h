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
// All-reduce over a row of 16 lanes by row-rotate DPP (v_max / v_min with a rotated operand: no LDS crossbar, 4 instructions).
template <int CTRL>
__device__ __forceinline__ int dpp_row(int v) { return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xf, 0xf, false); }
__device__ __forceinline__ float row16_max(float v) {          // (fmaxf: a NaN operand is skipped)
  v = fmaxf(v, __builtin_bit_cast(float, dpp_row<0x128>(__builtin_bit_cast(int, v))));     // row_ror:8
  v = fmaxf(v, __builtin_bit_cast(float, dpp_row<0x124>(__builtin_bit_cast(int, v))));     // row_ror:4
  v = fmaxf(v, __builtin_bit_cast(float, dpp_row<0x122>(__builtin_bit_cast(int, v))));     // row_ror:2
  v = fmaxf(v, __builtin_bit_cast(float, dpp_row<0x121>(__builtin_bit_cast(int, v))));     // row_ror:1
  return v;
}
__device__ __forceinline__ int row16_min_i(int v) {
  v = min(v, dpp_row<0x128>(v));
  v = min(v, dpp_row<0x124>(v));
  v = min(v, dpp_row<0x122>(v));
  v = min(v, dpp_row<0x121>(v));
  return v;
}

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
  if (h.io || h.p) {
    // ---- forward(): every head value is decoded and stored (io), the raw logits too (p).  In the filter form with p only the raw
    // logits are stored here; the rows are then filtered below.
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
          if (pimg) pimg[c_poff[j]] = r;
          if (iimg) iimg[c_ioff[j]] = yolo_decode_elem<false>(r, c_k[j], gx, gy, c_anchor[j], h.stride, h.no - 5);
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
    if (h.io) return;
  }
  // ---- FILTER form (detect(): io is never written): the row filter of non_max_suppression (reference utils/utils.py:212-218;
  // csrc/nms.hip nms_filter_kernel, whose arithmetic this repeats on the SAME decoded values - yolo_decode_elem<false> - so that the
  // compact NMS form stays bit-equal to head + io + nms_filter) over the wave's own PPW x na rows.
  // Round 5: only what can decide a row is decoded.  Rounds 3-4 decoded all na * (5 + nc) values of every pixel (exp + rcp each) and
  // let two lanes per row scan the 80 decoded class scores one after the other: ~25 M vector instructions per 32 SPP-640 images,
  // which - not the conv, not the stores - is what most of the 80x80 head's 0.115 ms were.  Now a row is handled by a GROUP OF 16
  // LANES (four rows per wave and pass, class c on lane c % 16, reductions by row-rotate DPP: no LDS crossbar, no loop over classes):
  //   1. objectness first: conf = s(obj) * best <= s(obj) (best <= 1, one fp32 product), so a row with s(obj) <= conf_thres cannot
  //      survive whatever its classes are - a pass without a candidate row only stores its ~0 keys (a NaN objectness fails the `<=`
  //      and takes the full path, where the finite test drops it as the reference does);
  //   2. the maximum class LOGIT l* of the row (a sigmoid is monotone up to its rounding), then only the classes whose logit lies in
  //      a window below l* are decoded and the reference's "first maximum of the decoded scores" is taken among them.  The window is
  //      wide enough for every class that can tie with or exceed the decoded maximum: two scores differ by more than the decode's
  //      error (2^-21 relative) once the logits are 0.25 apart below l* = 12 (s'(12) = 6e-6 per unit), and above that everything
  //      from logit 11 up is decoded (saturated scores do tie: s(17) = s(30) = 1.0f, and the FIRST of them wins as in torch.max);
  //   3. a NaN class logit anywhere drops the row (utils.py:218 via isfinite; +-inf logits decode to 1 / 0 and are fine), nc = 1
  //      has the reference's constant class score 1 (yolo_layer.py:95-96);
  //   4. box (x, y, w, h) and conf are decoded by the group's first lane, for candidates only.
  // Every row leaves its key (class | ~conf | io row; ~0 if it does not survive) at its place in `row_keys`, a survivor also its record
  // (x, y, w, h, class score) in `rec`.
  using namespace yolo_nms;
  static_assert((PPW & (PPW - 1)) == 0, "rows are numbered anchor * PPW + pixel");
  const int nc = h.no - 5;
  const int grp = lane >> 4, sub = lane & 15;
  const int ni = (nc + 15) >> 4;                                           // class slots per lane (<= 8: 5 + nc <= 128)
  // (opaque scalars + selects, as in head_lanes: an indexed read of the kernel-argument table becomes a scratch load)
  float sw[4], sh[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    sw[i] = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, h.anchor_w[i])));
    sh[i] = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, h.anchor_h[i])));
  }
  for (int r0 = 0; r0 < PPW * h.na; r0 += 4) {
    const int rl = r0 + grp;                                                // this group's row: anchor rl / PPW, pixel rl % PPW
    const int an = rl / PPW, pi = rl & (PPW - 1);
    const int mm = m0 + wave * PPW + pi;
    const bool live = an < h.na && mm < a.M;
    const float* const row = stg + (wave * PPW + pi) * DP + (live ? an : 0) * h.no;
    const float s_obj = yolo_decode_elem<false>(row[4], 4, 0, 0, 0.f, h.stride, nc);
    const bool cand = live && !(s_obj <= h.conf_thres);
    const int bimg = live ? mm / hw_out : 0;
    const int cell = mm - bimg * hw_out;
    const int iorow = h.io_row_offset + an * hw_out + cell;
    const long slot = (long)bimg * h.io_rows_total + iorow;
    if (__ballot(cand) == 0) {                                              // (wave-uniform) nothing in this pass can survive
      if (live && sub == 0) h.row_keys[slot] = ~0ull;
      continue;
    }
    float best = 1.f;                                                       // (nc == 1: the reference's constant class score)
    int arg = 0;
    bool cls_finite = true;
    if (nc > 1) {
      float l[8];
      float lmax = -__builtin_inff();
      bool nan_here = false;
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int c = sub + 16 * i;
        l[i] = (i < ni && c < nc) ? row[5 + c] : -__builtin_inff();
        nan_here |= l[i] != l[i];
        lmax = fmaxf(lmax, l[i]);                                           // (fmaxf skips NaNs: such a row is dropped anyway)
      }
      lmax = row16_max(lmax);
      cls_finite = row16_max(nan_here ? 1.f : 0.f) == 0.f;
      const float window = lmax > 12.f ? 11.f : lmax - 0.25f;
      float dbest = -1.f;
      int cbest = 1 << 20;
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        if (i < ni && l[i] >= window) {                                      // (a lane's classes in ascending order: strict > keeps the first)
          const float dv = yolo_decode_elem<false>(l[i], 5 + sub + 16 * i, 0, 0, 0.f, h.stride, nc);
          if (dv > dbest) dbest = dv, cbest = sub + 16 * i;
        }
      }
      best = row16_max(dbest);
      arg = row16_min_i(dbest == best ? cbest : (1 << 20));
    }
    if (sub == 0 && live) {
      bool keep = false;
      float conf = 0.f;
      f32x4 box = {0.f, 0.f, 0.f, 0.f};
      if (cand) {
        const int gy = cell / d.wo, gx = cell - gy * d.wo;
        const float aw = an == 0 ? sw[0] : an == 1 ? sw[1] : an == 2 ? sw[2] : sw[3];
        const float ah = an == 0 ? sh[0] : an == 1 ? sh[1] : an == 2 ? sh[2] : sh[3];
        box[0] = yolo_decode_elem<false>(row[0], 0, gx, gy, 0.f, h.stride, nc);
        box[1] = yolo_decode_elem<false>(row[1], 1, gx, gy, 0.f, h.stride, nc);
        box[2] = yolo_decode_elem<false>(row[2], 2, gx, gy, aw, h.stride, nc);
        box[3] = yolo_decode_elem<false>(row[3], 3, gx, gy, ah, h.stride, nc);
        conf = s_obj * best;                                                                  // utils.py:213
        const bool all_finite = cls_finite && finite_f(best) && finite_f(box[0]) && finite_f(box[1]) && finite_f(box[2]) && finite_f(box[3]) && finite_f(conf);
        keep = (conf > h.conf_thres) && (box[2] > h.min_wh) && (box[3] > h.min_wh) && all_finite;   // :216-218
      }
      h.row_keys[slot] = keep ? make_key(arg, conf, iorow) : ~0ull;        // one key per row (no atomics): nms_merge compacts them
      if (keep) {
        float* const rp = h.rec + slot * kRecFloats;
        *reinterpret_cast<f32x4*>(rp) = box;
        rp[4] = best;
      }
    }
  }
}

}  // namespace yolo_conv
