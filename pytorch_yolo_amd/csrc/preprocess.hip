// Pre-processing in front of the hot path: the reference's loader turns a decoded uint8 HWC
// image into the float32 NCHW tensor the model eats with
//   LetterBox (utils/augs.py:7-94: cv2.resize INTER_AREA by one ratio, cv2.copyMakeBorder BORDER_REPLICATE)
//   -> _convert_img_for_net (utils/dataset_csv.py:79-87: float32, /= 255, HWC -> CHW)
//   -> equalize_shapes (utils/dataset_csv.py:146-171: centre on a 0.5 canvas of the batch's max size).
// One kernel does all three per image: a thread owns one canvas pixel, finds its place (canvas fill / replicate
// border / resized image) and evaluates the INTER_AREA sum for it directly from the source image — nothing
// intermediate is written.  The arithmetic follows oracle/preprocess.py operation for operation (tables in double,
// weights rounded to float32, separate float32 multiply and add, x pass inside y pass, round half to even), so the
// device result equals the oracle bit for bit.  HBM-bound: reads the source image once (taps of neighbouring
// outputs overlap in L2), writes 4 B x c per canvas pixel.
#include "common.h"

namespace {

struct LetterboxArgs {
  const uint8_t* src;
  float* dst_f32;     // mode 1: [c, dst_h, dst_w] float32 (/255), canvas filled with `fill` outside the rectangle
  uint8_t* dst_u8;    // mode 0: [th, tw, c] uint8 = LetterBox.apply's image
  int h, w, c, src_pitch;
  int dst_h, dst_w, rh, rw, top, left, th, tw, off_y, off_x;
  double scale;       // 1 / resize_ratio (source pixels per resized pixel)
  float fill;
};

struct Tap {
  int first;          // index of the first source sample
  int n;              // number of samples
  bool head, tail;    // the first / last sample carries a partial weight
  float w_head, w_mid, w_tail;
};

// computeResizeAreaTab (scale >= 1) or the bilinear variant OpenCV uses for INTER_AREA when enlarging (scale < 1)
__device__ __forceinline__ Tap make_tap(int d, int ssize, double scale) {
  Tap t;
  if (scale >= 1.0) {
    const double f1 = d * scale, f2 = f1 + scale;
    const double cell = fmin(scale, (double)ssize - f1);
    int s1 = (int)ceil(f1), s2 = (int)floor(f2);
    s2 = min(s2, ssize - 1);
    s1 = min(s1, s2);
    t.head = s1 - f1 > 1e-3;
    t.tail = f2 - s2 > 1e-3;
    t.first = t.head ? s1 - 1 : s1;
    t.n = (t.head ? 1 : 0) + (s2 - s1) + (t.tail ? 1 : 0);
    t.w_mid = (float)(1.0 / cell);
    t.w_head = (float)((s1 - f1) / cell);
    t.w_tail = (float)(fmin(fmin(f2 - s2, 1.0), cell) / cell);
  } else {
    const double inv = 1.0 / scale;
    int sx = (int)floor(d * scale);
    double fx = (double)(d + 1) - (double)(sx + 1) * inv;
    fx = fx <= 0.0 ? 0.0 : fx - floor(fx);
    if (sx < 0) {
      sx = 0;
      fx = 0.0;
    }
    if (sx >= ssize - 1) {
      sx = ssize - 1;
      fx = 0.0;
    }
    t.first = sx;
    t.n = 2;
    t.head = t.tail = true;
    t.w_head = (float)(1.0 - fx);
    t.w_tail = (float)fx;
    t.w_mid = 0.f;
  }
  return t;
}

__device__ __forceinline__ float tap_weight(const Tap& t, int i) {
  if (t.head && i == 0) return t.w_head;          // (head first: a lone head sample is not the tail)
  if (t.tail && i == t.n - 1) return t.w_tail;
  return t.w_mid;
}

template <int MODE>
__global__ __launch_bounds__(256) void letterbox_kernel(const LetterboxArgs a) {
  const int W = MODE ? a.dst_w : a.tw, H = MODE ? a.dst_h : a.th;
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (long)W * H) return;
  const int y = (int)(idx / W), x = (int)(idx - (long)y * W);
  const long plane = (long)a.dst_h * a.dst_w;
  int ly = y, lx = x;                                   // position inside the letterbox rectangle
  if (MODE) {
    ly -= a.off_y;
    lx -= a.off_x;
    if ((unsigned)ly >= (unsigned)a.th || (unsigned)lx >= (unsigned)a.tw) {
      for (int ch = 0; ch < a.c; ++ch) a.dst_f32[ch * plane + idx] = a.fill;     // equalize_shapes canvas
      return;
    }
  }
  const int ry = min(max(ly - a.top, 0), a.rh - 1), rx = min(max(lx - a.left, 0), a.rw - 1);   // BORDER_REPLICATE
  const Tap ty = make_tap(ry, a.h, a.scale), tx = make_tap(rx, a.w, a.scale);
  const bool up = a.scale < 1.0;
  float tot[4] = {0.f, 0.f, 0.f, 0.f};
  for (int iy = 0; iy < ty.n; ++iy) {
    const int sy = up ? (iy == 0 ? ty.first : min(ty.first + 1, a.h - 1)) : ty.first + iy;
    const uint8_t* const row = a.src + (long)sy * a.src_pitch;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    for (int ix = 0; ix < tx.n; ++ix) {
      const int sx = up ? (ix == 0 ? tx.first : min(tx.first + 1, a.w - 1)) : tx.first + ix;
      const float alpha = tap_weight(tx, ix);
      for (int ch = 0; ch < a.c; ++ch) acc[ch] = __fadd_rn(acc[ch], __fmul_rn((float)row[sx * a.c + ch], alpha));
    }
    const float beta = tap_weight(ty, iy);
    for (int ch = 0; ch < a.c; ++ch) tot[ch] = __fadd_rn(tot[ch], __fmul_rn(acc[ch], beta));
  }
  for (int ch = 0; ch < a.c; ++ch) {
    const float r = fminf(fmaxf(rintf(tot[ch]), 0.f), 255.f);        // saturate_cast<uchar>: round half to even
    if (MODE) a.dst_f32[ch * plane + idx] = __fdiv_rn(r, 255.f);     // dataset_csv.py:82-83
    else a.dst_u8[idx * a.c + ch] = (uint8_t)r;
  }
}

}  // namespace

extern "C" int yolo_letterbox_u8_fwd(const uint8_t* src, int h, int w, int c, int src_pitch, double resize_ratio, int rh, int rw,
                                     int top, int left, int th, int tw, uint8_t* dst_u8, float* dst_f32, int dst_h, int dst_w,
                                     int off_y, int off_x, float fill, yolo_stream_t s) {
  YOLO_REQUIRE(src && (dst_u8 || dst_f32) && !(dst_u8 && dst_f32), "letterbox: exactly one of dst_u8 / dst_f32");
  YOLO_REQUIRE(h > 0 && w > 0 && c >= 1 && c <= 4 && src_pitch >= w * c, "letterbox: bad source (h %d w %d c %d pitch %d)", h, w, c,
               src_pitch);
  YOLO_REQUIRE(resize_ratio > 0.0 && rh > 0 && rw > 0 && top >= 0 && left >= 0 && top + rh <= th && left + rw <= tw,
               "letterbox: resized image %dx%d at (%d,%d) does not fit the %dx%d rectangle", rh, rw, top, left, th, tw);
  LetterboxArgs a;
  a.src = src;
  a.dst_f32 = dst_f32;
  a.dst_u8 = dst_u8;
  a.h = h;
  a.w = w;
  a.c = c;
  a.src_pitch = src_pitch;
  a.dst_h = dst_h;
  a.dst_w = dst_w;
  a.rh = rh;
  a.rw = rw;
  a.top = top;
  a.left = left;
  a.th = th;
  a.tw = tw;
  a.off_y = off_y;
  a.off_x = off_x;
  a.scale = 1.0 / resize_ratio;
  a.fill = fill;
  if (dst_f32) {
    YOLO_REQUIRE(off_y >= 0 && off_x >= 0 && off_y + th <= dst_h && off_x + tw <= dst_w,
                 "letterbox: rectangle %dx%d at (%d,%d) does not fit the %dx%d canvas", th, tw, off_y, off_x, dst_h, dst_w);
    const long total = (long)dst_h * dst_w;
    hipLaunchKernelGGL(letterbox_kernel<1>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)s, a);
  } else {
    const long total = (long)th * tw;
    hipLaunchKernelGGL(letterbox_kernel<0>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)s, a);
  }
  return yolo_check_launch("yolo_letterbox_u8_fwd");
}
