// Element kernels the EfficientNet-B0 encoder of YOLOv3TinyEfficient needs beyond the Darknet / MobileNetV2 set
// (reference models/yolov3_tiny_efficient.py:13-72; the blocks themselves live in efficientnet_pytorch 0.2.0's MBConvBlock):
//   * depthwise k x k convolution (k = 3 or 5) with an explicit leading pad: TensorFlow "same" padding puts the odd pad row /
//     column at the bottom / right (stride 2 on an even map: pad 0 + 1 for k = 3, 1 + 2 for k = 5), everything beyond the
//     image reads zero;
//   * squeeze-and-excitation: global average pool, the two 1x1 convs on the pooled vector (swish between them), sigmoid,
//     channel-wise rescale of the block's tensor.
// All are HBM-bound 16-byte-per-lane streams over NHWC bf16 like pointwise.hip; arithmetic in fp32.
#include "common.h"

namespace {

__device__ __forceinline__ float act_f(float v, int act) {
  if (act == YOLO_ACT_SWISH) return v / (1.f + expf(-v));          // x * sigmoid(x)
  if (act == YOLO_ACT_LEAKY01) return fmaxf(v, 0.1f * v);
  if (act == YOLO_ACT_RELU6) return fminf(fmaxf(v, 0.f), 6.f);
  if (act == YOLO_ACT_RELU) return fmaxf(v, 0.f);
  return v;
}

inline unsigned blocks_for(long total) { return (unsigned)((total + 255) / 256); }

// one thread = 8 channels of one output pixel; w: f32 [k*k][c] tap-major, bias f32 [c]
__global__ __launch_bounds__(256) void dwconv_kernel(const bf16_t* __restrict__ x, const float* __restrict__ wt,
                                                     const float* __restrict__ bias, bf16_t* __restrict__ y, int h, int w, int c,
                                                     int in_ct, int in_co, int ho, int wo, int out_ct, int out_co, int k, int stride,
                                                     int pad, int act, long total) {
  const long t = (long)blockIdx.x * 256 + threadIdx.x;
  if (t >= total) return;
  const int cg = c >> 3;
  const int g = (int)(t % cg);
  long p = t / cg;
  const int ow = (int)(p % wo);
  p /= wo;
  const int oh = (int)(p % ho);
  const long b = p / ho;
  float acc[8];
  {
    const f32x4 b0 = *reinterpret_cast<const f32x4*>(bias + g * 8), b1 = *reinterpret_cast<const f32x4*>(bias + g * 8 + 4);
#pragma unroll
    for (int e = 0; e < 4; ++e) acc[e] = b0[e], acc[4 + e] = b1[e];
  }
  for (int i = 0; i < k; ++i) {
    const int hi = oh * stride - pad + i;
    if ((unsigned)hi >= (unsigned)h) continue;
    for (int j = 0; j < k; ++j) {
      const int wi = ow * stride - pad + j;
      if ((unsigned)wi >= (unsigned)w) continue;
      const bf16x8 v = *reinterpret_cast<const bf16x8*>(x + ((b * h + hi) * w + wi) * in_ct + in_co + g * 8);
      const float* wp = wt + (long)(i * k + j) * c + g * 8;
      const f32x4 w0 = *reinterpret_cast<const f32x4*>(wp), w1 = *reinterpret_cast<const f32x4*>(wp + 4);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        acc[e] = fmaf((float)v[e], w0[e], acc[e]);
        acc[4 + e] = fmaf((float)v[4 + e], w1[e], acc[4 + e]);
      }
    }
  }
  bf16x8 o;
#pragma unroll
  for (int e = 0; e < 8; ++e) o[e] = (bf16_t)act_f(acc[e], act);
  *reinterpret_cast<bf16x8*>(y + ((b * ho + oh) * wo + ow) * out_ct + out_co + g * 8) = o;
}

// ---- squeeze-and-excitation ---------------------------------------------------------------------------------------------
constexpr int kSeMaxSplits = 32;   // pixel ranges a map is cut into for the pooling pass (partial sums, added in a fixed order)

// (1) per (image, group of <= 32 8-channel chunks, pixel range): partial sums over the range.  Block = 256 threads:
//     thread = (pixel stripe t / cgb, chunk t % cgb); the stripes meet in LDS in a fixed order.
__global__ __launch_bounds__(256) void se_partial_kernel(const bf16_t* __restrict__ x, float* __restrict__ partial, int hw, int c,
                                                         int in_ct, int in_co, int cgb, int splits) {
  __shared__ float part[256][8];
  const int cg = c >> 3;
  const int groups = (cg + cgb - 1) / cgb;
  int bid = blockIdx.x;
  const int sp = bid % splits;
  bid /= splits;
  const int b = bid / groups, grp = bid % groups;
  const int lc = threadIdx.x % cgb, stripe = threadIdx.x / cgb, nstripes = 256 / cgb;
  const int g = grp * cgb + lc;
  const int per = (hw + splits - 1) / splits, p_lo = sp * per, p_hi = min(hw, p_lo + per);
  float s[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (g < cg)
    for (int p = p_lo + stripe; p < p_hi; p += nstripes) {
      const bf16x8 v = *reinterpret_cast<const bf16x8*>(x + ((long)b * hw + p) * in_ct + in_co + g * 8);
#pragma unroll
      for (int e = 0; e < 8; ++e) s[e] += (float)v[e];
    }
#pragma unroll
  for (int e = 0; e < 8; ++e) part[threadIdx.x][e] = s[e];
  __syncthreads();
  if (stripe == 0 && g < cg) {
    float tot[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int st = 0; st < nstripes; ++st)            // fixed order: deterministic
#pragma unroll
      for (int e = 0; e < 8; ++e) tot[e] += part[st * cgb + lc][e];
#pragma unroll
    for (int e = 0; e < 8; ++e) partial[((long)b * splits + sp) * c + g * 8 + e] = tot[e];
  }
}

// (2) per image: mean = sum of the partials / hw; hidden = swish(W1 mean + b1) (sq values); scale = sigmoid(W2 hidden + b2).
//     W1: f32 [sq][c], W2T: f32 [sq][c] (the expand weight transposed: lanes read consecutive channels); sq <= 64.
__global__ __launch_bounds__(1024) void se_fc_kernel(const float* __restrict__ partial, float* __restrict__ mean, const float* __restrict__ w1,
                                                     const float* __restrict__ b1, const float* __restrict__ w2t, const float* __restrict__ b2,
                                                     float* __restrict__ scale, int c, int sq, int splits, float inv_hw) {
  __shared__ float hid[64];
  const int b = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float* const m = mean + (long)b * c;
  for (int i = threadIdx.x; i < c; i += 1024) {
    float t = 0.f;
    for (int sp = 0; sp < splits; ++sp) t += partial[((long)b * splits + sp) * c + i];
    m[i] = t * inv_hw;
  }
  __syncthreads();
  for (int j = wave; j < sq; j += 16) {              // one wave per hidden unit: lanes stride the channels
    float s = 0.f;
    for (int i = lane; i < c; i += 64) s = fmaf(w1[(long)j * c + i], m[i], s);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
    if (lane == 0) {
      const float v = s + b1[j];
      hid[j] = v / (1.f + expf(-v));
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < c; i += 1024) {
    float s = b2[i];
    for (int j = 0; j < sq; ++j) s = fmaf(w2t[(long)j * c + i], hid[j], s);
    scale[(long)b * c + i] = 1.f / (1.f + expf(-s));
  }
}

// (3) y = x * scale[image][channel]
__global__ __launch_bounds__(256) void se_scale_kernel(const bf16_t* __restrict__ x, const float* __restrict__ scale,
                                                       bf16_t* __restrict__ y, int hw, int c, int in_ct, int in_co, int out_ct,
                                                       int out_co, long total) {
  const long t = (long)blockIdx.x * 256 + threadIdx.x;
  if (t >= total) return;
  const int cg = c >> 3;
  const int g = (int)(t % cg);
  const long p = t / cg, b = p / hw;
  const bf16x8 v = *reinterpret_cast<const bf16x8*>(x + p * in_ct + in_co + g * 8);
  const float* sp = scale + b * c + g * 8;
  const f32x4 s0 = *reinterpret_cast<const f32x4*>(sp), s1 = *reinterpret_cast<const f32x4*>(sp + 4);
  bf16x8 o;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    o[e] = (bf16_t)((float)v[e] * s0[e]);
    o[4 + e] = (bf16_t)((float)v[4 + e] * s1[e]);
  }
  *reinterpret_cast<bf16x8*>(y + p * out_ct + out_co + g * 8) = o;
}

}  // namespace

extern "C" int yolo_dwconv_fwd(const void* x, const float* w, const float* bias, void* y, int n, int h, int w_, int c, int in_c_total,
                               int in_c_offset, int ho, int wo, int out_c_total, int out_c_offset, int ksize, int stride, int pad,
                               int act, yolo_stream_t s) {
  YOLO_REQUIRE(x && w && bias && y && n > 0 && c > 0 && c % 8 == 0, "dwconv: bad arguments");
  YOLO_REQUIRE((ksize == 3 || ksize == 5) && (stride == 1 || stride == 2) && pad >= 0 && pad < ksize, "dwconv: k %d stride %d pad %d", ksize,
               stride, pad);
  // the last output's window must start inside the padded image and need at most k - 1 - pad rows of trailing zeros
  YOLO_REQUIRE(ho >= 1 && wo >= 1 && (ho - 1) * stride - pad < h && (wo - 1) * stride - pad < w_ &&
                   (ho - 1) * stride - pad + ksize <= h + ksize - 1 && (wo - 1) * stride - pad + ksize <= w_ + ksize - 1,
               "dwconv: output size %dx%d inconsistent with input %dx%d k%d s%d pad %d", ho, wo, h, w_, ksize, stride, pad);
  YOLO_REQUIRE(in_c_total % 8 == 0 && in_c_offset % 8 == 0 && out_c_total % 8 == 0 && out_c_offset % 8 == 0 &&
                   in_c_offset + c <= in_c_total && out_c_offset + c <= out_c_total,
               "dwconv: views must be 8-channel aligned");
  const long total = (long)n * ho * wo * (c / 8);
  hipLaunchKernelGGL(dwconv_kernel, dim3(blocks_for(total)), dim3(256), 0, (hipStream_t)s, (const bf16_t*)x, w, bias, (bf16_t*)y, h,
                     w_, c, in_c_total, in_c_offset, ho, wo, out_c_total, out_c_offset, ksize, stride, pad, act, total);
  return yolo_check_launch("yolo_dwconv_fwd");
}

extern "C" size_t yolo_se_workspace_bytes(int n, int c) { return (size_t)n * c * (2 + kSeMaxSplits) * sizeof(float); }

extern "C" int yolo_se_fwd(const void* x, void* y, int n, int h, int w_, int c, int in_c_total, int in_c_offset, int out_c_total,
                           int out_c_offset, const float* w1, const float* b1, const float* w2, const float* b2, int squeeze,
                           void* workspace, size_t ws_bytes, yolo_stream_t s) {
  YOLO_REQUIRE(x && y && w1 && b1 && w2 && b2 && workspace && n > 0 && c > 0 && c % 8 == 0, "se: bad arguments");
  YOLO_REQUIRE(squeeze >= 1 && squeeze <= 64, "se: %d squeezed channels (1..64)", squeeze);
  YOLO_REQUIRE(ws_bytes >= yolo_se_workspace_bytes(n, c), "se: workspace too small");
  YOLO_REQUIRE(in_c_total % 8 == 0 && in_c_offset % 8 == 0 && out_c_total % 8 == 0 && out_c_offset % 8 == 0 &&
                   in_c_offset + c <= in_c_total && out_c_offset + c <= out_c_total,
               "se: views must be 8-channel aligned");
  float* const mean = (float*)workspace;
  float* const scale = mean + (size_t)n * c;
  float* const partial = scale + (size_t)n * c;
  const int hw = h * w_, cg = c / 8;
  const int cgb = cg < 32 ? (cg >= 16 ? 16 : cg >= 8 ? 8 : cg >= 4 ? 4 : cg >= 2 ? 2 : 1) : 32;    // power of two <= 32: 256 % cgb == 0
  const int groups = (cg + cgb - 1) / cgb;
  // enough workgroups for the chip: the 208x208 x 32-channel map of the first block is ONE channel group per image
  int splits = 1;
  while (splits < kSeMaxSplits && (long)n * groups * splits < 512 && hw / (splits * 2) >= 256) splits *= 2;
  hipLaunchKernelGGL(se_partial_kernel, dim3((unsigned)(n * groups * splits)), dim3(256), 0, (hipStream_t)s, (const bf16_t*)x, partial, hw,
                     c, in_c_total, in_c_offset, cgb, splits);
  int rc = yolo_check_launch("yolo_se_fwd(pool)");
  if (rc) return rc;
  hipLaunchKernelGGL(se_fc_kernel, dim3((unsigned)n), dim3(1024), 0, (hipStream_t)s, partial, mean, w1, b1, w2, b2, scale, c, squeeze, splits,
                     1.f / (float)hw);
  rc = yolo_check_launch("yolo_se_fwd(fc)");
  if (rc) return rc;
  const long total = (long)n * hw * cg;
  hipLaunchKernelGGL(se_scale_kernel, dim3(blocks_for(total)), dim3(256), 0, (hipStream_t)s, (const bf16_t*)x, scale, (bf16_t*)y, hw,
                     c, in_c_total, in_c_offset, out_c_total, out_c_offset, total);
  return yolo_check_launch("yolo_se_fwd(scale)");
}
