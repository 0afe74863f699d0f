// Once-per-document conditioning path (step-invariant): the VGG-style conv pyramid of
// idf/cross_model.py:18-95 as im2col + exact-fp32 MFMA GEMM (bias + ReLU in the GEMM epilogue),
// 2x2 max-pool, and the bilinear (align_corners=True) resize of the last level to the coordinate
// grid (idf/cross_model.py:590-593).  Activations are channels-last [H, W, C] f32, which is both
// the GEMM's natural output and the layout the per-step feature warp gathers from.
#include "common.h"
#include "mfma.h"

namespace dvd {

// rows of the 3x3 / pad 1 / stride 1 convolution as a GEMM operand:
//   out[(y*W + x)*ldo + (ky*3 + kx)*C + c] = in[c*sc + (y+ky-1)*sy + (x+kx-1)*sx]   (0 outside)
// columns [9*C, ldo) are zeroed (K padding of the GEMM).
__global__ void __launch_bounds__(256) im2col3x3_kernel(const float* __restrict__ in, long sc, long sy, long sx,
                                                        float* __restrict__ out, int ldo, int c, int h, int w,
                                                        long total) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  const long pix = idx / ldo;
  const int col = (int)(idx % ldo);
  float v = 0.f;
  if (col < 9 * c) {
    const int tap = col / c, ch = col % c;
    const int y = (int)(pix / w) + tap / 3 - 1, x = (int)(pix % w) + tap % 3 - 1;
    if (y >= 0 && y < h && x >= 0 && x < w) v = in[ch * sc + y * sy + x * sx];
  }
  out[idx] = v;
}

// the same matrix from a CHANNELS-LAST source [H, W, C], C % 4 == 0: one thread per (pixel, 4 consecutive columns) - the four
// columns are four channels of one tap, 16 contiguous bytes on both sides (the element-wise kernel above ran at 1.5 TB/s of
// stores; this form is bound by HBM)
__global__ void __launch_bounds__(256) im2col3x3_nhwc4_kernel(const float* __restrict__ in, float* __restrict__ out,
                                                              int ldo, int c, int h, int w, long total4) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total4) return;
  const int l4 = ldo >> 2;
  const long pix = idx / l4;
  const int col = (int)(idx - pix * l4) << 2;
  floatx4 v = {0.f, 0.f, 0.f, 0.f};
  if (col < 9 * c) {
    const int tap = col / c, ch = col - tap * c;
    const int py = (int)(pix / w), px = (int)(pix - (long)py * w);
    const int y = py + tap / 3 - 1, x = px + tap % 3 - 1;
    if (y >= 0 && y < h && x >= 0 && x < w) v = *(const floatx4*)(in + ((size_t)y * w + x) * c + ch);
  }
  *(floatx4*)(out + pix * ldo + col) = v;
}

// channels-last 2x2 / stride 2 max-pool: in [H, W, C] -> out [H/2, W/2, C]
__global__ void __launch_bounds__(256) maxpool2_nhwc_kernel(const float* __restrict__ in, float* __restrict__ out,
                                                            int c, int h, int w, long total4) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total4) return;
  const int c4 = c >> 2;
  const int ch = (int)(idx % c4) * 4;
  const long pix = idx / c4;
  const int wo = w >> 1;
  const int yo = (int)(pix / wo), xo = (int)(pix % wo);
  const float* p = in + (((size_t)2 * yo) * w + 2 * xo) * c + ch;
  const floatx4 a = *(const floatx4*)p, b = *(const floatx4*)(p + c);
  const floatx4 d = *(const floatx4*)(p + (size_t)w * c), e = *(const floatx4*)(p + (size_t)w * c + c);
  floatx4 r;
#pragma unroll
  for (int k = 0; k < 4; ++k) r[k] = fmaxf(fmaxf(a[k], b[k]), fmaxf(d[k], e[k]));
  *(floatx4*)(out + pix * c + ch) = r;
}

// channels-last bilinear resize, align_corners=True (ATen upsample_bilinear2d: src = scale*dst,
// scale = (in-1)/(out-1)); in [Hin, Win, C] -> out [Hout, Wout, C]
__global__ void __launch_bounds__(256) resize_bilinear_nhwc_kernel(const float* __restrict__ in,
                                                                   float* __restrict__ out, int c, int hin, int win,
                                                                   int hout, int wout, long total4) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total4) return;
  const int c4 = c >> 2;
  const int ch = (int)(idx % c4) * 4;
  const long pix = idx / c4;
  const int yo = (int)(pix / wout), xo = (int)(pix % wout);
  const float sy = hout > 1 ? (float)(hin - 1) / (float)(hout - 1) : 0.f;
  const float sx = wout > 1 ? (float)(win - 1) / (float)(wout - 1) : 0.f;
  const float fy = sy * (float)yo, fx = sx * (float)xo;
  const int y0 = min((int)fy, hin - 1), x0 = min((int)fx, win - 1);
  const int y1 = min(y0 + 1, hin - 1), x1 = min(x0 + 1, win - 1);
  const float ly1 = fy - (float)y0, lx1 = fx - (float)x0, ly0 = 1.f - ly1, lx0 = 1.f - lx1;
  const floatx4 a = *(const floatx4*)(in + ((size_t)y0 * win + x0) * c + ch);
  const floatx4 b = *(const floatx4*)(in + ((size_t)y0 * win + x1) * c + ch);
  const floatx4 d = *(const floatx4*)(in + ((size_t)y1 * win + x0) * c + ch);
  const floatx4 e = *(const floatx4*)(in + ((size_t)y1 * win + x1) * c + ch);
  floatx4 r;
#pragma unroll
  for (int k = 0; k < 4; ++k) r[k] = ly0 * (lx0 * a[k] + lx1 * b[k]) + ly1 * (lx0 * d[k] + lx1 * e[k]);
  *(floatx4*)(out + pix * c + ch) = r;
}

// [H, W, C] channels-last -> [C, H, W] planar (API boundary: the reference returns feat as NCHW)
__global__ void __launch_bounds__(256) nhwc_to_nchw_kernel(const float* __restrict__ in, float* __restrict__ out,
                                                           int c, long hw, long total) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  const int ch = (int)(idx / hw);
  const long pix = idx % hw;
  out[idx] = in[pix * c + ch];
}

}  // namespace dvd

using namespace dvd;

extern "C" int dvd_im2col3x3(const float* in, long sc, long sy, long sx, float* out, int ldo, int c, int h, int w,
                             void* stream) {
  DVD_REQUIRE(in && out, "im2col3x3: null pointer");
  DVD_REQUIRE(c > 0 && h > 0 && w > 0 && ldo >= 9 * c, "im2col3x3: bad shape");
  const long total = (long)h * w * ldo;
  if (sc == 1 && sx == c && sy == (long)w * c && c % 4 == 0 && ldo % 4 == 0 && ((uintptr_t)in % 16) == 0 &&
      ((uintptr_t)out % 16) == 0) {
    im2col3x3_nhwc4_kernel<<<cdiv(total / 4, 256), 256, 0, (hipStream_t)stream>>>(in, out, ldo, c, h, w, total / 4);
    return check_launch("im2col3x3(nhwc)");
  }
  im2col3x3_kernel<<<cdiv(total, 256), 256, 0, (hipStream_t)stream>>>(in, sc, sy, sx, out, ldo, c, h, w, total);
  return check_launch("im2col3x3");
}

extern "C" int dvd_maxpool2_nhwc(const float* in, float* out, int c, int h, int w, void* stream) {
  DVD_REQUIRE(in && out, "maxpool2: null pointer");
  DVD_REQUIRE(c % 4 == 0 && h % 2 == 0 && w % 2 == 0 && h > 0 && w > 0, "maxpool2: bad shape");
  const long total4 = (long)(h / 2) * (w / 2) * (c / 4);
  maxpool2_nhwc_kernel<<<cdiv(total4, 256), 256, 0, (hipStream_t)stream>>>(in, out, c, h, w, total4);
  return check_launch("maxpool2");
}

extern "C" int dvd_resize_bilinear_nhwc(const float* in, float* out, int c, int hin, int win, int hout, int wout,
                                        void* stream) {
  DVD_REQUIRE(in && out, "resize_bilinear: null pointer");
  DVD_REQUIRE(c % 4 == 0 && hin > 0 && win > 0 && hout > 0 && wout > 0, "resize_bilinear: bad shape");
  const long total4 = (long)hout * wout * (c / 4);
  resize_bilinear_nhwc_kernel<<<cdiv(total4, 256), 256, 0, (hipStream_t)stream>>>(in, out, c, hin, win, hout, wout,
                                                                                 total4);
  return check_launch("resize_bilinear");
}

extern "C" int dvd_nhwc_to_nchw(const float* in, float* out, int c, int h, int w, void* stream) {
  DVD_REQUIRE(in && out, "nhwc_to_nchw: null pointer");
  const long total = (long)c * h * w;
  nhwc_to_nchw_kernel<<<cdiv(total, 256), 256, 0, (hipStream_t)stream>>>(in, out, c, (long)h * w, total);
  return check_launch("nhwc_to_nchw");
}
