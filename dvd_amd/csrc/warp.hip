// Bilinear warps of the DvD sampling path: the per-step feature warp / drop-in grid_sample
// (datasets/utils/warping.py:50-73) and the fused full-resolution unwarp tail
// (train_settings/dvd/evaluation.py:301-306 + utils_flow/visualization_utils.py:75-77).
//
// All of these are HBM/L2-bound gathers: one lane per output pixel along x so that the grid
// reads, the taps of a smooth warp and the stores of a wave are contiguous runs.
#include "common.h"
#include "mfma.h"

namespace dvd {

// Unnormalise with align_corners=True: ((g + 1) / 2) * (size - 1)
__device__ __forceinline__ float unnorm(float g, int size) { return ((g + 1.f) * 0.5f) * (float)(size - 1); }

// Branch-free bilinear taps (zeros padding): out-of-range taps get weight 0 and a clamped (always valid)
// address, so all four loads of a pixel issue unconditionally and back-to-back.
struct Taps {
  int o00, o01, o10, o11;      // element offsets (row * pitch + col) of the four taps, clamped in range
  float w00, w01, w10, w11;
};

__device__ __forceinline__ Taps make_taps(float gx, float gy, int hin, int win, int pitch) {
  Taps t;
  const float ix = unnorm(gx, win), iy = unnorm(gy, hin);
  float fx = floorf(ix), fy = floorf(iy);
  const float ex = fx + 1.f, ey = fy + 1.f;
  float w00 = (ex - ix) * (ey - iy), w01 = (ix - fx) * (ey - iy);
  float w10 = (ex - ix) * (iy - fy), w11 = (ix - fx) * (iy - fy);
  // keep the int conversion safe for wild / non-finite coordinates
  fx = fminf(fmaxf(fx, -2.f), (float)win);
  fy = fminf(fmaxf(fy, -2.f), (float)hin);
  if (!(ix == ix)) fx = -2.f;
  if (!(iy == iy)) fy = -2.f;
  const int x0 = (int)fx, y0 = (int)fy;
  const bool x0ok = x0 >= 0 && x0 < win, x1ok = x0 + 1 >= 0 && x0 + 1 < win;
  const bool y0ok = y0 >= 0 && y0 < hin, y1ok = y0 + 1 >= 0 && y0 + 1 < hin;
  t.w00 = (x0ok && y0ok) ? w00 : 0.f;
  t.w01 = (x1ok && y0ok) ? w01 : 0.f;
  t.w10 = (x0ok && y1ok) ? w10 : 0.f;
  t.w11 = (x1ok && y1ok) ? w11 : 0.f;
  const int xc0 = min(max(x0, 0), win - 1), xc1 = min(max(x0 + 1, 0), win - 1);
  const int yc0 = min(max(y0, 0), hin - 1), yc1 = min(max(y0 + 1, 0), hin - 1);
  t.o00 = yc0 * pitch + xc0;
  t.o01 = yc0 * pitch + xc1;
  t.o10 = yc1 * pitch + xc0;
  t.o11 = yc1 * pitch + xc1;
  return t;
}

constexpr int PX = 4;   // output pixels per thread: PX consecutive ROWS at one x.  A wave-instruction then still
                        // reads/writes one contiguous run (1 px per lane) while 4x as many loads are in flight,
                        // and vertically adjacent pixels share their tap rows in L1.

// ---------------------------------------------------------------------------------------
// Drop-in grid_sample, NCHW f32.  One thread per (n, 4 y's, x).
// ---------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) grid_sample_nchw_kernel(const float* __restrict__ src,
                                                               const float* __restrict__ grid,
                                                               float* __restrict__ out, int c, int hin, int win,
                                                               int h, int w, int src_batch_div) {
  const int n = blockIdx.z;
  const int y0 = blockIdx.y * PX;
  const int x = blockIdx.x * blockDim.x + threadIdx.x;
  if (x >= w) return;
  const size_t hw = (size_t)h * w;
  const float* g = grid + (size_t)n * 2 * hw + x;
  Taps t[PX];
#pragma unroll
  for (int k = 0; k < PX; ++k) {
    const int yy = min(y0 + k, h - 1);
    t[k] = make_taps(g[(size_t)yy * w], g[hw + (size_t)yy * w], hin, win, win);
  }
  const size_t plane = (size_t)hin * win;
  const float* s = src + (size_t)(n / src_batch_div) * c * plane;
  float* o = out + (size_t)n * c * hw + x;
  for (int ch = 0; ch < c; ++ch) {
    const float* p = s + (size_t)ch * plane;
    float v[PX][4];
#pragma unroll
    for (int k = 0; k < PX; ++k) {
      v[k][0] = p[t[k].o00]; v[k][1] = p[t[k].o01]; v[k][2] = p[t[k].o10]; v[k][3] = p[t[k].o11];
    }
#pragma unroll
    for (int k = 0; k < PX; ++k)
      if (y0 + k < h)
        o[(size_t)ch * hw + (size_t)(y0 + k) * w] =
            ((v[k][0] * t[k].w00 + v[k][1] * t[k].w01) + v[k][2] * t[k].w10) + v[k][3] * t[k].w11;
  }
}

// ---------------------------------------------------------------------------------------
// Fused unwarp tail.  grid(i,j) is computed on the fly from the coarse flow (L2-resident:
// 2*G*G floats), so HBM traffic is src taps + output only.
// ---------------------------------------------------------------------------------------
struct UpParams {
  int g, h, w;
  float sy, sx;   // (g-1)/(h-1), (g-1)/(w-1): ATen area_pixel_compute_scale, align_corners=True
  float inv_w1, inv_h1;
  float scale;
};

__device__ __forceinline__ void flow_grid_at(const float* __restrict__ flow, const UpParams& p, int i, int j,
                                             float& gx, float& gy) {
  // F.interpolate(bilinear, align_corners=True): src = scale * dst
  float fy = p.sy * (float)i, fx = p.sx * (float)j;
  int y0 = (int)fy, x0 = (int)fx;
  y0 = min(y0, p.g - 1);
  x0 = min(x0, p.g - 1);
  int y1 = min(y0 + 1, p.g - 1), x1 = min(x0 + 1, p.g - 1);
  float ly1 = fy - (float)y0, lx1 = fx - (float)x0;
  float ly0 = 1.f - ly1, lx0 = 1.f - lx1;
  const int gg = p.g * p.g;
  float v[2];
#pragma unroll
  for (int ch = 0; ch < 2; ++ch) {
    const float* f = flow + ch * gg;
    v[ch] = ly0 * (lx0 * f[y0 * p.g + x0] + lx1 * f[y0 * p.g + x1]) +
            ly1 * (lx0 * f[y1 * p.g + x0] + lx1 * f[y1 * p.g + x1]);
  }
  float bx = (float)j * p.inv_w1, by = (float)i * p.inv_h1;
  gx = (((v[0] + bx) * 1.f) * 2.f - 1.f) * p.scale;
  gy = (((v[1] + by) * 1.f) * 2.f - 1.f) * p.scale;
}

__global__ void __launch_bounds__(256) unwarp_grid_kernel(const float* __restrict__ flow, float* __restrict__ grid,
                                                          UpParams p) {
  const int i = blockIdx.y;
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= p.w) return;
  float gx, gy;
  flow_grid_at(flow, p, i, j, gx, gy);
  grid[(size_t)i * p.w + j] = gx;
  grid[(size_t)p.h * p.w + (size_t)i * p.w + j] = gy;
}

__global__ void __launch_bounds__(256) unwarp_f32_kernel(const float* __restrict__ flow,
                                                         const float* __restrict__ src, float* __restrict__ out,
                                                         UpParams p) {
  const int i0 = blockIdx.y * PX;
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= p.w) return;
  Taps t[PX];
#pragma unroll
  for (int k = 0; k < PX; ++k) {
    float gx, gy;
    flow_grid_at(flow, p, min(i0 + k, p.h - 1), j, gx, gy);
    t[k] = make_taps(gx, gy, p.h, p.w, p.w);
  }
  const size_t plane = (size_t)p.h * p.w;
  float v[3][PX][4];
#pragma unroll
  for (int ch = 0; ch < 3; ++ch) {
    const float* q = src + (size_t)ch * plane;
#pragma unroll
    for (int k = 0; k < PX; ++k) {
      v[ch][k][0] = q[t[k].o00]; v[ch][k][1] = q[t[k].o01]; v[ch][k][2] = q[t[k].o10]; v[ch][k][3] = q[t[k].o11];
    }
  }
#pragma unroll
  for (int k = 0; k < PX; ++k) {
    if (i0 + k >= p.h) break;
    float* o = out + ((size_t)(i0 + k) * p.w + j) * 3;
#pragma unroll
    for (int ch = 0; ch < 3; ++ch)
      o[ch] = ((v[ch][k][0] * t[k].w00 + v[ch][k][1] * t[k].w01) + v[ch][k][2] * t[k].w10) + v[ch][k][3] * t[k].w11;
  }
}

__global__ void __launch_bounds__(256) unwarp_u8_kernel(const float* __restrict__ flow,
                                                        const uint8_t* __restrict__ src, uint8_t* __restrict__ out,
                                                        UpParams p) {
  const int i0 = blockIdx.y * PX;
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= p.w) return;
  Taps t[PX];
#pragma unroll
  for (int k = 0; k < PX; ++k) {
    float gx, gy;
    flow_grid_at(flow, p, min(i0 + k, p.h - 1), j, gx, gy);
    t[k] = make_taps(gx, gy, p.h, p.w, p.w);
  }
  uint8_t v[PX][4][3];
#pragma unroll
  for (int k = 0; k < PX; ++k) {
    const int offs[4] = {t[k].o00, t[k].o01, t[k].o10, t[k].o11};
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      const uint8_t* q = src + (size_t)offs[a] * 3;
      v[k][a][0] = q[0]; v[k][a][1] = q[1]; v[k][a][2] = q[2];
    }
  }
#pragma unroll
  for (int k = 0; k < PX; ++k) {
    if (i0 + k >= p.h) break;
    uint8_t* o = out + ((size_t)(i0 + k) * p.w + j) * 3;
#pragma unroll
    for (int ch = 0; ch < 3; ++ch) {
      const float a = (((float)v[k][0][ch] * t[k].w00 + (float)v[k][1][ch] * t[k].w01) +
                       (float)v[k][2][ch] * t[k].w10) + (float)v[k][3][ch] * t[k].w11;
      o[ch] = (uint8_t)(int)a;   // truncation, as numpy .astype(uint8) for 0 <= a < 256
    }
  }
}

static UpParams make_up(int g, int h, int w, float scale) {
  UpParams p;
  p.g = g;
  p.h = h;
  p.w = w;
  p.sy = h > 1 ? (float)(g - 1) / (float)(h - 1) : 0.f;
  p.sx = w > 1 ? (float)(g - 1) / (float)(w - 1) : 0.f;
  p.inv_w1 = w > 1 ? 1.f / (float)(w - 1) : 0.f;
  p.inv_h1 = h > 1 ? 1.f / (float)(h - 1) : 0.f;
  p.scale = scale;
  return p;
}

}  // namespace dvd

using namespace dvd;

extern "C" int dvd_grid_sample_bilinear_zeros_ac(const float* src, const float* grid, float* out, int n, int c,
                                                 int hin, int win, int h, int w, int src_batch_div, void* stream) {
  DVD_REQUIRE(src && grid && out, "grid_sample: null pointer");
  DVD_REQUIRE(n >= 0 && c > 0 && hin > 0 && win > 0 && h > 0 && w > 0 && src_batch_div > 0,
              "grid_sample: bad shape n=%d c=%d in=%dx%d out=%dx%d", n, c, hin, win, h, w);
  if (n == 0) return DVD_OK;
  DVD_REQUIRE(h <= 65535 && n <= 65535, "grid_sample: h or n exceeds the 65535 grid limit");
  DVD_REQUIRE(cdiv(h, 4) <= 65535, "grid_sample: h too large");
  const int bx = w >= 256 ? 256 : (w > 64 ? 128 : 64);
  dim3 grd(cdiv(w, bx), cdiv(h, 4), n);
  grid_sample_nchw_kernel<<<grd, bx, 0, (hipStream_t)stream>>>(src, grid, out, c, hin, win, h, w, src_batch_div);
  return check_launch("grid_sample");
}

static int unwarp_args(const void* flow, const void* src, const void* out, int g, int h, int w) {
  DVD_REQUIRE(flow && src && out, "unwarp: null pointer");
  DVD_REQUIRE(g >= 2 && h >= 1 && w >= 1 && h <= 65535, "unwarp: bad shape g=%d h=%d w=%d", g, h, w);
  return DVD_OK;
}

extern "C" int dvd_unwarp_grid(const float* flow, int g, float* grid_out, int h, int w, float scale,
                               void* stream) {
  if (int e = unwarp_args(flow, flow, grid_out, g, h, w)) return e;
  dim3 grd(cdiv(w, 256), h);
  unwarp_grid_kernel<<<grd, 256, 0, (hipStream_t)stream>>>(flow, grid_out, make_up(g, h, w, scale));
  return check_launch("unwarp_grid");
}

extern "C" int dvd_unwarp_f32(const float* flow, int g, const float* src_chw, float* out_hwc, int h, int w,
                              float scale, void* stream) {
  if (int e = unwarp_args(flow, src_chw, out_hwc, g, h, w)) return e;
  dim3 grd(cdiv(w, 256), cdiv(h, 4));
  unwarp_f32_kernel<<<grd, 256, 0, (hipStream_t)stream>>>(flow, src_chw, out_hwc, make_up(g, h, w, scale));
  return check_launch("unwarp_f32");
}

extern "C" int dvd_unwarp_u8(const float* flow, int g, const uint8_t* src_hwc, uint8_t* out_hwc, int h, int w,
                             float scale, void* stream) {
  if (int e = unwarp_args(flow, src_hwc, out_hwc, g, h, w)) return e;
  dim3 grd(cdiv(w, 256), cdiv(h, 4));
  unwarp_u8_kernel<<<grd, 256, 0, (hipStream_t)stream>>>(flow, src_hwc, out_hwc, make_up(g, h, w, scale));
  return check_launch("unwarp_u8");
}
