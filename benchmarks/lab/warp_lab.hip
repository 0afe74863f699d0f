// Stand-alone laboratory for the full-resolution gather (K20): variants of the drop-in grid_sample kernel
// timed against a pure-copy ceiling of the same byte count.  Not part of the product library.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 warp_lab.hip -o warp_lab && ./warp_lab
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));
struct __attribute__((packed, aligned(4))) pf2 { float a, b; };

__device__ __forceinline__ float unnorm(float g, int size) { return ((g + 1.f) * 0.5f) * (float)(size - 1); }

// ---------------- V0: copy ceiling: read nin float4, write nout float4 --------------------------------------
__global__ void __launch_bounds__(256) copy_kernel(const f4* __restrict__ a, const f4* __restrict__ b, f4* __restrict__ o,
                                                   size_t n3) {
  // per index i < n3: reads a[3i..3i+2] (src 12 B/px x4 px), b[2i..2i+1] (grid), writes o[3i..3i+2]
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n3) return;
  f4 s0 = a[i], s1 = a[i + n3], s2 = a[i + 2 * n3], g0 = b[i], g1 = b[i + n3];
  o[i] = s0 + g0; o[i + n3] = s1 + g1; o[i + 2 * n3] = s2 + g0;
}

// ---------------- V1: the shipped structure: 1 px in x per lane, 4 rows per thread ---------------------------
struct Taps { int o00, o01, o10, o11; float w00, w01, w10, w11; };
__device__ __forceinline__ Taps make_taps(float gx, float gy, int hin, int win, int pitch) {
  Taps t;
  const float ix = unnorm(gx, win), iy = unnorm(gy, hin);
  float fx = floorf(ix), fy = floorf(iy);
  const float ex = fx + 1.f, ey = fy + 1.f;
  float w00 = (ex - ix) * (ey - iy), w01 = (ix - fx) * (ey - iy);
  float w10 = (ex - ix) * (iy - fy), w11 = (ix - fx) * (iy - fy);
  fx = fminf(fmaxf(fx, -2.f), (float)win);
  fy = fminf(fmaxf(fy, -2.f), (float)hin);
  if (!(ix == ix)) fx = -2.f;
  if (!(iy == iy)) fy = -2.f;
  const int x0 = (int)fx, y0 = (int)fy;
  const bool x0ok = x0 >= 0 && x0 < win, x1ok = x0 + 1 >= 0 && x0 + 1 < win;
  const bool y0ok = y0 >= 0 && y0 < hin, y1ok = y0 + 1 >= 0 && y0 + 1 < hin;
  t.w00 = (x0ok && y0ok) ? w00 : 0.f; t.w01 = (x1ok && y0ok) ? w01 : 0.f;
  t.w10 = (x0ok && y1ok) ? w10 : 0.f; t.w11 = (x1ok && y1ok) ? w11 : 0.f;
  const int xc0 = min(max(x0, 0), win - 1), xc1 = min(max(x0 + 1, 0), win - 1);
  const int yc0 = min(max(y0, 0), hin - 1), yc1 = min(max(y0 + 1, 0), hin - 1);
  t.o00 = yc0 * pitch + xc0; t.o01 = yc0 * pitch + xc1; t.o10 = yc1 * pitch + xc0; t.o11 = yc1 * pitch + xc1;
  return t;
}
constexpr int PX = 4;
__global__ void __launch_bounds__(256) v1_kernel(const float* __restrict__ src, const float* __restrict__ grid,
                                                 float* __restrict__ out, int c, int hin, int win, int h, int w) {
  const int y0 = blockIdx.y * PX;
  const int x = blockIdx.x * blockDim.x + threadIdx.x;
  if (x >= w) return;
  const size_t hw = (size_t)h * w;
  const float* g = grid + x;
  Taps t[PX];
#pragma unroll
  for (int k = 0; k < PX; ++k) {
    const int yy = min(y0 + k, h - 1);
    t[k] = make_taps(g[(size_t)yy * w], g[hw + (size_t)yy * w], hin, win, win);
  }
  const size_t plane = (size_t)hin * win;
  float* o = out + x;
  for (int ch = 0; ch < c; ++ch) {
    const float* p = src + (size_t)ch * plane;
    float v[PX][4];
#pragma unroll
    for (int k = 0; k < PX; ++k) { v[k][0] = p[t[k].o00]; v[k][1] = p[t[k].o01]; v[k][2] = p[t[k].o10]; v[k][3] = p[t[k].o11]; }
#pragma unroll
    for (int k = 0; k < PX; ++k)
      if (y0 + k < h)
        o[(size_t)ch * hw + (size_t)(y0 + k) * w] = ((v[k][0] * t[k].w00 + v[k][1] * t[k].w01) + v[k][2] * t[k].w10) + v[k][3] * t[k].w11;
  }
}

// ---------------- V2: 4 px in x per thread, float4 grid loads / stores, 8-byte pair gathers ------------------
// Pair taps: the two x-neighbours of a row are one 8-byte load at column bx = clamp(x0, 0, win-2); weights are
// re-targeted so that an out-of-range tap contributes 0 exactly as zeros padding does.
struct PTaps { int r0, r1; float a00, a01, a10, a11; };   // row offsets (yc*pitch+bx) and the weights of (bx, bx+1)
__device__ __forceinline__ PTaps make_ptaps(float gx, float gy, int hin, int win, int pitch) {
  PTaps t;
  const float ix = unnorm(gx, win), iy = unnorm(gy, hin);
  float fx = floorf(ix), fy = floorf(iy);
  const float ex = fx + 1.f, ey = fy + 1.f;
  const float wx0 = ex - ix, wx1 = ix - fx, wy0 = ey - iy, wy1 = iy - fy;
  fx = fminf(fmaxf(fx, -2.f), (float)win);
  fy = fminf(fmaxf(fy, -2.f), (float)hin);
  if (!(ix == ix)) fx = -2.f;
  if (!(iy == iy)) fy = -2.f;
  const int x0 = (int)fx, y0 = (int)fy;
  const int bx = min(max(x0, 0), win - 2);
  // weight landing on column bx / bx+1
  const float cl = (x0 == bx) ? wx0 : ((x0 + 1 == bx) ? wx1 : 0.f);
  const float cr = (x0 + 1 == bx + 1) ? wx1 : ((x0 == bx + 1) ? wx0 : 0.f);
  const bool y0ok = y0 >= 0 && y0 < hin, y1ok = y0 + 1 >= 0 && y0 + 1 < hin;
  t.a00 = y0ok ? cl * wy0 : 0.f;
  t.a01 = y0ok ? cr * wy0 : 0.f;
  t.a10 = y1ok ? cl * wy1 : 0.f;
  t.a11 = y1ok ? cr * wy1 : 0.f;
  const int yc0 = min(max(y0, 0), hin - 1), yc1 = min(max(y0 + 1, 0), hin - 1);
  t.r0 = yc0 * pitch + bx;
  t.r1 = yc1 * pitch + bx;
  return t;
}
__device__ __forceinline__ f2 ld2(const float* p) {
  pf2 v = *reinterpret_cast<const pf2*>(p);
  return f2{v.a, v.b};
}
__device__ __forceinline__ f2 ld2b(const float* base, unsigned byte_off) {
  pf2 v = *reinterpret_cast<const pf2*>(reinterpret_cast<const char*>(base) + byte_off);
  return f2{v.a, v.b};
}
template <int ROWS>
__global__ void __launch_bounds__(256) v2_kernel(const float* __restrict__ src, const float* __restrict__ grid,
                                                 float* __restrict__ out, int c, int hin, int win, int h, int w) {
  // block = 64 lanes x 4 waves: wave wv handles rows y = (blockIdx.y*4 + wv)*ROWS .. +ROWS-1; lane handles 4 px in x
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int x = (blockIdx.x * 64 + lane) * 4;
  const int yb = (blockIdx.y * 4 + wv) * ROWS;
  if (x >= w || yb >= h) return;
  const size_t hw = (size_t)h * w;
  const size_t plane = (size_t)hin * win;
#pragma unroll
  for (int r = 0; r < ROWS; ++r) {
    const int y = yb + r;
    if (y >= h) break;
    const f4 gx = *reinterpret_cast<const f4*>(grid + (size_t)y * w + x);
    const f4 gy = *reinterpret_cast<const f4*>(grid + hw + (size_t)y * w + x);
    PTaps t[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) t[k] = make_ptaps(gx[k], gy[k], hin, win, win);
    for (int ch = 0; ch < c; ++ch) {
      const float* p = src + (size_t)ch * plane;
      f2 u[4], d[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) { u[k] = ld2(p + t[k].r0); d[k] = ld2(p + t[k].r1); }
      f4 o;
#pragma unroll
      for (int k = 0; k < 4; ++k) o[k] = ((u[k][0] * t[k].a00 + u[k][1] * t[k].a01) + d[k][0] * t[k].a10) + d[k][1] * t[k].a11;
      *reinterpret_cast<f4*>(out + (size_t)ch * hw + (size_t)y * w + x) = o;
    }
  }
}


// ---------------- V3: like V2 but all ROWS rows' grid loads are issued before any gather (explicit pipelining),
// streaming (non-temporal) grid loads and output stores ------------------------------------------------------
template <int ROWS, bool NT>
__global__ void __launch_bounds__(256) v3_kernel(const float* __restrict__ src, const float* __restrict__ grid,
                                                 float* __restrict__ out, int c, int hin, int win, int h, int w) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int x = (blockIdx.x * 64 + lane) * 4;
  const int yb = (blockIdx.y * 4 + wv) * ROWS;
  if (x >= w || yb >= h) return;
  const size_t hw = (size_t)h * w;
  const size_t plane = (size_t)hin * win;
  f4 gx[ROWS], gy[ROWS];
#pragma unroll
  for (int r = 0; r < ROWS; ++r) {
    const int y = min(yb + r, h - 1);
    const f4* px = reinterpret_cast<const f4*>(grid + (size_t)y * w + x);
    const f4* py = reinterpret_cast<const f4*>(grid + hw + (size_t)y * w + x);
    gx[r] = NT ? __builtin_nontemporal_load(px) : *px;
    gy[r] = NT ? __builtin_nontemporal_load(py) : *py;
  }
#pragma unroll
  for (int r = 0; r < ROWS; ++r) {
    const int y = yb + r;
    PTaps t[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) t[k] = make_ptaps(gx[r][k], gy[r][k], hin, win, win);
    f2 u[3][4], d[3][4];
#pragma unroll
    for (int ch = 0; ch < 3; ++ch)
#pragma unroll
      for (int k = 0; k < 4; ++k) { u[ch][k] = ld2(src + ch * plane + t[k].r0); d[ch][k] = ld2(src + ch * plane + t[k].r1); }
    if (y < h) {
#pragma unroll
      for (int ch = 0; ch < 3; ++ch) {
        f4 o;
#pragma unroll
        for (int k = 0; k < 4; ++k) o[k] = ((u[ch][k][0] * t[k].a00 + u[ch][k][1] * t[k].a01) + d[ch][k][0] * t[k].a10) + d[ch][k][1] * t[k].a11;
        f4* po = reinterpret_cast<f4*>(out + (size_t)ch * hw + (size_t)y * w + x);
        if (NT) __builtin_nontemporal_store(o, po); else *po = o;
      }
    }
  }
}

// ---------------- V4: lane-consecutive pixels (px = lane + 64*k): every gather instruction covers one contiguous
// run; dword grid loads and stores ----------------------------------------------------------------------------
template <int KP>
__global__ void __launch_bounds__(256) v4_kernel(const float* __restrict__ src, const float* __restrict__ grid,
                                                 float* __restrict__ out, int c, int hin, int win, int h, int w) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int x0 = blockIdx.x * 64 * KP + lane;
  const int y = blockIdx.y * 4 + wv;
  if (y >= h) return;
  const size_t hw = (size_t)h * w;
  const size_t plane = (size_t)hin * win;
  float gx[KP], gy[KP];
#pragma unroll
  for (int k = 0; k < KP; ++k) {
    const int x = min(x0 + 64 * k, w - 1);
    gx[k] = grid[(size_t)y * w + x];
    gy[k] = grid[hw + (size_t)y * w + x];
  }
  PTaps t[KP];
#pragma unroll
  for (int k = 0; k < KP; ++k) t[k] = make_ptaps(gx[k], gy[k], hin, win, win);
  f2 u[3][KP], d[3][KP];
#pragma unroll
  for (int ch = 0; ch < 3; ++ch)
#pragma unroll
    for (int k = 0; k < KP; ++k) { u[ch][k] = ld2(src + ch * plane + t[k].r0); d[ch][k] = ld2(src + ch * plane + t[k].r1); }
#pragma unroll
  for (int ch = 0; ch < 3; ++ch)
#pragma unroll
    for (int k = 0; k < KP; ++k)
      if (x0 + 64 * k < w)
        out[(size_t)ch * hw + (size_t)y * w + x0 + 64 * k] =
            ((u[ch][k][0] * t[k].a00 + u[ch][k][1] * t[k].a01) + d[ch][k][0] * t[k].a10) + d[ch][k][1] * t[k].a11;
}


// ---------------- V5: V4(KP=1) with an XCD-aware block map: the blocks one XCD receives (linear id % 8) cover one
// contiguous band of rows, so vertically shared tap rows hit in that XCD's L2 ----------------------------------
template <int WAVES>
__global__ void __launch_bounds__(WAVES * 64) v5_kernel(const float* __restrict__ src, const float* __restrict__ grid,
                                                 float* __restrict__ out, int c, int hin, int win, int h, int w, int nbx, int nby) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int total = nbx * nby;
  const int L = blockIdx.x;
  const int per = (total + 7) / 8;
  const int v = (L & 7) * per + (L >> 3);
  if (v >= total) return;
  const int by = v / nbx, bx = v - by * nbx;
  const int x = bx * 64 + lane;
  const int y = by * WAVES + wv;
  if (y >= h) return;
  const size_t hw = (size_t)h * w;
  const size_t plane = (size_t)hin * win;
  const int xc = min(x, w - 1);
  const float gx = grid[(size_t)y * w + xc], gy = grid[hw + (size_t)y * w + xc];
  const PTaps t = make_ptaps(gx, gy, hin, win, win);
  f2 u[3], d[3];
#pragma unroll
  for (int ch = 0; ch < 3; ++ch) { u[ch] = ld2(src + ch * plane + t.r0); d[ch] = ld2(src + ch * plane + t.r1); }
  if (x < w)
#pragma unroll
    for (int ch = 0; ch < 3; ++ch)
      out[(size_t)ch * hw + (size_t)y * w + x] = ((u[ch][0] * t.a00 + u[ch][1] * t.a01) + d[ch][0] * t.a10) + d[ch][1] * t.a11;
}


// ---------------- V6: V4(KP) with streaming hints; V7: grid computed analytically (no dependent grid load) ------
template <int KP, int MODE>
__global__ void __launch_bounds__(256) v6_kernel(const float* __restrict__ src, const float* __restrict__ grid,
                                                 float* __restrict__ out, int c, int hin, int win, int h, int w) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int x0 = blockIdx.x * 64 * KP + lane;
  const int y = blockIdx.y * 4 + wv;
  if (y >= h) return;
  const size_t hw = (size_t)h * w;
  const size_t plane = (size_t)hin * win;
  float gx[KP], gy[KP];
#pragma unroll
  for (int k = 0; k < KP; ++k) {
    const int x = min(x0 + 64 * k, w - 1);
    if (MODE == 1) {   // analytic grid: same smooth field as the host builds, no memory dependence
      float u = (float)x * (1.f / 2479.f), v = (float)y * (1.f / 3507.f);
      gx[k] = ((u + 0.04f * __sinf(3.1f * v + 0.5f) * __cosf(2.3f * u)) * 2.f - 1.f) * 0.987f;
      gy[k] = ((v + 0.05f * __sinf(2.7f * u + 0.3f) * __cosf(1.9f * v)) * 2.f - 1.f) * 0.987f;
    } else {
      gx[k] = __builtin_nontemporal_load(grid + (size_t)y * w + x);
      gy[k] = __builtin_nontemporal_load(grid + hw + (size_t)y * w + x);
    }
  }
  PTaps t[KP];
#pragma unroll
  for (int k = 0; k < KP; ++k) t[k] = make_ptaps(gx[k], gy[k], hin, win, win);
  f2 u[3][KP], d[3][KP];
#pragma unroll
  for (int ch = 0; ch < 3; ++ch)
#pragma unroll
    for (int k = 0; k < KP; ++k) { u[ch][k] = ld2(src + ch * plane + t[k].r0); d[ch][k] = ld2(src + ch * plane + t[k].r1); }
#pragma unroll
  for (int ch = 0; ch < 3; ++ch)
#pragma unroll
    for (int k = 0; k < KP; ++k)
      if (x0 + 64 * k < w)
        __builtin_nontemporal_store(((u[ch][k][0] * t[k].a00 + u[ch][k][1] * t[k].a01) + d[ch][k][0] * t[k].a10) + d[ch][k][1] * t[k].a11,
                                    out + (size_t)ch * hw + (size_t)y * w + x0 + 64 * k);
}
// dword copy with the same stream structure (5 dword loads, 3 dword stores per px): ceiling for 4-byte lanes
__global__ void __launch_bounds__(256) copy4_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ o, size_t n) {
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  float s0 = a[i], s1 = a[i + n], s2 = a[i + 2 * n], g0 = b[i], g1 = b[i + n];
  o[i] = s0 + g0; o[i + n] = s1 + g1; o[i + 2 * n] = s2 + g0;
}
// dependent copy: the src index depends on the loaded grid value (identity in effect)
__global__ void __launch_bounds__(256) copydep_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ o, size_t n) {
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  float g0 = b[i], g1 = b[i + n];
  size_t j = i + (size_t)(int)(fminf(fabsf(g0 * g1), 0.5f));     // always i (|g| < 1 -> product < 1 -> min(.,0.5) -> int 0)
  float s0 = a[j], s1 = a[j + n], s2 = a[j + 2 * n];
  o[i] = s0 + g0; o[i + n] = s1 + g1; o[i + 2 * n] = s2 + g0;
}


// ---------------- V8: V6 with uniform (scalar) plane/row bases and 32-bit lane offsets ------------------------
template <int KP>
__global__ void __launch_bounds__(256) v8_kernel(const float* __restrict__ src, const float* __restrict__ grid,
                                                 float* __restrict__ out, int c, int hin, int win, int h, int w) {
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int y = blockIdx.y * 4 + wv;          // wave-uniform
  if (y >= h) return;
  const unsigned x0 = blockIdx.x * 64 * KP + lane;
  const size_t hw = (size_t)h * w;
  const unsigned plane = (unsigned)hin * (unsigned)win;
  const float* grow = grid + (size_t)y * w;     // uniform
  float gx[KP], gy[KP];
#pragma unroll
  for (int k = 0; k < KP; ++k) {
    const unsigned x = min(x0 + 64u * k, (unsigned)w - 1u);
    gx[k] = __builtin_nontemporal_load(reinterpret_cast<const float*>(reinterpret_cast<const char*>(grow) + x * 4u));
    gy[k] = __builtin_nontemporal_load(reinterpret_cast<const float*>(reinterpret_cast<const char*>(grow + hw) + x * 4u));
  }
  PTaps t[KP];
#pragma unroll
  for (int k = 0; k < KP; ++k) t[k] = make_ptaps(gx[k], gy[k], hin, win, win);
  f2 u[3][KP], d[3][KP];
#pragma unroll
  for (int ch = 0; ch < 3; ++ch) {
    const float* pc = src + (size_t)ch * plane;  // uniform
#pragma unroll
    for (int k = 0; k < KP; ++k) { u[ch][k] = ld2b(pc, (unsigned)t[k].r0 * 4u); d[ch][k] = ld2b(pc, (unsigned)t[k].r1 * 4u); }
  }
  float* orow = out + (size_t)y * w;             // uniform
#pragma unroll
  for (int ch = 0; ch < 3; ++ch)
#pragma unroll
    for (int k = 0; k < KP; ++k)
      if (x0 + 64u * k < (unsigned)w)
        __builtin_nontemporal_store(((u[ch][k][0] * t[k].a00 + u[ch][k][1] * t[k].a01) + d[ch][k][0] * t[k].a10) + d[ch][k][1] * t[k].a11,
                                    reinterpret_cast<float*>(reinterpret_cast<char*>(orow + (size_t)ch * hw) + (x0 + 64u * k) * 4u));
}


// ---------------- V9: 2-D tile per wave (TW x 64/TW pixels), 4 waves side by side in x ------------------------
template <int TW>
__global__ void __launch_bounds__(256) v9_kernel(const float* __restrict__ src, const float* __restrict__ grid,
                                                 float* __restrict__ out, int c, int hin, int win, int h, int w) {
  constexpr int TH = 64 / TW;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int x = (blockIdx.x * 4 + wv) * TW + (lane % TW);
  const int y = blockIdx.y * TH + lane / TW;
  if (x >= w || y >= h) return;
  const size_t hw = (size_t)h * w;
  const unsigned plane = (unsigned)hin * (unsigned)win;
  const unsigned pix = (unsigned)y * (unsigned)w + (unsigned)x;
  const float gx = __builtin_nontemporal_load(reinterpret_cast<const float*>(reinterpret_cast<const char*>(grid) + pix * 4u));
  const float gy = __builtin_nontemporal_load(reinterpret_cast<const float*>(reinterpret_cast<const char*>(grid + hw) + pix * 4u));
  const PTaps t = make_ptaps(gx, gy, hin, win, win);
  f2 u[3], d[3];
#pragma unroll
  for (int ch = 0; ch < 3; ++ch) { u[ch] = ld2b(src + (size_t)ch * plane, (unsigned)t.r0 * 4u); d[ch] = ld2b(src + (size_t)ch * plane, (unsigned)t.r1 * 4u); }
#pragma unroll
  for (int ch = 0; ch < 3; ++ch)
    __builtin_nontemporal_store(((u[ch][0] * t.a00 + u[ch][1] * t.a01) + d[ch][0] * t.a10) + d[ch][1] * t.a11,
                                reinterpret_cast<float*>(reinterpret_cast<char*>(out + (size_t)ch * hw) + pix * 4u));
}


// ---------------- V10: wave per (KR consecutive rows x 64 px): lane owns KR vertically adjacent pixels ---------
template <int KR>
__global__ void __launch_bounds__(256) v10_kernel(const float* __restrict__ src, const float* __restrict__ grid,
                                                 float* __restrict__ out, int c, int hin, int win, int h, int w) {
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int yb = (blockIdx.y * 4 + wv) * KR;          // wave-uniform
  if (yb >= h) return;
  const unsigned x = min(blockIdx.x * 64u + lane, (unsigned)w - 1u);
  const bool live = blockIdx.x * 64u + lane < (unsigned)w;
  const size_t hw = (size_t)h * w;
  const unsigned plane = (unsigned)hin * (unsigned)win;
  PTaps t[KR];
#pragma unroll
  for (int k = 0; k < KR; ++k) {
    const float* grow = grid + (size_t)min(yb + k, h - 1) * w;
    const float gx = __builtin_nontemporal_load(reinterpret_cast<const float*>(reinterpret_cast<const char*>(grow) + x * 4u));
    const float gy = __builtin_nontemporal_load(reinterpret_cast<const float*>(reinterpret_cast<const char*>(grow + hw) + x * 4u));
    t[k] = make_ptaps(gx, gy, hin, win, win);
  }
  f2 u[3][KR], d[3][KR];
#pragma unroll
  for (int ch = 0; ch < 3; ++ch) {
    const float* pc = src + (size_t)ch * plane;
#pragma unroll
    for (int k = 0; k < KR; ++k) { u[ch][k] = ld2b(pc, (unsigned)t[k].r0 * 4u); d[ch][k] = ld2b(pc, (unsigned)t[k].r1 * 4u); }
  }
#pragma unroll
  for (int ch = 0; ch < 3; ++ch)
#pragma unroll
    for (int k = 0; k < KR; ++k)
      if (live && yb + k < h)
        __builtin_nontemporal_store(((u[ch][k][0] * t[k].a00 + u[ch][k][1] * t[k].a01) + d[ch][k][0] * t[k].a10) + d[ch][k][1] * t[k].a11,
                                    reinterpret_cast<float*>(reinterpret_cast<char*>(out + (size_t)ch * hw + (size_t)(yb + k) * w) + x * 4u));
}

// ---------------- host --------------------------------------------------------------------------------------
static void cpu_ref(const std::vector<float>& src, const std::vector<float>& grid, std::vector<float>& out, int c, int hin,
                    int win, int h, int w, int y_lo, int y_hi) {
  for (int y = y_lo; y < y_hi; ++y)
    for (int x = 0; x < w; ++x) {
      float gx = grid[(size_t)y * w + x], gy = grid[(size_t)h * w + (size_t)y * w + x];
      float ix = ((gx + 1.f) * 0.5f) * (float)(win - 1), iy = ((gy + 1.f) * 0.5f) * (float)(hin - 1);
      float fx = floorf(ix), fy = floorf(iy);
      int x0 = (int)fx, y0 = (int)fy;
      float w00 = (fx + 1.f - ix) * (fy + 1.f - iy), w01 = (ix - fx) * (fy + 1.f - iy);
      float w10 = (fx + 1.f - ix) * (iy - fy), w11 = (ix - fx) * (iy - fy);
      for (int ch = 0; ch < c; ++ch) {
        auto at = [&](int yy, int xx) { return (yy >= 0 && yy < hin && xx >= 0 && xx < win) ? src[(size_t)ch * hin * win + (size_t)yy * win + xx] : 0.f; };
        out[(size_t)ch * h * w + (size_t)y * w + x] = ((at(y0, x0) * w00 + at(y0, x0 + 1) * w01) + at(y0 + 1, x0) * w10) + at(y0 + 1, x0 + 1) * w11;
      }
    }
}

int main(int argc, char** argv) {
  const int H = 3508, W = 2480, C = 3, NSET = 3, ITERS = 30;
  const float AMP = argc > 1 ? (float)atof(argv[1]) : 1.f;
  printf("field amplitude x%g\n", AMP);
  const size_t hw = (size_t)H * W;
  std::vector<float> hsrc(C * hw), hgrid(2 * hw), hout(C * hw), href(C * hw);
  unsigned s = 12345;
  for (auto& v : hsrc) { s = s * 1664525u + 1013904223u; v = (float)(s >> 8) * (255.f / 16777216.f); }
  for (int y = 0; y < H; ++y)
    for (int x = 0; x < W; ++x) {
      float u = (float)x / (W - 1), v = (float)y / (H - 1);
      float fx = AMP * 0.04f * sinf(3.1f * v + 0.5f) * cosf(2.3f * u), fy = AMP * 0.05f * sinf(2.7f * u + 0.3f) * cosf(1.9f * v);
      hgrid[(size_t)y * W + x] = ((u + fx) * 2.f - 1.f) * 0.987f;
      hgrid[hw + (size_t)y * W + x] = ((v + fy) * 2.f - 1.f) * 0.987f;
    }
  float *src[NSET], *grid[NSET], *out[NSET];
  for (int i = 0; i < NSET; ++i) {
    CK(hipMalloc(&src[i], C * hw * 4)); CK(hipMalloc(&grid[i], 2 * hw * 4)); CK(hipMalloc(&out[i], C * hw * 4));
    CK(hipMemcpy(src[i], hsrc.data(), C * hw * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(grid[i], hgrid.data(), 2 * hw * 4, hipMemcpyHostToDevice));
  }
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const double bytes = 32.0 * hw;
  const int ylo = 1700, yhi = 1708;
  cpu_ref(hsrc, hgrid, href, C, H, W, H, W, ylo, yhi);
  std::vector<float> edge_ref(C * hw);
  auto run = [&](const char* name, auto launch, bool check) {
    for (int i = 0; i < 3; ++i) launch(i % NSET);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int i = 0; i < ITERS; ++i) launch(i % NSET);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= ITERS;
    double maxerr = -1; int wx = 0, wy = 0, wc = 0;
    if (check) {
      CK(hipMemcpy(hout.data(), out[0], C * hw * 4, hipMemcpyDeviceToHost));
      maxerr = 0;
      for (int ch = 0; ch < C; ++ch)
        for (int y = ylo; y < yhi; ++y)
          for (int x = 0; x < W; ++x) {
            size_t i = (size_t)ch * hw + (size_t)y * W + x;
            double e = fabs((double)hout[i] - href[i]);
            if (e > maxerr) { maxerr = e; wx = x; wy = y; wc = ch; }
          }
      if (maxerr > 1e-3) {
        size_t i = (size_t)wc * hw + (size_t)wy * W + wx;
        float gx = hgrid[(size_t)wy * W + wx], gy = hgrid[hw + (size_t)wy * W + wx];
        printf("   worst at ch %d y %d x %d: got %.6f ref %.6f  ix %.6f iy %.6f\n", wc, wy, wx, hout[i], href[i],
               ((gx + 1.f) * 0.5f) * (W - 1), ((gy + 1.f) * 0.5f) * (H - 1));
      }
    }
    printf("%-28s %8.4f ms  %7.1f GB/s (32 B/px)  maxerr %g\n", name, ms, bytes / ms * 1e-6, maxerr);
  };
  const size_t n3 = hw / 4;   // float4 count per plane
  run("V0 copy ceiling", [&](int k) { copy_kernel<<<(n3 + 255) / 256, 256>>>((const f4*)src[k], (const f4*)grid[k], (f4*)out[k], n3); }, false);
  run("V1 shipped (1px x 4rows)", [&](int k) { v1_kernel<<<dim3((W + 255) / 256, (H + 3) / 4), 256>>>(src[k], grid[k], out[k], C, H, W, H, W); }, true);
  run("V2 4px pair-gather ROWS=1", [&](int k) { v2_kernel<1><<<dim3((W / 4 + 63) / 64, (H + 3) / 4), 256>>>(src[k], grid[k], out[k], C, H, W, H, W); }, true);
  run("V2 4px pair-gather ROWS=2", [&](int k) { v2_kernel<2><<<dim3((W / 4 + 63) / 64, (H + 7) / 8), 256>>>(src[k], grid[k], out[k], C, H, W, H, W); }, true);
  run("V2 4px pair-gather ROWS=4", [&](int k) { v2_kernel<4><<<dim3((W / 4 + 63) / 64, (H + 15) / 16), 256>>>(src[k], grid[k], out[k], C, H, W, H, W); }, true);
  run("V3 pipelined ROWS=1 nt", [&](int k) { v3_kernel<1, true><<<dim3((W / 4 + 63) / 64, (H + 3) / 4), 256>>>(src[k], grid[k], out[k], C, H, W, H, W); }, true);
  run("V3 pipelined ROWS=2", [&](int k) { v3_kernel<2, false><<<dim3((W / 4 + 63) / 64, (H + 7) / 8), 256>>>(src[k], grid[k], out[k], C, H, W, H, W); }, true);
  run("V3 pipelined ROWS=2 nt", [&](int k) { v3_kernel<2, true><<<dim3((W / 4 + 63) / 64, (H + 7) / 8), 256>>>(src[k], grid[k], out[k], C, H, W, H, W); }, true);
  run("V3 pipelined ROWS=4 nt", [&](int k) { v3_kernel<4, true><<<dim3((W / 4 + 63) / 64, (H + 15) / 16), 256>>>(src[k], grid[k], out[k], C, H, W, H, W); }, true);
  run("V4 lane-consecutive KP=1", [&](int k) { v4_kernel<1><<<dim3((W + 63) / 64, (H + 3) / 4), 256>>>(src[k], grid[k], out[k], C, H, W, H, W); }, true);
  run("V4 lane-consecutive KP=2", [&](int k) { v4_kernel<2><<<dim3((W + 127) / 128, (H + 3) / 4), 256>>>(src[k], grid[k], out[k], C, H, W, H, W); }, true);
  run("V4 lane-consecutive KP=4", [&](int k) { v4_kernel<4><<<dim3((W + 255) / 256, (H + 3) / 4), 256>>>(src[k], grid[k], out[k], C, H, W, H, W); }, true);
  { int nbx = (W + 63) / 64, nby = (H + 3) / 4; int tot = ((nbx * nby + 7) / 8) * 8;
    run("V5 xcd-band 4 waves", [&](int k) { v5_kernel<4><<<tot, 256>>>(src[k], grid[k], out[k], C, H, W, H, W, nbx, nby); }, true); }
  { int nbx = (W + 63) / 64, nby = (H + 7) / 8; int tot = ((nbx * nby + 7) / 8) * 8;
    run("V5 xcd-band 8 waves", [&](int k) { v5_kernel<8><<<tot, 512>>>(src[k], grid[k], out[k], C, H, W, H, W, nbx, nby); }, true); }
  { int nbx = (W + 63) / 64, nby = (H + 15) / 16; int tot = ((nbx * nby + 7) / 8) * 8;
    run("V5 xcd-band 16 waves", [&](int k) { v5_kernel<16><<<tot, 1024>>>(src[k], grid[k], out[k], C, H, W, H, W, nbx, nby); }, true); }
  run("V6 KP=1 nt", [&](int k) { v6_kernel<1, 0><<<dim3((W + 63) / 64, (H + 3) / 4), 256>>>(src[k], grid[k], out[k], C, H, W, H, W); }, true);
  run("V6 KP=2 nt", [&](int k) { v6_kernel<2, 0><<<dim3((W + 127) / 128, (H + 3) / 4), 256>>>(src[k], grid[k], out[k], C, H, W, H, W); }, true);
  run("V7 KP=1 analytic grid", [&](int k) { v6_kernel<1, 1><<<dim3((W + 63) / 64, (H + 3) / 4), 256>>>(src[k], grid[k], out[k], C, H, W, H, W); }, false);
  run("V7 KP=2 analytic grid", [&](int k) { v6_kernel<2, 1><<<dim3((W + 127) / 128, (H + 3) / 4), 256>>>(src[k], grid[k], out[k], C, H, W, H, W); }, false);
  run("copy4 (dword lanes)", [&](int k) { copy4_kernel<<<(hw + 255) / 256, 256>>>(src[k], grid[k], out[k], hw); }, false);
  run("copy4 dependent index", [&](int k) { copydep_kernel<<<(hw + 255) / 256, 256>>>(src[k], grid[k], out[k], hw); }, false);
  run("V8 KP=1 scalar bases", [&](int k) { v8_kernel<1><<<dim3((W + 63) / 64, (H + 3) / 4), 256>>>(src[k], grid[k], out[k], C, H, W, H, W); }, true);
  run("V8 KP=2 scalar bases", [&](int k) { v8_kernel<2><<<dim3((W + 127) / 128, (H + 3) / 4), 256>>>(src[k], grid[k], out[k], C, H, W, H, W); }, true);
  run("V9 tile 32x2", [&](int k) { v9_kernel<32><<<dim3((W + 127) / 128, (H + 1) / 2), 256>>>(src[k], grid[k], out[k], C, H, W, H, W); }, true);
  run("V9 tile 16x4", [&](int k) { v9_kernel<16><<<dim3((W + 63) / 64, (H + 3) / 4), 256>>>(src[k], grid[k], out[k], C, H, W, H, W); }, true);
  run("V9 tile 8x8", [&](int k) { v9_kernel<8><<<dim3((W + 31) / 32, (H + 7) / 8), 256>>>(src[k], grid[k], out[k], C, H, W, H, W); }, true);
  run("V10 KR=2", [&](int k) { v10_kernel<2><<<dim3((W + 63) / 64, (H + 7) / 8), 256>>>(src[k], grid[k], out[k], C, H, W, H, W); }, true);
  run("V10 KR=4", [&](int k) { v10_kernel<4><<<dim3((W + 63) / 64, (H + 15) / 16), 256>>>(src[k], grid[k], out[k], C, H, W, H, W); }, true);
  return 0;
}
