// Bilinear warps of the DvD sampling path: the per-step feature warp / drop-in grid_sample
// (datasets/utils/warping.py:50-73) and the fused full-resolution unwarp tail
// (train_settings/dvd/evaluation.py:301-306 + utils_flow/visualization_utils.py:75-77).
//
// All of these are HBM/L2-bound gathers.  Fast kernels: one WAVE per run of an output row (lane-consecutive pixels,
// so grid reads, the taps of a smooth warp and the stores of every wave instruction are contiguous runs), the two
// x-neighbours of a tap row fetched as ONE 8-byte (f32) / 12-byte (u8 RGB pair) load, wave-uniform plane/row bases in
// SGPRs with 32-bit lane offsets, streaming (nt) grid loads and output stores.  Measured on MI355X at 3508x2480x3 f32
// (benchmarks/lab/warp_lab.hip): 4.4 TB/s on the 32 B/px drop-in contract against 5.6 TB/s for a plain copy of the same
// streams; the 1-pixel-per-lane-x-4-rows scalar kernels (kept below for win < 2, unaligned widths and > 4 GiB planes)
// reach 3.5 TB/s.
#include "common.h"
#include "mfma.h"

namespace dvd {

// Unnormalise with align_corners=True in ATen's order (aten/src/ATen/native/cpu/GridSamplerKernel.cpp, ComputeLocation:
// (g + 1) * scaling_factor, scaling_factor = (size - 1) / 2; two roundings).  Round 5: every arithmetic step of the warps
// follows the CPU kernels the reference runs (found by bit-exact probing of torch 2.10 CPU, tests/tools/aten_order_probe.py),
// so that the f32 results - and with them the truncated u8 bytes - are the reference's, not merely close to them.
__device__ __forceinline__ float unnorm(float g, int size) { return (g + 1.f) * ((float)(size - 1) * 0.5f); }

// The four-tap sum in ATen's order: nw * w_nw, then fused multiply-adds of ne, sw, se (GridSamplerKernel.cpp's
// `nw_val * nw + ne_val * ne + sw_val * sw + se_val * se`, whose additions the CPU build contracts into FMAs).
__device__ __forceinline__ float tap_sum(float v_nw, float w_nw, float v_ne, float w_ne, float v_sw, float w_sw, float v_se,
                                         float w_se) {
  return fmaf(v_se, w_se, fmaf(v_sw, w_sw, fmaf(v_ne, w_ne, __fmul_rn(v_nw, w_nw))));
}

// Branch-free bilinear taps (zeros padding): out-of-range taps get weight 0 and a clamped (always valid)
// address, so all four loads of a pixel issue unconditionally and back-to-back.  A NaN / infinite coordinate gives a
// NaN output pixel, as F.grid_sample does.
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
  if (!(fabsf(ix) < __builtin_inff()) || !(fabsf(iy) < __builtin_inff()))
    t.w00 = t.w01 = t.w10 = t.w11 = __builtin_nanf("");   // non-finite coordinate: NaN out, as ATen (inf - inf weights)
  const int xc0 = min(max(x0, 0), win - 1), xc1 = min(max(x0 + 1, 0), win - 1);
  const int yc0 = min(max(y0, 0), hin - 1), yc1 = min(max(y0 + 1, 0), hin - 1);
  t.o00 = yc0 * pitch + xc0;
  t.o01 = yc0 * pitch + xc1;
  t.o10 = yc1 * pitch + xc0;
  t.o11 = yc1 * pitch + xc1;
  return t;
}


// Pair taps: the two x-neighbours of a tap row are one load at column bx = clamp(x0, 0, win-2).  The weights are
// re-targeted onto (bx, bx+1) so an out-of-range tap contributes exactly 0 (zeros padding); in range they are the
// same products, blended in the same order, as the scalar kernels - results are bit-identical.
struct PTaps {
  int r0, r1;                  // element offsets (row * pitch + bx) of the upper / lower tap row, clamped in range
  float a00, a01, a10, a11;    // weights of (r0, bx), (r0, bx+1), (r1, bx), (r1, bx+1)
};

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
  const float cl = (x0 == bx) ? wx0 : ((x0 + 1 == bx) ? wx1 : 0.f);        // weight landing on column bx
  const float cr = (x0 == bx) ? wx1 : ((x0 == bx + 1) ? wx0 : 0.f);        // ... on column bx + 1
  const bool y0ok = y0 >= 0 && y0 < hin, y1ok = y0 + 1 >= 0 && y0 + 1 < hin;
  t.a00 = y0ok ? cl * wy0 : 0.f;
  t.a01 = y0ok ? cr * wy0 : 0.f;
  t.a10 = y1ok ? cl * wy1 : 0.f;
  t.a11 = y1ok ? cr * wy1 : 0.f;
  if (!(fabsf(ix) < __builtin_inff()) || !(fabsf(iy) < __builtin_inff()))
    t.a00 = t.a01 = t.a10 = t.a11 = __builtin_nanf("");   // non-finite coordinate: NaN out, as ATen (inf - inf weights)
  const int yc0 = min(max(y0, 0), hin - 1), yc1 = min(max(y0 + 1, 0), hin - 1);
  t.r0 = yc0 * pitch + bx;
  t.r1 = yc1 * pitch + bx;
  return t;
}

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x3 __attribute__((ext_vector_type(3)));
struct __attribute__((packed, aligned(4))) PackedF2 { float a, b; };
struct __attribute__((packed, aligned(4))) PackedF3 { float a, b, c; };
struct __attribute__((packed, aligned(4))) PackedU3 { uint32_t a, b, c; };

// 8-byte load at a 4-byte-aligned address: wave-uniform base + 32-bit lane byte offset (saddr + voffset form)
__device__ __forceinline__ f32x2 load_pair(const float* base, uint32_t byte_off) {
  const PackedF2 v = *reinterpret_cast<const PackedF2*>(reinterpret_cast<const char*>(base) + byte_off);
  return f32x2{v.a, v.b};
}
__device__ __forceinline__ float blend(f32x2 u, f32x2 d, const PTaps& t) {
  return tap_sum(u[0], t.a00, u[1], t.a01, d[0], t.a10, d[1], t.a11);
}

constexpr int KR = 4;    // rows per wave of the f32 fast kernels (vertically adjacent pixels per lane)
constexpr int CG = 3;    // channels whose gathers are in flight together in the drop-in kernel

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
            tap_sum(v[k][0], t[k].w00, v[k][1], t[k].w01, v[k][2], t[k].w10, v[k][3], t[k].w11);
  }
}


// Fast path (win >= 2, plane < 4 GiB).  Wave = 64 consecutive x  x  KR consecutive rows (a lane owns KR vertically
// adjacent pixels, whose tap rows overlap in L1); block = 4 waves stacked in y.
__global__ void __launch_bounds__(256) grid_sample_rows_kernel(const float* __restrict__ src,
                                                               const float* __restrict__ grid,
                                                               float* __restrict__ out, int c, int hin, int win,
                                                               int h, int w, int src_batch_div) {
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int n = blockIdx.z;
  const int yb = (blockIdx.y * 4 + wv) * KR;            // wave-uniform
  if (yb >= h) return;
  const bool live = blockIdx.x * 64u + lane < (uint32_t)w;
  const uint32_t xb = min(blockIdx.x * 64u + lane, (uint32_t)w - 1u) * 4u;
  const size_t hw = (size_t)h * w;
  const size_t plane = (size_t)hin * win;
  const float* g = grid + (size_t)n * 2 * hw;
  PTaps t[KR];
#pragma unroll
  for (int k = 0; k < KR; ++k) {
    const float* grow = g + (size_t)min(yb + k, h - 1) * w;
    const float gx = __builtin_nontemporal_load(reinterpret_cast<const float*>(reinterpret_cast<const char*>(grow) + xb));
    const float gy = __builtin_nontemporal_load(reinterpret_cast<const float*>(reinterpret_cast<const char*>(grow + hw) + xb));
    t[k] = make_ptaps(gx, gy, hin, win, win);
  }
  const float* s = src + (size_t)(n / src_batch_div) * c * plane;
  float* o = out + (size_t)n * c * hw;
  for (int ch0 = 0; ch0 < c; ch0 += CG) {
    f32x2 u[CG][KR], d[CG][KR];
#pragma unroll
    for (int j = 0; j < CG; ++j) {
      const float* pc = s + (size_t)min(ch0 + j, c - 1) * plane;     // uniform
#pragma unroll
      for (int k = 0; k < KR; ++k) {
        u[j][k] = load_pair(pc, (uint32_t)t[k].r0 * 4u);
        d[j][k] = load_pair(pc, (uint32_t)t[k].r1 * 4u);
      }
    }
#pragma unroll
    for (int j = 0; j < CG; ++j) {
      if (ch0 + j >= c) break;
#pragma unroll
      for (int k = 0; k < KR; ++k)
        if (live && yb + k < h)
          __builtin_nontemporal_store(
              blend(u[j][k], d[j][k], t[k]),
              reinterpret_cast<float*>(reinterpret_cast<char*>(o + (size_t)(ch0 + j) * hw + (size_t)(yb + k) * w) + xb));
    }
  }
}

// ---------------------------------------------------------------------------------------
// Drop-in grid_sample through LDS-staged source tiles (win % 4 == 0, planes < 4 GiB).
//
// The direct gathers above issue, per wave instruction, 64 eight-byte loads whose addresses follow the warp: under
// shear they fall into up to ~45 different source rows, i.e. ~45 L1 line accesses for 512 useful bytes - the address
// path, not HBM, bounds them (46-50 % of the 8 TB/s figure).  Here a workgroup owns a 32 x 32 OUTPUT tile:
//   1. every thread reads the grid for its 4 consecutive pixels (one 16-byte streaming load per plane) and builds
//      their pair taps;
//   2. the tile's source footprint - the bounding box of all tap rows / columns - is reduced over the workgroup;
//   3. the box is copied into LDS with fully coalesced 16-byte loads along source rows (x origin aligned down to 4 px);
//   4. the taps are gathered from LDS (two adjacent dwords per tap row) and the outputs streamed out.
// A tile whose box exceeds the LDS budget (2048 px per plane: ~8 % of the tiles of the bench's flow, whose local shear
// reaches 35 degrees; any wildly non-smooth grid) takes the direct gather for that tile; arithmetic (make_ptaps / blend)
// is shared with the row kernels, so all paths give the same bits.
// Measured on MI355X (8 documents of 3508 x 2480 per launch, the bench's flow; benchmarks/warp_time.py): 4.2-4.6 TB/s
// against 3.8 for the row kernel and 5.1 for a device copy of the same 2.2 GB - and the same with an identity flow: the
// gather no longer costs anything, what remains is the streaming rate of eight interleaved planes.  What mattered, in
// order: 16 bytes per lane on every streaming access (the first version read the grid and wrote the output 4 bytes per
// lane like the row kernel and ran at 3.0-3.7 TB/s whatever the tile shape), workgroups per CU (cap 2048 / 6 per CU
// beats cap 3072 / 4 per CU by 3 %), square tiles (64 x 16: -8 %, 128 x 8: -25 %: larger footprints, fewer workgroups);
// a persistent variant that prefetched the next tile's grid values was slower than one tile per workgroup.
// Workgroup ids are mapped to tiles in XCD BANDS: id % 8 is the XCD the hardware places a workgroup on, and each XCD
// walks a contiguous row-major range of tiles, so the halo rows / columns neighbouring tiles share are re-read from
// that XCD's own L2 instead of being fetched over the fabric by eight different L2s.
// ---------------------------------------------------------------------------------------
constexpr int LTW = 32, LTH = 32;   // output tile: 32 x 32 pixels = 256 threads x 4 CONSECUTIVE x (one 16-byte grid load
                                    // per plane and one 16-byte store per channel per thread: every streaming access of
                                    // the kernel is 16 bytes per lane, 8 lanes per 128-byte row segment)
constexpr int LCG = 3;              // planes staged together
constexpr int LCAP_P = 2048;        // floats per plane held in LDS (rows x 4-px-aligned width of the footprint, compact):
                                    // 3 * 2048 * 4 B = 24 KiB, 72 VGPRs  ->  6 workgroups (24 waves) per CU

struct PTapsB {
  int bx, yc0, yc1;               // left tap column (pair base), upper / lower tap row - all clamped in range
  float a00, a01, a10, a11;
};

__device__ __forceinline__ PTapsB make_ptaps_box(float gx, float gy, int hin, int win) {
  // identical arithmetic to make_ptaps (the weights must be the same bits); returns the coordinates un-flattened
  PTapsB t;
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
  const float cl = (x0 == bx) ? wx0 : ((x0 + 1 == bx) ? wx1 : 0.f);
  const float cr = (x0 == bx) ? wx1 : ((x0 == bx + 1) ? wx0 : 0.f);
  const bool y0ok = y0 >= 0 && y0 < hin, y1ok = y0 + 1 >= 0 && y0 + 1 < hin;
  t.a00 = y0ok ? cl * wy0 : 0.f;
  t.a01 = y0ok ? cr * wy0 : 0.f;
  t.a10 = y1ok ? cl * wy1 : 0.f;
  t.a11 = y1ok ? cr * wy1 : 0.f;
  if (!(fabsf(ix) < __builtin_inff()) || !(fabsf(iy) < __builtin_inff()))
    t.a00 = t.a01 = t.a10 = t.a11 = __builtin_nanf("");
  t.bx = bx;
  t.yc0 = min(max(y0, 0), hin - 1);
  t.yc1 = min(max(y0 + 1, 0), hin - 1);
  return t;
}

__device__ __forceinline__ float blend_b(f32x2 u, f32x2 d, const PTapsB& t) {
  return tap_sum(u[0], t.a00, u[1], t.a01, d[0], t.a10, d[1], t.a11);
}

__device__ __forceinline__ int wave_min(int v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = min(v, __shfl_xor(v, o, 64));
  return v;
}
__device__ __forceinline__ int wave_max(int v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = max(v, __shfl_xor(v, o, 64));
  return v;
}

__device__ __forceinline__ float4 load_f4(const float* base, uint32_t byte_off) {
  // 16-byte load: wave-uniform base (SGPR pair) + 32-bit lane byte offset
  return *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(base) + byte_off);
}

__device__ __forceinline__ float load_f1_nt(const float* base, uint32_t byte_off) {
  return __builtin_nontemporal_load(reinterpret_cast<const float*>(reinterpret_cast<const char*>(base) + byte_off));
}
__device__ __forceinline__ void store_f1_nt(float* base, uint32_t byte_off, float v) {
  __builtin_nontemporal_store(v, reinterpret_cast<float*>(reinterpret_cast<char*>(base) + byte_off));
}

__device__ __forceinline__ float4 load_f4_nt(const float* base, uint32_t byte_off) {
  typedef float v4 __attribute__((ext_vector_type(4)));
  const v4 v = __builtin_nontemporal_load(reinterpret_cast<const v4*>(reinterpret_cast<const char*>(base) + byte_off));
  return make_float4(v[0], v[1], v[2], v[3]);
}
__device__ __forceinline__ void store_f4_nt(float* base, uint32_t byte_off, float4 q) {
  typedef float v4 __attribute__((ext_vector_type(4)));
  __builtin_nontemporal_store(v4{q.x, q.y, q.z, q.w}, reinterpret_cast<v4*>(reinterpret_cast<char*>(base) + byte_off));
}

// every global access below is "wave-uniform 64-bit base (SGPR pair) + 32-bit lane byte offset", 16 bytes per lane
template <int TW, int LCAP, int NT, int DMA = 0>
__global__ void __launch_bounds__(256) grid_sample_lds_kernel(
    const float* __restrict__ src, const float* __restrict__ grid, float* __restrict__ out, int c, int hin, int win, int h,
    int w, int src_batch_div, int ntx, int nty, unsigned tiles_total, unsigned tiles_per_xcd) {
  __shared__ __attribute__((aligned(16))) float box[LCG * LCAP];
  __shared__ int red[4][4];
  // ---- tile of this workgroup (XCD bands) ----
  const unsigned seq = blockIdx.x >> 3, xcd = blockIdx.x & 7u;
  const unsigned tile = xcd * tiles_per_xcd + seq;
  if (seq >= tiles_per_xcd || tile >= tiles_total) return;                  // workgroup-uniform
  const unsigned per_img = (unsigned)ntx * (unsigned)nty;
  const int n = (int)(tile / per_img);
  const unsigned rem = tile - (unsigned)n * per_img;
  const int ty = (int)(rem / (unsigned)ntx), tx = (int)(rem - (unsigned)ty * (unsigned)ntx);
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  constexpr int TPR = TW / 4, TH = 256 / TPR;                               // threads per tile row, tile rows
  const int x = tx * TW + 4 * (tid % TPR), y = ty * TH + (tid / TPR);       // 4 consecutive pixels of one row
  const size_t hw = (size_t)h * w;
  const size_t plane = (size_t)hin * win;
  const float* g = grid + (size_t)n * 2 * hw;                               // uniform
  // byte offset of this thread's 4 pixels inside one [h, w] plane (grid and output share it); a dead thread (beyond the
  // right / bottom edge: w % 4 == 0, so a thread is live or dead as a whole) is clamped onto live pixels for the loads
  const bool live = x < w && y < h;
  const uint32_t poff = ((uint32_t)min(y, h - 1) * (uint32_t)w + (uint32_t)min(x, w - 4)) * 4u;
  const float4 gx4 = NT ? load_f4_nt(g, poff) : load_f4(g, poff), gy4 = NT ? load_f4_nt(g + hw, poff) : load_f4(g + hw, poff);
  const float gxs[4] = {gx4.x, gx4.y, gx4.z, gx4.w}, gys[4] = {gy4.x, gy4.y, gy4.z, gy4.w};
  PTapsB t[4];
  int xmin = win, xmax = 0, ymin = hin, ymax = 0;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    t[k] = make_ptaps_box(gxs[k], gys[k], hin, win);
    xmin = min(xmin, t[k].bx); xmax = max(xmax, t[k].bx + 1);
    ymin = min(ymin, t[k].yc0); ymax = max(ymax, t[k].yc1);
  }
  xmin = wave_min(xmin); xmax = wave_max(xmax); ymin = wave_min(ymin); ymax = wave_max(ymax);
  if (lane == 0) { red[wv][0] = xmin; red[wv][1] = xmax; red[wv][2] = ymin; red[wv][3] = ymax; }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    xmin = min(xmin, red[i][0]); xmax = max(xmax, red[i][1]);
    ymin = min(ymin, red[i][2]); ymax = max(ymax, red[i][3]);
  }
  // wave-uniform from here on (SGPRs): the box geometry and every plane base derived from it
  xmin = __builtin_amdgcn_readfirstlane(xmin); xmax = __builtin_amdgcn_readfirstlane(xmax);
  ymin = __builtin_amdgcn_readfirstlane(ymin); ymax = __builtin_amdgcn_readfirstlane(ymax);
  const int x0a = xmin & ~3;                           // 16-byte aligned box origin (win % 4 == 0)
  const int bw4 = ((xmax - x0a) >> 2) + 1;             // float4 per box row
  const int bh = ymax - ymin + 1;
  const int bw = bw4 * 4;                              // LDS row pitch (floats): the box is stored compactly
  const int nvec = bh * bw4;
  const bool staged = nvec * 4 <= LCAP && bw4 <= 64;   // workgroup-uniform
  const float* s = src + (size_t)(n / src_batch_div) * c * plane;
  float* o = out + (size_t)n * c * hw;

  if (staged) {
    // tap positions inside the LDS image (floats), and the copy pattern: thread i moves float4 number i, i + 256, ... of
    // each plane's [bh][bw4] image (row = i / bw4 by an exact reciprocal multiply: i < 1024, bw4 <= 64)
    constexpr int ITER = (LCAP / 4 + 255) / 256;
    int l0[4], l1[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      l0[k] = (t[k].yc0 - ymin) * bw + (t[k].bx - x0a);
      l1[k] = (t[k].yc1 - ymin) * bw + (t[k].bx - x0a);
    }
    const uint32_t inv = (65536u + (uint32_t)bw4 - 1u) / (uint32_t)bw4;      // uniform
    uint32_t voff[ITER];
    bool vok[ITER];
#pragma unroll
    for (int it = 0; it < ITER; ++it) {
      const uint32_t idx = (uint32_t)tid + 256u * it;
      const uint32_t row = (idx * inv) >> 16, c4 = idx - row * (uint32_t)bw4;
      vok[it] = (int)idx < nvec;
      voff[it] = vok[it] ? (row * (uint32_t)win + 4u * c4) * 4u : 0u;
    }
    const float* s0 = s + (size_t)ymin * win + x0a;                           // uniform
    for (int ch0 = 0; ch0 < c; ch0 += LCG) {
      if constexpr (DMA) {
        // lab (round 5): the box goes global -> LDS directly (global_load_lds_dwordx4: lane-linear 1-KiB chunks = exactly the
        // compact [bh][bw4] float4 image): no staging registers, no ds_write.  A lane beyond the box writes its (unused) slot.
        if (ch0) __syncthreads();                      // the previous channel group's taps have been read
        typedef __attribute__((address_space(3))) void* lptr_t;
        const unsigned lds0 = (unsigned)(uintptr_t)(lptr_t)box;
#pragma unroll
        for (int j = 0; j < LCG; ++j) {
          const float* pc = s0 + (size_t)min(ch0 + j, c - 1) * plane;         // uniform
#pragma unroll
          for (int it = 0; it < ITER; ++it) {
            const unsigned dst = __builtin_amdgcn_readfirstlane(lds0 + (unsigned)(j * LCAP * 4 + (wv * 64 + 256 * it) * 16));
            asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %0" ::"s"(pc), "s"(dst), "v"(voff[it]) : "memory");
          }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
      } else {
      float4 v[LCG][ITER];
#pragma unroll
      for (int j = 0; j < LCG; ++j) {
        const float* pc = s0 + (size_t)min(ch0 + j, c - 1) * plane;           // uniform
#pragma unroll
        for (int it = 0; it < ITER; ++it) v[j][it] = load_f4(pc, voff[it]);
      }
      if (ch0) __syncthreads();                        // the previous channel group's taps have been read
#pragma unroll
      for (int j = 0; j < LCG; ++j)
#pragma unroll
        for (int it = 0; it < ITER; ++it)
          if (vok[it]) *reinterpret_cast<float4*>(&box[j * LCAP + (tid + 256 * it) * 4]) = v[j][it];
      __syncthreads();
      }
#pragma unroll
      for (int j = 0; j < LCG; ++j) {
        if (ch0 + j >= c) break;
        float r[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const float* b0 = &box[j * LCAP + l0[k]];
          const float* b1 = &box[j * LCAP + l1[k]];
          const f32x2 u{b0[0], b0[1]}, d{b1[0], b1[1]};
          r[k] = blend_b(u, d, t[k]);
        }
        if (live) {
          float* oc = o + (size_t)(ch0 + j) * hw;                             // uniform
          if (NT) store_f4_nt(oc, poff, make_float4(r[0], r[1], r[2], r[3]));
          else *reinterpret_cast<float4*>(reinterpret_cast<char*>(oc) + poff) = make_float4(r[0], r[1], r[2], r[3]);
        }
      }
    }
  } else {
    // ---- footprint too large for LDS: direct gather for this tile ----
    uint32_t g0[4], g1[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      g0[k] = ((uint32_t)t[k].yc0 * (uint32_t)win + (uint32_t)t[k].bx) * 4u;
      g1[k] = ((uint32_t)t[k].yc1 * (uint32_t)win + (uint32_t)t[k].bx) * 4u;
    }
    for (int ch = 0; ch < c; ++ch) {
      const float* pc = s + (size_t)ch * plane;                               // uniform
      float r[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) r[k] = blend_b(load_pair(pc, g0[k]), load_pair(pc, g1[k]), t[k]);
      if (live) store_f4_nt(o + (size_t)ch * hw, poff, make_float4(r[0], r[1], r[2], r[3]));
    }
  }
}

// ---------------------------------------------------------------------------------------
// Fused unwarp tail.  grid(i,j) is computed on the fly from the coarse flow (L2-resident:
// 2*G*G floats), so HBM traffic is src taps + output only.
// ---------------------------------------------------------------------------------------
struct UpParams {
  int g, h, w;
  float sy, sx;     // (g-1)/(h-1), (g-1)/(w-1): ATen area_pixel_compute_scale, align_corners=True
  float sby, sbx;   // 511/(h-1), 511/(w-1): the same for the 512 x 512 base grid (train_settings/dvd/evaluation.py:304)
  float scale;
  int small;        // h + w <= 128: torch's CPU dispatch takes its channels-last kernel, whose sum has another order (below)
};

// One axis of F.interpolate(bilinear, align_corners=True) as ATen's CPU kernel computes it (UpSampleKernel.cpp,
// compute_source_index_and_lambda): src = scale * dst; i0 = min(floor(src), in - 1); l1 = clamp(src - i0, 0, 1); l0 = 1 - l1.
__device__ __forceinline__ void interp_axis(float scale, int dst, int in, int& i0, int& i1, float& l0, float& l1) {
  const float s = __fmul_rn(scale, (float)dst);
  i0 = min((int)s, in - 1);
  i1 = min(i0 + 1, in - 1);
  l1 = fminf(fmaxf(__fsub_rn(s, (float)i0), 0.f), 1.f);
  l0 = __fsub_rn(1.f, l1);
}
// ... and its four-value sum.  torch picks one of two CPU kernels by the OUTPUT size (UpSampleKernel.cpp,
// _use_vectorized_kernel_cond_2d; both orders found by bit-exact probing, oracle/aten_order.py restates them):
//   h + w > 128   separable: t = fma(left, lx0, right * lx1) per row, then fma(t_upper, ly0, t_lower * ly1)
//   h + w <= 128  channels-last kernel: weights w_rc = l_y(r) * l_x(c), sum fma(d, w11, fma(c, w10, fma(a, w00, b * w01)))
__device__ __forceinline__ float interp_sum(float a, float b, float c, float d, float lx0, float lx1, float ly0, float ly1,
                                            int small) {
  if (small) {
    const float w00 = __fmul_rn(ly0, lx0), w01 = __fmul_rn(ly0, lx1), w10 = __fmul_rn(ly1, lx0), w11 = __fmul_rn(ly1, lx1);
    return fmaf(d, w11, fmaf(c, w10, fmaf(a, w00, __fmul_rn(b, w01))));
  }
  const float t0 = fmaf(a, lx0, __fmul_rn(b, lx1)), t1 = fmaf(c, lx0, __fmul_rn(d, lx1));
  return fmaf(t0, ly0, __fmul_rn(t1, ly1));
}

// The per-AXIS terms of one output coordinate: the flow's and the 512-grid base's source indices and lambdas, and the base
// grid's two values k / 511 (an IEEE division each).  They depend on the column (or the row) alone, so a kernel that walks
// down a column block computes its x terms ONCE (unwarp_u8_band_kernel); flow_grid_at composes them per pixel.
struct AxisTerm {
  int f0, f1;            // flow rows / columns
  float l0, l1;          // ... and their lambdas
  float bl0, bl1;        // lambdas of the 512-grid base
  float c0, c1;          // its two values: index / 511
};
__device__ __forceinline__ AxisTerm axis_term(float scale, float bscale, int dst, int g) {
  AxisTerm t;
  interp_axis(scale, dst, g, t.f0, t.f1, t.l0, t.l1);
  int b0, b1;
  interp_axis(bscale, dst, 512, b0, b1, t.bl0, t.bl1);
  t.c0 = __fdiv_rn((float)b0, 511.f);
  t.c1 = __fdiv_rn((float)b1, 511.f);
  return t;
}
__device__ __forceinline__ void flow_grid_from(const float* __restrict__ flow, const UpParams& p, const AxisTerm& y,
                                               const AxisTerm& x, float& gx, float& gy) {
  // sample = F.interpolate(flow, (H, W), bilinear, align_corners=True)                       (evaluation.py:301)
  const int gg = p.g * p.g;
  float v[2];
#pragma unroll
  for (int ch = 0; ch < 2; ++ch) {
    const float* f = flow + ch * gg;
    v[ch] = interp_sum(f[y.f0 * p.g + x.f0], f[y.f0 * p.g + x.f1], f[y.f1 * p.g + x.f0], f[y.f1 * p.g + x.f1], x.l0, x.l1, y.l0,
                       y.l1, p.small);
  }
  // base = F.interpolate(coords_grid_tensor((512, 512)) / 511., (H, W), bilinear, align_corners=True)   (evaluation.py:304):
  // channel 0 holds column / 511, channel 1 row / 511 of the 512 x 512 grid - interpolated like any other image, so the
  // value is NOT j / (W - 1) to the last bit
  const float bx = interp_sum(x.c0, x.c1, x.c0, x.c1, x.bl0, x.bl1, y.bl0, y.bl1, p.small);
  const float by = interp_sum(y.c0, y.c0, y.c1, y.c1, x.bl0, x.bl1, y.bl0, y.bl1, p.small);
  // sample = (((sample + base) * 1) * 2 - 1) * 0.987                                         (evaluation.py:306)
  gx = __fmul_rn(__fsub_rn(__fmul_rn(__fmul_rn(__fadd_rn(v[0], bx), 1.f), 2.f), 1.f), p.scale);
  gy = __fmul_rn(__fsub_rn(__fmul_rn(__fmul_rn(__fadd_rn(v[1], by), 1.f), 2.f), 1.f), p.scale);
}
__device__ __forceinline__ void flow_grid_at(const float* __restrict__ flow, const UpParams& p, int i, int j,
                                             float& gx, float& gy) {
  flow_grid_from(flow, p, axis_term(p.sy, p.sby, i, p.g), axis_term(p.sx, p.sbx, j, p.g), gx, gy);
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
  flow += (size_t)blockIdx.z * 2 * p.g * p.g;        // one document per grid z (batched launch)
  src += (size_t)blockIdx.z * 3 * p.h * p.w;
  out += (size_t)blockIdx.z * 3 * p.h * p.w;
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
      o[ch] = tap_sum(v[ch][k][0], t[k].w00, v[ch][k][1], t[k].w01, v[ch][k][2], t[k].w10, v[ch][k][3], t[k].w11);
  }
}

__global__ void __launch_bounds__(256) unwarp_u8_kernel(const float* __restrict__ flow,
                                                        const uint8_t* __restrict__ src, uint8_t* __restrict__ out,
                                                        UpParams p) {
  flow += (size_t)blockIdx.z * 2 * p.g * p.g;        // one document per grid z (batched launch)
  src += (size_t)blockIdx.z * 3 * p.h * p.w;
  out += (size_t)blockIdx.z * 3 * p.h * p.w;
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
      const float a = tap_sum((float)v[k][0][ch], t[k].w00, (float)v[k][1][ch], t[k].w01, (float)v[k][2][ch], t[k].w10,
                              (float)v[k][3][ch], t[k].w11);
      o[ch] = (uint8_t)(int)a;   // truncation, as numpy .astype(uint8) for 0 <= a < 256
    }
  }
}


// Fast fused f32 tail (w >= 2, plane < 4 GiB): same wave shape; the output is HWC, so a lane's three channels are one
// 12-byte store and a wave instruction writes 768 contiguous bytes.
__global__ void __launch_bounds__(256) unwarp_f32_rows_kernel(const float* __restrict__ flow,
                                                              const float* __restrict__ src,
                                                              float* __restrict__ out, UpParams p) {
  flow += (size_t)blockIdx.z * 2 * p.g * p.g;        // one document per grid z (batched launch)
  src += (size_t)blockIdx.z * 3 * p.h * p.w;
  out += (size_t)blockIdx.z * 3 * p.h * p.w;
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int ib = (blockIdx.y * 4 + wv) * KR;
  if (ib >= p.h) return;
  const bool live = blockIdx.x * 64u + lane < (uint32_t)p.w;
  const uint32_t j = min(blockIdx.x * 64u + lane, (uint32_t)p.w - 1u);
  PTaps t[KR];
#pragma unroll
  for (int k = 0; k < KR; ++k) {
    float gx, gy;
    flow_grid_at(flow, p, min(ib + k, p.h - 1), (int)j, gx, gy);
    t[k] = make_ptaps(gx, gy, p.h, p.w, p.w);
  }
  const size_t plane = (size_t)p.h * p.w;
  f32x2 u[3][KR], d[3][KR];
#pragma unroll
  for (int ch = 0; ch < 3; ++ch)
#pragma unroll
    for (int k = 0; k < KR; ++k) {
      u[ch][k] = load_pair(src + ch * plane, (uint32_t)t[k].r0 * 4u);
      d[ch][k] = load_pair(src + ch * plane, (uint32_t)t[k].r1 * 4u);
    }
#pragma unroll
  for (int k = 0; k < KR; ++k)
    if (live && ib + k < p.h) {
      PackedF3 o{blend(u[0][k], d[0][k], t[k]), blend(u[1][k], d[1][k], t[k]), blend(u[2][k], d[2][k], t[k])};
      *reinterpret_cast<PackedF3*>(reinterpret_cast<char*>(out + (size_t)(ib + k) * p.w * 3) + j * 12u) = o;
    }
}

// Fast fused u8 tail (w % 4 == 0, image < 4 GiB): a lane owns 4 consecutive pixels of a row.  The RGB pair of a tap
// row is 6 consecutive bytes at byte offset 3*(row*w + bx): one 12-byte load at the 4-byte-aligned address below it
// (moved back inside the image at its very end) and two v_alignbyte; the lane's 12 output bytes are one 12-byte store.
__device__ __forceinline__ void load_rgb_pair(const uint8_t* __restrict__ src, uint32_t off, uint32_t last_base,
                                              uint32_t& lo, uint32_t& hi) {
  // bytes [off, off + 6): lo = bytes 0..3, hi = bytes 4..5 (upper half unspecified)
  const uint32_t base = min(off & ~3u, last_base);
  uint32_t sh = off - base;                                 // 0..6
  const PackedU3 v = *reinterpret_cast<const PackedU3*>(src + base);
  uint32_t d0 = v.a, d1 = v.b, d2 = v.c;
  if (sh >= 4u) { d0 = d1; d1 = d2; d2 = 0u; sh -= 4u; }
  lo = __builtin_amdgcn_alignbyte(d1, d0, sh);
  hi = __builtin_amdgcn_alignbyte(d2, d1, sh);
}

__global__ void __launch_bounds__(256) unwarp_u8_rows_kernel(const float* __restrict__ flow,
                                                             const uint8_t* __restrict__ src,
                                                             uint8_t* __restrict__ out, UpParams p) {
  flow += (size_t)blockIdx.z * 2 * p.g * p.g;        // one document per grid z (batched launch)
  src += (size_t)blockIdx.z * 3 * p.h * p.w;
  out += (size_t)blockIdx.z * 3 * p.h * p.w;
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int i = blockIdx.y * 4 + wv;
  if (i >= p.h) return;
  const uint32_t j0 = (blockIdx.x * 64 + lane) * 4;
  if (j0 >= (uint32_t)p.w) return;
  const uint32_t last_base = (uint32_t)p.h * (uint32_t)p.w * 3u - 12u;
  PTaps t[4];
  uint32_t ulo[4], uhi[4], dlo[4], dhi[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    float gx, gy;
    flow_grid_at(flow, p, i, (int)(j0 + k), gx, gy);
    t[k] = make_ptaps(gx, gy, p.h, p.w, p.w);
  }
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    load_rgb_pair(src, (uint32_t)t[k].r0 * 3u, last_base, ulo[k], uhi[k]);
    load_rgb_pair(src, (uint32_t)t[k].r1 * 3u, last_base, dlo[k], dhi[k]);
  }
  uint32_t ob[12];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    // left pixel = bytes 0,1,2; right pixel = bytes 3 (lo), 4,5 (hi)
    const float ul[3] = {(float)(ulo[k] & 255u), (float)((ulo[k] >> 8) & 255u), (float)((ulo[k] >> 16) & 255u)};
    const float ur[3] = {(float)(ulo[k] >> 24), (float)(uhi[k] & 255u), (float)((uhi[k] >> 8) & 255u)};
    const float dl[3] = {(float)(dlo[k] & 255u), (float)((dlo[k] >> 8) & 255u), (float)((dlo[k] >> 16) & 255u)};
    const float dr[3] = {(float)(dlo[k] >> 24), (float)(dhi[k] & 255u), (float)((dhi[k] >> 8) & 255u)};
#pragma unroll
    for (int ch = 0; ch < 3; ++ch) {
      const float a = tap_sum(ul[ch], t[k].a00, ur[ch], t[k].a01, dl[ch], t[k].a10, dr[ch], t[k].a11);
      ob[k * 3 + ch] = (uint32_t)(int)a & 255u;             // truncation, as numpy .astype(uint8) for 0 <= a < 256
    }
  }
  PackedU3 o;
  o.a = ob[0] | (ob[1] << 8) | (ob[2] << 16) | (ob[3] << 24);
  o.b = ob[4] | (ob[5] << 8) | (ob[6] << 16) | (ob[7] << 24);
  o.c = ob[8] | (ob[9] << 8) | (ob[10] << 16) | (ob[11] << 24);
  *reinterpret_cast<PackedU3*>(out + ((size_t)i * p.w + j0) * 3) = o;
}

#ifdef DVD_LAB
// Round 6, MEASURED AND REJECTED - lab only (DVD_WARP_U8_UB = 2 | 4 | 8 | 16, DVD_WARP_U8_UNROLL): the same
// tail walking DOWN a column block.  unwarp_u8_rows_kernel spends ~150 VALU instructions per pixel for 6 bytes of traffic, and
// a third of them are the per-COLUMN terms of the up-sampling (two source-index / lambda computations and two IEEE divisions by
// 511 per pixel) that every row recomputes; here a wave keeps its 256 columns (4 per lane) and loops over UBT rows - the
// column terms computed once, the row terms once per row and lane.  Same operations in the same order on the same values
// (flow_grid_from): the same bytes (tests/test_gpu_ops.py).  And SLOWER the taller the band (8 x 3508 x 2480, MI355X,
// profiles/r6_u8_band_variants.txt): rows 0.441 ms, UBT 2: 0.496, 4: 0.543, 8: 0.628, 16: 0.755, unrolled or not.  So the
// instruction count is NOT what bounds the row kernel (round 5 called it VALU-bound): a pixel is one dependent chain - coarse-flow
// loads -> taps -> source gather -> blend -> store - and the kernel runs at the rate its resident waves overlap those chains;
// a wave that walks 8 rows holds its registers 8 chains long and there are 8x fewer waves to overlap.
constexpr int UB = 8;
template <int UBT, bool UNROLL>
__global__ void __launch_bounds__(256) unwarp_u8_band_kernel(const float* __restrict__ flow, const uint8_t* __restrict__ src,
                                                             uint8_t* __restrict__ out, UpParams p) {
  flow += (size_t)blockIdx.z * 2 * p.g * p.g;        // one document per grid z (batched launch)
  src += (size_t)blockIdx.z * 3 * p.h * p.w;
  out += (size_t)blockIdx.z * 3 * p.h * p.w;
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int i0 = (blockIdx.y * 4 + wv) * UBT;
  if (i0 >= p.h) return;
  const uint32_t j0 = (blockIdx.x * 64 + lane) * 4;
  if (j0 >= (uint32_t)p.w) return;
  const uint32_t last_base = (uint32_t)p.h * (uint32_t)p.w * 3u - 12u;
  AxisTerm xt[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) xt[k] = axis_term(p.sx, p.sbx, (int)(j0 + k), p.g);
  const int i1 = min(i0 + UBT, p.h);
#pragma unroll(UNROLL ? UBT : 1)
  for (int i = i0; i < i1; ++i) {
    const AxisTerm yt = axis_term(p.sy, p.sby, i, p.g);
    PTaps t[4];
    uint32_t ulo[4], uhi[4], dlo[4], dhi[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      float gx, gy;
      flow_grid_from(flow, p, yt, xt[k], gx, gy);
      t[k] = make_ptaps(gx, gy, p.h, p.w, p.w);
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      load_rgb_pair(src, (uint32_t)t[k].r0 * 3u, last_base, ulo[k], uhi[k]);
      load_rgb_pair(src, (uint32_t)t[k].r1 * 3u, last_base, dlo[k], dhi[k]);
    }
    uint32_t ob[12];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float ul[3] = {(float)(ulo[k] & 255u), (float)((ulo[k] >> 8) & 255u), (float)((ulo[k] >> 16) & 255u)};
      const float ur[3] = {(float)(ulo[k] >> 24), (float)(uhi[k] & 255u), (float)((uhi[k] >> 8) & 255u)};
      const float dl[3] = {(float)(dlo[k] & 255u), (float)((dlo[k] >> 8) & 255u), (float)((dlo[k] >> 16) & 255u)};
      const float dr[3] = {(float)(dlo[k] >> 24), (float)(dhi[k] & 255u), (float)((dhi[k] >> 8) & 255u)};
#pragma unroll
      for (int ch = 0; ch < 3; ++ch) {
        const float a = tap_sum(ul[ch], t[k].a00, ur[ch], t[k].a01, dl[ch], t[k].a10, dr[ch], t[k].a11);
        ob[k * 3 + ch] = (uint32_t)(int)a & 255u;           // truncation, as numpy .astype(uint8) for 0 <= a < 256
      }
    }
    PackedU3 o;
    o.a = ob[0] | (ob[1] << 8) | (ob[2] << 16) | (ob[3] << 24);
    o.b = ob[4] | (ob[5] << 8) | (ob[6] << 16) | (ob[7] << 24);
    o.c = ob[8] | (ob[9] << 8) | (ob[10] << 16) | (ob[11] << 24);
    *reinterpret_cast<PackedU3*>(out + ((size_t)i * p.w + j0) * 3) = o;
  }
}
#endif  // DVD_LAB

static UpParams make_up(int g, int h, int w, float scale) {
  UpParams p;
  p.g = g;
  p.h = h;
  p.w = w;
  p.sy = h > 1 ? (float)(g - 1) / (float)(h - 1) : 0.f;
  p.sx = w > 1 ? (float)(g - 1) / (float)(w - 1) : 0.f;
  p.sby = h > 1 ? 511.f / (float)(h - 1) : 0.f;
  p.sbx = w > 1 ? 511.f / (float)(w - 1) : 0.f;
  p.scale = scale;
  p.small = (h + w <= 128) ? 1 : 0;
  return p;
}

}  // namespace dvd

using namespace dvd;

// The 1-pixel-per-lane kernels are the fallback for shapes the row kernels do not take (win < 2, w % 4 != 0 for u8,
// planes >= 4 GiB).  Lab build only: DVD_WARP_SCALAR=1 forces them (A/B runs, fast-vs-fallback bit-identity checks).
static bool scalar_warp() {
#ifdef DVD_LAB
  const char* e = getenv("DVD_WARP_SCALAR");
  return e && e[0] == '1';
#else
  return false;
#endif
}
static int lds_tile_variant() {     // lab only: 1 = 32 x 32 tiles (4 px per thread, 36 KiB of LDS: 4 workgroups per CU), ...
#ifdef DVD_LAB
  const char* e = getenv("DVD_WARP_LDSVAR");
  return e ? atoi(e) : 0;
#else
  return 0;
#endif
}
// drop-in grid_sample: 0 = shape-chosen kernel (product), 1 = skip the LDS-tile kernel (row kernel), 2 = scalar kernel
static int warp_variant() {
#ifdef DVD_LAB
  if (scalar_warp()) return 2;
  const char* e = getenv("DVD_WARP_NOLDS");
  return (e && e[0] == '1') ? 1 : 0;
#else
  return 0;
#endif
}

extern "C" int dvd_grid_sample_bilinear_zeros_ac(const float* src, const float* grid, float* out, int n, int c,
                                                 int hin, int win, int h, int w, int src_batch_div, void* stream) {
  DVD_REQUIRE(src && grid && out, "grid_sample: null pointer");
  DVD_REQUIRE(n >= 0 && c > 0 && hin > 0 && win > 0 && h > 0 && w > 0 && src_batch_div > 0,
              "grid_sample: bad shape n=%d c=%d in=%dx%d out=%dx%d", n, c, hin, win, h, w);
  if (n == 0) return DVD_OK;
  DVD_REQUIRE(h <= 65535 && n <= 65535, "grid_sample: h or n exceeds the 65535 grid limit");
  DVD_REQUIRE(cdiv(h, 4) <= 65535, "grid_sample: h too large");
  const bool planes32 = (size_t)hin * win * 4 < (1ull << 32) && (size_t)w * 4 < (1ull << 32);
  const bool out32 = (size_t)h * w * 4 < (1ull << 32);
  // LDS-staged tiles: 16-byte row loads need win % 4 == 0 and a 16-byte aligned source; chosen from the SHAPE only
  if (win >= 4 && win % 4 == 0 && w % 4 == 0 && ((uintptr_t)src % 16) == 0 && ((uintptr_t)grid % 16) == 0 &&
      ((uintptr_t)out % 16) == 0 && planes32 && out32 && warp_variant() == 0) {
#define LDS_LAUNCH(TW_, CAP_, NT_) LDS_LAUNCH_D(TW_, CAP_, NT_, 0)
#define LDS_LAUNCH_D(TW_, CAP_, NT_, DMA_)                                                                            \
  {                                                                                                                   \
    const int ntx = cdiv(w, TW_), nty = cdiv(h, 1024 / TW_);                                                          \
    const size_t total = (size_t)ntx * nty * n;                                                                       \
    DVD_REQUIRE(total < (1ull << 31) - 8, "grid_sample: too many tiles");                                             \
    const unsigned per_xcd = (unsigned)((total + 7) / 8);                                                             \
    grid_sample_lds_kernel<TW_, CAP_, NT_, DMA_><<<per_xcd * 8u, 256, 0, (hipStream_t)stream>>>(                       \
        src, grid, out, c, hin, win, h, w, src_batch_div, ntx, nty, (unsigned)total, per_xcd);                        \
  }
#ifdef DVD_LAB
    switch (lds_tile_variant()) {      // lab: tile shapes / LDS budgets / streaming hints that were measured
      case 1: LDS_LAUNCH(32, 3072, 1) return check_launch("grid_sample(lds 32x32 cap 3072 nt)");
      case 2: LDS_LAUNCH(32, 3072, 0) return check_launch("grid_sample(lds 32x32 cap 3072)");
      case 3: LDS_LAUNCH(64, 4096, 0) return check_launch("grid_sample(lds 64x16 cap 4096)");
      case 4: LDS_LAUNCH(64, 3072, 0) return check_launch("grid_sample(lds 64x16 cap 3072)");
      case 5: LDS_LAUNCH(128, 4096, 0) return check_launch("grid_sample(lds 128x8 cap 4096)");
      case 6: LDS_LAUNCH(32, 1536, 0) return check_launch("grid_sample(lds 32x32 cap 1536)");
      case 7: LDS_LAUNCH(32, 1024, 0) return check_launch("grid_sample(lds 32x32 cap 1024)");
      case 8: LDS_LAUNCH_D(32, 2048, 0, 1) return check_launch("grid_sample(lds 32x32 cap 2048 dma)");
      case 9: LDS_LAUNCH_D(32, 1536, 0, 1) return check_launch("grid_sample(lds 32x32 cap 1536 dma)");
      case 10: LDS_LAUNCH_D(32, 1024, 0, 1) return check_launch("grid_sample(lds 32x32 cap 1024 dma)");
      default: break;
    }
#endif
    LDS_LAUNCH(LTW, LCAP_P, 0)
#undef LDS_LAUNCH
#undef LDS_LAUNCH_D
    return check_launch("grid_sample(lds)");
  }
  if (win >= 2 && planes32 && warp_variant() != 2) {
    dim3 grd(cdiv(w, 64), cdiv(h, 4 * KR), n);
    grid_sample_rows_kernel<<<grd, 256, 0, (hipStream_t)stream>>>(src, grid, out, c, hin, win, h, w, src_batch_div);
    return check_launch("grid_sample");
  }
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

extern "C" int dvd_unwarp_f32_batch(const float* flow, int g, const float* src_chw, float* out_hwc, int n, int h,
                                    int w, float scale, void* stream) {
  if (int e = unwarp_args(flow, src_chw, out_hwc, g, h, w)) return e;
  DVD_REQUIRE(n >= 0 && n <= 65535, "unwarp: bad batch %d", n);
  if (n == 0) return DVD_OK;
  if (w >= 2 && (size_t)h * w * 12 < (1ull << 32) && !scalar_warp()) {
    dim3 grd(cdiv(w, 64), cdiv(h, 4 * KR), n);
    unwarp_f32_rows_kernel<<<grd, 256, 0, (hipStream_t)stream>>>(flow, src_chw, out_hwc, make_up(g, h, w, scale));
    return check_launch("unwarp_f32");
  }
  dim3 grd(cdiv(w, 256), cdiv(h, 4), n);
  unwarp_f32_kernel<<<grd, 256, 0, (hipStream_t)stream>>>(flow, src_chw, out_hwc, make_up(g, h, w, scale));
  return check_launch("unwarp_f32");
}

extern "C" int dvd_unwarp_f32(const float* flow, int g, const float* src_chw, float* out_hwc, int h, int w,
                              float scale, void* stream) {
  return dvd_unwarp_f32_batch(flow, g, src_chw, out_hwc, 1, h, w, scale, stream);
}

extern "C" int dvd_unwarp_u8_batch(const float* flow, int g, const uint8_t* src_hwc, uint8_t* out_hwc, int n, int h,
                                   int w, float scale, void* stream) {
  if (int e = unwarp_args(flow, src_hwc, out_hwc, g, h, w)) return e;
  DVD_REQUIRE(n >= 0 && n <= 65535, "unwarp: bad batch %d", n);
  if (n == 0) return DVD_OK;
  if (w % 4 == 0 && (size_t)h * w * 3 < (1ull << 32) && (size_t)h * w * 3 >= 12 && !scalar_warp()) {
#ifdef DVD_LAB
    if (const char* e = getenv("DVD_WARP_U8_UB")) {      // lab: band height x unrolled or not
      const int ub = atoi(e);
      const bool un = getenv("DVD_WARP_U8_UNROLL") != nullptr;
      const UpParams up = make_up(g, h, w, scale);
      dim3 grd(cdiv(w, 256), cdiv(h, 4 * ub), n);
#define UBL(U_, N_) unwarp_u8_band_kernel<U_, N_><<<grd, 256, 0, (hipStream_t)stream>>>(flow, src_hwc, out_hwc, up)
      if (ub == 2) { if (un) UBL(2, true); else UBL(2, false); }
      else if (ub == 4) { if (un) UBL(4, true); else UBL(4, false); }
      else if (ub == 16) { if (un) UBL(16, true); else UBL(16, false); }
      else { if (un) UBL(8, true); else UBL(8, false); }
#undef UBL
      return check_launch("unwarp_u8(lab band)");
    }
#endif
    dim3 grd(cdiv(w, 256), cdiv(h, 4), n);
    unwarp_u8_rows_kernel<<<grd, 256, 0, (hipStream_t)stream>>>(flow, src_hwc, out_hwc, make_up(g, h, w, scale));
    return check_launch("unwarp_u8");
  }
  dim3 grd(cdiv(w, 256), cdiv(h, 4), n);
  unwarp_u8_kernel<<<grd, 256, 0, (hipStream_t)stream>>>(flow, src_hwc, out_hwc, make_up(g, h, w, scale));
  return check_launch("unwarp_u8");
}

extern "C" int dvd_unwarp_u8(const float* flow, int g, const uint8_t* src_hwc, uint8_t* out_hwc, int h, int w,
                             float scale, void* stream) {
  return dvd_unwarp_u8_batch(flow, g, src_hwc, out_hwc, 1, h, w, scale, stream);
}


#ifdef DVD_LAB
// ---------------------------------------------------------------------------------------
// LAB: the streaming CEILING of the drop-in grid_sample's access pattern (round-3 VERDICT, weak 10: "the ceiling quoted
// against it is a hipMemcpy").  Same launch geometry as grid_sample_lds_kernel (32 x 32 tiles in XCD bands, 256 threads x
// 4 consecutive pixels), same streams - two grid planes and c source planes read, c output planes written, every access
// 16 bytes per lane - but NO gather: a thread's source address is its own output address (needs hin == h, win == w).  What
// this kernel reaches is what eight interleaved 16-byte-per-lane plane streams reach on the box; the distance from it to
// grid_sample_lds_kernel is the price of the dependent grid -> footprint -> LDS round trip.
// ---------------------------------------------------------------------------------------
template <int NT, int TW>
__global__ void __launch_bounds__(256) stream_copy_planes_kernel(const float* __restrict__ src, const float* __restrict__ grid,
                                                                float* __restrict__ out, int c, int h, int w, int ntx, int nty,
                                                                unsigned tiles_total, unsigned tiles_per_xcd) {
  const unsigned seq = blockIdx.x >> 3, xcd = blockIdx.x & 7u;
  const unsigned tile = xcd * tiles_per_xcd + seq;
  if (seq >= tiles_per_xcd || tile >= tiles_total) return;
  const unsigned per_img = (unsigned)ntx * (unsigned)nty;
  const int n = (int)(tile / per_img);
  const unsigned rem = tile - (unsigned)n * per_img;
  const int ty = (int)(rem / (unsigned)ntx), tx = (int)(rem - (unsigned)ty * (unsigned)ntx);
  const int tid = threadIdx.x;
  constexpr int TPR = TW / 4, TH = 256 / TPR;
  const int x = tx * TW + 4 * (tid % TPR), y = ty * TH + (tid / TPR);
  const size_t hw = (size_t)h * w;
  const bool live = x < w && y < h;
  const uint32_t poff = ((uint32_t)min(y, h - 1) * (uint32_t)w + (uint32_t)min(x, w - 4)) * 4u;
  const float* g = grid + (size_t)n * 2 * hw;
  const float4 gx = NT ? load_f4_nt(g, poff) : load_f4(g, poff), gy = NT ? load_f4_nt(g + hw, poff) : load_f4(g + hw, poff);
  const float bias = (gx.x + gx.y + gx.z + gx.w + gy.x + gy.y + gy.z + gy.w) * 1e-30f;   // keeps the grid loads alive
  const float* s = src + (size_t)n * c * hw;
  float* o = out + (size_t)n * c * hw;
  for (int ch = 0; ch < c; ++ch) {
    float4 v = load_f4(s + (size_t)ch * hw, poff);
    v.x += bias; v.y += bias; v.z += bias; v.w += bias;
    if (live) {
      if (NT) store_f4_nt(o + (size_t)ch * hw, poff, v);
      else *reinterpret_cast<float4*>(reinterpret_cast<char*>(o + (size_t)ch * hw) + poff) = v;
    }
  }
}

extern "C" int dvd_lab_stream_copy_planes(const float* src, const float* grid, float* out, int n, int c, int h, int w,
                                          int nontemporal, int tile_w, void* stream) {
  DVD_REQUIRE(src && grid && out && n > 0 && c > 0 && w % 4 == 0, "lab_stream_copy_planes: bad arguments");
  DVD_REQUIRE(tile_w == 32 || tile_w == 64 || tile_w == 128 || tile_w == 256, "lab_stream_copy_planes: tile_w in {32,64,128,256}");
  const int ntx = cdiv(w, tile_w), nty = cdiv(h, 1024 / tile_w);
  const size_t total = (size_t)ntx * nty * n;
  const unsigned per_xcd = (unsigned)((total + 7) / 8);
#define SC_LAUNCH(NT_, TW_)                                                                                          \
  stream_copy_planes_kernel<NT_, TW_><<<per_xcd * 8u, 256, 0, (hipStream_t)stream>>>(src, grid, out, c, h, w, ntx, nty, \
                                                                                   (unsigned)total, per_xcd)
  if (nontemporal) {
    if (tile_w == 32) SC_LAUNCH(1, 32); else if (tile_w == 64) SC_LAUNCH(1, 64); else if (tile_w == 128) SC_LAUNCH(1, 128);
    else SC_LAUNCH(1, 256);
  } else {
    if (tile_w == 32) SC_LAUNCH(0, 32); else if (tile_w == 64) SC_LAUNCH(0, 64); else if (tile_w == 128) SC_LAUNCH(0, 128);
    else SC_LAUNCH(0, 256);
  }
#undef SC_LAUNCH
  return check_launch("lab_stream_copy_planes");
}
#endif
