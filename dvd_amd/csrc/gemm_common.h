// Shared by the GEMM translation units (gemm.hip, gemm_t384.hip): the argument block, the persistent kernels' tile walk and
// the GELU of the epilogues.
#pragma once
#include "common.h"
#include "mfma.h"

namespace dvd {

struct GemmArgs {
  const void* A;      // [M,K] lda
  const void* B;      // [N,K] ldb
  const void* Blo;    // optional low part of a split weight: B_true = B + lo_scale * Blo (same layout as B)
  const void* Alo;    // ... or on the A side (swapped GEMMs put the weight in A); at most one of the two
  float lo_scale;
  float* C32;         // optional [M,N] ldc
  _Float16* C16;      // optional [M,N] ldc16
  const float* bias;  // optional, per column (bias_row = 0) or per row (bias_row = 1)
  const float* res;   // optional residual [M,N] ldres (fp32), added last
  const float* gate;  // optional [M / gate_rows, N] ldgate: out = gate * (acc + bias)
  const float* pos;   // optional [pos_rows, N] ldpos: out += pos[row % pos_rows]
  long sA, sB, sC32, sC16, sBias, sRes, sGate;  // batch strides in elements (blockIdx.y)
  int M, N, K;
  int lda, ldb, ldc, ldc16, ldres, ldgate, ldpos;
  int gate_rows, pos_rows;
  int act;       // 0 none, 1 GELU(tanh), 2 ReLU
  int bias_row;  // bias indexed by row instead of column
  int ntm, ntn;  // tile counts
  int walk;          // gemm_nt_t384_kernel: tile walk variant (tile_coords)
  int stagger;       // large-tile kernel: start-up delay quantum (x4096 cycles) of the first round of workgroups
  int vec_epilogue;  // 1: LDS-staged row-contiguous stores (needs N % 8 == 0 and 16-byte aligned rows)
  int debug;     // timing ablations only (DVD_GEMM_DEBUG): 1 = no operand loads in the K loop, 2 = no MFMAs,
                 // 3 = per-wave s_memtime stamps (start, first tile landed, K loop done, epilogue done) -> stamps
  unsigned long long* stamps;
  // gemm_nt_kernel<f32, CONV>: A is not a matrix but a channels-last map [M = images * cv_h * cv_w pixels, cv_ca] (plus an
  // optional second source cv_b [M, cv_cb], the channel concatenation) and column k of row m is tap k / (cv_ca + cv_cb) of
  // the cv_ks x cv_ks window (dilation cv_dil, zero outside the map), channel k % (cv_ca + cv_cb): the im2col matrix, unbuilt
  const float* cv_b;
  int cv_ca, cv_cb, cv_h, cv_w, cv_ks, cv_dil;
};

// conv (cv_ks x cv_ks, stride 1, 'same' padding) + bias (+ReLU: act 2) of a channels-last f32 map as an implicit GEMM on the
// exact-f32 128 x 128 kernel: the arithmetic - and the bits - of im2col + dvd_gemm_nt(f32).  ca, cb multiples of 16,
// kp == ks * ks * (ca + cb).  Returns DVD_OK or an error code (error.hip).
int launch_gemm_conv_f32(const float* a, int ca, const float* b, int cb, int h, int w, int ks, int dil, long rows,
                         const float* wgt, int kp, const float* bias, float* out, int cout, int act, void* stream);

// Tile walk of the persistent kernels.  Virtual id `vid` runs on XCD vid % 8 (hardware round-robin) as that XCD's k-th
// tile, k = vid / 8; an XCD owns a contiguous range of row panels.  At any time an XCD's 32 CUs work on 32 consecutive k.
//  * default: row-major inside the XCD's range - the 32 tiles cover ~32/ntn row panels x all N tiles, so per 32 tiles the
//    XCD's 4 MB L2 sees (32/ntn) A panels + ntn W panels;
//  * few row panels (ntm < 8) and many N tiles: column-major (see below);
//  * wide outputs (ntn a multiple of 4, >= 8; needs ntm % 8 == 0): blocks of 8 row panels x 4 N tiles - 8 A panels (0.79 MB
//    each at K = 1536) + 4 W panels (1.57 MB each, hi + lo), the minimum of a_bytes * rows + w_bytes * cols at rows * cols
//    = 32; at N = 3072 the row-major walk streamed all 12 W panels (18.9 MB) through L2 for every 32 tiles.
// The order changes which workgroup computes a tile, never a tile's arithmetic.
__device__ __forceinline__ void tile_coords(int vid, int ntm, int ntn, int& tm, int& tn, int walk = 0) {
  const int nwg = ntm * ntn;
  const int q = nwg / 8, rr = nwg % 8, xcd = vid % 8, k = vid / 8;
  if (walk == 1 && ntn == 6 && (ntm & 7) == 0) {
    // six N tiles (the N = 1536 GEMMs) in two groups of three: an XCD's 32 concurrent tiles are ~10.7 row panels x 3 W panels
    // - the three W panels (2.4 MB at K = 1536) stay in its 4 MB L2 while the A panels stream through, each read twice overall;
    // row-major over all six keeps 4.7 MB of W panels + 6.3 MB of A panels in flight and re-fetches W about every other time
    const int rows = ntm >> 3, per = rows * 3;
    const int cg = k / per, rem = k - cg * per;
    tm = xcd * rows + rem / 3;
    tn = 3 * cg + rem % 3;
    return;
  }
  if (rr == 0 && (ntm & 7) == 0 && (ntn & 3) == 0 && ntn >= 8) {
    const int rows = ntm >> 3;               // row panels per XCD
    const int grp = 8 * ntn;                 // tiles in a group of 8 row panels (a multiple of 32)
    const int g = k / grp, rem = k - g * grp;
    if (g < (rows >> 3)) {
      const int b = rem >> 5, i = rem & 31;
      tm = xcd * rows + g * 8 + (i & 7);
      tn = 4 * b + (i >> 3);
    } else {                                 // the last rows % 8 row panels of the XCD: row-major
      const int k2 = k - (rows >> 3) * grp;
      tm = xcd * rows + (rows & ~7) + k2 / ntn;
      tn = k2 % ntn;
    }
    return;
  }
  const int id = (xcd < rr ? xcd * (q + 1) : rr * (q + 1) + (xcd - rr) * q) + k;
  if (ntm < 8 && ntn >= 32) {      // few row panels, many N tiles (the V^T projections: weights on the A side, M = 1536):
    tn = id / ntm;                 // column-major - the 32 tiles an XCD works on share ALL ntm A panels and 32 / ntm B
    tm = id % ntm;                 // panels (9 MB at ntm = 6) instead of one A panel and 32 B panels (26 MB)
    return;
  }
  tm = id / ntn;
  tn = id % ntn;
}


// GELU(tanh) (timm Mlp's act layer, idf/cross_model.py:163-174):  0.5 x (1 + tanh u) = x sigmoid(2 u) = x / (1 + exp(-2 u)),
// u = k0 (x + k1 x^3).  Written with ONE exponential and one reciprocal (v_exp_f32 / v_rcp_f32, ~1 ulp each: 1e-7 relative
// on a value that is then rounded to f16) instead of tanhf, whose library expansion is ~4x the instructions and a branch:
// fc1's epilogue applies it to 2 G elements per evaluation and was half of that GEMM's 3.8 ms (K = 384: six K slabs per
// tile).  Range: for x <= -10.5 the tanh form is exactly -0 in fp32 arithmetic (1 + tanh u rounds to 0) while
// exp(-2u) overflows here (x / inf = -0 for finite x, but -inf / inf would be NaN), so that range returns -0 explicitly -
// also at -inf, where torch's own formula 0.5 x (1 + tanh u) evaluates -inf * 0 = NaN (no finite activation gets there);
// x -> +inf gives x, NaN propagates.  tests/test_gpu_gemm.py::test_gelu_epilogue_range checks [-12, 12], the exp-overflow
// region and +-1e4 / +inf against torch's gelu(approximate='tanh').
__device__ __forceinline__ float gelu_tanh(float x) {
  const float k0 = 0.7978845608028654f, k1 = 0.044715f;
  const float u = k0 * (x + k1 * x * x * x);
  const float y = __fdividef(x, 1.f + __expf(-2.f * u));
  return x < -10.5f ? -0.f : y;
}

// gemm_t384.hip: the 384 x 256 kernel (lab builds: dbg selects a timing ablation / the stamp build)
int launch_gemm_t384(const GemmArgs& p, int batch, int dbg, void* stream);

}  // namespace dvd
