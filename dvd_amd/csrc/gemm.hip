// MFMA GEMM for the denoiser:  C[M,N] = A[M,K] . B[N,K]^T  (+ fused epilogue), both operands
// K-contiguous ("NT"), i.e. activations [tokens, K] against nn.Linear / 1x1-conv weights [N, K].
//
//   f16 variant : v_mfma_f32_32x32x16_f16, fp32 accumulate   (per-step GEMMs: K8-K10, K13-K14)
//   f32 variant : v_mfma_f32_32x32x2_f32, exact fp32          (once-per-document work: K3-K5 and
//                                                              the conv pyramid as im2col GEMM)
//
// Tile 128x128 per 256-thread workgroup (4 waves as 2x2, each 64x64 = 2x2 MFMA tiles), K-step 64
// halfs / 16 floats, LDS double-buffered with register-staged prefetch (global loads of tile k+1
// are issued before the MFMAs of tile k, written to the other buffer after them: one barrier per
// K-step).  LDS rows are padded by 16 B so the 16 rows of a ds_read_b128 lane group fall on 16
// different 16-B slots (conflict-free, guide section 2).  Two workgroups per CU.
//
// The epilogue replaces what the reference does in separate ATen ops after each Linear:
// bias, GELU(tanh) / ReLU, positional-embedding add, adaLN gate, residual add, f16/f32 stores.
#include "common.h"
#include "mfma.h"
#include <type_traits>

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
};

__device__ __forceinline__ float gelu_tanh(float x) {
  const float k0 = 0.7978845608028654f, k1 = 0.044715f;
  float u = k0 * (x + k1 * x * x * x);
  return 0.5f * x * (1.f + tanhf(u));
}

template <bool F32>
__global__ void __launch_bounds__(256, 2) gemm_nt_kernel(GemmArgs p) {
  using T = typename std::conditional<F32, float, _Float16>::type;
  constexpr int BK = F32 ? 16 : 64;                 // elements per K-step
  constexpr int ROWB = BK * (int)sizeof(T);          // payload bytes per LDS row (64 / 128)
  constexpr int LROW = ROWB + 16;                    // padded row pitch
  constexpr int CH = ROWB / 16;                      // 16-B chunks per row
  constexpr int NLD = 128 * CH / 256;                // chunks per thread per operand
  constexpr int EPC = 16 / (int)sizeof(T);           // elements per chunk
  __shared__ __attribute__((aligned(16))) char smem[2][2][128 * LROW];

  // XCD-aware tile order: consecutive ids inside one XCD walk the N tiles of one M panel.
  const int nwg = p.ntm * p.ntn;
  int id = blockIdx.x;
  {
    const int q = nwg / 8, r = nwg % 8, xcd = id % 8, k = id / 8;
    id = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + k;
  }
  const int tm = id / p.ntn, tn = id % p.ntn;
  const int bm0 = tm * 128, bn0 = tn * 128;
  const int z = blockIdx.y;
  const T* A = (const T*)p.A + z * p.sA;
  const T* B = (const T*)p.B + z * p.sB;
  const T* Blo = p.Blo ? (const T*)p.Blo + z * p.sB : nullptr;
  const T* Alo = p.Alo ? (const T*)p.Alo + z * p.sA : nullptr;

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, h = lane >> 5;
  const int wr = wave >> 1, wc = wave & 1;

  // per-thread staging slots
  const T* ga[NLD];
  const T* gb[NLD];
  const T* gbl[NLD];
  const T* gal[NLD];
  int lofs[NLD];
#pragma unroll
  for (int i = 0; i < NLD; ++i) {
    const int c = tid + 256 * i;
    const int row = c / CH, ch = c % CH;
    const int ra = min(bm0 + row, p.M - 1), rb = min(bn0 + row, p.N - 1);
    ga[i] = A + (size_t)ra * p.lda + ch * EPC;
    gb[i] = B + (size_t)rb * p.ldb + ch * EPC;
    gbl[i] = Blo ? Blo + (size_t)rb * p.ldb + ch * EPC : gb[i];
    gal[i] = Alo ? Alo + (size_t)ra * p.lda + ch * EPC : ga[i];
    lofs[i] = row * LROW + ch * 16;
  }
  // Split weights: the low parts are accumulated FIRST, the accumulator is scaled by lo_scale (a power of
  // two: exact), then the high parts are added -> one accumulator, fp32-grade weights at 2x the MFMAs.
  const int nkk = p.K / BK;
  const int nlo = (Blo || Alo) ? nkk : 0;
  const int nk = nkk + nlo;

  // Register-staged prefetch, TWO K-tiles deep: tile j waits in register set (j & 1) for two full iterations
  // before it is written to LDS, so a global load has ~2 x (16 MFMA x 2 waves) of matrix time to land.  With
  // one-tile-deep staging every iteration stalled on L2/HBM latency (2 workgroups per CU cannot hide it).
  u32x4 ra0[NLD], rb0[NLD], ra1[NLD], rb1[NLD];
#define DVD_GLOAD(RA, RB, t_)                                                    \
  {                                                                              \
    const int tt_ = (t_);                                                        \
    const bool lo_ = tt_ < nlo;                                                  \
    const size_t kofs_ = (size_t)(lo_ ? tt_ : tt_ - nlo) * BK;                   \
    _Pragma("unroll") for (int i = 0; i < NLD; ++i) {                            \
      RA[i] = *(const u32x4*)((lo_ ? gal[i] : ga[i]) + kofs_);                   \
      RB[i] = *(const u32x4*)((lo_ ? gbl[i] : gb[i]) + kofs_);                   \
    }                                                                            \
  }
#define DVD_LSTORE(RA, RB, buf_)                                                 \
  _Pragma("unroll") for (int i = 0; i < NLD; ++i) {                              \
    *(u32x4*)(&smem[buf_][0][lofs[i]]) = RA[i];                                  \
    *(u32x4*)(&smem[buf_][1][lofs[i]]) = RB[i];                                  \
  }
#define DVD_COMPUTE(buf_)                                                                          \
  {                                                                                                \
    const char* sa = &smem[buf_][0][(64 * wr + r) * LROW];                                         \
    const char* sb = &smem[buf_][1][(64 * wc + r) * LROW];                                         \
    if constexpr (!F32) {                                                                          \
      _Pragma("unroll") for (int s = 0; s < 4; ++s) {                                              \
        half8 a[2], b[2];                                                                          \
        _Pragma("unroll") for (int m = 0; m < 2; ++m)                                              \
          a[m] = *(const half8*)(sa + m * 32 * LROW + (16 * s + 8 * h) * 2);                       \
        _Pragma("unroll") for (int n = 0; n < 2; ++n)                                              \
          b[n] = *(const half8*)(sb + n * 32 * LROW + (16 * s + 8 * h) * 2);                       \
        _Pragma("unroll") for (int m = 0; m < 2; ++m)                                              \
          _Pragma("unroll") for (int n = 0; n < 2; ++n) acc[m][n] = mfma32_f16(a[m], b[n], acc[m][n]); \
      }                                                                                            \
    } else {                                                                                       \
      /* lane-half h owns k = 8h .. 8h+7 of the 16-deep step (any k assignment is valid if A and B agree) */ \
      float a[2][8], b[2][8];                                                                      \
      _Pragma("unroll") for (int m = 0; m < 2; ++m) {                                              \
        const floatx4 lo = *(const floatx4*)(sa + m * 32 * LROW + 32 * h);                         \
        const floatx4 hi = *(const floatx4*)(sa + m * 32 * LROW + 32 * h + 16);                    \
        _Pragma("unroll") for (int e = 0; e < 4; ++e) { a[m][e] = lo[e]; a[m][4 + e] = hi[e]; }     \
      }                                                                                            \
      _Pragma("unroll") for (int n = 0; n < 2; ++n) {                                              \
        const floatx4 lo = *(const floatx4*)(sb + n * 32 * LROW + 32 * h);                         \
        const floatx4 hi = *(const floatx4*)(sb + n * 32 * LROW + 32 * h + 16);                    \
        _Pragma("unroll") for (int e = 0; e < 4; ++e) { b[n][e] = lo[e]; b[n][4 + e] = hi[e]; }     \
      }                                                                                            \
      _Pragma("unroll") for (int s = 0; s < 8; ++s)                                                \
        _Pragma("unroll") for (int m = 0; m < 2; ++m)                                              \
          _Pragma("unroll") for (int n = 0; n < 2; ++n) acc[m][n] = mfma32_f32(a[m][s], b[n][s], acc[m][n]); \
    }                                                                                              \
  }
#define DVD_LOSCALE(kt_)                                                                           \
  if (nlo && (kt_) == nlo - 1) {                                                                   \
    _Pragma("unroll") for (int m = 0; m < 2; ++m)                                                  \
      _Pragma("unroll") for (int n = 0; n < 2; ++n)                                                \
        _Pragma("unroll") for (int i = 0; i < 16; ++i) acc[m][n][i] *= p.lo_scale;                 \
  }

  floatx16 acc[2][2];
#pragma unroll
  for (int m = 0; m < 2; ++m)
#pragma unroll
    for (int n = 0; n < 2; ++n)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[m][n][i] = 0.f;

  // Loads and LDS stores are issued UNCONDITIONALLY (tile index clamped to the last tile: a couple of
  // redundant tile loads per workgroup) so that the number of loads in flight is the same on every path and
  // hipcc can emit the counted vmcnt(8) that lets the younger register set stay in flight across the store.
  const int last = nk - 1;
  DVD_GLOAD(ra0, rb0, 0)
  DVD_GLOAD(ra1, rb1, min(1, last))
  DVD_LSTORE(ra0, rb0, 0)
  __syncthreads();
  DVD_GLOAD(ra0, rb0, min(2, last))
  const int pairs = nk >> 1;
  for (int pp = 0; pp < pairs; ++pp) {
    const int kt = 2 * pp;
    // even step: tile kt is in LDS[0]; tile kt+1 waits in set 1, tile kt+2 in set 0
    DVD_COMPUTE(0)
    DVD_LOSCALE(kt)
    DVD_LSTORE(ra1, rb1, 1)
    __syncthreads();
    DVD_GLOAD(ra1, rb1, min(kt + 3, last))
    // odd step: tile kt+1 is in LDS[1]
    DVD_COMPUTE(1)
    DVD_LOSCALE(kt + 1)
    DVD_LSTORE(ra0, rb0, 0)
    __syncthreads();
    DVD_GLOAD(ra0, rb0, min(kt + 4, last))
  }
  if (nk & 1) {   // odd tile count (no split weights): the last tile is already in LDS[0]
    DVD_COMPUTE(0)
    DVD_LOSCALE(nk - 1)
  }
#undef DVD_GLOAD
#undef DVD_LSTORE
#undef DVD_COMPUTE
#undef DVD_LOSCALE

  // ---------------- epilogue ----------------
  float* C32 = p.C32 ? p.C32 + z * p.sC32 : nullptr;
  _Float16* C16 = p.C16 ? p.C16 + z * p.sC16 : nullptr;
  const float* bias = p.bias ? p.bias + z * p.sBias : nullptr;
  const float* res = p.res ? p.res + z * p.sRes : nullptr;
  const float* gate = p.gate ? p.gate + z * p.sGate : nullptr;
#pragma unroll
  for (int n = 0; n < 2; ++n) {
    const int col = bn0 + 64 * wc + 32 * n + r;
    if (col >= p.N) continue;
    const float bcol = (bias && !p.bias_row) ? bias[col] : 0.f;
#pragma unroll
    for (int m = 0; m < 2; ++m) {
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int row = bm0 + 64 * wr + 32 * m + cd_row(i, h);
        if (row >= p.M) continue;
        float v = acc[m][n][i] + bcol;
        if (bias && p.bias_row) v += bias[row];
        if (p.act == 1) v = gelu_tanh(v);
        else if (p.act == 2) v = fmaxf(v, 0.f);
        if (p.pos) v += p.pos[(size_t)(row % p.pos_rows) * p.ldpos + col];
        if (gate) v *= gate[(size_t)(row / p.gate_rows) * p.ldgate + col];
        if (res) v += res[(size_t)row * p.ldres + col];
        if (C32) C32[(size_t)row * p.ldc + col] = v;
        if (C16) C16[(size_t)row * p.ldc16 + col] = (_Float16)v;
      }
    }
  }
}

}  // namespace dvd

using namespace dvd;

extern "C" int dvd_gemm_nt(const dvd_gemm_desc* d, void* stream) {
  DVD_REQUIRE(d && d->A && d->B && (d->C32 || d->C16), "gemm: null pointer");
  DVD_REQUIRE(d->dtype == 0 || d->dtype == 1, "gemm: dtype must be 0 (f16) or 1 (f32)");
  const int bk = d->dtype == 1 ? 16 : 64, esz = d->dtype == 1 ? 4 : 2;
  DVD_REQUIRE(d->M > 0 && d->N > 0 && d->K > 0 && d->K % bk == 0, "gemm: K=%d must be a positive multiple of %d",
              d->K, bk);
  DVD_REQUIRE(d->batch >= 1 && d->batch <= 65535, "gemm: bad batch %d", d->batch);
  DVD_REQUIRE(((size_t)d->lda * esz) % 16 == 0 && ((size_t)d->ldb * esz) % 16 == 0 &&
                  ((uintptr_t)d->A % 16) == 0 && ((uintptr_t)d->B % 16) == 0 &&
                  ((size_t)d->strideA * esz) % 16 == 0 && ((size_t)d->strideB * esz) % 16 == 0,
              "gemm: operands must be 16-byte aligned (lda=%d ldb=%d)", d->lda, d->ldb);
  DVD_REQUIRE(!d->gate || d->gate_rows > 0, "gemm: gate needs gate_rows");
  DVD_REQUIRE(!d->pos || d->pos_rows > 0, "gemm: pos needs pos_rows");
  GemmArgs p;
  DVD_REQUIRE(!(d->B_lo && d->A_lo), "gemm: only one operand may be split");
  DVD_REQUIRE((!d->B_lo && !d->A_lo) ||
                  (d->dtype == 0 && ((uintptr_t)d->B_lo % 16) == 0 && ((uintptr_t)d->A_lo % 16) == 0),
              "gemm: split operands need dtype f16, 16-byte aligned");
  p.Blo = d->B_lo; p.Alo = d->A_lo; p.lo_scale = d->lo_scale;
  p.A = d->A; p.B = d->B; p.C32 = d->C32; p.C16 = (_Float16*)d->C16;
  p.bias = d->bias; p.res = d->res; p.gate = d->gate; p.pos = d->pos;
  p.sA = d->strideA; p.sB = d->strideB; p.sC32 = d->strideC32; p.sC16 = d->strideC16;
  p.sBias = d->strideBias; p.sRes = d->strideRes; p.sGate = d->strideGate;
  p.M = d->M; p.N = d->N; p.K = d->K;
  p.lda = d->lda; p.ldb = d->ldb; p.ldc = d->ldc; p.ldc16 = d->ldc16; p.ldres = d->ldres;
  p.ldgate = d->ldgate; p.ldpos = d->ldpos;
  p.gate_rows = d->gate_rows; p.pos_rows = d->pos_rows;
  p.act = d->act; p.bias_row = d->bias_row;
  p.ntm = cdiv(d->M, 128); p.ntn = cdiv(d->N, 128);
  dim3 grid(p.ntm * p.ntn, d->batch);
  if (d->dtype == 1)
    gemm_nt_kernel<true><<<grid, 256, 0, (hipStream_t)stream>>>(p);
  else
    gemm_nt_kernel<false><<<grid, 256, 0, (hipStream_t)stream>>>(p);
  return check_launch("gemm_nt");
}
