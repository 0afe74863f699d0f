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
#include "gemm_common.h"
#include <stdlib.h>
#include <string.h>
#include <type_traits>

#ifndef T384_WALK
#define T384_WALK 0         // tile walk of gemm_nt_t384_kernel for six N tiles: 0 row-major, 1 two groups of three (tile_coords)
#endif
// gemm_nt_ring128_kernel takes the f16 128 x 128-tile problems with at most this many tiles (profiles/r5_gemm_ring128.txt)
#define RING128_MAX_TILES 256
// ... and gemm_nt_ring256_kernel those with N % 256 == 0 whose 128 x 256 tiles fill most of the chip once (with fewer tiles
// the 128 x 128 ring kernel on twice as many CUs is faster: profiles/r5_gemm_ring256.txt)
#define RING256_MAX_TILES 256
#define RING256_MIN_TILES 160
#ifndef T384_STAGGER
#define T384_STAGGER 0      // start-up delay quantum of gemm_nt_t384_kernel (x 1024 cycles x 0..15 per workgroup); see the kernel
#endif
#ifndef DVD_GEMM_SPREAD_FIRST
#define DVD_GEMM_SPREAD_FIRST 0
#endif
#ifndef DVD_GEMM_SPREAD_STEP
#define DVD_GEMM_SPREAD_STEP 1
#endif

namespace dvd {

// Epilogue of one 32x32 accumulator tile (rows row0 + cd_row(i,h), column col).  Kept as a function so the
// callers' tile loops stay small enough to be fully unrolled (a partially unrolled epilogue indexes the
// accumulator array dynamically, which homes ALL accumulators in scratch memory).
__device__ __forceinline__ void epilogue_tile(const GemmArgs& p, const floatx16& t, int row0, int col, int h,
                                              float bcol, float* C32, _Float16* C16, const float* bias,
                                              const float* res, const float* gate) {
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int row = row0 + cd_row(i, h);
    if (row < p.M) {
      float v = t[i] + bcol;
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

// ---------------------------------------------------------------------------------------------------------
// Epilogue of one wave's 64 x 64 block (2 x 2 accumulator tiles), staged through a private 16 KiB LDS region so
// that global traffic is row-contiguous 16-byte (f32) / 8-byte (f16) accesses per lane: 4 rows x 256 B (resp.
// 128 B) per wave-instruction.  The MFMA C layout has one COLUMN per lane; storing it directly costs one 2-byte
// (f16) store per element and took as long as the whole K loop (measured with s_memtime stamps: 50.6 % of a
// wave's lifetime in the 256 x 256 kernel).  All epilogue math (bias, activation, pos, gate, residual) runs on
// the read-back side, 4 consecutive columns per lane.
//   stage : this wave's LDS region, 64 rows x 64 floats (256-B pitch: conflict-free for both access patterns)
//   t00.. : accumulator tiles (rows 0-31 | 32-63) x (cols 0-31 | 32-63) of the block at (row0, col0)
// ---------------------------------------------------------------------------------------------------------
// EPI selects what the store loop may touch besides LDS (compile-time, so that the loop body contains NO global load):
//   0  plain: column bias + activation only                      -> the 8 row-contiguous stores are fire-and-forget
//   1  + residual: rows prefetched through a rolling 4-iteration window, always older than the stores they overlap
//   3  + row bias (transposed outputs): 8 scalars before the loop
//   2  general (pos / gate / residual with row bias): loads inside the loop (small GEMMs only)
// Why it matters: on CDNA4 vmcnt counts stores too and retires in order, so ANY wait for a load inside the loop - even
// the wait the compiler places after a conditional load that is skipped at run time - also waits for the previous
// iteration's global store to be acknowledged (~1300 cycles): the epilogue took 23 000 cycles per 256 x 256 tile, a
// quarter of the kernel, 16 stores x 1300 (s_memtime stamps, identical to the cycle with and without other CUs storing).
template <int EPI, int NIT = 8>   // NIT 8-row iterations: 8 = the whole 64 x 64 block, 4 = a 32-row half of it
__device__ __forceinline__ void epilogue_readback(const GemmArgs& p, float* stage, int row0, int col0, int lane,
                                                  float* C32, _Float16* C16, const float* bias, const float* res,
                                                  const float* gate);

template <int EPI>
__device__ __forceinline__ void epilogue_block64(const GemmArgs& p, float* stage, const floatx16& t00,
                                                 const floatx16& t01, const floatx16& t10, const floatx16& t11,
                                                 int row0, int col0, int lane, float* C32, _Float16* C16,
                                                 const float* bias, const float* res, const float* gate,
                                                 unsigned long long* tmid = nullptr) {
  const int r = lane & 31, h = lane >> 5;
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int rr = cd_row(i, h);
    stage[rr * 64 + r] = t00[i];
    stage[rr * 64 + 32 + r] = t01[i];
    stage[(32 + rr) * 64 + r] = t10[i];
    stage[(32 + rr) * 64 + 32 + r] = t11[i];
  }
  if (tmid) {                                   // diagnostic builds: LDS writes retired
    __builtin_amdgcn_s_waitcnt(0xC07F);
    *tmid = __builtin_amdgcn_s_memtime();
  }
  epilogue_readback<EPI>(p, stage, row0, col0, lane, C32, C16, bias, res, gate);
}

// The same 64 x 64 block held as 4 x 4 accumulator tiles of v_mfma_f32_16x16x32_f16 (register e of lane l = row
// 4 (l >> 4) + e, column l & 15 of its tile): staged into the same LDS image, then the shared read-back.
template <int EPI>
__device__ __forceinline__ void epilogue_block64_m16(const GemmArgs& p, float* stage, const floatx4 (&t)[8][4], int mt0,
                                                     int row0, int col0, int lane, float* C32, _Float16* C16,
                                                     const float* bias, const float* res, const float* gate) {
  const int c16 = lane & 15, q = lane >> 4;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int e = 0; e < 4; ++e) stage[(16 * i + 4 * q + e) * 64 + 16 * j + c16] = t[mt0 + i][j][e];
  epilogue_readback<EPI>(p, stage, row0, col0, lane, C32, C16, bias, res, gate);
}

template <int EPI, int NIT>
__device__ __forceinline__ void epilogue_readback(const GemmArgs& p, float* stage, int row0, int col0, int lane,
                                                  float* C32, _Float16* C16, const float* bias, const float* res,
                                                  const float* gate) {
  static_assert(NIT == 8 || NIT == 4, "whole block or half block");
  // same-wave LDS accesses execute in order; the compiler inserts the lgkmcnt wait for the reads below.
  // Read back 8 consecutive columns per lane: 8 rows x 256 B (f32, two 16-B stores) / 128 B (f16, one 16-B store)
  // per wave-instruction (guide T21).
  const int c8 = (lane & 7) * 8;
  const int col = col0 + c8;
  const bool colok = col < p.N;
  float bc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (bias && !p.bias_row && colok) {
    const floatx4 b0 = *(const floatx4*)(bias + col), b1 = *(const floatx4*)(bias + col + 4);
#pragma unroll
    for (int e = 0; e < 4; ++e) { bc[e] = b0[e]; bc[4 + e] = b1[e]; }
  }
  // EPI 1: residual rows through a rolling window of 4 iterations - the load of iteration it + 4 is issued right after
  // iteration it has consumed its slot and BEFORE iteration it's stores, so (vmcnt being in-order) waiting for it never
  // waits for a store.  EPI 3: the 8 row-bias scalars up front.  All of these loads are unconditional (a conditional
  // load is a control-flow merge, after which the compiler waits vmcnt(0)).
  floatx4 rs0[4], rs1[4];
  float brow[8];
  const int cc = colok ? col : 0;
  auto res_row = [&](int it) { return res + (size_t)min(row0 + it * 8 + (lane >> 3), p.M - 1) * p.ldres + cc; };
  if constexpr (EPI == 1) {
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      rs0[it] = *(const floatx4*)res_row(it);
      rs1[it] = *(const floatx4*)(res_row(it) + 4);
    }
  }
  if constexpr (EPI == 3) {
#pragma unroll
    for (int it = 0; it < NIT; ++it) brow[it] = bias[min(row0 + it * 8 + (lane >> 3), p.M - 1)];
  }
  const bool relu = p.act == 2;
  // EPI 1 needs compile-time window slots (it & 3): unrolled by 4.  The other flavours are unrolled by 2 only: fully
  // unrolled, the compiler hoists every iteration's addresses out of the persistent tile loop and spills them, and each
  // scratch reload is a vmcnt(0) - i.e. a wait for the previous store again.
  constexpr int EPI_UNROLL = EPI == 1 ? 4 : (EPI == 3 ? NIT : 2);
#pragma unroll EPI_UNROLL
  for (int it = 0; it < NIT; ++it) {
    const int lr = it * 8 + (lane >> 3);
    const int row = row0 + lr;
    const floatx4 s0 = *(const floatx4*)(stage + lr * 64 + c8), s1 = *(const floatx4*)(stage + lr * 64 + c8 + 4);
    floatx4 rcur0 = {0.f, 0.f, 0.f, 0.f}, rcur1 = {0.f, 0.f, 0.f, 0.f};
    if constexpr (EPI == 1) {
      rcur0 = rs0[it & 3];
      rcur1 = rs1[it & 3];
      if (it + 4 < NIT) {
        rs0[it & 3] = *(const floatx4*)res_row(it + 4);
        rs1[it & 3] = *(const floatx4*)(res_row(it + 4) + 4);
      }
    }
    float br = 0.f;
    if constexpr (EPI == 3) br = brow[it];
    if constexpr (EPI == 2) br = (bias && p.bias_row) ? bias[min(row, p.M - 1)] : 0.f;
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = (e < 4 ? s0[e] : s1[e - 4]) + bc[e] + br;
    if (p.act == 1) {                              // one uniform branch per iteration (tanhf branches internally)
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = gelu_tanh(v[e]);
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = relu ? fmaxf(v[e], 0.f) : v[e];
    if (row < p.M && colok) {
      if constexpr (EPI == 2) {
        if (p.pos) {
          const float* pp = p.pos + (size_t)(row % p.pos_rows) * p.ldpos + col;
          const floatx4 p0 = *(const floatx4*)pp, p1 = *(const floatx4*)(pp + 4);
#pragma unroll
          for (int e = 0; e < 4; ++e) { v[e] += p0[e]; v[4 + e] += p1[e]; }
        }
        if (gate) {
          const float* gp = gate + (size_t)(row / p.gate_rows) * p.ldgate + col;
          const floatx4 g0 = *(const floatx4*)gp, g1 = *(const floatx4*)(gp + 4);
#pragma unroll
          for (int e = 0; e < 4; ++e) { v[e] *= g0[e]; v[4 + e] *= g1[e]; }
        }
        if (res) {
          const float* rp = res + (size_t)row * p.ldres + col;
          const floatx4 r0 = *(const floatx4*)rp, r1 = *(const floatx4*)(rp + 4);
#pragma unroll
          for (int e = 0; e < 4; ++e) { v[e] += r0[e]; v[4 + e] += r1[e]; }
        }
      }
      if constexpr (EPI == 1) {
#pragma unroll
        for (int e = 0; e < 4; ++e) { v[e] += rcur0[e]; v[4 + e] += rcur1[e]; }
      }
      if (C32) {
        float* cp = C32 + (size_t)row * p.ldc + col;
        const floatx4 o0 = {v[0], v[1], v[2], v[3]}, o1 = {v[4], v[5], v[6], v[7]};
        *(floatx4*)cp = o0;
        *(floatx4*)(cp + 4) = o1;
      }
      if (C16) {
        half8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = (_Float16)v[e];
        *(half8*)(C16 + (size_t)row * p.ldc16 + col) = o;
      }
    }
  }
}

// run-time (wave-uniform) choice of the epilogue flavour
#define DVD_EPILOGUE_BLOCK64(...)                                                           \
  {                                                                                         \
    const bool brow_ = bias && p.bias_row;                                                  \
    if (p.pos || gate || (res && brow_)) epilogue_block64<2>(__VA_ARGS__);                  \
    else if (res) epilogue_block64<1>(__VA_ARGS__);                                         \
    else if (brow_) epilogue_block64<3>(__VA_ARGS__);                                       \
    else epilogue_block64<0>(__VA_ARGS__);                                                  \
  }

template <bool F32, int PD, bool CONV = false>   // CONV (f32 only): the A operand is an unbuilt im2col matrix (GemmArgs::cv_*)
__global__ void __launch_bounds__(256, 2) gemm_nt_kernel(GemmArgs p) {
  static_assert(!CONV || F32, "implicit-GEMM operand: exact-f32 kernel only");
  using T = typename std::conditional<F32, float, _Float16>::type;
  constexpr int BK = F32 ? 16 : 64;                 // elements per K-step
  constexpr int ROWB = BK * (int)sizeof(T);          // payload bytes per LDS row (64 / 128)
  constexpr int LROW = ROWB + 16;                    // padded row pitch
  constexpr int CH = ROWB / 16;                      // 16-B chunks per row
  constexpr int NLD = 128 * CH / 256;                // chunks per thread per operand
  constexpr int EPC = 16 / (int)sizeof(T);           // elements per chunk
  // each [stage][operand] slab is at least 16 KiB so that the four slabs double as the epilogue's 4 x 16 KiB
  // per-wave staging regions (f32 operands alone would only need 10 KiB per slab)
  constexpr int SLAB = 128 * LROW > 16384 ? 128 * LROW : 16384;
  __shared__ __attribute__((aligned(16))) char smem[2][2][SLAB];

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
  // CONV: the slot's pixel (clamped row ra of the im2col matrix = pixel ra of the batch of maps) and its 4-float chunk
  long cpix[NLD];
  int cy[NLD], cx[NLD], cch[NLD];
  bool cok[PD][NLD];
  if constexpr (CONV) {
    const long hw = (long)p.cv_h * p.cv_w;
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
      const int c = tid + 256 * i;
      const int row = c / CH;
      cch[i] = (c % CH) * EPC;
      cpix[i] = min(bm0 + row, p.M - 1);
      const long pl = cpix[i] % hw;
      cy[i] = (int)(pl / p.cv_w);
      cx[i] = (int)(pl - (long)cy[i] * p.cv_w);
    }
  }
  // Split weights: the low parts are accumulated FIRST, the accumulator is scaled by lo_scale (a power of
  // two: exact), then the high parts are added -> one accumulator, fp32-grade weights at 2x the MFMAs.
  const int nkk = p.K / BK;
  const int nlo = (Blo || Alo) ? nkk : 0;
  const int nk = nkk + nlo;

  // Register-staged prefetch, PD K-tiles deep: tile j waits in register set (j % PD) for PD full iterations before it is
  // written to LDS, so a global load has ~PD x (16 MFMA x 2 waves) of matrix time to land.  With one-tile-deep staging every
  // iteration stalled on L2/HBM latency (2 workgroups per CU cannot hide it).  PD = 2 is the product; PD = 4 (lab build,
  // DVD_GEMM_PD4) measured +-3 % at the sampler's small row counts (profiles/r5_gemm_small_variants.txt), as did hoisting a
  // K-tile's 16 fragment reads above its MFMAs (+8 %: slower): at M = 2048 (192 workgroups, one wave per SIMD) an iteration
  // is a chain of dependent latencies - fragment read, MFMA, LDS store, barrier - that neither change shortens.
  u32x4 ra[PD][NLD], rb[PD][NLD];
#define DVD_GLOAD(s_, t_)                                                        \
  {                                                                              \
    const int tt_ = (t_);                                                        \
    const bool lo_ = tt_ < nlo;                                                  \
    const size_t kofs_ = (size_t)(lo_ ? tt_ : tt_ - nlo) * BK;                   \
    if constexpr (CONV) {                                                        \
      /* K-tile tt_ = 16 channels of ONE tap of ONE source (ca, cb multiples of 16): wave-uniform tap arithmetic, then a  */ \
      /* border test per slot.  Taps outside the map load the slot's own pixel (every load is issued: counted waits) and  */ \
      /* are zeroed when the tile goes to LDS.                                                                            */ \
      const int cc_ = p.cv_ca + p.cv_cb, cpt_ = cc_ >> 4;                        \
      const int tap_ = tt_ / cpt_, c0_ = (tt_ - tap_ * cpt_) << 4;               \
      const int rad_ = p.cv_ks >> 1;                                             \
      const int dy_ = (tap_ / p.cv_ks - rad_) * p.cv_dil, dx_ = (tap_ % p.cv_ks - rad_) * p.cv_dil; \
      const bool fst_ = c0_ < p.cv_ca;                                           \
      const float* sb_ = fst_ ? (const float*)p.A : p.cv_b;                      \
      const int sc_ = fst_ ? p.cv_ca : p.cv_cb, so_ = fst_ ? c0_ : c0_ - p.cv_ca; \
      _Pragma("unroll") for (int i = 0; i < NLD; ++i) {                          \
        const int yy_ = cy[i] + dy_, xx_ = cx[i] + dx_;                          \
        const bool ok_ = yy_ >= 0 && yy_ < p.cv_h && xx_ >= 0 && xx_ < p.cv_w;    \
        const long q_ = ok_ ? cpix[i] + (long)dy_ * p.cv_w + dx_ : cpix[i];       \
        ra[s_][i] = *(const u32x4*)(sb_ + q_ * sc_ + so_ + cch[i]);              \
        cok[s_][i] = ok_;                                                        \
        rb[s_][i] = *(const u32x4*)(gb[i] + kofs_);                              \
      }                                                                          \
    } else {                                                                     \
      _Pragma("unroll") for (int i = 0; i < NLD; ++i) {                          \
        ra[s_][i] = *(const u32x4*)((lo_ ? gal[i] : ga[i]) + kofs_);             \
        rb[s_][i] = *(const u32x4*)((lo_ ? gbl[i] : gb[i]) + kofs_);             \
      }                                                                          \
    }                                                                            \
  }
#define DVD_LSTORE(s_, buf_)                                                     \
  _Pragma("unroll") for (int i = 0; i < NLD; ++i) {                              \
    if constexpr (CONV) {                                                        \
      const u32x4 z_ = {0u, 0u, 0u, 0u};                                         \
      *(u32x4*)(&smem[buf_][0][lofs[i]]) = cok[s_][i] ? ra[s_][i] : z_;          \
    } else {                                                                     \
      *(u32x4*)(&smem[buf_][0][lofs[i]]) = ra[s_][i];                            \
    }                                                                            \
    *(u32x4*)(&smem[buf_][1][lofs[i]]) = rb[s_][i];                              \
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

  // Loads and LDS stores are issued UNCONDITIONALLY (tile index clamped to the last tile: a few redundant tile loads per
  // workgroup) so that the number of loads in flight is the same on every path and hipcc can emit the counted vmcnt that
  // lets the younger register sets stay in flight across the store.
  // step j: tile j is in LDS[j & 1]; tiles j+1 .. j+PD wait in sets (j+1) % PD .. (j+PD) % PD
  const int last = nk - 1;
#define DVD_STEP(j_, ph_)                                                        \
  {                                                                              \
    DVD_COMPUTE((ph_) & 1)                                                       \
    DVD_LOSCALE(j_)                                                              \
    DVD_LSTORE(((ph_) + 1) % PD, ((ph_) + 1) & 1)                                \
    __syncthreads();                                                             \
    DVD_GLOAD(((ph_) + 1) % PD, min((j_) + 1 + PD, last))                        \
  }
  static_assert(PD == 2 || PD == 4, "the step loop is unrolled by 4");
  DVD_GLOAD(0, 0)
  DVD_GLOAD(1, min(1, last))
  if constexpr (PD == 4) {
    DVD_GLOAD(2 % PD, min(2, last))
    DVD_GLOAD(3 % PD, min(3, last))
  }
  DVD_LSTORE(0, 0)
  __syncthreads();
  DVD_GLOAD(0, min(PD, last))
  int kt = 0;
  for (; kt + 4 <= nk; kt += 4) {
    DVD_STEP(kt, 0)
    DVD_STEP(kt + 1, 1)
    DVD_STEP(kt + 2, 2)
    DVD_STEP(kt + 3, 3)
  }
  if (kt < nk) {
    DVD_STEP(kt, 0)
    if (kt + 1 < nk) {
      DVD_STEP(kt + 1, 1)
      if (kt + 2 < nk) DVD_STEP(kt + 2, 2)
    }
  }
#undef DVD_STEP
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
  if (p.vec_epilogue) {
    // the K loop ended with a barrier (or the tail compute): make sure every wave is done reading operands
    __syncthreads();
    float* stage = (float*)(&smem[0][0][0]) + wave * (64 * 64);     // 4 x 16 KiB of the 72 KiB operand buffers
    DVD_EPILOGUE_BLOCK64(p, stage, acc[0][0], acc[0][1], acc[1][0], acc[1][1], bm0 + 64 * wr, bn0 + 64 * wc, lane, C32,
                         C16, bias, res, gate)
  } else {
#pragma unroll
    for (int n = 0; n < 2; ++n) {
      const int col = bn0 + 64 * wc + 32 * n + r;
      if (col >= p.N) continue;
      const float bcol = (bias && !p.bias_row) ? bias[col] : 0.f;
#pragma unroll
      for (int m = 0; m < 2; ++m)
        epilogue_tile(p, acc[m][n], bm0 + 64 * wr + 32 * m, col, h, bcol, C32, C16, bias, res, gate);
    }
  }
}

#ifdef DVD_LAB
// ================================================================================================
// LAB (DVD_GEMM_W8=1; measured, not adopted: profiles/r5_gemm_w8.txt).  The 128 x 128 tile of gemm_nt_kernel<false> by EIGHT
// waves - two per SIMD: each 64 x 64 block of the tile belongs to a pair of waves, 64 x 32 each, so two chains of dependent
// latencies (fragment read, MFMA, LDS store, barrier) interleave on every SIMD while the tile's global and LDS-store traffic
// stays that of one tile.  Same operands per output in the same order: the bits of gemm_nt_kernel<false>, tested as such
// (test_gemm_eight_wave_kernel).  Epilogue: a pair stages its block into one 64 x 64 LDS region; each wave reads back 32 rows
// of it (epilogue_readback<EPI, 4>).  Result at the sampler's smallest row counts (2048 rows, 192-384 workgroups): 3-4 %
// faster; nothing from 4096 rows on.  With prefetch depth and fragment-read placement also without effect, what bounds the
// 128 x 128 tile there is not a latency chain inside the workgroup: the tile moves 32 KiB per 64-deep step for 2 x 128 x 128
// x 64 FLOP - 64 FLOP per byte - and the kernel draws 8-12 TB/s from L2 at every row count (the guide's L2 streaming peak
// is 17-19 TB/s with nothing else going on).
// ================================================================================================
__global__ void __launch_bounds__(512, 2) gemm_nt_w8_kernel(GemmArgs p) {
  using T = _Float16;
  constexpr int BK = 64, LROW = 128 + 16, CH = 8, NLD = 128 * CH / 512, EPC = 8, SLAB = 128 * LROW;
  __shared__ __attribute__((aligned(16))) char smem[2][2][SLAB];

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
  const int wc = wave & 1, wr = (wave >> 1) & 1, wn = wave >> 2;     // block (wr, wc) of the tile, 32-column half wn of it

  const T* ga[NLD];
  const T* gb[NLD];
  const T* gbl[NLD];
  const T* gal[NLD];
  int lofs[NLD];
#pragma unroll
  for (int i = 0; i < NLD; ++i) {
    const int c = tid + 512 * i;
    const int row = c / CH, ch = c % CH;
    const int ra = min(bm0 + row, p.M - 1), rb = min(bn0 + row, p.N - 1);
    ga[i] = A + (size_t)ra * p.lda + ch * EPC;
    gb[i] = B + (size_t)rb * p.ldb + ch * EPC;
    gbl[i] = Blo ? Blo + (size_t)rb * p.ldb + ch * EPC : gb[i];
    gal[i] = Alo ? Alo + (size_t)ra * p.lda + ch * EPC : ga[i];
    lofs[i] = row * LROW + ch * 16;
  }
  const int nkk = p.K / BK;
  const int nlo = (Blo || Alo) ? nkk : 0;
  const int nk = nkk + nlo;

  u32x4 ra[2][NLD], rb[2][NLD];       // register-staged prefetch, two K-tiles deep, as in gemm_nt_kernel
#define DVD_GLOAD(s_, t_)                                                        \
  {                                                                              \
    const int tt_ = (t_);                                                        \
    const bool lo_ = tt_ < nlo;                                                  \
    const size_t kofs_ = (size_t)(lo_ ? tt_ : tt_ - nlo) * BK;                   \
    _Pragma("unroll") for (int i = 0; i < NLD; ++i) {                            \
      ra[s_][i] = *(const u32x4*)((lo_ ? gal[i] : ga[i]) + kofs_);               \
      rb[s_][i] = *(const u32x4*)((lo_ ? gbl[i] : gb[i]) + kofs_);               \
    }                                                                            \
  }
#define DVD_LSTORE(s_, buf_)                                                     \
  _Pragma("unroll") for (int i = 0; i < NLD; ++i) {                              \
    *(u32x4*)(&smem[buf_][0][lofs[i]]) = ra[s_][i];                              \
    *(u32x4*)(&smem[buf_][1][lofs[i]]) = rb[s_][i];                              \
  }
#define DVD_COMPUTE(buf_)                                                                          \
  {                                                                                                \
    const char* sa = &smem[buf_][0][(64 * wr + r) * LROW];                                         \
    const char* sb = &smem[buf_][1][(64 * wc + 32 * wn + r) * LROW];                               \
    _Pragma("unroll") for (int s = 0; s < 4; ++s) {                                                \
      half8 a[2];                                                                                  \
      _Pragma("unroll") for (int m = 0; m < 2; ++m)                                                \
        a[m] = *(const half8*)(sa + m * 32 * LROW + (16 * s + 8 * h) * 2);                         \
      const half8 b = *(const half8*)(sb + (16 * s + 8 * h) * 2);                                  \
      _Pragma("unroll") for (int m = 0; m < 2; ++m) acc[m] = mfma32_f16(a[m], b, acc[m]);          \
    }                                                                                              \
  }
#define DVD_LOSCALE(kt_)                                                                           \
  if (nlo && (kt_) == nlo - 1) {                                                                   \
    _Pragma("unroll") for (int m = 0; m < 2; ++m)                                                  \
      _Pragma("unroll") for (int i = 0; i < 16; ++i) acc[m][i] *= p.lo_scale;                      \
  }
  floatx16 acc[2];
#pragma unroll
  for (int m = 0; m < 2; ++m)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[m][i] = 0.f;

  const int last = nk - 1;
#define DVD_STEP(j_, ph_)                                                        \
  {                                                                              \
    DVD_COMPUTE((ph_) & 1)                                                       \
    DVD_LOSCALE(j_)                                                              \
    DVD_LSTORE(((ph_) + 1) & 1, ((ph_) + 1) & 1)                                 \
    __syncthreads();                                                             \
    DVD_GLOAD(((ph_) + 1) & 1, min((j_) + 3, last))                              \
  }
  DVD_GLOAD(0, 0)
  DVD_GLOAD(1, min(1, last))
  DVD_LSTORE(0, 0)
  __syncthreads();
  DVD_GLOAD(0, min(2, last))
  int kt = 0;
  for (; kt + 2 <= nk; kt += 2) {
    DVD_STEP(kt, 0)
    DVD_STEP(kt + 1, 1)
  }
  if (kt < nk) DVD_STEP(kt, 0)
#undef DVD_STEP
#undef DVD_GLOAD
#undef DVD_LSTORE
#undef DVD_COMPUTE
#undef DVD_LOSCALE

  float* C32 = p.C32 ? p.C32 + z * p.sC32 : nullptr;
  _Float16* C16 = p.C16 ? p.C16 + z * p.sC16 : nullptr;
  const float* bias = p.bias ? p.bias + z * p.sBias : nullptr;
  const float* res = p.res ? p.res + z * p.sRes : nullptr;
  const float* gate = p.gate ? p.gate + z * p.sGate : nullptr;
  if (p.vec_epilogue) {
    __syncthreads();                               // every wave is done reading operands
    float* stage = (float*)(&smem[0][0][0]) + (wave & 3) * (64 * 64);     // the pair's 64 x 64 region (4 x 16 KiB)
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int rr = cd_row(i, h);
      stage[rr * 64 + 32 * wn + r] = acc[0][i];
      stage[(32 + rr) * 64 + 32 * wn + r] = acc[1][i];
    }
    __syncthreads();                               // both halves of the block are staged
    float* half = stage + wn * (32 * 64);
    const int row0 = bm0 + 64 * wr + 32 * wn, col0 = bn0 + 64 * wc;
    const bool brow_ = bias && p.bias_row;
    if (p.pos || gate || (res && brow_)) epilogue_readback<2, 4>(p, half, row0, col0, lane, C32, C16, bias, res, gate);
    else if (res) epilogue_readback<1, 4>(p, half, row0, col0, lane, C32, C16, bias, res, gate);
    else if (brow_) epilogue_readback<3, 4>(p, half, row0, col0, lane, C32, C16, bias, res, gate);
    else epilogue_readback<0, 4>(p, half, row0, col0, lane, C32, C16, bias, res, gate);
  } else {
    const int col = bn0 + 64 * wc + 32 * wn + r;
    if (col < p.N) {
      const float bcol = (bias && !p.bias_row) ? bias[col] : 0.f;
#pragma unroll
      for (int m = 0; m < 2; ++m) epilogue_tile(p, acc[m], bm0 + 64 * wr + 32 * m, col, h, bcol, C32, C16, bias, res, gate);
    }
  }
}

#endif  // DVD_LAB

// ================================================================================================
// Exact-f32 GEMM for NARROW outputs (N <= 64): the pre-stage conv nets' shape family - U2NETP's convs have 16 or 64 output
// channels (and 1 for the side maps) on maps of 81 ... 82 944 pixels, where the 128 x 128 tile above spends 7/8 of its
// (slow: 1/16 of the f16 rate) f32 MFMAs on padding columns and a 100-row map leaves one workgroup running a 36-step K
// loop alone.  Here a wave owns 32 rows x one or two 32-column MFMA tiles: operands go global -> registers directly (16 bytes per
// lane: lane (r, h) holds k = k0 + 4h .. 4h + 3 of row r - any k assignment is valid as long as A and B agree), 4
// v_mfma_f32_32x32x2_f32 per 8-deep chunk, two chunks in flight; bias (+ReLU) epilogue straight from the accumulators.
// ================================================================================================
template <int NT>   // NT 32-column tiles per wave: N <= 32 * NT
__global__ void __launch_bounds__(256) gemm_f32_narrow_kernel(GemmArgs p) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = lane & 31, h = lane >> 5;
  const int m0 = (blockIdx.x * 4 + wave) * 32;
  if (m0 >= p.M) return;
  const int z = blockIdx.y;
  const float* A = (const float*)p.A + z * p.sA;
  const float* B = (const float*)p.B + z * p.sB;
  const float* ap = A + (size_t)min(m0 + r, p.M - 1) * p.lda + 4 * h;
  const float* bp[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) bp[t] = B + (size_t)min(32 * t + r, p.N - 1) * p.ldb + 4 * h;
  floatx16 acc[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;
  // K % 16 == 0 (checked by the host): 16-deep chunks (two 8-deep halves), chunk n in register set n % NS.  The refill
  // issued beside chunk n's MFMAs is chunk n + PF and goes to the set chunk n - 1 has just left, so PF chunks are in flight
  // and no load targets registers that are still being read (round 5; rounds 2-4 prefetched one chunk through a register
  // copy, which the compiler had to wait for: every chunk cost a memory round trip on the nets' small maps).
  constexpr int PF = 3, NS = PF + 1;
  const int nch = p.K >> 4;
  floatx4 A0[NS], A1[NS], B0[NS][NT], B1[NS][NT];
#define DVD_FETCH(s_, n_)                                                                            \
  {                                                                                                  \
    const int kk_ = min((n_), nch - 1) << 4;                                                         \
    A0[s_] = *(const floatx4*)(ap + kk_);                                                            \
    A1[s_] = *(const floatx4*)(ap + kk_ + 8);                                                        \
    _Pragma("unroll") for (int t = 0; t < NT; ++t) {                                                 \
      B0[s_][t] = *(const floatx4*)(bp[t] + kk_);                                                    \
      B1[s_][t] = *(const floatx4*)(bp[t] + kk_ + 8);                                                \
    }                                                                                                \
  }
#define DVD_CONSUME(s_, n_)                                                                          \
  {                                                                                                  \
    DVD_FETCH(((s_) + PF) % NS, (n_) + PF)                                                           \
    __builtin_amdgcn_sched_barrier(0);   /* the refill is issued HERE, not clustered with later ones */ \
    _Pragma("unroll") for (int e = 0; e < 4; ++e)                                                    \
      _Pragma("unroll") for (int t = 0; t < NT; ++t) acc[t] = mfma32_f32(A0[s_][e], B0[s_][t][e], acc[t]); \
    _Pragma("unroll") for (int e = 0; e < 4; ++e)                                                    \
      _Pragma("unroll") for (int t = 0; t < NT; ++t) acc[t] = mfma32_f32(A1[s_][e], B1[s_][t][e], acc[t]); \
    __builtin_amdgcn_sched_barrier(0);                                                               \
  }
  DVD_FETCH(0, 0)
  DVD_FETCH(1, 1)
  DVD_FETCH(2, 2)
  int n0 = 0;
  for (; n0 + NS <= nch; n0 += NS) {
    DVD_CONSUME(0, n0)
    DVD_CONSUME(1, n0 + 1)
    DVD_CONSUME(2, n0 + 2)
    DVD_CONSUME(3, n0 + 3)
    asm volatile("" ::: "memory");         // keeps the last refill on this side of the back edge
  }
  if (n0 < nch) {
    DVD_CONSUME(0, n0)
    if (n0 + 1 < nch) {
      DVD_CONSUME(1, n0 + 1)
      if (n0 + 2 < nch) DVD_CONSUME(2, n0 + 2)
    }
  }
#undef DVD_CONSUME
#undef DVD_FETCH
  static_assert(NS == 4, "the chunk loop is unrolled by 4");
  // epilogue: every value first (one wait for the bias and the trailing prefetches), then the stores back to back - with the
  // bias add inside the row test the compiler waited for all memory operations, i.e. the previous store, before each store
  float* C = p.C32 + z * p.sC32;
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const int col = 32 * t + r;
    const float bcol = p.bias ? (p.bias + z * p.sBias)[min(col, p.N - 1)] : 0.f;
    float v[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      v[i] = acc[t][i] + bcol;
      v[i] = p.act == 2 ? fmaxf(v[i], 0.f) : v[i];
      asm volatile("" : "+v"(v[i]));       // the value exists before the row tests
    }
    if (col >= p.N) continue;
    float* cp = C + (size_t)m0 * p.ldc + col;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int lr = (i & 3) + 8 * (i >> 2) + 4 * h;
      if (m0 + lr < p.M) cp[(size_t)lr * p.ldc] = v[i];
    }
  }
}

// ================================================================================================
// Large-tile f16 kernel for the decoder GEMMs (92 % of the per-step GEMM FLOPs: M = all tokens, N and K in
// {1536, 2048, 3072}): 256 x 256 tile per 512-thread workgroup (8 waves as 2 x 4, each 128 x 64 = 4 x 2 MFMA
// 32x32 tiles, 128 accumulator registers), K-step 64.
//   * operand tiles go global -> LDS directly (global_load_lds_dwordx4 from inline asm, SGPR base + 32-bit
//     VGPR offset, no staging registers / ds_write), two 64 KiB stages;
//   * LDS rows are 128 B and unpadded (an LDS-DMA image is lane-linear); bank conflicts are removed by the XOR
//     swizzle chunk ^= (row >> 1) & 7 applied to the per-lane SOURCE address and to the fragment reads;
//   * per 16-deep k-step a wave reads 6 fragments for 8 MFMAs (0.75 ds_read_b128 per MFMA instead of 1.0) and
//     the reads of step s+1 are pinned between the MFMAs of step s (sched_barrier) so LDS latency is covered;
//   * one barrier per K-step with 32 MFMAs per wave (2 waves per SIMD) between barriers: 4x the matrix work per
//     synchronisation of the 128 x 128 kernel, whose waves spent > 50 % of their cycles parked (SQ_WAIT_ANY).
// Same split-weight accumulation order and the same epilogue as gemm_nt_kernel.
// ================================================================================================
// one 1-KiB LDS-DMA load (issued between MFMA groups, see the K loop)
__device__ __forceinline__ void glds_one4(const char* gbase, unsigned voff, unsigned lds) {
  unsigned keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\t"
      "s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %3, %1\n\t"
      "s_mov_b32 m0, %0"
      : "=&s"(keep)
      : "s"(gbase), "s"(lds), "v"(voff)
      : "memory", "scc");
}

template <int N>
__device__ __forceinline__ void glds_group4(const char* gbase, const unsigned (&voff)[N], unsigned lds) {
  static_assert(N == 4, "unsupported group size");
  unsigned keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\t"
      "s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %3, %1\n\t"
      "s_add_u32 m0, %2, 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %4, %1\n\t"
      "s_add_u32 m0, %2, 0x800\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %5, %1\n\t"
      "s_add_u32 m0, %2, 0xc00\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %6, %1\n\t"
      "s_mov_b32 m0, %0"
      : "=&s"(keep)
      : "s"(gbase), "s"(lds), "v"(voff[0]), "v"(voff[1]), "v"(voff[2]), "v"(voff[3])
      : "memory", "scc");
}

// ================================================================================================
// gemm_nt_ring128_kernel (round 5): the 128 x 128 tile of gemm_nt_kernel<f16> for problems with FEW tiles - the sampler at the
// reference's operating point: 2048 rows, 192-384 workgroups, one per CU, one wave per SIMD.  There the register-staged
// kernel runs the parts of a K-tile one after the other (PMC: matrix pipe 29 %, parked 28 %, LDS issue 28 %, VALU 12 % of a
// wave's cycles; profiles/r5_gemm_small_pmc.txt) and nothing overlaps them.  Here, with the means of the large kernels:
//   * operand tiles go global -> LDS by LDS-DMA (no staging registers, no ds_write, no address VALU per tile): 64-deep slabs
//     of 2 x 16 KiB, 128-byte rows, chunk ^= (row >> 1) & 7 on the source side and in the fragment reads (gemm_nt_big_kernel);
//   * a ring of FIVE slabs = all 160 KiB: slab kt + 4 is requested at the top of slab kt and has to have landed by the end
//     of slab kt + 2 (three slabs of matrix time for an L2 round trip), the wait is counted (vmcnt(16)), never 0;
//   * the barrier at the end of slab kt publishes slab kt + 2, so slab kt + 1 is complete while slab kt is multiplied:
//     fragments are read TWO k-steps ahead (four register sets), the last two steps of a slab reading the first two of the
//     next one - no wave starts a slab cold;
//   * the slab's eight LDS-DMA pieces are issued one after every pair of MFMAs, not as a burst (see 'Measured' below);
//   * one barrier per slab (16 MFMAs per wave), a bare s_barrier: nothing in flight at it needs an lgkmcnt(0).
// Same operands per output in the same order (low parts, scale, high parts; k ascending in 16-deep steps) and the same
// epilogues: the bits of gemm_nt_kernel<f16>, tested as such.  K % 64 == 0, 16-byte aligned operands and rows.
// Measured (profiles/r5_gemm_ring128.txt): 23-25 % faster than the register-staged kernel up to 256 tiles (one workgroup per
// CU: 29.6 vs 38.7 us at 2048 x 1536 x 1536 with a weight pair), equal from 384 tiles on, where that kernel has two workgroups
// per CU and this one (160 KiB of LDS) one - hence RING128_MAX_TILES.  s_memtime stamps: 855 cycles per slab for 512 of matrix
// work, the same for ONE tile on an idle chip as for 192 - the bound is inside the workgroup: a wave waits at each of its 8
// LDS-DMA instructions until the CU's DMA path accepts the piece (~107 cycles per piece and wave = 38 B/clk per CU of the 47
// the path streams); issued as a burst at the top of the slab those waits cost 1160 cycles per slab, spread between the MFMA
// pairs 855.  Fragment reads one or two k-steps ahead, __syncthreads or a bare s_barrier: no difference.
// ================================================================================================
__global__ void __launch_bounds__(256, 1) gemm_nt_ring128_kernel(GemmArgs p) {
  constexpr int BK = 64, TILE = 128 * 128, SLAB = 2 * TILE, NS = 5;     // bytes: one operand tile, one slab (A | B), ring slots
  extern __shared__ __attribute__((aligned(16))) char smem[];
  typedef __attribute__((address_space(3))) void* lptr_t;
  const int nwg = p.ntm * p.ntn;
  int id = blockIdx.x;
  {
    const int q = nwg / 8, rr = nwg % 8, xcd = id % 8, k = id / 8;       // XCD-aware tile order, as gemm_nt_kernel
    id = (xcd < rr ? xcd * (q + 1) : rr * (q + 1) + (xcd - rr) * q) + k;
  }
  const int tm = id / p.ntn, tn = id % p.ntn;
  const int bm0 = tm * 128, bn0 = tn * 128;
  const int z = blockIdx.y;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 31, h = lane >> 5;
  const int wr = wave >> 1, wc = wave & 1;
  const _Float16* A = (const _Float16*)p.A + z * p.sA;
  const _Float16* B = (const _Float16*)p.B + z * p.sB;
  const _Float16* Alo = p.Alo ? (const _Float16*)p.Alo + z * p.sA : nullptr;
  const _Float16* Blo = p.Blo ? (const _Float16*)p.Blo + z * p.sB : nullptr;

  // per-lane source offsets (bytes, relative to the tile's first row at k = 0) of this wave's 4 + 4 one-KiB pieces (8 rows each)
  unsigned aoff[4], boff[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row = 8 * (4 * wave + i) + (lane >> 3), pos = lane & 7;
    const int logical = pos ^ ((row >> 1) & 7);
    const int ra = min(bm0 + row, p.M - 1) - bm0, rb = min(bn0 + row, p.N - 1) - bn0;
    aoff[i] = (unsigned)(ra * (p.lda * 2) + logical * 16);
    boff[i] = (unsigned)(rb * (p.ldb * 2) + logical * 16);
  }
  const char* Atile = (const char*)(A + (size_t)bm0 * p.lda);
  const char* Btile = (const char*)(B + (size_t)bn0 * p.ldb);
  const char* Alotile = Alo ? (const char*)(Alo + (size_t)bm0 * p.lda) : Atile;
  const char* Blotile = Blo ? (const char*)(Blo + (size_t)bn0 * p.ldb) : Btile;
  const unsigned lds0 = (unsigned)(uintptr_t)(lptr_t)smem;

  int frag[4];
#pragma unroll
  for (int s4 = 0; s4 < 4; ++s4) frag[s4] = r * 128 + (((2 * s4 + h) ^ ((r >> 1) & 7)) * 16);
  const int a_base = wr * 64 * 128;
  const int b_base = TILE + wc * 64 * 128;

  floatx16 acc[2][2];
#pragma unroll
  for (int m = 0; m < 2; ++m)
#pragma unroll
    for (int n = 0; n < 2; ++n)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[m][n][i] = 0.f;
  const int nkk = p.K / BK;
  const int nlo = (Blo || Alo) ? nkk : 0;
  const int nk = nkk + nlo;
  const int last = nk - 1;

#define RING_ISSUE(t_, slot_)                                                                      \
  {                                                                                                \
    const int tt_ = min((t_), last);                                                               \
    const bool lo_ = tt_ < nlo;                                                                    \
    const size_t kb_ = (size_t)(lo_ ? tt_ : tt_ - nlo) * (BK * 2);                                 \
    glds_group4<4>((lo_ ? Alotile : Atile) + kb_, aoff, lds0 + (slot_) * SLAB + (4 * wave) * 1024);        \
    glds_group4<4>((lo_ ? Blotile : Btile) + kb_, boff, lds0 + (slot_) * SLAB + TILE + (4 * wave) * 1024); \
  }
#define SB() __builtin_amdgcn_sched_barrier(0)
#ifdef DVD_LAB
  unsigned long long ts0 = 0, ts1 = 0, ts2 = 0;        // lab: s_memtime stamps (benchmarks/gemm_ring128_stamps.py)
  if (p.stamps) ts0 = __builtin_amdgcn_s_memtime();
#endif
  RING_ISSUE(0, 0)
  RING_ISSUE(1, 1)
  RING_ISSUE(2, 2)
  RING_ISSUE(3, 3)
  asm volatile("s_waitcnt vmcnt(16)" ::: "memory");   // slabs 0 and 1 (this wave's pieces); 2 and 3 stay in flight
  __syncthreads();
#ifdef DVD_LAB
  if (p.stamps) ts1 = __builtin_amdgcn_s_memtime();
#endif
  // k-step s4 of every slab multiplies from register set s4; the reads issued between its MFMAs are those of the step TWO
  // ahead (set (s4 + 2) & 3: this slab's, or the next slab's first two) - 8 MFMAs = 256 cycles between a read and its use,
  // with one wave per SIMD nothing else covers an LDS round trip
  half8 fa[4][2], fb[4][2];
#pragma unroll
  for (int s4 = 0; s4 < 2; ++s4) {
#pragma unroll
    for (int m = 0; m < 2; ++m) fa[s4][m] = *(const half8*)(smem + a_base + m * 32 * 128 + frag[s4]);
#pragma unroll
    for (int n = 0; n < 2; ++n) fb[s4][n] = *(const half8*)(smem + b_base + n * 32 * 128 + frag[s4]);
  }
  SB();
  int cur = 0;
  for (int kt = 0; kt < nk; ++kt) {
    int nxt = cur + 1; if (nxt >= NS) nxt -= NS;
    int free_slot = cur + 4; if (free_slot >= NS) free_slot -= NS;      // slab kt - 1's slot: everyone left it at the last barrier
    // slab kt + 4's eight pieces, ONE after every pair of MFMAs: a piece is accepted every ~90 cycles when the CU's four
    // waves all feed the LDS-DMA path (47 B/clk per CU), and a wave waits at the instruction until it is - issued as a burst
    // at the top of the slab they cost 700 cycles in front of its 512 cycles of MFMAs (stamps: 1160 per slab, whatever the
    // problem size, even for a single tile); spread, the matrix pipe works through the waits
    const int tt_n = min(kt + 4, last);
    const bool lo_n = tt_n < nlo;
    const size_t kb_n = (size_t)(lo_n ? tt_n : tt_n - nlo) * (BK * 2);
    const char* a_n = (lo_n ? Alotile : Atile) + kb_n;
    const char* b_n = (lo_n ? Blotile : Btile) + kb_n;
    const unsigned lds_n = lds0 + free_slot * SLAB + (4 * wave) * 1024;
    const char* base = smem + cur * SLAB;
    const char* nbase = smem + nxt * SLAB;
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4) {
      const int ns = (s4 + 2) & 3;
      const char* rb = s4 < 2 ? base : nbase;
#pragma unroll
      for (int m = 0; m < 2; ++m) {
#pragma unroll
        for (int n = 0; n < 2; ++n) acc[m][n] = mfma32_f16(fa[s4][m], fb[s4][n], acc[m][n]);
        const int pc = 2 * s4 + m;                   // piece 0..7 of the next slab: A 0-3, B 0-3
        if (pc < 4) glds_one4(a_n, aoff[pc], lds_n + pc * 1024);
        else glds_one4(b_n, boff[pc - 4], lds_n + TILE + (pc - 4) * 1024);
        SB();
      }
      // (after the step's MFMAs have been issued: its own operands are dead, set ns was last read two steps ago)
#pragma unroll
      for (int m = 0; m < 2; ++m) {
        fa[ns][m] = *(const half8*)(rb + a_base + m * 32 * 128 + frag[ns]);
        fb[ns][m] = *(const half8*)(rb + b_base + m * 32 * 128 + frag[ns]);
      }
      SB();
    }
    if (nlo && kt == nlo - 1) {
#pragma unroll
      for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
          for (int i = 0; i < 16; ++i) acc[m][n][i] *= p.lo_scale;
    }
    // slab kt + 2 landed (kt + 3 and kt + 4 stay in flight) ... for everyone; and everyone is past slab kt.  A bare s_barrier:
    // the fragment reads still in flight here are slab kt + 1's (its slot is not reused before the NEXT barrier), so there is
    // nothing for an lgkmcnt(0) - which __syncthreads() would add, one exposed LDS round trip per slab - to protect
    asm volatile("s_waitcnt vmcnt(16)\n\ts_barrier" ::: "memory");
    cur = nxt;
  }
#undef RING_ISSUE
#undef SB
#ifdef DVD_LAB
  if (p.stamps) ts2 = __builtin_amdgcn_s_memtime();
#endif
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the clamped tail requests: nobody may still be writing LDS
  __syncthreads();

  // ---------------- epilogue: gemm_nt_kernel's ----------------
  float* C32 = p.C32 ? p.C32 + z * p.sC32 : nullptr;
  _Float16* C16 = p.C16 ? p.C16 + z * p.sC16 : nullptr;
  const float* bias = p.bias ? p.bias + z * p.sBias : nullptr;
  const float* res = p.res ? p.res + z * p.sRes : nullptr;
  const float* gate = p.gate ? p.gate + z * p.sGate : nullptr;
  if (p.vec_epilogue) {
    float* stage = (float*)smem + wave * (64 * 64);
    DVD_EPILOGUE_BLOCK64(p, stage, acc[0][0], acc[0][1], acc[1][0], acc[1][1], bm0 + 64 * wr, bn0 + 64 * wc, lane, C32,
                         C16, bias, res, gate)
  } else {
#pragma unroll
    for (int n = 0; n < 2; ++n) {
      const int col = bn0 + 64 * wc + 32 * n + r;
      if (col >= p.N) continue;
      const float bcol = (bias && !p.bias_row) ? bias[col] : 0.f;
#pragma unroll
      for (int m = 0; m < 2; ++m)
        epilogue_tile(p, acc[m][n], bm0 + 64 * wr + 32 * m, col, h, bcol, C32, C16, bias, res, gate);
    }
  }
#ifdef DVD_LAB
  if (p.stamps && lane == 0) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    unsigned long long* o = p.stamps + ((size_t)blockIdx.x * 4 + wave) * 8;
    o[0] = ts0; o[1] = ts1; o[2] = ts2; o[3] = __builtin_amdgcn_s_memtime();
  }
#endif
}

// ================================================================================================
// gemm_nt_ring256_kernel (round 5): gemm_nt_ring128_kernel's scheme on a 128 x 256 tile with EIGHT waves (2 x 4, 64 x 64 each,
// two per SIMD), for the same few-tile problems when N % 256 == 0.  ring128 is bound by the rate at which a CU's LDS-DMA
// path accepts its tile's pieces (32 KiB per 64-deep slab for 512 cycles of matrix work per SIMD); this tile needs 48 KiB for
// twice the work - 25 % fewer bytes per FLOP - and two waves per SIMD cover each other's waits.  K in 32-deep half slabs
// (A 8 KiB | B 16 KiB: 24 one-KiB pieces of 16 rows x 64 B, three per wave) through a ring of SIX = 144 KiB: half slab j + 4
// is requested during half slab j (one piece after each of its first three MFMA pairs) and has to have landed by the end
// of j + 2 (counted wait: vmcnt(6)); the barrier at the end of j publishes j + 2, so j + 1 is complete while j is
// multiplied and its first fragments are read under j's last MFMAs.  64-byte LDS rows, chunk ^= (row >> 2) & 3 on the
// source side and in the fragment reads (gemm_nt_split128_kernel's image).  Same operands per output in the same order
// (low parts, scale, high parts; k ascending in 16-deep steps) and gemm_nt_kernel's epilogues: its bits, tested as such.
// ================================================================================================
__global__ void __launch_bounds__(512, 1) gemm_nt_ring256_kernel(GemmArgs p) {
  constexpr int BK = 32, TA = 128 * 64, TB = 256 * 64, SLOT = TA + TB, NS = 6;     // bytes
  extern __shared__ __attribute__((aligned(16))) char smem[];
  typedef __attribute__((address_space(3))) void* lptr_t;
  const int nwg = p.ntm * p.ntn;
  int id = blockIdx.x;
  {
    const int q = nwg / 8, rr = nwg % 8, xcd = id % 8, k = id / 8;       // XCD-aware tile order, as gemm_nt_kernel
    id = (xcd < rr ? xcd * (q + 1) : rr * (q + 1) + (xcd - rr) * q) + k;
  }
  const int tm = id / p.ntn, tn = id % p.ntn;
  const int bm0 = tm * 128, bn0 = tn * 256;
  const int z = blockIdx.y;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 31, h = lane >> 5;
  const int wr = wave >> 2, wc = wave & 3;
  const _Float16* A = (const _Float16*)p.A + z * p.sA;
  const _Float16* B = (const _Float16*)p.B + z * p.sB;
  const _Float16* Alo = p.Alo ? (const _Float16*)p.Alo + z * p.sA : nullptr;
  const _Float16* Blo = p.Blo ? (const _Float16*)p.Blo + z * p.sB : nullptr;

  // per-lane source offsets of this wave's pieces (16 rows x 64 B each): A piece `wave`, B pieces 2 wave and 2 wave + 1
  unsigned aoff, boff[2];
  {
    const int row = 16 * wave + (lane >> 2), pos = lane & 3;
    const int ra = min(bm0 + row, p.M - 1) - bm0;
    aoff = (unsigned)(ra * (p.lda * 2) + (pos ^ ((row >> 2) & 3)) * 16);
  }
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int row = 16 * (2 * wave + i) + (lane >> 2), pos = lane & 3;
    const int rb = min(bn0 + row, p.N - 1) - bn0;
    boff[i] = (unsigned)(rb * (p.ldb * 2) + (pos ^ ((row >> 2) & 3)) * 16);
  }
  const char* Atile = (const char*)(A + (size_t)bm0 * p.lda);
  const char* Btile = (const char*)(B + (size_t)bn0 * p.ldb);
  const char* Alotile = Alo ? (const char*)(Alo + (size_t)bm0 * p.lda) : Atile;
  const char* Blotile = Blo ? (const char*)(Blo + (size_t)bn0 * p.ldb) : Btile;
  const unsigned lds0 = (unsigned)(uintptr_t)(lptr_t)smem;
  const unsigned lds_a = lds0 + wave * 1024, lds_b = lds0 + TA + (2 * wave) * 1024;

  int frag[2];
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) frag[ks] = r * 64 + (((2 * ks + h) ^ ((r >> 2) & 3)) * 16);
  const int a_base = wr * 64 * 64;
  const int b_base = TA + wc * 64 * 64;

  floatx16 acc[2][2];
#pragma unroll
  for (int m = 0; m < 2; ++m)
#pragma unroll
    for (int n = 0; n < 2; ++n)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[m][n][i] = 0.f;
  const int nkk = p.K / BK;
  const int nlo = (Blo || Alo) ? nkk : 0;
  const int nk = nkk + nlo;
  const int last = nk - 1;

#define R256_SRC(t_)                                                                               \
  const int tt_ = min((t_), last);                                                                 \
  const bool lo_ = tt_ < nlo;                                                                      \
  const size_t kb_ = (size_t)(lo_ ? tt_ : tt_ - nlo) * (BK * 2);                                   \
  const char* a_s = (lo_ ? Alotile : Atile) + kb_;                                                 \
  const char* b_s = (lo_ ? Blotile : Btile) + kb_;
#define SB() __builtin_amdgcn_sched_barrier(0)
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    R256_SRC(t)
    glds_one4(a_s, aoff, lds_a + t * SLOT);
    glds_one4(b_s, boff[0], lds_b + t * SLOT);
    glds_one4(b_s, boff[1], lds_b + t * SLOT + 1024);
  }
  asm volatile("s_waitcnt vmcnt(6)" ::: "memory");    // half slabs 0 and 1 (this wave's pieces); 2 and 3 stay in flight
  __syncthreads();
  // k-step ks of every half slab multiplies from register set ks; the reads issued between its MFMAs are the next half
  // slab's same step (one half slab = 8 MFMAs ahead)
  half8 fa[2][2], fb[2][2];
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
    for (int m = 0; m < 2; ++m) fa[ks][m] = *(const half8*)(smem + a_base + m * 32 * 64 + frag[ks]);
#pragma unroll
    for (int n = 0; n < 2; ++n) fb[ks][n] = *(const half8*)(smem + b_base + n * 32 * 64 + frag[ks]);
  }
  SB();
  int cur = 0;
  for (int kt = 0; kt < nk; ++kt) {
    int nxt = cur + 1; if (nxt >= NS) nxt -= NS;
    int free_slot = cur + 4; if (free_slot >= NS) free_slot -= NS;      // half slab kt - 2's slot
    R256_SRC(kt + 4)
    const unsigned la = lds_a + free_slot * SLOT, lb = lds_b + free_slot * SLOT;
    const char* nbase = smem + nxt * SLOT;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
      for (int m = 0; m < 2; ++m) {
#pragma unroll
        for (int n = 0; n < 2; ++n) acc[m][n] = mfma32_f16(fa[ks][m], fb[ks][n], acc[m][n]);
        const int pc = 2 * ks + m;                   // one piece of half slab kt + 4 after each of the first three MFMA pairs
        if (pc == 0) glds_one4(a_s, aoff, la);
        else if (pc == 1) glds_one4(b_s, boff[0], lb);
        else if (pc == 2) glds_one4(b_s, boff[1], lb + 1024);
        SB();
      }
      // this step's operands are dead: the same step of the NEXT half slab (published at the last barrier) into their set
#pragma unroll
      for (int m = 0; m < 2; ++m) {
        fa[ks][m] = *(const half8*)(nbase + a_base + m * 32 * 64 + frag[ks]);
        fb[ks][m] = *(const half8*)(nbase + b_base + m * 32 * 64 + frag[ks]);
      }
      SB();
    }
    if (nlo && kt == nlo - 1) {
#pragma unroll
      for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
          for (int i = 0; i < 16; ++i) acc[m][n][i] *= p.lo_scale;
    }
    // half slab kt + 2 landed (kt + 3 and kt + 4 stay in flight) ... for everyone; and everyone is past half slab kt.  The
    // fragment reads in flight here are half slab kt + 1's (its slot is reused after the barrier at the end of kt + 1 at
    // the earliest, by which time they have been consumed): a bare s_barrier
    asm volatile("s_waitcnt vmcnt(6)\n\ts_barrier" ::: "memory");
    cur = nxt;
  }
#undef R256_SRC
#undef SB
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the clamped tail requests: nobody may still be writing LDS
  __syncthreads();

  // ---------------- epilogue: gemm_nt_kernel's ----------------
  float* C32 = p.C32 ? p.C32 + z * p.sC32 : nullptr;
  _Float16* C16 = p.C16 ? p.C16 + z * p.sC16 : nullptr;
  const float* bias = p.bias ? p.bias + z * p.sBias : nullptr;
  const float* res = p.res ? p.res + z * p.sRes : nullptr;
  const float* gate = p.gate ? p.gate + z * p.sGate : nullptr;
  if (p.vec_epilogue) {
    float* stage = (float*)smem + wave * (64 * 64);                  // 8 x 16 KiB of the 144 KiB
    DVD_EPILOGUE_BLOCK64(p, stage, acc[0][0], acc[0][1], acc[1][0], acc[1][1], bm0 + 64 * wr, bn0 + 64 * wc, lane, C32,
                         C16, bias, res, gate)
  } else {
#pragma unroll
    for (int n = 0; n < 2; ++n) {
      const int col = bn0 + 64 * wc + 32 * n + r;
      if (col >= p.N) continue;
      const float bcol = (bias && !p.bias_row) ? bias[col] : 0.f;
#pragma unroll
      for (int m = 0; m < 2; ++m)
        epilogue_tile(p, acc[m][n], bm0 + 64 * wr + 32 * m, col, h, bcol, C32, C16, bias, res, gate);
    }
  }
}

template <int DBG>   // DBG: timing ablations (DVD_GEMM_DEBUG) 1 = no operand loads in the K loop, 2 = no MFMAs, 5 = no fragment
                     // reads inside a slab, 6 = no end-of-slab wait + barrier, 7 = MFMAs only (1 + 5 + 6): profiles/r5_gemm_ablation.txt
__global__ void __launch_bounds__(512, 2) gemm_nt_big_kernel(GemmArgs p) {
  constexpr int BK = 64, TILE = 256 * 128;   // bytes of one operand tile (256 rows x 64 halfs)
  extern __shared__ __attribute__((aligned(16))) char smem[];   // [2 stages][A | B]
  typedef __attribute__((address_space(3))) void* lptr_t;

  // PERSISTENT: gridDim.x <= 256 workgroups (one per CU, a multiple of 8 so a workgroup stays on "its" XCD
  // group) walk the tile list.  The epilogue's global stores are fire-and-forget, so they drain while the same
  // workgroup is already loading / multiplying its next tile; with one workgroup per tile every CU ran K loop and
  // (HBM-write-bound, 25 % of its time) epilogue in lockstep and the two phases never overlapped.
  const int nwg = p.ntm * p.ntn;
  const int z = blockIdx.y;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 31, h = lane >> 5;
  const int wr = wave >> 2, wc = wave & 3;
  for (int vid = blockIdx.x; vid < nwg; vid += gridDim.x) {
  int tm, tn;
  tile_coords(vid, p.ntm, p.ntn, tm, tn);
  tm = __builtin_amdgcn_readfirstlane(tm); tn = __builtin_amdgcn_readfirstlane(tn);   // wave-uniform: keep in SGPRs
  const int bm0 = tm * 256, bn0 = tn * 256;
  const _Float16* A = (const _Float16*)p.A + z * p.sA;
  const _Float16* B = (const _Float16*)p.B + z * p.sB;
  const _Float16* Alo = p.Alo ? (const _Float16*)p.Alo + z * p.sA : nullptr;
  const _Float16* Blo = p.Blo ? (const _Float16*)p.Blo + z * p.sB : nullptr;

  // per-lane source offsets (bytes, relative to the tile's first row at k = 0) of this wave's 4 + 4 loads
  unsigned aoff[4], boff[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row = 8 * (4 * wave + i) + (lane >> 3), pos = lane & 7;
    const int logical = pos ^ ((row >> 1) & 7);
    const int ra = min(bm0 + row, p.M - 1) - bm0, rb = min(bn0 + row, p.N - 1) - bn0;
    aoff[i] = (unsigned)ra * (unsigned)(p.lda * 2) + logical * 16;
    boff[i] = (unsigned)rb * (unsigned)(p.ldb * 2) + logical * 16;
  }
  const char* Atile = (const char*)(A + (size_t)bm0 * p.lda);
  const char* Btile = (const char*)(B + (size_t)bn0 * p.ldb);
  const char* Alotile = Alo ? (const char*)(Alo + (size_t)bm0 * p.lda) : Atile;
  const char* Blotile = Blo ? (const char*)(Blo + (size_t)bn0 * p.ldb) : Btile;
  const unsigned lds0 = (unsigned)(uintptr_t)(lptr_t)smem;

  // per-lane fragment read offsets: row r of a 32-row block, k-step s -> chunk (2s + h) ^ ((r >> 1) & 7)
  int frag[4];
#pragma unroll
  for (int s4 = 0; s4 < 4; ++s4) frag[s4] = r * 128 + (((2 * s4 + h) ^ ((r >> 1) & 7)) * 16);
  const int a_base = wr * 128 * 128;            // this wave's first A row, bytes
  const int b_base = TILE + wc * 64 * 128;      // this wave's first B row

  floatx16 acc[4][2];
#pragma unroll
  for (int m = 0; m < 4; ++m)
#pragma unroll
    for (int n = 0; n < 2; ++n)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[m][n][i] = 0.f;

  const int nkk = p.K / BK;
  const int nlo = (Blo || Alo) ? nkk : 0;
  const int nk = nkk + nlo;
  // (a per-workgroup K rotation was tried against suspected L2-channel hot-spotting: no gain, and it makes the
  //  fp32 summation order depend on the tile id, so it is not used)

#define BIG_ISSUE(t_, stage_)                                                                      \
  {                                                                                                \
    const int tt_ = (t_);                                                                          \
    const bool lo_ = tt_ < nlo;                                                                    \
    const size_t kb_ = (size_t)(lo_ ? tt_ : tt_ - nlo) * (BK * 2);                                 \
    glds_group4<4>((lo_ ? Alotile : Atile) + kb_, aoff, lds0 + (stage_) * 2 * TILE + (4 * wave) * 1024);        \
    glds_group4<4>((lo_ ? Blotile : Btile) + kb_, boff, lds0 + (stage_) * 2 * TILE + TILE + (4 * wave) * 1024); \
  }
#define SB() __builtin_amdgcn_sched_barrier(0)
#define LDFRAG(set_, s4_)                                                                          \
  {                                                                                                \
    _Pragma("unroll") for (int m = 0; m < 4; ++m)                                                  \
      fa[set_][m] = *(const half8*)(base + a_base + m * 32 * 128 + frag[s4_]);                     \
    _Pragma("unroll") for (int n = 0; n < 2; ++n)                                                  \
      fb[set_][n] = *(const half8*)(base + b_base + n * 32 * 128 + frag[s4_]);                     \
  }

  // De-synchronise the chip: with one workgroup per CU every CU would otherwise run its K loop and then its
  // (HBM-write-bound) epilogue in lockstep, so the output bursts of all 256 CUs collide while HBM idles during
  // the K loops.  The first round of workgroups starts with a different delay per CU; later rounds stay spread.
  if (p.stagger && blockIdx.x < 256) {
    const int slots = (blockIdx.x * 5) & 15;
    for (int i = 0; i < slots * p.stagger; ++i) __builtin_amdgcn_s_sleep(64);   // 64 * 64 cycles each
  }
  unsigned long long t0 = 0, t1 = 0, t2 = 0;
  if constexpr (DBG == 3) t0 = __builtin_amdgcn_s_memtime();
  BIG_ISSUE(0, 0)
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if constexpr (DBG == 3) t1 = __builtin_amdgcn_s_memtime();
  int cur = 0;
  // Next-slab loads: all 8 of a wave in a burst at the top of the slab (default; with two waves per SIMD the burst of
  // one wave hides under the MFMAs of the other).  DBG 4 (DVD_GEMM_SPREAD=1) spreads them over the slab instead (one
  // 1-KiB load after every second group of two MFMAs): measured 3 % SLOWER here (811 vs 840 TF/s at N=K=1536), unlike
  // the one-wave-per-SIMD attention kernel where spreading gained 10 %.
  constexpr bool SPREAD = (DBG == 4);
  for (int kt = 0; kt < nk; ++kt) {
    // waves 0-3 (one per SIMD) issue their burst here, waves 4-7 after the first of the four k-steps: the two waves of
    // a SIMD then do not sit in their issue stalls at the same time
    if (!SPREAD && DBG != 1 && DBG != 7 && kt + 1 < nk && wave < 4) BIG_ISSUE(kt + 1, cur ^ 1)
    const int tnx = min(kt + 1, nk - 1);
    const bool lo_nx = tnx < nlo;
    const size_t kb_nx = (size_t)(lo_nx ? tnx : tnx - nlo) * (BK * 2);
    const char* a_nx = (lo_nx ? Alotile : Atile) + kb_nx;
    const char* b_nx = (lo_nx ? Blotile : Btile) + kb_nx;
    const unsigned lds_nx = lds0 + (cur ^ 1) * 2 * TILE + (4 * wave) * 1024;
    const char* base = smem + cur * 2 * TILE;
    if constexpr (DBG == 2) {   // ablation: loads + barrier only
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      cur ^= 1;
      continue;
    }
    half8 fa[2][4], fb[2][2];
    LDFRAG(0, 0)
    if constexpr (DBG == 5 || DBG == 7) { LDFRAG(1, 1) }   // ablation: both register sets filled once per slab, none inside
    SB();
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4) {
      const int cs = s4 & 1;
      // 8 MFMAs of k-step s4; the 6 fragment reads of step s4+1 are spread between them
#pragma unroll
      for (int m = 0; m < 4; ++m) {
#pragma unroll
        for (int n = 0; n < 2; ++n) acc[m][n] = mfma32_f16(fa[cs][m], fb[cs][n], acc[m][n]);
        if (s4 < 3 && DBG != 5 && DBG != 7) {
          fa[cs ^ 1][m] = *(const half8*)(base + a_base + m * 32 * 128 + frag[s4 + 1]);
          if (m < 2) fb[cs ^ 1][m] = *(const half8*)(base + b_base + m * 32 * 128 + frag[s4 + 1]);
        }
        if constexpr (SPREAD) {
          const int g = s4 * 4 + m;                       // 16 MFMA groups per slab; loads after groups SPREAD_AT(k)
          constexpr int FIRST = DVD_GEMM_SPREAD_FIRST, STEP = DVD_GEMM_SPREAD_STEP;
          if (g >= FIRST && (g - FIRST) % STEP == 0 && (g - FIRST) / STEP < 8) {
            const int k = (g - FIRST) / STEP;
            if (k < 4) glds_one4(a_nx, aoff[k & 3], lds_nx + (k & 3) * 1024);
            else glds_one4(b_nx, boff[k & 3], lds_nx + TILE + (k & 3) * 1024);
          }
        }
        SB();
      }
      if (s4 == 0) {
        if (!SPREAD && DBG != 1 && DBG != 7 && kt + 1 < nk && wave >= 4) BIG_ISSUE(kt + 1, cur ^ 1)
        SB();
      }
    }
    if (nlo && kt == nlo - 1) {
#pragma unroll
      for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
          for (int i = 0; i < 16; ++i) acc[m][n][i] *= p.lo_scale;
    }
    if constexpr (DBG != 6 && DBG != 7) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
    }
    cur ^= 1;
  }
#undef BIG_ISSUE
#undef SB
#undef LDFRAG

  if constexpr (DBG == 3) t2 = __builtin_amdgcn_s_memtime();
  unsigned long long te[3] = {0, 0, 0};
  // ---------------- epilogue (same semantics as gemm_nt_kernel) ----------------
  float* C32 = p.C32 ? p.C32 + z * p.sC32 : nullptr;
  _Float16* C16 = p.C16 ? p.C16 + z * p.sC16 : nullptr;
  const float* bias = p.bias ? p.bias + z * p.sBias : nullptr;
  const float* res = p.res ? p.res + z * p.sRes : nullptr;
  const float* gate = p.gate ? p.gate + z * p.sGate : nullptr;
  // the large-tile kernel carries only the two staged flavours its callers need (plain / residual); the rare pos /
  // gate / row-bias epilogues at N % 256 == 0 take the scalar path below (fewer live values across the K loop)
  if (p.vec_epilogue && !p.pos && !gate && !(bias && p.bias_row)) {
    float* stage = (float*)smem + wave * (64 * 64);                  // 8 x 16 KiB = the whole 128 KiB
    const int row0 = bm0 + 128 * wr, col0 = bn0 + 64 * wc;
    // Launder the lane id: everything the epilogue derives from it (staging offsets, row / column of a lane) is then
    // recomputed here instead of being hoisted above the K loop of the persistent tile loop, kept live across it and
    // SPILLED - each scratch reload is a serialized ~500-cycle vmcnt(0) wait (24 of them: 12 000 cycles per tile).
    int lane_e = lane;
    asm volatile("" : "+v"(lane_e));
#define lane lane_e
    if (res) {
      epilogue_block64<1>(p, stage, acc[0][0], acc[0][1], acc[1][0], acc[1][1], row0, col0, lane, C32, C16, bias, res,
                          gate, DBG == 3 ? &te[0] : nullptr);
      if constexpr (DBG == 3) te[1] = __builtin_amdgcn_s_memtime();
      epilogue_block64<1>(p, stage, acc[2][0], acc[2][1], acc[3][0], acc[3][1], row0 + 64, col0, lane, C32, C16, bias,
                          res, gate, DBG == 3 ? &te[2] : nullptr);
    } else {
      epilogue_block64<0>(p, stage, acc[0][0], acc[0][1], acc[1][0], acc[1][1], row0, col0, lane, C32, C16, bias, res,
                          gate, DBG == 3 ? &te[0] : nullptr);
      if constexpr (DBG == 3) te[1] = __builtin_amdgcn_s_memtime();
      epilogue_block64<0>(p, stage, acc[2][0], acc[2][1], acc[3][0], acc[3][1], row0 + 64, col0, lane, C32, C16, bias,
                          res, gate, DBG == 3 ? &te[2] : nullptr);
    }
#undef lane
  } else {
  // written out (not a loop): hipcc refuses to fully unroll an 8 x 16-element epilogue loop and would then
  // index acc[][] dynamically
#define BIG_EP(m_, n_)                                                                                    \
  {                                                                                                       \
    const int col = bn0 + 64 * wc + 32 * (n_) + r;                                                        \
    if (col < p.N) {                                                                                      \
      const float bcol = (bias && !p.bias_row) ? bias[col] : 0.f;                                         \
      epilogue_tile(p, acc[m_][n_], bm0 + 128 * wr + 32 * (m_), col, h, bcol, C32, C16, bias, res, gate); \
    }                                                                                                     \
  }
  BIG_EP(0, 0) BIG_EP(0, 1) BIG_EP(1, 0) BIG_EP(1, 1) BIG_EP(2, 0) BIG_EP(2, 1) BIG_EP(3, 0) BIG_EP(3, 1)
#undef BIG_EP
  }
  if constexpr (DBG == 3) {
    const unsigned long long t3 = __builtin_amdgcn_s_memtime();      // stores issued, not yet drained
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long t4 = __builtin_amdgcn_s_memtime();      // stores drained
    if (lane == 0 && p.stamps && vid < 256 * 64) {
      unsigned long long* o = p.stamps + ((size_t)vid * 8 + wave) * 8;
      o[0] = t0; o[1] = t1; o[2] = t2; o[3] = t3; o[4] = t4; o[5] = te[0]; o[6] = te[1]; o[7] = te[2];
    }
  }
  __syncthreads();   // every wave has read its staging region back: the next tile's LDS-DMA may overwrite it
  }  // persistent tile loop
}


#ifdef DVD_LAB

// ================================================================================================
// Round 4: the 256 x 256 tile with ONE wave per SIMD ("big2").  Four waves (2 x 2), each 128 x 128 of the tile = 4 x 4
// accumulators = all 256 AGPRs, so a k-step reads 8 fragments for 16 MFMAs (0.5 ds_read_b128 per MFMA; the 8-wave kernel:
// 0.75) and nothing competes with the wave for its SIMD.  What the second wave per SIMD bought the 8-wave kernel - cover
// for LDS latency, LDS-DMA issue and the barrier - is bought here by the schedule, pinned like the attention kernel's
// (flash_attn_r64p_kernel, attention.hip): the K loop is a sequence of `asm volatile` statements of four MFMAs each, with
// the fragment reads of the NEXT half slab and the LDS-DMA pieces of the half slab three ahead in their gaps.
//   * K is walked in HALF slabs of 32 (two k-steps, 32 MFMAs per wave): [A 256 x 32 | B 256 x 32] = 32 KiB, a ring of four
//     buffers.  LDS rows are 64 B, 16-byte chunk c of row r at c ^ ((r >> 2) & 3) (conflict-free ds_read_b128, the
//     attention kernel's V^T image); one LDS-DMA piece = 16 rows.
//   * iteration j (half slab j in registers): 8 MFMAs | vmcnt(8) + THE barrier | 24 MFMAs with the 16 fragment reads of
//     half slab j + 1 and this wave's 8 pieces of half slab j + 3.  At the barrier half slab j + 1 has landed (its pieces
//     were issued two iterations ago) and every wave has left iteration j - 1, whose buffer the new pieces overwrite.  The
//     LDS-DMA queue is never drained inside the K loop (counted vmcnt), the LDS queue is waited for with counted lgkmcnt.
// Plain and residual epilogues through the shared LDS staging (epilogue_block64), tile walk of the 8-wave kernel.  One
// weight tensor only (the dithered weights of large grids); a (hi, lo) pair takes the split kernels.
// MEASURED (MI355X, benchmarks/gemm_time.py, same process order A/B/A/B, M = 331 776; profiles/archive/r4_gemm_big2.txt) and
// REJECTED: 943-961 TF/s at K = 1536 (N = 3072 / 2048 / 1536) and 1031-1036 at K = 2048 against 1015-1041 and 1077-1081
// for the 8-wave kernel: 4-9 % SLOWER, bit-identical results.  The K loop is not where the 8-wave kernel loses: a 256 x 256
// tile needs 32 KiB of operands per 1024 matrix cycles = 32 B/clk, the CU's whole L2 -> LDS rate, whatever
// the wave layout; what the one-wave layout adds is an epilogue run by four waves instead of eight (twice the staging
// and store work per wave, half the stores in flight) and a per-tile prologue nobody hides.  LAB ONLY (DVD_GEMM_BIG2=1).
// ================================================================================================
namespace big2 {
#define B2_MF "v_mfma_f32_32x32x16_f16 "
#define B2_RD(i_) "ds_read_b128 %[n" #i_ "], %[addr] offset:%[o" #i_ "]\n\t"
#define B2_DMA(i_) "s_mov_b32 m0, %[l" #i_ "]\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %[v" #i_ "], %[gb]\n\t"
// four MFMAs of one accumulator row: acc[n] += a . b[n]
__device__ __forceinline__ void mm4(floatx16& c0, floatx16& c1, floatx16& c2, floatx16& c3, const half8& a, const half8& b0,
                                    const half8& b1, const half8& b2, const half8& b3) {
  asm volatile(B2_MF "%[c0], %[a], %[b0], %[c0]\n\t" B2_MF "%[c1], %[a], %[b1], %[c1]\n\t" B2_MF "%[c2], %[a], %[b2], %[c2]\n\t"
               B2_MF "%[c3], %[a], %[b3], %[c3]"
               : [c0] "+a"(c0), [c1] "+a"(c1), [c2] "+a"(c2), [c3] "+a"(c3)
               : [a] "v"(a), [b0] "v"(b0), [b1] "v"(b1), [b2] "v"(b2), [b3] "v"(b3));
}
// the same behind a counted LDS wait
template <int LG>
__device__ __forceinline__ void mm4_wait(floatx16& c0, floatx16& c1, floatx16& c2, floatx16& c3, const half8& a, const half8& b0,
                                         const half8& b1, const half8& b2, const half8& b3) {
  asm volatile("s_waitcnt lgkmcnt(%[lg])\n\t" B2_MF "%[c0], %[a], %[b0], %[c0]\n\t" B2_MF "%[c1], %[a], %[b1], %[c1]\n\t"
               B2_MF "%[c2], %[a], %[b2], %[c2]\n\t" B2_MF "%[c3], %[a], %[b3], %[c3]"
               : [c0] "+a"(c0), [c1] "+a"(c1), [c2] "+a"(c2), [c3] "+a"(c3)
               : [a] "v"(a), [b0] "v"(b0), [b1] "v"(b1), [b2] "v"(b2), [b3] "v"(b3), [lg] "i"(LG));
}
// four MFMAs, each followed by one fragment read of the next half slab (four 32-row blocks of ONE operand and k-step:
// one address register, immediates 0 / 2048 / 4096 / 6144), and two LDS-DMA pieces (after the 2nd and the 4th MFMA)
template <int LG>   // LG >= 0: counted LDS wait in front
__device__ __forceinline__ void mm4_rd4_dma2(floatx16& c0, floatx16& c1, floatx16& c2, floatx16& c3, const half8& a,
                                             const half8& b0, const half8& b1, const half8& b2, const half8& b3, half8& n0,
                                             half8& n1, half8& n2, half8& n3, unsigned addr, const char* gb, unsigned l0,
                                             unsigned v0, unsigned l1, unsigned v1) {
  if constexpr (LG >= 0) {
    asm volatile("s_waitcnt lgkmcnt(%[lg])\n\t" B2_MF "%[c0], %[a], %[b0], %[c0]\n\t" B2_RD(0) B2_MF "%[c1], %[a], %[b1], %[c1]\n\t"
                 B2_RD(1) B2_DMA(0) B2_MF "%[c2], %[a], %[b2], %[c2]\n\t" B2_RD(2) B2_MF "%[c3], %[a], %[b3], %[c3]\n\t" B2_RD(3)
                 B2_DMA(1)
                 : [c0] "+a"(c0), [c1] "+a"(c1), [c2] "+a"(c2), [c3] "+a"(c3), [n0] "=&v"(n0), [n1] "=&v"(n1), [n2] "=&v"(n2),
                   [n3] "=&v"(n3)
                 : [a] "v"(a), [b0] "v"(b0), [b1] "v"(b1), [b2] "v"(b2), [b3] "v"(b3), [addr] "v"(addr), [o0] "i"(0),
                   [o1] "i"(2048), [o2] "i"(4096), [o3] "i"(6144), [gb] "s"(gb), [l0] "s"(l0), [v0] "v"(v0), [l1] "s"(l1),
                   [v1] "v"(v1), [lg] "i"(LG)
                 : "memory");
  } else {
    asm volatile(B2_MF "%[c0], %[a], %[b0], %[c0]\n\t" B2_RD(0) B2_MF "%[c1], %[a], %[b1], %[c1]\n\t" B2_RD(1) B2_DMA(0)
                 B2_MF "%[c2], %[a], %[b2], %[c2]\n\t" B2_RD(2) B2_MF "%[c3], %[a], %[b3], %[c3]\n\t" B2_RD(3) B2_DMA(1)
                 : [c0] "+a"(c0), [c1] "+a"(c1), [c2] "+a"(c2), [c3] "+a"(c3), [n0] "=&v"(n0), [n1] "=&v"(n1), [n2] "=&v"(n2),
                   [n3] "=&v"(n3)
                 : [a] "v"(a), [b0] "v"(b0), [b1] "v"(b1), [b2] "v"(b2), [b3] "v"(b3), [addr] "v"(addr), [o0] "i"(0),
                   [o1] "i"(2048), [o2] "i"(4096), [o3] "i"(6144), [gb] "s"(gb), [l0] "s"(l0), [v0] "v"(v0), [l1] "s"(l1),
                   [v1] "v"(v1)
                 : "memory");
  }
}
__device__ __forceinline__ void dma_piece(const char* gb, unsigned voff, unsigned lds) {
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %0" ::"s"(gb), "s"(lds), "v"(voff) : "memory");
}
struct Frags {        // one half slab's operands of a wave: [k-step][32-row block]
  half8 a[2][4], b[2][4];
};
constexpr int HALF = 32768, BOFF = 16384, NBUF = 4;

// One iteration: the 32 MFMAs of the half slab in `cur`, the reads of the next one into `nxt`, the pieces of the half slab
// three ahead.  rd_*: fragment read bases (k-step 0 / 1 of A, of B) inside the NEXT half slab's buffer.
__device__ __forceinline__ void half_slab(floatx16 (&acc)[4][4], const Frags& cur, Frags& nxt, unsigned rd_a0, unsigned rd_a1,
                                          unsigned rd_b0, unsigned rd_b1, const char* a_src, const char* b_src,
                                          const unsigned (&aoff)[4], const unsigned (&boff)[4], unsigned lds_dst) {
#define ROW(m_) acc[m_][0], acc[m_][1], acc[m_][2], acc[m_][3]
#define OPS(s_, m_) cur.a[s_][m_], cur.b[s_][0], cur.b[s_][1], cur.b[s_][2], cur.b[s_][3]
  mm4_wait<8>(ROW(0), OPS(0, 0));                 // this half slab's k-step-0 fragments have arrived
  mm4(ROW(1), OPS(0, 1));
  // half slab j + 1 has landed in every wave's view; every wave has left the previous iteration
  asm volatile("s_waitcnt vmcnt(8)\n\ts_barrier" ::: "memory");
  mm4_rd4_dma2<-1>(ROW(2), OPS(0, 2), nxt.b[0][0], nxt.b[0][1], nxt.b[0][2], nxt.b[0][3], rd_b0, a_src, lds_dst, aoff[0],
                   lds_dst + 1024, aoff[1]);
  mm4_rd4_dma2<-1>(ROW(3), OPS(0, 3), nxt.a[0][0], nxt.a[0][1], nxt.a[0][2], nxt.a[0][3], rd_a0, a_src, lds_dst + 2048, aoff[2],
                   lds_dst + 3072, aoff[3]);
  // k-step 1: its fragments were read an iteration ago; the 8 reads just issued may still be in flight
  mm4_rd4_dma2<8>(ROW(0), OPS(1, 0), nxt.b[1][0], nxt.b[1][1], nxt.b[1][2], nxt.b[1][3], rd_b1, b_src, lds_dst + BOFF, boff[0],
                  lds_dst + BOFF + 1024, boff[1]);
  mm4_rd4_dma2<-1>(ROW(1), OPS(1, 1), nxt.a[1][0], nxt.a[1][1], nxt.a[1][2], nxt.a[1][3], rd_a1, b_src, lds_dst + BOFF + 2048,
                   boff[2], lds_dst + BOFF + 3072, boff[3]);
  mm4(ROW(2), OPS(1, 2));
  mm4(ROW(3), OPS(1, 3));
#undef ROW
#undef OPS
}
}  // namespace big2

__global__ void __launch_bounds__(256, 1) gemm_nt_big2_kernel(GemmArgs p) {
  using namespace big2;
  extern __shared__ __attribute__((aligned(16))) char smem[];   // 4 x [A half | B half]; then the epilogue's staging
  typedef __attribute__((address_space(3))) void* lptr_t;
  const int nwg = p.ntm * p.ntn;
  const int z = blockIdx.y;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 31, h = lane >> 5;
  const int wr = wave >> 1, wc = wave & 1;
  const unsigned lds0 = (unsigned)(uintptr_t)(lptr_t)smem;
  // fragment read bases inside buffer 0: row (128 w + r) of the operand half, chunk (2 s + h) ^ ((r >> 2) & 3)
  const unsigned fa0 = lds0 + (128 * wr + r) * 64 + ((h ^ ((r >> 2) & 3)) * 16), fa1 = fa0 ^ 32;
  const unsigned fb0 = lds0 + BOFF + (128 * wc + r) * 64 + ((h ^ ((r >> 2) & 3)) * 16), fb1 = fb0 ^ 32;
  const int nh = p.K / 32;                         // half slabs
  for (int vid = blockIdx.x; vid < nwg; vid += gridDim.x) {
    int tm, tn;
    tile_coords(vid, p.ntm, p.ntn, tm, tn);
    tm = __builtin_amdgcn_readfirstlane(tm); tn = __builtin_amdgcn_readfirstlane(tn);
    const int bm0 = tm * 256, bn0 = tn * 256;
    const _Float16* A = (const _Float16*)p.A + z * p.sA;
    const _Float16* B = (const _Float16*)p.B + z * p.sB;
    // per-lane source offsets of this wave's 4 + 4 pieces of a half slab: piece i = rows 16 (4 wave + i) .. + 15, LDS slot
    // (row, pos = lane & 3) <- chunk pos ^ ((row >> 2) & 3); rows clamped into the matrix (ragged last tiles)
    unsigned aoff[4], boff[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int row = 16 * (4 * wave + i) + (lane >> 2), pos = lane & 3;
      const int logical = pos ^ ((row >> 2) & 3);
      const int ra = min(bm0 + row, p.M - 1) - bm0, rb = min(bn0 + row, p.N - 1) - bn0;
      aoff[i] = (unsigned)ra * (unsigned)(p.lda * 2) + logical * 16;
      boff[i] = (unsigned)rb * (unsigned)(p.ldb * 2) + logical * 16;
    }
    const char* Atile = (const char*)(A + (size_t)bm0 * p.lda);
    const char* Btile = (const char*)(B + (size_t)bn0 * p.ldb);
    // the 16 accumulators are zeroed IN the accumulator file, one MFMA each (0 . 0 + 0: the operand fragment is a zero
    // VGPR quad): written as `acc = 0` the compiler builds 256 zero VGPRs first and spills them around the prologue
    floatx16 acc[4][4];
    {
      const half8 zf = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
      for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int n = 0; n < 4; ++n)
          asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_f16 %0, %1, %1, 0" : "=a"(acc[m][n]) : "v"(zf));
    }
    const unsigned pdst = lds0 + (4 * wave) * 1024;         // this wave's pieces inside a buffer's A half
    // ---- prologue: half slabs 0, 1, 2 -> buffers 0, 1, 2 (clamped re-loads of the last one when K is short)
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const size_t kb = (size_t)min(j, nh - 1) * 64;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        dma_piece(Atile + kb, aoff[i], pdst + j * HALF + i * 1024);
        dma_piece(Btile + kb, boff[i], pdst + j * HALF + BOFF + i * 1024);
      }
    }
    asm volatile("s_waitcnt vmcnt(16)\n\ts_barrier" ::: "memory");      // half slab 0 has landed
    Frags f0, f1;
    // the first half slab's fragments, in the order the loop's counted waits expect: B k0, A k0, B k1, A k1
#define B2_PRIME(dst_, base_, m_) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst_) : "v"(base_), "i"((m_) * 2048))
    B2_PRIME(f0.b[0][0], fb0, 0); B2_PRIME(f0.b[0][1], fb0, 1); B2_PRIME(f0.b[0][2], fb0, 2); B2_PRIME(f0.b[0][3], fb0, 3);
    B2_PRIME(f0.a[0][0], fa0, 0); B2_PRIME(f0.a[0][1], fa0, 1); B2_PRIME(f0.a[0][2], fa0, 2); B2_PRIME(f0.a[0][3], fa0, 3);
    B2_PRIME(f0.b[1][0], fb1, 0); B2_PRIME(f0.b[1][1], fb1, 1); B2_PRIME(f0.b[1][2], fb1, 2); B2_PRIME(f0.b[1][3], fb1, 3);
    B2_PRIME(f0.a[1][0], fa1, 0); B2_PRIME(f0.a[1][1], fa1, 1); B2_PRIME(f0.a[1][2], fa1, 2); B2_PRIME(f0.a[1][3], fa1, 3);
#undef B2_PRIME
    // ---- K loop, two half slabs per trip (the fragment sets alternate by name)
    int rdbuf = 1;                                   // buffer of half slab j + 1
    for (int j = 0; j < nh; j += 2) {
#pragma unroll
      for (int par = 0; par < 2; ++par) {
        const int jj = j + par;
        if (jj < nh) {                               // wave-uniform (K / 32 may be odd)
          const unsigned ro = (unsigned)rdbuf * HALF;
          const int wrbuf = (rdbuf + 2) & 3;         // buffer of half slab j + 3 = the one half slab j - 1 was read from
          const size_t kb = (size_t)min(jj + 3, nh - 1) * 64;
          if (par == 0)
            half_slab(acc, f0, f1, fa0 + ro, fa1 + ro, fb0 + ro, fb1 + ro, Atile + kb, Btile + kb, aoff, boff, pdst + wrbuf * HALF);
          else
            half_slab(acc, f1, f0, fa0 + ro, fa1 + ro, fb0 + ro, fb1 + ro, Atile + kb, Btile + kb, aoff, boff, pdst + wrbuf * HALF);
          rdbuf = (rdbuf + 1) & 3;
        }
      }
    }
    // the ring still has pieces and reads in flight (clamped re-loads): drain them before LDS becomes the epilogue's
    // staging area; the last MFMAs must have written their accumulators before the compiler's code reads them
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_nop 15\n\ts_nop 7\n\ts_barrier" ::: "memory");

    // ---------------- epilogue (the 8-wave kernel's flavours) ----------------
    float* C32 = p.C32 ? p.C32 + z * p.sC32 : nullptr;
    _Float16* C16 = p.C16 ? p.C16 + z * p.sC16 : nullptr;
    const float* bias = p.bias ? p.bias + z * p.sBias : nullptr;
    const float* res = p.res ? p.res + z * p.sRes : nullptr;
    const float* gate = p.gate ? p.gate + z * p.sGate : nullptr;
    float* stage = (float*)smem + wave * (64 * 64);
    int lane_e = lane;
    asm volatile("" : "+v"(lane_e));                 // see gemm_nt_big_kernel: keep epilogue addressing out of the K loop
#pragma unroll
    for (int mb = 0; mb < 2; ++mb)
#pragma unroll
      for (int nb = 0; nb < 2; ++nb) {
        __builtin_amdgcn_sched_barrier(0);   // one 64 x 64 block at a time: 64 accumulator values in VGPRs, not 256
        const int row0 = bm0 + 128 * wr + 64 * mb, col0 = bn0 + 128 * wc + 64 * nb;
        if (res)
          epilogue_block64<1>(p, stage, acc[2 * mb][2 * nb], acc[2 * mb][2 * nb + 1], acc[2 * mb + 1][2 * nb],
                              acc[2 * mb + 1][2 * nb + 1], row0, col0, lane_e, C32, C16, bias, res, gate);
        else
          epilogue_block64<0>(p, stage, acc[2 * mb][2 * nb], acc[2 * mb][2 * nb + 1], acc[2 * mb + 1][2 * nb],
                              acc[2 * mb + 1][2 * nb + 1], row0, col0, lane_e, C32, C16, bias, res, gate);
      }
    __syncthreads();   // every wave has read its staging region back: the next tile's LDS-DMA may overwrite it
  }
}

// ================================================================================================
// The same 256 x 256 x 64 kernel on v_mfma_f32_16x16x32_f16.  Both MFMA shapes do the same FLOPs per cycle, read the
// same number of fragments per FLOP on this wave tile (12 ds_read_b128 per 32-deep step) and use the same LDS image; but
// these loops run power-limited (1.3-1.5 GHz of the 2.4 GHz the peak assumes), and the clock the chip holds under a
// matrix-dense loop depends on the MFMA shape: MI355X_MICROARCH.md 'DVFS give-back' (7) measures the 16x16x32 loop at
// 1.12-1.15x the FLOP/s of the 32x32x16 loop at equal cycles, operands re-read from LDS included.
//   A / B fragment of lane l (c = l & 15, q = l >> 4): row c of a 16-row tile, k = 32 s + 8 q .. + 7  ->  16-byte chunk
//   4 s + q of the 128-byte LDS row, XOR-swizzled by (row >> 1) & 7 like every other reader of this image (a
//   ds_read_b128 lane group {0-3, 12-15, 20-27} covers rows 0-3 and 12-15 at chunk q0 and rows 4-11 at chunk q0 ^ 1: all
//   sixteen 16-byte bank slots).  Accumulator register e of lane l = row 4 q + e, column c of its 16 x 16 tile.
// Plain (un-split) operands only.  MEASURED AND REJECTED (round 3, M = 331 776, benchmarks/gemm_time.py): 990-1043 TF/s
// against 1030-1095 TF/s for gemm_nt_big_kernel on the same box and shapes - 1 to 7 % SLOWER: the clock advantage of the
// bare loop does not survive 64 instead of 32 MFMA issues per slab and per wave.  Lab build only (DVD_GEMM_M16=1).
// ================================================================================================
__device__ __forceinline__ floatx4 mfma16_f16(half8 a, half8 b, floatx4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
}

__global__ void __launch_bounds__(512, 2) gemm_nt_big16_kernel(GemmArgs p) {
  constexpr int BK = 64, TILE = 256 * 128;   // bytes of one operand tile (256 rows x 64 halfs)
  extern __shared__ __attribute__((aligned(16))) char smem[];   // [2 stages][A | B]
  typedef __attribute__((address_space(3))) void* lptr_t;
  const int nwg = p.ntm * p.ntn;
  const int z = blockIdx.y;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 2, wc = wave & 3;
  for (int vid = blockIdx.x; vid < nwg; vid += gridDim.x) {
  int tm, tn;
  tile_coords(vid, p.ntm, p.ntn, tm, tn);
  tm = __builtin_amdgcn_readfirstlane(tm); tn = __builtin_amdgcn_readfirstlane(tn);
  const int bm0 = tm * 256, bn0 = tn * 256;
  const _Float16* A = (const _Float16*)p.A + z * p.sA;
  const _Float16* B = (const _Float16*)p.B + z * p.sB;

  unsigned aoff[4], boff[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row = 8 * (4 * wave + i) + (lane >> 3), pos = lane & 7;
    const int logical = pos ^ ((row >> 1) & 7);
    const int ra = min(bm0 + row, p.M - 1) - bm0, rb = min(bn0 + row, p.N - 1) - bn0;
    aoff[i] = (unsigned)ra * (unsigned)(p.lda * 2) + logical * 16;
    boff[i] = (unsigned)rb * (unsigned)(p.ldb * 2) + logical * 16;
  }
  const char* Atile = (const char*)(A + (size_t)bm0 * p.lda);
  const char* Btile = (const char*)(B + (size_t)bn0 * p.ldb);
  const unsigned lds0 = (unsigned)(uintptr_t)(lptr_t)smem;

  const int c16 = lane & 15, q = lane >> 4;
  int frag[2];
#pragma unroll
  for (int s2 = 0; s2 < 2; ++s2) frag[s2] = c16 * 128 + (((4 * s2 + q) ^ ((c16 >> 1) & 7)) * 16);
  const int a_base = wr * 128 * 128;            // this wave's first A row, bytes
  const int b_base = TILE + wc * 64 * 128;      // this wave's first B row

  floatx4 acc[8][4];
#pragma unroll
  for (int m = 0; m < 8; ++m)
#pragma unroll
    for (int n = 0; n < 4; ++n)
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[m][n][i] = 0.f;
  const int nk = p.K / BK;

#define BIG_ISSUE(t_, stage_)                                                                      \
  {                                                                                                \
    const size_t kb_ = (size_t)(t_) * (BK * 2);                                                    \
    glds_group4<4>(Atile + kb_, aoff, lds0 + (stage_) * 2 * TILE + (4 * wave) * 1024);             \
    glds_group4<4>(Btile + kb_, boff, lds0 + (stage_) * 2 * TILE + TILE + (4 * wave) * 1024);      \
  }
#define SB() __builtin_amdgcn_sched_barrier(0)
  BIG_ISSUE(0, 0)
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  int cur = 0;
  for (int kt = 0; kt < nk; ++kt) {
    // waves 0-3 (one per SIMD) issue their burst here, waves 4-7 after the first of the two k-steps (see gemm_nt_big_kernel)
    if (kt + 1 < nk && wave < 4) BIG_ISSUE(kt + 1, cur ^ 1)
    const char* base = smem + cur * 2 * TILE;
    half8 fa[2][8], fb[2][4];
#pragma unroll
    for (int m = 0; m < 8; ++m) fa[0][m] = *(const half8*)(base + a_base + m * 16 * 128 + frag[0]);
#pragma unroll
    for (int n = 0; n < 4; ++n) fb[0][n] = *(const half8*)(base + b_base + n * 16 * 128 + frag[0]);
    SB();
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) {
      // 32 MFMAs of one 32-deep k-step; the 12 fragment reads of the next step are spread between the first 8 groups
#pragma unroll
      for (int m = 0; m < 8; ++m) {
#pragma unroll
        for (int n = 0; n < 4; ++n) acc[m][n] = mfma16_f16(fa[s2][m], fb[s2][n], acc[m][n]);
        if (s2 == 0) {
          fa[1][m] = *(const half8*)(base + a_base + m * 16 * 128 + frag[1]);
          if (m < 4) fb[1][m] = *(const half8*)(base + b_base + m * 16 * 128 + frag[1]);
        }
        SB();
      }
      if (s2 == 0) {
        if (kt + 1 < nk && wave >= 4) BIG_ISSUE(kt + 1, cur ^ 1)
        SB();
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    cur ^= 1;
  }
#undef BIG_ISSUE
#undef SB

  // ---------------- epilogue (same semantics as gemm_nt_big_kernel) ----------------
  float* C32 = p.C32 ? p.C32 + z * p.sC32 : nullptr;
  _Float16* C16 = p.C16 ? p.C16 + z * p.sC16 : nullptr;
  const float* bias = p.bias ? p.bias + z * p.sBias : nullptr;
  const float* res = p.res ? p.res + z * p.sRes : nullptr;
  const float* gate = p.gate ? p.gate + z * p.sGate : nullptr;
  {
    float* stage = (float*)smem + wave * (64 * 64);                  // 8 x 16 KiB = the whole 128 KiB
    const int row0 = bm0 + 128 * wr, col0 = bn0 + 64 * wc;
    int lane_e = lane;
    asm volatile("" : "+v"(lane_e));                                 // see gemm_nt_big_kernel
    if (res) {
      epilogue_block64_m16<1>(p, stage, acc, 0, row0, col0, lane_e, C32, C16, bias, res, gate);
      epilogue_block64_m16<1>(p, stage, acc, 4, row0 + 64, col0, lane_e, C32, C16, bias, res, gate);
    } else if (bias && p.bias_row) {
      epilogue_block64_m16<3>(p, stage, acc, 0, row0, col0, lane_e, C32, C16, bias, res, gate);
      epilogue_block64_m16<3>(p, stage, acc, 4, row0 + 64, col0, lane_e, C32, C16, bias, res, gate);
    } else {
      epilogue_block64_m16<0>(p, stage, acc, 0, row0, col0, lane_e, C32, C16, bias, res, gate);
      epilogue_block64_m16<0>(p, stage, acc, 4, row0 + 64, col0, lane_e, C32, C16, bias, res, gate);
    }
  }
  __syncthreads();   // every wave has read its staging region back: the next tile's LDS-DMA may overwrite it
  }  // persistent tile loop
}


#endif  // DVD_LAB

// ================================================================================================
// Split-weight GEMM in ONE pass over K:  C = A . (B + Blo)^T  with Blo stored UNSCALED (f16 subnormals allowed: the
// f16 MFMA honours them on gfx950, checked in benchmarks/lab/denorm_lab.hip), so both products go into the same fp32
// accumulators and no second sweep / rescale is needed.  Per 32-deep K slab a workgroup loads three 16-KiB tiles
// [A | B | Blo] instead of [A | B] twice: 25 % fewer bytes through the 33 B/clk L2->LDS path, one A fragment read feeds
// four MFMAs instead of two (0.5 instead of 0.75 ds_read_b128 per MFMA), same 32 MFMAs per wave between barriers, and
// three 48-KiB stages (prefetch distance two slabs) fit the LDS.  Same 256 x 256 tile, wave layout, persistent tile
// walk and epilogue as gemm_nt_big_kernel.  64-byte LDS rows: 16-byte chunk c of row r lives at c ^ ((r >> 2) & 3).
// ================================================================================================
// SPLIT = false (lab build, DVD_GEMM_RING=1): the same 3-stage skewed pipeline on ONE weight tensor - a stage is [A | B], a
// wave has 8 MFMAs per half slab; measured against gemm_nt_big_kernel for the dithered weights.
template <bool SPLIT>
__global__ void __launch_bounds__(512, 2) gemm_nt_split_kernel(GemmArgs p) {
  constexpr int BK = 32, TILE = 256 * 64, STAGE = 3 * TILE, NST = 3;
  extern __shared__ __attribute__((aligned(16))) char smem[];   // [3 stages][A | B | Blo]
  typedef __attribute__((address_space(3))) void* lptr_t;
  const int nwg = p.ntm * p.ntn;
  const int z = blockIdx.y;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 31, h = lane >> 5;
  const int wr = wave >> 2, wc = wave & 3;
  for (int vid = blockIdx.x; vid < nwg; vid += gridDim.x) {
  int tm, tn;
  tile_coords(vid, p.ntm, p.ntn, tm, tn);
  tm = __builtin_amdgcn_readfirstlane(tm); tn = __builtin_amdgcn_readfirstlane(tn);   // wave-uniform: keep in SGPRs
  const int bm0 = tm * 256, bn0 = tn * 256;
  const _Float16* A = (const _Float16*)p.A + z * p.sA;
  const _Float16* B = (const _Float16*)p.B + z * p.sB;
  const _Float16* Blo = SPLIT ? (const _Float16*)p.Blo + z * p.sB : B;

  // per-lane source offsets of this wave's 2 + 2 + 2 loads (1 KiB = 16 rows x 64 B each)
  unsigned aoff[2], boff[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int row = 16 * (2 * wave + i) + (lane >> 2), pos = lane & 3;
    const int logical = pos ^ ((row >> 2) & 3);
    const int ra = min(bm0 + row, p.M - 1) - bm0, rb = min(bn0 + row, p.N - 1) - bn0;
    aoff[i] = (unsigned)ra * (unsigned)(p.lda * 2) + logical * 16;
    boff[i] = (unsigned)rb * (unsigned)(p.ldb * 2) + logical * 16;
  }
  const char* Atile = (const char*)(A + (size_t)bm0 * p.lda);
  const char* Btile = (const char*)(B + (size_t)bn0 * p.ldb);
  const char* Ltile = (const char*)(Blo + (size_t)bn0 * p.ldb);
  const unsigned lds0 = (unsigned)(uintptr_t)(lptr_t)smem;

  int frag[2];
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) frag[ks] = r * 64 + (((2 * ks + h) ^ ((r >> 2) & 3)) * 16);
  const int a_base = wr * 128 * 64;
  const int b_base = TILE + wc * 64 * 64;
  const int l_base = 2 * TILE + wc * 64 * 64;

  floatx16 acc[4][2];
#pragma unroll
  for (int m = 0; m < 4; ++m)
#pragma unroll
    for (int n = 0; n < 2; ++n)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[m][n][i] = 0.f;
  const int nk = p.K / BK;

#define SPLIT_ISSUE(t_, stage_)                                                                          \
  {                                                                                                      \
    const size_t kb_ = (size_t)(t_) * (BK * 2);                                                          \
    const unsigned l_ = lds0 + (stage_) * STAGE + (2 * wave) * 1024;                                     \
    glds_one4(Atile + kb_, aoff[0], l_);            glds_one4(Atile + kb_, aoff[1], l_ + 1024);          \
    glds_one4(Btile + kb_, boff[0], l_ + TILE);     glds_one4(Btile + kb_, boff[1], l_ + TILE + 1024);   \
    if constexpr (SPLIT) { glds_one4(Ltile + kb_, boff[0], l_ + 2 * TILE); glds_one4(Ltile + kb_, boff[1], l_ + 2 * TILE + 1024); } \
  }
#define SB() __builtin_amdgcn_sched_barrier(0)
  // prologue: slabs 0 and 1 (clamped: K = 32 has a single slab)
  SPLIT_ISSUE(0, 0)
  SPLIT_ISSUE(min(1, nk - 1), 1)
  if constexpr (SPLIT) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");   // slab 0 landed (this wave's part); slab 1 may be in flight
  else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
  __syncthreads();
  // The loop is skewed by half a slab so that no wave ever starts a slab with cold fragment registers: the barrier
  // that publishes slab kt + 1 sits in the MIDDLE of slab kt, and the second half of slab kt already reads the first
  // fragments of slab kt + 1 between its MFMAs.  (With the barrier at the end of the slab both waves of a SIMD stall
  // together on ~150 cycles of LDS latency + the barrier, once per 32 MFMAs.)
  half8 fa[2][4], fh[2][2], fl[2][2];
#define SPLIT_READ(set_, base_, ks_)                                                                  \
  {                                                                                                   \
    _Pragma("unroll") for (int m = 0; m < 4; ++m) fa[set_][m] = *(const half8*)((base_) + a_base + m * 32 * 64 + frag[ks_]); \
    _Pragma("unroll") for (int n = 0; n < 2; ++n) {                                                   \
      fh[set_][n] = *(const half8*)((base_) + b_base + n * 32 * 64 + frag[ks_]);                      \
      if constexpr (SPLIT) fl[set_][n] = *(const half8*)((base_) + l_base + n * 32 * 64 + frag[ks_]); \
    }                                                                                                 \
  }
  // 16 MFMAs of one k-step (8 hi then 8 lo: an accumulator is revisited after 8 others); the 8 fragment reads of the
  // NEXT k-step (set nset_, from nbase_ at k-step nks_) are spread between the first 8
#define SPLIT_KSTEP(set_, nset_, nbase_, nks_)                                                        \
  {                                                                                                   \
    _Pragma("unroll") for (int m = 0; m < 4; ++m) {                                                   \
      _Pragma("unroll") for (int n = 0; n < 2; ++n) acc[m][n] = mfma32_f16(fa[set_][m], fh[set_][n], acc[m][n]); \
      fa[nset_][m] = *(const half8*)((nbase_) + a_base + m * 32 * 64 + frag[nks_]);                   \
      if (m < 2) fh[nset_][m] = *(const half8*)((nbase_) + b_base + m * 32 * 64 + frag[nks_]);        \
      else if constexpr (SPLIT) fl[nset_][m - 2] = *(const half8*)((nbase_) + l_base + (m - 2) * 32 * 64 + frag[nks_]); \
      SB();                                                                                           \
    }                                                                                                 \
    if constexpr (SPLIT) {                                                                            \
      _Pragma("unroll") for (int m = 0; m < 4; ++m) {                                                 \
        _Pragma("unroll") for (int n = 0; n < 2; ++n) acc[m][n] = mfma32_f16(fa[set_][m], fl[set_][n], acc[m][n]); \
        SB();                                                                                         \
      }                                                                                               \
    }                                                                                                 \
  }
  SPLIT_READ(0, smem, 0)
  SB();
  int cur = 0;
  for (int kt = 0; kt < nk; ++kt) {
    const char* base = smem + cur * STAGE;
    int n1 = cur + 1; if (n1 >= NST) n1 -= NST;
    int n2 = cur + 2; if (n2 >= NST) n2 -= NST;
    SPLIT_KSTEP(0, 1, base, 1)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // slab kt + 1 (issued half a slab ago or in the prologue) landed
    __syncthreads();                                   // ... for everyone; and everyone is past slab kt - 1
    // -> the stage slab kt - 1 used.  The two waves of a SIMD issue their six loads at different points of the second
    // half (right after the barrier / between its hi and lo MFMAs), so one's issue stalls sit under the other's MFMAs
    // (+1..3 %; spreading the six loads one by one behind run-time wave-group tests was slower again)
    if (wave < 4) SPLIT_ISSUE(min(kt + 2, nk - 1), n2)
    SB();
    {
      const char* nb = smem + n1 * STAGE;              // second half: already reads slab kt + 1's first fragments
#pragma unroll
      for (int m = 0; m < 4; ++m) {
#pragma unroll
        for (int n = 0; n < 2; ++n) acc[m][n] = mfma32_f16(fa[1][m], fh[1][n], acc[m][n]);
        fa[0][m] = *(const half8*)(nb + a_base + m * 32 * 64 + frag[0]);
        if (m < 2) fh[0][m] = *(const half8*)(nb + b_base + m * 32 * 64 + frag[0]);
        else if constexpr (SPLIT) fl[0][m - 2] = *(const half8*)(nb + l_base + (m - 2) * 32 * 64 + frag[0]);
        SB();
      }
      if (wave >= 4) SPLIT_ISSUE(min(kt + 2, nk - 1), n2)
      SB();
      if constexpr (SPLIT) {
#pragma unroll
        for (int m = 0; m < 4; ++m) {
#pragma unroll
          for (int n = 0; n < 2; ++n) acc[m][n] = mfma32_f16(fa[1][m], fl[1][n], acc[m][n]);
          SB();
        }
      }
    }
    cur = n1;
  }
#undef SPLIT_READ
#undef SPLIT_KSTEP
#undef SPLIT_ISSUE
#undef SB
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // the clamped tail loads: nobody may still be writing LDS
  __syncthreads();

  // ---------------- epilogue (same as gemm_nt_big_kernel) ----------------
  float* C32 = p.C32 ? p.C32 + z * p.sC32 : nullptr;
  _Float16* C16 = p.C16 ? p.C16 + z * p.sC16 : nullptr;
  const float* bias = p.bias ? p.bias + z * p.sBias : nullptr;
  const float* res = p.res ? p.res + z * p.sRes : nullptr;
  const float* gate = p.gate ? p.gate + z * p.sGate : nullptr;
  if (p.vec_epilogue && !p.pos && !gate && !(bias && p.bias_row)) {
    float* stage = (float*)smem + wave * (64 * 64);                  // 8 x 16 KiB of the 144 KiB
    const int row0 = bm0 + 128 * wr, col0 = bn0 + 64 * wc;
    int lane_e = lane;
    asm volatile("" : "+v"(lane_e));                                 // see gemm_nt_big_kernel
    if (res) {
      epilogue_block64<1>(p, stage, acc[0][0], acc[0][1], acc[1][0], acc[1][1], row0, col0, lane_e, C32, C16, bias, res, gate);
      epilogue_block64<1>(p, stage, acc[2][0], acc[2][1], acc[3][0], acc[3][1], row0 + 64, col0, lane_e, C32, C16, bias, res, gate);
    } else {
      epilogue_block64<0>(p, stage, acc[0][0], acc[0][1], acc[1][0], acc[1][1], row0, col0, lane_e, C32, C16, bias, res, gate);
      epilogue_block64<0>(p, stage, acc[2][0], acc[2][1], acc[3][0], acc[3][1], row0 + 64, col0, lane_e, C32, C16, bias, res, gate);
    }
  } else {
#define BIG_EP(m_, n_)                                                                                    \
  {                                                                                                       \
    const int col = bn0 + 64 * wc + 32 * (n_) + r;                                                        \
    if (col < p.N) {                                                                                      \
      const float bcol = (bias && !p.bias_row) ? bias[col] : 0.f;                                         \
      epilogue_tile(p, acc[m_][n_], bm0 + 128 * wr + 32 * (m_), col, h, bcol, C32, C16, bias, res, gate); \
    }                                                                                                     \
  }
  BIG_EP(0, 0) BIG_EP(0, 1) BIG_EP(1, 0) BIG_EP(1, 1) BIG_EP(2, 0) BIG_EP(2, 1) BIG_EP(3, 0) BIG_EP(3, 1)
#undef BIG_EP
  }
  __syncthreads();   // staging regions read back: the next tile's loads may overwrite them
  }  // persistent tile loop
}

// Same kernel for N % 128 == 0 (the DiT block's N = 384 / 1152 projections): 256 x 128 tiles, a wave owns 64 x 64, the
// general (pos / gate / residual) staged epilogue is available because only 64 accumulator registers are live.
// SPLIT = false: the same pipeline on ONE weight tensor (the engine's dithered weights): no Blo tile, no lo MFMAs - a stage
// is [A | B] and a wave has 8 MFMAs per half slab.
template <bool SPLIT>
__global__ void __launch_bounds__(512, 2) gemm_nt_split128_kernel(GemmArgs p) {
  constexpr int BK = 32, TILE = 256 * 64, TILEB = 128 * 64, STAGE = TILE + 2 * TILEB, NST = 3;
  extern __shared__ __attribute__((aligned(16))) char smem[];   // [3 stages][A 16K | B 8K | Blo 8K] (>= 128 KiB for the epilogue)
  typedef __attribute__((address_space(3))) void* lptr_t;
  const int nwg = p.ntm * p.ntn;
  const int z = blockIdx.y;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 31, h = lane >> 5;
  const int wr = wave >> 1, wc = wave & 1;
  for (int vid = blockIdx.x; vid < nwg; vid += gridDim.x) {
  int tm, tn;
  tile_coords(vid, p.ntm, p.ntn, tm, tn);
  tm = __builtin_amdgcn_readfirstlane(tm); tn = __builtin_amdgcn_readfirstlane(tn);   // wave-uniform: keep in SGPRs
  const int bm0 = tm * 256, bn0 = tn * 128;
  const _Float16* A = (const _Float16*)p.A + z * p.sA;
  const _Float16* B = (const _Float16*)p.B + z * p.sB;
  const _Float16* Blo = SPLIT ? (const _Float16*)p.Blo + z * p.sB : B;

  // per-lane source offsets of this wave's 2 + 1 + 1 loads (1 KiB = 16 rows x 64 B each)
  unsigned aoff[2], boff;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int row = 16 * (2 * wave + i) + (lane >> 2), pos = lane & 3;
    const int ra = min(bm0 + row, p.M - 1) - bm0;
    aoff[i] = (unsigned)ra * (unsigned)(p.lda * 2) + (pos ^ ((row >> 2) & 3)) * 16;
  }
  {
    const int row = 16 * wave + (lane >> 2), pos = lane & 3;
    const int rb = min(bn0 + row, p.N - 1) - bn0;
    boff = (unsigned)rb * (unsigned)(p.ldb * 2) + (pos ^ ((row >> 2) & 3)) * 16;
  }
  const char* Atile = (const char*)(A + (size_t)bm0 * p.lda);
  const char* Btile = (const char*)(B + (size_t)bn0 * p.ldb);
  const char* Ltile = (const char*)(Blo + (size_t)bn0 * p.ldb);
  const unsigned lds0 = (unsigned)(uintptr_t)(lptr_t)smem;

  int frag[2];
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) frag[ks] = r * 64 + (((2 * ks + h) ^ ((r >> 2) & 3)) * 16);
  const int a_base = wr * 64 * 64;
  const int b_base = TILE + wc * 64 * 64;
  const int l_base = TILE + TILEB + wc * 64 * 64;

  floatx16 acc[2][2];
#pragma unroll
  for (int m = 0; m < 2; ++m)
#pragma unroll
    for (int n = 0; n < 2; ++n)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[m][n][i] = 0.f;
  const int nk = p.K / BK;

#define SPLIT_ISSUE(t_, stage_)                                                                          \
  {                                                                                                      \
    const size_t kb_ = (size_t)(t_) * (BK * 2);                                                          \
    const unsigned l_ = lds0 + (stage_) * STAGE;                                                         \
    glds_one4(Atile + kb_, aoff[0], l_ + (2 * wave) * 1024); glds_one4(Atile + kb_, aoff[1], l_ + (2 * wave + 1) * 1024); \
    glds_one4(Btile + kb_, boff, l_ + TILE + wave * 1024);                                               \
    if constexpr (SPLIT) glds_one4(Ltile + kb_, boff, l_ + TILE + TILEB + wave * 1024);                  \
  }
#define SB() __builtin_amdgcn_sched_barrier(0)
  // prologue: slabs 0 and 1 (clamped: K = 32 has a single slab)
  SPLIT_ISSUE(0, 0)
  SPLIT_ISSUE(min(1, nk - 1), 1)
  if constexpr (SPLIT) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");   // slab 0 landed (this wave's part); slab 1 may be in flight
  else asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
  __syncthreads();
  // The loop is skewed by half a slab so that no wave ever starts a slab with cold fragment registers: the barrier
  // that publishes slab kt + 1 sits in the MIDDLE of slab kt, and the second half of slab kt already reads the first
  // fragments of slab kt + 1 between its MFMAs.  (With the barrier at the end of the slab both waves of a SIMD stall
  // together on ~150 cycles of LDS latency + the barrier, once per 32 MFMAs.)
  half8 fa[2][2], fh[2][2], fl[2][2];
#define SPLIT_READ(set_, base_, ks_)                                                                  \
  {                                                                                                   \
    _Pragma("unroll") for (int m = 0; m < 2; ++m) fa[set_][m] = *(const half8*)((base_) + a_base + m * 32 * 64 + frag[ks_]); \
    _Pragma("unroll") for (int n = 0; n < 2; ++n) {                                                   \
      fh[set_][n] = *(const half8*)((base_) + b_base + n * 32 * 64 + frag[ks_]);                      \
      if constexpr (SPLIT) fl[set_][n] = *(const half8*)((base_) + l_base + n * 32 * 64 + frag[ks_]); \
    }                                                                                                 \
  }
  // 16 MFMAs of one k-step (8 hi then 8 lo: an accumulator is revisited after 8 others); the 8 fragment reads of the
  // NEXT k-step (set nset_, from nbase_ at k-step nks_) are spread between the first 8
#define SPLIT_KSTEP(set_, nset_, nbase_, nks_)                                                        \
  {                                                                                                   \
    _Pragma("unroll") for (int m = 0; m < 2; ++m) {                                                   \
      _Pragma("unroll") for (int n = 0; n < 2; ++n) acc[m][n] = mfma32_f16(fa[set_][m], fh[set_][n], acc[m][n]); \
      fa[nset_][m] = *(const half8*)((nbase_) + a_base + m * 32 * 64 + frag[nks_]);                   \
      fh[nset_][m] = *(const half8*)((nbase_) + b_base + m * 32 * 64 + frag[nks_]);                   \
      if constexpr (SPLIT) fl[nset_][m] = *(const half8*)((nbase_) + l_base + m * 32 * 64 + frag[nks_]); \
      SB();                                                                                           \
    }                                                                                                 \
    if constexpr (SPLIT) {                                                                            \
      _Pragma("unroll") for (int m = 0; m < 2; ++m) {                                                 \
        _Pragma("unroll") for (int n = 0; n < 2; ++n) acc[m][n] = mfma32_f16(fa[set_][m], fl[set_][n], acc[m][n]); \
        SB();                                                                                         \
      }                                                                                               \
    }                                                                                                 \
  }
  SPLIT_READ(0, smem, 0)
  SB();
  int cur = 0;
  for (int kt = 0; kt < nk; ++kt) {
    const char* base = smem + cur * STAGE;
    int n1 = cur + 1; if (n1 >= NST) n1 -= NST;
    int n2 = cur + 2; if (n2 >= NST) n2 -= NST;
    SPLIT_KSTEP(0, 1, base, 1)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // slab kt + 1 (issued half a slab ago or in the prologue) landed
    __syncthreads();                                   // ... for everyone; and everyone is past slab kt - 1
    // -> the stage slab kt - 1 used.  The two waves of a SIMD issue their six loads at different points of the second
    // half (right after the barrier / between its hi and lo MFMAs), so one's issue stalls sit under the other's MFMAs
    // (+1..3 %; spreading the six loads one by one behind run-time wave-group tests was slower again)
    if (wave < 4) SPLIT_ISSUE(min(kt + 2, nk - 1), n2)
    SB();
    {
      const char* nb = smem + n1 * STAGE;              // second half: already reads slab kt + 1's first fragments
#pragma unroll
      for (int m = 0; m < 2; ++m) {
#pragma unroll
        for (int n = 0; n < 2; ++n) acc[m][n] = mfma32_f16(fa[1][m], fh[1][n], acc[m][n]);
        fa[0][m] = *(const half8*)(nb + a_base + m * 32 * 64 + frag[0]);
        fh[0][m] = *(const half8*)(nb + b_base + m * 32 * 64 + frag[0]);
        if constexpr (SPLIT) fl[0][m] = *(const half8*)(nb + l_base + m * 32 * 64 + frag[0]);
        SB();
      }
      if (wave >= 4) SPLIT_ISSUE(min(kt + 2, nk - 1), n2)
      SB();
      if constexpr (SPLIT) {
#pragma unroll
        for (int m = 0; m < 2; ++m) {
#pragma unroll
          for (int n = 0; n < 2; ++n) acc[m][n] = mfma32_f16(fa[1][m], fl[1][n], acc[m][n]);
          SB();
        }
      }
    }
    cur = n1;
  }
#undef SPLIT_READ
#undef SPLIT_KSTEP
#undef SPLIT_ISSUE
#undef SB
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // the clamped tail loads: nobody may still be writing LDS
  __syncthreads();

  // ---------------- epilogue (same as gemm_nt_big_kernel) ----------------
  float* C32 = p.C32 ? p.C32 + z * p.sC32 : nullptr;
  _Float16* C16 = p.C16 ? p.C16 + z * p.sC16 : nullptr;
  const float* bias = p.bias ? p.bias + z * p.sBias : nullptr;
  const float* res = p.res ? p.res + z * p.sRes : nullptr;
  const float* gate = p.gate ? p.gate + z * p.sGate : nullptr;
  if (p.vec_epilogue) {
    float* stage = (float*)smem + wave * (64 * 64);                  // 8 x 16 KiB
    int lane_e = lane;
    asm volatile("" : "+v"(lane_e));                                 // see gemm_nt_big_kernel
    DVD_EPILOGUE_BLOCK64(p, stage, acc[0][0], acc[0][1], acc[1][0], acc[1][1], bm0 + 64 * wr, bn0 + 64 * wc, lane_e, C32, C16,
                         bias, res, gate)
  } else {
#define BIG_EP(m_, n_)                                                                                    \
  {                                                                                                       \
    const int col = bn0 + 64 * wc + 32 * (n_) + r;                                                        \
    if (col < p.N) {                                                                                      \
      const float bcol = (bias && !p.bias_row) ? bias[col] : 0.f;                                         \
      epilogue_tile(p, acc[m_][n_], bm0 + 64 * wr + 32 * (m_), col, h, bcol, C32, C16, bias, res, gate);  \
    }                                                                                                     \
  }
  BIG_EP(0, 0) BIG_EP(0, 1) BIG_EP(1, 0) BIG_EP(1, 1)
#undef BIG_EP
  }
  __syncthreads();   // staging regions read back: the next tile's loads may overwrite them
  }  // persistent tile loop
}

}  // namespace dvd

using namespace dvd;

#ifdef DVD_LAB
static unsigned long long* g_gemm_stamps = nullptr;
// lab builds only (DVD_GEMM_DEBUG=3): where the per-wave s_memtime stamps of the large-tile kernel go
extern "C" int dvd_gemm_debug_stamps(void* dev_u64) { g_gemm_stamps = (unsigned long long*)dev_u64; return DVD_OK; }
#endif

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
  p.cv_b = nullptr; p.cv_ca = p.cv_cb = p.cv_h = p.cv_w = p.cv_ks = p.cv_dil = 0;

  // The product library has no switches: the kernel is a function of the descriptor alone.  The lab build
  // (-DDVD_LAB, benchmarks/lab/libdvd_hip_lab.so) keeps the A/B switches of the experiments that were measured.
#ifdef DVD_LAB
  { const char* dbg = getenv("DVD_GEMM_DEBUG"); p.debug = dbg ? atoi(dbg) : 0; }
  { const char* sg = getenv("DVD_GEMM_STAGGER"); p.stagger = sg ? atoi(sg) : 0; }   // measured: no effect
  p.stamps = g_gemm_stamps;
  p.walk = 0;
  const bool lab_scalar_epi = getenv("DVD_GEMM_SCALAR_EPILOGUE"), lab_v1 = getenv("DVD_GEMM_V1"),
             lab_twopass = getenv("DVD_GEMM_TWOPASS"), lab_nonpersistent = getenv("DVD_GEMM_NONPERSISTENT"),
             lab_spread = getenv("DVD_GEMM_SPREAD"), lab_no_t384 = getenv("DVD_GEMM_NO_T384") || p.debug;
#else
  p.debug = 0; p.stagger = 0; p.stamps = nullptr; p.walk = 0;
  constexpr bool lab_scalar_epi = false, lab_v1 = false, lab_twopass = false, lab_nonpersistent = false, lab_no_t384 = false;
#endif
  {
    auto al = [](const void* q, size_t a) { return ((uintptr_t)q % a) == 0; };
    bool ok = d->N % 8 == 0 && !lab_scalar_epi;
    if (d->C32) ok = ok && d->ldc % 4 == 0 && d->strideC32 % 4 == 0 && al(d->C32, 16);
    if (d->C16) ok = ok && d->ldc16 % 8 == 0 && d->strideC16 % 8 == 0 && al(d->C16, 16);
    if (d->bias && !d->bias_row) ok = ok && al(d->bias, 16) && d->strideBias % 4 == 0;
    if (d->pos) ok = ok && d->ldpos % 4 == 0 && al(d->pos, 16);
    if (d->gate) ok = ok && d->ldgate % 4 == 0 && d->strideGate % 4 == 0 && al(d->gate, 16);
    if (d->res) ok = ok && d->ldres % 4 == 0 && d->strideRes % 4 == 0 && al(d->res, 16);
    p.vec_epilogue = ok ? 1 : 0;
  }
  // large-tile kernel for the big f16 GEMMs (decoder): N a multiple of 256, at least a few row tiles
  // A document must get the same bits whether it is sampled alone or in a batch.  Two ways this holds here: (a) within a
  // problem family the kernel is chosen from (dtype, N, K, split) only, never from M - same kernel, same fp32 summation order;
  // (b) where the choice DOES follow the row count (small_tiles 1 vs 2 below: the engine switches at 16 384 token rows; the
  // 384 x 256 kernel of round 5 vs gemm_nt_big_kernel), the kernels on both sides accumulate every output in the same MFMA
  // sequence and run the same epilogue arithmetic - bit-identical by construction, and tested as such (tests/test_gpu_gemm.py).
  // small_tiles 1: the caller's problem family is small: 128x128 tiles everywhere.  small_tiles 2: the same family with MANY
  // rows (a large batch of small grids): shapes with N % 256 == 0 take the 256x256 kernel in its TWO-SWEEP form - low parts
  // first, scale, high parts, k ascending in 16-deep MFMA steps: exactly the 128x128 kernel's accumulation sequence, so a
  // document still gets the same bits alone (small_tiles 1) and in a large batch (tests/test_gpu_gemm.py) - everything else
  // stays on the 128x128 kernel.
  const bool small = d->small_tiles == 1, small_many = d->small_tiles == 2;
  const bool big = d->dtype == 0 && d->N % 256 == 0 && !lab_v1 && !small;
  if (d->dtype == 0 && d->N % 128 == 0 && d->N % 256 != 0 && !d->A_lo && (!d->B_lo || d->lo_scale == 1.f) &&
      d->K % 32 == 0 && !lab_twopass && !lab_v1 && !small && !small_many) {
    // 256 x 128 tiles for the DiT block's 384-wide GEMMs: (hi, lo) in one pass, or ONE (dithered) weight tensor
    p.ntm = cdiv(d->M, 256); p.ntn = d->N / 128;
    constexpr int LDS = 8 * 16384;                    // 3 stages x 32 KiB, rounded up to the epilogue's 8 x 16 KiB
    static DeviceOnce once_s1;
    if (const auto bit = DeviceOnce::current_bit(); once_s1.need(bit)) {
      (void)hipFuncSetAttribute((const void*)gemm_nt_split128_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
      (void)hipFuncSetAttribute((const void*)gemm_nt_split128_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
      once_s1.done(bit);
    }
    int nblk = p.ntm * p.ntn;
    if (nblk > 256) nblk = 256;
    if (d->B_lo) gemm_nt_split128_kernel<true><<<dim3(nblk, d->batch), 512, LDS, (hipStream_t)stream>>>(p);
    else gemm_nt_split128_kernel<false><<<dim3(nblk, d->batch), 512, LDS, (hipStream_t)stream>>>(p);
    return check_launch("gemm_nt(128-wide)");
  }
  if (big && d->B_lo && !d->A_lo && d->lo_scale == 1.f && d->K % 32 == 0 && !lab_twopass && !small_many) {
    p.ntm = cdiv(d->M, 256); p.ntn = d->N / 256;
    constexpr int LDS = 3 * 3 * 256 * 64;
    static DeviceOnce once_s;
    if (const auto bit = DeviceOnce::current_bit(); once_s.need(bit)) {
      (void)hipFuncSetAttribute((const void*)gemm_nt_split_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
      once_s.done(bit);
    }
    int nblk = p.ntm * p.ntn;
    if (nblk > 256) nblk = 256;
    gemm_nt_split_kernel<true><<<dim3(nblk, d->batch), 512, LDS, (hipStream_t)stream>>>(p);
    return check_launch("gemm_nt(split)");
  }
#ifdef DVD_LAB
  if (big && !d->B_lo && !d->A_lo && d->K % 32 == 0 && getenv("DVD_GEMM_RING")) {
    // lab: the split kernel's 3-stage skewed pipeline on one weight tensor, for A/B runs against gemm_nt_big_kernel
    p.ntm = cdiv(d->M, 256); p.ntn = d->N / 256;
    constexpr int LDS = 3 * 3 * 256 * 64;
    static DeviceOnce once_r;
    if (const auto bit = DeviceOnce::current_bit(); once_r.need(bit)) {
      (void)hipFuncSetAttribute((const void*)gemm_nt_split_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
      once_r.done(bit);
    }
    int nblk = p.ntm * p.ntn;
    if (nblk > 256) nblk = 256;
    gemm_nt_split_kernel<false><<<dim3(nblk, d->batch), 512, LDS, (hipStream_t)stream>>>(p);
    return check_launch("gemm_nt(ring, lab)");
  }
#endif
#ifdef DVD_LAB
  if (big && !d->B_lo && !d->A_lo && p.vec_epilogue && !d->pos && !d->gate && getenv("DVD_GEMM_M16")) {
    // lab: the 16x16x32-MFMA variant of the 256 x 256 kernel (measured 1-7 % slower, see gemm_nt_big16_kernel)
    p.ntm = cdiv(d->M, 256); p.ntn = d->N / 256;
    constexpr int LDS = 2 * 2 * 256 * 128;
    static DeviceOnce once16;
    if (const auto bit = DeviceOnce::current_bit(); once16.need(bit)) {
      (void)hipFuncSetAttribute((const void*)gemm_nt_big16_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
      once16.done(bit);
    }
    int nblk = p.ntm * p.ntn;
    if (nblk > 256) nblk = 256;
    gemm_nt_big16_kernel<<<dim3(nblk, d->batch), 512, LDS, (hipStream_t)stream>>>(p);
    return check_launch("gemm_nt(big16)");
  }
#endif
#ifdef DVD_LAB
  if (big && !d->B_lo && !d->A_lo && d->K % 32 == 0 && p.vec_epilogue && !d->pos && !d->gate && !(d->bias && d->bias_row) &&
      getenv("DVD_GEMM_BIG2")) {
    p.ntm = cdiv(d->M, 256); p.ntn = d->N / 256;
    constexpr int LDS = big2::NBUF * big2::HALF;
    static DeviceOnce once2;
    if (const auto bit = DeviceOnce::current_bit(); once2.need(bit)) {
      (void)hipFuncSetAttribute((const void*)gemm_nt_big2_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
      once2.done(bit);
    }
    int nblk = p.ntm * p.ntn;
    if (nblk > 256) nblk = 256;
    gemm_nt_big2_kernel<<<dim3(nblk, d->batch), 256, LDS, (hipStream_t)stream>>>(p);
    return check_launch("gemm_nt(big2)");
  }
#endif
  if (big && !d->B_lo && !d->A_lo && !small_many && p.vec_epilogue && !d->pos && !d->gate && !(d->bias && d->bias_row) &&
      !(d->res && d->act == 1) && d->K % 128 == 0 && d->K >= 256 && !lab_no_t384) {
    // round 5: 384 x 256 tiles, 4-slot half-slab ring, generated K loop; bit-identical to gemm_nt_big_kernel
    p.ntm = cdiv(d->M, 384); p.ntn = d->N / 256;
    int tdbg = 0;
    p.stagger = T384_STAGGER;
    p.walk = T384_WALK;
#ifdef DVD_LAB
    if (const char* e = getenv("DVD_GEMM_T384_DBG")) tdbg = atoi(e);
    if (const char* e = getenv("DVD_GEMM_T384_STAGGER")) p.stagger = atoi(e);
    if (getenv("DVD_GEMM_T384_PRIO")) p.debug |= 0x100;
    if (const char* e = getenv("DVD_GEMM_T384_WALK")) p.walk = atoi(e);
    if (getenv("DVD_GEMM_T384_NT")) p.debug |= 0x200;
    if (getenv("DVD_GEMM_T384_RES_PHASED")) p.debug |= 0x400;
    if (getenv("DVD_GEMM_T384_NOSTORE")) p.debug |= 0x800;
#endif
    return launch_gemm_t384(p, d->batch, tdbg, stream);
  }
  if (big) {
    p.ntm = cdiv(d->M, 256); p.ntn = d->N / 256;
    constexpr int LDS = 2 * 2 * 256 * 128;
    static DeviceOnce once;
    if (const auto bit = DeviceOnce::current_bit(); once.need(bit)) {
      (void)hipFuncSetAttribute((const void*)gemm_nt_big_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
#ifdef DVD_LAB
      (void)hipFuncSetAttribute((const void*)gemm_nt_big_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
      (void)hipFuncSetAttribute((const void*)gemm_nt_big_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
      (void)hipFuncSetAttribute((const void*)gemm_nt_big_kernel<3>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
      (void)hipFuncSetAttribute((const void*)gemm_nt_big_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
      (void)hipFuncSetAttribute((const void*)gemm_nt_big_kernel<5>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
      (void)hipFuncSetAttribute((const void*)gemm_nt_big_kernel<6>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
      (void)hipFuncSetAttribute((const void*)gemm_nt_big_kernel<7>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
#endif
      once.done(bit);
    }
    int nblk = p.ntm * p.ntn;
    if (nblk > 256 && !lab_nonpersistent) nblk = 256;
    dim3 gridb(nblk, d->batch);
#ifdef DVD_LAB
    if (p.debug == 1) gemm_nt_big_kernel<1><<<gridb, 512, LDS, (hipStream_t)stream>>>(p);
    else if (p.debug == 2) gemm_nt_big_kernel<2><<<gridb, 512, LDS, (hipStream_t)stream>>>(p);
    else if (p.debug == 3) gemm_nt_big_kernel<3><<<gridb, 512, LDS, (hipStream_t)stream>>>(p);
    else if (p.debug == 5) gemm_nt_big_kernel<5><<<gridb, 512, LDS, (hipStream_t)stream>>>(p);
    else if (p.debug == 6) gemm_nt_big_kernel<6><<<gridb, 512, LDS, (hipStream_t)stream>>>(p);
    else if (p.debug == 7) gemm_nt_big_kernel<7><<<gridb, 512, LDS, (hipStream_t)stream>>>(p);
    else if (lab_spread) gemm_nt_big_kernel<4><<<gridb, 512, LDS, (hipStream_t)stream>>>(p);
    else
#endif
    gemm_nt_big_kernel<0><<<gridb, 512, LDS, (hipStream_t)stream>>>(p);
    return check_launch("gemm_nt(big)");
  }
  if (d->dtype == 1 && d->N <= 64 && d->C32 && !d->C16 && !d->pos && !d->gate && !d->res && !d->bias_row &&
      (d->act == 0 || d->act == 2) && !d->A_lo && !d->B_lo) {
    const dim3 grd(cdiv(d->M, 128), d->batch);
    if (d->N <= 32) gemm_f32_narrow_kernel<1><<<grd, 256, 0, (hipStream_t)stream>>>(p);
    else gemm_f32_narrow_kernel<2><<<grd, 256, 0, (hipStream_t)stream>>>(p);
    return check_launch("gemm_nt(f32 narrow)");
  }
  p.ntm = cdiv(d->M, 128); p.ntn = cdiv(d->N, 128);
  dim3 grid(p.ntm * p.ntn, d->batch);
  {
    // f16 problems with few 128 x 128 tiles: the LDS-DMA ring kernel (same bits; see gemm_nt_ring128_kernel)
    auto al16 = [](const void* q) { return ((uintptr_t)q % 16) == 0; };
    bool ring = d->dtype != 1 && d->K % 64 == 0 && d->K >= 128 && d->lda % 8 == 0 && d->ldb % 8 == 0 && d->strideA % 8 == 0 &&
                d->strideB % 8 == 0 && al16(d->A) && al16(d->B) && al16(d->A_lo) && al16(d->B_lo) &&
                (long)p.ntm * p.ntn <= RING128_MAX_TILES && (long)d->M * d->lda < (1l << 30) && (long)d->N * d->ldb < (1l << 30);
#ifdef DVD_LAB
    if (const char* e = getenv("DVD_GEMM_RING128")) ring = ring && atoi(e) != 0;
#endif
    bool ring256 = d->dtype != 1 && d->N % 256 == 0 && d->K % 32 == 0 && d->K >= 128 && d->lda % 8 == 0 && d->ldb % 8 == 0 &&
                   d->strideA % 8 == 0 && d->strideB % 8 == 0 && al16(d->A) && al16(d->B) && al16(d->A_lo) && al16(d->B_lo) &&
                   (long)p.ntm * (d->N / 256) <= RING256_MAX_TILES && (long)p.ntm * (d->N / 256) >= RING256_MIN_TILES &&
                   (long)d->M * d->lda < (1l << 30) &&
                   (long)d->N * d->ldb < (1l << 30);
#ifdef DVD_LAB
    if (const char* e = getenv("DVD_GEMM_RING256")) {       // 0: off; 2: also below RING256_MIN_TILES (tests, A/B runs)
      if (atoi(e) == 0) ring256 = false;
      if (atoi(e) == 2) ring256 = d->dtype != 1 && d->N % 256 == 0 && d->K >= 128 && d->lda % 8 == 0 && d->ldb % 8 == 0 && d->strideA % 8 == 0 &&
                                  d->strideB % 8 == 0 && al16(d->A) && al16(d->B) && al16(d->A_lo) && al16(d->B_lo) &&
                                  (long)p.ntm * (d->N / 256) <= RING256_MAX_TILES;
    }
#endif
    if (ring256) {
      constexpr int LDS = 6 * (128 + 256) * 64;
      static DeviceOnce once_r256;
      if (const auto bit = DeviceOnce::current_bit(); once_r256.need(bit)) {
        (void)hipFuncSetAttribute((const void*)gemm_nt_ring256_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
        once_r256.done(bit);
      }
      p.ntn = d->N / 256;
      gemm_nt_ring256_kernel<<<dim3(p.ntm * p.ntn, d->batch), 512, LDS, (hipStream_t)stream>>>(p);
      return check_launch("gemm_nt(ring256)");
    }
    if (ring) {
      constexpr int LDS = 5 * 2 * 128 * 128;
      static DeviceOnce once_r128;
      if (const auto bit = DeviceOnce::current_bit(); once_r128.need(bit)) {
        (void)hipFuncSetAttribute((const void*)gemm_nt_ring128_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
        once_r128.done(bit);
      }
      gemm_nt_ring128_kernel<<<grid, 256, LDS, (hipStream_t)stream>>>(p);
      return check_launch("gemm_nt(ring128)");
    }
  }
#ifdef DVD_LAB
  if (const char* e = getenv("DVD_GEMM_W8"); e && atoi(e) != 0 && d->dtype != 1) {   // lab: two waves per SIMD on each tile
    gemm_nt_w8_kernel<<<grid, 512, 0, (hipStream_t)stream>>>(p);
    return check_launch("gemm_nt(w8, lab)");
  }
  if (getenv("DVD_GEMM_PD4")) {       // lab: register prefetch four tiles deep (measured: no gain, see gemm_nt_kernel)
    if (d->dtype == 1) gemm_nt_kernel<true, 4><<<grid, 256, 0, (hipStream_t)stream>>>(p);
    else gemm_nt_kernel<false, 4><<<grid, 256, 0, (hipStream_t)stream>>>(p);
    return check_launch("gemm_nt(pd4, lab)");
  }
#endif
  if (d->dtype == 1)
    gemm_nt_kernel<true, 2><<<grid, 256, 0, (hipStream_t)stream>>>(p);
  else
    gemm_nt_kernel<false, 2><<<grid, 256, 0, (hipStream_t)stream>>>(p);
  return check_launch("gemm_nt");
}

int dvd::launch_gemm_conv_f32(const float* a, int ca, const float* b, int cb, int h, int w, int ks, int dil, long rows,
                              const float* wgt, int kp, const float* bias, float* out, int cout, int act, void* stream) {
  DVD_REQUIRE(a && wgt && bias && out && (b || cb == 0), "gemm_conv_f32: null pointer");
  DVD_REQUIRE(ca > 0 && ca % 16 == 0 && cb >= 0 && cb % 16 == 0 && (ks & 1) == 1 && dil > 0 && kp == ks * ks * (ca + cb),
              "gemm_conv_f32: bad shape (ca %d, cb %d, ks %d, kp %d)", ca, cb, ks, kp);
  DVD_REQUIRE(h > 0 && w > 0 && rows > 0 && rows % ((long)h * w) == 0 && rows < (1l << 31) && cout > 0,
              "gemm_conv_f32: bad map (%d x %d, %ld rows)", h, w, rows);
  DVD_REQUIRE(((uintptr_t)a % 16) == 0 && ((uintptr_t)b % 16) == 0 && ((uintptr_t)wgt % 16) == 0,
              "gemm_conv_f32: operands must be 16-byte aligned");
  GemmArgs p;
  memset(&p, 0, sizeof(p));
  p.A = a; p.cv_b = b; p.cv_ca = ca; p.cv_cb = cb; p.cv_h = h; p.cv_w = w; p.cv_ks = ks; p.cv_dil = dil;
  p.B = wgt; p.ldb = kp; p.C32 = out; p.ldc = cout; p.bias = bias; p.act = act; p.lo_scale = 1.f;
  p.M = (int)rows; p.N = cout; p.K = kp;
  p.vec_epilogue = (cout % 8 == 0 && ((uintptr_t)out % 16) == 0 && ((uintptr_t)bias % 16) == 0) ? 1 : 0;
  p.ntm = cdiv(p.M, 128); p.ntn = cdiv(p.N, 128);
  gemm_nt_kernel<true, 2, true><<<dim3(p.ntm * p.ntn, 1), 256, 0, (hipStream_t)stream>>>(p);
  return check_launch("gemm_conv_f32");
}
