// Flash-style multi-head attention for the denoiser (f16 operands, fp32 softmax / accumulate):
//   O = softmax(scale * Q K^T) V      per (batch, head), never materialising the T x T scores.
// Replaces the three attention flavours on the path:
//   nn.MultiheadAttention core   6 heads x 64   (idf/cross_model.py:203-205,237-265)
//   timm Attention core          6 heads x 64   (idf/cross_model.py:163-165,268-289)
//   SATRN ScaledDotProductAttention 6 heads x 256, temperature 16, all-ones mask
//                                               (idf/cross_attn.py:73-83,197-221)
// At the BASELINE grid (T = 20736 tokens) these are 83 % of all FLOPs of a denoiser step.
//
// Structure (one 256-thread workgroup = 4 waves = 128 query rows; one wave owns 32 query rows):
//   * "swapped" first product  S^T[key][q] = K . Q^T  (A = K rows from LDS, B = Q fragments in
//     registers), so the softmax axis (keys) lies in the accumulator registers of a lane and the
//     query on the lane: row max / sum are register reductions + ONE cross-half exchange, and the
//     online-softmax rescale factor is a per-lane scalar.
//   * the S^T accumulator, converted to f16, IS the B operand of the second product
//     O^T[d][q] = V^T . P^T  (guide section 3 "accumulator tile as the next MFMA's operand"); the k-order
//     permutation this implies is absorbed by reading K rows in a bit-swapped order (kappa), so
//     the V^T operand is a plain 16-byte LDS read.
//   * V is consumed TRANSPOSED ([head_dim, T] per batch, keys contiguous): the projection GEMM
//     writes it that way for free by swapping its operands (dvd_gemm_nt with A = W_v).
//   * K / V^T tiles of 64 keys are double-buffered in LDS (rows padded by 16 B: conflict-free
//     ds_read_b128), global loads for tile t+1 are issued before the MFMAs of tile t.
//   * XCD-aware workgroup order: the q-blocks that run concurrently on one XCD belong to the same
//     (batch, head), so its K/V stream is served from that XCD's L2.
#include "common.h"
#include "mfma.h"
#include <stdlib.h>

#ifndef DVD_ATTN64_OCC
#define DVD_ATTN64_OCC 2   /* 3 waves per SIMD fit (140 VGPRs) and measure the same 838 TF/s; 4 spill */
#endif
#ifndef DVD_ATTN64_ROWSUM
#define DVD_ATTN64_ROWSUM 0
#endif
#ifndef DVD_ATTN_PF
#define DVD_ATTN_PF 4
#endif

namespace dvd {

struct AttnArgs {
  const _Float16* Q;
  const _Float16* K;
  const _Float16* Vt;
  _Float16* O;
  long sQ, sK, sVt, sO;  // batch strides (elements)
  int ldq, ldk, ldvt, ldo;
  int heads, batch, tq, tk;
  int kv_div;            // kv batch = b / kv_div
  int nqb;               // query blocks per (b, h)
  float c;               // scale * log2(e)
  unsigned long long* stamps;   // diagnostic builds only
};

__device__ __forceinline__ int kappa(int r) {  // swap bits 2 and 3
  return (r & ~12) | ((r & 4) << 1) | ((r & 8) >> 1);
}

// max over the two 32-lane halves of a wave without the LDS: v_permlane32_swap (new on gfx950) exchanges the upper half
// of its first operand with the lower half of its second; with both = x the two results hold x.lo and x.hi in every
// lane.  (__shfl_xor(x, 32) compiles to ds_bpermute_b32, an LDS instruction whose lgkmcnt wait also drains every
// fragment read in flight.)
__device__ __forceinline__ float half_swap_max(float x) {
  float a = x, b = x;
  asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
  return fmaxf(a, b);
}

template <int D>
__global__ void __launch_bounds__(256, (D == 256 ? 1 : 2)) flash_attn_kernel(AttnArgs p) {
  constexpr int KB = 64;                 // keys per tile
  constexpr int KP = 2 * D + 16;         // K tile row pitch (bytes)
  constexpr int VP = 2 * KB + 16;        // V^T tile row pitch (bytes)
  constexpr int KBYTES = KB * KP, VBYTES = D * VP;
  constexpr int KCH = D / 8;             // 16-B chunks per K row
  constexpr int NK = KB * KCH / 256;     // K chunks per thread   (8 / 2)
  constexpr int NV = D * 8 / 256;        // V^T chunks per thread (8 / 2)
  constexpr int KS = D / 16;             // k-steps of the first product
  constexpr int DT = D / 32;             // 32-row tiles of O^T
  constexpr float RESCALE_THR = 10.f;
  extern __shared__ __attribute__((aligned(16))) char smem[];   // [2][KBYTES + VBYTES]

  const int nwg = gridDim.x;
  int id = blockIdx.x;
  {
    const int q = nwg / 8, rr = nwg % 8, xcd = id % 8, k = id / 8;
    id = (xcd < rr ? xcd * (q + 1) : rr * (q + 1) + (xcd - rr) * q) + k;
  }
  const int qb = id % p.nqb;
  const int bh = id / p.nqb;
  const int head = bh % p.heads, b = bh / p.heads;
  const int kvb = b / p.kv_div;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, h = lane >> 5;

  const _Float16* Qg = p.Q + b * p.sQ + (size_t)head * D;
  const _Float16* Kg = p.K + kvb * p.sK + (size_t)head * D;
  const _Float16* Vg = p.Vt + kvb * p.sVt + (size_t)head * D * p.ldvt;

  // ---- Q fragments (B operand of S^T = K . Q^T): lane (q = r, half h) holds Q[q][16 ks + 8 h + j]
  const int qrow = min(qb * 128 + wave * 32 + r, p.tq - 1);
  half8 qf[KS];
  {
    const _Float16* qp = Qg + (size_t)qrow * p.ldq + 8 * h;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) qf[ks] = *(const half8*)(qp + 16 * ks);
  }

  floatx16 o[DT];
#pragma unroll
  for (int dt = 0; dt < DT; ++dt)
#pragma unroll
    for (int i = 0; i < 16; ++i) o[dt][i] = 0.f;
  float m_run = -1e30f, l_run = 0.f;

  // ---- staging slots
  u32x4 kreg[NK], vreg[NV];
  int k_lofs[NK], v_lofs[NV];
  int k_row[NK], k_col[NK], v_row[NV], v_col[NV];
#pragma unroll
  for (int i = 0; i < NK; ++i) {
    const int cidx = tid + 256 * i;
    k_row[i] = cidx / KCH;
    k_col[i] = (cidx % KCH) * 8;
    k_lofs[i] = k_row[i] * KP + (cidx % KCH) * 16;
  }
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int cidx = tid + 256 * i;
    v_row[i] = cidx / 8;
    v_col[i] = (cidx % 8) * 8;
    v_lofs[i] = KBYTES + v_row[i] * VP + (cidx % 8) * 16;
  }
  const int nt = (p.tk + KB - 1) / KB;

#define DVD_ATTN_GLOAD(t_)                                                                  \
  {                                                                                         \
    const int key0 = (t_) * KB;                                                             \
    _Pragma("unroll") for (int i = 0; i < NK; ++i) {                                        \
      const int kr = min(key0 + k_row[i], p.tk - 1);                                        \
      kreg[i] = *(const u32x4*)(Kg + (size_t)kr * p.ldk + k_col[i]);                        \
    }                                                                                       \
    if (key0 + KB <= p.tk) { /* wave-uniform fast path: no per-chunk predication */         \
      _Pragma("unroll") for (int i = 0; i < NV; ++i)                                        \
        vreg[i] = *(const u32x4*)(Vg + (size_t)v_row[i] * p.ldvt + key0 + v_col[i]);        \
    } else {                                                                                \
      _Pragma("unroll") for (int i = 0; i < NV; ++i) {                                      \
        const int kc = key0 + v_col[i];                                                     \
        u32x4 v = {0u, 0u, 0u, 0u};                                                         \
        if (kc < p.tk) v = *(const u32x4*)(Vg + (size_t)v_row[i] * p.ldvt + kc);            \
        vreg[i] = v;                                                                        \
      }                                                                                     \
    }                                                                                       \
  }
#define DVD_ATTN_LSTORE(buf_)                                                               \
  {                                                                                         \
    char* base = smem + (buf_) * (KBYTES + VBYTES);                                         \
    _Pragma("unroll") for (int i = 0; i < NK; ++i) *(u32x4*)(base + k_lofs[i]) = kreg[i];   \
    _Pragma("unroll") for (int i = 0; i < NV; ++i) *(u32x4*)(base + v_lofs[i]) = vreg[i];   \
  }

  DVD_ATTN_GLOAD(0)
  DVD_ATTN_LSTORE(0)
  __syncthreads();

  const int kr_ofs = kappa(r) * KP + 16 * h;   // byte offset of this lane's K fragment (kb = 0, ks = 0)
  const int vr_ofs = KBYTES + r * VP + 16 * h; // byte offset of this lane's V^T fragment (dt = 0, kb = 0, s = 0)
  int cur = 0;
  for (int t = 0; t < nt; ++t) {
    const bool more = t + 1 < nt;
    if (more) DVD_ATTN_GLOAD(t + 1)
    const char* base = smem + cur * (KBYTES + VBYTES);

    // ---- S^T = K . Q^T   (two 32-key blocks)
    floatx16 s[2];
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
      for (int i = 0; i < 16; ++i) s[kb][i] = 0.f;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        const half8 kf = *(const half8*)(base + kr_ofs + kb * 32 * KP + ks * 32);
        s[kb] = mfma32_f16(kf, qf[ks], s[kb]);
      }
    }

    // ---- online softmax over the 64 keys of this tile (query = lane column)
    if (t == nt - 1 && (p.tk % KB) != 0) {
      const int key0 = t * KB;
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int i = 0; i < 16; ++i)
          if (key0 + 32 * kb + kappa(cd_row(i, h)) >= p.tk) s[kb][i] = -1e30f;
    }
    float mx = -1e30f;
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int i = 0; i < 16; ++i) mx = fmaxf(mx, s[kb][i]);
    mx *= p.c;                                  // c > 0
    mx = half_swap_max(mx);
    // Deferred rescale (guide T13): keep the running max stale while this tile's max exceeds it by less than
    // RESCALE_THR (log2 units).  P then lies in (0, 2^THR] instead of (0, 1]: f16 keeps the same RELATIVE
    // precision there, and the fp32 accumulators O^T / l have ample range.  The decision is wave-uniform and
    // taken BEFORE this tile's P is formed, so O^T, l and P always share one reference max.
    if (__any(mx - m_run > RESCALE_THR)) {
      const float m_new = fmaxf(m_run, mx);
      const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
      m_run = m_new;
      l_run *= alpha;
#pragma unroll
      for (int dt = 0; dt < DT; ++dt)
#pragma unroll
        for (int i = 0; i < 16; ++i) o[dt][i] *= alpha;
    }
    float rs = 0.f;
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const float pv = __builtin_amdgcn_exp2f(fmaf(s[kb][i], p.c, -m_run));
        s[kb][i] = pv;
        rs += pv;
      }
    l_run += rs;                                // per-lane partial (this half's keys)

    // ---- O^T += V^T . P^T
    half8 pf[2][2];
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) pf[kb][s2] = pack_acc_f16(s[kb], s2);
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) {
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
          const half8 vf = *(const half8*)(base + vr_ofs + dt * 32 * VP + (32 * kb + 16 * s2) * 2);
          o[dt] = mfma32_f16(vf, pf[kb][s2], o[dt]);
        }
    }

    if (more) DVD_ATTN_LSTORE(cur ^ 1)
    __syncthreads();
    cur ^= 1;
  }

  // ---- epilogue: O[q][d] = O^T[d][q] / l
  const float l_tot = l_run + __shfl_xor(l_run, 32);
  const float inv = 1.f / l_tot;
  const int qglob = qb * 128 + wave * 32 + r;
  if (qglob < p.tq) {
    _Float16* op = p.O + b * p.sO + (size_t)qglob * p.ldo + (size_t)head * D;
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) {
        half4 v;
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = (_Float16)(o[dt][4 * g4 + j] * inv);
        *(half4*)(op + 32 * dt + 8 * g4 + 4 * h) = v;
      }
  }
}


// Direct global->LDS loads issued from inline asm (guide 5.7): hipcc then does not know an LDS-DMA is in
// flight - with the builtin form its waitcnt pass degrades EVERY LDS wait in the loop to lgkmcnt(0), which
// serialises the fragment-read pipeline - and the SGPR-base + 32-bit-VGPR-offset address form costs no VALU.
// N loads of 1 KiB each to LDS addresses lds, lds+1024, ...; completion is tracked by the caller's vmcnt(0).
template <int N>
__device__ __forceinline__ void glds_group(const char* gbase, const unsigned (&voff)[N], unsigned lds) {
  static_assert(N == 2 || N == 8, "unsupported group size");
  unsigned keep;
  if constexpr (N == 8) {
    asm volatile(
        "s_mov_b32 %0, m0\n\t"
        "s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %3, %1\n\t"
        "s_add_u32 m0, %2, 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %4, %1\n\t"
        "s_add_u32 m0, %2, 0x800\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %5, %1\n\t"
        "s_add_u32 m0, %2, 0xc00\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %6, %1\n\t"
        "s_add_u32 m0, %2, 0x1000\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %7, %1\n\t"
        "s_add_u32 m0, %2, 0x1400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %8, %1\n\t"
        "s_add_u32 m0, %2, 0x1800\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %9, %1\n\t"
        "s_add_u32 m0, %2, 0x1c00\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %10, %1\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep)
        : "s"(gbase), "s"(lds), "v"(voff[0]), "v"(voff[1]), "v"(voff[2]), "v"(voff[3]), "v"(voff[4]), "v"(voff[5]),
          "v"(voff[6]), "v"(voff[7])
        : "memory", "scc");
  } else {
    asm volatile(
        "s_mov_b32 %0, m0\n\t"
        "s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %3, %1\n\t"
        "s_add_u32 m0, %2, 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %4, %1\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep)
        : "s"(gbase), "s"(lds), "v"(voff[0]), "v"(voff[1])
        : "memory", "scc");
  }
}

// One 1-KiB LDS-DMA load (see glds_group).  Issued one at a time between MFMAs: sixteen of them back to back at
// the top of a tile cost ~120 cycles EACH with the matrix pipe idle (36 % of the tile, s_memtime stamps);
// spread over the tile the memory pipeline drains between them and most of the issue time hides under an MFMA.
__device__ __forceinline__ void glds_one(const char* gbase, unsigned voff, unsigned lds) {
  unsigned keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\t"
      "s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %3, %1\n\t"
      "s_mov_b32 m0, %0"
      : "=&s"(keep)
      : "s"(gbase), "s"(lds), "v"(voff)
      : "memory");
}


// O^T *= alpha with the accumulators LEFT IN AGPRs.  Written as `o[dt][i] *= alpha` the compiler keeps O in AGPRs for
// the MFMAs but copies all 128 registers to VGPRs at the top of EVERY tile for the (rare) rescale branch
// (128 v_accvgpr_read per tile, ~10 % of the tile); here the round trip exists only inside the branch.
__device__ __forceinline__ void scale_acc_in_agpr(floatx16& acc, float alpha) {
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    float x = acc[i], tmp;
    asm volatile("v_accvgpr_read_b32 %1, %0\n\tv_mul_f32 %1, %1, %2\n\ts_nop 0\n\tv_accvgpr_write_b32 %0, %1"
                 : "+a"(x), "=&v"(tmp)
                 : "v"(alpha));
    acc[i] = x;
  }
}

// ================================================================================================
// Fast path (tk % 64 == 0: every shape the engine produces): K / V^T tiles go global -> LDS DIRECTLY
// (global_load_lds_dwordx4, 1 KiB per wave-instruction, no staging registers, no ds_write).  The LDS
// image of such a load is lane-linear, so bank conflicts are avoided by an XOR swizzle applied to the
// per-lane SOURCE address and, identically, to the fragment reads (guide rule 21):
//   512-byte rows (K, head_dim 256): 16-byte chunk c of row r lives at chunk  c ^ (r & 15)
//   128-byte rows (K at head_dim 64, V^T always):                  at chunk  c ^ ((r >> 1) & 7)
// which makes the 16 rows of every ds_read_b128 lane group hit 16 distinct 16-byte bank slots.
// The swizzled read offsets are per-lane constants precomputed once (KS + 4 VGPRs).
// Without staging registers the wave's working set (Q 64 + S^T 32 + P 16 + fragments) fits the 256
// architectural VGPRs and the 128 O^T accumulators stay in AGPRs untouched by the VALU: the v1
// structure spent ~400 v_accvgpr moves per tile shuffling spilled state.
// ================================================================================================
template <int D, int DBG>   // DBG bit 0: accumulate per-phase s_memtime deltas (DVD_ATTN_DEBUG=1); bit 1: bulk load issue (DVD_ATTN_BULK=1)
__global__ void __launch_bounds__(256, (D == 256 ? 1 : DVD_ATTN64_OCC)) flash_attn_glds_kernel(AttnArgs p) {
  constexpr int KB = 64;
  constexpr int KROWB = 2 * D;             // K row bytes (512 / 128)
  constexpr int KCPR = KROWB / 16;         // chunks per K row (32 / 8)
  constexpr int KBYTES = KB * KROWB;       // 32768 / 8192
  constexpr int VBYTES = D * 128;          // V^T tile: D rows x 64 keys x 2 B
  constexpr int BUF = KBYTES + VBYTES;
  constexpr int KINST = KBYTES / 4096;     // 1-KiB loads per wave per tile (8 / 2)
  constexpr int VINST = VBYTES / 4096;
  constexpr int KS = D / 16;
  constexpr int DT = D / 32;
  constexpr float RESCALE_THR = 10.f;
  extern __shared__ __attribute__((aligned(16))) char smem[];   // [2][BUF]
  typedef const __attribute__((address_space(1))) void* gptr_t;
  typedef __attribute__((address_space(3))) void* lptr_t;

  const int nwg = gridDim.x;
  int id = blockIdx.x;
  {
    const int q = nwg / 8, rr = nwg % 8, xcd = id % 8, k = id / 8;
    id = (xcd < rr ? xcd * (q + 1) : rr * (q + 1) + (xcd - rr) * q) + k;
  }
  const int qb = id % p.nqb;
  const int bh = id / p.nqb;
  const int head = bh % p.heads, b = bh / p.heads;
  const int kvb = b / p.kv_div;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 31, h = lane >> 5;

  const _Float16* Qg = p.Q + b * p.sQ + (size_t)head * D;
  const char* Kg = (const char*)(p.K + kvb * p.sK + (size_t)head * D);
  const char* Vg = (const char*)(p.Vt + kvb * p.sVt + (size_t)head * D * p.ldvt);

  const int qrow = min(qb * 128 + wave * 32 + r, p.tq - 1);
  half8 qf[KS];
  {
    const _Float16* qp = Qg + (size_t)qrow * p.ldq + 8 * h;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) qf[ks] = *(const half8*)(qp + 16 * ks);
  }
  // The compiler cannot see the inline-asm vmcnt(0) below; left alone it sinks its own waits for these KS loads into
  // the tile loop as vmcnt(15), vmcnt(14), ... before the MFMAs that first use qf[ks] - which, vmcnt being in-order,
  // makes every tile's S^T phase wait for the LDS-DMA loads of the NEXT tile issued just before it.  Retire the Q
  // loads here with a wait the compiler does track (vmcnt(0); expcnt / lgkmcnt untouched).
  __builtin_amdgcn_s_waitcnt(0x0F70);
  // head_dim 64 is VALU-issue-bound (per 64-key tile and wave: 16 MFMAs against 32 v_exp + ~100 other VALU, two
  // waves per SIMD), so the softmax argument  s * c - m  is taken off the VALU entirely: Q is scaled by c = scale *
  // log2(e) once per workgroup (fp32 multiply, one f16 rounding), and -m_run enters through the C operand of the first
  // MFMA of each S^T chain (a 16-register block `negm`, rewritten only in the rare rescale branch) - the accumulators
  // come out as  s * c - m_run  and feed v_exp directly: 32 v_fma per tile (15 % of the issue slots) are gone.
  constexpr bool NEGM = (D == 64);
  if constexpr (NEGM) {
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
#pragma unroll
      for (int j = 0; j < 8; ++j) qf[ks][j] = (_Float16)((float)qf[ks][j] * p.c);
  }

  // ---- per-lane SOURCE offsets (bytes) of the direct-to-LDS loads, swizzled
  unsigned koff[KINST], voff[VINST];
#pragma unroll
  for (int i = 0; i < KINST; ++i) {
    const int q = (KINST * wave + i) * 64 + lane;
    const int row = q / KCPR, pos = q % KCPR;
    const int f = (KCPR == 32) ? (row & 15) : ((row >> 1) & 7);
    koff[i] = (unsigned)row * (unsigned)(p.ldk * 2) + (unsigned)((pos ^ f) * 16);
  }
#pragma unroll
  for (int i = 0; i < VINST; ++i) {
    const int q = (VINST * wave + i) * 64 + lane;
    const int row = q / 8, pos = q % 8;
    voff[i] = (unsigned)row * (unsigned)(p.ldvt * 2) + (unsigned)((pos ^ ((row >> 1) & 7)) * 16);
  }
  // ---- per-lane fragment READ offsets, same swizzle
  const int kr = kappa(r);
  const int fk = (KCPR == 32) ? (kr & 15) : ((kr >> 1) & 7);
  int kfrag[KS];
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) kfrag[ks] = kr * KROWB + (((2 * ks + h) ^ fk) * 16);
  const int fv = (r >> 1) & 7;
  int vfrag[4];
#pragma unroll
  for (int c = 0; c < 4; ++c) vfrag[c] = KBYTES + r * 128 + (((2 * c + h) ^ fv) * 16);   // c = 2 kb + s2

  floatx16 o[DT];
#pragma unroll
  for (int dt = 0; dt < DT; ++dt)
#pragma unroll
    for (int i = 0; i < 16; ++i) o[dt][i] = 0.f;
  float m_run = NEGM ? 0.f : -1e30f, l_run = 0.f;
  floatx16 negm;                    // NEGM: all 16 registers = -m_run (the lane's query row owns all of them)
#pragma unroll
  for (int i = 0; i < 16; ++i) negm[i] = 0.f;
  // EXPERIMENT (off): take the softmax row sums off the VALU - an all-ones V^T row block makes one extra MFMA per P chunk
  // accumulate sum_k P[k, q] into osum (4 MFMAs per tile instead of 32 v_add_f32).  Correct, but head_dim 64 measured
  // the same with it (861 vs 895 TF/s, inside the box-to-box noise), as it did with two query row blocks per wave and
  // with three waves per SIMD: none of LDS traffic, barrier count, VALU adds or occupancy is what holds it at ~36 %.
  constexpr bool MFMA_ROWSUM = (D == 64) && DVD_ATTN64_ROWSUM;
  floatx16 osum;
  half8 ones8;
#pragma unroll
  for (int i = 0; i < 16; ++i) osum[i] = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i) ones8[i] = (_Float16)1.f;
  const int nt = p.tk / KB;
  const size_t ktile = (size_t)KB * p.ldk * 2;   // bytes between K tiles

  const unsigned lds0 = (unsigned)(uintptr_t)(lptr_t)smem;   // LDS byte address of the dynamic region
#define DVD_GLDS_ISSUE(t_, buf_)                                                                           \
  {                                                                                                        \
    glds_group<KINST>(Kg + (size_t)(t_) * ktile, koff, lds0 + (buf_) * BUF + (KINST * wave) * 1024);        \
    glds_group<VINST>(Vg + (size_t)(t_) * (KB * 2), voff, lds0 + (buf_) * BUF + KBYTES + (VINST * wave) * 1024); \
  }

  DVD_GLDS_ISSUE(0, 0)
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  // ---- main loop.  The instruction stream is pinned with sched_barrier(0) into groups of
  //      { 1 MFMA, 1 LDS fragment read for the MFMA four groups later, a few VALU } so that
  //      (a) four ds_read_b128 are always in flight ahead of their consumer (hipcc otherwise emits
  //          read -> lgkmcnt(0) -> MFMA pairs and the matrix pipe idles on LDS latency), the compiler
  //          still places the counted s_waitcnt and every MFMA hazard itself;
  //      (b) the exp2 / sum / f16-pack of P chunk c+1 issues in the gaps of the 8 MFMAs of chunk c.
#define SB() __builtin_amdgcn_sched_barrier(0)
// S^T step i works on key block (i & 1) and k-step (i >> 1): the two 16-deep accumulation chains alternate, so
// consecutive MFMAs never depend on each other (a single dependent chain ran at ~58 cycles per MFMA, not 32)
#define KLOAD(i_) fr[(i_) & FM] = *(const half8*)(base + kfrag[(i_) >> 1] + ((i_) & 1) * 32 * KROWB)
#define VLOAD(j_) fr[(j_) & FM] = *(const half8*)(base + vfrag[(j_) / DT] + ((j_) % DT) * 32 * 128)
  // fragment reads in flight ahead of their MFMA: LDS latency grows under the concurrent LDS-DMA writes
  constexpr int PF = (D == 256) ? DVD_ATTN_PF : 4, FM = 2 * PF - 1;
  constexpr int NS = 2 * KS;     // MFMAs of the first product per tile
  constexpr int NP = 4 * DT;     // MFMAs of the second product per tile
  int cur = 0;
  unsigned long long acc_t[5] = {0, 0, 0, 0, 0};
#define STAMP(k_)                                                              \
  if constexpr (DBG & 1) {                                                    \
    __builtin_amdgcn_sched_barrier(0);                                         \
    const unsigned long long now_ = __builtin_amdgcn_s_memtime();              \
    __builtin_amdgcn_s_waitcnt(0xC07F);                                        \
    acc_t[k_] += now_ - tprev;                                                 \
    tprev = now_;                                                              \
    __builtin_amdgcn_sched_barrier(0);                                         \
  }
  unsigned long long tprev = 0;
  if constexpr (DBG & 1) tprev = __builtin_amdgcn_s_memtime();
  // Next-tile loads.  SPREAD (default): one 1-KiB LDS-DMA load after every 3rd MFMA, S^T product first - the CU's
  // L2->LDS path moves ~33 B/clk, so the 16 loads of a wave issued back to back stall ~120 cycles EACH with the
  // matrix pipe idle (30 % of the tile); spread over the tile the path is ~60 % busy and an issue rarely stalls.  The
  // tile index is clamped instead of branching (the last tile is re-loaded into the idle buffer, harmlessly).
  // BULK (DBG 2/3, DVD_ATTN_BULK=1): the burst at the top of the tile, kept for A/B runs.
  constexpr bool SPREAD = (DBG < 2);
  constexpr bool STAMPS = (DBG & 1);
  for (int t = 0; t < nt; ++t) {
    if constexpr (!SPREAD) {
      if (t + 1 < nt) DVD_GLDS_ISSUE(t + 1, cur ^ 1)
    }
    const int tn = min(t + 1, nt - 1);
    const char* kg_next = Kg + (size_t)tn * ktile;
    const char* vg_next = Vg + (size_t)tn * (KB * 2);
    const unsigned lds_next = lds0 + (cur ^ 1) * BUF;
#define GLDS_K(i_) glds_one(kg_next, koff[i_], lds_next + (KINST * wave + (i_)) * 1024)
#define GLDS_V(i_) glds_one(vg_next, voff[i_], lds_next + KBYTES + (VINST * wave + (i_)) * 1024)
    // load g of the tile (K loads first) goes after MFMA number GAP*g + GAP-1 of the tile's 64 (S^T then PV): all are
    // issued in the first ~3/4 of the tile so that the last has landed (L2 latency ~500 cycles) by the tile's barrier
    constexpr int NG = KINST + VINST, GAP = 3;
#define GLDS_ANY(g_) if ((g_) < KINST) { GLDS_K((g_) < KINST ? (g_) : 0); } else { GLDS_V((g_) >= KINST ? (g_) - KINST : 0); }
    STAMP(0)
    const char* base = smem + cur * BUF;
    half8 fr[FM + 1];
    floatx16 s[2];
    if constexpr (!NEGM) {
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int i = 0; i < 16; ++i) s[kb][i] = 0.f;
    }
#pragma unroll
    for (int i = 0; i < PF; ++i) KLOAD(i);
    SB();
#pragma unroll
    for (int i = 0; i < NS; ++i) {
      if (NEGM && i < 2) {
        // D = A.B + (-m_run) with C != D: issued from inline asm (early-clobber D), because the builtin makes hipcc copy
        // negm into the second chain's accumulator first (8 v_mov_b64 per tile).  The consumer of each result is the
        // chain's next MFMA, two MFMAs later, reading it as SrcC at exactly the same registers.
        asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %3" : "=&v"(s[i & 1]) : "v"(fr[i & FM]), "v"(qf[i >> 1]), "v"(negm));
      } else {
        s[i & 1] = mfma32_f16(fr[i & FM], qf[i >> 1], s[i & 1]);
      }
      if (i + PF < NS) { KLOAD(i + PF); } else { VLOAD(i + PF - NS); }
      if constexpr (SPREAD) {
        if (i % GAP == GAP - 1 && i / GAP < NG) { GLDS_ANY(i / GAP) }
      }
      SB();
    }

    STAMP(1)
    float mx = -1e30f;
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int i = 0; i < 16; ++i) mx = fmaxf(mx, s[kb][i]);
    if constexpr (!NEGM) mx *= p.c;
    mx = half_swap_max(mx);                    // NEGM: already (row max) * c - m_run
    bool resc;
    if constexpr (NEGM) resc = t == 0 || __any(mx > RESCALE_THR);    // the first tile always sets the reference maximum
    else resc = __any(mx - m_run > RESCALE_THR);
    if (resc) {                                // deferred rescale, see flash_attn_kernel
      float alpha;
      if constexpr (NEGM) {
        const float delta = t == 0 ? mx : fmaxf(mx, 0.f);
        alpha = __builtin_amdgcn_exp2f(-delta);
        m_run += delta;
#pragma unroll
        for (int i = 0; i < 16; ++i) negm[i] = -m_run;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
          for (int i = 0; i < 16; ++i) s[kb][i] -= delta;          // this tile's scores were formed against the old m_run
      } else {
        const float m_new = fmaxf(m_run, mx);
        alpha = __builtin_amdgcn_exp2f(m_run - m_new);
        m_run = m_new;
      }
      l_run *= alpha;
      if constexpr (MFMA_ROWSUM) {
#pragma unroll
        for (int i = 0; i < 16; ++i) osum[i] *= alpha;
      }
      if constexpr (D == 256) {
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) scale_acc_in_agpr(o[dt], alpha);
      } else {                      // head_dim 64: the 32 accumulators live in VGPRs, plain VALU multiply
#pragma unroll
        for (int dt = 0; dt < DT; ++dt)
#pragma unroll
          for (int i = 0; i < 16; ++i) o[dt][i] *= alpha;
      }
    }
    float rs = 0.f;
    half8 pf[4];
    // chunk c of P = registers 8 (c & 1) .. +7 of s[c >> 1]
#define PEXP(c_, e_)                                                                             \
  {                                                                                              \
    const float sv_ = s[(c_) >> 1][8 * ((c_) & 1) + (e_)];                                       \
    const float pv_ = __builtin_amdgcn_exp2f(NEGM ? sv_ : fmaf(sv_, p.c, -m_run));               \
    if constexpr (!MFMA_ROWSUM) rs += pv_;                                                       \
    pf[c_][e_] = (_Float16)pv_;                                                                  \
  }
#pragma unroll
    for (int e = 0; e < 8; ++e) PEXP(0, e)
    STAMP(2)
    SB();
#pragma unroll
    for (int j = 0; j < NP; ++j) {
      const int c = j / DT, dt = j % DT;
      o[dt] = mfma32_f16(fr[j & FM], pf[c], o[dt]);
      if constexpr (MFMA_ROWSUM) {
        if (dt == DT - 1) osum = mfma32_f16(ones8, pf[c], osum);
      }
      if (j + PF < NP) VLOAD(j + PF);
      if (c < 3) {                          // P chunk c+1: DT MFMA gaps for 8 elements
        constexpr int per = (8 + DT - 1) / DT;
#pragma unroll
        for (int e = dt * per; e < (dt + 1) * per && e < 8; ++e) PEXP(c + 1, e)
      }
      if constexpr (SPREAD) {
        if ((NS + j) % GAP == GAP - 1 && (NS + j) / GAP < NG) { GLDS_ANY((NS + j) / GAP) }
      }
      SB();
    }
    l_run += rs;
    STAMP(3)

    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // next tile has landed (this wave's loads) ...
    __syncthreads();                                   // ... and everyone's; also: all reads of `cur` are done
    STAMP(4)
    cur ^= 1;
  }
  if constexpr (DBG & 1) {
    if (lane == 0 && p.stamps) {
      unsigned long long* o_ = p.stamps + ((size_t)blockIdx.x * 4 + wave) * 5;
      for (int k = 0; k < 5; ++k) o_[k] = acc_t[k];
    }
  }
#undef STAMP
#undef GLDS_K
#undef GLDS_V
#undef GLDS_ANY
#undef SB
#undef KLOAD
#undef VLOAD
#undef PEXP

  const float l_tot = MFMA_ROWSUM ? osum[0] : l_run + __shfl_xor(l_run, 32);
  const float inv = 1.f / l_tot;
  const int qglob = qb * 128 + wave * 32 + r;
  if (qglob < p.tq) {
    _Float16* op = p.O + b * p.sO + (size_t)qglob * p.ldo + (size_t)head * D;
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) {
        half4 v;
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = (_Float16)(o[dt][4 * g4 + j] * inv);
        *(half4*)(op + 32 * dt + 8 * g4 + 4 * h) = v;
      }
  }
}


// ================================================================================================
// head_dim 256, TWO 32-row query blocks per wave, 32-key tiles ("r64").  flash_attn_glds_kernel<256> is bound by the CU's
// LDS port: every MFMA of every wave needs its own 1-KiB fragment (4 waves x 1 KiB per 32-cycle MFMA = 128 B/clk = the
// port's peak; benchmarks/lab/mix_lab.hip: 33.2 cycles per MFMA with fragment reads alone, 42.9 with the tile refill on
// top).  Here one K / V^T fragment read feeds TWO MFMAs (query rows r and 32 + r), which needs O^T for 64 rows = all 256
// AGPRs and Q for 64 rows = 128 VGPRs; the rest fits only because the key tile is halved to 32 keys (S^T 32 + P 16
// registers for both row blocks) and the swizzled fragment offsets are recomputed with one v_xor instead of being kept.
// A workgroup = 4 waves = 256 query rows, so K / V^T are also streamed from L2 half as often.  Same fragment maps,
// deferred-rescale online softmax (one running max / sum per row block) and per-element arithmetic as the 32-row kernel.
// LDS image: K tile 32 keys x 512 B, chunk c of row r at c ^ (r & 15); V^T tile 256 rows x 64 B, chunk c of row r at
// c ^ ((r >> 2) & 3); two 32-KiB buffers.
// ================================================================================================
template <int DBG>
__global__ void __launch_bounds__(256, 1) flash_attn_r64_kernel(AttnArgs p) {
  constexpr int D = 256, KB = 32, KROWB = 512, KBYTES = KB * KROWB, VROWB = 64, VBYTES = D * VROWB, BUF = KBYTES + VBYTES;
  constexpr int KS = 16, DT = 8, RB = 2;
  constexpr float RESCALE_THR = 10.f;
  extern __shared__ __attribute__((aligned(16))) char smem[];   // [2][BUF]
  typedef __attribute__((address_space(3))) void* lptr_t;

  const int nwg = gridDim.x;
  int id = blockIdx.x;
  {
    const int q = nwg / 8, rr = nwg % 8, xcd = id % 8, k = id / 8;
    id = (xcd < rr ? xcd * (q + 1) : rr * (q + 1) + (xcd - rr) * q) + k;
  }
  const int qb = id % p.nqb;          // 256-row query blocks
  const int bh = id / p.nqb;
  const int head = bh % p.heads, b = bh / p.heads;
  const int kvb = b / p.kv_div;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 31, h = lane >> 5;

  const _Float16* Qg = p.Q + b * p.sQ + (size_t)head * D;
  const char* Kg = (const char*)(p.K + kvb * p.sK + (size_t)head * D);
  const char* Vg = (const char*)(p.Vt + kvb * p.sVt + (size_t)head * D * p.ldvt);

  half8 qf[RB][KS];
#pragma unroll
  for (int rb = 0; rb < RB; ++rb) {
    const int qrow = min(qb * 256 + wave * 64 + rb * 32 + r, p.tq - 1);
    const _Float16* qp = Qg + (size_t)qrow * p.ldq + 8 * h;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) qf[rb][ks] = *(const half8*)(qp + 16 * ks);
  }
  __builtin_amdgcn_s_waitcnt(0x0F70);   // retire the Q loads with a wait the compiler tracks (see flash_attn_glds_kernel)

  // per-lane SOURCE offsets of this wave's 4 + 4 direct-to-LDS loads per tile
  unsigned koff[4], voff[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int q = (4 * wave + i) * 64 + lane;
    const int krow = q >> 5, kpos = q & 31;
    koff[i] = (unsigned)krow * (unsigned)(p.ldk * 2) + (unsigned)((kpos ^ (krow & 15)) * 16);
    const int vrow = q >> 2, vpos = q & 3;
    voff[i] = (unsigned)vrow * (unsigned)(p.ldvt * 2) + (unsigned)((vpos ^ ((vrow >> 2) & 3)) * 16);
  }
  // fragment read offsets, recomputed per read from two per-lane bases:
  //   K, k-step ks :  kr*512 + (((2 ks + h) ^ (kr & 15)) * 16)   =  kbase ^ (ks * 32)      (kr & 15 only touches bits 4..7,
  //                                                                   2 ks * 16 = ks * 32 touches bits 5..8: plain XOR)
  //   V^T, chunk c, d block dt :  KBYTES + (32 dt + r)*64 + (((2 c + h) ^ ((r >> 2) & 3)) * 16)  =  (vbase ^ (c * 32)) + dt * 2048
  const int kr = kappa(r);
  const int kbase = kr * KROWB + ((h ^ (kr & 15)) * 16);
  const int vbase = KBYTES + r * VROWB + ((h ^ ((r >> 2) & 3)) * 16);

  floatx16 o[RB][DT];
#pragma unroll
  for (int rb = 0; rb < RB; ++rb)
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
      for (int i = 0; i < 16; ++i) o[rb][dt][i] = 0.f;
  float m_run[RB] = {-1e30f, -1e30f}, l_run[RB] = {0.f, 0.f};
  const int nt = p.tk / KB;
  const size_t ktile = (size_t)KB * p.ldk * 2;
  const unsigned lds0 = (unsigned)(uintptr_t)(lptr_t)smem;

#pragma unroll
  for (int i = 0; i < 4; ++i) {
    glds_one(Kg, koff[i], lds0 + (4 * wave + i) * 1024);
    glds_one(Vg, voff[i], lds0 + KBYTES + (4 * wave + i) * 1024);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

#define SB() __builtin_amdgcn_sched_barrier(0)
#define KFRAG(ks_) (*(const half8*)(base + (kbase ^ ((ks_) * 32))))
#define VFRAG(c_, dt_) (*(const half8*)(base + (vbase ^ ((c_) * 32)) + (dt_) * 2048))
  unsigned long long acc_t[5] = {0, 0, 0, 0, 0};
  unsigned long long tprev = 0;
#define STAMP(k_)                                                              \
  if constexpr (DBG & 1) {                                                     \
    __builtin_amdgcn_sched_barrier(0);                                         \
    const unsigned long long now_ = __builtin_amdgcn_s_memtime();              \
    __builtin_amdgcn_s_waitcnt(0xC07F);                                        \
    acc_t[k_] += now_ - tprev;                                                 \
    tprev = now_;                                                              \
    __builtin_amdgcn_sched_barrier(0);                                         \
  }
  if constexpr (DBG & 1) tprev = __builtin_amdgcn_s_memtime();
  int cur = 0;
  for (int t = 0; t < nt; ++t) {
    const int tn = min(t + 1, nt - 1);
    const char* kg_next = Kg + (size_t)tn * ktile;
    const char* vg_next = Vg + (size_t)tn * (KB * 2);
    const unsigned lds_next = lds0 + (cur ^ 1) * BUF + (4 * wave) * 1024;
    const char* base = smem + cur * BUF;
    STAMP(0)
    half8 fr[4];
    floatx16 s[RB];
#pragma unroll
    for (int f = 0; f < 4; ++f) fr[f] = KFRAG(f);
    SB();
    // ---- S^T = K.Q^T : 16 K fragments, two MFMAs each; the next tile's 8 loads go out after every second fragment
#pragma unroll
    for (int f = 0; f < KS; ++f) {
      // S^T accumulates in ARCHITECTURAL VGPRs, by inline asm: left to the compiler these MFMAs get AGPR destinations,
      // and as O^T owns all 256 AGPRs it then evicts two O^T accumulators to VGPRs and back every tile (64 moves).
      // The two chains alternate, so a dependent MFMA is always one independent 8-pass MFMA behind its producer.
      if (f == 0) {      // first step: C = 0 as an inline constant instead of 32 v_mov per tile
        asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, 0" : "=v"(s[0]) : "v"(fr[0]), "v"(qf[0][0]));
        asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, 0" : "=v"(s[1]) : "v"(fr[0]), "v"(qf[1][0]));
      } else {
        asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(s[0]) : "v"(fr[f & 3]), "v"(qf[0][f]));
        asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(s[1]) : "v"(fr[f & 3]), "v"(qf[1][f]));
      }
      if (f + 4 < KS) fr[f & 3] = KFRAG(f + 4);
      else fr[f & 3] = VFRAG(0, f + 4 - KS);
      if (f & 1) {
        if (f < 8) glds_one(kg_next, koff[f >> 1], lds_next + (f >> 1) * 1024);
        else glds_one(vg_next, voff[(f >> 1) - 4], lds_next + KBYTES + ((f >> 1) - 4) * 1024);
      }
      SB();
    }
    asm volatile("s_nop 15\n\ts_nop 7" ::: "memory");      // the last two MFMAs' results must have landed before the VALU reads them
    STAMP(1)
    // ---- online softmax, per row block
    float mx[RB];
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) {
      float m = -1e30f;
#pragma unroll
      for (int i = 0; i < 16; ++i) m = fmaxf(m, s[rb][i]);
      mx[rb] = half_swap_max(m * p.c);
    }
    if (__any(fmaxf(mx[0] - m_run[0], mx[1] - m_run[1]) > RESCALE_THR)) {     // deferred rescale
#pragma unroll
      for (int rb = 0; rb < RB; ++rb) {
        const float m_new = fmaxf(m_run[rb], mx[rb]);
        const float alpha = __builtin_amdgcn_exp2f(m_run[rb] - m_new);
        m_run[rb] = m_new;
        l_run[rb] *= alpha;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) scale_acc_in_agpr(o[rb][dt], alpha);
      }
    }
    float rs[RB] = {0.f, 0.f};
    half8 pf[RB][2];
#define PEXP4(rb_, c_, e_)                                                                          \
  {                                                                                                 \
    const float pv_ = __builtin_amdgcn_exp2f(fmaf(s[rb_][8 * (c_) + (e_)], p.c, -m_run[rb_]));      \
    rs[rb_] += pv_;                                                                                 \
    pf[rb_][c_][e_] = (_Float16)pv_;                                                                \
  }
#pragma unroll
    for (int e = 0; e < 8; ++e) { PEXP4(0, 0, e) PEXP4(1, 0, e) }
    STAMP(2)
    SB();
    // ---- O^T += V^T.P : 16 V^T fragments (chunk c, d block dt), two MFMAs each; chunk 1's exps in chunk 0's gaps
#pragma unroll
    for (int f = 0; f < 2 * DT; ++f) {
      const int c = f >> 3, dt = f & 7;
      o[0][dt] = mfma32_f16(fr[f & 3], pf[0][c], o[0][dt]);
      o[1][dt] = mfma32_f16(fr[f & 3], pf[1][c], o[1][dt]);
      if (f + 4 < 2 * DT) fr[f & 3] = VFRAG((f + 4) >> 3, (f + 4) & 7);
      if (c == 0) { PEXP4(0, 1, dt) PEXP4(1, 1, dt) }
      SB();
    }
    l_run[0] += rs[0];
    l_run[1] += rs[1];
    STAMP(3)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    STAMP(4)
    cur ^= 1;
  }
  if constexpr (DBG & 1) {
    if (lane == 0 && p.stamps) {
      unsigned long long* o_ = p.stamps + ((size_t)blockIdx.x * 4 + wave) * 5;
      for (int k = 0; k < 5; ++k) o_[k] = acc_t[k];
    }
  }
#undef STAMP
#undef SB
#undef KFRAG
#undef VFRAG
#undef PEXP4

#pragma unroll
  for (int rb = 0; rb < RB; ++rb) {
    const float l_tot = l_run[rb] + __shfl_xor(l_run[rb], 32);
    const float inv = 1.f / l_tot;
    const int qglob = qb * 256 + wave * 64 + rb * 32 + r;
    if (qglob < p.tq) {
      _Float16* op = p.O + b * p.sO + (size_t)qglob * p.ldo + (size_t)head * D;
#pragma unroll
      for (int dt = 0; dt < DT; ++dt)
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
          half4 v;
#pragma unroll
          for (int j = 0; j < 4; ++j) v[j] = (_Float16)(o[rb][dt][4 * g4 + j] * inv);
          *(half4*)(op + 32 * dt + 8 * g4 + 4 * h) = v;
        }
    }
  }
}


#ifdef DVD_LAB
#include "../../benchmarks/lab/csrc/attention_lab.inc"
#endif

}  // namespace dvd

using namespace dvd;

#ifdef DVD_LAB
static unsigned long long* g_attn_stamps = nullptr;
// lab builds only (DVD_ATTN_DEBUG=1): per-wave accumulated s_memtime deltas of the 5 phases of a KV tile
extern "C" int dvd_attn_debug_stamps(void* dev_u64) { g_attn_stamps = (unsigned long long*)dev_u64; return DVD_OK; }
#endif

// 64 query rows per wave (256 per workgroup) from this many query rows on: a (batch, head) then has >= 22 workgroups of
// its own.  The choice depends on the problem's SHAPE only (head_dim, tq, tk) - never on the batch - so a document
// takes the same kernel, and the same online-softmax tile order, alone or in a batch (bit-identical results).
static constexpr int R64_MIN_TQ = 5376;

extern "C" const char* dvd_flash_attn_kernel_name(int head_dim, int tq, int tk) {
  if (head_dim != 64 && head_dim != 256) return "";
  if (tk % 64 != 0) return head_dim == 256 ? "flash_attn_kernel<256>" : "flash_attn_kernel<64>";
  if (head_dim == 256) return tq >= R64_MIN_TQ ? "flash_attn_r64_kernel<0>" : "flash_attn_glds_kernel<256, 0>";
  return "flash_attn_glds_kernel<64, 0>";
}

template <typename KernelT>
static void allow_lds(KernelT kernel, int bytes) {
  (void)hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
}

extern "C" int dvd_flash_attn(const dvd_attn_desc* d, void* stream) {
  DVD_REQUIRE(d && d->Q && d->K && d->Vt && d->O, "flash_attn: null pointer");
  DVD_REQUIRE(d->head_dim == 64 || d->head_dim == 256, "flash_attn: head_dim %d not in {64,256}", d->head_dim);
  DVD_REQUIRE(d->heads > 0 && d->batch > 0 && d->tq > 0 && d->tk > 0 && d->kv_batch_div > 0, "flash_attn: bad shape");
  DVD_REQUIRE(d->tk % 8 == 0, "flash_attn: tk=%d must be a multiple of 8", d->tk);
  DVD_REQUIRE(d->ldq % 8 == 0 && d->ldk % 8 == 0 && d->ldvt % 8 == 0 && d->ldo % 4 == 0 &&
                  d->strideQ % 8 == 0 && d->strideK % 8 == 0 && d->strideVt % 8 == 0 && d->strideO % 4 == 0 &&
                  ((uintptr_t)d->Q % 16) == 0 && ((uintptr_t)d->K % 16) == 0 && ((uintptr_t)d->Vt % 16) == 0 &&
                  ((uintptr_t)d->O % 8) == 0,
              "flash_attn: operands must be 16-byte aligned");
  AttnArgs p;
  p.Q = (const _Float16*)d->Q; p.K = (const _Float16*)d->K; p.Vt = (const _Float16*)d->Vt; p.O = (_Float16*)d->O;
  p.sQ = d->strideQ; p.sK = d->strideK; p.sVt = d->strideVt; p.sO = d->strideO;
  p.ldq = d->ldq; p.ldk = d->ldk; p.ldvt = d->ldvt; p.ldo = d->ldo;
  p.heads = d->heads; p.batch = d->batch; p.tq = d->tq; p.tk = d->tk;
  p.kv_div = d->kv_batch_div;
  p.nqb = cdiv(d->tq, 128);
  p.c = d->scale * 1.4426950408889634f;
  p.stamps = nullptr;
  const long nwg = (long)p.nqb * d->heads * d->batch;
  DVD_REQUIRE(nwg < (1l << 31), "flash_attn: grid too large");
  hipStream_t st = (hipStream_t)stream;
  bool fast = d->tk % 64 == 0;      // the LDS-DMA kernels stream whole 64-key tiles; ragged key counts take the
                                    // register-staged kernel (masked tail)
  bool r64 = d->head_dim == 256 && d->tq >= R64_MIN_TQ;
  // hipFuncSetAttribute is per DEVICE: remember which devices have been set up (one bit each), lock-free - the
  // attribute call is idempotent, so two threads racing on a device's first launch both set it and both are right
  static DeviceOnce attr_done;
  const unsigned long long dev_bit = DeviceOnce::current_bit();
  const bool first_on_device = attr_done.need(dev_bit);
  if (first_on_device) {
    allow_lds(flash_attn_glds_kernel<256, 0>, 2 * (64 * 512 + 256 * 128));
    allow_lds(flash_attn_r64_kernel<0>, 2 * (32 * 512 + 256 * 64));
    allow_lds(flash_attn_kernel<256>, 2 * (64 * (2 * 256 + 16) + 256 * (2 * 64 + 16)));
    attr_done.done(dev_bit);
  }
#ifdef DVD_LAB
  // ---- lab build: every experiment and diagnostic variant behind its environment switch ----
  p.stamps = g_attn_stamps;
  const bool dbg = getenv("DVD_ATTN_DEBUG"), bulk = getenv("DVD_ATTN_BULK");
  if (getenv("DVD_ATTN_V1")) fast = false;
  if (getenv("DVD_ATTN_R64")) r64 = d->head_dim == 256;
  if (getenv("DVD_ATTN_R32") || getenv("DVD_ATTN_PIPE") || bulk) r64 = false;
  if (first_on_device) {
    constexpr int LDS = 2 * (64 * 512 + 256 * 128);
    allow_lds(flash_attn_dsplit_kernel, LDS);
    allow_lds(flash_attn_glds_kernel<256, 1>, LDS);
    allow_lds(flash_attn_glds_kernel<256, 2>, LDS);
    allow_lds(flash_attn_glds_kernel<256, 3>, LDS);
    allow_lds(flash_attn_pipe_kernel<0>, LDS);
    allow_lds(flash_attn_pipe_kernel<1>, LDS);
    allow_lds(flash_attn_r64_kernel<1>, 2 * (32 * 512 + 256 * 64));
  }
  if (fast && d->head_dim == 256) {
    constexpr int LDS = 2 * (64 * 512 + 256 * 128);
    if (getenv("DVD_ATTN_DSPLIT")) {   // measured slower (732 vs 812 TF/s)
      flash_attn_dsplit_kernel<<<(unsigned)nwg, 512, LDS, st>>>(p);
      return check_launch("flash_attn(lab dsplit)");
    }
    if (r64 && dbg) {
      p.nqb = cdiv(d->tq, 256);
      flash_attn_r64_kernel<1><<<(unsigned)((long)p.nqb * d->heads * d->batch), 256, 2 * (32 * 512 + 256 * 64), st>>>(p);
      return check_launch("flash_attn(lab r64 stamps)");
    }
    if (!r64 && getenv("DVD_ATTN_PIPE")) {   // slower (830 vs 975 TF/s): see the kernel's header
      if (dbg) flash_attn_pipe_kernel<1><<<(unsigned)nwg, 256, LDS, st>>>(p);
      else flash_attn_pipe_kernel<0><<<(unsigned)nwg, 256, LDS, st>>>(p);
      return check_launch("flash_attn(lab pipe)");
    }
    if (!r64 && (dbg || bulk)) {
      if (dbg && bulk) flash_attn_glds_kernel<256, 3><<<(unsigned)nwg, 256, LDS, st>>>(p);
      else if (bulk) flash_attn_glds_kernel<256, 2><<<(unsigned)nwg, 256, LDS, st>>>(p);
      else flash_attn_glds_kernel<256, 1><<<(unsigned)nwg, 256, LDS, st>>>(p);
      return check_launch("flash_attn(lab glds variant)");
    }
  } else if (fast) {
    constexpr int LDS = 2 * (64 * 128 + 64 * 128);
    if (getenv("DVD_ATTN_64X2")) {   // head_dim 64, two query row blocks per wave: measured = (872 vs 895 TF/s)
      p.nqb = cdiv(d->tq, 256);
      flash_attn_glds64x2_kernel<<<(unsigned)((long)p.nqb * d->heads * d->batch), 256, LDS, st>>>(p);
      return check_launch("flash_attn(lab 64x2)");
    }
    if (dbg || bulk) {
      if (dbg) flash_attn_glds_kernel<64, 1><<<(unsigned)nwg, 256, LDS, st>>>(p);
      else flash_attn_glds_kernel<64, 2><<<(unsigned)nwg, 256, LDS, st>>>(p);
      return check_launch("flash_attn(lab glds64 variant)");
    }
  }
#endif
  // ---- product dispatch: four kernels, chosen by (head_dim, tq, tk) ----
  if (fast && r64) {
    p.nqb = cdiv(d->tq, 256);
    flash_attn_r64_kernel<0><<<(unsigned)((long)p.nqb * d->heads * d->batch), 256, 2 * (32 * 512 + 256 * 64), st>>>(p);
  } else if (fast && d->head_dim == 256) {
    flash_attn_glds_kernel<256, 0><<<(unsigned)nwg, 256, 2 * (64 * 512 + 256 * 128), st>>>(p);
  } else if (fast) {
    flash_attn_glds_kernel<64, 0><<<(unsigned)nwg, 256, 2 * (64 * 128 + 64 * 128), st>>>(p);
  } else if (d->head_dim == 256) {
    flash_attn_kernel<256><<<(unsigned)nwg, 256, 2 * (64 * (2 * 256 + 16) + 256 * (2 * 64 + 16)), st>>>(p);
  } else {
    flash_attn_kernel<64><<<(unsigned)nwg, 256, 2 * (64 * (2 * 64 + 16) + 64 * (2 * 64 + 16)), st>>>(p);
  }
  return check_launch("flash_attn");
}
