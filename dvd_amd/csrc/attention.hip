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
  int xcd_map;           // lab builds only (DVD_ATTN_XCDMAP, r64x): 0 product map, 1 no remap, 2 two problems interleaved
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
  // (Round 4 tried the loop unrolled by the LDS buffer - every LDS address base + immediate, the next tile's pointers advanced
  // by an add, 34 -> ~15 SALU and no v_add_u32 per tile, same bits - after the decoder kernel's ablations had shown what such
  // glue costs ITS single wave per SIMD: head_dim 64, two waves per SIMD, measured 11.36 vs 11.23 ms, i.e. slightly slower, like
  // the v_dot2 / v_pk_add row sums.  This kernel runs at 1.45 GHz under the power cap; removing issue work does not speed it up.)
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
    if constexpr (!NEGM) mx *= p.c;            // NEGM: already (row max) * c - m_run
    // The deferred-rescale test is LANE-LOCAL (round 4): a lane holds 32 of its query's 64 keys and m_run is kept identical
    // in both halves of a row, so "some lane's own maximum exceeds m_run + THR" is exactly "some row's maximum does"; the
    // cross-half exchange (v_mov + v_permlane32_swap + 3 v_max + 3 s_nop per tile in a VALU-issue-bound loop) moves into the
    // rare branch.
    bool resc;
    if constexpr (NEGM) resc = t == 0 || __any(mx > RESCALE_THR);    // the first tile always sets the reference maximum
    else resc = __any(mx - m_run > RESCALE_THR);
    if (resc) {                                // deferred rescale, see flash_attn_kernel
      mx = half_swap_max(mx);                  // the row's maximum over all 64 keys, the same in both halves
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
    // chunk c of P = registers 8 (c & 1) .. +7 of s[c >> 1].  Row sums stay one v_add_f32 per element: 16 v_dot2c_f32_f16 on the
    // packed words instead of the 32 adds (17 fewer VALU instructions per tile) measured 1.4 % SLOWER at head_dim 64 (10.447 vs
    // 10.304 ms, round 4), 16 v_pk_add_f32 on pairs of exponentials 2.3 % slower (10.342 vs 10.108 ms): at 1.45 GHz the kernel is held
    // by the power cap, not by its instruction count (benchmarks/lab/valu_lab.hip: two waves per SIMD issue VALU work concurrently).
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
// head_dim 256, 64 query rows per wave, SOFTWARE-PIPELINED across key tiles ("r64p", round 4).
//
// What rocm's compiler made of flash_attn_r64_kernel (read in the ISA, round 4): its PV phase uses the MFMA builtin, and a
// builtin MFMA is a pure value to the instruction selector - sched_barrier(0) pins the machine scheduler, not the order
// the selection DAG is linearised in - so ALL 32 exponentials (+ 32 v_fma, 10 v_pk_add_f32) of a tile were emitted ahead
// of the first PV MFMA: ~600 cycles per 32-key tile with the matrix pipe idle, on top of the LDS latency exposed after
// every barrier.  Here the tile loop is a sequence of `asm volatile` statements (they keep their source order; the
// compiler still allocates the registers), one statement per K / V^T fragment = two MFMAs + their VALU fillers + the
// next fragment read, and the dependency S^T -> softmax -> PV is cut by pipelining across tiles:
//
//   iteration t :  phase 1  S^T(t+1) = K(t+1).Q^T     32 MFMAs | VALU in the gaps: exp2 / row sum / f16 pack of tile t
//                                                               (chunk 0 of both row blocks, chunk 1 of row block 0);
//                                                               this wave's 4 LDS-DMA pieces of K(t+2)
//                  phase 2a O^T += V^T(t).P(t) chunk 0 16 MFMAs | chunk 1 of row block 1
//                  phase 2b                    chunk 1 16 MFMAs | lane-local row maximum of S^T(t+1) and the
//                                                               deferred-rescale test; the 4 LDS-DMA pieces of V^T(t+2)
//                  [rare]   rescale O^T, l, m          (after PV(t) has been issued: O^T, l and P(t) share one reference)
//                  vmcnt(4) + ONE barrier              (the V^T pieces just issued stay in flight: a 3-deep V^T ring)
//
// so the matrix pipe always has independent MFMAs while a tile's softmax runs on the VALU.  S^T is double-buffered in
// architectural VGPRs (2 x 32), the loop is unrolled by two so both buffers have compile-time names; Q (128 VGPRs) and
// O^T (all 256 AGPRs) as in flash_attn_r64_kernel.  The row maximum is checked LANE-LOCALLY (a lane owns 16 of its query's
// 32 keys; m_run is kept identical in both halves of a row), so the cross-half exchange exists only in the rare branch.
// Measured (MI355X, s_memtime stamps + PMC, profiles/archive/r4_*): 3574 -> ~2900 cycles per tile, MFMA busy 58.7 -> 68.6 %, but the
// chip holds 1.70 instead of 1.87 GHz under the denser stream (it runs at its power cap): +7 % wall for -14 % cycles.
//
// Why statements are merged: hipcc's hazard recogniser counts an inline-asm statement as ZERO wait states and assumes a
// dst-forwarding hazard whenever a statement reads a VGPR that an earlier statement wrote with no compiler instruction in
// between - it then pads with s_nop 0 (4 issue cycles each; the first version had 43 per tile).  One statement per
// fragment keeps the def -> use chains (row sums, exponentials -> packs) inside a statement or a whole statement apart.
//
// LDS: K tiles are stored as 16 two-row pieces (one 1-KiB LDS-DMA each) at a pitch of 1056 B with the 16-byte chunks of
// the odd row XOR-ed by 1: conflict-free ds_read_b128 like the old 16-way XOR swizzle, but a fragment address is ONE
// per-lane base + an immediate (the old image cost a v_xor per read).  V^T tiles as in flash_attn_r64_kernel (two bases).
// [K0 | K1 | V0 | V1 | V2] = 2 x 16896 + 3 x 16384 B; the K ring is indexed by immediates, the V^T ring by two rotating
// address registers (its offsets would not fit ds_read's 16-bit immediate).  The LDS-DMA source offsets of a wave's four
// pieces differ by a wave-uniform stride, so ONE per-lane offset register serves them (the base pointer moves in SGPRs).
// Fragment maps, kappa key order, deferred-rescale rule and per-element arithmetic are flash_attn_r64_kernel's.
// ================================================================================================
namespace r64p {
constexpr int KPIECE = 1056, KBYTES = 16 * KPIECE, VBYTES = 256 * 64, KSLOTS = 3, VSLOTS = 3, VBASE = KSLOTS * KBYTES;
constexpr int LDS_BYTES = KSLOTS * KBYTES + VSLOTS * VBYTES;
constexpr float RESCALE_THR = 10.f;

// One 1-KiB LDS-DMA piece (prologue form).  M0 (the LDS destination) is written in the statement that uses it and NOT
// restored: nothing else in this kernel reads M0 (checked in the ISA).
__device__ __forceinline__ void glds_piece(const char* gbase, unsigned voff, unsigned lds) {
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %0" ::"s"(gbase), "s"(lds), "v"(voff) : "memory");
}

#define R64P_MFMA0(acc_, a_, b_) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, 0" : "=&v"(acc_) : "v"(a_), "v"(b_))
#define R64P_MFMA(acc_, a_, b_) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc_) : "v"(a_), "v"(b_))
#define R64P_LDS(dst_, addr_, off_) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst_) : "v"(addr_), "i"(off_))
#define R64P_WAIT_LGKM(n_) asm volatile("s_waitcnt lgkmcnt(%0)" ::"i"(n_))

// ---- text pieces of a step statement.  A step = one K / V^T fragment = two MFMAs; its shape is
//        s_waitcnt lgkmcnt(2) | MFMA a | ds_read (fragment three steps ahead) [LDS-DMA piece] fillers A | MFMA b | fillers B
//      The read and the DMA piece sit in the FIRST gap: the second one also carries the next step's wait and the s_nop
//      hipcc pads every statement with, and a gap hides ~24 issue cycles (fma 4, exp 8, add 4, ds_read 4, ...) beside
//      its MFMA's own 8; with the read behind the second MFMA that gap ran 40+ cycles (stamps: 43 cycles per MFMA).
#define R64P_W "s_waitcnt lgkmcnt(2)\n\t"
#define R64P_MF "v_mfma_f32_32x32x16_f16 "
#define R64P_RD "ds_read_b128 %[nf], %[addr] offset:%[off]\n\t"
#define R64P_DMA "s_mov_b32 m0, %[lds]\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %[voff], %[gb]\n\t"
// exp unit: e = exp2(s * c - m), then the row-sum add of an OLDER unit's result (never of the exponential just issued:
// trans-use hazard, one wait state)
#define R64P_EU(e_, s_, m_, acc_, old_) \
  "v_fma_f32 %[" e_ "], %[" s_ "], %[c], -%[" m_ "]\n\tv_exp_f32_e32 %[" e_ "], %[" e_ "]\n\tv_add_f32_e32 %[" acc_ "], %[" acc_ "], %[" old_ "]\n\t"

// ---- phase 1, first fragment: the two S^T chains start from C = 0; no row-sum add is pending yet
template <int OFF>
__device__ __forceinline__ void p1_first(floatx16& sn0, floatx16& sn1, const half8& fr, half8& nf, const half8& q0,
                                         const half8& q1, float sa, float ma, float& ea, float sb, float mb, float& eb, float c,
                                         float& acc_a, unsigned addr) {
  asm volatile(R64P_W R64P_MF "%[sn0], %[fr], %[q0], 0\n\t" R64P_RD
               "v_fma_f32 %[ea], %[sa], %[c], -%[ma]\n\tv_exp_f32_e32 %[ea], %[ea]\n\t"
               R64P_MF "%[sn1], %[fr], %[q1], 0\n\t" R64P_EU("eb", "sb", "mb", "acca", "ea")
               : [sn0] "=&v"(sn0), [sn1] "=&v"(sn1), [nf] "=&v"(nf), [ea] "=&v"(ea), [eb] "=&v"(eb), [acca] "+v"(acc_a)
               : [fr] "v"(fr), [q0] "v"(q0), [q1] "v"(q1), [sa] "v"(sa), [sb] "v"(sb), [ma] "v"(ma), [mb] "v"(mb), [c] "s"(c),
                 [addr] "v"(addr), [off] "i"(OFF));
}
// ---- phase 1, a fragment with two exp units in its two MFMA gaps
template <int OFF>
__device__ __forceinline__ void p1_eu2(floatx16& sn0, floatx16& sn1, const half8& fr, half8& nf, const half8& q0, const half8& q1,
                                       float sa, float ma, float& ea, float sb, float mb, float& eb, float c, float old,
                                       float& acc_old, float& acc_a, unsigned addr) {
  asm volatile(R64P_W R64P_MF "%[sn0], %[fr], %[q0], %[sn0]\n\t" R64P_RD R64P_EU("ea", "sa", "ma", "acco", "old")
               R64P_MF "%[sn1], %[fr], %[q1], %[sn1]\n\t" R64P_EU("eb", "sb", "mb", "acca", "ea")
               : [sn0] "+v"(sn0), [sn1] "+v"(sn1), [nf] "=&v"(nf), [ea] "=&v"(ea), [eb] "=&v"(eb), [acco] "+v"(acc_old),
                 [acca] "+v"(acc_a)
               : [fr] "v"(fr), [q0] "v"(q0), [q1] "v"(q1), [sa] "v"(sa), [sb] "v"(sb), [ma] "v"(ma), [mb] "v"(mb), [c] "s"(c),
                 [old] "v"(old), [addr] "v"(addr), [off] "i"(OFF));
}
// ---- phase 1, a fragment without exp units: one LDS-DMA piece in its first gap, the four f16 packs of one P fragment
//      (8 exponentials of one row block -> 4 words) in its second
template <int OFF>
__device__ __forceinline__ u32x4 p1_dma_pack(floatx16& sn0, floatx16& sn1, const half8& fr, half8& nf, const half8& q0,
                                             const half8& q1, unsigned addr, const char* gb, unsigned lds, unsigned voff, float e0,
                                             float e1, float e2, float e3, float e4, float e5, float e6, float e7) {
  // each packed word overwrites the register of its first exponential (no extra registers: the kernel sits at 256)
  asm volatile(R64P_W R64P_MF "%[sn0], %[fr], %[q0], %[sn0]\n\t" R64P_RD R64P_DMA R64P_MF "%[sn1], %[fr], %[q1], %[sn1]\n\t"
               "v_cvt_pk_f16_f32 %[e0], %[e0], %[e1]\n\tv_cvt_pk_f16_f32 %[e2], %[e2], %[e3]\n\t"
               "v_cvt_pk_f16_f32 %[e4], %[e4], %[e5]\n\tv_cvt_pk_f16_f32 %[e6], %[e6], %[e7]"
               : [sn0] "+v"(sn0), [sn1] "+v"(sn1), [nf] "=&v"(nf), [e0] "+v"(e0), [e2] "+v"(e2), [e4] "+v"(e4), [e6] "+v"(e6)
               : [fr] "v"(fr), [q0] "v"(q0), [q1] "v"(q1), [addr] "v"(addr), [off] "i"(OFF), [gb] "s"(gb), [lds] "s"(lds),
                 [voff] "v"(voff), [e1] "v"(e1), [e3] "v"(e3), [e5] "v"(e5), [e7] "v"(e7)
               : "memory");
  return u32x4{__builtin_bit_cast(unsigned, e0), __builtin_bit_cast(unsigned, e2), __builtin_bit_cast(unsigned, e4),
               __builtin_bit_cast(unsigned, e6)};
}
template <int OFF>
__device__ __forceinline__ void p1_dma(floatx16& sn0, floatx16& sn1, const half8& fr, half8& nf, const half8& q0, const half8& q1,
                                       unsigned addr, const char* gb, unsigned lds, unsigned voff) {
  asm volatile(R64P_W R64P_MF "%[sn0], %[fr], %[q0], %[sn0]\n\t" R64P_RD R64P_DMA R64P_MF "%[sn1], %[fr], %[q1], %[sn1]"
               : [sn0] "+v"(sn0), [sn1] "+v"(sn1), [nf] "=&v"(nf)
               : [fr] "v"(fr), [q0] "v"(q0), [q1] "v"(q1), [addr] "v"(addr), [off] "i"(OFF), [gb] "s"(gb), [lds] "s"(lds),
                 [voff] "v"(voff)
               : "memory");
}
// four f16 packs (one P fragment = 8 elements of one row block) as one statement
__device__ __forceinline__ u32x4 pack8(float e0, float e1, float e2, float e3, float e4, float e5, float e6, float e7) {
  unsigned w0, w1, w2, w3;
  asm volatile("v_cvt_pk_f16_f32 %0, %4, %5\n\tv_cvt_pk_f16_f32 %1, %6, %7\n\tv_cvt_pk_f16_f32 %2, %8, %9\n\t"
               "v_cvt_pk_f16_f32 %3, %10, %11"
               : "=&v"(w0), "=&v"(w1), "=&v"(w2), "=&v"(w3)
               : "v"(e0), "v"(e1), "v"(e2), "v"(e3), "v"(e4), "v"(e5), "v"(e6), "v"(e7));
  return u32x4{w0, w1, w2, w3};
}
// ---- phase 2a: two PV MFMAs (O^T in AGPRs) + one exp unit
template <int OFF>
__device__ __forceinline__ void p2_eu1(floatx16& o0, floatx16& o1, const half8& fr, half8& nf, const u32x4& p0, const u32x4& p1,
                                       float sa, float ma, float& ea, float c, float old, float& acc_old, unsigned addr) {
  asm volatile(R64P_W R64P_MF "%[o0], %[fr], %[p0], %[o0]\n\t" R64P_RD R64P_EU("ea", "sa", "ma", "acco", "old")
               R64P_MF "%[o1], %[fr], %[p1], %[o1]"
               : [o0] "+a"(o0), [o1] "+a"(o1), [nf] "=&v"(nf), [ea] "=&v"(ea), [acco] "+v"(acc_old)
               : [fr] "v"(fr), [p0] "v"(p0), [p1] "v"(p1), [sa] "v"(sa), [ma] "v"(ma), [c] "s"(c), [old] "v"(old),
                 [addr] "v"(addr), [off] "i"(OFF));
}
// ---- phase 2b.  The lane-local maximum of S^T(next) runs in four chains (two per row block).
// first step: the chains start (elements 0-2 and 7-9 of each row block)
template <int OFF>
__device__ __forceinline__ void p2_max_first(floatx16& o0, floatx16& o1, const half8& fr, half8& nf, const u32x4& p0,
                                             const u32x4& p1, const floatx16& s0, const floatx16& s1, float& a0, float& b0,
                                             float& a1, float& b1, unsigned addr) {
  asm volatile(R64P_W R64P_MF "%[o0], %[fr], %[p0], %[o0]\n\t" R64P_RD
               "v_max3_f32 %[a0], %[x0], %[x1], %[x2]\n\tv_max3_f32 %[b0], %[x3], %[x4], %[x5]\n\t"
               R64P_MF "%[o1], %[fr], %[p1], %[o1]\n\t"
               "v_max3_f32 %[a1], %[y0], %[y1], %[y2]\n\tv_max3_f32 %[b1], %[y3], %[y4], %[y5]"
               : [o0] "+a"(o0), [o1] "+a"(o1), [nf] "=&v"(nf), [a0] "=&v"(a0), [b0] "=&v"(b0), [a1] "=&v"(a1), [b1] "=&v"(b1)
               : [fr] "v"(fr), [p0] "v"(p0), [p1] "v"(p1), [x0] "v"(s0[0]), [x1] "v"(s0[1]), [x2] "v"(s0[2]), [x3] "v"(s0[7]),
                 [x4] "v"(s0[8]), [x5] "v"(s0[9]), [y0] "v"(s1[0]), [y1] "v"(s1[1]), [y2] "v"(s1[2]), [y3] "v"(s1[7]),
                 [y4] "v"(s1[8]), [y5] "v"(s1[9]), [addr] "v"(addr), [off] "i"(OFF));
}
// a step that extends the four chains by two elements each: a <- max3(a, xa0, xa1), b <- max3(b, xb0, xb1)
template <int OFF>
__device__ __forceinline__ void p2_max_step(floatx16& o0, floatx16& o1, const half8& fr, half8& nf, const u32x4& p0,
                                            const u32x4& p1, float& a0, float& b0, float& a1, float& b1, float xa0, float xa1,
                                            float xb0, float xb1, float ya0, float ya1, float yb0, float yb1, unsigned addr) {
  asm volatile(R64P_W R64P_MF "%[o0], %[fr], %[p0], %[o0]\n\t" R64P_RD
               "v_max3_f32 %[a0], %[a0], %[xa0], %[xa1]\n\tv_max3_f32 %[b0], %[b0], %[xb0], %[xb1]\n\t"
               R64P_MF "%[o1], %[fr], %[p1], %[o1]\n\t"
               "v_max3_f32 %[a1], %[a1], %[ya0], %[ya1]\n\tv_max3_f32 %[b1], %[b1], %[yb0], %[yb1]"
               : [o0] "+a"(o0), [o1] "+a"(o1), [nf] "=&v"(nf), [a0] "+v"(a0), [b0] "+v"(b0), [a1] "+v"(a1), [b1] "+v"(b1)
               : [fr] "v"(fr), [p0] "v"(p0), [p1] "v"(p1), [xa0] "v"(xa0), [xa1] "v"(xa1), [xb0] "v"(xb0), [xb1] "v"(xb1),
                 [ya0] "v"(ya0), [ya1] "v"(ya1), [yb0] "v"(yb0), [yb1] "v"(yb1), [addr] "v"(addr), [off] "i"(OFF));
}
// The remaining steps of phase 2b carry one LDS-DMA piece of V^T(t+2) in their first gap and their VALU in the second.
#define R64P_P2_DMA_STEP(name_, valu_, outs_, ins_, ...)                                                                   \
  template <int OFF>                                                                                                       \
  __device__ __forceinline__ void name_(floatx16& o0, floatx16& o1, const half8& fr, half8& nf, const u32x4& p0,           \
                                        const u32x4& p1, unsigned addr, const char* gb, unsigned lds, unsigned voff,        \
                                        __VA_ARGS__) {                                                                     \
    asm volatile(R64P_W R64P_MF "%[o0], %[fr], %[p0], %[o0]\n\t" R64P_RD R64P_DMA R64P_MF "%[o1], %[fr], %[p1], %[o1]\n\t" \
                 valu_                                                                                                     \
                 : [o0] "+a"(o0), [o1] "+a"(o1), [nf] "=&v"(nf) outs_                                                      \
                 : [fr] "v"(fr), [p0] "v"(p0), [p1] "v"(p1), [addr] "v"(addr), [off] "i"(OFF), [gb] "s"(gb), [lds] "s"(lds), \
                   [voff] "v"(voff) ins_                                                                                   \
                 : "memory");                                                                                              \
  }
#define R64P_COMMA ,
// join: a <- max3(a, b, x14) per row block
R64P_P2_DMA_STEP(p2_join, "v_max3_f32 %[a0], %[a0], %[b0], %[x14]\n\tv_max3_f32 %[a1], %[a1], %[b1], %[y14]",
                 R64P_COMMA[a0] "+v"(a0) R64P_COMMA[a1] "+v"(a1),
                 R64P_COMMA[b0] "v"(b0) R64P_COMMA[b1] "v"(b1) R64P_COMMA[x14] "v"(x14) R64P_COMMA[y14] "v"(y14), float& a0, float b0,
                 float& a1, float b1, float x14, float y14)
// last element, then d = max(a0 * c - thr0, a1 * c - thr1): > 0 in some lane <=> a row's maximum grew by more than THR
R64P_P2_DMA_STEP(p2_last, "v_max_f32_e32 %[a0], %[a0], %[x15]\n\tv_max_f32_e32 %[a1], %[a1], %[y15]",
                 R64P_COMMA[a0] "+v"(a0) R64P_COMMA[a1] "+v"(a1), R64P_COMMA[x15] "v"(x15) R64P_COMMA[y15] "v"(y15), float& a0,
                 float& a1, float x15, float y15)
R64P_P2_DMA_STEP(p2_test,
                 "v_fma_f32 %[d], %[a0], %[c], -%[thr0]\n\tv_fma_f32 %[t], %[a1], %[c], -%[thr1]\n\tv_max_f32_e32 %[d], %[d], %[t]",
                 R64P_COMMA[d] "=&v"(d) R64P_COMMA[t] "=&v"(t),
                 R64P_COMMA[a0] "v"(a0) R64P_COMMA[a1] "v"(a1) R64P_COMMA[c] "s"(c) R64P_COMMA[thr0] "v"(thr0) R64P_COMMA[thr1] "v"(thr1),
                 float a0, float a1, float c, float thr0, float thr1, float& d, float& t)
// lane mask of d > 0, and the first half of the row-sum update
R64P_P2_DMA_STEP(p2_mask,
                 "v_cmp_lt_f32_e64 %[mask], 0, %[d]\n\tv_add_f32_e32 %[l0], %[l0], %[ra0]\n\tv_add_f32_e32 %[l1], %[l1], %[ra1]",
                 R64P_COMMA[mask] "=&s"(mask) R64P_COMMA[l0] "+v"(l0) R64P_COMMA[l1] "+v"(l1),
                 R64P_COMMA[d] "v"(d) R64P_COMMA[ra0] "v"(ra0) R64P_COMMA[ra1] "v"(ra1), float d, unsigned long long& mask, float& l0,
                 float ra0, float& l1, float ra1)
// last step of an iteration: the second half of the row-sum update (no DMA piece)
template <int OFF>
__device__ __forceinline__ void p2_lsum(floatx16& o0, floatx16& o1, const half8& fr, half8& nf, const u32x4& p0, const u32x4& p1,
                                        float& l0, float rb0, float& l1, float rb1, unsigned addr) {
  asm volatile(R64P_W R64P_MF "%[o0], %[fr], %[p0], %[o0]\n\t" R64P_RD R64P_MF "%[o1], %[fr], %[p1], %[o1]\n\t"
               "v_add_f32_e32 %[l0], %[l0], %[rb0]\n\tv_add_f32_e32 %[l1], %[l1], %[rb1]"
               : [o0] "+a"(o0), [o1] "+a"(o1), [nf] "=&v"(nf), [l0] "+v"(l0), [l1] "+v"(l1)
               : [fr] "v"(fr), [p0] "v"(p0), [p1] "v"(p1), [rb0] "v"(rb0), [rb1] "v"(rb1), [addr] "v"(addr), [off] "i"(OFF));
}

struct Soft {            // per row block: running max, its rescale threshold m + THR, running sum (lane-partial)
  float m, thr, l;
};

#define R64P_STAMP(k_)                                                                         \
  if constexpr (DBG_ != 0) {                                                                   \
    unsigned long long now_;                                                                   \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(now_)::"memory");               \
    acc_t[k_] += now_ - tprev;                                                                 \
    tprev = now_;                                                                              \
  }

// One pipelined iteration t.  SC = S^T(t) (complete, reference max valid): its P is formed and consumed here; SN = S^T(t+1),
// written here.  kcur / knext: K fragment base addresses of the ring slots of K(t+1) (phase 1) and K(t+2) (pre-read at the
// end of phase 2b); vrd0 / vrd1: V^T(t) fragment bases (chunk 0 / 1).  fr: the four-slot fragment ring, in flight ACROSS
// iterations: step n waits for fragment n, and reads fragment n + 3 into the slot step n - 1 has just consumed.
template <int DBG_>
__device__ __forceinline__ void tile_iter(unsigned long long (&acc_t)[6], unsigned long long& tprev, half8 (&fr)[4],
                                          floatx16 (&SC)[2], floatx16 (&SN)[2], const half8 (&qf)[2][16], floatx16 (&o)[2][8],
                                          Soft (&sm)[2], float c, unsigned kcur, unsigned knext, unsigned vrd0, unsigned vrd1,
                                          const char* kg_next, const char* vg_next, unsigned kstride4, unsigned vstride4,
                                          unsigned koff0, unsigned voff0, unsigned lds_k_dst, unsigned lds_v_dst) {
  float e[32];                    // exponentials by exp-unit number
  float ra[2] = {0.f, 0.f}, rb[2] = {0.f, 0.f};   // row sums: adds issued in an MFMA pair's first / second gap
  // exp unit u (0..31) -> (row block, S^T element): chunk 0 of rb 0 and rb 1 interleaved (u < 16), then chunk 1 of rb 0
  // (16..23), then chunk 1 of rb 1 (24..31)
#define URB(u_) ((u_) < 16 ? ((u_) & 1) : ((u_) < 24 ? 0 : 1))
#define UEL(u_) ((u_) < 16 ? ((u_) >> 1) : ((u_) < 24 ? 8 + (u_) - 16 : 8 + (u_) - 24))
#define USC(u_) SC[URB(u_)][UEL(u_)]
#define UM(u_) sm[URB(u_)].m
  // ---------------- phase 1: S^T(next) = K . Q^T; exp units 0..23 of the current tile in the first 24 gaps ----------------
  u32x4 p00, p10, p01, p11;
#define P1_EU2(f_)                                                                                                          \
  p1_eu2<((f_) + 3) * 32>(SN[0], SN[1], fr[(f_) & 3], fr[((f_) + 3) & 3], qf[0][f_], qf[1][f_], USC(2 * (f_)), UM(2 * (f_)),     \
                          e[2 * (f_)], USC(2 * (f_) + 1), UM(2 * (f_) + 1), e[2 * (f_) + 1], c, e[2 * (f_) - 1],                \
                          rb[URB(2 * (f_) - 1)], ra[URB(2 * (f_))], kcur);
  p1_first<3 * 32>(SN[0], SN[1], fr[0], fr[3], qf[0][0], qf[1][0], USC(0), UM(0), e[0], USC(1), UM(1), e[1], c, ra[0], kcur);
  P1_EU2(1)
  P1_EU2(2)
  P1_EU2(3)
  P1_EU2(4)
  P1_EU2(5)
  P1_EU2(6)
  P1_EU2(7)                        // units 14, 15: chunk 0 is complete
  P1_EU2(8)
  P1_EU2(9)
  P1_EU2(10)
  P1_EU2(11)                       // units 22, 23: chunk 1 of row block 0 is complete
#undef P1_EU2
  // the last four fragments carry this wave's 4 LDS-DMA pieces of K(t+3) (they have more than an iteration to land) and the
  // f16 packs of the three P fragments formed so far; the ring runs on into phase 2: V^T fragments (chunk 0, d blocks
  // 0..2) behind the last K fragment
  p00 = p1_dma_pack<15 * 32>(SN[0], SN[1], fr[0], fr[3], qf[0][12], qf[1][12], kcur, kg_next, lds_k_dst, koff0, e[0], e[2], e[4],
                             e[6], e[8], e[10], e[12], e[14]);
  p10 = p1_dma_pack<0 * 2048>(SN[0], SN[1], fr[1], fr[0], qf[0][13], qf[1][13], vrd0, kg_next + kstride4, lds_k_dst + KPIECE, koff0,
                              e[1], e[3], e[5], e[7], e[9], e[11], e[13], e[15]);
  p01 = p1_dma_pack<1 * 2048>(SN[0], SN[1], fr[2], fr[1], qf[0][14], qf[1][14], vrd0, kg_next + 2 * (size_t)kstride4,
                              lds_k_dst + 2 * KPIECE, koff0, e[16], e[17], e[18], e[19], e[20], e[21], e[22], e[23]);
  p1_dma<2 * 2048>(SN[0], SN[1], fr[3], fr[2], qf[0][15], qf[1][15], vrd0, kg_next + 3 * (size_t)kstride4,
                   lds_k_dst + 3 * KPIECE, koff0);
  R64P_STAMP(0)
  // ---------------- phase 2a: O^T += V^T . P chunk 0 (d blocks g = 0..7); exp units 24..31 ----------------
#define P2A(g_, addr_, off_)                                                                                               \
  p2_eu1<off_>(o[0][g_], o[1][g_], fr[(g_) & 3], fr[((g_) + 3) & 3], p00, p10, USC(24 + (g_)), sm[1].m, e[24 + (g_)], c,    \
               e[23 + (g_)], rb[URB(23 + (g_))], addr_);
  P2A(0, vrd0, 3 * 2048)
  P2A(1, vrd0, 4 * 2048)
  P2A(2, vrd0, 5 * 2048)
  P2A(3, vrd0, 6 * 2048)
  P2A(4, vrd0, 7 * 2048)
  P2A(5, vrd1, 0 * 2048)
  P2A(6, vrd1, 1 * 2048)
  P2A(7, vrd1, 2 * 2048)
#undef P2A
  R64P_STAMP(1)
  // ---------------- the iteration's ONE barrier.  Every LDS-DMA piece but the four K(t+3) pieces just issued has landed
  // (K(t+2) and V^T(t+1): issued an iteration ago) and becomes visible to every wave; every wave has finished phase 1 of
  // this iteration and phase 2b of the previous one, so the K slot the next phase 1 refills and the V^T slot phase 2b
  // refills below are free.
  if constexpr (DBG_ != 0) {
    asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    R64P_STAMP(2)
    asm volatile("s_barrier" ::: "memory");
    R64P_STAMP(3)
  } else {
    asm volatile("s_waitcnt vmcnt(4)\n\ts_barrier" ::: "memory");
  }
  asm volatile("v_add_f32_e32 %0, %0, %1" : "+v"(rb[1]) : "v"(e[31]));     // the last unit's row-sum add
  p11 = pack8(e[24], e[25], e[26], e[27], e[28], e[29], e[30], e[31]);
  // ---------------- phase 2b: chunk 1; lane-local maximum of S^T(next) in four chains, the rescale test, l += row sums;
  // this wave's 4 LDS-DMA pieces of V^T(t+2); the first three K fragments of the next iteration ----------------
  float a0, b0, a1, b1, dtest, ttmp;
  unsigned long long mask;
  p2_max_first<3 * 2048>(o[0][0], o[1][0], fr[0], fr[3], p01, p11, SN[0], SN[1], a0, b0, a1, b1, vrd1);
  p2_max_step<4 * 2048>(o[0][1], o[1][1], fr[1], fr[0], p01, p11, a0, b0, a1, b1, SN[0][3], SN[0][4], SN[0][10], SN[0][11],
                        SN[1][3], SN[1][4], SN[1][10], SN[1][11], vrd1);
  p2_max_step<5 * 2048>(o[0][2], o[1][2], fr[2], fr[1], p01, p11, a0, b0, a1, b1, SN[0][5], SN[0][6], SN[0][12], SN[0][13],
                        SN[1][5], SN[1][6], SN[1][12], SN[1][13], vrd1);
  p2_join<6 * 2048>(o[0][3], o[1][3], fr[3], fr[2], p01, p11, vrd1, vg_next, lds_v_dst, voff0, a0, b0, a1, b1, SN[0][14],
                    SN[1][14]);
  p2_last<7 * 2048>(o[0][4], o[1][4], fr[0], fr[3], p01, p11, vrd1, vg_next + vstride4, lds_v_dst + 1024, voff0, a0, a1,
                    SN[0][15], SN[1][15]);
  p2_test<0 * 32>(o[0][5], o[1][5], fr[1], fr[0], p01, p11, knext, vg_next + 2 * (size_t)vstride4, lds_v_dst + 2 * 1024, voff0,
                  a0, a1, c, sm[0].thr, sm[1].thr, dtest, ttmp);
  p2_mask<1 * 32>(o[0][6], o[1][6], fr[2], fr[1], p01, p11, knext, vg_next + 3 * (size_t)vstride4, lds_v_dst + 3 * 1024, voff0,
                  dtest, mask, sm[0].l, ra[0], sm[1].l, ra[1]);
  p2_lsum<2 * 32>(o[0][7], o[1][7], fr[3], fr[2], p01, p11, sm[0].l, rb[0], sm[1].l, rb[1], knext);
  R64P_STAMP(4)
  // deferred rescale (rare): some lane saw its row's maximum over its 16 keys of the next tile exceed m + THR
  if (mask != 0) {
    asm volatile("s_nop 15\n\ts_nop 7" ::: "memory");          // the last PV MFMAs must have written O^T
    const float mloc[2] = {a0, a1};
#pragma unroll
    for (int r_ = 0; r_ < 2; ++r_) {
      const float mx = half_swap_max(mloc[r_] * c);            // the row's maximum over all 32 keys, same in both halves
      const float m_new = fmaxf(sm[r_].m, mx);
      const float alpha = __builtin_amdgcn_exp2f(sm[r_].m - m_new);
      sm[r_].m = m_new;
      sm[r_].thr = m_new + RESCALE_THR;
      sm[r_].l *= alpha;
#pragma unroll
      for (int dt = 0; dt < 8; ++dt) scale_acc_in_agpr(o[r_][dt], alpha);
    }
  }
  R64P_STAMP(5)
#undef URB
#undef UEL
#undef USC
#undef UM
}
}  // namespace r64p

template <int DBG>
__global__ void __launch_bounds__(256, 1) flash_attn_r64p_kernel(AttnArgs p) {
  using namespace r64p;
  constexpr int D = 256, KB = 32, KS = 16, DT = 8, RB = 2;
  extern __shared__ __attribute__((aligned(16))) char smem[];   // [K0 | K1 | V0 | V1 | V2]
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

  // LDS-DMA source offsets.  K piece j = 4 wave + i holds key rows 2j, 2j + 1: LDS slot (e = lane >> 5, pos = lane & 31)
  // <- row 2j + e, chunk pos ^ e; the pieces of one wave are 2 rows = `kstride4` bytes apart, so the per-lane offset is
  // the one of piece 4 wave and the base pointer advances.  V^T piece j holds d rows 16j .. 16j + 15: lane -> row
  // 16j + (lane >> 2), chunk (lane & 3) ^ ((lane >> 4) & 3) (the row's bits 2,3 are the lane's bits 4,5 for every j).
  const unsigned kstride4 = (unsigned)(2 * p.ldk * 2), vstride4 = (unsigned)(16 * p.ldvt * 2);
  const unsigned koff0 = (unsigned)(8 * wave + (lane >> 5)) * (unsigned)(p.ldk * 2) + (unsigned)(((lane & 31) ^ (lane >> 5)) * 16);
  const unsigned voff0 = (unsigned)(64 * wave + (lane >> 2)) * (unsigned)(p.ldvt * 2) + (unsigned)(((lane & 3) ^ ((lane >> 4) & 3)) * 16);
  const unsigned lds0 = (unsigned)(uintptr_t)(lptr_t)smem;
  const int kr = kappa(r);
  const unsigned kaddr = lds0 + (kr >> 1) * KPIECE + (kr & 1) * 512 + ((h ^ (kr & 1)) * 16);        // slot 0; + ks * 32
  const unsigned vrel0 = lds0 + VBASE + r * 64 + ((h ^ ((r >> 2) & 3)) * 16);                       // chunk 0, slot 0; + dt * 2048
  const unsigned vrel1 = vrel0 ^ 32;                                                                // chunk 1

  floatx16 o[RB][DT];
#pragma unroll
  for (int rb = 0; rb < RB; ++rb)
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
      for (int i = 0; i < 16; ++i) o[rb][dt][i] = 0.f;
  Soft sm[RB] = {{-1e30f, -1e30f, 0.f}, {-1e30f, -1e30f, 0.f}};
  const int nt = p.tk / KB;              // even (tk % 64 == 0)
  const size_t ktile = (size_t)KB * p.ldk * 2;
  const unsigned kdst = lds0 + (4 * wave) * KPIECE, vdst = lds0 + VBASE + (4 * wave) * 1024;

  // ---- prologue: K(0), K(1), K(2) -> K slots 0, 1, 2; V(0), V(1) -> V slots 0, 1
  {
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const char* kj = Kg + (size_t)min(j, nt - 1) * ktile;
      const char* vj = Vg + (size_t)min(j, nt - 1) * (KB * 2);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        glds_piece(kj + (size_t)i * kstride4, koff0, kdst + j * KBYTES + i * KPIECE);
        if (j < 2) glds_piece(vj + (size_t)i * vstride4, voff0, vdst + j * VBYTES + i * 1024);
      }
    }
  }
  asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");

  floatx16 sA[RB], sB[RB];
  {   // S^T(0) -> sA, un-pipelined; its row maximum sets the reference
    half8 f0[4];
#pragma unroll
    for (int f = 0; f < 4; ++f) R64P_LDS(f0[f], kaddr, f * 32);
#pragma unroll
    for (int f = 0; f < 16; ++f) {
      if (f < 13) R64P_WAIT_LGKM(3);
      else if (f == 13) R64P_WAIT_LGKM(2);
      else if (f == 14) R64P_WAIT_LGKM(1);
      else R64P_WAIT_LGKM(0);
      if (f == 0) { R64P_MFMA0(sA[0], f0[0], qf[0][0]); R64P_MFMA0(sA[1], f0[0], qf[1][0]); }
      else { R64P_MFMA(sA[0], f0[f & 3], qf[0][f]); R64P_MFMA(sA[1], f0[f & 3], qf[1][f]); }
      if (f + 4 < 16) R64P_LDS(f0[f & 3], kaddr, (f + 4) * 32);
    }
    // every wave has read K(0) before any wave's first iteration overwrites K slot 0 with K(3)
    asm volatile("s_nop 15\n\ts_nop 7\n\ts_barrier" ::: "memory");
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) {
      float m = -1e30f;
#pragma unroll
      for (int i = 0; i < 16; ++i) m = fmaxf(m, sA[rb][i]);
      m = half_swap_max(m * p.c);
      sm[rb].m = m;
      sm[rb].thr = m + RESCALE_THR;
    }
  }

  unsigned long long acc_t[6] = {0, 0, 0, 0, 0, 0};
  unsigned long long tprev = 0;
  if constexpr (DBG != 0) asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tprev)::"memory");
  // Rings.  Iteration t reads K(t+1) from K slot (t+1) % 3, pre-reads K(t+2) from slot (t+2) % 3 and refills slot t % 3
  // with K(t+3); it reads V^T(t) from V slot t % 3 and refills slot (t+2) % 3 with V^T(t+2).
  unsigned kcur = kaddr + KBYTES, knext = kaddr + 2 * KBYTES;
  unsigned vrd0 = vrel0, vrd1 = vrel1;
  int slot = 0;                           // t % 3 (wave-uniform)
  // the fragment ring of the loop: the first three fragments of K(1) for iteration 0's phase 1
  half8 fr[4];
#pragma unroll
  for (int f = 0; f < 3; ++f) R64P_LDS(fr[f], kcur, f * 32);
  for (int t = 0; t < nt; t += 2) {
#pragma unroll
    for (int par = 0; par < 2; ++par) {
      const int tt = t + par;
      const int tk3 = min(tt + 3, nt - 1), tv2 = min(tt + 2, nt - 1);
      const int vwr = slot == 0 ? 2 : slot - 1;                              // (t + 2) % 3
      const char* kg_next = Kg + (size_t)tk3 * ktile;
      const char* vg_next = Vg + (size_t)tv2 * (KB * 2);
      if (par == 0)   // even tile: P(t) from sA, S(t+1) -> sB
        tile_iter<DBG>(acc_t, tprev, fr, sA, sB, qf, o, sm, p.c, kcur, knext, vrd0, vrd1, kg_next, vg_next, kstride4, vstride4,
                       koff0, voff0, kdst + slot * KBYTES, vdst + vwr * VBYTES);
      else            // odd tile: P(t) from sB, S(t+1) -> sA
        tile_iter<DBG>(acc_t, tprev, fr, sB, sA, qf, o, sm, p.c, kcur, knext, vrd0, vrd1, kg_next, vg_next, kstride4, vstride4,
                       koff0, voff0, kdst + slot * KBYTES, vdst + vwr * VBYTES);
      // rotate: K(t+2)'s slot becomes the phase-1 slot, the slot just refilled (t % 3) the pre-read slot
      kcur = knext;
      knext = kaddr + slot * KBYTES;
      const int vstep = slot == 2 ? -2 * VBYTES : VBYTES;                    // V read slot (t + 1) % 3
      vrd0 += vstep;
      vrd1 += vstep;
      slot = slot == 2 ? 0 : slot + 1;
    }
  }
  // drain the LDS-DMA and the fragment reads still in flight (the last iterations re-load clamped tiles): LDS must not be
  // written after the workgroup has ended; and the last PV MFMAs must have written O^T before the compiler's code reads it
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_nop 15\n\ts_nop 7" ::: "memory");
  if constexpr (DBG != 0) {
    if (lane == 0 && p.stamps) {
      unsigned long long* o_ = p.stamps + ((size_t)blockIdx.x * 4 + wave) * 6;
      for (int k = 0; k < 6; ++k) o_[k] = acc_t[k];
    }
  }

#pragma unroll
  for (int rb = 0; rb < RB; ++rb) {
    const float l_tot = sm[rb].l + __shfl_xor(sm[rb].l, 32);
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


// ================================================================================================
// r64m (round 4, second step; superseded as the production kernel by its 16x16x32 sibling r64x below; lab switch
// DVD_ATTN_R64M) = r64p's data movement and pipeline with
// HAND-ALLOCATED registers, ONE exp unit per fragment step, and the whole key-tile loop as ONE generated asm statement.
//  * r64p keeps 24 of a tile's 32 exponential units in phase 1 (two per fragment step) and none in phase 2b.  Letting a
//    tile's first 8 units run one tile EARLY (in the previous tile's phase 2b) gives each of the 32 steps exactly one, but
//    with compiler-allocated registers that cost 8+ live values and hipcc answered with 193 scratch reloads of Q inside
//    the loop (the kernel sits at 256 + 256).  Here an exponential OVERWRITES the S^T element it consumes and a packed
//    word is written straight into its P fragment register: the softmax owns no register at all.
//  * The registers v[32:255] and a[0:255] belong to the statements (the kernel is compiled with amdgpu_num_vgpr(32));
//    tests/test_abi.py checks on the ISA that the compiler's code never touches them.
//  * No glue: the MFMA-only ABLATION of the first version ran 2412 cycles per tile for 2048 cycles of MFMA - ~55 scalar
//    instructions of address arithmetic per tile, issued one per ~6 cycles by the SIMD's only wave, cost more than the
//    whole softmax.  Six tile variants (t % 6) turn every LDS address into base + immediate; the loop, the pointer
//    advance and the rare rescale are inside the statement.  Stamps (benchmarks/attn_stamps_r64m.py): 2126 cycles per
//    tile (r64p ~2400), MFMA-only ablation 2052; PMC: 45.8 M cycles, MFMA busy 88 % (r64p: 49.0 M, 82 %).
//  * What the cycles buy is bounded by the POWER CAP: -6.4 % cycles came out as -1.9 % time (the clock fell from 1.51 to
//    1.44 GHz); over all variants of this kernel, throughput ~ (MFMA duty)^0.44.
// gen_attn_r64m.py holds the register plan, the schedule and the generator of attn_r64m_body.inc.
// ================================================================================================
#include "attn_r64m_body.inc"
#ifdef DVD_LAB
#include "../../benchmarks/lab/csrc/attn_r64m_abl.inc"
#endif

template <int RB_, int... DT_>
__device__ __forceinline__ void r64m_store_row(_Float16* op, float inv, std::integer_sequence<int, DT_...>) {
  auto tile = [&](auto dt_c) {
    constexpr int dt = decltype(dt_c)::value;
    const floatx16 x = r64m_read_o<16 * (8 * RB_ + dt)>();
#pragma unroll
    for (int g4 = 0; g4 < 4; ++g4) {
      half4 v;
#pragma unroll
      for (int j = 0; j < 4; ++j) v[j] = (_Float16)(x[4 * g4 + j] * inv);
      *(half4*)(op + 32 * dt + 8 * g4) = v;
    }
  };
  (tile(std::integral_constant<int, DT_>{}), ...);
}

__device__ __forceinline__ const char* uniform_ptr(const char* q) {      // tell the compiler the pointer is wave-uniform
  const unsigned long long v = (unsigned long long)q;
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
  return (const char*)(((unsigned long long)hi << 32) | lo);
}

template <int DBG>
__global__ void __launch_bounds__(256, 1) __attribute__((amdgpu_num_vgpr(R64M_COMPILER_VGPRS))) flash_attn_r64m_kernel(AttnArgs p) {
  using namespace r64p;
  constexpr int D = 256, KB = 32, DT = 8, RB = 2;
#ifdef DVD_LAB
  unsigned long long ts[4] = {0, 0, 0, 0};   // lab: s_memtime at kernel entry | loop entry | loop exit | kernel exit (OUTSIDE the loop)
  if (p.stamps) ts[0] = __builtin_readcyclecounter();
#endif
  extern __shared__ __attribute__((aligned(16))) char smem[];   // [K0 | K1 | K2 | V0 | V1 | V2]
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

  {
    const int q0 = min(qb * 256 + wave * 64 + r, p.tq - 1), q1 = min(qb * 256 + wave * 64 + 32 + r, p.tq - 1);
    r64m_load_q(Qg + (size_t)q0 * p.ldq + 8 * h, Qg + (size_t)q1 * p.ldq + 8 * h);
  }
  const unsigned kstride4 = (unsigned)(2 * p.ldk * 2), vstride4 = (unsigned)(16 * p.ldvt * 2);
  const unsigned koff0 = (unsigned)(8 * wave + (lane >> 5)) * (unsigned)(p.ldk * 2) + (unsigned)(((lane & 31) ^ (lane >> 5)) * 16);
  const unsigned voff0 = (unsigned)(64 * wave + (lane >> 2)) * (unsigned)(p.ldvt * 2) + (unsigned)(((lane & 3) ^ ((lane >> 4) & 3)) * 16);
  const unsigned lds0 = (unsigned)(uintptr_t)(lptr_t)smem;
  const int kr = kappa(r);
  const unsigned kaddr = lds0 + (kr >> 1) * KPIECE + (kr & 1) * 512 + ((h ^ (kr & 1)) * 16);        // slot 0; + ks * 32
  const unsigned vrel0 = lds0 + VBASE + r * 64 + ((h ^ ((r >> 2) & 3)) * 16);                       // chunk 0, slot 0; + dt * 2048
  const unsigned vrel1 = vrel0 ^ 32;                                                                // chunk 1

  r64m_zero_o();                         // O^T: a[0:255], owned by the statements like v[32:255]
  Soft sm[RB] = {{-1e30f, -1e30f, 0.f}, {-1e30f, -1e30f, 0.f}};
  const int nt = p.tk / KB;              // even (tk % 64 == 0)
  const size_t ktile = (size_t)KB * p.ldk * 2;
  const unsigned kdst = lds0 + (4 * wave) * KPIECE, vdst = lds0 + VBASE + (4 * wave) * 1024;

  // ---- prologue: K(0), K(1), K(2) -> K slots 0, 1, 2; V(0), V(1) -> V slots 0, 1
  {
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const char* kj = Kg + (size_t)min(j, nt - 1) * ktile;
      const char* vj = Vg + (size_t)min(j, nt - 1) * (KB * 2);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        glds_piece(kj + (size_t)i * kstride4, koff0, kdst + j * KBYTES + i * KPIECE);
        if (j < 2) glds_piece(vj + (size_t)i * vstride4, voff0, vdst + j * VBYTES + i * 1024);
      }
    }
  }
  asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");

  float a0, a1;
  r64m_prologue_s0(kaddr, a0, a1);                                   // S^T(0) -> buffer A; lane-local maxima
  asm volatile("s_barrier" ::: "memory");                                // every wave has read K(0) before K slot 0 is refilled
  {
    const float mloc[2] = {a0, a1};
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) {
      const float m = half_swap_max(mloc[rb] * p.c);
      sm[rb].m = m;
      sm[rb].thr = m + RESCALE_THR;
    }
  }
  // Rings.  Tile t reads K(t+1) from K slot (t+1) % 3, pre-reads K(t+2) from slot (t+2) % 3 and refills slot t % 3 with
  // K(t+3); it reads V^T(t) from V slot t % 3 and refills slot (t+2) % 3 with V^T(t+2).  A tile comes in six variants
  // (t % 6): every LDS address in them is a loop-invariant base + an immediate; the DMA source pairs start at kg / vg and
  // advance inside the loop (no further once they reach the last tile, which is then simply re-loaded).
  unsigned koff[4], voff[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    koff[i] = koff0 + i * kstride4;
    voff[i] = voff0 + i * vstride4;
  }
  const char* kg = uniform_ptr(Kg + (size_t)min(3, nt - 1) * ktile);
  const char* vg = uniform_ptr(Vg + (size_t)min(2, nt - 1) * (KB * 2));
  const int klim = nt - 4, vlim = nt - 3;
  const unsigned kstep = (unsigned)ktile, vstep = KB * 2;
  float e0, e1;                          // side sums of a tile's early exp units (0..6), joined to l by the tile itself
  r64m_prologue_units(kaddr, p.c, sm[0].m, sm[1].m, e0, e1);             // units 0..7 of tile 0; the ring: K(1) fragments 0..2
#ifdef DVD_LAB
  if (p.stamps) ts[1] = __builtin_readcyclecounter();
#endif
#define R64M_LOOP_ARGS sm[0].l, sm[1].l, sm[0].m, sm[1].m, sm[0].thr, sm[1].thr, e0, e1, kg, vg, nt, kaddr, vrel0, vrel1, koff, voff, p.c, \
                       kdst, vdst, kstep, vstep, klim, vlim
  // the whole key-tile loop, the rare rescale block and the drain: ONE statement (gen_attn_r64m.py)
#ifdef DVD_LAB
  if constexpr (DBG == 1) r64m_loop_m16(R64M_LOOP_ARGS);
  else if constexpr (DBG == 2) r64m_loop_nobar(R64M_LOOP_ARGS);
  else if constexpr (DBG == 3) r64m_loop_mfmaonly_nobar(R64M_LOOP_ARGS);
  else if constexpr (DBG == 4) r64m_loop_mfmaonly(R64M_LOOP_ARGS);
  else if constexpr (DBG == 5) r64m_loop_noeu(R64M_LOOP_ARGS);
  else if constexpr (DBG == 6) r64m_loop_nodma(R64M_LOOP_ARGS);
  else if constexpr (DBG == 7) r64m_loop_noread(R64M_LOOP_ARGS);
  else
#endif
    r64m_loop(R64M_LOOP_ARGS);
#undef R64M_LOOP_ARGS
#ifdef DVD_LAB
  if (p.stamps) ts[2] = __builtin_readcyclecounter();
#endif

#pragma unroll
  for (int rb = 0; rb < RB; ++rb) {
    const float l_tot = sm[rb].l + __shfl_xor(sm[rb].l, 32);
    const float inv = 1.f / l_tot;
    const int qglob = qb * 256 + wave * 64 + rb * 32 + r;
    if (qglob < p.tq) {
      _Float16* op = p.O + b * p.sO + (size_t)qglob * p.ldo + (size_t)head * D + 4 * h;
      if (rb == 0) r64m_store_row<0>(op, inv, std::make_integer_sequence<int, 8>{});
      else r64m_store_row<1>(op, inv, std::make_integer_sequence<int, 8>{});
    }
  }
#ifdef DVD_LAB
  if (p.stamps) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    ts[3] = __builtin_readcyclecounter();
    if (lane == 0) {
      unsigned long long* o_ = p.stamps + ((size_t)blockIdx.x * 4 + wave) * 6;
      o_[0] = ts[1] - ts[0];
      o_[1] = ts[2] - ts[1];
      o_[2] = ts[3] - ts[2];
      o_[3] = ts[0];
      o_[4] = ts[3];
      // HW_ID (register 4, all 32 bits) | XCC_ID (register 20, bits 3:0) << 32: which CU ran this workgroup
      o_[5] = (unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 4) |
              ((unsigned long long)__builtin_amdgcn_s_getreg((3 << 11) | 20) << 32);
    }
  }
#endif
}


// ================================================================================================
// r64x (round 4, third step; the PRODUCTION kernel for head_dim 256 at tq >= R64_MIN_TQ) = r64m on
// v_mfma_f32_16x16x32_f16.  Same pipeline across key tiles, rings, LDS-DMA and register ownership (the statements own
// v[28:255] and a[0:255]); what changes with the MFMA shape - the fragment maps, the key order inside a tile, the K / V^T LDS
// images, four query blocks of 16 rows per wave instead of two of 32, the rare block inside the PV phase - is described in
// gen_attn_r64x.py, which generates attn_r64x_body.inc.
// Why it wins although it needs MORE cycles (2366 per tile against r64m's 2126; MFMA minimum 2048): every MFMA kernel of the
// step runs against the board's power cap, and the chip holds 1.75-1.8 GHz on this shape where it holds 1.45-1.5 on
// 32x32x16: 29.97 vs 32.36 ms on one box, 30.27 vs 30.77 on another (+1.6 .. 8 %).  Getting there took three measurements:
// beside 128 MFMA issues per tile no VALU instruction is free any more (first version 2711 cycles); v_dot2c_f32_f16,
// v_pk_fma_f32 and v_pk_add_f32 do not overlap with the matrix pipe at all (benchmarks/lab/opsel_lab.hip: 34 cycles per MFMA +
// instruction pair where v_fma / v_add / v_cvt_pk / v_max3 / v_exp take 17-18) - scalar row sums: 2460, scalar arguments: 2366.
// ================================================================================================
#include "attn_r64x_body.inc"
#ifdef DVD_LAB
#include "../../benchmarks/lab/csrc/attn_r64x_abl.inc"
#endif

template <int QB_>
__device__ __forceinline__ void r64x_store_rows(_Float16* op, float inv) {      // op: the lane's query row of block QB_, + 4 g dims
#pragma unroll
  for (int d4 = 0; d4 < 4; ++d4) {
    floatx16 x;
    if (d4 == 0) x = r64m_read_o<64 * QB_ + 0>();
    else if (d4 == 1) x = r64m_read_o<64 * QB_ + 16>();
    else if (d4 == 2) x = r64m_read_o<64 * QB_ + 32>();
    else x = r64m_read_o<64 * QB_ + 48>();
#pragma unroll
    for (int k = 0; k < 4; ++k) {                      // dims 16 (4 d4 + k) + 4 g .. + 3
      half4 v;
#pragma unroll
      for (int j = 0; j < 4; ++j) v[j] = (_Float16)(x[4 * k + j] * inv);
      *(half4*)(op + 16 * (4 * d4 + k)) = v;
    }
  }
}

template <int DBG>
__global__ void __launch_bounds__(256, 1) __attribute__((amdgpu_num_vgpr(R64X_COMPILER_VGPRS))) flash_attn_r64x_kernel(AttnArgs p) {
  using namespace r64p;
  constexpr int D = 256, KB = 32;
#ifdef DVD_LAB
  unsigned long long ts[4] = {0, 0, 0, 0};
  if (p.stamps) ts[0] = __builtin_readcyclecounter();
#endif
  extern __shared__ __attribute__((aligned(16))) char smem[];   // [K0 | K1 | K2 | V0 | V1 | V2]
  typedef __attribute__((address_space(3))) void* lptr_t;

  const int nwg = gridDim.x;
  int id = blockIdx.x;
  {
    const int q = nwg / 8, rr = nwg % 8, xcd = id % 8, k = id / 8;
    id = (xcd < rr ? xcd * (q + 1) : rr * (q + 1) + (xcd - rr) * q) + k;
  }
  int qb = id % p.nqb;                // 256-row query blocks
  int bh = id / p.nqb;
#ifdef DVD_LAB
  // VERDICT r4 item 7 (profiles/r5_attn_xcd_map.txt): the product map above gives an XCD a contiguous range of ids, i.e. the
  // 81 query blocks of a (sample, head) run on ONE XCD in consecutive residency slots.  1: no remap - consecutive query
  // blocks land on different XCDs, every XCD streams every problem's K / V.  2: two problems interleaved on an XCD's slots.
  if (p.xcd_map == 1) { qb = (int)blockIdx.x % p.nqb; bh = (int)blockIdx.x / p.nqb; }
  if (p.xcd_map == 2) {
    const int pair = id / (2 * p.nqb), r2 = id % (2 * p.nqb);
    if (2 * pair + 1 < nwg / p.nqb) { bh = 2 * pair + (r2 & 1); qb = r2 >> 1; }
  }
#endif
  const int head = bh % p.heads, b = bh / p.heads;
  const int kvb = b / p.kv_div;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int c16 = lane & 15, g = lane >> 4;

  const _Float16* Qg = p.Q + b * p.sQ + (size_t)head * D;
  const char* Kg = (const char*)(p.K + kvb * p.sK + (size_t)head * D);
  const char* Vg = (const char*)(p.Vt + kvb * p.sVt + (size_t)head * D * p.ldvt);

  {
    const int qrow = qb * 256 + wave * 64 + c16;
    const unsigned rowb = (unsigned)(p.ldq * 2);
    const unsigned q0 = (unsigned)min(qrow, p.tq - 1) * rowb + 16 * g, q1 = (unsigned)min(qrow + 16, p.tq - 1) * rowb + 16 * g;
    const unsigned q2 = (unsigned)min(qrow + 32, p.tq - 1) * rowb + 16 * g, q3 = (unsigned)min(qrow + 48, p.tq - 1) * rowb + 16 * g;
    r64x_load_q((const _Float16*)uniform_ptr((const char*)Qg), q0, q1, q2, q3);
  }
  // LDS-DMA sources.  K: the wave's four 1-KiB pieces are the A-rows 8 w .. 8 w + 7 of the tile = natural keys
  // 16 (w & 1) + 4 (w >> 1) + {0, 2, 8, 10} (+ lane >> 5); chunks of the odd row XOR 1 (r64p's piece image).
  // V^T: 16 dim rows per piece, chunk swizzle (-(row >> 2)) & 3.
  unsigned koff[4], voff[4];
  {
    const unsigned krow0 = 16 * (wave & 1) + 4 * (wave >> 1) + (lane >> 5);
    const unsigned kchunk = (unsigned)(((lane & 31) ^ (lane >> 5)) * 16);
    const unsigned kadd[4] = {0, 2, 8, 10};
    const unsigned vchunk = (unsigned)(((lane & 3) ^ ((0u - (unsigned)(lane >> 4)) & 3)) * 16);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      koff[j] = (krow0 + kadd[j]) * (unsigned)(p.ldk * 2) + kchunk;
      voff[j] = (unsigned)(64 * wave + 16 * j + (lane >> 2)) * (unsigned)(p.ldvt * 2) + vchunk;
    }
  }
  const unsigned lds0 = (unsigned)(uintptr_t)(lptr_t)smem;
  // fragment reads: K A-row c16 of key block kb2 = LDS row 16 kb2 + c16 (+ imm: slot, kb2, ks); V^T dim row 16 db + c16
  const unsigned kaddr = lds0 + (c16 >> 1) * KPIECE + (c16 & 1) * 512 + ((g ^ (c16 & 1)) * 16);
  const unsigned vrel = lds0 + VBASE + c16 * 64 + ((g ^ ((0u - (unsigned)(c16 >> 2)) & 3)) * 16);

  r64m_zero_o();
  const int nt = p.tk / KB;              // even (tk % 64 == 0)
  const size_t ktile = (size_t)KB * p.ldk * 2;
  const unsigned kdst = lds0 + (4 * wave) * KPIECE, vdst = lds0 + VBASE + (4 * wave) * 1024;

  // ---- prologue: K(0), K(1), K(2) -> K slots 0, 1, 2; V(0), V(1) -> V slots 0, 1
  {
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const char* kj = Kg + (size_t)min(j, nt - 1) * ktile;
      const char* vj = Vg + (size_t)min(j, nt - 1) * (KB * 2);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        glds_piece(kj, koff[i], kdst + j * KBYTES + i * KPIECE);
        if (j < 2) glds_piece(vj, voff[i], vdst + j * VBYTES + i * 1024);
      }
    }
  }
  asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");

  float a[4];
  r64x_prologue_s0(kaddr, a[0], a[1], a[2], a[3]);                      // S^T(0) -> buffer A; lane-local maxima
  asm volatile("s_barrier" ::: "memory");                                // every wave has read K(0) before K slot 0 is refilled
  float m[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {                                           // a query's 32 keys live in lanes c16 + 16 g, g = 0..3
    float x = a[q] * p.c;
    x = fmaxf(x, __shfl_xor(x, 16));
    x = fmaxf(x, __shfl_xor(x, 32));
    m[q] = x;
  }
  const char* kg = uniform_ptr(Kg + (size_t)min(3, nt - 1) * ktile);
  const char* vg = uniform_ptr(Vg + (size_t)min(2, nt - 1) * (KB * 2));
  const int klim = nt - 4, vlim = nt - 3;
  const unsigned kstep = (unsigned)ktile, vstep = KB * 2;
  r64x_prologue_units(kaddr, p.c, m[0], m[1], m[2], m[3]);              // m -> v[28:31]; tile 0: arguments, exp units 0..15
  float l[4] = {0.f, 0.f, 0.f, 0.f};
#ifdef DVD_LAB
  if (p.stamps) ts[1] = __builtin_readcyclecounter();
#endif
#define R64X_LOOP_ARGS l[0], l[1], l[2], l[3], kg, vg, nt, kaddr, vrel, koff, voff, p.c, kdst, vdst, kstep, vstep, klim, vlim
#ifdef DVD_LAB
  if constexpr (DBG == 1) r64x_loop_novalu(R64X_LOOP_ARGS);
  else if constexpr (DBG == 2) r64x_loop_nobar(R64X_LOOP_ARGS);
  else if constexpr (DBG == 3) r64x_loop_mfmaonly(R64X_LOOP_ARGS);
  else
#endif
    r64x_loop(R64X_LOOP_ARGS);
#undef R64X_LOOP_ARGS
#ifdef DVD_LAB
  if (p.stamps) ts[2] = __builtin_readcyclecounter();
#endif

  // the lane index is recomputed (mbcnt) so that nothing per-lane has to stay live across the loop: with 28 VGPRs the compiler
  // otherwise parks such values in AGPRs - which belong to the statements (tests/test_abi.py checks the ISA for exactly that)
  const int lane_e = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
  const int c16e = lane_e & 15, ge = lane_e >> 4;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    float lt = l[q];
    lt += __shfl_xor(lt, 16);
    lt += __shfl_xor(lt, 32);
    const float inv = 1.f / lt;
    const int qglob = qb * 256 + wave * 64 + 16 * q + c16e;
    if (qglob < p.tq) {
      _Float16* op = p.O + b * p.sO + (size_t)qglob * p.ldo + (size_t)head * D + 4 * ge;
      if (q == 0) r64x_store_rows<0>(op, inv);
      else if (q == 1) r64x_store_rows<1>(op, inv);
      else if (q == 2) r64x_store_rows<2>(op, inv);
      else r64x_store_rows<3>(op, inv);
    }
  }
#ifdef DVD_LAB
  if (p.stamps) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    ts[3] = __builtin_readcyclecounter();
    if (lane == 0) {
      unsigned long long* o_ = p.stamps + ((size_t)blockIdx.x * 4 + wave) * 6;
      o_[0] = ts[1] - ts[0];
      o_[1] = ts[2] - ts[1];
      o_[2] = ts[3] - ts[2];
      o_[3] = ts[0];
      o_[4] = ts[3];
      o_[5] = (unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 4) |
              ((unsigned long long)__builtin_amdgcn_s_getreg((3 << 11) | 20) << 32);
    }
  }
#endif
}


// ================================================================================================
// h64m: head_dim 64 on the decoder kernel's recipe (round 4; gen_attn_h64m.py generates attn_h64m_body.inc): 64 query rows
// per wave, S^T(t+1) and the softmax of tile t overlapped by pipelining across 32-key tiles, the loop one asm statement with
// hand-allocated registers - and TWO waves per SIMD (160 VGPRs + 64 AGPRs per wave), because head_dim 64 has four exp units
// beside every pair of MFMAs.  K image: natural key rows of 128 B, chunks XOR-swizzled by (row >> 1) & 7, rows read in kappa
// order (one fragment base per ks); V^T image as in the decoder kernel (64-byte rows, chunk ^ ((row >> 2) & 3)).
// MEASURED (lab switch DVD_ATTN_H64M; profiles/archive/r4_attn_h64m_*): correct on the first run and EXACTLY as fast as the
// compiler-scheduled flash_attn_glds_kernel<64> (10.32 vs 10.32 ms, 10.49 vs 10.62 on another box): 17.3 M vs 16.4 M cycles,
// MFMA busy 58 vs 61 %.  Its ablations say why neither moves: without the exp units 11.1 M cycles (91 % busy), i.e. the
// softmax VALU work is not hidden at all - each of a tile's 132 VALU instructions costs the SIMD ~2.4 cycles on top of
// the MFMAs, whichever kernel issues them, however they are ordered (interleaving the units' dependent fma -> exp pairs:
// no change).  Head_dim 64 is bound by the SUM of matrix and VALU time, not by its schedule.  Lab build only; its sibling
// on the other MFMA shape, h64x below, is the product kernel.
// ================================================================================================
#include "attn_h64m_body.inc"
#ifdef DVD_LAB
#include "../../benchmarks/lab/csrc/attn_h64m_abl.inc"
#endif

namespace h64m {
constexpr int KBYTES = 4096, VBYTES = 4096, VBASE = 3 * KBYTES, LDS_BYTES = 3 * KBYTES + 3 * VBYTES;
}

template <int DBG>
__global__ void __launch_bounds__(256, 2) __attribute__((amdgpu_num_vgpr(H64M_COMPILER_VGPRS))) flash_attn_h64m_kernel(AttnArgs p) {
  using namespace h64m;
  constexpr int D = 64, KB = 32;
  constexpr float RESCALE_THR = 10.f;
  extern __shared__ __attribute__((aligned(16))) char smem[];   // [K0 | K1 | K2 | V0 | V1 | V2], 4 KiB each
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
  {
    const int q0 = min(qb * 256 + wave * 64 + r, p.tq - 1), q1 = min(qb * 256 + wave * 64 + 32 + r, p.tq - 1);
    h64m_load_q(Qg + (size_t)q0 * p.ldq + 8 * h, Qg + (size_t)q1 * p.ldq + 8 * h);
  }
  // LDS-DMA sources: this wave's 1-KiB piece of a K tile = key rows 8 w .. 8 w + 7 (8 chunks each), of a V^T tile = dim rows
  // 16 w .. 16 w + 15 (4 chunks each); the swizzles are applied to the SOURCE chunk
  const int krow = 8 * wave + (lane >> 3);
  const unsigned koff = (unsigned)krow * (unsigned)(p.ldk * 2) + (unsigned)(((lane & 7) ^ ((krow >> 1) & 7)) * 16);
  const unsigned voff = (unsigned)(16 * wave + (lane >> 2)) * (unsigned)(p.ldvt * 2) + (unsigned)(((lane & 3) ^ ((lane >> 4) & 3)) * 16);
  const unsigned lds0 = (unsigned)(uintptr_t)(lptr_t)smem;
  const int kr = kappa(r);
  unsigned kf[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) kf[ks] = lds0 + kr * 128 + (((2 * ks + h) ^ ((kr >> 1) & 7)) * 16);
  const unsigned vrel0 = lds0 + VBASE + r * 64 + ((h ^ ((r >> 2) & 3)) * 16);
  const unsigned vrel1 = vrel0 ^ 32;

  h64m_zero_o();
  r64p::Soft sm[2] = {{-1e30f, -1e30f, 0.f}, {-1e30f, -1e30f, 0.f}};
  const int nt = p.tk / KB;              // even (tk % 64 == 0)
  const size_t ktile = (size_t)KB * p.ldk * 2;
  const unsigned kdst = lds0 + wave * 1024, vdst = lds0 + VBASE + wave * 1024;
#pragma unroll
  for (int j = 0; j < 3; ++j) {          // prologue: K(0..2) -> K slots 0..2, V^T(0..1) -> V slots 0, 1
    r64p::glds_piece(Kg + (size_t)min(j, nt - 1) * ktile, koff, kdst + j * KBYTES);
    if (j < 2) r64p::glds_piece(Vg + (size_t)min(j, nt - 1) * (KB * 2), voff, vdst + j * VBYTES);
  }
  asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");

  float a0, a1;
  h64m_prologue_s0(kf, a0, a1);                                           // S^T(0) -> buffer A; lane-local maxima
  asm volatile("s_barrier" ::: "memory");                                // every wave has read K(0) before K slot 0 is refilled
  {
    const float mloc[2] = {a0, a1};
#pragma unroll
    for (int rb = 0; rb < 2; ++rb) {
      const float m = half_swap_max(mloc[rb] * p.c);
      sm[rb].m = m;
      sm[rb].thr = m + RESCALE_THR;
    }
  }
  const char* kg = uniform_ptr(Kg + (size_t)min(3, nt - 1) * ktile);
  const char* vg = uniform_ptr(Vg + (size_t)min(2, nt - 1) * (KB * 2));
  const int klim = nt - 4, vlim = nt - 3;
  const unsigned kstep = (unsigned)ktile, vstep = KB * 2;
  float e0, e1;
  h64m_prologue_units(kf, p.c, sm[0].m, sm[1].m, e0, e1);                // units 0..7 of tile 0; the ring: K(1) fragments 0..2
#define H64M_LOOP_ARGS sm[0].l, sm[1].l, sm[0].m, sm[1].m, sm[0].thr, sm[1].thr, e0, e1, kg, vg, nt, kf, vrel0, vrel1, koff, voff, p.c, \
                       kdst, vdst, kstep, vstep, klim, vlim
#ifdef DVD_LAB
  if constexpr (DBG == 1) h64m_loop_noeu(H64M_LOOP_ARGS);
  else if constexpr (DBG == 2) h64m_loop_m16(H64M_LOOP_ARGS);
  else if constexpr (DBG == 3) h64m_loop_mfmaonly(H64M_LOOP_ARGS);
  else
#endif
    h64m_loop(H64M_LOOP_ARGS);
#undef H64M_LOOP_ARGS

  const int lane_e = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));     // nothing per-lane kept live across the loop
  const int re = lane_e & 31, he = lane_e >> 5;
#pragma unroll
  for (int rb = 0; rb < 2; ++rb) {
    const float l_tot = sm[rb].l + __shfl_xor(sm[rb].l, 32);
    const float inv = 1.f / l_tot;
    const int qglob = qb * 256 + wave * 64 + rb * 32 + re;
    if (qglob < p.tq) {
      _Float16* op = p.O + b * p.sO + (size_t)qglob * p.ldo + (size_t)head * D + 4 * he;
#pragma unroll
      for (int dt = 0; dt < 2; ++dt) {
        floatx16 x;
        if (rb == 0 && dt == 0) x = r64m_read_o<0>();
        else if (rb == 0) x = r64m_read_o<16>();
        else if (dt == 0) x = r64m_read_o<32>();
        else x = r64m_read_o<48>();
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
          half4 v;
#pragma unroll
          for (int j = 0; j < 4; ++j) v[j] = (_Float16)(x[4 * g4 + j] * inv);
          *(half4*)(op + 32 * dt + 8 * g4) = v;
        }
      }
    }
  }
}


// ================================================================================================
// h64x (round 4; the PRODUCTION kernel for head_dim 64 at tq >= R64_MIN_TQ): head_dim 64 on v_mfma_f32_16x16x32_f16 = r64x's
// fragment maps at h64m's geometry (gen_attn_h64x.py generates attn_h64x_body.inc): 64 query rows per wave as four query
// blocks of 16, 32-key tiles of 8 steps, two waves per SIMD (176 VGPRs + 64 AGPRs), K tiles in A-row order with 128-byte
// rows, one LDS-DMA piece per wave, stream and tile.  Priced first by an ablation of h64m (two 16x16x32 MFMAs per 32x32x16:
// 10.01 vs 10.53 ms), then built: correct on its first run; against flash_attn_glds_kernel<64> on four boxes +0.8, +2.2,
// +4.0, +5.2 % (9.80-10.25 vs 10.13-10.66 ms; profiles/archive/r4_attn_h64x_*).  19.6 M cycles against 16.7 M - the softmax VALU
// work hides even less beside 16-cycle MFMAs - at 1.96 instead of 1.65 GHz.  Then the softmax argument was taken off the
// VALU as in flash_attn_glds_kernel<64> (Q pre-scaled by c, -m through the C operand of each chain's first MFMA: the two
// waves per SIMD leave 16 registers for the tuples): 32 of 135 VALU instructions per tile gone, 9.11 vs 10.07 ms on one
// box (profiles/archive/r4_attn_h64x_negm_ab.txt) = 1161 TF/s.
// ================================================================================================
// Round 6 (LM = true, gen_attn_h64x.py's second body h64l): the row sums l leave the VALU - V^T gets a fifth dim block of ones,
// l = ones . P falls out of four more PV MFMAs per tile (a[64:79]) and is the sum of the same f16-rounded P that O^T takes.
typedef unsigned uintx4 __attribute__((ext_vector_type(4)));
#include "attn_h64x_body.inc"
#include "attn_h64l_body.inc"
#ifdef DVD_LAB
#include "../../benchmarks/lab/csrc/attn_h64x_abl.inc"
#include "../../benchmarks/lab/csrc/attn_h64l_abl.inc"
#endif

template <int DBG, bool LM>
__global__ void __launch_bounds__(256, 2) __attribute__((amdgpu_num_vgpr(H64X_COMPILER_VGPRS))) flash_attn_h64x_kernel(AttnArgs p) {
  using namespace h64m;                 // LDS geometry: 3 + 3 slots of 4 KiB
  constexpr int D = 64, KB = 32;
  extern __shared__ __attribute__((aligned(16))) char smem[];
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
  const int c16 = lane & 15, g = lane >> 4;

  const _Float16* Qg = p.Q + b * p.sQ + (size_t)head * D;
  const char* Kg = (const char*)(p.K + kvb * p.sK + (size_t)head * D);
  const char* Vg = (const char*)(p.Vt + kvb * p.sVt + (size_t)head * D * p.ldvt);
  {
    const int qrow = qb * 256 + wave * 64 + c16;
    const unsigned rowb = (unsigned)(p.ldq * 2);
    const unsigned q0 = (unsigned)min(qrow, p.tq - 1) * rowb + 16 * g, q1 = (unsigned)min(qrow + 16, p.tq - 1) * rowb + 16 * g;
    const unsigned q2 = (unsigned)min(qrow + 32, p.tq - 1) * rowb + 16 * g, q3 = (unsigned)min(qrow + 48, p.tq - 1) * rowb + 16 * g;
    if constexpr (LM) h64l_load_q((const _Float16*)uniform_ptr((const char*)Qg), q0, q1, q2, q3, p.c);
    else h64x_load_q((const _Float16*)uniform_ptr((const char*)Qg), q0, q1, q2, q3, p.c);  // Q * c (fp32 multiply, one f16 rounding)
  }
  // LDS-DMA sources.  K: this wave's piece = the tile's A-rows 8 w .. 8 w + 7 (row 16 kb2 + i holds the natural key
  // 8 (i >> 2) + 4 kb2 + (i & 3)), 8 chunks each, chunk ^ ((row >> 1) & 7).  V^T: dim rows 16 w .. 16 w + 15, 4 chunks each.
  unsigned koff, voff;
  {
    const int i = 8 * (wave & 1) + (lane >> 3), kb2 = wave >> 1;
    const int key = 8 * (i >> 2) + 4 * kb2 + (i & 3);
    koff = (unsigned)key * (unsigned)(p.ldk * 2) + (unsigned)(((lane & 7) ^ ((i >> 1) & 7)) * 16);
    voff = (unsigned)(16 * wave + (lane >> 2)) * (unsigned)(p.ldvt * 2) + (unsigned)(((lane & 3) ^ ((0u - (unsigned)(lane >> 4)) & 3)) * 16);
  }
  const unsigned lds0 = (unsigned)(uintptr_t)(lptr_t)smem;
  // fragment reads: K A-row c16 of key block kb2 (+ imm: slot, kb2 * 2048), chunk 4 ks + g; V^T dim row 16 db + c16, chunk g
  const unsigned kf0 = lds0 + c16 * 128 + (((0 + g) ^ ((c16 >> 1) & 7)) * 16);
  const unsigned kf1 = lds0 + c16 * 128 + (((4 + g) ^ ((c16 >> 1) & 7)) * 16);
  const unsigned vrel = lds0 + VBASE + c16 * 64 + ((g ^ ((0u - (unsigned)(c16 >> 2)) & 3)) * 16);

  if constexpr (LM) h64l_zero_o();
  else h64x_zero_o();
  const int nt = p.tk / KB;              // even (tk % 64 == 0)
  const size_t ktile = (size_t)KB * p.ldk * 2;
  const unsigned kdst = lds0 + wave * 1024, vdst = lds0 + VBASE + wave * 1024;
#pragma unroll
  for (int j = 0; j < 3; ++j) {          // prologue: K(0..2) -> K slots 0..2, V^T(0..1) -> V slots 0, 1
    r64p::glds_piece(Kg + (size_t)min(j, nt - 1) * ktile, koff, kdst + j * KBYTES);
    if (j < 2) r64p::glds_piece(Vg + (size_t)min(j, nt - 1) * (KB * 2), voff, vdst + j * VBYTES);
  }
  asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");

  float a[4];
  if constexpr (LM) h64l_prologue_s0(kf0, kf1, a[0], a[1], a[2], a[3]);
  else h64x_prologue_s0(kf0, kf1, a[0], a[1], a[2], a[3]);               // S^T(0) -> buffer A; lane-local maxima
  asm volatile("s_barrier" ::: "memory");                                // every wave has read K(0) before K slot 0 is refilled
  float m[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {                                           // a query's 32 keys live in lanes c16 + 16 g, g = 0..3
    float x = a[q];                                                       // Q is pre-scaled: the scores are s * c
    x = fmaxf(x, __shfl_xor(x, 16));
    x = fmaxf(x, __shfl_xor(x, 32));
    m[q] = x;
  }
  const char* kg = uniform_ptr(Kg + (size_t)min(3, nt - 1) * ktile);
  const char* vg = uniform_ptr(Vg + (size_t)min(2, nt - 1) * (KB * 2));
  const int klim = nt - 4, vlim = nt - 3;
  const unsigned kstep = (unsigned)ktile, vstep = KB * 2;
  if constexpr (LM) h64l_prologue_units(kf0, kf1, m[0], m[1], m[2], m[3]);
  else h64x_prologue_units(kf0, kf1, m[0], m[1], m[2], m[3]);            // -m -> v[160:175]; tile 0: scores - m, exp units 0..15
  float l[4] = {0.f, 0.f, 0.f, 0.f};
#define H64X_TAIL_ARGS kg, vg, nt, kf0, kf1, vrel, koff, voff, kdst, vdst, kstep, vstep, klim, vlim
  if constexpr (LM) {
    const uintx4 ones = {0x3c003c00u, 0x3c003c00u, 0x3c003c00u, 0x3c003c00u};      // the A fragment of the ones block: 16 x 32 f16 1.0
#ifdef DVD_LAB
    if constexpr (DBG == 1) h64l_loop_novalu(ones, H64X_TAIL_ARGS);
    else if constexpr (DBG == 2) h64l_loop_nobar(ones, H64X_TAIL_ARGS);
    else if constexpr (DBG == 3) h64l_loop_mfmaonly(ones, H64X_TAIL_ARGS);
    else
#endif
      h64l_loop(ones, H64X_TAIL_ARGS);
    l[0] = h64l_read_l<0>(); l[1] = h64l_read_l<1>(); l[2] = h64l_read_l<2>(); l[3] = h64l_read_l<3>();
  } else {
#ifdef DVD_LAB
    if constexpr (DBG == 1) h64x_loop_novalu(l[0], l[1], l[2], l[3], H64X_TAIL_ARGS);
    else if constexpr (DBG == 2) h64x_loop_nobar(l[0], l[1], l[2], l[3], H64X_TAIL_ARGS);
    else if constexpr (DBG == 3) h64x_loop_mfmaonly(l[0], l[1], l[2], l[3], H64X_TAIL_ARGS);
    else
#endif
      h64x_loop(l[0], l[1], l[2], l[3], H64X_TAIL_ARGS);
  }
#undef H64X_TAIL_ARGS

  const int lane_e = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));     // nothing per-lane kept live across the loop
  const int c16e = lane_e & 15, ge = lane_e >> 4;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    float lt = l[q];
    if constexpr (!LM) {               // LM: the l tile already holds the sum over all the keys
      lt += __shfl_xor(lt, 16);
      lt += __shfl_xor(lt, 32);
    }
    const float inv = 1.f / lt;
    const int qglob = qb * 256 + wave * 64 + 16 * q + c16e;
    if (qglob < p.tq) {
      _Float16* op = p.O + b * p.sO + (size_t)qglob * p.ldo + (size_t)head * D + 4 * ge;
      floatx16 x;                      // tiles (db = 0..3, q): a[16 q + 4 db + r] = dim 16 db + 4 g + r
      if (q == 0) x = r64m_read_o<0>();
      else if (q == 1) x = r64m_read_o<16>();
      else if (q == 2) x = r64m_read_o<32>();
      else x = r64m_read_o<48>();
#pragma unroll
      for (int db = 0; db < 4; ++db) {
        half4 v;
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = (_Float16)(x[4 * db + j] * inv);
        *(half4*)(op + 16 * db) = v;
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

// 64 query rows per wave (256 per workgroup) from this many query rows on: a (batch, head) then has >= 21 workgroups of
// its own, so that ONE document (two samples) already fills the 256 CUs.  At 16 samples the generated kernels also win
// below it (profiles/archive/r4_attn_midsize_threshold.txt: head_dim 256 +12 .. 38 % from T = 1024 on, head_dim 64 +5 .. 17 % from
// T = 2304 on), but with one document their 256-row workgroups would leave half of the CUs idle - and the choice must not
// depend on the batch.  The choice depends on the problem's SHAPE only (head_dim, tq, tk) - never on the batch - so a document
// takes the same kernel, and the same online-softmax tile order, alone or in a batch (bit-identical results).
static constexpr int R64_MIN_TQ = 5376;

extern "C" const char* dvd_flash_attn_kernel_name(int head_dim, int tq, int tk) {
  if (head_dim != 64 && head_dim != 256) return "";
  if (tk % 64 != 0) return head_dim == 256 ? "flash_attn_kernel<256>" : "flash_attn_kernel<64>";
  if (head_dim == 256) return tq >= R64_MIN_TQ ? "flash_attn_r64x_kernel<0>" : "flash_attn_glds_kernel<256, 0>";
  return tq >= R64_MIN_TQ ? "flash_attn_h64x_kernel<0, true>" : "flash_attn_glds_kernel<64, 0>";
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
  p.xcd_map = 0;
  const long nwg = (long)p.nqb * d->heads * d->batch;
  DVD_REQUIRE(nwg < (1l << 31), "flash_attn: grid too large");
  hipStream_t st = (hipStream_t)stream;
  bool fast = d->tk % 64 == 0;      // the LDS-DMA kernels stream whole 64-key tiles; ragged key counts take the
                                    // register-staged kernel (masked tail)
  bool r64 = d->tq >= R64_MIN_TQ;      // 64 query rows per wave (256 per workgroup): the generated kernels r64x / h64x
  // hipFuncSetAttribute is per DEVICE: remember which devices have been set up (one bit each), lock-free - the
  // attribute call is idempotent, so two threads racing on a device's first launch both set it and both are right
  static DeviceOnce attr_done;
  const unsigned long long dev_bit = DeviceOnce::current_bit();
  const bool first_on_device = attr_done.need(dev_bit);
  if (first_on_device) {
    allow_lds(flash_attn_glds_kernel<256, 0>, 2 * (64 * 512 + 256 * 128));
    allow_lds(flash_attn_r64x_kernel<0>, r64p::LDS_BYTES);
    allow_lds(flash_attn_h64x_kernel<0, true>, h64m::LDS_BYTES);
    allow_lds(flash_attn_kernel<256>, 2 * (64 * (2 * 256 + 16) + 256 * (2 * 64 + 16)));
    attr_done.done(dev_bit);
  }
#ifdef DVD_LAB
  // ---- lab build: every experiment and diagnostic variant behind its environment switch ----
  p.stamps = g_attn_stamps;
  if (const char* xm = getenv("DVD_ATTN_XCDMAP")) p.xcd_map = atoi(xm);
  const bool dbg = getenv("DVD_ATTN_DEBUG"), bulk = getenv("DVD_ATTN_BULK");
  if (getenv("DVD_ATTN_V1")) fast = false;
  if (getenv("DVD_ATTN_R64") || getenv("DVD_ATTN_R64P") || getenv("DVD_ATTN_R64M") || getenv("DVD_ATTN_R64M_ABL") || getenv("DVD_ATTN_R64X_ABL") ||
      getenv("DVD_ATTN_R64OLD"))
    r64 = true;
  if (getenv("DVD_ATTN_R32") || getenv("DVD_ATTN_PIPE") || getenv("DVD_ATTN_GLDS64") || bulk) r64 = false;
  if (first_on_device) {
    constexpr int LDS = 2 * (64 * 512 + 256 * 128);
    allow_lds(flash_attn_dsplit_kernel, LDS);
    allow_lds(flash_attn_glds_kernel<256, 1>, LDS);
    allow_lds(flash_attn_glds_kernel<256, 2>, LDS);
    allow_lds(flash_attn_glds_kernel<256, 3>, LDS);
    allow_lds(flash_attn_pipe_kernel<0>, LDS);
    allow_lds(flash_attn_pipe_kernel<1>, LDS);
    allow_lds(flash_attn_r64_kernel<0>, 2 * (32 * 512 + 256 * 64));
    allow_lds(flash_attn_r64_kernel<1>, 2 * (32 * 512 + 256 * 64));
    allow_lds(flash_attn_r64p_kernel<0>, r64p::LDS_BYTES);
    allow_lds(flash_attn_r64p_kernel<1>, r64p::LDS_BYTES);
    allow_lds(flash_attn_r64m_kernel<0>, r64p::LDS_BYTES);
    allow_lds(flash_attn_r64x_kernel<1>, r64p::LDS_BYTES);
    allow_lds(flash_attn_r64x_kernel<2>, r64p::LDS_BYTES);
    allow_lds(flash_attn_r64x_kernel<3>, r64p::LDS_BYTES);
    allow_lds(flash_attn_r64m_kernel<1>, r64p::LDS_BYTES);
    allow_lds(flash_attn_r64m_kernel<2>, r64p::LDS_BYTES);
    allow_lds(flash_attn_r64m_kernel<3>, r64p::LDS_BYTES);
    allow_lds(flash_attn_r64m_kernel<4>, r64p::LDS_BYTES);
    allow_lds(flash_attn_r64m_kernel<5>, r64p::LDS_BYTES);
    allow_lds(flash_attn_r64m_kernel<6>, r64p::LDS_BYTES);
    allow_lds(flash_attn_r64m_kernel<7>, r64p::LDS_BYTES);
  }
  if (fast && d->head_dim == 256) {
    constexpr int LDS = 2 * (64 * 512 + 256 * 128);
    if (getenv("DVD_ATTN_DSPLIT")) {   // measured slower (732 vs 812 TF/s)
      flash_attn_dsplit_kernel<<<(unsigned)nwg, 512, LDS, st>>>(p);
      return check_launch("flash_attn(lab dsplit)");
    }
    if (r64 && getenv("DVD_ATTN_R64X_ABL")) {
      // TIMING ABLATIONS of the production kernel's loop (garbage results): 1 no VALU | 2 no barrier | 3 MFMAs only
      p.nqb = cdiv(d->tq, 256);
      const unsigned g = (unsigned)((long)p.nqb * d->heads * d->batch);
      switch (atoi(getenv("DVD_ATTN_R64X_ABL"))) {
        case 1: flash_attn_r64x_kernel<1><<<g, 256, r64p::LDS_BYTES, st>>>(p); break;
        case 2: flash_attn_r64x_kernel<2><<<g, 256, r64p::LDS_BYTES, st>>>(p); break;
        case 3: flash_attn_r64x_kernel<3><<<g, 256, r64p::LDS_BYTES, st>>>(p); break;
        default: flash_attn_r64x_kernel<0><<<g, 256, r64p::LDS_BYTES, st>>>(p);
      }
      return check_launch("flash_attn(lab r64x ablation)");
    }
    if (r64 && (getenv("DVD_ATTN_R64M") || getenv("DVD_ATTN_R64M_ABL"))) {
      // round 4's second step: the same pipeline on the 32x32x16 MFMA (superseded by r64x: 1.6-8 % slower by box).
      // DVD_ATTN_R64M_ABL: timing ablations of ITS loop (garbage results): 1 16x16x32 MFMAs, same FLOPs (power) | 2 no barrier
      // | 3 MFMAs only, no barrier | 4 MFMAs only | 5 no softmax VALU | 6 no LDS-DMA | 7 no fragment reads
      p.nqb = cdiv(d->tq, 256);
      const unsigned g = (unsigned)((long)p.nqb * d->heads * d->batch);
      switch (getenv("DVD_ATTN_R64M_ABL") ? atoi(getenv("DVD_ATTN_R64M_ABL")) : 0) {
        case 1: flash_attn_r64m_kernel<1><<<g, 256, r64p::LDS_BYTES, st>>>(p); break;
        case 2: flash_attn_r64m_kernel<2><<<g, 256, r64p::LDS_BYTES, st>>>(p); break;
        case 3: flash_attn_r64m_kernel<3><<<g, 256, r64p::LDS_BYTES, st>>>(p); break;
        case 4: flash_attn_r64m_kernel<4><<<g, 256, r64p::LDS_BYTES, st>>>(p); break;
        case 5: flash_attn_r64m_kernel<5><<<g, 256, r64p::LDS_BYTES, st>>>(p); break;
        case 6: flash_attn_r64m_kernel<6><<<g, 256, r64p::LDS_BYTES, st>>>(p); break;
        case 7: flash_attn_r64m_kernel<7><<<g, 256, r64p::LDS_BYTES, st>>>(p); break;
        default: flash_attn_r64m_kernel<0><<<g, 256, r64p::LDS_BYTES, st>>>(p);
      }
      return check_launch("flash_attn(lab r64m)");
    }
    if (r64 && getenv("DVD_ATTN_R64P")) {     // round 4's first step (compiler-allocated registers), with or without stamps
      p.nqb = cdiv(d->tq, 256);
      const unsigned g = (unsigned)((long)p.nqb * d->heads * d->batch);
      if (dbg) flash_attn_r64p_kernel<1><<<g, 256, r64p::LDS_BYTES, st>>>(p);
      else flash_attn_r64p_kernel<0><<<g, 256, r64p::LDS_BYTES, st>>>(p);
      return check_launch("flash_attn(lab r64p)");
    }
    if (r64 && getenv("DVD_ATTN_R64OLD")) {   // rounds 1-3's production kernel (lab include), with or without its stamps
      p.nqb = cdiv(d->tq, 256);
      const unsigned g = (unsigned)((long)p.nqb * d->heads * d->batch);
      if (dbg) flash_attn_r64_kernel<1><<<g, 256, 2 * (32 * 512 + 256 * 64), st>>>(p);
      else flash_attn_r64_kernel<0><<<g, 256, 2 * (32 * 512 + 256 * 64), st>>>(p);
      return check_launch("flash_attn(lab r64 old)");
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
    if (getenv("DVD_ATTN_H64X") || (r64 && (getenv("DVD_ATTN_H64X_ABL") || getenv("DVD_ATTN_H64X_NOLM")))) {   // the production kernel forced at any size; _ABL: ablations
      p.nqb = cdiv(d->tq, 256);
      const unsigned g = (unsigned)((long)p.nqb * d->heads * d->batch);
      const int abl = getenv("DVD_ATTN_H64X_ABL") ? atoi(getenv("DVD_ATTN_H64X_ABL")) : 0;
      if (getenv("DVD_ATTN_H64X_NOLM")) {     // round 4's body: the row sums on the VALU (superseded by the LM body in round 6)
        switch (abl) {
          case 1: flash_attn_h64x_kernel<1, false><<<g, 256, h64m::LDS_BYTES, st>>>(p); break;
          case 2: flash_attn_h64x_kernel<2, false><<<g, 256, h64m::LDS_BYTES, st>>>(p); break;
          case 3: flash_attn_h64x_kernel<3, false><<<g, 256, h64m::LDS_BYTES, st>>>(p); break;
          default: flash_attn_h64x_kernel<0, false><<<g, 256, h64m::LDS_BYTES, st>>>(p);
        }
        return check_launch("flash_attn(lab h64x, row sums on the VALU)");
      }
      switch (abl) {
        case 1: flash_attn_h64x_kernel<1, true><<<g, 256, h64m::LDS_BYTES, st>>>(p); break;
        case 2: flash_attn_h64x_kernel<2, true><<<g, 256, h64m::LDS_BYTES, st>>>(p); break;
        case 3: flash_attn_h64x_kernel<3, true><<<g, 256, h64m::LDS_BYTES, st>>>(p); break;
        default: flash_attn_h64x_kernel<0, true><<<g, 256, h64m::LDS_BYTES, st>>>(p);
      }
      return check_launch("flash_attn(lab h64x)");
    }
    if (getenv("DVD_ATTN_H64M")) {   // head_dim 64 on the decoder kernel's recipe (generated loop, two waves per SIMD); _ABL: ablations
      p.nqb = cdiv(d->tq, 256);
      const unsigned g = (unsigned)((long)p.nqb * d->heads * d->batch);
      switch (getenv("DVD_ATTN_H64M_ABL") ? atoi(getenv("DVD_ATTN_H64M_ABL")) : 0) {
        case 1: flash_attn_h64m_kernel<1><<<g, 256, h64m::LDS_BYTES, st>>>(p); break;
        case 2: flash_attn_h64m_kernel<2><<<g, 256, h64m::LDS_BYTES, st>>>(p); break;
        case 3: flash_attn_h64m_kernel<3><<<g, 256, h64m::LDS_BYTES, st>>>(p); break;
        default: flash_attn_h64m_kernel<0><<<g, 256, h64m::LDS_BYTES, st>>>(p);
      }
      return check_launch("flash_attn(lab h64m)");
    }
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
  // ---- product dispatch: six kernels, chosen by (head_dim, tq, tk) ----
  if (fast && r64 && d->head_dim == 256) {
    p.nqb = cdiv(d->tq, 256);
    flash_attn_r64x_kernel<0><<<(unsigned)((long)p.nqb * d->heads * d->batch), 256, r64p::LDS_BYTES, st>>>(p);
  } else if (fast && d->head_dim == 256) {
    flash_attn_glds_kernel<256, 0><<<(unsigned)nwg, 256, 2 * (64 * 512 + 256 * 128), st>>>(p);
  } else if (fast && r64) {            // head_dim 64 at production sizes: the generated 16x16x32 loop, 256-row workgroups
    p.nqb = cdiv(d->tq, 256);
    flash_attn_h64x_kernel<0, true><<<(unsigned)((long)p.nqb * d->heads * d->batch), 256, h64m::LDS_BYTES, st>>>(p);
  } else if (fast) {
    flash_attn_glds_kernel<64, 0><<<(unsigned)nwg, 256, 2 * (64 * 128 + 64 * 128), st>>>(p);
  } else if (d->head_dim == 256) {
    flash_attn_kernel<256><<<(unsigned)nwg, 256, 2 * (64 * (2 * 256 + 16) + 256 * (2 * 64 + 16)), st>>>(p);
  } else {
    flash_attn_kernel<64><<<(unsigned)nwg, 256, 2 * (64 * (2 * 64 + 16) + 64 * (2 * 64 + 16)), st>>>(p);
  }
  return check_launch("flash_attn");
}
