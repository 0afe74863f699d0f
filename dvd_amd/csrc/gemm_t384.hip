// ================================================================================================
// Round 5: gemm_nt_t384_kernel - 384 x 256 tiles, K in 32-deep half slabs through a FOUR-slot LDS ring
// (all 160 KiB), the K loop ONE generated asm statement with hand-allocated registers (gen_gemm_t384.py: tile, ring, register
// plan, schedule, why).  What the ablations of gemm_nt_big_kernel said (profiles/r5_gemm_ablation.txt): its MFMAs alone take
// 1.98 ms and its loads alone 1.50 ms of the 3.18 ms the whole N = 3072 launch takes - the two barely overlap, because a
// slab's 64 KiB are requested by all 256 CUs at once at the top of the slab and waited for (vmcnt(0) + barrier) at its bottom,
// one HBM/fabric round trip per slab; the L2 -> LDS path itself streams 47 B/clk per CU when it is kept busy
// (benchmarks/lab/l2path_lab.hip), not the 33 that round 1's tile-synchronous probe measured.  Here a half slab's pieces
// are in flight for 2.5 iterations and the queue is never drained inside the loop.
// Same MFMA instruction, same ascending-k order per accumulator and the same epilogue arithmetic as gemm_nt_big_kernel:
// bit-identical results (tests/test_gpu_gemm.py).  Plain / residual epilogues (column bias, ReLU / GELU); N % 256 == 0,
// K % 128 == 0, K >= 256; any M (rows clamped on load, masked on store).
// ================================================================================================
#include "gemm_common.h"

namespace dvd {

#include "gemm_t384_body.inc"
#ifdef DVD_LAB
#include "../../benchmarks/lab/csrc/gemm_t384_abl.inc"
#endif

namespace t384 {
constexpr int HALF = 40960, BOFF = 24576, LDS_BYTES = 4 * HALF;
constexpr int STAGE = 2 * 32 * 64 + 128;   // floats per wave: two 32 x 64 blocks + the wave's 128 column biases

__device__ __forceinline__ const char* uniform_ptr(const char* q) {      // tell the compiler the pointer is wave-uniform
  const unsigned long long v = (unsigned long long)q;
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
  return (const char*)(((unsigned long long)hi << 32) | lo);
}
__device__ __forceinline__ void dma_piece(const char* gb, unsigned voff, unsigned lds) {
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %0" ::"s"(gb), "s"(lds), "v"(voff) : "memory");
}

// One 32 x 64 block (accumulators t0 | t1 = columns 0-31 | 32-63) staged through the wave's LDS region and written as
// row-contiguous 16-byte accesses, 8 consecutive columns per lane: epilogue_readback's arithmetic in its order (gemm.hip), so
// the bits are gemm_nt_big_kernel's.  The residual rows ride a rolling window that runs ACROSS the six blocks of a wave tile
// (slot g % 2 of global iteration g = 4 block + it): a load is always issued before the stores it overlaps, so waiting
// for it never waits for a store (vmcnt retires in order and counts stores); the column biases come from LDS.
__device__ __forceinline__ void stage2(float* stage, const floatx16& t0, const floatx16& t1, int lane) {
  const int r = lane & 31, h = lane >> 5;
#pragma unroll
  for (int i = 0; i < 16; ++i) stage[cd_row(i, h) * 64 + r] = t0[i];
#pragma unroll
  for (int i = 0; i < 16; ++i) stage[cd_row(i, h) * 64 + 32 + r] = t1[i];
}

template <int EPI, int BLK>
__device__ __forceinline__ void block(const GemmArgs& p, float* stage, const float* stage_bias, const floatx16& t0, const floatx16& t1,
                                      int row_w, int col_w, int lane_in, float* C32, _Float16* C16, const float* res, bool has_bias,
                                      floatx4 (&rs0)[2], floatx4 (&rs1)[2]) {
  constexpr int MB = BLK >> 1, NB = BLK & 1;
  int lane = lane_in;
  asm volatile("" : "+v"(lane));           // per-block addressing is recomputed here, not hoisted above the six blocks and spilled
  if constexpr (BLK >= 2) stage2(stage, t0, t1, lane);
  const int c8 = (lane & 7) * 8;
  const int col = col_w + 64 * NB + c8;
  float bc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (has_bias) {
    const float* bl = stage_bias + 64 * NB + c8;
    const floatx4 b0 = *(const floatx4*)bl, b1 = *(const floatx4*)(bl + 4);
#pragma unroll
    for (int e = 0; e < 4; ++e) { bc[e] = b0[e]; bc[4 + e] = b1[e]; }
  }
  const bool relu = p.act == 2;
#pragma unroll
  for (int it = 0; it < 4; ++it) {
    constexpr int G0 = 4 * BLK;
    const int lr = it * 8 + (lane >> 3);
    const int row = row_w + 32 * MB + lr;
    const floatx4 s0 = *(const floatx4*)(stage + lr * 64 + c8), s1 = *(const floatx4*)(stage + lr * 64 + c8 + 4);
    floatx4 rcur0 = {0.f, 0.f, 0.f, 0.f}, rcur1 = {0.f, 0.f, 0.f, 0.f};
    if constexpr (EPI == 1) {
      rcur0 = rs0[(G0 + it) & 1];
      rcur1 = rs1[(G0 + it) & 1];
      if (G0 + it + 2 < 24) {             // global iteration g + 2 (possibly in the next block), issued BEFORE this iteration's stores
        const int g2 = G0 + it + 2, blk2 = g2 >> 2, it2 = g2 & 3;
        const int row2 = row_w + 32 * (blk2 >> 1) + it2 * 8 + (lane >> 3), col2 = col_w + 64 * (blk2 & 1) + c8;
        const float* rp = res + (size_t)min(row2, p.M - 1) * p.ldres + col2;
        rs0[(G0 + it) & 1] = *(const floatx4*)rp;
        rs1[(G0 + it) & 1] = *(const floatx4*)(rp + 4);
      }
    }
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = (e < 4 ? s0[e] : s1[e - 4]) + bc[e] + 0.f;
    if constexpr (EPI == 0) {             // GELU + residual takes gemm_nt_big_kernel (no caller in the engine)
      if (p.act == 1) {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = gelu_tanh(v[e]);
      }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = relu ? fmaxf(v[e], 0.f) : v[e];
    if (row < p.M) {
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

template <int EPI>
__device__ __forceinline__ void epilogue(const GemmArgs& p, float* stage, const floatx16 (&acc)[12], int row_w, int col_w, int lane,
                                         float* C32, _Float16* C16, const float* bias, const float* res) {
  // the four accumulators that live in VGPRs (row m = 0) go to the two staging areas first: 64 registers free for what follows
  float* const sa = stage;
  float* const sb = stage + 32 * 64;
  float* const sbias = stage + 2 * 32 * 64;
  stage2(sa, acc[0], acc[1], lane);
  stage2(sb, acc[2], acc[3], lane);
  if (bias) {                               // the wave's 128 column biases -> LDS: no global load inside the store sequence
    if (lane < 32) *(floatx4*)(sbias + 4 * lane) = *(const floatx4*)(bias + col_w + 4 * lane);
  }
  floatx4 rs0[2], rs1[2];
  if constexpr (EPI == 1) {
#pragma unroll
    for (int g = 0; g < 2; ++g) {
      const float* rp = res + (size_t)min(row_w + g * 8 + (lane >> 3), p.M - 1) * p.ldres + col_w + (lane & 7) * 8;
      rs0[g] = *(const floatx4*)rp;
      rs1[g] = *(const floatx4*)(rp + 4);
    }
  }
  const bool hb = bias != nullptr;
  block<EPI, 0>(p, sa, sbias, acc[0], acc[1], row_w, col_w, lane, C32, C16, res, hb, rs0, rs1);
  block<EPI, 1>(p, sb, sbias, acc[2], acc[3], row_w, col_w, lane, C32, C16, res, hb, rs0, rs1);
  block<EPI, 2>(p, sa, sbias, acc[4], acc[5], row_w, col_w, lane, C32, C16, res, hb, rs0, rs1);
  block<EPI, 3>(p, sb, sbias, acc[6], acc[7], row_w, col_w, lane, C32, C16, res, hb, rs0, rs1);
  block<EPI, 4>(p, sa, sbias, acc[8], acc[9], row_w, col_w, lane, C32, C16, res, hb, rs0, rs1);
  block<EPI, 5>(p, sb, sbias, acc[10], acc[11], row_w, col_w, lane, C32, C16, res, hb, rs0, rs1);
}
}  // namespace t384

template <int DBG>   // lab: 1 no LDS-DMA in the loop, 2 no fragment reads, 3 no barrier, 4 MFMAs only, 5 s_memtime stamps
__global__ void __launch_bounds__(512, 2) gemm_nt_t384_kernel(GemmArgs p) {
  using namespace t384;
  extern __shared__ __attribute__((aligned(1024))) char smem[];   // 4 x [A half slab | B half slab]; then the epilogue's staging
  typedef __attribute__((address_space(3))) void* lptr_t;
  const int nwg = p.ntm * p.ntn;
  const int z = blockIdx.y;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 1, wc = wave & 1;
  const unsigned lds0 = (unsigned)(uintptr_t)(lptr_t)smem;
  const int nloop = p.K / 128 - 2;
  for (int vid = blockIdx.x; vid < nwg; vid += gridDim.x) {
    int tm, tn;
    tile_coords(vid, p.ntm, p.ntn, tm, tn);
    tm = __builtin_amdgcn_readfirstlane(tm); tn = __builtin_amdgcn_readfirstlane(tn);
    const int bm0 = tm * 384, bn0 = tn * 256;
#ifdef DVD_LAB
    unsigned long long t0 = 0, t1 = 0, t2 = 0;
    if constexpr (DBG == 5) t0 = __builtin_amdgcn_s_memtime();
#endif
    floatx16 acc[12];
    {
      const _Float16* A = (const _Float16*)p.A + z * p.sA;
      const _Float16* B = (const _Float16*)p.B + z * p.sB;
      // per-lane source offsets of the wave's 3 + 2 pieces of a half slab: piece = 16 rows x 64 B, LDS slot (row, pos = lane & 3)
      // <- chunk pos ^ ((row >> 2) & 3); rows clamped into the matrix (ragged last tiles)
      unsigned va[3], vb[2];
      const int pos = lane & 3;
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        const int row = 16 * (3 * wave + i) + (lane >> 2);
        const int ra = min(bm0 + row, p.M - 1) - bm0;
        va[i] = (unsigned)ra * (unsigned)(p.lda * 2) + (pos ^ ((row >> 2) & 3)) * 16;
      }
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int row = 16 * (2 * wave + i) + (lane >> 2);
        const int rb = min(bn0 + row, p.N - 1) - bn0;
        vb[i] = (unsigned)rb * (unsigned)(p.ldb * 2) + (pos ^ ((row >> 2) & 3)) * 16;
      }
      const char* Atile = uniform_ptr((const char*)(A + (size_t)bm0 * p.lda));
      const char* Btile = uniform_ptr((const char*)(B + (size_t)bn0 * p.ldb));
      const unsigned pda = lds0 + (3 * wave) * 1024, pdb = lds0 + BOFF + (2 * wave) * 1024;
      // prologue: half slabs 0, 1, 2 -> slots 0, 1, 2
#pragma unroll
      for (int j = 0; j < 3; ++j) {
#pragma unroll
        for (int i = 0; i < 3; ++i) dma_piece(Atile + j * 64, va[i], pda + j * HALF + i * 1024);
#pragma unroll
        for (int i = 0; i < 2; ++i) dma_piece(Btile + j * 64, vb[i], pdb + j * HALF + i * 1024);
      }
      // fragment read bases in slot 0: row r of the wave's first 32-row block, chunk (2 s + h) ^ ((r >> 2) & 3)
      const int r = lane & 31, h = lane >> 5;
      const unsigned ch = (unsigned)((h ^ ((r >> 2) & 3)) * 16);
      const unsigned fa0 = lds0 + (96 * wr + r) * 64 + ch, fa1 = fa0 ^ 32;
      const unsigned fb0 = lds0 + BOFF + (128 * wc + r) * 64 + ch, fb1 = fb0 ^ 32;
      asm volatile("s_waitcnt vmcnt(10)\n\ts_barrier" ::: "memory");      // half slab 0 has landed
#ifdef DVD_LAB
      if constexpr (DBG == 5) t1 = __builtin_amdgcn_s_memtime();
#endif
#define T384_ARGS acc, Atile + 3 * 64, Btile + 3 * 64, nloop, pda, pdb, va[0], va[1], va[2], vb[0], vb[1], fa0, fa1, fb0, fb1
#ifdef DVD_LAB
      if constexpr (DBG == 1) t384_loop_nodma(T384_ARGS);
      else if constexpr (DBG == 2) t384_loop_noread(T384_ARGS);
      else if constexpr (DBG == 3) t384_loop_nobar(T384_ARGS);
      else if constexpr (DBG == 4) t384_loop_mfmaonly(T384_ARGS);
      else
#endif
        t384_loop(T384_ARGS);
#undef T384_ARGS
    }
    // every wave has read its last fragments: LDS becomes the epilogue's staging area
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
#ifdef DVD_LAB
    if constexpr (DBG == 5) t2 = __builtin_amdgcn_s_memtime();
#endif
    {
      float* C32 = p.C32 ? p.C32 + z * p.sC32 : nullptr;
      _Float16* C16 = p.C16 ? p.C16 + z * p.sC16 : nullptr;
      const float* bias = p.bias ? p.bias + z * p.sBias : nullptr;
      const float* res = p.res ? p.res + z * p.sRes : nullptr;
      // the lane index is recomputed so that nothing per-lane has to stay live across the loop statement
      const int lane_e = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
      float* stage = (float*)smem + wave * STAGE;
      const int row_w = bm0 + 96 * wr, col_w = bn0 + 128 * wc;
      if (res) epilogue<1>(p, stage, acc, row_w, col_w, lane_e, C32, C16, bias, res);
      else epilogue<0>(p, stage, acc, row_w, col_w, lane_e, C32, C16, bias, res);
    }
#ifdef DVD_LAB
    if constexpr (DBG == 5) {
      const unsigned long long t3 = __builtin_amdgcn_s_memtime();
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      const unsigned long long t4 = __builtin_amdgcn_s_memtime();
      if (lane == 0 && p.stamps && vid < 256 * 64) {
        unsigned long long* o = p.stamps + ((size_t)vid * 8 + wave) * 8;
        o[0] = t0; o[1] = t1; o[2] = t2; o[3] = t3; o[4] = t4; o[5] = t2; o[6] = t2; o[7] = t2;
      }
    }
#endif
    __syncthreads();   // every wave has read its staging region back: the next tile's LDS-DMA may overwrite it
  }
}

int launch_gemm_t384(const GemmArgs& p, int batch, int dbg, void* stream) {
  static DeviceOnce once_t;
  if (const auto bit = DeviceOnce::current_bit(); once_t.need(bit)) {
    (void)hipFuncSetAttribute((const void*)gemm_nt_t384_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, t384::LDS_BYTES);
#ifdef DVD_LAB
    (void)hipFuncSetAttribute((const void*)gemm_nt_t384_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, t384::LDS_BYTES);
    (void)hipFuncSetAttribute((const void*)gemm_nt_t384_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, t384::LDS_BYTES);
    (void)hipFuncSetAttribute((const void*)gemm_nt_t384_kernel<3>, hipFuncAttributeMaxDynamicSharedMemorySize, t384::LDS_BYTES);
    (void)hipFuncSetAttribute((const void*)gemm_nt_t384_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize, t384::LDS_BYTES);
    (void)hipFuncSetAttribute((const void*)gemm_nt_t384_kernel<5>, hipFuncAttributeMaxDynamicSharedMemorySize, t384::LDS_BYTES);
#endif
    once_t.done(bit);
  }
  int nblk = p.ntm * p.ntn;
  if (nblk > 256) nblk = 256;
  const dim3 grid(nblk, batch);
  hipStream_t st = (hipStream_t)stream;
#ifdef DVD_LAB
  switch (dbg) {
    case 1: gemm_nt_t384_kernel<1><<<grid, 512, t384::LDS_BYTES, st>>>(p); break;
    case 2: gemm_nt_t384_kernel<2><<<grid, 512, t384::LDS_BYTES, st>>>(p); break;
    case 3: gemm_nt_t384_kernel<3><<<grid, 512, t384::LDS_BYTES, st>>>(p); break;
    case 4: gemm_nt_t384_kernel<4><<<grid, 512, t384::LDS_BYTES, st>>>(p); break;
    case 5: gemm_nt_t384_kernel<5><<<grid, 512, t384::LDS_BYTES, st>>>(p); break;
    default: gemm_nt_t384_kernel<0><<<grid, 512, t384::LDS_BYTES, st>>>(p);
  }
#else
  (void)dbg;
  gemm_nt_t384_kernel<0><<<grid, 512, t384::LDS_BYTES, st>>>(p);
#endif
  return check_launch("gemm_nt(t384)");
}

}  // namespace dvd
