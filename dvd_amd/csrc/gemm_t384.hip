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
// X16 (round 6): the accumulators come out of the 16x16x32 loop as 16 x 16 quads (gen_gemm_t384.py, x_iteration): register
// 4 q + j of a 32 x 32 block, q = 2 mi + ni, is row 16 mi + 4 g + j, column 16 ni + c for lane (c = l & 15, g = l >> 4).  One
// v_permlane16_swap_b32 of the registers (mi, ni = 0, j) and (mi, ni = 1, j) - [x0 x1 x2 x3], [y0 y1 y2 y3] by 16-lane group ->
// [x0 y0 x2 y2], [x1 y1 x3 y3] (benchmarks/lab/permlane_lab.hip) - leaves register i = 8 mi + 4 y + j holding, for lane
// (r = l & 31, h = l >> 5), row 16 mi + 4 y + j + 8 h and column r: two rows of 32 consecutive columns per register, the shape
// every epilogue below was written for (the 32x32x16 layout: row 8 (i >> 2) + (i & 3) + 4 h).
template <bool X16> __device__ __forceinline__ constexpr int acc_row(int i, int h) {
  return X16 ? 16 * (i >> 3) + 4 * ((i >> 2) & 1) + (i & 3) + 8 * h : cd_row(i, h);
}
template <bool X16> constexpr int HROWS = X16 ? 8 : 4;        // row distance between the two lane halves of a register
// As asm statements: with __builtin_amdgcn_permlane16_swap this hipcc drops the instruction's SECOND result (it re-uses the source
// register for the next swap's operand: found on the ISA - rows 4-7 / 12-15 of every block came out as copies of rows 0-3 / 8-11).
// Four independent swaps per statement (one half tile: registers j and 4 + j), one s_nop pair around them: the VALU-write ->
// permlane-read and permlane-write -> VALU-read hazards the compiler would pad itself.
__device__ __forceinline__ void swap16x4(float& x0, float& y0, float& x1, float& y1, float& x2, float& y2, float& x3, float& y3) {
  asm("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\tv_permlane16_swap_b32 %2, %3\n\tv_permlane16_swap_b32 %4, %5\n\t"
      "v_permlane16_swap_b32 %6, %7\n\ts_nop 1"
      : "+v"(x0), "+v"(y0), "+v"(x1), "+v"(y1), "+v"(x2), "+v"(y2), "+v"(x3), "+v"(y3));
}
template <bool X16> __device__ __forceinline__ floatx16 rows32(const floatx16& t) {
  floatx16 o = t;
  if constexpr (X16) {
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) {
      float x[4], y[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) { x[j] = o[8 * mi + j]; y[j] = o[8 * mi + 4 + j]; }
      swap16x4(x[0], y[0], x[1], y[1], x[2], y[2], x[3], y[3]);
#pragma unroll
      for (int j = 0; j < 4; ++j) { o[8 * mi + j] = x[j]; o[8 * mi + 4 + j] = y[j]; }
    }
  }
  return o;
}

__device__ __forceinline__ void dma_piece(const char* gb, unsigned voff, unsigned lds) {
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %0" ::"s"(gb), "s"(lds), "v"(voff) : "memory");
}

// One 32 x 64 block (accumulators t0 | t1 = columns 0-31 | 32-63) staged through the wave's LDS region and written as
// row-contiguous 16-byte accesses, 8 consecutive columns per lane: epilogue_readback's arithmetic in its order (gemm.hip), so
// the bits are gemm_nt_big_kernel's.  The residual rows ride a rolling window that runs ACROSS the six blocks of a wave tile
// (slot g % 2 of global iteration g = 4 block + it): a load is always issued before the stores it overlaps, so waiting
// for it never waits for a store (vmcnt retires in order and counts stores); the column biases come from LDS.
template <bool X16>
__device__ __forceinline__ void stage2(float* stage, const floatx16& a0, const floatx16& a1, int lane) {
  const int r = lane & 31, h = lane >> 5;
  const floatx16 t0 = rows32<X16>(a0), t1 = rows32<X16>(a1);
#pragma unroll
  for (int i = 0; i < 16; ++i) stage[acc_row<X16>(i, h) * 64 + r] = t0[i];
#pragma unroll
  for (int i = 0; i < 16; ++i) stage[acc_row<X16>(i, h) * 64 + 32 + r] = t1[i];
}

template <int EPI, int BLK, bool X16>
__device__ __forceinline__ void block(const GemmArgs& p, float* stage, const float* stage_bias, const floatx16& t0, const floatx16& t1,
                                      int row_w, int col_w, int lane_in, float* C32, _Float16* C16, const float* res, bool has_bias,
                                      floatx4 (&rs0)[2], floatx4 (&rs1)[2]) {
  constexpr int MB = BLK >> 1, NB = BLK & 1;
  int lane = lane_in;
  asm volatile("" : "+v"(lane));           // per-block addressing is recomputed here, not hoisted above the six blocks and spilled
  if constexpr (BLK >= 2) stage2<X16>(stage, t0, t1, lane);
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

template <int EPI, bool X16>
__device__ __forceinline__ void epilogue(const GemmArgs& p, float* stage, const floatx16 (&acc)[12], int row_w, int col_w, int lane,
                                         float* C32, _Float16* C16, const float* bias, const float* res) {
  // the four accumulators that live in VGPRs (row m = 0) go to the two staging areas first: 64 registers free for what follows
  float* const sa = stage;
  float* const sb = stage + 32 * 64;
  float* const sbias = stage + 2 * 32 * 64;
  stage2<X16>(sa, acc[0], acc[1], lane);
  stage2<X16>(sb, acc[2], acc[3], lane);
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
  block<EPI, 0, X16>(p, sa, sbias, acc[0], acc[1], row_w, col_w, lane, C32, C16, res, hb, rs0, rs1);
  block<EPI, 1, X16>(p, sb, sbias, acc[2], acc[3], row_w, col_w, lane, C32, C16, res, hb, rs0, rs1);
  block<EPI, 2, X16>(p, sa, sbias, acc[4], acc[5], row_w, col_w, lane, C32, C16, res, hb, rs0, rs1);
  block<EPI, 3, X16>(p, sb, sbias, acc[6], acc[7], row_w, col_w, lane, C32, C16, res, hb, rs0, rs1);
  block<EPI, 4, X16>(p, sa, sbias, acc[8], acc[9], row_w, col_w, lane, C32, C16, res, hb, rs0, rs1);
  block<EPI, 5, X16>(p, sb, sbias, acc[10], acc[11], row_w, col_w, lane, C32, C16, res, hb, rs0, rs1);
}

// ---- fast epilogue 1: f32 output (optional residual, optional ReLU / bias), NO LDS.  Register i of a 32 x 32 accumulator is
// one row per lane half and 32 consecutive columns across the lanes: a global_store_dword of it writes two full 128-byte
// row segments (MI355X_MICROARCH.md: "two 128-B segments in two rows: full rate"), the residual is read the same way.  The
// residual of tile t + 1 is requested before tile t's stores are issued (vmcnt counts stores and retires in order).
// Arithmetic per element as in epilogue_readback: ((acc + bias) + 0), ReLU, + residual.
#ifndef T384_RES_DEPTH
#define T384_RES_DEPTH 3      // half tiles between a residual load and its use (direct_f32)
#endif
#ifndef T384_RES_TWOPHASE
#define T384_RES_TWOPHASE 0   // the residual flavour's epilogue: 0 = interleaved loads and stores (product), 1 = two phases (round 6:
                              // measured and rejected, see twophase_res_f32)
#endif
typedef __attribute__((address_space(1))) float gfloat;
typedef __attribute__((address_space(1))) char gchar;

template <bool RES, bool FULL, bool X16>
__device__ __forceinline__ void direct_f32(const GemmArgs& p, const floatx16 (&acc)[12], int row_w, int col_w, int lane, float* C32,
                                           const float* bias, const float* res) {
  const int r = lane & 31, h = lane >> 5;
  float bv[4] = {0.f, 0.f, 0.f, 0.f};
  if (bias) {
#pragma unroll
    for (int n = 0; n < 4; ++n) bv[n] = bias[col_w + 32 * n + r];
  }
  const bool relu = p.act == 2;
  // every address = wave-uniform base of the half tile and row (SGPR pair) + ONE per-lane 32-bit offset (row 4 h, column r):
  // global_load/store_dword v, v_off, s[base]
  const unsigned oc = (unsigned)((HROWS<X16> * h) * p.ldc + r) * 4u, orr = (unsigned)((HROWS<X16> * h) * p.ldres + r) * 4u;
  const gchar* cb = (const gchar*)uniform_ptr((const char*)(C32 + (size_t)row_w * p.ldc + col_w));
  const gchar* rb = RES ? (const gchar*)uniform_ptr((const char*)(res + (size_t)row_w * p.ldres + col_w)) : nullptr;
  const int rows_left = p.M - row_w - HROWS<X16> * h;          // rows of this lane half that exist (ragged last row tile)
  // row (within the half tile of 16) of register k of a step, lane half 0: 32x32x16 layout (k & 3) + 8 (k >> 2); X16 (after
  // the permlane16 swap) (k & 3) + 4 (k >> 2)
  auto krow = [](int k) { return (k & 3) + (X16 ? 4 : 8) * (k >> 2); };
  // half tiles (registers 8 u .. 8 u + 7 of accumulator t): 24 steps; the residual of a step is requested T384_RES_DEPTH steps
  // before it is used, always before the stores of the step that issues it (waiting for a load also waits for every OLDER
  // store: vmcnt retires in order and counts stores).  Two steps ahead a step cost 1375 cycles = 33 k per tile
  // (profiles/r5_gemm_t384_epilogue_kind.txt); what bounds it is the bytes in flight, see phased_res_f32 below.  The window
  // is three steps deep (T384_RES_DEPTH): what the 128 VGPRs hold beside the four VGPR-resident accumulators without spill
  // reloads inside the store stream - deeper windows (4, 5, and 2 -> 5 once those accumulators are stored) were compiled and
  // all spilled INTO the stream, where every reload is one more wait for a store.  The residual flavour therefore stays
  // epilogue-bound (30 % of a tile, 810-925 TF/s against 1040-1140 for the f16 flavour: profiles/r5_gemm_t384_res_ab.txt);
  // it is 2-3 % faster than gemm_nt_big_kernel's, whose staged epilogue pays the same acknowledgements.
  constexpr int W = T384_RES_DEPTH + 1;
  float rv[W][8];
  auto load_res = [&](int s, float (&dst)[8]) {
    const int t = s >> 1, u = s & 1, m = t >> 2, n = t & 3;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int ro = 32 * m + 16 * u + krow(k);
      const gchar* sb = rb + ((size_t)ro * p.ldres + 32 * n) * 4;               // wave-uniform
      dst[k] = (FULL || ro < rows_left) ? *(const gfloat*)(sb + (size_t)orr) : 0.f;
    }
  };
  constexpr int D = T384_RES_DEPTH;
  if constexpr (RES) {
#pragma unroll
    for (int s = 0; s < D; ++s) load_res(s, rv[s]);
  }
#pragma unroll
  for (int s = 0; s < 24; ++s) {
    const int t = s >> 1, u = s & 1, m = t >> 2, n = t & 3;
    __builtin_amdgcn_sched_barrier(0);     // one half tile at a time: the scheduler must not pull all the read-outs forward
    if constexpr (RES) {
      if (s + D < 24) load_res(s + D, rv[(s + D) % W]);
    }
    float v[8], a8[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) a8[k] = acc[t][8 * u + k];
    if constexpr (X16) {                           // two 16 x 16 quads -> rows of 32 columns (see rows32)
      swap16x4(a8[0], a8[4], a8[1], a8[5], a8[2], a8[6], a8[3], a8[7]);
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) {                 // (no `+ 0.f`: an accumulator that starts at +0 is never -0; bv = 0 without a bias)
      float x = a8[k] + bv[n];
      x = relu ? fmaxf(x, 0.f) : x;
      if constexpr (RES) x += rv[s % W][k];
      v[k] = x;
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int ro = 32 * m + 16 * u + krow(k);
      gchar* sb = const_cast<gchar*>(cb) + ((size_t)ro * p.ldc + 32 * n) * 4;    // wave-uniform
#ifdef DVD_LAB
      if (p.debug & 0x800) {                       // lab (DVD_GEMM_T384_NOSTORE): the residual epilogue's loads alone (garbage output):
        if (v[k] == 123.456f) *(gfloat*)(sb + (size_t)oc) = v[k];      // the store exists, its condition never holds
      } else if (p.debug & 0x200) {                // lab (DVD_GEMM_T384_NT): streaming stores - are they acknowledged sooner?
        if (FULL || ro < rows_left) __builtin_nontemporal_store(v[k], (gfloat*)(sb + (size_t)oc));
      } else
#endif
      if (FULL || ro < rows_left) *(gfloat*)(sb + (size_t)oc) = v[k];
    }
  }
}

// ---- the residual flavour in TWO PHASES (round 6; MEASURED AND REJECTED - compiled only with -DT384_RES_TWOPHASE=1).
// gfx950 has ONE in-order vmcnt for loads and stores, so in the interleaved form above every wait for a residual value also
// waits for the acknowledgement of every older store of the wave; the kernel with its stores never taken is 0.3 ms per launch
// faster (profiles/r6_gemm_res_loads_alone.txt).  Here the accumulators themselves are the buffer that would break the coupling:
// phase A reads the residual (the same window) and writes relu(acc + bias) + residual BACK INTO THE ACCUMULATOR REGISTERS (one
// asm statement per half tile with the eight registers as in/out operands: as compiled C++ the update spilled ~50 registers
// into the K loop) - no store is issued, a wait stands behind loads only; phase B is the f32 flavour's store stream.  Same
// arithmetic per element: same bits (tested).  Measured (profiles/r6_gemm_res_twophase.txt, one epilogue per build): fc 1.97-1.98
// ms with a 3-step window, 1.90-1.93 with 2 steps (fewer spill reloads inside the store stream), against 1.90-1.92 interleaved - no
// gain.  So the 0.3 ms are not the load-behind-store waits either: taking the stores away removes 2 GB of writes per launch from a
// kernel whose every phase runs against the power cap and the fabric, which a re-ordering inside the epilogue does not.
template <bool FULL, bool X16>
__device__ __forceinline__ void twophase_res_f32(const GemmArgs& p, floatx16 (&acc)[12], int row_w, int col_w, int lane, float* C32,
                                                 const float* bias, const float* res) {
  const int r = lane & 31, h = lane >> 5;
  float bv[4] = {0.f, 0.f, 0.f, 0.f};
  if (bias) {
#pragma unroll
    for (int n = 0; n < 4; ++n) bv[n] = bias[col_w + 32 * n + r];
  }
  const bool relu = p.act == 2;
  const float lo = relu ? 0.f : -__builtin_inff();
  const unsigned oc = (unsigned)((HROWS<X16> * h) * p.ldc + r) * 4u, orr = (unsigned)((HROWS<X16> * h) * p.ldres + r) * 4u;
  const gchar* cb = (const gchar*)uniform_ptr((const char*)(C32 + (size_t)row_w * p.ldc + col_w));
  const gchar* rb = (const gchar*)uniform_ptr((const char*)(res + (size_t)row_w * p.ldres + col_w));
  const int rows_left = p.M - row_w - HROWS<X16> * h;
  auto krow = [](int k) { return (k & 3) + (X16 ? 4 : 8) * (k >> 2); };
  constexpr int D = T384_RES_DEPTH, W = D + 1;
  float rv[W][8];
  auto load_res = [&](int s, float (&dst)[8]) {
    const int t = s >> 1, u = s & 1, m = t >> 2, n = t & 3;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int ro = 32 * m + 16 * u + krow(k);
      const gchar* sb = rb + ((size_t)ro * p.ldres + 32 * n) * 4;               // wave-uniform
      dst[k] = (FULL || ro < rows_left) ? *(const gfloat*)(sb + (size_t)orr) : 0.f;
    }
  };
#pragma unroll
  for (int s = 0; s < D; ++s) load_res(s, rv[s]);
#pragma unroll
  for (int s = 0; s < 24; ++s) {                     // phase A: loads only
    const int t = s >> 1, u = s & 1, n = t & 3;
    __builtin_amdgcn_sched_barrier(0);
    if (s + D < 24) load_res(s + D, rv[(s + D) % W]);
    float a8[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) a8[k] = acc[t][8 * u + k];
    if constexpr (X16) {
      // ONE statement per half tile updates its eight accumulator registers IN PLACE (they are its in/out operands, in whichever
      // file they live: "+a" for the AGPR-resident rows m = 1, 2, "+v" for row m = 0): swap to rows of 32 columns, + bias,
      // max with lo (0 for ReLU, -inf otherwise: max(x, -inf) = x), + residual.  As compiled C++ the same update made the
      // compiler keep old and new values of the 192 accumulators alive side by side and spill ~50 registers into the K loop.
      const float* q = rv[s % W];
      if (t >= 4) {
        float t0, t1, t2, t3, t4, t5, t6, t7;
        asm volatile(
            "v_accvgpr_read_b32 %8, %0\n\tv_accvgpr_read_b32 %9, %1\n\tv_accvgpr_read_b32 %10, %2\n\tv_accvgpr_read_b32 %11, %3\n\t"
            "v_accvgpr_read_b32 %12, %4\n\tv_accvgpr_read_b32 %13, %5\n\tv_accvgpr_read_b32 %14, %6\n\tv_accvgpr_read_b32 %15, %7\n\t"
            "s_nop 1\n\t"
            "v_permlane16_swap_b32 %8, %12\n\tv_permlane16_swap_b32 %9, %13\n\tv_permlane16_swap_b32 %10, %14\n\tv_permlane16_swap_b32 %11, %15\n\t"
            "s_nop 1\n\t"
            "v_add_f32 %8, %8, %16\n\tv_add_f32 %9, %9, %16\n\tv_add_f32 %10, %10, %16\n\tv_add_f32 %11, %11, %16\n\t"
            "v_add_f32 %12, %12, %16\n\tv_add_f32 %13, %13, %16\n\tv_add_f32 %14, %14, %16\n\tv_add_f32 %15, %15, %16\n\t"
            "v_max_f32 %8, %8, %17\n\tv_max_f32 %9, %9, %17\n\tv_max_f32 %10, %10, %17\n\tv_max_f32 %11, %11, %17\n\t"
            "v_max_f32 %12, %12, %17\n\tv_max_f32 %13, %13, %17\n\tv_max_f32 %14, %14, %17\n\tv_max_f32 %15, %15, %17\n\t"
            "v_add_f32 %8, %8, %18\n\tv_add_f32 %9, %9, %19\n\tv_add_f32 %10, %10, %20\n\tv_add_f32 %11, %11, %21\n\t"
            "v_add_f32 %12, %12, %22\n\tv_add_f32 %13, %13, %23\n\tv_add_f32 %14, %14, %24\n\tv_add_f32 %15, %15, %25\n\t"
            "v_accvgpr_write_b32 %0, %8\n\tv_accvgpr_write_b32 %1, %9\n\tv_accvgpr_write_b32 %2, %10\n\tv_accvgpr_write_b32 %3, %11\n\t"
            "v_accvgpr_write_b32 %4, %12\n\tv_accvgpr_write_b32 %5, %13\n\tv_accvgpr_write_b32 %6, %14\n\tv_accvgpr_write_b32 %7, %15"
            : "+a"(a8[0]), "+a"(a8[1]), "+a"(a8[2]), "+a"(a8[3]), "+a"(a8[4]), "+a"(a8[5]), "+a"(a8[6]), "+a"(a8[7]),
              "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3), "=&v"(t4), "=&v"(t5), "=&v"(t6), "=&v"(t7)
            : "v"(bv[n]), "v"(lo), "v"(q[0]), "v"(q[1]), "v"(q[2]), "v"(q[3]), "v"(q[4]), "v"(q[5]), "v"(q[6]), "v"(q[7]));
      } else {
        asm volatile(
            "s_nop 1\n\t"
            "v_permlane16_swap_b32 %0, %4\n\tv_permlane16_swap_b32 %1, %5\n\tv_permlane16_swap_b32 %2, %6\n\tv_permlane16_swap_b32 %3, %7\n\t"
            "s_nop 1\n\t"
            "v_add_f32 %0, %0, %8\n\tv_add_f32 %1, %1, %8\n\tv_add_f32 %2, %2, %8\n\tv_add_f32 %3, %3, %8\n\t"
            "v_add_f32 %4, %4, %8\n\tv_add_f32 %5, %5, %8\n\tv_add_f32 %6, %6, %8\n\tv_add_f32 %7, %7, %8\n\t"
            "v_max_f32 %0, %0, %9\n\tv_max_f32 %1, %1, %9\n\tv_max_f32 %2, %2, %9\n\tv_max_f32 %3, %3, %9\n\t"
            "v_max_f32 %4, %4, %9\n\tv_max_f32 %5, %5, %9\n\tv_max_f32 %6, %6, %9\n\tv_max_f32 %7, %7, %9\n\t"
            "v_add_f32 %0, %0, %10\n\tv_add_f32 %1, %1, %11\n\tv_add_f32 %2, %2, %12\n\tv_add_f32 %3, %3, %13\n\t"
            "v_add_f32 %4, %4, %14\n\tv_add_f32 %5, %5, %15\n\tv_add_f32 %6, %6, %16\n\tv_add_f32 %7, %7, %17"
            : "+v"(a8[0]), "+v"(a8[1]), "+v"(a8[2]), "+v"(a8[3]), "+v"(a8[4]), "+v"(a8[5]), "+v"(a8[6]), "+v"(a8[7])
            : "v"(bv[n]), "v"(lo), "v"(q[0]), "v"(q[1]), "v"(q[2]), "v"(q[3]), "v"(q[4]), "v"(q[5]), "v"(q[6]), "v"(q[7]));
      }
#pragma unroll
      for (int k = 0; k < 8; ++k) acc[t][8 * u + k] = a8[k];
    } else {
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        float x = a8[k] + bv[n];
        x = relu ? fmaxf(x, 0.f) : x;
        acc[t][8 * u + k] = x + rv[s % W][k];
      }
    }
  }
#pragma unroll
  for (int s = 0; s < 24; ++s) {                     // phase B: stores only
    const int t = s >> 1, u = s & 1, m = t >> 2, n = t & 3;
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int ro = 32 * m + 16 * u + krow(k);
      gchar* sb = const_cast<gchar*>(cb) + ((size_t)ro * p.ldc + 32 * n) * 4;    // wave-uniform
      if (FULL || ro < rows_left) *(gfloat*)(sb + (size_t)oc) = acc[t][8 * u + k];
    }
  }
}

#ifdef DVD_LAB
// ---- the residual flavour, PHASED (f32 output = residual + relu(acc + bias), the decoder's fc / conv2): five phases of two or
// three accumulators.  A phase finishes its tiles from the residual rows requested one phase earlier, then requests the next
// phase's residual values and only THEN issues its own stores - so the loads a phase waits for are older than the
// stores issued beside them, and the only stores they are younger than were issued a whole phase (~2.5 k cycles: about one
// store acknowledgement) earlier.  The interleaved form above waits 24 times per tile, each time on stores three small steps
// old: 30-37 k cycles per tile; this one waits five times.
// MEASURED (profiles/r5_gemm_t384_res_phased.txt) and NOT adopted: 37.4 k cycles per tile against 37.6 k for the interleaved form, the
// same wall time - so the acknowledgement order is not what holds the residual epilogue.  With 6-12 KB of residual in flight per
// wave the epilogue reads 393 KB per CU at ~16 B/clk: the latency of an HBM read under every CU's mixed read / write burst times
// the bytes the free registers can keep in flight; a start-up stagger of the workgroups (which would spread the bursts) moves it
// by 2 % (`r5_gemm_t384_stagger_res.txt`).  LAB ONLY (DVD_GEMM_T384_RES_PHASED).
template <bool FULL>
__device__ __forceinline__ void phased_res_f32(const GemmArgs& p, const floatx16 (&acc)[12], int row_w, int col_w, int lane, float* C32,
                                               const float* bias, const float* res) {
  const int r = lane & 31, h = lane >> 5;
  float bv[4] = {0.f, 0.f, 0.f, 0.f};
  if (bias) {
#pragma unroll
    for (int n = 0; n < 4; ++n) bv[n] = bias[col_w + 32 * n + r];
  }
  const bool relu = p.act == 2;
  const unsigned oc = (unsigned)((4 * h) * p.ldc + r) * 4u, orr = (unsigned)((4 * h) * p.ldres + r) * 4u;
  const gchar* cb = (const gchar*)uniform_ptr((const char*)(C32 + (size_t)row_w * p.ldc + col_w));
  const gchar* rb = (const gchar*)uniform_ptr((const char*)(res + (size_t)row_w * p.ldres + col_w));
  const int rows_left = p.M - row_w - 4 * h;
  float rv[3][16];
  auto load_tile = [&](int t, float (&dst)[16]) {
    const int m = t >> 2, n = t & 3;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int ro = 32 * m + (i & 3) + 8 * (i >> 2);
      const gchar* sb = rb + ((size_t)ro * p.ldres + 32 * n) * 4;               // wave-uniform
      dst[i] = (FULL || ro < rows_left) ? *(const gfloat*)(sb + (size_t)orr) : 0.f;
    }
  };
  // phases of 2, 2, 3, 3, 2 accumulators: the first two are the VGPR-resident ones, finished in place, and are kept small so
  // that (accumulators still in VGPRs) + (results waiting for their stores) + (the next phase's residual) stays under ~100
  constexpr int PH[6] = {0, 2, 4, 7, 10, 12};
#pragma unroll
  for (int t = PH[0]; t < PH[1]; ++t) load_tile(t, rv[t - PH[0]]);
#pragma unroll
  for (int ph = 0; ph < 5; ++ph) {
    float out[3][16];
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int t = PH[ph]; t < PH[ph + 1]; ++t) {
      const int j = t - PH[ph], n = t & 3;
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        float x = acc[t][i] + bv[n];
        x = relu ? fmaxf(x, 0.f) : x;
        out[j][i] = x + rv[j][i];
      }
    }
    __builtin_amdgcn_sched_barrier(0);     // the next phase's residual is requested BEFORE this phase's stores
    if (ph < 4) {
#pragma unroll
      for (int t = PH[ph + 1]; t < PH[ph + 2]; ++t) load_tile(t, rv[t - PH[ph + 1]]);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int t = PH[ph]; t < PH[ph + 1]; ++t) {
      const int j = t - PH[ph], m = t >> 2, n = t & 3;
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int ro = 32 * m + (i & 3) + 8 * (i >> 2);
        gchar* sb = const_cast<gchar*>(cb) + ((size_t)ro * p.ldc + 32 * n) * 4;    // wave-uniform
        if (FULL || ro < rows_left) *(gfloat*)(sb + (size_t)oc) = out[j][i];
      }
    }
  }
}

#endif  // DVD_LAB

// ---- fast epilogue 2: f16 output only (bias, ReLU / GELU), NO LDS either.  Bias, activation and the f16 rounding happen in
// the accumulator layout (a lane = one column); neighbouring lanes then trade one value per register PAIR through a DPP
// quad permute, so that an even lane holds (row a: columns r, r + 1) and its odd neighbour (row a + 1: columns r - 1, r) as
// one packed dword each: a global_store_dword writes four 64-byte row segments, 8 stores per 32 x 32 accumulator.
template <bool FULL, bool X16>
__device__ __forceinline__ void packed_f16(const GemmArgs& p, const floatx16 (&acc)[12], int row_w, int col_w, int lane,
                                           _Float16* C16, const float* bias) {
  const int r = lane & 31, h = lane >> 5;
  const bool odd = r & 1;
  float bv[4] = {0.f, 0.f, 0.f, 0.f};
  if (bias) {
#pragma unroll
    for (int n = 0; n < 4; ++n) bv[n] = bias[col_w + 32 * n + r];
  }
  const int act = p.act;
  const unsigned sel = odd ? 0x03020706u : 0x05040100u;      // v_perm_b32 (S0 = neighbour's word, S1 = own): see the pack below
  // After the exchange an even lane 2 L holds (row a: columns 2 L, 2 L + 1) and the odd lane 2 L + 1 (row a + 1: the same columns).
  // Stored like that, neighbouring lanes alternate between two rows and the store unit sees 64 separate 4-byte accesses (15 cycles
  // per store instruction, measured: the f16 epilogue took 11.7 k cycles whatever its VALU work was).  One ds_bpermute_b32 (the
  // LDS crossbar, no LDS memory) sorts the words so that lanes 0-15 of a half hold row a and lanes 16-31 row a + 1: a
  // store is then four runs of 16 consecutive lanes = four contiguous 64-byte row segments.
  const int sub = r >> 4, lc = r & 15;                                   // after the sort: row a + sub, columns 2 lc, 2 lc + 1
  const int bp_src = (32 * h + 2 * lc + sub) * 4;                        // ds_bpermute address: the lane that holds them before it
  // per-lane byte offsets of the 8 packed registers of a tile (register pair (2 j, 2 j + 1) = rows a, a + 1): the same for all tiles
  unsigned oc[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) oc[j] = (unsigned)((acc_row<X16>(2 * j, h) + sub) * p.ldc16 + 2 * lc) * 2u;
  const gchar* cb = (const gchar*)uniform_ptr((const char*)(C16 + (size_t)row_w * p.ldc16 + col_w));
  const int rows_left = p.M - row_w;
#pragma unroll
  for (int t = 0; t < 12; ++t) {
    const int m = t >> 2, n = t & 3;
    __builtin_amdgcn_sched_barrier(0);     // one accumulator at a time: the scheduler must not pull all twelve read-outs forward
    // (the old kernel's `+ 0.f`s are dropped: an accumulator that starts at +0 is never -0, so they change no bit)
    float v[16];
    {
      const floatx16 a32 = rows32<X16>(acc[t]);
#pragma unroll
      for (int i = 0; i < 16; ++i) v[i] = a32[i] + bv[n];       // bv = 0 without a bias: x + 0 = x for every x that is not -0
    }
    if (act == 1) {
#pragma unroll
      for (int i = 0; i < 16; ++i) v[i] = gelu_tanh(v[i]);
    }
    if (act == 2) {
#pragma unroll
      for (int i = 0; i < 16; ++i) v[i] = fmaxf(v[i], 0.f);
    }
    gchar* sb = const_cast<gchar*>(cb) + ((size_t)(32 * m) * p.ldc16 + 32 * n) * 2;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      // pack the lane's two rows of its column, trade the packed word with the neighbour lane (one DPP quad permute), and
      // pick by a per-lane byte selector: even lane = (row a: own column | neighbour's), odd lane = (row a + 1: neighbour's | own)
      const half2v own = {(_Float16)v[2 * j], (_Float16)v[2 * j + 1]};
      const unsigned pw = __builtin_bit_cast(unsigned, own);
      const unsigned qw = (unsigned)__builtin_amdgcn_mov_dpp((int)pw, 0xB1, 0xF, 0xF, true);
      const unsigned pk = (unsigned)__builtin_amdgcn_ds_bpermute(bp_src, (int)__builtin_amdgcn_perm(qw, pw, sel));
      const int ro = 32 * m + acc_row<X16>(2 * j, h) + sub;
      if (FULL || ro < rows_left) *(__attribute__((address_space(1))) unsigned*)(sb + (size_t)oc[j]) = pk;
    }
  }
}
}  // namespace t384

// FL: the epilogue compiled into this instance (one per instance keeps the function small and its registers un-spilled):
//   0 f16 output only: packed_f16       1 f32 output only: direct_f32       2 f32 output only + residual: direct_f32
//   3 anything else without residual: the staged generic path       4 ... with residual
// FULL: M % 384 == 0, no row masks.  DBG (lab): 1 no LDS-DMA in the loop, 2 no fragment reads, 3 no barrier, 4 MFMAs only,
// 5 s_memtime stamps, 6 two 16x16x32 MFMAs per 32x32x16 (pricing, garbage math)
template <int DBG, int FL, bool FULL, bool X16>
__global__ void __launch_bounds__(512, 2) gemm_nt_t384_kernel(GemmArgs p) {
  using namespace t384;
  // XT: the ring never drains between the tiles of a workgroup (gen_gemm_t384.py, loop_stmt(xt=True)): the next tile's first
  // three half slabs land while this tile's epilogue runs.  Needs an epilogue that leaves LDS alone (FL 0-2) and per-lane DMA
  // offsets that do not depend on the tile (no clamped rows: FULL).
  constexpr bool XT = FULL && FL <= 2;
  extern __shared__ __attribute__((aligned(1024))) char smem[];   // 4 x [A half slab | B half slab]; (FL 3, 4: then the staging)
  typedef __attribute__((address_space(3))) void* lptr_t;
  const int nwg = p.ntm * p.ntn;
  const int z = blockIdx.y;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 1, wc = wave & 1;
  const unsigned lds0 = (unsigned)(uintptr_t)(lptr_t)smem;
  const int nloop = p.K / 128 - 2;
  const _Float16* A = (const _Float16*)p.A + z * p.sA;
  const _Float16* B = (const _Float16*)p.B + z * p.sB;
  const unsigned pda = lds0 + (3 * wave) * 1024, pdb = lds0 + BOFF + (2 * wave) * 1024;
  bool first = true;
#ifdef DVD_LAB
  // lab (DVD_GEMM_T384_PRIO): static priority for the second-dispatched half of the workgroup (MI355X_MICROARCH.md, two waves
  // per SIMD, item 4)
  if ((p.debug & 0x100) && wave >= 4) __builtin_amdgcn_s_setprio(1);
#endif
  // De-synchronise the chip: every CU runs the same tiles at the same pace, so all 256 epilogues would store at the same
  // moment - 50 MB bursts against the HBM write rate (the epilogue of the f16 flavour measured 11.4 k cycles = 5.7 TB/s
  // chip-wide, while the kernel's AVERAGE write rate is 0.7 TB/s).  A start-up delay of up to 15 quanta spreads the
  // workgroups of the first round over part of a tile period; without a barrier between tiles (XT) they stay spread.
  if (p.stagger > 0) {
    const int slots = (blockIdx.x * 5) & 15;
    for (int i = 0; i < slots * p.stagger; ++i) __builtin_amdgcn_s_sleep(16);   // 16 x 64 cycles each
  }
  for (int vid = blockIdx.x; vid < nwg; vid += gridDim.x) {
    int tm, tn;
    tile_coords(vid, p.ntm, p.ntn, tm, tn, p.walk);
    tm = __builtin_amdgcn_readfirstlane(tm); tn = __builtin_amdgcn_readfirstlane(tn);
    const int bm0 = tm * 384, bn0 = tn * 256;
#ifdef DVD_LAB
    unsigned long long t0 = 0, t1 = 0, t2 = 0;
    if constexpr (DBG == 5) t0 = __builtin_amdgcn_s_memtime();
#endif
    floatx16 acc[12];
    {
      // per-lane source offsets of the wave's 3 + 2 pieces of a half slab: piece = 16 rows x 64 B, LDS slot (row, pos = lane & 3)
      // <- chunk pos ^ ((row >> 2) & 3); rows clamped into the matrix (ragged last tiles; FULL: no row is ever clamped)
      // (everything per-lane is derived here from a laundered lane id: hoisted out of the tile loop it would stay live across
      //  the epilogue, which has no register to spare)
      int lane_l = lane;
      asm volatile("" : "+v"(lane_l));
#define lane lane_l
      // fragment read bases in slot 0: row r of the wave's first 32-row block, chunk (2 s + h) ^ ((r >> 2) & 3)
      // X16: fragment = 16 rows x 64 B, lane l reads chunk (l >> 4) of row (l & 15); image swizzle (-(row >> 2)) & 3 (gen_gemm_t384.py)
      const unsigned ch = X16 ? (unsigned)(((lane >> 4) ^ ((0u - (unsigned)((lane & 15) >> 2)) & 3u)) * 16)
                              : (unsigned)((((lane >> 5) ^ (((lane & 31) >> 2) & 3))) * 16);
      const int frow = X16 ? (lane & 15) : (lane & 31);
      const unsigned fa0 = lds0 + (96 * wr + frow) * 64 + ch, fa1 = fa0 ^ 32;
      const unsigned fb0 = lds0 + BOFF + (128 * wc + frow) * 64 + ch, fb1 = fb0 ^ 32;
      unsigned va[3], vb[2];
      const int pos = lane & 3;
      auto swz = [](int row) { return X16 ? (int)((0u - (unsigned)(row >> 2)) & 3u) : ((row >> 2) & 3); };
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        const int row = 16 * (3 * wave + i) + (lane >> 2);
        const int ra = FULL ? row : min(bm0 + row, p.M - 1) - bm0;
        va[i] = (unsigned)ra * (unsigned)(p.lda * 2) + (pos ^ swz(row)) * 16;
      }
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int row = 16 * (2 * wave + i) + (lane >> 2);
        vb[i] = (unsigned)row * (unsigned)(p.ldb * 2) + (pos ^ swz(row)) * 16;     // N % 256 == 0: never clamped
      }
      (void)fa1; (void)fb1;
      const char* Atile = uniform_ptr((const char*)(A + (size_t)bm0 * p.lda));
      const char* Btile = uniform_ptr((const char*)(B + (size_t)bn0 * p.ldb));
      if (!XT || first) {
        // prologue: half slabs 0, 1, 2 -> slots 0, 1, 2
#pragma unroll
        for (int j = 0; j < 3; ++j) {
#pragma unroll
          for (int i = 0; i < 3; ++i) dma_piece(Atile + j * 64, va[i], pda + j * HALF + i * 1024);
#pragma unroll
          for (int i = 0; i < 2; ++i) dma_piece(Btile + j * 64, vb[i], pdb + j * HALF + i * 1024);
        }
        asm volatile("s_waitcnt vmcnt(10)\n\ts_barrier" ::: "memory");      // half slab 0 has landed
        first = false;
      }
#ifdef DVD_LAB
      if constexpr (DBG == 5) t1 = __builtin_amdgcn_s_memtime();
#endif
#define T384_ARGS acc, Atile + 3 * 64, Btile + 3 * 64, nloop, pda, pdb, va[0], va[1], va[2], vb[0], vb[1], fa0, fa1, fb0, fb1
#define T384X_ARGS acc, Atile + 3 * 64, Btile + 3 * 64, nloop, pda, pdb, va[0], va[1], va[2], vb[0], vb[1], fa0, fb0
      if constexpr (X16 && XT) {
        const int vnext = vid + (int)gridDim.x < nwg ? vid + (int)gridDim.x : vid;
        int tm2, tn2;
        tile_coords(vnext, p.ntm, p.ntn, tm2, tn2, p.walk);
        tm2 = __builtin_amdgcn_readfirstlane(tm2); tn2 = __builtin_amdgcn_readfirstlane(tn2);
        const char* Anext = uniform_ptr((const char*)(A + (size_t)tm2 * 384 * p.lda));
        const char* Bnext = uniform_ptr((const char*)(B + (size_t)tn2 * 256 * p.ldb));
        t384x_loop_xt(T384X_ARGS, Anext, Bnext);
      } else if constexpr (X16) {
        t384x_loop(T384X_ARGS);
      } else if constexpr (XT) {
        // the next tile of this workgroup (the last one re-loads its own first half slabs: valid addresses, never read)
        const int vnext = vid + (int)gridDim.x < nwg ? vid + (int)gridDim.x : vid;
        int tm2, tn2;
        tile_coords(vnext, p.ntm, p.ntn, tm2, tn2, p.walk);
        tm2 = __builtin_amdgcn_readfirstlane(tm2); tn2 = __builtin_amdgcn_readfirstlane(tn2);
        const char* Anext = uniform_ptr((const char*)(A + (size_t)tm2 * 384 * p.lda));
        const char* Bnext = uniform_ptr((const char*)(B + (size_t)tn2 * 256 * p.ldb));
#ifdef DVD_LAB
        if constexpr (DBG == 1) t384_loop_xt_nodma(T384_ARGS, Anext, Bnext);
        else if constexpr (DBG == 2) t384_loop_xt_noread(T384_ARGS, Anext, Bnext);
        else if constexpr (DBG == 3) t384_loop_xt_nobar(T384_ARGS, Anext, Bnext);
        else if constexpr (DBG == 4) t384_loop_xt_mfmaonly(T384_ARGS, Anext, Bnext);
        else if constexpr (DBG == 6) t384_loop_xt_m16(T384_ARGS, Anext, Bnext);
        else
#endif
          t384_loop_xt(T384_ARGS, Anext, Bnext);
      } else {
#ifdef DVD_LAB
        if constexpr (DBG == 1) t384_loop_nodma(T384_ARGS);
        else if constexpr (DBG == 2) t384_loop_noread(T384_ARGS);
        else if constexpr (DBG == 3) t384_loop_nobar(T384_ARGS);
        else if constexpr (DBG == 4) t384_loop_mfmaonly(T384_ARGS);
        else if constexpr (DBG == 6) t384_loop_m16(T384_ARGS);
        else
#endif
          t384_loop(T384_ARGS);
      }
#undef T384_ARGS
#undef T384X_ARGS
#undef lane
    }
    if constexpr (!XT) {
      // every wave has read its last fragments: LDS becomes the epilogue's staging area
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    }
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
      if constexpr (FL == 0) packed_f16<FULL, X16>(p, acc, row_w, col_w, lane_e, C16, bias);
      else if constexpr (FL == 1) direct_f32<false, FULL, X16>(p, acc, row_w, col_w, lane_e, C32, bias, res);
      else if constexpr (FL == 2) {
#ifdef DVD_LAB
        if constexpr (!X16) {
          if (p.debug & 0x400) { phased_res_f32<FULL>(p, acc, row_w, col_w, lane_e, C32, bias, res); }    // lab: the phased form (measured equal)
          else direct_f32<true, FULL, X16>(p, acc, row_w, col_w, lane_e, C32, bias, res);
        } else
#endif
        {
          // ONE residual epilogue per kernel instance (two inlined side by side spill into each other and into the K loop:
          // measured).  T384_RES_TWOPHASE = 0 compiles round 5's interleaved form instead (benchmarks/lab/alt/ A/B builds).
          if constexpr (T384_RES_TWOPHASE != 0) twophase_res_f32<FULL, X16>(p, acc, row_w, col_w, lane_e, C32, bias, res);
          else direct_f32<true, FULL, X16>(p, acc, row_w, col_w, lane_e, C32, bias, res);
        }
      }
      else if constexpr (FL == 3) epilogue<0, X16>(p, stage, acc, row_w, col_w, lane_e, C32, C16, bias, res);
      else epilogue<1, X16>(p, stage, acc, row_w, col_w, lane_e, C32, C16, bias, res);
    }
#ifdef DVD_LAB
    if constexpr (DBG == 5) {
      const unsigned long long t3 = __builtin_amdgcn_s_memtime();
      if (lane == 0 && p.stamps && vid < 256 * 64) {
        unsigned long long* o = p.stamps + ((size_t)vid * 8 + wave) * 8;
        o[0] = t0; o[1] = t1; o[2] = t2; o[3] = t3; o[4] = t3; o[5] = t2; o[6] = t2; o[7] = t2;
      }
    }
#endif
    if constexpr (FL >= 3) __syncthreads();   // every wave has read its staging region back: the next tile's LDS-DMA may overwrite it
  }
  // no LDS-DMA may land after the workgroup has ended (XT: the last tile's look-ahead pieces are still in flight)
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

template <int DBG, bool X16>
static void launch_fl(const GemmArgs& p, int fl, bool full, dim3 grid, hipStream_t st) {
#define T384_GO(FL_, FULL_) gemm_nt_t384_kernel<DBG, FL_, FULL_, X16><<<grid, 512, t384::LDS_BYTES, st>>>(p)
  if (full) {
    switch (fl) { case 0: T384_GO(0, true); break; case 1: T384_GO(1, true); break; case 2: T384_GO(2, true); break;
                  case 3: T384_GO(3, true); break; default: T384_GO(4, true); }
  } else {
    switch (fl) { case 0: T384_GO(0, false); break; case 1: T384_GO(1, false); break; case 2: T384_GO(2, false); break;
                  case 3: T384_GO(3, false); break; default: T384_GO(4, false); }
  }
#undef T384_GO
}
template <int DBG, bool X16>
static void allow_lds() {
#define T384_AL(FL_, FULL_) (void)hipFuncSetAttribute((const void*)gemm_nt_t384_kernel<DBG, FL_, FULL_, X16>, hipFuncAttributeMaxDynamicSharedMemorySize, t384::LDS_BYTES)
  T384_AL(0, true); T384_AL(1, true); T384_AL(2, true); T384_AL(3, true); T384_AL(4, true);
  T384_AL(0, false); T384_AL(1, false); T384_AL(2, false); T384_AL(3, false); T384_AL(4, false);
#undef T384_AL
}

// T384_X16: the product's K loop.  1 = v_mfma_f32_16x16x32_f16 (round 6), 0 = round 5's 32x32x16 loop (bit-identical to
// gemm_nt_big_kernel).  The lab build carries both (DVD_GEMM_T384_M32=1 selects the old one) and the timing ablations of the old one.
#ifndef T384_X16
#define T384_X16 1
#endif
int launch_gemm_t384(const GemmArgs& p, int batch, int dbg, void* stream) {
  static DeviceOnce once_t;
  if (const auto bit = DeviceOnce::current_bit(); once_t.need(bit)) {
    allow_lds<0, T384_X16 != 0>();
#ifdef DVD_LAB
    allow_lds<0, T384_X16 == 0>();
    allow_lds<1, false>(); allow_lds<2, false>(); allow_lds<3, false>(); allow_lds<4, false>(); allow_lds<5, false>(); allow_lds<6, false>();
    allow_lds<5, true>();
#endif
    once_t.done(bit);
  }
  int nblk = p.ntm * p.ntn;
  if (nblk > 256) nblk = 256;
#ifdef DVD_LAB
  if (const char* e = getenv("DVD_GEMM_T384_NBLK")) nblk = atoi(e) < nblk ? atoi(e) : nblk;   // lab: fewer persistent workgroups
#endif
  const dim3 grid(nblk, batch);
  hipStream_t st = (hipStream_t)stream;
  // the epilogue flavour is a function of the descriptor (never of the data): see gemm_nt_t384_kernel
  const int fl = (p.C16 && !p.C32 && !p.res) ? 0 : (p.C32 && !p.C16 && p.act != 1) ? (p.res ? 2 : 1) : (p.res ? 4 : 3);
  const bool full = p.M % 384 == 0;
#ifdef DVD_LAB
  const bool use_x16 = getenv("DVD_GEMM_T384_M32") ? false : (getenv("DVD_GEMM_T384_X16") ? true : (T384_X16 != 0));
  if (use_x16) {
    if (dbg == 5) launch_fl<5, true>(p, fl, full, grid, st);
    else launch_fl<0, true>(p, fl, full, grid, st);
    return check_launch("gemm_nt(t384 x16)");
  }
  switch (dbg) {
    case 1: launch_fl<1, false>(p, fl, full, grid, st); break;
    case 2: launch_fl<2, false>(p, fl, full, grid, st); break;
    case 3: launch_fl<3, false>(p, fl, full, grid, st); break;
    case 4: launch_fl<4, false>(p, fl, full, grid, st); break;
    case 5: launch_fl<5, false>(p, fl, full, grid, st); break;
    case 6: launch_fl<6, false>(p, fl, full, grid, st); break;      // round 6: the 16x16x32 PRICING ablation (garbage math)
    default: launch_fl<0, false>(p, fl, full, grid, st);
  }
#else
  (void)dbg;
  launch_fl<0, T384_X16 != 0>(p, fl, full, grid, st);
#endif
  return check_launch("gemm_nt(t384)");
}

}  // namespace dvd
