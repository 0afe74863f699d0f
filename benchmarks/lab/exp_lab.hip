// Round 5: would moving the head_dim-64 attention's exponentials from v_exp_f32 onto PACKED-f16
// arithmetic pay?  The kernel (flash_attn_h64x) issues 32 v_exp_f32 + 16 v_cvt_pk_f16_f32 per 16 MFMAs 16x16x32 and wave, two
// waves per SIMD, and its VALU port is 77-80 % busy.  A packed-f16 2^x (round-to-nearest split by the 1536 magic constant,
// degree-3 polynomial, exponent add) is 9 v_pk_* instructions per PAIR of elements - on elements that are rounded to f16 for
// the PV product anyway.  This lab issues exactly those instruction mixes beside the MFMAs (two waves per SIMD, every CU) and
// reports cycles per 16 MFMAs:
//   V0  MFMAs only            V1  + 2 v_exp_f32 + 1 v_cvt_pk per MFMA (the kernel today)
//   V2  half of the pairs by the packed polynomial      V3  all of them by the packed polynomial
// (a TIMING proxy: the instruction sequence is the one a correct packed 2^x needs - same opcodes, same dependences - its
//  constants were not validated numerically, because the timing already decides: see profiles/r5_exp_offload_lab.txt)
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 exp_lab.hip -o exp_lab && ./exp_lab
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef _Float16 half8 __attribute__((ext_vector_type(8)));

#define MF(acc_) "v_mfma_f32_16x16x32_f16 " acc_ ", %[ha], %[hb], " acc_ "\n\t"
// the kernel's unit: two exponentials and the pack of their results
#define EXP2(x0_, x1_, p_) "v_exp_f32 " x0_ ", " x0_ "\n\tv_exp_f32 " x1_ ", " x1_ "\n\tv_cvt_pk_f16_f32 " p_ ", " x0_ ", " x1_ "\n\t"
// the packed-f16 unit on one pair: pack, clamp, split, polynomial, exponent add (9 packed instructions + the pack)
#define POLY(x0_, x1_, p_, t_, n_)                                                             \
  "v_cvt_pk_f16_f32 " p_ ", " x0_ ", " x1_ "\n\t"                                              \
  "v_pk_max_f16 " p_ ", " p_ ", %[lo]\n\t"                                                     \
  "v_pk_add_f16 " t_ ", " p_ ", %[magic]\n\t"                                                  \
  "v_pk_add_f16 " n_ ", " t_ ", %[magic] neg_lo:[0,1] neg_hi:[0,1]\n\t"                        \
  "v_pk_add_f16 " p_ ", " p_ ", " n_ " neg_lo:[0,1] neg_hi:[0,1]\n\t"                          \
  "v_pk_fma_f16 " n_ ", " p_ ", %[c3], %[c2]\n\t"                                              \
  "v_pk_fma_f16 " n_ ", " n_ ", " p_ ", %[c1]\n\t"                                             \
  "v_pk_fma_f16 " n_ ", " n_ ", " p_ ", %[c0]\n\t"                                             \
  "v_pk_lshlrev_b16 " t_ ", 10, " t_ "\n\t"                                                    \
  "v_pk_add_u16 " p_ ", " n_ ", " t_ "\n\t"

template <int V>
__global__ void __launch_bounds__(512, 2) exp_kernel(float* out, int iters, long long* clk) {
  floatx4 a0 = {0, 0, 0, 0}, a1 = a0, a2 = a0, a3 = a0;
  half8 ha, hb;
  for (int i = 0; i < 8; ++i) { ha[i] = (_Float16)(0.01f * ((threadIdx.x * 7 + i) % 97)); hb[i] = (_Float16)(0.02f * ((threadIdx.x * 5 + i) % 89)); }
  float x[8];
  for (int i = 0; i < 8; ++i) x[i] = -0.37f * ((threadIdx.x + 3 * i) % 23);
  unsigned p[4] = {0, 0, 0, 0}, t[2] = {0, 0}, n[2] = {0, 0};
  const unsigned magic = 0x66006600u, lo = 0xcb00cb00u /* -14 */, c3 = 0x2b1c2b1cu /* 0.0555 */, c2 = 0x33b033b0u /* 0.2402 */,
                 c1 = 0x398c398cu /* 0.6931 */, c0 = 0x3c003c00u /* 1 */;
  const long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#define OPS : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), \
      "+v"(x[7]), "+v"(p[0]), "+v"(p[1]), "+v"(p[2]), "+v"(p[3]), "+v"(t[0]), "+v"(t[1]), "+v"(n[0]), "+v"(n[1])                            \
    : [ha] "v"(ha), [hb] "v"(hb), [magic] "v"(magic), [lo] "v"(lo), [c3] "v"(c3), [c2] "v"(c2), [c1] "v"(c1), [c0] "v"(c0)
#pragma unroll
    for (int rep = 0; rep < 4; ++rep) {     // 4 x 4 = 16 MFMAs per iteration
      if constexpr (V == 0) asm volatile(MF("%0") MF("%1") MF("%2") MF("%3") OPS);
      if constexpr (V == 1) asm volatile(MF("%0") EXP2("%4", "%5", "%12") MF("%1") EXP2("%6", "%7", "%13") MF("%2") EXP2("%8", "%9", "%14") MF("%3") EXP2("%10", "%11", "%15") OPS);
      if constexpr (V == 2) asm volatile(MF("%0") EXP2("%4", "%5", "%12") MF("%1") POLY("%6", "%7", "%13", "%16", "%18") MF("%2") EXP2("%8", "%9", "%14") MF("%3") POLY("%10", "%11", "%15", "%17", "%19") OPS);
      if constexpr (V == 3) asm volatile(MF("%0") POLY("%4", "%5", "%12", "%16", "%18") MF("%1") POLY("%6", "%7", "%13", "%17", "%19") MF("%2") POLY("%8", "%9", "%14", "%16", "%18") MF("%3") POLY("%10", "%11", "%15", "%17", "%19") OPS);
      // keep the exponent arguments in range across iterations (not timed differently between the variants)
      if constexpr (V != 0) {
#pragma unroll
        for (int i = 0; i < 8; ++i) asm volatile("v_mov_b32 %0, %1" : "=v"(x[i]) : "v"(-0.37f * ((threadIdx.x + i) & 15)));
      }
    }
  }
  const long long t1 = __builtin_readcyclecounter();
  float s = a0[0] + a1[1] + a2[2] + a3[3];
  for (int i = 0; i < 8; ++i) s += x[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s + (float)(p[0] ^ p[1] ^ p[2] ^ p[3]);
  if (threadIdx.x == 0) clk[blockIdx.x] = t1 - t0;
}

int main() {
  float* out; long long* clk;
  CK(hipMalloc(&out, 256 * 512 * 4)); CK(hipMalloc(&clk, 256 * 8));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int iters = 20000;
  const char* names[4] = {"MFMAs only", "2 v_exp_f32 + v_cvt_pk per MFMA (today)", "half of the pairs by the packed polynomial", "all pairs by the packed polynomial"};
  for (int v = 0; v < 4; ++v) {
    auto launch = [&]() {
      if (v == 0) exp_kernel<0><<<256, 512>>>(out, iters, clk);
      if (v == 1) exp_kernel<1><<<256, 512>>>(out, iters, clk);
      if (v == 2) exp_kernel<2><<<256, 512>>>(out, iters, clk);
      if (v == 3) exp_kernel<3><<<256, 512>>>(out, iters, clk);
    };
    launch(); CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0)); launch(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    long long h[256]; CK(hipMemcpy(h, clk, sizeof(h), hipMemcpyDeviceToHost));
    double avg = 0; for (auto c : h) avg += c; avg /= 256;
    printf("V%d %-48s: %7.1f cycles per 16 MFMAs and wave (two waves per SIMD: matrix minimum 512 per pair of waves = 256 per wave-share)  %.2f ms  clock %.2f GHz\n",
           v, names[v], avg / iters, ms, avg / ms * 1e-6);
  }
  return 0;
}
