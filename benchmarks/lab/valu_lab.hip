// Issue cost of the softmax's VALU instructions on gfx950, one wave per SIMD (the attention kernels' regime): does a
// transcendental block the VALU for its 16 cycles or overlap with ordinary VALU work; what do the packed f32 forms cost?
// Every variant is a loop of independent instructions on distinct registers (no dependency stalls), timed with s_memtime.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 valu_lab.hip -o valu_lab
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

#define EXP8  "v_exp_f32_e32 %0, %0\n v_exp_f32_e32 %1, %1\n v_exp_f32_e32 %2, %2\n v_exp_f32_e32 %3, %3\n v_exp_f32_e32 %4, %4\n v_exp_f32_e32 %5, %5\n v_exp_f32_e32 %6, %6\n v_exp_f32_e32 %7, %7\n"
#define EXPH8 "v_exp_f16_e32 %0, %0\n v_exp_f16_e32 %1, %1\n v_exp_f16_e32 %2, %2\n v_exp_f16_e32 %3, %3\n v_exp_f16_e32 %4, %4\n v_exp_f16_e32 %5, %5\n v_exp_f16_e32 %6, %6\n v_exp_f16_e32 %7, %7\n"
#define FMA8  "v_fma_f32 %8, %8, %16, %16\n v_fma_f32 %9, %9, %16, %16\n v_fma_f32 %10, %10, %16, %16\n v_fma_f32 %11, %11, %16, %16\n v_fma_f32 %12, %12, %16, %16\n v_fma_f32 %13, %13, %16, %16\n v_fma_f32 %14, %14, %16, %16\n v_fma_f32 %15, %15, %16, %16\n"
#define ADD8  "v_add_f32_e32 %8, %8, %16\n v_add_f32_e32 %9, %9, %16\n v_add_f32_e32 %10, %10, %16\n v_add_f32_e32 %11, %11, %16\n v_add_f32_e32 %12, %12, %16\n v_add_f32_e32 %13, %13, %16\n v_add_f32_e32 %14, %14, %16\n v_add_f32_e32 %15, %15, %16\n"
#define CVT8  "v_cvt_pk_f16_f32 %8, %8, %16\n v_cvt_pk_f16_f32 %9, %9, %16\n v_cvt_pk_f16_f32 %10, %10, %16\n v_cvt_pk_f16_f32 %11, %11, %16\n v_cvt_pk_f16_f32 %12, %12, %16\n v_cvt_pk_f16_f32 %13, %13, %16\n v_cvt_pk_f16_f32 %14, %14, %16\n v_cvt_pk_f16_f32 %15, %15, %16\n"
#define MAX8  "v_max3_f32 %8, %8, %16, %9\n v_max3_f32 %9, %9, %16, %10\n v_max3_f32 %10, %10, %16, %11\n v_max3_f32 %11, %11, %16, %12\n v_max3_f32 %12, %12, %16, %13\n v_max3_f32 %13, %13, %16, %14\n v_max3_f32 %14, %14, %16, %15\n v_max3_f32 %15, %15, %16, %8\n"
// interleaved: one exp, one fma, ... (so an overlap, if the hardware has one, can happen)
#define MIX8  "v_exp_f32_e32 %0, %0\n v_fma_f32 %8, %8, %16, %16\n v_exp_f32_e32 %1, %1\n v_fma_f32 %9, %9, %16, %16\n v_exp_f32_e32 %2, %2\n v_fma_f32 %10, %10, %16, %16\n v_exp_f32_e32 %3, %3\n v_fma_f32 %11, %11, %16, %16\n v_exp_f32_e32 %4, %4\n v_fma_f32 %12, %12, %16, %16\n v_exp_f32_e32 %5, %5\n v_fma_f32 %13, %13, %16, %16\n v_exp_f32_e32 %6, %6\n v_fma_f32 %14, %14, %16, %16\n v_exp_f32_e32 %7, %7\n v_fma_f32 %15, %15, %16, %16\n"
#define MIX3  "v_exp_f32_e32 %0, %0\n v_fma_f32 %8, %8, %16, %16\n v_fma_f32 %9, %9, %16, %16\n v_fma_f32 %10, %10, %16, %16\n v_exp_f32_e32 %1, %1\n v_fma_f32 %11, %11, %16, %16\n v_fma_f32 %12, %12, %16, %16\n v_fma_f32 %13, %13, %16, %16\n"

typedef float float2v __attribute__((ext_vector_type(2)));

template <int V>
__global__ void __launch_bounds__(512, 1) valu_kernel(float* out, int iters, long long* clk) {
  float e[8], f[8];
  for (int i = 0; i < 8; ++i) { e[i] = -0.001f * (threadIdx.x + i); f[i] = 0.5f + 0.001f * i; }
  float c = 0.999f;
  unsigned sa = 1, sb = 2;
  float2v p[8], q = {0.999f, 0.998f};
  for (int i = 0; i < 8; ++i) p[i] = float2v{0.5f + 0.01f * i, 0.25f};
  __shared__ float4 lds[1024];
  typedef float floatx16 __attribute__((ext_vector_type(16)));
  typedef _Float16 half8 __attribute__((ext_vector_type(8)));
  floatx16 acc0 = {0}, acc1 = {0};
  half8 ha = {1, 2, 3, 4, 5, 6, 7, 8}, hb = {1, 1, 1, 1, 2, 2, 2, 2};
  typedef float floatx4 __attribute__((ext_vector_type(4)));
  floatx4 r0 = {0}, r1 = {0}, r2 = {0}, r3 = {0};
  lds[threadIdx.x] = float4{1.f, 2.f, 3.f, 4.f};
  lds[threadIdx.x + 256] = float4{1.f, 2.f, 3.f, 4.f};
  __syncthreads();
  const unsigned la = (unsigned)(uintptr_t)(__attribute__((address_space(3))) void*)&lds[threadIdx.x & 63];
  const long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
   for (int rep = 0; rep < 8; ++rep) {
#define OPS "+v"(e[0]), "+v"(e[1]), "+v"(e[2]), "+v"(e[3]), "+v"(e[4]), "+v"(e[5]), "+v"(e[6]), "+v"(e[7]), "+v"(f[0]), "+v"(f[1]), "+v"(f[2]), "+v"(f[3]), "+v"(f[4]), "+v"(f[5]), "+v"(f[6]), "+v"(f[7]) : "v"(c)
    if constexpr (V == 0) asm volatile(EXP8 : OPS);
    if constexpr (V == 1) asm volatile(FMA8 : OPS);
    if constexpr (V == 2) asm volatile(EXP8 FMA8 : OPS);
    if constexpr (V == 3) asm volatile(MIX8 : OPS);
    if constexpr (V == 4) asm volatile(MIX3 : OPS);
    if constexpr (V == 5) asm volatile(EXPH8 : OPS);
    if constexpr (V == 6) asm volatile(ADD8 : OPS);
    if constexpr (V == 7) asm volatile(CVT8 : OPS);
    if constexpr (V == 8) asm volatile(MAX8 : OPS);
    if constexpr (V == 9)
      asm volatile("v_pk_fma_f32 %0, %0, %8, %8\n v_pk_fma_f32 %1, %1, %8, %8\n v_pk_fma_f32 %2, %2, %8, %8\n v_pk_fma_f32 %3, %3, %8, %8\n"
                   "v_pk_fma_f32 %4, %4, %8, %8\n v_pk_fma_f32 %5, %5, %8, %8\n v_pk_fma_f32 %6, %6, %8, %8\n v_pk_fma_f32 %7, %7, %8, %8\n"
                   : "+v"(p[0]), "+v"(p[1]), "+v"(p[2]), "+v"(p[3]), "+v"(p[4]), "+v"(p[5]), "+v"(p[6]), "+v"(p[7]) : "v"(q));
    if constexpr (V == 10)
      asm volatile("v_pk_add_f32 %0, %0, %8\n v_pk_add_f32 %1, %1, %8\n v_pk_add_f32 %2, %2, %8\n v_pk_add_f32 %3, %3, %8\n"
                   "v_pk_add_f32 %4, %4, %8\n v_pk_add_f32 %5, %5, %8\n v_pk_add_f32 %6, %6, %8\n v_pk_add_f32 %7, %7, %8\n"
                   : "+v"(p[0]), "+v"(p[1]), "+v"(p[2]), "+v"(p[3]), "+v"(p[4]), "+v"(p[5]), "+v"(p[6]), "+v"(p[7]) : "v"(q));
    if constexpr (V == 11)
      asm volatile("v_pk_mul_f32 %0, %0, %8\n v_pk_mul_f32 %1, %1, %8\n v_pk_mul_f32 %2, %2, %8\n v_pk_mul_f32 %3, %3, %8\n"
                   "v_pk_mul_f32 %4, %4, %8\n v_pk_mul_f32 %5, %5, %8\n v_pk_mul_f32 %6, %6, %8\n v_pk_mul_f32 %7, %7, %8\n"
                   : "+v"(p[0]), "+v"(p[1]), "+v"(p[2]), "+v"(p[3]), "+v"(p[4]), "+v"(p[5]), "+v"(p[6]), "+v"(p[7]) : "v"(q));
    if constexpr (V == 12)
      asm volatile("v_dot2c_f32_f16 %0, %8, %9\n v_dot2c_f32_f16 %1, %8, %9\n v_dot2c_f32_f16 %2, %8, %9\n v_dot2c_f32_f16 %3, %8, %9\n"
                   "v_dot2c_f32_f16 %4, %8, %9\n v_dot2c_f32_f16 %5, %8, %9\n v_dot2c_f32_f16 %6, %8, %9\n v_dot2c_f32_f16 %7, %8, %9\n"
                   : "+v"(e[0]), "+v"(e[1]), "+v"(e[2]), "+v"(e[3]), "+v"(e[4]), "+v"(e[5]), "+v"(e[6]), "+v"(e[7]) : "v"(f[0]), "v"(f[1]));
    if constexpr (V == 13)      // 4 ds_read_b128 issued, then waited for together
      asm volatile("ds_read_b128 %0, %4\n ds_read_b128 %1, %4 offset:1024\n ds_read_b128 %2, %4 offset:2048\n ds_read_b128 %3, %4 offset:3072\n s_waitcnt lgkmcnt(0)"
                   : "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(r3) : "v"(la) : "memory");
    if constexpr (V == 14)      // per MFMA: 3 independent VALU (does the MFMA's issue + 3 VALU fit its 32 cycles?)
      asm volatile("v_mfma_f32_32x32x16_f16 %0, %2, %3, %0\n v_fma_f32 %4, %4, %10, %10\n v_fma_f32 %5, %5, %10, %10\n v_fma_f32 %6, %6, %10, %10\n"
                   "v_mfma_f32_32x32x16_f16 %1, %2, %3, %1\n v_fma_f32 %7, %7, %10, %10\n v_fma_f32 %8, %8, %10, %10\n v_fma_f32 %9, %9, %10, %10\n"
                   : "+v"(acc0), "+v"(acc1) : "v"(ha), "v"(hb), "v"(f[0]), "v"(f[1]), "v"(f[2]), "v"(f[3]), "v"(f[4]), "v"(f[5]), "v"(c));
    if constexpr (V == 15)      // per MFMA: 4 independent VALU
      asm volatile("v_mfma_f32_32x32x16_f16 %0, %2, %3, %0\n v_fma_f32 %4, %4, %10, %10\n v_fma_f32 %5, %5, %10, %10\n v_fma_f32 %6, %6, %10, %10\n v_fma_f32 %7, %7, %10, %10\n"
                   "v_mfma_f32_32x32x16_f16 %1, %2, %3, %1\n v_fma_f32 %8, %8, %10, %10\n v_fma_f32 %9, %9, %10, %10\n v_fma_f32 %4, %4, %10, %10\n v_fma_f32 %5, %5, %10, %10\n"
                   : "+v"(acc0), "+v"(acc1) : "v"(ha), "v"(hb), "v"(f[0]), "v"(f[1]), "v"(f[2]), "v"(f[3]), "v"(f[4]), "v"(f[5]), "v"(c));
    if constexpr (V == 16)      // per MFMA: exp + 2 VALU + ds_read
      asm volatile("v_mfma_f32_32x32x16_f16 %0, %2, %3, %0\n v_exp_f32_e32 %4, %4\n v_fma_f32 %5, %5, %10, %10\n v_fma_f32 %6, %6, %10, %10\n ds_read_b128 %11, %13\n"
                   "v_mfma_f32_32x32x16_f16 %1, %2, %3, %1\n v_exp_f32_e32 %7, %7\n v_fma_f32 %8, %8, %10, %10\n v_fma_f32 %9, %9, %10, %10\n ds_read_b128 %12, %13 offset:1024\n s_waitcnt lgkmcnt(1)\n"
                   : "+v"(acc0), "+v"(acc1) : "v"(ha), "v"(hb), "v"(f[0]), "v"(f[1]), "v"(f[2]), "v"(f[3]), "v"(f[4]), "v"(f[5]), "v"(c), "v"(r0), "v"(r1), "v"(la) : "memory");
    if constexpr (V == 17)      // 16x16x32: per MFMA (16 cycles) one VALU
      asm volatile("v_mfma_f32_16x16x32_f16 %0, %2, %3, %0\n v_fma_f32 %4, %4, %8, %8\n v_mfma_f32_16x16x32_f16 %1, %2, %3, %1\n v_fma_f32 %5, %5, %8, %8\n"
                   "v_mfma_f32_16x16x32_f16 %0, %2, %3, %0\n v_fma_f32 %6, %6, %8, %8\n v_mfma_f32_16x16x32_f16 %1, %2, %3, %1\n v_fma_f32 %7, %7, %8, %8\n"
                   : "+v"(r0), "+v"(r1) : "v"(ha), "v"(hb), "v"(f[0]), "v"(f[1]), "v"(f[2]), "v"(f[3]), "v"(c));
    if constexpr (V == 18)      // 16x16x32 bare, 4 per iteration
      asm volatile("v_mfma_f32_16x16x32_f16 %0, %2, %3, %0\n v_mfma_f32_16x16x32_f16 %1, %2, %3, %1\n v_mfma_f32_16x16x32_f16 %0, %2, %3, %0\n v_mfma_f32_16x16x32_f16 %1, %2, %3, %1\n"
                   : "+v"(r0), "+v"(r1) : "v"(ha), "v"(hb));
    if constexpr (V == 20)      // AGPR accumulators, 3 VALU per MFMA
      asm volatile("v_mfma_f32_32x32x16_f16 %0, %2, %3, %0\n v_fma_f32 %4, %4, %10, %10\n v_fma_f32 %5, %5, %10, %10\n v_fma_f32 %6, %6, %10, %10\n"
                   "v_mfma_f32_32x32x16_f16 %1, %2, %3, %1\n v_fma_f32 %7, %7, %10, %10\n v_fma_f32 %8, %8, %10, %10\n v_fma_f32 %9, %9, %10, %10\n"
                   : "+a"(acc0), "+a"(acc1) : "v"(ha), "v"(hb), "v"(f[0]), "v"(f[1]), "v"(f[2]), "v"(f[3]), "v"(f[4]), "v"(f[5]), "v"(c));
    if constexpr (V == 21)      // VGPR accumulators, 1 VALU per MFMA
      asm volatile("v_mfma_f32_32x32x16_f16 %0, %2, %3, %0\n v_fma_f32 %4, %4, %10, %10\n"
                   "v_mfma_f32_32x32x16_f16 %1, %2, %3, %1\n v_fma_f32 %7, %7, %10, %10\n"
                   : "+v"(acc0), "+v"(acc1) : "v"(ha), "v"(hb), "v"(f[0]), "v"(f[1]), "v"(f[2]), "v"(f[3]), "v"(f[4]), "v"(f[5]), "v"(c));
    if constexpr (V == 22)      // VGPR accumulators, 2 VALU per MFMA
      asm volatile("v_mfma_f32_32x32x16_f16 %0, %2, %3, %0\n v_fma_f32 %4, %4, %10, %10\n v_fma_f32 %5, %5, %10, %10\n"
                   "v_mfma_f32_32x32x16_f16 %1, %2, %3, %1\n v_fma_f32 %7, %7, %10, %10\n v_fma_f32 %8, %8, %10, %10\n"
                   : "+v"(acc0), "+v"(acc1) : "v"(ha), "v"(hb), "v"(f[0]), "v"(f[1]), "v"(f[2]), "v"(f[3]), "v"(f[4]), "v"(f[5]), "v"(c));
    if constexpr (V == 23)      // AGPR accumulators, 2 VALU per MFMA
      asm volatile("v_mfma_f32_32x32x16_f16 %0, %2, %3, %0\n v_fma_f32 %4, %4, %10, %10\n v_fma_f32 %5, %5, %10, %10\n"
                   "v_mfma_f32_32x32x16_f16 %1, %2, %3, %1\n v_fma_f32 %7, %7, %10, %10\n v_fma_f32 %8, %8, %10, %10\n"
                   : "+a"(acc0), "+a"(acc1) : "v"(ha), "v"(hb), "v"(f[0]), "v"(f[1]), "v"(f[2]), "v"(f[3]), "v"(f[4]), "v"(f[5]), "v"(c));
    if constexpr (V == 24)      // AGPR accumulators, 1 VALU per MFMA
      asm volatile("v_mfma_f32_32x32x16_f16 %0, %2, %3, %0\n v_fma_f32 %4, %4, %10, %10\n"
                   "v_mfma_f32_32x32x16_f16 %1, %2, %3, %1\n v_fma_f32 %7, %7, %10, %10\n"
                   : "+a"(acc0), "+a"(acc1) : "v"(ha), "v"(hb), "v"(f[0]), "v"(f[1]), "v"(f[2]), "v"(f[3]), "v"(f[4]), "v"(f[5]), "v"(c));
    if constexpr (V == 25)      // AGPR accumulators, exp + fma + add-like per MFMA pair = the attention step
      asm volatile("v_mfma_f32_32x32x16_f16 %0, %2, %3, %0\n v_fma_f32 %4, %4, %10, %10\n v_exp_f32_e32 %5, %5\n v_add_f32_e32 %6, %6, %10\n"
                   "v_mfma_f32_32x32x16_f16 %1, %2, %3, %1\n"
                   : "+a"(acc0), "+a"(acc1) : "v"(ha), "v"(hb), "v"(f[0]), "v"(f[1]), "v"(f[2]), "v"(f[3]), "v"(f[4]), "v"(f[5]), "v"(c));
    if constexpr (V == 26)      // same with VGPR accumulators
      asm volatile("v_mfma_f32_32x32x16_f16 %0, %2, %3, %0\n v_fma_f32 %4, %4, %10, %10\n v_exp_f32_e32 %5, %5\n v_add_f32_e32 %6, %6, %10\n"
                   "v_mfma_f32_32x32x16_f16 %1, %2, %3, %1\n"
                   : "+v"(acc0), "+v"(acc1) : "v"(ha), "v"(hb), "v"(f[0]), "v"(f[1]), "v"(f[2]), "v"(f[3]), "v"(f[4]), "v"(f[5]), "v"(c));
    if constexpr (V == 27)      // the head_dim-64 density: per MFMA 6 v_fma + 2 v_exp (AGPR accumulators)
      asm volatile("v_mfma_f32_32x32x16_f16 %0, %2, %3, %0\n v_fma_f32 %4, %4, %10, %10\n v_fma_f32 %5, %5, %10, %10\n v_exp_f32_e32 %6, %6\n v_fma_f32 %7, %7, %10, %10\n"
                   "v_fma_f32 %8, %8, %10, %10\n v_exp_f32_e32 %9, %9\n v_fma_f32 %4, %4, %10, %10\n v_fma_f32 %5, %5, %10, %10\n"
                   "v_mfma_f32_32x32x16_f16 %1, %2, %3, %1\n v_fma_f32 %7, %7, %10, %10\n v_fma_f32 %8, %8, %10, %10\n v_exp_f32_e32 %6, %6\n v_fma_f32 %4, %4, %10, %10\n"
                   "v_fma_f32 %5, %5, %10, %10\n v_exp_f32_e32 %9, %9\n v_fma_f32 %7, %7, %10, %10\n v_fma_f32 %8, %8, %10, %10\n"
                   : "+a"(acc0), "+a"(acc1) : "v"(ha), "v"(hb), "v"(f[0]), "v"(f[1]), "v"(f[2]), "v"(f[3]), "v"(f[4]), "v"(f[5]), "v"(c));
    if constexpr (V == 28)      // the same VALU work without the MFMAs
      asm volatile("v_fma_f32 %4, %4, %10, %10\n v_fma_f32 %5, %5, %10, %10\n v_exp_f32_e32 %6, %6\n v_fma_f32 %7, %7, %10, %10\n"
                   "v_fma_f32 %8, %8, %10, %10\n v_exp_f32_e32 %9, %9\n v_fma_f32 %4, %4, %10, %10\n v_fma_f32 %5, %5, %10, %10\n"
                   "v_fma_f32 %7, %7, %10, %10\n v_fma_f32 %8, %8, %10, %10\n v_exp_f32_e32 %6, %6\n v_fma_f32 %4, %4, %10, %10\n"
                   "v_fma_f32 %5, %5, %10, %10\n v_exp_f32_e32 %9, %9\n v_fma_f32 %7, %7, %10, %10\n v_fma_f32 %8, %8, %10, %10\n"
                   : "+a"(acc0), "+a"(acc1) : "v"(ha), "v"(hb), "v"(f[0]), "v"(f[1]), "v"(f[2]), "v"(f[3]), "v"(f[4]), "v"(f[5]), "v"(c));
    if constexpr (V == 19)      // 8 SALU
      asm volatile("s_add_u32 %0, %0, 1\n s_add_u32 %1, %1, 1\n s_add_u32 %0, %0, 1\n s_add_u32 %1, %1, 1\n s_add_u32 %0, %0, 1\n s_add_u32 %1, %1, 1\n s_add_u32 %0, %0, 1\n s_add_u32 %1, %1, 1\n"
                   : "+s"(sa), "+s"(sb) : : "scc");
   }
  }
  const long long t1 = __builtin_readcyclecounter();
  float s = acc0[0] + acc1[3] + r0[0] + r1[1] + r2[2] + r3[3] + (float)(sa + sb);
  for (int i = 0; i < 8; ++i) s += e[i] + f[i] + p[i][0] + p[i][1];
  out[blockIdx.x * 256 + (threadIdx.x & 255)] = s;
  if (threadIdx.x == 0) clk[blockIdx.x] = t1 - t0;
}

static int g_threads = 256;   // 256 = one wave per SIMD, 512 = two
template <int V>
static void run(const char* name, int n_instr, float* out, long long* clk) {
  const int iters = 4000;      // x 8 unrolled repetitions of the pattern per loop trip
  valu_kernel<V><<<256, g_threads>>>(out, iters, clk);
  CK(hipDeviceSynchronize());
  valu_kernel<V><<<256, g_threads>>>(out, iters, clk);
  CK(hipDeviceSynchronize());
  std::vector<long long> c(256);
  CK(hipMemcpy(c.data(), clk, 256 * 8, hipMemcpyDeviceToHost));
  double cyc = 0; for (int i = 0; i < 256; ++i) cyc += c[i];
  cyc /= 256;
  printf("%-44s %7.2f cycles per iteration = %5.2f per instruction\n", name, cyc / iters / 8, cyc / iters / 8 / n_instr);
}

int main(int argc, char** argv) {
  if (argc > 1) g_threads = atoi(argv[1]);
  printf("%d threads per workgroup = %d wave(s) per SIMD; cycles are per wave\n", g_threads, g_threads / 256);
  float* out; long long* clk;
  CK(hipMalloc(&out, 256 * 256 * 4)); CK(hipMalloc(&clk, 256 * 8));
  run<0>("8 v_exp_f32", 8, out, clk);
  run<1>("8 v_fma_f32", 8, out, clk);
  run<2>("8 v_exp_f32 then 8 v_fma_f32", 16, out, clk);
  run<3>("8 x (v_exp_f32, v_fma_f32) interleaved", 16, out, clk);
  run<4>("2 x (v_exp_f32, 3 v_fma_f32)", 8, out, clk);
  run<5>("8 v_exp_f16", 8, out, clk);
  run<6>("8 v_add_f32", 8, out, clk);
  run<7>("8 v_cvt_pk_f16_f32", 8, out, clk);
  run<8>("8 v_max3_f32", 8, out, clk);
  run<9>("8 v_pk_fma_f32", 8, out, clk);
  run<10>("8 v_pk_add_f32", 8, out, clk);
  run<11>("8 v_pk_mul_f32", 8, out, clk);
  run<12>("8 v_dot2c_f32_f16", 8, out, clk);
  run<13>("4 ds_read_b128 + wait for all", 4, out, clk);
  run<14>("2 x (MFMA 32x32x16, 3 v_fma)", 2, out, clk);
  run<15>("2 x (MFMA 32x32x16, 4 v_fma)", 2, out, clk);
  run<16>("2 x (MFMA 32x32x16, exp, 2 v_fma, ds_read)", 2, out, clk);
  run<17>("4 x (MFMA 16x16x32, 1 v_fma)", 4, out, clk);
  run<18>("4 x MFMA 16x16x32 bare", 4, out, clk);
  run<19>("8 s_add_u32", 8, out, clk);
  run<21>("2 x (MFMA 32x32x16 VGPR acc, 1 v_fma)", 2, out, clk);
  run<22>("2 x (MFMA 32x32x16 VGPR acc, 2 v_fma)", 2, out, clk);
  run<24>("2 x (MFMA 32x32x16 AGPR acc, 1 v_fma)", 2, out, clk);
  run<23>("2 x (MFMA 32x32x16 AGPR acc, 2 v_fma)", 2, out, clk);
  run<20>("2 x (MFMA 32x32x16 AGPR acc, 3 v_fma)", 2, out, clk);
  run<26>("MFMA, fma, exp, add, MFMA  (VGPR acc)", 2, out, clk);
  run<25>("MFMA, fma, exp, add, MFMA  (AGPR acc)", 2, out, clk);
  run<27>("2 x (MFMA 32x32x16, 6 v_fma + 2 v_exp)  [per MFMA]", 2, out, clk);
  run<28>("2 x (6 v_fma + 2 v_exp), no MFMA        [per group]", 2, out, clk);
  return 0;
}
