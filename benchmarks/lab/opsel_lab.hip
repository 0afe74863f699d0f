// What do op_sel / op_sel_hi do on src2 of v_pk_fma_f32 on gfx950, and what does v_dot2c_f32_f16 cost with a literal?
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 opsel_lab.hip -o opsel_lab
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
typedef float float2v __attribute__((ext_vector_type(2)));

__global__ void opsel_kernel(float* out) {
  float2v x = {1.f, 2.f}, c = {10.f, 100.f}, m = {3.f, 5.f}, r;
  int k = 0;
#define T(OPS)                                                                                 \
  asm volatile("v_pk_fma_f32 %0, %1, %2, %3 " OPS : "=v"(r) : "v"(x), "v"(c), "v"(m));         \
  out[k++] = r[0]; out[k++] = r[1];
  T("")                                              // (1*10+3, 2*100+5) = 13, 205
  T("op_sel_hi:[1,1,0]")                             // src2 low for both: 13, 203
  T("op_sel:[0,0,1]")                                // src2 high for the low result: 15, 205 ?
  T("op_sel:[0,0,1] op_sel_hi:[1,1,1]")              // 15, 205
  T("op_sel:[0,0,1] op_sel_hi:[1,1,0]")              // swapped: 15, 203
  T("op_sel:[0,1,0]")                                // src1 high for the low result: 103, 205
  T("op_sel:[1,0,0]")                                // src0 high for the low result: 23, 205
  T("op_sel_hi:[1,1,0] neg_lo:[0,0,1] neg_hi:[0,0,1]")   // 10-3 = 7, 200-3 = 197
  T("op_sel:[0,0,1] neg_lo:[0,0,1] neg_hi:[0,0,1]")      // 10-5 = 5, 200-5 = 195 ?
  // the same with src1 in an SGPR pair (the attention kernel's form)
  float2v cs = {10.f, 100.f};
#define TS(OPS)                                                                                \
  asm volatile("v_pk_fma_f32 %0, %1, %2, %3 " OPS : "=v"(r) : "v"(x), "s"(cs), "v"(m));        \
  out[k++] = r[0]; out[k++] = r[1];
  TS("op_sel_hi:[1,1,0] neg_lo:[0,0,1] neg_hi:[0,0,1]")  // 7 197
  TS("op_sel:[0,0,1] neg_lo:[0,0,1] neg_hi:[0,0,1]")     // 5 195
  TS("op_sel:[0,0,1] op_sel_hi:[1,1,1] neg_lo:[0,0,1] neg_hi:[0,0,1]")     // 5 195
  // in place (dst = src0), as the kernel does
  float2v y = x;
  asm volatile("v_pk_fma_f32 %0, %0, %1, %2 op_sel:[0,0,1] neg_lo:[0,0,1] neg_hi:[0,0,1]" : "+v"(y) : "s"(cs), "v"(m));
  out[k++] = y[0]; out[k++] = y[1];                      // 5 195
}

template <int V>
__global__ void __launch_bounds__(256, 1) dot_kernel(float* out, int iters, long long* clk) {
  typedef float floatx4 __attribute__((ext_vector_type(4)));
  typedef float floatx16 __attribute__((ext_vector_type(16)));
  typedef _Float16 half8 __attribute__((ext_vector_type(8)));
  floatx4 a0 = {0}, a1 = {0}, a2 = {0}, a3 = {0};
  floatx16 b0 = {0}, b1 = {0};
  half8 ha = {1, 2, 3, 4, 5, 6, 7, 8}, hb = {1, 1, 1, 1, 2, 2, 2, 2};
  float2v p2[4] = {{1.f, 2.f}, {1.5f, 2.5f}, {0.5f, 0.25f}, {3.f, 4.f}};
  float e[8], f = 0.5f + threadIdx.x;
  for (int i = 0; i < 8; ++i) e[i] = 0.001f * (threadIdx.x + i);
  const long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int rep = 0; rep < 8; ++rep) {
      if constexpr (V == 0)
        asm volatile("v_dot2c_f32_f16 %0, %8, %8\n v_dot2c_f32_f16 %1, %8, %8\n v_dot2c_f32_f16 %2, %8, %8\n v_dot2c_f32_f16 %3, %8, %8\n"
                     "v_dot2c_f32_f16 %4, %8, %8\n v_dot2c_f32_f16 %5, %8, %8\n v_dot2c_f32_f16 %6, %8, %8\n v_dot2c_f32_f16 %7, %8, %8\n"
                     : "+v"(e[0]), "+v"(e[1]), "+v"(e[2]), "+v"(e[3]), "+v"(e[4]), "+v"(e[5]), "+v"(e[6]), "+v"(e[7]) : "v"(f));
      if constexpr (V == 1)
        asm volatile("v_dot2c_f32_f16 %0, 0x3c003c00, %8\n v_dot2c_f32_f16 %1, 0x3c003c00, %8\n v_dot2c_f32_f16 %2, 0x3c003c00, %8\n v_dot2c_f32_f16 %3, 0x3c003c00, %8\n"
                     "v_dot2c_f32_f16 %4, 0x3c003c00, %8\n v_dot2c_f32_f16 %5, 0x3c003c00, %8\n v_dot2c_f32_f16 %6, 0x3c003c00, %8\n v_dot2c_f32_f16 %7, 0x3c003c00, %8\n"
                     : "+v"(e[0]), "+v"(e[1]), "+v"(e[2]), "+v"(e[3]), "+v"(e[4]), "+v"(e[5]), "+v"(e[6]), "+v"(e[7]) : "v"(f));
      if constexpr (V == 2)      // ONE accumulator (the kernels' row sum): a dependent chain
        asm volatile("v_dot2c_f32_f16 %0, %1, %1\n v_dot2c_f32_f16 %0, %1, %1\n v_dot2c_f32_f16 %0, %1, %1\n v_dot2c_f32_f16 %0, %1, %1\n"
                     "v_dot2c_f32_f16 %0, %1, %1\n v_dot2c_f32_f16 %0, %1, %1\n v_dot2c_f32_f16 %0, %1, %1\n v_dot2c_f32_f16 %0, %1, %1\n"
                     : "+v"(e[0]) : "v"(f));
      if constexpr (V == 4)      // beside MFMAs: MFMA 16x16x32, dot2c (independent accumulators)
        asm volatile("v_mfma_f32_16x16x32_f16 %8, %10, %11, %8\n v_dot2c_f32_f16 %0, 0x3c003c00, %12\n v_mfma_f32_16x16x32_f16 %9, %10, %11, %9\n v_dot2c_f32_f16 %1, 0x3c003c00, %12\n"
                     "v_mfma_f32_16x16x32_f16 %13, %10, %11, %13\n v_dot2c_f32_f16 %2, 0x3c003c00, %12\n v_mfma_f32_16x16x32_f16 %14, %10, %11, %14\n v_dot2c_f32_f16 %3, 0x3c003c00, %12\n"
                     : "+v"(e[0]), "+v"(e[1]), "+v"(e[2]), "+v"(e[3]), "+v"(e[4]), "+v"(e[5]), "+v"(e[6]), "+v"(e[7]), "+v"(a0), "+v"(a1)
                     : "v"(ha), "v"(hb), "v"(f), "v"(a2), "v"(a3));
      if constexpr (V == 5)      // the same with v_add_f32
        asm volatile("v_mfma_f32_16x16x32_f16 %8, %10, %11, %8\n v_add_f32 %0, %0, %12\n v_mfma_f32_16x16x32_f16 %9, %10, %11, %9\n v_add_f32 %1, %1, %12\n"
                     "v_mfma_f32_16x16x32_f16 %13, %10, %11, %13\n v_add_f32 %2, %2, %12\n v_mfma_f32_16x16x32_f16 %14, %10, %11, %14\n v_add_f32 %3, %3, %12\n"
                     : "+v"(e[0]), "+v"(e[1]), "+v"(e[2]), "+v"(e[3]), "+v"(e[4]), "+v"(e[5]), "+v"(e[6]), "+v"(e[7]), "+v"(a0), "+v"(a1)
                     : "v"(ha), "v"(hb), "v"(f), "v"(a2), "v"(a3));
      if constexpr (V == 6)      // MFMA 32x32x16 + dot2c
        asm volatile("v_mfma_f32_32x32x16_f16 %8, %10, %11, %8\n v_dot2c_f32_f16 %0, 0x3c003c00, %12\n v_dot2c_f32_f16 %1, 0x3c003c00, %12\n"
                     "v_mfma_f32_32x32x16_f16 %9, %10, %11, %9\n v_dot2c_f32_f16 %2, 0x3c003c00, %12\n v_dot2c_f32_f16 %3, 0x3c003c00, %12\n"
                     : "+v"(e[0]), "+v"(e[1]), "+v"(e[2]), "+v"(e[3]), "+v"(e[4]), "+v"(e[5]), "+v"(e[6]), "+v"(e[7]), "+v"(b0), "+v"(b1)
                     : "v"(ha), "v"(hb), "v"(f));
      if constexpr (V == 7)      // MFMA 32x32x16 + add
        asm volatile("v_mfma_f32_32x32x16_f16 %8, %10, %11, %8\n v_add_f32 %0, %0, %12\n v_add_f32 %1, %1, %12\n"
                     "v_mfma_f32_32x32x16_f16 %9, %10, %11, %9\n v_add_f32 %2, %2, %12\n v_add_f32 %3, %3, %12\n"
                     : "+v"(e[0]), "+v"(e[1]), "+v"(e[2]), "+v"(e[3]), "+v"(e[4]), "+v"(e[5]), "+v"(e[6]), "+v"(e[7]), "+v"(b0), "+v"(b1)
                     : "v"(ha), "v"(hb), "v"(f));
      if constexpr (V == 10)
        asm volatile("v_mfma_f32_16x16x32_f16 %8, %10, %11, %8\n v_fma_f32 %0, %0, %12, %12\n"
                     "v_mfma_f32_16x16x32_f16 %9, %10, %11, %9\n v_fma_f32 %1, %1, %12, %12\n"
                     "v_mfma_f32_16x16x32_f16 %13, %10, %11, %13\n v_fma_f32 %2, %2, %12, %12\n"
                     "v_mfma_f32_16x16x32_f16 %14, %10, %11, %14\n v_fma_f32 %3, %3, %12, %12\n"
                     : "+v"(e[0]), "+v"(e[1]), "+v"(e[2]), "+v"(e[3]), "+v"(e[4]), "+v"(e[5]), "+v"(e[6]), "+v"(e[7]), "+v"(a0), "+v"(a1)
                     : "v"(ha), "v"(hb), "v"(f), "v"(a2), "v"(a3), "v"(p2[0]), "v"(p2[1]), "v"(p2[2]), "v"(p2[3]));
      if constexpr (V == 11)
        asm volatile("v_mfma_f32_16x16x32_f16 %8, %10, %11, %8\n v_exp_f32 %0, %0\n"
                     "v_mfma_f32_16x16x32_f16 %9, %10, %11, %9\n v_exp_f32 %1, %1\n"
                     "v_mfma_f32_16x16x32_f16 %13, %10, %11, %13\n v_exp_f32 %2, %2\n"
                     "v_mfma_f32_16x16x32_f16 %14, %10, %11, %14\n v_exp_f32 %3, %3\n"
                     : "+v"(e[0]), "+v"(e[1]), "+v"(e[2]), "+v"(e[3]), "+v"(e[4]), "+v"(e[5]), "+v"(e[6]), "+v"(e[7]), "+v"(a0), "+v"(a1)
                     : "v"(ha), "v"(hb), "v"(f), "v"(a2), "v"(a3), "v"(p2[0]), "v"(p2[1]), "v"(p2[2]), "v"(p2[3]));
      if constexpr (V == 12)
        asm volatile("v_mfma_f32_16x16x32_f16 %8, %10, %11, %8\n v_cvt_pk_f16_f32 %0, %0, %12\n"
                     "v_mfma_f32_16x16x32_f16 %9, %10, %11, %9\n v_cvt_pk_f16_f32 %1, %1, %12\n"
                     "v_mfma_f32_16x16x32_f16 %13, %10, %11, %13\n v_cvt_pk_f16_f32 %2, %2, %12\n"
                     "v_mfma_f32_16x16x32_f16 %14, %10, %11, %14\n v_cvt_pk_f16_f32 %3, %3, %12\n"
                     : "+v"(e[0]), "+v"(e[1]), "+v"(e[2]), "+v"(e[3]), "+v"(e[4]), "+v"(e[5]), "+v"(e[6]), "+v"(e[7]), "+v"(a0), "+v"(a1)
                     : "v"(ha), "v"(hb), "v"(f), "v"(a2), "v"(a3), "v"(p2[0]), "v"(p2[1]), "v"(p2[2]), "v"(p2[3]));
      if constexpr (V == 13)
        asm volatile("v_mfma_f32_16x16x32_f16 %8, %10, %11, %8\n v_max3_f32 %0, %0, %12, %12\n"
                     "v_mfma_f32_16x16x32_f16 %9, %10, %11, %9\n v_max3_f32 %1, %1, %12, %12\n"
                     "v_mfma_f32_16x16x32_f16 %13, %10, %11, %13\n v_max3_f32 %2, %2, %12, %12\n"
                     "v_mfma_f32_16x16x32_f16 %14, %10, %11, %14\n v_max3_f32 %3, %3, %12, %12\n"
                     : "+v"(e[0]), "+v"(e[1]), "+v"(e[2]), "+v"(e[3]), "+v"(e[4]), "+v"(e[5]), "+v"(e[6]), "+v"(e[7]), "+v"(a0), "+v"(a1)
                     : "v"(ha), "v"(hb), "v"(f), "v"(a2), "v"(a3), "v"(p2[0]), "v"(p2[1]), "v"(p2[2]), "v"(p2[3]));
      if constexpr (V == 14)
        asm volatile("v_mfma_f32_16x16x32_f16 %8, %10, %11, %8\n v_pk_fma_f32 %15, %15, %15, %15\n"
                     "v_mfma_f32_16x16x32_f16 %9, %10, %11, %9\n v_pk_fma_f32 %16, %16, %16, %16\n"
                     "v_mfma_f32_16x16x32_f16 %13, %10, %11, %13\n v_pk_fma_f32 %17, %17, %17, %17\n"
                     "v_mfma_f32_16x16x32_f16 %14, %10, %11, %14\n v_pk_fma_f32 %18, %18, %18, %18\n"
                     : "+v"(e[0]), "+v"(e[1]), "+v"(e[2]), "+v"(e[3]), "+v"(e[4]), "+v"(e[5]), "+v"(e[6]), "+v"(e[7]), "+v"(a0), "+v"(a1)
                     : "v"(ha), "v"(hb), "v"(f), "v"(a2), "v"(a3), "v"(p2[0]), "v"(p2[1]), "v"(p2[2]), "v"(p2[3]));
      if constexpr (V == 15)
        asm volatile("v_mfma_f32_16x16x32_f16 %8, %10, %11, %8\n v_pk_add_f32 %15, %15, %15\n"
                     "v_mfma_f32_16x16x32_f16 %9, %10, %11, %9\n v_pk_add_f32 %16, %16, %16\n"
                     "v_mfma_f32_16x16x32_f16 %13, %10, %11, %13\n v_pk_add_f32 %17, %17, %17\n"
                     "v_mfma_f32_16x16x32_f16 %14, %10, %11, %14\n v_pk_add_f32 %18, %18, %18\n"
                     : "+v"(e[0]), "+v"(e[1]), "+v"(e[2]), "+v"(e[3]), "+v"(e[4]), "+v"(e[5]), "+v"(e[6]), "+v"(e[7]), "+v"(a0), "+v"(a1)
                     : "v"(ha), "v"(hb), "v"(f), "v"(a2), "v"(a3), "v"(p2[0]), "v"(p2[1]), "v"(p2[2]), "v"(p2[3]));
      if constexpr (V == 16)
        asm volatile("v_mfma_f32_16x16x32_f16 %8, %10, %11, %8\n v_pk_mul_f16 %0, %0, %12\n"
                     "v_mfma_f32_16x16x32_f16 %9, %10, %11, %9\n v_pk_mul_f16 %1, %1, %12\n"
                     "v_mfma_f32_16x16x32_f16 %13, %10, %11, %13\n v_pk_mul_f16 %2, %2, %12\n"
                     "v_mfma_f32_16x16x32_f16 %14, %10, %11, %14\n v_pk_mul_f16 %3, %3, %12\n"
                     : "+v"(e[0]), "+v"(e[1]), "+v"(e[2]), "+v"(e[3]), "+v"(e[4]), "+v"(e[5]), "+v"(e[6]), "+v"(e[7]), "+v"(a0), "+v"(a1)
                     : "v"(ha), "v"(hb), "v"(f), "v"(a2), "v"(a3), "v"(p2[0]), "v"(p2[1]), "v"(p2[2]), "v"(p2[3]));
      if constexpr (V == 3)      // the same chain of v_add_f32
        asm volatile("v_add_f32 %0, %0, %1\n v_add_f32 %0, %0, %1\n v_add_f32 %0, %0, %1\n v_add_f32 %0, %0, %1\n"
                     "v_add_f32 %0, %0, %1\n v_add_f32 %0, %0, %1\n v_add_f32 %0, %0, %1\n v_add_f32 %0, %0, %1\n"
                     : "+v"(e[0]) : "v"(f));
    }
  }
  const long long t1 = __builtin_readcyclecounter();
  float s = 0.f;
  for (int i = 0; i < 8; ++i) s += e[i];
  s += a0[0] + a1[1] + a2[2] + a3[3] + b0[0] + b1[5];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0) clk[blockIdx.x] = t1 - t0;
}

template <int V>
static void run(const char* name, float* out, long long* clk) {
  const int iters = 4000;
  dot_kernel<V><<<256, 256>>>(out, iters, clk);
  CK(hipDeviceSynchronize());
  dot_kernel<V><<<256, 256>>>(out, iters, clk);
  CK(hipDeviceSynchronize());
  long long c[256];
  CK(hipMemcpy(c, clk, sizeof(c), hipMemcpyDeviceToHost));
  double cyc = 0; for (int i = 0; i < 256; ++i) cyc += c[i];
  printf("%-52s %6.2f cycles per instruction\n", name, cyc / 256 / iters / 64);
}

int main() {
  float* out; long long* clk;
  CK(hipMalloc(&out, 256 * 256 * 4)); CK(hipMalloc(&clk, 256 * 8));
  opsel_kernel<<<1, 1>>>(out);
  float h[26];
  CK(hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost));
  const char* names[9] = {"plain                          expect 13 205", "op_sel_hi:[1,1,0]              expect 13 203",
                          "op_sel:[0,0,1]                 expect 15 205", "op_sel:[0,0,1] hi:[1,1,1]      expect 15 205",
                          "op_sel:[0,0,1] hi:[1,1,0]      expect 15 203", "op_sel:[0,1,0]                 expect 103 205",
                          "op_sel:[1,0,0]                 expect 23 205", "hi:[1,1,0] neg src2            expect 7 197",
                          "op_sel:[0,0,1] neg src2        expect 5 195"};
  for (int i = 0; i < 9; ++i) printf("v_pk_fma_f32 %s   got %g %g\n", names[i], h[2 * i], h[2 * i + 1]);
  const char* names2[4] = {"SGPR src1, hi:[1,1,0] neg      expect 7 197", "SGPR src1, op_sel:[0,0,1] neg  expect 5 195",
                           "SGPR src1, sel+hi explicit neg expect 5 195", "SGPR src1, in place, sel neg   expect 5 195"};
  for (int i = 0; i < 4; ++i) printf("v_pk_fma_f32 %s   got %g %g\n", names2[i], h[18 + 2 * i], h[19 + 2 * i]);
  run<0>("8 independent v_dot2c_f32_f16, VGPR operands", out, clk);
  run<1>("8 independent v_dot2c_f32_f16, literal (1, 1)", out, clk);
  run<2>("8 v_dot2c_f32_f16 into ONE accumulator", out, clk);
  run<3>("8 v_add_f32 into ONE accumulator", out, clk);
  printf("(the next four: cycles per 8 issued instructions / 8, i.e. x 2 = per MFMA + VALU pair (16x16) and x 3 per MFMA + 2 VALU (32x32))\n");
  run<4>("4 x (MFMA 16x16x32, v_dot2c literal)", out, clk);
  run<5>("4 x (MFMA 16x16x32, v_add_f32)", out, clk);
  run<10>("4 x (MFMA 16x16x32, v_fma_f32)", out, clk);
  run<11>("4 x (MFMA 16x16x32, v_exp_f32)", out, clk);
  run<12>("4 x (MFMA 16x16x32, v_cvt_pk_f16_f32)", out, clk);
  run<13>("4 x (MFMA 16x16x32, v_max3_f32)", out, clk);
  run<14>("4 x (MFMA 16x16x32, v_pk_fma_f32)", out, clk);
  run<15>("4 x (MFMA 16x16x32, v_pk_add_f32)", out, clk);
  run<16>("4 x (MFMA 16x16x32, v_pk_mul_f16)", out, clk);
  run<6>("2 x (MFMA 32x32x16, 2 v_dot2c literal)", out, clk);
  run<7>("2 x (MFMA 32x32x16, 2 v_add_f32)", out, clk);
  return 0;
}
