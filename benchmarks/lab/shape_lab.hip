// Which f16 MFMA shape does THIS box run faster under its power cap?  (MI355X_MICROARCH.md 'DVFS give-back' (7): the
// 16x16x32 loop delivered 1.12-1.15x the FLOP/s of the 32x32x16 loop at equal cycles on random data.)  Bare loops,
// operands in registers, one wave per SIMD (256 threads x 256 workgroups), 128 accumulator registers per wave in both:
//   A: 8 x v_mfma_f32_32x32x16_f16 per iteration     B: 32 x v_mfma_f32_16x16x32_f16 per iteration   (same FLOPs)
// and the same with a filler of VALU work (4 v_exp + 8 v_fma per 8 / 32 MFMAs: the attention kernel's density), because an
// MFMA of either shape holds the SIMD's vector issue for 8 cycles - twice as often per FLOP in the 16x16 shape.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 shape_lab.hip -o shape_lab
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef float floatx4 __attribute__((ext_vector_type(4)));

template <int SHAPE, int VALU>
__global__ void __launch_bounds__(256, 1) shape_kernel(const half8* __restrict__ g, float* __restrict__ out, int iters,
                                                       long long* clk, long long* rt) {
  const int lane = threadIdx.x & 63;
  half8 a[4], b[4];
  for (int i = 0; i < 4; ++i) { a[i] = g[(blockIdx.x * 8 + i) * 64 + lane]; b[i] = g[(blockIdx.x * 8 + 4 + i) * 64 + lane]; }
  float e[4] = {0.1f * lane, 0.2f, 0.3f, 0.4f};
  float s = 0.f;
  const long long t0 = __builtin_readcyclecounter();
  const long long r0 = __builtin_amdgcn_s_memrealtime();
  if constexpr (SHAPE == 32) {
    floatx16 acc[8];
    for (int k = 0; k < 8; ++k) for (int i = 0; i < 16; ++i) acc[k][i] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc[j]) : "v"(a[j & 3]), "v"(b[(j >> 1) & 3]));
        if (VALU && (j & 1)) asm volatile("v_fma_f32 %0, %0, %1, %1\n\tv_exp_f32_e32 %0, %0\n\tv_fma_f32 %0, %0, %1, %1" : "+v"(e[j >> 1]) : "v"(e[(j >> 1) ^ 1]));
      }
    }
    for (int k = 0; k < 8; ++k) for (int i = 0; i < 16; ++i) s += acc[k][i];
  } else {
    floatx4 acc[32];
    for (int k = 0; k < 32; ++k) for (int i = 0; i < 4; ++i) acc[k][i] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int j = 0; j < 32; ++j) {
        asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc[j]) : "v"(a[j & 3]), "v"(b[(j >> 2) & 3]));
        if (VALU && (j & 7) == 7) asm volatile("v_fma_f32 %0, %0, %1, %1\n\tv_exp_f32_e32 %0, %0\n\tv_fma_f32 %0, %0, %1, %1" : "+v"(e[j >> 3]) : "v"(e[(j >> 3) ^ 1]));
      }
    }
    for (int k = 0; k < 32; ++k) for (int i = 0; i < 4; ++i) s += acc[k][i];
  }
  const long long t1 = __builtin_readcyclecounter();
  const long long r1 = __builtin_amdgcn_s_memrealtime();
  out[blockIdx.x * 256 + threadIdx.x] = s + e[0] + e[1] + e[2] + e[3];
  if (threadIdx.x == 0) { clk[blockIdx.x] = t1 - t0; rt[blockIdx.x] = r1 - r0; }
}

template <int SHAPE, int VALU>
static void run(const char* name, const half8* g, float* out, long long* clk, long long* rt) {
  const int iters = 200000;
  for (int w = 0; w < 2; ++w) shape_kernel<SHAPE, VALU><<<256, 256>>>(g, out, iters / 4, clk, rt);
  CK(hipDeviceSynchronize());
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  CK(hipEventRecord(e0));
  shape_kernel<SHAPE, VALU><<<256, 256>>>(g, out, iters, clk, rt);
  CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  std::vector<long long> c(256), r(256);
  CK(hipMemcpy(c.data(), clk, 256 * 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(r.data(), rt, 256 * 8, hipMemcpyDeviceToHost));
  double cyc = 0, rtm = 0; for (int i = 0; i < 256; ++i) { cyc += c[i]; rtm += r[i]; }
  cyc /= 256; rtm /= 256;
  const double flop = 256.0 * 4 * iters * 262144.0;
  printf("%-44s %8.2f ms  %7.0f TF/s   %.2f cycles per 32x32x16-equivalent MFMA   in-kernel clock %.2f GHz\n", name, ms,
         flop / ms / 1e9, cyc / (iters * 8.0), cyc / rtm * 0.1);
}

int main() {
  half8* g; float* out; long long *clk, *rt;
  std::vector<_Float16> h(256 * 8 * 64 * 8);
  srand(1);
  for (auto& v : h) v = (_Float16)((rand() / (float)RAND_MAX) * 2.f - 1.f);
  CK(hipMalloc(&g, h.size() * 2)); CK(hipMemcpy(g, h.data(), h.size() * 2, hipMemcpyHostToDevice));
  CK(hipMalloc(&out, 256 * 256 * 4)); CK(hipMalloc(&clk, 256 * 8)); CK(hipMalloc(&rt, 256 * 8));
  for (int rep = 0; rep < 2; ++rep) {
    run<32, 0>("32x32x16 bare", g, out, clk, rt);
    run<16, 0>("16x16x32 bare", g, out, clk, rt);
    run<32, 1>("32x32x16 + 1 exp, 2 fma per 2 MFMA", g, out, clk, rt);
    run<16, 1>("16x16x32 + 1 exp, 2 fma per 8 MFMA (same)", g, out, clk, rt);
  }
  return 0;
}
