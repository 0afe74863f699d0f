// How far apart must two 32x32x16 f16 MFMAs on the SAME accumulator be issued to run at the pipe's rate?  (Round 4: the
// attention kernel's MFMA-only ablation ran 37.7 cycles per MFMA where the bare 8-accumulator loop of shape_lab.hip runs
// 33.5; its S^T phase is TWO chains of 16 dependent MFMAs.)  One wave per SIMD, 256 workgroups, CH independent chains issued
// round-robin (so a chain's consecutive MFMAs are CH issue slots apart), accumulators in VGPRs or AGPRs.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 chain_lab.hip -o chain_lab
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float floatx16 __attribute__((ext_vector_type(16)));

template <int CH, int AG>
__global__ void __launch_bounds__(256, 1) chain_kernel(const half8* __restrict__ g, float* __restrict__ out, int iters, long long* clk) {
  const int lane = threadIdx.x & 63;
  half8 a[4], b[4];
  for (int i = 0; i < 4; ++i) { a[i] = g[(blockIdx.x * 8 + i) * 64 + lane]; b[i] = g[(blockIdx.x * 8 + 4 + i) * 64 + lane]; }
  floatx16 acc[CH];
  for (int k = 0; k < CH; ++k) for (int i = 0; i < 16; ++i) acc[k][i] = 0.f;
  const long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      if constexpr (AG) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(acc[j % CH]) : "v"(a[j & 3]), "v"(b[(j >> 2) & 3]));
      else asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc[j % CH]) : "v"(a[j & 3]), "v"(b[(j >> 2) & 3]));
    }
  }
  const long long t1 = __builtin_readcyclecounter();
  float s = 0.f;
  for (int k = 0; k < CH; ++k) for (int i = 0; i < 16; ++i) s += acc[k][i];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0) clk[blockIdx.x] = t1 - t0;
}

template <int CH, int AG>
static void run(const half8* g, float* out, long long* clk) {
  const int iters = 100000;
  chain_kernel<CH, AG><<<256, 256>>>(g, out, iters / 4, clk);
  CK(hipDeviceSynchronize());
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  CK(hipEventRecord(e0));
  chain_kernel<CH, AG><<<256, 256>>>(g, out, iters, clk);
  CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  std::vector<long long> c(256);
  CK(hipMemcpy(c.data(), clk, 256 * 8, hipMemcpyDeviceToHost));
  double cyc = 0; for (int i = 0; i < 256; ++i) cyc += c[i];
  cyc /= 256;
  printf("%2d chain(s), %s accumulators: %6.2f cycles per MFMA   %7.0f TF/s   clock %.2f GHz\n", CH, AG ? "AGPR" : "VGPR",
         cyc / (iters * 16.0), 256.0 * 4 * iters * 16 * 32768.0 / ms / 1e9, cyc / (ms * 1e-3) / 1e9);
}

int main() {
  half8* g; float* out; long long* clk;
  std::vector<_Float16> h(256 * 8 * 64 * 8);
  srand(1);
  for (auto& v : h) v = (_Float16)((rand() / (float)RAND_MAX) * 2.f - 1.f);
  CK(hipMalloc(&g, h.size() * 2)); CK(hipMemcpy(g, h.data(), h.size() * 2, hipMemcpyHostToDevice));
  CK(hipMalloc(&out, 256 * 256 * 4)); CK(hipMalloc(&clk, 256 * 8));
  for (int rep = 0; rep < 2; ++rep) {
    run<1, 0>(g, out, clk); run<2, 0>(g, out, clk); run<4, 0>(g, out, clk); run<8, 0>(g, out, clk);
    run<1, 1>(g, out, clk); run<2, 1>(g, out, clk); run<4, 1>(g, out, clk); run<8, 1>(g, out, clk); run<16, 1>(g, out, clk);
  }
  return 0;
}
