// Can a CU feed its matrix pipe from LDS and from L1/L2 (plain global loads into VGPRs) at the same time?
// 4 waves per CU (one per SIMD, like the head-dim-256 attention kernel).  Per MFMA "slot" each wave needs one 1-KiB
// A fragment.  Variants: all fragments from LDS (today's kernel: 4 x 1 KiB per 32-cycle slot = the LDS port's peak),
// every 2nd / every 4th fragment by a fully coalesced 1-KiB global_load_dwordx4 from an L2-resident buffer instead.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 mix_lab.hip -o mix_lab
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float floatx16 __attribute__((ext_vector_type(16)));

template <int GEVERY>   // 0: LDS only; n: every n-th fragment from global
__global__ void __launch_bounds__(256, 1) mix_kernel(const half8* __restrict__ g, float* __restrict__ out, int iters, long long* clk) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int i = threadIdx.x; i < 65536 / 16; i += 256) ((half8*)smem)[i] = g[i];
  __syncthreads();
  floatx16 acc[8];
  for (int k = 0; k < 8; ++k) for (int i = 0; i < 16; ++i) acc[k][i] = 0.f;
  half8 b; for (int i = 0; i < 8; ++i) b[i] = (_Float16)1.f;
  half8 fr[8];
  const char* lbase = smem + lane * 16;
  const half8* gbase = g + lane + wave * 64;
  // 4 fragments in flight
  for (int j = 0; j < 4; ++j) fr[j] = *(const half8*)(lbase + j * 1024);
  const long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int j = 0; j < 32; ++j) {
      acc[j & 7] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fr[j & 7], b, acc[j & 7], 0, 0, 0);
      const int n = j + 4;
      if (GEVERY != 0 && (n % GEVERY) == 0) fr[n & 7] = gbase[(size_t)((it * 32 + n) & 1023) * 256];   // 1 KiB per wave, 4 KiB rows
      else fr[n & 7] = *(const half8*)(lbase + ((n * 1024 + wave * 4096) & 65535 & ~1023));
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  const long long t1 = __builtin_readcyclecounter();
  float s = 0.f;
  for (int k = 0; k < 8; ++k) for (int i = 0; i < 16; ++i) s += acc[k][i];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0) clk[blockIdx.x] = t1 - t0;
}


__device__ __forceinline__ void glds_one(const char* gbase, unsigned voff, unsigned lds) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %3, %1\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "s"(gbase), "s"(lds), "v"(voff) : "memory", "scc");
}
// LDS fragments + LDS-DMA traffic into the OTHER half of the buffer: one 1-KiB load per GAP MFMAs per wave
template <int GAP>
__global__ void __launch_bounds__(256, 1) dma_kernel(const half8* __restrict__ g, float* __restrict__ out, int iters, long long* clk) {
  extern __shared__ __attribute__((aligned(1024))) char smem[];   // 128 KiB: [0,64K) read, [64K,128K) DMA target
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  for (int i = threadIdx.x; i < 65536 / 16; i += 256) ((half8*)smem)[i] = g[i];
  __syncthreads();
  floatx16 acc[8];
  for (int k = 0; k < 8; ++k) for (int i = 0; i < 16; ++i) acc[k][i] = 0.f;
  half8 b; for (int i = 0; i < 8; ++i) b[i] = (_Float16)1.f;
  half8 fr[8];
  const char* lbase = smem + lane * 16;
  const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(size_t)smem) + 65536 + wave * 16384;
  const unsigned voff = lane * 16 + wave * 16384;
  for (int j = 0; j < 4; ++j) fr[j] = *(const half8*)(lbase + j * 1024);
  const long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
    const char* gt = (const char*)g + (size_t)(it & 31) * 65536;
#pragma unroll
    for (int j = 0; j < 32; ++j) {
      acc[j & 7] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fr[j & 7], b, acc[j & 7], 0, 0, 0);
      const int n = j + 4;
      fr[n & 7] = *(const half8*)(lbase + ((n * 1024 + wave * 4096) & 65535 & ~1023));
      if (GAP != 0 && (j % GAP) == GAP - 1) glds_one(gt, voff + (j / GAP) * 1024, lds0 + ((j / GAP) & 15) * 1024);
      __builtin_amdgcn_sched_barrier(0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  const long long t1 = __builtin_readcyclecounter();
  float s = 0.f;
  for (int k = 0; k < 8; ++k) for (int i = 0; i < 16; ++i) s += acc[k][i];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0) clk[blockIdx.x] = t1 - t0;
}

// LDS fragments + register-staged refill of the other half: per GAP MFMAs one 1-KiB global_load_dwordx4 into a staging
// register and one ds_write_b128 of the register loaded 16 steps earlier (16 loads in flight = 64 VGPRs)
template <int GAP>
__global__ void __launch_bounds__(256, 1) stage_kernel(const half8* __restrict__ g, float* __restrict__ out, int iters, long long* clk) {
  extern __shared__ __attribute__((aligned(1024))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  for (int i = threadIdx.x; i < 65536 / 16; i += 256) ((half8*)smem)[i] = g[i];
  __syncthreads();
  floatx16 acc[8];
  for (int k = 0; k < 8; ++k) for (int i = 0; i < 16; ++i) acc[k][i] = 0.f;
  half8 b; for (int i = 0; i < 8; ++i) b[i] = (_Float16)1.f;
  half8 fr[8];
  constexpr int NS = 32 / GAP;            // loads per 32-MFMA block, all in flight
  half8 st[NS];
  const char* lbase = smem + lane * 16;
  char* wbase = smem + 65536 + wave * 16384 + lane * 16;
  const half8* gw = g + (size_t)wave * 1024 + lane;
  for (int j = 0; j < 4; ++j) fr[j] = *(const half8*)(lbase + j * 1024);
  for (int k = 0; k < NS; ++k) st[k] = gw[k * 64];
  const long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
    const half8* gt = gw + (size_t)((it + 1) & 31) * 4096;
#pragma unroll
    for (int j = 0; j < 32; ++j) {
      acc[j & 7] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fr[j & 7], b, acc[j & 7], 0, 0, 0);
      const int n = j + 4;
      fr[n & 7] = *(const half8*)(lbase + ((n * 1024 + wave * 4096) & 65535 & ~1023));
      if ((j % GAP) == GAP - 1) {
        const int k = j / GAP;
        *(half8*)(wbase + (k & 15) * 1024) = st[k];      // loaded one block (32 MFMAs ~ 1100 cycles) ago
        st[k] = gt[k * 64];
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  const long long t1 = __builtin_readcyclecounter();
  float s = 0.f;
  for (int k = 0; k < 8; ++k) for (int i = 0; i < 16; ++i) s += acc[k][i];
  for (int k = 0; k < NS; ++k) s += (float)st[k][0];
  out[blockIdx.x * 256 + threadIdx.x] = s + (float)((half8*)wbase)[0][0];
  if (threadIdx.x == 0) clk[blockIdx.x] = t1 - t0;
}

int main() {
  half8* g; float* out; long long* clk;
  const size_t gbytes = (size_t)1024 * 256 * 16 + 65536 + 32 * 65536;     // 4 MiB + : L2-resident, every CU streams the same lines
  CK(hipMalloc(&g, gbytes)); CK(hipMemset(g, 0, gbytes)); CK(hipMalloc(&out, 256 * 256 * 4)); CK(hipMalloc(&clk, 256 * 8));
  const int iters = 2000;
  auto run = [&](const char* name, auto kern) {
    CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
    kern<<<256, 256, 65536>>>(g, out, iters, clk); CK(hipDeviceSynchronize());
    kern<<<256, 256, 65536>>>(g, out, iters, clk); CK(hipDeviceSynchronize());
    long long h[256]; CK(hipMemcpy(h, clk, sizeof(h), hipMemcpyDeviceToHost));
    double avg = 0; for (auto c : h) avg += c; avg /= 256;
    printf("%-44s %.1f cycles per MFMA (matrix pipe needs 32)\n", name, avg / ((double)iters * 32));
  };
  run("all fragments from LDS", mix_kernel<0>);
  run("every 4th fragment from global (L2 hit)", mix_kernel<4>);
  run("every 2nd fragment from global (L2 hit)", mix_kernel<2>);
  run("every fragment from global (L2 hit)", mix_kernel<1>);
  auto run2 = [&](const char* name, auto kern) {
    CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 131072));
    kern<<<256, 256, 131072>>>(g, out, iters, clk); CK(hipDeviceSynchronize());
    kern<<<256, 256, 131072>>>(g, out, iters, clk); CK(hipDeviceSynchronize());
    long long h[256]; CK(hipMemcpy(h, clk, sizeof(h), hipMemcpyDeviceToHost));
    double avg = 0; for (auto c : h) avg += c; avg /= 256;
    printf("%-44s %.1f cycles per MFMA\n", name, avg / ((double)iters * 32));
  };
  run2("LDS fragments, no DMA, vmcnt(0) per 32", dma_kernel<0>);
  run2("LDS fragments + 1 LDS-DMA KiB per 4 MFMA", dma_kernel<4>);
  run2("LDS fragments + 1 LDS-DMA KiB per 2 MFMA", dma_kernel<2>);
  run2("LDS fragments + 1 staged KiB per 4 MFMA", stage_kernel<4>);
  run2("LDS fragments + 1 staged KiB per 2 MFMA", stage_kernel<2>);
  return 0;
}
