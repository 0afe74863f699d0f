// Per-CU load-path throughput on gfx950: how many bytes/clk one CU can pull from L2 (all CUs hammering the same
// 64-KiB tiles, as the attention kernel does) through
//   (a) global_load_lds_dwordx4   (LDS-DMA)
//   (b) global_load_dwordx4 into VGPRs
//   (c) (b) + ds_write_b128 into LDS (register-staged tile fill)
// One workgroup of 4 waves per CU (80 KiB LDS forces 1 WG/CU... 2 at most), grid = 256.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 ldpath_lab.hip -o ldpath_lab && ./ldpath_lab
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int TILE = 65536;      // bytes per "tile"
constexpr int NT = 256;          // tiles streamed per workgroup (region = 16 MiB, L2/MALL resident after the first pass)

__device__ __forceinline__ void glds8(const char* gbase, const unsigned (&voff)[8], unsigned lds) {
  unsigned keep;
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
}

template <int MODE, int SHIFT>
__global__ void __launch_bounds__(256) ld_kernel(const char* __restrict__ g, unsigned* __restrict__ sink, long long* __restrict__ clk, int active) {
  extern __shared__ __attribute__((aligned(1024))) char lds[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  unsigned acc = 0;
  unsigned voff[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) voff[j] = (wave * 16 + j) * 1024 + lane * 16;
  const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(size_t)lds);
  __syncthreads();
  const long long t0 = __builtin_readcyclecounter();
  for (int t = 0; t < NT; ++t) {
    const char* gt = g + (size_t)((t + (blockIdx.x >> SHIFT)) % NT) * TILE;
    if (wave >= active) {
    } else if (MODE == 3) {          // half the tile by LDS-DMA, the other half through registers + ds_write
      glds8(gt, voff, lds0 + (wave * 16) * 1024);
      u32x4 v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = *reinterpret_cast<const u32x4*>(gt + (wave * 16 + 8 + j) * 1024 + lane * 16);
#pragma unroll
      for (int j = 0; j < 8; ++j) *reinterpret_cast<u32x4*>(lds + (wave * 16 + 8 + j) * 1024 + lane * 16) = v[j];
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else if (MODE == 0) {
      glds8(gt, voff, lds0 + (wave * 16) * 1024);
      glds8(gt + 8192, voff, lds0 + (wave * 16 + 8) * 1024);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else {
      u32x4 v[16];
#pragma unroll
      for (int j = 0; j < 16; ++j) v[j] = *reinterpret_cast<const u32x4*>(gt + (wave * 16 + j) * 1024 + lane * 16);
      if (MODE == 1) {
#pragma unroll
        for (int j = 0; j < 16; ++j) acc ^= v[j][0] ^ v[j][1] ^ v[j][2] ^ v[j][3];
      } else {
#pragma unroll
        for (int j = 0; j < 16; ++j) *reinterpret_cast<u32x4*>(lds + (wave * 16 + j) * 1024 + lane * 16) = v[j];
      }
    }
    if (MODE != 1) {
      __syncthreads();
      acc ^= *reinterpret_cast<unsigned*>(lds + ((lane * 4 + t * 64) & (TILE - 4)));
      __syncthreads();
    }
  }
  const long long t1 = __builtin_readcyclecounter();
  sink[blockIdx.x * 256 + threadIdx.x] = acc;
  if (threadIdx.x == 0) clk[blockIdx.x] = t1 - t0;
}

int main() {
  char* g; unsigned* sink; long long* clk;
  CK(hipMalloc(&g, (size_t)NT * TILE)); CK(hipMemset(g, 1, (size_t)NT * TILE));
  CK(hipMalloc(&sink, 256 * 256 * 4)); CK(hipMalloc(&clk, 256 * 8));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const char* names[4] = {"LDS-DMA global_load_lds_dwordx4", "global_load_dwordx4 -> VGPR", "global_load_dwordx4 -> VGPR -> ds_write_b128", "half LDS-DMA + half VGPR -> ds_write"};
  constexpr int SH = 10;   // all workgroups stream the same tile sequence (L2 hits), like the q-blocks of one head
  for (int active : {4, 2, 1})
  for (int wgs : {256}) {
    for (int mode = 0; mode < 4; ++mode) {
      auto launch = [&]() {
        if (mode == 0) ld_kernel<0, SH><<<wgs, 256, TILE>>>(g, sink, clk, active);
        if (mode == 1) ld_kernel<1, SH><<<wgs, 256, TILE>>>(g, sink, clk, active);
        if (mode == 2) ld_kernel<2, SH><<<wgs, 256, TILE>>>(g, sink, clk, active);
        if (mode == 3) ld_kernel<3, SH><<<wgs, 256, TILE>>>(g, sink, clk, active);
      };
      launch(); CK(hipDeviceSynchronize());
      CK(hipEventRecord(e0)); launch(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      std::vector<long long> h(wgs > 256 ? 256 : wgs);
      CK(hipMemcpy(h.data(), clk, h.size() * 8, hipMemcpyDeviceToHost));
      double avg = 0; for (auto c : h) avg += c; avg /= h.size();
      double bytes = (double)NT * TILE;
      // readcyclecounter = s_memtime: 100 MHz constant clock on gfx9?  report both wall-clock and counter based
      printf("%-48s active waves %d wgs=%3d  %.3f ms  per-WG %.1f GB/s  chip %.1f TB/s  (memtime ticks/tile %.0f)\n", names[mode], active, wgs, ms,
             bytes * active / 4 / ms * 1e-6, bytes * active / 4 * wgs / ms * 1e-9, avg / NT);
    }
  }
  return 0;
}
