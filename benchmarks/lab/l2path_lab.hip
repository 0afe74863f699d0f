// Round 5: is the ~33 B/clk L2 -> LDS rate of ldpath_lab a PER-CU limit or the XCD's L2 shared by its 32 CUs?
// Every workgroup (8 waves, one per CU when wgs <= 256) streams L2-resident 40-KiB half slabs into a 4-deep LDS ring by
// LDS-DMA with a counted vmcnt (5 pieces per wave and half slab, the round-5 GEMM's shape) and one barrier per half slab;
// the number of ACTIVE workgroups is swept (hardware dispatch is round-robin over the 8 XCDs, so `wgs` active workgroups
// = wgs / 8 CUs per XCD).  If the per-CU rate rises when fewer CUs pull, the limit is shared; if not, it is the CU's own path.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 l2path_lab.hip -o l2path_lab && ./l2path_lab
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

constexpr int HALF = 40960;      // bytes per half slab (384 x 32 + 256 x 32 halfs)
constexpr int NSLAB = 384;       // distinct half slabs in the streamed region (15 MiB: L2 / Infinity-Cache resident)
constexpr int ITER = 2048;       // half slabs streamed per workgroup

__device__ __forceinline__ void piece(const char* gbase, unsigned voff, unsigned lds) {
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %0" :: "s"(gbase), "s"(lds), "v"(voff) : "memory");
}

template <int PIECES>   // pieces per wave and half slab (5 = 40 KiB by 8 waves; 4 = 32 KiB)
__global__ void __launch_bounds__(512) stream_kernel(const char* __restrict__ g, long long* __restrict__ clk, int spread) {
  extern __shared__ __attribute__((aligned(1024))) char lds[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(size_t)lds);
  unsigned voff[PIECES];
#pragma unroll
  for (int j = 0; j < PIECES; ++j) voff[j] = (wave * PIECES + j) * 1024 + lane * 16;
  __syncthreads();
  const long long t0 = __builtin_readcyclecounter();
  // workgroups of one XCD walk the SAME slab sequence (an operand panel shared in that XCD's L2), shifted per XCD
  const int xcd = blockIdx.x & 7;
  for (int t = 0; t < ITER; ++t) {
    const char* gt = g + (size_t)((t + xcd * 37 + (spread ? (blockIdx.x >> 3) * 3 : 0)) % NSLAB) * HALF;
    const unsigned slot = lds0 + (t & 3) * HALF + wave * PIECES * 1024;
#pragma unroll
    for (int j = 0; j < PIECES; ++j) piece(gt, voff[j], slot + j * 1024);
    // counted wait: the pieces of half slab t - 2 have landed (2 x PIECES younger ones stay in flight)
    if (PIECES == 5) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    __builtin_amdgcn_s_barrier();
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  const long long t1 = __builtin_readcyclecounter();
  if (threadIdx.x == 0) clk[blockIdx.x] = t1 - t0;
}

int main() {
  char* g; long long* clk;
  CK(hipMalloc(&g, (size_t)NSLAB * HALF)); CK(hipMemset(g, 1, (size_t)NSLAB * HALF));
  CK(hipMalloc(&clk, 1024 * 8));
  CK(hipFuncSetAttribute((const void*)stream_kernel<5>, hipFuncAttributeMaxDynamicSharedMemorySize, 4 * HALF));
  CK(hipFuncSetAttribute((const void*)stream_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize, 4 * HALF));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int spread : {0, 1})
  for (int pieces : {5, 4})
  for (int wgs : {256, 128, 64, 32, 16, 8}) {
    auto launch = [&]() {
      if (pieces == 5) stream_kernel<5><<<wgs, 512, 4 * HALF>>>(g, clk, spread);
      else stream_kernel<4><<<wgs, 512, 4 * HALF>>>(g, clk, spread);
    };
    launch(); CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0)); launch(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<long long> h(wgs);
    CK(hipMemcpy(h.data(), clk, h.size() * 8, hipMemcpyDeviceToHost));
    double avg = 0; for (auto c : h) avg += c; avg /= h.size();
    const double bytes = (double)ITER * pieces * 8 * 1024;
    printf("pieces %d  %s  active WGs %3d (%2d per XCD): %.3f ms  per-CU %.1f GB/s = %.1f B/clk (s_memtime cycles)  chip %.2f TB/s  clock %.2f GHz\n",
           pieces, spread ? "per-CU distinct slabs" : "XCD-shared slabs   ", wgs, wgs / 8, ms, bytes / ms * 1e-6, bytes / avg,
           bytes * wgs / ms * 1e-9, avg / ms * 1e-6);
  }
  return 0;
}
