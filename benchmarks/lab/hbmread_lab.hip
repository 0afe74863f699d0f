// Round 6: how fast can ONE CU read HBM-resident data that nothing else touches - and does it depend on the bytes it keeps in
// flight, on the access width, or on how many other CUs read at the same time?  (The residual epilogue of gemm_nt_t384_kernel
// reads 393 KB per tile at 13.5-16 B/clk per CU with 6 KB per wave in flight, whether 32 or 256 workgroups run: profiles/
// r6_gemm_res_epilogue_probe.txt.)  Every workgroup (8 waves = one CU's worth at <= 256 workgroups) streams its own private
// stripe of a 4-GiB buffer once (no reuse: misses L2 and the Infinity Cache) in one of three forms:
//   dword   global_load_dword   per lane: 256 B per wave instruction (two 128-B row segments: the epilogue's access)
//   x4      global_load_dwordx4 per lane: 1 KiB per wave instruction
//   dma     global_load_lds_dwordx4:      1 KiB per wave instruction into LDS (no register window)
// with D instructions per wave issued back to back before the wave waits for the older batch (double-buffered: between D and 2 D
// in flight).   hipcc --offload-arch=gfx950 -O3 -std=c++17 hbmread_lab.hip -o hbmread_lab && ./hbmread_lab
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

typedef float floatx4 __attribute__((ext_vector_type(4)));
constexpr size_t STRIPE = 12u << 20;        // bytes per workgroup (12 MiB: 256 workgroups = 3 GiB)

template <int D, int W>   // W = 1: dword, 4: dwordx4
__global__ void __launch_bounds__(512) reg_kernel(const char* __restrict__ g, float* __restrict__ sink, long long* __restrict__ clk) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const char* base = g + (size_t)blockIdx.x * STRIPE;
  constexpr int BYTES = 64 * 4 * W;                    // per wave instruction
  constexpr size_t PER_WAVE = STRIPE / 8;
  const char* p = base + (size_t)wave * PER_WAVE + lane * 4 * W;
  const int nbatch = (int)(PER_WAVE / ((size_t)BYTES * D));
  float acc = 0.f;
  const long long t0 = __builtin_readcyclecounter();
  floatx4 cur[D], nxt[D];
  auto load = [&](const char* q, floatx4& dst) {
    if (W == 4) dst = *(const floatx4*)q;
    else { dst[0] = *(const float*)q; }
  };
#pragma unroll
  for (int j = 0; j < D; ++j) load(p + (size_t)j * BYTES, cur[j]);
  for (int b = 1; b < nbatch; ++b) {
    const char* q = p + (size_t)b * BYTES * D;
#pragma unroll
    for (int j = 0; j < D; ++j) load(q + (size_t)j * BYTES, nxt[j]);
#pragma unroll
    for (int j = 0; j < D; ++j) { acc += cur[j][0]; if (W == 4) acc += cur[j][3]; }
#pragma unroll
    for (int j = 0; j < D; ++j) cur[j] = nxt[j];
  }
#pragma unroll
  for (int j = 0; j < D; ++j) acc += cur[j][0];
  const long long t1 = __builtin_readcyclecounter();
  if (acc == 123.456f) sink[0] = acc;
  if (threadIdx.x == 0) clk[blockIdx.x] = t1 - t0;
}

template <int D>
__global__ void __launch_bounds__(512) dma_kernel(const char* __restrict__ g, long long* __restrict__ clk) {
  extern __shared__ __attribute__((aligned(1024))) char lds[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(size_t)lds) + wave * (2 * D * 1024);
  constexpr size_t PER_WAVE = STRIPE / 8;
  const char* base = g + (size_t)blockIdx.x * STRIPE + (size_t)wave * PER_WAVE;
  const unsigned voff = lane * 16;
  const int nbatch = (int)(PER_WAVE / (1024u * D));
  const long long t0 = __builtin_readcyclecounter();
  for (int b = 0; b < nbatch; ++b) {
    const char* q = (const char*)__builtin_amdgcn_readfirstlane((unsigned)((size_t)(base + (size_t)b * 1024 * D) & 0xffffffffu)) ;
    (void)q;
    const char* qq = base + (size_t)b * 1024 * D;
    const unsigned long long v = (unsigned long long)qq;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
    const char* qs = (const char*)(((unsigned long long)hi << 32) | lo);
    const unsigned slot = lds0 + (b & 1) * D * 1024;
#pragma unroll
    for (int j = 0; j < D; ++j)
      asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %0 offset:0" :: "s"(qs + j * 1024), "s"(slot + j * 1024), "v"(voff) : "memory");
    // the previous batch has landed (this batch's D pieces stay in flight)
    if (D == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else if (D == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  const long long t1 = __builtin_readcyclecounter();
  if (threadIdx.x == 0) clk[blockIdx.x] = t1 - t0;
}

// The residual epilogue's ADDRESS PATTERN (f32 stream [M, 1536], a workgroup = a 384 x 256 tile, wave (wr, wc) = its 96 x 128
// block at rows 96 wr, columns 128 wc), D instructions per wave issued before the older batch is waited for:
//   PAT 0  the epilogue's: one instruction = two 128-byte row segments (rows rho, rho + 8 of one 32-column block), order
//          (m, n, u, k) - a row's four neighbouring segments are touched ~2 steps apart
//   PAT 1  row-major pieces: one instruction (dwordx4) = two 512-byte row runs (rows rho, rho + 8, all four column blocks)
// Both read every byte of the tile's 96 x 128 x 4 wave blocks exactly once; tiles walk down the matrix.
template <int D, int PAT>
__global__ void __launch_bounds__(512) tile_kernel(const char* __restrict__ g, float* __restrict__ sink, long long* __restrict__ clk, int tiles) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, wr = wave >> 1, wc = wave & 1;
  constexpr int LD = 1536 * 4;                                   // bytes per matrix row
  float acc = 0.f;
  const long long t0 = __builtin_readcyclecounter();
  for (int t = 0; t < tiles; ++t) {
    // tile (tm, tn): 6 column tiles per row of tiles; workgroup b takes tiles b, b + gridDim.x, ...
    const int vid = blockIdx.x + t * gridDim.x, tm = vid / 6, tn = vid % 6;
    const char* wb = g + ((size_t)tm * 384 + 96 * wr) * LD + ((size_t)tn * 256 + 128 * wc) * 4;
    if (PAT == 0) {
      const char* lp = wb + (size_t)(8 * (lane >> 5)) * LD + (lane & 31) * 4;
      float cur[D], nxt[D];
      auto addr = [&](int i) {        // instruction i of 192: (m, n, u, k)
        const int k = i & 7, u = (i >> 3) & 1, n = (i >> 4) & 3, m = i >> 6;
        return lp + (size_t)(32 * m + 16 * u + (k & 3) + 4 * (k >> 2)) * LD + 128 * n;
      };
#pragma unroll
      for (int j = 0; j < D; ++j) cur[j] = *(const float*)addr(j);
      for (int b = 1; b < 192 / D; ++b) {
#pragma unroll
        for (int j = 0; j < D; ++j) nxt[j] = *(const float*)addr(b * D + j);
#pragma unroll
        for (int j = 0; j < D; ++j) acc += cur[j];
#pragma unroll
        for (int j = 0; j < D; ++j) cur[j] = nxt[j];
      }
#pragma unroll
      for (int j = 0; j < D; ++j) acc += cur[j];
    } else {
      const char* lp = wb + (size_t)(8 * (lane >> 5)) * LD + (lane & 31) * 16;
      floatx4 cur[D], nxt[D];
      auto addr = [&](int i) {        // piece i of 48: (m, i'): rows 32 m + 16 (i' >> 3) + 4 ((i' >> 2) & 1) + (i' & 3), + 8 for the upper half
        const int ip = i & 15, m = i >> 4;
        return lp + (size_t)(32 * m + 16 * (ip >> 3) + 4 * ((ip >> 2) & 1) + (ip & 3)) * LD;
      };
#pragma unroll
      for (int j = 0; j < D; ++j) cur[j] = *(const floatx4*)addr(j);
      for (int b = 1; b < 48 / D; ++b) {
#pragma unroll
        for (int j = 0; j < D; ++j) nxt[j] = *(const floatx4*)addr(b * D + j);
#pragma unroll
        for (int j = 0; j < D; ++j) acc += cur[j][0] + cur[j][3];
#pragma unroll
        for (int j = 0; j < D; ++j) cur[j] = nxt[j];
      }
#pragma unroll
      for (int j = 0; j < D; ++j) acc += cur[j][0];
    }
  }
  const long long t1 = __builtin_readcyclecounter();
  if (acc == 123.456f) sink[0] = acc;
  if (threadIdx.x == 0) clk[blockIdx.x] = t1 - t0;
}

template <typename F> static void run(const char* what, int d, int wgs, F launch, long long* clk, hipEvent_t e0, hipEvent_t e1) {
  launch(); CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0)); launch(); CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  long long h[256]; CK(hipMemcpy(h, clk, wgs * 8, hipMemcpyDeviceToHost));
  double cyc = 0; for (int i = 0; i < wgs; ++i) cyc += (double)h[i]; cyc /= wgs;
  const double bytes = (double)STRIPE;
  printf("%-6s D=%2d  %3d workgroups: %7.3f ms  %6.0f GB/s chip  %5.1f B/clk per CU (s_memrealtime-free cycle counter: %0.0f kcycles per stripe)\n",
         what, d, wgs, ms, bytes * wgs / ms / 1e6, bytes / cyc, cyc / 1e3);
}

int main() {
  char* g; float* sink; long long* clk;
  CK(hipMalloc(&g, STRIPE * 256)); CK(hipMemset(g, 0, STRIPE * 256));
  CK(hipMalloc(&sink, 64)); CK(hipMalloc(&clk, 256 * 8));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  CK(hipFuncSetAttribute((const void*)dma_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  CK(hipFuncSetAttribute((const void*)dma_kernel<8>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  CK(hipFuncSetAttribute((const void*)dma_kernel<10>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  for (int wgs : {256, 32}) {
#define REG(D_, W_) run(W_ == 1 ? "dword" : "x4", D_, wgs, [&]() { reg_kernel<D_, W_><<<wgs, 512>>>(g, sink, clk); }, clk, e0, e1)
    REG(8, 1); REG(16, 1); REG(24, 1); REG(48, 1);
    REG(2, 4); REG(4, 4); REG(8, 4); REG(16, 4);
#undef REG
    run("dma", 4, wgs, [&]() { dma_kernel<4><<<wgs, 512, 8 * 2 * 4 * 1024>>>(g, clk); }, clk, e0, e1);
    run("dma", 8, wgs, [&]() { dma_kernel<8><<<wgs, 512, 8 * 2 * 8 * 1024>>>(g, clk); }, clk, e0, e1);
    run("dma", 10, wgs, [&]() { dma_kernel<10><<<wgs, 512, 8 * 2 * 10 * 1024>>>(g, clk); }, clk, e0, e1);
  }
  // the tile patterns: the buffer seen as an f32 matrix [M, 1536]; 256 (32) workgroups x `tiles` tiles of 393 KB each
  {
    const size_t total = STRIPE * 256;
    const int tiles_all = (int)(total / (384 * 1536 * 4) * 6);          // tiles in the buffer
    for (int wgs : {256, 32}) {
      const int tiles = tiles_all / 256;                                 // per workgroup (the 32-workgroup run reads an eighth)
      auto rep = [&](const char* what, int d, auto launch) {
        launch(); CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0)); launch(); CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("%-34s D=%2d  %3d workgroups x %d tiles: %7.3f ms  %6.0f GB/s chip  %5.1f GB/s per CU\n", what, d, wgs, tiles, ms,
               393216.0 * tiles * wgs / ms / 1e6, 393216.0 * tiles / ms / 1e6);
      };
      rep("epilogue pattern (2 x 128 B)", 8, [&]() { tile_kernel<8, 0><<<wgs, 512>>>(g, sink, clk, tiles); });
      rep("epilogue pattern (2 x 128 B)", 24, [&]() { tile_kernel<24, 0><<<wgs, 512>>>(g, sink, clk, tiles); });
      rep("epilogue pattern (2 x 128 B)", 48, [&]() { tile_kernel<48, 0><<<wgs, 512>>>(g, sink, clk, tiles); });
      rep("row-major pieces (2 x 512 B)", 4, [&]() { tile_kernel<4, 1><<<wgs, 512>>>(g, sink, clk, tiles); });
      rep("row-major pieces (2 x 512 B)", 8, [&]() { tile_kernel<8, 1><<<wgs, 512>>>(g, sink, clk, tiles); });
      rep("row-major pieces (2 x 512 B)", 12, [&]() { tile_kernel<12, 1><<<wgs, 512>>>(g, sink, clk, tiles); });
    }
  }
  return 0;
}
