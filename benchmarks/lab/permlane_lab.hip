// Lab probe (round 6): what v_permlane16_swap_b32 does to two registers, lane by lane - the epilogue of the 16x16x32 GEMM loop
// turns two 16 x 16 accumulator quads (lane = column c, row group g) into rows of 32 consecutive columns with it.
// build: hipcc --offload-arch=gfx950 -O2 -o benchmarks/lab/permlane_lab benchmarks/lab/permlane_lab.hip
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(unsigned* o) {
  unsigned x = 100 + threadIdx.x, y = 200 + threadIdx.x;
  auto r = __builtin_amdgcn_permlane16_swap(x, y, false, false);
  o[threadIdx.x] = r[0];
  o[64 + threadIdx.x] = r[1];
}
int main() {
  unsigned* d; unsigned h[128];
  hipMalloc(&d, sizeof(h));
  k<<<1, 64>>>(d);
  hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  for (int r = 0; r < 2; ++r) {
    printf("r[%d] by 16-lane group:", r);
    for (int g = 0; g < 4; ++g) printf("  [%u..%u]", h[64 * r + 16 * g], h[64 * r + 16 * g + 15]);
    printf("\n");
  }
  return 0;
}
