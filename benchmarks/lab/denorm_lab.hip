// Does v_mfma_f32_32x32x16_f16 honour f16 subnormal inputs on gfx950, or flush them?
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float floatx16 __attribute__((ext_vector_type(16)));
__global__ void k(float* out, float bval) {
  half8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (_Float16)1024.f; b[i] = (_Float16)bval; }
  floatx16 c;
  for (int i = 0; i < 16; ++i) c[i] = 0.f;
  c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
  if (threadIdx.x == 0) { out[0] = c[0]; out[1] = (float)b[0]; }
}
int main() {
  float* d; hipMalloc(&d, 16);
  for (float bv : {9.5367431640625e-07f /* 2^-20 */, 5.9604644775390625e-08f /* 2^-24 */, 6.103515625e-05f /* 2^-14 normal */}) {
    k<<<1, 64>>>(d, bv);
    float h[2]; hipMemcpy(h, d, 8, hipMemcpyDeviceToHost);
    printf("b = %g (as f16 %g): mfma sum over k=16 of 1024*b = %g, expected %g\n", bv, h[1], h[0], 16.0 * 1024.0 * bv);
  }
  return 0;
}
