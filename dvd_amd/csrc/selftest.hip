// Hardware self-tests of the MFMA fragment layouts the GEMM and attention kernels rely on.
// Exact small-integer data, asymmetric operands (a transposed write cannot hide).
#include "common.h"
#include "mfma.h"

namespace dvd {

// out[0 .. 1024)    : D = A[32x16] . B[16x32]           (f16 MFMA)
// out[1024 .. 2048) : Y = Vt[32x32] . X, X = K[32x16] . Qt[16x32]  (accumulator as next B operand)
// out[2048 .. 3072) : D = A32[32x2] . B32[2x32]         (f32 MFMA)
__global__ void __launch_bounds__(64) mfma_selftest_kernel(const _Float16* __restrict__ A,   // [32][16]
                                                           const _Float16* __restrict__ B,   // [16][32]
                                                           const _Float16* __restrict__ Vt,  // [32 d][32 key]
                                                           float* __restrict__ out) {
  const int l = threadIdx.x, r = l & 31, h = l >> 5;
  half8 a, b;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    a[j] = A[r * 16 + 8 * h + j];
    b[j] = B[(8 * h + j) * 32 + r];
  }
  floatx16 d = {0};
  d = mfma32_f16(a, b, d);
#pragma unroll
  for (int i = 0; i < 16; ++i) out[cd_row(i, h) * 32 + r] = d[i];

  // chained: X[key][q] with K := A (rows = keys), Qt := B (cols = q)
  floatx16 y = {0};
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    half8 xb = pack_acc_f16(d, s);
    half8 va;
#pragma unroll
    for (int j = 0; j < 8; ++j) va[j] = Vt[r * 32 + 16 * s + 8 * (j >> 2) + 4 * h + (j & 3)];
    y = mfma32_f16(va, xb, y);
  }
#pragma unroll
  for (int i = 0; i < 16; ++i) out[1024 + cd_row(i, h) * 32 + r] = y[i];

  // f32 MFMA: A32[i][k] = A[i][k] (k < 2), B32[k][j] = B[k][j]
  floatx16 e = {0};
  e = mfma32_f32((float)A[r * 16 + h], (float)B[h * 32 + r], e);
#pragma unroll
  for (int i = 0; i < 16; ++i) out[2048 + cd_row(i, h) * 32 + r] = e[i];
}

}  // namespace dvd

extern "C" int dvd_selftest_mfma(const void* a16, const void* b16, const void* vt16, float* out3072, void* stream) {
  DVD_REQUIRE(a16 && b16 && vt16 && out3072, "selftest_mfma: null pointer");
  dvd::mfma_selftest_kernel<<<1, 64, 0, (hipStream_t)stream>>>((const _Float16*)a16, (const _Float16*)b16,
                                                              (const _Float16*)vt16, out3072);
  return dvd::check_launch("selftest_mfma");
}
