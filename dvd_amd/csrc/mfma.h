// MFMA fragment helpers for gfx950 (wave64).  Layouts (guide section 3, verified on hardware by
// dvd_selftest_mfma / tests/test_gpu_selftest.py):
//   v_mfma_f32_32x32x16_f16:  lane l, r = l & 31, h = l >> 5
//     A fragment element j (0..7) = A[row r][k = 8h + j]
//     B fragment element j        = B[k = 8h + j][col r]
//     C/D register i (0..15)      = D[row (i & 3) + 8 (i >> 2) + 4 h][col r]
//   v_mfma_f32_32x32x2_f32:  A = A[row r][k = h], B = B[k = h][col r], same C/D map.
#pragma once
#include <hip/hip_runtime.h>

namespace dvd {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef _Float16 half2v __attribute__((ext_vector_type(2)));
typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ floatx16 mfma32_f16(half8 a, half8 b, floatx16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ floatx16 mfma32_f32(float a, float b, floatx16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}

// row of C/D register i for lane-half h
__device__ __forceinline__ int cd_row(int i, int h) { return (i & 3) + 8 * (i >> 2) + 4 * h; }

// Pack 8 consecutive accumulator registers [8s, 8s+8) of a 32x32 f32 tile into the f16 fragment of
// k-step s of a following 32x32x16 MFMA that contracts over the tile's ROW index.  Fragment element j
// of lane-half h then carries tile row 16 s + 8 (j >> 2) + 4 h + (j & 3).
__device__ __forceinline__ half8 pack_acc_f16(const floatx16& x, int s) {
  half8 r;
#pragma unroll
  for (int j = 0; j < 8; ++j) r[j] = (_Float16)x[8 * s + j];
  return r;
}

}  // namespace dvd
