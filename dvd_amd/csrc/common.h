// Shared host/device helpers for libdvd_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>

#include "../../include/dvd_hip.h"

namespace dvd {

void set_error(const char* fmt, ...);

inline int check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    set_error("%s: %s", what, hipGetErrorString(e));
    return DVD_E_LAUNCH;
  }
  return DVD_OK;
}

#define DVD_REQUIRE(cond, ...)          \
  do {                                  \
    if (!(cond)) {                      \
      dvd::set_error(__VA_ARGS__);      \
      return DVD_E_ARG;                 \
    }                                   \
  } while (0)

static inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }

constexpr int kWave = 64;

// non-contracted fp32 arithmetic (bit-compatible with separately rounded tensor ops)
__device__ __forceinline__ float mul_rn(float a, float b) { return __fmul_rn(a, b); }
__device__ __forceinline__ float add_rn(float a, float b) { return __fadd_rn(a, b); }
__device__ __forceinline__ float sub_rn(float a, float b) { return __fsub_rn(a, b); }
__device__ __forceinline__ float div_rn(float a, float b) { return __fdiv_rn(a, b); }

}  // namespace dvd
