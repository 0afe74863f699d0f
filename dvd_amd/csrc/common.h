// Shared host/device helpers for libdvd_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>

#include <atomic>

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

// One-time per-DEVICE setup (hipFuncSetAttribute is a per-device property): one bit per device, lock-free.  The guarded
// calls are idempotent, so two threads racing on a device's first launch both make them and both are right.
struct DeviceOnce {
  std::atomic<unsigned long long> mask{0};
  static unsigned long long current_bit() {
    int dev = 0;
    (void)hipGetDevice(&dev);
    return 1ull << (dev & 63);
  }
  bool need(unsigned long long b) const { return !(mask.load(std::memory_order_acquire) & b); }
  void done(unsigned long long b) { mask.fetch_or(b, std::memory_order_release); }
};

static inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }

constexpr int kWave = 64;

// non-contracted fp32 arithmetic (bit-compatible with separately rounded tensor ops)
__device__ __forceinline__ float mul_rn(float a, float b) { return __fmul_rn(a, b); }
__device__ __forceinline__ float add_rn(float a, float b) { return __fadd_rn(a, b); }
__device__ __forceinline__ float sub_rn(float a, float b) { return __fsub_rn(a, b); }
__device__ __forceinline__ float div_rn(float a, float b) { return __fdiv_rn(a, b); }

}  // namespace dvd
