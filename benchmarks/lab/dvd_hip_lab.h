/* Extra entry points of the LAB build of the library (make -C dvd_amd/csrc lab -> benchmarks/lab/libdvd_hip_lab.so,
 * compiled with -DDVD_LAB).  The lab library is the product library plus the measured-and-rejected experiment kernels
 * and the DVD_* environment switches that select them; it is used by benchmarks/ and by tests/tools/lab_checks.py only.
 * The product library (dvd_amd/libdvd_hip.so, include/dvd_hip.h) reads no environment variable. */
#ifndef DVD_HIP_LAB_H
#define DVD_HIP_LAB_H
#ifdef __cplusplus
extern "C" {
#endif
/* DVD_GEMM_DEBUG=3: device buffer [workgroups*8*4] u64 receiving per-wave s_memtime stamps of the large-tile GEMM */
int dvd_gemm_debug_stamps(void* dev_u64);
/* DVD_ATTN_DEBUG=1: device buffer [workgroups*4*5] u64 receiving per-wave phase times of the attention kernels */
int dvd_attn_debug_stamps(void* dev_u64);
#ifdef __cplusplus
}
#endif
#endif
