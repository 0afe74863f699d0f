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
/* the streaming ceiling of the drop-in grid_sample's access pattern: same tiles, same 2 + c + c plane streams at 16 bytes per
 * lane, no gather (source address = output address; hin == h, win == w) - warp.hip, benchmarks/warp_time.py */
int dvd_lab_stream_copy_planes(const float* src, const float* grid, float* out, int n, int c, int h, int w, int nontemporal,
                               int tile_w /* 32: the gather kernel's 32 x 32 tiles; 64 / 128 / 256: wider, flatter tiles */, void* stream);
#ifdef __cplusplus
}
#endif
#endif
