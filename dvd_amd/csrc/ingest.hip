// Image ingest of the sampling path: what Doc_benchmark.__getitem__ does on the CPU
// (datasets/doc_dataset/doc_benchmark.py:75-97) after the file is decoded:
//   img_ori = cv2.imread(path)[:, :, ::-1]             BGR -> RGB, uint8 [H,W,3]
//   img     = cv2.resize(img_ori, (512, 512))          INTER_LINEAR on uint8
//   source_image = ArrayToTensor(img) / 255.           float32 [3,512,512] in 0..1
// cv2 (opencv-python, un-pinned in the reference's requirements.txt) is absent from this image, so the resize
// restates OpenCV's published 8-bit INTER_LINEAR algorithm (imgproc/src/resize.cpp: half-pixel centres, 11-bit
// fixed-point coefficients rounded to nearest-even, horizontal pass into int32, vertical pass
// ((b0*(S0>>4))>>16) + ((b1*(S1>>4))>>16) + 2) >> 2; an exactly-2x downscale takes cv::resize's INTER_AREA switch).  Integer arithmetic throughout: bit-exact against
// oracle/ingest_oracle.py; parity with a real cv2 build is UNPINNED.
#include "common.h"

namespace dvd {

struct ResizeAxis {   // per destination index: source index and the two 11-bit coefficients
  int s;
  short a0, a1;
};

// coefficients of one axis; launched once per axis (<= 512 threads of work)
__global__ void resize_axis_kernel(ResizeAxis* __restrict__ tab, int ssize, int dsize) {
  const int d = blockIdx.x * blockDim.x + threadIdx.x;
  if (d >= dsize) return;
  const double scale = (double)ssize / (double)dsize;
  float f = (float)(((double)d + 0.5) * scale - 0.5);
  int s = (int)floorf(f);
  f -= (float)s;
  if (s < 0) { f = 0.f; s = 0; }
  if (s >= ssize - 1) { f = 0.f; s = ssize - 1; }
  // saturate_cast<short>(float * 2048): round to nearest, ties to even (cvRound)
  tab[d].s = s;
  tab[d].a0 = (short)__float2int_rn((1.f - f) * 2048.f);
  tab[d].a1 = (short)__float2int_rn(f * 2048.f);
}

__global__ void __launch_bounds__(256) ingest_resize_kernel(const uint8_t* __restrict__ src, int h, int w, int swap_rb,
                                                            const ResizeAxis* __restrict__ tx,
                                                            const ResizeAxis* __restrict__ ty, float* __restrict__ out,
                                                            int osize) {
  const int dx = blockIdx.x * blockDim.x + threadIdx.x, dy = blockIdx.y;
  if (dx >= osize) return;
  if (h == 2 * osize && w == 2 * osize) {
    // exactly 2x in both axes: cv::resize switches INTER_LINEAR to INTER_AREA, whose 8-bit fast path is the rounded
    // mean of the 2x2 block (resizeAreaFast_: (S00 + S01 + S10 + S11 + 2) >> 2)
    const uint8_t* r0 = src + ((size_t)(2 * dy) * w + 2 * dx) * 3;
    const uint8_t* r1 = r0 + (size_t)w * 3;
#pragma unroll
    for (int ch = 0; ch < 3; ++ch) {
      const int sc = swap_rb ? 2 - ch : ch;
      const int u = ((int)r0[sc] + (int)r0[3 + sc] + (int)r1[sc] + (int)r1[3 + sc] + 2) >> 2;
      out[((size_t)ch * osize + dy) * osize + dx] = __fdiv_rn((float)u, 255.f);
    }
    return;
  }
  const ResizeAxis ax = tx[dx], ay = ty[dy];
  const int x0 = ax.s, x1 = min(ax.s + 1, w - 1), y0 = ay.s, y1 = min(ay.s + 1, h - 1);
  const uint8_t* r0 = src + ((size_t)y0 * w) * 3;
  const uint8_t* r1 = src + ((size_t)y1 * w) * 3;
#pragma unroll
  for (int ch = 0; ch < 3; ++ch) {
    const int sc = swap_rb ? 2 - ch : ch;
    const int S0 = (int)r0[(size_t)x0 * 3 + sc] * ax.a0 + (int)r0[(size_t)x1 * 3 + sc] * ax.a1;   // horizontal pass, row y0
    const int S1 = (int)r1[(size_t)x0 * 3 + sc] * ax.a0 + (int)r1[(size_t)x1 * 3 + sc] * ax.a1;   // row y1
    const int v = ((((int)ay.a0 * (S0 >> 4)) >> 16) + (((int)ay.a1 * (S1 >> 4)) >> 16) + 2) >> 2;
    const int u = min(max(v, 0), 255);
    out[((size_t)ch * osize + dy) * osize + dx] = __fdiv_rn((float)u, 255.f);
  }
}

__global__ void __launch_bounds__(256) swap_rb_kernel(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst, long px) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= px) return;
  const uint8_t b = src[i * 3], g = src[i * 3 + 1], r = src[i * 3 + 2];
  dst[i * 3] = r; dst[i * 3 + 1] = g; dst[i * 3 + 2] = b;
}

}  // namespace dvd

using namespace dvd;

extern "C" long dvd_ingest_scratch_bytes(int out_size) { return (long)out_size * 2 * (long)sizeof(ResizeAxis); }

extern "C" int dvd_ingest_u8(const uint8_t* src_hwc, int h, int w, int swap_rb, float* y_chw, int out_size,
                             uint8_t* rgb_hwc_out, void* scratch, void* stream) {
  DVD_REQUIRE(src_hwc && y_chw && scratch, "ingest_u8: null pointer");
  DVD_REQUIRE(h >= 1 && w >= 1 && out_size >= 1 && out_size <= 65535, "ingest_u8: bad shape %dx%d -> %d", h, w, out_size);
  hipStream_t st = (hipStream_t)stream;
  ResizeAxis* tx = (ResizeAxis*)scratch;
  ResizeAxis* ty = tx + out_size;
  resize_axis_kernel<<<cdiv(out_size, 256), 256, 0, st>>>(tx, w, out_size);
  resize_axis_kernel<<<cdiv(out_size, 256), 256, 0, st>>>(ty, h, out_size);
  dim3 grd(cdiv(out_size, 256), out_size);
  ingest_resize_kernel<<<grd, 256, 0, st>>>(src_hwc, h, w, swap_rb ? 1 : 0, tx, ty, y_chw, out_size);
  if (rgb_hwc_out) {
    const long px = (long)h * w;
    if (swap_rb) {
      swap_rb_kernel<<<cdiv(px, 256), 256, 0, st>>>(src_hwc, rgb_hwc_out, px);
    } else if (rgb_hwc_out != src_hwc) {
      if (hipMemcpyAsync(rgb_hwc_out, src_hwc, (size_t)px * 3, hipMemcpyDeviceToDevice, st) != hipSuccess) {
        set_error("ingest_u8: device copy failed");
        return DVD_E_LAUNCH;
      }
    }
  }
  return check_launch("ingest_u8");
}
