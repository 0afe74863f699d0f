// Fused scheduler step: DDIM (idf/gaussian_diffusion.py:434-438,470-489) or DDPM posterior
// (idf/gaussian_diffusion.py:270-292) + the next step's feature-warp grid (:622), and the
// hypothesis mean + clamp (:639-640).  Elementwise, latency-bound (2.65 MB per sample at G=288).
//
// Arithmetic uses explicitly non-contracted fp32 ops in the reference's operation order so
// results are bit-identical to the separately-rounded tensor ops of the CPU path.
#include "common.h"

namespace dvd {

__global__ void __launch_bounds__(256) sched_step_kernel(dvd_sched_coef c, const float* __restrict__ x_t,
                                                         const float* __restrict__ x0,
                                                         const float* __restrict__ noise,
                                                         float* __restrict__ x_prev,
                                                         float* __restrict__ next_grid, long total, int g) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  const float xt = x_t[idx], p0 = x0[idx];
  float mean;
  if (c.kind == 0) {
    // eps = (sqrt_recip * x_t - x0) / sqrt_recipm1
    float eps = div_rn(sub_rn(mul_rn(c.c_recip, xt), p0), c.c_recipm1);
    // x0 * sqrt(abar_prev) + sqrt(1 - abar_prev - sigma^2) * eps
    mean = add_rn(mul_rn(p0, c.sqrt_abar_prev), mul_rn(c.dir_coef, eps));
  } else {
    mean = add_rn(mul_rn(c.coef1, p0), mul_rn(c.coef2, xt));
  }
  float nz = noise ? noise[idx] : 0.f;
  x_prev[idx] = add_rn(mean, mul_rn(c.sigma, nz));
  if (next_grid) {
    // layout [N,2,G,G]; channel 0 -> x base j/(G-1), channel 1 -> y base i/(G-1)
    const int gg = g * g;
    const int rem = (int)(idx % (2 * gg));
    const int ch = rem / gg;
    const int pix = rem - ch * gg;
    const int i = pix / g, j = pix - i * g;
    const float base = (float)(ch == 0 ? j : i) / (float)(g - 1);
    next_grid[idx] = sub_rn(mul_rn(add_rn(p0, base), 2.f), 1.f);
  }
}

__global__ void __launch_bounds__(256) hyp_mean_clamp_kernel(const float* __restrict__ x0, float* __restrict__ out,
                                                             int n_hyp, int per_doc, long total) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  const long d = idx / per_doc;
  const int e = (int)(idx - d * per_doc);
  const float* p = x0 + d * n_hyp * per_doc + e;
  float s = 0.f;
  for (int h = 0; h < n_hyp; ++h) s = add_rn(s, p[(long)h * per_doc]);
  s = div_rn(s, (float)n_hyp);
  out[idx] = fminf(fmaxf(s, -1.f), 1.f);
}

}  // namespace dvd

using namespace dvd;

extern "C" int dvd_sched_step(const dvd_sched_coef* coef, const float* x_t, const float* x0, const float* noise,
                              float* x_prev, float* next_grid, int n, int g, void* stream) {
  DVD_REQUIRE(coef && x_t && x0 && x_prev, "sched_step: null pointer");
  DVD_REQUIRE(coef->kind == 0 || coef->kind == 1, "sched_step: kind must be 0 (DDIM) or 1 (DDPM)");
  DVD_REQUIRE(n >= 0 && g >= 2, "sched_step: bad shape n=%d g=%d", n, g);
  DVD_REQUIRE(noise || coef->sigma == 0.f, "sched_step: sigma != 0 needs a noise tensor");
  const long total = (long)n * 2 * g * g;
  if (total == 0) return DVD_OK;
  sched_step_kernel<<<cdiv(total, 256), 256, 0, (hipStream_t)stream>>>(*coef, x_t, x0, noise, x_prev, next_grid,
                                                                     total, g);
  return check_launch("sched_step");
}

extern "C" int dvd_hyp_mean_clamp(const float* x0, float* out, int docs, int n_hyp, int g, void* stream) {
  DVD_REQUIRE(x0 && out, "hyp_mean_clamp: null pointer");
  DVD_REQUIRE(docs >= 0 && n_hyp >= 1 && g >= 2, "hyp_mean_clamp: bad shape");
  const int per_doc = 2 * g * g;
  const long total = (long)docs * per_doc;
  if (total == 0) return DVD_OK;
  hyp_mean_clamp_kernel<<<cdiv(total, 256), 256, 0, (hipStream_t)stream>>>(x0, out, n_hyp, per_doc, total);
  return check_launch("hyp_mean_clamp");
}
