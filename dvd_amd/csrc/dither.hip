// Temporal dithering of the f16 weight rounding.
//
// The per-step GEMM weights exist as f16 pairs W = hi + lo (weights.py).  Rounding W to ONE f16 (hi alone) is a
// perturbation that is IDENTICAL at every denoising step: its effect on x0 drifts linearly with the step count
// (DESIGN.md section 5), which is why round 1 added the lo pass (2x the GEMM MFMAs).  This kernel makes the rounding
// zero-mean over the steps instead: before evaluation number `step` every weight is re-rounded to one of its two f16
// neighbours,
//     w_step = up   if  u(elem, step) < (W - down) / (up - down)   else down,      E_step[w_step] = W exactly,
// with u(elem, step) = hash(elem) + step * 2^32/phi  (mod 2^32): a golden-ratio Kronecker sequence per element, whose
// running mean over S steps converges like log(S)/S (not 1/sqrt(S) as independent random rounding would).  The GEMMs
// then run ONE pass on w_step.  Integer arithmetic + one correctly rounded fp32 division: bit-reproducible on the host
// (tests/test_gpu_gemm.py restates it in numpy).
//
// No counterpart in the reference (fp32 PyTorch); the reference lines the dithered GEMMs compute are those of the
// per-step nn.Linear / 1x1 convs (idf/cross_attn.py:197-221,52-57; idf/cross_model.py:163-174,237-292).
#include "common.h"

namespace dvd {

__device__ __forceinline__ unsigned dither_hash(unsigned g) {
  unsigned h = g * 0x9E3779B1u;
  h ^= h >> 15; h *= 0x85EBCA77u;
  h ^= h >> 13; h *= 0xC2B2AE3Du;
  h ^= h >> 16;
  return h;
}

__device__ __forceinline__ unsigned short dither_one(unsigned short hb, unsigned short lb, unsigned g, unsigned phase) {
  if ((lb & 0x7fffu) == 0) return hb;                               // W is an f16 already
  unsigned short nb;
  if ((hb & 0x7fffu) == 0) nb = (unsigned short)((lb & 0x8000u) | 1u);          // from +-0 towards lo's sign
  else nb = ((hb ^ lb) & 0x8000u) ? (unsigned short)(hb - 1) : (unsigned short)(hb + 1);   // towards / away from zero
  if ((nb & 0x7c00u) == 0x7c00u) return hb;                         // would step onto inf: keep the nearest
  const float hf = __half2float(__ushort_as_half(hb)), lf = __half2float(__ushort_as_half(lb));
  const float nf = __half2float(__ushort_as_half(nb));
  float frac = __fdiv_rn(fabsf(lf), fabsf(__fsub_rn(nf, hf)));
  frac = fminf(frac, 0.99999994f);
  const unsigned thr = (unsigned)(frac * 4294967296.0f);
  const unsigned u = dither_hash(g) + phase;
  return u < thr ? nb : hb;
}

__global__ void __launch_bounds__(256) dither_f16_kernel(const uint4* __restrict__ hi, const uint4* __restrict__ lo,
                                                         uint4* __restrict__ out, long n8, unsigned elem0, unsigned phase) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n8) return;
  const uint4 h = hi[i], l = lo[i];
  const unsigned hw[4] = {h.x, h.y, h.z, h.w}, lw[4] = {l.x, l.y, l.z, l.w};
  unsigned ow[4];
  const unsigned g0 = elem0 + (unsigned)(i * 8);
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const unsigned short a = dither_one((unsigned short)(hw[k] & 0xffffu), (unsigned short)(lw[k] & 0xffffu), g0 + 2 * k, phase);
    const unsigned short b = dither_one((unsigned short)(hw[k] >> 16), (unsigned short)(lw[k] >> 16), g0 + 2 * k + 1, phase);
    ow[k] = (unsigned)a | ((unsigned)b << 16);
  }
  out[i] = make_uint4(ow[0], ow[1], ow[2], ow[3]);
}

}  // namespace dvd

using namespace dvd;

extern "C" int dvd_dither_f16(const void* hi, const void* lo, void* out, long nelem, unsigned elem0, unsigned step,
                              void* stream) {
  DVD_REQUIRE(hi && lo && out, "dither_f16: null pointer");
  DVD_REQUIRE(nelem > 0 && nelem % 8 == 0, "dither_f16: nelem=%ld must be a positive multiple of 8", nelem);
  DVD_REQUIRE(((uintptr_t)hi % 16) == 0 && ((uintptr_t)lo % 16) == 0 && ((uintptr_t)out % 16) == 0,
              "dither_f16: pointers must be 16-byte aligned");
  const long n8 = nelem / 8;
  const unsigned phase = step * 0x9E3779B9u;          // step * 2^32 / golden ratio  (mod 2^32)
  dither_f16_kernel<<<(unsigned)((n8 + 255) / 256), 256, 0, (hipStream_t)stream>>>((const uint4*)hi, (const uint4*)lo,
                                                                                   (uint4*)out, n8, elem0, phase);
  return check_launch("dither_f16");
}
