// A small convolutional-network executor for the PRE-STAGE conditioning nets of the sampling path
// (SURVEY 8(f) rank 1): the two U2NETP document-mask nets and the text-line UNet that run once per
// document before the diffusion loop (train_settings/dvd/evaluation.py:162-216;
// train_settings/models/geotr/geotr_core.py:24-46,48-330,745-845; unet_model.py:4-37; unet_parts.py:8-77).
//
// The host describes a net as a flat op list over numbered activation slots (dvd_cn_op, include/dvd_hip.h);
// this file plans the slots in one caller-provided workspace and enqueues, per op:
//   conv (3x3 dilated or 1x1, optional channel-concat of two sources, eval-mode BatchNorm folded into the
//         weights on the host, bias, optional ReLU) = one gather kernel (im2col with the concat and the
//         dilation folded in) + the library's exact-f32 MFMA GEMM (v_mfma_f32_32x32x2_f32: fp32 products,
//         fp32 accumulate - the conditioning reused by every denoising step carries no low-precision error);
//   max-pool 2x2 / stride 2 (ceil_mode), bilinear resize (align_corners on or off), add, sigmoid.
// Activations are channels-last f32 ([H*W, C] row-major = the GEMM's C matrix, so a conv writes its output
// with no transpose).  Everything only enqueues on the caller's stream; nothing allocates or synchronises.
#include <string.h>

#include <algorithm>
#include <vector>

#include "common.h"
#include "gemm_common.h"

namespace dvd {

struct CnSlot { int h = 0, w = 0, c = 0; size_t off = 0; bool set = false; };

struct ConvNet {
  std::vector<dvd_cn_op> ops;
  std::vector<CnSlot> slots;
  std::vector<int> kpad;       // per op (conv only): padded K
  std::vector<int> ksplit;     // per op (conv only): K slices (1 = none)
  size_t part_off = 0, part_bytes = 0;   // split-K partial sums
  size_t col_off = 0, col_bytes = 0, need_bytes = 0;
  long weight_floats = 0;
  int in_c, in_h, in_w;
  int batch = 1;                // images per run: every slot holds [batch * h * w, c] rows (image-major)
};

static inline size_t al256(size_t b) { return (b + 255) & ~(size_t)255; }

// in: [B*H*W, C] -> out: planar [B,C,H,W] (all images of a batch in one launch)
__global__ void __launch_bounds__(256) nhwc_to_nchw_batched_kernel(const float* __restrict__ in, float* __restrict__ out, int c,
                                                                   long hw, long total) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  const long plane = i / hw, q = i - plane * hw;          // plane = b * c + ch
  const long b = plane / c;
  const int ch = (int)(plane - b * c);
  out[i] = in[(b * hw + q) * c + ch];
}

// in: planar [B,C,H,W] -> out: [B*H*W, C]
__global__ void __launch_bounds__(256) nchw_to_nhwc_kernel(const float* __restrict__ in, float* __restrict__ out, int c,
                                                          long hw, long total) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  const long p = i / c;
  const int ch = (int)(i - p * c);
  const long b = p / hw, q = p - b * hw;
  out[i] = in[(b * c + ch) * hw + q];
}

// col[p, tap * (ca + cb) + ch] = src(ch)[y + (ky - r) * dil, x + (kx - r) * dil] (zero outside); columns >= ks*ks*(ca+cb)
// are zero.  One thread per (pixel, 4 consecutive K entries): channels-last sources make the reads contiguous.
__global__ void __launch_bounds__(256) im2col_cat_kernel(const float* __restrict__ a, int ca, const float* __restrict__ b,
                                                         int cb, float* __restrict__ col, int kp, int ks, int dil,
                                                         int h, int w, long total4) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= total4) return;
  const int kq = kp / 4;
  const long p = i / kq;
  const int k0 = (int)(i - p * kq) * 4;
  const long hw = (long)h * w;
  const long img0 = (p / hw) * hw;                       // first row of this pixel's image
  const long pl = p - img0;
  const int y = (int)(pl / w), x = (int)(pl - (long)y * w);
  const int cc = ca + cb, kreal = ks * ks * cc, r = ks / 2;
  float v[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int k = k0 + j;
    float val = 0.f;
    if (k < kreal) {
      const int tap = k / cc, ch = k - tap * cc;
      const int yy = y + (tap / ks - r) * dil, xx = x + (tap % ks - r) * dil;
      if (yy >= 0 && yy < h && xx >= 0 && xx < w) {
        const long q = img0 + (long)yy * w + xx;
        val = ch < ca ? a[q * ca + ch] : b[q * cb + (ch - ca)];
      }
    }
    v[j] = val;
  }
  *reinterpret_cast<float4*>(col + p * kp + k0) = make_float4(v[0], v[1], v[2], v[3]);
}


// below this many rows a 33..64-channel narrow convolution runs as two 32-column halves (convnet_run, dvd_conv3x3_nhwc)
constexpr long NARROW_SPLIT_ROWS = 16384;

typedef float cn_floatx4 __attribute__((ext_vector_type(4)));
typedef float cn_floatx16 __attribute__((ext_vector_type(16)));

// Implicit-GEMM conv for narrow outputs (cout <= 64) over channel counts that are multiples of 16: the arithmetic of
// im2col_cat_kernel + gemm_f32_narrow_kernel - same K order (tap-major, channels of source a then b), same 8-deep chunks
// per lane half, same MFMA sequence, so the same bits - without materialising the [H*W, 9*cin] matrix: a wave owns 32
// consecutive pixels, lane (r, hh) reads 4 consecutive channels of pixel r's tap straight from the channels-last source
// (zeros outside the map).  One kernel instead of a gather (4.3 ms of a document's 13.6 ms of pre-stage kernel time) + a GEMM.
template <int NT>
__global__ void __launch_bounds__(256) conv_f32_narrow_kernel(const float* __restrict__ a, int ca,
                                                              const float* __restrict__ b, int cb,
                                                              const float* __restrict__ wgt, int kp,
                                                              const float* __restrict__ bias, float* __restrict__ out,
                                                              int cout, int ks, int dil, int h, int w, int act,
                                                              long rows) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = lane & 31, hh = lane >> 5;
  const long hw = (long)h * w;
  const long m0 = ((long)blockIdx.x * 4 + wave) * 32;     // rows = batch * hw, image-major: a wave may straddle two images
  if (m0 >= rows) return;
  const long pix = min(m0 + r, rows - 1);
  const long img0 = (pix / hw) * hw;
  const long pl = pix - img0;
  const int y = (int)(pl / w), x = (int)(pl - (long)y * w);
  const int cc = ca + cb, rad = ks / 2;
  const float* wp[NT];
  const int col0 = 32 * NT * blockIdx.y;     // blockIdx.y: which group of NT 32-column tiles (small maps split the columns)
#pragma unroll
  for (int t = 0; t < NT; ++t) wp[t] = wgt + (size_t)min(col0 + 32 * t + r, cout - 1) * kp + 4 * hh;
  cn_floatx16 acc[NT];
  float bcol[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    bcol[t] = bias[min(col0 + 32 * t + r, cout - 1)];
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;
  }
  const cn_floatx4 zero4 = {0.f, 0.f, 0.f, 0.f};
  // The K loop runs over 16-channel chunks n = tap * (cc / 16) + c0 / 16 in the order of the materialised matrix.  Operands
  // of chunk n + PF are requested before chunk n's MFMAs are issued (round 5): on the nets' small maps (36 x 36 and below:
  // a handful of workgroups on 256 CUs) nothing else hides a load, and without the prefetch every chunk cost one L2 round
  // trip on top of its 8 MFMAs.  Same chunks, same order, same MFMA sequence: same bits.
  constexpr int PF = 3, NS = PF + 1;       // chunk n lives in register set n % NS; the refill issued beside chunk n's MFMAs
                                           // goes to the set chunk n - 1 has just left (never to registers still being read)
  const int cpt = cc >> 4, nch = ks * ks * cpt;
  cn_floatx4 A0[NS], A1[NS], B0[NS][NT], B1[NS][NT];
  bool IN[NS];
  auto fetch = [&](int n, cn_floatx4& a0, cn_floatx4& a1, cn_floatx4 (&b0)[NT], cn_floatx4 (&b1)[NT], bool& inside) {
    const int tap = n / cpt, c0 = (n - tap * cpt) << 4;
    const int yy = y + (tap / ks - rad) * dil, xx = x + (tap % ks - rad) * dil;
    const bool ok = yy >= 0 && yy < h && xx >= 0 && xx < w;
    // taps outside the map read the lane's own pixel (always valid) and are zeroed when the chunk is consumed: every
    // lane issues every load, so the number of loads in flight is known and the waits are counted
    const long q = ok ? img0 + (long)yy * w + xx : pix;
    // channels c0 + 4 hh .. + 3 and c0 + 8 + 4 hh .. + 3 (ca % 8 == 0: a chunk never straddles the two sources)
    const int c1 = c0 + 4 * hh, c2 = c0 + 8 + 4 * hh;
    const bool s1 = c1 < ca, s2 = c2 < ca;            // selects, not branches: one multiply per address
    const float* p1 = (s1 ? a : b) + q * (s1 ? ca : cb) + (s1 ? c1 : c1 - ca);
    const float* p2 = (s2 ? a : b) + q * (s2 ? ca : cb) + (s2 ? c2 : c2 - ca);
    a0 = *(const cn_floatx4*)p1;
    a1 = *(const cn_floatx4*)p2;
    inside = ok;
    const int k0 = n << 4;
#pragma unroll
    for (int t = 0; t < NT; ++t) { b0[t] = *(const cn_floatx4*)(wp[t] + k0); b1[t] = *(const cn_floatx4*)(wp[t] + k0 + 8); }
  };
#pragma unroll
  for (int s = 0; s < PF; ++s) fetch(min(s, nch - 1), A0[s], A1[s], B0[s], B1[s], IN[s]);
#define DVD_CONSUME(s_, nxt_)                                                                                       \
  {                                                                                                                 \
    constexpr int f_ = ((s_) + PF) % NS;                                                                            \
    fetch(min((nxt_), nch - 1), A0[f_], A1[f_], B0[f_], B1[f_], IN[f_]);                                            \
    __builtin_amdgcn_sched_barrier(0);     /* the refill is issued here, ahead of this chunk's MFMAs */                \
    const cn_floatx4 a0 = IN[s_] ? A0[s_] : zero4, a1 = IN[s_] ? A1[s_] : zero4;                                    \
    _Pragma("unroll") for (int e = 0; e < 4; ++e)                                                                   \
      _Pragma("unroll") for (int t = 0; t < NT; ++t)                                                                \
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[e], B0[s_][t][e], acc[t], 0, 0, 0);                        \
    _Pragma("unroll") for (int e = 0; e < 4; ++e)                                                                   \
      _Pragma("unroll") for (int t = 0; t < NT; ++t)                                                                \
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[e], B1[s_][t][e], acc[t], 0, 0, 0);                        \
  }
  int n0 = 0;
  for (; n0 + NS <= nch; n0 += NS) {       // whole groups: no branch between the waits
    DVD_CONSUME(0, n0 + PF)
    DVD_CONSUME(1, n0 + 1 + PF)
    DVD_CONSUME(2, n0 + 2 + PF)
    DVD_CONSUME(3, n0 + 3 + PF)
    asm volatile("" ::: "memory");         // keeps the last refill on this side of the back edge
  }
  if (n0 < nch) {
    DVD_CONSUME(0, n0 + PF)
    if (n0 + 1 < nch) {
      DVD_CONSUME(1, n0 + 1 + PF)
      if (n0 + 2 < nch) DVD_CONSUME(2, n0 + 2 + PF)
    }
  }
#undef DVD_CONSUME
  static_assert(NS == 4, "the chunk loop is unrolled by 4");
  // epilogue: every value first (one wait for the bias and the trailing prefetches), then the stores back to back - with the
  // bias add inside the row test the compiler waited for ALL memory operations, i.e. the previous store, before each store
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const int col = col0 + 32 * t + r;
    float v[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      v[i] = acc[t][i] + bcol[t];
      v[i] = act == 2 ? fmaxf(v[i], 0.f) : v[i];
      asm volatile("" : "+v"(v[i]));       // the value exists before the row tests
    }
    if (col >= cout) continue;
    float* op = out + m0 * cout + col;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int lr = (i & 3) + 8 * (i >> 2) + 4 * hh;
      if (m0 + lr < rows) op[(long)lr * cout] = v[i];
    }
  }
}

// LAB (DVD_CONV_LDS=bytes): dynamic LDS per workgroup of the narrow conv kernels - unused by them, an occupancy cap for
// the L1-footprint experiment of round 5 (profiles/r5_conv_narrow_occupancy.txt).  0 in the product library.
static inline unsigned narrow_lds() {
#ifdef DVD_LAB
  if (const char* e = getenv("DVD_CONV_LDS")) {
    static bool once = false;
    if (!once) {
      (void)hipFuncSetAttribute((const void*)conv_f32_narrow_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      (void)hipFuncSetAttribute((const void*)conv_f32_narrow_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      once = true;
    }
    return (unsigned)atoi(e);
  }
#endif
  return 0;
}

// nn.MaxPool2d(2, stride=2, ceil_mode): windows are clipped at the border
__global__ void __launch_bounds__(256) maxpool2_ceil_kernel(const float* __restrict__ in, float* __restrict__ out, int c,
                                                            int h, int w, int ho, int wo, long total) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  const int ch = (int)(i % c);
  const long p = i / c;
  const long img = p / ((long)ho * wo), pl = p - img * ((long)ho * wo);
  const int x = (int)(pl % wo), y = (int)(pl / wo);
  const int y0 = 2 * y, x0 = 2 * x, y1 = min(y0 + 1, h - 1), x1 = min(x0 + 1, w - 1);
  const float* s = in + img * (long)h * w * c;
  const float m0 = fmaxf(s[((long)y0 * w + x0) * c + ch], s[((long)y0 * w + x1) * c + ch]);
  const float m1 = fmaxf(s[((long)y1 * w + x0) * c + ch], s[((long)y1 * w + x1) * c + ch]);
  out[i] = fmaxf(m0, m1);
}

// F.interpolate(mode='bilinear') source index (ATen area_pixel_compute_source_index) and weights
__device__ __forceinline__ void bil_index(int dst, int in_size, int out_size, int align, int& i0, int& i1, float& l1) {
  float src;
  if (align) {
    const float scale = out_size > 1 ? (float)(in_size - 1) / (float)(out_size - 1) : 0.f;
    src = scale * (float)dst;
  } else {
    const float scale = (float)in_size / (float)out_size;
    src = fmaxf(scale * ((float)dst + 0.5f) - 0.5f, 0.f);
  }
  i0 = min((int)src, in_size - 1);
  i1 = min(i0 + 1, in_size - 1);
  l1 = src - (float)i0;
}

// channels-last resize; ATen's blend order: l0y * (l0x * v00 + l1x * v01) + l1y * (l0x * v10 + l1x * v11)
__global__ void __launch_bounds__(256) resize_nhwc_kernel(const float* __restrict__ in, float* __restrict__ out, int c,
                                                          int hin, int win, int hout, int wout, int align, long total) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  const int ch = (int)(i % c);
  const long p = i / c;
  const long img = p / ((long)hout * wout), pl = p - img * ((long)hout * wout);
  const int x = (int)(pl % wout), y = (int)(pl / wout);
  int y0, y1, x0, x1;
  float ly, lx;
  bil_index(y, hin, hout, align, y0, y1, ly);
  bil_index(x, win, wout, align, x0, x1, lx);
  const float hy = 1.f - ly, hx = 1.f - lx;
  const float* s = in + img * (long)hin * win * c;
  const float v00 = s[((long)y0 * win + x0) * c + ch], v01 = s[((long)y0 * win + x1) * c + ch];
  const float v10 = s[((long)y1 * win + x0) * c + ch], v11 = s[((long)y1 * win + x1) * c + ch];
  out[i] = hy * (hx * v00 + lx * v01) + ly * (hx * v10 + lx * v11);
}

// planar resize: `planes` independent [hin, win] images
__global__ void __launch_bounds__(256) resize_planar_kernel(const float* __restrict__ in, float* __restrict__ out,
                                                            int hin, int win, int hout, int wout, int align, long total) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  const int x = (int)(i % wout);
  const long q = i / wout;
  const int y = (int)(q % hout);
  const long pl = q / hout;
  int y0, y1, x0, x1;
  float ly, lx;
  bil_index(y, hin, hout, align, y0, y1, ly);
  bil_index(x, win, wout, align, x0, x1, lx);
  const float hy = 1.f - ly, hx = 1.f - lx;
  const float* s = in + pl * (long)hin * win;
  out[i] = hy * (hx * s[(long)y0 * win + x0] + lx * s[(long)y0 * win + x1]) +
           ly * (hx * s[(long)y1 * win + x0] + lx * s[(long)y1 * win + x1]);
}

__global__ void __launch_bounds__(256) add_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                  float* __restrict__ out, long n) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i < n) out[i] = a[i] + b[i];
}

__global__ void __launch_bounds__(256) sigmoid_kernel(const float* __restrict__ a, float* __restrict__ out, long n) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i < n) out[i] = 1.f / (1.f + expf(-a[i]));
}

// out[ch, p] = (d0[p] > thr ? 1 : 0) * x[ch, p]        (Seg.forward, geotr_core.py:989-990)
__global__ void __launch_bounds__(256) mask_mul_kernel(const float* __restrict__ d0, const float* __restrict__ x,
                                                       float* __restrict__ out, float* __restrict__ mask_out, int c,
                                                       long hw, float thr) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= hw) return;
  const long img = blockIdx.y;                            // one image per grid row: d0 [N,1,hw], x / out [N,c,hw]
  d0 += img * hw; x += img * c * hw; out += img * c * hw;
  const float m = d0[i] > thr ? 1.f : 0.f;
  if (mask_out) mask_out[img * hw + i] = m;
  for (int ch = 0; ch < c; ++ch) out[(long)ch * hw + i] = m * x[(long)ch * hw + i];
}

// out[i] = act(sum_s part[s][i] + bias[i % c]), slices added in order
__global__ void __launch_bounds__(256) splitk_reduce_kernel(const float* __restrict__ part, int S, long mn,
                                                            const float* __restrict__ bias, int c, int act,
                                                            float* __restrict__ out) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= mn) return;
  float v = part[i];
  for (int s = 1; s < S; ++s) v += part[(long)s * mn + i];
  v += bias[i % c];
  out[i] = act == 2 ? fmaxf(v, 0.f) : v;
}

}  // namespace dvd

using namespace dvd;

extern "C" int dvd_nhwc_to_nchw(const float*, float*, int, int, int, void*);

extern "C" int dvd_convnet_create(const dvd_cn_op* ops, int n_ops, int n_slots, int in_c, int in_h, int in_w,
                                  void** handle) {
  return dvd_convnet_create_batched(ops, n_ops, n_slots, in_c, in_h, in_w, 1, handle);
}

// `batch` images per run: every activation slot holds the images one after the other ([batch * h * w, c] rows), so each
// op is ONE launch over the whole batch (a conv's GEMM has M = batch * h * w).  Kernel choices and the split-K factor
// are functions of the per-image shape only, so an image gives the same bits alone or in a batch.
extern "C" int dvd_convnet_create_batched(const dvd_cn_op* ops, int n_ops, int n_slots, int in_c, int in_h, int in_w,
                                          int batch, void** handle) {
  DVD_REQUIRE(ops && handle && n_ops > 0 && n_slots > 1, "convnet_create: bad arguments");
  DVD_REQUIRE(in_c > 0 && in_h > 0 && in_w > 0 && batch > 0 && batch <= 1024, "convnet_create: bad input shape / batch");
  ConvNet* n = new ConvNet();
  n->batch = batch;
  n->ops.assign(ops, ops + n_ops);
  n->slots.resize(n_slots);
  n->kpad.assign(n_ops, 0);
  n->ksplit.assign(n_ops, 1);
  size_t partmax = 0;
  n->in_c = in_c; n->in_h = in_h; n->in_w = in_w;
  n->slots[0].h = in_h; n->slots[0].w = in_w; n->slots[0].c = in_c; n->slots[0].set = true;
  size_t off = 0, colmax = 0;
  n->slots[0].off = 0;
  const size_t nb = (size_t)batch;
  off = al256(nb * in_h * in_w * in_c * 4);
  long wf = 0;
  auto fail = [&](const char* msg, int i) {
    set_error("convnet_create: op %d: %s", i, msg);
    delete n;
    return DVD_E_ARG;
  };
  for (int i = 0; i < n_ops; ++i) {
    const dvd_cn_op& o = n->ops[i];
    if (o.a < 0 || o.a >= n_slots || !n->slots[o.a].set) return fail("input slot not defined", i);
    if (o.dst <= 0 || o.dst >= n_slots || n->slots[o.dst].set) return fail("bad or already written destination slot", i);
    if (o.b >= n_slots || (o.b >= 0 && !n->slots[o.b].set)) return fail("second input slot not defined", i);
    const CnSlot& a = n->slots[o.a];
    CnSlot d;
    d.set = true;
    switch (o.op) {
      case DVD_CN_CONV: {
        if ((o.ks != 1 && o.ks != 3) || o.dil < 1 || o.cout < 1) return fail("bad conv parameters", i);
        int cin = a.c;
        if (o.b >= 0) {
          const CnSlot& b = n->slots[o.b];
          if (b.h != a.h || b.w != a.w) return fail("concat sources differ in size", i);
          cin += b.c;
        }
        const int kp = (o.ks * o.ks * cin + 15) / 16 * 16;
        n->kpad[i] = kp;
        if (o.w_off != wf) return fail("weights must be packed in op order (w_off mismatch)", i);
        wf += (long)o.cout * kp + (o.cout + 3) / 4 * 4;   // bias padded to 16 bytes: every conv's weights stay 16-byte aligned
        d.h = a.h; d.w = a.w; d.c = o.cout;
        {
          // Small maps with wide channels (the UNet's deep layers: 324 ... 1296 pixels, K up to 9216) give the 128 x 128
          // tile kernel a handful of workgroups and a K loop of hundreds of barrier-separated steps.  Such a conv is cut
          // into S slices of K, run as ONE batched GEMM launch (batch stride = K / S along both operands) into S partial
          // sums, and a small kernel adds them in slice order (deterministic), then bias and activation.
          const long tiles = (long)((a.h * a.w + 127) / 128) * ((o.cout + 127) / 128);
          int S = 1;
          if (o.cout > 64 && kp >= 1024 && tiles < 128) {
            const int units = kp / 16;
            for (int c = 2; c <= units && tiles * c <= 256; ++c)
              if (units % c == 0 && kp / c >= 128) S = c;
          }
          n->ksplit[i] = S;
          if (S > 1) partmax = std::max(partmax, nb * S * a.h * a.w * o.cout * 4);
          // the im2col matrix is sized only for the convs that write one (the same two predicates convnet_run takes the
          // implicit-GEMM kernel / the direct 1x1 read on): the 16 -> 64-channel 3x3 convs at full resolution would
          // otherwise reserve 382 MB per image that nothing ever touches
          const int cb = o.b >= 0 ? n->slots[o.b].c : 0;
          const bool implicit = o.cout <= 64 && a.c % 8 == 0 && cb % 8 == 0 && (a.c + cb) % 16 == 0 && S == 1;
          const bool direct = o.ks == 1 && o.b < 0 && a.c == kp;
          if (!implicit && !direct) colmax = std::max(colmax, nb * a.h * a.w * kp * 4);
        }
        break;
      }
      case DVD_CN_POOL:
        d.h = o.flag ? (a.h + 1) / 2 : a.h / 2; d.w = o.flag ? (a.w + 1) / 2 : a.w / 2; d.c = a.c;
        if (d.h < 1 || d.w < 1) return fail("pool of a 1-pixel map", i);
        break;
      case DVD_CN_RESIZE:
        if (o.b >= 0) { d.h = n->slots[o.b].h; d.w = n->slots[o.b].w; } else { d.h = o.h; d.w = o.w; }
        if (d.h < 1 || d.w < 1) return fail("bad resize target", i);
        d.c = a.c;
        break;
      case DVD_CN_ADD: {
        if (o.b < 0) return fail("add needs two inputs", i);
        const CnSlot& b = n->slots[o.b];
        if (b.h != a.h || b.w != a.w || b.c != a.c) return fail("add operands differ in shape", i);
        d.h = a.h; d.w = a.w; d.c = a.c;
        break;
      }
      case DVD_CN_SIGMOID:
        d.h = a.h; d.w = a.w; d.c = a.c;
        break;
      default:
        return fail("unknown op", i);
    }
    d.off = off;
    off += al256(nb * d.h * d.w * d.c * 4);
    n->slots[o.dst] = d;
  }
  n->col_off = off; n->col_bytes = al256(colmax);
  n->part_off = off + n->col_bytes; n->part_bytes = al256(partmax);
  n->need_bytes = off + n->col_bytes + n->part_bytes;
  n->weight_floats = wf;
  *handle = n;
  return DVD_OK;
}

extern "C" int dvd_convnet_destroy(void* handle) {
  delete (ConvNet*)handle;
  return DVD_OK;
}

extern "C" long dvd_convnet_workspace_bytes(void* handle) { return handle ? (long)((ConvNet*)handle)->need_bytes : -1; }
extern "C" long dvd_convnet_weight_floats(void* handle) { return handle ? ((ConvNet*)handle)->weight_floats : -1; }

extern "C" int dvd_convnet_slot_shape(void* handle, int slot, int* h, int* w, int* c) {
  DVD_REQUIRE(handle && h && w && c, "convnet_slot_shape: null pointer");
  ConvNet* n = (ConvNet*)handle;
  DVD_REQUIRE(slot >= 0 && slot < (int)n->slots.size() && n->slots[slot].set, "convnet_slot_shape: slot %d not defined", slot);
  *h = n->slots[slot].h; *w = n->slots[slot].w; *c = n->slots[slot].c;
  return DVD_OK;
}

extern "C" int dvd_conv3x3_nhwc(const float* in, int c, const float* wgt, int kp, const float* bias, float* out, int cout,
                                int h, int w, int relu, void* stream) {
  DVD_REQUIRE(in && wgt && bias && out, "conv3x3_nhwc: null pointer");
  DVD_REQUIRE(c > 0 && c % 16 == 0 && cout > 0 && kp >= 9 * c && kp % 4 == 0 && h > 0 && w > 0,
              "conv3x3_nhwc: bad shape (c %d, cout %d, kp %d)", c, cout, kp);
  DVD_REQUIRE(((uintptr_t)in % 16) == 0 && ((uintptr_t)wgt % 16) == 0, "conv3x3_nhwc: operands must be 16-byte aligned");
  const long rows = (long)h * w;
  if (cout > 64) return launch_gemm_conv_f32(in, c, nullptr, 0, h, w, 3, 1, rows, wgt, kp, bias, out, cout, relu ? 2 : 0, stream);
  const dim3 grd(cdiv(rows, 128));
  hipStream_t st = (hipStream_t)stream;
  if (cout <= 32 || rows < NARROW_SPLIT_ROWS)
    conv_f32_narrow_kernel<1><<<dim3(grd.x, cout <= 32 ? 1 : 2), 256, narrow_lds(), st>>>(in, c, nullptr, 0, wgt, kp, bias, out, cout, 3, 1, h, w,
                                                                                          relu ? 2 : 0, rows);
  else
    conv_f32_narrow_kernel<2><<<grd, 256, narrow_lds(), st>>>(in, c, nullptr, 0, wgt, kp, bias, out, cout, 3, 1, h, w, relu ? 2 : 0, rows);
  return check_launch("conv3x3_nhwc");
}

extern "C" int dvd_convnet_run(void* handle, const float* in_nchw, const float* weights, void* workspace,
                               long workspace_bytes, int n_out, const int* out_slots, float* const* out_nchw,
                               void* stream) {
  DVD_REQUIRE(handle && in_nchw && weights && workspace, "convnet_run: null pointer");
  DVD_REQUIRE(n_out >= 0 && (n_out == 0 || (out_slots && out_nchw)), "convnet_run: bad outputs");
  ConvNet* n = (ConvNet*)handle;
  DVD_REQUIRE((size_t)workspace_bytes >= n->need_bytes && ((uintptr_t)workspace % 256) == 0 &&
                  ((uintptr_t)weights % 16) == 0,
              "convnet_run: need %zu workspace bytes, 256-byte aligned (got %ld)", n->need_bytes, workspace_bytes);
  hipStream_t st = (hipStream_t)stream;
  char* ws = (char*)workspace;
  auto P = [&](int slot) { return (float*)(ws + n->slots[slot].off); };
  float* col = (float*)(ws + n->col_off);
  const long B = n->batch;
  {
    const long hw = (long)n->in_h * n->in_w, total = B * hw * n->in_c;
    nchw_to_nhwc_kernel<<<cdiv(total, 256), 256, 0, st>>>(in_nchw, P(0), n->in_c, hw, total);
  }
  for (size_t i = 0; i < n->ops.size(); ++i) {
    const dvd_cn_op& o = n->ops[i];
    const CnSlot& a = n->slots[o.a];
    const CnSlot& d = n->slots[o.dst];
    const long nd = B * d.h * d.w * d.c;
    const long rows = B * a.h * a.w;                      // GEMM rows of a conv over slot a
    switch (o.op) {
      case DVD_CN_CONV: {
        const int kp = n->kpad[i];
        const int cb = o.b >= 0 ? n->slots[o.b].c : 0;
        if (o.cout <= 64 && a.c % 8 == 0 && cb % 8 == 0 && (a.c + cb) % 16 == 0 && n->ksplit[i] == 1) {
          // narrow output over 16-aligned channels: implicit GEMM, no im2col matrix
          const dim3 grd(cdiv(rows, 128));
          const float* wgt = weights + o.w_off;
          if (o.cout <= 32 || rows < NARROW_SPLIT_ROWS)
            // one 32-column tile per wave; a 33..64-channel layer on a small map gets its two tiles from two workgroups
            // (twice the waves on a chip that is mostly idle there; per output the same chain of MFMAs: same bits)
            conv_f32_narrow_kernel<1><<<dim3(grd.x, o.cout <= 32 ? 1 : 2), 256, narrow_lds(), st>>>(P(o.a), a.c, o.b >= 0 ? P(o.b) : nullptr, cb, wgt, kp,
                                                           wgt + (long)o.cout * kp, P(o.dst), o.cout, o.ks, o.dil, a.h, a.w,
                                                           o.act, rows);
          else
            conv_f32_narrow_kernel<2><<<grd, 256, narrow_lds(), st>>>(P(o.a), a.c, o.b >= 0 ? P(o.b) : nullptr, cb, wgt, kp,
                                                           wgt + (long)o.cout * kp, P(o.dst), o.cout, o.ks, o.dil, a.h, a.w,
                                                           o.act, rows);
          break;
        }
        if (a.c % 16 == 0 && cb % 16 == 0 && kp == o.ks * o.ks * (a.c + cb) && n->ksplit[i] == 1 && rows < (1l << 31) &&
            !(o.ks == 1 && o.b < 0)) {
          // wide output over 16-aligned channels (round 5): the 128 x 128 exact-f32 GEMM gathers its A tiles from the map(s)
          // itself - the im2col matrix (up to 9216 columns per pixel) is never written or read
          if (int e = launch_gemm_conv_f32(P(o.a), a.c, o.b >= 0 ? P(o.b) : nullptr, cb, a.h, a.w, o.ks, o.dil, rows,
                                           weights + o.w_off, kp, weights + o.w_off + (long)o.cout * kp, P(o.dst), o.cout, o.act,
                                           stream))
            return e;
          break;
        }
        const float* A = P(o.a);
        int lda = a.c;
        if (!(o.ks == 1 && o.b < 0 && a.c == kp)) {   // a 1x1 conv over a 16-aligned single source reads the slot directly
          const long total4 = rows * (kp / 4);
          im2col_cat_kernel<<<cdiv(total4, 256), 256, 0, st>>>(P(o.a), a.c, o.b >= 0 ? P(o.b) : nullptr, cb, col, kp,
                                                              o.ks, o.dil, a.h, a.w, total4);
          A = col; lda = kp;
        }
        dvd_gemm_desc g;
        memset(&g, 0, sizeof(g));
        DVD_REQUIRE(rows < (1l << 31), "convnet_run: batch too large for one GEMM");
        g.dtype = 1; g.M = (int)rows; g.N = o.cout; g.K = kp; g.batch = 1;
        g.A = A; g.lda = lda; g.B = weights + o.w_off; g.ldb = kp;
        g.C32 = P(o.dst); g.ldc = o.cout;
        g.bias = weights + o.w_off + (long)o.cout * kp; g.act = o.act;
        g.lo_scale = 1.f;
        const int S = n->ksplit[i];
        if (S > 1) {                    // split-K: S partial products in one batched launch, then reduce + bias + act
          float* part = (float*)(ws + n->part_off);
          const long mn = (long)g.M * g.N;
          g.K = kp / S; g.batch = S; g.strideA = kp / S; g.strideB = kp / S;
          g.C32 = part; g.strideC32 = mn; g.bias = nullptr; g.act = 0;
          if (int e = dvd_gemm_nt(&g, stream)) return e;
          splitk_reduce_kernel<<<cdiv(mn, 256), 256, 0, st>>>(part, S, mn, weights + o.w_off + (long)o.cout * kp, o.cout,
                                                             o.act, P(o.dst));
          break;
        }
        if (int e = dvd_gemm_nt(&g, stream)) return e;
        break;
      }
      case DVD_CN_POOL:
        maxpool2_ceil_kernel<<<cdiv(nd, 256), 256, 0, st>>>(P(o.a), P(o.dst), a.c, a.h, a.w, d.h, d.w, nd);
        break;
      case DVD_CN_RESIZE:
        resize_nhwc_kernel<<<cdiv(nd, 256), 256, 0, st>>>(P(o.a), P(o.dst), a.c, a.h, a.w, d.h, d.w, o.flag, nd);
        break;
      case DVD_CN_ADD:
        add_kernel<<<cdiv(nd, 256), 256, 0, st>>>(P(o.a), P(o.b), P(o.dst), nd);
        break;
      case DVD_CN_SIGMOID:
        sigmoid_kernel<<<cdiv(nd, 256), 256, 0, st>>>(P(o.a), P(o.dst), nd);
        break;
    }
  }
  for (int k = 0; k < n_out; ++k) {
    const int s = out_slots[k];
    DVD_REQUIRE(s >= 0 && s < (int)n->slots.size() && n->slots[s].set && out_nchw[k], "convnet_run: bad output %d", k);
    const long hw = (long)n->slots[s].h * n->slots[s].w, total = B * hw * n->slots[s].c;
    nhwc_to_nchw_batched_kernel<<<cdiv(total, 256), 256, 0, st>>>(P(s), out_nchw[k], n->slots[s].c, hw, total);
  }
  return check_launch("convnet_run");
}

extern "C" int dvd_resize_bilinear_nchw(const float* in, float* out, long planes, int hin, int win, int hout, int wout,
                                        int align_corners, void* stream) {
  DVD_REQUIRE(in && out, "resize_bilinear_nchw: null pointer");
  DVD_REQUIRE(planes >= 0 && hin > 0 && win > 0 && hout > 0 && wout > 0, "resize_bilinear_nchw: bad shape");
  const long total = planes * hout * wout;
  if (total == 0) return DVD_OK;
  resize_planar_kernel<<<cdiv(total, 256), 256, 0, (hipStream_t)stream>>>(in, out, hin, win, hout, wout,
                                                                         align_corners ? 1 : 0, total);
  return check_launch("resize_bilinear_nchw");
}

extern "C" int dvd_threshold_mask_mul(const float* d0, const float* x_nchw, float* out_nchw, float* mask_out, int c,
                                      long hw, float thr, void* stream) {
  return dvd_threshold_mask_mul_batch(d0, x_nchw, out_nchw, mask_out, 1, c, hw, thr, stream);
}

extern "C" int dvd_threshold_mask_mul_batch(const float* d0, const float* x_nchw, float* out_nchw, float* mask_out, int n,
                                            int c, long hw, float thr, void* stream) {
  DVD_REQUIRE(d0 && x_nchw && out_nchw && c > 0 && hw > 0 && n > 0 && n <= 65535, "threshold_mask_mul: bad arguments");
  mask_mul_kernel<<<dim3(cdiv(hw, 256), n), 256, 0, (hipStream_t)stream>>>(d0, x_nchw, out_nchw, mask_out, c, hw, thr);
  return check_launch("threshold_mask_mul");
}
