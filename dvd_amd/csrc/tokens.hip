// Token-side (HBM-bound) kernels of the denoiser: everything between the MFMA GEMMs / attention.
// One wave (64 lanes) owns one token row wherever a row reduction is needed (LayerNorm), so
// reductions are DPP/shuffle-only and every global access is a contiguous 16-byte-per-lane run.
//
//   embed_obs_ln     K1+K7  patchify x_t (conv k=s=2, 2->384) + bias + pos-embed; LN (no affine)
//   layernorm_rows   K7/K9/K10/K13/K14 LayerNorm (+affine) (+adaLN modulate) -> f16 GEMM operand
//   build_r_rows     K18+K5r feature warp + concat[init_flow, init_feat] + 2x2 patch rows (f16)
//   patch_rows       K3/K5  2x2 patch rows of a [C,G,G] or [G,G,C] map (f32)
//   dwconv3x3        K14    depthwise 3x3 + folded BN + ReLU on the token grid
//   colmean / posenc K12    adaptive 2-D positional encoding
//   small_linear     K2/K6/K12/K15 tiny-M linears (t-embed MLP, adaLN, pos-enc scale MLPs)
//   final_tokens     K15+K16 LN -> LN -> modulate -> Linear 1536->8 -> unpatchify + init_flow
#include "common.h"
#include "mfma.h"

namespace dvd {

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

// ----------------------------------------------------------------------------------------------
// K1 + K7: x_tok = PatchEmbed(x_t) + pos ; xq = LN(x_tok)  (idf/cross_model.py:571,238)
// x [N,2,G,G] f32; w [384,8] (conv weight flattened c*4+p*2+q); pos [T,384]
// ----------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) embed_obs_ln_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                           const float* __restrict__ bias,
                                                           const float* __restrict__ pos, float* __restrict__ tok32,
                                                           _Float16* __restrict__ ln16, int g, long rows) {
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int lane = threadIdx.x & 63;
  const int side = g >> 1, T = side * side;
  const int n = (int)(row / T), t = (int)(row % T);
  const int ty = t / side, tx = t % side;
  float pv[8];
#pragma unroll
  for (int c = 0; c < 2; ++c)
#pragma unroll
    for (int p = 0; p < 2; ++p)
#pragma unroll
      for (int q = 0; q < 2; ++q)
        pv[c * 4 + p * 2 + q] = x[(((size_t)n * 2 + c) * g + (2 * ty + p)) * g + 2 * tx + q];
  float v[6];
  float s = 0.f;
#pragma unroll
  for (int j = 0; j < 6; ++j) {
    const int o = lane + 64 * j;
    float acc = bias[o];
#pragma unroll
    for (int k = 0; k < 8; ++k) acc += w[o * 8 + k] * pv[k];
    acc += pos[(size_t)t * 384 + o];
    v[j] = acc;
    s += acc;
  }
  const float mean = wave_sum(s) * (1.f / 384.f);
  float q2 = 0.f;
#pragma unroll
  for (int j = 0; j < 6; ++j) {
    const float d = v[j] - mean;
    q2 += d * d;
  }
  const float rstd = rsqrtf(wave_sum(q2) * (1.f / 384.f) + 1e-6f);
#pragma unroll
  for (int j = 0; j < 6; ++j) {
    const int o = lane + 64 * j;
    tok32[row * 384 + o] = v[j];
    ln16[row * 384 + o] = (_Float16)((v[j] - mean) * rstd);
  }
}

// ----------------------------------------------------------------------------------------------
// LayerNorm over C (384 or 1536) channels of token rows, optional affine, optional adaLN modulate
// y = LN(x) * (1 + scale[sample]) + shift[sample]   (idf/cross_model.py:13-14); f16 output.
// Batched (blockIdx.y): in += z*sIn, out += z*sOut (stream slices of the 1536-wide buffer).
// ----------------------------------------------------------------------------------------------
template <int C>
__global__ void __launch_bounds__(256) layernorm_rows_kernel(const float* __restrict__ in, int ldin, long sIn,
                                                             _Float16* __restrict__ out, int ldout, long sOut,
                                                             const float* __restrict__ gamma,
                                                             const float* __restrict__ beta,
                                                             const float* __restrict__ shift,
                                                             const float* __restrict__ scale, int ldmod,
                                                             int mod_rows, float eps, long rows) {
  constexpr int PER = C / 64;  // 6 or 24 floats per lane, as float2 / float4 runs
  constexpr int V = (C == 384) ? 2 : 4;
  constexpr int NV = PER / V;
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int lane = threadIdx.x & 63;
  const float* ip = in + blockIdx.y * sIn + row * ldin;
  float v[PER];
  float s = 0.f;
  typedef float fvec __attribute__((ext_vector_type(V)));
  typedef _Float16 hvec __attribute__((ext_vector_type(V)));
#pragma unroll
  for (int k = 0; k < NV; ++k) {
    const int c0 = (k * 64 + lane) * V;
    const fvec x = *(const fvec*)(ip + c0);          // 8-/16-byte loads: rows are 16-byte aligned (host check)
#pragma unroll
    for (int e = 0; e < V; ++e) {
      v[k * V + e] = x[e];
      s += x[e];
    }
  }
  const float mean = wave_sum(s) * (1.f / C);
  float q2 = 0.f;
#pragma unroll
  for (int k = 0; k < PER; ++k) {
    const float d = v[k] - mean;
    q2 += d * d;
  }
  const float rstd = rsqrtf(wave_sum(q2) * (1.f / C) + eps);
  _Float16* op = out + blockIdx.y * sOut + row * ldout;
  const float* sh = shift ? shift + (row / mod_rows) * ldmod : nullptr;
  const float* sc = scale ? scale + (row / mod_rows) * ldmod : nullptr;
#pragma unroll
  for (int k = 0; k < NV; ++k) {
    const int c0 = (k * 64 + lane) * V;
    fvec g = {}, bt = {}, scv = {}, shv = {};
    if (gamma) { g = *(const fvec*)(gamma + c0); bt = *(const fvec*)(beta + c0); }
    if (sc) { scv = *(const fvec*)(sc + c0); shv = *(const fvec*)(sh + c0); }
    hvec o;
#pragma unroll
    for (int e = 0; e < V; ++e) {
      float y = (v[k * V + e] - mean) * rstd;
      if (gamma) y = y * g[e] + bt[e];
      if (sc) y = y * (1.f + scv[e]) + shv[e];
      o[e] = (_Float16)y;
    }
    *(hvec*)(op + c0) = o;
  }
}

// ----------------------------------------------------------------------------------------------
// K18 + K5r: rows of the r-embedder GEMM.  For sample n, token (ty,tx), patch position pq:
//   row[pq*258 + 0..1]   = init_flow[n, c, 2ty+p, 2tx+q]
//   row[pq*258 + 2..257] = init_feat = warp ? bilinear(feat_doc, grid(x0_prev)) : feat_doc
// feat is channels-last [docs, G, G, 256] f32; flow [N,2,G,G] (x0_prev = init_flow);
// the warp grid is (x0_prev + base)*2-1 (idf/gaussian_diffusion.py:618-624), zeros padding.
// Output f16 [N*T, ldo] with ldo >= 1032 (pad columns zeroed).   One wave per (token, pq).
// mode: 0 = init_feat is zero, 1 = feat itself (t > 600, idf/cross_model.py:597-598), 2 = warped feat,
// 3 = an explicit init_feat tensor [N,256,G,G] (NCHW) supplied by the caller (direct model() calls).
// ----------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) build_r_rows_kernel(const float* __restrict__ feat,
                                                           const float* __restrict__ init_feat_nchw,
                                                           const float* __restrict__ flow,
                                                           _Float16* __restrict__ out, int ldo, int g, int n_hyp,
                                                           int mode, long items) {
  const long item = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (item >= items) return;
  const int lane = threadIdx.x & 63;
  const int side = g >> 1, T = side * side;
  const int pq = (int)(item & 3);
  const long tokrow = item >> 2;
  const int n = (int)(tokrow / T), t = (int)(tokrow % T);
  const int y = 2 * (t / side) + (pq >> 1), x = 2 * (t % side) + (pq & 1);
  const size_t gg = (size_t)g * g;
  const float fx = flow[((size_t)n * 2 + 0) * gg + (size_t)y * g + x];
  const float fy = flow[((size_t)n * 2 + 1) * gg + (size_t)y * g + x];
  _Float16* orow = out + tokrow * ldo + pq * 258;
  const float* fd = feat + (size_t)(n / n_hyp) * gg * 256;
  float v[4] = {0.f, 0.f, 0.f, 0.f};   // channels 4*lane .. 4*lane+3
  if (mode == 1) {
    const floatx4 f = *(const floatx4*)(fd + ((size_t)y * g + x) * 256 + 4 * lane);
    v[0] = f[0]; v[1] = f[1]; v[2] = f[2]; v[3] = f[3];
  } else if (mode == 3) {
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = init_feat_nchw[((size_t)n * 256 + 4 * lane + e) * gg + (size_t)y * g + x];
  } else if (mode == 2) {
    const float inv = 1.f / (float)(g - 1);
    const float gx = sub_rn(mul_rn(add_rn(fx, (float)x * inv), 2.f), 1.f);
    const float gy = sub_rn(mul_rn(add_rn(fy, (float)y * inv), 2.f), 1.f);
    const float ix = ((gx + 1.f) * 0.5f) * (float)(g - 1), iy = ((gy + 1.f) * 0.5f) * (float)(g - 1);
    float x0f = floorf(ix), y0f = floorf(iy);
    const float wx1 = ix - x0f, wy1 = iy - y0f, wx0 = (x0f + 1.f) - ix, wy0 = (y0f + 1.f) - iy;
    x0f = fminf(fmaxf(x0f, -2.f), (float)g);
    y0f = fminf(fmaxf(y0f, -2.f), (float)g);
    if (!(ix == ix)) x0f = -2.f;
    if (!(iy == iy)) y0f = -2.f;
    const int x0 = (int)x0f, y0 = (int)y0f;
    const float wts[4] = {wx0 * wy0, wx1 * wy0, wx0 * wy1, wx1 * wy1};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int xx = x0 + (k & 1), yy = y0 + (k >> 1);
      if (xx >= 0 && xx < g && yy >= 0 && yy < g) {
        const floatx4 f = *(const floatx4*)(fd + ((size_t)yy * g + xx) * 256 + 4 * lane);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] += f[e] * wts[k];
      }
    }
  }
  // columns 2 + 4*lane + e : 2-byte aligned only -> scalar f16 stores (coalesced across lanes)
#pragma unroll
  for (int e = 0; e < 4; ++e) orow[2 + 4 * lane + e] = (_Float16)v[e];
  if (lane == 0) {
    orow[0] = (_Float16)fx;
    orow[1] = (_Float16)fy;
  }
  if (pq == 3) {  // zero the K padding once per row
    for (int c = 1032 + lane; c < ldo; c += 64) out[tokrow * ldo + c] = (_Float16)0.f;
  }
}

// ----------------------------------------------------------------------------------------------
// 2x2 patch rows of a feature map with arbitrary element strides (NCHW or channels-last), f32:
//   out[(n*T + t)*ldo + pq*C + c] = in[n*sn + c*sc + (2ty+p)*sy + (2tx+q)*sx]
// ----------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) patch_rows_kernel(const float* __restrict__ in, long sn, long sc, long sy,
                                                         long sx, float* __restrict__ out, int ldo, int c, int g,
                                                         long total) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  const int k4 = 4 * c;
  const long row = idx / k4;
  const int col = (int)(idx % k4);
  const int pq = col / c, ch = col % c;
  const int side = g >> 1, T = side * side;
  const int n = (int)(row / T), t = (int)(row % T);
  const int y = 2 * (t / side) + (pq >> 1), x = 2 * (t % side) + (pq & 1);
  out[row * ldo + col] = in[n * sn + ch * sc + y * sy + x * sx];
}

// ----------------------------------------------------------------------------------------------
// K14 middle: depthwise 3x3 (pad 1) + folded BatchNorm + ReLU on the side x side token grid
// (idf/cross_attn.py:33-41).  in/out f16 [N*T, C] token-major; w [9, C] f32 tap-major (BN scale
// folded in), b [C].  One thread per (token, 8 channels): 16-byte loads/stores.
// ----------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) dwconv3x3_kernel(const _Float16* __restrict__ in,
                                                        _Float16* __restrict__ out, const float* __restrict__ w,
                                                        const float* __restrict__ b, int side, int c, long total) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  const int c8 = c >> 3;
  const long tok = idx / c8;
  const int ch = (int)(idx % c8) * 8;
  const int T = side * side;
  const int t = (int)(tok % T);
  const int ty = t / side, tx = t % side;
  float acc[8];
  {
    const floatx4 b0 = *(const floatx4*)(b + ch), b1 = *(const floatx4*)(b + ch + 4);
#pragma unroll
    for (int e = 0; e < 4; ++e) { acc[e] = b0[e]; acc[4 + e] = b1[e]; }
  }
#pragma unroll
  for (int dy = -1; dy <= 1; ++dy) {
    const int yy = ty + dy;
    if (yy < 0 || yy >= side) continue;
#pragma unroll
    for (int dx = -1; dx <= 1; ++dx) {
      const int xx = tx + dx;
      if (xx < 0 || xx >= side) continue;
      const half8 v = *(const half8*)(in + (tok + (long)dy * side + dx) * c + ch);
      const float* wp = w + ((dy + 1) * 3 + (dx + 1)) * c + ch;
      const floatx4 w0 = *(const floatx4*)wp, w1 = *(const floatx4*)(wp + 4);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        acc[e] += (float)v[e] * w0[e];
        acc[4 + e] += (float)v[4 + e] * w1[e];
      }
    }
  }
  half8 o;
#pragma unroll
  for (int e = 0; e < 8; ++e) o[e] = (_Float16)fmaxf(acc[e], 0.f);
  *(half8*)(out + tok * c + ch) = o;
}

// Same op, one thread per (4 consecutive tokens of a grid row x TY consecutive rows, 8 channels): each input row's 6 taps
// are loaded ONCE and feed the up-to-three output rows they belong to (a sliding window over y), so a thread issues
// (TY + 2) x 6 sixteen-byte loads for 4 TY outputs - 4.5 loads per output at TY = 1 (round 1's kernel), 3.0 at TY = 2,
// 2.25 at TY = 4 (the one-token kernel above: 9).  Every output still sums its taps in the order dy = -1, 0, 1 / dx = -1, 0,
// 1 of dwconv3x3_kernel, skipping the same out-of-range taps, so the results are bit-identical.
constexpr int DW_TX_P = 2, DW_TY_P = 4;      // product: 2 x 4 tokens per thread - measured (MI355X, 16 x 144 x 144 x 2048):
                                             // 4x1 1.26 ms (round 1's kernel), 4x2 1.20, 2x2 1.10, 2x4 1.07, 2x8 1.29, 1x4 1.52;
                                             // with round 4's interior path: 4x1 1.06, 4x2 0.97, 2x2 1.09, 2x4 0.97, 2x8 1.35
template <int DW_TX, int TY>
__global__ void __launch_bounds__(256) dwconv3x3_tile_kernel(const _Float16* __restrict__ in,
                                                             _Float16* __restrict__ out, const float* __restrict__ w,
                                                             const float* __restrict__ b, int side, int c, long total) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  const int c8 = c >> 3;
  const int ch = (int)(idx % c8) * 8;
  long rest = idx / c8;
  const int xg = (side + DW_TX - 1) / DW_TX, yg = (side + TY - 1) / TY;
  const int tx0 = (int)(rest % xg) * DW_TX;
  rest /= xg;
  const int ty0 = (int)(rest % yg) * TY;
  const long n = rest / yg;
  const long img_tok = n * side * side;                  // token index of (n, 0, 0)
  float acc[TY][DW_TX][8];
  {
    const floatx4 b0 = *(const floatx4*)(b + ch), b1 = *(const floatx4*)(b + ch + 4);
#pragma unroll
    for (int j = 0; j < TY; ++j)
#pragma unroll
      for (int k = 0; k < DW_TX; ++k)
#pragma unroll
        for (int e = 0; e < 4; ++e) { acc[j][k][e] = b0[e]; acc[j][k][4 + e] = b1[e]; }
  }
  // INTERIOR tiles (round 4): every lane of a wave works on the same tile position (the channel group is the fastest index),
  // so "no tap of this tile leaves the image" is wave-uniform.  The general path below guards every load and every tap with
  // a lane mask - 128 exec-masked branches, the 24 loads of a thread issued in groups of 2-6 with a wait after each - while
  // 92 % of the tiles at 144 x 144 need no guard at all: they take a branch-free copy with all loads issued up front.  Same taps
  // in the same order (dy, dx ascending per output): bit-identical.
  const bool interior = tx0 >= 1 && tx0 + DW_TX < side && ty0 >= 1 && ty0 + TY < side;
  if (interior) {
    half8 v[TY + 2][DW_TX + 2];
#pragma unroll
    for (int rr = 0; rr < TY + 2; ++rr)
#pragma unroll
      for (int i = 0; i < DW_TX + 2; ++i)
        v[rr][i] = *(const half8*)(in + (img_tok + (long)(ty0 - 1 + rr) * side + (tx0 - 1 + i)) * c + ch);
#pragma unroll
    for (int rr = 0; rr < TY + 2; ++rr)                   // the same accumulation order as below: input rows ascending
#pragma unroll
      for (int j = 0; j < TY; ++j) {
        const int dy = rr - 1 - j;
        if (dy < -1 || dy > 1) continue;
#pragma unroll
        for (int dx = -1; dx <= 1; ++dx) {
          const float* wp = w + ((dy + 1) * 3 + (dx + 1)) * c + ch;
          const floatx4 w0 = *(const floatx4*)wp, w1 = *(const floatx4*)(wp + 4);
#pragma unroll
          for (int k = 0; k < DW_TX; ++k)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              acc[j][k][e] += (float)v[rr][k + 1 + dx][e] * w0[e];
              acc[j][k][4 + e] += (float)v[rr][k + 1 + dx][4 + e] * w1[e];
            }
        }
      }
#pragma unroll
    for (int j = 0; j < TY; ++j)
#pragma unroll
      for (int k = 0; k < DW_TX; ++k) {
        half8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = (_Float16)fmaxf(acc[j][k][e], 0.f);
        *(half8*)(out + (img_tok + (long)(ty0 + j) * side + tx0 + k) * c + ch) = o;
      }
    return;
  }
#pragma unroll
  for (int rr = -1; rr <= TY; ++rr) {                    // input row ty0 + rr feeds output rows j = rr - 1, rr, rr + 1
    const int yy = ty0 + rr;
    if (yy < 0 || yy >= side) continue;
    half8 v[DW_TX + 2];
#pragma unroll
    for (int i = 0; i < DW_TX + 2; ++i) {
      const int xx = tx0 - 1 + i;
      if (xx >= 0 && xx < side) v[i] = *(const half8*)(in + (img_tok + (long)yy * side + xx) * c + ch);
      else
#pragma unroll
        for (int e = 0; e < 8; ++e) v[i][e] = (_Float16)0.f;
    }
#pragma unroll
    for (int j = 0; j < TY; ++j) {
      const int dy = rr - j;                             // this input row is tap row dy of output row ty0 + j
      if (dy < -1 || dy > 1) continue;
      if (ty0 + j >= side) continue;
#pragma unroll
      for (int dx = -1; dx <= 1; ++dx) {
        const float* wp = w + ((dy + 1) * 3 + (dx + 1)) * c + ch;
        const floatx4 w0 = *(const floatx4*)wp, w1 = *(const floatx4*)(wp + 4);
#pragma unroll
        for (int k = 0; k < DW_TX; ++k) {
          const int xx = tx0 + k + dx;
          if (xx < 0 || xx >= side) continue;          // same skipped taps as the scalar kernel (keeps -0/+0 identical)
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            acc[j][k][e] += (float)v[k + 1 + dx][e] * w0[e];
            acc[j][k][4 + e] += (float)v[k + 1 + dx][4 + e] * w1[e];
          }
        }
      }
    }
  }
#pragma unroll
  for (int j = 0; j < TY; ++j) {
    if (ty0 + j >= side) break;
#pragma unroll
    for (int k = 0; k < DW_TX; ++k) {
      if (tx0 + k >= side) break;
      half8 o;
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] = (_Float16)fmaxf(acc[j][k][e], 0.f);
      *(half8*)(out + (img_tok + (long)(ty0 + j) * side + tx0 + k) * c + ch) = o;
    }
  }
}

// Round 6: the same op as a SLIDING WINDOW down the image.  One wave = 512 channels (16 bytes per lane) of
// a strip of TX token columns; it walks SY rows down, holding input rows y - 1, y, y + 1 (TX + 2 columns each) in registers and
// the row after them in flight: per output row ONE new input row is loaded - (TX + 2) / TX loads per output instead of the
// tile kernel's 3 - the 72 weights of a lane's 8 channels stay in registers for the whole strip, and every column / row test is
// wave-uniform (scalar branches, no exec masks).  What the tile kernel left on the table was not instructions but LOCALITY: its
// 2 x 4 tiles re-read every input token three times through L2, and with workgroup i on XCD i mod 8 the tiles that share a halo
// sat on eight different L2s (fabric traffic ~3x the input).  Here the units are numbered so that the workgroups resident on
// one XCD at a time are x-adjacent strips of one band of rows: the two halo columns a strip shares with its neighbours are
// hits in THAT XCD's L2, the band's y halo is 2 rows in SY.  Per output the taps are summed in the order dy = -1, 0, 1 /
// dx = -1, 0, 1 from the bias, out-of-range taps skipped, each step one FMA of the f16 input with the f32 weight - the order
// and the arithmetic of dwconv3x3_kernel: bit-identical (tests/test_gpu_tokens.py).  Needs c % 512 == 0 (the decoder FFN: 2048).
// One strip: LEFT / RIGHT = the strip touches the image's left / right edge (the halo column there does not exist: its taps are
// skipped STATICALLY; its loads are clamped onto an existing column so that every load is unconditional and the waits are counted).
template <int TX, bool NT, bool LEFT, bool RIGHT>
__device__ __forceinline__ void dwconv_strip(const _Float16* __restrict__ img, _Float16* __restrict__ oimg, const float (&wt)[9][8],
                                             const float (&bs)[8], int side, int c, int x0, int y0, int y1) {
  const int ncol = RIGHT ? side - x0 : TX;           // output columns of this strip (the last strip may be narrower)
  half8 R[4][TX + 2];                                // rows y - 1, y, y + 1 and the one in flight, rotating
  size_t coff[TX + 2];                               // element offsets of the window's columns (clamped into the image)
#pragma unroll
  for (int i = 0; i < TX + 2; ++i) coff[i] = (size_t)min(max(x0 - 1 + i, 0), side - 1) * c;
  auto load_row = [&](int y, half8 (&dst)[TX + 2]) {
    const _Float16* rp = img + (size_t)min(max(y, 0), side - 1) * side * c;
#pragma unroll
    for (int i = 0; i < TX + 2; ++i) dst[i] = *(const half8*)(rp + coff[i]);
  };
  load_row(y0 - 1, R[0]);
  load_row(y0, R[1]);
  load_row(y0 + 1, R[2]);
  for (int yy = y0; yy < y1; yy += 4) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int y = yy + u;
      if (y < y1) {
        load_row(y + 2, R[(u + 3) & 3]);             // lands while this row and the next are computed
        float acc[TX][8];
#pragma unroll
        for (int k = 0; k < TX; ++k)
#pragma unroll
          for (int e = 0; e < 8; ++e) acc[k][e] = bs[e];
#pragma unroll
        for (int dy = -1; dy <= 1; ++dy) {
          if (dy == -1 && y == 0) continue;          // wave-uniform: the image's first / last row
          if (dy == 1 && y == side - 1) continue;
          const half8 (&row)[TX + 2] = R[(u + 1 + dy) & 3];
#pragma unroll
          for (int dx = -1; dx <= 1; ++dx) {
            const int t = (dy + 1) * 3 + (dx + 1);
#pragma unroll
            for (int k = 0; k < TX; ++k) {
              if (LEFT && k + dx < 0) continue;                    // column x0 - 1 does not exist
              if (RIGHT && k + dx >= ncol) continue;               // ... nor column x0 + ncol (wave-uniform)
#pragma unroll
              for (int e = 0; e < 8; ++e) acc[k][e] = __builtin_fmaf((float)row[k + 1 + dx][e], wt[t][e], acc[k][e]);
            }
          }
        }
        _Float16* op = oimg + ((size_t)y * side + x0) * c;
#pragma unroll
        for (int k = 0; k < TX; ++k) {
          if (RIGHT && k >= ncol) continue;
          half8 o;
#pragma unroll
          for (int e = 0; e < 8; ++e) o[e] = (_Float16)fmaxf(acc[k][e], 0.f);
          if (NT) __builtin_nontemporal_store(o, (half8*)(op + (size_t)k * c));
          else *(half8*)(op + (size_t)k * c) = o;
        }
      }
    }
  }
}

template <int TX, bool NT>
__global__ void __launch_bounds__(256) dwconv3x3_strip_kernel(const _Float16* __restrict__ in, _Float16* __restrict__ out,
                                                              const float* __restrict__ w, const float* __restrict__ b, int side,
                                                              int c, int sy, int xstrips, int ybands, int nwaves) {
  // XCD-aware numbering: hardware block id -> logical id such that the blocks one XCD runs are consecutive logical ids
  int id = blockIdx.x;
  {
    const int nwg = gridDim.x, q = nwg / 8, rr = nwg % 8, xcd = id % 8, k = id / 8;
    id = (xcd < rr ? xcd * (q + 1) : rr * (q + 1) + (xcd - rr) * q) + k;
  }
  const int lane = threadIdx.x & 63;
  const int wid = __builtin_amdgcn_readfirstlane(id * 4 + (int)(threadIdx.x >> 6));
  if (wid >= nwaves) return;
  const int chunks = c >> 9;                         // 512-channel chunks per token
  const int chunk = wid % chunks;
  int unit = wid / chunks;
  const int xs = unit % xstrips;
  unit /= xstrips;
  const int yb = unit % ybands;
  const int n = unit / ybands;
  const int x0 = xs * TX, y0 = yb * sy, y1 = min(y0 + sy, side);
  const int ch = chunk * 512 + lane * 8;
  const _Float16* img = in + (size_t)n * side * side * c + ch;
  _Float16* oimg = out + (size_t)n * side * side * c + ch;
  float wt[9][8], bs[8];
#pragma unroll
  for (int t = 0; t < 9; ++t) {
    const floatx4 w0 = *(const floatx4*)(w + t * c + ch), w1 = *(const floatx4*)(w + t * c + ch + 4);
#pragma unroll
    for (int e = 0; e < 4; ++e) { wt[t][e] = w0[e]; wt[t][4 + e] = w1[e]; }
  }
  {
    const floatx4 b0 = *(const floatx4*)(b + ch), b1 = *(const floatx4*)(b + ch + 4);
#pragma unroll
    for (int e = 0; e < 4; ++e) { bs[e] = b0[e]; bs[4 + e] = b1[e]; }
  }
  const bool left = x0 == 0, right = x0 + TX >= side;          // wave-uniform
  if (!left && !right) dwconv_strip<TX, NT, false, false>(img, oimg, wt, bs, side, c, x0, y0, y1);
  else if (left && !right) dwconv_strip<TX, NT, true, false>(img, oimg, wt, bs, side, c, x0, y0, y1);
  else if (!left) dwconv_strip<TX, NT, false, true>(img, oimg, wt, bs, side, c, x0, y0, y1);
  else dwconv_strip<TX, NT, true, true>(img, oimg, wt, bs, side, c, x0, y0, y1);
}

// ----------------------------------------------------------------------------------------------
// K12: adaptive 2-D positional encoding (idf/cross_attn.py:143-157)
// colsum_partial: part[n][chunk][c] = sum over the chunk's tokens of z[n,t,c]   (deterministic 2-stage)
// colmean_final : pooled[n][c] = (sum_chunks part) / T
// posenc_add    : z[n,t,c] += hs[n,c]*htab[ty][c] + ws[n,c]*wtab[tx][c]
// ----------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) colsum_partial_kernel(const float* __restrict__ z, float* __restrict__ part,
                                                             int T, int c, int chunks) {
  const int n = blockIdx.z, chunk = blockIdx.y;
  const int col = blockIdx.x * 256 + threadIdx.x;
  if (col >= c) return;
  const int per = (T + chunks - 1) / chunks;
  const int t0 = chunk * per, t1 = min(T, t0 + per);
  const float* p = z + ((size_t)n * T + t0) * c + col;
  float s = 0.f;
  for (int t = t0; t < t1; ++t, p += c) s += *p;
  part[((size_t)n * chunks + chunk) * c + col] = s;
}

__global__ void __launch_bounds__(256) colmean_final_kernel(const float* __restrict__ part,
                                                            float* __restrict__ pooled, int T, int c, int chunks,
                                                            int total) {
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= total) return;
  const int n = idx / c, col = idx % c;
  float s = 0.f;
  for (int k = 0; k < chunks; ++k) s += part[((size_t)n * chunks + k) * c + col];
  pooled[idx] = s / (float)T;
}

__global__ void __launch_bounds__(256) posenc_add_kernel(float* __restrict__ z, const float* __restrict__ hs,
                                                         const float* __restrict__ ws,
                                                         const float* __restrict__ htab,
                                                         const float* __restrict__ wtab, int side, int c,
                                                         long total4) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total4) return;
  const int c4 = c >> 2;
  const long tok = idx / c4;
  const int ch = (int)(idx % c4) * 4;
  const int T = side * side;
  const int n = (int)(tok / T), t = (int)(tok % T);
  const int ty = t / side, tx = t % side;
  floatx4 v = *(floatx4*)(z + tok * c + ch);
  const floatx4 a = *(const floatx4*)(hs + (size_t)n * c + ch), hb = *(const floatx4*)(htab + (size_t)ty * c + ch);
  const floatx4 b = *(const floatx4*)(ws + (size_t)n * c + ch), wb = *(const floatx4*)(wtab + (size_t)tx * c + ch);
#pragma unroll
  for (int e = 0; e < 4; ++e) v[e] = (v[e] + a[e] * hb[e]) + b[e] * wb[e];
  *(floatx4*)(z + tok * c + ch) = v;
}

// ----------------------------------------------------------------------------------------------
// Tiny-M linear:  y[m][o] = act_out( sum_k W[o][k] * act_in(x[m][(k % kmod)]) + b[o] ),  m < M <= 8 per pass.
// One wave per output feature; W rows are read once per group of 8 samples.
// act_in: 0 none, 1 SiLU, 2 timestep sinusoid (x is t[m]; k<half -> cos(t f_k), else sin(t f_{k-half})
//         with f_k = exp(-ln(1e4) k / half), idf/cross_model.py:111-129)
// act_out: 0 none, 1 SiLU, 2 ReLU, 3 sigmoid.   kmod: input width before tiling (t.repeat(1,4), :331).
// ----------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) small_linear_kernel(const float* __restrict__ x, int ldx,
                                                           const float* __restrict__ w,
                                                           const float* __restrict__ b, float* __restrict__ y,
                                                           int ldy, int M, int K, int N, int kmod, int act_in,
                                                           int act_out) {
  const int o = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (o >= N) return;
  const int lane = threadIdx.x & 63;
  const float* wr = w + (size_t)o * K;
  // groups of 8 samples: blockIdx.y of them run side by side (round 5: at 64 samples one wave per output feature walked
  // eight groups one after the other, 217 us per call at the reference's operating point with 32 documents per batch)
  for (int m0 = 8 * blockIdx.y; m0 < M; m0 += 8 * gridDim.y) {
    float acc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = 0.f;
    for (int k = lane; k < K; k += 64) {
      const float wv = wr[k];
      const int kk = k % kmod;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        if (m0 + j < M) {
          float xv;
          if (act_in == 2) {
            const int half = kmod >> 1;
            const float t = x[(m0 + j) * ldx];
            const int kf = kk < half ? kk : kk - half;
            const float fr = expf(-9.210340371976184f * (float)kf / (float)half);
            xv = kk < half ? cosf(t * fr) : sinf(t * fr);
          } else {
            xv = x[(size_t)(m0 + j) * ldx + kk];
            if (act_in == 1) xv = xv / (1.f + expf(-xv));
          }
          acc[j] += wv * xv;
        }
      }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float s = wave_sum(acc[j]);
      if (lane == 0 && m0 + j < M) {
        float v = s + (b ? b[o] : 0.f);
        if (act_out == 1) v = v / (1.f + expf(-v));
        else if (act_out == 2) v = fmaxf(v, 0.f);
        else if (act_out == 3) v = 1.f / (1.f + expf(-v));
        y[(size_t)(m0 + j) * ldy + o] = v;
      }
    }
  }
}

// ----------------------------------------------------------------------------------------------
// K15 + K16: z -> LN(affine, 1e-5) [decoder.layer_norm, idf/cross_attn.py:457] -> LN(no affine, 1e-6)
// -> modulate(shift, scale) -> Linear 1536->8 + bias [FinalLayer2, idf/cross_model.py:329-336]
// -> unpatchify nhwpqc->nchpwq [:553-566] -> + init_flow [:645-646].   One wave per token.
// ----------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) final_tokens_kernel(const float* __restrict__ z,
                                                           const float* __restrict__ gamma,
                                                           const float* __restrict__ beta,
                                                           const float* __restrict__ shift,
                                                           const float* __restrict__ scale, int ldmod,
                                                           int mod_rows, const float* __restrict__ w,
                                                           const float* __restrict__ b,
                                                           const float* __restrict__ init_flow,
                                                           float* __restrict__ x0, float* __restrict__ tok8, int g,
                                                           long rows) {
  constexpr int C = 1536, PER = 24;
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int lane = threadIdx.x & 63;
  const float* ip = z + row * C;
  float v[PER];
  float s = 0.f;
#pragma unroll
  for (int k = 0; k < 6; ++k) {
    const floatx4 f = *(const floatx4*)(ip + (k * 64 + lane) * 4);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      v[k * 4 + e] = f[e];
      s += f[e];
    }
  }
  float mean = wave_sum(s) * (1.f / C);
  float q2 = 0.f;
#pragma unroll
  for (int k = 0; k < PER; ++k) {
    const float d = v[k] - mean;
    q2 += d * d;
  }
  float rstd = rsqrtf(wave_sum(q2) * (1.f / C) + 1e-5f);
  s = 0.f;
#pragma unroll
  for (int k = 0; k < 6; ++k)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int c = (k * 64 + lane) * 4 + e;
      v[k * 4 + e] = (v[k * 4 + e] - mean) * rstd * gamma[c] + beta[c];
      s += v[k * 4 + e];
    }
  mean = wave_sum(s) * (1.f / C);
  q2 = 0.f;
#pragma unroll
  for (int k = 0; k < PER; ++k) {
    const float d = v[k] - mean;
    q2 += d * d;
  }
  rstd = rsqrtf(wave_sum(q2) * (1.f / C) + 1e-6f);
  const float* sh = shift + (row / mod_rows) * ldmod;
  const float* sc = scale + (row / mod_rows) * ldmod;
  float dot[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) dot[j] = 0.f;
#pragma unroll
  for (int k = 0; k < 6; ++k)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int c = (k * 64 + lane) * 4 + e;
      const float y = ((v[k * 4 + e] - mean) * rstd) * (1.f + sc[c]) + sh[c];
#pragma unroll
      for (int j = 0; j < 8; ++j) dot[j] += y * w[j * C + c];
    }
#pragma unroll
  for (int j = 0; j < 8; ++j) dot[j] = wave_sum(dot[j]);
  if (lane < 8) {
    float o = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j)
      if (lane == j) o = dot[j];
    o += b[lane];
    if (tok8) tok8[row * 8 + lane] = o;
    const int side = g >> 1, T = side * side;
    const int n = (int)(row / T), t = (int)(row % T);
    const int p = lane >> 2, q = (lane >> 1) & 1, c = lane & 1;   // token channel = (p*2 + q)*2 + c
    const size_t at = (((size_t)n * 2 + c) * g + 2 * (t / side) + p) * g + 2 * (t % side) + q;
    x0[at] = o + (init_flow ? init_flow[at] : 0.f);
  }
}

}  // namespace dvd

using namespace dvd;

#define LAUNCH_ROWS(rows) cdiv((rows), 4), 256, 0, (hipStream_t)stream

extern "C" int dvd_embed_obs_ln(const float* x, const float* w, const float* bias, const float* pos, float* tok32,
                                void* ln16, int n, int g, void* stream) {
  DVD_REQUIRE(x && w && bias && pos && tok32 && ln16, "embed_obs_ln: null pointer");
  DVD_REQUIRE(n > 0 && g >= 2 && g % 2 == 0, "embed_obs_ln: bad shape");
  const long rows = (long)n * (g / 2) * (g / 2);
  embed_obs_ln_kernel<<<LAUNCH_ROWS(rows)>>>(x, w, bias, pos, tok32, (_Float16*)ln16, g, rows);
  return check_launch("embed_obs_ln");
}

extern "C" int dvd_layernorm_rows(const float* in, int ldin, long stride_in, void* out16, int ldout, long stride_out,
                                  int batch, long rows, int c, const float* gamma, const float* beta,
                                  const float* shift, const float* scale, int ldmod, int mod_rows, float eps,
                                  void* stream) {
  DVD_REQUIRE(in && out16, "layernorm_rows: null pointer");
  DVD_REQUIRE(c == 384 || c == 1536, "layernorm_rows: C=%d not in {384,1536}", c);
  DVD_REQUIRE((gamma == nullptr) == (beta == nullptr) && (shift == nullptr) == (scale == nullptr),
              "layernorm_rows: affine / modulate pointers must come in pairs");
  DVD_REQUIRE(!shift || mod_rows > 0, "layernorm_rows: mod_rows");
  DVD_REQUIRE(rows > 0 && batch > 0 && batch < 65536, "layernorm_rows: bad rows/batch");
  DVD_REQUIRE(ldin % 4 == 0 && stride_in % 2 == 0 && ldout % 4 == 0 && stride_out % 2 == 0 && ldmod % 4 == 0 &&
                  ((uintptr_t)in % 16) == 0 && ((uintptr_t)out16 % 8) == 0 && ((uintptr_t)gamma % 16) == 0 &&
                  ((uintptr_t)beta % 16) == 0 && ((uintptr_t)shift % 16) == 0 && ((uintptr_t)scale % 16) == 0,
              "layernorm_rows: rows must be 16-byte aligned (vector loads)");
  dim3 grid(cdiv(rows, 4), batch);
  if (c == 384)
    layernorm_rows_kernel<384><<<grid, 256, 0, (hipStream_t)stream>>>(in, ldin, stride_in, (_Float16*)out16, ldout,
                                                                       stride_out, gamma, beta, shift, scale, ldmod,
                                                                       mod_rows > 0 ? mod_rows : 1, eps, rows);
  else
    layernorm_rows_kernel<1536><<<grid, 256, 0, (hipStream_t)stream>>>(in, ldin, stride_in, (_Float16*)out16, ldout,
                                                                        stride_out, gamma, beta, shift, scale, ldmod,
                                                                        mod_rows > 0 ? mod_rows : 1, eps, rows);
  return check_launch("layernorm_rows");
}

extern "C" int dvd_build_r_rows(const float* feat_nhwc, const float* init_feat_nchw, const float* flow, void* out16,
                                int ldo, int n, int g, int n_hyp, int mode, void* stream) {
  DVD_REQUIRE(feat_nhwc && flow && out16, "build_r_rows: null pointer");
  DVD_REQUIRE(ldo >= 1032 && n > 0 && g >= 2 && g % 2 == 0 && n_hyp > 0 && mode >= 0 && mode <= 3,
              "build_r_rows: bad arguments");
  DVD_REQUIRE(mode != 3 || init_feat_nchw, "build_r_rows: mode 3 needs init_feat");
  const long items = (long)n * (g / 2) * (g / 2) * 4;
  build_r_rows_kernel<<<LAUNCH_ROWS(items)>>>(feat_nhwc, init_feat_nchw, flow, (_Float16*)out16, ldo, g, n_hyp, mode,
                                              items);
  return check_launch("build_r_rows");
}

extern "C" int dvd_patch_rows(const float* in, long sn, long sc, long sy, long sx, float* out, int ldo, int n, int c,
                              int g, void* stream) {
  DVD_REQUIRE(in && out, "patch_rows: null pointer");
  DVD_REQUIRE(n > 0 && c > 0 && g >= 2 && g % 2 == 0 && ldo >= 4 * c, "patch_rows: bad shape");
  const long total = (long)n * (g / 2) * (g / 2) * 4 * c;
  patch_rows_kernel<<<cdiv(total, 256), 256, 0, (hipStream_t)stream>>>(in, sn, sc, sy, sx, out, ldo, c, g, total);
  return check_launch("patch_rows");
}

extern "C" int dvd_dwconv3x3(const void* in16, void* out16, const float* w9c, const float* b, int n, int side, int c,
                             void* stream) {
  DVD_REQUIRE(in16 && out16 && w9c && b, "dwconv3x3: null pointer");
  DVD_REQUIRE(n > 0 && side > 0 && c % 8 == 0, "dwconv3x3: bad shape");
#ifdef DVD_LAB
  if (getenv("DVD_DWCONV_V1")) {   // lab build: the one-token-per-thread kernel (same tap order, bit-identical, 1.5x slower)
    const long total1 = (long)n * side * side * (c / 8);
    dwconv3x3_kernel<<<cdiv(total1, 256), 256, 0, (hipStream_t)stream>>>((const _Float16*)in16, (_Float16*)out16, w9c, b,
                                                                         side, c, total1);
    return check_launch("dwconv3x3(lab v1)");
  }
#endif
  // product rule (round 6): the sliding-window kernel - 3 columns x 24 rows per wave, streaming stores: 0.505-0.51 ms at
  // 16 x 144 x 144 x 2048 (5.3-5.4 TB/s) against the tile kernel's 0.97 (profiles/r6_dwconv_strip.txt) - wherever its waves fill
  // the chip; small maps keep the tile kernel (more, shorter threads).  All kernels give the same bits, so the rule may look
  // at the batch.
  int stx = (c % 512 == 0 && (long)n * side * side >= 16384) ? 3 : 0, ssy = 24;
  bool snt = true;
  (void)snt;
#ifdef DVD_LAB
  // lab: DVD_DWCONV_STRIP = 0 (tile kernel) | 2 | 3 | 4 columns; DVD_DWCONV_SY the band height; DVD_DWCONV_NT = 0 | 1
  if (getenv("DVD_DWCONV_TY") || getenv("DVD_DWCONV_TX")) stx = 0;
  if (const char* e = getenv("DVD_DWCONV_STRIP")) stx = (c % 512 == 0) ? atoi(e) : 0;
  if (const char* e = getenv("DVD_DWCONV_SY")) ssy = atoi(e);
  if (const char* e = getenv("DVD_DWCONV_NT")) snt = atoi(e) != 0;
#endif
  if (stx >= 2 && stx <= 4 && ssy >= 1) {
    const int xstrips = cdiv(side, stx), ybands = cdiv(side, ssy);
    const int nwaves = n * ybands * xstrips * (c / 512);
    const dim3 g(cdiv(nwaves, 4));
#define DW_STRIP(TX_, NT_) dwconv3x3_strip_kernel<TX_, NT_><<<g, 256, 0, (hipStream_t)stream>>>((const _Float16*)in16, (_Float16*)out16, w9c, b, side, c, ssy, xstrips, ybands, nwaves)
#ifdef DVD_LAB
    if (stx == 2) { if (snt) DW_STRIP(2, true); else DW_STRIP(2, false); }
    else if (stx == 4) { if (snt) DW_STRIP(4, true); else DW_STRIP(4, false); }
    else if (!snt) DW_STRIP(3, false);
    else
#endif
    DW_STRIP(3, true);
#undef DW_STRIP
    return check_launch("dwconv3x3(strip)");
  }
  int ty = DW_TY_P, tx = DW_TX_P;
#ifdef DVD_LAB
  if (const char* e = getenv("DVD_DWCONV_TY")) ty = atoi(e);      // lab: 1 = round 1's row kernel, 2, 4
  if (const char* e = getenv("DVD_DWCONV_TX")) tx = atoi(e);      // lab: 2 or 4 tokens along the row
#endif
  const long total = (long)n * cdiv(side, ty) * cdiv(side, tx) * (c / 8);
  const dim3 grd(cdiv(total, 256));
#define DW_LAUNCH(TX_, TY_) dwconv3x3_tile_kernel<TX_, TY_><<<grd, 256, 0, (hipStream_t)stream>>>((const _Float16*)in16, (_Float16*)out16, w9c, b, side, c, total)
#ifdef DVD_LAB
  if (tx == 2 && ty == 2) DW_LAUNCH(2, 2);
  else if (tx == 2 && ty == 4) DW_LAUNCH(2, 4);
  else if (tx == 2 && ty == 8) DW_LAUNCH(2, 8);
  else if (tx == 1 && ty == 4) DW_LAUNCH(1, 4);
  else if (tx == 1 && ty == 8) DW_LAUNCH(1, 8);
  else if (tx == 2) DW_LAUNCH(2, 1);
  else if (ty == 1) DW_LAUNCH(4, 1);
  else if (ty == 4) DW_LAUNCH(4, 4);
  else
#endif
  DW_LAUNCH(DW_TX_P, DW_TY_P);
#undef DW_LAUNCH
  return check_launch("dwconv3x3");
}

extern "C" int dvd_colmean(const float* z, float* partial, float* pooled, int n, int t, int c, int chunks,
                           void* stream) {
  DVD_REQUIRE(z && partial && pooled, "colmean: null pointer");
  DVD_REQUIRE(n > 0 && n < 65536 && t > 0 && c > 0 && chunks > 0 && chunks < 65536, "colmean: bad shape");
  dim3 grid(cdiv(c, 256), chunks, n);
  colsum_partial_kernel<<<grid, 256, 0, (hipStream_t)stream>>>(z, partial, t, c, chunks);
  colmean_final_kernel<<<cdiv((long)n * c, 256), 256, 0, (hipStream_t)stream>>>(partial, pooled, t, c, chunks, n * c);
  return check_launch("colmean");
}

extern "C" int dvd_posenc_add(float* z, const float* hs, const float* ws, const float* htab, const float* wtab,
                              int n, int side, int c, void* stream) {
  DVD_REQUIRE(z && hs && ws && htab && wtab, "posenc_add: null pointer");
  DVD_REQUIRE(n > 0 && side > 0 && c % 4 == 0, "posenc_add: bad shape");
  const long total4 = (long)n * side * side * (c / 4);
  posenc_add_kernel<<<cdiv(total4, 256), 256, 0, (hipStream_t)stream>>>(z, hs, ws, htab, wtab, side, c, total4);
  return check_launch("posenc_add");
}

extern "C" int dvd_small_linear(const float* x, int ldx, const float* w, const float* b, float* y, int ldy, int m,
                                int k, int n, int kmod, int act_in, int act_out, void* stream) {
  DVD_REQUIRE(x && w && y, "small_linear: null pointer");
  DVD_REQUIRE(m > 0 && k > 0 && n > 0 && kmod > 0 && act_in >= 0 && act_in <= 2 && act_out >= 0 && act_out <= 3,
              "small_linear: bad arguments");
  small_linear_kernel<<<dim3(cdiv(n, 4), cdiv(m, 8)), 256, 0, (hipStream_t)stream>>>(x, ldx, w, b, y, ldy, m, k, n, kmod, act_in,
                                                                                       act_out);
  return check_launch("small_linear");
}

extern "C" int dvd_final_tokens(const float* z, const float* gamma, const float* beta, const float* shift,
                                const float* scale, int ldmod, int mod_rows, const float* w8, const float* b8,
                                const float* init_flow, float* x0, float* tok8, int n, int g, void* stream) {
  DVD_REQUIRE(z && gamma && beta && shift && scale && w8 && b8 && x0, "final_tokens: null pointer");
  DVD_REQUIRE(n > 0 && g >= 2 && g % 2 == 0 && mod_rows > 0, "final_tokens: bad shape");
  const long rows = (long)n * (g / 2) * (g / 2);
  final_tokens_kernel<<<LAUNCH_ROWS(rows)>>>(z, gamma, beta, shift, scale, ldmod, mod_rows, w8, b8, init_flow, x0,
                                             tok8, g, rows);
  return check_launch("final_tokens");
}
