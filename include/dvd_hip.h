/* dvd_hip.h - C ABI of libdvd_hip.so, the MI355X (gfx950) engine for the DvD
 * coordinate-diffusion sampling path.
 *
 * The reference (hanquansanren/DvD) is pure Python on PyTorch and has no FFI of its own;
 * the functions below sit UNDER the Python callables the reference's plug-in surface uses
 *  and each names the reference call site it replaces.  Paths are relative
 * to the reference root; idf/ = train_settings/dvd/improved_diffusion/.
 *
 * Conventions
 *   - extern "C", plain pointers and sizes, no exceptions cross the boundary.
 *   - Every function returns 0 on success or a negative DVD_E_* code; dvd_last_error()
 *     returns a thread-local message for the last failure.
 *   - All tensor pointers are BORROWED DEVICE pointers, valid for the duration of the call;
 *     functions only ENQUEUE work on `stream` (a hipStream_t passed as void*) and never
 *     synchronise, allocate or free device memory (workspaces are caller-provided), so every
 *     entry point may be captured into a hipGraph.
 *   - "tok" layouts are token-major [rows, channels] with `ld` = row stride in elements.
 */
#ifndef DVD_HIP_H
#define DVD_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DVD_OK 0
#define DVD_E_ARG (-1)     /* bad argument (null pointer, unsupported size/alignment) */
#define DVD_E_LAUNCH (-2)  /* HIP reported a launch error */
#define DVD_E_STATE (-3)   /* engine used out of order (weights missing, docs not prepared) */

const char* dvd_last_error(void);
/* library/ABI version: major*1000 + minor */
int dvd_version(void);

/* ------------------------------------------------------------------------------------------
 * Warps
 * ---------------------------------------------------------------------------------------- */

/* Drop-in for register_model2 / SpatialTransformer2.forward
 * (datasets/utils/warping.py:14-23,50-73): F.grid_sample(src, grid.permute(0,2,3,1),
 * mode='bilinear', padding_mode='zeros', align_corners=True).
 * src [N,C,Hin,Win] f32, grid [N,2,H,W] f32 (channel 0 = x, 1 = y, normalised [-1,1]),
 * out [N,C,H,W] f32.  src_batch_div: source image of sample n is n / src_batch_div
 * (1 = reference behaviour; H_hyp lets the hypotheses of one document share one source). */
int dvd_grid_sample_bilinear_zeros_ac(const float* src, const float* grid, float* out,
                                      int n, int c, int hin, int win, int h, int w,
                                      int src_batch_div, void* stream);

/* Fused full-resolution unwarp tail (train_settings/dvd/evaluation.py:301-306 +
 * utils_flow/visualization_utils.py:75-77):
 *   s    = bilinear_upsample(flow[2,G,G] -> H x W, align_corners=True)
 *   grid = ((s + base) * 2 - 1) * scale,  base_x = j/(W-1), base_y = i/(H-1)
 *   out  = grid_sample(src, grid, bilinear, zeros, align_corners=True)
 * f32 variant: src [3,H,W] f32 planar (values 0..255), out [H,W,3] f32 (HWC, what the
 * reference hands to numpy); u8 variant: src [H,W,3] u8, out [H,W,3] u8 with C-style
 * truncation of the f32 result (numpy .astype(uint8) on in-range values). */
int dvd_unwarp_f32(const float* flow, int g, const float* src_chw, float* out_hwc,
                   int h, int w, float scale, void* stream);
int dvd_unwarp_u8(const float* flow, int g, const uint8_t* src_hwc, uint8_t* out_hwc,
                  int h, int w, float scale, void* stream);
/* The same for n documents of one size in ONE launch (flow [n,2,G,G], src / out [n,...] contiguous):
 * the per-document loop of run_evaluation_docunet (evaluation.py:245-306) for a batch of documents. */
int dvd_unwarp_f32_batch(const float* flow, int g, const float* src_chw, float* out_hwc,
                         int n, int h, int w, float scale, void* stream);
int dvd_unwarp_u8_batch(const float* flow, int g, const uint8_t* src_hwc, uint8_t* out_hwc,
                        int n, int h, int w, float scale, void* stream);
/* Materialise the full-resolution sampling grid only ([2,H,W] f32), i.e. evaluation.py:301-306. */
int dvd_unwarp_grid(const float* flow, int g, float* grid_out, int h, int w, float scale, void* stream);

/* ------------------------------------------------------------------------------------------
 * Scheduler step (idf/gaussian_diffusion.py:434-438,445-491 ddim_sample; :270-292 posterior
 * mean for the DDPM variant).  One fused elementwise kernel over [n_elem] floats:
 *   DDIM: eps = (c_recip*x_t - x0)/c_recipm1 ; x_prev = x0*sqrt_abar_prev + dir_coef*eps + sigma*noise
 *   DDPM: x_prev = coef1*x0 + coef2*x_t + sigma*noise
 * Coefficients are the float32 casts of the float64 schedule tables (`_extract_into_tensor`,
 * :1181-1197); sigma already contains the (t != 0) mask.  noise may be NULL when sigma == 0.
 * If next_grid != NULL it also receives (x0 + base_G)*2 - 1 ([N,2,G,G], idf/gaussian_diffusion.py:622),
 * the sampling grid of the next step's feature warp. */
typedef struct {
  int kind;             /* 0 = DDIM, 1 = DDPM */
  float c_recip;        /* sqrt(1/abar_t)        DDIM */
  float c_recipm1;      /* sqrt(1/abar_t - 1)    DDIM */
  float sqrt_abar_prev; /* sqrt(abar_{t-1})      DDIM */
  float dir_coef;       /* sqrt(1 - abar_{t-1} - sigma^2) DDIM */
  float coef1;          /* posterior_mean_coef1  DDPM */
  float coef2;          /* posterior_mean_coef2  DDPM */
  float sigma;          /* noise scale incl. (t != 0) mask */
} dvd_sched_coef;

int dvd_sched_step(const dvd_sched_coef* coef, const float* x_t, const float* x0, const float* noise,
                   float* x_prev, float* next_grid, int n, int g, void* stream);

/* mean over the H hypotheses of each document + clamp to [-1,1]
 * (idf/gaussian_diffusion.py:639-640): x0 [docs*H,2,G,G] -> out [docs,2,G,G]. */
int dvd_hyp_mean_clamp(const float* x0, float* out, int docs, int n_hyp, int g, void* stream);

/* ------------------------------------------------------------------------------------------
 * GEMM with fused epilogue:  C[M,N] = epi(A[M,K] . B[N,K]^T)  - replaces every nn.Linear / 1x1 conv
 * / patch-embed conv on the path together with the ATen ops that follow it in the reference:
 *   timm Attention.qkv/proj, Mlp.fc1(+GELU tanh)/fc2 (idf/cross_model.py:163-174,268-292),
 *   nn.MultiheadAttention in/out projections (:203-205,237-265), PatchEmbed + pos-embed (:571-605),
 *   decoder linear_q/k/v/fc and conv1/conv2 + folded BatchNorm + ReLU (idf/cross_attn.py:52-57,197-221),
 *   adaLN gate * (.) + residual (idf/cross_model.py:268-292).
 * dtype 0: A,B f16, fp32 accumulate (MFMA 32x32x16 f16);  dtype 1: A,B f32, exact fp32 MFMA.
 *   out = acc (+ bias[col] or bias[row]) -> act -> (+ pos[row % pos_rows][col]) -> (* gate[row / gate_rows][col])
 *         -> (+ res[row][col]);  stored to C32 (f32) and/or C16 (f16).
 * B_lo/lo_scale: weights split into two f16 parts (hi + lo_scale * lo) give fp32-grade weights at twice the MFMA
 * work; used for every per-step weight because f16 weight rounding is a SYSTEMATIC error that accumulates
 * linearly over the diffusion steps (precision).
 * Batched over `batch` with element strides (0 = shared operand).  K % 64 == 0 (f16) / % 16 (f32);
 * rows of A/B 16-byte aligned. */
typedef struct {
  int dtype;
  int M, N, K, batch;
  const void* A; int lda; long strideA;
  const void* B; int ldb; long strideB;
  const void* B_lo; const void* A_lo; float lo_scale; /* optional split weight (f16 only), on ONE side:
                                                          X_true = X + lo_scale * X_lo */
  float* C32; int ldc; long strideC32;
  void* C16; int ldc16; long strideC16;
  const float* bias; int bias_row; long strideBias;
  int act;                                   /* 0 none, 1 GELU(tanh), 2 ReLU */
  const float* pos; int ldpos; int pos_rows;
  const float* gate; int ldgate; int gate_rows; long strideGate;
  const float* res; int ldres; long strideRes;
  int small_tiles;                           /* f16 only: 1 = 128x128 tiles for every shape.  The caller's statement about
                                                its PROBLEM FAMILY (the engine derives 1 from the grid alone): at a few
                                                thousand rows the 256x256 persistent kernels put < 50 workgroups on 256 CUs
                                                and run one workgroup's K loop latency-bound.  2 = the same family with MANY
                                                rows (the engine: >= 16 384 token rows, i.e. the choice between 1 and 2 DOES
                                                depend on the batch): N % 256 == 0 shapes take the 256x256 kernel in the
                                                two-sweep form whose accumulation sequence IS the 128x128 kernel's - the
                                                same bits, which is what keeps 'alone == in a batch' true
                                                (tests/test_gpu_gemm.py::test_gemm_small_family_engine_shapes_same_bits,
                                                tests/test_gpu_engine.py::test_large_batch_of_small_grids_matches_single) */
} dvd_gemm_desc;

int dvd_gemm_nt(const dvd_gemm_desc* desc, void* stream);

/* ------------------------------------------------------------------------------------------
 * Flash attention core  O = softmax(scale * Q K^T) V  per (batch, head); f16 in/out, fp32 softmax.
 * Replaces the score/softmax/value products inside nn.MultiheadAttention (idf/cross_model.py:237-265),
 * timm Attention (:268-289) and SATRN ScaledDotProductAttention (idf/cross_attn.py:73-83; its
 * all-ones mask is a no-op and dropout is identity in eval mode).
 *   Q  [batch, tq, heads*head_dim]  row stride ldq  (may point into a fused qkv buffer)
 *   K  [kv_batch, tk, heads*head_dim] row stride ldk
 *   Vt [kv_batch, heads*head_dim, tk] row stride ldvt  (V TRANSPOSED: keys contiguous)
 *   O  [batch, tq, heads*head_dim]  row stride ldo
 * kv batch of query batch b is b / kv_batch_div (hypotheses of one document share its conditioning
 * K/V).  head_dim in {64, 256}; tk % 8 == 0. */
typedef struct {
  int head_dim, heads, batch, tq, tk, kv_batch_div;
  const void* Q; int ldq; long strideQ;
  const void* K; int ldk; long strideK;
  const void* Vt; int ldvt; long strideVt;
  void* O; int ldo; long strideO;
  float scale;
} dvd_attn_desc;

int dvd_flash_attn(const dvd_attn_desc* desc, void* stream);
/* Name of the kernel dvd_flash_attn launches for a problem shape (the choice depends on head_dim, tq and tk only -
 * never on the batch or the environment).  Measurement tooling matches rocprofv3 / PMC records against it. */
const char* dvd_flash_attn_kernel_name(int head_dim, int tq, int tk);

/* ------------------------------------------------------------------------------------------
 * Token-side kernels (each also reachable on its own for parity tests).  "rows" = tokens of all
 * samples, token-major; f16 outputs feed the GEMM / attention operands.
 * ---------------------------------------------------------------------------------------- */

/* obs_embedder (PatchEmbed k=s=2, 2->384) + bias + pos-embed, and cross_norm LayerNorm (no affine,
 * eps 1e-6) of the result (idf/cross_model.py:571,238).  x [N,2,G,G]; w [384,8]; pos [T,384];
 * tok32 [N*T,384] f32; ln16 [N*T,384] f16. */
int dvd_embed_obs_ln(const float* x, const float* w, const float* bias, const float* pos, float* tok32,
                     void* ln16, int n, int g, void* stream);

/* LayerNorm over c in {384,1536} channels (+affine gamma/beta) (+adaLN modulate y*(1+scale)+shift with
 * row / mod_rows selecting the modulation row) -> f16.  nn.LayerNorm + modulate() call sites:
 * idf/cross_model.py:13-14,268-292; idf/cross_attn.py:386-391.  Batched over `batch` slices. */
int dvd_layernorm_rows(const float* in, int ldin, long stride_in, void* out16, int ldout, long stride_out,
                       int batch, long rows, int c, const float* gamma, const float* beta, const float* shift,
                       const float* scale, int ldmod, int mod_rows, float eps, void* stream);

/* Rows of the r_embedder GEMM: per token and 2x2 patch position the 258 channels
 * cat([init_flow, init_feat]) (idf/cross_model.py:596-603) where init_feat is, by `mode`,
 * 0: zeros, 1: feat itself (t > 600), 2: bilinear warp of feat by (init_flow + base)*2-1
 * (idf/gaussian_diffusion.py:618-624), 3: the explicit tensor init_feat_nchw [N,256,G,G].
 * feat is channels-last [docs,G,G,256]; out f16 [N*T, ldo>=1032]. */
int dvd_build_r_rows(const float* feat_nhwc, const float* init_feat_nchw, const float* flow, void* out16, int ldo,
                     int n, int g, int n_hyp, int mode, void* stream);

/* 2x2 patch rows of a map with arbitrary element strides (PatchEmbed operand, idf/cross_model.py:585,594,605):
 * out[(n*T+t)*ldo + (p*2+q)*c + ch] = in[n*sn + ch*sc + (2ty+p)*sy + (2tx+q)*sx]. */
int dvd_patch_rows(const float* in, long sn, long sc, long sy, long sx, float* out, int ldo, int n, int c, int g,
                   void* stream);

/* Depthwise 3x3 (pad 1) + folded eval-mode BatchNorm + ReLU on the side x side token grid
 * (LocalityAwareFeedforward.depthwise_conv, idf/cross_attn.py:33-41,54).  f16 [n*side*side, c]; w [9,c]. */
int dvd_dwconv3x3(const void* in16, void* out16, const float* w9c, const float* b, int n, int side, int c,
                  void* stream);

/* Adaptive2DPositionalEncoding (idf/cross_attn.py:143-157): token mean per sample (deterministic two-stage),
 * then z += hs[n,c]*htab[ty,c] + ws[n,c]*wtab[tx,c]. */
int dvd_colmean(const float* z, float* partial, float* pooled, int n, int t, int c, int chunks, void* stream);
int dvd_posenc_add(float* z, const float* hs, const float* ws, const float* htab, const float* wtab, int n, int side,
                   int c, void* stream);

/* Tiny-M linear y = act_out(W . act_in(x) + b): TimestepEmbedder (act_in 2 = sinusoid of x[m], idf/cross_model.py:111-139),
 * adaLN_modulation (act_in 1 = SiLU, :175-177,325-327, kmod tiles the input: t.repeat(1,4) :331), pos-enc scale MLPs
 * (act_out 2 ReLU / 3 sigmoid, idf/cross_attn.py:136-141). */
int dvd_small_linear(const float* x, int ldx, const float* w, const float* b, float* y, int ldy, int m, int k, int n,
                     int kmod, int act_in, int act_out, void* stream);

/* decoder.layer_norm + FinalLayer2 + unpatchify + init_flow (idf/cross_attn.py:457; idf/cross_model.py:329-336,
 * 553-566,645-646).  z [N*T,1536]; x0 [N,2,G,G]; tok8 (optional) [N*T,8]. */
int dvd_final_tokens(const float* z, const float* gamma, const float* beta, const float* shift, const float* scale,
                     int ldmod, int mod_rows, const float* w8, const float* b8, const float* init_flow, float* x0,
                     float* tok8, int n, int g, void* stream);

/* Conv pyramid pieces (idf/cross_model.py:18-95), channels-last f32. */
int dvd_im2col3x3(const float* in, long sc, long sy, long sx, float* out, int ldo, int c, int h, int w, void* stream);
/* 3x3 / pad 1 / stride 1 convolution + bias (+ReLU when relu != 0) of a channels-last map in [h, w, c], c % 16 == 0, as an
 * implicit GEMM: the arithmetic of dvd_im2col3x3 + dvd_gemm_nt (f32) on weights wgt [cout, kp] (K order tap-major; kp >= 9 c
 * for cout <= 64, kp == 9 c above) - same K-tiles, same MFMA sequence, same bits - without the [h w, kp] matrix.  At most 64
 * output channels: operands straight to registers (conv_f32_narrow_kernel); more: the exact-f32 128 x 128 GEMM gathering its
 * A tiles from the map (gemm_nt_kernel<f32, CONV>). */
int dvd_conv3x3_nhwc(const float* in, int c, const float* wgt, int kp, const float* bias, float* out, int cout, int h, int w,
                     int relu, void* stream);
int dvd_maxpool2_nhwc(const float* in, float* out, int c, int h, int w, void* stream);
int dvd_resize_bilinear_nhwc(const float* in, float* out, int c, int hin, int win, int hout, int wout, void* stream);
int dvd_nhwc_to_nchw(const float* in, float* out, int c, int h, int w, void* stream);

/* ------------------------------------------------------------------------------------------
 * Pre-stage conditioning nets: the U2NETP document-mask nets and the text-line
 * UNet that produce mask_cat / mask_y512 / line_msk once per document, before the diffusion loop
 * (train_settings/dvd/evaluation.py:162-216; train_settings/models/geotr/geotr_core.py:24-46,48-330,
 * 745-845,984-1019; unet_model.py:4-37; unet_parts.py:8-77).
 * A net is a flat list of ops over numbered activation slots (slot 0 = the input); the host builds the
 * list from the architecture and packs the weights (eval-mode BatchNorm folded into conv weight + bias,
 * conv weight as [cout, kpad] with K order (ky*ks+kx)*cin + c, kpad = ks*ks*cin rounded up to 16,
 * followed by the bias [cout] padded to a multiple of 4 floats; convs packed in op order).  Activations are channels-last f32; convs run
 * on the exact-f32 MFMA GEMM. */
enum { DVD_CN_CONV = 0, DVD_CN_POOL = 1, DVD_CN_RESIZE = 2, DVD_CN_ADD = 3, DVD_CN_SIGMOID = 4 };
typedef struct {
  int op;           /* DVD_CN_* */
  int a, b;         /* input slots; b = -1 when unused.  conv: b is concatenated after a along channels
                       (torch.cat((a, b), 1)); add: second operand; resize: the slot whose size is the target */
  int dst;          /* output slot (each slot is written once) */
  int ks, dil;      /* conv: kernel size 1 or 3, dilation (padding = dil * (ks / 2)) */
  int cout, act;    /* conv: output channels; 0 none, 2 ReLU */
  long w_off;       /* conv: offset, in floats, of this conv's [cout, kpad] weights (+ bias) in the blob */
  int h, w;         /* resize with b = -1: explicit target size */
  int flag;         /* resize: align_corners; pool: ceil_mode */
} dvd_cn_op;

int dvd_convnet_create(const dvd_cn_op* ops, int n_ops, int n_slots, int in_c, int in_h, int in_w, void** handle);
/* The same net for `batch` images per run (the documents of a batch, evaluation.py:162-216 is called per document in the
 * reference): every op is ONE launch over all images (a conv's GEMM has M = batch * h * w rows).  Kernel choices depend on
 * the per-image shape only: an image gives the same bits alone or in a batch. */
int dvd_convnet_create_batched(const dvd_cn_op* ops, int n_ops, int n_slots, int in_c, int in_h, int in_w, int batch,
                               void** handle);
int dvd_convnet_destroy(void* handle);
long dvd_convnet_workspace_bytes(void* handle);
long dvd_convnet_weight_floats(void* handle);
int dvd_convnet_slot_shape(void* handle, int slot, int* h, int* w, int* c);
/* One forward pass: in_nchw [batch, in_c, in_h, in_w] f32 planar (batch = 1 for dvd_convnet_create); the requested slots
 * are written to out_nchw[k] as planar [batch, c, h, w].  workspace: >= dvd_convnet_workspace_bytes, 256-byte aligned. */
int dvd_convnet_run(void* handle, const float* in_nchw, const float* weights, void* workspace, long workspace_bytes,
                    int n_out, const int* out_slots, float* const* out_nchw, void* stream);
/* F.interpolate(mode='bilinear', align_corners=...) on `planes` independent [hin, win] f32 images
 * (evaluation.py:162,199-204,210; geotr_core.py:992,1010). */
int dvd_resize_bilinear_nchw(const float* in, float* out, long planes, int hin, int win, int hout, int wout,
                             int align_corners, void* stream);
/* mskx = (d0 > thr).float() * x  (Seg.forward, geotr_core.py:989-990): d0 [hw], x / out [c, hw];
 * mask_out (optional) receives the 0/1 mask. */
int dvd_threshold_mask_mul(const float* d0, const float* x_nchw, float* out_nchw, float* mask_out, int c, long hw,
                           float thr, void* stream);
/* the same for n images in one launch: d0 [n, hw], x / out [n, c, hw], mask_out [n, hw] */
int dvd_threshold_mask_mul_batch(const float* d0, const float* x_nchw, float* out_nchw, float* mask_out, int n, int c,
                                 long hw, float thr, void* stream);

/* ------------------------------------------------------------------------------------------
 * Image ingest (datasets/doc_dataset/doc_benchmark.py:75-97 after the decode):
 *   y = cv2.resize(rgb, (out, out)) / 255.  as planar f32 [3,out,out]  (OpenCV 8-bit INTER_LINEAR fixed point),
 *   rgb_hwc_out (optional) = the full-resolution image in RGB order (what the unwarp tail samples).
 * src_hwc [h,w,3] uint8 on the device; swap_rb: the source is BGR (cv2.imread order).
 * scratch: dvd_ingest_scratch_bytes(out_size) device bytes. */
long dvd_ingest_scratch_bytes(int out_size);
int dvd_ingest_u8(const uint8_t* src_hwc, int h, int w, int swap_rb, float* y_chw, int out_size,
                  uint8_t* rgb_hwc_out, void* scratch, void* stream);

/* Temporal dithering of the f16 weight rounding (no reference counterpart: the reference is fp32; the GEMMs that consume
 * the result are the per-step nn.Linear / 1x1 convs, idf/cross_attn.py:197-221,52-57, idf/cross_model.py:163-174).
 * hi, lo [nelem] f16 with W = hi + lo (hi = round-to-nearest f16 of W); out [nelem] f16 = W re-rounded to one of its two
 * f16 neighbours so that the MEAN over consecutive `step`s is W: up iff hash(elem0 + i) + step * 2^32/phi (mod 2^32) <
 * 2^32 * (W - down)/(up - down).  nelem % 8 == 0, 16-byte aligned.  The engine calls it before each evaluation when
 * its "dither" option is on. */
int dvd_dither_f16(const void* hi, const void* lo, void* out, long nelem, unsigned elem0, unsigned step, void* stream);

/* ------------------------------------------------------------------------------------------
 * Engine: DiT.forward of the live model (idf/cross_model.py:568-647) for docs x n_hyp samples.
 *   create -> workspace_bytes -> bind_workspace -> set_tensor(every tensor of tensor_info) ->
 *   prepare_docs (once per batch of documents) -> denoise_step (once per diffusion step).
 * The engine owns no device memory.  Tensor names/sizes are enumerated by tensor_info; the host-side
 * packer (dvd_amd/weights.py) derives them from a reference-named state_dict (model1852000.pt keys).
 * ---------------------------------------------------------------------------------------- */
int dvd_engine_create(int grid, int docs, int n_hyp, void** handle);
int dvd_engine_destroy(void* handle);
long dvd_engine_workspace_bytes(void* handle);
int dvd_engine_bind_workspace(void* handle, void* workspace, long bytes);
int dvd_engine_tensor_count(void* handle);
int dvd_engine_tensor_info(void* handle, int index, const char** name, int* dtype /*0 f32, 1 f16*/, long* nelem);
int dvd_engine_set_tensor(void* handle, const char* name, const void* dev_ptr, long nelem);
/* options (per handle; no environment variable is read anywhere in the library):
 *   "split_weights" (default 1): use the (hi, lo) f16 weight pairs -> fp32-grade weights, 2x GEMM MFMAs;
 *   "dither"        (default: 1 when the grid takes the 256-wide GEMM kernels, i.e. grid >= 66, else 0): the weights of
 *                                the 256-wide per-step GEMMs are re-rounded to ONE f16 before every evaluation with a
 *                                step-dependent sub-ulp offset (dvd_dither_f16: zero-mean over the steps) and those
 *                                GEMMs run one pass - half the MFMAs of the split; the other GEMMs keep the (hi, lo) pair
 *                                (no effect on small-tile grids, which keep the pair everywhere);
 *   "dither_step"  (default 0) : the dithering phase of every following denoise_step until it is set again.  The engine keeps
 *                                NO running counter: an evaluation is a pure function of its inputs and the handle's
 *                                options (the reference's model() is pure, idf/cross_model.py:568-647).  A roll-out sets it
 *                                to its loop index (0 at the first step: S-1-i at timestep index i) before every
 *                                evaluation, which is what makes the rounding zero-mean over the steps; left constant the
 *                                evaluations are still correct, each with plain f16 weight rounding;
 *   "ffn_lo"        (default 1): 0 drops the lo pass of the decoder FFN's two 1x1 convs only (-4.8 % step time;
 *                                measured coordinate error on synthetic weights 1.2e-4 -> 3.3e-4: opt-in);
 *   "graphs"        (default 0): replay each denoiser evaluation as a captured hipGraph (bit-identical results;
 *                                the Python engine turns it on for grids <= 128);
 *   "small_tiles"   (default: 1 when (grid/2)^2 <= 1024 tokens, i.e. grid <= 64, else 0): 128x128 GEMM tiles for
 *                                the per-step GEMMs - a function of the grid only, so a document takes the same kernels
 *                                and the same summation order alone or in a batch. */
int dvd_engine_set_option(void* handle, const char* name, int value);
/* y512 [docs,3,512,512] (0..1), mask_cat [docs,1,512,512], mask_y512 [docs,384,G,G], line_msk [docs,64,G,G]
 * (kwargs of the denoiser call, train_settings/dvd/evaluation.py:106-115). */
int dvd_engine_prepare_docs(void* handle, const float* y512, const float* mask_cat, const float* mask_y512,
                            const float* line_msk, void* stream);
/* feat as the reference returns it: [docs,256,G,G] (idf/cross_model.py:647). */
int dvd_engine_feat_nchw(void* handle, float* out, void* stream);
/* x_t, init_flow, x0_out [N,2,G,G].  t_embed = value fed to the timestep embedder after the override rule
 * (idf/cross_model.py:575-580); feat_mode as in dvd_build_r_rows. */
int dvd_engine_denoise_step(void* handle, const float* x_t, float t_embed, int feat_mode, const float* init_flow,
                            const float* init_feat_nchw /* feat_mode 3 only, else NULL */, float* x0_out, void* stream);
/* Per-launch timing of the dominant kernel (the head_dim-256 decoder attention) with HIP events recorded on
 * the launch stream: profile(1) arms it, profile_read returns the number of timed launches and their summed
 * duration since the last read (synchronises on the events).  Used by bench.py's roofline leg. */
int dvd_engine_profile(void* handle, int enable);
int dvd_engine_profile_read(void* handle, int* launches, double* total_ms);
/* Internal activation buffers by name (parity tests only); debug_stop makes denoise_step return after
 * stage k: 1 cross-attention streams, 2 DiT block, 3 decoder pos-enc, 4+j decoder layer j (0 = run all). */
int dvd_engine_debug_buffer(void* handle, const char* name, void** ptr, long* bytes);
int dvd_engine_debug_stop(void* handle, int stage);

/* ------------------------------------------------------------------------------------------
 * Hardware self-test of the MFMA fragment layouts the kernels rely on (exact integer data).
 * a16 [32,16], b16 [16,32], vt16 [32,32] f16; out [3072] f32 = {A.B, Vt.(A.B) via accumulator-as-
 * operand, f32-MFMA A[:, :2].B[:2, :]}.  No reference counterpart (test infrastructure). */
int dvd_selftest_mfma(const void* a16, const void* b16, const void* vt16, float* out3072, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* DVD_HIP_H */
