// The denoiser engine: one handle = one (device, stream) working set for `docs` documents x
// `n_hyp` hypotheses at coordinate-grid size G.  It owns no device memory: weights are bound by
// name (dvd_engine_set_tensor) and all activations live in a caller-provided workspace.
//
// What it computes is DiT.forward of the live configuration (idf/cross_model.py:568-647 with
// train_mode 'stage_1_dit_cross', tv=True), restructured for the hardware:
//   * only blocks[-1] is evaluated: the reference's block loop never feeds x forward (:615-616),
//     so blocks 0..10 are dead compute (verified bit-identical);
//   * everything that does not depend on the denoising step - conv pyramid, c/m/l patch embeddings
//     and their cross-attention K/V projections - is computed once per DOCUMENT in
//     dvd_engine_prepare_docs and shared by the document's hypotheses;
//   * the four conditioning streams run as batched launches and write straight into the 1536-wide
//     token buffer the decoder consumes (no torch.cat / transpose, :623).
#include <string.h>

#include <string>
#include <vector>

#include "common.h"
#include "mfma.h"

extern "C" {
int dvd_embed_obs_ln(const float*, const float*, const float*, const float*, float*, void*, int, int, void*);
int dvd_layernorm_rows(const float*, int, long, void*, int, long, int, long, int, const float*, const float*,
                       const float*, const float*, int, int, float, void*);
int dvd_build_r_rows(const float*, const float*, const float*, void*, int, int, int, int, int, void*);
int dvd_patch_rows(const float*, long, long, long, long, float*, int, int, int, int, void*);
int dvd_dwconv3x3(const void*, void*, const float*, const float*, int, int, int, void*);
int dvd_colmean(const float*, float*, float*, int, int, int, int, void*);
int dvd_posenc_add(float*, const float*, const float*, const float*, const float*, int, int, int, void*);
int dvd_small_linear(const float*, int, const float*, const float*, float*, int, int, int, int, int, int, int, void*);
int dvd_final_tokens(const float*, const float*, const float*, const float*, const float*, int, int, const float*,
                     const float*, const float*, float*, float*, int, int, void*);
int dvd_im2col3x3(const float*, long, long, long, float*, int, int, int, int, void*);
int dvd_maxpool2_nhwc(const float*, float*, int, int, int, void*);
int dvd_resize_bilinear_nhwc(const float*, float*, int, int, int, int, int, void*);
int dvd_nhwc_to_nchw(const float*, float*, int, int, int, void*);
}

namespace dvd {

constexpr int HID = 384, DEC = 1536, FFN = 2048, RK = 1088 /* 4*258 = 1032 padded to 64 */;
constexpr int COLCHUNKS = 64;

struct TensorSpec {
  std::string name;
  int dtype;  // 0 f32, 1 f16
  long nelem;
};

static const int PYR_CIN[7] = {4, 64, 64, 128, 128, 256, 256};
static const int PYR_COUT[7] = {64, 64, 128, 128, 256, 256, 256};
static inline int pyr_kpad(int i) { return ((9 * PYR_CIN[i] + 63) / 64) * 64; }

static std::vector<TensorSpec> tensor_specs(int G) {
  const long T = (long)(G / 2) * (G / 2), side = G / 2;
  std::vector<TensorSpec> v;
  auto f32 = [&](const std::string& n, long e) { v.push_back({n, 0, e}); };
  // every f16 weight comes as a (hi, lo) pair: W = hi + lo, lo unscaled  (see dvd_gemm_desc.B_lo)
  auto f16 = [&](const std::string& n, long e) { v.push_back({n, 1, e}); v.push_back({n + "_lo", 1, e}); };
  f32("obs_w", HID * 8); f32("obs_b", HID); f32("pos", T * HID);
  f16("r_w16", (long)HID * RK); f32("r_b", HID);
  f32("c_w", (long)HID * 1024); f32("c_b", HID);
  f32("m_w", (long)HID * 1536); f32("m_b", HID);
  f32("l_w", (long)HID * 256); f32("l_b", HID);
  f32("t_w0", HID * 256); f32("t_b0", HID); f32("t_w2", HID * HID); f32("t_b2", HID);
  f32("ada_w", 6 * HID * HID); f32("ada_b", 6 * HID);
  f16("ca_wq16", HID * HID); f32("ca_bq", HID);
  f16("ca_wk16", HID * HID); f32("ca_bk", HID); f16("ca_wv16", HID * HID); f32("ca_bv", HID);
  f32("ca_wk32", HID * HID); f32("ca_wv32", HID * HID);
  f16("ca_wo16", HID * HID); f32("ca_bo", HID);
  f16("sa_wqk16", 2 * HID * HID); f32("sa_bqk", 2 * HID); f16("sa_wv16", HID * HID); f32("sa_bv", HID);
  f16("sa_wp16", HID * HID); f32("sa_bp", HID);
  f16("fc1_w16", 4 * HID * HID); f32("fc1_b", 4 * HID); f16("fc2_w16", 4 * HID * HID); f32("fc2_b", HID);
  for (const char* hw : {"pe_h", "pe_w"}) {
    f32(std::string(hw) + "0_w", (long)DEC * DEC); f32(std::string(hw) + "0_b", DEC);
    f32(std::string(hw) + "2_w", (long)DEC * DEC); f32(std::string(hw) + "2_b", DEC);
  }
  f32("pe_htab", side * DEC); f32("pe_wtab", side * DEC);
  for (int j = 0; j < 6; ++j) {
    const std::string p = "d" + std::to_string(j) + "_";
    f32(p + "n1w", DEC); f32(p + "n1b", DEC);
    f16(p + "wqk16", 2L * DEC * DEC); f16(p + "wv16", (long)DEC * DEC); f16(p + "wfc16", (long)DEC * DEC);
    f32(p + "n2w", DEC); f32(p + "n2b", DEC);
    f16(p + "c1w16", (long)FFN * DEC); f32(p + "c1b", FFN);
    f32(p + "dww", 9 * FFN); f32(p + "dwb", FFN);
    f16(p + "c2w16", (long)DEC * FFN); f32(p + "c2b", DEC);
  }
  f32("dec_nw", DEC); f32("dec_nb", DEC);
  f32("fin_ada_w", 2L * DEC * DEC); f32("fin_ada_b", 2 * DEC); f32("fin_w", 8 * DEC); f32("fin_b", 8);
  for (int i = 0; i < 7; ++i) {
    f32("pyr" + std::to_string(i) + "_w", (long)PYR_COUT[i] * pyr_kpad(i));
    f32("pyr" + std::to_string(i) + "_b", PYR_COUT[i]);
  }
  return v;
}

struct Buf {
  std::string name;
  size_t off, bytes;
};

struct Engine {
  int G, side, docs, hyp, N;
  long T, NT;
  std::vector<TensorSpec> specs;
  std::vector<const void*> wptr;
  char* ws = nullptr;
  size_t ws_bytes = 0, need_bytes = 0;
  std::vector<Buf> bufs;
  bool prepared = false;
  bool split_weights = true;   // use the lo parts (fp32-grade weights, 2x GEMM MFMAs)
  // optional per-launch timing of the dominant kernel (decoder attention) with HIP events on the launch stream
  bool prof_on = false;
  std::vector<hipEvent_t> prof_ev;
  size_t prof_used = 0;
  int debug_stop = 0;  // parity tests: return from denoise_step after stage k (0 = run everything)

  const void* W(const char* name) const {
    for (size_t i = 0; i < specs.size(); ++i)
      if (specs[i].name == name) return wptr[i];
    return nullptr;
  }
  const float* Wf(const std::string& n) const { return (const float*)W(n.c_str()); }
  const void* Wh(const std::string& n) const { return W(n.c_str()); }
  const void* Wl(const std::string& n) const { return split_weights ? W((n + "_lo").c_str()) : nullptr; }
  char* B(const char* name) const {
    for (auto& b : bufs)
      if (b.name == name) return ws + b.off;
    return nullptr;
  }
  long Bn(const char* name) const {
    for (auto& b : bufs)
      if (b.name == name) return (long)b.bytes;
    return 0;
  }
};

static void plan(Engine* e) {
  size_t off = 0;
  auto add = [&](const char* n, size_t bytes) {
    bytes = (bytes + 255) & ~(size_t)255;
    e->bufs.push_back({n, off, bytes});
    off += bytes;
  };
  const size_t T = e->T, NT = e->NT, docs = e->docs, N = e->N, G = e->G;
  // persistent per-document conditioning
  add("feat", docs * G * G * 256 * 4);
  for (const char* s : {"kc16", "km16", "kl16", "vtc16", "vtm16", "vtl16"}) add(s, docs * T * HID * 2);
  // small per-step vectors
  add("tbuf", 256); add("th", HID * 4); add("cvec", HID * 4); add("mod", 6 * HID * 4); add("finmod", 2 * DEC * 4);
  add("pooled", N * DEC * 4); add("petmp", N * DEC * 4); add("hs", N * DEC * 4); add("wsc", N * DEC * 4);
  add("part", N * COLCHUNKS * DEC * 4);
  const size_t scratch0 = off;
  // per-step activations
  add("xtok32", NT * HID * 4); add("xq16", NT * HID * 2); add("arows16", NT * RK * 2); add("rtok16", NT * HID * 2);
  add("q16", NT * HID * 2); add("kr16", NT * HID * 2); add("vtr16", NT * HID * 2);
  add("att16", NT * DEC * 2); add("z", NT * DEC * 4); add("h16", NT * DEC * 2); add("qk16", NT * 2 * DEC * 2);
  add("vt16", NT * DEC * 2); add("mlp16", NT * 4 * DEC * 2);
  const size_t step_end = off;
  // prepare-time scratch aliases the per-step region (prepare never overlaps a step)
  off = scratch0;
  add("p_cat4", (size_t)4 * 512 * 512 * 4);
  add("p_col", (size_t)512 * 512 * 576 * 4);
  add("p_actA", (size_t)512 * 512 * 64 * 4);
  add("p_actB", (size_t)512 * 512 * 64 * 4);
  add("p_rows", T * 1536 * 4);
  add("p_tok32", T * HID * 4);
  e->need_bytes = off > step_end ? off : step_end;
}

static int gemm(int dtype, int M, int N, int K, int batch, const void* A, int lda, long sA, const void* Bm, int ldb,
                long sB, float* C32, int ldc, long sC32, void* C16, int ldc16, long sC16, const float* bias,
                int bias_row, int act, const float* pos, int pos_rows, const float* gate, int gate_rows,
                const float* res, int ldres, long sRes, void* stream, const void* Alo = nullptr,
                const void* Blo = nullptr) {
  dvd_gemm_desc d;
  memset(&d, 0, sizeof(d));
  d.A_lo = Alo; d.B_lo = Blo; d.lo_scale = 1.f;
  d.dtype = dtype; d.M = M; d.N = N; d.K = K; d.batch = batch;
  d.A = A; d.lda = lda; d.strideA = sA;
  d.B = Bm; d.ldb = ldb; d.strideB = sB;
  d.C32 = C32; d.ldc = ldc; d.strideC32 = sC32;
  d.C16 = C16; d.ldc16 = ldc16; d.strideC16 = sC16;
  d.bias = bias; d.bias_row = bias_row; d.act = act;
  d.pos = pos; d.ldpos = N; d.pos_rows = pos_rows;
  d.gate = gate; d.ldgate = N; d.gate_rows = gate_rows;
  d.res = res; d.ldres = ldres; d.strideRes = sRes;
  return dvd_gemm_nt(&d, stream);
}

static int attn(int hd, int heads, int batch, int tq, int tk, int kv_div, const void* Q, int ldq, long sQ,
                const void* K, int ldk, long sK, const void* Vt, int ldvt, long sVt, void* O, int ldo, long sO,
                float scale, void* stream) {
  dvd_attn_desc d;
  memset(&d, 0, sizeof(d));
  d.head_dim = hd; d.heads = heads; d.batch = batch; d.tq = tq; d.tk = tk; d.kv_batch_div = kv_div;
  d.Q = Q; d.ldq = ldq; d.strideQ = sQ; d.K = K; d.ldk = ldk; d.strideK = sK;
  d.Vt = Vt; d.ldvt = ldvt; d.strideVt = sVt; d.O = O; d.ldo = ldo; d.strideO = sO; d.scale = scale;
  return dvd_flash_attn(&d, stream);
}

__global__ void set_scalar_kernel(float* p, float v) { *p = v; }

#define TRY(x)            \
  do {                    \
    int _e = (x);         \
    if (_e) return _e;    \
  } while (0)

}  // namespace dvd

using namespace dvd;

extern "C" int dvd_engine_create(int grid, int docs, int n_hyp, void** handle) {
  DVD_REQUIRE(handle, "engine_create: null handle");
  DVD_REQUIRE(grid >= 4 && grid % 2 == 0 && ((grid / 2) * (grid / 2)) % 8 == 0,
              "engine_create: grid=%d must be even with (grid/2)^2 a multiple of 8", grid);
  DVD_REQUIRE(docs > 0 && n_hyp > 0, "engine_create: docs/n_hyp must be positive");
  Engine* e = new Engine();
  e->G = grid; e->side = grid / 2; e->docs = docs; e->hyp = n_hyp; e->N = docs * n_hyp;
  e->T = (long)e->side * e->side; e->NT = e->T * e->N;
  e->specs = tensor_specs(grid);
  e->wptr.assign(e->specs.size(), nullptr);
  plan(e);
  *handle = e;
  return DVD_OK;
}

extern "C" int dvd_engine_destroy(void* handle) {
  delete (Engine*)handle;
  return DVD_OK;
}

extern "C" long dvd_engine_workspace_bytes(void* handle) { return handle ? (long)((Engine*)handle)->need_bytes : -1; }

extern "C" int dvd_engine_bind_workspace(void* handle, void* ws, long bytes) {
  DVD_REQUIRE(handle && ws, "engine_bind_workspace: null pointer");
  Engine* e = (Engine*)handle;
  DVD_REQUIRE((size_t)bytes >= e->need_bytes && ((uintptr_t)ws % 256) == 0,
              "engine_bind_workspace: need %zu bytes, 256-byte aligned (got %ld)", e->need_bytes, bytes);
  e->ws = (char*)ws; e->ws_bytes = bytes; e->prepared = false;
  return DVD_OK;
}

extern "C" int dvd_engine_tensor_count(void* handle) { return handle ? (int)((Engine*)handle)->specs.size() : -1; }

extern "C" int dvd_engine_tensor_info(void* handle, int index, const char** name, int* dtype, long* nelem) {
  DVD_REQUIRE(handle && name && dtype && nelem, "engine_tensor_info: null pointer");
  Engine* e = (Engine*)handle;
  DVD_REQUIRE(index >= 0 && index < (int)e->specs.size(), "engine_tensor_info: index out of range");
  *name = e->specs[index].name.c_str(); *dtype = e->specs[index].dtype; *nelem = e->specs[index].nelem;
  return DVD_OK;
}

extern "C" int dvd_engine_set_tensor(void* handle, const char* name, const void* dev_ptr, long nelem) {
  DVD_REQUIRE(handle && name && dev_ptr, "engine_set_tensor: null pointer");
  Engine* e = (Engine*)handle;
  for (size_t i = 0; i < e->specs.size(); ++i)
    if (e->specs[i].name == name) {
      DVD_REQUIRE(e->specs[i].nelem == nelem, "engine_set_tensor: %s expects %ld elements, got %ld", name,
                  e->specs[i].nelem, nelem);
      DVD_REQUIRE(((uintptr_t)dev_ptr % 16) == 0, "engine_set_tensor: %s must be 16-byte aligned", name);
      e->wptr[i] = dev_ptr;
      return DVD_OK;
    }
  set_error("engine_set_tensor: unknown tensor '%s'", name);
  return DVD_E_ARG;
}

extern "C" int dvd_engine_profile(void* handle, int enable) {
  DVD_REQUIRE(handle, "engine_profile: null handle");
  Engine* e = (Engine*)handle;
  if (enable && e->prof_ev.empty()) {
    e->prof_ev.resize(2 * 4096);
    for (auto& ev : e->prof_ev)
      if (hipEventCreate(&ev) != hipSuccess) { set_error("engine_profile: hipEventCreate failed"); return DVD_E_LAUNCH; }
  }
  e->prof_on = enable != 0;
  e->prof_used = 0;
  return DVD_OK;
}

extern "C" int dvd_engine_profile_read(void* handle, int* launches, double* total_ms) {
  DVD_REQUIRE(handle && launches && total_ms, "engine_profile_read: null pointer");
  Engine* e = (Engine*)handle;
  *launches = 0; *total_ms = 0.0;
  for (size_t i = 0; i + 1 < e->prof_used; i += 2) {
    float ms = 0.f;
    if (hipEventSynchronize(e->prof_ev[i + 1]) != hipSuccess ||
        hipEventElapsedTime(&ms, e->prof_ev[i], e->prof_ev[i + 1]) != hipSuccess) {
      set_error("engine_profile_read: event query failed");
      return DVD_E_LAUNCH;
    }
    *total_ms += ms; *launches += 1;
  }
  e->prof_used = 0;
  return DVD_OK;
}

extern "C" int dvd_engine_set_option(void* handle, const char* name, int value) {
  DVD_REQUIRE(handle && name, "engine_set_option: null pointer");
  Engine* e = (Engine*)handle;
  if (strcmp(name, "split_weights") == 0) { e->split_weights = value != 0; return DVD_OK; }
  set_error("engine_set_option: unknown option '%s'", name);
  return DVD_E_ARG;
}

extern "C" int dvd_engine_debug_stop(void* handle, int stage) {
  DVD_REQUIRE(handle && stage >= 0, "engine_debug_stop: bad arguments");
  ((Engine*)handle)->debug_stop = stage;
  return DVD_OK;
}

extern "C" int dvd_engine_debug_buffer(void* handle, const char* name, void** ptr, long* bytes) {
  DVD_REQUIRE(handle && name && ptr && bytes, "engine_debug_buffer: null pointer");
  Engine* e = (Engine*)handle;
  DVD_REQUIRE(e->ws, "engine_debug_buffer: no workspace bound");
  *ptr = e->B(name); *bytes = e->Bn(name);
  DVD_REQUIRE(*ptr, "engine_debug_buffer: unknown buffer '%s'", name);
  return DVD_OK;
}

static int require_ready(Engine* e, bool need_prepared) {
  if (!e->ws) { set_error("engine: no workspace bound"); return DVD_E_STATE; }
  for (size_t i = 0; i < e->specs.size(); ++i)
    if (!e->wptr[i]) { set_error("engine: weight tensor '%s' not set", e->specs[i].name.c_str()); return DVD_E_STATE; }
  if (need_prepared && !e->prepared) { set_error("engine: dvd_engine_prepare_docs has not run"); return DVD_E_STATE; }
  return DVD_OK;
}

// ------------------------------------------------------------------------------------------------
// Once per batch of documents (step-invariant conditioning, idf/cross_model.py:584-594,604-605 and
// the K/V halves of the shared cross_attn in_proj, :237-257)
// ------------------------------------------------------------------------------------------------
extern "C" int dvd_engine_prepare_docs(void* handle, const float* y512, const float* mask_cat,
                                       const float* mask_y512, const float* line_msk, void* stream) {
  DVD_REQUIRE(handle && y512 && mask_cat && mask_y512 && line_msk, "engine_prepare_docs: null pointer");
  Engine* e = (Engine*)handle;
  TRY(require_ready(e, false));
  hipStream_t st = (hipStream_t)stream;
  const int G = e->G;
  const long T = e->T;
  float* cat4 = (float*)e->B("p_cat4");
  float* col = (float*)e->B("p_col");
  float* actA = (float*)e->B("p_actA");
  float* actB = (float*)e->B("p_actB");
  float* rows = (float*)e->B("p_rows");
  float* tok32 = (float*)e->B("p_tok32");
  const float* pos = e->Wf("pos");
  for (int d = 0; d < e->docs; ++d) {
    // cat([y512, mask_cat], dim=1)  (:586-587)
    if (hipMemcpyAsync(cat4, y512 + (size_t)d * 3 * 512 * 512, (size_t)3 * 512 * 512 * 4, hipMemcpyDeviceToDevice, st) != hipSuccess ||
        hipMemcpyAsync(cat4 + 3 * 512 * 512, mask_cat + (size_t)d * 512 * 512, (size_t)512 * 512 * 4, hipMemcpyDeviceToDevice, st) != hipSuccess) {
      set_error("engine_prepare_docs: device copy failed");
      return DVD_E_LAUNCH;
    }
    // conv pyramid (:18-95): conv3x3+ReLU as im2col + f32 GEMM, channels-last activations
    struct L { int idx, hw; bool pool; };
    const L layers[7] = {{0, 512, false}, {1, 512, true}, {2, 256, false}, {3, 256, true},
                         {4, 128, false}, {5, 128, false}, {6, 128, true}};
    const float* cur = cat4;
    long sc = 512L * 512, sy = 512, sx = 1;   // first layer reads the planar input
    float* act[2] = {actA, actB};
    int wi = 0;
    for (const L& l : layers) {
      const int cin = PYR_CIN[l.idx], cout = PYR_COUT[l.idx], kp = pyr_kpad(l.idx);
      TRY(dvd_im2col3x3(cur, sc, sy, sx, col, kp, cin, l.hw, l.hw, stream));
      const std::string wn = "pyr" + std::to_string(l.idx);
      float* outp = act[wi];
      wi ^= 1;
      TRY(gemm(1, l.hw * l.hw, cout, kp, 1, col, kp, 0, e->Wf(wn + "_w"), kp, 0, outp, cout, 0, nullptr, 0, 0,
               e->Wf(wn + "_b"), 0, /*relu*/ 2, nullptr, 0, nullptr, 0, nullptr, 0, 0, stream));
      int hw = l.hw;
      if (l.pool) {
        TRY(dvd_maxpool2_nhwc(outp, act[wi], cout, hw, hw, stream));
        outp = act[wi];
        wi ^= 1;
        hw /= 2;
      }
      cur = outp;
      sc = 1; sy = (long)hw * cout; sx = cout;
    }
    // cur = level_3 output [64,64,256]; resize to the coordinate grid (:590-593 generalised)
    float* feat = (float*)e->B("feat") + (size_t)d * G * G * 256;
    if (G == 64) {
      if (hipMemcpyAsync(feat, cur, (size_t)64 * 64 * 256 * 4, hipMemcpyDeviceToDevice, st) != hipSuccess) {
        set_error("engine_prepare_docs: device copy failed");
        return DVD_E_LAUNCH;
      }
    } else {
      TRY(dvd_resize_bilinear_nhwc(cur, feat, 256, 64, 64, G, G, stream));
    }
    // c / m / l patch embeddings (+pos) and their K / V^T projections (shared cross_attn weights)
    struct S { const float* src; long sn, sc, sy, sx; int c; const char* w; const char* b; const char* k16; const char* vt16; };
    const S streams[3] = {
        {feat, 0, 1, (long)G * 256, 256, 256, "c_w", "c_b", "kc16", "vtc16"},
        {mask_y512 + (size_t)d * 384 * G * G, 0, (long)G * G, G, 1, 384, "m_w", "m_b", "km16", "vtm16"},
        {line_msk + (size_t)d * 64 * G * G, 0, (long)G * G, G, 1, 64, "l_w", "l_b", "kl16", "vtl16"}};
    for (const S& s : streams) {
      const int K4 = 4 * s.c;
      TRY(dvd_patch_rows(s.src, s.sn, s.sc, s.sy, s.sx, rows, K4, 1, s.c, G, stream));
      TRY(gemm(1, (int)T, HID, K4, 1, rows, K4, 0, e->Wf(s.w), K4, 0, tok32, HID, 0, nullptr, 0, 0, e->Wf(s.b), 0, 0,
               pos, (int)T, nullptr, 0, nullptr, 0, 0, stream));
      _Float16* k16 = (_Float16*)e->B(s.k16) + (size_t)d * T * HID;
      _Float16* vt16 = (_Float16*)e->B(s.vt16) + (size_t)d * T * HID;
      TRY(gemm(1, (int)T, HID, HID, 1, tok32, HID, 0, e->Wf("ca_wk32"), HID, 0, nullptr, 0, 0, k16, HID, 0,
               e->Wf("ca_bk"), 0, 0, nullptr, 0, nullptr, 0, nullptr, 0, 0, stream));
      TRY(gemm(1, HID, (int)T, HID, 1, e->Wf("ca_wv32"), HID, 0, tok32, HID, 0, nullptr, 0, 0, vt16, (int)T, 0,
               e->Wf("ca_bv"), 1, 0, nullptr, 0, nullptr, 0, nullptr, 0, 0, stream));
    }
  }
  e->prepared = true;
  return DVD_OK;
}

extern "C" int dvd_engine_feat_nchw(void* handle, float* out, void* stream) {
  DVD_REQUIRE(handle && out, "engine_feat_nchw: null pointer");
  Engine* e = (Engine*)handle;
  TRY(require_ready(e, true));
  for (int d = 0; d < e->docs; ++d)
    TRY(dvd_nhwc_to_nchw((const float*)e->B("feat") + (size_t)d * e->G * e->G * 256,
                         out + (size_t)d * 256 * e->G * e->G, 256, e->G, e->G, stream));
  return DVD_OK;
}

// ------------------------------------------------------------------------------------------------
// One denoiser evaluation for all docs*n_hyp samples (DiT.forward, idf/cross_model.py:568-647).
//   x_t, init_flow, x0_out : [N,2,G,G] f32.   t_embed: the (batch-global) value fed to the timestep
//   embedder after the reference's override rule (:575-580).   feat_mode: 1 -> init_feat = feat
//   (t_model > 600, :597-598), 2 -> init_feat = grid_sample(feat, (init_flow + base)*2-1)
//   (idf/gaussian_diffusion.py:618-624), 0 -> init_feat = 0, 3 -> init_feat_nchw [N,256,G,G] given explicitly.
// ------------------------------------------------------------------------------------------------
extern "C" int dvd_engine_denoise_step(void* handle, const float* x_t, float t_embed, int feat_mode,
                                       const float* init_flow, const float* init_feat_nchw, float* x0_out,
                                       void* stream) {
  DVD_REQUIRE(handle && x_t && init_flow && x0_out, "engine_denoise_step: null pointer");
  DVD_REQUIRE(feat_mode >= 0 && feat_mode <= 3 && (feat_mode != 3 || init_feat_nchw),
              "engine_denoise_step: feat_mode %d (3 needs init_feat)", feat_mode);
  Engine* e = (Engine*)handle;
  TRY(require_ready(e, true));
  hipStream_t st = (hipStream_t)stream;
  const int G = e->G, N = e->N, side = e->side, hyp = e->hyp;
  const int T = (int)e->T;
  const long NT = e->NT;
  DVD_REQUIRE(4 * NT < (1L << 31), "engine_denoise_step: batch too large for 32-bit row indices (NT=%ld)", NT);

  float* tbuf = (float*)e->B("tbuf");
  float* th = (float*)e->B("th");
  float* cvec = (float*)e->B("cvec");
  float* mod = (float*)e->B("mod");
  float* finmod = (float*)e->B("finmod");
  float* xtok32 = (float*)e->B("xtok32");
  _Float16* xq16 = (_Float16*)e->B("xq16");
  _Float16* arows16 = (_Float16*)e->B("arows16");
  _Float16* rtok16 = (_Float16*)e->B("rtok16");
  _Float16* q16 = (_Float16*)e->B("q16");
  _Float16* kr16 = (_Float16*)e->B("kr16");
  _Float16* vtr16 = (_Float16*)e->B("vtr16");
  _Float16* att16 = (_Float16*)e->B("att16");
  float* z = (float*)e->B("z");
  _Float16* h16 = (_Float16*)e->B("h16");
  _Float16* qk16 = (_Float16*)e->B("qk16");
  _Float16* vt16 = (_Float16*)e->B("vt16");
  _Float16* mlp16 = (_Float16*)e->B("mlp16");

  // --- timestep embedding and the two adaLN tables (one row: t is batch-global in sampling) ---
  set_scalar_kernel<<<1, 1, 0, st>>>(tbuf, t_embed);
  TRY(dvd_small_linear(tbuf, 1, e->Wf("t_w0"), e->Wf("t_b0"), th, HID, 1, 256, HID, 256, 2, 1, stream));
  TRY(dvd_small_linear(th, HID, e->Wf("t_w2"), e->Wf("t_b2"), cvec, HID, 1, HID, HID, HID, 0, 0, stream));
  TRY(dvd_small_linear(cvec, HID, e->Wf("ada_w"), e->Wf("ada_b"), mod, 6 * HID, 1, HID, 6 * HID, HID, 1, 0, stream));
  TRY(dvd_small_linear(cvec, HID, e->Wf("fin_ada_w"), e->Wf("fin_ada_b"), finmod, 2 * DEC, 1, DEC, 2 * DEC, HID, 1, 0,
                       stream));
  const float *sh_a = mod, *sc_a = mod + HID, *g_a = mod + 2 * HID, *sh_m = mod + 3 * HID, *sc_m = mod + 4 * HID,
              *g_m = mod + 5 * HID;

  // --- tokens ---
  TRY(dvd_embed_obs_ln(x_t, e->Wf("obs_w"), e->Wf("obs_b"), e->Wf("pos"), xtok32, xq16, N, G, stream));
  TRY(dvd_build_r_rows((const float*)e->B("feat"), init_feat_nchw, init_flow, arows16, RK, N, G, hyp, feat_mode,
                       stream));
  TRY(gemm(0, (int)NT, HID, RK, 1, arows16, RK, 0, e->Wh("r_w16"), RK, 0, nullptr, 0, 0, rtok16, HID, 0, e->Wf("r_b"), 0,
           0, e->Wf("pos"), T, nullptr, 0, nullptr, 0, 0, stream, nullptr, e->Wl("r_w16")));

  // --- parallel cross-attention of the shared query against the 4 streams (:237-265) ---
  TRY(gemm(0, (int)NT, HID, HID, 1, xq16, HID, 0, e->Wh("ca_wq16"), HID, 0, nullptr, 0, 0, q16, HID, 0, e->Wf("ca_bq"),
           0, 0, nullptr, 0, nullptr, 0, nullptr, 0, 0, stream, nullptr, e->Wl("ca_wq16")));
  TRY(gemm(0, (int)NT, HID, HID, 1, rtok16, HID, 0, e->Wh("ca_wk16"), HID, 0, nullptr, 0, 0, kr16, HID, 0,
           e->Wf("ca_bk"), 0, 0, nullptr, 0, nullptr, 0, nullptr, 0, 0, stream, nullptr, e->Wl("ca_wk16")));
  TRY(gemm(0, HID, T, HID, N, e->Wh("ca_wv16"), HID, 0, rtok16, HID, (long)T * HID, nullptr, 0, 0, vtr16, T,
           (long)HID * T, e->Wf("ca_bv"), 1, 0, nullptr, 0, nullptr, 0, nullptr, 0, 0, stream, e->Wl("ca_wv16"), nullptr));
  {
    const char* kn[3] = {"kc16", "km16", "kl16"};
    const char* vn[3] = {"vtc16", "vtm16", "vtl16"};
    for (int s = 0; s < 3; ++s)
      TRY(attn(64, 6, N, T, T, hyp, q16, HID, (long)T * HID, e->B(kn[s]), HID, (long)T * HID, e->B(vn[s]), T,
               (long)HID * T, att16 + (size_t)s * NT * HID, HID, (long)T * HID, 0.125f, stream));
    TRY(attn(64, 6, N, T, T, 1, q16, HID, (long)T * HID, kr16, HID, (long)T * HID, vtr16, T, (long)HID * T,
             att16 + (size_t)3 * NT * HID, HID, (long)T * HID, 0.125f, stream));
  }
  // x_s = x + out_proj(attn_s)  -> z[:, 384 s : 384 (s+1)]   (stream order cond, msk6, line, r == cat order :623)
  TRY(gemm(0, (int)NT, HID, HID, 4, att16, HID, NT * HID, e->Wh("ca_wo16"), HID, 0, z, DEC, HID, nullptr, 0, 0,
           e->Wf("ca_bo"), 0, 0, nullptr, 0, nullptr, 0, xtok32, HID, 0, stream, nullptr, e->Wl("ca_wo16")));

  if (e->debug_stop == 1) return check_launch("engine_denoise_step(stop 1)");
  // --- per stream: gated self-attention (:268-289) ---
  TRY(dvd_layernorm_rows(z, DEC, HID, h16, HID, NT * HID, 4, NT, HID, nullptr, nullptr, sh_a, sc_a, 0, (int)NT, 1e-6f,
                         stream));
  TRY(gemm(0, (int)(4 * NT), 2 * HID, HID, 1, h16, HID, 0, e->Wh("sa_wqk16"), HID, 0, nullptr, 0, 0, qk16, 2 * HID, 0,
           e->Wf("sa_bqk"), 0, 0, nullptr, 0, nullptr, 0, nullptr, 0, 0, stream, nullptr, e->Wl("sa_wqk16")));
  TRY(gemm(0, HID, T, HID, 4 * N, e->Wh("sa_wv16"), HID, 0, h16, HID, (long)T * HID, nullptr, 0, 0, vt16, T,
           (long)HID * T, e->Wf("sa_bv"), 1, 0, nullptr, 0, nullptr, 0, nullptr, 0, 0, stream, e->Wl("sa_wv16"), nullptr));
  TRY(attn(64, 6, 4 * N, T, T, 1, qk16, 2 * HID, (long)T * 2 * HID, qk16 + HID, 2 * HID, (long)T * 2 * HID, vt16, T,
           (long)HID * T, att16, HID, (long)T * HID, 0.125f, stream));
  TRY(gemm(0, (int)NT, HID, HID, 4, att16, HID, NT * HID, e->Wh("sa_wp16"), HID, 0, z, DEC, HID, nullptr, 0, 0,
           e->Wf("sa_bp"), 0, 0, nullptr, 0, g_a, (int)NT, z, DEC, HID, stream, nullptr, e->Wl("sa_wp16")));
  // --- per stream: gated MLP (:271-292) ---
  TRY(dvd_layernorm_rows(z, DEC, HID, h16, HID, NT * HID, 4, NT, HID, nullptr, nullptr, sh_m, sc_m, 0, (int)NT, 1e-6f,
                         stream));
  TRY(gemm(0, (int)(4 * NT), 4 * HID, HID, 1, h16, HID, 0, e->Wh("fc1_w16"), HID, 0, nullptr, 0, 0, mlp16, 4 * HID, 0,
           e->Wf("fc1_b"), 0, /*gelu*/ 1, nullptr, 0, nullptr, 0, nullptr, 0, 0, stream, nullptr, e->Wl("fc1_w16")));
  TRY(gemm(0, (int)NT, HID, 4 * HID, 4, mlp16, 4 * HID, NT * 4 * HID, e->Wh("fc2_w16"), 4 * HID, 0, z, DEC, HID,
           nullptr, 0, 0, e->Wf("fc2_b"), 0, 0, nullptr, 0, g_m, (int)NT, z, DEC, HID, stream, nullptr, e->Wl("fc2_w16")));

  if (e->debug_stop == 2) return check_launch("engine_denoise_step(stop 2)");
  // --- decoder: adaptive 2-D positional encoding (idf/cross_attn.py:143-157) ---
  float* pooled = (float*)e->B("pooled");
  float* petmp = (float*)e->B("petmp");
  float* hs = (float*)e->B("hs");
  float* wsc = (float*)e->B("wsc");
  TRY(dvd_colmean(z, (float*)e->B("part"), pooled, N, T, DEC, COLCHUNKS, stream));
  TRY(dvd_small_linear(pooled, DEC, e->Wf("pe_h0_w"), e->Wf("pe_h0_b"), petmp, DEC, N, DEC, DEC, DEC, 0, 2, stream));
  TRY(dvd_small_linear(petmp, DEC, e->Wf("pe_h2_w"), e->Wf("pe_h2_b"), hs, DEC, N, DEC, DEC, DEC, 0, 3, stream));
  TRY(dvd_small_linear(pooled, DEC, e->Wf("pe_w0_w"), e->Wf("pe_w0_b"), petmp, DEC, N, DEC, DEC, DEC, 0, 2, stream));
  TRY(dvd_small_linear(petmp, DEC, e->Wf("pe_w2_w"), e->Wf("pe_w2_b"), wsc, DEC, N, DEC, DEC, DEC, 0, 3, stream));
  TRY(dvd_posenc_add(z, hs, wsc, e->Wf("pe_htab"), e->Wf("pe_wtab"), N, side, DEC, stream));

  if (e->debug_stop == 3) return check_launch("engine_denoise_step(stop 3)");
  // --- decoder layers (idf/cross_attn.py:377-396) ---
  _Float16* f1 = mlp16;
  _Float16* f2 = mlp16 + (size_t)NT * FFN;
  for (int j = 0; j < 6; ++j) {
    const std::string p = "d" + std::to_string(j) + "_";
    TRY(dvd_layernorm_rows(z, DEC, 0, h16, DEC, 0, 1, NT, DEC, e->Wf(p + "n1w"), e->Wf(p + "n1b"), nullptr, nullptr, 0,
                           1, 1e-5f, stream));
    TRY(gemm(0, (int)NT, 2 * DEC, DEC, 1, h16, DEC, 0, e->Wh(p + "wqk16"), DEC, 0, nullptr, 0, 0, qk16, 2 * DEC, 0,
             nullptr, 0, 0, nullptr, 0, nullptr, 0, nullptr, 0, 0, stream, nullptr, e->Wl(p + "wqk16")));
    TRY(gemm(0, DEC, T, DEC, N, e->Wh(p + "wv16"), DEC, 0, h16, DEC, (long)T * DEC, nullptr, 0, 0, vt16, T,
             (long)DEC * T, nullptr, 0, 0, nullptr, 0, nullptr, 0, nullptr, 0, 0, stream, e->Wl(p + "wv16"), nullptr));
    const bool timed = e->prof_on && e->prof_used + 2 <= e->prof_ev.size();
    if (timed) (void)hipEventRecord(e->prof_ev[e->prof_used], st);
    TRY(attn(256, 6, N, T, T, 1, qk16, 2 * DEC, (long)T * 2 * DEC, qk16 + DEC, 2 * DEC, (long)T * 2 * DEC, vt16, T,
             (long)DEC * T, att16, DEC, (long)T * DEC, 0.0625f, stream));
    if (timed) { (void)hipEventRecord(e->prof_ev[e->prof_used + 1], st); e->prof_used += 2; }
    TRY(gemm(0, (int)NT, DEC, DEC, 1, att16, DEC, 0, e->Wh(p + "wfc16"), DEC, 0, z, DEC, 0, nullptr, 0, 0, nullptr, 0,
             0, nullptr, 0, nullptr, 0, z, DEC, 0, stream, nullptr, e->Wl(p + "wfc16")));
    TRY(dvd_layernorm_rows(z, DEC, 0, h16, DEC, 0, 1, NT, DEC, e->Wf(p + "n2w"), e->Wf(p + "n2b"), nullptr, nullptr, 0,
                           1, 1e-5f, stream));
    TRY(gemm(0, (int)NT, FFN, DEC, 1, h16, DEC, 0, e->Wh(p + "c1w16"), DEC, 0, nullptr, 0, 0, f1, FFN, 0,
             e->Wf(p + "c1b"), 0, 2, nullptr, 0, nullptr, 0, nullptr, 0, 0, stream, nullptr, e->Wl(p + "c1w16")));
    TRY(dvd_dwconv3x3(f1, f2, e->Wf(p + "dww"), e->Wf(p + "dwb"), N, side, FFN, stream));
    TRY(gemm(0, (int)NT, DEC, FFN, 1, f2, FFN, 0, e->Wh(p + "c2w16"), FFN, 0, z, DEC, 0, nullptr, 0, 0,
             e->Wf(p + "c2b"), 0, 2, nullptr, 0, nullptr, 0, z, DEC, 0, stream, nullptr, e->Wl(p + "c2w16")));
    if (e->debug_stop == 4 + j) return check_launch("engine_denoise_step(stop 4+j)");
  }

  // --- final LayerNorm + FinalLayer2 + unpatchify + init_flow (:457; idf/cross_model.py:329-336,553-566,645-646) ---
  TRY(dvd_final_tokens(z, e->Wf("dec_nw"), e->Wf("dec_nb"), finmod, finmod + DEC, 0, (int)NT, e->Wf("fin_w"),
                       e->Wf("fin_b"), init_flow, x0_out, nullptr, N, G, stream));
  return check_launch("engine_denoise_step");
}
