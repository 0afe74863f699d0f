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
#include <algorithm>

#include <string>
#include <vector>

#include "common.h"
#include "gemm_common.h"
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
int dvd_conv3x3_nhwc(const float*, int, const float*, int, const float*, float*, int, int, int, int, void*);
int dvd_maxpool2_nhwc(const float*, float*, int, int, int, void*);
int dvd_resize_bilinear_nhwc(const float*, float*, int, int, int, int, int, void*);
int dvd_nhwc_to_nchw(const float*, float*, int, int, int, void*);
int dvd_dither_f16(const void*, const void*, void*, long, unsigned, unsigned, void*);
}

namespace dvd {

constexpr int HID = 384, DEC = 1536, FFN = 2048, RK = 1088 /* 4*258 = 1032 padded to 64 */;
constexpr int COLCHUNKS = 64;

struct TensorSpec {
  std::string name;
  int dtype;  // 0 f32, 1 f16
  long nelem;
  bool wide;  // f16 weight of a GEMM that takes the 256-wide kernels on large grids (gets a dithered copy, see Engine)
};

static const int PYR_CIN[7] = {4, 64, 64, 128, 128, 256, 256};
static const int PYR_COUT[7] = {64, 64, 128, 128, 256, 256, 256};
static inline int pyr_kpad(int i) { return ((9 * PYR_CIN[i] + 63) / 64) * 64; }

static std::vector<TensorSpec> tensor_specs(int G) {
  const long T = (long)(G / 2) * (G / 2), side = G / 2;
  std::vector<TensorSpec> v;
  auto f32 = [&](const std::string& n, long e) { v.push_back({n, 0, e, false}); };
  // every f16 weight comes as a (hi, lo) pair: W = hi + lo, lo unscaled  (see dvd_gemm_desc.B_lo).  `wide`: its GEMM has
  // N % 256 == 0 (B-side weights: N = output features; A-side V^T projections: N = tokens) and runs ONE pass on a dithered
  // copy on large grids.  The DiT block's 384-wide GEMMs keep the (hi, lo) pair: measured (round 3), dithering them too
  // saves 0.3 % of the step (they are memory-bound) and costs 8-35 % more roll-out error.
  auto f16 = [&](const std::string& n, long e, bool wide = false) {
    v.push_back({n, 1, e, wide});
    v.push_back({n + "_lo", 1, e, false});
  };
  f32("obs_w", HID * 8); f32("obs_b", HID); f32("pos", T * HID);
  f16("r_w16", (long)HID * RK); f32("r_b", HID);
  f32("c_w", (long)HID * 1024); f32("c_b", HID);
  f32("m_w", (long)HID * 1536); f32("m_b", HID);
  f32("l_w", (long)HID * 256); f32("l_b", HID);
  f32("t_w0", HID * 256); f32("t_b0", HID); f32("t_w2", HID * HID); f32("t_b2", HID);
  f32("ada_w", 6 * HID * HID); f32("ada_b", 6 * HID);
  f16("ca_wq16", HID * HID); f32("ca_bq", HID);
  f16("ca_wk16", HID * HID); f32("ca_bk", HID); f16("ca_wv16", HID * HID, true); f32("ca_bv", HID);
  f32("ca_wk32", HID * HID); f32("ca_wv32", HID * HID);
  f16("ca_wo16", HID * HID); f32("ca_bo", HID);
  f16("sa_wqk16", 2 * HID * HID, true); f32("sa_bqk", 2 * HID); f16("sa_wv16", HID * HID, true); f32("sa_bv", HID);
  f16("sa_wp16", HID * HID); f32("sa_bp", HID);
  f16("fc1_w16", 4 * HID * HID, true); f32("fc1_b", 4 * HID); f16("fc2_w16", 4 * HID * HID); f32("fc2_b", HID);
  for (const char* hw : {"pe_h", "pe_w"}) {
    f32(std::string(hw) + "0_w", (long)DEC * DEC); f32(std::string(hw) + "0_b", DEC);
    f32(std::string(hw) + "2_w", (long)DEC * DEC); f32(std::string(hw) + "2_b", DEC);
  }
  f32("pe_htab", side * DEC); f32("pe_wtab", side * DEC);
  for (int j = 0; j < 6; ++j) {
    const std::string p = "d" + std::to_string(j) + "_";
    f32(p + "n1w", DEC); f32(p + "n1b", DEC);
    f16(p + "wqk16", 2L * DEC * DEC, true); f16(p + "wv16", (long)DEC * DEC, true); f16(p + "wfc16", (long)DEC * DEC, true);
    f32(p + "n2w", DEC); f32(p + "n2b", DEC);
    f16(p + "c1w16", (long)FFN * DEC, true); f32(p + "c1b", FFN);
    f32(p + "dww", 9 * FFN); f32(p + "dwb", FFN);
    f16(p + "c2w16", (long)DEC * FFN, true); f32(p + "c2b", DEC);
  }
  f32("dec_nw", DEC); f32("dec_nb", DEC);
  f32("fin_ada_w", 2L * DEC * DEC); f32("fin_ada_b", 2 * DEC); f32("fin_w", 8 * DEC); f32("fin_b", 8);
  for (int i = 0; i < 7; ++i) {
    f32("pyr" + std::to_string(i) + "_w", (long)PYR_COUT[i] * pyr_kpad(i));
    f32("pyr" + std::to_string(i) + "_b", PYR_COUT[i]);
  }
  return v;
}

// Every weight tensor and workspace buffer the step touches is addressed by an INDEX resolved once, in
// dvd_engine_create: no string is built, hashed or compared on the launch path.
#define DVD_W_LIST(X)                                                                                              \
  X(obs_w) X(obs_b) X(pos) X(r_w16) X(r_b) X(c_w) X(c_b) X(m_w) X(m_b) X(l_w) X(l_b) X(t_w0) X(t_b0) X(t_w2) X(t_b2) \
  X(ada_w) X(ada_b) X(ca_wq16) X(ca_bq) X(ca_wk16) X(ca_bk) X(ca_wv16) X(ca_bv) X(ca_wk32) X(ca_wv32) X(ca_wo16)   \
  X(ca_bo) X(sa_wqk16) X(sa_bqk) X(sa_wv16) X(sa_bv) X(sa_wp16) X(sa_bp) X(fc1_w16) X(fc1_b) X(fc2_w16) X(fc2_b)   \
  X(pe_h0_w) X(pe_h0_b) X(pe_h2_w) X(pe_h2_b) X(pe_w0_w) X(pe_w0_b) X(pe_w2_w) X(pe_w2_b) X(pe_htab) X(pe_wtab)    \
  X(dec_nw) X(dec_nb) X(fin_ada_w) X(fin_ada_b) X(fin_w) X(fin_b)
#define DVD_WD_LIST(X) \
  X(n1w) X(n1b) X(wqk16) X(wv16) X(wfc16) X(n2w) X(n2b) X(c1w16) X(c1b) X(dww) X(dwb) X(c2w16) X(c2b)
#define DVD_B_LIST(X)                                                                                              \
  X(feat) X(kc16) X(km16) X(kl16) X(vtc16) X(vtm16) X(vtl16) X(tbuf) X(th) X(cvec) X(mod) X(finmod) X(pooled)      \
  X(petmp) X(hs) X(wsc) X(part) X(xtok32) X(xq16) X(arows16) X(rtok16) X(q16) X(kr16) X(vtr16) X(att16) X(z) X(h16) \
  X(qk16) X(vt16) X(mlp16) X(p_cat4) X(p_col) X(p_actA) X(p_actB) X(p_rows) X(p_tok32) X(w16dith)

struct WIdx {
#define X(n) int n = -1;
  DVD_W_LIST(X)
  struct Dec { DVD_WD_LIST(X) } d[6];
#undef X
  int pyr_w[7], pyr_b[7];
};
struct BIdx {
#define X(n) int n = -1;
  DVD_B_LIST(X)
#undef X
};

struct Buf {
  std::string name;
  size_t off, bytes;
};

struct GraphKey {
  int feat_mode;
  const void *x_t, *init_flow, *init_feat, *x0_out;
  bool operator==(const GraphKey& o) const {
    return feat_mode == o.feat_mode && x_t == o.x_t && init_flow == o.init_flow && init_feat == o.init_feat &&
           x0_out == o.x0_out;
  }
};
struct GraphEntry {
  GraphKey key;
  hipGraphExec_t exec;   // nullptr: the key has been run eagerly once (lazy one-time initialisation done), not yet captured
  unsigned long last_use;
};

struct Engine {
  int G, side, docs, hyp, N;
  long T, NT;
  std::vector<TensorSpec> specs;
  std::vector<const void*> wptr;
  char* ws = nullptr;
  size_t ws_bytes = 0, need_bytes = 0;
  std::vector<Buf> bufs;
  WIdx wi;
  BIdx bi;
  bool prepared = false;
  // ---- options (dvd_engine_set_option; fixed for the life of a captured graph) ----
  bool split_weights = true;   // use the lo parts (fp32-grade weights, 2x GEMM MFMAs)
  bool ffn_lo = true;          // keep the lo parts of the decoder FFN's two 1x1 convs (see dvd_engine_set_option)
  bool small_tiles = false;    // 128x128 GEMM tiles for the per-step GEMMs (small grids; set from the grid in create)
  // temporal dithering of the f16 weight rounding (dither.hip): the 256-wide GEMMs read a per-evaluation re-rounded
  // single-f16 copy of their weight (buffer w16dith) and run ONE pass; everything else keeps the (hi, lo) pair
  bool dither = false;
  unsigned dither_step = 0;            // dithering phase of every evaluation until the option is set again (NO hidden counter:
                                       // an evaluation is a pure function of its inputs and the handle's options)
  std::vector<long> dith_off;          // per weight index: element offset of its copy inside w16dith, or -1
  std::vector<int> dith_list;          // weight indices that have a copy
  long dith_total = 0;
  bool use_graphs = false;     // replay a denoiser evaluation as one hipGraph (launch-bound small grids)
  // optional per-launch timing of the dominant kernel (decoder attention) with HIP events on the launch stream:
  // a ring of event pairs, drained into running totals when it wraps, so EVERY launch of the timed region counts
  bool prof_on = false;
  std::vector<hipEvent_t> prof_ev;
  size_t prof_head = 0, prof_inflight = 0;   // pairs
  double prof_total_ms = 0.0;
  long prof_count = 0;
  int debug_stop = 0;  // parity tests: return from denoise_step after stage k (0 = run everything)
  // hipGraph cache of denoiser evaluations, keyed by what a captured launch sequence bakes in
  std::vector<GraphEntry> graphs;
  hipStream_t cap_stream = nullptr;

  const float* F(int i) const { return (const float*)wptr[i]; }
  // `wide`: the GEMM that consumes weight i takes the 256-wide kernels (its N is a multiple of 256 on a large grid)
  bool dithered(int i, bool wide) const { return dither && wide && !small_tiles && dith_off[i] >= 0; }
  const void* H(int i, bool wide = false) const {
    return dithered(i, wide) ? (const void*)((const _Float16*)(ws + bufs[bi.w16dith].off) + dith_off[i]) : wptr[i];
  }
  const void* L(int i, bool wide = false) const {                      // (hi, lo) pairs are adjacent
    return (split_weights && !dithered(i, wide)) ? wptr[i + 1] : nullptr;
  }
  const void* Lffn(int i, bool wide = false) const { return ffn_lo ? L(i, wide) : nullptr; }
  char* B(int i) const { return ws + bufs[i].off; }
  int find_w(const std::string& name) const {
    for (size_t i = 0; i < specs.size(); ++i)
      if (specs[i].name == name) return (int)i;
    return -1;
  }
  int find_b(const std::string& name) const {
    for (size_t i = 0; i < bufs.size(); ++i)
      if (bufs[i].name == name) return (int)i;
    return -1;
  }
  unsigned long graph_clock = 0;
  void drop_graphs() {
    bool any = false;
    for (auto& g : graphs) any = any || g.exec;
    if (any) (void)hipDeviceSynchronize();   // an exec may still be in flight on the caller's stream (rare path)
    for (auto& g : graphs)
      if (g.exec) (void)hipGraphExecDestroy(g.exec);
    graphs.clear();
  }
  // Evict ONE entry when the cache is full: a key that was only ever run eagerly if there is one (single-step entry
  // points pass freshly allocated tensors and leave such never-replayed keys behind), else the least recently used
  // exec - never the whole cache, so the sampler's captured evaluations survive.
  void evict_one(hipStream_t st) {
    size_t victim = 0;
    bool found_eager = false;
    for (size_t i = 0; i < graphs.size(); ++i) {
      if (!graphs[i].exec) {
        if (!found_eager || graphs[i].last_use < graphs[victim].last_use) victim = i;
        found_eager = true;
      } else if (!found_eager && graphs[i].last_use < graphs[victim].last_use) {
        victim = i;
      }
    }
    if (graphs[victim].exec) {
      (void)hipStreamSynchronize(st);        // it may still be running
      (void)hipGraphExecDestroy(graphs[victim].exec);
    }
    graphs.erase(graphs.begin() + (long)victim);
  }
};

static const int PROF_PAIRS = 4096;

static int prof_drain_one(Engine* e) {
  float ms = 0.f;
  hipEvent_t a = e->prof_ev[2 * e->prof_head], b = e->prof_ev[2 * e->prof_head + 1];
  if (hipEventSynchronize(b) != hipSuccess || hipEventElapsedTime(&ms, a, b) != hipSuccess) {
    set_error("engine_profile: event query failed");
    return DVD_E_LAUNCH;
  }
  e->prof_total_ms += ms; e->prof_count += 1;
  e->prof_head = (e->prof_head + 1) % PROF_PAIRS; e->prof_inflight -= 1;
  return DVD_OK;
}

static bool resolve_indices(Engine* e) {
  bool ok = true;
#define X(n) ok = ok && (e->wi.n = e->find_w(#n)) >= 0;
  DVD_W_LIST(X)
#undef X
  for (int j = 0; j < 6; ++j) {
    const std::string p = "d" + std::to_string(j) + "_";
#define X(n) ok = ok && (e->wi.d[j].n = e->find_w(p + #n)) >= 0;
    DVD_WD_LIST(X)
#undef X
  }
  for (int i = 0; i < 7; ++i) {
    ok = ok && (e->wi.pyr_w[i] = e->find_w("pyr" + std::to_string(i) + "_w")) >= 0;
    ok = ok && (e->wi.pyr_b[i] = e->find_w("pyr" + std::to_string(i) + "_b")) >= 0;
  }
#define X(n) ok = ok && (e->bi.n = e->find_b(#n)) >= 0;
  DVD_B_LIST(X)
#undef X
  return ok;
}

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
  // dithered single-f16 copies of the weights of the 256-wide GEMMs (persistent: rewritten before every evaluation)
  e->dith_off.assign(e->specs.size(), -1);
  e->dith_list.clear();
  e->dith_total = 0;
  for (size_t i = 0; i + 1 < e->specs.size(); ++i) {
    if (!e->specs[i].wide) continue;
    e->dith_off[i] = e->dith_total;
    e->dith_list.push_back((int)i);
    e->dith_total += (e->specs[i].nelem + 127) / 128 * 128;
  }
  add("w16dith", (size_t)e->dith_total * 2);
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
  if (dtype == 2) { d.dtype = 0; d.small_tiles = 1; }     // dtype 2 = f16 operands on 128x128 tiles (small grids)
  if (dtype == 3) { d.dtype = 0; d.small_tiles = 2; }     // dtype 3 = the same family, many rows (see dvd_gemm_desc)
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
  // a few thousand rows per GEMM put < 50 of the 256x256 persistent tiles on 256 CUs: small grids take 128x128 tiles.
  // Decided from the GRID only (never the batch): a document gets the same kernels alone or in a batch.
  e->small_tiles = e->T <= 1024;
  e->dither = !e->small_tiles;
  if (!resolve_indices(e)) {
    delete e;
    set_error("engine_create: internal tensor/buffer table mismatch");
    return DVD_E_STATE;
  }
  *handle = e;
  return DVD_OK;
}

extern "C" int dvd_engine_destroy(void* handle) {
  Engine* e = (Engine*)handle;
  if (!e) return DVD_OK;
  e->drop_graphs();
  if (e->cap_stream) (void)hipStreamDestroy(e->cap_stream);
  for (auto& ev : e->prof_ev) (void)hipEventDestroy(ev);
  delete e;
  return DVD_OK;
}

extern "C" long dvd_engine_workspace_bytes(void* handle) { return handle ? (long)((Engine*)handle)->need_bytes : -1; }

extern "C" int dvd_engine_bind_workspace(void* handle, void* ws, long bytes) {
  DVD_REQUIRE(handle && ws, "engine_bind_workspace: null pointer");
  Engine* e = (Engine*)handle;
  DVD_REQUIRE((size_t)bytes >= e->need_bytes && ((uintptr_t)ws % 256) == 0,
              "engine_bind_workspace: need %zu bytes, 256-byte aligned (got %ld)", e->need_bytes, bytes);
  e->ws = (char*)ws; e->ws_bytes = bytes; e->prepared = false;
  e->drop_graphs();   // captured launches bake the workspace addresses in
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
      if (e->wptr[i] != dev_ptr) e->drop_graphs();   // captured launches bake the weight addresses in
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
    e->prof_ev.resize(2 * PROF_PAIRS);
    for (auto& ev : e->prof_ev)
      if (hipEventCreate(&ev) != hipSuccess) { set_error("engine_profile: hipEventCreate failed"); return DVD_E_LAUNCH; }
  }
  e->prof_on = enable != 0;
  e->prof_head = 0; e->prof_inflight = 0; e->prof_total_ms = 0.0; e->prof_count = 0;
  return DVD_OK;
}

extern "C" int dvd_engine_profile_read(void* handle, int* launches, double* total_ms) {
  DVD_REQUIRE(handle && launches && total_ms, "engine_profile_read: null pointer");
  Engine* e = (Engine*)handle;
  while (e->prof_inflight) TRY(prof_drain_one(e));
  *launches = (int)e->prof_count; *total_ms = e->prof_total_ms;
  e->prof_total_ms = 0.0; e->prof_count = 0;
  return DVD_OK;
}

extern "C" int dvd_engine_set_option(void* handle, const char* name, int value) {
  DVD_REQUIRE(handle && name, "engine_set_option: null pointer");
  Engine* e = (Engine*)handle;
  if (strcmp(name, "split_weights") == 0) { e->split_weights = value != 0; e->drop_graphs(); return DVD_OK; }
  if (strcmp(name, "small_tiles") == 0) { e->small_tiles = value != 0; e->drop_graphs(); return DVD_OK; }
  if (strcmp(name, "ffn_lo") == 0) { e->ffn_lo = value != 0; e->drop_graphs(); return DVD_OK; }
  if (strcmp(name, "dither") == 0) { e->dither = value != 0; e->drop_graphs(); return DVD_OK; }
  if (strcmp(name, "dither_step") == 0) { e->dither_step = (unsigned)value; return DVD_OK; }
  if (strcmp(name, "graphs") == 0) {
    e->use_graphs = value != 0;
    if (!e->use_graphs) e->drop_graphs();
    return DVD_OK;
  }
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
  const int i = e->find_b(name);
  DVD_REQUIRE(i >= 0, "engine_debug_buffer: unknown buffer '%s'", name);
  *ptr = e->B(i); *bytes = (long)e->bufs[i].bytes;
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
  float* cat4 = (float*)e->B(e->bi.p_cat4);
  float* col = (float*)e->B(e->bi.p_col);
  float* actA = (float*)e->B(e->bi.p_actA);
  float* actB = (float*)e->B(e->bi.p_actB);
  const float* pos = e->F(e->wi.pos);
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
      float* outp = act[wi];
      wi ^= 1;
      if (sc == 1 && cin % 16 == 0 && kp == 9 * cin) {
        // every layer but the first (4 planar input channels): implicit GEMM with the arithmetic - and the bits - of the
        // im2col + GEMM pair it replaces (level_1's second conv alone wrote and read a 604 MB matrix)
        TRY(dvd_conv3x3_nhwc(cur, cin, e->F(e->wi.pyr_w[l.idx]), kp, e->F(e->wi.pyr_b[l.idx]), outp, cout, l.hw, l.hw,
                                    1, stream));
      } else {
        TRY(dvd_im2col3x3(cur, sc, sy, sx, col, kp, cin, l.hw, l.hw, stream));
        TRY(gemm(1, l.hw * l.hw, cout, kp, 1, col, kp, 0, e->F(e->wi.pyr_w[l.idx]), kp, 0, outp, cout, 0, nullptr, 0, 0,
                 e->F(e->wi.pyr_b[l.idx]), 0, /*relu*/ 2, nullptr, 0, nullptr, 0, nullptr, 0, 0, stream));
      }
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
    float* feat = (float*)e->B(e->bi.feat) + (size_t)d * G * G * 256;
    if (G == 64) {
      if (hipMemcpyAsync(feat, cur, (size_t)64 * 64 * 256 * 4, hipMemcpyDeviceToDevice, st) != hipSuccess) {
        set_error("engine_prepare_docs: device copy failed");
        return DVD_E_LAUNCH;
      }
    } else {
      TRY(dvd_resize_bilinear_nhwc(cur, feat, 256, 64, 64, G, G, stream));
    }
  }
  // c / m / l patch embeddings (+pos) and their K / V^T projections (shared cross_attn weights), for GROUPS of documents
  // (round 5): per document these are 1024-row GEMMs at the reference's grid - 24 workgroups on 256 CUs, nine of them per
  // document, 20 of a 32-document batch's 254 ms.  A group's patch rows and tokens live in the prepare-time scratch, which is
  // free once the pyramid loop is done; the group is as large as it allows (all 32 documents at G = 64, 5 at G = 288).
  // Row-stacked GEMMs and a batch dimension for the transposed V projection: per-row arithmetic unchanged, same bits.
  {
    // the prepare-time scratch from the im2col region to its end (p_col, p_actA, p_actB, p_rows, p_tok32 are laid out in
    // this order by plan() and none of them is live here); p_rows + p_tok32 alone hold one document at any grid
    const size_t per_doc = (size_t)T * (1536 + HID) * 4;
    {   // the grouping below treats the five regions as ONE contiguous scratch - say so where a re-ordered
        // plan() would otherwise underflow `avail` and let rows_g / tok_g run into live buffers
      const int chain[5] = {e->bi.p_col, e->bi.p_actA, e->bi.p_actB, e->bi.p_rows, e->bi.p_tok32};
      for (int i = 0; i + 1 < 5; ++i)
        DVD_REQUIRE(e->bufs[chain[i]].off < e->bufs[chain[i + 1]].off &&
                        e->bufs[chain[i]].off + e->bufs[chain[i]].bytes <= e->bufs[chain[i + 1]].off &&
                        e->bufs[chain[i + 1]].off - (e->bufs[chain[i]].off + e->bufs[chain[i]].bytes) < 4096,
                    "engine_prepare_docs: the prepare-time scratch regions are not laid out back to back (plan() changed?)");
    }
    const size_t avail = e->bufs[e->bi.p_tok32].off + e->bufs[e->bi.p_tok32].bytes - e->bufs[e->bi.p_col].off;
    DVD_REQUIRE(avail >= per_doc, "engine_prepare_docs: prepare scratch smaller than one document's rows");
    const int gmax = (int)std::max<size_t>(1, std::min<size_t>((size_t)e->docs, avail / per_doc));
    const WIdx& wx = e->wi;
    const BIdx& bx = e->bi;
    for (int d0 = 0; d0 < e->docs; d0 += gmax) {
      const int gd = std::min(gmax, e->docs - d0);
      float* rows_g = col;
      float* tok_g = col + (size_t)gd * T * 1536;
      struct S { const float* src; long sn, sc, sy, sx; int c; int w, b, k16, vt16; };
      const S streams[3] = {
          {(const float*)e->B(bx.feat) + (size_t)d0 * G * G * 256, (long)G * G * 256, 1, (long)G * 256, 256, 256, wx.c_w, wx.c_b,
           bx.kc16, bx.vtc16},
          {mask_y512 + (size_t)d0 * 384 * G * G, 384L * G * G, (long)G * G, G, 1, 384, wx.m_w, wx.m_b, bx.km16, bx.vtm16},
          {line_msk + (size_t)d0 * 64 * G * G, 64L * G * G, (long)G * G, G, 1, 64, wx.l_w, wx.l_b, bx.kl16, bx.vtl16}};
      for (const S& s : streams) {
        const int K4 = 4 * s.c;
        DVD_REQUIRE((long)gd * T < (1l << 31), "engine_prepare_docs: group too large");
        TRY(dvd_patch_rows(s.src, s.sn, s.sc, s.sy, s.sx, rows_g, K4, gd, s.c, G, stream));
        TRY(gemm(1, (int)(gd * T), HID, K4, 1, rows_g, K4, 0, e->F(s.w), K4, 0, tok_g, HID, 0, nullptr, 0, 0, e->F(s.b), 0, 0,
                 pos, (int)T, nullptr, 0, nullptr, 0, 0, stream));
        _Float16* k16 = (_Float16*)e->B(s.k16) + (size_t)d0 * T * HID;
        _Float16* vt16 = (_Float16*)e->B(s.vt16) + (size_t)d0 * T * HID;
        TRY(gemm(1, (int)(gd * T), HID, HID, 1, tok_g, HID, 0, e->F(e->wi.ca_wk32), HID, 0, nullptr, 0, 0, k16, HID, 0,
                 e->F(e->wi.ca_bk), 0, 0, nullptr, 0, nullptr, 0, nullptr, 0, 0, stream));
        TRY(gemm(1, HID, (int)T, HID, gd, e->F(e->wi.ca_wv32), HID, 0, tok_g, HID, (long)T * HID, nullptr, 0, 0, vt16, (int)T,
                 (long)T * HID, e->F(e->wi.ca_bv), 1, 0, nullptr, 0, nullptr, 0, nullptr, 0, 0, stream));
      }
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
    TRY(dvd_nhwc_to_nchw((const float*)e->B(e->bi.feat) + (size_t)d * e->G * e->G * 256,
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
// The launch sequence of one evaluation.  The embedded timestep is read from `tbuf` (written by the caller before this
// sequence), so the sequence depends on (feat_mode, the four I/O pointers) only and can be captured into a hipGraph.
static int enqueue_step(Engine* e, const float* x_t, int feat_mode, const float* init_flow,
                        const float* init_feat_nchw, float* x0_out, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  const int G = e->G, N = e->N, side = e->side, hyp = e->hyp;
  const int T = (int)e->T;
  const long NT = e->NT;
  // gemm() dtype code: f16 operands; on small grids 128x128 tiles, and from 16 384 token rows on (a batch of >= 8 documents
  // x 2 hypotheses at G = 64) the 256x256 kernel in the bit-compatible two-sweep form for the 256-wide shapes
  const int F16 = e->small_tiles ? (e->NT >= 16384 ? 3 : 2) : 0;
  constexpr int F16S = 2;                     // ... and 128x128 tiles on every grid (a function of the GEMM, not of the batch)

  float* tbuf = (float*)e->B(e->bi.tbuf);
  float* th = (float*)e->B(e->bi.th);
  float* cvec = (float*)e->B(e->bi.cvec);
  float* mod = (float*)e->B(e->bi.mod);
  float* finmod = (float*)e->B(e->bi.finmod);
  float* xtok32 = (float*)e->B(e->bi.xtok32);
  _Float16* xq16 = (_Float16*)e->B(e->bi.xq16);
  _Float16* arows16 = (_Float16*)e->B(e->bi.arows16);
  _Float16* rtok16 = (_Float16*)e->B(e->bi.rtok16);
  _Float16* q16 = (_Float16*)e->B(e->bi.q16);
  _Float16* kr16 = (_Float16*)e->B(e->bi.kr16);
  _Float16* vtr16 = (_Float16*)e->B(e->bi.vtr16);
  _Float16* att16 = (_Float16*)e->B(e->bi.att16);
  float* z = (float*)e->B(e->bi.z);
  _Float16* h16 = (_Float16*)e->B(e->bi.h16);
  _Float16* qk16 = (_Float16*)e->B(e->bi.qk16);
  _Float16* vt16 = (_Float16*)e->B(e->bi.vt16);
  _Float16* mlp16 = (_Float16*)e->B(e->bi.mlp16);

  // --- timestep embedding and the two adaLN tables (one row: t is batch-global in sampling) ---
  TRY(dvd_small_linear(tbuf, 1, e->F(e->wi.t_w0), e->F(e->wi.t_b0), th, HID, 1, 256, HID, 256, 2, 1, stream));
  TRY(dvd_small_linear(th, HID, e->F(e->wi.t_w2), e->F(e->wi.t_b2), cvec, HID, 1, HID, HID, HID, 0, 0, stream));
  TRY(dvd_small_linear(cvec, HID, e->F(e->wi.ada_w), e->F(e->wi.ada_b), mod, 6 * HID, 1, HID, 6 * HID, HID, 1, 0, stream));
  TRY(dvd_small_linear(cvec, HID, e->F(e->wi.fin_ada_w), e->F(e->wi.fin_ada_b), finmod, 2 * DEC, 1, DEC, 2 * DEC, HID, 1, 0,
                       stream));
  const float *sh_a = mod, *sc_a = mod + HID, *g_a = mod + 2 * HID, *sh_m = mod + 3 * HID, *sc_m = mod + 4 * HID,
              *g_m = mod + 5 * HID;

  // --- tokens ---
  TRY(dvd_embed_obs_ln(x_t, e->F(e->wi.obs_w), e->F(e->wi.obs_b), e->F(e->wi.pos), xtok32, xq16, N, G, stream));
  TRY(dvd_build_r_rows((const float*)e->B(e->bi.feat), init_feat_nchw, init_flow, arows16, RK, N, G, hyp, feat_mode,
                       stream));
  TRY(gemm(F16, (int)NT, HID, RK, 1, arows16, RK, 0, e->H(e->wi.r_w16), RK, 0, nullptr, 0, 0, rtok16, HID, 0, e->F(e->wi.r_b), 0,
           0, e->F(e->wi.pos), T, nullptr, 0, nullptr, 0, 0, stream, nullptr, e->L(e->wi.r_w16)));

  // --- parallel cross-attention of the shared query against the 4 streams (:237-265) ---
  TRY(gemm(F16, (int)NT, HID, HID, 1, xq16, HID, 0, e->H(e->wi.ca_wq16), HID, 0, nullptr, 0, 0, q16, HID, 0, e->F(e->wi.ca_bq),
           0, 0, nullptr, 0, nullptr, 0, nullptr, 0, 0, stream, nullptr, e->L(e->wi.ca_wq16)));
  TRY(gemm(F16, (int)NT, HID, HID, 1, rtok16, HID, 0, e->H(e->wi.ca_wk16), HID, 0, nullptr, 0, 0, kr16, HID, 0,
           e->F(e->wi.ca_bk), 0, 0, nullptr, 0, nullptr, 0, nullptr, 0, 0, stream, nullptr, e->L(e->wi.ca_wk16)));
  // (the two 384 x T x 384 V^T projections take the 128 x 128 kernel on every grid: K = 384 is six 64-deep slabs and a
  //  token panel is read by two row tiles only, so the 256 x 256 kernel - one workgroup per CU, every slab's loads waited
  //  for at its end - runs them latency-bound: 2.13 vs 0.93 ms and 0.58 vs 0.21 ms at the bench shape, benchmarks/gemm_vt384_time.py)
  TRY(gemm(F16S, HID, T, HID, N, e->H(e->wi.ca_wv16, true), HID, 0, rtok16, HID, (long)T * HID, nullptr, 0, 0, vtr16, T,
           (long)HID * T, e->F(e->wi.ca_bv), 1, 0, nullptr, 0, nullptr, 0, nullptr, 0, 0, stream, e->L(e->wi.ca_wv16, true), nullptr));
  {
    const int kn[3] = {e->bi.kc16, e->bi.km16, e->bi.kl16};
    const int vn[3] = {e->bi.vtc16, e->bi.vtm16, e->bi.vtl16};
    for (int s = 0; s < 3; ++s)
      TRY(attn(64, 6, N, T, T, hyp, q16, HID, (long)T * HID, e->B(kn[s]), HID, (long)T * HID, e->B(vn[s]), T,
               (long)HID * T, att16 + (size_t)s * NT * HID, HID, (long)T * HID, 0.125f, stream));
    TRY(attn(64, 6, N, T, T, 1, q16, HID, (long)T * HID, kr16, HID, (long)T * HID, vtr16, T, (long)HID * T,
             att16 + (size_t)3 * NT * HID, HID, (long)T * HID, 0.125f, stream));
  }
  // x_s = x + out_proj(attn_s)  -> z[:, 384 s : 384 (s+1)]   (stream order cond, msk6, line, r == cat order :623)
  TRY(gemm(F16, (int)NT, HID, HID, 4, att16, HID, NT * HID, e->H(e->wi.ca_wo16), HID, 0, z, DEC, HID, nullptr, 0, 0,
           e->F(e->wi.ca_bo), 0, 0, nullptr, 0, nullptr, 0, xtok32, HID, 0, stream, nullptr, e->L(e->wi.ca_wo16)));

  if (e->debug_stop == 1) return check_launch("engine_denoise_step(stop 1)");
  // --- per stream: gated self-attention (:268-289) ---
  TRY(dvd_layernorm_rows(z, DEC, HID, h16, HID, NT * HID, 4, NT, HID, nullptr, nullptr, sh_a, sc_a, 0, (int)NT, 1e-6f,
                         stream));
  TRY(gemm(F16, (int)(4 * NT), 2 * HID, HID, 1, h16, HID, 0, e->H(e->wi.sa_wqk16, true), HID, 0, nullptr, 0, 0, qk16, 2 * HID, 0,
           e->F(e->wi.sa_bqk), 0, 0, nullptr, 0, nullptr, 0, nullptr, 0, 0, stream, nullptr, e->L(e->wi.sa_wqk16, true)));
  TRY(gemm(F16S, HID, T, HID, 4 * N, e->H(e->wi.sa_wv16, true), HID, 0, h16, HID, (long)T * HID, nullptr, 0, 0, vt16, T,
           (long)HID * T, e->F(e->wi.sa_bv), 1, 0, nullptr, 0, nullptr, 0, nullptr, 0, 0, stream, e->L(e->wi.sa_wv16, true), nullptr));
  TRY(attn(64, 6, 4 * N, T, T, 1, qk16, 2 * HID, (long)T * 2 * HID, qk16 + HID, 2 * HID, (long)T * 2 * HID, vt16, T,
           (long)HID * T, att16, HID, (long)T * HID, 0.125f, stream));
  TRY(gemm(F16, (int)NT, HID, HID, 4, att16, HID, NT * HID, e->H(e->wi.sa_wp16), HID, 0, z, DEC, HID, nullptr, 0, 0,
           e->F(e->wi.sa_bp), 0, 0, nullptr, 0, g_a, (int)NT, z, DEC, HID, stream, nullptr, e->L(e->wi.sa_wp16)));
  // --- per stream: gated MLP (:271-292) ---
  TRY(dvd_layernorm_rows(z, DEC, HID, h16, HID, NT * HID, 4, NT, HID, nullptr, nullptr, sh_m, sc_m, 0, (int)NT, 1e-6f,
                         stream));
  TRY(gemm(F16, (int)(4 * NT), 4 * HID, HID, 1, h16, HID, 0, e->H(e->wi.fc1_w16, true), HID, 0, nullptr, 0, 0, mlp16, 4 * HID, 0,
           e->F(e->wi.fc1_b), 0, /*gelu*/ 1, nullptr, 0, nullptr, 0, nullptr, 0, 0, stream, nullptr, e->L(e->wi.fc1_w16, true)));
  TRY(gemm(F16, (int)NT, HID, 4 * HID, 4, mlp16, 4 * HID, NT * 4 * HID, e->H(e->wi.fc2_w16), 4 * HID, 0, z, DEC, HID,
           nullptr, 0, 0, e->F(e->wi.fc2_b), 0, 0, nullptr, 0, g_m, (int)NT, z, DEC, HID, stream, nullptr, e->L(e->wi.fc2_w16)));

  if (e->debug_stop == 2) return check_launch("engine_denoise_step(stop 2)");
  // --- decoder: adaptive 2-D positional encoding (idf/cross_attn.py:143-157) ---
  float* pooled = (float*)e->B(e->bi.pooled);
  float* petmp = (float*)e->B(e->bi.petmp);
  float* hs = (float*)e->B(e->bi.hs);
  float* wsc = (float*)e->B(e->bi.wsc);
  TRY(dvd_colmean(z, (float*)e->B(e->bi.part), pooled, N, T, DEC, COLCHUNKS, stream));
  TRY(dvd_small_linear(pooled, DEC, e->F(e->wi.pe_h0_w), e->F(e->wi.pe_h0_b), petmp, DEC, N, DEC, DEC, DEC, 0, 2, stream));
  TRY(dvd_small_linear(petmp, DEC, e->F(e->wi.pe_h2_w), e->F(e->wi.pe_h2_b), hs, DEC, N, DEC, DEC, DEC, 0, 3, stream));
  TRY(dvd_small_linear(pooled, DEC, e->F(e->wi.pe_w0_w), e->F(e->wi.pe_w0_b), petmp, DEC, N, DEC, DEC, DEC, 0, 2, stream));
  TRY(dvd_small_linear(petmp, DEC, e->F(e->wi.pe_w2_w), e->F(e->wi.pe_w2_b), wsc, DEC, N, DEC, DEC, DEC, 0, 3, stream));
  TRY(dvd_posenc_add(z, hs, wsc, e->F(e->wi.pe_htab), e->F(e->wi.pe_wtab), N, side, DEC, stream));

  if (e->debug_stop == 3) return check_launch("engine_denoise_step(stop 3)");
  // --- decoder layers (idf/cross_attn.py:377-396) ---
  _Float16* f1 = mlp16;
  _Float16* f2 = mlp16 + (size_t)NT * FFN;
  for (int j = 0; j < 6; ++j) {
    const WIdx::Dec& dw = e->wi.d[j];
    TRY(dvd_layernorm_rows(z, DEC, 0, h16, DEC, 0, 1, NT, DEC, e->F(dw.n1w), e->F(dw.n1b), nullptr, nullptr, 0,
                           1, 1e-5f, stream));
    TRY(gemm(F16, (int)NT, 2 * DEC, DEC, 1, h16, DEC, 0, e->H(dw.wqk16, true), DEC, 0, nullptr, 0, 0, qk16, 2 * DEC, 0,
             nullptr, 0, 0, nullptr, 0, nullptr, 0, nullptr, 0, 0, stream, nullptr, e->L(dw.wqk16, true)));
    TRY(gemm(F16, DEC, T, DEC, N, e->H(dw.wv16, true), DEC, 0, h16, DEC, (long)T * DEC, nullptr, 0, 0, vt16, T,
             (long)DEC * T, nullptr, 0, 0, nullptr, 0, nullptr, 0, nullptr, 0, 0, stream, e->L(dw.wv16, true), nullptr));
    const bool timed = e->prof_on;
    size_t slot = 0;
    if (timed) {
      if (e->prof_inflight == (size_t)PROF_PAIRS) TRY(prof_drain_one(e));   // ring full: fold the oldest pair into the totals
      slot = (e->prof_head + e->prof_inflight) % PROF_PAIRS;
      (void)hipEventRecord(e->prof_ev[2 * slot], st);
    }
    TRY(attn(256, 6, N, T, T, 1, qk16, 2 * DEC, (long)T * 2 * DEC, qk16 + DEC, 2 * DEC, (long)T * 2 * DEC, vt16, T,
             (long)DEC * T, att16, DEC, (long)T * DEC, 0.0625f, stream));
    if (timed) { (void)hipEventRecord(e->prof_ev[2 * slot + 1], st); e->prof_inflight += 1; }
    TRY(gemm(F16, (int)NT, DEC, DEC, 1, att16, DEC, 0, e->H(dw.wfc16, true), DEC, 0, z, DEC, 0, nullptr, 0, 0, nullptr, 0,
             0, nullptr, 0, nullptr, 0, z, DEC, 0, stream, nullptr, e->L(dw.wfc16, true)));
    TRY(dvd_layernorm_rows(z, DEC, 0, h16, DEC, 0, 1, NT, DEC, e->F(dw.n2w), e->F(dw.n2b), nullptr, nullptr, 0,
                           1, 1e-5f, stream));
    TRY(gemm(F16, (int)NT, FFN, DEC, 1, h16, DEC, 0, e->H(dw.c1w16, true), DEC, 0, nullptr, 0, 0, f1, FFN, 0,
             e->F(dw.c1b), 0, 2, nullptr, 0, nullptr, 0, nullptr, 0, 0, stream, nullptr, e->Lffn(dw.c1w16, true)));
    TRY(dvd_dwconv3x3(f1, f2, e->F(dw.dww), e->F(dw.dwb), N, side, FFN, stream));
    TRY(gemm(F16, (int)NT, DEC, FFN, 1, f2, FFN, 0, e->H(dw.c2w16, true), FFN, 0, z, DEC, 0, nullptr, 0, 0,
             e->F(dw.c2b), 0, 2, nullptr, 0, nullptr, 0, z, DEC, 0, stream, nullptr, e->Lffn(dw.c2w16, true)));
    if (e->debug_stop == 4 + j) return check_launch("engine_denoise_step(stop 4+j)");
  }

  // --- final LayerNorm + FinalLayer2 + unpatchify + init_flow (:457; idf/cross_model.py:329-336,553-566,645-646) ---
  TRY(dvd_final_tokens(z, e->F(e->wi.dec_nw), e->F(e->wi.dec_nb), finmod, finmod + DEC, 0, (int)NT, e->F(e->wi.fin_w),
                       e->F(e->wi.fin_b), init_flow, x0_out, nullptr, N, G, stream));
  return check_launch("engine_denoise_step");
}

extern "C" int dvd_engine_denoise_step(void* handle, const float* x_t, float t_embed, int feat_mode,
                                       const float* init_flow, const float* init_feat_nchw, float* x0_out,
                                       void* stream) {
  DVD_REQUIRE(handle && x_t && init_flow && x0_out, "engine_denoise_step: null pointer");
  DVD_REQUIRE(feat_mode >= 0 && feat_mode <= 3 && (feat_mode != 3 || init_feat_nchw),
              "engine_denoise_step: feat_mode %d (3 needs init_feat)", feat_mode);
  Engine* e = (Engine*)handle;
  TRY(require_ready(e, true));
  DVD_REQUIRE(4 * e->NT < (1L << 31), "engine_denoise_step: batch too large for 32-bit row indices (NT=%ld)", e->NT);
  hipStream_t st = (hipStream_t)stream;
  set_scalar_kernel<<<1, 1, 0, st>>>((float*)e->B(e->bi.tbuf), t_embed);
  if (e->dither && !e->small_tiles) {
    // re-round the weights of the 256-wide GEMMs for this evaluation (outside any captured graph: the phase changes)
    _Float16* dst = (_Float16*)e->B(e->bi.w16dith);
    for (int i : e->dith_list)
      TRY(dvd_dither_f16(e->wptr[i], e->wptr[i + 1], dst + e->dith_off[i], e->specs[i].nelem, (unsigned)e->dith_off[i],
                         e->dither_step, stream));
  }
  if (!e->use_graphs || e->prof_on || e->debug_stop)
    return enqueue_step(e, x_t, feat_mode, init_flow, init_feat_nchw, x0_out, stream);

  // ---- hipGraph replay: ~110 launches of a small-grid evaluation become one graph launch.  A key's first use runs
  // eagerly (every lazy one-time initialisation happens outside capture), its second use captures, later uses replay.
  const GraphKey key{feat_mode, x_t, init_flow, feat_mode == 3 ? init_feat_nchw : nullptr, x0_out};
  GraphEntry* ge = nullptr;
  for (auto& g : e->graphs)
    if (g.key == key) { ge = &g; break; }
  e->graph_clock += 1;
  if (!ge) {
    if (e->graphs.size() >= 32) e->evict_one(st);
    e->graphs.push_back({key, nullptr, e->graph_clock});
    return enqueue_step(e, x_t, feat_mode, init_flow, init_feat_nchw, x0_out, stream);
  }
  ge->last_use = e->graph_clock;
  if (!ge->exec) {
    if (!e->cap_stream && hipStreamCreateWithFlags(&e->cap_stream, hipStreamNonBlocking) != hipSuccess) {
      set_error("engine_denoise_step: cannot create the capture stream");
      return DVD_E_LAUNCH;
    }
    hipGraph_t graph = nullptr;
    if (hipStreamBeginCapture(e->cap_stream, hipStreamCaptureModeRelaxed) != hipSuccess) {
      set_error("engine_denoise_step: hipStreamBeginCapture failed");
      return DVD_E_LAUNCH;
    }
    const int rc = enqueue_step(e, x_t, feat_mode, init_flow, init_feat_nchw, x0_out, e->cap_stream);
    const hipError_t ec = hipStreamEndCapture(e->cap_stream, &graph);
    if (rc != DVD_OK) { if (graph) (void)hipGraphDestroy(graph); return rc; }
    if (ec != hipSuccess || !graph || hipGraphInstantiate(&ge->exec, graph, nullptr, nullptr, 0) != hipSuccess) {
      if (graph) (void)hipGraphDestroy(graph);
      ge->exec = nullptr;
      set_error("engine_denoise_step: graph capture/instantiate failed (%s)", hipGetErrorString(ec));
      return DVD_E_LAUNCH;
    }
    (void)hipGraphDestroy(graph);
  }
  if (hipGraphLaunch(ge->exec, st) != hipSuccess) {
    set_error("engine_denoise_step: hipGraphLaunch failed");
    return DVD_E_LAUNCH;
  }
  return check_launch("engine_denoise_step(graph)");
}
