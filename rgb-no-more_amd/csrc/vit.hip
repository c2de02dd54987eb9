// Composite stages of the JPEG-ViT (reference: models/plainvit.py:493-529 TransformerEncoderBlock,
// :157-218 PatchEmbedding_DCT_Group, :542-557 ClassificationHead): each is one C-ABI call that enqueues
// its kernels on the caller's stream, so the Python side pays one ctypes call per autograd node.
#include "common.h"
#include "../../include/rgbnm.h"

#define TRY(x)                 \
  do {                         \
    int rc__ = (x);            \
    if (rc__ != 0) return rc__; \
  } while (0)

static inline size_t max_sz(size_t a, size_t b) { return a > b ? a : b; }

#include <string.h>
#include <atomic>
#include <mutex>
#include <vector>
#include "internal.h"
namespace {
std::mutex g_trace_mu;                    // trace records / event pool: measurement feature, any thread may launch
// ---- optional per-kernel timing with HIP events recorded on the launch stream (option "trace" = bit mask of tags)
struct TraceRec { hipEvent_t a, b; int tag; double flops, bytes; };
std::vector<TraceRec> g_recs;
std::vector<hipEvent_t> g_event_pool;     // events are created once and recycled (hipEventCreate costs ~10 us)
hipEvent_t take_event() {
  if (!g_event_pool.empty()) { hipEvent_t e = g_event_pool.back(); g_event_pool.pop_back(); return e; }
  hipEvent_t e = nullptr;
  if (hipEventCreate(&e) != hipSuccess) return nullptr;
  return e;
}
// process-wide configuration switches (A/B experiments): atomics, so reading them from concurrent calls is race-free;
// they select between parity-tested kernels and are meant to be set before work is enqueued
struct Opt { const char* name; std::atomic<int> value; };
Opt g_opts[] = {{"nt_staged", 1}, {"tn_tr", 1}, {"tn_pipe", 1}, {"attn_v2", 1}, {"nt_wres", 1}, {"nt_kpipe", 1}, {"attn_persist", 1}, {"ln_fuse", 1}, {"tn_group", 2}, {"trace", 0}, {"tn_wgs", 512}, {"mlp_fuse", 1}, {"nt_cold", 0}, {"mlp_bwd", 1}, {"nt_small", 1}, {"mlp_dmast", 1}, {"gelu_table", 1}, {"ln_rows", 1}, {"kp8", 1}, {"win_xcd", 1}, {"kp_persist", 1}, {"fwd_chain", 1}, {"bwd_chain", 1}, {"tn_direct", 1}, {"tn_pack", 1}, {"tn_wide", 1}};
}

int rgbnm_trace_begin(int tag, double flops, double bytes, hipStream_t st) {
  int mask = 0;
  for (auto& o : g_opts) if (!strcmp(o.name, "trace")) mask = o.value.load(std::memory_order_relaxed);
  if (!((mask >> tag) & 1)) return -1;
  std::lock_guard<std::mutex> lk(g_trace_mu);
  TraceRec r;
  r.a = take_event();
  r.b = take_event();
  if (!r.a || !r.b) return -1;
  r.tag = tag; r.flops = flops; r.bytes = bytes;
  (void)hipEventRecord(r.a, st);
  g_recs.push_back(r);
  return (int)g_recs.size() - 1;
}
void rgbnm_trace_end(int slot, hipStream_t st) {
  if (slot < 0) return;
  std::lock_guard<std::mutex> lk(g_trace_mu);
  if (slot >= 0 && slot < (int)g_recs.size()) (void)hipEventRecord(g_recs[slot].b, st);
}

extern "C" {

int rgbnm_abi_version(void) { return RGBNM_ABI_VERSION; }

int rgbnm_set_option(const char* name, int value) {
  for (auto& o : g_opts)
    if (name && !strcmp(o.name, name)) { o.value.store(value, std::memory_order_relaxed); return RGBNM_OK; }
  return RGBNM_EINVAL;
}
int rgbnm_get_option(const char* name) {
  for (auto& o : g_opts)
    if (name && !strcmp(o.name, name)) return o.value.load(std::memory_order_relaxed);
  return -1;
}

int rgbnm_trace_reserve(int n_events) {
  if (n_events < 0 || n_events > (1 << 16)) return RGBNM_EINVAL;
  std::lock_guard<std::mutex> lk(g_trace_mu);
  while ((int)g_event_pool.size() < n_events) {
    hipEvent_t e = nullptr;
    if (hipEventCreate(&e) != hipSuccess) return RGBNM_ELAUNCH;
    g_event_pool.push_back(e);
  }
  return RGBNM_OK;
}

int rgbnm_trace_collect(int tag, double* ms_total, double* flops_total, double* bytes_total, int* count) {
  double ms = 0, fl = 0, by = 0;
  int n = 0;
  std::lock_guard<std::mutex> lk(g_trace_mu);
  std::vector<TraceRec> keep;
  for (auto& r : g_recs) {
    if (r.tag != tag) { keep.push_back(r); continue; }
    if (hipEventSynchronize(r.b) != hipSuccess) return RGBNM_ELAUNCH;
    float t = 0.f;
    if (hipEventElapsedTime(&t, r.a, r.b) != hipSuccess) return RGBNM_ELAUNCH;
    ms += t; fl += r.flops; by += r.bytes; ++n;
    g_event_pool.push_back(r.a);
    g_event_pool.push_back(r.b);
  }
  g_recs.swap(keep);
  if (ms_total) *ms_total = ms;
  if (flops_total) *flops_total = fl;
  if (bytes_total) *bytes_total = by;
  if (count) *count = n;
  return RGBNM_OK;
}

const char* rgbnm_strerror(int code) {
  switch (code) {
    case 0: return "ok";
    case RGBNM_EINVAL: return "invalid argument (null pointer, bad shape, alignment or dtype)";
    case RGBNM_ELAUNCH: return "HIP kernel launch failed";
    case RGBNM_EWORKSPACE: return "workspace too small";
    default: return "unknown rgbnm error";
  }
}

// dx = dres + LayerNorm'(A . W^T) in one launch (gemm_nt_kpipe.hip) + two jobs for the batched reduction.
// false: not eligible (fp32, E != 192, small M, option ln_fuse = 0) -> the caller runs GEMM and LayerNorm separately.
static bool fused_dx_lnbwd(int dt, const void* A, int lda, const void* Wt, int ldw, const void* x, const float* gamma,
                           const float* mean, const float* rstd, const void* dres, void* dx, float* dgamma,
                           float* dbeta, int M, int E, int K, void* ws, size_t ws_bytes, hipStream_t st) {
  if (dt != RGBNM_DT_BF16 || E != 192 || !rgbnm_get_option("ln_fuse") || !rgbnm_get_option("nt_kpipe")) return false;
  if (ws_bytes < (size_t)cdiv(M, cdiv(M, 512)) * 2 * E * sizeof(float)) return false;    // up to 512 row panels
  int npanels = 0;
  const int rc = rgbnm_launch_nt_kpipe_lnbwd(A, lda, Wt, ldw, x, E, gamma, mean, rstd, dres, E, dx, E, (float*)ws,
                                             &npanels, M, E, K, st);
  if (rc != RGBNM_OK) return false;
  RgbnmReduceJob j;
  j.part = (const float*)ws; j.stride = 2LL * E; j.out = dgamma; j.n = E; j.S = npanels; j.cols = 1; j.perm_heads = 0;
  j.accumulate = 0; j.epw = 8;
  if (rgbnm_reduce_submit(j, st) != RGBNM_OK) return false;
  j.part = (const float*)ws + E; j.out = dbeta;
  return rgbnm_reduce_submit(j, st) == RGBNM_OK;
}

// du = (dy . W2) * gelu'(u), dx = dy + LayerNorm'(du . W1) in one launch (mlp_fused.hip) + two jobs for the batched reduction.
// false: not eligible (fp32, E != 192, small M, option mlp_bwd / ln_fuse = 0) -> the caller runs the separate launches.
static bool fused_mlp_bwd(int dt, const void* dy, const void* w2t, const void* w1t, const void* gp, void* du, const void* x,
                          const float* gamma, const float* mean, const float* rstd, void* dx, float* dgamma, float* dbeta,
                          int M, int E, void* ws, size_t ws_bytes, hipStream_t st) {
  if (dt != RGBNM_DT_BF16 || E != 192 || !rgbnm_get_option("mlp_bwd") || !rgbnm_get_option("ln_fuse")) return false;
  if (ws_bytes < (size_t)cdiv(M, cdiv(M, 256)) * 2 * E * sizeof(float)) return false;
  int npanels = 0;
  const int rc = rgbnm_launch_mlp_bwd(dy, E, w2t, w1t, gp, 4 * E, du, 4 * E, x, E, gamma, mean, rstd, dx, E, (float*)ws,
                                      &npanels, M, E, 4 * E, st);
  if (rc != RGBNM_OK) return false;
  RgbnmReduceJob j;
  j.part = (const float*)ws; j.stride = 2LL * E; j.out = dgamma; j.n = E; j.S = npanels; j.cols = 1; j.perm_heads = 0;
  j.accumulate = 0; j.epw = 8;
  if (rgbnm_reduce_submit(j, st) != RGBNM_OK) return false;
  j.part = (const float*)ws + E; j.out = dbeta;
  return rgbnm_reduce_submit(j, st) == RGBNM_OK;
}

// workspace regions of one block backward: fc2 dW | fc1 dW | proj dW | qkv dW | LN2 | LN1 ; returns the total
static size_t block_ws_offsets(int M, int E, int I, size_t (&off)[7]) {
  const size_t sz[6] = {rgbnm_gemm_tn_workspace(M, E, 4 * E), rgbnm_gemm_tn_workspace(M, 4 * E, E),
                        rgbnm_gemm_tn_workspace(M, E, I),     rgbnm_gemm_tn_workspace(M, 3 * I, E),
                        rgbnm_layernorm_bwd_workspace(M, E),  rgbnm_layernorm_bwd_workspace(M, E)};
  off[0] = 0;
  for (int i = 0; i < 6; ++i) off[i + 1] = off[i] + ((sz[i] + 255) & ~(size_t)255);
  return off[6];
}

size_t rgbnm_vit_workspace(const rgbnm_vit_cfg* c) { return rgbnm_vit_workspace_ex(c, 1024); }

size_t rgbnm_vit_workspace_ex(const rgbnm_vit_cfg* c, int n_classes) {
  if (!c) return 0;
  if (n_classes < 1024) n_classes = 1024;
  const int M = c->B * c->N, E = c->E, I = c->heads * 64;
  // a block's backward keeps the partials of its four weight-gradient GEMMs and two LayerNorms side by side until
  // the one batched reduction at its end (block_ws_offsets below)
  size_t off[7];
  size_t w = block_ws_offsets(M, E, I, off);
  w = max_sz(w, rgbnm_gemm_tn_workspace(M, E, 384));
  w = max_sz(w, rgbnm_gemm_tn_workspace(c->B, n_classes, E));
  w = max_sz(w, rgbnm_layernorm_bwd_workspace(M, E));
  w = max_sz(w, (size_t)c->B * 2 * E * sizeof(float));
  return w;
}

int rgbnm_vit_ln_chain(const rgbnm_vit_cfg* c) {
  if (!c) return 0;
  return c->dtype == RGBNM_DT_BF16 && c->E == 192 && c->B * c->N >= 8192 && rgbnm_get_option("ln_fuse") &&
         rgbnm_get_option("nt_kpipe");
}

int rgbnm_vit_block_fwd_chain(const rgbnm_vit_cfg* c, const rgbnm_block_params* p, const rgbnm_block_acts* a, int flags,
                              const rgbnm_block_params* next_p, const rgbnm_block_acts* next_a, void* st) {
  if (!c || !p || !a) return RGBNM_EINVAL;
  const int dt = c->dtype, M = c->B * c->N, E = c->E, I = c->heads * 64;
  const bool chain = rgbnm_vit_ln_chain(c) != 0;
  if ((flags & 1) && !chain) return RGBNM_EINVAL;
  if (!(flags & 1))
    TRY(rgbnm_layernorm_fwd(dt, a->x_in, p->ln1_g, p->ln1_b, a->xn1, a->mean1, a->rstd1, M, E, c->ln_eps, st));
  TRY(rgbnm_gemm_nt(dt, RGBNM_EPI_NONE, a->xn1, E, p->wqkv, E, a->qkv, 3 * I, p->bqkv_perm, 0, 0, 0, 0, 0, 0, M,
                    3 * I, E, 0, st));
  TRY(rgbnm_attention_fwd(dt, a->qkv, a->attn, a->lse, c->B, c->N, c->heads, c->attn_scale, st));
  // x_mid = x_in + proj(attn) ; xn2 = LN2(x_mid): one launch when chaining is on
  if (!chain || rgbnm_launch_nt_kpipe_res_ln(a->attn, I, p->wproj, I, p->bproj, a->x_in, E, a->x_mid, E, p->ln2_g,
                                             p->ln2_b, a->xn2, E, a->mean2, a->rstd2, c->ln_eps, M, E, I,
                                             (hipStream_t)st) != RGBNM_OK) {
    TRY(rgbnm_gemm_nt(dt, RGBNM_EPI_RES, a->attn, I, p->wproj, I, a->x_mid, E, p->bproj, a->x_in, E, 0, 0, 0, 0, M, E,
                      I, 0, st));
    TRY(rgbnm_layernorm_fwd(dt, a->x_mid, p->ln2_g, p->ln2_b, a->xn2, a->mean2, a->rstd2, M, E, c->ln_eps, st));
  }
  const bool to_next = chain && next_p && next_a;
  if (to_next && next_a->x_in != a->x_out) return RGBNM_EINVAL;
  // the whole FeedForwardBlock (+ residual [+ the next block's LN1]) in one launch when eligible (mlp_fused.hip)
  if (dt == RGBNM_DT_BF16 && rgbnm_get_option("mlp_fuse")) {
    const int rc = rgbnm_launch_mlp_fwd(a->xn2, E, p->w1, p->b1, p->w2, p->b2, a->x_mid, E, a->gl, a->u, 4 * E, a->x_out, E,
                                        to_next ? next_p->ln1_g : nullptr, to_next ? next_p->ln1_b : nullptr,
                                        to_next ? next_a->xn1 : nullptr, E, to_next ? next_a->mean1 : nullptr,
                                        to_next ? next_a->rstd1 : nullptr, c->ln_eps, M, E, 4 * E, (hipStream_t)st);
    if (rc <= 0) return rc;                                  // done (0) or a real error (< 0); 1 = not eligible
  }
  TRY(rgbnm_gemm_nt(dt, RGBNM_EPI_GELU, a->xn2, E, p->w1, E, a->gl, 4 * E, p->b1, 0, 0, a->u, 4 * E, 0, 0, M, 4 * E,
                    E, 0, st));
  // x_out = x_mid + fc2(gl) [; next block's xn1 = LN1(x_out)]
  if (to_next) {
    const int rc = rgbnm_launch_nt_kpipe_res_ln(a->gl, 4 * E, p->w2, 4 * E, p->b2, a->x_mid, E, a->x_out, E,
                                                next_p->ln1_g, next_p->ln1_b, next_a->xn1, E, next_a->mean1,
                                                next_a->rstd1, c->ln_eps, M, E, 4 * E, (hipStream_t)st);
    if (rc != RGBNM_OK) return rc < 0 ? rc : RGBNM_EINVAL;    // rgbnm_vit_ln_chain() promised eligibility
    return RGBNM_OK;
  }
  TRY(rgbnm_gemm_nt(dt, RGBNM_EPI_RES, a->gl, 4 * E, p->w2, 4 * E, a->x_out, E, p->b2, a->x_mid, E, 0, 0, 0, 0, M, E,
                    4 * E, 0, st));
  return RGBNM_OK;
}

int rgbnm_vit_block_fwd(const rgbnm_vit_cfg* c, const rgbnm_block_params* p, const rgbnm_block_acts* a, void* st) {
  return rgbnm_vit_block_fwd_chain(c, p, a, 0, nullptr, nullptr, st);
}

int rgbnm_vit_block_bwd(const rgbnm_vit_cfg* c, const rgbnm_block_params* p, const rgbnm_block_acts* a,
                        const rgbnm_block_grads* g, const rgbnm_block_scratch* s, const void* dy, void* dx, void* st) {
  if (!c || !p || !a || !g || !s || !dy || !dx) return RGBNM_EINVAL;
  const int dt = c->dtype, M = c->B * c->N, E = c->E, I = c->heads * 64;
  size_t off[7];
  if (s->ws_bytes < block_ws_offsets(M, E, I, off)) return RGBNM_EWORKSPACE;
  char* wsb = (char*)s->ws;
#define WS(i) (wsb + off[i]), (off[(i) + 1] - off[i])
  rgbnm_reduce_defer_begin();     // the 12 partial reductions of this block run as one launch at the end
  bool tn_open = false;           // a grouping opened below and not yet closed (only an error leaves it so)
  const int rc = [&]() -> int {
  // ---- MLP branch: x_out = x_mid + fc2(gelu(fc1(LN2(x_mid)))) ------------------------------------
  // tn_group: the four dW GEMMs run one per launch (0), as pairs fc2 + fc1 / proj + qkv (1) or all in one launch at the
  // end of the block (2): T tiles in a launch -> 256 / T splits of the token axis -> that many fp32 partial tiles
  // E = 192: all four in one launch measured best (21 tiles, 12 token splits).  E = 384: 72 tiles leave 3 splits on 216 of the 256
  // CUs; as pairs (48 tiles x 5 splits, 24 x 10: 240 CUs each, the first pair launched right behind the MLP data path whose
  // outputs it reads) the JPEG-S step is 1 % faster (13.26 -> 13.13 ms, interleaved)
  int group = rgbnm_get_option("tn_group");
  // (192 x 384 tiles, option tn_wide: a block's four GEMMs are 24 tiles -- one launch, 10 token splits, 240 CUs)
  const bool wide4 = rgbnm_get_option("tn_wide") && E % 384 == 0 && I % 384 == 0 && M % 64 == 0;
  if (group == 2 && E > 192 && !wide4) group = 1;
  if (group) { rgbnm_tn_defer_begin(); tn_open = true; }
  TRY(rgbnm_gemm_tn(dt, dy, E, a->gl, 4 * E, g->dw2, g->db2, M, E, 4 * E, 0, 0, WS(0), st));
  // du = (dy . W2) * gelu'(u) and dx_mid = dy + LN2'(du . W1) in ONE launch when eligible (mlp_fused.hip, option mlp_bwd)
  const bool mlp_bwd_fused = fused_mlp_bwd(dt, dy, p->w2_t, p->w1_t, a->u, s->du, a->x_mid, p->ln2_g, a->mean2, a->rstd2,
                                           s->dx_mid, g->dln2_g, g->dln2_b, M, E, WS(4), (hipStream_t)st);
  if (!mlp_bwd_fused)
    TRY(rgbnm_gemm_nt(dt, RGBNM_EPI_DGELU, dy, E, p->w2_t, E, s->du, 4 * E, 0, a->u, 4 * E, 0, 0, 0, 0, M, 4 * E, E, 0,
                      st));
  TRY(rgbnm_gemm_tn(dt, s->du, 4 * E, a->xn2, E, g->dw1, g->db1, M, 4 * E, E, 0, 0, WS(1), st));
  if (group == 1) { tn_open = false; TRY(rgbnm_tn_defer_flush((hipStream_t)st)); }
  // dx_mid = dy + LN2'(du . W1): LayerNorm backward fused into the GEMM epilogue when eligible
  if (mlp_bwd_fused) {
  } else if (!fused_dx_lnbwd(dt, s->du, 4 * E, p->w1_t, 4 * E, a->x_mid, p->ln2_g, a->mean2, a->rstd2, dy, s->dx_mid,
                      g->dln2_g, g->dln2_b, M, E, 4 * E, WS(4), (hipStream_t)st)) {
    TRY(rgbnm_gemm_nt(dt, RGBNM_EPI_NONE, s->du, 4 * E, p->w1_t, 4 * E, s->dxn, E, 0, 0, 0, 0, 0, 0, 0, M, E, 4 * E,
                      0, st));
    TRY(rgbnm_layernorm_bwd(dt, s->dxn, a->x_mid, p->ln2_g, a->mean2, a->rstd2, dy, s->dx_mid, g->dln2_g, g->dln2_b,
                            M, E, 0, WS(4), st));
  }
  // ---- attention branch: x_mid = x_in + proj(attn(qkv(LN1(x_in)))) -------------------------------
  if (group == 1) { rgbnm_tn_defer_begin(); tn_open = true; }
  TRY(rgbnm_gemm_tn(dt, s->dx_mid, E, a->attn, I, g->dwproj, g->dbproj, M, E, I, 0, 0, WS(2), st));
  TRY(rgbnm_gemm_nt(dt, RGBNM_EPI_NONE, s->dx_mid, E, p->wproj_t, E, s->dattn, I, 0, 0, 0, 0, 0, 0, 0, M, I, E, 0, st));
  TRY(rgbnm_attention_bwd(dt, a->qkv, a->attn, s->dattn, a->lse, s->dqkv, c->B, c->N, c->heads, c->attn_scale, st));
  TRY(rgbnm_gemm_tn(dt, s->dqkv, 3 * I, a->xn1, E, g->dwqkv, g->dbqkv, M, 3 * I, E, c->heads, 0, WS(3), st));
  if (group) { tn_open = false; TRY(rgbnm_tn_defer_flush((hipStream_t)st)); }
  if (!fused_dx_lnbwd(dt, s->dqkv, 3 * I, p->wqkv_t, 3 * I, a->x_in, p->ln1_g, a->mean1, a->rstd1, s->dx_mid, dx,
                      g->dln1_g, g->dln1_b, M, E, 3 * I, WS(5), (hipStream_t)st)) {
    TRY(rgbnm_gemm_nt(dt, RGBNM_EPI_NONE, s->dqkv, 3 * I, p->wqkv_t, 3 * I, s->dxn, E, 0, 0, 0, 0, 0, 0, 0, M, E,
                      3 * I, 0, st));
    TRY(rgbnm_layernorm_bwd(dt, s->dxn, a->x_in, p->ln1_g, a->mean1, a->rstd1, s->dx_mid, dx, g->dln1_g, g->dln1_b, M,
                            E, 0, WS(5), st));
  }
  return RGBNM_OK;
  }();
#undef WS
  const int rt = tn_open ? rgbnm_tn_defer_flush((hipStream_t)st) : RGBNM_OK;      // only an error leaves the grouping open
  const int rf = rgbnm_reduce_defer_flush((hipStream_t)st);
  return rc != RGBNM_OK ? rc : (rt != RGBNM_OK ? rt : rf);
}

// The weight / bias / LayerNorm-parameter gradients of n blocks after rgbnm_vit_chain_bwd (vit_chain_bwd.hip) has run their data
// path: ALL their weight-gradient GEMMs as one grouped launch (4 n jobs, 21 n tiles: the more blocks, the fewer token splits
// -- twelve blocks: 252 tiles, no split) and the batched reduction.  Every block brings its own workspace (scratch->ws).
int rgbnm_vit_blocks_bwd_dw(const rgbnm_vit_cfg* c, int n, const rgbnm_block_acts* const* a, const rgbnm_block_grads* const* g,
                            const rgbnm_block_scratch* const* s, const void* const* dy, const float* const* part2,
                            const float* const* part1, void* st) {
  return rgbnm_vit_blocks_bwd_dw_pe(c, n, a, g, s, dy, part2, part1, nullptr, nullptr, nullptr, nullptr, nullptr, 0, st);
}

// ... with the patch embedding's weight gradient (rgbnm_patch_embed_bwd: dW = dx0^T . feat, the same token axis) as one more job of
// the same grouped launch: 252 + 4 = 256 tiles at B = 256, still no token split -- its own launch (21 us), its partial sums and
// their share of the reduction are gone.  pe_dx0 == NULL: blocks only.
int rgbnm_vit_blocks_bwd_dw_pe(const rgbnm_vit_cfg* c, int n, const rgbnm_block_acts* const* a, const rgbnm_block_grads* const* g,
                               const rgbnm_block_scratch* const* s, const void* const* dy, const float* const* part2,
                               const float* const* part1, const void* pe_dx0, const void* pe_feat, float* pe_dw, float* pe_db,
                               void* pe_ws, size_t pe_ws_bytes, void* st) {
  if (!c || n < 1 || n > 12 || !a || !g || !s || !dy || !part2 || !part1) return RGBNM_EINVAL;
  if (pe_dx0 && (!pe_feat || !pe_dw || !pe_ws)) return RGBNM_EINVAL;
  const int dt = c->dtype, M = c->B * c->N, E = c->E, I = c->heads * 64;
  size_t off[7];
  const size_t need = block_ws_offsets(M, E, I, off);
  for (int i = 0; i < n; ++i) {
    if (!a[i] || !g[i] || !s[i] || !dy[i] || !part2[i] || !part1[i]) return RGBNM_EINVAL;
    if (s[i]->ws_bytes < need) return RGBNM_EWORKSPACE;
    for (int k = 0; k < i; ++k)
      if (s[k]->ws == s[i]->ws) return RGBNM_EINVAL;          // the partial sums of the grouped blocks live side by side
  }
  rgbnm_reduce_defer_begin();
  bool tn_open = false;
  const int rc = [&]() -> int {
    const int group = rgbnm_get_option("tn_group");
    if (group) { rgbnm_tn_defer_begin_n(group == 1 ? 2 : 4 * n + (pe_dx0 ? 1 : 0)); tn_open = true; }
    for (int i = 0; i < n; ++i) {
      char* wsb = (char*)s[i]->ws;
#define WS(k) (wsb + off[k]), (off[(k) + 1] - off[k])
      TRY(rgbnm_gemm_tn(dt, dy[i], E, a[i]->gl, 4 * E, g[i]->dw2, g[i]->db2, M, E, 4 * E, 0, 0, WS(0), st));
      TRY(rgbnm_gemm_tn(dt, s[i]->du, 4 * E, a[i]->xn2, E, g[i]->dw1, g[i]->db1, M, 4 * E, E, 0, 0, WS(1), st));
      TRY(rgbnm_gemm_tn(dt, s[i]->dx_mid, E, a[i]->attn, I, g[i]->dwproj, g[i]->dbproj, M, E, I, 0, 0, WS(2), st));
      TRY(rgbnm_gemm_tn(dt, s[i]->dqkv, 3 * I, a[i]->xn1, E, g[i]->dwqkv, g[i]->dbqkv, M, 3 * I, E, c->heads, 0, WS(3), st));
#undef WS
    }
    if (pe_dx0) TRY(rgbnm_gemm_tn(dt, pe_dx0, E, pe_feat, 384, pe_dw, pe_db, M, E, 384, 0, 0, pe_ws, pe_ws_bytes, st));
    if (group) { tn_open = false; TRY(rgbnm_tn_defer_flush((hipStream_t)st)); }
    for (int i = 0; i < n; ++i) {       // per-image partial sums of the LayerNorm parameter gradients (one panel per image)
      RgbnmReduceJob j;
      j.stride = 2LL * E; j.n = E; j.S = c->B; j.cols = 1; j.perm_heads = 0; j.accumulate = 0; j.epw = 8;
      j.part = part2[i]; j.out = g[i]->dln2_g;
      TRY(rgbnm_reduce_submit(j, (hipStream_t)st));
      j.part = part2[i] + E; j.out = g[i]->dln2_b;
      TRY(rgbnm_reduce_submit(j, (hipStream_t)st));
      j.part = part1[i]; j.out = g[i]->dln1_g;
      TRY(rgbnm_reduce_submit(j, (hipStream_t)st));
      j.part = part1[i] + E; j.out = g[i]->dln1_b;
      TRY(rgbnm_reduce_submit(j, (hipStream_t)st));
    }
    return RGBNM_OK;
  }();
  const int rt = tn_open ? rgbnm_tn_defer_flush((hipStream_t)st) : RGBNM_OK;
  const int rf = rgbnm_reduce_defer_flush((hipStream_t)st);
  return rc != RGBNM_OK ? rc : (rt != RGBNM_OK ? rt : rf);
}

int rgbnm_vit_block_bwd_dw(const rgbnm_vit_cfg* c, const rgbnm_block_acts* a, const rgbnm_block_grads* g,
                           const rgbnm_block_scratch* s, const void* dy, const float* part2, const float* part1, void* st) {
  return rgbnm_vit_blocks_bwd_dw(c, 1, &a, &g, &s, &dy, &part2, &part1, st);
}

int rgbnm_patch_embed_fwd_mix(const rgbnm_vit_cfg* c, int in_dtype, const void* y, const void* cbcr, const float* lam_dev,
                              const float* conv16, const void* wpe, const float* bpe, const float* pos, void* feat, void* x0, int Hb,
                              int Wb, void* st) {
  if (!c) return RGBNM_EINVAL;
  const int M = c->B * c->N;
  if ((Hb / 2) * (Wb / 2) != c->N) return RGBNM_EINVAL;
  TRY(rgbnm_subblock_embed_mix(in_dtype, c->dtype, y, cbcr, lam_dev, conv16, feat, c->B, Hb, Wb, 0, st));
  TRY(rgbnm_gemm_nt(c->dtype, RGBNM_EPI_POS, feat, 384, wpe, 384, x0, c->E, bpe, 0, 0, 0, 0, pos, c->N, M, c->E, 384,
                    0, st));
  return RGBNM_OK;
}

int rgbnm_patch_embed_fwd(const rgbnm_vit_cfg* c, int in_dtype, const void* y, const void* cbcr, const float* conv16,
                          const void* wpe, const float* bpe, const float* pos, void* feat, void* x0, int Hb, int Wb,
                          void* st) {
  return rgbnm_patch_embed_fwd_mix(c, in_dtype, y, cbcr, nullptr, conv16, wpe, bpe, pos, feat, x0, Hb, Wb, st);
}

int rgbnm_patch_embed_bwd(const rgbnm_vit_cfg* c, const void* dx0, const void* feat, float* dwpe, float* dbpe,
                          void* ws, size_t ws_bytes, void* st) {
  if (!c) return RGBNM_EINVAL;
  return rgbnm_gemm_tn(c->dtype, dx0, c->E, feat, 384, dwpe, dbpe, c->B * c->N, c->E, 384, 0, 0, ws, ws_bytes, st);
}

int rgbnm_head_fwd(const rgbnm_vit_cfg* c, const rgbnm_head_params* p, const rgbnm_head_acts* a, void* st) {
  if (!c || !p || !a) return RGBNM_EINVAL;
  const int dt = c->dtype, E = c->E, C = p->n_classes;
  TRY(rgbnm_head_pool_fwd(dt, a->x, p->ln_g, p->ln_b, a->pooled, a->mean, a->rstd, c->B, c->N, E, c->ln_eps, st));
  TRY(rgbnm_gemm_nt(dt, RGBNM_EPI_TANH, a->pooled, E, p->w1, E, a->h1, E, p->b1, 0, 0, 0, 0, 0, 0, c->B, E, E, 0, st));
  TRY(rgbnm_gemm_nt(dt, RGBNM_EPI_NONE, a->h1, E, p->w2, E, a->logits, C, p->b2, 0, 0, 0, 0, 0, 0, c->B, C, E, 1, st));
  return RGBNM_OK;
}

size_t rgbnm_head_bwd_workspace(const rgbnm_vit_cfg* c, int n_classes) {
  if (!c) return 0;
  return ((rgbnm_gemm_tn_workspace(c->B, n_classes, c->E) + 255) & ~(size_t)255) + ((rgbnm_gemm_tn_workspace(c->B, c->E, c->E) + 255) & ~(size_t)255) +
         (size_t)c->B * 2 * c->E * sizeof(float);
}

int rgbnm_head_bwd(const rgbnm_vit_cfg* c, const rgbnm_head_params* p, const rgbnm_head_acts* a,
                   const rgbnm_head_grads* g, const void* dlogits, void* da, void* dpooled, void* dx, void* ws,
                   size_t ws_bytes, void* st) {
  if (!c || !p || !a || !g || !dlogits || !da || !dpooled || !dx) return RGBNM_EINVAL;
  const int dt = c->dtype, E = c->E, C = p->n_classes, B = c->B;
  // three split-sum producers: side by side when the workspace has room for all of them (rgbnm_head_bwd_workspace: required
  // inside a held-reduction bracket, where the partial sums live until rgbnm_reduce_hold_end), else one after the other in place
  const size_t s1 = (rgbnm_gemm_tn_workspace(B, C, E) + 255) & ~(size_t)255, s2 = (rgbnm_gemm_tn_workspace(B, E, E) + 255) & ~(size_t)255;
  const size_t s3 = (size_t)B * 2 * E * sizeof(float);
  const bool apart = ws_bytes >= s1 + s2 + s3;
  char* w1 = (char*)ws;
  char* w2 = apart ? w1 + s1 : w1;
  char* w3 = apart ? w2 + s2 : w1;
  const size_t b1 = apart ? s1 : ws_bytes, b2 = apart ? s2 : ws_bytes, b3 = apart ? ws_bytes - s1 - s2 : ws_bytes;
  // both weight-gradient GEMMs (same row count) as ONE grouped launch behind the dtanh product that makes the second one's operand
  // -- when their partial sums have regions of their own (a shared region would be overwritten by the second job)
  const bool grouped = apart && rgbnm_get_option("tn_group") != 0;
  if (grouped) rgbnm_tn_defer_begin();
  const int rc = [&]() -> int {
    TRY(rgbnm_gemm_tn(dt, dlogits, C, a->h1, E, g->dw2, g->db2, B, C, E, 0, 0, w1, b1, st));
    TRY(rgbnm_gemm_nt(dt, RGBNM_EPI_DTANH, dlogits, C, p->w2_t, C, da, E, 0, a->h1, E, 0, 0, 0, 0, B, E, C, 0, st));
    TRY(rgbnm_gemm_tn(dt, da, E, a->pooled, E, g->dw1, g->db1, B, E, E, 0, 0, w2, b2, st));
    return RGBNM_OK;
  }();
  const int rt = grouped ? rgbnm_tn_defer_flush((hipStream_t)st) : RGBNM_OK;
  if (rc != RGBNM_OK) return rc;
  if (rt != RGBNM_OK) return rt;
  TRY(rgbnm_gemm_nt(dt, RGBNM_EPI_NONE, da, E, p->w1_t, E, dpooled, E, 0, 0, 0, 0, 0, 0, 0, B, E, E, 0, st));
  TRY(rgbnm_head_pool_bwd(dt, dpooled, a->x, p->ln_g, a->mean, a->rstd, dx, g->dln_g, g->dln_b, B, c->N, E, 0, w3, b3, st));
  return RGBNM_OK;
}

}  // extern "C"
