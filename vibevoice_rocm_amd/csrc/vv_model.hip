// vv_model.hip — composite operators: each call enqueues the full launch sequence of one component of the
// per-frame loop (LLM step, diffusion-head sampling, acoustic decoder, semantic/acoustic encoder, connectors)
// on the caller's stream.  No host synchronisation, no allocation: capturable into one hipGraph per frame.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>

#include "vv_hip.h"
#include "vv_common.h"

namespace {

struct Carver {   // carve 256-byte aligned float buffers out of a caller-provided workspace
  char* p;
  explicit Carver(void* base) : p(reinterpret_cast<char*>(base)) {}
  float* take(size_t n_floats) {
    float* r = reinterpret_cast<float*>(p);
    p += (n_floats * sizeof(float) + 255) & ~(size_t)255;
    return r;
  }
};
inline size_t al(size_t n_floats) { return (n_floats * sizeof(float) + 255) & ~(size_t)255; }

inline vv_lin_args lin_base(const float* x, int64_t ldx, int m, const void* w, int n, int k, int wdt, float* out, int64_t ldo) {
  vv_lin_args a;
  memset(&a, 0, sizeof(a));
  a.x = x; a.ldx = ldx; a.m = m; a.w = w; a.n = n; a.k = k; a.wdt = wdt; a.out = out; a.ldo = ldo;
  return a;
}

}  // namespace

// ---------------------------------------------------------------------------------------------------------------
// Qwen2 decoder stack
// ---------------------------------------------------------------------------------------------------------------
// weight-only fp8 companions (vv_w8) replace the bf16 matrix on the weight-streaming GEMVs (<= 2 activation rows)
static inline void use_w8(vv_lin_args& a, const vv_w8& q, const vv_w8* q2 = nullptr) {
  if (a.m > 2 || !q.q || !q.scale || (a.w2 && (!q2 || !q2->q || !q2->scale))) return;
  a.w = q.q; a.wscale = q.scale; a.wdt = VV_FP8;
  if (a.w2) { a.w2 = q2->q; a.w2scale = q2->scale; }
}

// from this many rows on, a bf16-weight LLM forward is a prompt prefill: activations are cast to bf16 once per GEMM and both
// operands stream from global on the matrix cores with 128-row tiles (every weight fragment reused by 4 row tiles)
#define VV_PREFILL_ROWS 64
// decode attention over long contexts splits the keys over up to this many blocks per (row, q head) (vv_attn_decode.hip)
#define VV_ATT_ROWS 8
#define VV_ATT_MAX_SPLIT 16        // per-head kernel
#define VV_ATT_PART_SPLITS 128     // capacity of the partials workspace: the grouped kernel spreads long contexts over up to 128 splits per (row, KV head)

// split-K workspace of the row-batched decode step (vv_gemv_rows.hip): the widest non-dual matrix is the one that splits K
static size_t rows_part_floats(const vv_llm* m) {
  const int qkvd = (m->heads + 2 * m->kv_heads) * m->head_dim;
  const int n = qkvd > m->hidden ? qkvd : m->hidden;
  const size_t a = vv_gemv_rows_part_floats(n, 0), b = vv_gemv_rows_part_floats(m->inter, 1) / 8;     // the dual kernel splits K at most two ways
  return a > b ? a : b;
}
static size_t rows_tickets(const vv_llm* m) {
  const int qkvd = (m->heads + 2 * m->kv_heads) * m->head_dim;
  const int n = qkvd > m->inter ? qkvd : m->inter;
  return vv_gemv_rows_tickets(n > m->hidden ? n : m->hidden);
}

extern "C" size_t vv_llm_ws_bytes(const vv_llm* m, int R) {
  if (!m || R <= 0) return 0;
  const size_t qkv = (size_t)(m->heads + 2 * m->kv_heads) * m->head_dim;
  const size_t widest = (size_t)(m->inter > m->hidden ? m->inter : m->hidden);
  return al((size_t)R * m->hidden) + al((size_t)R * qkv) + al((size_t)R * m->heads * m->head_dim) + al((size_t)R * m->inter) +
         al((size_t)R * m->head_dim) + (R >= VV_PREFILL_ROWS ? al((size_t)R * widest / 2 + 64) : 0) +
         al((size_t)VV_ATT_ROWS * m->heads * VV_ATT_PART_SPLITS * (m->head_dim + 2)) + al((size_t)VV_ATT_ROWS * m->heads) +   // split-key decode attention
         ((R > 2 && R <= 8) ? al(rows_part_floats(m)) + al(rows_tickets(m)) : 0);                                            // split-K partials of the 3..8-row GEMV
}

extern "C" int vv_llm_forward(const vv_llm* m, const vv_kv* kv, const float* x, int64_t ldx, int R, const int* lens,
                              const int* cache_rows, float* out, int64_t ldo, void* ws, vv_stream_t stream) {
  if (!m || !kv || !x || !lens || !ws || !m->layer) return vv_set_error(VV_E_ARG, "vv_llm_forward: null pointer");
  if (R <= 0) return vv_set_error(VV_E_ARG, "vv_llm_forward: R=%d", R);
  if (kv->layers != m->layers || kv->kv_heads != m->kv_heads || kv->head_dim != m->head_dim)
    return vv_set_error(VV_E_ARG, "vv_llm_forward: kv cache shape does not match the model");
  hipStream_t s = (hipStream_t)stream;
  const int H = m->hidden, d = m->head_dim, qd = m->heads * d, qkvd = (m->heads + 2 * m->kv_heads) * d;
  Carver c(ws);
  float* h = c.take((size_t)R * H);
  float* qkv = c.take((size_t)R * qkvd);
  float* att = c.take((size_t)R * qd);
  float* act = c.take((size_t)R * m->inter);
  float* rope = c.take((size_t)R * d);
  // layer 0 reads the input embeddings where they lie (prologue and residual operand): no copy into the workspace
  const float* hin = x;
  int64_t ldh = ldx;
  VV_TRY(vv_rope_table(lens, m->inv_freq, R, d, rope, stream));
  const bool prefill = (R >= VV_PREFILL_ROWS) && m->wdt == VV_BF16 && H % 16 == 0 && m->inter % 16 == 0 && qd % 16 == 0;
  void* xb = prefill ? (void*)c.take((size_t)R * (m->inter > H ? m->inter : H) / 2 + 64) : nullptr;
  float* att_part = c.take((size_t)VV_ATT_ROWS * m->heads * VV_ATT_PART_SPLITS * (d + 2));
  int* att_tickets = reinterpret_cast<int*>(c.take((size_t)VV_ATT_ROWS * m->heads));
  // contexts beyond ~1K keys: one block per (row, q head) would pull all of its K / V through a single CU (~95 GB/s: 8 us at S = 3000)
  const bool decode = (cache_rows == nullptr && R <= kv->rows);
  int att_split = 1;
  if (decode && R <= VV_ATT_ROWS && kv->s_max > 1024) {
    att_split = (kv->s_max + 511) / 512;
    if (att_split > VV_ATT_MAX_SPLIT) att_split = VV_ATT_MAX_SPLIT;
    hipError_t e = hipMemsetAsync(att_tickets, 0, (size_t)VV_ATT_ROWS * m->heads * sizeof(int), s);
    if (e != hipSuccess) return vv_set_error(VV_E_HIP, "vv_llm_forward: %s", hipGetErrorString(e));
  }
  void* xb2 = prefill ? (void*)act : nullptr;      // bf16 SwiGLU output [R, inter] lives in the (otherwise unused) fp32 act buffer
  // 3..8 rows (dialogues batched into the row dimension): the matrix-core GEMV with its split-K workspace, fragment-major weights when the model has them
  const bool rows8 = decode && R > 2 && R <= 8 && m->wdt == VV_BF16;
  float* rpart = nullptr;
  int* rtick = nullptr;
  size_t rpart_n = 0, rtick_n = 0;
  if (rows8) {
    rpart_n = rows_part_floats(m); rtick_n = rows_tickets(m);
    rpart = c.take(rpart_n);
    rtick = reinterpret_cast<int*>(c.take(rtick_n));
    hipError_t e = hipMemsetAsync(rtick, 0, rtick_n * sizeof(int), s);
    if (e != hipSuccess) return vv_set_error(VV_E_HIP, "vv_llm_forward: %s", hipGetErrorString(e));
  }
  if (rows8) {      // the residual stream lives in h from the start: the split-K projections then accumulate into it in place (vv_gemv_rows.hip)
    VV_TRY(vv_copy_rows(x, ldx, h, H, R, H, stream));
    hin = h; ldh = H;
  }
  auto lin = [&](vv_lin_args& a, const void* f1, const void* f2) -> int {
    if (!rows8) return vv_linear(&a, stream);
    return vv_linear_ws(&a, f1, f2, rpart, rpart_n, rtick, rtick_n, stream);
  };
  for (int l = 0; l < m->layers; ++l) {
    const vv_llm_layer& L = m->layer[l];
    vv_lin_args a;
    if (prefill) {
      VV_TRY(vv_cast_rows_bf16(hin, ldh, R, H, VV_PRO_RMSNORM, L.ln1, m->rms_eps, xb, H, stream));
      a = lin_base((const float*)xb, H, R, L.wqkv, qkvd, H, m->wdt, qkv, qkvd);
      a.flags = VV_LIN_X_BF16; a.bias = L.bqkv;
    } else {
      a = lin_base(hin, ldh, R, L.wqkv, qkvd, H, m->wdt, qkv, qkvd);
      a.pro = VV_PRO_RMSNORM; a.norm_w = L.ln1; a.eps = m->rms_eps; a.bias = L.bqkv;
      if (!rows8) use_w8(a, L.q_qkv);
    }
    VV_TRY(lin(a, L.f_qkv, nullptr));
    if (decode) {
      VV_TRY(vv_attn_decode_ws(qkv, qkvd, R, m->heads, kv, l, rope, lens, att, qd, att_part, att_tickets, att_split, VV_ATT_PART_SPLITS, stream));   // decode: RoPE + append fused
    } else {
      VV_TRY(vv_rope_store(qkv, qkvd, R, m->heads, kv, l, rope, lens, cache_rows, stream));
      VV_TRY(vv_attn(qkv, qkvd, R, m->heads, kv, l, lens, cache_rows, att, qd, stream));
    }
    if (prefill) {
      VV_TRY(vv_cast_rows_bf16(att, qd, R, qd, VV_PRO_NONE, nullptr, 0.f, xb, qd, stream));
      a = lin_base((const float*)xb, qd, R, L.wo, H, qd, m->wdt, h, H);
      a.flags = VV_LIN_X_BF16;
    } else {
      a = lin_base(att, qd, R, L.wo, H, qd, m->wdt, h, H);
      if (!rows8) use_w8(a, L.q_o);
    }
    a.res = hin; a.ldres = ldh;
    VV_TRY(lin(a, L.f_o, nullptr));
    hin = h; ldh = H;
    if (prefill) {
      VV_TRY(vv_cast_rows_bf16(h, H, R, H, VV_PRO_RMSNORM, L.ln2, m->rms_eps, xb, H, stream));
      a = lin_base((const float*)xb, H, R, L.wgate, m->inter, H, m->wdt, act, m->inter);
      a.flags = VV_LIN_X_BF16;
    } else {
      a = lin_base(h, H, R, L.wgate, m->inter, H, m->wdt, act, m->inter);
      a.pro = VV_PRO_RMSNORM; a.norm_w = L.ln2; a.eps = m->rms_eps;
    }
    a.w2 = L.wup; a.act = VV_ACT_SWIGLU;
    if (!prefill && !rows8) use_w8(a, L.q_gate, &L.q_up);
    if (prefill) {            // the SwiGLU output is handed to the down projection in bf16: no separate cast pass
      a.out = reinterpret_cast<float*>(xb2); a.ldo = m->inter; a.flags |= VV_LIN_OUT_BF16;
    }
    VV_TRY(lin(a, L.f_gate, L.f_up));
    if (prefill) {
      a = lin_base((const float*)xb2, m->inter, R, L.wdown, H, m->inter, m->wdt, h, H);
      a.flags = VV_LIN_X_BF16;
    } else {
      a = lin_base(act, m->inter, R, L.wdown, H, m->inter, m->wdt, h, H);
      if (!rows8) use_w8(a, L.q_down);
    }
    a.res = h; a.ldres = H;
    VV_TRY(lin(a, L.f_down, nullptr));
  }
  if (!out) return 0;       // the caller runs the final norm itself (vv_llm_tail: norm + logits + token + bookkeeping in one launch)
  return vv_rmsnorm_rows(h, H, m->final_norm, m->rms_eps, R, H, out, ldo, s);
}

// ---------------------------------------------------------------------------------------------------------------
// diffusion head + DPM-Solver++ sampling
// ---------------------------------------------------------------------------------------------------------------
extern "C" size_t vv_head_ws_bytes(const vv_head* h, int n_steps) {
  if (!h || n_steps <= 0) return 0;
  const size_t R = 2 * (size_t)n_steps > 8 ? 2 * (size_t)n_steps : 8;
  const size_t D = h->D;
  return al(8 * D) /*c0*/ + al(R * D) /*c*/ + (size_t)h->layers * al(R * 3 * D) + al(R * 2 * D) + al(8 * D) /*hcur*/ +
         al(8 * (size_t)h->ffn) + al(8 * (size_t)h->latent) /*v*/ + 4 * al(h->latent) /*x, x0 history: double-buffered*/ +
         al(8 * D) /*second hidden-row buffer*/ + 2 * al(D + (size_t)h->latent) /*fused solver state X, M*/;
}

// shared body: rows R (<= 8), modulation tables mod[l] [*, 3D] / modf [*, 2D] with row offset `mrow`
static int head_body(const vv_head* h, const float* x, int64_t ldx, int R, float* const* mod, const float* modf, int64_t mrow,
                     float* hcur, float* act, float* v, vv_stream_t stream) {
  const int D = h->D;
  vv_lin_args a;
  if (x) {                                   // x == NULL: hcur = noisy_images_proj(x) was already produced (vv_dpm_proj)
    a = lin_base(x, ldx, R, h->noisy_proj, D, h->latent, h->wdt, hcur, D);
    VV_TRY(vv_linear(&a, stream));
  }
  for (int l = 0; l < h->layers; ++l) {
    const vv_head_layer& L = h->layer[l];
    const float* ml = mod[l] + mrow * 3 * D;
    a = lin_base(hcur, D, R, L.wgate, h->ffn, D, h->wdt, act, h->ffn);
    a.pro = VV_PRO_RMSNORM; a.norm_w = L.norm_w; a.eps = h->eps;
    a.mod_shift = ml; a.mod_scale = ml + D; a.ld_mod = 3 * D;
    a.w2 = L.wup; a.act = VV_ACT_SWIGLU;
    use_w8(a, L.q_gate, &L.q_up);
    VV_TRY(vv_linear(&a, stream));
    a = lin_base(act, h->ffn, R, L.wdown, D, h->ffn, h->wdt, hcur, D);
    a.gate = ml + 2 * D; a.gate_ld = 3 * D; a.res = hcur; a.ldres = D;
    use_w8(a, L.q_down);
    VV_TRY(vv_linear(&a, stream));
  }
  const float* mf = modf + mrow * 2 * D;
  a = lin_base(hcur, D, R, h->final_linear, h->latent, D, h->wdt, v, h->latent);
  a.pro = VV_PRO_RMSNORM; a.norm_w = nullptr; a.eps = h->eps;
  a.mod_shift = mf; a.mod_scale = mf + D; a.ld_mod = 2 * D;
  return vv_linear(&a, stream);
}

// c: fp32 rows (presilu false: SiLU applied as the GEMM prologue) or, with c_bf16, SiLU'd rows already rounded to bf16 (the operand the
// matrix-core path would round to anyway; the >= 32-tile adaLN GEMMs then run on the LDS-tiled kernel)
static int head_modulations(const vv_head* h, const float* c, int rows, float* const* mod, float* modf, bool presilu, bool c_bf16, vv_stream_t stream) {
  const int D = h->D;
  if (presilu && c_bf16) {     // the per-frame form (rows = 2 * n_steps bf16 conditioning rows): every matrix in one launch (vv_fused.hip)
    const int one = vv_head_modulations_fused(h, c, rows, mod, modf, (hipStream_t)stream);
    if (one) return one < 0 ? one : 0;
  }
  for (int l = 0; l < h->layers; ++l) {
    vv_lin_args a = lin_base(c, D, rows, h->layer[l].adaln, 3 * D, D, h->wdt, mod[l], 3 * D);
    a.pro = presilu ? VV_PRO_NONE : VV_PRO_SILU;
    if (c_bf16) a.flags = VV_LIN_X_BF16;
    VV_TRY(vv_linear(&a, stream));
  }
  vv_lin_args a = lin_base(c, D, rows, h->final_adaln, 2 * D, D, h->wdt, modf, 2 * D);
  a.pro = presilu ? VV_PRO_NONE : VV_PRO_SILU;
  if (c_bf16) a.flags = VV_LIN_X_BF16;
  return vv_linear(&a, stream);
}

extern "C" int vv_head_forward(const vv_head* h, const float* x, const float* temb_rows, const float* cond, int R, float* v,
                               void* ws, vv_stream_t stream) {
  if (!h || !x || !temb_rows || !cond || !v || !ws) return vv_set_error(VV_E_ARG, "vv_head_forward: null pointer");
  if (R <= 0 || R > 8 || h->layers > 16) return vv_set_error(VV_E_ARG, "vv_head_forward: R=%d (1..8)", R);
  const int D = h->D;
  Carver cv(ws);
  float* c0 = cv.take(8 * (size_t)D);
  float* c = cv.take(8 * (size_t)D);
  float* mod[16];
  for (int l = 0; l < h->layers; ++l) mod[l] = cv.take(8 * 3 * (size_t)D);
  float* modf = cv.take(8 * 2 * (size_t)D);
  float* hcur = cv.take(8 * (size_t)D);
  float* act = cv.take(8 * (size_t)h->ffn);
  vv_lin_args a = lin_base(cond, h->cond_dim, R, h->cond_proj, D, h->cond_dim, h->wdt, c0, D);
  VV_TRY(vv_linear(&a, stream));
  // c[r] = c0[r] + temb_rows[r]: rows_b = R with a single "step" whose b-row is row r -> use add_rows twice-free form
  for (int r = 0; r < R; ++r) VV_TRY(vv_add_rows(c0 + (size_t)r * D, D, temb_rows + (size_t)r * D, D, c + (size_t)r * D, 1, 1, D, stream));
  VV_TRY(head_modulations(h, c, R, mod, modf, false, false, stream));
  return head_body(h, x, h->latent, R, mod, modf, 0, hcur, act, v, stream);
}

extern "C" int vv_head_sample(const vv_head* h, const float* cond2, int64_t ld_cond, const float* noise, const float* temb,
                              const vv_dpm_coef* coef, int n_steps, float cfg_scale, float* latent_out, void* ws,
                              const float* sde_noise, vv_stream_t stream) {
  if (!h || !cond2 || !noise || !temb || !coef || !latent_out || !ws) return vv_set_error(VV_E_ARG, "vv_head_sample: null pointer");
  if (n_steps <= 0 || h->layers > 16) return vv_set_error(VV_E_ARG, "vv_head_sample: bad n_steps/layers");
  hipStream_t s = (hipStream_t)stream;
  const int D = h->D;
  const size_t R = 2 * (size_t)n_steps > 8 ? 2 * (size_t)n_steps : 8;
  Carver cv(ws);
  float* c0 = cv.take(8 * (size_t)D);
  float* c = cv.take(R * D);
  float* mod[16];
  for (int l = 0; l < h->layers; ++l) mod[l] = cv.take(R * 3 * D);
  float* modf = cv.take(R * 2 * D);
  float* hcur = cv.take(8 * (size_t)D);
  float* act = cv.take(8 * (size_t)h->ffn);
  float* v = cv.take(8 * (size_t)h->latent);
  float* xb[2] = {cv.take(h->latent), cv.take(h->latent)};       // x and x0-history are double-buffered per step
  float* mb[2] = {cv.take(h->latent), cv.take(h->latent)};
  float* hcur2 = cv.take(8 * (size_t)D);
  float* Xs = cv.take(D + (size_t)h->latent);
  float* Ms = cv.take(D + (size_t)h->latent);
  // step-invariant work hoisted out of the loop: cond_proj, silu(cond_proj(cond) + t_emb(t_i)), all adaLN modulations
  const bool cb = h->wdt == VV_BF16 && 2 * n_steps > 8 && D % 32 == 0 && ((uintptr_t)c % 16 == 0);
  bool ode0 = !sde_noise;
  for (int i = 0; i < n_steps && ode0; ++i) ode0 = coef[i].cn == 0.f;
  // the ODE solver on the fused boundary: cond_proj, the row expansion and the solver-state initialisation are one launch (vv_fused.hip)
  int pre = 0;
  if (cb && ode0 && vv_head_boundary_supported(h)) {
    pre = vv_head_pre_fused(h, cond2, ld_cond, temb, n_steps, c, noise, Xs, Ms, hcur, D, s);
    if (pre < 0) return pre;
  }
  vv_lin_args a;
  if (!pre) {
    a = lin_base(cond2, ld_cond, 2, h->cond_proj, D, h->cond_dim, h->wdt, c0, D);
    VV_TRY(vv_linear(&a, stream));
    if (cb) VV_TRY(vv_add_rows_silu_bf16(c0, D, temb, D, c, 2 * n_steps, 2, D, stream));
    else VV_TRY(vv_add_rows_silu(c0, D, temb, D, c, 2 * n_steps, 2, D, stream));
  }
  VV_TRY(head_modulations(h, c, 2 * n_steps, mod, modf, true, cb, stream));
  bool ode = !sde_noise;
  for (int i = 0; i < n_steps && ode; ++i) ode = coef[i].cn == 0.f;          // the SDE solver (variance noise per step) keeps the three-launch boundary
  if (ode && vv_head_boundary_supported(h)) {
    // one launch per step boundary: FinalLayer + CFG + DPM-Solver++ update + next noisy_images_proj as ONE GEMV over G = [P F ; F]
    // with the solver in its epilogue (vv_fused.hip).  The hidden rows alternate between two buffers: the boundary kernel of step i
    // reads the rows the layers of step i worked on while it writes the rows step i + 1 starts from.
    float* hb[2] = {hcur, hcur2};
    if (!pre) VV_TRY(vv_head_init_fused(h, noise, Xs, Ms, hb[0], D, s));
    for (int i = 0; i < n_steps; ++i) {
      float* hc = hb[i & 1];
      for (int l = 0; l < h->layers; ++l) {
        const vv_head_layer& L = h->layer[l];
        const float* ml = mod[l] + (size_t)2 * i * 3 * D;
        a = lin_base(hc, D, 2, L.wgate, h->ffn, D, h->wdt, act, h->ffn);
        a.pro = VV_PRO_RMSNORM; a.norm_w = L.norm_w; a.eps = h->eps;
        a.mod_shift = ml; a.mod_scale = ml + D; a.ld_mod = 3 * D;
        a.w2 = L.wup; a.act = VV_ACT_SWIGLU; a.flags = VV_LIN_W_REUSED;
        use_w8(a, L.q_gate, &L.q_up);
        VV_TRY(vv_linear(&a, stream));
        a = lin_base(act, h->ffn, 2, L.wdown, D, h->ffn, h->wdt, hc, D);
        a.gate = ml + 2 * D; a.gate_ld = 3 * D; a.res = hc; a.ldres = D; a.flags = VV_LIN_W_REUSED;
        use_w8(a, L.q_down);
        VV_TRY(vv_linear(&a, stream));
      }
      const float* mf = modf + (size_t)2 * i * 2 * D;
      VV_TRY(vv_head_boundary_fused(h, hc, D, mf, mf + D, 2 * D, cfg_scale, &coef[i], Xs, Ms, hb[(i + 1) & 1], D, latent_out, s));
    }
    return 0;
  }
  for (int i = 0; i <= n_steps; ++i) {
    // step boundary: solver update of the previous step's v (none before step 0) fused with this step's noisy_images_proj
    const float* xin = (i == 0) ? noise : xb[(i - 1) & 1];
    float* xout = (i == n_steps) ? latent_out : xb[i & 1];
    VV_TRY(vv_dpm_proj(i == 0 ? nullptr : v, h->latent, cfg_scale, i == 0 ? nullptr : &coef[i - 1], xin, mb[(i - 1) & 1], xout, mb[i & 1],
                       i == n_steps ? nullptr : h->noisy_proj, h->wdt, h->latent, D, hcur, D, 2,
                       (sde_noise && i > 0) ? sde_noise + (size_t)(i - 1) * h->latent : nullptr, stream));
    if (i == n_steps) break;
    VV_TRY(head_body(h, nullptr, 0, 2, mod, modf, 2 * (int64_t)i, hcur, act, v, stream));
  }
  (void)s;
  return 0;
}

// B utterances per call: rows [2 B] everywhere the single-utterance sampler has 2; conditioning rows laid out [step][2 B]
extern "C" size_t vv_head_ws_bytes_batch(const vv_head* h, int n_steps, int B) {
  if (!h || n_steps <= 0 || B <= 0 || B > 4) return 0;
  const size_t R = (size_t)2 * B * n_steps, D = h->D;
  return al(8 * D) /*c0*/ + al(R * D) /*c (bf16 rows fit)*/ + (size_t)h->layers * al(R * 3 * D) + al(R * 2 * D) + 2 * al(8 * D) /*hidden rows x 2*/ +
         al(8 * (size_t)h->ffn) + 2 * al((size_t)B * (D + h->latent)) /*solver state X, M*/ +
         al(vv_gemv_rows_part_floats(h->D, 0)) + al(vv_gemv_rows_tickets(h->ffn));
}

extern "C" int vv_head_sample_batch(const vv_head* h, const float* cond, int64_t ld_cond, const float* noise, int64_t ld_noise, const float* temb,
                                    const vv_dpm_coef* coef, int n_steps, float cfg_scale, float* latent_out, int64_t ld_latent, int B, void* ws,
                                    vv_stream_t stream) {
  if (!h || !cond || !noise || !temb || !coef || !latent_out || !ws) return vv_set_error(VV_E_ARG, "vv_head_sample_batch: null pointer");
  if (n_steps <= 0 || h->layers > 16 || B <= 0 || B > 4) return vv_set_error(VV_E_ARG, "vv_head_sample_batch: n_steps=%d layers=%d B=%d (1..4)", n_steps, h->layers, B);
  for (int i = 0; i < n_steps; ++i)
    if (coef[i].cn != 0.f) return vv_set_error(VV_E_UNSUPPORTED, "vv_head_sample_batch: the SDE solver is served per utterance (vv_head_sample)");
  if (h->wdt != VV_BF16 || !vv_head_boundary_supported(h) || h->D % 32)
    return vv_set_error(VV_E_UNSUPPORTED, "vv_head_sample_batch: needs bf16 weights and the fused solver boundary (vv_head.fused_g)");
  hipStream_t s = (hipStream_t)stream;
  const int D = h->D, R2 = 2 * B;
  const size_t R = (size_t)R2 * n_steps;
  Carver cv(ws);
  float* c0 = cv.take(8 * (size_t)D);
  float* c = cv.take(R * D);
  float* mod[16];
  for (int l = 0; l < h->layers; ++l) mod[l] = cv.take(R * 3 * D);
  float* modf = cv.take(R * 2 * D);
  float* hb[2] = {cv.take(8 * (size_t)D), cv.take(8 * (size_t)D)};
  float* act = cv.take(8 * (size_t)h->ffn);
  const int64_t sst = D + h->latent;
  float* Xs = cv.take((size_t)B * sst);
  float* Ms = cv.take((size_t)B * sst);
  const size_t rpart_n = vv_gemv_rows_part_floats(h->D, 0), rtick_n = vv_gemv_rows_tickets(h->ffn);
  float* rpart = cv.take(rpart_n);
  int* rtick = reinterpret_cast<int*>(cv.take(rtick_n));
  if (hipMemsetAsync(rtick, 0, rtick_n * sizeof(int), s) != hipSuccess) return vv_set_error(VV_E_HIP, "vv_head_sample_batch: memset");
  // step-invariant work: cond_proj on all rows, silu(cond_proj(cond) + t_emb(t_i)) as bf16 rows [step][2 B], every adaLN modulation
  vv_lin_args a = lin_base(cond, ld_cond, R2, h->cond_proj, D, h->cond_dim, h->wdt, c0, D);
  VV_TRY(vv_linear_ws(&a, nullptr, nullptr, rpart, rpart_n, rtick, rtick_n, stream));
  VV_TRY(vv_add_rows_silu_bf16(c0, D, temb, D, c, (int)R, R2, D, stream));
  VV_TRY(head_modulations(h, c, (int)R, mod, modf, true, true, stream));
  for (int b = 0; b < B; ++b)
    VV_TRY(vv_head_init_fused(h, noise + (size_t)b * ld_noise, Xs + (size_t)b * sst, Ms + (size_t)b * sst, hb[0] + (size_t)2 * b * D, D, s));
  for (int i = 0; i < n_steps; ++i) {
    float* hc = hb[i & 1];
    for (int l = 0; l < h->layers; ++l) {
      const vv_head_layer& L = h->layer[l];
      const float* ml = mod[l] + (size_t)R2 * i * 3 * D;
      a = lin_base(hc, D, R2, L.wgate, h->ffn, D, h->wdt, act, h->ffn);
      a.pro = VV_PRO_RMSNORM; a.norm_w = L.norm_w; a.eps = h->eps;
      a.mod_shift = ml; a.mod_scale = ml + D; a.ld_mod = 3 * D;
      a.w2 = L.wup; a.act = VV_ACT_SWIGLU; a.flags = VV_LIN_W_REUSED;
      VV_TRY(vv_linear_ws(&a, L.f_gate, L.f_up, rpart, rpart_n, rtick, rtick_n, stream));
      a = lin_base(act, h->ffn, R2, L.wdown, D, h->ffn, h->wdt, hc, D);
      a.gate = ml + 2 * D; a.gate_ld = 3 * D; a.res = hc; a.ldres = D; a.flags = VV_LIN_W_REUSED;
      VV_TRY(vv_linear_ws(&a, L.f_down, nullptr, rpart, rpart_n, rtick, rtick_n, stream));
    }
    const float* mf = modf + (size_t)R2 * i * 2 * D;
    VV_TRY(vv_head_boundary_batch(h, hc, D, mf, mf + D, 2 * D, cfg_scale, &coef[i], Xs, Ms, sst, hb[(i + 1) & 1], D, latent_out, ld_latent, B, s));
  }
  return 0;
}

// ---------------------------------------------------------------------------------------------------------------
// causal conv tokenizer (decoder / encoder)
// ---------------------------------------------------------------------------------------------------------------
static inline int conv_ctx_of(const vv_conv& c) { return c.transposed ? 1 : c.kk - c.stride; }

// max floats any [rows, C] activation (incl. left context and right padding) can take for an input of t_in steps
static void convnet_sizes(const vv_convnet* net, int64_t t_in, int decoder, size_t* act_elems, size_t* hid_elems) {
  size_t amax = 0, hmax = 0;
  int64_t T = t_in;
  for (int i = 0; i < net->n_stages; ++i) {
    const vv_conv& cv = net->sample[i];
    const int ctx = conv_ctx_of(cv);
    size_t in_el = (size_t)(T + ctx + cv.kk + cv.stride) * cv.cin;   // generous right padding for ragged tails
    if (in_el > amax) amax = in_el;
    if (cv.transposed) T = T * cv.stride; else T = (T + cv.stride - 1) / cv.stride;
    const size_t C = cv.cout;
    size_t el = (size_t)(T + 8) * C;
    if (el > amax) amax = el;
    if (net->n_blocks[i] > 0 && (size_t)T * 4 * C > hmax) hmax = (size_t)T * 4 * C;
  }
  size_t in_el = (size_t)(T + 8 + net->head.kk) * net->head.cin;
  if (in_el > amax) amax = in_el;
  (void)decoder;
  *act_elems = amax + 64;
  *hid_elems = hmax + 64;
}

// streaming nets (every conv carries a state buffer): each conv's padded input [ctx + T, cin] gets a region of its own, so the left
// contexts of ALL convs can be put in place by one launch before the first conv runs and collected by one launch after the last
static bool convnet_streaming(const vv_convnet* net) {
  for (int i = 0; i <= net->n_stages; ++i) {
    const vv_conv& cv = (i == net->n_stages) ? net->head : net->sample[i];
    if (conv_ctx_of(cv) > 0 && !cv.state) return false;
  }
  return true;
}
static size_t convnet_pad_floats(const vv_convnet* net, int64_t t_in, size_t* offs /* [n_stages + 1] or null */) {
  size_t tot = 0;
  int64_t T = t_in;
  for (int i = 0; i <= net->n_stages; ++i) {
    const vv_conv& cv = (i == net->n_stages) ? net->head : net->sample[i];
    if (offs) offs[i] = tot;
    tot += (((size_t)(T + conv_ctx_of(cv) + 2) * cv.cin) + 63) & ~(size_t)63;
    if (cv.transposed) T = T * cv.stride; else T = (T + cv.stride - 1) / cv.stride;
  }
  return tot;
}

// scratch histories of the one-row stage's blocks (<= 16 blocks x 6 rows x 2048 channels), see run_blocks
#define VV_ROW_BLOCKS 16
#define VV_ROW_HIST_FLOATS ((size_t)VV_ROW_BLOCKS * 6 * 2048)
#define VV_CTX_ITEMS (VV_MAX_STAGES + 1 + 16)   // capacity of a net's streaming-state move list (= vv_conv_ctx_batch's limit, vv_fused.hip)

extern "C" size_t vv_convnet_ws_bytes(const vv_convnet* net, int64_t t_in, int decoder) {
  if (!net || t_in <= 0) return 0;
  size_t a, h;
  convnet_sizes(net, t_in, decoder, &a, &h);
  return 2 * al(a) + al(h) + (convnet_streaming(net) ? al(convnet_pad_floats(net, t_in, nullptr)) + al(VV_ROW_HIST_FLOATS) : 0);
}

// Runs the stage's blocks on the ping-pong buffers cur / other.  The LAST block writes its result `nctx` rows into a buffer so
// that it becomes the next conv's padded input; *pad_out is that buffer.  Unfused block (mixer -> other, lin1 -> hid, lin2): the
// dead input buffer is the only one free, so the result goes there, shifted.  Fused block (one launch reads its input while
// other row tiles already write): the result must not overlap the input; `other` is free (no mixer output) and takes it.
static int run_blocks(const vv_convnet* net, int stage, int64_t T, int C, float*& cur, float*& other, float* hid,
                      int nctx, float** pad_out, vv_stream_t stream, float* next_pad = nullptr, float* row_hist = nullptr,
                      vv_conv_ctx_item* items = nullptr, int* n_items = nullptr, int* row_stage = nullptr) {
  // row_hist / items (streaming nets): scratch for the new histories of the one-row stage's blocks and the list of state moves that the
  // net's closing vv_conv_ctx_batch performs (a block's history may only be replaced once every workgroup that reads it is done)
  // next_pad (streaming nets): the next conv's own padded-input region; the stage result goes to next_pad + nctx rows
  const int nb = net->n_blocks[stage];
  for (int j = 0; j < nb; ++j) {
    const vv_block& B = net->blocks[stage][j];
    const bool last = (j == nb - 1);
    if (!vv_convffn_prefers(net->wdt, (int)T, C)) {   // narrow stages with many rows: the whole block as one launch (vv_block1d.hip)
      float* dst = last ? (next_pad ? next_pad : other) + (size_t)nctx * C : other;
      const int fused = vv_launch_block1d(B, net->wdt, cur, dst, (int)T, C, net->eps, (hipStream_t)stream);
      if (fused < 0) return fused;
      if (fused) {
        if (last) { *pad_out = next_pad ? next_pad : other; cur = nullptr; }
        else { float* t = cur; cur = other; other = t; }
        continue;
      }
    }
    float* final_dst = last ? (next_pad ? next_pad : cur) + (size_t)nctx * C : nullptr;
    if (last) *pad_out = next_pad ? next_pad : cur;
    // one-row fast path: row_hist holds one scratch history per block index j, so only ONE stage of a net may take it (the shipped nets
    // have a single T == 1 stage; a second one falls through to the general path below), and the item list must have room
    if (T == 1 && row_hist && items && n_items && j < VV_ROW_BLOCKS && B.hist && !B.q_w1.q && *n_items < VV_CTX_ITEMS &&
        (*row_stage < 0 || *row_stage == stage)) {
      // the one-row stage (C = 2048): mixer + RMSNorm + first GEMM + GELU as one launch (vv_convffn.hip), then the weight-streaming GEMV
      float* hn = row_hist + (size_t)j * 6 * C;
      int one = vv_launch_ffn_in_row_hs(B, net->wdt, cur, other, hid, hn, C, net->eps, (hipStream_t)stream);
      if (one == 0) one = vv_launch_ffn_in_row(B, net->wdt, cur, other, hid, hn, C, net->eps, (hipStream_t)stream);
      if (one < 0) return one;
      if (one) {
        *row_stage = stage;
        items[(*n_items)++] = vv_conv_ctx_item{hn, B.hist, 6, 0, C, B.dw_w, B.hs};   // the scatter also refreshes hs from the rows it stores
        float* dst1 = (last && final_dst) ? final_dst : other;
        vv_lin_args a1 = lin_base(hid, 4 * C, 1, B.w2, C, 4 * C, net->wdt, dst1, C);
        a1.bias = B.b2; a1.gate = B.ffn_gamma; a1.gate_ld = 0; a1.res = other; a1.ldres = C;
        use_w8(a1, B.q_w2);
        VV_TRY(vv_linear(&a1, stream));
        if (dst1 == other) { float* t = cur; cur = other; other = t; }
        else { cur = nullptr; }
        continue;
      }
    }
    if (B.hs && B.hist && items && n_items) {
      if (*n_items >= VV_CTX_ITEMS) return vv_set_error(VV_E_UNSUPPORTED, "convnet: more than %d streaming-state moves in one net", VV_CTX_ITEMS);
      items[(*n_items)++] = vv_conv_ctx_item{B.hist, B.hist, 6, 0, C, B.dw_w, B.hs};   // this block's history changes below on a general path: hs follows at the closing scatter
    }
    {   // middle stages of a streaming frame (C = 256 / 512): mixer + first GEMM, second GEMM (vv_convffn.hip); the bf16 hidden tile
        // fills the first half of `hid`, the scratch history sits behind it
      float* dst2 = (last && final_dst) ? final_dst : other;
      const int two = vv_launch_convffn(B, net->wdt, cur, other, hid, reinterpret_cast<float*>(reinterpret_cast<char*>(hid) + (size_t)T * 4 * C * 2),
                                        dst2, (int)T, C, net->eps, (hipStream_t)stream);
      if (two < 0) return two;
      if (two) {
        if (dst2 == other) { float* t = cur; cur = other; other = t; }
        else { cur = nullptr; }
        continue;
      }
    }
    VV_TRY(vv_block_mixer(cur, other, (int)T, C, B.norm_w, net->eps, B.dw_w, B.dw_b, B.gamma, B.hist, stream));
    // T > 8 rows on bf16 weights: both FFN linears run on the matrix cores and the 4C-wide hidden activation is handed
    // over in bf16 (half the bytes, and the second GEMM reads its fragments straight from it: no LDS staging)
    const bool handoff = net->wdt == VV_BF16 && T > 8 && C % 16 == 0 && ((uintptr_t)B.w1 % 16 == 0) && ((uintptr_t)B.w2 % 16 == 0);
    vv_lin_args a;
    const long tiles1 = (long)((4 * C) / 128) * ((T + 127) / 128);   // 128 x 128 output tiles of the first FFN GEMM
    if (handoff && C % 32 == 0 && (4 * C) % 128 == 0 && T >= 64 && tiles1 >= 24) {
      // whole-utterance sequences (enough output tiles for the LDS-tiled 128 x 128 kernel; a streaming frame never has them):
      // RMSNorm + bf16 cast once, then both FFN GEMMs stream bf16 activations.  The cast rows live in the half of `hid`
      // that the bf16 hidden tile leaves free.
      void* xb = reinterpret_cast<char*>(hid) + (size_t)T * 4 * C * 2;
      VV_TRY(vv_cast_rows_bf16(other, C, (int)T, C, VV_PRO_RMSNORM, B.ffn_norm_w, net->eps, xb, C, stream));
      a = lin_base((const float*)xb, C, (int)T, B.w1, 4 * C, C, net->wdt, hid, 4 * C);
      a.flags = VV_LIN_X_BF16; a.bias = B.b1; a.act = VV_ACT_GELU;
    } else {
      a = lin_base(other, C, (int)T, B.w1, 4 * C, C, net->wdt, hid, 4 * C);
      a.pro = VV_PRO_RMSNORM; a.norm_w = B.ffn_norm_w; a.eps = net->eps; a.bias = B.b1; a.act = VV_ACT_GELU;
    }
    if (handoff) a.flags |= VV_LIN_OUT_BF16;
    use_w8(a, B.q_w1);
    VV_TRY(vv_linear(&a, stream));
    float* dst = (j == nb - 1 && final_dst) ? final_dst : other;
    a = lin_base(hid, 4 * C, (int)T, B.w2, C, 4 * C, net->wdt, dst, C);
    if (handoff) a.flags |= VV_LIN_X_BF16;
    a.bias = B.b2; a.gate = B.ffn_gamma; a.gate_ld = 0; a.res = other; a.ldres = C;
    use_w8(a, B.q_w2);
    VV_TRY(vv_linear(&a, stream));
    if (dst == other) { float* t = cur; cur = other; other = t; }   // result now in `cur`
    else { cur = nullptr; }                                           // result went to final_dst
  }
  return 0;
}

// prepare the left context of a conv's padded input buffer `pad` (= [ctx rows | T rows already written by the producer])
static int conv_left_ctx(const vv_conv& cv, float* pad, int64_t T, vv_stream_t stream) {
  const int ctx = conv_ctx_of(cv);
  if (ctx <= 0) return 0;
  if (cv.state) return vv_conv_ctx(pad, cv.state, ctx, (int)T, cv.cin, stream);
  hipError_t e = hipMemsetAsync(pad, 0, (size_t)ctx * cv.cin * 4, (hipStream_t)stream);
  if (e != hipSuccess) return vv_set_error(VV_E_HIP, "conv_left_ctx: %s", hipGetErrorString(e));
  return 0;
}

extern "C" int vv_decoder_forward(const vv_convnet* net, const float* latent, int T0, float pre_scale, float pre_bias, float* wav,
                                  void* ws, vv_stream_t stream) {
  if (!net || !latent || !wav || !ws) return vv_set_error(VV_E_ARG, "vv_decoder_forward: null pointer");
  if (T0 <= 0 || net->n_stages < 1 || net->n_stages > VV_MAX_STAGES) return vv_set_error(VV_E_ARG, "vv_decoder_forward: bad T/stages");
  size_t ael, hel;
  convnet_sizes(net, T0, 1, &ael, &hel);
  Carver cvr(ws);
  float* A = cvr.take(ael);
  float* Bf = cvr.take(ael);
  float* hid = cvr.take(hel);
  int64_t T = T0;
  // stem input: padded [6 + T, vae] in A
  const vv_conv& stem = net->sample[0];
  if (stem.transposed || stem.stride != 1) return vv_set_error(VV_E_ARG, "vv_decoder_forward: stem must be a stride-1 conv");
  // streaming: every conv reads its own padded-input region; all left contexts are placed by ONE launch up front and collected by
  // ONE launch at the end (they were one dependent launch per conv: 8 per frame and net)
  const bool streaming = convnet_streaming(net);
  size_t poff[VV_MAX_STAGES + 1];
  float* pads = nullptr;
  vv_conv_ctx_item items[VV_CTX_ITEMS];
  int n_items = 0, row_stage = -1;
  float* row_hist = nullptr;
  if (streaming) {
    pads = cvr.take(convnet_pad_floats(net, T0, poff));
    row_hist = cvr.take(VV_ROW_HIST_FLOATS);
    int64_t Ti = T0;
    for (int i = 0; i <= net->n_stages; ++i) {
      const vv_conv& cv = (i == net->n_stages) ? net->head : net->sample[i];
      if (conv_ctx_of(cv) > 0) items[n_items++] = vv_conv_ctx_item{pads + poff[i], cv.state, conv_ctx_of(cv), (int)Ti, cv.cin};
      if (cv.transposed) Ti *= cv.stride;
    }
    // the scaled latent frame rides along with the context gather: one launch instead of two in front of the stem conv
    vv_conv_ctx_item gi[VV_MAX_STAGES + 2];
    for (int i = 0; i < n_items; ++i) gi[i] = items[i];
    gi[n_items] = vv_conv_ctx_item{pads + poff[0] + (size_t)conv_ctx_of(stem) * stem.cin, const_cast<float*>(latent), (int)T, 0, stem.cin, nullptr, nullptr, 1,
                                   pre_scale, pre_bias};
    VV_TRY(vv_conv_ctx_batch(gi, n_items + 1, 0, (hipStream_t)stream));
  }
  float* pad = streaming ? pads + poff[0] : A;
  if (!streaming) VV_TRY(vv_affine(latent, pre_scale, pre_bias, pad + (size_t)conv_ctx_of(stem) * stem.cin, (int64_t)T * stem.cin, stream));
  float* cur = nullptr;     // current activation buffer, `pad` holds the next conv's input
  float* other = nullptr;
  for (int i = 0; i < net->n_stages; ++i) {
    const vv_conv& cv = net->sample[i];
    if (!streaming) VV_TRY(conv_left_ctx(cv, pad, T, stream));
    float* outb = (pad == A) ? Bf : A;
    vv_lin_args a;
    if (cv.transposed) {
      a = lin_base(pad, cv.cin, (int)T, cv.w, cv.stride * cv.cout, 2 * cv.cin, net->wdt, outb, (int64_t)cv.stride * cv.cout);
      a.bias = cv.b;
      VV_TRY(vv_linear(&a, stream));
      T *= cv.stride;
    } else {
      a = lin_base(pad, (int64_t)cv.stride * cv.cin, (int)T, cv.w, cv.cout, cv.kk * cv.cin, net->wdt, outb, cv.cout);
      a.bias = cv.b;
      VV_TRY(vv_linear(&a, stream));
    }
    cur = outb;
    other = (cur == A) ? Bf : A;
    const int C = cv.cout;
    const vv_conv& nxt = (i + 1 < net->n_stages) ? net->sample[i + 1] : net->head;
    const int nctx = conv_ctx_of(nxt);
    float* next_pad = streaming ? pads + poff[i + 1] : nullptr;
    if (net->n_blocks[i] > 0) {
      // the last block writes straight into the next conv's padded input; which buffer that is depends on block parity:
      // block j reads cur -> writes other, then they swap.  The last block's mixer output sits in `other_last`, its
      // result may go anywhere except that buffer and hid: use the buffer holding the (dead) input of that block.
      VV_TRY(run_blocks(net, i, T, C, cur, other, hid, nctx, &pad, stream, next_pad, row_hist, items, &n_items, &row_stage));
    } else {
      float* dstb = next_pad ? next_pad : other;
      hipError_t e = hipMemcpyAsync(dstb + (size_t)nctx * C, cur, (size_t)T * C * 4, hipMemcpyDeviceToDevice, (hipStream_t)stream);
      if (e != hipSuccess) return vv_set_error(VV_E_HIP, "vv_decoder_forward: %s", hipGetErrorString(e));
      pad = dstb;
    }
  }
  const vv_conv& hd = net->head;
  if (!streaming) VV_TRY(conv_left_ctx(hd, pad, T, stream));
  vv_lin_args a = lin_base(pad, (int64_t)hd.stride * hd.cin, (int)T, hd.w, hd.cout, hd.kk * hd.cin, net->wdt, wav, hd.cout);
  a.bias = hd.b;
  VV_TRY(vv_linear(&a, stream));
  if (streaming) VV_TRY(vv_conv_ctx_batch(items, n_items, 1, (hipStream_t)stream));
  return 0;
}

extern "C" int vv_encoder_forward(const vv_convnet* net, const float* wav, int64_t T0, float* feat, void* ws, vv_stream_t stream) {
  if (!net || !wav || !feat || !ws) return vv_set_error(VV_E_ARG, "vv_encoder_forward: null pointer");
  if (T0 <= 0 || net->n_stages < 1 || net->n_stages > VV_MAX_STAGES) return vv_set_error(VV_E_ARG, "vv_encoder_forward: bad T/stages");
  hipStream_t s = (hipStream_t)stream;
  size_t ael, hel;
  convnet_sizes(net, T0, 0, &ael, &hel);
  Carver cvr(ws);
  float* A = cvr.take(ael);
  float* Bf = cvr.take(ael);
  float* hid = cvr.take(hel);
  int64_t T = T0;
  const vv_conv& stem = net->sample[0];
  // streaming frames (T0 a multiple of the hop, every conv with a state buffer): dedicated padded-input regions, all left contexts
  // gathered / scattered by one launch each (see vv_decoder_forward)
  bool streaming = convnet_streaming(net);
  {
    int64_t Ti = T0;
    for (int i = 0; i < net->n_stages && streaming; ++i) { if (Ti % net->sample[i].stride) streaming = false; Ti /= net->sample[i].stride; }
  }
  size_t poff[VV_MAX_STAGES + 1];
  float* pads = nullptr;
  vv_conv_ctx_item items[VV_CTX_ITEMS];
  int n_items = 0, row_stage = -1;
  float* row_hist = nullptr;
  if (streaming) {
    pads = cvr.take(convnet_pad_floats(net, T0, poff));
    row_hist = cvr.take(VV_ROW_HIST_FLOATS);
    int64_t Ti = T0;
    for (int i = 0; i <= net->n_stages; ++i) {
      const vv_conv& cv = (i == net->n_stages) ? net->head : net->sample[i];
      if (conv_ctx_of(cv) > 0) items[n_items++] = vv_conv_ctx_item{pads + poff[i], cv.state, conv_ctx_of(cv), (int)Ti, cv.cin};
      Ti = (Ti + cv.stride - 1) / cv.stride;
    }
    // the waveform chunk rides along with the context gather (no separate copy in front of the stem conv)
    vv_conv_ctx_item gi[VV_MAX_STAGES + 2];
    for (int i = 0; i < n_items; ++i) gi[i] = items[i];
    gi[n_items] = vv_conv_ctx_item{pads + poff[0] + (size_t)conv_ctx_of(stem) * stem.cin, const_cast<float*>(wav), (int)T, 0, stem.cin, nullptr, nullptr, 1, 1.0f, 0.0f};
    VV_TRY(vv_conv_ctx_batch(gi, n_items + 1, 0, s));
  }
  float* pad = streaming ? pads + poff[0] : A;
  hipError_t e = hipSuccess;
  if (!streaming) e = hipMemcpyAsync(pad + (size_t)conv_ctx_of(stem) * stem.cin, wav, (size_t)T * stem.cin * 4, hipMemcpyDeviceToDevice, s);
  if (e != hipSuccess) return vv_set_error(VV_E_HIP, "vv_encoder_forward: %s", hipGetErrorString(e));
  float* cur = nullptr;
  float* other = nullptr;
  for (int i = 0; i <= net->n_stages; ++i) {
    const bool is_head = (i == net->n_stages);
    const vv_conv& cv = is_head ? net->head : net->sample[i];
    if (cv.transposed) return vv_set_error(VV_E_ARG, "vv_encoder_forward: transposed conv in an encoder");
    const int ctx = conv_ctx_of(cv);
    if (!streaming) VV_TRY(conv_left_ctx(cv, pad, T, stream));
    // output length as the reference's non-streaming padding rule gives it (modular_vibevoice_tokenizer.py:127-133);
    // for streaming frames T is a multiple of the stride and there is no tail.
    const int64_t Tout = (T + cv.stride - 1) / cv.stride;
    const int64_t need = (Tout - 1) * cv.stride + cv.kk;        // rows of the padded buffer the conv reads
    const int64_t have = ctx + T;
    if (need > have) {
      e = hipMemsetAsync(pad + (size_t)have * cv.cin, 0, (size_t)(need - have) * cv.cin * 4, s);
      if (e != hipSuccess) return vv_set_error(VV_E_HIP, "vv_encoder_forward: %s", hipGetErrorString(e));
    }
    float* outb = is_head ? feat : ((pad == A) ? Bf : A);
    vv_lin_args a = lin_base(pad, (int64_t)cv.stride * cv.cin, (int)Tout, cv.w, cv.cout, cv.kk * cv.cin, net->wdt, outb, cv.cout);
    a.bias = cv.b;
    {   // whole-utterance sequences: the padded input is cast to bf16 once (into `hid`, free between stages) and the conv - a GEMM
        // over overlapping rows - runs on the LDS-tiled kernel; a streaming frame never has the >= 24 output tiles this needs
      const long tiles = (long)(cv.cout / 128) * ((Tout + 127) / 128);
      const size_t in_el = (size_t)need * cv.cin;
      if (net->wdt == VV_BF16 && !is_head && cv.cout % 128 == 0 && (cv.kk * cv.cin) % 32 == 0 && (cv.stride * cv.cin) % 8 == 0 && cv.cin >= 128 && cv.cin % 8 == 0 &&
          tiles >= 24 && in_el * 2 <= hel * 4 && ((uintptr_t)cv.w % 16 == 0)) {
        VV_TRY(vv_cast_rows_bf16(pad, cv.cin, (int)need, cv.cin, VV_PRO_NONE, nullptr, 0.f, hid, cv.cin, stream));
        a.x = hid; a.flags = VV_LIN_X_BF16;
      } else if (net->wdt == VV_BF16 && is_head && Tout >= 64 && cv.cout % 64 == 0 && (cv.kk * cv.cin) % 128 == 0 && cv.kk * cv.cin >= 8192 &&
                 (cv.stride * cv.cin) % 8 == 0 && cv.cin % 8 == 0 && in_el * 2 <= hel * 4 && ((uintptr_t)cv.w % 16 == 0)) {
        // the head conv of a long sequence (few output channels, K = 7 x 2048): same cast, 64 x 64 tiles over the long K
        VV_TRY(vv_cast_rows_bf16(pad, cv.cin, (int)need, cv.cin, VV_PRO_NONE, nullptr, 0.f, hid, cv.cin, stream));
        a.x = hid; a.flags = VV_LIN_X_BF16;
      }
    }
    VV_TRY(vv_linear(&a, stream));
    if (is_head) break;
    T = Tout;
    cur = outb;
    other = (cur == A) ? Bf : A;
    const int C = cv.cout;
    const vv_conv& nxt = (i + 1 < net->n_stages) ? net->sample[i + 1] : net->head;
    const int nctx = conv_ctx_of(nxt);
    float* next_pad = streaming ? pads + poff[i + 1] : nullptr;
    if (net->n_blocks[i] > 0) {
      VV_TRY(run_blocks(net, i, T, C, cur, other, hid, nctx, &pad, stream, next_pad, row_hist, items, &n_items, &row_stage));
    } else {
      float* dstb = next_pad ? next_pad : other;
      e = hipMemcpyAsync(dstb + (size_t)nctx * C, cur, (size_t)T * C * 4, hipMemcpyDeviceToDevice, s);
      if (e != hipSuccess) return vv_set_error(VV_E_HIP, "vv_encoder_forward: %s", hipGetErrorString(e));
      pad = dstb;
    }
  }
  if (streaming) VV_TRY(vv_conv_ctx_batch(items, n_items, 1, s));
  return 0;
}

extern "C" int vv_convnet_reset(const vv_convnet* net, vv_stream_t stream) {
  if (!net) return vv_set_error(VV_E_ARG, "vv_convnet_reset: null");
  hipStream_t s = (hipStream_t)stream;
  for (int i = 0; i <= net->n_stages; ++i) {
    const vv_conv& cv = (i == net->n_stages) ? net->head : net->sample[i];
    if (cv.state) {
      hipError_t e = hipMemsetAsync(cv.state, 0, (size_t)conv_ctx_of(cv) * cv.cin * 4, s);
      if (e != hipSuccess) return vv_set_error(VV_E_HIP, "vv_convnet_reset: %s", hipGetErrorString(e));
    }
    if (i < net->n_stages) {
      for (int j = 0; j < net->n_blocks[i]; ++j) {
        const vv_block& B = net->blocks[i][j];
        if (B.hist) {
          hipError_t e = hipMemsetAsync(B.hist, 0, (size_t)6 * net->sample[i].cout * 4, s);
          if (e != hipSuccess) return vv_set_error(VV_E_HIP, "vv_convnet_reset: %s", hipGetErrorString(e));
        }
        if (B.hs) {
          hipError_t e = hipMemsetAsync(B.hs, 0, (size_t)net->sample[i].cout * 4, s);
          if (e != hipSuccess) return vv_set_error(VV_E_HIP, "vv_convnet_reset: %s", hipGetErrorString(e));
        }
      }
    }
  }
  return 0;
}

// ---------------------------------------------------------------------------------------------------------------
// SpeechConnector
// ---------------------------------------------------------------------------------------------------------------
extern "C" int vv_connector_forward(const vv_connector* c, const float* x, int R, float* out, int accumulate, float* ws, vv_stream_t stream) {
  if (!c || !x || !out || !ws || R <= 0) return vv_set_error(VV_E_ARG, "vv_connector_forward: bad args");
  vv_lin_args a = lin_base(x, c->din, R, c->fc1, c->hidden, c->din, c->wdt, ws, c->hidden);
  a.bias = c->b1;
  VV_TRY(vv_linear(&a, stream));
  a = lin_base(ws, c->hidden, R, c->fc2, c->hidden, c->hidden, c->wdt, out, c->hidden);
  a.pro = VV_PRO_RMSNORM; a.norm_w = c->norm_w; a.eps = 1e-6f; a.bias = c->b2;
  if (accumulate) { a.res = out; a.ldres = c->hidden; }
  return vv_linear(&a, stream);
}

// acoustic + semantic connector of ONE frame, summed, stored to rows 0 .. rows_out - 1 of out (the next step's input embedding of the positive
// and the negative branch are the same vector, modeling_vibevoice_inference.py:665-673).  ws: 2 * hidden floats.
extern "C" int vv_connector_pair(const vv_connector* ac, const vv_connector* sem, const float* latent, const float* semfeat, float* out, int64_t ldo,
                                 int rows_out, float* ws, vv_stream_t stream) {
  if (!ac || !sem || !latent || !semfeat || !out || !ws || rows_out < 1) return vv_set_error(VV_E_ARG, "vv_connector_pair: bad args");
  const int one = vv_launch_connector_pair(ac, sem, latent, semfeat, out, ldo, rows_out, ws, (hipStream_t)stream);
  if (one) return one < 0 ? one : 0;
  VV_TRY(vv_connector_forward(ac, latent, 1, out, 0, ws, stream));
  VV_TRY(vv_connector_forward(sem, semfeat, 1, out, 1, ws, stream));
  for (int r = 1; r < rows_out; ++r) VV_TRY(vv_copy_rows(out, 0, out + r * ldo, ldo, 1, ac->hidden, stream));
  return 0;
}

extern "C" size_t vv_sizeof(const char* name) {
  if (!name) return 0;
#define S(t) if (!strcmp(name, #t)) return sizeof(t);
  S(vv_w8) S(vv_lin_args) S(vv_kv) S(vv_llm_layer) S(vv_llm) S(vv_head_layer) S(vv_head) S(vv_dpm_coef) S(vv_block) S(vv_conv) S(vv_convnet) S(vv_connector) S(vv_prof_entry)
#undef S
  return 0;
}
