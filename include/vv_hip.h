/*
 * vv_hip.h — C ABI of libvv_hip.so: the MI355X (gfx950) kernels of the VibeVoice per-frame hot path.
 *
 * The reference (beecave-homelab/VibeVoice-ROCm) has no native/FFI layer at all: its hot
 * path is Python calling torch ops.  This ABI is therefore the boundary *we* define underneath the reference's
 * Python class API; each entry point names the reference Python code whose arithmetic it replaces
 * (paths relative to the reference root).  The Python host (vibevoice_rocm_amd/) binds it with ctypes.
 *
 * Conventions
 *   - extern "C", plain pointers and ints only.  All data pointers are DEVICE pointers unless marked (host).
 *   - every function returns 0 on success, a negative VV_E_* code otherwise; vv_last_error() gives the text
 *     (thread-local).  Nothing throws, nothing allocates device memory: workspaces are caller-provided and sized
 *     with the matching *_ws_bytes() query.  No hidden synchronisation: everything is enqueued on `stream`
 *     and is capturable into a hipGraph (vv_graph_*).
 *   - activations are fp32.  Matrix weights are fp32 or bf16 (`wdt`), vectors (norms, biases, layer scales,
 *     depthwise taps) are always fp32.  KV cache is fp32 or bf16 (`kvdt`).
 *   - convolutional activations are channels-last [T, C] (the reference uses [B, C, T]).
 */
#ifndef VV_HIP_H
#define VV_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* vv_stream_t; /* hipStream_t */

enum { VV_F32 = 0, VV_BF16 = 1, VV_FP8 = 2 /* e4m3fn bytes + per-output-row fp32 scale (vv_lin_args.wscale); streaming GEMV (m <= 2) only */ };
enum { VV_OK = 0, VV_E_ARG = -1, VV_E_HIP = -2, VV_E_UNSUPPORTED = -3 };
enum { VV_PRO_NONE = 0, VV_PRO_RMSNORM = 1, VV_PRO_SILU = 2 };
enum { VV_ACT_NONE = 0, VV_ACT_GELU = 1, VV_ACT_SWIGLU = 2 };
/* vv_lin_args.flags: bf16 activation hand-off between two matrix-core GEMMs (x / out point at bf16 [m, ld] arrays, no
 * prologue on a bf16 x), a hint that the weights are re-read soon (keep them cacheable instead of streaming them
 * non-temporally: the diffusion head's matrices are reused by every one of the N solver steps of a frame), and
 * VV_LIN_W_FRAG: w / w2 point at the fragment-major copies of the matrices (vv_llm_layer.f_*; 3..8 rows, bf16, n % 16 == 0,
 * k % 32 == 0 - any other call with this flag is an error) */
enum { VV_LIN_X_BF16 = 1, VV_LIN_OUT_BF16 = 2, VV_LIN_W_REUSED = 4, VV_LIN_W_FRAG = 8 };

const char* vv_last_error(void);
int vv_abi_version(void);
int vv_tune(const char* key, int value); /* developer tuning hook (grid-size overrides for micro-benchmarks) */
int vv_init(void); /* one-time per-process kernel attribute setup; call before any graph capture */

/* ------------------------------------------------------------------------------------------------------------
 * vv_linear — out[m, n] = epilogue( sum_k W[n, k] * prologue(x)[m, k] )
 * One fused primitive for every Linear / dense Conv1d / ConvTranspose1d of the path:
 *   nn.Linear (q/k/v/o/gate/up/down, heads, connectors, FFN)      x rows contiguous, ldx = k
 *   SConv1d dense, kernel kk, stride s on channels-last data       x = padded [ctx+T, C] buffer, ldx = s*C, k = kk*C
 *   SConvTranspose1d with kk = 2 s                                  x = [1+T, C] buffer, ldx = C, k = 2*C, n = s*C_out
 * prologue (applied to each x row before the product):
 *   VV_PRO_RMSNORM: x * rsqrt(mean(x^2) + eps) [* norm_w] then, if mod_scale != NULL, * (1 + mod_scale[m]) + mod_shift[m]
 *                   (Qwen2RMSNorm; diffusion-head RMSNorm + modulate, modular_vibevoice_diffusion_head.py:31-45;
 *                   ConvRMSNorm, modular_vibevoice_tokenizer.py:77-91; LlamaRMSNorm in SpeechConnector)
 *   VV_PRO_SILU:    silu(x)  (adaLN_modulation / TimestepEmbedder, modular_vibevoice_diffusion_head.py:58-63,152-156)
 * epilogue: + bias[n]; VV_ACT_GELU (erf form, FFN modular_vibevoice_tokenizer.py:589) or VV_ACT_SWIGLU
 *   (silu(W x) * (W2 x), FeedForwardNetwork :116-123 / Qwen2MLP); then * gate (gate_ld == 0: per-channel vector
 *   gate[n] = layer scale gamma; gate_ld > 0: per-row gate[m*gate_ld + n] = adaLN gate); then + res[m*ldres + n].
 * `out` may alias `res`.  m <= 8 rows stream the weights once through a GEMV kernel, larger m use a tiled GEMM.
 * ------------------------------------------------------------------------------------------------------------ */
typedef struct vv_lin_args {
  const float* x;
  int64_t ldx;
  int m;
  int pro;
  const float* norm_w;
  float eps;
  const float* mod_shift;
  const float* mod_scale;
  int64_t ld_mod;
  const void* w;
  const void* w2;
  const float* bias;
  int n, k, wdt;
  int act;
  const float* gate;
  int64_t gate_ld;
  const float* res;
  int64_t ldres;
  float* out;
  int64_t ldo;
  int flags;   /* VV_LIN_* */
  const float* wscale;   /* wdt == VV_FP8: out = (W8 x) * wscale[n] (+ bias ...); else ignored */
  const float* w2scale;
} vv_lin_args;

int vv_linear(const vv_lin_args* a, vv_stream_t stream);

/* ------------------------------------------------------------------------------------------------------------
 * Qwen2 attention pieces (third-party math: transformers.models.qwen2.modeling_qwen2; reference call sites
 * vibevoice/modular/modeling_vibevoice.py:187-199, modeling_vibevoice_inference.py:226-237).
 * KV cache layout: k, v = [layers][rows][kv_heads][s_max][head_dim], dtype kvdt.
 * Each of the R query rows carries lens[r] (device int: its absolute position == number of tokens cached before
 * it) and cache_rows[r] (device int or NULL = r): decode uses rows {positive, negative}; prefill uses R prompt
 * tokens that all append to cache row 0 with lens = pos0 + r.
 * vv_rope_store: RoPE (half rotation, fp32 cos/sin from inv_freq) on q and k of qkv[R, (heads+2 kv_heads)*d]
 *   in place, and store k, v at slot lens[r] of the cache.
 * vv_attn: out[r, h*d ..] = softmax(q k^T / sqrt(d)) v over slots 0..lens[r] (inclusive), GQA, fp32 softmax.
 * ------------------------------------------------------------------------------------------------------------ */
typedef struct vv_kv {
  void* k;
  void* v;
  int kvdt, layers, rows, kv_heads, s_max, head_dim;
  void* vt;   /* optional transposed value cache in 32-key tiles, [layers][rows][kv_heads][s_max / 32][head_dim][32] (same dtype, same size as v):
                 element (d, s) of a head lives at (s / 32) * 32 * head_dim + d * 32 + s % 32, so one tile is 8 KB of contiguous memory whose rows are
                 the A-operand fragments of the matrix-core P.V product.  vv_rope_store and vv_attn_decode keep it in step with v; with it (bf16,
                 head_dim 128, s_max % 32 == 0) prompt-sized vv_attn calls and split-key decode steps run both attention products on the matrix
                 cores without a transpose; NULL: the VALU kernels.  Need not be zero-initialised. */
} vv_kv;

/* rope_table[R][head_dim/2][2] = {cos, sin}(lens[r] * inv_freq[i]): computed once per step, shared by all layers */
int vv_rope_table(const int* lens, const float* inv_freq, int R, int head_dim, float* rope_table, vv_stream_t stream);
int vv_rope_store(float* qkv, int64_t ld_qkv, int R, int heads, const vv_kv* kv, int layer, const float* rope_table,
                  const int* lens, const int* cache_rows, vv_stream_t stream);
int vv_attn(const float* qkv, int64_t ld_qkv, int R, int heads, const vv_kv* kv, int layer, const int* lens,
            const int* cache_rows, float* out, int64_t ldo, vv_stream_t stream);
/* vv_attn_decode: vv_rope_store + vv_attn fused for the per-frame step (row r appends to cache row r): qkv is the raw
 * projection (pre-RoPE); q and the new k are rotated in registers, k/v appended at slot lens[r], attention over 0..lens[r]. */
int vv_attn_decode(const float* qkv, int64_t ld_qkv, int R, int heads, const vv_kv* kv, int layer, const float* rope_table,
                   const int* lens, float* out, int64_t ldo, vv_stream_t stream);

/* ------------------------------------------------------------------------------------------------------------
 * Block1D first half on channels-last data (modular_vibevoice_tokenizer.py:924-932 with the streaming
 * SConv1d of :327-382, depthwise k=7):   out = x + gamma * (dwconv7(RMSNorm_c(x)) + b)
 * hist[6, C] holds the previous 6 NORMALISED rows (the conv's cached input) and is updated in place;
 * hist == NULL means non-streaming (zero left context, nothing stored).
 * ------------------------------------------------------------------------------------------------------------ */
int vv_block_mixer(const float* x, float* out, int T, int C, const float* norm_w, float eps, const float* dw_w,
                   const float* dw_b, const float* gamma, float* hist, vv_stream_t stream);

/* Streaming context for the dense convs: pad[0:ctx] <- state; state <- last ctx rows of [state ; pad[ctx:ctx+T]].
 * (SConv1d :364-380 keeps the last ctx inputs; SConvTranspose1d :538-547 needs only the previous input, ctx=1.) */
int vv_conv_ctx(float* pad, float* state, int ctx, int T, int C, vv_stream_t stream);

/* small elementwise helpers */
int vv_affine(const float* x, float a, float b, float* out, int64_t n, vv_stream_t stream);            /* out = a*x + b */
int vv_add_rows(const float* a, int64_t lda, const float* b, int64_t ldb, float* out, int rows, int rows_b, int n,
                vv_stream_t stream); /* out[r, :] = a[r / rows_b... see .hip]: c[step*rb + j] = a[j] + b[step]  */
int vv_embed_row(const void* table, int wdt, int64_t hidden, const int* token, float* out, vv_stream_t stream);
int vv_gather_rows(const void* table, int wdt, int64_t hidden, const int* ids_host, int n, void* out, vv_stream_t stream); /* ids (host) */
/* token = ids[argmax logits] (first max in ascending id order, as torch.argmax over the masked vocabulary does,
 * modeling_vibevoice_inference.py:53-66,486-496); *forced_token >= 0 (device int, may be NULL) overrides the choice. */
int vv_argmax_ids(const float* logits, int n, const int* ids, int* token_out, const int* forced_token, vv_stream_t stream);
/* out[r, :] = bf16(prologue(x[r, :])) (prologue: VV_PRO_NONE or VV_PRO_RMSNORM with optional weight): the activation operand of a
 * many-row matrix-core GEMM (vv_linear with VV_LIN_X_BF16), e.g. the prompt prefill */
int vv_cast_rows_bf16(const float* x, int64_t ldx, int rows, int n, int pro, const float* norm_w, float eps, void* out, int64_t ldo,
                      vv_stream_t stream);
int vv_copy_rows(const float* x, int64_t ldx, float* out, int64_t ldo, int rows, int n, vv_stream_t stream); /* ldx may be 0 (broadcast) */
/* one DPM-Solver++ step fused with classifier-free guidance, per latent element
 * (modeling_vibevoice_inference.py:704-707 + vibevoice/schedule/dpm_solver.py:581-584,669-677,738-764):
 *   eps = v_unc + cfg*(v_cond - v_unc); x0 = alpha_s*x - sigma_s*eps;
 *   order 1: x = cx*x - cd*x0;   order 2: x = cx*x - cd*x0 - 0.5*cd*rinv*(x0 - m_prev);   m_prev = x0 */
int vv_dpm_step(const float* v, int64_t ldv, int n_samples, int latent, float cfg_scale, float alpha_s, float sigma_s,
                float cx, float cd, float rinv, int order, float* x, float* m_prev, vv_stream_t stream);
/* positions after a decode step: lens[0] += 1; token == tok_start: lens[1] = 0 (negative branch refreshed, modeling_vibevoice_inference.py:547-563);
 * token == tok_diffusion: lens[1] += 1 (the negative step is committed, :575-587), *frame_counter += 1.  tok_start < 0 selects
 * refresh_negative=False (:501-515): lens[1] += 1 for every token, never reset. */
int vv_advance_lens(int* lens, const int* token, int tok_start, int tok_diffusion, int* frame_counter, vv_stream_t stream);

/* ------------------------------------------------------------------------------------------------------------
 * Composite operators (one call enqueues the whole launch sequence of a component).
 * ------------------------------------------------------------------------------------------------------------ */
/* Optional weight-only fp8 companion of a matrix for the batch-1/2 decode GEMVs: e4m3fn bytes [N, K] and one
 * fp32 scale per output row.  q == NULL: not quantised.  The bf16 matrix stays the operand of every GEMM-shaped use (prefill, T > 2). */
typedef struct vv_w8 { const void* q; const float* scale; } vv_w8;

typedef struct vv_llm_layer {
  const float* ln1;  /* input_layernorm.weight */
  const float* ln2;  /* post_attention_layernorm.weight */
  const void* wqkv;  /* rows q | k | v concatenated: [(heads + 2 kv_heads) * d, hidden] */
  const float* bqkv;
  const void* wo;    /* [hidden, heads*d] */
  const void* wgate; /* [inter, hidden] */
  const void* wup;
  const void* wdown; /* [hidden, inter] */
  vv_w8 q_qkv, q_o, q_gate, q_up, q_down;
  /* optional fragment-major copies of the five matrices (NULL: none) for a row-batched decode step (4..8 rows = {positive, negative} x 2..4
   * dialogues, vv_gemv_rows.hip): [N / 16][K / 32][4][16][8], i.e. element (16 g + n, 32 j + 8 c + e) of the row-major matrix at
   * ((g * K/32 + j) * 64 + 16 c + n) * 8 + e - one matrix-core B fragment per 1 KB of contiguous memory.  N % 16 == 0, K % 32 == 0. */
  const void* f_qkv; const void* f_o; const void* f_gate; const void* f_up; const void* f_down;
} vv_llm_layer;

typedef struct vv_llm {
  int wdt, hidden, inter, layers, heads, kv_heads, head_dim;
  float rms_eps;
  const float* inv_freq;       /* [head_dim/2] */
  const float* final_norm;     /* [hidden] */
  const vv_llm_layer* layer;   /* (host) array [layers] */
} vv_llm;

size_t vv_llm_ws_bytes(const vv_llm* m, int R);
/* R rows of input embeddings x[R, hidden] -> final-normed hidden states out[R, hidden]; appends to the KV cache.
 * Replaces Qwen2Model.forward for both the per-frame decode (R = 2: positive + negative branch sharing one weight
 * pass; modeling_vibevoice_inference.py:478-480 and :581-583) and the prompt prefill (R = L0; :478 at step 0). */
int vv_llm_forward(const vv_llm* m, const vv_kv* kv, const float* x, int64_t ldx, int R, const int* lens,
                   const int* cache_rows, float* out, int64_t ldo, void* ws, vv_stream_t stream);
/* out == NULL: the final RMSNorm is left to the caller; the un-normalised last hidden rows then stay at the START of ws as [R, hidden] fp32.
 * vv_llm_tail finishes a decode step in one launch: out[R, hidden] = final RMSNorm(h) (the {condition, negative condition} of the
 * diffusion head), logits[nv] = w_valid[nv, hidden] . out[0] (the constrained vocabulary rows of lm_head, VibeVoiceTokenConstraintProcessor
 * modeling_vibevoice_inference.py:53-66), token = ids[argmax] (first maximum in ascending id order) unless *forced_token >= 0, and the
 * position bookkeeping of vv_advance_lens (lens == NULL: skipped). */
int vv_llm_tail(const vv_llm* m, const float* h, int64_t ldh, int R, float* out, int64_t ldo, const void* w_valid, int nv, const int* ids,
                float* logits_out, int* token_out, const int* forced_token, int* lens, int tok_start, int tok_diffusion, int* frame_counter,
                vv_stream_t stream);

/* One decode step for B dialogues batched into the row dimension: vv_llm_forward with R = 2 B rows (dialogue b = rows 2 b, 2 b + 1 of x, lens and
 * of a KV cache with rows >= 2 B) streams the weights once for all of them; vv_llm_tail_batch then does for every dialogue what vv_llm_tail does
 * for one: h = the un-normalised rows at the start of the LLM workspace, out[2 B, hidden], logits_out[B][8], token_out[B], forced_token[B] or NULL,
 * lens[2 B], frame_counter[B]; active[B] (or NULL = all): a dialogue with active == 0 has finished - its rows are computed and dropped, its
 * positions stay where they are. */
int vv_llm_tail_batch(const vv_llm* m, const float* h, int64_t ldh, int B, float* out, int64_t ldo, const void* w_valid, int nv, const int* ids,
                      float* logits_out, int* token_out, const int* forced_token, int* lens, int tok_start, int tok_diffusion, int* frame_counter,
                      const int* active, vv_stream_t stream);

typedef struct vv_head_layer {
  const float* norm_w;
  const void* wgate;  /* [ffn, D] */
  const void* wup;
  const void* wdown;  /* [D, ffn] */
  const void* adaln;  /* [3D, D]: shift | scale | gate */
  vv_w8 q_gate, q_up, q_down;
  const void* f_gate; const void* f_up; const void* f_down;   /* optional fragment-major copies (see vv_llm_layer) */
} vv_head_layer;

typedef struct vv_head {
  int wdt, D, ffn, layers, latent, cond_dim;
  float eps;
  int flags;                 /* reserved, 0 */
  const void* noisy_proj;    /* [D, latent] */
  const void* cond_proj;     /* [D, cond_dim] */
  const void* final_adaln;   /* [2D, D]: shift | scale */
  const void* final_linear;  /* [latent, D] */
  const vv_head_layer* layer; /* (host) array */
  const float* fused_g;      /* optional [D + latent, D] fp32: [noisy_proj x final_linear ; final_linear].  When set, vv_head_sample runs a solver-step
                                boundary (FinalLayer + CFG + DPM-Solver++ update + next noisy_images_proj: all linear in the modulated hidden state) as
                                ONE GEMV over it with the solver as epilogue; NULL keeps the three-launch form */
} vv_head;

typedef struct vv_dpm_coef { float alpha_s, sigma_s, cx, cd, rinv; int order; float cn; /* variance-noise coefficient: 0 for the ODE solver, sigma_t sqrt(1 - e^-2h) for sde-dpmsolver++ */ } vv_dpm_coef;

/* out[(i*rows_b + j), :] = silu(a[j, :] + b[i, :]): the adaLN input silu(cond_proj(cond) + t_emb(t_i)) for all steps at once */
int vv_add_rows_silu(const float* a, int64_t lda, const float* b, int64_t ldb, float* out, int rows, int rows_b, int n, vv_stream_t stream);
/* same, rounded to bf16 [rows, n] (the VV_LIN_X_BF16 operand of the hoisted adaLN GEMMs) */
int vv_add_rows_silu_bf16(const float* a, int64_t lda, const float* b, int64_t ldb, void* out, int rows, int rows_b, int n, vv_stream_t stream);
/* step boundary of the sampler in one launch: x_out = DPM-Solver++/CFG update of (v, x_in, m_in) [v == NULL: x_out = x_in],
 * m_out = x0 prediction, and h[r, :] = W x_out for r < rows (the next step's noisy_images_proj; W == NULL: skipped).
 * x_in/x_out and m_in/m_out must be distinct buffers. */
int vv_dpm_proj(const float* v, int64_t ldv, float cfg_scale, const vv_dpm_coef* coef, const float* x_in, const float* m_in,
                float* x_out, float* m_out, const void* w, int wdt, int latent, int D, float* h, int64_t ldh, int rows,
                const float* step_noise /* [latent] variance noise of this step (coef->cn != 0) or NULL */, vv_stream_t stream);

size_t vv_head_ws_bytes(const vv_head* h, int n_steps);
/* sample_speech_tokens for ONE utterance (modeling_vibevoice_inference.py:695-708) with
 * VibeVoiceDiffusionHead.forward (modular_vibevoice_diffusion_head.py:254-280) and the DPM-Solver++ step
 * (vibevoice/schedule/dpm_solver.py:935-1022) fused into one launch sequence:
 *   cond2[2, cond_dim] = {positive, negative} LLM hidden states; noise[latent]; temb[n_steps, D] = t_embedder(t_i)
 *   (step-invariant, precomputed by the host with vv_linear); coef (host) per step.  latent_out[latent].
 * cond_proj and every adaLN modulation are hoisted out of the step loop (they do not depend on x). */
int vv_head_sample(const vv_head* h, const float* cond2, int64_t ld_cond, const float* noise, const float* temb,
                   const vv_dpm_coef* coef, int n_steps, float cfg_scale, float* latent_out, void* ws,
                   const float* sde_noise /* [n_steps, latent] per-step variance noise of the SDE solver (dpm_solver.py:993-998) or NULL */,
                   vv_stream_t stream);
/* sample_speech_tokens for B utterances at once (B <= 4; the ODE solver on the fused boundary, bf16 weights): cond[2 B, cond_dim] = rows
 * {positive, negative} of utterance b at 2 b, 2 b + 1; noise[b * ld_noise ..], latent_out[b * ld_latent ..].  Every head matrix is streamed once
 * per solver step for all utterances.  ws: vv_head_ws_bytes_batch(h, n_steps, B) bytes. */
size_t vv_head_ws_bytes_batch(const vv_head* h, int n_steps, int B);
int vv_head_sample_batch(const vv_head* h, const float* cond, int64_t ld_cond, const float* noise, int64_t ld_noise, const float* temb,
                         const vv_dpm_coef* coef, int n_steps, float cfg_scale, float* latent_out, int64_t ld_latent, int B, void* ws,
                         vv_stream_t stream);
/* VibeVoiceDiffusionHead.forward alone (parity tests): x[R, latent], temb_rows[R, D], cond[R, cond_dim] -> v[R, latent] */
int vv_head_forward(const vv_head* h, const float* x, const float* temb_rows, const float* cond, int R, float* v,
                    void* ws, vv_stream_t stream);

typedef struct vv_block {
  const float* gamma; const float* ffn_gamma; const float* norm_w; const float* ffn_norm_w;
  const float* dw_w;  /* [C, 7] */
  const float* dw_b;
  const void* w1; const float* b1;  /* [4C, C] */
  const void* w2; const float* b2;  /* [C, 4C] */
  float* hist;                      /* [6, C] streaming state or NULL */
  vv_w8 q_w1, q_w2;                 /* used when the block runs as GEMVs (T <= 2 rows: stage 0 of a streaming frame) */
  const float* dw_last;             /* optional [C]: dw_w[:, 6] packed (the tap of the newest row); with hs: one-row frames skip the 7-row window */
  float* hs;                        /* optional [C] streaming state next to hist: sum_k<6 dw_w[c, k] * hist[k, c], kept in step with hist by
                                       the composites (zero when hist is zero); NULL = recompute from hist every frame */
} vv_block;

/* One whole Block1D (mixer + FFN, vibevoice/modular/modular_vibevoice_tokenizer.py:555-600) as a single launch: x[T, C] -> out[T, C]
 * (out != x).  Covered: bf16 weights, C = 32 / 64 / 128, T >= 32; anything else returns VV_E_UNSUPPORTED (the composites then
 * run vv_block_mixer + two vv_linear).  b->hist (or NULL) is the streaming state, updated in place. */
int vv_block1d(const struct vv_block* b, int wdt, const float* x, float* out, int T, int C, float eps, vv_stream_t stream);

/* The same Block1D for the middle stages of a streaming frame (bf16 weights; C = 256 / 512 with 3 <= T <= 256, C = 1024 with T <= 64,
 * C = 128 with T <= 1024) as two launches (mixer +
 * first FFN GEMM, second FFN GEMM): x[T, C] -> out[T, C] (out != x).  ws: vv_block_mid_ws_bytes(T, C) bytes of scratch, 16-byte aligned.
 * Other shapes return VV_E_UNSUPPORTED.  b->hist as above. */
size_t vv_block_mid_ws_bytes(int T, int C);
int vv_block_mid(const struct vv_block* b, int wdt, const float* x, float* out, void* ws, int T, int C, float eps, vv_stream_t stream);

typedef struct vv_conv {
  const void* w;      /* re-laid: SConv1d [cout, kk*cin] with k index = tap*cin + ci;
                         SConvTranspose1d [s*cout, 2*cin] with row = r*cout + co, k = j*cin + ci (j=0: previous input) */
  const float* b;     /* SConv1d [cout];  SConvTranspose1d [s*cout] (bias tiled over r) */
  int cin, cout, kk, stride, transposed;
  float* state;       /* [ctx, cin] streaming state or NULL; ctx = kk - stride (conv) or 1 (transposed) */
} vv_conv;

#define VV_MAX_STAGES 8
typedef struct vv_convnet {
  int wdt, n_stages;
  float eps;
  vv_conv sample[VV_MAX_STAGES];          /* stem + up/down-sampling convs, one per stage */
  int n_blocks[VV_MAX_STAGES];
  const vv_block* blocks[VV_MAX_STAGES];  /* (host) arrays */
  vv_conv head;
} vv_convnet;

size_t vv_convnet_ws_bytes(const vv_convnet* net, int64_t t_in, int decoder);
/* TokenizerDecoder.forward (modular_vibevoice_tokenizer.py:914-951): latent[T, vae_dim] -> wav[T*hop] (streaming
 * when the net carries state buffers).  pre_scale/pre_bias fold `latent / scale - bias`
 * (modeling_vibevoice_inference.py:634) into the stem input: x = pre_scale*latent + pre_bias. */
int vv_decoder_forward(const vv_convnet* net, const float* latent, int T, float pre_scale, float pre_bias, float* wav,
                       void* ws, vv_stream_t stream);
/* TokenizerEncoder.forward (:776-813): wav[T] -> feat[ceil(T/hop), vae_dim]; streaming (T multiple of hop, net has
 * state) or whole-utterance non-streaming with the reference's zero left/right padding (:384-418). */
int vv_encoder_forward(const vv_convnet* net, const float* wav, int64_t T, float* feat, void* ws, vv_stream_t stream);
/* zero every state buffer of the net == VibeVoiceTokenizerStreamingCache.set_to_zero (:234-241) */
int vv_convnet_reset(const vv_convnet* net, vv_stream_t stream);

typedef struct vv_connector { int wdt, din, hidden; const void* fc1; const float* b1; const float* norm_w; const void* fc2; const float* b2; } vv_connector;
/* SpeechConnector.forward (modeling_vibevoice.py:58-69): out[R,hidden] (+)= fc2(RMSNorm(fc1 x)); accumulate != 0 adds
 * into `out` (acoustic + semantic sum, modeling_vibevoice_inference.py:665-667).  ws: R*hidden floats. */
int vv_connector_forward(const vv_connector* c, const float* x, int R, float* out, int accumulate, float* ws, vv_stream_t stream);
/* Both connectors of one generated frame: out[r, :] = acoustic_connector(latent) + semantic_connector(semfeat) for r < rows_out
 * (modeling_vibevoice_inference.py:665-670; the positive and the negative branch consume the same embedding, :575-579).  Two launches
 * (fc1 of both, fc2 of both with the RMSNorms in the prologue) when the weights are bf16, else the two connectors in turn.  ws: 2 * hidden floats. */
int vv_connector_pair(const vv_connector* ac, const vv_connector* sem, const float* latent, const float* semfeat, float* out, int64_t ldo,
                      int rows_out, float* ws, vv_stream_t stream);

/* ------------------------------------------------------------------------------------------------------------
 * hipGraph capture of whatever the caller enqueues between begin/end on `stream`.
 * ------------------------------------------------------------------------------------------------------------ */
int vv_graph_begin(vv_stream_t stream);
int vv_graph_end(vv_stream_t stream, void** graph_exec_out);
int vv_graph_launch(void* graph_exec, vv_stream_t stream);
int vv_graph_destroy(void* graph_exec);

/* ------------------------------------------------------------------------------------------------------------
 * Per-launch timing of vv_linear with HIP events recorded on the launch stream (eager mode only, never during
 * graph capture).  vv_prof_begin arms it; vv_prof_end synchronises and aggregates by (m, n, k, dual, wdt).
 * ------------------------------------------------------------------------------------------------------------ */
typedef struct vv_prof_entry { int m, n, k, dual, wdt, count; double total_ms; } vv_prof_entry;
int vv_prof_begin(int max_records);
int vv_prof_end(vv_prof_entry* out, int max_out, int* n_out);

/* struct sizes, for the ctypes mirror's self-check */
size_t vv_sizeof(const char* struct_name);

#ifdef __cplusplus
}
#endif
#endif /* VV_HIP_H */
