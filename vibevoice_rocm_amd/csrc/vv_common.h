// internal helpers shared by the .hip translation units (not part of the C ABI)
#ifndef VV_COMMON_H
#define VV_COMMON_H
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdint.h>
#include "vv_hip.h"

int vv_set_error(int code, const char* fmt, ...);

#define VV_CHECK_LAUNCH(name)                                                                    \
  do {                                                                                           \
    hipError_t e_ = hipGetLastError();                                                           \
    if (e_ != hipSuccess) return vv_set_error(VV_E_HIP, "%s: %s", name, hipGetErrorString(e_));  \
  } while (0)

#define VV_TRY(expr)            \
  do {                          \
    int rc_ = (expr);           \
    if (rc_) return rc_;        \
  } while (0)

int vv_launch_gemv_stream(const vv_lin_args& a, hipStream_t s);   // vv_gemv_stream.hip: 1 = launched, 0 = not covered
// vv_gemv_rows.hip: 3..8 activation rows on the matrix cores; 1 launched, 0 not covered, < 0 error.  part / tickets: split-K workspace
// (vv_gemv_rows_part_floats / vv_gemv_rows_tickets give the sizes; tickets zero on entry, left zero) or null
int vv_launch_gemv_rows(const vv_lin_args& a, float* part, size_t part_floats, int* tickets, size_t n_tickets, hipStream_t s);
size_t vv_gemv_rows_part_floats(int n, int dual);
size_t vv_gemv_rows_tickets(int n);
int vv_gemv_rows_init();
void vv_gemv_rows_set(int on, int blocks, int pers);            // tuning hooks (negative / zero: keep)
int vv_linear_ws(const vv_lin_args* a, const void* f1, const void* f2, float* part, size_t part_floats, int* tickets, size_t n_tickets,
                 vv_stream_t stream);   // vv_kernels.hip: f1 / f2 = fragment-major copies of a->w / a->w2 or null
int vv_launch_mfma_gemm(const vv_lin_args& a, hipStream_t s);     // vv_mfma_gemm.hip: 1 launched, 0 not covered, <0 error
int vv_mfma_gemm_init();
// vv_attn_decode.hip: bf16 KV cache, head_dim 128; part / tickets = split-key workspace ([R, heads, nsplit, 130] floats, [R, heads] zeroed ints) or null
// part_cap: splits the partials workspace has room for (>= nsplit); the grouped kernel may use more splits than the per-head kernel's nsplit
int vv_launch_attn_decode(const float* qkv, int64_t ld_qkv, int R, int heads, const vv_kv* kv, int layer, const float2* rope, const int* lens, float* out,
                          int64_t ldo, float* part, int* tickets, int nsplit, int part_cap, hipStream_t s);
int vv_attn_decode_ws(const float* qkv, int64_t ld_qkv, int R, int heads, const vv_kv* kv, int layer, const float* rope_table, const int* lens, float* out,
                      int64_t ldo, float* part, int* tickets, int nsplit, int part_cap, vv_stream_t stream);
// vv_attn_prefill.hip: matrix-core prompt attention (bf16 cache + kv->vt, head_dim 128); 1 launched, 0 not covered, < 0 error
int vv_launch_attn_prefill(const float* qkv, int64_t ld_qkv, int R, int heads, const vv_kv* kv, int layer, const int* lens, const int* cache_rows,
                           float* out, int64_t ldo, hipStream_t s);
// vv_fused.hip
int vv_head_init_fused(const vv_head* h, const float* noise, float* Xs, float* Ms, float* h0, int64_t ldh, hipStream_t s);
bool vv_head_boundary_supported(const vv_head* h);
int vv_head_boundary_fused(const vv_head* h, const float* hrows, int64_t ldh, const float* shift, const float* scale, int64_t ld_mod, float cfg,
                           const vv_dpm_coef* k, float* Xs, float* Ms, float* h_out, int64_t ldh_out, float* latent_out, hipStream_t s);
int vv_head_boundary_batch(const vv_head* h, const float* hrows, int64_t ldh, const float* shift, const float* scale, int64_t ld_mod, float cfg,
                           const vv_dpm_coef* k, float* Xs, float* Ms, int64_t state_stride, float* h_out, int64_t ldh_out, float* latent_out,
                           int64_t latent_stride, int B, hipStream_t s);   // B dialogues per launch: rows 2 b, 2 b + 1; state / sample of b at b * stride
int vv_fused_init();
int vv_head_pre_fused(const vv_head* h, const float* cond2, int64_t ld_cond, const float* temb, int n_steps, void* c_bf16, const float* noise,
                      float* Xs, float* Ms, float* h0, int64_t ldh, hipStream_t s);   // cond_proj + silu(c0 + temb) rows (bf16) + solver-state init in one launch; 1 launched, 0 not covered
int vv_launch_connector_pair(const vv_connector* ac, const vv_connector* sem, const float* latent, const float* semfeat, float* out, int64_t ldo, int rows_out,
                             float* ws, hipStream_t s);   // 1 launched, 0 not covered
int vv_head_modulations_fused(const vv_head* h, const void* c_bf16, int rows, float* const* mod, float* modf, hipStream_t s);   // 1 launched, 0 not covered
struct vv_conv_ctx_item { float* pad; float* state; int ctx, T, C; const float* dw_w; float* hs; int affine; float scale, bias; };   // dw_w / hs (scatter only): also hs[c] = sum_k<6 dw_w[c, k] * new state[k, c]; affine (gather only): pad = state * scale + bias (the net's input rides along)
int vv_conv_ctx_batch(const vv_conv_ctx_item* items, int n, int scatter, hipStream_t s);   // scatter 0: pad[0:ctx] <- state; 1: state <- pad[T : T + ctx]
int vv_block1d_init();                                            // vv_block1d.hip
int vv_launch_block1d(const vv_block& B, int wdt, const float* x, float* out, int T, int C, float eps, hipStream_t s);   // 1 launched, 0 not covered
void vv_block1d_set_fused(int on);
void vv_block1d_set_blocks(int b);
int vv_convffn_init();                                            // vv_convffn.hip
int vv_launch_convffn(const vv_block& B, int wdt, const float* x, float* y, void* hidden, float* hist_new, float* out, int T, int C, float eps,
                      hipStream_t s);                            // 1 launched, 0 not covered
void vv_convffn_set(int on);
int vv_launch_ffn_in_row(const vv_block& B, int wdt, const float* x, float* y, float* hidden, float* hist_new, int C, float eps, hipStream_t s);
void vv_convffn_set_t1(int on);
int vv_launch_ffn_in_row_hs(const vv_block& B, int wdt, const float* x, float* y, float* hidden, float* hist_new, int C, float eps, hipStream_t s);
void vv_convffn_set_t1hs(int on);
void vv_convffn_set_rows(int c, int rows);
void vv_convffn_set_c128(int on);
bool vv_convffn_prefers(int wdt, int T, int C);
int vv_launch_skinny(const vv_lin_args& a, hipStream_t s);       // resampling convs of a streaming frame: 1 launched, 0 not covered
void vv_skinny_set(int on, int min_m, int max_m);
int vv_rmsnorm_rows(const float* x, int64_t ldx, const float* w, float eps, int rows, int n, float* out, int64_t ldo, hipStream_t s);

#ifdef __HIPCC__
// GELU (exact-erf form) with erf from Abramowitz & Stegun 7.1.26: |erf error| <= 1.5e-7, one v_exp + one v_rcp + 6 FMA instead
// of libm's erff (~4x the instructions).  The reciprocal is the hardware's v_rcp_f32 (1 ulp): __frcp_rn expands to the IEEE division
// sequence (v_div_scale / v_div_fmas / v_div_fixup + Newton steps, ~10 more instructions per GELU), and these kernels are bound by
// per-CU vector throughput (32 - 64 GELUs per lane per tile: tools/experiments/block1d_x3.hip.inc has the phase timings).  Used only where the result is rounded to bf16 (2^-9 relative) right away - the hidden
// activation between the two FFN GEMMs of a conv block, where every lane evaluates dozens of them and erff was the longest
// phase of the kernel.  fp32 outputs keep erff.
__device__ __forceinline__ float vv_gelu_as(float v) {
  const float x = v * 0.70710678118654752440f, ax = fabsf(x);
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, ax, 1.0f));
  const float poly = t * fmaf(t, fmaf(t, fmaf(t, fmaf(t, 1.061405429f, -1.453152027f), 1.421413741f), -0.284496736f), 0.254829592f);
  const float erf_abs = 1.0f - poly * __expf(-ax * ax);
  return 0.5f * v * (1.0f + copysignf(erf_abs, x));
}

// Wave-wide (64 lanes) sum, result in every lane.  DPP row operations reduce each 16-lane row at VALU speed (4 dependent
// v_add with a DPP operand), the four row sums are combined through readlane.  The usual __shfl_xor butterfly is six
// dependent ds_bpermute round trips (~100+ cycles each): with one or two waves per SIMD, as in the weight-streaming
// kernels, that latency sits on the critical path of every row group.  Fixed summation order: deterministic.
__device__ __forceinline__ float vv_wave_sum(float v) {
#define VV_DPP_ADD(ctrl) v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), ctrl, 0xF, 0xF, true))
  VV_DPP_ADD(0xB1);    // quad_perm [1,0,3,2]: lane ^ 1
  VV_DPP_ADD(0x4E);    // quad_perm [2,3,0,1]: lane ^ 2
  VV_DPP_ADD(0x141);   // row_half_mirror: the other quad of the 8-lane half row
  VV_DPP_ADD(0x140);   // row_mirror: the other half of the 16-lane row
#undef VV_DPP_ADD
  const float r0 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 0));
  const float r1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 16));
  const float r2 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 32));
  const float r3 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 48));
  return (r0 + r1) + (r2 + r3);
}
#endif

#endif
