// vv_fused.hip — launch-count cuts on the per-frame latency chain (a dependent launch costs >= 1.6 us even when it is trivial, 4-5 us
// once it has to pull its predecessor's output through L2: tools/mb_chain.cpp).
//
//   head_boundary_kernel  one launch per solver step for  FinalLayer (RMSNorm + modulate + linear D -> latent)  +  CFG  +  the
//                         DPM-Solver++ update  +  the next step's noisy_images_proj (latent -> D).  Everything after the modulate is
//                         LINEAR in y, so with  G = [P F ; F]  (P = noisy_images_proj [D, latent], F = final linear [latent, D]; G is
//                         (D + latent) x D, fp32, built once at load time) the state  X = [P x ; x]  obeys
//                             z = G (y_u + s (y_c - y_u)),   x0 = alpha_s X - sigma_s z,   X' = cx X - cd x0 - cd/2 rinv (x0 - M),  M' = x0
//                         element by element: one M = 1 GEMV over G with the solver as its epilogue replaces a GEMV (N = latent), the
//                         update kernel and a second tiny GEMV (reference: modular_vibevoice_diffusion_head.py:184-188,272-279,
//                         modeling_vibevoice_inference.py:704-707, schedule/dpm_solver.py:581-584,669-677,738-764).
//   head_init_kernel      X0 = [P noise ; noise], M = 0, both rows of the first step's hidden state.
//   llm_tail_kernel       final RMSNorm of the decode rows + the 4-5 constrained logits + argmax / forced token + position bookkeeping
//                         (four launches -> one; reference: modeling_vibevoice_inference.py:53-66,241-242,486-499).
//   conv_ctx_gather / conv_ctx_scatter   the left-context handling of ALL streaming convs of a tokenizer in two launches per frame
//                         instead of one per conv (modular_vibevoice_tokenizer.py:364-380,538-547).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "vv_hip.h"
#include "vv_common.h"

namespace {

typedef unsigned short bf16_t;
__device__ __forceinline__ float bf2f(bf16_t v) { return __uint_as_float((unsigned)v << 16); }
__device__ __forceinline__ float ldw(const float* p) { return *p; }
__device__ __forceinline__ float ldw(const bf16_t* p) { return bf2f(*p); }

// ---------------------------------------------------------------------------------------------------------------
// diffusion head: solver-step boundary
// ---------------------------------------------------------------------------------------------------------------
template <typename WT>
__global__ __launch_bounds__(256) void head_init_kernel(const WT* P, const float* noise, int D, int latent, float* Xs, float* Ms, float* h0, int64_t ldh) {
  extern __shared__ float nz[];
  for (int i = threadIdx.x; i < latent; i += blockDim.x) nz[i] = noise[i];
  __syncthreads();
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= D + latent) return;
  float s;
  if (n < D) {
    s = 0.f;
    const WT* pr = P + (int64_t)n * latent;
    for (int j = 0; j < latent; ++j) s = fmaf(ldw(pr + j), nz[j], s);
    h0[n] = s;
    h0[ldh + n] = s;
  } else {
    s = nz[n - D];
  }
  Xs[n] = s;
  Ms[n] = 0.f;
}

struct BoundaryArgs {
  const float* G;          // [D + latent, D] fp32
  const float* h; int64_t ldh;                      // [2, D] hidden rows after the last head layer {cond, uncond}
  const float* shift; const float* scale; int64_t ld_mod;   // final adaLN rows of this step: row r at + r * ld_mod
  float eps, cfg;
  vv_dpm_coef k;
  float* Xs; float* Ms;    // [D + latent] solver state (P x ; x) and previous x0 prediction, updated in place (one owner per element)
  float* h_out; int64_t ldh_out;                    // [2, D] next step's hidden rows (both = P x'), a DIFFERENT buffer than h
  float* latent_out;       // [latent] x' (the last step's value is the sample)
  int D, latent;
};

template <int KU>
__global__ __launch_bounds__(256) void head_boundary_kernel(const BoundaryArgs a) {
  __shared__ __attribute__((aligned(16))) float ys[KU * 512];
  __shared__ float red[4][2];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int D = a.D, N = a.D + a.latent;
  constexpr int NCH = (KU * 128 + 255) / 256;
  const int nchunks = D >> 2;
  // activation side first (see gemv_stream_kernel): h rows, shift / scale rows of both branches
  float4 hv[2][NCH], sv[2][NCH], cv[2][NCH];
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    const int ch = tid + c * 256;
    const int kk = ch < nchunks ? ch * 4 : 0;
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      hv[r][c] = *reinterpret_cast<const float4*>(a.h + r * a.ldh + kk);
      sv[r][c] = *reinterpret_cast<const float4*>(a.shift + r * a.ld_mod + kk);
      cv[r][c] = *reinterpret_cast<const float4*>(a.scale + r * a.ld_mod + kk);
    }
  }
  const int gstride = gridDim.x * 4;
  int g = blockIdx.x * 4 + wave;
  float4 cur[KU][2], nxt[KU][2];
  auto issue = [&](float4 (&b)[KU][2], int row) {
    const bool live = row < N;
#pragma unroll
    for (int u = 0; u < KU; ++u) {
      const int k0 = u * 512 + lane * 8;
      const int64_t off = (live && k0 < D) ? (int64_t)row * D + k0 : 0;
      b[u][0] = *reinterpret_cast<const float4*>(a.G + off);
      b[u][1] = *reinterpret_cast<const float4*>(a.G + off + 4);
    }
  };
  float xs_cur = a.Xs[g < N ? g : 0], ms_cur = a.Ms[g < N ? g : 0];
  issue(cur, g);
  float xs_nxt = a.Xs[g + gstride < N ? g + gstride : 0], ms_nxt = a.Ms[g + gstride < N ? g + gstride : 0];
  issue(nxt, g + gstride);
  __builtin_amdgcn_sched_barrier(0);
#define VV_FENCE4(v) asm volatile("" : "+v"((v).x), "+v"((v).y), "+v"((v).z), "+v"((v).w))
#pragma unroll
  for (int c = 0; c < NCH; ++c)
#pragma unroll
    for (int r = 0; r < 2; ++r) { VV_FENCE4(hv[r][c]); VV_FENCE4(sv[r][c]); VV_FENCE4(cv[r][c]); }
#undef VV_FENCE4
  float ss[2];
#pragma unroll
  for (int r = 0; r < 2; ++r) {
    float s1 = 0.f;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      const float4 v = hv[r][c];
      s1 += (tid + c * 256 < nchunks) ? (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w) : 0.f;
    }
    ss[r] = vv_wave_sum(s1);
  }
  if (lane == 0) { red[wave][0] = ss[0]; red[wave][1] = ss[1]; }
  __syncthreads();
  {
    const float rs0 = rsqrtf(((red[0][0] + red[1][0]) + (red[2][0] + red[3][0])) / (float)D + a.eps);
    const float rs1 = rsqrtf(((red[0][1] + red[1][1]) + (red[2][1] + red[3][1])) / (float)D + a.eps);
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      const int ch = tid + c * 256;
      if (ch < KU * 128) {
        float4 y = make_float4(0.f, 0.f, 0.f, 0.f);
        if (ch < nchunks) {
          const float4 h0 = hv[0][c], h1 = hv[1][c];
          // FinalLayer: modulate(norm_final(h), shift, scale) per branch, then classifier-free guidance on the (linear) remainder
          const float yc0 = h0.x * rs0 * (1.0f + cv[0][c].x) + sv[0][c].x, yu0 = h1.x * rs1 * (1.0f + cv[1][c].x) + sv[1][c].x;
          const float yc1 = h0.y * rs0 * (1.0f + cv[0][c].y) + sv[0][c].y, yu1 = h1.y * rs1 * (1.0f + cv[1][c].y) + sv[1][c].y;
          const float yc2 = h0.z * rs0 * (1.0f + cv[0][c].z) + sv[0][c].z, yu2 = h1.z * rs1 * (1.0f + cv[1][c].z) + sv[1][c].z;
          const float yc3 = h0.w * rs0 * (1.0f + cv[0][c].w) + sv[0][c].w, yu3 = h1.w * rs1 * (1.0f + cv[1][c].w) + sv[1][c].w;
          y = make_float4(yu0 + a.cfg * (yc0 - yu0), yu1 + a.cfg * (yc1 - yu1), yu2 + a.cfg * (yc2 - yu2), yu3 + a.cfg * (yc3 - yu3));
        }
        *reinterpret_cast<float4*>(&ys[ch * 4]) = y;
      }
    }
  }
  __syncthreads();
  float yr[KU][8];
#pragma unroll
  for (int u = 0; u < KU; ++u) {
    const float4 p = *reinterpret_cast<const float4*>(&ys[u * 512 + lane * 8]);
    const float4 q = *reinterpret_cast<const float4*>(&ys[u * 512 + lane * 8 + 4]);
    yr[u][0] = p.x; yr[u][1] = p.y; yr[u][2] = p.z; yr[u][3] = p.w; yr[u][4] = q.x; yr[u][5] = q.y; yr[u][6] = q.z; yr[u][7] = q.w;
  }
  while (g < N) {
    float acc = 0.f;
#pragma unroll
    for (int u = 0; u < KU; ++u) {
      const float w[8] = {cur[u][0].x, cur[u][0].y, cur[u][0].z, cur[u][0].w, cur[u][1].x, cur[u][1].y, cur[u][1].z, cur[u][1].w};
#pragma unroll
      for (int j = 0; j < 8; ++j) acc = fmaf(w[j], yr[u][j], acc);
    }
    const float z = vv_wave_sum(acc);
    if (lane == 0) {
      const float x0 = a.k.alpha_s * xs_cur - a.k.sigma_s * z;
      float xt = a.k.cx * xs_cur - a.k.cd * x0;
      if (a.k.order == 2) xt -= 0.5f * a.k.cd * (a.k.rinv * (x0 - ms_cur));
      a.Xs[g] = xt;
      a.Ms[g] = x0;
      if (g < D) { a.h_out[g] = xt; a.h_out[a.ldh_out + g] = xt; }
      else a.latent_out[g - D] = xt;
    }
    g += gstride;
#pragma unroll
    for (int u = 0; u < KU; ++u) { cur[u][0] = nxt[u][0]; cur[u][1] = nxt[u][1]; }
    xs_cur = xs_nxt; ms_cur = ms_nxt;
    if (g + gstride < N) { xs_nxt = a.Xs[g + gstride]; ms_nxt = a.Ms[g + gstride]; issue(nxt, g + gstride); }
  }
}

template <int KU> void launch_boundary(const BoundaryArgs& a, hipStream_t s) {
  const int n = a.D + a.latent;
  int blocks = (n + 3) / 4;
  if (blocks > 512) blocks = 512;
  hipLaunchKernelGGL((head_boundary_kernel<KU>), dim3(blocks), dim3(256), 0, s, a);
}

// ---------------------------------------------------------------------------------------------------------------
// LLM step tail
// ---------------------------------------------------------------------------------------------------------------
template <typename WT>
__global__ __launch_bounds__(256) void llm_tail_kernel(const float* h, int64_t ldh, int R, int H, const float* norm_w, float eps, float* out, int64_t ldo,
                                                       const WT* w_valid, int nv, const int* ids, float* logits_out, int* token_out, const int* forced,
                                                       int* lens, int tok_start, int tok_diff, int* frame_ctr) {
  extern __shared__ float hn0[];          // [H] normalised row 0 (the positive branch: the only row logits are taken from)
  __shared__ float red[4];
  __shared__ float lg[16];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int r = 0; r < R; ++r) {
    const float* xr = h + (int64_t)r * ldh;
    float s = 0.f;
    for (int c = tid; c < H; c += 256) { const float v = xr[c]; s = fmaf(v, v, s); }
    s = vv_wave_sum(s);
    __syncthreads();
    if (lane == 0) red[wave] = s;
    __syncthreads();
    const float rstd = rsqrtf(((red[0] + red[1]) + (red[2] + red[3])) / (float)H + eps);
    for (int c = tid; c < H; c += 256) {
      const float v = xr[c] * rstd * norm_w[c];
      out[(int64_t)r * ldo + c] = v;
      if (r == 0) hn0[c] = v;
    }
  }
  __syncthreads();
  for (int i = wave; i < nv; i += 4) {                // one wave per constrained vocabulary row
    const WT* wr = w_valid + (int64_t)i * H;
    float s = 0.f;
    for (int c = lane; c < H; c += 64) s = fmaf(ldw(wr + c), hn0[c], s);
    s = vv_wave_sum(s);
    if (lane == 0) { lg[i] = s; logits_out[i] = s; }
  }
  __syncthreads();
  if (tid == 0) {
    int best = 0;
    for (int i = 1; i < nv; ++i)
      if (lg[i] > lg[best] || (lg[i] == lg[best] && ids[i] < ids[best])) best = i;     // first maximum in ascending id order, as torch.argmax over the masked vocabulary
    const int f = forced ? *forced : -1;
    const int t = f >= 0 ? f : ids[best];
    *token_out = t;
    if (lens) {
      lens[0] += 1;
      if (t == tok_start) lens[1] = 0;
      else if (t == tok_diff) { lens[1] += 1; if (frame_ctr) *frame_ctr += 1; }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------
// streaming conv contexts of a whole tokenizer
// ---------------------------------------------------------------------------------------------------------------
struct CtxItem { float* pad; float* state; int ctx, T, C; };
struct CtxArgs { CtxItem it[VV_MAX_STAGES + 1]; int n; };

// gather: pad[0 : ctx] <- state for every conv (before any of them runs)
__global__ __launch_bounds__(256) void conv_ctx_gather_kernel(const CtxArgs a) {
  const CtxItem it = a.it[blockIdx.y];
  const int n = it.ctx * it.C;
  for (int i = (blockIdx.x * 256 + threadIdx.x) * 4; i < n; i += gridDim.x * 1024) {
    if (i + 3 < n && (n & 3) == 0) *reinterpret_cast<float4*>(it.pad + i) = *reinterpret_cast<const float4*>(it.state + i);
    else for (int j = i; j < n && j < i + 4; ++j) it.pad[j] = it.state[j];
  }
}
// scatter: state <- last ctx rows of [state ; T new rows] = pad rows [T, T + ctx) (after every conv has consumed its input)
__global__ __launch_bounds__(256) void conv_ctx_scatter_kernel(const CtxArgs a) {
  const CtxItem it = a.it[blockIdx.y];
  const int n = it.ctx * it.C;
  const float* src = it.pad + (int64_t)it.T * it.C;
  for (int i = (blockIdx.x * 256 + threadIdx.x) * 4; i < n; i += gridDim.x * 1024) {
    if (i + 3 < n && (n & 3) == 0 && (((int64_t)it.T * it.C) & 3) == 0) *reinterpret_cast<float4*>(it.state + i) = *reinterpret_cast<const float4*>(src + i);
    else for (int j = i; j < n && j < i + 4; ++j) it.state[j] = src[j];
  }
}

}  // namespace

// internal entry points (vv_common.h)
int vv_head_init_fused(const vv_head* h, const float* noise, float* Xs, float* Ms, float* h0, int64_t ldh, hipStream_t s) {
  const int n = h->D + h->latent;
  const int blocks = (n + 255) / 256;
  const size_t lds = (size_t)h->latent * sizeof(float);
  if (h->wdt == VV_F32) hipLaunchKernelGGL((head_init_kernel<float>), dim3(blocks), dim3(256), lds, s, (const float*)h->noisy_proj, noise, h->D, h->latent, Xs, Ms, h0, ldh);
  else hipLaunchKernelGGL((head_init_kernel<bf16_t>), dim3(blocks), dim3(256), lds, s, (const bf16_t*)h->noisy_proj, noise, h->D, h->latent, Xs, Ms, h0, ldh);
  VV_CHECK_LAUNCH("vv_head_init_fused");
  return 0;
}

bool vv_head_boundary_supported(const vv_head* h) {
  return h->fused_g != nullptr && h->D % 8 == 0 && h->D <= 4096 && ((uintptr_t)h->fused_g % 16 == 0);
}

int vv_head_boundary_fused(const vv_head* h, const float* hrows, int64_t ldh, const float* shift, const float* scale, int64_t ld_mod, float cfg,
                           const vv_dpm_coef* k, float* Xs, float* Ms, float* h_out, int64_t ldh_out, float* latent_out, hipStream_t s) {
  BoundaryArgs a;
  a.G = h->fused_g; a.h = hrows; a.ldh = ldh; a.shift = shift; a.scale = scale; a.ld_mod = ld_mod; a.eps = h->eps; a.cfg = cfg; a.k = *k;
  a.Xs = Xs; a.Ms = Ms; a.h_out = h_out; a.ldh_out = ldh_out; a.latent_out = latent_out; a.D = h->D; a.latent = h->latent;
  switch ((h->D + 511) / 512) {
    case 1: launch_boundary<1>(a, s); break;
    case 2: launch_boundary<2>(a, s); break;
    case 3: launch_boundary<3>(a, s); break;
    case 4: launch_boundary<4>(a, s); break;
    case 5: launch_boundary<5>(a, s); break;
    case 6: launch_boundary<6>(a, s); break;
    case 7: launch_boundary<7>(a, s); break;
    case 8: launch_boundary<8>(a, s); break;
    default: return vv_set_error(VV_E_UNSUPPORTED, "vv_head_boundary_fused: D=%d", h->D);
  }
  VV_CHECK_LAUNCH("vv_head_boundary_fused");
  return 0;
}

extern "C" int vv_llm_tail(const vv_llm* m, const float* h, int64_t ldh, int R, float* out, int64_t ldo, const void* w_valid, int nv, const int* ids,
                           float* logits_out, int* token_out, const int* forced_token, int* lens, int tok_start, int tok_diffusion,
                           int* frame_counter, vv_stream_t stream) {
  if (!m || !h || !out || !w_valid || !ids || !logits_out || !token_out) return vv_set_error(VV_E_ARG, "vv_llm_tail: null pointer");
  if (R <= 0 || nv <= 0 || nv > 16) return vv_set_error(VV_E_ARG, "vv_llm_tail: R=%d nv=%d (nv <= 16)", R, nv);
  hipStream_t s = (hipStream_t)stream;
  const size_t lds = (size_t)m->hidden * sizeof(float);
  if (m->wdt == VV_F32)
    hipLaunchKernelGGL((llm_tail_kernel<float>), dim3(1), dim3(256), lds, s, h, ldh, R, m->hidden, m->final_norm, m->rms_eps, out, ldo, (const float*)w_valid, nv, ids,
                       logits_out, token_out, forced_token, lens, tok_start, tok_diffusion, frame_counter);
  else
    hipLaunchKernelGGL((llm_tail_kernel<bf16_t>), dim3(1), dim3(256), lds, s, h, ldh, R, m->hidden, m->final_norm, m->rms_eps, out, ldo, (const bf16_t*)w_valid, nv, ids,
                       logits_out, token_out, forced_token, lens, tok_start, tok_diffusion, frame_counter);
  VV_CHECK_LAUNCH("vv_llm_tail");
  return 0;
}

int vv_conv_ctx_batch(const vv_conv_ctx_item* items, int n, int scatter, hipStream_t s) {
  if (n <= 0) return 0;
  if (n > VV_MAX_STAGES + 1) return vv_set_error(VV_E_ARG, "vv_conv_ctx_batch: too many convs");
  CtxArgs a;
  a.n = n;
  int mx = 0;
  for (int i = 0; i < n; ++i) {
    a.it[i].pad = items[i].pad; a.it[i].state = items[i].state; a.it[i].ctx = items[i].ctx; a.it[i].T = items[i].T; a.it[i].C = items[i].C;
    if (items[i].ctx * items[i].C > mx) mx = items[i].ctx * items[i].C;
  }
  int bx = (mx + 1023) / 1024;
  if (bx > 16) bx = 16;
  if (scatter) hipLaunchKernelGGL(conv_ctx_scatter_kernel, dim3(bx, n), dim3(256), 0, s, a);
  else hipLaunchKernelGGL(conv_ctx_gather_kernel, dim3(bx, n), dim3(256), 0, s, a);
  VV_CHECK_LAUNCH("vv_conv_ctx_batch");
  return 0;
}
