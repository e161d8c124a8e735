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
  // dialogues batched into one launch (blockIdx.y = b): hidden / modulation rows 2 b, 2 b + 1; solver state and sample of b at these strides
  int64_t state_stride, latent_stride;
};

template <int KU>
__global__ __launch_bounds__(256) void head_boundary_kernel(BoundaryArgs a) {
  {
    const int b = blockIdx.y;
    a.h += 2 * b * a.ldh; a.shift += 2 * b * a.ld_mod; a.scale += 2 * b * a.ld_mod;
    a.Xs += b * a.state_stride; a.Ms += b * a.state_stride; a.h_out += 2 * b * a.ldh_out; a.latent_out += b * a.latent_stride;
  }
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

template <int KU> void launch_boundary(const BoundaryArgs& a, int B, hipStream_t s) {
  const int n = a.D + a.latent;
  int blocks = (n + 3) / 4;
  if (blocks > 512) blocks = 512;
  if (B > 1 && blocks > 256) blocks = 256;         // B dialogues share the chip
  hipLaunchKernelGGL((head_boundary_kernel<KU>), dim3(blocks, B), dim3(256), 0, s, a);
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
      if (tok_start < 0) { lens[1] += 1; if (t == tok_diff && frame_ctr) *frame_ctr += 1; }      // refresh_negative=False: the negative row consumes every token
      else if (t == tok_start) lens[1] = 0;
      else if (t == tok_diff) { lens[1] += 1; if (frame_ctr) *frame_ctr += 1; }
    }
  }
}

// The decode shape (R <= 2 rows, <= 8 constrained ids, H <= 4096): every operand - the rows, the norm weight, this thread's 4-column
// slices of the constrained vocabulary rows, ids, forced token - is requested in one burst, then two barriers: row statistics,
// logit partials.  The general kernel above walks rows and vocabulary rows one after the other (5-6 dependent trips, 21 us).
template <typename WT>
__global__ __launch_bounds__(256) void llm_tail_fast_kernel(const float* h, int64_t ldh, int R, int H, const float* norm_w, float eps, float* out, int64_t ldo,
                                                            const WT* w_valid, int nv, const int* ids, float* logits_out, int* token_out, const int* forced,
                                                            int* lens, int tok_start, int tok_diff, int* frame_ctr, const int* active) {
  {   // dialogues of a row-batched step: block b owns rows 2 b, 2 b + 1 and the b-th token / forced token / counters
    const int b = blockIdx.x;
    h += (int64_t)2 * b * ldh; out += (int64_t)2 * b * ldo; logits_out += 8 * b; token_out += b;
    if (forced) forced += b;
    if (lens) lens += 2 * b;
    if (frame_ctr) frame_ctr += b;
    if (active) active += b;
  }
  constexpr int NC = 4, NV = 8;                    // column chunks (of 4) per thread per row, constrained ids
  __shared__ float red[4][2];
  __shared__ float lgp[4][NV];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int H4 = H >> 2;
  float4 xv[2][NC], nw[NC], wv[NV][NC];
#pragma unroll
  for (int c = 0; c < NC; ++c) {
    const int ch = tid + 256 * c;
    const int k = ch < H4 ? 4 * ch : 0;
#pragma unroll
    for (int r = 0; r < 2; ++r) xv[r][c] = *reinterpret_cast<const float4*>(h + (int64_t)(r < R ? r : 0) * ldh + k);
    nw[c] = *reinterpret_cast<const float4*>(norm_w + k);
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const WT* wr = w_valid + (int64_t)(i < nv ? i : 0) * H + k;
      if constexpr (sizeof(WT) == 2) {
        const uint2 p = *reinterpret_cast<const uint2*>(wr);
        wv[i][c] = make_float4(__uint_as_float(p.x << 16), __uint_as_float(p.x & 0xffff0000u), __uint_as_float(p.y << 16), __uint_as_float(p.y & 0xffff0000u));
      } else {
        wv[i][c] = *reinterpret_cast<const float4*>(wr);
      }
    }
  }
  int idv[NV];
#pragma unroll
  for (int i = 0; i < NV; ++i) idv[i] = ids[i < nv ? i : 0];
  const int fv = forced ? *forced : -1;
  const int live = active ? *active : 1;
  const int l0 = lens ? lens[0] : 0, l1 = lens ? lens[1] : 0, fc = frame_ctr ? *frame_ctr : 0;
  float ss[2] = {0.f, 0.f};
#pragma unroll
  for (int c = 0; c < NC; ++c) {
    const bool cv = tid + 256 * c < H4;
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      const float4 v = xv[r][c];
      ss[r] += cv ? (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w) : 0.f;
    }
  }
  ss[0] = vv_wave_sum(ss[0]); ss[1] = vv_wave_sum(ss[1]);
  if (lane == 0) { red[wave][0] = ss[0]; red[wave][1] = ss[1]; }
  __syncthreads();
  float lg[NV];
#pragma unroll
  for (int i = 0; i < NV; ++i) lg[i] = 0.f;
#pragma unroll
  for (int r = 0; r < 2; ++r) {
    if (r >= R) break;
    const float rstd = rsqrtf(((red[0][r] + red[1][r]) + (red[2][r] + red[3][r])) / (float)H + eps);
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      const int ch = tid + 256 * c;
      if (ch >= H4) continue;
      const float4 v = make_float4(xv[r][c].x * rstd * nw[c].x, xv[r][c].y * rstd * nw[c].y, xv[r][c].z * rstd * nw[c].z, xv[r][c].w * rstd * nw[c].w);
      *reinterpret_cast<float4*>(out + (int64_t)r * ldo + 4 * ch) = v;
      if (r == 0) {
#pragma unroll
        for (int i = 0; i < NV; ++i) lg[i] += (wv[i][c].x * v.x + wv[i][c].y * v.y) + (wv[i][c].z * v.z + wv[i][c].w * v.w);
      }
    }
  }
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const float t = vv_wave_sum(lg[i]);
    if (lane == 0) lgp[wave][i] = t;
  }
  __syncthreads();
  if (tid == 0) {
    float l[NV];
    int best = 0;
    for (int i = 0; i < nv; ++i) {
      l[i] = (lgp[0][i] + lgp[1][i]) + (lgp[2][i] + lgp[3][i]);
      logits_out[i] = l[i];
      if (i && (l[i] > l[best] || (l[i] == l[best] && idv[i] < idv[best]))) best = i;
    }
    const int t = fv >= 0 ? fv : idv[best];
    *token_out = t;
    if (lens && live) {                              // a finished dialogue of the batch keeps its positions (its rows are computed and dropped)
      lens[0] = l0 + 1;
      if (tok_start < 0) { lens[1] = l1 + 1; if (t == tok_diff && frame_ctr) *frame_ctr = fc + 1; }   // refresh_negative=False: the negative row consumes every token
      else if (t == tok_start) lens[1] = 0;
      else if (t == tok_diff) { lens[1] = l1 + 1; if (frame_ctr) *frame_ctr = fc + 1; }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------
// streaming conv contexts of a whole tokenizer
// ---------------------------------------------------------------------------------------------------------------
struct CtxItem { float* pad; float* state; int ctx, T, C; const float* dw_w; float* hs; int affine; float scale, bias; };
struct CtxArgs { CtxItem it[VV_MAX_STAGES + 1 + 16]; int n; };   // convs of a net + the one-row stage's block histories (vv_model.hip)

// gather: pad[0 : ctx] <- state for every conv (before any of them runs)
__global__ __launch_bounds__(256) void conv_ctx_gather_kernel(const CtxArgs a) {
  const CtxItem it = a.it[blockIdx.y];
  const int n = it.ctx * it.C;
  if (it.affine) {               // the net's own input (latent frame / waveform chunk) into its stem's padded buffer: y = x * scale + bias
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) it.pad[i] = fmaf(it.state[i], it.scale, it.bias);
    return;
  }
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
  if (it.hs) {     // a Block1D history of the one-row stage: its depthwise-conv contribution to the next frame, from the rows just stored
    for (int c = blockIdx.x * 256 + threadIdx.x; c < it.C; c += gridDim.x * 256) {
      float s = 0.f;
#pragma unroll
      for (int k = 0; k < 6; ++k) s = fmaf(it.dw_w[(size_t)c * 7 + k], src[(size_t)k * it.C + c], s);
      it.hs[c] = s;
    }
  }
}


// ---- all adaLN modulations of a frame in ONE launch ---------------------------------------------------------------------------------
// mod[l][r, :] = adaLN_l c[r, :] for the 2 * n_steps conditioning rows c (bf16), every layer l and the final layer: 5 skinny GEMMs
// (40 rows x 4608 / 3072 outputs, K = 1536 at 1.5B) that ran as five launches of the LDS-tiled kernel on 72 workgroups each
// (12-15 us apiece, 65 us per frame).  Here workgroup bx owns 16 output channels of ONE of the matrices and NM row tiles of 16 rows on
// mfma_f32_16x16x32_bf16, K split over its 4 waves; weight and activation fragments are all requested up front, no LDS image.
struct AdaTab { const bf16_t* w[17]; float* out[17]; int nblk[17]; int ldo[17]; int n; };

typedef __bf16 ad_bf16x8 __attribute__((ext_vector_type(8)));
typedef float ad_f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int ad_u32x4 __attribute__((ext_vector_type(4)));

template <int NST, int NM>
__global__ __launch_bounds__(256) void adaln_all_kernel(const bf16_t* __restrict__ c, int rows, int D, const AdaTab tab) {
  __shared__ float red[4 * NM * 4 * 64];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  int bx = blockIdx.x, l = 0;
  while (l + 1 < tab.n && bx >= tab.nblk[l]) { bx -= tab.nblk[l]; ++l; }
  const int n0 = bx * 16, m0 = blockIdx.y * (16 * NM);
  const int r16 = lane & 15, kq = lane >> 4;
  const int kb = wave * (32 * NST) + 8 * kq;
  const bf16_t* wr = tab.w[l] + (int64_t)(n0 + r16) * D + kb;
  ad_u32x4 wf[NST], xf[NM][NST];
#pragma unroll
  for (int s = 0; s < NST; ++s) wf[s] = *reinterpret_cast<const ad_u32x4*>(wr + 32 * s);
#pragma unroll
  for (int t = 0; t < NM; ++t) {
    const bf16_t* xr = c + (int64_t)min(m0 + 16 * t + r16, rows - 1) * D + kb;
#pragma unroll
    for (int s = 0; s < NST; ++s) xf[t][s] = *reinterpret_cast<const ad_u32x4*>(xr + 32 * s);
  }
  ad_f32x4 acc[NM];
#pragma unroll
  for (int t = 0; t < NM; ++t) acc[t] = ad_f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int s = 0; s < NST; ++s)
#pragma unroll
    for (int t = 0; t < NM; ++t)
      acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(ad_bf16x8, wf[s]), __builtin_bit_cast(ad_bf16x8, xf[t][s]), acc[t], 0, 0, 0);
#pragma unroll
  for (int t = 0; t < NM; ++t)
#pragma unroll
    for (int i = 0; i < 4; ++i) red[((wave * NM + t) * 4 + i) * 64 + lane] = acc[t][i];
  __syncthreads();
  const int ei = tid >> 6, el = tid & 63;                          // accumulator register ei of lane el: channel n0 + 4 (el >> 4) + ei, row (el & 15)
  float* out = tab.out[l];
  const int ldo = tab.ldo[l];
#pragma unroll
  for (int t = 0; t < NM; ++t) {
    const int m = m0 + 16 * t + (el & 15);
    if (m < rows) {
      float v = 0.f;
#pragma unroll
      for (int w4 = 0; w4 < 4; ++w4) v += red[((w4 * NM + t) * 4 + ei) * 64 + el];      // fixed order: deterministic
      out[(int64_t)m * ldo + n0 + 4 * (el >> 4) + ei] = v;
    }
  }
}

// The same with the conditioning rows resident in LDS (D = 1536: <= 48 rows x 3 KB): in the kernel above every workgroup pulls all of c
// through L2 for 16 output channels (1344 workgroups x 147 KB = 3x the weight bytes).  Here one workgroup per CU stages its row group
// once and walks output blocks bx, bx + grid, ...: weights stream from memory one block ahead, activation fragments come from LDS.
template <int NST>
__global__ __launch_bounds__(256) void adaln_lds_kernel(const bf16_t* __restrict__ c, int rows, const AdaTab tab, int total) {
  constexpr int D = 4 * 32 * NST, P = D + 8, NM = 3, RG = 40;     // 40 rows per group (2 x 20 solver steps): 40 x 3 KB + partials fit the 160 KB LDS
  extern __shared__ __attribute__((aligned(16))) unsigned char ad_smem[];
  bf16_t* xs = reinterpret_cast<bf16_t*>(ad_smem);                                   // [nr][P]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int m0 = blockIdx.y * RG;
  const int nr = min(RG, rows - m0);
  float* red = reinterpret_cast<float*>(ad_smem + (size_t)RG * P * 2);              // [2][4 waves][NM][4][64]
  const int r16 = lane & 15, kq = lane >> 4;
  const int kb = wave * (32 * NST) + 8 * kq;
  auto locate = [&](int bx, int& l, int& n0) { l = 0; while (l + 1 < tab.n && bx >= tab.nblk[l]) { bx -= tab.nblk[l]; ++l; } n0 = bx * 16; };
  ad_u32x4 wcur[NST], wnxt[NST], wnn[NST];                         // weights run TWO output blocks ahead of the MFMAs
  int bx = blockIdx.x, l, n0;
  locate(bx, l, n0);
  {
    const bf16_t* wr = tab.w[l] + (int64_t)(n0 + r16) * D + kb;
#pragma unroll
    for (int s = 0; s < NST; ++s) wcur[s] = *reinterpret_cast<const ad_u32x4*>(wr + 32 * s);
    int l1, n1;
    locate(min(bx + (int)gridDim.x, total - 1), l1, n1);
    const bf16_t* wr1 = tab.w[l1] + (int64_t)(n1 + r16) * D + kb;
#pragma unroll
    for (int s = 0; s < NST; ++s) wnxt[s] = *reinterpret_cast<const ad_u32x4*>(wr1 + 32 * s);
  }
  {   // stage the row group: 16-byte chunks, 12 per thread per pass
    constexpr int CPR = D / 8;                                    // chunks per row
    const int nch = nr * CPR;
    for (int e0 = 0; e0 < nch; e0 += 256 * 12) {
      ad_u32x4 v[12];
#pragma unroll
      for (int i = 0; i < 12; ++i) {
        const int e = min(e0 + tid + 256 * i, nch - 1);
        const int r = e / CPR, cc = e - r * CPR;
        v[i] = *reinterpret_cast<const ad_u32x4*>(c + (int64_t)(m0 + r) * D + 8 * cc);
      }
#pragma unroll
      for (int i = 0; i < 12; ++i) {
        const int e = e0 + tid + 256 * i;
        if (e < nch) { const int r = e / CPR, cc = e - r * CPR; *reinterpret_cast<ad_u32x4*>(xs + r * P + 8 * cc) = v[i]; }
      }
    }
  }
  __syncthreads();
  const bf16_t* xf[NM];
#pragma unroll
  for (int t = 0; t < NM; ++t) xf[t] = xs + min(16 * t + r16, nr - 1) * P + kb;
  const int ei = tid >> 6, el = tid & 63;
  int par = 0;
  for (; bx < total; bx += gridDim.x) {
    const int bn = bx + 2 * gridDim.x;
    if (bn < total) {
      int l2, n2;
      locate(bn, l2, n2);
      const bf16_t* wr = tab.w[l2] + (int64_t)(n2 + r16) * D + kb;
#pragma unroll
      for (int s = 0; s < NST; ++s) wnn[s] = *reinterpret_cast<const ad_u32x4*>(wr + 32 * s);
    }
    ad_f32x4 acc[NM];
#pragma unroll
    for (int t = 0; t < NM; ++t) acc[t] = ad_f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < NST; ++s)
#pragma unroll
      for (int t = 0; t < NM; ++t) {
        const ad_u32x4 xb = *reinterpret_cast<const ad_u32x4*>(xf[t] + 32 * s);
        acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(ad_bf16x8, wcur[s]), __builtin_bit_cast(ad_bf16x8, xb), acc[t], 0, 0, 0);
      }
    float* rp = red + par * (4 * NM * 4 * 64);
#pragma unroll
    for (int t = 0; t < NM; ++t)
#pragma unroll
      for (int i = 0; i < 4; ++i) rp[((wave * NM + t) * 4 + i) * 64 + lane] = acc[t][i];
    __syncthreads();                     // one barrier per block: the partials alternate between two buffers
    locate(bx, l, n0);
    float* out = tab.out[l];
    const int ldo = tab.ldo[l];
#pragma unroll
    for (int t = 0; t < NM; ++t) {
      const int m = 16 * t + (el & 15);
      if (m < nr) {
        float v = 0.f;
#pragma unroll
        for (int w4 = 0; w4 < 4; ++w4) v += rp[((w4 * NM + t) * 4 + ei) * 64 + el];      // fixed order: deterministic
        out[(int64_t)(m0 + m) * ldo + n0 + 4 * (el >> 4) + ei] = v;
      }
    }
    par ^= 1;
#pragma unroll
    for (int s = 0; s < NST; ++s) { wcur[s] = wnxt[s]; wnxt[s] = wnn[s]; }
  }
}


// ---------------------------------------------------------------------------------------------------------------
// both SpeechConnectors of a frame in two launches (modeling_vibevoice.py:58-69, modeling_vibevoice_inference.py:665-667):
//   x_next = (fc2_a RMSNorm(fc1_a latent + b1_a) + b2_a) + (fc2_s RMSNorm(fc1_s sem + b1_s) + b2_s), stored to `rows_out` rows
// They ran as 4 dependent GEMV launches + a row copy (24 us for 9.8 MB); fc1 of both is one launch (one output per thread, K = 64 / 128),
// fc2 of both is one M = 1 GEMV over the two [H, H] matrices with the two RMSNorms in its prologue and the sum + both output rows in its epilogue.
// ---------------------------------------------------------------------------------------------------------------
struct ConnPairArgs {
  const bf16_t* fc1_a; const float* b1_a; int din_a; const bf16_t* fc1_s; const float* b1_s; int din_s;
  const float* xa; const float* xs;                     // latent [din_a], semantic features [din_s]
  float* t;                                             // [2, H] fc1 outputs (workspace)
  const float* nw_a; const float* nw_s; const bf16_t* fc2_a; const bf16_t* fc2_s; const float* b2_a; const float* b2_s;
  float* out; int64_t ldo; int rows_out; int H; float eps;
};

__global__ __launch_bounds__(256) void conn_fc1_pair_kernel(const ConnPairArgs a) {
  __shared__ __attribute__((aligned(16))) float xin[512];
  const int tid = threadIdx.x;
  for (int i = tid; i < a.din_a + a.din_s; i += 256) xin[i] = i < a.din_a ? a.xa[i] : a.xs[i - a.din_a];
  __syncthreads();
  const int n = blockIdx.x * 256 + tid;
  if (n >= 2 * a.H) return;
  const bool sem = n >= a.H;
  const int row = sem ? n - a.H : n, din = sem ? a.din_s : a.din_a;
  const bf16_t* w = (sem ? a.fc1_s : a.fc1_a) + (int64_t)row * din;
  const float* x = xin + (sem ? a.din_a : 0);
  float acc = 0.f;
  for (int k = 0; k < din; k += 8) {
    const uint4 v = *reinterpret_cast<const uint4*>(w + k);
    const unsigned u[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      acc = fmaf(__uint_as_float(u[j] << 16), x[k + 2 * j], acc);
      acc = fmaf(__uint_as_float(u[j] & 0xffff0000u), x[k + 2 * j + 1], acc);
    }
  }
  a.t[n] = acc + (sem ? a.b1_s : a.b1_a)[row];
}

template <int KU>       // H = KU * 512
__global__ __launch_bounds__(256) void conn_fc2_pair_kernel(const ConnPairArgs a, int rows_per_wave) {
  __shared__ __attribute__((aligned(16))) float ys[2][KU * 512];
  __shared__ float red[4][2];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int H = a.H;
  constexpr int NCH = (KU * 128 + 255) / 256;
  const int nchunks = H >> 2;
  float4 tv[2][NCH], nv[2][NCH];
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    const int ch = tid + c * 256;
    const int kk = ch < nchunks ? ch * 4 : 0;
    tv[0][c] = *reinterpret_cast<const float4*>(a.t + kk);
    tv[1][c] = *reinterpret_cast<const float4*>(a.t + H + kk);
    nv[0][c] = *reinterpret_cast<const float4*>(a.nw_a + kk);
    nv[1][c] = *reinterpret_cast<const float4*>(a.nw_s + kk);
  }
  // this wave's rows: gw, gw + nwaves, ... (at most 2 at the shipped shapes); the first row's weights are requested before the prologue
  const int nwaves = gridDim.x * 4, gw = blockIdx.x * 4 + wave;
  uint4 wa[KU], wsm[KU];
  auto issue = [&](int row) {
    const bool live = row < H;
#pragma unroll
    for (int u = 0; u < KU; ++u) {
      const int64_t off = live ? (int64_t)row * H + u * 512 + lane * 8 : 0;
      wa[u] = *reinterpret_cast<const uint4*>(a.fc2_a + off);
      wsm[u] = *reinterpret_cast<const uint4*>(a.fc2_s + off);
    }
  };
  issue(gw);
  float ss[2];
#pragma unroll
  for (int r = 0; r < 2; ++r) {
    float s1 = 0.f;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      const float4 v = tv[r][c];
      s1 += (tid + c * 256 < nchunks) ? (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w) : 0.f;
    }
    ss[r] = vv_wave_sum(s1);
  }
  if (lane == 0) { red[wave][0] = ss[0]; red[wave][1] = ss[1]; }
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");      // LDS-only barrier: the weight loads stay in flight
#pragma unroll
  for (int r = 0; r < 2; ++r) {
    const float rs = rsqrtf(((red[0][r] + red[1][r]) + (red[2][r] + red[3][r])) / (float)H + a.eps);
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      const int ch = tid + c * 256;
      if (ch < KU * 128) {
        const float4 v = tv[r][c], w = nv[r][c];
        *reinterpret_cast<float4*>(&ys[r][ch * 4]) = ch < nchunks ? make_float4(v.x * rs * w.x, v.y * rs * w.y, v.z * rs * w.z, v.w * rs * w.w) : make_float4(0.f, 0.f, 0.f, 0.f);
      }
    }
  }
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
  float ya[KU][8], yb[KU][8];
#pragma unroll
  for (int u = 0; u < KU; ++u) {
#pragma unroll
    for (int h2 = 0; h2 < 2; ++h2) {
      const float4 p = *reinterpret_cast<const float4*>(&ys[0][u * 512 + lane * 8 + 4 * h2]), q = *reinterpret_cast<const float4*>(&ys[1][u * 512 + lane * 8 + 4 * h2]);
      ya[u][4 * h2] = p.x; ya[u][4 * h2 + 1] = p.y; ya[u][4 * h2 + 2] = p.z; ya[u][4 * h2 + 3] = p.w;
      yb[u][4 * h2] = q.x; yb[u][4 * h2 + 1] = q.y; yb[u][4 * h2 + 2] = q.z; yb[u][4 * h2 + 3] = q.w;
    }
  }
  for (int i = 0; i < rows_per_wave; ++i) {
    const int row = gw + i * nwaves;
    float sa = 0.f, sb = 0.f;
#pragma unroll
    for (int u = 0; u < KU; ++u) {
      const unsigned ua[4] = {wa[u].x, wa[u].y, wa[u].z, wa[u].w}, ub[4] = {wsm[u].x, wsm[u].y, wsm[u].z, wsm[u].w};
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        sa = fmaf(__uint_as_float(ua[j] << 16), ya[u][2 * j], sa); sa = fmaf(__uint_as_float(ua[j] & 0xffff0000u), ya[u][2 * j + 1], sa);
        sb = fmaf(__uint_as_float(ub[j] << 16), yb[u][2 * j], sb); sb = fmaf(__uint_as_float(ub[j] & 0xffff0000u), yb[u][2 * j + 1], sb);
      }
    }
    if (i + 1 < rows_per_wave) issue(row + nwaves);
    sa = vv_wave_sum(sa);
    sb = vv_wave_sum(sb);
    if (lane == 0 && row < H) {
      const float v = (sa + a.b2_a[row]) + (sb + a.b2_s[row]);      // acoustic_embed + semantic_embed
      for (int r = 0; r < a.rows_out; ++r) a.out[(int64_t)r * a.ldo + row] = v;
    }
  }
}


// ---------------------------------------------------------------------------------------------------------------
// head_pre_kernel: everything in front of the solver loop that does not need the adaLN matrices, in ONE launch (it was three: the cond_proj
// GEMV, the silu(c0 + t_emb(t_i)) row expansion, the solver-state initialisation - 4.7 + 5.3 + 4.9 us of a frame for < 1 us of work):
//   workgroups [0, gemv_blocks): c0 = cond_proj cond for the two CFG rows (a wave owns output pairs n, n + 1), and straight from the wave's
//                                registers c[(i * 2 + j), n] = bf16(silu(c0[j, n] + temb[i, n])) for every solver step i (lane = step)
//   the rest:                    X0 = [P noise ; noise], M = 0, both rows of the first step's hidden state (head_init_kernel's body)
// Reference: modular_vibevoice_diffusion_head.py:262-270 (cond_proj, t_embedder, c = cond + t), :152-156 (SiLU in front of adaLN_modulation).
// ---------------------------------------------------------------------------------------------------------------
struct HeadPreArgs {
  const bf16_t* Wc; const float* cond; int64_t ld_cond; int K;       // cond_proj [D, K], cond [2, K]
  const float* temb; int n_steps; bf16_t* c_out;                    // temb [n_steps, D] -> c_out [2 n_steps, D]
  const bf16_t* P; const float* noise; int D, latent; float* Xs; float* Ms; float* h0; int64_t ldh;
  int gemv_blocks;
};

template <int KU>      // K = KU * 512
__global__ __launch_bounds__(256) void head_pre_kernel(const HeadPreArgs a) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if ((int)blockIdx.x >= a.gemv_blocks) {        // solver-state initialisation
    __shared__ float nz[256];
    for (int i = tid; i < a.latent; i += 256) nz[i] = a.noise[i];
    __syncthreads();
    const int n = ((int)blockIdx.x - a.gemv_blocks) * 256 + tid;
    if (n >= a.D + a.latent) return;
    float s;
    if (n < a.D) {
      s = 0.f;
      const bf16_t* pr = a.P + (int64_t)n * a.latent;
      for (int j = 0; j < a.latent; ++j) s = fmaf(bf2f(pr[j]), nz[j], s);
      a.h0[n] = s;
      a.h0[a.ldh + n] = s;
    } else {
      s = nz[n - a.D];
    }
    a.Xs[n] = s;
    a.Ms[n] = 0.f;
    return;
  }
  const int D = a.D, npairs = D >> 1;
  const int nwaves = a.gemv_blocks * 4;
  int g = blockIdx.x * 4 + wave;
  float x[2][KU][8];
#pragma unroll
  for (int r = 0; r < 2; ++r)
#pragma unroll
    for (int u = 0; u < KU; ++u) {
      const float4 p = *reinterpret_cast<const float4*>(a.cond + r * a.ld_cond + u * 512 + lane * 8), q = *reinterpret_cast<const float4*>(a.cond + r * a.ld_cond + u * 512 + lane * 8 + 4);
      x[r][u][0] = p.x; x[r][u][1] = p.y; x[r][u][2] = p.z; x[r][u][3] = p.w; x[r][u][4] = q.x; x[r][u][5] = q.y; x[r][u][6] = q.z; x[r][u][7] = q.w;
    }
  for (; g < npairs; g += nwaves) {
    const int n = 2 * g;
    uint4 w[2][KU];
#pragma unroll
    for (int o = 0; o < 2; ++o)
#pragma unroll
      for (int u = 0; u < KU; ++u) w[o][u] = *reinterpret_cast<const uint4*>(a.Wc + (int64_t)(n + o) * a.K + u * 512 + lane * 8);
    // the step rows this lane will write: requested now, they arrive while the dot products run
    float te[2][2];                                // [pass][output]: steps lane and lane + 64
#pragma unroll
    for (int ps = 0; ps < 2; ++ps) {
      const int i = min(lane + 64 * ps, a.n_steps - 1);
      const float2 t2 = *reinterpret_cast<const float2*>(a.temb + (int64_t)i * D + n);
      te[ps][0] = t2.x; te[ps][1] = t2.y;
    }
    float acc[2][2];                               // [output][row]
#pragma unroll
    for (int o = 0; o < 2; ++o) {
      acc[o][0] = 0.f; acc[o][1] = 0.f;
#pragma unroll
      for (int u = 0; u < KU; ++u) {
        const unsigned uu[4] = {w[o][u].x, w[o][u].y, w[o][u].z, w[o][u].w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float w0 = __uint_as_float(uu[j] << 16), w1 = __uint_as_float(uu[j] & 0xffff0000u);
          acc[o][0] = fmaf(w0, x[0][u][2 * j], acc[o][0]); acc[o][0] = fmaf(w1, x[0][u][2 * j + 1], acc[o][0]);
          acc[o][1] = fmaf(w0, x[1][u][2 * j], acc[o][1]); acc[o][1] = fmaf(w1, x[1][u][2 * j + 1], acc[o][1]);
        }
      }
    }
    float c0[2][2];
#pragma unroll
    for (int o = 0; o < 2; ++o)
#pragma unroll
      for (int r = 0; r < 2; ++r) c0[o][r] = vv_wave_sum(acc[o][r]);
#pragma unroll
    for (int ps = 0; ps < 2; ++ps) {
      const int i = lane + 64 * ps;
      if (i < a.n_steps) {
#pragma unroll
        for (int r = 0; r < 2; ++r) {
          const float v0 = c0[0][r] + te[ps][0], v1 = c0[1][r] + te[ps][1];
          const float s0 = v0 / (1.0f + expf(-v0)), s1 = v1 / (1.0f + expf(-v1));
          const unsigned u0 = __float_as_uint(s0), u1 = __float_as_uint(s1);        // round to nearest even, as kv_store<bf16_t>
          const unsigned pk = ((u0 + 0x7fffu + ((u0 >> 16) & 1u)) >> 16) | (((u1 + 0x7fffu + ((u1 >> 16) & 1u)) >> 16) << 16);
          *reinterpret_cast<unsigned*>(a.c_out + ((int64_t)(i * 2 + r)) * D + n) = pk;
        }
      }
    }
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
  return vv_head_boundary_batch(h, hrows, ldh, shift, scale, ld_mod, cfg, k, Xs, Ms, 0, h_out, ldh_out, latent_out, 0, 1, s);
}

// B dialogues in one launch: hidden rows [2 B] (dialogue b: rows 2 b, 2 b + 1), modulation rows likewise, solver state / sample of b at b * stride
int vv_head_boundary_batch(const vv_head* h, const float* hrows, int64_t ldh, const float* shift, const float* scale, int64_t ld_mod, float cfg,
                           const vv_dpm_coef* k, float* Xs, float* Ms, int64_t state_stride, float* h_out, int64_t ldh_out, float* latent_out,
                           int64_t latent_stride, int B, hipStream_t s) {
  BoundaryArgs a;
  a.state_stride = state_stride; a.latent_stride = latent_stride;
  a.G = h->fused_g; a.h = hrows; a.ldh = ldh; a.shift = shift; a.scale = scale; a.ld_mod = ld_mod; a.eps = h->eps; a.cfg = cfg; a.k = *k;
  a.Xs = Xs; a.Ms = Ms; a.h_out = h_out; a.ldh_out = ldh_out; a.latent_out = latent_out; a.D = h->D; a.latent = h->latent;
  switch ((h->D + 511) / 512) {
    case 1: launch_boundary<1>(a, B, s); break;
    case 2: launch_boundary<2>(a, B, s); break;
    case 3: launch_boundary<3>(a, B, s); break;
    case 4: launch_boundary<4>(a, B, s); break;
    case 5: launch_boundary<5>(a, B, s); break;
    case 6: launch_boundary<6>(a, B, s); break;
    case 7: launch_boundary<7>(a, B, s); break;
    case 8: launch_boundary<8>(a, B, s); break;
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
  auto a16 = [](const void* q) { return ((uintptr_t)q % 16) == 0; };
  if (R <= 2 && nv <= 8 && m->hidden % 4 == 0 && m->hidden <= 4096 && a16(h) && ldh % 4 == 0 && a16(out) && ldo % 4 == 0 && a16(m->final_norm) &&
      ((uintptr_t)w_valid % (m->wdt == VV_F32 ? 16 : 8)) == 0 && (m->wdt == VV_F32 || m->hidden % 4 == 0)) {
    if (m->wdt == VV_F32)
      hipLaunchKernelGGL((llm_tail_fast_kernel<float>), dim3(1), dim3(256), 0, s, h, ldh, R, m->hidden, m->final_norm, m->rms_eps, out, ldo, (const float*)w_valid, nv,
                         ids, logits_out, token_out, forced_token, lens, tok_start, tok_diffusion, frame_counter, (const int*)nullptr);
    else
      hipLaunchKernelGGL((llm_tail_fast_kernel<bf16_t>), dim3(1), dim3(256), 0, s, h, ldh, R, m->hidden, m->final_norm, m->rms_eps, out, ldo, (const bf16_t*)w_valid,
                         nv, ids, logits_out, token_out, forced_token, lens, tok_start, tok_diffusion, frame_counter, (const int*)nullptr);
    VV_CHECK_LAUNCH("vv_llm_tail");
    return 0;
  }
  if (m->wdt == VV_F32)
    hipLaunchKernelGGL((llm_tail_kernel<float>), dim3(1), dim3(256), lds, s, h, ldh, R, m->hidden, m->final_norm, m->rms_eps, out, ldo, (const float*)w_valid, nv, ids,
                       logits_out, token_out, forced_token, lens, tok_start, tok_diffusion, frame_counter);
  else
    hipLaunchKernelGGL((llm_tail_kernel<bf16_t>), dim3(1), dim3(256), lds, s, h, ldh, R, m->hidden, m->final_norm, m->rms_eps, out, ldo, (const bf16_t*)w_valid, nv, ids,
                       logits_out, token_out, forced_token, lens, tok_start, tok_diffusion, frame_counter);
  VV_CHECK_LAUNCH("vv_llm_tail");
  return 0;
}

// The tail of a row-batched decode step (B dialogues, rows {positive, negative} x B): block b does what vv_llm_tail does for dialogue b.
extern "C" int vv_llm_tail_batch(const vv_llm* m, const float* h, int64_t ldh, int B, float* out, int64_t ldo, const void* w_valid, int nv, const int* ids,
                                 float* logits_out, int* token_out, const int* forced_token, int* lens, int tok_start, int tok_diffusion,
                                 int* frame_counter, const int* active, vv_stream_t stream) {
  if (!m || !h || !out || !w_valid || !ids || !logits_out || !token_out) return vv_set_error(VV_E_ARG, "vv_llm_tail_batch: null pointer");
  if (B <= 0 || B > 4 || nv <= 0 || nv > 8) return vv_set_error(VV_E_ARG, "vv_llm_tail_batch: B=%d (1..4) nv=%d (<= 8)", B, nv);
  auto a16 = [](const void* q) { return ((uintptr_t)q % 16) == 0; };
  if (!(m->hidden % 4 == 0 && m->hidden <= 4096 && a16(h) && ldh % 4 == 0 && a16(out) && ldo % 4 == 0 && a16(m->final_norm) &&
        ((uintptr_t)w_valid % (m->wdt == VV_F32 ? 16 : 8)) == 0))
    return vv_set_error(VV_E_UNSUPPORTED, "vv_llm_tail_batch: hidden=%d / alignment not covered", m->hidden);
  hipStream_t s = (hipStream_t)stream;
  if (m->wdt == VV_F32)
    hipLaunchKernelGGL((llm_tail_fast_kernel<float>), dim3(B), dim3(256), 0, s, h, ldh, 2, m->hidden, m->final_norm, m->rms_eps, out, ldo, (const float*)w_valid, nv,
                       ids, logits_out, token_out, forced_token, lens, tok_start, tok_diffusion, frame_counter, active);
  else
    hipLaunchKernelGGL((llm_tail_fast_kernel<bf16_t>), dim3(B), dim3(256), 0, s, h, ldh, 2, m->hidden, m->final_norm, m->rms_eps, out, ldo, (const bf16_t*)w_valid,
                       nv, ids, logits_out, token_out, forced_token, lens, tok_start, tok_diffusion, frame_counter, active);
  VV_CHECK_LAUNCH("vv_llm_tail_batch");
  return 0;
}

int vv_conv_ctx_batch(const vv_conv_ctx_item* items, int n, int scatter, hipStream_t s) {
  if (n <= 0) return 0;
  if (n > VV_MAX_STAGES + 1 + 16) return vv_set_error(VV_E_ARG, "vv_conv_ctx_batch: too many items");
  CtxArgs a;
  a.n = n;
  int mx = 0;
  for (int i = 0; i < n; ++i) {
    a.it[i].pad = items[i].pad; a.it[i].state = items[i].state; a.it[i].ctx = items[i].ctx; a.it[i].T = items[i].T; a.it[i].C = items[i].C;
    a.it[i].dw_w = scatter ? items[i].dw_w : nullptr; a.it[i].hs = scatter ? items[i].hs : nullptr;
    a.it[i].affine = scatter ? 0 : items[i].affine; a.it[i].scale = items[i].scale; a.it[i].bias = items[i].bias;
    if (items[i].ctx * items[i].C > mx) mx = items[i].ctx * items[i].C;
  }
  int bx = (mx + 1023) / 1024;
  if (bx > 16) bx = 16;
  if (scatter) hipLaunchKernelGGL(conv_ctx_scatter_kernel, dim3(bx, n), dim3(256), 0, s, a);
  else hipLaunchKernelGGL(conv_ctx_gather_kernel, dim3(bx, n), dim3(256), 0, s, a);
  VV_CHECK_LAUNCH("vv_conv_ctx_batch");
  return 0;
}

static constexpr size_t ADA_LDS = (size_t)40 * (1536 + 8) * 2 + (size_t)2 * 4 * 3 * 4 * 64 * 4;

int vv_fused_init() {     // before any graph capture
  if (hipFuncSetAttribute(reinterpret_cast<const void*>(&adaln_lds_kernel<12>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ADA_LDS) != hipSuccess)
    return vv_set_error(VV_E_HIP, "vv_fused_init: cannot raise the LDS limit");
  return 0;
}

// 1 = launched, 0 = not covered (caller runs one vv_linear per matrix)
int vv_head_modulations_fused(const vv_head* h, const void* c_bf16, int rows, float* const* mod, float* modf, hipStream_t s) {
  const int D = h->D;
  if (h->wdt != VV_BF16 || h->layers > 16 || rows < 1 || ((uintptr_t)c_bf16 % 16) || ((uintptr_t)h->final_adaln % 16)) return 0;
  if (D != 4 * 32 * 12 && D != 4 * 32 * 28) return 0;
  AdaTab tab;
  int total = 0;
  for (int l = 0; l < h->layers; ++l) {
    if ((uintptr_t)h->layer[l].adaln % 16) return 0;
    tab.w[l] = reinterpret_cast<const bf16_t*>(h->layer[l].adaln); tab.out[l] = mod[l]; tab.nblk[l] = 3 * D / 16; tab.ldo[l] = 3 * D;
    total += tab.nblk[l];
  }
  const int f = h->layers;
  tab.w[f] = reinterpret_cast<const bf16_t*>(h->final_adaln); tab.out[f] = modf; tab.nblk[f] = 2 * D / 16; tab.ldo[f] = 2 * D;
  total += tab.nblk[f];
  tab.n = f + 1;
  for (int l = tab.n; l < 17; ++l) { tab.w[l] = nullptr; tab.out[l] = nullptr; tab.nblk[l] = 0; tab.ldo[l] = 0; }
  const bf16_t* c = reinterpret_cast<const bf16_t*>(c_bf16);
  if (D == 4 * 32 * 12) hipLaunchKernelGGL((adaln_lds_kernel<12>), dim3(total < 256 ? total : 256, (rows + 39) / 40), dim3(256), ADA_LDS, s, c, rows, tab, total);
  else hipLaunchKernelGGL((adaln_all_kernel<28, 1>), dim3(total, (rows + 15) / 16), dim3(256), 0, s, c, rows, D, tab);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return vv_set_error(VV_E_HIP, "vv_head_modulations_fused: %s", hipGetErrorString(e));
  return 1;
}

// 1 = launched, 0 = not covered (caller runs the two connectors one after the other)
int vv_launch_connector_pair(const vv_connector* ac, const vv_connector* sem, const float* latent, const float* semfeat, float* out, int64_t ldo, int rows_out,
                             float* ws, hipStream_t s) {
  const int H = ac->hidden;
  if (ac->wdt != VV_BF16 || sem->wdt != VV_BF16 || sem->hidden != H || H % 512 || H / 512 > 7 || ac->din % 8 || sem->din % 8 || ac->din + sem->din > 512) return 0;
  auto a16 = [](const void* q) { return q && ((uintptr_t)q % 16) == 0; };
  if (!a16(ac->fc1) || !a16(sem->fc1) || !a16(ac->fc2) || !a16(sem->fc2) || !a16(ac->norm_w) || !a16(sem->norm_w) || !a16(ws) || !ac->b1 || !sem->b1 || !ac->b2 || !sem->b2) return 0;
  ConnPairArgs a;
  a.fc1_a = (const bf16_t*)ac->fc1; a.b1_a = ac->b1; a.din_a = ac->din; a.fc1_s = (const bf16_t*)sem->fc1; a.b1_s = sem->b1; a.din_s = sem->din;
  a.xa = latent; a.xs = semfeat; a.t = ws;
  a.nw_a = ac->norm_w; a.nw_s = sem->norm_w; a.fc2_a = (const bf16_t*)ac->fc2; a.fc2_s = (const bf16_t*)sem->fc2; a.b2_a = ac->b2; a.b2_s = sem->b2;
  a.out = out; a.ldo = ldo; a.rows_out = rows_out; a.H = H; a.eps = 1e-6f;
  hipLaunchKernelGGL(conn_fc1_pair_kernel, dim3((2 * H + 255) / 256), dim3(256), 0, s, a);
  const int blocks = 192, nwaves = blocks * 4, rpw = (H + nwaves - 1) / nwaves;
  switch (H / 512) {
#define VV_C2(KU) case KU: hipLaunchKernelGGL((conn_fc2_pair_kernel<KU>), dim3(blocks), dim3(256), 0, s, a, rpw); break;
    VV_C2(1) VV_C2(2) VV_C2(3) VV_C2(4) VV_C2(5) VV_C2(6) VV_C2(7)
#undef VV_C2
  }
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return vv_set_error(VV_E_HIP, "vv_connector_pair: %s", hipGetErrorString(e));
  return 1;
}

// 1 = launched (cond_proj + silu row expansion + solver-state init), 0 = not covered
int vv_head_pre_fused(const vv_head* h, const float* cond2, int64_t ld_cond, const float* temb, int n_steps, void* c_bf16, const float* noise,
                      float* Xs, float* Ms, float* h0, int64_t ldh, hipStream_t s) {
  const int D = h->D, K = h->cond_dim;
  if (h->wdt != VV_BF16 || K % 512 || K / 512 > 7 || D % 2 || n_steps > 128 || h->latent > 256) return 0;
  auto a16 = [](const void* q) { return q && ((uintptr_t)q % 16) == 0; };
  if (!a16(h->cond_proj) || !a16(cond2) || (ld_cond % 4) || !a16(temb) || !a16(c_bf16) || !h->noisy_proj) return 0;
  HeadPreArgs a;
  a.Wc = (const bf16_t*)h->cond_proj; a.cond = cond2; a.ld_cond = ld_cond; a.K = K; a.temb = temb; a.n_steps = n_steps; a.c_out = (bf16_t*)c_bf16;
  a.P = (const bf16_t*)h->noisy_proj; a.noise = noise; a.D = D; a.latent = h->latent; a.Xs = Xs; a.Ms = Ms; a.h0 = h0; a.ldh = ldh;
  a.gemv_blocks = 192;
  const int init_blocks = (D + h->latent + 255) / 256;
  switch (K / 512) {
#define VV_HP(KU) case KU: hipLaunchKernelGGL((head_pre_kernel<KU>), dim3(a.gemv_blocks + init_blocks), dim3(256), 0, s, a); break;
    VV_HP(1) VV_HP(2) VV_HP(3) VV_HP(4) VV_HP(5) VV_HP(6) VV_HP(7)
#undef VV_HP
  }
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return vv_set_error(VV_E_HIP, "vv_head_pre_fused: %s", hipGetErrorString(e));
  return 1;
}
