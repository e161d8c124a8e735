// vv_kernels.hip — gfx950 (MI355X / CDNA4) kernels of the VibeVoice per-frame hot path and their launchers.
//
// Everything on this path at batch 1-2 is HBM-bandwidth bound (~2.5 flop/byte), so the kernels are
// built around 64-wide wavefronts streaming each weight byte exactly once with 16-byte loads per lane, staging the
// (tiny) activation vectors in LDS, reducing with wavefront shuffles, and fusing every elementwise neighbour
// (RMSNorm / adaLN modulate / SiLU prologues; bias / GELU / SwiGLU / layer-scale / residual epilogues) into the
// producing kernel.  MFMA is deliberately not used here: at M <= 8 rows the matrix cores cannot be fed.
#include <hip/hip_runtime.h>
#include <hip/hip_bf16.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <math.h>

#include "vv_hip.h"
#include "vv_common.h"

// ---------------------------------------------------------------------------------------------------------------
// error plumbing
// ---------------------------------------------------------------------------------------------------------------
static thread_local char g_err[512] = "";

int vv_set_error(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return code;
}

extern "C" const char* vv_last_error(void) { return g_err; }
extern "C" int vv_abi_version(void) { return 6; }   // 6: fragment-major weight copies (vv_llm_layer.f_*, vv_head_layer.f_*, VV_LIN_W_FRAG), vv_llm_tail_batch, vv_head_sample_batch; 5: vv_kv.vt in 32-key tiles, vv_attn_decode maintains it, vv_advance_lens tok_start < 0; 2: vv_w8 fp8 companions, vv_dpm_coef.cn + variance-noise arguments; 3: vv_head.fused_g, vv_llm_tail, vv_llm_forward(out = NULL); 4: vv_block.dw_last / hs, vv_block_mid
int vv_mixer_init();
extern "C" int vv_init(void) {
  VV_TRY(vv_mixer_init());
  VV_TRY(vv_mfma_gemm_init());
  VV_TRY(vv_block1d_init());
  VV_TRY(vv_convffn_init());
  VV_TRY(vv_gemv_rows_init());
  return vv_fused_init();
}
// process-wide split-K scratch of the 3..8-row matrix-core GEMV for PUBLIC vv_linear calls (micro-benchmarks and tests only, switched on by
// vv_tune("gemv_rows_scratch", 1): one stream at a time).  The composites hand vv_linear_ws their own workspace instead.
static float* g_rows_part = nullptr;
static int* g_rows_tk = nullptr;
static const size_t G_ROWS_PART_FLOATS = (size_t)4 << 20, G_ROWS_TICKETS = 4096;
static int rows_scratch(int on) {
  if (on && !g_rows_part) {
    if (hipMalloc(&g_rows_part, G_ROWS_PART_FLOATS * sizeof(float)) != hipSuccess || hipMalloc(&g_rows_tk, G_ROWS_TICKETS * sizeof(int)) != hipSuccess ||
        hipMemset(g_rows_tk, 0, G_ROWS_TICKETS * sizeof(int)) != hipSuccess)
      return vv_set_error(VV_E_HIP, "vv_tune: gemv_rows_scratch allocation failed");
  }
  if (!on && g_rows_part) { (void)hipFree(g_rows_part); (void)hipFree(g_rows_tk); g_rows_part = nullptr; g_rows_tk = nullptr; }
  return 0;
}
void vv_gemv_rows_set_dbg(int d);
void vv_gemv_rows_set_atomic(int on);
void vv_gemv_stream_set_blocks(int b);
void vv_gemv_stream_set_waves(int w);
void vv_gemv_stream_set_opt(int o);
void vv_gemv_stream_set_long(int cap, int ku);
void vv_gemv_stream_set_dual_rw(int r);
void vv_gemv_stream_set_small_rw(int r);
void vv_mfma_set_mt(int mt);
void vv_mfma_set_tiled_rows(int r);
void vv_mfma_set_tiled_bk128(int on);
void vv_mfma_set_tiled_small(int t);
void vv_mfma_set_tiled_dual_bk64(int on);
void vv_mfma_set_tiled_small_dual(int t);
void vv_mfma_set_tiled_small_k(int k);
void vv_mixer_set_rows(int on);
void vv_mfma_set_mt_prefill(int mt);
extern int g_attn_group;
void vv_attn_decode_set_gqa(int on);
void vv_attn_decode_set_gqa_keys(int k);
extern "C" int vv_tune(const char* key, int value) {   // developer tuning hooks (not part of the stable ABI surface)
  if (key && !strcmp(key, "gemv_long_cap")) { vv_gemv_stream_set_long(value, 0); return 0; }
  if (key && !strcmp(key, "gemv_long_ku")) { vv_gemv_stream_set_long(0, value); return 0; }
  if (key && !strcmp(key, "gemv_blocks")) { vv_gemv_stream_set_blocks(value); return 0; }
  if (key && !strcmp(key, "gemv_waves")) { vv_gemv_stream_set_waves(value); return 0; }
  if (key && !strcmp(key, "gemv_rows")) { vv_gemv_rows_set(value, 0, -1); return 0; }
  if (key && !strcmp(key, "gemv_rows_blocks")) { vv_gemv_rows_set(-1, value, -1); return 0; }
  if (key && !strcmp(key, "gemv_rows_pers")) { vv_gemv_rows_set(-1, 0, value); return 0; }
  if (key && !strcmp(key, "gemv_rows_scratch")) return rows_scratch(value);
  if (key && !strcmp(key, "gemv_rows_dbg")) { vv_gemv_rows_set_dbg(value); return 0; }
  if (key && !strcmp(key, "gemv_rows_atomic")) { vv_gemv_rows_set_atomic(value); return 0; }
  if (key && !strcmp(key, "gemv_opt")) { vv_gemv_stream_set_opt(value); return 0; }
  if (key && !strcmp(key, "gemv_dual_rw")) { vv_gemv_stream_set_dual_rw(value); return 0; }
  if (key && !strcmp(key, "gemv_small_rw")) { vv_gemv_stream_set_small_rw(value); return 0; }
  if (key && !strcmp(key, "mixer_rows")) { vv_mixer_set_rows(value); return 0; }
  if (key && !strcmp(key, "block1d_fused")) { vv_block1d_set_fused(value); return 0; }
  if (key && !strcmp(key, "block1d_blocks")) { vv_block1d_set_blocks(value); return 0; }
  if (key && !strcmp(key, "convffn")) { vv_convffn_set(value); return 0; }
  if (key && !strcmp(key, "convffn_t1")) { vv_convffn_set_t1(value); return 0; }
  if (key && !strcmp(key, "convffn_t1hs")) { vv_convffn_set_t1hs(value); return 0; }
  if (key && !strcmp(key, "convffn_rows512")) { vv_convffn_set_rows(512, value); return 0; }
  if (key && !strcmp(key, "convffn_rows256")) { vv_convffn_set_rows(256, value); return 0; }
  if (key && !strcmp(key, "convffn_rows128")) { vv_convffn_set_rows(128, value); return 0; }
  if (key && !strcmp(key, "convffn_c128")) { vv_convffn_set_c128(value); return 0; }
  if (key && !strcmp(key, "skinny")) { vv_skinny_set(value, 0, 0); return 0; }
  if (key && !strcmp(key, "skinny_min_m")) { vv_skinny_set(1, value, 0); return 0; }
  if (key && !strcmp(key, "skinny_max_m")) { vv_skinny_set(1, 0, value); return 0; }
  if (key && !strcmp(key, "mfma_tiled_bk128")) { vv_mfma_set_tiled_bk128(value); return 0; }
  if (key && !strcmp(key, "attn_group")) { g_attn_group = value; return 0; }
  if (key && !strcmp(key, "attn_gqa")) { vv_attn_decode_set_gqa(value); return 0; }
  if (key && !strcmp(key, "attn_gqa_keys")) { vv_attn_decode_set_gqa_keys(value); return 0; }
  if (key && !strcmp(key, "mfma_tiled_small_k")) { vv_mfma_set_tiled_small_k(value); return 0; }
  if (key && !strcmp(key, "mfma_tiled_dual_bk64")) { vv_mfma_set_tiled_dual_bk64(value); return 0; }
  if (key && !strcmp(key, "mfma_tiled_small_dual")) { vv_mfma_set_tiled_small_dual(value); return 0; }
  if (key && !strcmp(key, "mfma_tiled_small")) { vv_mfma_set_tiled_small(value); return 0; }
  if (key && !strcmp(key, "mfma_tiled_rows")) { vv_mfma_set_tiled_rows(value); return 0; }
  if (key && !strcmp(key, "mfma_mt")) { vv_mfma_set_mt(value); return 0; }
  if (key && !strcmp(key, "mfma_mt_prefill")) { vv_mfma_set_mt_prefill(value); return 0; }
  return vv_set_error(VV_E_ARG, "vv_tune: unknown key");
}

// ---------------------------------------------------------------------------------------------------------------
// device helpers
// ---------------------------------------------------------------------------------------------------------------
typedef unsigned short bf16_t;

__device__ __forceinline__ float bf2f(unsigned int u16) { return __uint_as_float(u16 << 16); }
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}
__device__ __forceinline__ float wave_sum_dpp(float v) { return vv_wave_sum(v); }   // all 64 lanes active
__device__ __forceinline__ float silu_f(float v) { return v / (1.0f + expf(-v)); }
__device__ __forceinline__ float gelu_f(float v) { return 0.5f * v * (1.0f + erff(v * 0.70710678118654752440f)); }

template <typename WT> struct WL;
template <> struct WL<float> {
  static __device__ __forceinline__ void load8(const float* p, float (&o)[8]) {
    const float4 a = *reinterpret_cast<const float4*>(p);
    const float4 b = *reinterpret_cast<const float4*>(p + 4);
    o[0] = a.x; o[1] = a.y; o[2] = a.z; o[3] = a.w; o[4] = b.x; o[5] = b.y; o[6] = b.z; o[7] = b.w;
  }
  static __device__ __forceinline__ void load4(const float* p, float (&o)[4]) {
    const float4 a = *reinterpret_cast<const float4*>(p);
    o[0] = a.x; o[1] = a.y; o[2] = a.z; o[3] = a.w;
  }
  static __device__ __forceinline__ float load1(const float* p) { return *p; }
};
template <> struct WL<bf16_t> {
  static __device__ __forceinline__ void load8(const bf16_t* p, float (&o)[8]) {
    const uint4 v = *reinterpret_cast<const uint4*>(p);
    o[0] = __uint_as_float(v.x << 16); o[1] = __uint_as_float(v.x & 0xffff0000u);
    o[2] = __uint_as_float(v.y << 16); o[3] = __uint_as_float(v.y & 0xffff0000u);
    o[4] = __uint_as_float(v.z << 16); o[5] = __uint_as_float(v.z & 0xffff0000u);
    o[6] = __uint_as_float(v.w << 16); o[7] = __uint_as_float(v.w & 0xffff0000u);
  }
  static __device__ __forceinline__ void load4(const bf16_t* p, float (&o)[4]) {
    const uint2 v = *reinterpret_cast<const uint2*>(p);
    o[0] = __uint_as_float(v.x << 16); o[1] = __uint_as_float(v.x & 0xffff0000u);
    o[2] = __uint_as_float(v.y << 16); o[3] = __uint_as_float(v.y & 0xffff0000u);
  }
  static __device__ __forceinline__ float load1(const bf16_t* p) { return bf2f(*p); }
};

__device__ __forceinline__ void lin_epilogue(const vv_lin_args& a, int m, int n, float v, float v2) {
  if (a.bias) v += a.bias[n];
  if (a.act == VV_ACT_GELU) v = gelu_f(v);
  else if (a.act == VV_ACT_SWIGLU) v = silu_f(v) * v2;
  if (a.gate) v *= a.gate_ld ? a.gate[(int64_t)m * a.gate_ld + n] : a.gate[n];
  if (a.res) v += a.res[(int64_t)m * a.ldres + n];
  a.out[(int64_t)m * a.ldo + n] = v;
}

// ---------------------------------------------------------------------------------------------------------------
// GEMV: M <= 8 rows of activations against a streamed weight matrix.
//   block = 256 threads = 4 waves.  KSPLIT == 1: each wave owns RW whole weight rows (large N);
//   KSPLIT == 4: the block owns RW rows and its 4 waves split K (small N, long K) and combine through LDS.
//   x is staged (with its prologue applied) into LDS in chunks of KC elements so LDS use stays <= 32 KB.
// ---------------------------------------------------------------------------------------------------------------
#define GEMV_THREADS 256

template <int M>
__device__ __forceinline__ void row_stats(const vv_lin_args& a, float* red, float (&rstd)[M]) {
  // rstd[m] = rsqrt(mean(x[m]^2) + eps), every thread gets all M values
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  float ss[M];
#pragma unroll
  for (int m = 0; m < M; ++m) {
    const float* xr = a.x + (int64_t)m * a.ldx;
    float s = 0.f;
    for (int k = tid; k < a.k; k += GEMV_THREADS) { const float v = xr[k]; s += v * v; }
    ss[m] = wave_sum(s);
  }
  if (lane == 0) {
#pragma unroll
    for (int m = 0; m < M; ++m) red[wave * M + m] = ss[m];
  }
  __syncthreads();
#pragma unroll
  for (int m = 0; m < M; ++m) {
    const float t = red[m] + red[M + m] + red[2 * M + m] + red[3 * M + m];
    rstd[m] = rsqrtf(t / (float)a.k + a.eps);
  }
  __syncthreads();
}

template <int M>
__device__ __forceinline__ void stage_chunk(const vv_lin_args& a, float* xs, int kc0, int kc, int kcp, const float (&rstd)[M]) {
  // xs[m][0..kcp) <- prologue(x[m][kc0 .. kc0+kc)), zero padded to kcp
  for (int idx = threadIdx.x; idx < M * kcp; idx += GEMV_THREADS) {
    const int m = idx / kcp, kk = idx - m * kcp;
    float v = 0.f;
    if (kk < kc) {
      const int k = kc0 + kk;
      v = a.x[(int64_t)m * a.ldx + k];
      if (a.pro == VV_PRO_RMSNORM) {
        v *= rstd[m];
        if (a.norm_w) v *= a.norm_w[k];
        if (a.mod_scale) v = v * (1.0f + a.mod_scale[(int64_t)m * a.ld_mod + k]) + a.mod_shift[(int64_t)m * a.ld_mod + k];
      } else if (a.pro == VV_PRO_SILU) {
        v = silu_f(v);
      }
    }
    xs[idx] = v;
  }
}

template <typename WT, int M, bool DUAL, int KSPLIT, int RW>
__global__ __launch_bounds__(GEMV_THREADS) void gemv_kernel(const vv_lin_args a, const int KC) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* red = smem;                 // [4*M] stats scratch, then [4][RW][M][2] split-K scratch
  float* xs = smem + 64 * 4;         // [M][KCP]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int K = a.k, N = a.n;
  const WT* __restrict__ W = reinterpret_cast<const WT*>(a.w);
  const WT* __restrict__ W2 = reinterpret_cast<const WT*>(a.w2);

  float rstd[M];
#pragma unroll
  for (int m = 0; m < M; ++m) rstd[m] = 1.f;
  if (a.pro == VV_PRO_RMSNORM) row_stats<M>(a, red, rstd);

  const int row0 = (KSPLIT == 1 ? (blockIdx.x * 4 + wave) : blockIdx.x) * RW;
  float acc[RW][M], acc2[DUAL ? RW : 1][M];
#pragma unroll
  for (int r = 0; r < RW; ++r)
#pragma unroll
    for (int m = 0; m < M; ++m) { acc[r][m] = 0.f; if (DUAL) acc2[r][m] = 0.f; }

  for (int kc0 = 0; kc0 < K; kc0 += KC) {
    const int kc = min(KC, K - kc0);
    const int kcp = (kc + 7) & ~7;
    if (kc0 > 0) __syncthreads();
    stage_chunk<M>(a, xs, kc0, kc, kcp, rstd);
    __syncthreads();
    const int kstart = (KSPLIT == 1 ? lane : (wave * 64 + lane)) * 8;
    const int kstep = (KSPLIT == 1 ? 64 : 256) * 8;
#pragma unroll 2
    for (int k = kstart; k < kc; k += kstep) {
      float w[RW][8], w2[DUAL ? RW : 1][8];
#pragma unroll
      for (int r = 0; r < RW; ++r) {
        const int n = min(row0 + r, N - 1);
        WL<WT>::load8(W + (int64_t)n * K + kc0 + k, w[r]);
        if (DUAL) WL<WT>::load8(W2 + (int64_t)n * K + kc0 + k, w2[r]);
      }
#pragma unroll
      for (int m = 0; m < M; ++m) {
        const float4 x0 = *reinterpret_cast<const float4*>(xs + m * kcp + k);
        const float4 x1 = *reinterpret_cast<const float4*>(xs + m * kcp + k + 4);
        const float xv[8] = {x0.x, x0.y, x0.z, x0.w, x1.x, x1.y, x1.z, x1.w};
#pragma unroll
        for (int r = 0; r < RW; ++r) {
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            acc[r][m] = fmaf(w[r][j], xv[j], acc[r][m]);
            if (DUAL) acc2[r][m] = fmaf(w2[r][j], xv[j], acc2[r][m]);
          }
        }
      }
    }
  }
#pragma unroll
  for (int r = 0; r < RW; ++r)
#pragma unroll
    for (int m = 0; m < M; ++m) { acc[r][m] = wave_sum(acc[r][m]); if (DUAL) acc2[r][m] = wave_sum(acc2[r][m]); }

  if (KSPLIT == 1) {
    if (lane == 0) {
#pragma unroll
      for (int r = 0; r < RW; ++r)
        if (row0 + r < N) {
#pragma unroll
          for (int m = 0; m < M; ++m) lin_epilogue(a, m, row0 + r, acc[r][m], DUAL ? acc2[r][m] : 0.f);
        }
    }
  } else {
    __syncthreads();   // xs/red reuse
    float* part = smem;  // [4][RW*M*2]
    if (lane == 0) {
#pragma unroll
      for (int r = 0; r < RW; ++r)
#pragma unroll
        for (int m = 0; m < M; ++m) {
          part[(wave * RW * M + r * M + m) * 2 + 0] = acc[r][m];
          part[(wave * RW * M + r * M + m) * 2 + 1] = DUAL ? acc2[r][m] : 0.f;
        }
    }
    __syncthreads();
    if (tid < RW * M) {
      const int r = tid / M, m = tid - r * M;
      if (row0 + r < N) {
        float s = 0.f, s2 = 0.f;
#pragma unroll
        for (int w4 = 0; w4 < 4; ++w4) { s += part[(w4 * RW * M + tid) * 2]; s2 += part[(w4 * RW * M + tid) * 2 + 1]; }
        lin_epilogue(a, m, row0 + r, s, s2);
      }
    }
  }
}

// generic (any K, any alignment) fallback: one wave per output row, scalar loads.  Only tiny shapes land here.
template <typename WT>
__global__ __launch_bounds__(GEMV_THREADS) void gemv_generic_kernel(const vv_lin_args a) {
  __shared__ float red[4];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const WT* W = reinterpret_cast<const WT*>(a.w);
  const WT* W2 = reinterpret_cast<const WT*>(a.w2);
  for (int m = 0; m < a.m; ++m) {
    const float* xr = a.x + (int64_t)m * a.ldx;
    float rstd = 1.f;
    if (a.pro == VV_PRO_RMSNORM) {
      float s = 0.f;
      for (int k = tid; k < a.k; k += GEMV_THREADS) s += xr[k] * xr[k];
      s = wave_sum(s);
      __syncthreads();
      if (lane == 0) red[wave] = s;
      __syncthreads();
      rstd = rsqrtf((red[0] + red[1] + red[2] + red[3]) / (float)a.k + a.eps);
    }
    for (int n = blockIdx.x * 4 + wave; n < a.n; n += gridDim.x * 4) {
      float s = 0.f, s2 = 0.f;
      for (int k = lane; k < a.k; k += 64) {
        float v = xr[k];
        if (a.pro == VV_PRO_RMSNORM) {
          v *= rstd;
          if (a.norm_w) v *= a.norm_w[k];
          if (a.mod_scale) v = v * (1.0f + a.mod_scale[(int64_t)m * a.ld_mod + k]) + a.mod_shift[(int64_t)m * a.ld_mod + k];
        } else if (a.pro == VV_PRO_SILU) v = silu_f(v);
        s = fmaf(WL<WT>::load1(W + (int64_t)n * a.k + k), v, s);
        if (W2) s2 = fmaf(WL<WT>::load1(W2 + (int64_t)n * a.k + k), v, s2);
      }
      s = wave_sum(s); s2 = wave_sum(s2);
      if (lane == 0) lin_epilogue(a, m, n, s, s2);
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------
// GEMM (M > 8): TMT x TNT output tile per 256-thread block (64x64 with a 4x4 micro-tile, or 32x32 with 2x2 when the
// problem is too small to fill the chip with big tiles), K stepped by 16 through LDS with the next K-slab's global
// loads already in flight in registers while the current one is multiplied.
// A rows may overlap (ldx < k): that is how dense Conv1d / ConvTranspose1d read their im2col view in place.
// ---------------------------------------------------------------------------------------------------------------
#define TK 16

template <typename WT, bool DUAL, bool VEC, int TMT, int TNT>
__global__ __launch_bounds__(256) void gemm_kernel(const vv_lin_args a) {
  constexpr int MI = TMT / 16, NJ = TNT / 16;
  constexpr bool BOTH = (TMT == 64);            // 64x64: every thread stages one A slot and one W slot; 32x32: half/half
  __shared__ __attribute__((aligned(16))) float As[TK][TMT + 4];
  __shared__ __attribute__((aligned(16))) float Ws[TK][TNT + 4];
  __shared__ __attribute__((aligned(16))) float Ws2[DUAL ? TK : 1][TNT + 4];
  __shared__ float rs[TMT];
  const int tid = threadIdx.x;
  const int tx = tid & 15, ty = tid >> 4;
  const int m0 = blockIdx.y * TMT, n0 = blockIdx.x * TNT;
  const int M = a.m, N = a.n, K = a.k;
  const WT* __restrict__ W = reinterpret_cast<const WT*>(a.w);
  const WT* __restrict__ W2 = reinterpret_cast<const WT*>(a.w2);

  if (a.pro == VV_PRO_RMSNORM) {
    constexpr int TPR = 256 / TMT;              // threads per row: 4 or 8
    const int r = tid / TPR, q = tid % TPR;
    float s = 0.f;
    if (m0 + r < M) {
      const float* xr = a.x + (int64_t)(m0 + r) * a.ldx;
      for (int k = q; k < K; k += TPR) { const float v = xr[k]; s += v * v; }
    }
#pragma unroll
    for (int o = 1; o < TPR; o <<= 1) s += __shfl_xor(s, o);
    if (q == 0) rs[r] = rsqrtf(s / (float)K + a.eps);
    __syncthreads();
  }

  float acc[MI][NJ], acc2[DUAL ? MI : 1][NJ];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j) { acc[i][j] = 0.f; if (DUAL) acc2[i][j] = 0.f; }

  // staging roles
  const bool doA = BOTH || tid < 128;
  const bool doW = BOTH || tid >= 128;
  const int st = BOTH ? tid : (tid & 127);
  const int lr = st >> 2, lk = (st & 3) * 4;

  auto load_slab = [&](int k0, float (&av)[4], float (&wv)[4], float (&wv2)[4]) {
#pragma unroll
    for (int i = 0; i < 4; ++i) { av[i] = 0.f; wv[i] = 0.f; wv2[i] = 0.f; }
    const int kb = k0 + lk;
    if (doA && m0 + lr < M) {
      const float* xr = a.x + (int64_t)(m0 + lr) * a.ldx + kb;
      if (VEC && kb + 3 < K) {
        const float4 v = *reinterpret_cast<const float4*>(xr);
        av[0] = v.x; av[1] = v.y; av[2] = v.z; av[3] = v.w;
      } else {
#pragma unroll
        for (int i = 0; i < 4; ++i) if (kb + i < K) av[i] = xr[i];
      }
      if (a.pro == VV_PRO_RMSNORM) {
        const float r = rs[lr];
#pragma unroll
        for (int i = 0; i < 4; ++i) if (kb + i < K) {
          float v = av[i] * r;
          if (a.norm_w) v *= a.norm_w[kb + i];
          if (a.mod_scale) v = v * (1.0f + a.mod_scale[(int64_t)(m0 + lr) * a.ld_mod + kb + i]) + a.mod_shift[(int64_t)(m0 + lr) * a.ld_mod + kb + i];
          av[i] = v;
        }
      } else if (a.pro == VV_PRO_SILU) {
#pragma unroll
        for (int i = 0; i < 4; ++i) av[i] = silu_f(av[i]);
      }
    }
    if (doW && n0 + lr < N) {
      const WT* wr = W + (int64_t)(n0 + lr) * K + kb;
      if (VEC && kb + 3 < K) {
        WL<WT>::load4(wr, wv);
        if (DUAL) WL<WT>::load4(W2 + (int64_t)(n0 + lr) * K + kb, wv2);
      } else {
#pragma unroll
        for (int i = 0; i < 4; ++i) if (kb + i < K) {
          wv[i] = WL<WT>::load1(wr + i);
          if (DUAL) wv2[i] = WL<WT>::load1(W2 + (int64_t)(n0 + lr) * K + kb + i);
        }
      }
    }
  };
  auto store_slab = [&](const float (&av)[4], const float (&wv)[4], const float (&wv2)[4]) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      if (doA) As[lk + i][lr] = av[i];
      if (doW) { Ws[lk + i][lr] = wv[i]; if (DUAL) Ws2[lk + i][lr] = wv2[i]; }
    }
  };

  float av[4], wv[4], wv2[4];
  load_slab(0, av, wv, wv2);
  store_slab(av, wv, wv2);
  __syncthreads();
  for (int k0 = 0; k0 < K; k0 += TK) {
    const bool more = k0 + TK < K;
    if (more) load_slab(k0 + TK, av, wv, wv2);          // in flight while this slab is multiplied
#pragma unroll
    for (int kk = 0; kk < TK; ++kk) {
      float ar[MI], br[NJ], br2[NJ];
      if constexpr (MI == 4) { const float4 v = *reinterpret_cast<const float4*>(&As[kk][ty * 4]); ar[0] = v.x; ar[1] = v.y; ar[MI - 2] = v.z; ar[MI - 1] = v.w; }
      else { const float2 v = *reinterpret_cast<const float2*>(&As[kk][ty * 2]); ar[0] = v.x; ar[1] = v.y; }
      if constexpr (NJ == 4) { const float4 v = *reinterpret_cast<const float4*>(&Ws[kk][tx * 4]); br[0] = v.x; br[1] = v.y; br[NJ - 2] = v.z; br[NJ - 1] = v.w; }
      else { const float2 v = *reinterpret_cast<const float2*>(&Ws[kk][tx * 2]); br[0] = v.x; br[1] = v.y; }
      if (DUAL) {
        if constexpr (NJ == 4) { const float4 v = *reinterpret_cast<const float4*>(&Ws2[kk][tx * 4]); br2[0] = v.x; br2[1] = v.y; br2[NJ - 2] = v.z; br2[NJ - 1] = v.w; }
        else { const float2 v = *reinterpret_cast<const float2*>(&Ws2[kk][tx * 2]); br2[0] = v.x; br2[1] = v.y; }
      }
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j) { acc[i][j] = fmaf(ar[i], br[j], acc[i][j]); if (DUAL) acc2[i][j] = fmaf(ar[i], br2[j], acc2[i][j]); }
    }
    if (more) {
      __syncthreads();
      store_slab(av, wv, wv2);
      __syncthreads();
    }
  }
#pragma unroll
  for (int i = 0; i < MI; ++i) {
    const int m = m0 + ty * MI + i;
    if (m >= M) continue;
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const int n = n0 + tx * NJ + j;
      if (n < N) lin_epilogue(a, m, n, acc[i][j], DUAL ? acc2[i][j] : 0.f);
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------
// launcher
// ---------------------------------------------------------------------------------------------------------------
template <typename WT, int M, bool DUAL>
static int launch_gemv(const vv_lin_args& a, hipStream_t s) {
  const int K = a.k, N = a.n;
  int KC = (8192 / M) & ~1023;                 // <= 32 KB of staged x
  if (KC < 1024) KC = 1024;
  if (KC > K) KC = (K + 7) & ~7;
  const size_t lds = (size_t)(256 + M * KC) * sizeof(float);
  const bool splitk = (N < 4096 && K >= 2048);
  if (splitk) {
    if (N >= 1024) { hipLaunchKernelGGL((gemv_kernel<WT, M, DUAL, 4, 2>), dim3((N + 1) / 2), dim3(GEMV_THREADS), lds, s, a, KC); }
    else           { hipLaunchKernelGGL((gemv_kernel<WT, M, DUAL, 4, 1>), dim3(N), dim3(GEMV_THREADS), lds, s, a, KC); }
  } else {
    if ((int64_t)N >= 8192 && M <= 4 && !(DUAL && M > 2)) {
      hipLaunchKernelGGL((gemv_kernel<WT, M, DUAL, 1, 2>), dim3((N + 7) / 8), dim3(GEMV_THREADS), lds, s, a, KC);
    } else {
      hipLaunchKernelGGL((gemv_kernel<WT, M, DUAL, 1, 1>), dim3((N + 3) / 4), dim3(GEMV_THREADS), lds, s, a, KC);
    }
  }
  return 0;
}

template <typename WT, bool DUAL>
static int launch_gemv_m(const vv_lin_args& a, hipStream_t s) {
  switch (a.m) {
    case 1: return launch_gemv<WT, 1, DUAL>(a, s);
    case 2: return launch_gemv<WT, 2, DUAL>(a, s);
    case 3: return launch_gemv<WT, 3, DUAL>(a, s);
    case 4: return launch_gemv<WT, 4, DUAL>(a, s);
    case 5: return launch_gemv<WT, 5, DUAL>(a, s);
    case 6: return launch_gemv<WT, 6, DUAL>(a, s);
    case 7: return launch_gemv<WT, 7, DUAL>(a, s);
    default: return launch_gemv<WT, 8, DUAL>(a, s);
  }
}

template <typename WT>
static int launch_linear(const vv_lin_args& a, hipStream_t s) {
  const bool dual = a.w2 != nullptr;
  const size_t wsz = sizeof(WT);
  const bool w_al16 = ((uintptr_t)a.w % 16 == 0) && (!dual || (uintptr_t)a.w2 % 16 == 0);
  if (vv_launch_skinny(a, s)) return 0;                          // a few rows x K = 512..2560, plain epilogue: the resampling convs (vv_convffn.hip)
  if (a.m <= 8) {
    if (a.m > 2 && g_rows_part) {                               // 3..8 rows on the matrix cores (process-wide scratch: see rows_scratch)
      const int rc = vv_launch_gemv_rows(a, g_rows_part, G_ROWS_PART_FLOATS, g_rows_tk, G_ROWS_TICKETS, s);
      if (rc) return rc < 0 ? rc : 0;
    }
    if (a.flags & VV_LIN_W_FRAG)                                // only the 3..8-row matrix-core GEMV reads the fragment-major layout
      return vv_set_error(VV_E_UNSUPPORTED, "vv_linear: VV_LIN_W_FRAG weights are read by the 3..8-row matrix-core GEMV only (m=%d n=%d k=%d not covered)", a.m, a.n, a.k);
    if (vv_launch_gemv_stream(a, s)) return 0;                  // bf16 weight-streaming fast path (<= 4 rows; 5..8 rows when K splits to <= 2 units per wave)
    if (a.m > 4 && a.wdt == VV_BF16 && a.ldx != 0) {
      // 5..8 rows not covered above: two streaming passes of <= 4 rows (the LDS-staged kernel below is LDS-bandwidth bound at M = 8)
      vv_lin_args lo = a, hi = a;
      lo.m = 4;
      hi.m = a.m - 4;
      hi.x = a.x + 4 * a.ldx;
      hi.out = a.out + 4 * a.ldo;
      if (a.res) hi.res = a.res + 4 * a.ldres;
      if (a.gate && a.gate_ld) hi.gate = a.gate + 4 * a.gate_ld;
      if (a.mod_scale) { hi.mod_scale = a.mod_scale + 4 * a.ld_mod; hi.mod_shift = a.mod_shift + 4 * a.ld_mod; }
      if (vv_launch_gemv_stream(lo, s)) {
        if (vv_launch_gemv_stream(hi, s)) return 0;
        return vv_set_error(VV_E_HIP, "vv_linear: split GEMV second half not covered");
      }
    }
    const bool fast = (a.k % 8 == 0) && w_al16 && ((a.k * wsz) % 16 == 0);
    if (fast) return dual ? launch_gemv_m<WT, true>(a, s) : launch_gemv_m<WT, false>(a, s);
    int blocks = (a.n + 3) / 4;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL((gemv_generic_kernel<WT>), dim3(blocks), dim3(GEMV_THREADS), 0, s, a);
    return 0;
  }
  {
    const int rc = vv_launch_mfma_gemm(a, s);     // bf16 weights: matrix-core path
    if (rc < 0) return rc;
    if (rc == 1) return 0;
    if (a.flags & (VV_LIN_X_BF16 | VV_LIN_OUT_BF16)) return vv_set_error(VV_E_UNSUPPORTED, "vv_linear: bf16 hand-off not covered for this shape/alignment");
  }
  const bool vec = (a.k % 4 == 0) && (a.ldx % 4 == 0) && ((uintptr_t)a.x % 16 == 0) && w_al16;
  const long big_tiles = (long)((a.n + 63) / 64) * ((a.m + 63) / 64);
  const bool small = big_tiles < 192;              // too few 64x64 tiles to fill 256 CUs: use 32x32 tiles
  const int tm = small ? 32 : 64;
  dim3 grid((a.n + tm - 1) / tm, (a.m + tm - 1) / tm);
  if (grid.y > 65535u) return vv_set_error(VV_E_UNSUPPORTED, "vv_linear: m=%d rows exceed one launch (split the call)", a.m);
#define VV_GEMM_LAUNCH(D, V)                                                                                  \
  do {                                                                                                        \
    if (small) hipLaunchKernelGGL((gemm_kernel<WT, D, V, 32, 32>), grid, dim3(256), 0, s, a);                 \
    else hipLaunchKernelGGL((gemm_kernel<WT, D, V, 64, 64>), grid, dim3(256), 0, s, a);                       \
  } while (0)
  if (dual) { if (vec) VV_GEMM_LAUNCH(true, true); else VV_GEMM_LAUNCH(true, false); }
  else { if (vec) VV_GEMM_LAUNCH(false, true); else VV_GEMM_LAUNCH(false, false); }
#undef VV_GEMM_LAUNCH
  return 0;
}

// ---------------------------------------------------------------------------------------------------------------
// optional per-launch timing of vv_linear with HIP events (eager mode only; used by bench.py's roofline leg)
// ---------------------------------------------------------------------------------------------------------------
#include <vector>
namespace {
struct ProfRec { hipEvent_t e0, e1; int m, n, k, dual, wdt; };
std::vector<ProfRec> g_prof;
bool g_prof_on = false;
size_t g_prof_cap = 0;
}
extern "C" int vv_prof_begin(int max_records) {
  if (max_records <= 0) return vv_set_error(VV_E_ARG, "vv_prof_begin: max_records");
  g_prof.clear();
  g_prof.reserve(max_records);
  g_prof_cap = (size_t)max_records;
  g_prof_on = true;
  return 0;
}
extern "C" int vv_prof_end(vv_prof_entry* out, int max_out, int* n_out) {
  g_prof_on = false;
  if (!out || !n_out) return vv_set_error(VV_E_ARG, "vv_prof_end: null");
  int n = 0;
  for (auto& r : g_prof) {
    float ms = 0.f;
    hipError_t e = hipEventSynchronize(r.e1);
    if (e == hipSuccess) e = hipEventElapsedTime(&ms, r.e0, r.e1);
    (void)hipEventDestroy(r.e0);
    (void)hipEventDestroy(r.e1);
    if (e != hipSuccess) continue;
    int j = 0;
    for (; j < n; ++j)
      if (out[j].m == r.m && out[j].n == r.n && out[j].k == r.k && out[j].dual == r.dual && out[j].wdt == r.wdt) break;
    if (j == n) {
      if (n >= max_out) continue;
      out[n].m = r.m; out[n].n = r.n; out[n].k = r.k; out[n].dual = r.dual; out[n].wdt = r.wdt; out[n].count = 0; out[n].total_ms = 0.0;
      ++n;
    }
    out[j].count += 1;
    out[j].total_ms += ms;
  }
  g_prof.clear();
  *n_out = n;
  return 0;
}

extern "C" int vv_linear(const vv_lin_args* a, vv_stream_t stream) {
  if (!a || !a->x || !a->w || !a->out) return vv_set_error(VV_E_ARG, "vv_linear: null pointer");
  if (a->m <= 0 || a->n <= 0 || a->k <= 0) return vv_set_error(VV_E_ARG, "vv_linear: bad shape m=%d n=%d k=%d", a->m, a->n, a->k);
  if (a->act == VV_ACT_SWIGLU && !a->w2) return vv_set_error(VV_E_ARG, "vv_linear: SWIGLU needs w2");
  if (a->act != VV_ACT_SWIGLU && a->w2) return vv_set_error(VV_E_ARG, "vv_linear: w2 given without SWIGLU");
  if (a->mod_scale && (!a->mod_shift || a->pro != VV_PRO_RMSNORM)) return vv_set_error(VV_E_ARG, "vv_linear: modulate needs RMSNORM prologue and shift");
  if (a->m > 8 && a->ldx == 0) return vv_set_error(VV_E_ARG, "vv_linear: broadcast rows (ldx=0) only for m<=8");
  if ((a->flags & VV_LIN_W_FRAG) && (a->wdt != VV_BF16 || a->m < 3 || a->m > 8 || a->n % 16 || a->k % 32))
    return vv_set_error(VV_E_ARG, "vv_linear: VV_LIN_W_FRAG needs bf16 weights, 3..8 rows, n %% 16 == 0 and k %% 32 == 0");
  if ((a->flags & (VV_LIN_X_BF16 | VV_LIN_OUT_BF16)) && (a->wdt != VV_BF16 || a->m <= 8 || a->k % 16))
    return vv_set_error(VV_E_ARG, "vv_linear: bf16 activation hand-off needs bf16 weights, m > 8 and k %% 16 == 0");
  hipStream_t s = (hipStream_t)stream;
  int rc;
  ProfRec pr;
  const bool prof = g_prof_on && g_prof.size() < g_prof_cap;
  if (prof) {
    pr.m = a->m; pr.n = a->n; pr.k = a->k; pr.dual = a->w2 != nullptr; pr.wdt = a->wdt;
    if (hipEventCreate(&pr.e0) != hipSuccess || hipEventCreate(&pr.e1) != hipSuccess) return vv_set_error(VV_E_HIP, "vv_linear: event create");
    (void)hipEventRecord(pr.e0, s);
  }
  if (a->wdt == VV_F32) rc = launch_linear<float>(*a, s);
  else if (a->wdt == VV_BF16) rc = launch_linear<bf16_t>(*a, s);
  else if (a->wdt == VV_FP8) {
    // weight-only fp8 exists for the weight-streaming GEMV alone (<= 2 rows); GEMM-shaped calls use the bf16 matrix
    rc = vv_launch_gemv_stream(*a, s) ? 0 : vv_set_error(VV_E_UNSUPPORTED, "vv_linear: fp8 weights need m <= 2, k %% 8 == 0, scales and 8-byte aligned rows (m=%d k=%d)", a->m, a->k);
  }
  else return vv_set_error(VV_E_ARG, "vv_linear: bad wdt %d", a->wdt);
  if (rc) return rc;
  if (prof) { (void)hipEventRecord(pr.e1, s); g_prof.push_back(pr); }
  VV_CHECK_LAUNCH("vv_linear");
  return 0;
}

// vv_linear for the composites of a row-batched step: 3..8 rows go to the matrix-core GEMV with the caller's split-K workspace, on the
// fragment-major copies f1 / f2 of w / w2 when the model has them; shapes that kernel does not cover take vv_linear on the row-major matrices
int vv_linear_ws(const vv_lin_args* a, const void* f1, const void* f2, float* part, size_t part_floats, int* tickets, size_t n_tickets, vv_stream_t stream) {
  if (a && a->wdt == VV_BF16 && a->m > 2 && a->m <= 8 && a->x && a->w && a->out) {
    vv_lin_args b = *a;
    if (f1 && (!b.w2 || f2)) { b.w = f1; if (b.w2) b.w2 = f2; b.flags |= VV_LIN_W_FRAG; }
    const int rc = vv_launch_gemv_rows(b, part, part_floats, tickets, n_tickets, (hipStream_t)stream);
    if (rc < 0) return rc;
    if (rc == 1) { VV_CHECK_LAUNCH("vv_linear(rows)"); return 0; }
  }
  return vv_linear(a, stream);
}

// ---------------------------------------------------------------------------------------------------------------
// RoPE + KV store, decode attention
// ---------------------------------------------------------------------------------------------------------------
template <typename KT> __device__ __forceinline__ void kv_store(KT* p, float v);
template <> __device__ __forceinline__ void kv_store<float>(float* p, float v) { *p = v; }
template <> __device__ __forceinline__ void kv_store<bf16_t>(bf16_t* p, float v) {
  const __hip_bfloat16 b = __float2bfloat16(v);
  *p = *reinterpret_cast<const bf16_t*>(&b);
}

// cos/sin of (position * inv_freq) for every row of a step: computed ONCE per LLM step (accurate cosf/sinf with large
// arguments cost hundreds of instructions) and shared by all layers' RoPE.  table[r][i] = {cos, sin}, i < head_dim/2.
__global__ void rope_table_kernel(const int* lens, const float* inv_freq, int half, float2* table) {
  const int r = blockIdx.x;
  const float pos = (float)lens[r];
  for (int i = threadIdx.x; i < half; i += blockDim.x) {
    const float ang = pos * inv_freq[i];
    table[(int64_t)r * half + i] = make_float2(cosf(ang), sinf(ang));
  }
}
extern "C" int vv_rope_table(const int* lens, const float* inv_freq, int R, int head_dim, float* table, vv_stream_t stream) {
  if (!lens || !inv_freq || !table || R <= 0 || head_dim <= 0 || head_dim % 2) return vv_set_error(VV_E_ARG, "vv_rope_table: bad args");
  hipLaunchKernelGGL(rope_table_kernel, dim3(R), dim3(64), 0, (hipStream_t)stream, lens, inv_freq, head_dim / 2, reinterpret_cast<float2*>(table));
  VV_CHECK_LAUNCH("vv_rope_table");
  return 0;
}

template <typename KT>
__global__ __launch_bounds__(256) void rope_store_kernel(float* qkv, int64_t ld, int heads, vv_kv kv, int layer,
                                                         const float2* rope, const int* lens, const int* cache_rows) {
  const int r = blockIdx.x;
  const int d = kv.head_dim, half = d >> 1;
  const int pos = lens[r];
  const int crow = cache_rows ? cache_rows[r] : r;
  float* row = qkv + (int64_t)r * ld;
  const int nq = heads * half, nk = kv.kv_heads * half;
  KT* kc = reinterpret_cast<KT*>(kv.k);
  KT* vc = reinterpret_cast<KT*>(kv.v);
  const int64_t base = (((int64_t)layer * kv.rows + crow) * kv.kv_heads) * kv.s_max * d;
  for (int idx = threadIdx.x; idx < nq + nk; idx += blockDim.x) {
    const bool isq = idx < nq;
    const int j = isq ? idx : idx - nq;
    const int h = j / half, i = j - h * half;
    float* p = row + (isq ? 0 : heads * d) + h * d;
    const float2 cs = rope[(int64_t)r * half + i];
    const float c = cs.x, s = cs.y;
    const float x1 = p[i], x2 = p[i + half];
    const float y1 = x1 * c - x2 * s;         // q*cos + rotate_half(q)*sin, first half: -x2
    const float y2 = x2 * c + x1 * s;         // second half: +x1
    if (isq) { p[i] = y1; p[i + half] = y2; }
    else {
      KT* dst = kc + base + ((int64_t)h * kv.s_max + pos) * d;
      kv_store<KT>(dst + i, y1);
      kv_store<KT>(dst + i + half, y2);
    }
  }
  const float* vrow = row + (heads + kv.kv_heads) * d;
  KT* vtc = reinterpret_cast<KT*>(kv.vt);                   // optional transposed copy in 32-key tiles [.., s_max / 32, d, 32] for the matrix-core attention kernels
  for (int idx = threadIdx.x; idx < kv.kv_heads * d; idx += blockDim.x) {
    const int h = idx / d, i = idx - h * d;
    kv_store<KT>(vc + base + ((int64_t)h * kv.s_max + pos) * d + i, vrow[idx]);
    if (vtc) kv_store<KT>(vtc + base + (int64_t)h * kv.s_max * d + (int64_t)(pos >> 5) * 32 * d + i * 32 + (pos & 31), vrow[idx]);
  }
}

extern "C" int vv_rope_store(float* qkv, int64_t ld_qkv, int R, int heads, const vv_kv* kv, int layer, const float* rope_table,
                             const int* lens, const int* cache_rows, vv_stream_t stream) {
  const float2* inv_freq = reinterpret_cast<const float2*>(rope_table);
  if (!qkv || !kv || !inv_freq || !lens) return vv_set_error(VV_E_ARG, "vv_rope_store: null pointer");
  if (layer < 0 || layer >= kv->layers || R <= 0) return vv_set_error(VV_E_ARG, "vv_rope_store: bad layer/R");
  if (kv->head_dim % 2) return vv_set_error(VV_E_ARG, "vv_rope_store: odd head_dim");
  hipStream_t s = (hipStream_t)stream;
  if (kv->kvdt == VV_F32) hipLaunchKernelGGL((rope_store_kernel<float>), dim3(R), dim3(256), 0, s, qkv, ld_qkv, heads, *kv, layer, inv_freq, lens, cache_rows);
  else hipLaunchKernelGGL((rope_store_kernel<bf16_t>), dim3(R), dim3(256), 0, s, qkv, ld_qkv, heads, *kv, layer, inv_freq, lens, cache_rows);
  VV_CHECK_LAUNCH("vv_rope_store");
  return 0;
}

template <typename KT, int EPL>
__device__ __forceinline__ void load_epl(const KT* p, float (&o)[EPL]) {
  if constexpr (EPL == 8) WL<KT>::load8(p, o);
  else WL<KT>::load4(p, o);
}

typedef unsigned int att_raw __attribute__((ext_vector_type(4)));     // 16 bytes of cache per lane: 8 bf16 or 4 fp32
template <typename KT, int EPL>
__device__ __forceinline__ void unpack_epl(const att_raw v, float (&o)[EPL]) {
  if constexpr (EPL == 8) {
    o[0] = __uint_as_float(v.x << 16); o[1] = __uint_as_float(v.x & 0xffff0000u);
    o[2] = __uint_as_float(v.y << 16); o[3] = __uint_as_float(v.y & 0xffff0000u);
    o[4] = __uint_as_float(v.z << 16); o[5] = __uint_as_float(v.z & 0xffff0000u);
    o[6] = __uint_as_float(v.w << 16); o[7] = __uint_as_float(v.w & 0xffff0000u);
  } else {
    o[0] = __uint_as_float(v.x); o[1] = __uint_as_float(v.y); o[2] = __uint_as_float(v.z); o[3] = __uint_as_float(v.w);
  }
}
// sum over the G lanes that share a key (G = 16: one DPP row, four VALU-speed steps instead of four ds_bpermute round trips)
__device__ __forceinline__ float group_dot_sum(float v, int G) {
  if (G == 16) {
#define VV_DPP_ADD(ctrl) v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), ctrl, 0xF, 0xF, true))
    VV_DPP_ADD(0xB1); VV_DPP_ADD(0x4E); VV_DPP_ADD(0x141); VV_DPP_ADD(0x140);
#undef VV_DPP_ADD
    return v;
  }
  for (int o = G >> 1; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

// One block per (query row, q head).  G lanes share one key (G * EPL == head_dim, EPL = 16 B of cache per lane),
// so a wave covers 64/G keys per step with fully coalesced 16-byte loads; online softmax per lane group,
// groups merged through LDS at the end.
template <typename KT, int EPL>
__global__ __launch_bounds__(256) void attn_kernel(const float* qkv, int64_t ld, int heads, vv_kv kv, int layer,
                                                   const int* lens, const int* cache_rows, float* out, int64_t ldo) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int r = blockIdx.y, h = blockIdx.x;
  const int d = kv.head_dim;
  const int G = d / EPL;                       // lanes per key
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int gl = lane % G, gi = lane / G;      // lane within group, group within wave
  const int KPW = 64 / G;                      // keys per wave step
  const int n_keys = lens[r] + 1;
  const int crow = cache_rows ? cache_rows[r] : r;
  const int kvh = h / (heads / kv.kv_heads);
  const int64_t base = ((((int64_t)layer * kv.rows + crow) * kv.kv_heads + kvh) * kv.s_max) * d;
  const KT* kc = reinterpret_cast<const KT*>(kv.k) + base;
  const KT* vc = reinterpret_cast<const KT*>(kv.v) + base;
  const float scale = rsqrtf((float)d);
  float q[EPL];
  const float* qp = qkv + (int64_t)r * ld + h * d + gl * EPL;
#pragma unroll
  for (int j = 0; j < EPL; ++j) q[j] = qp[j] * scale;
  float mmax = -INFINITY, lsum = 0.f, acc[EPL];
#pragma unroll
  for (int j = 0; j < EPL; ++j) acc[j] = 0.f;
  for (int s0 = wave * KPW; s0 < n_keys; s0 += 4 * KPW) {
    const int s = s0 + gi;
    const bool valid = s < n_keys;
    float kx[EPL], vx[EPL];
    const int sc = valid ? s : n_keys - 1;
    load_epl<KT, EPL>(kc + (int64_t)sc * d + gl * EPL, kx);
    load_epl<KT, EPL>(vc + (int64_t)sc * d + gl * EPL, vx);
    float dot = 0.f;
#pragma unroll
    for (int j = 0; j < EPL; ++j) dot = fmaf(q[j], kx[j], dot);
    for (int o = G >> 1; o > 0; o >>= 1) dot += __shfl_xor(dot, o);
    if (valid) {
      const float mn = fmaxf(mmax, dot);
      const float corr = expf(mmax - mn);      // exp(-inf) = 0 on the first key
      const float p = expf(dot - mn);
      lsum = lsum * corr + p;
#pragma unroll
      for (int j = 0; j < EPL; ++j) acc[j] = fmaf(p, vx[j], acc[j] * corr);
      mmax = mn;
    }
  }
  // merge the 4*KPW groups: sm = [ngroups][2 + d]
  const int ng = 4 * KPW;
  const int g = wave * KPW + gi;
  float* rec = sm + (int64_t)g * (d + 2);
  if (gl == 0) { rec[0] = mmax; rec[1] = lsum; }
#pragma unroll
  for (int j = 0; j < EPL; ++j) rec[2 + gl * EPL + j] = acc[j];
  __syncthreads();
  for (int i = tid; i < d; i += blockDim.x) {
    float M = -INFINITY;
    for (int gg = 0; gg < ng; ++gg) M = fmaxf(M, sm[(int64_t)gg * (d + 2)]);
    float num = 0.f, den = 0.f;
    for (int gg = 0; gg < ng; ++gg) {
      const float* rr = sm + (int64_t)gg * (d + 2);
      const float wgt = (rr[0] == -INFINITY) ? 0.f : expf(rr[0] - M);
      den = fmaf(rr[1], wgt, den);
      num = fmaf(rr[2 + i], wgt, num);
    }
    out[(int64_t)r * ldo + h * d + i] = num / den;
  }
}

// Prompt attention: one block per (query row, KV head).  The NQ query heads of a GQA group share every key / value load (the
// per-head kernel above re-reads the cache once per q head: at 330 rows x 12 heads that is ~340 MB of L2 traffic per layer and
// the whole cost of the kernel); loads run one batch of keys ahead of the arithmetic.
template <typename KT, int EPL, int NQ, int GT>      // GT: lanes per key when known at compile time (head_dim 128), 0 = from kv.head_dim
__global__ __launch_bounds__(256) void attn_group_kernel(const float* qkv, int64_t ld, vv_kv kv, int layer,
                                                         const int* lens, const int* cache_rows, float* out, int64_t ldo) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  constexpr int U = 2;
  const int r = blockIdx.y, kvh = blockIdx.x;
  const int d = GT ? GT * EPL : kv.head_dim;
  const int G = GT ? GT : d / EPL;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int gl = lane % G, gi = lane / G;
  const int KPW = 64 / G;
  const int n_keys = lens[r] + 1;
  const int crow = cache_rows ? cache_rows[r] : r;
  const int64_t base = ((((int64_t)layer * kv.rows + crow) * kv.kv_heads + kvh) * kv.s_max) * d;
  const KT* kc = reinterpret_cast<const KT*>(kv.k) + base + gl * EPL;
  const KT* vc = reinterpret_cast<const KT*>(kv.v) + base + gl * EPL;
  const float scale = rsqrtf((float)d);
  float q[NQ][EPL], acc[NQ][EPL], mmax[NQ], lsum[NQ];
#pragma unroll
  for (int hq = 0; hq < NQ; ++hq) {
    const float* qp = qkv + (int64_t)r * ld + (kvh * NQ + hq) * d + gl * EPL;
#pragma unroll
    for (int j = 0; j < EPL; ++j) { q[hq][j] = qp[j] * scale; acc[hq][j] = 0.f; }
    mmax[hq] = -INFINITY; lsum[hq] = 0.f;
  }
  const int stride = 4 * KPW;
  att_raw kr[U], vr[U];
  auto issue = [&](int s0) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int sc = min(s0 + u * stride + gi, n_keys - 1);
      kr[u] = *reinterpret_cast<const att_raw*>(kc + (int64_t)sc * d);
      vr[u] = *reinterpret_cast<const att_raw*>(vc + (int64_t)sc * d);
    }
  };
  issue(wave * KPW);
  for (int s0 = wave * KPW; s0 < n_keys; s0 += U * stride) {
    att_raw kc_[U], vc_[U];
#pragma unroll
    for (int u = 0; u < U; ++u) { kc_[u] = kr[u]; vc_[u] = vr[u]; }
    if (s0 + U * stride < n_keys) issue(s0 + U * stride);
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const bool valid = s0 + u * stride + gi < n_keys;
      float kx[EPL], vx[EPL];
      unpack_epl<KT, EPL>(kc_[u], kx);
      unpack_epl<KT, EPL>(vc_[u], vx);
#pragma unroll
      for (int hq = 0; hq < NQ; ++hq) {
        float dot = 0.f;
#pragma unroll
        for (int j = 0; j < EPL; ++j) dot = fmaf(q[hq][j], kx[j], dot);
        dot = group_dot_sum(dot, G);
        // the running maximum moves O(log n) times: the rescale is skipped (wave-uniformly) when no key group's maximum moved
        const float mn = valid ? fmaxf(mmax[hq], dot) : mmax[hq];
        if (__builtin_amdgcn_ballot_w64(mn != mmax[hq]) != 0) {
          const float corr = __expf(mmax[hq] - mn);      // exp(-inf) = 0 on the first key; 1 where the maximum stayed
          lsum[hq] *= (mn == mmax[hq]) ? 1.f : corr;
#pragma unroll
          for (int j = 0; j < EPL; ++j) acc[hq][j] *= (mn == mmax[hq]) ? 1.f : corr;
          mmax[hq] = mn;
        }
        const float pr = valid ? __expf(dot - mn) : 0.f;
        lsum[hq] += pr;
#pragma unroll
        for (int j = 0; j < EPL; ++j) acc[hq][j] = fmaf(pr, vx[j], acc[hq][j]);
      }
    }
  }
  // merge the 4*KPW key groups per q head: sm = [ngroups][NQ][2 + d]
  const int ng = 4 * KPW;
  const int g = wave * KPW + gi;
#pragma unroll
  for (int hq = 0; hq < NQ; ++hq) {
    float* rec = sm + ((int64_t)g * NQ + hq) * (d + 2);
    if (gl == 0) { rec[0] = mmax[hq]; rec[1] = lsum[hq]; }
#pragma unroll
    for (int j = 0; j < EPL; ++j) rec[2 + gl * EPL + j] = acc[hq][j];
  }
  __syncthreads();
  // the merge weight of a (key group, q head) pair is computed once, not once per output element
  float* wts = sm + (int64_t)ng * NQ * (d + 2);                       // [NQ][ng]
  if (tid < NQ * ng) {
    const int hq = tid / ng, gg = tid - hq * ng;
    float M = -INFINITY;
    for (int g2 = 0; g2 < ng; ++g2) M = fmaxf(M, sm[((int64_t)g2 * NQ + hq) * (d + 2)]);
    const float mg = sm[((int64_t)gg * NQ + hq) * (d + 2)];
    wts[tid] = (mg == -INFINITY) ? 0.f : expf(mg - M);
  }
  __syncthreads();
  for (int i = tid; i < NQ * d; i += blockDim.x) {
    const int hq = i / d, e = i - hq * d;
    float num = 0.f, den = 0.f;
    for (int gg = 0; gg < ng; ++gg) {
      const float* rr = sm + ((int64_t)gg * NQ + hq) * (d + 2);
      const float wgt = wts[hq * ng + gg];
      den = fmaf(rr[1], wgt, den);
      num = fmaf(rr[2 + e], wgt, num);
    }
    out[(int64_t)r * ldo + (kvh * NQ + hq) * d + e] = num / den;
  }
}

template <int NQ>
bool launch_attn_group(const float* qkv, int64_t ld, const vv_kv* kv, int layer, const int* lens, const int* cache_rows, float* out,
                       int64_t ldo, int R, hipStream_t s) {
  const int d = kv->head_dim, epl = kv->kvdt == VV_F32 ? 4 : 8, ng = 4 * (64 / (d / epl));
  const size_t lds = ((size_t)ng * NQ * (d + 2) + (size_t)NQ * ng) * sizeof(float);
  if (lds > 65536 || NQ * ng > 256) return false;
  dim3 grid(kv->kv_heads, R);
  if (kv->kvdt == VV_F32) {
    if (d == 128) hipLaunchKernelGGL((attn_group_kernel<float, 4, NQ, 32>), grid, dim3(256), lds, s, qkv, ld, *kv, layer, lens, cache_rows, out, ldo);
    else hipLaunchKernelGGL((attn_group_kernel<float, 4, NQ, 0>), grid, dim3(256), lds, s, qkv, ld, *kv, layer, lens, cache_rows, out, ldo);
  } else {
    if (d == 128) hipLaunchKernelGGL((attn_group_kernel<bf16_t, 8, NQ, 16>), grid, dim3(256), lds, s, qkv, ld, *kv, layer, lens, cache_rows, out, ldo);
    else hipLaunchKernelGGL((attn_group_kernel<bf16_t, 8, NQ, 0>), grid, dim3(256), lds, s, qkv, ld, *kv, layer, lens, cache_rows, out, ldo);
  }
  return true;
}

int g_attn_group = 1;      // tuning hook "attn_group": 0 = one block per q head for every row count

extern "C" int vv_attn(const float* qkv, int64_t ld_qkv, int R, int heads, const vv_kv* kv, int layer, const int* lens,
                       const int* cache_rows, float* out, int64_t ldo, vv_stream_t stream) {
  if (!qkv || !kv || !lens || !out) return vv_set_error(VV_E_ARG, "vv_attn: null pointer");
  if (layer < 0 || layer >= kv->layers || R <= 0 || heads % kv->kv_heads) return vv_set_error(VV_E_ARG, "vv_attn: bad layer/R/heads");
  const int d = kv->head_dim;
  const int epl = kv->kvdt == VV_F32 ? 4 : 8;
  if (d % epl || 64 % (d / epl) || d / epl > 64) return vv_set_error(VV_E_UNSUPPORTED, "vv_attn: head_dim %d unsupported for kv dtype %d", d, kv->kvdt);
  const int G = d / epl, ng = 4 * (64 / G);
  const size_t lds = (size_t)ng * (d + 2) * sizeof(float);
  if (lds > 65536) return vv_set_error(VV_E_UNSUPPORTED, "vv_attn: LDS %zu too large", lds);
  hipStream_t s = (hipStream_t)stream;
  {   // prompt rows against a bf16 cache with a transposed value copy: both products on the matrix cores (vv_attn_prefill.hip)
    const int rc = vv_launch_attn_prefill(qkv, ld_qkv, R, heads, kv, layer, lens, cache_rows, out, ldo, s);
    if (rc) return rc < 0 ? rc : 0;
  }
  if (g_attn_group && R >= 8 && ((uintptr_t)qkv % 16 == 0) && (ld_qkv % 4 == 0)) {      // prompt-sized row counts: share K/V loads across the GQA group
    const int nq = heads / kv->kv_heads;
    bool done = false;
    if (nq == 2) done = launch_attn_group<2>(qkv, ld_qkv, kv, layer, lens, cache_rows, out, ldo, R, s);
    else if (nq == 6) done = launch_attn_group<6>(qkv, ld_qkv, kv, layer, lens, cache_rows, out, ldo, R, s);
    else if (nq == 7) done = launch_attn_group<7>(qkv, ld_qkv, kv, layer, lens, cache_rows, out, ldo, R, s);
    if (done) { VV_CHECK_LAUNCH("vv_attn"); return 0; }
  }
  dim3 grid(heads, R);
  if (kv->kvdt == VV_F32) hipLaunchKernelGGL((attn_kernel<float, 4>), grid, dim3(256), lds, s, qkv, ld_qkv, heads, *kv, layer, lens, cache_rows, out, ldo);
  else hipLaunchKernelGGL((attn_kernel<bf16_t, 8>), grid, dim3(256), lds, s, qkv, ld_qkv, heads, *kv, layer, lens, cache_rows, out, ldo);
  VV_CHECK_LAUNCH("vv_attn");
  return 0;
}

// ---------------------------------------------------------------------------------------------------------------
// Decode attention with RoPE and the KV append fused in (rows own distinct cache rows: the per-frame {positive, negative}
// step).  q and the new k are rotated in registers; keys 0..pos-1 come from the cache, the new token's k/v straight from
// the projection buffer (so no block depends on another block's cache write); the first q head of each kv group appends
// k, v at slot pos.  512 threads, 4 keys in flight per lane group: S ~ 500 is four round trips instead of thirty.
// ---------------------------------------------------------------------------------------------------------------
#define ATT_UNR 4
template <typename KT, int EPL, int GT>      // GT: lanes per key when known at compile time (head_dim 128), 0 = from kv.head_dim
__global__ __launch_bounds__(1024) void attn_fused_kernel(const float* qkv, int64_t ld, int heads, vv_kv kv, int layer,
                                                         const float2* rope, const int* lens, float* out, int64_t ldo) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int r = blockIdx.y, h = blockIdx.x;
  const int d = GT ? GT * EPL : kv.head_dim, half = d >> 1;
  const int G = GT ? GT : d / EPL;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int NW = 16;
  const int gl = lane % G, gi = lane / G;
  const int KPW = 64 / G;
  const int pos = lens[r];
  const int gsz = heads / kv.kv_heads;
  const int kvh = h / gsz;
  const int64_t base = ((((int64_t)layer * kv.rows + r) * kv.kv_heads + kvh) * kv.s_max) * d;
  KT* kcw = reinterpret_cast<KT*>(kv.k) + base;
  KT* vcw = reinterpret_cast<KT*>(kv.v) + base;
  const KT* kc = kcw;
  const KT* vc = vcw;
  const float scale = rsqrtf((float)d);
  const float* row = qkv + (int64_t)r * ld;
  const int e0 = gl * EPL;
  // The cached keys / values do not depend on this step's q: the first batch (ATT_UNR keys per lane group, raw 16-byte slices)
  // is requested before q, the new k and the RoPE table are even loaded, and batch i+1 is requested before batch i is used.
  const int stride = NW * KPW * ATT_UNR;
  att_raw kraw[2][ATT_UNR], vraw[2][ATT_UNR];
  auto issue_kv = [&](int buf, int s0) {
#pragma unroll
    for (int u = 0; u < ATT_UNR; ++u) {
      const int sidx = s0 + u * KPW + gi;
      const int sc = sidx < pos ? sidx : 0;
      kraw[buf][u] = *reinterpret_cast<const att_raw*>(kc + (int64_t)sc * d + e0);
      vraw[buf][u] = *reinterpret_cast<const att_raw*>(vc + (int64_t)sc * d + e0);
    }
  };
  const int s_first = (wave * ATT_UNR) * KPW;
  if (s_first < pos) issue_kv(0, s_first);
  // rotate this lane's slice of q and of the new k (half rotation: pair (i, i + d/2))
  const bool lo = e0 < half;
  const int pe0 = lo ? e0 + half : e0 - half;
  float q[EPL], kn[EPL], vn[EPL];
  {
    const float* qp = row + h * d;
    const float* kp = row + (heads + kvh) * d;
    const float* vp = row + (heads + kv.kv_heads + kvh) * d;
#pragma unroll
    for (int j = 0; j < EPL; ++j) {
      const int fi = (lo ? e0 : pe0) + j;                 // frequency index in [0, d/2)
      const float2 cs = rope[(int64_t)r * half + fi];
      const float c = cs.x, sn = cs.y;
      const float qa = qp[e0 + j], qb = qp[pe0 + j];
      const float ka = kp[e0 + j], kb = kp[pe0 + j];
      q[j] = (lo ? qa * c - qb * sn : qa * c + qb * sn) * scale;
      kn[j] = lo ? ka * c - kb * sn : ka * c + kb * sn;
      vn[j] = vp[e0 + j];
    }
  }
  if (h % gsz == 0 && wave == 0 && gi == 0) {             // one writer per (row, kv head)
#pragma unroll
    for (int j = 0; j < EPL; ++j) {
      kv_store<KT>(kcw + (int64_t)pos * d + e0 + j, kn[j]);
      kv_store<KT>(vcw + (int64_t)pos * d + e0 + j, vn[j]);
    }
  }
  float mmax = -INFINITY, lsum = 0.f, acc[EPL];
#pragma unroll
  for (int j = 0; j < EPL; ++j) acc[j] = 0.f;
  auto update = [&](float dot, const float (&vx)[EPL]) {
    const float mn = fmaxf(mmax, dot);
    const float corr = expf(mmax - mn);
    const float p = expf(dot - mn);
    lsum = lsum * corr + p;
#pragma unroll
    for (int j = 0; j < EPL; ++j) acc[j] = fmaf(p, vx[j], acc[j] * corr);
    mmax = mn;
  };
  auto consume = [&](int buf, int s0) {
#pragma unroll
    for (int u = 0; u < ATT_UNR; ++u) {
      float kx[EPL], vx[EPL];
      unpack_epl<KT, EPL>(kraw[buf][u], kx);
      unpack_epl<KT, EPL>(vraw[buf][u], vx);
      float dot = 0.f;
#pragma unroll
      for (int j = 0; j < EPL; ++j) dot = fmaf(q[j], kx[j], dot);
      dot = group_dot_sum(dot, G);
      if (s0 + u * KPW + gi < pos) update(dot, vx);
    }
  };
  for (int s0 = s_first; s0 < pos; s0 += 2 * stride) {   // buffers have fixed roles: no register rotation
    if (s0 + stride < pos) issue_kv(1, s0 + stride);
    consume(0, s0);
    if (s0 + stride >= pos) break;
    if (s0 + 2 * stride < pos) issue_kv(0, s0 + 2 * stride);
    consume(1, s0 + stride);
  }
  {   // the new token itself (every block needs it; the cache copy may not be written yet by its owner block)
    float dot = 0.f;
#pragma unroll
    for (int j = 0; j < EPL; ++j) dot = fmaf(q[j], kn[j], dot);
    dot = group_dot_sum(dot, G);
    if (wave == 0 && gi == 0) update(dot, vn);
  }
  // merge the lane groups of this wave with shuffles (each step pairs groups `o` lanes apart), then the waves through LDS
  for (int o = G; o < 64; o <<= 1) {
    const float m2 = __shfl_xor(mmax, o), l2 = __shfl_xor(lsum, o);
    const float mn = fmaxf(mmax, m2);
    const float c1 = (mmax == -INFINITY) ? 0.f : expf(mmax - mn);
    const float c2 = (m2 == -INFINITY) ? 0.f : expf(m2 - mn);
    lsum = lsum * c1 + l2 * c2;
#pragma unroll
    for (int j = 0; j < EPL; ++j) { const float a2 = __shfl_xor(acc[j], o); acc[j] = acc[j] * c1 + a2 * c2; }
    mmax = mn;
  }
  float* rec = sm + (int64_t)wave * (d + 2);
  if (gi == 0) {
    if (gl == 0) { rec[0] = mmax; rec[1] = lsum; }
#pragma unroll
    for (int j = 0; j < EPL; ++j) rec[2 + e0 + j] = acc[j];
  }
  __syncthreads();
  for (int i = tid; i < d; i += blockDim.x) {
    float M = -INFINITY;
    for (int gg = 0; gg < NW; ++gg) M = fmaxf(M, sm[(int64_t)gg * (d + 2)]);
    float num = 0.f, den = 0.f;
    for (int gg = 0; gg < NW; ++gg) {
      const float* rr = sm + (int64_t)gg * (d + 2);
      const float wgt = (rr[0] == -INFINITY) ? 0.f : expf(rr[0] - M);
      den = fmaf(rr[1], wgt, den);
      num = fmaf(rr[2 + i], wgt, num);
    }
    out[(int64_t)r * ldo + h * d + i] = num / den;
  }
}

extern "C" int vv_attn_decode(const float* qkv, int64_t ld_qkv, int R, int heads, const vv_kv* kv, int layer, const float* rope_table,
                              const int* lens, float* out, int64_t ldo, vv_stream_t stream) {
  return vv_attn_decode_ws(qkv, ld_qkv, R, heads, kv, layer, rope_table, lens, out, ldo, nullptr, nullptr, 1, 1, stream);
}

int vv_attn_decode_ws(const float* qkv, int64_t ld_qkv, int R, int heads, const vv_kv* kv, int layer, const float* rope_table, const int* lens, float* out,
                      int64_t ldo, float* part, int* tickets, int nsplit, int part_cap, vv_stream_t stream) {
  const float2* inv_freq = reinterpret_cast<const float2*>(rope_table);
  if (!qkv || !kv || !lens || !out || !inv_freq) return vv_set_error(VV_E_ARG, "vv_attn_decode: null pointer");
  if (layer < 0 || layer >= kv->layers || R <= 0 || R > kv->rows || heads % kv->kv_heads) return vv_set_error(VV_E_ARG, "vv_attn_decode: bad layer/R/heads");
  const int d = kv->head_dim;
  const int epl = kv->kvdt == VV_F32 ? 4 : 8;
  if (d % 2 || d % epl || 64 % (d / epl) || d / epl > 64 || (d / 2) % epl) return vv_set_error(VV_E_UNSUPPORTED, "vv_attn_decode: head_dim %d unsupported", d);
  const size_t lds = (size_t)16 * (d + 2) * sizeof(float);
  if (lds > 65536) return vv_set_error(VV_E_UNSUPPORTED, "vv_attn_decode: LDS %zu too large", lds);
  hipStream_t s = (hipStream_t)stream;
  {   // the product shape (bf16 cache, head_dim 128): vv_attn_decode.hip
    const int rc = vv_launch_attn_decode(qkv, ld_qkv, R, heads, kv, layer, inv_freq, lens, out, ldo, part, tickets, nsplit, part_cap, s);
    if (rc) return rc < 0 ? rc : 0;
  }
  dim3 grid(heads, R);
  if (kv->kvdt == VV_F32) {
    if (d == 128) hipLaunchKernelGGL((attn_fused_kernel<float, 4, 32>), grid, dim3(1024), lds, s, qkv, ld_qkv, heads, *kv, layer, inv_freq, lens, out, ldo);
    else hipLaunchKernelGGL((attn_fused_kernel<float, 4, 0>), grid, dim3(1024), lds, s, qkv, ld_qkv, heads, *kv, layer, inv_freq, lens, out, ldo);
  } else {
    if (d == 128) hipLaunchKernelGGL((attn_fused_kernel<bf16_t, 8, 16>), grid, dim3(1024), lds, s, qkv, ld_qkv, heads, *kv, layer, inv_freq, lens, out, ldo);
    else hipLaunchKernelGGL((attn_fused_kernel<bf16_t, 8, 0>), grid, dim3(1024), lds, s, qkv, ld_qkv, heads, *kv, layer, inv_freq, lens, out, ldo);
  }
  VV_CHECK_LAUNCH("vv_attn_decode");
  return 0;
}

// ---------------------------------------------------------------------------------------------------------------
// Block1D mixer: out = x + gamma * (dwconv7(RMSNorm_c(x)) + b), channels-last, streaming history of normalised rows
// ---------------------------------------------------------------------------------------------------------------
#define MIX_TR 32                      // rows per tile
#define MIX_OWN ((MIX_TR + 6 + 3) / 4) // rows (incl. the 6-row halo) a lane can own when a wave covers one row per step
__global__ __launch_bounds__(256) void block_mixer_kernel(const float* __restrict__ x, float* __restrict__ out, int T, int C,
                                                          const float* norm_w, float eps, const float* dw_w, const float* dw_b,
                                                          const float* gamma, float* hist, int TR, int CS) {
  // grid = (row tiles of TR rows, channel slices of CS <= 64 channels).  A lane owns one channel of the slice and the rows
  // {wave*rpw + rl + 4*rpw*i}: everything it needs from memory (its channel's parameters, its x values / history values,
  // and the full rows for the RMS statistic) is requested in one burst at kernel entry, so the kernel is a single memory
  // round trip, a barrier, the 7-tap conv out of LDS and the store.
  extern __shared__ __attribute__((aligned(16))) float sm[];   // [TR + 6][CS] normalised rows, then [TR + 6] rstd
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int t0 = blockIdx.x * TR;
  const int tr = min(TR, T - t0);
  const int nrow = tr + 6;
  const int c0 = blockIdx.y * CS;
  const int cs = min(CS, C - c0);
  const int rpw = 64 / CS;
  const int cl = lane % CS, rl = lane / CS;
  const bool cv = cl < cs;
  const int cg = c0 + (cv ? cl : 0);
  float* rstd_s = sm + (int64_t)(TR + 6) * CS;
  const float nw = norm_w[cg], bb = dw_b[cg], gm = gamma[cg];
  float tap[7];
#pragma unroll
  for (int k = 0; k < 7; ++k) tap[k] = dw_w[cg * 7 + k];
  // own values: row rr = wave*rpw + rl + 4*rpw*i  (t = t0 - 6 + rr; t < 0 -> history row 6 + t, already normalised)
  float own[MIX_OWN];
#pragma unroll
  for (int i = 0; i < MIX_OWN; ++i) {
    const int rr = wave * rpw + rl + 4 * rpw * i;
    const int t = t0 - 6 + rr;
    own[i] = 0.f;
    if (rr < nrow && cv) own[i] = (t < 0) ? (hist ? hist[(int64_t)(6 + t) * C + cg] : 0.f) : x[(int64_t)t * C + cg];
  }
  // row statistics: a wave keeps 4 rows in flight, 16-byte loads, loads batched ahead of the adds (an un-unrolled
  // load->add loop would serialise one L2 round trip per iteration)
  const bool c4 = (C % 4 == 0);
  for (int rb = wave * 4; rb < nrow; rb += 16) {
    float ss[4] = {0.f, 0.f, 0.f, 0.f};
    const float* xr[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int t = t0 - 6 + rb + u;
      xr[u] = (rb + u < nrow && t >= 0) ? x + (int64_t)t * C : nullptr;
    }
    if (c4) {
#pragma unroll 4
      for (int c = lane * 4; c < C; c += 256) {
#pragma unroll
        for (int u = 0; u < 4; ++u) if (xr[u]) {
          const float4 v = *reinterpret_cast<const float4*>(xr[u] + c);
          ss[u] += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
        }
      }
    } else {
#pragma unroll 4
      for (int c = lane; c < C; c += 64) {
#pragma unroll
        for (int u = 0; u < 4; ++u) if (xr[u]) { const float v = xr[u][c]; ss[u] = fmaf(v, v, ss[u]); }
      }
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const float tot = wave_sum_dpp(ss[u]);
      if (lane == 0 && rb + u < nrow) rstd_s[rb + u] = xr[u] ? rsqrtf(tot / (float)C + eps) : 1.f;   // history rows are stored normalised
    }
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < MIX_OWN; ++i) {
    const int rr = wave * rpw + rl + 4 * rpw * i;
    if (rr < nrow && cv) {
      const int t = t0 - 6 + rr;
      sm[(int64_t)rr * CS + cl] = (t < 0) ? own[i] : own[i] * rstd_s[rr] * nw;
    }
  }
  __syncthreads();
  // outputs: row tt (sm row tt + 6) is owned by the lane that owns sm row tt + 6: wave/rl pattern shifted by 6 rows, so the
  // residual x value is re-read from L1/L2 only when the owner differs (cheap: this block has just touched that line)
  for (int tt = wave * rpw + rl; tt < tr; tt += 4 * rpw) {
    if (!cv) continue;
    float s = bb;
#pragma unroll
    for (int k = 0; k < 7; ++k) s = fmaf(tap[k], sm[(int64_t)(tt + k) * CS + cl], s);
    const int64_t o = (int64_t)(t0 + tt) * C + cg;
    out[o] = x[o] + gm * s;
  }
  if (hist && blockIdx.x == 0) {
    // new history = last 6 rows of [old history ; normalised x rows], for this block's channel slice.  Row tile 0 is the
    // only reader of hist (values captured before the first barrier), so it is also the only writer; rows outside its tile
    // (T > TR) are re-normalised here.
    for (int j = wave; j < 6; j += 4) {
      const int src = T - 6 + j;             // row of x; negative -> old history row j+T
      float* dst = hist + (int64_t)j * C + c0;
      if (src < 0) {
        for (int c = lane; c < cs; c += 64) dst[c] = sm[(int64_t)(j + T) * CS + c];
      } else if (src < tr) {
        for (int c = lane; c < cs; c += 64) dst[c] = sm[(int64_t)(src + 6) * CS + c];
      } else {
        const float* xs = x + (int64_t)src * C;
        float q = 0.f;
        for (int c = lane; c < C; c += 64) { const float v = xs[c]; q = fmaf(v, v, q); }
        const float rstd = rsqrtf(wave_sum_dpp(q) / (float)C + eps);
        for (int c = lane; c < cs; c += 64) dst[c] = xs[c0 + c] * rstd * norm_w[c0 + c];
      }
    }
  }
}

// ---- block mixer, few-rows form (streaming frames of the wide stages: T <= 256 rows, 256 <= C <= 2048) -----------------------------
// The sliced kernel above is shaped for long sequences (voice prompts): per workgroup it makes ~6 dependent global round trips and
// recomputes full-row statistics once per channel slice.  A streaming frame of the wide stages is T = 1 / 8 / 40 / 200 rows: here a
// workgroup owns <= 8 rows x ALL channels, fetches its whole window (6 halo rows) with one batch of loads, reduces each row with
// DPP wave sums + fixed-order partials in LDS, keeps normalised and raw rows in LDS and writes the result: one global round trip
// plus the store.  Row tile 0 is the only reader and writer of the history, as above.
#define MIXR_NI 16                     // float4 a thread may hold: (TR + 6) * C / 4 / 256
__global__ __launch_bounds__(256) void mixer_rows_kernel(const float* __restrict__ x, float* __restrict__ out, int T, int C,
                                                         const float* __restrict__ norm_w, float eps, const float* __restrict__ dw_w,
                                                         const float* __restrict__ dw_b, const float* __restrict__ gamma, float* hist, int TR) {
  extern __shared__ __attribute__((aligned(16))) float msm[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int C4 = C >> 2;
  const int t0 = blockIdx.x * TR;
  const int rows = min(TR, T - t0), nrow = rows + 6;
  float* xn = msm;                                   // [TR + 6][C] normalised window
  float* xraw = xn + (size_t)(TR + 6) * C;           // [TR][C] raw rows of this tile (residual)
  float* part = xraw + (size_t)TR * C;               // [TR + 6][8] per-row partial sums of squares (C4 / 64 slots)
  const int nel = nrow * C4;                         // float4 elements of the window
  const int nslot = C4 >> 6;
  // C <= 1024: 256 % C4 == 0, a thread sees ONE column (4 channels) in every pass: its norm weight, taps, bias and layer scale
  // are requested up front, next to the window loads, instead of after the reductions
  const bool colfix = (256 % C4) == 0;
  const int cfix = 4 * (tid % C4);
  float4 nwf = make_float4(1.f, 1.f, 1.f, 1.f), dbf = nwf, gmf = nwf;
  float tapf[28];
  if (colfix) {
    nwf = *reinterpret_cast<const float4*>(norm_w + cfix);
    dbf = *reinterpret_cast<const float4*>(dw_b + cfix);
    gmf = *reinterpret_cast<const float4*>(gamma + cfix);
#pragma unroll
    for (int q = 0; q < 7; ++q) {
      const float4 tq = *reinterpret_cast<const float4*>(dw_w + (size_t)cfix * 7 + 4 * q);
      tapf[4 * q] = tq.x; tapf[4 * q + 1] = tq.y; tapf[4 * q + 2] = tq.z; tapf[4 * q + 3] = tq.w;
    }
  }
  float4 v[MIXR_NI];
#pragma unroll
  for (int i = 0; i < MIXR_NI; ++i) {
    const int e = tid + 256 * i;
    v[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (e < nel) {
      const int w = e / C4, c4 = e - w * C4, t = t0 - 6 + w;
      if (t >= 0) v[i] = *reinterpret_cast<const float4*>(x + (int64_t)t * C + 4 * c4);
      else if (hist) v[i] = *reinterpret_cast<const float4*>(hist + (int64_t)(6 + t) * C + 4 * c4);     // already normalised
    }
  }
#pragma unroll
  for (int i = 0; i < MIXR_NI; ++i) {
    const int e0 = 256 * i + 64 * wave;              // first element of this wave's 64: one row per wave step (C4 is a multiple of 64)
    if (e0 < nel) {
      const float s = wave_sum_dpp(v[i].x * v[i].x + v[i].y * v[i].y + v[i].z * v[i].z + v[i].w * v[i].w);
      const int w = e0 / C4;
      if (lane == 0) part[w * 8 + ((e0 - w * C4) >> 6)] = s;
    }
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < MIXR_NI; ++i) {
    const int e = tid + 256 * i;
    if (e < nel) {
      const int w = e / C4, c4 = e - w * C4, t = t0 - 6 + w;
      float4 o = v[i];
      if (t >= 0) {
        float ss = 0.f;
        for (int sl = 0; sl < nslot; ++sl) ss += part[w * 8 + sl];          // fixed order: deterministic
        const float rstd = rsqrtf(ss / (float)C + eps);
        const float4 nw = colfix ? nwf : *reinterpret_cast<const float4*>(norm_w + 4 * c4);
        o = make_float4(v[i].x * rstd * nw.x, v[i].y * rstd * nw.y, v[i].z * rstd * nw.z, v[i].w * rstd * nw.w);
        if (w >= 6) *reinterpret_cast<float4*>(xraw + (size_t)(w - 6) * C + 4 * c4) = v[i];
      }
      *reinterpret_cast<float4*>(xn + (size_t)w * C + 4 * c4) = o;
    }
  }
  __syncthreads();
  for (int o = tid; o < rows * C4; o += 256) {
    const int tt = o / C4, c4 = o - tt * C4, c0 = 4 * c4;
    float tap[28];
    if (colfix) {
#pragma unroll
      for (int q = 0; q < 28; ++q) tap[q] = tapf[q];
    } else {
#pragma unroll
      for (int q = 0; q < 7; ++q) {
        const float4 tq = *reinterpret_cast<const float4*>(dw_w + (size_t)c0 * 7 + 4 * q);
        tap[4 * q] = tq.x; tap[4 * q + 1] = tq.y; tap[4 * q + 2] = tq.z; tap[4 * q + 3] = tq.w;
      }
    }
    float4 s = colfix ? dbf : *reinterpret_cast<const float4*>(dw_b + c0);
#pragma unroll
    for (int k = 0; k < 7; ++k) {
      const float4 a = *reinterpret_cast<const float4*>(xn + (size_t)(tt + k) * C + c0);
      s.x = fmaf(tap[k], a.x, s.x); s.y = fmaf(tap[7 + k], a.y, s.y); s.z = fmaf(tap[14 + k], a.z, s.z); s.w = fmaf(tap[21 + k], a.w, s.w);
    }
    const float4 gm = colfix ? gmf : *reinterpret_cast<const float4*>(gamma + c0);
    const float4 r = *reinterpret_cast<const float4*>(xraw + (size_t)tt * C + c0);
    *reinterpret_cast<float4*>(out + (int64_t)(t0 + tt) * C + c0) = make_float4(r.x + gm.x * s.x, r.y + gm.y * s.y, r.z + gm.z * s.z, r.w + gm.w * s.w);
  }
  if (hist && blockIdx.x == 0) {
    // new history = last 6 rows of [old history ; normalised x rows]: window row (src + 6) when this tile's window holds it
    for (int j = wave; j < 6; j += 4) {
      const int src = T - 6 + j;
      float* dst = hist + (int64_t)j * C;
      if (src < rows) {
        for (int c = lane * 4; c < C; c += 256) *reinterpret_cast<float4*>(dst + c) = *reinterpret_cast<const float4*>(xn + (size_t)(src + 6) * C + c);
      } else {
        const float* xs = x + (int64_t)src * C;
        float q = 0.f;
        for (int c = lane * 4; c < C; c += 256) { const float4 a = *reinterpret_cast<const float4*>(xs + c); q += a.x * a.x + a.y * a.y + a.z * a.z + a.w * a.w; }
        const float rstd = rsqrtf(wave_sum_dpp(q) / (float)C + eps);
        for (int c = lane * 4; c < C; c += 256) {
          const float4 a = *reinterpret_cast<const float4*>(xs + c), nw = *reinterpret_cast<const float4*>(norm_w + c);
          *reinterpret_cast<float4*>(dst + c) = make_float4(a.x * rstd * nw.x, a.y * rstd * nw.y, a.z * rstd * nw.z, a.w * rstd * nw.w);
        }
      }
    }
  }
}

int vv_mixer_init() {
  if (hipFuncSetAttribute(reinterpret_cast<const void*>(mixer_rows_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024) != hipSuccess)
    return vv_set_error(VV_E_HIP, "vv_mixer_init: cannot raise the LDS limit");
  return 0;
}

static int g_mixer_rows = 1;
void vv_mixer_set_rows(int on) { g_mixer_rows = on; }

extern "C" int vv_block_mixer(const float* x, float* out, int T, int C, const float* norm_w, float eps, const float* dw_w,
                              const float* dw_b, const float* gamma, float* hist, vv_stream_t stream) {
  if (!x || !out || !norm_w || !dw_w || !dw_b || !gamma) return vv_set_error(VV_E_ARG, "vv_block_mixer: null pointer");
  if (x == out) return vv_set_error(VV_E_ARG, "vv_block_mixer: in-place not allowed (halo rows)");
  if (T <= 0 || C <= 0) return vv_set_error(VV_E_ARG, "vv_block_mixer: bad shape");
  {   // 16..256 rows x wide channels: one workgroup per 8 rows (mixer_rows_kernel); fewer rows keep the channel-sliced kernel
      // (a single workgroup is slower than 32 slices: 13 vs 7 us at T = 1, C = 2048)
    auto a16 = [](const void* q) { return ((uintptr_t)q % 16) == 0; };
    if (g_mixer_rows && T >= 16 && T <= 256 && C % 256 == 0 && C >= 256 && C <= 2048 && a16(x) && a16(out) && a16(norm_w) && a16(dw_w) && a16(dw_b) &&
        a16(gamma) && (!hist || a16(hist))) {
      int TRr = 8;
      while (TRr > 1 && ((TRr + 6) * (C / 4) > 256 * MIXR_NI || TRr / 2 >= T)) TRr >>= 1;
      if ((TRr + 6) * (C / 4) <= 256 * MIXR_NI && !(hist && T > TRr && TRr < 6)) {
        const size_t ldsb = ((size_t)(TRr + 6) * C + (size_t)TRr * C + (size_t)(TRr + 6) * 8) * sizeof(float);
        hipLaunchKernelGGL(mixer_rows_kernel, dim3((T + TRr - 1) / TRr), dim3(256), ldsb, (hipStream_t)stream, x, out, T, C, norm_w, eps,
                           dw_w, dw_b, gamma, hist, TRr);
        VV_CHECK_LAUNCH("vv_block_mixer");
        return 0;
      }
    }
  }
  int CS = 64;                                   // channel slice: power of two <= 64 that covers narrow stages exactly
  while (CS > 1 && CS / 2 >= C) CS >>= 1;
  int TR = T < MIX_TR ? T : MIX_TR;
  const size_t lds = (size_t)((TR + 6) * CS + (TR + 6)) * sizeof(float);
  if (hist && TR < 6 && T > TR)   // row tiles 1..5 would read hist while tile 0 rewrites it
    return vv_set_error(VV_E_UNSUPPORTED, "vv_block_mixer: streaming with C=%d needs T<=%d rows per call (got %d)", C, TR, T);
  dim3 grid((T + TR - 1) / TR, (C + CS - 1) / CS);
  if (grid.y > 65535u || grid.x > 2147483647u) return vv_set_error(VV_E_UNSUPPORTED, "vv_block_mixer: grid too large");
  hipLaunchKernelGGL(block_mixer_kernel, grid, dim3(256), lds, (hipStream_t)stream, x, out, T, C, norm_w, eps,
                     dw_w, dw_b, gamma, hist, TR, CS);
  VV_CHECK_LAUNCH("vv_block_mixer");
  return 0;
}

// pad[0:ctx] <- state; state <- last ctx rows of [state ; pad[ctx : ctx+T]]   (single block: ctx*C is tiny)
// Row r of the new state is row T + r of the virtual sequence V = [state ; new rows]: an OLD state row while T + r < ctx, a new row
// (pad[T + r], untouched by the prefix copy) otherwise - so every load of both copies is issued before the one barrier.
#define CTX_NI 8
__global__ __launch_bounds__(1024) void conv_ctx_kernel(float* pad, float* state, int ctx, int T, int C) {
  const int n = ctx * C;
  float a[CTX_NI], b[CTX_NI];
  const int tid = threadIdx.x;
  for (int base = 0; base < n; base += 1024 * CTX_NI) {
#pragma unroll
    for (int u = 0; u < CTX_NI; ++u) {
      const int i = base + tid + 1024 * u;
      if (i < n) {
        const int r = i / C;
        a[u] = state[i];
        b[u] = (T + r < ctx) ? state[(int64_t)T * C + i] : pad[(int64_t)T * C + i];
      }
    }
    __syncthreads();                                   // all reads of the old state are done before it is rewritten
#pragma unroll
    for (int u = 0; u < CTX_NI; ++u) {
      const int i = base + tid + 1024 * u;
      if (i < n) { pad[i] = a[u]; state[i] = b[u]; }
    }
    __syncthreads();
  }
}

extern "C" int vv_conv_ctx(float* pad, float* state, int ctx, int T, int C, vv_stream_t stream) {
  if (!pad || !state || ctx <= 0 || T <= 0 || C <= 0) return vv_set_error(VV_E_ARG, "vv_conv_ctx: bad args");
  hipLaunchKernelGGL(conv_ctx_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, pad, state, ctx, T, C);
  VV_CHECK_LAUNCH("vv_conv_ctx");
  return 0;
}

// ---------------------------------------------------------------------------------------------------------------
// small elementwise kernels
// ---------------------------------------------------------------------------------------------------------------
__global__ void affine_kernel(const float* x, float a, float b, float* out, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) out[i] = fmaf(a, x[i], b);
}
extern "C" int vv_affine(const float* x, float a, float b, float* out, int64_t n, vv_stream_t stream) {
  if (!x || !out || n <= 0) return vv_set_error(VV_E_ARG, "vv_affine: bad args");
  int blocks = (int)((n + 255) / 256); if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(affine_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, x, a, b, out, n);
  VV_CHECK_LAUNCH("vv_affine");
  return 0;
}

// out[(i*rows_b + j), :] = a[j, :] + b[i, :]   for i < rows/rows_b, j < rows_b      (c = cond_proj(cond)[j] + t_emb[i])
__global__ void add_rows_kernel(const float* a, int64_t lda, const float* b, int64_t ldb, float* out, int rows, int rows_b, int n) {
  const int r = blockIdx.x, i = r / rows_b, j = r - i * rows_b;
  for (int c = threadIdx.x; c < n; c += blockDim.x) out[(int64_t)r * n + c] = a[(int64_t)j * lda + c] + b[(int64_t)i * ldb + c];
}
__global__ void add_rows_silu_kernel(const float* a, int64_t lda, const float* b, int64_t ldb, float* out, int rows, int rows_b, int n) {
  const int r = blockIdx.x, i = r / rows_b, j = r - i * rows_b;
  for (int c = threadIdx.x; c < n; c += blockDim.x) out[(int64_t)r * n + c] = silu_f(a[(int64_t)j * lda + c] + b[(int64_t)i * ldb + c]);
}
extern "C" int vv_add_rows_silu(const float* a, int64_t lda, const float* b, int64_t ldb, float* out, int rows, int rows_b, int n, vv_stream_t stream) {
  if (!a || !b || !out || rows <= 0 || rows_b <= 0 || rows % rows_b || n <= 0) return vv_set_error(VV_E_ARG, "vv_add_rows_silu: bad args");
  hipLaunchKernelGGL(add_rows_silu_kernel, dim3(rows), dim3(256), 0, (hipStream_t)stream, a, lda, b, ldb, out, rows, rows_b, n);
  VV_CHECK_LAUNCH("vv_add_rows_silu");
  return 0;
}
// bf16 output: the operand of the hoisted adaLN GEMMs (rounded to bf16 by the matrix-core path anyway)
__global__ void add_rows_silu_bf16_kernel(const float* a, int64_t lda, const float* b, int64_t ldb, bf16_t* out, int rows, int rows_b, int n) {
  const int r = blockIdx.x, i = r / rows_b, j = r - i * rows_b;
  for (int c = threadIdx.x; c < n; c += blockDim.x) kv_store<bf16_t>(out + (int64_t)r * n + c, silu_f(a[(int64_t)j * lda + c] + b[(int64_t)i * ldb + c]));
}
extern "C" int vv_add_rows_silu_bf16(const float* a, int64_t lda, const float* b, int64_t ldb, void* out, int rows, int rows_b, int n, vv_stream_t stream) {
  if (!a || !b || !out || rows <= 0 || rows_b <= 0 || rows % rows_b || n <= 0) return vv_set_error(VV_E_ARG, "vv_add_rows_silu_bf16: bad args");
  hipLaunchKernelGGL(add_rows_silu_bf16_kernel, dim3(rows), dim3(256), 0, (hipStream_t)stream, a, lda, b, ldb, (bf16_t*)out, rows, rows_b, n);
  VV_CHECK_LAUNCH("vv_add_rows_silu_bf16");
  return 0;
}
extern "C" int vv_add_rows(const float* a, int64_t lda, const float* b, int64_t ldb, float* out, int rows, int rows_b, int n, vv_stream_t stream) {
  if (!a || !b || !out || rows <= 0 || rows_b <= 0 || rows % rows_b || n <= 0) return vv_set_error(VV_E_ARG, "vv_add_rows: bad args");
  hipLaunchKernelGGL(add_rows_kernel, dim3(rows), dim3(256), 0, (hipStream_t)stream, a, lda, b, ldb, out, rows, rows_b, n);
  VV_CHECK_LAUNCH("vv_add_rows");
  return 0;
}

template <typename WT>
__global__ void embed_row_kernel(const WT* table, int64_t hidden, const int* token, float* out) {
  const int64_t t = *token;
  for (int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; c < hidden; c += (int64_t)gridDim.x * blockDim.x)
    out[c] = WL<WT>::load1(table + t * hidden + c);
}
extern "C" int vv_embed_row(const void* table, int wdt, int64_t hidden, const int* token, float* out, vv_stream_t stream) {
  if (!table || !token || !out || hidden <= 0) return vv_set_error(VV_E_ARG, "vv_embed_row: bad args");
  const int blocks = (int)((hidden + 255) / 256);
  if (wdt == VV_F32) hipLaunchKernelGGL((embed_row_kernel<float>), dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const float*)table, hidden, token, out);
  else hipLaunchKernelGGL((embed_row_kernel<bf16_t>), dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)table, hidden, token, out);
  VV_CHECK_LAUNCH("vv_embed_row");
  return 0;
}

extern "C" int vv_gather_rows(const void* table, int wdt, int64_t hidden, const int* ids_host, int n, void* out, vv_stream_t stream) {
  if (!table || !ids_host || !out || n <= 0) return vv_set_error(VV_E_ARG, "vv_gather_rows: bad args");
  const size_t esz = wdt == VV_F32 ? 4 : 2;
  for (int i = 0; i < n; ++i) {
    hipError_t e = hipMemcpyAsync((char*)out + (size_t)i * hidden * esz, (const char*)table + (size_t)ids_host[i] * hidden * esz,
                                  hidden * esz, hipMemcpyDeviceToDevice, (hipStream_t)stream);
    if (e != hipSuccess) return vv_set_error(VV_E_HIP, "vv_gather_rows: %s", hipGetErrorString(e));
  }
  return 0;
}

// token = ids[argmax(logits)] with first-max-wins in ascending id order (torch.argmax over the masked full vocabulary,
// modeling_vibevoice_inference.py:486-496); forced_token >= 0 overrides the choice (bench / fixture schedules).
__global__ void argmax_ids_kernel(const float* logits, int n, const int* ids, int* token_out, const int* forced) {
  if (threadIdx.x == 0) {
    int best = 0;
    for (int i = 1; i < n; ++i) {
      if (logits[i] > logits[best] || (logits[i] == logits[best] && ids[i] < ids[best])) best = i;
    }
    const int f = forced ? *forced : -1;
    *token_out = f >= 0 ? f : ids[best];
  }
}
extern "C" int vv_argmax_ids(const float* logits, int n, const int* ids, int* token_out, const int* forced_token, vv_stream_t stream) {
  if (!logits || !ids || !token_out || n <= 0) return vv_set_error(VV_E_ARG, "vv_argmax_ids: bad args");
  hipLaunchKernelGGL(argmax_ids_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, logits, n, ids, token_out, forced_token);
  VV_CHECK_LAUNCH("vv_argmax_ids");
  return 0;
}

__global__ void dpm_step_kernel(const float* v, int64_t ldv, int ns, int latent, float cfg, float alpha_s, float sigma_s, float cx,
                                float cd, float rinv, int order, float* x, float* m_prev) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= ns * latent) return;
  const int smp = i / latent, c = i - smp * latent;
  const float vc = v[(int64_t)smp * ldv + c], vu = v[(int64_t)(ns + smp) * ldv + c];
  const float eps = vu + cfg * (vc - vu);
  const float xx = x[i];
  const float x0 = alpha_s * xx - sigma_s * eps;
  float xn = cx * xx - cd * x0;
  if (order == 2) xn -= 0.5f * cd * (rinv * (x0 - m_prev[i]));
  x[i] = xn;
  m_prev[i] = x0;
}
extern "C" int vv_dpm_step(const float* v, int64_t ldv, int n_samples, int latent, float cfg_scale, float alpha_s, float sigma_s,
                           float cx, float cd, float rinv, int order, float* x, float* m_prev, vv_stream_t stream) {
  if (!v || !x || !m_prev || n_samples <= 0 || latent <= 0) return vv_set_error(VV_E_ARG, "vv_dpm_step: bad args");
  const int n = n_samples * latent;
  hipLaunchKernelGGL(dpm_step_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, v, ldv, n_samples, latent, cfg_scale,
                     alpha_s, sigma_s, cx, cd, rinv, order, x, m_prev);
  VV_CHECK_LAUNCH("vv_dpm_step");
  return 0;
}

// One launch per solver step boundary: the DPM-Solver++/CFG update of the previous step's output fused with the next
// step's noisy_images_proj (h = Wp x, K = latent = 64): every block recomputes the 64-element x (trivial), block 0 also
// stores x / x0 for the following step.  x and m_prev are double-buffered (x_in != x_out) so blocks that still read the
// old values never race with block 0's store.
template <typename WT>
__global__ __launch_bounds__(256) void dpm_proj_kernel(const float* v, int64_t ldv, int has_v, float cfg, vv_dpm_coef k,
                                                       const float* x_in, const float* m_in, float* x_out, float* m_out,
                                                       const WT* W, int latent, int D, float* h, int64_t ldh, int rows, const float* step_noise) {
  extern __shared__ float xs[];
  const int tid = threadIdx.x;
  for (int i = tid; i < latent; i += blockDim.x) {
    float xn = x_in[i];
    if (has_v) {
      const float vc = v[i], vu = v[ldv + i];
      const float eps = vu + cfg * (vc - vu);
      const float x0 = k.alpha_s * xn - k.sigma_s * eps;
      float xt = k.cx * xn - k.cd * x0;
      if (k.order == 2) xt -= 0.5f * k.cd * (k.rinv * (x0 - m_in[i]));
      if (step_noise) xt = fmaf(k.cn, step_noise[i], xt);      // SDE solver: variance noise of this step
      xn = xt;
      if (blockIdx.x == 0) m_out[i] = x0;
    }
    xs[i] = xn;
    if (blockIdx.x == 0) x_out[i] = xn;
  }
  __syncthreads();
  const int n = blockIdx.x * blockDim.x + tid;
  if (n < D && W) {
    const WT* wr = W + (int64_t)n * latent;
    float s = 0.f;
    for (int j = 0; j < latent; ++j) s = fmaf(WL<WT>::load1(wr + j), xs[j], s);
    for (int r = 0; r < rows; ++r) h[(int64_t)r * ldh + n] = s;
  }
}

extern "C" int vv_dpm_proj(const float* v, int64_t ldv, float cfg_scale, const vv_dpm_coef* coef, const float* x_in, const float* m_in,
                           float* x_out, float* m_out, const void* w, int wdt, int latent, int D, float* h, int64_t ldh, int rows,
                           const float* step_noise, vv_stream_t stream) {
  if (!x_in || !x_out || (v && (!coef || !m_in || !m_out)) || latent <= 0) return vv_set_error(VV_E_ARG, "vv_dpm_proj: bad args");
  if (x_in == x_out || (v && m_in == m_out)) return vv_set_error(VV_E_ARG, "vv_dpm_proj: x/m buffers must be double-buffered");
  if (w && (!h || D <= 0 || rows <= 0)) return vv_set_error(VV_E_ARG, "vv_dpm_proj: bad projection args");
  vv_dpm_coef k;
  memset(&k, 0, sizeof(k));
  if (v) k = *coef;
  if (v && k.cn != 0.f && !step_noise) return vv_set_error(VV_E_ARG, "vv_dpm_proj: the SDE solver step needs its variance noise");
  if (!v || k.cn == 0.f) step_noise = nullptr;
  const int blocks = w ? (D + 255) / 256 : 1;
  const size_t lds = (size_t)latent * sizeof(float);
  hipStream_t s = (hipStream_t)stream;
  if (wdt == VV_F32) hipLaunchKernelGGL((dpm_proj_kernel<float>), dim3(blocks), dim3(256), lds, s, v, ldv, v ? 1 : 0, cfg_scale, k, x_in, m_in, x_out, m_out, (const float*)w, latent, D, h, ldh, rows, step_noise);
  else hipLaunchKernelGGL((dpm_proj_kernel<bf16_t>), dim3(blocks), dim3(256), lds, s, v, ldv, v ? 1 : 0, cfg_scale, k, x_in, m_in, x_out, m_out, (const bf16_t*)w, latent, D, h, ldh, rows, step_noise);
  VV_CHECK_LAUNCH("vv_dpm_proj");
  return 0;
}

// device-side bookkeeping so a whole frame can be replayed as one graph:
// lens[0] (positive position) += 1; token == tok_start -> lens[1] = 0; token == tok_diffusion -> lens[1] += 1, frame += 1;
// tok_start < 0: lens[1] += 1 for every token (the negative branch is never refreshed)
__global__ void advance_lens_kernel(int* lens, const int* token, int tok_start, int tok_diff, int* frame) {
  if (threadIdx.x == 0) {
    const int t = *token;
    lens[0] += 1;
    if (tok_start < 0) { lens[1] += 1; if (t == tok_diff && frame) *frame += 1; }   // refresh_negative=False (modeling_vibevoice_inference.py:501-515)
    else if (t == tok_start) lens[1] = 0;
    else if (t == tok_diff) { lens[1] += 1; if (frame) *frame += 1; }
  }
}
extern "C" int vv_advance_lens(int* lens, const int* token, int tok_start, int tok_diffusion, int* frame_counter, vv_stream_t stream) {
  if (!lens || !token) return vv_set_error(VV_E_ARG, "vv_advance_lens: bad args");
  hipLaunchKernelGGL(advance_lens_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, lens, token, tok_start, tok_diffusion, frame_counter);
  VV_CHECK_LAUNCH("vv_advance_lens");
  return 0;
}

// rows of prologue(x) rounded to bf16: the activation operand of a matrix-core GEMM whose rows are many (prompt prefill), so
// the GEMM can stream both operands straight from global without staging.  prologue: RMSNorm (optional weight) or none.
__global__ __launch_bounds__(256) void cast_rows_bf16_kernel(const float* x, int64_t ldx, const float* norm_w, float eps, int pro, int n,
                                                             bf16_t* out, int64_t ldo) {
  __shared__ float red[4];
  const int r = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float* xr = x + (int64_t)r * ldx;
  float rstd = 1.f;
  if (pro == VV_PRO_RMSNORM) {
    float s = 0.f;
#pragma unroll 4
    for (int c = tid; c < n; c += 256) { const float v = xr[c]; s = fmaf(v, v, s); }
    s = wave_sum_dpp(s);
    if (lane == 0) red[wave] = s;
    __syncthreads();
    rstd = rsqrtf((red[0] + red[1] + red[2] + red[3]) / (float)n + eps);
  }
#pragma unroll 4
  for (int c = tid; c < n; c += 256) {
    float v = xr[c] * rstd;
    if (pro == VV_PRO_RMSNORM && norm_w) v *= norm_w[c];
    kv_store<bf16_t>(out + (int64_t)r * ldo + c, v);
  }
}
extern "C" int vv_cast_rows_bf16(const float* x, int64_t ldx, int rows, int n, int pro, const float* norm_w, float eps, void* out, int64_t ldo,
                                 vv_stream_t stream) {
  if (!x || !out || rows <= 0 || n <= 0 || (pro != VV_PRO_NONE && pro != VV_PRO_RMSNORM)) return vv_set_error(VV_E_ARG, "vv_cast_rows_bf16: bad args");
  hipLaunchKernelGGL(cast_rows_bf16_kernel, dim3(rows), dim3(256), 0, (hipStream_t)stream, x, ldx, norm_w, eps, pro, n, (bf16_t*)out, ldo);
  VV_CHECK_LAUNCH("vv_cast_rows_bf16");
  return 0;
}

__global__ void copy_rows_kernel(const float* x, int64_t ldx, float* out, int64_t ldo, int n) {
  const int r = blockIdx.y;
  for (int c = blockIdx.x * blockDim.x + threadIdx.x; c < n; c += gridDim.x * blockDim.x) out[(int64_t)r * ldo + c] = x[(int64_t)r * ldx + c];
}
// long rows (the streaming-state snapshot of a speculative frame: ~1 MB in one row): 16-byte accesses, 4 per thread in flight, enough
// workgroups to pull from every CU (64 workgroups of scalar loads took 10 us for it)
__global__ __launch_bounds__(256) void copy_rows4_kernel(const float4* x, int64_t ldx4, float4* out, int64_t ldo4, int n4) {
  const int r = blockIdx.y;
  const float4* xr = x + (int64_t)r * ldx4;
  float4* orow = out + (int64_t)r * ldo4;
  const int c0 = (blockIdx.x * 256 + threadIdx.x), st = gridDim.x * 256;
  for (int c = c0; c < n4; c += 4 * st) {
    float4 v[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = xr[min(c + i * st, n4 - 1)];
#pragma unroll
    for (int i = 0; i < 4; ++i) if (c + i * st < n4) orow[c + i * st] = v[i];
  }
}
extern "C" int vv_copy_rows(const float* x, int64_t ldx, float* out, int64_t ldo, int rows, int n, vv_stream_t stream) {
  if (!x || !out || rows <= 0 || n <= 0) return vv_set_error(VV_E_ARG, "vv_copy_rows: bad args");
  if (n >= 16384 && n % 4 == 0 && ldx % 4 == 0 && ldo % 4 == 0 && ((uintptr_t)x % 16) == 0 && ((uintptr_t)out % 16) == 0 && rows <= 65535) {
    int b4 = (n / 4 + 1023) / 1024; if (b4 > 1024) b4 = 1024;
    hipLaunchKernelGGL(copy_rows4_kernel, dim3(b4, rows), dim3(256), 0, (hipStream_t)stream, reinterpret_cast<const float4*>(x), ldx / 4,
                       reinterpret_cast<float4*>(out), ldo / 4, n / 4);
    VV_CHECK_LAUNCH("vv_copy_rows");
    return 0;
  }
  int bx = (n + 255) / 256; if (bx > 64) bx = 64;
  hipLaunchKernelGGL(copy_rows_kernel, dim3(bx, rows), dim3(256), 0, (hipStream_t)stream, x, ldx, out, ldo, n);
  VV_CHECK_LAUNCH("vv_copy_rows");
  return 0;
}

// rows of RMSNorm (final LLM norm): out = x * rsqrt(mean x^2 + eps) * w
__global__ __launch_bounds__(256) void rmsnorm_rows_kernel(const float* x, int64_t ldx, const float* w, float eps, int n, float* out, int64_t ldo) {
  __shared__ float red[4];
  const int r = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float* xr = x + (int64_t)r * ldx;
  float s = 0.f;
  for (int c = tid; c < n; c += blockDim.x) s += xr[c] * xr[c];
  s = wave_sum_dpp(s);
  if (lane == 0) red[wave] = s;
  __syncthreads();
  const float rstd = rsqrtf((red[0] + red[1] + red[2] + red[3]) / (float)n + eps);
  for (int c = tid; c < n; c += blockDim.x) out[(int64_t)r * ldo + c] = xr[c] * rstd * (w ? w[c] : 1.f);
}
int vv_rmsnorm_rows(const float* x, int64_t ldx, const float* w, float eps, int rows, int n, float* out, int64_t ldo, hipStream_t s) {
  hipLaunchKernelGGL(rmsnorm_rows_kernel, dim3(rows), dim3(256), 0, s, x, ldx, w, eps, n, out, ldo);
  VV_CHECK_LAUNCH("vv_rmsnorm_rows");
  return 0;
}

// ---------------------------------------------------------------------------------------------------------------
// hipGraph capture
// ---------------------------------------------------------------------------------------------------------------
extern "C" int vv_graph_begin(vv_stream_t stream) {
  hipError_t e = hipStreamBeginCapture((hipStream_t)stream, hipStreamCaptureModeThreadLocal);
  if (e != hipSuccess) return vv_set_error(VV_E_HIP, "hipStreamBeginCapture: %s", hipGetErrorString(e));
  return 0;
}
extern "C" int vv_graph_end(vv_stream_t stream, void** graph_exec_out) {
  hipGraph_t g = nullptr;
  hipError_t e = hipStreamEndCapture((hipStream_t)stream, &g);
  if (e != hipSuccess) return vv_set_error(VV_E_HIP, "hipStreamEndCapture: %s", hipGetErrorString(e));
  hipGraphExec_t ge = nullptr;
  e = hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
  (void)hipGraphDestroy(g);
  if (e != hipSuccess) return vv_set_error(VV_E_HIP, "hipGraphInstantiate: %s", hipGetErrorString(e));
  *graph_exec_out = (void*)ge;
  return 0;
}
extern "C" int vv_graph_launch(void* graph_exec, vv_stream_t stream) {
  hipError_t e = hipGraphLaunch((hipGraphExec_t)graph_exec, (hipStream_t)stream);
  if (e != hipSuccess) return vv_set_error(VV_E_HIP, "hipGraphLaunch: %s", hipGetErrorString(e));
  return 0;
}
extern "C" int vv_graph_destroy(void* graph_exec) {
  if (graph_exec) (void)hipGraphExecDestroy((hipGraphExec_t)graph_exec);
  return 0;
}
