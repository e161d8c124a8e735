// vv_convffn.hip — the middle stages of the streaming conv tokenizers (C = 256 / 512 channels, T = 200 / 40 rows per frame) as TWO
// launches per Block1D instead of three, each a single global-memory round trip.
//
// Reference: Block1D.forward (vibevoice/modular/modular_vibevoice_tokenizer.py:555-600):
//     y   = x + gamma     * (dwconv7_causal(RMSNorm(x)) + b)          (mixer)
//     out = y + ffn_gamma * (W2 gelu(W1 RMSNorm(y) + b1) + b2)        (FFN, hidden width 4C)
// At these sizes a frame holds a few hundred KB of activations against 1-4 MB of weights per block: every kernel is a latency
// chain, not a bandwidth or FLOP problem.  The three-launch form (mixer, GEMM, GEMM on the general vv_linear kernels) spent
// 24-39 us per block on ~1 us of work: each kernel made 3-5 DEPENDENT trips to L2 (row statistics, staging pass, weight batches,
// epilogue parameters).  Here:
//   ffn_in_kernel   grid (4C/32 hidden blocks, row tiles of 32): every workgroup requests its weight fragments, its per-channel
//                   parameters and the 38-row window (6 halo rows) in ONE burst, then runs mixer -> RMSNorm -> bf16 image in LDS ->
//                   MFMA (K split over the 4 waves) -> GELU -> bf16 hidden tile.  The mixer is recomputed by every hidden block
//                   of a row tile (a few hundred FMAs per thread: cheaper than a launch); workgroup j writes columns
//                   [8j, 8j + 8) of y, the last row tile's workgroup 0 the new streaming history (to a scratch row set: the
//                   other workgroups may still be reading the old one).
//   ffn_out_kernel  grid (C/16 output blocks, row tiles of 16) on mfma_f32_16x16x32_bf16: all weight and activation fragments of
//                   a wave's K quarter are requested up front (<= 128 VGPRs), combine through LDS, out = y + ffn_gamma (. + b2).
//                   Workgroup (0, 0) moves the scratch history into place.
#include <hip/hip_runtime.h>
#include <hip/hip_bf16.h>
#include <stdint.h>

#include "vv_hip.h"
#include "vv_common.h"

namespace {

typedef unsigned short bf16_t;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// phase timing of workgroup (0, 0) / thread 0, debug builds only (-DVV_CF_TIMING, tools/convffn_phase.py)
#ifdef VV_CF_TIMING
__device__ unsigned long long g_cf_t[8];
#define CSTAMP(i) do { if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) { const long long t_ = wall_clock64(); g_cf_t[i] += (unsigned long long)(t_ - tprev_); tprev_ = t_; } } while (0)
#else
#define CSTAMP(i) do { } while (0)
#endif

constexpr int TR2 = 16;         // rows per workgroup of ffn_out (one 16 x 16 MFMA tile)
constexpr int HALO = 6;         // causal depthwise kernel 7

__device__ __forceinline__ unsigned int pack2(float a, float b) {
  const __hip_bfloat16 x = __float2bfloat16(a), y = __float2bfloat16(b);
  return (unsigned int)(*reinterpret_cast<const bf16_t*>(&x)) | ((unsigned int)(*reinterpret_cast<const bf16_t*>(&y)) << 16);
}
__device__ __forceinline__ float sq4(const float4 v) { return v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w; }

template <int C, int TRV, int NT = 256> struct InLay {    // TRV: sequence rows per workgroup (<= 32; the MFMA tile stays 32 wide); NT: threads
  static constexpr int VW = C >= 256 ? 4 : 2;             // floats per thread per column group (C = 128: 64 threads x float2 keep a wave inside one row)
  static constexpr int F4 = C / 4;                        // float4 columns per row (history copy)
  static constexpr int FV = C / VW;                       // column groups per row
  static constexpr int TPR = FV < NT ? FV : NT;           // threads per row
  static constexpr int NWV = NT / 64;                     // waves: they split the window rows in the mixer part and K in the GEMM
  static constexpr int CPT = FV / TPR;                    // column groups per thread per row (2 at C = 2048)
  static constexpr int RP = NT / TPR;                     // window rows per pass of the workgroup
  static constexpr int NI = (TRV + HALO + RP - 1) / RP;   // passes over the window (TRV + 6 rows)
  static constexpr int WR = NI * RP;                      // window rows held in LDS (>= TRV + 6: no row guards on the LDS side)
  static constexpr int WPR = TPR / 64;                    // waves per row (partial sums of squares per row)
  static constexpr int P1 = C + 8;                        // bf16 pitch of the FFN input image
  static constexpr int ST = C / (16 * NWV);               // MFMA steps of one wave (K split over the waves)
  static constexpr size_t XN = (size_t)WR * C * 4;
  static constexpr size_t RED = (size_t)NWV * 16 * 64 * 4;   // K-split partial accumulators (alias the window once it is consumed)
  static constexpr size_t XH = (size_t)(WR - HALO) * P1 * 2;
  static constexpr size_t PART = (size_t)2 * WR * 8 * 4;
  static constexpr size_t LDS = (XN > RED ? XN : RED) + XH + PART;
};

// VW-wide (4 or 2 floats) loads / stores of a thread's column group
template <int VW> __device__ __forceinline__ void ldv(const float* p, float (&o)[VW]) {
  if constexpr (VW == 4) { const float4 t = *reinterpret_cast<const float4*>(p); o[0] = t.x; o[1] = t.y; o[2] = t.z; o[3] = t.w; }
  else { const float2 t = *reinterpret_cast<const float2*>(p); o[0] = t.x; o[1] = t.y; }
}
template <int VW> __device__ __forceinline__ void stv(float* p, const float (&o)[VW]) {
  if constexpr (VW == 4) *reinterpret_cast<float4*>(p) = make_float4(o[0], o[1], o[2], o[3]);
  else *reinterpret_cast<float2*>(p) = make_float2(o[0], o[1]);
}

// HF32: the hidden tile is written in fp32 (the T = 1 stage, whose second GEMM stays on the weight-streaming GEMV)
template <int C, int TRV, bool HF32, int NT>
__global__ __launch_bounds__(NT) void ffn_in_kernel(const float* __restrict__ x, float* __restrict__ y, void* __restrict__ hidden_v,
                                                     float* __restrict__ hist_new, int T, const vv_block B, float eps) {
  using L = InLay<C, TRV, NT>;
  constexpr int TR = TRV, CPT = L::CPT, VW = L::VW, NWV = L::NWV;
#ifdef VV_CF_TIMING
  long long tprev_ = wall_clock64();
#endif
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float* xn = reinterpret_cast<float*>(smem);                                        // [WR][C] normalised window
  float* red = reinterpret_cast<float*>(smem);                                       // aliases xn after the conv
  bf16_t* xh = reinterpret_cast<bf16_t*>(smem + (L::XN > L::RED ? L::XN : L::RED));  // [WR - 6][P1] RMSNorm(y) in bf16
  float* part = reinterpret_cast<float*>(smem + (L::XN > L::RED ? L::XN : L::RED) + L::XH);   // [2][WR][8]
  float* part2 = part + L::WR * 8;
  // The wave index goes through readfirstlane: everything derived from it (window row, validity, LDS rows) is then wave-uniform for the
  // compiler - scalar address arithmetic and scalar selects.  The mixer part of this kernel is bound by instruction issue (one wave per
  // SIMD, tools/convffn_phase.py), so guards are written as selects / 0-1 factors on always-valid addresses, not as branches.
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n0 = blockIdx.x * 32;
  const int t0 = blockIdx.y * TR;
  const int rows = min(TR, T - t0);
  const int lm = lane & 31, hk = (lane >> 5) * 8;

  // ---- everything this workgroup needs from memory, in one burst, in the order the phases need it (loads return in order) -------
  const int rloc = wave / L::WPR, slot = wave % L::WPR;            // TPR >= 64: a wave sits inside one window row
  int c0[CPT];
#pragma unroll
  for (int j = 0; j < CPT; ++j) c0[j] = VW * (tid % L::TPR + L::TPR * j);
  const float* hb = B.hist ? B.hist : x;                           // history rows of a stateless call read x row 0 and are zeroed below
  const float hkeep = B.hist ? 1.f : 0.f;
  float own[L::NI][CPT][VW];
#pragma unroll
  for (int i = 0; i < L::NI; ++i) {
    const int w = rloc + L::RP * i, t = t0 - HALO + w;
    const bool isx = t >= 0;
    const float* src = isx ? x + (int64_t)min(t, T - 1) * C : hb + (int64_t)(B.hist ? HALO + t : 0) * C;
    const float keep = (w < rows + HALO) ? (isx ? 1.f : hkeep) : 0.f;
#pragma unroll
    for (int j = 0; j < CPT; ++j) {
      ldv<VW>(src + c0[j], own[i][j]);
#pragma unroll
      for (int e = 0; e < VW; ++e) own[i][j][e] *= keep;
    }
  }
  float nw[CPT][VW], db[CPT][VW], gm[CPT][VW], fw[CPT][VW];
  float tap[CPT][VW * 7];                                          // tap[j][7 c + k] of channels c0[j] .. c0[j] + VW - 1
#pragma unroll
  for (int j = 0; j < CPT; ++j) {
    ldv<VW>(B.norm_w + c0[j], nw[j]);
    ldv<VW>(B.dw_b + c0[j], db[j]);
    ldv<VW>(B.gamma + c0[j], gm[j]);
    ldv<VW>(B.ffn_norm_w + c0[j], fw[j]);
#pragma unroll
    for (int q = 0; q < 7; ++q) {
      float tq[VW];
      ldv<VW>(B.dw_w + (size_t)c0[j] * 7 + VW * q, tq);
#pragma unroll
      for (int e = 0; e < VW; ++e) tap[j][VW * q + e] = tq[e];
    }
  }
  u32x4 wf[L::ST];
  {
    const bf16_t* wr = reinterpret_cast<const bf16_t*>(B.w1) + (int64_t)(n0 + lm) * C + wave * (C / NWV) + hk;
#pragma unroll
    for (int s = 0; s < L::ST; ++s) wf[s] = *reinterpret_cast<const u32x4*>(wr + s * 16);
  }
  const int eg = wave & 3;                                         // epilogue (waves 0..3): this thread finishes channels n0 + 8 eg + 4 (lane >> 5) + {0..3}
  const float4 b1v = *reinterpret_cast<const float4*>(B.b1 + n0 + 8 * eg + 4 * (lane >> 5));
  __builtin_amdgcn_sched_barrier(0);                               // nothing below is scheduled in front of these requests

  CSTAMP(0);                                       // address arithmetic + load issue
  // ---- 1. RMS statistic of the window rows ------------------------------------------------------------------------------------
#pragma unroll
  for (int i = 0; i < L::NI; ++i) {
    const int w = rloc + L::RP * i;
    float q = 0.f;
#pragma unroll
    for (int j = 0; j < CPT; ++j)
#pragma unroll
      for (int e = 0; e < VW; ++e) q = fmaf(own[i][j][e], own[i][j][e], q);
    const float s = vv_wave_sum(q);
    if (lane == 0) part[w * 8 + slot] = s;
  }
  __syncthreads();
  CSTAMP(1);                                       // loads landed, row statistics, barrier
#pragma unroll
  for (int i = 0; i < L::NI; ++i) {
    const int w = rloc + L::RP * i, t = t0 - HALO + w;
    float ss = part[w * 8];
#pragma unroll
    for (int sl = 1; sl < L::WPR; ++sl) ss += part[w * 8 + sl];
    const bool isx = t >= 0;                                       // history rows are stored normalised; rows past the sequence are zeros
    const float rstd = isx ? rsqrtf(ss / (float)C + eps) : 1.f;
#pragma unroll
    for (int j = 0; j < CPT; ++j) {
      float o[VW];
#pragma unroll
      for (int e = 0; e < VW; ++e) o[e] = own[i][j][e] * rstd * (isx ? nw[j][e] : 1.f);
      stv<VW>(xn + w * C + c0[j], o);
    }
  }
  __syncthreads();
  CSTAMP(2);                                       // normalised window to LDS

  // ---- 2. mixer, then the FFN's RMS statistic ------------------------------------------------------------------------------------
  float y1[L::NI][CPT][VW];
#pragma unroll
  for (int i = 0; i < L::NI; ++i) {
#pragma unroll
    for (int j = 0; j < CPT; ++j)
#pragma unroll
      for (int e = 0; e < VW; ++e) y1[i][j][e] = 0.f;
    if (L::RP * i + L::RP - 1 < HALO) continue;                    // (compile time) halo rows only: no output row in this pass
    const int w = rloc + L::RP * i, tt = w - HALO;
    const int tc = max(tt, 0);
    const float ok = (tt >= 0 && tt < rows) ? 1.f : 0.f;
    float q = 0.f;
#pragma unroll
    for (int j = 0; j < CPT; ++j) {
      float s[VW];
#pragma unroll
      for (int e = 0; e < VW; ++e) s[e] = db[j][e];
#pragma unroll
      for (int k = 0; k < 7; ++k) {
        float v[VW];
        ldv<VW>(xn + (tc + k) * C + c0[j], v);
#pragma unroll
        for (int e = 0; e < VW; ++e) s[e] = fmaf(tap[j][7 * e + k], v[e], s[e]);
      }
#pragma unroll
      for (int e = 0; e < VW; ++e) {
        y1[i][j][e] = (own[i][j][e] + gm[j][e] * s[e]) * ok;
        q = fmaf(y1[i][j][e], y1[i][j][e], q);
      }
    }
    const float s2 = vv_wave_sum(q);
    if (lane == 0) part2[w * 8 + slot] = s2;
  }
  // the new streaming history = the last 6 rows of [old history ; normalised rows]: all inside the LAST row tile's window
  if (hist_new && blockIdx.x == 0 && t0 + TR >= T) {
    for (int e = tid; e < HALO * L::F4; e += NT) {
      const int j = e / L::F4, c4 = e - j * L::F4;
      *reinterpret_cast<float4*>(hist_new + (size_t)j * C + 4 * c4) = *reinterpret_cast<const float4*>(xn + (size_t)(T - t0 + j) * C + 4 * c4);
    }
  }
  __syncthreads();
  CSTAMP(3);                                       // conv + second statistic
  {
    const int ys0 = blockIdx.x * (C / (int)gridDim.x);              // this workgroup's columns of y
#pragma unroll
    for (int i = 0; i < L::NI; ++i) {
      if (L::RP * i + L::RP - 1 < HALO) continue;
      const int w = rloc + L::RP * i, tt = w - HALO;
      if (tt >= 0) {                                                // (wave-uniform; rows past the end of the sequence hold zeros)
        float ss = part2[w * 8];
#pragma unroll
        for (int sl = 1; sl < L::WPR; ++sl) ss += part2[w * 8 + sl];
        const float rstd = rsqrtf(ss / (float)C + eps);
#pragma unroll
        for (int j = 0; j < CPT; ++j) {
          if constexpr (VW == 4) {
            uint2 p;
            p.x = pack2(y1[i][j][0] * rstd * fw[j][0], y1[i][j][1] * rstd * fw[j][1]);
            p.y = pack2(y1[i][j][2] * rstd * fw[j][2], y1[i][j][3] * rstd * fw[j][3]);
            *reinterpret_cast<uint2*>(xh + tt * L::P1 + c0[j]) = p;
          } else {
            *reinterpret_cast<unsigned int*>(xh + tt * L::P1 + c0[j]) = pack2(y1[i][j][0] * rstd * fw[j][0], y1[i][j][1] * rstd * fw[j][1]);
          }
          const bool mine = c0[j] >= ys0 && c0[j] < ys0 + C / (int)gridDim.x;
          if (mine && tt < rows) stv<VW>(y + (int64_t)(t0 + tt) * C + c0[j], y1[i][j]);
        }
      }
    }
  }
  __syncthreads();

  CSTAMP(4);                                       // bf16 image + y slice
  // ---- 3. hidden tile = gelu(W1 xh + b1): 32 channels x 32 rows, K split over the waves -------------------------------------------
  f32x16 acc;
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;
  {
    const bf16_t* xf = xh + min(lm, TR - 1) * L::P1 + wave * (C / NWV) + hk;   // tile columns past TR repeat a row (never stored)
#pragma unroll
    for (int s = 0; s < L::ST; ++s) {
      const u32x4 xb = *reinterpret_cast<const u32x4*>(xf + s * 16);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, wf[s]), __builtin_bit_cast(bf16x8, xb), acc, 0, 0, 0);
    }
  }
#pragma unroll
  for (int r = 0; r < 16; ++r) red[(wave * 16 + r) * 64 + lane] = acc[r];      // xn is dead: every read of it sits before the last barrier
  __syncthreads();
  CSTAMP(5);                                       // MFMA + partials
  {
    const int m = lane & 31;
    if (m < rows && wave < 4) {
      float v[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        float s = 0.f;
#pragma unroll
        for (int w4 = 0; w4 < NWV; ++w4) s += red[(w4 * 16 + 4 * eg + i) * 64 + lane];   // fixed order: deterministic
        v[i] = s;
      }
      const float g0 = vv_gelu_as(v[0] + b1v.x), g1 = vv_gelu_as(v[1] + b1v.y), g2 = vv_gelu_as(v[2] + b1v.z), g3 = vv_gelu_as(v[3] + b1v.w);
      const int64_t ho = (int64_t)(t0 + m) * (4 * C) + n0 + 8 * eg + 4 * (lane >> 5);
      if (HF32) {
        *reinterpret_cast<float4*>(reinterpret_cast<float*>(hidden_v) + ho) = make_float4(g0, g1, g2, g3);
      } else {
        uint2 p;
        p.x = pack2(g0, g1);
        p.y = pack2(g2, g3);
        *reinterpret_cast<uint2*>(reinterpret_cast<bf16_t*>(hidden_v) + ho) = p;
      }
    }
  }
  CSTAMP(6);                                       // combine + GELU + store
}


// ---- the one-row stage with the history part of the depthwise conv carried as state --------------------------------------------------------
// A frame is ONE row at C = 2048: conv(row) = b + sum_k<6 tap_k * hist_k + tap_6 * RMSNorm(x).  ffn_in_kernel<2048, 1> rebuilds the first two
// terms in each of its 256 workgroups from 6 history rows and 7 taps per channel: 105 KB of the 277 KB a workgroup requests.  With
// hs = sum_k<6 tap_k * hist_k kept next to the history (vv_block.hs, refreshed by the net's closing scatter from the rows it stores) and the
// newest row's tap packed (vv_block.dw_last) a workgroup needs 7 vectors of C floats and no window: y = x + gamma (b + hs + dw_last * xn).
__global__ __launch_bounds__(256) void ffn_in_row_kernel(const float* __restrict__ x, float* __restrict__ y, float* __restrict__ hidden,
                                                         float* __restrict__ hist_new, const vv_block B, float eps) {
  constexpr int C = 2048, P1 = C + 8, ST = C / 64;
  __shared__ __attribute__((aligned(16))) bf16_t xh[P1];
  __shared__ float red[4 * 16 * 64];
  __shared__ float part[8];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n0 = blockIdx.x * 32;
  const int lm = lane & 31, hk = (lane >> 5) * 8;
  float xv[2][4], hsv[2][4], dl[2][4], nw[2][4], db[2][4], gm[2][4], fw[2][4];
  int c0[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    c0[j] = 4 * (tid + 256 * j);
    ldv<4>(x + c0[j], xv[j]);
    ldv<4>(B.norm_w + c0[j], nw[j]);
    ldv<4>(B.hs + c0[j], hsv[j]);
    ldv<4>(B.dw_last + c0[j], dl[j]);
    ldv<4>(B.dw_b + c0[j], db[j]);
    ldv<4>(B.gamma + c0[j], gm[j]);
    ldv<4>(B.ffn_norm_w + c0[j], fw[j]);
  }
  const bool keeper = hist_new && blockIdx.x == 0;                 // workgroup 0 also shifts the history: rows 1..5 move up, the new row goes last
  float hrow[5][2][4];
  if (keeper) {
#pragma unroll
    for (int r = 0; r < 5; ++r)
#pragma unroll
      for (int j = 0; j < 2; ++j) ldv<4>(B.hist + (size_t)(r + 1) * C + c0[j], hrow[r][j]);
  }
  u32x4 wf[ST];
  {
    const bf16_t* wr = reinterpret_cast<const bf16_t*>(B.w1) + (int64_t)(n0 + lm) * C + wave * (C / 4) + hk;
#pragma unroll
    for (int s = 0; s < ST; ++s) wf[s] = *reinterpret_cast<const u32x4*>(wr + s * 16);
  }
  const int eg = wave;
  const float4 b1v = *reinterpret_cast<const float4*>(B.b1 + n0 + 8 * eg + 4 * (lane >> 5));
  __builtin_amdgcn_sched_barrier(0);
  float q = 0.f;
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int e = 0; e < 4; ++e) q = fmaf(xv[j][e], xv[j][e], q);
  q = vv_wave_sum(q);
  if (lane == 0) part[wave] = q;
  __syncthreads();
  const float rstd1 = rsqrtf(((part[0] + part[1]) + (part[2] + part[3])) / (float)C + eps);
  float yv[2][4], xn[2][4];
  float q2 = 0.f;
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      xn[j][e] = xv[j][e] * rstd1 * nw[j][e];
      const float sc = db[j][e] + hsv[j][e] + dl[j][e] * xn[j][e];
      yv[j][e] = xv[j][e] + gm[j][e] * sc;
      q2 = fmaf(yv[j][e], yv[j][e], q2);
    }
  q2 = vv_wave_sum(q2);
  if (lane == 0) part[4 + wave] = q2;
  if (keeper) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
#pragma unroll
      for (int r = 0; r < 5; ++r) stv<4>(hist_new + (size_t)r * C + c0[j], hrow[r][j]);
      stv<4>(hist_new + (size_t)5 * C + c0[j], xn[j]);
    }
  }
  __syncthreads();
  const float rstd2 = rsqrtf(((part[4] + part[5]) + (part[6] + part[7])) / (float)C + eps);
  {
    const int ys0 = blockIdx.x * (C / (int)gridDim.x);
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      uint2 p;
      p.x = pack2(yv[j][0] * rstd2 * fw[j][0], yv[j][1] * rstd2 * fw[j][1]);
      p.y = pack2(yv[j][2] * rstd2 * fw[j][2], yv[j][3] * rstd2 * fw[j][3]);
      *reinterpret_cast<uint2*>(xh + c0[j]) = p;
      if (c0[j] >= ys0 && c0[j] < ys0 + C / (int)gridDim.x) stv<4>(y + c0[j], yv[j]);
    }
  }
  __syncthreads();
  f32x16 acc;
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;
  {
    const bf16_t* xf = xh + wave * (C / 4) + hk;                   // every tile column reads the one row
#pragma unroll
    for (int s = 0; s < ST; ++s) {
      const u32x4 xb = *reinterpret_cast<const u32x4*>(xf + s * 16);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, wf[s]), __builtin_bit_cast(bf16x8, xb), acc, 0, 0, 0);
    }
  }
#pragma unroll
  for (int r = 0; r < 16; ++r) red[(wave * 16 + r) * 64 + lane] = acc[r];
  __syncthreads();
  if ((lane & 31) == 0) {
    float v[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      float s = 0.f;
#pragma unroll
      for (int w4 = 0; w4 < 4; ++w4) s += red[(w4 * 16 + 4 * eg + i) * 64 + lane];
      v[i] = s;
    }
    *reinterpret_cast<float4*>(hidden + n0 + 8 * eg + 4 * (lane >> 5)) =
        make_float4(vv_gelu_as(v[0] + b1v.x), vv_gelu_as(v[1] + b1v.y), vv_gelu_as(v[2] + b1v.z), vv_gelu_as(v[3] + b1v.w));
  }
}

// out[T, C] = res + ffn_gamma * (W2 hidden + b2), hidden bf16 [T, 4C]
template <int C, int NW>
__global__ __launch_bounds__(64 * NW) void ffn_out_kernel(const bf16_t* __restrict__ hidden, const float* res, float* out, const float* __restrict__ hist_new,
                                                          int T, const vv_block B) {
  constexpr int K = 4 * C, KW = K / NW, ST = KW / 32;              // a wave's share of K = KW columns = ST steps of 32
  __shared__ float red[NW * 4 * 64];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n0 = blockIdx.x * 16, t0 = blockIdx.y * TR2;
  const int r16 = lane & 15, kq = lane >> 4;
  u32x4 wf[ST], hf[ST];
  {
    const bf16_t* wr = reinterpret_cast<const bf16_t*>(B.w2) + (int64_t)(n0 + r16) * K + wave * KW + 8 * kq;
    const bf16_t* hr = hidden + (int64_t)min(t0 + r16, T - 1) * K + wave * KW + 8 * kq;
#pragma unroll
    for (int s = 0; s < ST; ++s) {
      wf[s] = *reinterpret_cast<const u32x4*>(wr + s * 32);
      hf[s] = *reinterpret_cast<const u32x4*>(hr + s * 32);
    }
  }
  // epilogue operands of this thread: accumulator register ei of lane el -> channel n0 + 4 (el >> 4) + ei, row t0 + (el & 15)
  const int ei = (tid >> 6) & 3, el = tid & 63;
  const int en = n0 + 4 * (el >> 4) + ei, em = t0 + (el & 15);
  const bool ev = em < T && tid < 256;
  const float b2v = B.b2[en], fgv = B.ffn_gamma[en];
  const float rv = ev ? res[(int64_t)em * C + en] : 0.f;
  // the streaming history the first kernel left in scratch moves into place (nobody reads B.hist any more in this block)
  const bool mover = hist_new && B.hist && blockIdx.x == 0 && blockIdx.y == 0;
  constexpr int HN = (HALO * C / 4 + 64 * NW - 1) / (64 * NW);
  float4 hv[HN];
  if (mover) {
#pragma unroll
    for (int i = 0; i < HN; ++i) {
      const int e = tid + 64 * NW * i;
      hv[i] = e < HALO * C / 4 ? *reinterpret_cast<const float4*>(hist_new + 4 * (size_t)e) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  }
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int s = 0; s < ST; ++s)
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wf[s]), __builtin_bit_cast(bf16x8, hf[s]), acc, 0, 0, 0);
#pragma unroll
  for (int i = 0; i < 4; ++i) red[(wave * 4 + i) * 64 + lane] = acc[i];
  __syncthreads();
  if (ev) {
    float s = 0.f;
#pragma unroll
    for (int w4 = 0; w4 < NW; ++w4) s += red[(w4 * 4 + ei) * 64 + el];         // fixed order: deterministic
    out[(int64_t)em * C + en] = rv + fgv * (s + b2v);
  }
  if (mover) {
#pragma unroll
    for (int i = 0; i < HN; ++i) {
      const int e = tid + 64 * NW * i;
      if (e < HALO * C / 4) *reinterpret_cast<float4*>(B.hist + 4 * (size_t)e) = hv[i];
    }
  }
}

// ---- resampling convs of a streaming frame as skinny GEMMs ------------------------------------------------------------------------------
// out[M, N] = x[M, K] W[N, K]^T + bias: fp32 rows (pitch ldx: the overlapping windows of a strided conv's padded input), bf16 weights,
// M <= a few hundred rows, K = 512 .. 2560.  The general kernels stage x through LDS in K chunks with a barrier pair per chunk
// (43 us for the 256 -> 512 stride-5 conv: 32 workgroups walking K = 2560).  Here a workgroup owns 16 rows x 16 channels on
// mfma_f32_16x16x32_bf16, K split over its 4 waves; a lane's B fragment is 8 consecutive floats of its row, converted in registers -
// no LDS image, no barrier before the MFMAs, every load of the kernel requested up front.
template <int NST, int NB>
__global__ __launch_bounds__(256) void skinny_kernel(const float* __restrict__ x, int64_t ldx, int M, const bf16_t* __restrict__ W, int K,
                                                     const float* __restrict__ bias, float* __restrict__ out, int64_t ldo) {
  // K = 4 waves x NB batches x NST steps of 32; a batch is one burst of loads (NB = 2 for K = 5120: 240 registers per burst)
  __shared__ float red[4 * 4 * 64];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n0 = blockIdx.x * 16, t0 = blockIdx.y * 16;
  const int r16 = lane & 15, kq = lane >> 4;
  const int kb = wave * (32 * NST * NB) + 8 * kq;
  const bf16_t* wr = W + (int64_t)(n0 + r16) * K + kb;
  const float* xr = x + (int64_t)min(t0 + r16, M - 1) * ldx + kb;
  const int ei = tid >> 6, el = tid & 63;
  const int en = n0 + 4 * (el >> 4) + ei, em = t0 + (el & 15);
  const float bv = bias ? bias[en] : 0.f;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int b = 0; b < NB; ++b) {
    u32x4 wf[NST];
    float4 xa[NST], xb[NST];
#pragma unroll
    for (int s = 0; s < NST; ++s) {
      wf[s] = *reinterpret_cast<const u32x4*>(wr + 32 * (b * NST + s));
      xa[s] = *reinterpret_cast<const float4*>(xr + 32 * (b * NST + s));
      xb[s] = *reinterpret_cast<const float4*>(xr + 32 * (b * NST + s) + 4);
    }
#pragma unroll
    for (int s = 0; s < NST; ++s) {
      u32x4 p;
      p.x = pack2(xa[s].x, xa[s].y); p.y = pack2(xa[s].z, xa[s].w); p.z = pack2(xb[s].x, xb[s].y); p.w = pack2(xb[s].z, xb[s].w);
      acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wf[s]), __builtin_bit_cast(bf16x8, p), acc, 0, 0, 0);
    }
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) red[(wave * 4 + i) * 64 + lane] = acc[i];
  __syncthreads();
  if (em < M) {
    float v = 0.f;
#pragma unroll
    for (int w4 = 0; w4 < 4; ++w4) v += red[(w4 * 4 + ei) * 64 + el];           // fixed order: deterministic
    out[(int64_t)em * ldo + en] = v + bv;
  }
}

int g_skinny = 1, g_skinny_min_m = 5, g_skinny_max_m = 2048;   // up to the T = 1600 rows of the 64 -> 32 / 32 -> 64 resampling convs

int g_on = 1;

// threads of ffn_in per channel count: the mixer part is instruction-issue bound, so the wide stages spread it over 8 waves
template <int C> constexpr int in_threads() { return (C == 512 || C == 1024) ? 512 : 256; }

template <int C, int TRV, int NW>
int launch_c(const vv_block& B, const float* x, float* y, void* hidden, float* hist_new, float* out, int T, float eps, hipStream_t s) {
  float* hn = B.hist ? hist_new : nullptr;
  constexpr int NT = in_threads<C>();
  constexpr size_t lds = InLay<C, TRV, NT>::LDS;
  hipLaunchKernelGGL((ffn_in_kernel<C, TRV, false, NT>), dim3(4 * C / 32, (T + TRV - 1) / TRV), dim3(NT), lds, s, x, y, hidden, hn, T, B, eps);
  hipLaunchKernelGGL((ffn_out_kernel<C, NW>), dim3(C / 16, (T + TR2 - 1) / TR2), dim3(64 * NW), 0, s, reinterpret_cast<const bf16_t*>(hidden), y, out, hn, T, B);
  return hipGetLastError() == hipSuccess ? 1 : vv_set_error(VV_E_HIP, "vv_convffn: launch failed");
}

// rows per workgroup of ffn_in.  Every hidden block of a row tile recomputes the tile's mixer, and with one wave per SIMD that part is
// bound by instruction issue (tools/convffn_phase.py: 9 of 12 us at C = 512 with 32-row tiles): short tiles cut it, at the price of
// re-reading W1 once per tile from L2.
int g_trv512 = 8, g_trv256 = 16, g_trv128 = 32;
int g_c128 = 1;       // the C = 128 stage on the two-launch kernels instead of the one-launch block kernel

}  // namespace

#ifdef VV_CF_TIMING
extern "C" int vv_convffn_debug_times(unsigned long long* out8, int reset) {
  if (hipMemcpyFromSymbol(out8, HIP_SYMBOL(g_cf_t), 8 * sizeof(unsigned long long)) != hipSuccess) return -1;
  if (reset) { unsigned long long z[8] = {0}; if (hipMemcpyToSymbol(HIP_SYMBOL(g_cf_t), z, sizeof(z)) != hipSuccess) return -1; }
  return 0;
}
#endif
void vv_convffn_set(int on) { g_on = on; }
void vv_convffn_set_rows(int c, int rows) { if (c == 512) g_trv512 = rows; else if (c == 256) g_trv256 = rows; else if (c == 128) g_trv128 = rows; }
void vv_convffn_set_c128(int on) { g_c128 = on; }
// the C = 128 stage of a streaming frame (T = 800 rows): two launches on 400 + 400 workgroups instead of the one-launch block kernel, whose
// 25 workgroups each pull the full 262 KB weight set through one CU
bool vv_convffn_prefers(int wdt, int T, int C) { return g_on && g_c128 && wdt == VV_BF16 && C == 128 && T >= 3 && T <= 1024; }
void vv_skinny_set(int on, int min_m, int max_m) { g_skinny = on; if (min_m > 0) g_skinny_min_m = min_m; if (max_m > 0) g_skinny_max_m = max_m; }

// 1 = launched, 0 = not covered
int vv_launch_skinny(const vv_lin_args& a, hipStream_t s) {
  if (!g_skinny || a.wdt != VV_BF16 || a.w2 || a.pro != VV_PRO_NONE || a.act != VV_ACT_NONE || a.gate || a.res || a.mod_scale || a.flags) return 0;
  if (a.m < g_skinny_min_m || a.m > g_skinny_max_m || a.n % 16 || a.ldx % 4 || a.ldx == 0 || ((uintptr_t)a.x % 16) || ((uintptr_t)a.w % 16)) return 0;
  dim3 grid(a.n / 16, (a.m + 15) / 16);
  const bf16_t* W = reinterpret_cast<const bf16_t*>(a.w);
#define VV_SK(NSTV, NBV) hipLaunchKernelGGL((skinny_kernel<NSTV, NBV>), grid, dim3(256), 0, s, a.x, a.ldx, a.m, W, a.k, a.bias, a.out, a.ldo)
  switch (a.k) {
    case 128: VV_SK(1, 1); break;
    case 256: VV_SK(2, 1); break;
    case 512: VV_SK(4, 1); break;
    case 1024: VV_SK(8, 1); break;
    case 2048: VV_SK(16, 1); break;
    case 2560: VV_SK(20, 1); break;
    case 5120: VV_SK(20, 2); break;
    default: return 0;
  }
#undef VV_SK
  return 1;
}

int vv_convffn_init() {
#define VV_CF_ATTR(CC, TT)                                                                                                                  \
  { constexpr int nt_ = in_threads<CC>(); constexpr int l_ = (int)InLay<CC, TT, nt_>::LDS;                                                 \
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&ffn_in_kernel<CC, TT, false, nt_>), hipFuncAttributeMaxDynamicSharedMemorySize, l_) != hipSuccess) \
      return vv_set_error(VV_E_HIP, "vv_convffn_init: cannot raise the LDS limit"); }
  VV_CF_ATTR(128, 16) VV_CF_ATTR(128, 32) VV_CF_ATTR(256, 8) VV_CF_ATTR(256, 16) VV_CF_ATTR(256, 32) VV_CF_ATTR(512, 8) VV_CF_ATTR(512, 16) VV_CF_ATTR(512, 32) VV_CF_ATTR(1024, 8)
#undef VV_CF_ATTR
  {
    constexpr int l1 = (int)InLay<2048, 1, 256>::LDS;
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&ffn_in_kernel<2048, 1, true, 256>), hipFuncAttributeMaxDynamicSharedMemorySize, l1) != hipSuccess)
      return vv_set_error(VV_E_HIP, "vv_convffn_init: cannot raise the LDS limit");
  }
  return 0;
}

int g_t1 = 1;
void vv_convffn_set_t1(int on) { g_t1 = on; }

// The first FFN half of a Block1D of the ONE-row stage (C = 2048, one row per frame): mixer + RMSNorm + W1 + GELU in one launch, hidden row
// in fp32 for the weight-streaming GEMV that follows; the new history goes to hist_new (6 * C floats of scratch that must survive until the
// caller has moved it into b.hist, after every workgroup of this launch is done).  1 = enqueued, 0 = not covered
int vv_launch_ffn_in_row(const vv_block& B, int wdt, const float* x, float* y, float* hidden, float* hist_new, int C, float eps, hipStream_t s) {
  if (!g_on || !g_t1 || wdt != VV_BF16 || C != 2048 || !B.hist) return 0;
  auto a16 = [](const void* q) { return q && ((uintptr_t)q % 16) == 0; };
  if (!a16(B.w1) || !a16(B.b1) || !a16(B.gamma) || !a16(B.norm_w) || !a16(B.ffn_norm_w) || !a16(B.dw_b) || !a16(B.dw_w) || !a16(x) || !a16(y) ||
      !a16(hidden) || !a16(hist_new) || !a16(B.hist) || x == y)
    return 0;
  constexpr size_t lds = InLay<2048, 1, 256>::LDS;
  hipLaunchKernelGGL((ffn_in_kernel<2048, 1, true, 256>), dim3(4 * 2048 / 32, 1), dim3(256), lds, s, x, y, hidden, hist_new, 1, B, eps);
  return hipGetLastError() == hipSuccess ? 1 : vv_set_error(VV_E_HIP, "vv_convffn: launch failed");
}

int g_t1hs = 1;
void vv_convffn_set_t1hs(int on) { g_t1hs = on; }

// the same with vv_block.hs / dw_last (see ffn_in_row_kernel); 1 = enqueued, 0 = not covered
int vv_launch_ffn_in_row_hs(const vv_block& B, int wdt, const float* x, float* y, float* hidden, float* hist_new, int C, float eps, hipStream_t s) {
  if (!g_on || !g_t1 || !g_t1hs || wdt != VV_BF16 || C != 2048 || !B.hist || !B.hs || !B.dw_last) return 0;
  auto a16 = [](const void* q) { return q && ((uintptr_t)q % 16) == 0; };
  if (!a16(B.w1) || !a16(B.b1) || !a16(B.gamma) || !a16(B.norm_w) || !a16(B.ffn_norm_w) || !a16(B.dw_b) || !a16(B.dw_last) || !a16(B.hs) || !a16(x) ||
      !a16(y) || !a16(hidden) || !a16(hist_new) || !a16(B.hist) || x == y)
    return 0;
  hipLaunchKernelGGL(ffn_in_row_kernel, dim3(4 * 2048 / 32), dim3(256), 0, s, x, y, hidden, hist_new, B, eps);
  return hipGetLastError() == hipSuccess ? 1 : vv_set_error(VV_E_HIP, "vv_convffn: launch failed");
}

// One Block1D of a middle stage: x[T, C] -> y[T, C] (mixer output, also the FFN residual) -> hidden (bf16 [T, 4C]) -> out[T, C]
// (out may be y).  hist_new: 6 * C floats of scratch.  1 = enqueued, 0 = not covered (caller runs mixer + two linears), < 0 = error
int vv_launch_convffn(const vv_block& B, int wdt, const float* x, float* y, void* hidden, float* hist_new, float* out, int T, int C, float eps,
                      hipStream_t s) {
  if (!g_on || wdt != VV_BF16 || T < 3 || (C != 128 && C != 256 && C != 512 && C != 1024) || (C == 1024 && T > 64)) return 0;
  if (C == 128 ? (!g_c128 || T > 1024) : T > 256) return 0;
  auto a16 = [](const void* q) { return q && ((uintptr_t)q % 16) == 0; };
  if (!a16(B.w1) || !a16(B.w2) || !a16(B.b1) || !B.b2 || !a16(B.gamma) || !B.ffn_gamma || !a16(B.norm_w) || !a16(B.ffn_norm_w) || !a16(B.dw_b) ||
      !a16(B.dw_w) || !a16(x) || !a16(y) || !a16(hidden) || !a16(hist_new) || !out || (B.hist && !a16(B.hist)))
    return 0;
  {   // workgroups read x (and halo rows) while others already write y / hidden: the ranges must be disjoint
    const uintptr_t xa = (uintptr_t)x, ya = (uintptr_t)y, bytes = (uintptr_t)T * C * 4;
    if (xa < ya + bytes && ya < xa + bytes) return 0;
  }
  if (C == 128) {
    if (g_trv128 == 16) return launch_c<128, 16, 4>(B, x, y, hidden, hist_new, out, T, eps, s);
    return launch_c<128, 32, 4>(B, x, y, hidden, hist_new, out, T, eps, s);
  }
  if (C == 256) {
    if (g_trv256 == 8) return launch_c<256, 8, 4>(B, x, y, hidden, hist_new, out, T, eps, s);
    if (g_trv256 == 16) return launch_c<256, 16, 4>(B, x, y, hidden, hist_new, out, T, eps, s);
    return launch_c<256, 32, 4>(B, x, y, hidden, hist_new, out, T, eps, s);
  }
  if (C == 512) {
    if (g_trv512 == 8) return launch_c<512, 8, 4>(B, x, y, hidden, hist_new, out, T, eps, s);
    if (g_trv512 == 16) return launch_c<512, 16, 4>(B, x, y, hidden, hist_new, out, T, eps, s);
    return launch_c<512, 32, 4>(B, x, y, hidden, hist_new, out, T, eps, s);
  }
  return launch_c<1024, 8, 8>(B, x, y, hidden, hist_new, out, T, eps, s);     // T = 8 rows per frame: 8-row tiles, K = 4096 over 8 waves
}

extern "C" size_t vv_block_mid_ws_bytes(int T, int C) { return (size_t)T * C * 4 + (size_t)T * 4 * C * 2 + (size_t)HALO * C * 4 + 64; }

extern "C" int vv_block_mid(const vv_block* b, int wdt, const float* x, float* out, void* ws, int T, int C, float eps, vv_stream_t stream) {
  if (!b || !x || !out || !ws || T <= 0) return vv_set_error(VV_E_ARG, "vv_block_mid: bad args");
  float* y = reinterpret_cast<float*>(ws);
  char* hidden = reinterpret_cast<char*>(ws) + (size_t)T * C * 4;
  float* hn = reinterpret_cast<float*>(hidden + (size_t)T * 4 * C * 2);
  const int rc = vv_launch_convffn(*b, wdt, x, y, hidden, hn, out, T, C, eps, (hipStream_t)stream);
  if (rc < 0) return rc;
  if (rc == 0) return vv_set_error(VV_E_UNSUPPORTED, "vv_block_mid: shape not covered (C=%d T=%d wdt=%d)", C, T, wdt);
  return 0;
}
