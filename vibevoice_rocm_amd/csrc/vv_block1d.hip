// vv_block1d.hip — one launch per Block1D of the conv tokenizers' narrow stages (C = 32 / 64 / 128 channels, bf16 weights).
//
// Reference: Block1D.forward (vibevoice/modular/modular_vibevoice_tokenizer.py:555-600):
//     x  = x  + gamma     * dwconv7_causal(RMSNorm(x))            (mixer)
//     x  = x  + ffn_gamma * W2 gelu(W1 RMSNorm(x) + b1) + b2       (FFN, hidden width 4C)
// These stages run T = 800 / 1600 / 3200 rows per frame against 8-130 KB of weights: as three launches (mixer, two GEMMs) each
// one is a ~6-8 us latency chain around a few hundred nanoseconds of work, and the 4C-wide hidden activation makes a round
// trip through global memory.  Here a workgroup owns 32 rows end to end: the normalised window (6 halo rows), the mixer output
// (FFN input and residual), its bf16 image and the bf16 hidden tile all live in LDS; both GEMMs run on the matrix cores
// (mfma_f32_32x32x16_bf16, weights straight from L2 as the A operand, activations from LDS as the B operand).
// Streaming state: hist = the last 6 normalised input rows of the previous call; row tile 0 is its only reader (before its
// first barrier) and its only writer (at its end), as in block_mixer_kernel.
#include <hip/hip_runtime.h>
#include <hip/hip_bf16.h>
#include <stdint.h>

#include "vv_hip.h"
#include "vv_common.h"

namespace {

typedef unsigned short bf16_t;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// phase timing of workgroup 0 / thread 0, debug builds only (-DVV_CF_TIMING, tools/convffn_phase.py)
#ifdef VV_CF_TIMING
__device__ unsigned long long g_b1_t[8];
#define BSTAMP(i) do { if (blockIdx.x == 0 && threadIdx.x == 0) { const long long t_ = wall_clock64(); g_b1_t[i] += (unsigned long long)(t_ - tprev_); tprev_ = t_; } } while (0)
#else
#define BSTAMP(i) do { } while (0)
#endif

constexpr int TR = 32;          // rows per workgroup (one MFMA tile)
constexpr int HALO = 6;         // causal depthwise kernel 7

__device__ __forceinline__ float gelu1(float v) { return vv_gelu_as(v); }   // result goes to bf16: see vv_common.h
__device__ __forceinline__ unsigned int pack2(float a, float b) {
  const __hip_bfloat16 x = __float2bfloat16(a), y = __float2bfloat16(b);
  return (unsigned int)(*reinterpret_cast<const bf16_t*>(&x)) | ((unsigned int)(*reinterpret_cast<const bf16_t*>(&y)) << 16);
}

// sum over an aligned group of G = 8 / 16 / 32 consecutive lanes (all lanes active)
template <int G>
__device__ __forceinline__ float group_sum(float v) {
#define VV_DPP_ADD(ctrl) v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), ctrl, 0xF, 0xF, true))
  VV_DPP_ADD(0xB1);                 // lane ^ 1
  VV_DPP_ADD(0x4E);                 // lane ^ 2
  VV_DPP_ADD(0x141);                // other quad of the 8-lane half row
  if (G >= 16) VV_DPP_ADD(0x140);   // other half of the 16-lane row
#undef VV_DPP_ADD
  if (G >= 32) v += __shfl_xor(v, 16);
  return v;
}

template <int C> struct Lay {
  static constexpr int F4 = C / 4;                       // float4 per row = threads per row
  static constexpr int RP = 256 / F4;                    // rows per pass of the 256 threads
  static constexpr int NI = (TR + HALO + RP - 1) / RP;   // passes over the 38-row window
  static constexpr int P1 = C + 8;                       // bf16 pitch of the FFN input image
  static constexpr int P2 = 4 * C + 8;                   // bf16 pitch of the hidden tile
  static constexpr int NB2 = C / 32;                     // 32-channel blocks of the FFN output
  static constexpr int KS = 4 / NB2;                     // waves that split K of the second GEMM
  static constexpr size_t XN = (size_t)(TR + HALO) * C * 4;
  static constexpr size_t X1 = (size_t)TR * C * 4;
  static constexpr size_t XH = (size_t)TR * P1 * 2;
  static constexpr size_t HID = (size_t)TR * P2 * 2;
  static constexpr size_t RED = (KS > 1) ? (size_t)(KS - 1) * NB2 * 16 * 64 * 4 : 0;
  static constexpr size_t LDS = XN + X1 + XH + HID + RED;
};

template <int C>
__global__ __launch_bounds__(256) void block1d_kernel(const float* __restrict__ x, float* __restrict__ out, int T, const vv_block B, float eps) {
  using L = Lay<C>;
#ifdef VV_CF_TIMING
  long long tprev_ = wall_clock64();
#endif
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float* xn = reinterpret_cast<float*>(smem);                                    // [TR + 6][C] normalised window
  float* x1s = reinterpret_cast<float*>(smem + L::XN);                           // [TR][C] mixer output (fp32: residual of the FFN)
  bf16_t* xh = reinterpret_cast<bf16_t*>(smem + L::XN + L::X1);                  // [TR][P1] RMSNorm(x1) in bf16
  bf16_t* hid = reinterpret_cast<bf16_t*>(smem + L::XN + L::X1 + L::XH);         // [TR][P2] gelu(W1 . + b1) in bf16
  float* red = reinterpret_cast<float*>(smem + L::XN + L::X1 + L::XH + L::HID);  // K-split partial accumulators
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n_tiles = (T + TR - 1) / TR;
  int tile = blockIdx.x;                                           // persistent: row tiles tile, tile + grid, ... with the weights loaded ONCE
  int t0 = tile * TR;
  int rows = min(TR, T - t0);
  const int cq = tid % L::F4, rloc = tid / L::F4;
  const int c0 = cq * 4;
  auto load_window = [&](float4 (&o)[L::NI], int tt0, int nrows) {
#pragma unroll
    for (int i = 0; i < L::NI; ++i) {
      const int w = rloc + L::RP * i, t = tt0 - HALO + w;
      o[i] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (w < nrows + HALO) {
        if (t >= 0) o[i] = *reinterpret_cast<const float4*>(x + (int64_t)t * C + c0);
        else if (B.hist) o[i] = *reinterpret_cast<const float4*>(B.hist + (int64_t)(HALO + t) * C + c0);   // already normalised
      }
    }
  };

  // ---- 1. window rows t0-6 .. t0+rows-1: raw values stay in registers, normalised values go to LDS -----------------------
  // Every load of the kernel is requested here, in the order the phases need them: loads return in order, so the window comes first
  // (with the weights first it sat behind 256 KB of them: 7.4 us before the first statistic at C = 128), then the per-channel
  // parameters of the mixer, then this wave's W1 fragments (C/32 blocks x C/16 steps x 16 B per lane, <= 128 VGPRs), W2 fragments
  // (<= 128 VGPRs) and the second GEMM's epilogue operands.  The later phases find their operands in registers instead of starting
  // another trip to L2 each.
  // Long sequences (a voice prompt: 40 000 rows at C = 128) used to launch one workgroup per row tile, each pulling the stage's whole
  // weight set (262 KB at C = 128) through its CU again: 330 MB of L2 traffic for 10 GFLOP.  Now at most 2 workgroups per CU stay
  // resident and walk the tiles with the weights in registers; the next tile's window is requested as soon as the mixer is done with
  // the current one.
  float4 own[L::NI];
  float ss[L::NI];
  load_window(own, t0, rows);
  const float4 nw = *reinterpret_cast<const float4*>(B.norm_w + c0);
  float tap[4][7];
  {
    float tq[28];                                                  // 28 consecutive floats: taps of channels c0 .. c0 + 3
#pragma unroll
    for (int q = 0; q < 7; ++q) {
      const float4 v = *reinterpret_cast<const float4*>(B.dw_w + (size_t)c0 * 7 + 4 * q);
      tq[4 * q] = v.x; tq[4 * q + 1] = v.y; tq[4 * q + 2] = v.z; tq[4 * q + 3] = v.w;
    }
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
      for (int k = 0; k < 7; ++k) tap[c][k] = tq[7 * c + k];
  }
  const float4 db = *reinterpret_cast<const float4*>(B.dw_b + c0);
  const float4 gm = *reinterpret_cast<const float4*>(B.gamma + c0);
  const float4 fw = *reinterpret_cast<const float4*>(B.ffn_norm_w + c0);
  const int hk = (lane >> 5) * 8;                                  // k offset of this lane inside a 16-wide MFMA step
  const int lm = lane & 31;
  constexpr int NBW = C / 32;                                      // W1 output blocks per wave
  constexpr int ST1 = C / 16;                                      // MFMA steps of the first GEMM
  u32x4 w1f[NBW][ST1];
  {
    const bf16_t* W1 = reinterpret_cast<const bf16_t*>(B.w1);
#pragma unroll
    for (int j = 0; j < NBW; ++j) {
      const bf16_t* wr = W1 + (int64_t)((wave * NBW + j) * 32 + lm) * C + hk;
#pragma unroll
      for (int s = 0; s < ST1; ++s) w1f[j][s] = *reinterpret_cast<const u32x4*>(wr + s * 16);
    }
  }
  // second GEMM: C/32 output blocks, K = 4C split over 4 / (C/32) waves
  constexpr int ST2 = (4 * C / 16) / L::KS;                        // MFMA steps of this wave in the second GEMM
  const int nblk = wave % L::NB2, kpart = wave / L::NB2;
  u32x4 w2f[ST2];
  {
    const bf16_t* w2r = reinterpret_cast<const bf16_t*>(B.w2) + (int64_t)(nblk * 32 + lm) * (4 * C) + kpart * ST2 * 16 + hk;
#pragma unroll
    for (int i = 0; i < ST2; ++i) w2f[i] = *reinterpret_cast<const u32x4*>(w2r + i * 16);
  }
  float4 b2v[4], fgv[4];                                           // epilogue of the second GEMM (channel runs of this lane)
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    const int n = nblk * 32 + 8 * g + 4 * (lane >> 5);
    b2v[g] = *reinterpret_cast<const float4*>(B.b2 + n);
    fgv[g] = *reinterpret_cast<const float4*>(B.ffn_gamma + n);
  }
  __builtin_amdgcn_sched_barrier(0);                               // nothing below is scheduled in front of these requests
  for (;;) {
#pragma unroll
  for (int i = 0; i < L::NI; ++i) ss[i] = own[i].x * own[i].x + own[i].y * own[i].y + own[i].z * own[i].z + own[i].w * own[i].w;
  BSTAMP(0);                                       // load issue + arrival of the window
#pragma unroll
  for (int i = 0; i < L::NI; ++i) ss[i] = group_sum<L::F4>(ss[i]);
#pragma unroll
  for (int i = 0; i < L::NI; ++i) {
    const int w = rloc + L::RP * i, t = t0 - HALO + w;
    if (w < TR + HALO) {
      float4 v = own[i];
      if (t >= 0 && w < rows + HALO) {
        const float rstd = rsqrtf(ss[i] / (float)C + eps);
        v.x *= rstd * nw.x; v.y *= rstd * nw.y; v.z *= rstd * nw.z; v.w *= rstd * nw.w;
      }
      *reinterpret_cast<float4*>(xn + w * C + c0) = v;
    }
  }
  __syncthreads();
  BSTAMP(1);                                       // statistics + normalised window

  // ---- 2. mixer + FFN RMSNorm: x1 = x + gamma (dwconv7(xn) + b); xh = bf16(x1 rstd ffn_norm_w) ----------------------------
  // first GEMM's bias runs: requested now, they arrive while the mixer computes
  float4 b1v[NBW][4];
#pragma unroll
  for (int j = 0; j < NBW; ++j)
#pragma unroll
    for (int g = 0; g < 4; ++g) b1v[j][g] = *reinterpret_cast<const float4*>(B.b1 + (wave * NBW + j) * 32 + 8 * g + 4 * (lane >> 5));
  {
    float4 x1[L::NI];
    float s2[L::NI];
#pragma unroll
    for (int i = 0; i < L::NI; ++i) {
      const int w = rloc + L::RP * i, tt = w - HALO;
      x1[i] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (tt >= 0 && tt < rows) {
        float4 s = db;
#pragma unroll
        for (int k = 0; k < 7; ++k) {
          const float4 v = *reinterpret_cast<const float4*>(xn + (tt + k) * C + c0);
          s.x = fmaf(tap[0][k], v.x, s.x); s.y = fmaf(tap[1][k], v.y, s.y); s.z = fmaf(tap[2][k], v.z, s.z); s.w = fmaf(tap[3][k], v.w, s.w);
        }
        x1[i] = make_float4(own[i].x + gm.x * s.x, own[i].y + gm.y * s.y, own[i].z + gm.z * s.z, own[i].w + gm.w * s.w);
      }
      s2[i] = x1[i].x * x1[i].x + x1[i].y * x1[i].y + x1[i].z * x1[i].z + x1[i].w * x1[i].w;
    }
#pragma unroll
    for (int i = 0; i < L::NI; ++i) s2[i] = group_sum<L::F4>(s2[i]);
#pragma unroll
    for (int i = 0; i < L::NI; ++i) {
      const int w = rloc + L::RP * i, tt = w - HALO;
      if (tt >= 0 && tt < TR) {                                   // rows past the end of the sequence: zeros
        const float rstd = rsqrtf(s2[i] / (float)C + eps);
        *reinterpret_cast<float4*>(x1s + tt * C + c0) = x1[i];
        uint2 p;
        p.x = pack2(x1[i].x * rstd * fw.x, x1[i].y * rstd * fw.y);
        p.y = pack2(x1[i].z * rstd * fw.z, x1[i].w * rstd * fw.w);
        *reinterpret_cast<uint2*>(xh + tt * L::P1 + c0) = p;
      }
    }
  }
  __syncthreads();
  BSTAMP(2);                                       // mixer + second norm
  const int tile_n = tile + gridDim.x;             // the raw window values are dead: the next tile's window streams in behind the two GEMMs
  if (tile_n < n_tiles) load_window(own, tile_n * TR, min(TR, T - tile_n * TR));

  // ---- 3. hidden = gelu(W1 xh + b1): 4C/32 output blocks, C/32 per wave; K = C ----------------------------------------------
  {
    const bf16_t* xf = xh + lm * L::P1 + hk;
#pragma unroll
    for (int j = 0; j < NBW; ++j) {
      f32x16 acc;
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[i] = 0.f;
#pragma unroll
      for (int s = 0; s < ST1; ++s) {
        const u32x4 xb = *reinterpret_cast<const u32x4*>(xf + s * 16);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, w1f[j][s]), __builtin_bit_cast(bf16x8, xb), acc, 0, 0, 0);
      }
      const int n0 = (wave * NBW + j) * 32;
#pragma unroll
      for (int g = 0; g < 4; ++g) {                              // acc[4g + i]: channel n0 + 8g + 4(lane >> 5) + i, row lane & 31
        const int n = n0 + 8 * g + 4 * (lane >> 5);
        const float4 b1 = b1v[j][g];
        uint2 p;
        p.x = pack2(gelu1(acc[4 * g] + b1.x), gelu1(acc[4 * g + 1] + b1.y));
        p.y = pack2(gelu1(acc[4 * g + 2] + b1.z), gelu1(acc[4 * g + 3] + b1.w));
        *reinterpret_cast<uint2*>(hid + lm * L::P2 + n) = p;
      }
    }
  }
  __syncthreads();
  BSTAMP(3);                                       // first GEMM + GELU

  // ---- 4. y = W2 hidden + b2; out = x1 + ffn_gamma y -----------------------------------------------------------------------------
  {
    const bf16_t* hf = hid + lm * L::P2 + kpart * ST2 * 16 + hk;
    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
#pragma unroll
    for (int i = 0; i < ST2; ++i) {
      const u32x4 hb = *reinterpret_cast<const u32x4*>(hf + i * 16);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, w2f[i]), __builtin_bit_cast(bf16x8, hb), acc, 0, 0, 0);
    }
    if (L::KS > 1) {                                               // fixed-order combine: deterministic
      if (kpart > 0) {
#pragma unroll
        for (int r = 0; r < 16; ++r) red[(((kpart - 1) * L::NB2 + nblk) * 16 + r) * 64 + lane] = acc[r];
      }
      __syncthreads();
      if (kpart == 0) {
#pragma unroll
        for (int kp = 1; kp < L::KS; ++kp)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[r] += red[(((kp - 1) * L::NB2 + nblk) * 16 + r) * 64 + lane];
      }
    }
    if (kpart == 0 && lm < rows) {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int n = nblk * 32 + 8 * g + 4 * (lane >> 5);
        const float4 b2 = b2v[g], fg = fgv[g];
        const float4 r1 = *reinterpret_cast<const float4*>(x1s + lm * C + n);
        float4 o;
        o.x = r1.x + fg.x * (acc[4 * g] + b2.x);
        o.y = r1.y + fg.y * (acc[4 * g + 1] + b2.y);
        o.z = r1.z + fg.z * (acc[4 * g + 2] + b2.z);
        o.w = r1.w + fg.w * (acc[4 * g + 3] + b2.w);
        *reinterpret_cast<float4*>(out + (int64_t)(t0 + lm) * C + n) = o;
      }
    }
  }

  BSTAMP(4);                                       // second GEMM + epilogue
  // ---- 5. streaming state: the last 6 normalised input rows (T >= 6) --------------------------------------------------------
  if (B.hist && tile == 0) {
    for (int j = wave; j < HALO; j += 4) {
      const int src = T - HALO + j;
      float* dst = B.hist + (int64_t)j * C;
      if (src < rows) {                                            // inside this tile's window
        for (int c = lane; c < C; c += 64) dst[c] = xn[(src + HALO) * C + c];
      } else {
        const float* xr = x + (int64_t)src * C;
        float v0 = lane < C ? xr[lane] : 0.f, v1 = (C > 64) ? xr[64 + lane] : 0.f;
        const float rstd = rsqrtf(vv_wave_sum(v0 * v0 + v1 * v1) / (float)C + eps);
        if (lane < C) dst[lane] = v0 * rstd * B.norm_w[lane];
        if (C > 64) dst[64 + lane] = v1 * rstd * B.norm_w[64 + lane];
      }
    }
  }
  if (tile_n >= n_tiles) break;
  tile = tile_n;
  t0 = tile * TR;
  rows = min(TR, T - t0);
  __syncthreads();                                 // every LDS buffer of this tile has been consumed
  }
}


int g_blocks = 256;           // persistent workgroups at C = 128 (1 per CU: 5.15 ms voice encode vs 5.18 at 512, 5.59 unbounded); tuning hook "block1d_blocks"

template <int C>
int launch_c(const float* x, float* out, int T, const vv_block& B, float eps, hipStream_t s) {
  const int n_tiles = (T + TR - 1) / TR;
  // C = 128: the weight set is 262 KB per workgroup - resident workgroups walk the tiles.  C = 64 / 32 (65 / 16 KB): one workgroup per tile
  // is faster (144 vs 176 us on a 40 000-row sequence: more independent workgroups per CU beat the saved re-fetch)
  const int cap = (C == 128) ? g_blocks : n_tiles;
  hipLaunchKernelGGL((block1d_kernel<C>), dim3(n_tiles < cap ? n_tiles : cap), dim3(256), Lay<C>::LDS, s, x, out, T, B, eps);
  return hipGetLastError() == hipSuccess ? 1 : vv_set_error(VV_E_HIP, "vv_block1d: launch failed");
}

int g_fused = 1;

}  // namespace

#ifdef VV_CF_TIMING
extern "C" int vv_block1d_debug_times(unsigned long long* out8, int reset) {
  if (hipMemcpyFromSymbol(out8, HIP_SYMBOL(g_b1_t), 8 * sizeof(unsigned long long)) != hipSuccess) return -1;
  if (reset) { unsigned long long z[8] = {0}; if (hipMemcpyToSymbol(HIP_SYMBOL(g_b1_t), z, sizeof(z)) != hipSuccess) return -1; }
  return 0;
}
#endif
void vv_block1d_set_fused(int on) { g_fused = on; }
void vv_block1d_set_blocks(int b) { if (b > 0) g_blocks = b; }

int vv_block1d_init() {
#define VV_ATTR(CC)                                                                                                  \
  if (hipFuncSetAttribute(reinterpret_cast<const void*>(&block1d_kernel<CC>), hipFuncAttributeMaxDynamicSharedMemorySize, \
                          (int)Lay<CC>::LDS) != hipSuccess)                                                          \
    return vv_set_error(VV_E_HIP, "vv_block1d_init: cannot raise the LDS limit");
  VV_ATTR(32) VV_ATTR(64) VV_ATTR(128)
#undef VV_ATTR
  return 0;
}

// 1 = the whole block was enqueued as one launch, 0 = not covered (caller runs mixer + two linears), < 0 = error
int vv_launch_block1d(const vv_block& B, int wdt, const float* x, float* out, int T, int C, float eps, hipStream_t s) {
  if (!g_fused || wdt != VV_BF16 || T < 32) return 0;
  {   // row tiles read their input (and halo) while others already write: the two [T, C] ranges must be disjoint
    const uintptr_t xa = (uintptr_t)x, oa = (uintptr_t)out, bytes = (uintptr_t)T * C * 4;
    if (xa < oa + bytes && oa < xa + bytes) return 0;
  }
  if (C != 32 && C != 64 && C != 128) return 0;
  auto a16 = [](const void* q) { return q && ((uintptr_t)q % 16) == 0; };
  if (!a16(B.w1) || !a16(B.w2) || !a16(B.b1) || !a16(B.b2) || !a16(B.gamma) || !a16(B.ffn_gamma) || !a16(B.norm_w) || !a16(B.ffn_norm_w) ||
      !a16(B.dw_b) || !a16(B.dw_w) || !a16(x) || !a16(out) || (B.hist && !a16(B.hist)))
    return 0;
  if (C == 32) return launch_c<32>(x, out, T, B, eps, s);
  if (C == 64) return launch_c<64>(x, out, T, B, eps, s);
  return launch_c<128>(x, out, T, B, eps, s);
}

extern "C" int vv_block1d(const vv_block* b, int wdt, const float* x, float* out, int T, int C, float eps, vv_stream_t stream) {
  if (!b || !x || !out || T <= 0) return vv_set_error(VV_E_ARG, "vv_block1d: bad args");
  const int rc = vv_launch_block1d(*b, wdt, x, out, T, C, eps, (hipStream_t)stream);
  if (rc < 0) return rc;
  if (rc == 0) return vv_set_error(VV_E_UNSUPPORTED, "vv_block1d: shape not covered by the fused kernel (C=%d T=%d wdt=%d)", C, T, wdt);
  return 0;
}
