// vv_mfma_gemm.hip — M > 8 rows against bf16 weights on the gfx950 matrix cores (mfma_f32_32x32x16_bf16).
//
// Used where the path really is GEMM shaped: the conv-tokenizer stages with T >= 8 time steps per frame (FFN linears,
// dense / transposed convs read in place as overlapping-row GEMMs), the hoisted adaLN modulation of all diffusion steps,
// the voice-prompt encoder and the LLM prompt prefill.  These problems are small and latency bound, not flop bound:
// the point of MFMA here is that one wave finishes a 32(out-channel) x 32(row) tile per 16 k-elements with a single
// 16-byte weight load per lane, so a tile's whole K loop is a short stream of independent loads instead of
// hundreds of LDS-staged barrier rounds.
//   D[n, m] = sum_k W[n, k] * Xhat[m, k]      A operand = weight fragment, straight from global (rows are K-contiguous)
//                                             B operand = activation fragment from an LDS image of the 32-row tile
//   Xhat = bf16( prologue(x) )  staged per workgroup (fp32 -> RMSNorm / modulate / SiLU -> bf16), row pitch padded by
//   16 B so the 32 rows of a ds_read_b128 land on distinct bank groups.
// Two wave layouts: 4 waves = 4 adjacent 32-channel blocks (wide N), or 4 waves = 4 K slices of one block combined
// through LDS in a fixed order (narrow N, long K): deterministic, no atomics.
// Activations are rounded to bf16 once at staging (the reference's bf16 run keeps them in bf16 throughout); the
// fp32-weight parity mode never comes here.
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

int g_mt_override = 0;   // tuning hook (vv_tune "mfma_mt")
int g_mt_prefill_xb = 4;   // bf16 activations streamed from global (no LDS chunks / barriers): 128-row strips
int g_mt_prefill = 2;      // measured on MI355X: 330-token prefill 23.3 ms (MT 1) / 20.7 ms (MT 2) / 30.3 ms (MT 4)

template <int MT> struct Tile {                 // MT 32-row tiles per workgroup share every weight fragment
  static constexpr int ROWS = 32 * MT;
  static constexpr int KCH = 1024 / MT;         // K elements of the activation tile resident in LDS at a time
  static constexpr int PITCH = KCH + 8;         // bf16 elements per LDS row (16-byte pad -> conflict-free ds_read_b128)
  static constexpr size_t XS_BYTES = (size_t)ROWS * PITCH * 2;
  static constexpr size_t RED_BYTES = 4 * 2 * 16 * 64 * 4;
  static constexpr size_t LDS = (XS_BYTES > RED_BYTES ? XS_BYTES : RED_BYTES) + ROWS * 4 + 64;
};

__device__ __forceinline__ float silu1(float v) { return v / (1.0f + expf(-v)); }
__device__ __forceinline__ float gelu1(float v) { return 0.5f * v * (1.0f + erff(v * 0.70710678118654752440f)); }
__device__ __forceinline__ unsigned int pack2(float a, float b) {
  const __hip_bfloat16 x = __float2bfloat16(a), y = __float2bfloat16(b);
  return (unsigned int)(*reinterpret_cast<const bf16_t*>(&x)) | ((unsigned int)(*reinterpret_cast<const bf16_t*>(&y)) << 16);
}
__device__ __forceinline__ void epi1(const vv_lin_args& a, int m, int n, float v, float v2) {
  if (a.bias) v += a.bias[n];
  if (a.act == VV_ACT_GELU) v = gelu1(v);
  else if (a.act == VV_ACT_SWIGLU) v = silu1(v) * v2;
  if (a.gate) v *= a.gate_ld ? a.gate[(int64_t)m * a.gate_ld + n] : a.gate[n];
  if (a.res) v += a.res[(int64_t)m * a.ldres + n];
  if (a.flags & VV_LIN_OUT_BF16) {
    const __hip_bfloat16 b = __float2bfloat16(v);
    reinterpret_cast<bf16_t*>(a.out)[(int64_t)m * a.ldo + n] = *reinterpret_cast<const bf16_t*>(&b);
  } else {
    a.out[(int64_t)m * a.ldo + n] = v;
  }
}

// four consecutive output channels of one row: 16-byte loads of bias / gate / residual and one 16-byte (fp32) or 8-byte (bf16)
// store when the row pitches allow it (the accumulator layout gives every lane 4 runs of 4 consecutive channels)
__device__ __forceinline__ void epi4(const vv_lin_args& a, bool vec_ok, int m, int n, const float (&vin)[4], const float (&v2)[4]) {
  if (!vec_ok || n + 3 >= a.n) {
#pragma unroll
    for (int i = 0; i < 4; ++i) if (n + i < a.n) epi1(a, m, n + i, vin[i], v2[i]);
    return;
  }
  float v[4] = {vin[0], vin[1], vin[2], vin[3]};
  if (a.bias) { const float4 b = *reinterpret_cast<const float4*>(a.bias + n); v[0] += b.x; v[1] += b.y; v[2] += b.z; v[3] += b.w; }
  if (a.act == VV_ACT_GELU) {
    if (a.flags & VV_LIN_OUT_BF16) {             // rounded to bf16 below: the 1.5e-7 erf approximation is exact at that precision
#pragma unroll
      for (int i = 0; i < 4; ++i) v[i] = vv_gelu_as(v[i]);
    } else {
#pragma unroll
      for (int i = 0; i < 4; ++i) v[i] = gelu1(v[i]);
    }
  } else if (a.act == VV_ACT_SWIGLU) {
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = silu1(v[i]) * v2[i];
  }
  if (a.gate) {
    const float4 g = *reinterpret_cast<const float4*>(a.gate + (a.gate_ld ? (int64_t)m * a.gate_ld : 0) + n);
    v[0] *= g.x; v[1] *= g.y; v[2] *= g.z; v[3] *= g.w;
  }
  if (a.res) {
    const float4 r = *reinterpret_cast<const float4*>(a.res + (int64_t)m * a.ldres + n);
    v[0] += r.x; v[1] += r.y; v[2] += r.z; v[3] += r.w;
  }
  if (a.flags & VV_LIN_OUT_BF16) {
    uint2 p;
    p.x = pack2(v[0], v[1]);
    p.y = pack2(v[2], v[3]);
    *reinterpret_cast<uint2*>(reinterpret_cast<bf16_t*>(a.out) + (int64_t)m * a.ldo + n) = p;
  } else {
    *reinterpret_cast<float4*>(a.out + (int64_t)m * a.ldo + n) = make_float4(v[0], v[1], v[2], v[3]);
  }
}

// phase timing of workgroup (0,0) / thread 0, debug builds only (-DVV_MFMA_TIMING, tools/mfma_phase_test.cpp)
#ifdef VV_MFMA_TIMING
__device__ unsigned long long g_mfma_t[8];
#define MSTAMP(i) do { if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) { const long long t_ = wall_clock64(); g_mfma_t[i] += (unsigned long long)(t_ - tprev_); tprev_ = t_; } } while (0)
#else
#define MSTAMP(i) do { } while (0)
#endif

template <bool DUAL, bool KSPLIT, bool XB, int MT>
__global__ __launch_bounds__(256) void mfma_linear_kernel(const vv_lin_args a) {
  using T = Tile<MT>;
#ifdef VV_MFMA_TIMING
  long long tprev_ = wall_clock64();
#endif
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  bf16_t* xs = reinterpret_cast<bf16_t*>(smem);                                             // [ROWS][PITCH]
  float* red = reinterpret_cast<float*>(smem);                                               // K-split combine scratch (aliases xs)
  float* rs = reinterpret_cast<float*>(smem + (T::XS_BYTES > T::RED_BYTES ? T::XS_BYTES : T::RED_BYTES));   // [ROWS] row rstd
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int M = a.m, N = a.n, K = a.k;
  const int m0 = blockIdx.y * T::ROWS;
  const int nblk = KSPLIT ? blockIdx.x : blockIdx.x * 4 + wave;
  const int n0 = nblk * 32;
  const bf16_t* __restrict__ W = reinterpret_cast<const bf16_t*>(a.w);
  const bf16_t* __restrict__ W2 = reinterpret_cast<const bf16_t*>(a.w2);

  // ---- per-row RMS statistic of the tile (whole K) -------------------------------------------------------------
  if (!XB && a.pro == VV_PRO_RMSNORM) {
    const int q = tid & 7;
    for (int r = tid >> 3; r < T::ROWS; r += 32) {
      float s = 0.f;
      if (m0 + r < M) {
        const float* xr = a.x + (int64_t)(m0 + r) * a.ldx;
#pragma unroll 8
        for (int k = q * 4; k < K; k += 32) {
          const float4 v = *reinterpret_cast<const float4*>(xr + k);
          s += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
        }
      }
      s += __shfl_xor(s, 1); s += __shfl_xor(s, 2); s += __shfl_xor(s, 4);
      if (q == 0) rs[r] = rsqrtf(s / (float)K + a.eps);
    }
  }

  MSTAMP(0);                                         // row statistics
  f32x16 acc[MT], acc2[DUAL ? MT : 1];
#pragma unroll
  for (int t = 0; t < MT; ++t)
#pragma unroll
    for (int i = 0; i < 16; ++i) { acc[t][i] = 0.f; if (DUAL) acc2[t][i] = 0.f; }
  const int wr = min(n0 + (lane & 31), N - 1);                      // weight row of this lane (clamped; masked at the store)
  const int hk = (lane >> 5) * 8;                                    // k offset of this lane inside a 16-wide step
  const bf16_t* wrow = W + (int64_t)wr * K + hk;
  const bf16_t* wrow2 = DUAL ? (W2 + (int64_t)wr * K + hk) : nullptr;
  const bf16_t* xfrag = xs + (lane & 31) * T::PITCH + hk;
  const bool active = n0 < N;

  if (XB) {
    // bf16 activations handed over by the producing GEMM: B fragments stream straight from global like the weights,
    // no LDS image, no barriers
    if (active) {
      const bf16_t* xrow[MT];
#pragma unroll
      for (int t = 0; t < MT; ++t) xrow[t] = reinterpret_cast<const bf16_t*>(a.x) + (int64_t)min(m0 + 32 * t + (lane & 31), M - 1) * a.ldx + hk;
      const int nsteps = K >> 4;
      int s_begin = 0, s_end = nsteps;
      if (KSPLIT) { const int per = (nsteps + 3) >> 2; s_begin = wave * per; s_end = min(nsteps, s_begin + per); }
      constexpr int UB = (MT == 1) ? 8 : 4;
      for (int sb = s_begin; sb < s_end; sb += UB) {          // UB k-steps of operands requested before the first MFMA
        u32x4 wa[UB], wb[DUAL ? UB : 1], xb[UB][MT];
#pragma unroll
        for (int i = 0; i < UB; ++i) {
          const int s = min(sb + i, s_end - 1);
          wa[i] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(wrow + s * 16));
          if (DUAL) wb[i] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(wrow2 + s * 16));
#pragma unroll
          for (int t = 0; t < MT; ++t) xb[i][t] = *reinterpret_cast<const u32x4*>(xrow[t] + s * 16);
        }
#pragma unroll
        for (int i = 0; i < UB; ++i) {
          if (sb + i >= s_end) break;
#pragma unroll
          for (int t = 0; t < MT; ++t) {
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, wa[i]), __builtin_bit_cast(bf16x8, xb[i][t]), acc[t], 0, 0, 0);
            if (DUAL) acc2[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, wb[i]), __builtin_bit_cast(bf16x8, xb[i][t]), acc2[t], 0, 0, 0);
          }
        }
      }
    }
  } else
  for (int kc0 = 0; kc0 < K; kc0 += T::KCH) {
    const int kc = min(T::KCH, K - kc0);
    __syncthreads();                                                 // rs visible / previous chunk fully consumed
    // ---- stage Xhat[ROWS, kc] ----------------------------------------------------------------------------------
    {
      const int q = tid & 7;
      for (int r = tid >> 3; r < T::ROWS; r += 32) {
        const bool rv = m0 + r < M;
        const float* xr = a.x + (int64_t)(rv ? m0 + r : 0) * a.ldx + kc0;
        const float rstd = (a.pro == VV_PRO_RMSNORM) ? rs[r] : 1.f;
        // loads are issued in batches of 8 ahead of the convert/store so the loop is not one L2 round trip per iteration
        for (int kb = q * 4; kb < kc; kb += 256) {
          float4 vv[8];
#pragma unroll
          for (int i = 0; i < 8; ++i) {
            const int k = kb + i * 32;
            vv[i] = (rv && k < kc) ? *reinterpret_cast<const float4*>(xr + k) : make_float4(0.f, 0.f, 0.f, 0.f);
          }
#pragma unroll
          for (int i = 0; i < 8; ++i) {
            const int k = kb + i * 32;
            if (k >= kc) break;
            float4 v = vv[i];
            if (a.pro == VV_PRO_RMSNORM) {
              v.x *= rstd; v.y *= rstd; v.z *= rstd; v.w *= rstd;
              if (a.norm_w) { const float4 w4 = *reinterpret_cast<const float4*>(a.norm_w + kc0 + k); v.x *= w4.x; v.y *= w4.y; v.z *= w4.z; v.w *= w4.w; }
              if (a.mod_scale && rv) {
                const int64_t mo = (int64_t)(m0 + r) * a.ld_mod + kc0 + k;
                v.x = v.x * (1.f + a.mod_scale[mo]) + a.mod_shift[mo];
                v.y = v.y * (1.f + a.mod_scale[mo + 1]) + a.mod_shift[mo + 1];
                v.z = v.z * (1.f + a.mod_scale[mo + 2]) + a.mod_shift[mo + 2];
                v.w = v.w * (1.f + a.mod_scale[mo + 3]) + a.mod_shift[mo + 3];
              }
            } else if (a.pro == VV_PRO_SILU) {
              v.x = silu1(v.x); v.y = silu1(v.y); v.z = silu1(v.z); v.w = silu1(v.w);
            }
            uint2 p;
            p.x = pack2(v.x, v.y);
            p.y = pack2(v.z, v.w);
            *reinterpret_cast<uint2*>(xs + r * T::PITCH + k) = p;
          }
        }
      }
    }
    __syncthreads();
    MSTAMP(1);                                       // staging (incl. the barrier before it)
    if (active) {
      const int nsteps = kc >> 4;
      int s_begin = 0, s_end = nsteps;
      if (KSPLIT) { const int per = (nsteps + 3) >> 2; s_begin = wave * per; s_end = min(nsteps, s_begin + per); }
      for (int sb = s_begin; sb < s_end; sb += 8) {           // 8 k-steps of weight fragments requested before the first MFMA
        u32x4 wa[8], wb[DUAL ? 8 : 1];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const int s = min(sb + i, s_end - 1);
          wa[i] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(wrow + kc0 + s * 16));
          if (DUAL) wb[i] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(wrow2 + kc0 + s * 16));
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          if (sb + i >= s_end) break;
#pragma unroll
          for (int t = 0; t < MT; ++t) {
            const u32x4 xb = *reinterpret_cast<const u32x4*>(xfrag + t * 32 * T::PITCH + (sb + i) * 16);
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, wa[i]), __builtin_bit_cast(bf16x8, xb), acc[t], 0, 0, 0);
            if (DUAL) acc2[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, wb[i]), __builtin_bit_cast(bf16x8, xb), acc2[t], 0, 0, 0);
          }
        }
      }
    }
  }

  MSTAMP(2);                                         // weight loads + MFMA
  // ---- epilogue: D[n = (reg&3) + 8*(reg>>2) + 4*(lane>>5)][m = lane&31]: a lane holds 4 runs of 4 consecutive channels ----
  const bool vec_ok = (a.ldo % 4 == 0) && ((uintptr_t)a.out % 16 == 0) && (!a.res || (a.ldres % 4 == 0 && (uintptr_t)a.res % 16 == 0)) &&
                      (!a.bias || (uintptr_t)a.bias % 16 == 0) && (!a.gate || ((uintptr_t)a.gate % 16 == 0 && a.gate_ld % 4 == 0));
  if (!KSPLIT) {
    if (!active) return;
#pragma unroll
    for (int t = 0; t < MT; ++t) {
      const int m = m0 + 32 * t + (lane & 31);
      if (m >= M) continue;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const float v[4] = {acc[t][4 * g], acc[t][4 * g + 1], acc[t][4 * g + 2], acc[t][4 * g + 3]};
        float v2[4] = {0.f, 0.f, 0.f, 0.f};
        if (DUAL) { v2[0] = acc2[t][4 * g]; v2[1] = acc2[t][4 * g + 1]; v2[2] = acc2[t][4 * g + 2]; v2[3] = acc2[t][4 * g + 3]; }
        epi4(a, vec_ok, m, n0 + 8 * g + 4 * (lane >> 5), v, v2);
      }
    }
  } else {
#pragma unroll
    for (int t = 0; t < MT; ++t) {
      __syncthreads();                                               // xs / previous tile's scratch no longer needed
#pragma unroll
      for (int reg = 0; reg < 16; ++reg) {
        red[((wave * 2 + 0) * 16 + reg) * 64 + lane] = acc[t][reg];
        if (DUAL) red[((wave * 2 + 1) * 16 + reg) * 64 + lane] = acc2[t][reg];
      }
      __syncthreads();
      {
        const int ln = tid & 63, g = tid >> 6;                        // thread -> (accumulator lane, run of 4 registers)
        const int m = m0 + 32 * t + (ln & 31);
        if (m < M) {
          float v[4] = {0.f, 0.f, 0.f, 0.f}, v2[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int w4 = 0; w4 < 4; ++w4) {                           // fixed order: deterministic
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              v[i] += red[((w4 * 2 + 0) * 16 + 4 * g + i) * 64 + ln];
              if (DUAL) v2[i] += red[((w4 * 2 + 1) * 16 + 4 * g + i) * 64 + ln];
            }
          }
          epi4(a, vec_ok, m, n0 + 8 * g + 4 * (ln >> 5), v, v2);
        }
      }
    }
  }
  MSTAMP(3);                                         // combine + epilogue
}

// ---- tiled GEMM for LONG row counts (M >= 1024 rows of bf16 activations, N % 128 == 0, K % 32 == 0) ------------------------------
// Whole-utterance voice-prompt encoding runs the conv FFNs over tens of thousands of rows: there the streaming kernel above
// re-reads every activation strip once per 32-channel block.  Here a workgroup owns a 128 x 128 output tile: both operands go
// global -> registers -> LDS in 128 x 32 slabs (double buffered, one barrier per K step), the 4 waves sit 2 x 2 and each holds
// 2 x 2 MFMA 32x32 accumulators.  (Not used for the 330-row prompt prefill: with a few dozen tiles the one-slab-deep K loop is a
// chain of L2 round trips and measured no faster than the streaming kernel.)
constexpr int TG_BM = 128, TG_BN = 128;

// BK = 32 (dual / short K) or 128 (long K: a down-projection over K = 8960 is 70 barrier rounds instead of 280; each round is
// dominated by the barrier + LDS hand-off, not by its 8 MFMAs)
template <bool DUAL, int BK, int TM>
__global__ __launch_bounds__(256) void mfma_tiled_kernel(const vv_lin_args a) {
  constexpr int PITCH = BK + 8;                    // bf16 pitch: the 32 rows of a fragment read hit distinct bank groups
  constexpr int NP = TM * BK / (8 * 256);          // 16-byte pieces per thread per operand slab (TM rows x BK)
  constexpr int FI = TM / 64;                      // 32 x 32 accumulators per wave per direction (waves sit 2 x 2)
  constexpr int PPR = BK / 8;                      // pieces per row
  constexpr int TG_D = (BK == 32) ? (DUAL ? 4 : 8) : (TM == 64 ? 4 : 2);   // slabs in flight in registers
  extern __shared__ __attribute__((aligned(16))) unsigned char tsm[];
  bf16_t* xs = reinterpret_cast<bf16_t*>(tsm);                       // [2][128 * PITCH]
  bf16_t* wsm = xs + 2 * TM * PITCH;                              // [2][128 * PITCH]
  bf16_t* ws2 = wsm + 2 * TM * PITCH;                             // [2][128 * PITCH] (dual only)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int M = a.m, K = a.k;
  const int n0 = blockIdx.x * TM, m0 = blockIdx.y * TM;
  const bf16_t* __restrict__ X = reinterpret_cast<const bf16_t*>(a.x);
  const bf16_t* __restrict__ W = reinterpret_cast<const bf16_t*>(a.w);
  const bf16_t* __restrict__ W2 = reinterpret_cast<const bf16_t*>(a.w2);
  // piece p = tid + 256 i of a slab: row p / PPR, k offset (p % PPR) * 8
  int prow[NP], pk[NP];
  const bf16_t* xg[NP]; const bf16_t* wg[NP]; const bf16_t* vg[NP];
#pragma unroll
  for (int i = 0; i < NP; ++i) {
    const int p = tid + 256 * i;
    prow[i] = p / PPR; pk[i] = (p % PPR) * 8;
    xg[i] = X + (int64_t)min(m0 + prow[i], M - 1) * a.ldx + pk[i];
    wg[i] = W + (int64_t)(n0 + prow[i]) * K + pk[i];
    vg[i] = DUAL ? W2 + (int64_t)(n0 + prow[i]) * K + pk[i] : nullptr;
  }
  struct Slab { u32x4 x[NP], w[NP], v[DUAL ? NP : 1]; };
  Slab R[TG_D];
  auto gload = [&](Slab& r, int k0) {
#pragma unroll
    for (int i = 0; i < NP; ++i) {
      r.x[i] = *reinterpret_cast<const u32x4*>(xg[i] + k0);
      r.w[i] = *reinterpret_cast<const u32x4*>(wg[i] + k0);
      if (DUAL) r.v[i] = *reinterpret_cast<const u32x4*>(vg[i] + k0);
    }
  };
  auto lstore = [&](const Slab& r, int buf) {
#pragma unroll
    for (int i = 0; i < NP; ++i) {
      *reinterpret_cast<u32x4*>(xs + buf * TM * PITCH + prow[i] * PITCH + pk[i]) = r.x[i];
      *reinterpret_cast<u32x4*>(wsm + buf * TM * PITCH + prow[i] * PITCH + pk[i]) = r.w[i];
      if (DUAL) *reinterpret_cast<u32x4*>(ws2 + buf * TM * PITCH + prow[i] * PITCH + pk[i]) = r.v[i];
    }
  };
  const int wn = wave & 1, wm = wave >> 1;
  const int fr = lane & 31, fk = (lane >> 5) * 8;
  f32x16 acc[FI][FI], acc2[DUAL ? FI : 1][DUAL ? FI : 1];
#pragma unroll
  for (int i = 0; i < FI; ++i)
#pragma unroll
    for (int j = 0; j < FI; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) { acc[i][j][r] = 0.f; if (DUAL) acc2[i][j][r] = 0.f; }
  const int nk = K / BK;
#pragma unroll
  for (int d = 0; d < TG_D; ++d) if (d < nk) gload(R[d], d * BK);
  lstore(R[0], 0);
  __syncthreads();
  auto multiply = [&](int buf) {
#pragma unroll
    for (int sub = 0; sub < BK / 16; ++sub) {
      u32x4 fa[FI], fb[FI], fa2[FI];
#pragma unroll
      for (int i = 0; i < FI; ++i) {
        fa[i] = *reinterpret_cast<const u32x4*>(wsm + buf * TM * PITCH + (wn * (TM / 2) + i * 32 + fr) * PITCH + sub * 16 + fk);
        if (DUAL) fa2[i] = *reinterpret_cast<const u32x4*>(ws2 + buf * TM * PITCH + (wn * (TM / 2) + i * 32 + fr) * PITCH + sub * 16 + fk);
        fb[i] = *reinterpret_cast<const u32x4*>(xs + buf * TM * PITCH + (wm * (TM / 2) + i * 32 + fr) * PITCH + sub * 16 + fk);
      }
#pragma unroll
      for (int i = 0; i < FI; ++i)
#pragma unroll
        for (int j = 0; j < FI; ++j) {
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fa[i]), __builtin_bit_cast(bf16x8, fb[j]), acc[i][j], 0, 0, 0);
          if (DUAL) acc2[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fa2[i]), __builtin_bit_cast(bf16x8, fb[j]), acc2[i][j], 0, 0, 0);
        }
    }
  };
  for (int ks = 0; ks < nk; ks += TG_D) {                            // unrolled by TG_D: slab registers have fixed roles
#pragma unroll
    for (int j = 0; j < TG_D; ++j) {
      const int it = ks + j;
      if (it < nk) {
        multiply(it & 1);
        if (it + 1 < nk) lstore(R[(j + 1) % TG_D], (it & 1) ^ 1);    // slab it+1: requested TG_D-1 steps ago
        if (it + TG_D < nk) gload(R[j], (it + TG_D) * BK);           // R[j] held slab `it`, stored to LDS one step ago
        __syncthreads();
      }
    }
  }
  const bool vec_ok = (a.ldo % 4 == 0) && ((uintptr_t)a.out % 16 == 0) && (!a.res || (a.ldres % 4 == 0 && (uintptr_t)a.res % 16 == 0)) &&
                      (!a.bias || (uintptr_t)a.bias % 16 == 0) && (!a.gate || ((uintptr_t)a.gate % 16 == 0 && a.gate_ld % 4 == 0));
#pragma unroll
  for (int i = 0; i < FI; ++i)
#pragma unroll
    for (int j = 0; j < FI; ++j) {
      const int m = m0 + wm * (TM / 2) + j * 32 + (lane & 31);
      if (m >= M) continue;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const float v[4] = {acc[i][j][4 * g], acc[i][j][4 * g + 1], acc[i][j][4 * g + 2], acc[i][j][4 * g + 3]};
        float v2[4] = {0.f, 0.f, 0.f, 0.f};
        if (DUAL) { v2[0] = acc2[i][j][4 * g]; v2[1] = acc2[i][j][4 * g + 1]; v2[2] = acc2[i][j][4 * g + 2]; v2[3] = acc2[i][j][4 * g + 3]; }
        epi4(a, vec_ok, m, n0 + wn * (TM / 2) + i * 32 + 8 * g + 4 * (lane >> 5), v, v2);
      }
    }
}

template <bool DUAL, int BK, int TM = 128>
constexpr size_t tiled_lds() { return (size_t)(DUAL ? 3 : 2) * 2 * TM * (BK + 8) * 2; }

constexpr size_t tiled_lds_q() { return tiled_lds<false, 128, 64>(); }
int g_tiled_bk128 = 1;     // long-K slabs for the non-dual tiled kernel (tuning hook "mfma_tiled_bk128")
int g_tiled_small = 200;  // below this many 128 x 128 tiles a long-K GEMM uses 64 x 64 tiles (tuning hook "mfma_tiled_small"; 0 = never)
int g_tiled_dual_bk64 = 1;
int g_tiled_small_k = 1024;     // shortest K for the 64 x 64 variant (tuning hook "mfma_tiled_small_k")
int g_tiled_small_dual = 256;   // same for the dual (SwiGLU gate/up) GEMM: prefill gate/up is 210 tiles (tuning hook "mfma_tiled_small_dual")
int g_tiled_rows = 32;     // fewest rows for the tiled kernel (tuning hook "mfma_tiled_rows"; 0 = never); it also needs >= 24 tiles

template <bool DUAL, bool KSPLIT, bool XB, int MT>
int launch(const vv_lin_args& a, hipStream_t s) {
  static bool attr_done = false;
  if (!attr_done) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&mfma_linear_kernel<DUAL, KSPLIT, XB, MT>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)Tile<MT>::LDS);
    if (e != hipSuccess) return vv_set_error(VV_E_HIP, "mfma_linear: hipFuncSetAttribute: %s", hipGetErrorString(e));
    attr_done = true;
  }
  const int nblocks = (a.n + 31) / 32, rtiles = (a.m + Tile<MT>::ROWS - 1) / Tile<MT>::ROWS;
  dim3 grid(KSPLIT ? nblocks : (nblocks + 3) / 4, rtiles);
  if (grid.y > 65535u) return vv_set_error(VV_E_UNSUPPORTED, "vv_linear: m=%d rows exceed one launch (split the call)", a.m);
  hipLaunchKernelGGL((mfma_linear_kernel<DUAL, KSPLIT, XB, MT>), grid, dim3(256), Tile<MT>::LDS, s, a);
  return 0;
}

template <bool DUAL, bool KSPLIT, bool XB>
int launch_mt(const vv_lin_args& a, hipStream_t s, int mt) {
  if (mt == 4) return launch<DUAL, KSPLIT, XB, 4>(a, s);
  if (mt == 2) return launch<DUAL, KSPLIT, XB, 2>(a, s);
  return launch<DUAL, KSPLIT, XB, 1>(a, s);
}

}  // namespace

// 1 = launched, 0 = shape/alignment not covered (caller falls back to the fp32 VALU GEMM), < 0 = error
int vv_launch_mfma_gemm(const vv_lin_args& a, hipStream_t s) {
  const bool xb = (a.flags & VV_LIN_X_BF16) != 0;
  if (a.wdt != VV_BF16 || a.m <= 8 || a.k % 16 || a.ldx % (xb ? 8 : 4)) return 0;
  if ((uintptr_t)a.w % 16 || (a.w2 && (uintptr_t)a.w2 % 16) || (uintptr_t)a.x % 16) return 0;
  if (xb && a.pro != VV_PRO_NONE) return vv_set_error(VV_E_ARG, "vv_linear: a bf16 x takes no prologue");
  if (a.norm_w && (uintptr_t)a.norm_w % 16) return 0;
  if ((a.k * 2) % 16) return 0;
  // the tiled kernel needs enough 128 x 128 tiles to occupy the chip's memory system (prefill: 36-210, voice-prompt encode: hundreds,
  // hoisted adaLN: 36); a conv-stage GEMM with 4 tiles stays on the streaming kernel (T = 200, C = 256: 8 us there, 25 us tiled)
  const bool enough_tiles = (long)(a.n / TG_BN) * ((a.m + TG_BM - 1) / TG_BM) >= 24;
  // a narrow output over a very long K (the 2048 -> 64 head conv of a whole-utterance encode: K = 14336, 203 rows): a handful of
  // 64 x 64 tiles, each a fast 128-column-slab K loop, instead of 14 streaming workgroups walking K in 16-element steps (216 -> 50 us)
  if (g_tiled_rows > 0 && g_tiled_small > 0 && xb && !a.w2 && !enough_tiles && a.m >= 64 && a.n % 64 == 0 && a.k % 128 == 0 && a.k >= 8192 &&
      a.ldx % 8 == 0) {
    dim3 gq(a.n / 64, (a.m + 63) / 64);
    hipLaunchKernelGGL((mfma_tiled_kernel<false, 128, 64>), gq, dim3(256), tiled_lds_q(), s, a);
    return 1;
  }
  if (g_tiled_rows > 0 && xb && a.m >= g_tiled_rows && enough_tiles && a.n % TG_BN == 0 && a.k % 32 == 0 && a.ldx % 8 == 0 && a.m <= 65535 * TG_BM) {
    dim3 grid(a.n / TG_BN, (a.m + TG_BM - 1) / TG_BM);
    const size_t lds_d = tiled_lds<true, 32>(), lds_l = tiled_lds<false, 128>(), lds_s = tiled_lds<false, 32>();
    const size_t lds_q = tiled_lds<false, 128, 64>(), lds_qd = tiled_lds<true, 32, 64>(), lds_qd64 = tiled_lds<true, 64, 64>();
    // a long-K GEMM on a few dozen 128 x 128 tiles (prefill down-projection: 36 tiles x 70 K rounds; the T = 200 stage of a voice-
    // prompt encode: 32 tiles x 64 rounds) leaves most CUs idle behind a serial K loop: 64 x 64 tiles give 4x the workgroups,
    // each with a quarter of the MFMA / LDS work per round and a 4-slab-deep register prefetch
    if (a.w2 && (long)grid.x * grid.y < g_tiled_small_dual) {
      dim3 gq(a.n / 64, (a.m + 63) / 64);
      if (a.k % 64 == 0 && g_tiled_dual_bk64) hipLaunchKernelGGL((mfma_tiled_kernel<true, 64, 64>), gq, dim3(256), lds_qd64, s, a);
      else hipLaunchKernelGGL((mfma_tiled_kernel<true, 32, 64>), gq, dim3(256), lds_qd, s, a);
      return 1;
    }
    if (!a.w2 && g_tiled_small > 0 && a.k % 128 == 0 && a.k >= g_tiled_small_k && (long)grid.x * grid.y < g_tiled_small) {
      dim3 gq(a.n / 64, (a.m + 63) / 64);
      hipLaunchKernelGGL((mfma_tiled_kernel<false, 128, 64>), gq, dim3(256), lds_q, s, a);
      return 1;
    }
    if (a.w2) hipLaunchKernelGGL((mfma_tiled_kernel<true, 32, 128>), grid, dim3(256), lds_d, s, a);
    else if (a.k % 128 == 0 && g_tiled_bk128) hipLaunchKernelGGL((mfma_tiled_kernel<false, 128, 128>), grid, dim3(256), lds_l, s, a);
    else hipLaunchKernelGGL((mfma_tiled_kernel<false, 32, 128>), grid, dim3(256), lds_s, s, a);
    return 1;
  }
  // rows per workgroup.  MT > 1 (each weight fragment reused by MT 32-row tiles) was measured SLOWER on every shape of this
  // path on MI355X (19.8 vs 23.4 audio-s/s, first chunk 55 vs 51 ms): these GEMMs are latency bound, and fewer / fatter
  // workgroups with shorter LDS-resident K chunks cost more than the saved weight re-reads (which hit L2 / Infinity Cache).
  // Exception: the prompt prefill (hundreds of rows against the big LLM matrices), where re-reading 55 MB of weights per
  // 32 rows is the cost.
  int mt = 1;
  if (g_mt_override > 0) mt = g_mt_override;
  else if (a.m >= 128 && a.k >= 1024 && a.n >= 1024) mt = xb ? g_mt_prefill_xb : g_mt_prefill;
  const long nblocks = (a.n + 31) / 32, rtiles = (a.m + 32 * mt - 1) / (32 * mt);
  // a wave's K loop is a serial chain of 16-element steps: split K over the workgroup's 4 waves whenever K is long, or when
  // there are too few tiles to fill the chip anyway
  const bool ksplit = a.k >= 512 || ((nblocks * rtiles < 256) && a.k >= 128);
  int rc;
  if (xb) {
    if (a.w2) rc = ksplit ? launch_mt<true, true, true>(a, s, mt) : launch_mt<true, false, true>(a, s, mt);
    else rc = ksplit ? launch_mt<false, true, true>(a, s, mt) : launch_mt<false, false, true>(a, s, mt);
  } else {
    if (a.w2) rc = ksplit ? launch_mt<true, true, false>(a, s, mt) : launch_mt<true, false, false>(a, s, mt);
    else rc = ksplit ? launch_mt<false, true, false>(a, s, mt) : launch_mt<false, false, false>(a, s, mt);
  }
  return rc ? rc : 1;
}

#ifdef VV_MFMA_TIMING
extern "C" int vv_mfma_debug_times(unsigned long long* out8, int reset) {
  if (hipMemcpyFromSymbol(out8, HIP_SYMBOL(g_mfma_t), 8 * sizeof(unsigned long long)) != hipSuccess) return -1;
  if (reset) { unsigned long long z[8] = {0}; if (hipMemcpyToSymbol(HIP_SYMBOL(g_mfma_t), z, sizeof(z)) != hipSuccess) return -1; }
  return 0;
}
#endif
void vv_mfma_set_mt(int mt) { g_mt_override = mt; }
void vv_mfma_set_tiled_rows(int r) { g_tiled_rows = r; }
void vv_mfma_set_tiled_bk128(int on) { g_tiled_bk128 = on; }
void vv_mfma_set_tiled_small(int t) { g_tiled_small = t; }
void vv_mfma_set_tiled_small_k(int k) { g_tiled_small_k = k; }
void vv_mfma_set_tiled_small_dual(int t) { g_tiled_small_dual = t; }
void vv_mfma_set_tiled_dual_bk64(int on) { g_tiled_dual_bk64 = on; }
void vv_mfma_set_mt_prefill(int mt) { g_mt_prefill = mt; g_mt_prefill_xb = mt; }

// graph capture must not see the one-time hipFuncSetAttribute calls: the library warms them here
int vv_mfma_gemm_init() {
  hipError_t e;
  const int lds_d = (int)tiled_lds<true, 32>(), lds_l = (int)tiled_lds<false, 128>(), lds_s = (int)tiled_lds<false, 32>();
  const int lds_q = (int)tiled_lds<false, 128, 64>(), lds_qd = (int)tiled_lds<true, 32, 64>(), lds_qd64 = (int)tiled_lds<true, 64, 64>();
  if (hipFuncSetAttribute(reinterpret_cast<const void*>(&mfma_tiled_kernel<true, 64, 64>), hipFuncAttributeMaxDynamicSharedMemorySize, lds_qd64) != hipSuccess ||
      hipFuncSetAttribute(reinterpret_cast<const void*>(&mfma_tiled_kernel<true, 32, 64>), hipFuncAttributeMaxDynamicSharedMemorySize, lds_qd) != hipSuccess ||
      hipFuncSetAttribute(reinterpret_cast<const void*>(&mfma_tiled_kernel<false, 128, 64>), hipFuncAttributeMaxDynamicSharedMemorySize, lds_q) != hipSuccess ||
      hipFuncSetAttribute(reinterpret_cast<const void*>(&mfma_tiled_kernel<false, 128, 128>), hipFuncAttributeMaxDynamicSharedMemorySize, lds_l) != hipSuccess ||
      hipFuncSetAttribute(reinterpret_cast<const void*>(&mfma_tiled_kernel<false, 32, 128>), hipFuncAttributeMaxDynamicSharedMemorySize, lds_s) != hipSuccess ||
      hipFuncSetAttribute(reinterpret_cast<const void*>(&mfma_tiled_kernel<true, 32, 128>), hipFuncAttributeMaxDynamicSharedMemorySize, lds_d) != hipSuccess)
    return vv_set_error(VV_E_HIP, "mfma init: cannot raise the LDS limit of the tiled kernel");
#define VV_ATTR1(D, S, X, MTV)                                                                                           \
  e = hipFuncSetAttribute(reinterpret_cast<const void*>(&mfma_linear_kernel<D, S, X, MTV>),                              \
                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)Tile<MTV>::LDS);                               \
  if (e != hipSuccess) return vv_set_error(VV_E_HIP, "mfma init: %s", hipGetErrorString(e));
#define VV_ATTR(D, S) VV_ATTR1(D, S, false, 1) VV_ATTR1(D, S, false, 2) VV_ATTR1(D, S, false, 4) \
                      VV_ATTR1(D, S, true, 1) VV_ATTR1(D, S, true, 2) VV_ATTR1(D, S, true, 4)
  VV_ATTR(false, false) VV_ATTR(false, true) VV_ATTR(true, false) VV_ATTR(true, true)
#undef VV_ATTR
#undef VV_ATTR1
  return 0;
}
