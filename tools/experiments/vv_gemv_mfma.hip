// vv_gemv_mfma.hip — the per-frame weight-streaming GEMVs (1..4 activation rows, bf16 weights) with the dot products on the matrix cores.
//
// out[m, n] = epilogue( sum_k prologue(x)[m, k] * W[n, k] )  for the decode shapes of the loop: Qwen2 q/k/v, o, gate/up (SwiGLU), down
// projections and the diffusion head's SwiGLU / down GEMVs (modeling_vibevoice_inference.py:478, modular_vibevoice_diffusion_head.py:
// 176-213): M = 2 rows (positive / negative CFG branch) against 5-55 MB of weights per call.
//
// The VALU form (vv_gemv_stream.hip) spends ~24 vector instructions per 16-byte weight load per lane (unpack 8 bf16 + 8 FMAs per
// activation row) and its time grows with the row count: 7.9 / 9.1 / 12.4 us at M = 1 / 2 / 4 for the head's SwiGLU GEMV, against 6.8 us
// for a kernel that only streams the bytes (tools/mb_rows.py) - the arithmetic, not the memory system, sets the pace.  Here the same
// 16-byte loads feed v_mfma_f32_4x4x4_16b_bf16 (16 independent 4x4x4 blocks per wave):
//   lane l = 4 b + q   weight row n0 + q (4 rows per wave step), k chunk b (16 chunks of 8 consecutive k = 128 k per step): one
//                      global_load_dwordx4 per lane covers 4 rows x 256 contiguous bytes
//   A operand          the lane's 8 weights as two 4-element halves (two MFMAs per load)
//   B operand          activation row q, same k chunk, kept in registers for the whole kernel as bf16 hi + lo halves: the product
//                      is exact to ~2^-17 relative, like the fp32-activation FMA form (4 MFMAs per load; rows q >= M hold zeros)
//   D registers        r = partial dot of weight row n0 + r with activation row q over this lane's chunks: summed over the 16 chunk
//                      lanes at the end of a row group (2 DPP rotates + 2 cross-row exchanges per register)
// Two matrices (SwiGLU gate / up) share the B operand.  Activations: the block stages prologue(x) (RMSNorm, norm weight, adaLN shift /
// scale) once in LDS, every lane converts its chunks from there.  Rows longer than 18 steps per wave split K over the block's 4 waves
// (KW = 4, partials through LDS, one row group per workgroup).
#include <hip/hip_runtime.h>
#include <hip/hip_bf16.h>
#include <stdint.h>
#include <math.h>

#include "vv_hip.h"
#include "vv_common.h"

namespace {

typedef unsigned short bf16_t;
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ float silu1(float v) { return v / (1.0f + expf(-v)); }
__device__ __forceinline__ float gelu1(float v) { return 0.5f * v * (1.0f + erff(v * 0.70710678118654752440f)); }
__device__ __forceinline__ unsigned bf16_rne(float f) {      // round to nearest even, finite inputs
  const unsigned u = __float_as_uint(f);
  return (u + 0x7fffu + ((u >> 16) & 1u)) >> 16;
}
// 8 floats -> 8 bf16 "hi" (4 dwords) and the bf16 of the remainders "lo"
__device__ __forceinline__ void split8(const float (&v)[8], u32x4& hi, u32x4& lo) {
  unsigned h[8], l[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    h[j] = bf16_rne(v[j]);
    l[j] = bf16_rne(v[j] - __uint_as_float(h[j] << 16));
  }
  hi.x = h[0] | (h[1] << 16); hi.y = h[2] | (h[3] << 16); hi.z = h[4] | (h[5] << 16); hi.w = h[6] | (h[7] << 16);
  lo.x = l[0] | (l[1] << 16); lo.y = l[2] | (l[3] << 16); lo.z = l[4] | (l[5] << 16); lo.w = l[6] | (l[7] << 16);
}
// 4 floats -> 4 bf16 "hi" and the bf16 of the remainders "lo" (two dwords each)
__device__ __forceinline__ void split4(const float4 v, uint2& hi, uint2& lo) {
  const unsigned h0 = bf16_rne(v.x), h1 = bf16_rne(v.y), h2 = bf16_rne(v.z), h3 = bf16_rne(v.w);
  const unsigned l0 = bf16_rne(v.x - __uint_as_float(h0 << 16)), l1 = bf16_rne(v.y - __uint_as_float(h1 << 16));
  const unsigned l2 = bf16_rne(v.z - __uint_as_float(h2 << 16)), l3 = bf16_rne(v.w - __uint_as_float(h3 << 16));
  hi = make_uint2(h0 | (h1 << 16), h2 | (h3 << 16));
  lo = make_uint2(l0 | (l1 << 16), l2 | (l3 << 16));
}
__device__ __forceinline__ s16x4 lo4(const u32x4 v) { const u32x2 t = {v.x, v.y}; return __builtin_bit_cast(s16x4, t); }
__device__ __forceinline__ s16x4 hi4(const u32x4 v) { const u32x2 t = {v.z, v.w}; return __builtin_bit_cast(s16x4, t); }
// sum over the 16 lanes that share (lane & 3): two rotates inside the 16-lane DPP row, then the other three rows
__device__ __forceinline__ float chunk_sum(float v) {
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x124, 0xF, 0xF, true));   // row_ror:4
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x128, 0xF, 0xF, true));   // row_ror:8
  v += __shfl_xor(v, 16);
  v += __shfl_xor(v, 32);
  return v;
}

// phase timing of workgroup 0 / thread 0, debug builds only (-DVV_CF_TIMING, tools/convffn_phase.py gemv)
#ifdef VV_CF_TIMING
__device__ unsigned long long g_gm_t[8];
#define GSTAMP(i) do { if (blockIdx.x == 0 && threadIdx.x == 0) { const long long t_ = wall_clock64(); g_gm_t[i] += (unsigned long long)(t_ - tprev_); tprev_ = t_; } } while (0)
#else
#define GSTAMP(i) do { } while (0)
#endif

// Workgroup barrier for LDS traffic only.  __syncthreads() carries a workgroup-scope fence that also drains every outstanding GLOBAL load
// (s_waitcnt vmcnt(0)): with the row group's weights requested up front the prologue then only starts once all of them have landed, and
// the kernel becomes "stream everything, then compute" (7.5 + 3.5 us instead of overlapping the two).  The prologue's barriers only
// order LDS writes and reads, so they wait for the LDS counter alone; the weight registers are not touched before their own wait.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// KW: waves of the block that split K (1: a wave owns whole row groups; 4: one row group per block iteration)
// NS: steps (of 128 k) per wave, compile-time bound; MR: activation rows held (2 or 4, >= a.m)
template <int KW, bool DUAL, int NS, int MR>
__global__ __launch_bounds__(256) void gemv_mfma_kernel(const vv_lin_args a, const int n_groups, const int steps_arg) {
  // steps_arg < 0 (tuning experiment "gemv_mfma_nopro", timing only): skip the activation prologue and run on whatever the LDS holds - the
  // time a consumer would take if its producer handed the activations over already normalised and split
  const bool nopro = steps_arg < 0;
  const int steps_total = nopro ? -steps_arg : steps_arg;
  // staged activation rows, already split: bf16 hi [MR][K] then bf16 lo [MR][K] (every thread converts only the chunks it loaded; each
  // lane's MFMA fragments are then two 16-byte LDS reads per step, no conversion in the wave's instruction stream)
  extern __shared__ __attribute__((aligned(16))) unsigned char xs_raw[];
  bf16_t* xh_s = reinterpret_cast<bf16_t*>(xs_raw);
  bf16_t* xl_s = xh_s + (size_t)MR * a.k;
  __shared__ float red[4 * 4];
  __shared__ float part[2][4][8][4];                                // KW = 4: [parity][wave][register][q]
#ifdef VV_CF_TIMING
  long long tprev_ = wall_clock64();
#endif
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int q = lane & 3, b = lane >> 2;
  const int K = a.k, N = a.n, M = a.m;
  const bf16_t* __restrict__ W = reinterpret_cast<const bf16_t*>(a.w);
  const bf16_t* __restrict__ W2 = reinterpret_cast<const bf16_t*>(a.w2);
  const bool reused = (a.flags & VV_LIN_W_REUSED) != 0;
  // this wave's steps [s0, s0 + cnt)
  const int per = (KW == 1) ? steps_total : (steps_total + KW - 1) / KW;
  const int s0 = (KW == 1) ? 0 : wave * per;
  const int cnt = max(0, min(per, steps_total - s0));
  const int gstride = (KW == 1) ? gridDim.x * 4 : gridDim.x;
  int g = (KW == 1) ? blockIdx.x * 4 + wave : blockIdx.x;
  u32x4 wa[NS], wb[DUAL ? NS : 1];
  auto issue = [&](int grp) {
    const int n = min(grp * 4 + q, N - 1);
    const bf16_t* p1 = W + (int64_t)n * K + (int64_t)s0 * 128 + 8 * b;
    const bf16_t* p2 = W2 + (int64_t)n * K + (int64_t)s0 * 128 + 8 * b;
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      const int so = (s < cnt ? s : 0) * 128;                      // steps past the end re-read step 0 (their activations are zero)
      if (reused) {
        wa[s] = *reinterpret_cast<const u32x4*>(p1 + so);
        if (DUAL) wb[s] = *reinterpret_cast<const u32x4*>(p2 + so);
      } else {
        wa[s] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(p1 + so));
        if (DUAL) wb[s] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(p2 + so));
      }
    }
  };
  // ---- activation rows -> LDS (whole block), the first row group's weights requested behind the activation loads -----------------
  if (nopro) {
    issue(g < n_groups ? g : 0);
  } else {
    const int K4 = K >> 2;
    const bool rms = a.pro == VV_PRO_RMSNORM;
    constexpr int NCH = (KW * NS + 7) / 8;                         // float4 chunks per thread per row (K <= KW * NS * 128)
    // Straight-line requests on always-valid addresses (a chunk past the row reads chunk 0, an absent operand reads x): with a guard
    // or a select per load the compiler consumed each value right away and the prologue became one memory round trip per chunk.
    // Norm weight and adaLN shift / scale go out with x, AHEAD of the weights: loads return in order.
    const bool has_nw = rms && a.norm_w, has_mod = rms && a.mod_scale;
    const float* nwp = has_nw ? a.norm_w : a.x;
    const float* scp = has_mod ? a.mod_scale : a.x;
    const float* shp = has_mod ? a.mod_shift : a.x;
    const int64_t ldm = has_mod ? a.ld_mod : 0;
    float4 xv[MR][NCH], nwv[NCH], scv[MR][NCH], shv[MR][NCH];
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      const int ch = tid + 256 * c;
      const int kk = ch < K4 ? 4 * ch : 0;
#pragma unroll
      for (int m = 0; m < MR; ++m) xv[m][c] = *reinterpret_cast<const float4*>(a.x + (int64_t)(m < M ? m : 0) * a.ldx + kk);
      nwv[c] = *reinterpret_cast<const float4*>(nwp + kk);
#pragma unroll
      for (int m = 0; m < MR; ++m) {
        scv[m][c] = *reinterpret_cast<const float4*>(scp + (int64_t)(m < M ? m : 0) * ldm + kk);
        shv[m][c] = *reinterpret_cast<const float4*>(shp + (int64_t)(m < M ? m : 0) * ldm + kk);
      }
    }
    // Every wave of the block has its activation requests in the CU's memory pipeline before any wave adds weight requests behind them
    // (one queue per CU: a wave that starts a little later would find its few activation lines behind 24 KB of another wave's weights).
    asm volatile("s_barrier" ::: "memory");
    issue(g < n_groups ? g : 0);
    __builtin_amdgcn_sched_barrier(0);
#define VV_FENCE4(v) asm volatile("" : "+v"((v).x), "+v"((v).y), "+v"((v).z), "+v"((v).w))
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      VV_FENCE4(nwv[c]);
#pragma unroll
      for (int m = 0; m < MR; ++m) { VV_FENCE4(xv[m][c]); VV_FENCE4(scv[m][c]); VV_FENCE4(shv[m][c]); }
    }
#undef VV_FENCE4
    if (!has_nw) {
#pragma unroll
      for (int c = 0; c < NCH; ++c) nwv[c] = make_float4(1.f, 1.f, 1.f, 1.f);
    }
    if (!has_mod) {
#pragma unroll
      for (int c = 0; c < NCH; ++c)
#pragma unroll
        for (int m = 0; m < MR; ++m) { scv[m][c] = make_float4(0.f, 0.f, 0.f, 0.f); shv[m][c] = make_float4(0.f, 0.f, 0.f, 0.f); }
    }
    GSTAMP(0);                                     // requests out
    if (rms) {
      float ss[MR];
#pragma unroll
      for (int m = 0; m < MR; ++m) {
        float s1 = 0.f;
#pragma unroll
        for (int c = 0; c < NCH; ++c)
          s1 += (tid + 256 * c < K4) ? (xv[m][c].x * xv[m][c].x + xv[m][c].y * xv[m][c].y) + (xv[m][c].z * xv[m][c].z + xv[m][c].w * xv[m][c].w) : 0.f;
        ss[m] = vv_wave_sum(s1);
      }
      if (lane == 0) {
#pragma unroll
        for (int m = 0; m < MR; ++m) red[wave * 4 + m] = ss[m];
      }
      GSTAMP(1);                                   // x landed, statistics
      lds_barrier();
      GSTAMP(2);                                   // first barrier (waits for every outstanding load of the block)
#pragma unroll
      for (int m = 0; m < MR; ++m) {
        if (m >= M) break;
        const float rstd = rsqrtf(((red[m] + red[4 + m]) + (red[8 + m] + red[12 + m])) / (float)K + a.eps);
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
          const int ch = tid + 256 * c;
          if (ch >= K4) break;
          float4 v = xv[m][c];
          v.x *= rstd; v.y *= rstd; v.z *= rstd; v.w *= rstd;
          const float4 w4 = nwv[c], sc = scv[m][c], sh = shv[m][c];
          v.x = v.x * w4.x * (1.0f + sc.x) + sh.x; v.y = v.y * w4.y * (1.0f + sc.y) + sh.y;
          v.z = v.z * w4.z * (1.0f + sc.z) + sh.z; v.w = v.w * w4.w * (1.0f + sc.w) + sh.w;
          uint2 h2, l2;
          split4(v, h2, l2);
          *reinterpret_cast<uint2*>(xh_s + (size_t)m * K + 4 * ch) = h2;
          *reinterpret_cast<uint2*>(xl_s + (size_t)m * K + 4 * ch) = l2;
        }
      }
    } else {
#pragma unroll
      for (int m = 0; m < MR; ++m) {
        if (m >= M) break;
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
          const int ch = tid + 256 * c;
          if (ch < K4) {
            uint2 h2, l2;
            split4(xv[m][c], h2, l2);
            *reinterpret_cast<uint2*>(xh_s + (size_t)m * K + 4 * ch) = h2;
            *reinterpret_cast<uint2*>(xl_s + (size_t)m * K + 4 * ch) = l2;
          }
        }
      }
    }
    GSTAMP(3);                                     // normalise + split + LDS
    lds_barrier();
  }
  GSTAMP(4);                                       // second barrier
  // ---- this lane's B fragments: activation row q, chunk b of every step ------------------------------------------------------------
  u32x4 xh[NS], xl[NS];
#pragma unroll
  for (int s = 0; s < NS; ++s) {
    xh[s] = u32x4{0u, 0u, 0u, 0u};
    xl[s] = xh[s];
    if (q < M && s < cnt) {
      const size_t o = (size_t)q * K + (size_t)(s0 + s) * 128 + 8 * b;
      xh[s] = *reinterpret_cast<const u32x4*>(xh_s + o);
      xl[s] = *reinterpret_cast<const u32x4*>(xl_s + o);
    }
  }
  GSTAMP(5);                                       // fragments from LDS
  // ---- row groups -------------------------------------------------------------------------------------------------------------------
  int parity = 0;
  while (g < n_groups) {
    // epilogue operands of the lane that will finish (row m = q, channels n0 .. n0 + 3): requested now, consumed after the MFMAs
    const int n0 = g * 4;
    const bool owner = (KW == 1 ? true : wave == 0) && b == 0 && q < M;
    float4 eb = make_float4(0.f, 0.f, 0.f, 0.f), eg = make_float4(1.f, 1.f, 1.f, 1.f), er = eb;
    if (owner) {
      if (a.bias) eb = *reinterpret_cast<const float4*>(a.bias + n0);
      if (a.gate) eg = *reinterpret_cast<const float4*>(a.gate + (a.gate_ld ? (int64_t)q * a.gate_ld : 0) + n0);
      if (a.res) er = *reinterpret_cast<const float4*>(a.res + (int64_t)q * a.ldres + n0);
    }
    f32x4 acc = {0.f, 0.f, 0.f, 0.f}, acc2 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      acc = __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(lo4(wa[s]), lo4(xh[s]), acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(hi4(wa[s]), hi4(xh[s]), acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(lo4(wa[s]), lo4(xl[s]), acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(hi4(wa[s]), hi4(xl[s]), acc, 0, 0, 0);
      if (DUAL) {
        acc2 = __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(lo4(wb[s]), lo4(xh[s]), acc2, 0, 0, 0);
        acc2 = __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(hi4(wb[s]), hi4(xh[s]), acc2, 0, 0, 0);
        acc2 = __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(lo4(wb[s]), lo4(xl[s]), acc2, 0, 0, 0);
        acc2 = __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(hi4(wb[s]), hi4(xl[s]), acc2, 0, 0, 0);
      }
    }
    GSTAMP(6);                                     // MFMAs (incl. the wait for the weights)
    const int gn = g + gstride;
    if (gn < n_groups) issue(gn);                                  // the next group's weights stream in behind the reduction / epilogue
    float v[4], v2[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) { v[r] = chunk_sum(acc[r]); v2[r] = DUAL ? chunk_sum(acc2[r]) : 0.f; }
    if (KW != 1) {
      if (b == 0) {
#pragma unroll
        for (int r = 0; r < 4; ++r) { part[parity][wave][r][q] = v[r]; part[parity][wave][4 + r][q] = v2[r]; }
      }
      lds_barrier();                                               // g is block-uniform; partials alternate between two buffers
      if (owner) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          v[r] = (part[parity][0][r][q] + part[parity][1][r][q]) + (part[parity][2][r][q] + part[parity][3][r][q]);
          v2[r] = (part[parity][0][4 + r][q] + part[parity][1][4 + r][q]) + (part[parity][2][4 + r][q] + part[parity][3][4 + r][q]);
        }
      }
      parity ^= 1;
    }
    if (owner && n0 < N) {
      const float bb[4] = {eb.x, eb.y, eb.z, eb.w}, gg[4] = {eg.x, eg.y, eg.z, eg.w}, rr[4] = {er.x, er.y, er.z, er.w};
      float o[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float t = v[r] + bb[r];
        if (a.act == VV_ACT_GELU) t = gelu1(t);
        else if (a.act == VV_ACT_SWIGLU) t = silu1(t) * v2[r];
        o[r] = t * gg[r] + rr[r];
      }
      *reinterpret_cast<float4*>(a.out + (int64_t)q * a.ldo + n0) = make_float4(o[0], o[1], o[2], o[3]);
    }
    g = gn;
    GSTAMP(7);                                     // reduction + epilogue
  }
}

// OFF by default.  Measured on MI355X against the VALU kernel (tools/mb_chain_lin.py ab gemv_mfma 0,1; us per launch inside a graph chain):
// head SwiGLU 11.1 vs 9.1, head down 8.0 vs 5.5, LLM SwiGLU 15.3 vs 13.8, LLM down 18.2 vs 9.0, qkv 5.4 vs 5.3, o 4.8 vs 4.0.  With one row
// group per wave the kernel is "request everything, then compute": the block-wide activation prologue (statistics, normalise, split:
// ~4.5 us of latency behind the CU's own weight traffic, tools/convffn_phase.py gemv) and the MFMA / reduction tail do not overlap with
// the stream the way the VALU kernel's steady-state loop does.  Even with the prologue removed (activations handed over already
// normalised and split - "gemv_mfma_nopro", timing only) it reaches 8.7 / 4.9 / 13.3 / 9.7 / 3.9 / 3.9 us: the upper bound of that
// redesign is ~150 us per frame, not the 450 us the row-count scaling of the VALU kernel suggested.  Kept for that next step and for
// batches of dialogues (its cost does not grow with the row count up to 4); vv_tune("gemv_mfma", 1) enables it.
int g_on = 0;
int g_nopro = 0;
int g_cap = 512;      // persistent workgroups (2 per CU)

template <int KW, bool DUAL, int NS>
void launch(const vv_lin_args& a, hipStream_t s, int steps) {
  const int n_groups = a.n / 4;
  const int work = (KW == 1) ? (n_groups + 3) / 4 : n_groups;
  const int blocks = work < g_cap ? work : g_cap;
  if (g_nopro) steps = -steps;
  if (a.m <= 2) hipLaunchKernelGGL((gemv_mfma_kernel<KW, DUAL, NS, 2>), dim3(blocks), dim3(256), (size_t)2 * a.k * sizeof(float), s, a, n_groups, steps);
  else hipLaunchKernelGGL((gemv_mfma_kernel<KW, DUAL, NS, 4>), dim3(blocks), dim3(256), (size_t)4 * a.k * sizeof(float), s, a, n_groups, steps);
}

}  // namespace

#ifdef VV_CF_TIMING
extern "C" int vv_gemv_mfma_debug_times(unsigned long long* out8, int reset) {
  if (hipMemcpyFromSymbol(out8, HIP_SYMBOL(g_gm_t), 8 * sizeof(unsigned long long)) != hipSuccess) return -1;
  if (reset) { unsigned long long z[8] = {0}; if (hipMemcpyToSymbol(HIP_SYMBOL(g_gm_t), z, sizeof(z)) != hipSuccess) return -1; }
  return 0;
}
#endif
void vv_gemv_mfma_set(int on, int cap) { g_on = on; if (cap > 0) g_cap = cap; }
void vv_gemv_mfma_set_nopro(int on) { g_nopro = on; }

int vv_gemv_mfma_init() {     // before any graph capture: the staged rows can exceed the default dynamic LDS limit
#define VV_GM_ATTR1(KWV, D, NSV, MRV)                                                                                                       \
  if (hipFuncSetAttribute(reinterpret_cast<const void*>(&gemv_mfma_kernel<KWV, D, NSV, MRV>), hipFuncAttributeMaxDynamicSharedMemorySize, \
                          KWV * NSV * 128 * MRV * 4) != hipSuccess)                                                                       \
    return vv_set_error(VV_E_HIP, "vv_gemv_mfma_init: cannot raise the LDS limit");
#define VV_GM_ATTR(KWV, D, NSV) VV_GM_ATTR1(KWV, D, NSV, 2) VV_GM_ATTR1(KWV, D, NSV, 4)
  VV_GM_ATTR(1, false, 12) VV_GM_ATTR(1, true, 12) VV_GM_ATTR(4, false, 9) VV_GM_ATTR(4, false, 18) VV_GM_ATTR(4, true, 7) VV_GM_ATTR(4, false, 7)
  VV_GM_ATTR(1, false, 16) VV_GM_ATTR(4, false, 16)
#undef VV_GM_ATTR
#undef VV_GM_ATTR1
  return 0;
}

// 1 = launched, 0 = not covered (caller falls back to the VALU kernel)
int vv_launch_gemv_mfma(const vv_lin_args& a, hipStream_t s) {
  if (!g_on || a.wdt != VV_BF16 || a.m < 1 || a.m > 4 || a.k % 128 || a.n % 4 || a.k > 9216) return 0;
  if (a.pro != VV_PRO_NONE && a.pro != VV_PRO_RMSNORM) return 0;
  if (a.mod_scale && (!a.mod_shift || a.pro != VV_PRO_RMSNORM)) return 0;
  if (a.flags & ~VV_LIN_W_REUSED) return 0;
  auto a16 = [](const void* q) { return ((uintptr_t)q % 16) == 0; };
  if (!a16(a.w) || (a.w2 && !a16(a.w2)) || !a16(a.x) || (a.m > 1 && a.ldx % 4) || !a16(a.out) || a.ldo % 4) return 0;
  if (a.m > 2 && (size_t)4 * a.k * 4 > 147456) return 0;
  if (a.norm_w && !a16(a.norm_w)) return 0;
  if (a.mod_scale && (!a16(a.mod_scale) || !a16(a.mod_shift) || a.ld_mod % 4)) return 0;
  if (a.bias && !a16(a.bias)) return 0;
  if (a.gate && (!a16(a.gate) || a.gate_ld % 4)) return 0;
  if (a.res && (!a16(a.res) || a.ldres % 4)) return 0;
  if (a.w2 && a.act != VV_ACT_SWIGLU) return 0;
  const int steps = a.k / 128;
  const bool dual = a.w2 != nullptr;
  if (steps == 12) { if (dual) launch<1, true, 12>(a, s, steps); else launch<1, false, 12>(a, s, steps); }
  else if (steps == 16 && !dual) launch<1, false, 16>(a, s, steps);
  else if (steps == 28) { if (dual) launch<4, true, 7>(a, s, steps); else launch<4, false, 7>(a, s, steps); }
  else if (steps <= 36 && steps > 28 && !dual) launch<4, false, 9>(a, s, steps);
  else if (steps <= 64 && steps > 36 && !dual) launch<4, false, 16>(a, s, steps);
  else if (steps <= 72 && steps > 64 && !dual) launch<4, false, 18>(a, s, steps);
  else return 0;
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return vv_set_error(VV_E_HIP, "vv_gemv_mfma: %s", hipGetErrorString(e));
  return 1;
}
