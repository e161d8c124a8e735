// vv_chain.hip — persistent "chained" kernel for the diffusion head's solver loop (batch-2 weight streaming).
//
// Why: a solver step is 10 dependent GEMVs of 0.2-28 MB each.  Launched one by one, every GEMV pays the kernel boundary
// (drain, ~1.7 us idle, refill of the memory pipe, the activation prologue in front of the first FMA): tools/mb_steady.py
// measures 2.3-3.0 TB/s per call against 5.0-6.1 TB/s for the same loop on a 16x taller matrix.  Here ONE grid (one
// 512-thread workgroup per CU) walks all phases of all solver steps:
//   * waves 0..5 of a workgroup STREAM weights.  As soon as a wave has stored its outputs of phase p it issues the loads of
//     all its weight rows of phase p+1 (up to 18 x 1 KB per wave, ~all of the next matrix device-wide): weights do not
//     depend on activations, so the memory pipe stays full while the hand-off below happens.
//   * waves 6..7 are HELPERS, one per activation row (conditional / unconditional branch).  They have no weight loads in
//     flight (loads return in order per wave, a poll behind 18 KB of prefetch would wait for all of it), so they see the
//     producers' stores one memory round trip after they land; they run the fused prologue (RMSNorm x adaLN modulate, or
//     the CFG + DPM-Solver++ update) once per workgroup and leave the activation rows in LDS for the streaming waves
//     (double-buffered: the only workgroup barrier per phase is "rows of phase q are in LDS").
// Hand-off between phases, without any cache maintenance instruction (a release/acquire fence pair per wave is an L2
// write-back + invalidate: measured ~30 us per phase):
//   * every hand-off access is a relaxed AGENT-scope atomic (sc1: performed at the device coherence point, never served
//     from / parked in one XCD's L2).  Weights and modulation tables are immutable while the kernel runs: plain loads.
//   * the residual stream and the head output are arrays of {fp32 value, epoch tag} 64-bit words: a consumer's load IS
//     its poll (data is valid when every tag equals the expected epoch), one round trip instead of store -> flag -> poll
//     -> load.
//   * the SwiGLU activations (4608 x 2, too large to poll) use a flag barrier: every workgroup stores the epoch into its
//     own flag word once the stores of its streaming waves have landed (counted in LDS); helpers read all flags with one
//     coalesced load per poll.  This is also
//     the only full barrier and what makes the tagged buffers safe to overwrite (a workgroup can only be one tagged phase
//     ahead of the slowest one).
//   * all workgroups must be co-resident: grid <= CU count (one workgroup per CU).  Two such kernels from different
//     streams may each get only part of the device and wait for each other: the path is opt-in (VV_HEAD_CHAIN) for
//     callers that own the GPU; every poll gives up after 2 s (error flag, output = NaN) instead of hanging the GPU.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <string.h>

#include "vv_hip.h"
#include "vv_common.h"

namespace {

typedef unsigned short bf16_t;
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned long long u64;

constexpr int SW = 6;                  // streaming waves per workgroup (6 x 256 CUs = 1536: 4608 SwiGLU rows = 3 per wave, 1536 down rows = 1 per wave)
constexpr int HW = 2;                  // helper waves (activation row 0 / 1)
constexpr int CT = (SW + HW) * 64;
constexpr int KU = 3;                  // mode R: 512-element K units of the register-resident activation fragment
constexpr int NB = 3;                  // mode R: row groups in flight per wave
constexpr int NT = 2 * KU * NB;        // 16-byte weight tiles a wave keeps in flight (mode L: units of one long row)
constexpr int MAX_LAYERS = 8;
constexpr int MAX_STEPS = 32;
constexpr int BAR_ERR = 256;           // flag words: [0, 256) one per workgroup, [256] error
constexpr int BAR_WORDS = 320;
constexpr long long POLL_LIMIT = 200000000LL;   // 2 s of the 100 MHz wall clock

__device__ __forceinline__ float wsum(float v) { return vv_wave_sum(v); }   // DPP row reduction, all 64 lanes active
__device__ __forceinline__ float silu1(float v) { return v / (1.0f + expf(-v)); }
__device__ __forceinline__ void unpack8(const u32x4 v, float (&o)[8]) {
  o[0] = __uint_as_float(v.x << 16); o[1] = __uint_as_float(v.x & 0xffff0000u);
  o[2] = __uint_as_float(v.y << 16); o[3] = __uint_as_float(v.y & 0xffff0000u);
  o[4] = __uint_as_float(v.z << 16); o[5] = __uint_as_float(v.z & 0xffff0000u);
  o[6] = __uint_as_float(v.w << 16); o[7] = __uint_as_float(v.w & 0xffff0000u);
}

// Every pointer reaches the kernel through the LDS copy of the descriptor, so the compiler only knows them as generic
// pointers: a plain dereference would be a FLAT load, which also counts on lgkmcnt - every wait for an LDS read would then
// wait for all prefetched weights.  All global accesses therefore go through address-space-1 pointers.
#define VV_GLOBAL __attribute__((address_space(1)))
template <typename T>
__device__ __forceinline__ T VV_GLOBAL* gp(T* p) { return (T VV_GLOBAL*)p; }
typedef float f32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 gld4(const float* p) {
  const f32x4 v = *(const f32x4 VV_GLOBAL*)p;
  return make_float4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ float gld1(const float* p) { return *(const float VV_GLOBAL*)p; }

// ---- hand-off accessors: relaxed, agent scope --------------------------------------------------------------------------
__device__ __forceinline__ unsigned ldu_co(const unsigned* p) { return __hip_atomic_load(gp(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ u64 ld64_co(const u64* p) { return __hip_atomic_load(gp(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ float ld_co(const float* p) { return __uint_as_float(ldu_co(reinterpret_cast<const unsigned*>(p))); }
__device__ __forceinline__ void stu_co(unsigned* p, unsigned v) { __hip_atomic_store(gp(p), v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_co(float* p, float v) { stu_co(reinterpret_cast<unsigned*>(p), __float_as_uint(v)); }
__device__ __forceinline__ void st64_co(u64* p, u64 v) { __hip_atomic_store(gp(p), v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ u64 tagged(float v, unsigned tag) { return (u64)__float_as_uint(v) | ((u64)tag << 32); }

// wave-uniform values that came through LDS: move them to SGPRs
__device__ __forceinline__ int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ float unif(float v) { return __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(v))); }
template <typename T>
__device__ __forceinline__ T* unip(T* p) {
  const uintptr_t a = reinterpret_cast<uintptr_t>(p);
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)a), hi = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
  return reinterpret_cast<T*>(((uintptr_t)hi << 32) | lo);
}

// one phase; every field is wave-uniform
struct Ph {
  int kind;                                 // 0: GEMV, 1: solver update + K = latent projection
  const bf16_t* w; const bf16_t* w2;        // w2 != null: dual (SwiGLU), a row group = {w row g, w2 row g}; else {w rows 2g, 2g+1}
  int n, k, kpad, longk, nt;                // longk: the row does not fit the register fragment: one row per wave step, x read from LDS
  const void* x; int ldx;                   // x_tag != 0: u64 {value, tag} elements; else fp32, valid once flag epoch x_flag is reached
  unsigned x_tag, x_flag;
  int pro; const float* norm_w; const float* mod_shift; const float* mod_scale; int ld_mod; float eps;
  int act; const float* gate; int gate_ld; const u64* res; int ldres;
  void* out; int ldo;                       // out_tag != 0: u64 {value, tag}; else fp32
  unsigned out_tag, out_flag;               // out_flag != 0: publish this flag epoch once the outputs have landed
  int step;
};

// ---- weight prefetch ---------------------------------------------------------------------------------------------------
// mode R tile index: slot * 2 KU + stream * KU + unit;  mode L: unit
// One 16-byte tile per lane of unit `u` of a weight row: scalar row/unit base + one shared 32-bit lane offset (the
// compiler keeps a single VGPR for all units instead of one 64-bit address each); only a partial last unit clamps its
// lanes into the row (their activations are 0).
__device__ __forceinline__ u32x4 wload(const bf16_t* row, int u, int k, int lane, int nt) {
  const char* base = reinterpret_cast<const char*>(row) + (size_t)u * 1024;
  unsigned voff = (unsigned)lane * 16u;
  if ((u + 1) * 512 > k) voff = min(voff, (unsigned)(k - u * 512 - 8) * 2u);
  const u32x4 VV_GLOBAL* q = (const u32x4 VV_GLOBAL*)(base + voff);
  return nt ? __builtin_nontemporal_load(q) : *q;
}

template <int SLOT>
__device__ __forceinline__ void issue_r(const Ph& p, int g, int ng, int lane, u32x4 (&b)[NT]) {
  if (g >= ng) return;
  const bool dual = p.w2 != nullptr;
  const int r0 = dual ? g : 2 * g;
  const int r1 = dual ? g : min(2 * g + 1, p.n - 1);
  const bf16_t* a0 = p.w + (int64_t)r0 * p.k;
  const bf16_t* a1 = (dual ? p.w2 : p.w) + (int64_t)r1 * p.k;
#pragma unroll
  for (int u = 0; u < KU; ++u) {
    if (u * 512 >= p.k) continue;
    b[SLOT * 2 * KU + u] = wload(a0, u, p.k, lane, p.nt);
    b[SLOT * 2 * KU + KU + u] = wload(a1, u, p.k, lane, p.nt);
  }
}

// epilogue operands of a long row (adaLN gate, residual): known before the activations are, so they travel with the weights
struct LEpi { float g[2], r[2]; };

__device__ __forceinline__ void issue_l(const Ph& p, int r, int lane, u32x4 (&b)[NT], LEpi& e) {
  if (r >= p.n) return;
#pragma unroll
  for (int m = 0; m < 2; ++m) {
    e.g[m] = p.gate ? gld1(p.gate + (int64_t)m * p.gate_ld + r) : 1.f;
    e.r[m] = p.res ? __uint_as_float((unsigned)ld64_co(p.res + (int64_t)m * p.ldres + r)) : 0.f;
  }
  const bf16_t* a0 = p.w + (int64_t)r * p.k;
#pragma unroll
  for (int u = 0; u < NT; ++u) {
    if (u * 512 >= p.k) continue;
    b[u] = wload(a0, u, p.k, lane, p.nt);
  }
}

__device__ __forceinline__ void prefetch(const Ph& p, int wg, int wave, int lane, u32x4 (&b)[NT], LEpi& e) {
  if (p.nt < 0) return;                                       // timing experiments: no weight traffic
  const int w0 = wg * SW + wave, ws = gridDim.x * SW;
  if (p.longk) {
    issue_l(p, w0, lane, b, e);
  } else {
    const int ng = p.w2 ? p.n : (p.n + 1) >> 1;
    issue_r<0>(p, w0, ng, lane, b);
    issue_r<1>(p, w0 + ws, ng, lane, b);
    issue_r<2>(p, w0 + 2 * ws, ng, lane, b);
  }
}

// ---- epilogue (one lane per output) ------------------------------------------------------------------------------------
__device__ __forceinline__ void epi(const Ph& p, int m, int n, float v, float v2) {
  if (p.act == VV_ACT_SWIGLU) v = silu1(v) * v2;
  if (p.gate) v *= gld1(p.gate + (int64_t)m * p.gate_ld + n);
  if (p.res) v += __uint_as_float((unsigned)ld64_co(p.res + (int64_t)m * p.ldres + n));
  if (p.out_tag) st64_co(reinterpret_cast<u64*>(p.out) + (int64_t)m * p.ldo + n, tagged(v, p.out_tag));
  else st_co(reinterpret_cast<float*>(p.out) + (int64_t)m * p.ldo + n, v);
}


// ---- streaming waves ---------------------------------------------------------------------------------------------------
template <int SLOT>
__device__ __forceinline__ void consume_r(const Ph& p, int g, int lane, const float (&xr)[2][KU][8], const u32x4 (&b)[NT]) {
  const bool dual = p.w2 != nullptr;
  float acc[2][2] = {{0.f, 0.f}, {0.f, 0.f}};
#pragma unroll
  for (int u = 0; u < KU; ++u) {
    if (u * 512 >= p.k) continue;
    float w0[8], w1[8];
    unpack8(b[SLOT * 2 * KU + u], w0);
    unpack8(b[SLOT * 2 * KU + KU + u], w1);
#pragma unroll
    for (int m = 0; m < 2; ++m) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        acc[0][m] = fmaf(w0[j], xr[m][u][j], acc[0][m]);
        acc[1][m] = fmaf(w1[j], xr[m][u][j], acc[1][m]);
      }
    }
  }
#pragma unroll
  for (int s = 0; s < 2; ++s)
#pragma unroll
    for (int m = 0; m < 2; ++m) acc[s][m] = wsum(acc[s][m]);
  if (lane == 0) {
    if (dual) {
      epi(p, 0, g, acc[0][0], acc[1][0]);
      epi(p, 1, g, acc[0][1], acc[1][1]);
    } else {
      epi(p, 0, 2 * g, acc[0][0], 0.f);
      epi(p, 1, 2 * g, acc[0][1], 0.f);
      if (2 * g + 1 < p.n) { epi(p, 0, 2 * g + 1, acc[1][0], 0.f); epi(p, 1, 2 * g + 1, acc[1][1], 0.f); }
    }
  }
}

__device__ __forceinline__ void stream_r(const Ph& p, int wg, int wave, int lane, const float* xs, u32x4 (&b)[NT]) {
  float xr[2][KU][8];
#pragma unroll
  for (int m = 0; m < 2; ++m)
#pragma unroll
    for (int u = 0; u < KU; ++u) {
      if (u * 512 < p.k) {                                    // padded with zeros up to kpad by the helpers
        const float4 a = *reinterpret_cast<const float4*>(xs + m * p.kpad + u * 512 + lane * 8);
        const float4 c = *reinterpret_cast<const float4*>(xs + m * p.kpad + u * 512 + lane * 8 + 4);
        xr[m][u][0] = a.x; xr[m][u][1] = a.y; xr[m][u][2] = a.z; xr[m][u][3] = a.w;
        xr[m][u][4] = c.x; xr[m][u][5] = c.y; xr[m][u][6] = c.z; xr[m][u][7] = c.w;
      } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) xr[m][u][j] = 0.f;
      }
    }
  const int ng = p.w2 ? p.n : (p.n + 1) >> 1;
  const int ws = gridDim.x * SW;
  int g = wg * SW + wave;
  for (;;) {                                                  // slots have fixed roles: no register rotation
    if (g >= ng) break;
    consume_r<0>(p, g, lane, xr, b); issue_r<0>(p, g + 3 * ws, ng, lane, b); g += ws;
    if (g >= ng) break;
    consume_r<1>(p, g, lane, xr, b); issue_r<1>(p, g + 3 * ws, ng, lane, b); g += ws;
    if (g >= ng) break;
    consume_r<2>(p, g, lane, xr, b); issue_r<2>(p, g + 3 * ws, ng, lane, b); g += ws;
  }
}

__device__ __forceinline__ void store_out(const Ph& p, int m, int n, float v) {
  if (p.out_tag) st64_co(reinterpret_cast<u64*>(p.out) + (int64_t)m * p.ldo + n, tagged(v, p.out_tag));
  else st_co(reinterpret_cast<float*>(p.out) + (int64_t)m * p.ldo + n, v);
}

__device__ __forceinline__ void stream_l(const Ph& p, int wg, int wave, int lane, const float* xs, u32x4 (&b)[NT], LEpi& e) {
  const int ws = gridDim.x * SW;
  for (int r = wg * SW + wave; r < p.n; r += ws) {
    float acc0 = 0.f, acc1 = 0.f;
#pragma unroll
    for (int u = 0; u < NT; ++u) {
      if (u * 512 >= p.k) continue;
      float w0[8];
      unpack8(b[u], w0);
      const float* x0 = xs + u * 512 + lane * 8;
      const float* x1 = x0 + p.kpad;
      const float4 a0 = *reinterpret_cast<const float4*>(x0), a1 = *reinterpret_cast<const float4*>(x0 + 4);
      const float4 c0 = *reinterpret_cast<const float4*>(x1), c1 = *reinterpret_cast<const float4*>(x1 + 4);
      acc0 = fmaf(w0[0], a0.x, acc0); acc0 = fmaf(w0[1], a0.y, acc0); acc0 = fmaf(w0[2], a0.z, acc0); acc0 = fmaf(w0[3], a0.w, acc0);
      acc0 = fmaf(w0[4], a1.x, acc0); acc0 = fmaf(w0[5], a1.y, acc0); acc0 = fmaf(w0[6], a1.z, acc0); acc0 = fmaf(w0[7], a1.w, acc0);
      acc1 = fmaf(w0[0], c0.x, acc1); acc1 = fmaf(w0[1], c0.y, acc1); acc1 = fmaf(w0[2], c0.z, acc1); acc1 = fmaf(w0[3], c0.w, acc1);
      acc1 = fmaf(w0[4], c1.x, acc1); acc1 = fmaf(w0[5], c1.y, acc1); acc1 = fmaf(w0[6], c1.z, acc1); acc1 = fmaf(w0[7], c1.w, acc1);
      __builtin_amdgcn_sched_barrier(0);        // keep the LDS reads of later units from being hoisted (registers)
    }
    acc0 = wsum(acc0); acc1 = wsum(acc1);
    if (lane == 0) { store_out(p, 0, r, fmaf(acc0, e.g[0], e.r[0])); store_out(p, 1, r, fmaf(acc1, e.g[1], e.r[1])); }
    issue_l(p, r + ws, lane, b, e);
  }
}

// ---- helper waves ------------------------------------------------------------------------------------------------------
__device__ __forceinline__ bool poll_failed(unsigned* bar, long long t0) {   // wave-uniform
  if (ldu_co(bar + BAR_ERR) != 0u) return true;
  if (wall_clock64() - t0 > POLL_LIMIT) { stu_co(bar + BAR_ERR, 1u); return true; }
  return false;
}

// all workgroups have published flag epoch `epoch` (lane i checks flags 4i .. 4i+3)
__device__ __forceinline__ void wait_flags(unsigned* bar, unsigned epoch, int lane) {
  const long long t0 = wall_clock64();
  const int nb = gridDim.x;
  for (;;) {
    const u64 a = ld64_co(reinterpret_cast<const u64*>(bar) + 2 * lane);
    const u64 b = ld64_co(reinterpret_cast<const u64*>(bar) + 2 * lane + 1);
    const unsigned f0 = (unsigned)a, f1 = (unsigned)(a >> 32), f2 = (unsigned)b, f3 = (unsigned)(b >> 32);
    const int i0 = 4 * lane;
    const bool ok = (i0 >= nb || f0 >= epoch) && (i0 + 1 >= nb || f1 >= epoch) && (i0 + 2 >= nb || f2 >= epoch) && (i0 + 3 >= nb || f3 >= epoch);
    if (__all(ok)) break;
    if (poll_failed(bar, t0)) break;
    __builtin_amdgcn_s_sleep(4);
  }
}

// helper wave `m` fetches activation row m of a GEMV phase: part 1 (may run before the workgroup's streaming waves have
// finished the previous phase: touches registers only).  Short rows (k <= 512 KU) are tagged and go through registers so
// RMSNorm + modulation can be applied; long rows are plain fp32 behind a flag barrier and are copied in part 2.
struct HelperRegs { float xv[KU][8]; float a[KU][8]; float sft[KU][8]; };   // y = x * rstd * a + sft

__device__ __forceinline__ void helper_fetch(const Ph& p, int m, int lane, unsigned* bar, HelperRegs& h) {
  if (p.longk || !p.x_tag) return;
  // the norm weight and the adaLN modulation are immutable: fetch them before (not after) the wait for the producers
#pragma unroll
  for (int u = 0; u < KU; ++u) {
    const int o = u * 512 + lane * 8;
    float nw[8], sh[8], sc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { nw[j] = 1.f; sh[j] = 0.f; sc[j] = 0.f; }
    if (p.pro == VV_PRO_RMSNORM && u * 512 < p.k && o < p.k) {
      if (p.norm_w) {
        const float4 a = gld4(p.norm_w + o), c = gld4(p.norm_w + o + 4);
        nw[0] = a.x; nw[1] = a.y; nw[2] = a.z; nw[3] = a.w; nw[4] = c.x; nw[5] = c.y; nw[6] = c.z; nw[7] = c.w;
      }
      if (p.mod_scale) {
        const int64_t mo = (int64_t)m * p.ld_mod + o;
        const float4 s0 = gld4(p.mod_shift + mo), s1 = gld4(p.mod_shift + mo + 4);
        const float4 c0 = gld4(p.mod_scale + mo), c1 = gld4(p.mod_scale + mo + 4);
        sh[0] = s0.x; sh[1] = s0.y; sh[2] = s0.z; sh[3] = s0.w; sh[4] = s1.x; sh[5] = s1.y; sh[6] = s1.z; sh[7] = s1.w;
        sc[0] = c0.x; sc[1] = c0.y; sc[2] = c0.z; sc[3] = c0.w; sc[4] = c1.x; sc[5] = c1.y; sc[6] = c1.z; sc[7] = c1.w;
      }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) { h.a[u][j] = nw[j] * (1.0f + sc[j]); h.sft[u][j] = sh[j]; }
  }
  const u64* xt = reinterpret_cast<const u64*>(p.x) + (int64_t)m * p.ldx;
  const long long t0 = wall_clock64();
  const int stride = p.k >> 6;                                 // sample: 64 elements spread over the row (512 B per poll, not 12 KB)
  for (;;) {
    const bool sok = (unsigned)(ld64_co(xt + min(lane * stride, p.k - 1)) >> 32) == p.x_tag;
    if (!__all(sok)) {
      if (poll_failed(bar, t0)) break;
      __builtin_amdgcn_s_sleep(8);
      continue;
    }
    bool ok = true;
#pragma unroll
    for (int u = 0; u < KU; ++u) {
      const int o = u * 512 + lane * 8;
      if (u * 512 < p.k && o < p.k) {
        u64 v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = ld64_co(xt + o + j);
#pragma unroll
        for (int j = 0; j < 8; ++j) { ok = ok && ((unsigned)(v[j] >> 32) == p.x_tag); h.xv[u][j] = __uint_as_float((unsigned)v[j]); }
      } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) h.xv[u][j] = 0.f;
      }
    }
    if (__all(ok)) break;
    if (poll_failed(bar, t0)) break;
    __builtin_amdgcn_s_sleep(2);
  }
}

// part 2: after the workgroup barrier that says "the streaming waves are done with the LDS rows of the previous phase"
__device__ __forceinline__ void helper_publish(const Ph& p, int m, int lane, unsigned* bar, HelperRegs& h, float* xs) {
  float* row = xs + m * p.kpad;
  if (p.longk || !p.x_tag) {                                  // plain fp32 row behind the flag barrier: straight copy
    if (p.x_flag) wait_flags(bar, p.x_flag, lane);
    const float* xrow = reinterpret_cast<const float*>(p.x) + (int64_t)m * p.ldx;
    // all loads of a batch of 6 units (24 x 8 bytes per lane) are issued before the first one is used: one round trip per
    // batch, not one per unit
    for (int u0 = 0; u0 * 512 < p.kpad; u0 += 6) {
      u64 q[6][4];
#pragma unroll
      for (int i = 0; i < 6; ++i) {
        const int o = (u0 + i) * 512 + lane * 8;
#pragma unroll
        for (int j = 0; j < 4; ++j) q[i][j] = (o < p.k) ? ld64_co(reinterpret_cast<const u64*>(xrow + o) + j) : 0ull;
      }
#pragma unroll
      for (int i = 0; i < 6; ++i) {
        const int o = (u0 + i) * 512 + lane * 8;
        if (o < p.kpad) {
          *reinterpret_cast<float4*>(row + o) = make_float4(__uint_as_float((unsigned)q[i][0]), __uint_as_float((unsigned)(q[i][0] >> 32)),
                                                            __uint_as_float((unsigned)q[i][1]), __uint_as_float((unsigned)(q[i][1] >> 32)));
          *reinterpret_cast<float4*>(row + o + 4) = make_float4(__uint_as_float((unsigned)q[i][2]), __uint_as_float((unsigned)(q[i][2] >> 32)),
                                                                __uint_as_float((unsigned)q[i][3]), __uint_as_float((unsigned)(q[i][3] >> 32)));
        }
      }
    }
    return;
  }
  float rstd = 1.f;
  if (p.pro == VV_PRO_RMSNORM) {
    float ss = 0.f;
#pragma unroll
    for (int u = 0; u < KU; ++u)
#pragma unroll
      for (int j = 0; j < 8; ++j) ss = fmaf(h.xv[u][j], h.xv[u][j], ss);
    rstd = rsqrtf(wsum(ss) / (float)p.k + p.eps);
  }
#pragma unroll
  for (int u = 0; u < KU; ++u) {
    if (u * 512 >= p.kpad) continue;
    const int o = u * 512 + lane * 8;
    float y[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) y[j] = (p.pro == VV_PRO_RMSNORM) ? fmaf(h.xv[u][j] * rstd, h.a[u][j], h.sft[u][j]) : h.xv[u][j];
    *reinterpret_cast<float4*>(row + o) = make_float4(y[0], y[1], y[2], y[3]);
    *reinterpret_cast<float4*>(row + o + 4) = make_float4(y[4], y[5], y[6], y[7]);
  }
}

// ---- diffusion head ----------------------------------------------------------------------------------------------------
struct HeadLayerDev { const float* norm_w; const bf16_t* wgate; const bf16_t* wup; const bf16_t* wdown; const float* mod; };
struct HeadChain {
  int D, ffn, layers, latent, n_steps;
  float eps, cfg;
  const bf16_t* noisy_proj; const bf16_t* final_linear;
  HeadLayerDev layer[MAX_LAYERS];            // mod: [2 n_steps, 3D] shift | scale | gate
  const float* modf;                         // [2 n_steps, 2D]
  const float* noise; float* latent_out;
  u64* hcur;                                 // tagged residual stream [2, D]
  float* act;                                // [2, ffn] (flag barrier)
  u64* v;                                    // tagged head output [2, latent]
  float* xb[2]; float* mb[2];
  unsigned* bar;
  int dbg_block, dbg_thread, dbg_mode;      // dbg_mode (timing experiments only): 1 = helpers skip the hand-off, 2 = no weight loads
  vv_dpm_coef coef[MAX_STEPS];
};

// phase ph -> description.  slot 0: solver update + noisy_images_proj, 1+2l: gate/up, 2+2l: down, 2L+1: final layer.
// Epochs: the residual stream is rewritten L+1 times per step (projection + every down layer); the flag barrier runs once
// per layer (SwiGLU output).
__device__ __forceinline__ void head_phase(const HeadChain& c, int ph, Ph& p) {
  const int layers = uni(c.layers), D = uni(c.D), ffn = uni(c.ffn), latent = uni(c.latent), n_steps = uni(c.n_steps);
  const int per = 2 * layers + 2;
  const int step = ph / per, slot = ph - step * per;
  const unsigned hbase = (unsigned)(step * (layers + 1));
  p.kind = 0; p.w2 = nullptr; p.pro = VV_PRO_NONE; p.act = VV_ACT_NONE; p.nt = 0; p.longk = 0;
  p.norm_w = nullptr; p.mod_shift = nullptr; p.mod_scale = nullptr; p.ld_mod = 0; p.eps = unif(c.eps);
  p.gate = nullptr; p.gate_ld = 0; p.res = nullptr; p.ldres = 0;
  p.x_tag = 0; p.x_flag = 0; p.out_tag = 0; p.out_flag = 0; p.step = step;
  if (slot == 0) {
    p.kind = 1;
    p.w = unip(c.noisy_proj); p.x = nullptr; p.ldx = 0; p.k = latent;
    p.n = (step == n_steps) ? 0 : D;
    p.out = unip(c.hcur); p.ldo = D; p.out_tag = hbase + 1;
  } else if (slot == per - 1) {
    const float* mf = unip(c.modf) + (int64_t)(2 * step) * 2 * D;
    p.w = unip(c.final_linear); p.x = unip(c.hcur); p.ldx = D; p.x_tag = hbase + 1 + layers; p.n = latent; p.k = D;
    p.pro = VV_PRO_RMSNORM; p.mod_shift = mf; p.mod_scale = mf + D; p.ld_mod = 2 * D;
    p.out = unip(c.v); p.ldo = latent; p.out_tag = (unsigned)step + 1;
  } else {
    const int l = (slot - 1) >> 1;
    const HeadLayerDev& Lr = c.layer[l];
    const float* ml = unip(Lr.mod) + (int64_t)(2 * step) * 3 * D;
    if ((slot - 1) & 1) {                     // down projection: act -> hcur += gate * (Wd act)
      p.w = unip(Lr.wdown); p.x = unip(c.act); p.ldx = ffn; p.x_flag = (unsigned)(step * layers + l + 1); p.n = D; p.k = ffn;
      p.longk = 1;
      p.gate = ml + 2 * D; p.gate_ld = 3 * D; p.res = unip(c.hcur); p.ldres = D;
      p.out = unip(c.hcur); p.ldo = D; p.out_tag = hbase + 2 + l;
    } else {                                  // SwiGLU: modulate(rmsnorm(hcur)) -> act
      p.w = unip(Lr.wgate); p.w2 = unip(Lr.wup); p.x = unip(c.hcur); p.ldx = D; p.x_tag = hbase + 1 + l; p.n = ffn; p.k = D;
      p.pro = VV_PRO_RMSNORM; p.norm_w = unip(Lr.norm_w); p.mod_shift = ml; p.mod_scale = ml + D; p.ld_mod = 3 * D;
      p.act = VV_ACT_SWIGLU;
      p.out = unip(c.act); p.ldo = ffn; p.out_flag = (unsigned)(step * layers + l + 1);
    }
  }
  p.kpad = (p.k + 511) & ~511;
  if (uni(c.dbg_mode) & 2) p.nt = -1;
}

// helper wave 0, solver phase: CFG + DPM-Solver++ update of step `step - 1` (none before step 0); lane i owns element i of
// the <= 64-element latent.  Returns the new x (registers only; v is tagged, so this is also the wait for the final layer).
__device__ __forceinline__ float dpm_fetch(const HeadChain& c, int step, int lane, bool& own_out, float& x0_out) {
  const int latent = uni(c.latent);
  const bool own = lane < latent;
  own_out = own;
  const float* x_in = (step == 0) ? unip(c.noise) : unip(c.xb[(step - 1) & 1]);
  float xn = own ? ld_co(x_in + lane) : 0.f;
  x0_out = 0.f;
  if (step > 0) {
    const vv_dpm_coef k = c.coef[step - 1];
    const u64* vt = unip(c.v);
    float vc = 0.f, vu = 0.f;
    const long long t0 = wall_clock64();
    for (;;) {
      bool ok = true;
      if (own) {
        const u64 a = ld64_co(vt + lane), b = ld64_co(vt + latent + lane);
        ok = ((unsigned)(a >> 32) == (unsigned)step) && ((unsigned)(b >> 32) == (unsigned)step);
        vc = __uint_as_float((unsigned)a); vu = __uint_as_float((unsigned)b);
      }
      if (__all(ok)) break;
      if (poll_failed(unip(c.bar), t0)) break;
      __builtin_amdgcn_s_sleep(4);
    }
    const float mprev = own ? ld_co(unip(c.mb[(step - 1) & 1]) + lane) : 0.f;
    const float eps = vu + unif(c.cfg) * (vc - vu);
    const float x0 = unif(k.alpha_s) * xn - unif(k.sigma_s) * eps;
    float xt = unif(k.cx) * xn - unif(k.cd) * x0;
    if (uni(k.order) == 2) xt -= 0.5f * unif(k.cd) * (unif(k.rinv) * (x0 - mprev));
    xn = xt;
    x0_out = x0;
  }
  return xn;
}

#ifdef VV_CHAIN_TIMING
// per-segment wall-clock sums of one chosen thread, kept in registers and written once at the end (a global += per stamp
// would wait for the wave's outstanding prefetch and distort what it measures)
__device__ unsigned long long g_chain_t[16];
#define TDECL long long tprev = wall_clock64(); unsigned tacc[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}
#define TSTAMP(i) do { const long long t_ = wall_clock64(); tacc[i] += (unsigned)(t_ - tprev); tprev = t_; } while (0)
#define TFLUSH() do { if (blockIdx.x == TB && threadIdx.x == TT) { for (int i_ = 0; i_ < 12; ++i_) g_chain_t[i_] += tacc[i_]; } } while (0)
#else
#define TDECL do { } while (0)
#define TSTAMP(i) do { } while (0)
#define TFLUSH() do { } while (0)
#endif

__global__ __launch_bounds__(CT) void head_chain_kernel(const HeadChain carg) {
  extern __shared__ __align__(16) float lds[];
  HeadChain& c = *reinterpret_cast<HeadChain*>(lds);              // the descriptor lives in LDS: phase setup never waits on the kernarg segment
  constexpr int DESC_FLOATS = (sizeof(HeadChain) + 15) / 16 * 4;
  unsigned* done = reinterpret_cast<unsigned*>(lds + DESC_FLOATS);  // streaming waves whose flagged stores have landed (monotonic)
  float* xs0 = lds + DESC_FLOATS + 4;                              // activation rows [2 buffers][2 rows][kmax_pad]: phase q uses buffer q & 1
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wg = blockIdx.x;
  {
    const unsigned* src = reinterpret_cast<const unsigned*>(&carg);
    for (int i = tid; i < (int)(sizeof(HeadChain) / 4); i += CT) reinterpret_cast<unsigned*>(lds)[i] = src[i];
    if (tid == 0) *done = 0u;
  }
  __syncthreads();
#ifdef VV_CHAIN_TIMING
  const int TB = uni(c.dbg_block), TT = uni(c.dbg_thread);
#endif
  TDECL;
  const int per = 2 * uni(c.layers) + 2;
  const int n_ph = uni(c.n_steps) * per + 1;
  const int kmax = max(uni(c.ffn), uni(c.D));
  const int xs_stride = 2 * ((kmax + 511) & ~511);
  unsigned* bar = unip(c.bar);
  Ph p;
  head_phase(c, 0, p);
  if (wave < SW) {
    // ======================================= streaming waves =======================================
    u32x4 b[NT];
    LEpi le;
    prefetch(p, wg, wave, lane, b, le);
    for (int ph = 0;;) {
      TSTAMP(0);                                                   // describe + issue of this phase's prefetch
      __syncthreads();                                             // the helpers have published this phase's rows in LDS
      TSTAMP(1);
      const float* xs = xs0 + (ph & 1) * xs_stride;
      if (p.longk) stream_l(p, wg, wave, lane, xs, b, le);
      else stream_r(p, wg, wave, lane, xs, b);
      TSTAMP(2);
      const bool flagged = p.out_flag != 0;
      if (++ph == n_ph) break;
      if (flagged) {                                               // outputs have landed before the workgroup's flag is published
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (lane == 0) atomicAdd(done, 1u);
      }
      TSTAMP(3);
      head_phase(c, ph, p);
      prefetch(p, wg, wave, lane, b, le);                          // the whole next phase goes in flight before the hand-off
    }
  } else {
    // ========================================= helper waves ========================================
    const int m = wave - SW;
    HelperRegs h;
    unsigned publish = 0, n_flagged = 0;
    for (int ph = 0;;) {
      float* xs = xs0 + (ph & 1) * xs_stride;
      // part 1: the previous phase's tagged outputs -> registers (this is the wait for the producers)
      float xn = 0.f, x0 = 0.f;
      bool own = false;
      const int dbg_mode = uni(c.dbg_mode);
      if (dbg_mode & 1) { for (int u = 0; u < KU; ++u) for (int j = 0; j < 8; ++j) h.xv[u][j] = 0.f; }
      else if (p.kind == 1) { if (m == 0) xn = dpm_fetch(c, p.step, lane, own, x0); }
      else helper_fetch(p, m, lane, bar, h);
      TSTAMP(4);
      if (publish && !(dbg_mode & 1)) {                            // the previous phase hands off through the flag barrier
        ++n_flagged;
        if (m == 0) {
          const long long t0 = wall_clock64();
          while (__hip_atomic_load(done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < n_flagged * SW) {
            __builtin_amdgcn_s_sleep(1);
            if (wall_clock64() - t0 > POLL_LIMIT) { stu_co(bar + BAR_ERR, 1u); break; }
          }
          if (lane == 0) stu_co(bar + wg, publish);
        }
      }
      TSTAMP(5);
      // part 2: LDS rows
      if (p.kind == 1) {
        if (m == 0) {
          const int n_steps = uni(c.n_steps);
          if (p.step == n_steps && ldu_co(bar + BAR_ERR) != 0u) xn = __int_as_float(0x7fc00000);   // a poll gave up: poison the result
          if (wg == 0 && own) {
            if (p.step > 0) st_co(unip(c.mb[p.step & 1]) + lane, x0);
            if (p.step == n_steps) *gp(unip(c.latent_out) + lane) = xn;
            else st_co(unip(c.xb[p.step & 1]) + lane, xn);
          }
          for (int o = lane; o < p.kpad; o += 64) { const float val = (o < p.k) ? __shfl(xn, o & 63) : 0.f; xs[o] = val; xs[p.kpad + o] = val; }
        }
      } else if (!(dbg_mode & 1)) {
        helper_publish(p, m, lane, bar, h, xs);
      }
      TSTAMP(6);
      publish = p.out_flag;
      __syncthreads();                                             // rows of phase ph are in LDS
      TSTAMP(7);
      if (++ph == n_ph) break;
      head_phase(c, ph, p);
    }
  }
  TFLUSH();
}

int g_chain_blocks = 0;        // 0: one workgroup per CU
int g_head_chain = 0;         // process-wide default (vv_tune "head_chain"); per call: vv_head.flags & VV_HEAD_CHAIN
int g_num_cu = 0;
int g_dbg_block = 0, g_dbg_thread = 0, g_dbg_mode = 0;

size_t head_chain_lds(int kmax) {
  const size_t desc = (sizeof(HeadChain) + 15) / 16 * 16;
  return desc + 16 + 2 * 2 * (size_t)((kmax + 511) & ~511) * sizeof(float);     // descriptor, counter, double-buffered rows
}

}  // namespace

#ifdef VV_CHAIN_TIMING
extern "C" int vv_chain_debug_times(unsigned long long* out8, int reset) {
  if (hipMemcpyFromSymbol(out8, HIP_SYMBOL(g_chain_t), 16 * sizeof(unsigned long long)) != hipSuccess) return -1;
  if (reset) { unsigned long long z[16] = {0}; if (hipMemcpyToSymbol(HIP_SYMBOL(g_chain_t), z, sizeof(z)) != hipSuccess) return -1; }
  return 0;
}
#endif
void vv_chain_set_dbg(int block, int thread) { g_dbg_block = block; g_dbg_thread = thread; }
void vv_chain_set_blocks(int b) { g_chain_blocks = b; }
void vv_chain_set_dbg_mode(int m) { g_dbg_mode = m; }
void vv_chain_set_head(int on) { g_head_chain = on; }

int vv_chain_init() {
  int dev = 0;
  hipDeviceProp_t prop;
  if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return vv_set_error(VV_E_HIP, "vv_chain_init: no device");
  g_num_cu = prop.multiProcessorCount;
  if (hipFuncSetAttribute(reinterpret_cast<const void*>(head_chain_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                          (int)head_chain_lds(512 * NT)) != hipSuccess)
    return vv_set_error(VV_E_HIP, "vv_chain_init: cannot raise the LDS limit");
  return 0;
}

// floats of workspace the chained kernel needs: flag words + tagged residual stream + tagged head output
size_t vv_head_chain_ws_floats(const vv_head* h) { return BAR_WORDS + 2 * (2 * (size_t)h->D + 2 * (size_t)h->latent) + 64; }

// 1 = the whole solver loop was enqueued as one persistent kernel, 0 = shape not covered (caller uses the per-GEMV path)
int vv_launch_head_chain(const vv_head* h, float* const* mod, const float* modf, const float* noise, const vv_dpm_coef* coef, int n_steps,
                         float cfg_scale, float* latent_out, float* act, float* const* xb, float* const* mb, float* chain_ws, hipStream_t s) {
  if (!(g_head_chain || (h->flags & VV_HEAD_CHAIN)) || g_num_cu <= 0) return 0;
  if (h->wdt != VV_BF16 || h->layers > MAX_LAYERS || h->layers < 1 || n_steps > MAX_STEPS) return 0;
  if (h->D % 8 || h->ffn % 8 || h->latent % 8 || h->latent > 64 || h->D > 512 * KU || h->ffn > 512 * NT || h->ffn <= 512 * KU) return 0;
  HeadChain c;
  memset(&c, 0, sizeof(c));
  c.D = h->D; c.ffn = h->ffn; c.layers = h->layers; c.latent = h->latent; c.n_steps = n_steps;
  c.eps = h->eps; c.cfg = cfg_scale;
  c.noisy_proj = (const bf16_t*)h->noisy_proj; c.final_linear = (const bf16_t*)h->final_linear;
  auto a16 = [](const void* q) { return ((uintptr_t)q % 16) == 0; };
  if (!a16(c.noisy_proj) || !a16(c.final_linear) || !a16(modf) || !a16(act) || !a16(chain_ws)) return 0;
  for (int l = 0; l < h->layers; ++l) {
    const vv_head_layer& L = h->layer[l];
    c.layer[l] = {L.norm_w, (const bf16_t*)L.wgate, (const bf16_t*)L.wup, (const bf16_t*)L.wdown, mod[l]};
    if (!a16(L.norm_w) || !a16(L.wgate) || !a16(L.wup) || !a16(L.wdown) || !a16(mod[l])) return 0;
  }
  c.modf = modf; c.noise = noise; c.latent_out = latent_out; c.act = act;
  c.bar = reinterpret_cast<unsigned*>(chain_ws);
  c.hcur = reinterpret_cast<u64*>(chain_ws + BAR_WORDS);
  c.v = c.hcur + 2 * (size_t)h->D;
  c.xb[0] = xb[0]; c.xb[1] = xb[1]; c.mb[0] = mb[0]; c.mb[1] = mb[1];
  c.dbg_block = g_dbg_block; c.dbg_thread = g_dbg_thread; c.dbg_mode = g_dbg_mode;
  for (int i = 0; i < n_steps; ++i) c.coef[i] = coef[i];
  int blocks = g_chain_blocks > 0 ? g_chain_blocks : g_num_cu;
  if (blocks > g_num_cu) blocks = g_num_cu;             // co-residency: never more than one workgroup per CU
  if (blocks > 256) blocks = 256;                       // flag words polled by one wave
  // flags and tags start from 0 on every launch (a replayed graph sees the previous launch's final epochs otherwise)
  if (hipMemsetAsync(chain_ws, 0, vv_head_chain_ws_floats(h) * sizeof(float), s) != hipSuccess) return vv_set_error(VV_E_HIP, "vv_head_sample: memset failed");
  const int kmax = h->ffn > h->D ? h->ffn : h->D;
  hipLaunchKernelGGL(head_chain_kernel, dim3(blocks), dim3(CT), head_chain_lds(kmax), s, c);
  if (hipGetLastError() != hipSuccess) return vv_set_error(VV_E_HIP, "vv_head_sample: chain launch failed");
  return 1;
}
