// vv_gemv_stream.hip — the batch-1/2 weight-streaming GEMV of the per-frame path (bf16 weights, M <= 4 rows).
//
// Roofline: HBM.  Every weight byte is read once per call with non-temporal 16-byte loads straight into VGPRs
// (no LDS round trip: nothing is shared between waves), the activation slice each lane needs lives in registers
// for the whole kernel (prologue = RMSNorm / adaLN modulate / SiLU fused, computed once per wave), and the
// workgroups are persistent: each wave walks its row groups with the next group's loads already in flight while
// it reduces the current one (register double buffering), so the memory pipe never drains between rows.
// Row sums use DPP row reductions (vv_wave_sum), every output's epilogue (bias / GELU / SwiGLU / gate / residual) runs in its own
// lane on operands fetched one row group ahead, M <= 2 has an fp8 (e4m3fn codes + row scale) instantiation, M = 8 covers the
// T = 8 conv stage when K splits to <= 2 units per wave.
//   KSPLIT == 1        a wave owns RW whole weight rows per step (block = 4 independent waves)        K <= 2560
//   KSPLIT == 4/8/16   the block's 4/8/16 waves split K (interleaved 512-element units) and combine through LDS
//                      in a fixed order (long K: 1.5B down-projections, every 7B matrix)
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "vv_hip.h"
#include "vv_common.h"

namespace {

typedef unsigned short bf16_t;
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float wsum(float v) { return vv_wave_sum(v); }   // DPP row reduction, all 64 lanes active
__device__ __forceinline__ float silu1(float v) { return v / (1.0f + expf(-v)); }
__device__ __forceinline__ float gelu1(float v) { return 0.5f * v * (1.0f + erff(v * 0.70710678118654752440f)); }

__device__ __forceinline__ void unpack8(const u32x4 v, float (&o)[8]) {
  o[0] = __uint_as_float(v.x << 16); o[1] = __uint_as_float(v.x & 0xffff0000u);
  o[2] = __uint_as_float(v.y << 16); o[3] = __uint_as_float(v.y & 0xffff0000u);
  o[4] = __uint_as_float(v.z << 16); o[5] = __uint_as_float(v.z & 0xffff0000u);
  o[6] = __uint_as_float(v.w << 16); o[7] = __uint_as_float(v.w & 0xffff0000u);
}

// 8 e4m3fn weights in the low 8 bytes of the tile register (fp8 weight-only mode: 8-byte loads per lane)
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void unpack8_f8(const u32x4 v, float (&o)[8]) {
  const f32x2 a = __builtin_amdgcn_cvt_pk_f32_fp8((int)v.x, false), b = __builtin_amdgcn_cvt_pk_f32_fp8((int)v.x, true);
  const f32x2 c = __builtin_amdgcn_cvt_pk_f32_fp8((int)v.y, false), d = __builtin_amdgcn_cvt_pk_f32_fp8((int)v.y, true);
  o[0] = a.x; o[1] = a.y; o[2] = b.x; o[3] = b.y; o[4] = c.x; o[5] = c.y; o[6] = d.x; o[7] = d.y;
}

// epilogue operands (bias / adaLN gate / residual) of one output: their addresses are known before the dot product is, so
// they are loaded one row group ahead, in front of the weight loads that follow them in the queue (loads return in order: an
// operand load issued at epilogue time would sit behind two row groups of prefetched weights)
typedef float vf2 __attribute__((ext_vector_type(2)));
struct EpiOp { float b, g, r, s, s2; };   // bias, gate, residual, fp8 row scales (raw loads; absent operands are ignored at use)
__device__ __forceinline__ EpiOp epi_load(const vv_lin_args& a, int m, int n) {
  // straight-line and use-free: an absent operand reads x[0] (always a valid address), the values are only looked at in epi_pre.
  // With a select per operand here the compiler waited for each dword right away - draining every load issued before it - and a
  // uniform branch per operand cut the issue sequence into blocks it then reordered.
  EpiOp e;
  const bool f8s = a.wdt == VV_FP8, f8d = f8s && a.w2;
  const float* pb = a.bias ? a.bias + n : a.x;
  const float* pg = a.gate ? a.gate + (a.gate_ld ? (int64_t)m * a.gate_ld + n : (int64_t)n) : a.x;
  const float* pr = a.res ? a.res + (int64_t)m * a.ldres + n : a.x;
  const float* ps = f8s ? a.wscale + n : a.x;
  const float* ps2 = f8d ? a.w2scale + n : a.x;
  e.b = *pb; e.g = *pg; e.r = *pr; e.s = *ps; e.s2 = *ps2;
  return e;
}
__device__ __forceinline__ EpiOp epi_load_cond(const vv_lin_args& a, int m, int n) {   // one uniform branch per operand (owner lanes only)
  EpiOp e;
  e.b = a.bias ? a.bias[n] : 0.f;
  e.g = a.gate ? (a.gate_ld ? a.gate[(int64_t)m * a.gate_ld + n] : a.gate[n]) : 1.f;
  e.r = a.res ? a.res[(int64_t)m * a.ldres + n] : 0.f;
  e.s = (a.wdt == VV_FP8) ? a.wscale[n] : 1.f;
  e.s2 = (a.wdt == VV_FP8 && a.w2) ? a.w2scale[n] : 1.f;
  return e;
}
__device__ __forceinline__ void epi_pre(const vv_lin_args& a, int m, int n, float v, float v2, const EpiOp& e) {
  if (a.wdt == VV_FP8) { v *= e.s; if (a.w2) v2 *= e.s2; }      // fp8 codes carry a row scale
  if (a.bias) v += e.b;
  if (a.act == VV_ACT_GELU) v = gelu1(v);
  else if (a.act == VV_ACT_SWIGLU) v = silu1(v) * v2;
  if (a.gate) v *= e.g;
  if (a.res) v += e.r;
  a.out[(int64_t)m * a.ldo + n] = v;
}

int g_blocks_override = 0;   // tuning hook (vv_tune)
int g_waves_override = 0;    // tuning hook "gemv_waves": waves per block of the whole-row (KSPLIT == 1) kernels, 3 .. 8
int g_opt = 13;              // tuning hook "gemv_opt": bit 0 / 1 = batched prologue for RMSNorm / no prologue, bit 2 = straight-line epilogue-operand loads, bit 3 = block-staged RMSNorm prologue (KSPLIT == 1)
int g_long_cap = 512;        // persistent blocks for K-split launches with K > 6144 (tuning hook)
int g_long_ku = 5;           // K units per wave allowed for rows longer than 12 units (tuning hook: 3 -> 8 waves split K)

template <int M, bool DUAL, int KSPLIT, int KU, int RW, bool F8>
__global__ __launch_bounds__(KSPLIT == 1 ? 512 : 64 * KSPLIT) void gemv_stream_kernel(const vv_lin_args a, const int n_groups, const int opt) {
  constexpr int NW = (KSPLIT == 1) ? 8 : KSPLIT;     // waves per block (KSPLIT == 1: at most; the launcher picks 3 .. 8 so that the row groups divide evenly)
  __shared__ float red[NW * M];
  __shared__ float part[2][NW][RW * M * 2];
  __shared__ __attribute__((aligned(16))) float xs[KSPLIT == 1 ? M * KU * 512 : 4];   // block-staged prologue result (KSPLIT == 1)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int nwv = (KSPLIT == 1) ? (int)(blockDim.x >> 6) : KSPLIT;
  const int K = a.k, N = a.n, mr = a.m;          // mr <= M real rows
  const bf16_t* __restrict__ W = reinterpret_cast<const bf16_t*>(a.w);
  const bf16_t* __restrict__ W2 = reinterpret_cast<const bf16_t*>(a.w2);

  // ---- this lane's k offsets and activation fragment (kept in registers for the whole kernel) -------------------
  int koff[KU];
  bool kval[KU];
#pragma unroll
  for (int u = 0; u < KU; ++u) {
    const int unit = (KSPLIT == 1) ? u : (wave + NW * u);
    koff[u] = unit * 512 + lane * 8;
    kval[u] = koff[u] < K;
    if (!kval[u]) koff[u] = 0;                    // any valid address; the activation there is forced to 0
  }
  const bool reused = (a.flags & VV_LIN_W_REUSED) != 0 || (opt & 16);    // opt bit 4 (tuning): every matrix with cacheable loads
  constexpr bool f8 = F8;                        // weight-only fp8 is its own instantiation: the bf16 kernels carry none of it
  // the first row group's weight loads are issued before the activation prologue so both latencies overlap
  const int gstride = (KSPLIT == 1) ? gridDim.x * nwv : gridDim.x;
  int g = (KSPLIT == 1) ? blockIdx.x * nwv + wave : blockIdx.x;
  u32x4 cur[RW][KU], cur2[DUAL ? RW : 1][KU];
  u32x4 nxt[RW][KU], nxt2[DUAL ? RW : 1][KU];
  auto issue = [&](u32x4 (&b)[RW][KU], u32x4 (&b2)[DUAL ? RW : 1][KU], int grp) {
    // a group past the end (the unconditional second issue of a wave that owns a single group) degenerates to one 16-byte
    // line per instruction: every lane reads element 0 of the matrix
    const bool live = grp < n_groups;
#pragma unroll
    for (int r = 0; r < RW; ++r) {
      const int n = min(grp * RW + r, N - 1);
#pragma unroll
      for (int u = 0; u < KU; ++u) {
        const int64_t off = live ? (int64_t)n * K + koff[u] : 0;
        if (f8) {                // e4m3fn bytes: this lane's 8 weights are 8 bytes
          const u32x2 t = __builtin_nontemporal_load(reinterpret_cast<const u32x2*>(reinterpret_cast<const unsigned char*>(a.w) + off));
          b[r][u].x = t.x; b[r][u].y = t.y;
          if (DUAL) {
            const u32x2 t2 = __builtin_nontemporal_load(reinterpret_cast<const u32x2*>(reinterpret_cast<const unsigned char*>(a.w2) + off));
            b2[r][u].x = t2.x; b2[r][u].y = t2.y;
          }
          continue;
        }
        const u32x4* p1 = reinterpret_cast<const u32x4*>(W + off);
        const u32x4* p2 = reinterpret_cast<const u32x4*>(W2 + off);
        if (reused) {            // weights re-read by the next solver step: leave them in L2 / Infinity Cache
          b[r][u] = *p1;
          if (DUAL) b2[r][u] = *p2;
        } else {                 // streamed once per frame: non-temporal, do not pollute the caches
          b[r][u] = __builtin_nontemporal_load(p1);
          if (DUAL) b2[r][u] = __builtin_nontemporal_load(p2);
        }
      }
    }
  };
  // epilogue: output (r, m) of a row group belongs to lane / thread  r * M + m  (KSPLIT == 1: lane of the wave that owns the
  // group; K split: thread of the block), so activations like GELU run once per output in parallel lanes instead of RW x M
  // times in lane 0, and each owner fetches its own operands one group ahead
  constexpr int NE = 1;
  const bool has_eo = a.bias || a.gate || a.res || a.wdt == VV_FP8;
  const int eid = (KSPLIT == 1) ? lane : tid;
  EpiOp eo_cur[NE], eo_nxt[NE];
  const int eo_id = eid < RW * M ? eid : 0;       // lanes that own no output fetch output 0's operands: no divergent branch around the loads
  const int eo_r = eo_id / M, eo_m = (eo_id - eo_r * M) < mr ? (eo_id - eo_r * M) : mr - 1;
  auto load_eo = [&](EpiOp (&e)[NE], int grp) {
    if (!has_eo) return;
    if (opt & 4) { e[0] = epi_load(a, eo_m, min((grp < n_groups ? grp : n_groups - 1) * RW + eo_r, N - 1)); return; }
    if (grp < n_groups && eid < RW * M) e[0] = epi_load_cond(a, eo_m, min(grp * RW + eo_r, N - 1));
  };
#pragma unroll
  for (int i = 0; i < NE; ++i) { eo_cur[i].b = 0.f; eo_cur[i].g = 1.f; eo_cur[i].r = 0.f; eo_cur[i].s = 1.f; eo_cur[i].s2 = 1.f; eo_nxt[i] = eo_cur[i]; }
  float xr[M][KU][8];
  // Activation side.  BATCHED (the per-frame shapes: M * KU <= 8, no SiLU prologue): every load the prologue needs - x rows, norm
  // weight, adaLN shift / scale - is issued in ONE batch AHEAD of the weight loads.  Loads return in order: these are L2 hits and
  // come back first, so the statistics and the normalisation run while the first two row groups of weights are still in flight.
  // (Issued behind the weights, and unit by unit with a full wait each, the prologue was 6-12 dependent round trips that only
  // started once the first weights had landed: +2 us on every kernel with a prologue, +4 us on the dual one.)
  // the SwiGLU (dual) kernels take the block-staged prologue, the others the per-wave batched one: one fast path per instantiation
  // keeps the register allocation of each kernel at what its own path needs
  constexpr bool CAN_STAGE = KSPLIT == 1 && (DUAL || M == 8);    // M = 8 (the T = 8 conv stage): 32 KB of activations per WAVE otherwise
  constexpr bool BATCHED = (M * KU <= 8) && !CAN_STAGE;
  // adaLN-modulated rows on a non-dual kernel (the head's 64-row final linear) keep the legacy path: the batch would pin 96 more registers
  const bool batched = BATCHED && ((a.pro == VV_PRO_RMSNORM && !a.mod_scale && (opt & 1)) || (a.pro == VV_PRO_NONE && (opt & 2)));
  bool staged = false;
  if constexpr (CAN_STAGE) staged = a.pro == VV_PRO_RMSNORM && (opt & 8);
  if (CAN_STAGE && staged) {
    // Prologue ONCE PER BLOCK (KSPLIT == 1, RMSNorm): thread t owns the 4-element chunks t, t + T, ... of every row; x, the norm weight
    // and the adaLN shift / scale chunks are requested first, then the first two row groups of weights; statistics through LDS, the
    // normalised rows land in LDS and every lane picks up its k slices from there.  With the prologue per WAVE every one of ~2000
    // waves pulled x, norm weight and modulation (42 KB for the head's SwiGLU GEMV: 86 MB against 28 MB of weights) through L2 at
    // the moment the weight stream saturates the fabric: the median wave saw its activations 4 us after entry (tools/gemv_lab.cpp).
    const int T = nwv * 64;
    constexpr int NCH = (KU * 128 + 191) / 192;     // 4-element chunks per thread per row (blocks have >= 3 waves)
    const bool has_nw = a.norm_w != nullptr, has_mod = a.mod_scale != nullptr;
    float4 xv[M][NCH], nv[NCH], sv[M][NCH], cv[M][NCH];
    const int nchunks = K >> 2;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      const int ch = tid + c * T;
      const int kk = ch < nchunks ? ch * 4 : 0;
#pragma unroll
      for (int m = 0; m < M; ++m) xv[m][c] = *reinterpret_cast<const float4*>(a.x + (int64_t)(m < mr ? m : mr - 1) * a.ldx + kk);
      if (has_nw) nv[c] = *reinterpret_cast<const float4*>(a.norm_w + kk);
      if (has_mod) {
#pragma unroll
        for (int m = 0; m < M; ++m) {
          const int64_t mo = (int64_t)(m < mr ? m : mr - 1) * a.ld_mod + kk;
          sv[m][c] = *reinterpret_cast<const float4*>(a.mod_shift + mo);
          cv[m][c] = *reinterpret_cast<const float4*>(a.mod_scale + mo);
        }
      }
    }
    load_eo(eo_cur, g);
    issue(cur, cur2, g);
    load_eo(eo_nxt, g + gstride);
    issue(nxt, nxt2, g + gstride);
    __builtin_amdgcn_sched_barrier(0);
#define VV_FENCE4(v) asm volatile("" : "+v"((v).x), "+v"((v).y), "+v"((v).z), "+v"((v).w))
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
#pragma unroll
      for (int m = 0; m < M; ++m) VV_FENCE4(xv[m][c]);
      if (has_nw) VV_FENCE4(nv[c]);
      if (has_mod) {
#pragma unroll
        for (int m = 0; m < M; ++m) { VV_FENCE4(sv[m][c]); VV_FENCE4(cv[m][c]); }
      }
    }
#undef VV_FENCE4
    float ss[M];
#pragma unroll
    for (int m = 0; m < M; ++m) {
      float s1 = 0.f;
#pragma unroll
      for (int c = 0; c < NCH; ++c) {
        const float4 v = xv[m][c];
        s1 += (tid + c * T < nchunks) ? (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w) : 0.f;
      }
      ss[m] = wsum(s1);
    }
    if (lane == 0) {
#pragma unroll
      for (int m = 0; m < M; ++m) red[wave * M + m] = ss[m];
    }
    __syncthreads();
#pragma unroll
    for (int m = 0; m < M; ++m) {
      float tot = 0.f;
      for (int w4 = 0; w4 < nwv; ++w4) tot += red[w4 * M + m];      // fixed order: deterministic
      const float rstd = rsqrtf(tot / (float)K + a.eps);
#pragma unroll
      for (int c = 0; c < NCH; ++c) {
        const int ch = tid + c * T;
        float4 v = xv[m][c];
        v.x *= rstd; v.y *= rstd; v.z *= rstd; v.w *= rstd;
        if (has_nw) { v.x *= nv[c].x; v.y *= nv[c].y; v.z *= nv[c].z; v.w *= nv[c].w; }
        if (has_mod) {
          v.x = v.x * (1.0f + cv[m][c].x) + sv[m][c].x; v.y = v.y * (1.0f + cv[m][c].y) + sv[m][c].y;
          v.z = v.z * (1.0f + cv[m][c].z) + sv[m][c].z; v.w = v.w * (1.0f + cv[m][c].w) + sv[m][c].w;
        }
        if (ch < KU * 128) *reinterpret_cast<float4*>(&xs[(m * KU * 128 + ch) * 4]) = ch < nchunks ? v : make_float4(0.f, 0.f, 0.f, 0.f);
      }
    }
    __syncthreads();
#pragma unroll
    for (int m = 0; m < M; ++m)
#pragma unroll
      for (int u = 0; u < KU; ++u) {
        const float4 p = *reinterpret_cast<const float4*>(&xs[m * KU * 512 + u * 512 + lane * 8]);
        const float4 q = *reinterpret_cast<const float4*>(&xs[m * KU * 512 + u * 512 + lane * 8 + 4]);
        xr[m][u][0] = p.x; xr[m][u][1] = p.y; xr[m][u][2] = p.z; xr[m][u][3] = p.w;
        xr[m][u][4] = q.x; xr[m][u][5] = q.y; xr[m][u][6] = q.z; xr[m][u][7] = q.w;
      }
  } else if (BATCHED && batched) {
    float4 xa[M][KU], xb[M][KU], na[KU], nb[KU];
#pragma unroll
    for (int m = 0; m < M; ++m) {
      const float* xrow = a.x + (int64_t)(m < mr ? m : mr - 1) * a.ldx;
#pragma unroll
      for (int u = 0; u < KU; ++u) {
        xa[m][u] = *reinterpret_cast<const float4*>(xrow + koff[u]);
        xb[m][u] = *reinterpret_cast<const float4*>(xrow + koff[u] + 4);
      }
    }
    const bool rms = a.pro == VV_PRO_RMSNORM;
    const bool has_nw = rms && a.norm_w;
    if (has_nw) {
#pragma unroll
      for (int u = 0; u < KU; ++u) {
        na[u] = *reinterpret_cast<const float4*>(a.norm_w + koff[u]);
        nb[u] = *reinterpret_cast<const float4*>(a.norm_w + koff[u] + 4);
      }
    }
    // the first two row groups go out unconditionally (a group past the end costs one 16-byte line per load instruction) so that
    // no branch separates them from the loads above, and nothing below is scheduled ahead of them
    load_eo(eo_cur, g);
    issue(cur, cur2, g);
    load_eo(eo_nxt, g + gstride);
    issue(nxt, nxt2, g + gstride);
    __builtin_amdgcn_sched_barrier(0);
    // everything the prologue computes with is made opaque HERE, behind the weight loads: hipcc otherwise hoists pieces of the
    // arithmetic (and the waits they need) into the blocks above, in front of the weight issue
#define VV_FENCE4(v) asm volatile("" : "+v"((v).x), "+v"((v).y), "+v"((v).z), "+v"((v).w))
#pragma unroll
    for (int u = 0; u < KU; ++u) {
#pragma unroll
      for (int m = 0; m < M; ++m) { VV_FENCE4(xa[m][u]); VV_FENCE4(xb[m][u]); }
      if (has_nw) { VV_FENCE4(na[u]); VV_FENCE4(nb[u]); }
    }
#undef VV_FENCE4
    float ss[M];
#pragma unroll
    for (int m = 0; m < M; ++m) {
      ss[m] = 0.f;
#pragma unroll
      for (int u = 0; u < KU; ++u) {
        const bool kv = kval[u];                         // lanes past K read a valid address and contribute zeros
        xr[m][u][0] = kv ? xa[m][u].x : 0.f; xr[m][u][1] = kv ? xa[m][u].y : 0.f; xr[m][u][2] = kv ? xa[m][u].z : 0.f; xr[m][u][3] = kv ? xa[m][u].w : 0.f;
        xr[m][u][4] = kv ? xb[m][u].x : 0.f; xr[m][u][5] = kv ? xb[m][u].y : 0.f; xr[m][u][6] = kv ? xb[m][u].z : 0.f; xr[m][u][7] = kv ? xb[m][u].w : 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) ss[m] = fmaf(xr[m][u][j], xr[m][u][j], ss[m]);
      }
    }
    if (rms) {
#pragma unroll
      for (int m = 0; m < M; ++m) ss[m] = wsum(ss[m]);   // KSPLIT == 1: every wave holds the whole row
      if (KSPLIT != 1) {
        if (lane == 0) {
#pragma unroll
          for (int m = 0; m < M; ++m) red[wave * M + m] = ss[m];
        }
        __syncthreads();
#pragma unroll
        for (int m = 0; m < M; ++m) {
          ss[m] = 0.f;
#pragma unroll
          for (int w4 = 0; w4 < nwv; ++w4) ss[m] += red[w4 * M + m];
        }
      }
#pragma unroll
      for (int m = 0; m < M; ++m) {
        const float rstd = rsqrtf(ss[m] / (float)K + a.eps);
#pragma unroll
        for (int u = 0; u < KU; ++u) {
          const float nw[8] = {na[u].x, na[u].y, na[u].z, na[u].w, nb[u].x, nb[u].y, nb[u].z, nb[u].w};
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            float v = xr[m][u][j] * rstd;
            if (has_nw) v *= nw[j];
            xr[m][u][j] = kval[u] ? v : 0.f;
          }
        }
      }
    }
  } else {
  if (g < n_groups) { load_eo(eo_cur, g); issue(cur, cur2, g); }
  if (g + gstride < n_groups) { load_eo(eo_nxt, g + gstride); issue(nxt, nxt2, g + gstride); }   // two row groups in flight before the prologue even starts
#pragma unroll
  for (int m = 0; m < M; ++m) {
    const float* xrow = a.x + (int64_t)(m < mr ? m : mr - 1) * a.ldx;
    float ss = 0.f;
#pragma unroll
    for (int u = 0; u < KU; ++u) {
      if (kval[u]) {
        const float4 p = *reinterpret_cast<const float4*>(xrow + koff[u]);
        const float4 q = *reinterpret_cast<const float4*>(xrow + koff[u] + 4);
        xr[m][u][0] = p.x; xr[m][u][1] = p.y; xr[m][u][2] = p.z; xr[m][u][3] = p.w;
        xr[m][u][4] = q.x; xr[m][u][5] = q.y; xr[m][u][6] = q.z; xr[m][u][7] = q.w;
      } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) xr[m][u][j] = 0.f;
      }
      if (a.pro == VV_PRO_SILU) {
#pragma unroll
        for (int j = 0; j < 8; ++j) xr[m][u][j] = silu1(xr[m][u][j]);
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) ss = fmaf(xr[m][u][j], xr[m][u][j], ss);
    }
    if (a.pro == VV_PRO_RMSNORM) {
      ss = wsum(ss);                               // KSPLIT == 1: every wave holds the whole row
      if (KSPLIT != 1) {
        if (lane == 0) red[wave * M + m] = ss;
        __syncthreads();
        ss = 0.f;
#pragma unroll
        for (int w4 = 0; w4 < nwv; ++w4) ss += red[w4 * M + m];
      }
      const float rstd = rsqrtf(ss / (float)K + a.eps);
#pragma unroll
      for (int u = 0; u < KU; ++u) {
        if (!kval[u]) continue;
        float nw[8], sh[8], sc[8];
        if (a.norm_w) {
          const float4 p = *reinterpret_cast<const float4*>(a.norm_w + koff[u]);
          const float4 q = *reinterpret_cast<const float4*>(a.norm_w + koff[u] + 4);
          nw[0] = p.x; nw[1] = p.y; nw[2] = p.z; nw[3] = p.w; nw[4] = q.x; nw[5] = q.y; nw[6] = q.z; nw[7] = q.w;
        }
        if (a.mod_scale) {
          const int64_t mo = (int64_t)(m < mr ? m : mr - 1) * a.ld_mod + koff[u];
          const float4 s0 = *reinterpret_cast<const float4*>(a.mod_shift + mo), s1 = *reinterpret_cast<const float4*>(a.mod_shift + mo + 4);
          const float4 c0 = *reinterpret_cast<const float4*>(a.mod_scale + mo), c1 = *reinterpret_cast<const float4*>(a.mod_scale + mo + 4);
          sh[0] = s0.x; sh[1] = s0.y; sh[2] = s0.z; sh[3] = s0.w; sh[4] = s1.x; sh[5] = s1.y; sh[6] = s1.z; sh[7] = s1.w;
          sc[0] = c0.x; sc[1] = c0.y; sc[2] = c0.z; sc[3] = c0.w; sc[4] = c1.x; sc[5] = c1.y; sc[6] = c1.z; sc[7] = c1.w;
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          float v = xr[m][u][j] * rstd;
          if (a.norm_w) v *= nw[j];
          if (a.mod_scale) v = v * (1.0f + sc[j]) + sh[j];
          xr[m][u][j] = v;
        }
      }
    }
  }

  }

  // ---- stream the weight rows ------------------------------------------------------------------------------------
  int parity = 0;
  while (g < n_groups) {
    const int gn = g + gstride;
    // The dot products run on packed fp32 FMAs (v_pk_fma_f32: two lanes of a register pair per instruction): even and odd k of a lane
    // accumulate separately and are added once per row group.  This loop, not the memory system, sets the kernel's pace (its time grows
    // 1.2-1.6 us per activation row, tools/mb_rows.py); half the FMA instructions per weight load is the cheapest cut.
    vf2 pacc[RW][M], pacc2[DUAL ? RW : 1][M];
#pragma unroll
    for (int r = 0; r < RW; ++r)
#pragma unroll
      for (int m = 0; m < M; ++m) { pacc[r][m] = vf2{0.f, 0.f}; if (DUAL) pacc2[r][m] = vf2{0.f, 0.f}; }
#pragma unroll
    for (int u = 0; u < KU; ++u) {
#pragma unroll
      for (int r = 0; r < RW; ++r) {
        float w[8], w2[8];
        if (f8) { unpack8_f8(cur[r][u], w); if (DUAL) unpack8_f8(cur2[r][u], w2); }
        else { unpack8(cur[r][u], w); if (DUAL) unpack8(cur2[r][u], w2); }
#pragma unroll
        for (int m = 0; m < M; ++m) {
#pragma unroll
          for (int j = 0; j < 8; j += 2) {
            const vf2 xp = {xr[m][u][j], xr[m][u][j + 1]};
            pacc[r][m] = __builtin_elementwise_fma(vf2{w[j], w[j + 1]}, xp, pacc[r][m]);
            if (DUAL) pacc2[r][m] = __builtin_elementwise_fma(vf2{w2[j], w2[j + 1]}, xp, pacc2[r][m]);
          }
        }
      }
    }
    float acc[RW][M], acc2[DUAL ? RW : 1][M];
#pragma unroll
    for (int r = 0; r < RW; ++r)
#pragma unroll
      for (int m = 0; m < M; ++m) {
        acc[r][m] = wsum(pacc[r][m].x + pacc[r][m].y);
        if (DUAL) acc2[r][m] = wsum(pacc2[r][m].x + pacc2[r][m].y);
      }
    if (KSPLIT == 1) {
      if (lane < RW * M) {                         // every lane holds all sums: lane r * M + m keeps (r, m)
        float v = acc[0][0], v2 = DUAL ? acc2[0][0] : 0.f;
#pragma unroll
        for (int r = 0; r < RW; ++r)
#pragma unroll
          for (int m = 0; m < M; ++m)
            if (lane == r * M + m) { v = acc[r][m]; if (DUAL) v2 = acc2[r][m]; }
        const int r = lane / M, m = lane - r * M;
        const int n = g * RW + r;
        if (n < N && m < mr) epi_pre(a, m, n, v, v2, eo_cur[0]);
      }
    } else {
      if (lane == 0) {
#pragma unroll
        for (int r = 0; r < RW; ++r)
#pragma unroll
          for (int m = 0; m < M; ++m) {
            part[parity][wave][(r * M + m) * 2] = acc[r][m];
            part[parity][wave][(r * M + m) * 2 + 1] = DUAL ? acc2[r][m] : 0.f;
          }
      }
      __syncthreads();                             // g is block-uniform when KSPLIT != 1
      if (tid < RW * M) {
        const int r = tid / M, m = tid - r * M;
        const int n = g * RW + r;
        if (n < N && m < mr) {
          float s = 0.f, s2 = 0.f;
#pragma unroll
          for (int w4 = 0; w4 < NW; ++w4) { s += part[parity][w4][tid * 2]; s2 += part[parity][w4][tid * 2 + 1]; }
          epi_pre(a, m, n, s, s2, eo_cur[0]);
        }
      }
      parity ^= 1;                                 // ping-pong: one barrier per group is enough
    }
#pragma unroll
    for (int i = 0; i < NE; ++i) eo_cur[i] = eo_nxt[i];
#pragma unroll
    for (int r = 0; r < RW; ++r)
#pragma unroll
      for (int u = 0; u < KU; ++u) { cur[r][u] = nxt[r][u]; if (DUAL) cur2[r][u] = nxt2[r][u]; }
    g = gn;
    if (g + gstride < n_groups) { load_eo(eo_nxt, g + gstride); issue(nxt, nxt2, g + gstride); }   // keep two groups in flight
  }
}

int g_dual_rw = 1;            // weight rows per wave step for the dual (SwiGLU) kernel: 1 keeps 4 waves/SIMD resident

template <int M, bool DUAL, int KSPLIT, int KU, int RW>
void launch_rw(const vv_lin_args& a, hipStream_t s) {
  // persistent grid, sized from measurements on MI355X (tools/mb_gemv.py): ~1.5-2 blocks per CU is the sweet spot for the
  // wave-per-row layout (more blocks only add prologue copies and a ragged last round), one block per row group when the
  // block's waves split K
  const int n_groups = (a.n + RW - 1) / RW;
  const int waves = (KSPLIT == 1 && g_waves_override >= 3 && g_waves_override <= 8) ? g_waves_override : 4;
  const int work = (KSPLIT == 1) ? (n_groups + waves - 1) / waves : n_groups;       // blocks if each wave did exactly one group
  // K split: every block reads all of x (M x K fp32 from L2); beyond ~6K columns that traffic rivals the weights, so long
  // rows use fewer, persistent blocks (n=1536 k=8960: 10.4 us at 768 blocks, 9.1 us at 512)
  const int cap = (KSPLIT == 1) ? (DUAL ? (RW == 1 ? 512 : 448) : 512) : (a.k > 6144 ? g_long_cap : 1024);
  int blocks = work < cap ? work : cap;
  if (g_blocks_override > 0) blocks = g_blocks_override < work ? g_blocks_override : work;
  const int threads = KSPLIT == 1 ? 64 * waves : 64 * KSPLIT;
  if constexpr (M <= 2) {                        // fp8 weights: decode rows only
    if (a.wdt == VV_FP8) {
      hipLaunchKernelGGL((gemv_stream_kernel<M, DUAL, KSPLIT, KU, RW, true>), dim3(blocks), dim3(threads), 0, s, a, n_groups, g_opt);
      return;
    }
  }
  hipLaunchKernelGGL((gemv_stream_kernel<M, DUAL, KSPLIT, KU, RW, false>), dim3(blocks), dim3(threads), 0, s, a, n_groups, g_opt);
}

int g_small_rw = 2;           // rows per wave step for narrow non-dual matrices (tuning hook)

template <int M, bool DUAL, int KSPLIT, int KU>
void launch_one(const vv_lin_args& a, hipStream_t s) {
  if (DUAL && g_dual_rw == 1) launch_rw<M, DUAL, KSPLIT, KU, 1>(a, s);
  else if (!DUAL && KSPLIT == 1 && a.n <= 4096 && g_small_rw == 1) launch_rw<M, DUAL, KSPLIT, KU, 1>(a, s);
  else launch_rw<M, DUAL, KSPLIT, KU, 2>(a, s);
}

template <int M, bool DUAL, int KSPLIT>
bool launch_kus(const vv_lin_args& a, hipStream_t s, int ku) {
  switch (ku) {
    case 1: if constexpr (KSPLIT == 1) { launch_one<M, DUAL, KSPLIT, 1>(a, s); return true; } else return false;
    case 2: launch_one<M, DUAL, KSPLIT, 2>(a, s); return true;
    case 3: launch_one<M, DUAL, KSPLIT, 3>(a, s); return true;
    case 4: if constexpr (!DUAL || KSPLIT == 1) { launch_one<M, DUAL, KSPLIT, 4>(a, s); return true; } else return false;
    case 5: if constexpr (!DUAL || KSPLIT == 1) { launch_one<M, DUAL, KSPLIT, 5>(a, s); return true; } else return false;
  }
  return false;
}

template <int M, bool DUAL>
bool launch_ku(const vv_lin_args& a, hipStream_t s, int ksplit, int ku) {
  switch (ksplit) {
    case 1: return launch_kus<M, DUAL, 1>(a, s, ku);
    case 4: return launch_kus<M, DUAL, 4>(a, s, ku);
    case 8: return launch_kus<M, DUAL, 8>(a, s, ku);
    case 16: if constexpr (!DUAL) return launch_kus<M, false, 16>(a, s, ku); else return false;
  }
  return false;
}

// 5..8 activation rows (the conv tokenizers' T = 8 stage, C = 1024): the fragment is 8 rows x KU x 8 floats, so K is split
// until KU <= 2; one pass over the weights instead of two 4-row passes
bool launch_m8(const vv_lin_args& a, hipStream_t s, int ksplit, int ku) {
  if (ksplit == 1) {
    if (ku == 1) launch_rw<8, false, 1, 1, 1>(a, s); else launch_rw<8, false, 1, 2, 1>(a, s);
    return true;
  }
  switch (ksplit) {
    case 4: launch_rw<8, false, 4, 2, 2>(a, s); return true;
    case 8: launch_rw<8, false, 8, 2, 2>(a, s); return true;
  }
  return false;   // 16 waves x 8 rows does not fit the 128-VGPR budget of a 1024-thread block: the caller splits the rows
}

}  // namespace

void vv_gemv_stream_set_blocks(int b) { g_blocks_override = b; }
void vv_gemv_stream_set_waves(int w) { g_waves_override = w; }
void vv_gemv_stream_set_opt(int o) { g_opt = o; }
void vv_gemv_stream_set_long(int cap, int ku) { if (cap > 0) g_long_cap = cap; if (ku > 0) g_long_ku = ku; }
void vv_gemv_stream_set_dual_rw(int r) { g_dual_rw = r; }
void vv_gemv_stream_set_small_rw(int r) { g_small_rw = r; }

// returns 1 when the call was launched here, 0 when the shape/alignment is not covered (caller falls back)
int vv_launch_gemv_stream(const vv_lin_args& a, hipStream_t s) {
  if ((a.wdt != VV_BF16 && a.wdt != VV_FP8) || a.m > 8 || a.k % 8) return 0;
  if (a.wdt == VV_FP8 && (a.m > 2 || !a.wscale || (a.w2 && !a.w2scale) || (uintptr_t)a.w % 8 || (a.w2 && (uintptr_t)a.w2 % 8))) return 0;
  if (a.m > 4) {
    if (a.w2 || a.wdt != VV_BF16) return 0;
    if ((uintptr_t)a.w % 16 || (uintptr_t)a.x % 16 || a.ldx % 4) return 0;
    if (a.norm_w && (uintptr_t)a.norm_w % 16) return 0;
    if (a.mod_scale && ((uintptr_t)a.mod_scale % 16 || (uintptr_t)a.mod_shift % 16 || a.ld_mod % 4)) return 0;
    const int units8 = (a.k + 511) / 512;
    int ks = 0, ku8 = 0;
    for (int w : {1, 4, 8}) {
      if ((units8 + w - 1) / w <= 2) { ks = w; ku8 = (units8 + w - 1) / w; break; }
    }
    if (!ks) return 0;
    if (ks > 1) ku8 = 2;
    return launch_m8(a, s, ks, ku8) ? 1 : 0;
  }
  if (a.wdt == VV_BF16 && ((uintptr_t)a.w % 16 || (a.w2 && (uintptr_t)a.w2 % 16))) return 0;
  if ((uintptr_t)a.x % 16) return 0;
  if (a.m > 1 && a.ldx % 4) return 0;
  if (a.norm_w && (uintptr_t)a.norm_w % 16) return 0;
  if (a.mod_scale && ((uintptr_t)a.mod_scale % 16 || (uintptr_t)a.mod_shift % 16 || a.ld_mod % 4)) return 0;
  const int units = (a.k + 511) / 512;
  const bool dual = a.w2 != nullptr;
  // smallest wave count whose per-wave slice fits the register-resident activation fragment (KU <= 5 units; <= 3 for the
  // dual kernel when K is split, its registers hold two weight streams)
  int ksplit = 1, ku = units;
  if (units > 5) {
    const int kumax = dual ? 3 : (units > 12 ? g_long_ku : 5);
    ksplit = 0;
    for (int w : {4, 8, 16}) {
      if ((units + w - 1) / w <= kumax) { ksplit = w; ku = (units + w - 1) / w; break; }
    }
    if (!ksplit) return 0;
    if (ku < 2) ku = 2;
  }
  bool ok;
  if (a.m == 1) ok = dual ? launch_ku<1, true>(a, s, ksplit, ku) : launch_ku<1, false>(a, s, ksplit, ku);
  else if (a.m == 2) ok = dual ? launch_ku<2, true>(a, s, ksplit, ku) : launch_ku<2, false>(a, s, ksplit, ku);
  else ok = dual ? launch_ku<4, true>(a, s, ksplit, ku) : launch_ku<4, false>(a, s, ksplit, ku);
  return ok ? 1 : 0;
}
