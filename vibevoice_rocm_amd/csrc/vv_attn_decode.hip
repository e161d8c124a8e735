// vv_attn_decode.hip — decode attention of the per-frame LLM step (bf16 KV cache, head_dim 128): RoPE(q, new k) + KV append + GQA attention
// for the R = 2 rows {positive, negative} (Qwen2 attention under modeling_vibevoice.py:187-199; reference call sites
// modeling_vibevoice_inference.py:478-480,581-583).
//
// Latency-bound (tools/attn_lab.cpp): the previous kernel spent 2.9 us until q / RoPE were in registers (56 scalar loads per lane),
// 5 us in a per-key online softmax with two expf per key on 16 waves sharing one CU, and 4.7 us merging lane groups with ds_bpermute
// chains plus a 16-step serial loop per output.  Here:
//   * q, new k, new v and the RoPE table come in with 14 16-byte loads per lane, issued before lens is even known;
//   * 16 lanes share a key (one DPP row: the dot product is 4 DPP adds), 4 keys per lane group are scored per batch and folded into the
//     running maximum ONCE per batch, in the log2 domain (v_exp_f32; the scale carries log2 e);
//   * all lane groups of the block merge through LDS in one step (every output thread folds the 16-32 partial maxima itself:
//     one barrier, no cross-lane chains);
//   * 512 threads per (row, q head): a CU pulls its 2 x S x 256 B of K/V at ~95 GB/s, which is the floor of this layout at
//     S ~ 500 (2.4 us); long contexts (cfg 4: S up to 4 500) split the keys over gridDim.z blocks whose partial (m, l, acc) meet
//     in the last-arriving block (agent-scope release / acquire around a ticket, cdna_hip_programming.md Guideline 16).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <math.h>
#include <type_traits>

#include "vv_hip.h"
#include "vv_common.h"

namespace {

typedef unsigned short bf16_t;
typedef unsigned int att_raw __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void unpack8(const att_raw v, float (&o)[8]) {
  o[0] = __uint_as_float(v.x << 16); o[1] = __uint_as_float(v.x & 0xffff0000u); o[2] = __uint_as_float(v.y << 16); o[3] = __uint_as_float(v.y & 0xffff0000u);
  o[4] = __uint_as_float(v.z << 16); o[5] = __uint_as_float(v.z & 0xffff0000u); o[6] = __uint_as_float(v.w << 16); o[7] = __uint_as_float(v.w & 0xffff0000u);
}
__device__ __forceinline__ float gsum16(float v) {       // sum over the 16 lanes of a DPP row, result in every lane of the row
#define VV_DPP_ADD(ctrl) v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), ctrl, 0xF, 0xF, true))
  VV_DPP_ADD(0xB1); VV_DPP_ADD(0x4E); VV_DPP_ADD(0x141); VV_DPP_ADD(0x140);
#undef VV_DPP_ADD
  return v;
}
__device__ __forceinline__ float ex2(float x) { return __builtin_amdgcn_exp2f(x); }
__device__ __forceinline__ unsigned bf16_bits(float f) {   // round to nearest even (finite inputs: projections of finite activations)
  const unsigned u = __float_as_uint(f);
  return (u + 0x7fffu + ((u >> 16) & 1u)) >> 16;
}

// phase timing of block (0,0,0) / thread 0, debug builds only (-DVV_CF_TIMING, tools/convffn_phase.py attn)
#ifdef VV_CF_TIMING
__device__ unsigned long long g_at_t[8];
#define ASTAMP(i) do { if (blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && threadIdx.x == 0) { const long long t_ = wall_clock64(); g_at_t[i] += (unsigned long long)(t_ - tprev_); tprev_ = t_; } } while (0)
#else
#define ASTAMP(i) do { } while (0)
#endif
#define ATT_UNR 4
template <int NW>
__global__ __launch_bounds__(NW * 64) void attn_decode_kernel(const float* qkv, int64_t ld, int heads, vv_kv kv, int layer, const float2* rope, const int* lens,
                                                             float* out, int64_t ldo, float* part, int* tickets) {
  constexpr int d = 128, half = 64, NG = NW * 4, EPL = 8;
  __shared__ __attribute__((aligned(16))) float sacc[NG][d];
  __shared__ float sm_[NG], sl_[NG];
  __shared__ int s_last;
#ifdef VV_CF_TIMING
  long long tprev_ = wall_clock64();
#endif
  const int r = blockIdx.y, h = blockIdx.x, split = blockIdx.z, nsplit = gridDim.z;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int gl = lane & 15, gi = lane >> 4, grp = wave * 4 + gi;
  const int e0 = gl * EPL;
  const bool lo = e0 < half;
  const int pe0 = lo ? e0 + half : e0 - half;               // half rotation pairs (i, i + d/2)
  const float* row = qkv + (int64_t)r * ld;
  const int gsz = heads / kv.kv_heads, kvh = h / gsz;
  const float* qp = row + h * d;
  const float* kp = row + (heads + kvh) * d;
  const float* vp = row + (heads + kv.kv_heads + kvh) * d;
  const float4 qa0 = *reinterpret_cast<const float4*>(qp + e0), qa1 = *reinterpret_cast<const float4*>(qp + e0 + 4);
  const float4 qb0 = *reinterpret_cast<const float4*>(qp + pe0), qb1 = *reinterpret_cast<const float4*>(qp + pe0 + 4);
  const float4 ka0 = *reinterpret_cast<const float4*>(kp + e0), ka1 = *reinterpret_cast<const float4*>(kp + e0 + 4);
  const float4 kb0 = *reinterpret_cast<const float4*>(kp + pe0), kb1 = *reinterpret_cast<const float4*>(kp + pe0 + 4);
  const float4 vn0 = *reinterpret_cast<const float4*>(vp + e0), vn1 = *reinterpret_cast<const float4*>(vp + e0 + 4);
  const float2* rp = rope + (int64_t)r * half + (lo ? e0 : pe0);
  const float4 r0 = *reinterpret_cast<const float4*>(rp), r1 = *reinterpret_cast<const float4*>(rp + 2);
  const float4 r2 = *reinterpret_cast<const float4*>(rp + 4), r3 = *reinterpret_cast<const float4*>(rp + 6);
  const int64_t base = ((((int64_t)layer * kv.rows + r) * kv.kv_heads + kvh) * kv.s_max) * d;
  bf16_t* kc = reinterpret_cast<bf16_t*>(kv.k) + base;
  bf16_t* vc = reinterpret_cast<bf16_t*>(kv.v) + base;
  att_raw kraw[2][ATT_UNR], vraw[2][ATT_UNR];
  // Unsplit contexts (the common case): the first batch of keys is requested together with q / k / v / rope, BEFORE the position is
  // known - slots past the position are masked (and their values zeroed: the cache behind the position may hold anything) when the
  // batch is consumed.  Waiting for lens[r] first put one more memory round trip in front of every layer's attention.
  const bool spec = nsplit == 1;
  if (spec) {
#pragma unroll
    for (int u = 0; u < ATT_UNR; ++u) {
      const int sc = min(grp + u * NG, kv.s_max - 1);
      kraw[0][u] = *reinterpret_cast<const att_raw*>(kc + (int64_t)sc * d + e0);
      vraw[0][u] = *reinterpret_cast<const att_raw*>(vc + (int64_t)sc * d + e0);
    }
  }
  const int pos = lens[r];
  const int per = (pos + nsplit - 1) / nsplit;              // this block's keys [ks, ke) of the pos cached ones
  const int ks = split * per, ke = min(pos, ks + per);
  auto issue_kv = [&](int buf, int s0) {
#pragma unroll
    for (int u = 0; u < ATT_UNR; ++u) {
      const int sidx = s0 + u * NG;
      const int sc = sidx < ke ? sidx : 0;
      kraw[buf][u] = *reinterpret_cast<const att_raw*>(kc + (int64_t)sc * d + e0);
      vraw[buf][u] = *reinterpret_cast<const att_raw*>(vc + (int64_t)sc * d + e0);
    }
  };
  const int s_first = ks + grp;
  if (!spec && s_first < ke) issue_kv(0, s_first);
  ASTAMP(0);                                       // requests out, position known
  const float qsc = rsqrtf((float)d) * 1.4426950408889634f;  // scores in the log2 domain
  const float cs[8] = {r0.x, r0.z, r1.x, r1.z, r2.x, r2.z, r3.x, r3.z}, sn[8] = {r0.y, r0.w, r1.y, r1.w, r2.y, r2.w, r3.y, r3.w};
  const float qa[8] = {qa0.x, qa0.y, qa0.z, qa0.w, qa1.x, qa1.y, qa1.z, qa1.w}, qb[8] = {qb0.x, qb0.y, qb0.z, qb0.w, qb1.x, qb1.y, qb1.z, qb1.w};
  const float ka[8] = {ka0.x, ka0.y, ka0.z, ka0.w, ka1.x, ka1.y, ka1.z, ka1.w}, kb[8] = {kb0.x, kb0.y, kb0.z, kb0.w, kb1.x, kb1.y, kb1.z, kb1.w};
  const float vn[8] = {vn0.x, vn0.y, vn0.z, vn0.w, vn1.x, vn1.y, vn1.z, vn1.w};
  float q[EPL], kn[EPL];
#pragma unroll
  for (int j = 0; j < EPL; ++j) {      // q * cos + rotate_half(q) * sin: first half takes -x2, second half +x1
    q[j] = (lo ? qa[j] * cs[j] - qb[j] * sn[j] : qa[j] * cs[j] + qb[j] * sn[j]) * qsc;
    kn[j] = lo ? ka[j] * cs[j] - kb[j] * sn[j] : ka[j] * cs[j] + kb[j] * sn[j];
  }
  ASTAMP(1);                                       // q / k / rope landed, RoPE
  float mmax = -INFINITY, lsum = 0.f, acc[EPL];
#pragma unroll
  for (int j = 0; j < EPL; ++j) acc[j] = 0.f;
  auto consume = [&](int buf, int s0, bool scrub) {
    float dot[ATT_UNR], vx[ATT_UNR][EPL];
    float bm = -INFINITY;
#pragma unroll
    for (int u = 0; u < ATT_UNR; ++u) {
      float kx[EPL];
      unpack8(kraw[buf][u], kx);
      unpack8(vraw[buf][u], vx[u]);
      const bool live = s0 + u * NG < ke;
      if (scrub && !live) {                          // a slot read ahead of the position: whatever bits it holds must not reach 0 * v
#pragma unroll
        for (int j = 0; j < EPL; ++j) vx[u][j] = 0.f;
      }
      float dd = 0.f;
#pragma unroll
      for (int j = 0; j < EPL; ++j) dd = fmaf(q[j], kx[j], dd);
      dd = gsum16(dd);
      dot[u] = live ? dd : -INFINITY;
      bm = fmaxf(bm, dot[u]);
    }
    if (bm == -INFINITY) return;                 // uniform over the lane group: no key in this batch
    const float mn = fmaxf(mmax, bm);
    const float corr = ex2(mmax - mn);           // exp2(-inf) = 0 on the first batch
    float ps = 0.f;
#pragma unroll
    for (int j = 0; j < EPL; ++j) acc[j] *= corr;
#pragma unroll
    for (int u = 0; u < ATT_UNR; ++u) {
      const float p = ex2(dot[u] - mn);          // masked keys: exp2(-inf) = 0
      ps += p;
#pragma unroll
      for (int j = 0; j < EPL; ++j) acc[j] = fmaf(p, vx[u][j], acc[j]);
    }
    lsum = lsum * corr + ps;
    mmax = mn;
  };
  const int stride = NG * ATT_UNR;
  for (int s0 = s_first; s0 < ke; s0 += 2 * stride) {       // two batches in flight, buffers with fixed roles
    if (s0 + stride < ke) issue_kv(1, s0 + stride);
    consume(0, s0, spec && s0 == s_first);
    if (s0 + stride >= ke) break;
    if (s0 + 2 * stride < ke) issue_kv(0, s0 + 2 * stride);
    consume(1, s0 + stride, false);
  }
  ASTAMP(2);                                       // key batches
  if (split == 0 && grp == 0) {
    // the new token itself, straight from the projection (its cache slot may not be written yet by the block that owns it)
    float dd = 0.f;
#pragma unroll
    for (int j = 0; j < EPL; ++j) dd = fmaf(q[j], kn[j], dd);
    dd = gsum16(dd);
    const float mn = fmaxf(mmax, dd), corr = ex2(mmax - mn), p = ex2(dd - mn);
    lsum = lsum * corr + p;
#pragma unroll
    for (int j = 0; j < EPL; ++j) acc[j] = fmaf(p, vn[j], acc[j] * corr);
    mmax = mn;
    if (h % gsz == 0) {                          // one writer per (row, kv head): append k, v at slot pos
      att_raw pk, pv;
      pk.x = bf16_bits(kn[0]) | (bf16_bits(kn[1]) << 16); pk.y = bf16_bits(kn[2]) | (bf16_bits(kn[3]) << 16);
      pk.z = bf16_bits(kn[4]) | (bf16_bits(kn[5]) << 16); pk.w = bf16_bits(kn[6]) | (bf16_bits(kn[7]) << 16);
      pv.x = bf16_bits(vn[0]) | (bf16_bits(vn[1]) << 16); pv.y = bf16_bits(vn[2]) | (bf16_bits(vn[3]) << 16);
      pv.z = bf16_bits(vn[4]) | (bf16_bits(vn[5]) << 16); pv.w = bf16_bits(vn[6]) | (bf16_bits(vn[7]) << 16);
      *reinterpret_cast<att_raw*>(kc + (int64_t)pos * d + e0) = pk;
      *reinterpret_cast<att_raw*>(vc + (int64_t)pos * d + e0) = pv;
      if (kv.vt) {                               // the transposed copy the matrix-core prompt attention reads stays in step with v
        bf16_t* vtc = reinterpret_cast<bf16_t*>(kv.vt) + base + (int64_t)(pos >> 5) * (32 * d) + (pos & 31);     // 32-key tiles [tile][d][32]
#pragma unroll
        for (int j = 0; j < EPL; ++j) vtc[(e0 + j) * 32] = (bf16_t)bf16_bits(vn[j]);
      }
    }
  }
  ASTAMP(3);                                       // new token + append
  // every lane group's (m, l, acc[128]) to LDS, then each output thread folds them itself: one barrier
  if (gl == 0) { sm_[grp] = mmax; sl_[grp] = lsum; }
  *reinterpret_cast<float4*>(&sacc[grp][e0]) = make_float4(acc[0], acc[1], acc[2], acc[3]);
  *reinterpret_cast<float4*>(&sacc[grp][e0 + 4]) = make_float4(acc[4], acc[5], acc[6], acc[7]);
  __syncthreads();
  float M = -INFINITY, num = 0.f, den = 0.f;
  if (tid < d) {
#pragma unroll
    for (int gg = 0; gg < NG; ++gg) M = fmaxf(M, sm_[gg]);
#pragma unroll 8
    for (int gg = 0; gg < NG; ++gg) {
      const float w = ex2(sm_[gg] - M);          // a group without keys has m = -inf: weight 0 (M is finite: split 0 holds the new token, every other split at least one key or it is skipped below)
      den = fmaf(w, sl_[gg], den);
      num = fmaf(w, sacc[gg][tid], num);
    }
  }
  if (nsplit == 1) {
    if (tid < d) out[(int64_t)r * ldo + h * d + tid] = num / den;
    ASTAMP(4);                                     // merge + store
    return;
  }
  // split keys: partial (M, den, num[128]) per block; the block that draws the last ticket folds them
  float* pp = part + (((int64_t)r * heads + h) * nsplit + split) * (d + 2);
  if (tid < d) {
    const bool empty = (ke <= ks) && split != 0;
    pp[2 + tid] = empty ? 0.f : num;
    if (tid == 0) { pp[0] = empty ? -INFINITY : M; pp[1] = empty ? 0.f : den; }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (tid == 0) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const int tk = __hip_atomic_fetch_add(&tickets[r * heads + h], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    s_last = (tk == nsplit - 1);
    if (s_last) {
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __hip_atomic_store(&tickets[r * heads + h], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);     // ready for the next launch
    }
  }
  __syncthreads();
  if (!s_last) return;
  // fold the nsplit partials: (m, den) of every split to LDS, then thread (split quarter, d) sums its quarter of the splits with its loads in
  // flight together and the quarters meet in LDS (a serial loop of 2 x nsplit dependent loads per output was most of the split path's 3.9 us)
  float* fm = &sacc[0][0];                                 // [2][nsplit] (m, den), then [NW / 2][d] quarter sums
  const float* p0 = part + (((int64_t)r * heads + h) * nsplit) * (d + 2);
  if (tid < nsplit) { fm[tid] = p0[tid * (d + 2)]; fm[nsplit + tid] = p0[tid * (d + 2) + 1]; }
  __syncthreads();
  float MM = -INFINITY;
  for (int sgi = 0; sgi < nsplit; ++sgi) MM = fmaxf(MM, fm[sgi]);
  constexpr int NQ = NW / 2;                               // NW * 64 threads = NQ quarters x 128 d
  const int dq = tid & (d - 1), sq = tid >> 7;
  float nn = 0.f;
#pragma unroll 4
  for (int sgi = sq; sgi < nsplit; sgi += NQ) nn = fmaf(ex2(fm[sgi] - MM), p0[sgi * (d + 2) + 2 + dq], nn);     // an empty split has m = -inf: weight 0, num 0
  float* qs = fm + 2 * nsplit;
  qs[sq * d + dq] = nn;
  __syncthreads();
  if (tid < d) {
    float dn = 0.f, tot = 0.f;
    for (int sgi = 0; sgi < nsplit; ++sgi) dn = fmaf(ex2(fm[sgi] - MM), fm[nsplit + sgi], dn);
#pragma unroll
    for (int qq = 0; qq < NQ; ++qq) tot += qs[qq * d + tid];
    out[(int64_t)r * ldo + h * d + tid] = tot / dn;
  }
}


// ---------------------------------------------------------------------------------------------------------------------------------------
// Grouped-query decode attention on the matrix cores: ONE workgroup per (row, KV head[, key split]) serves all G = heads / kv_heads q heads of
// the group from a single pass over K / V (the per-head kernel above gives every q head its own workgroup on consecutive block ids: six XCD
// L2s each fetch the same K / V - 3.06 MB fetched per launch for 0.4 MB of cache at S = 370, rocprofv3 FETCH_SIZE, r02).
//   scores   S^T[key, q] = K[key, :] . Q[q, :]     v_mfma_f32_16x16x32_bf16: A = 16 keys x 32 d straight from the key-major cache (one 16-B
//                                                  load per lane), B = Q^T: lane (n = lane & 15, g = lane >> 4) holds d = 32 s + 8 g .. + 8 of
//                                                  q head n, RoPE applied in registers, split into bf16 hi + lo parts (fp32-grade products);
//                                                  queries n >= G are zero columns
//   softmax  the accumulator has the query on the lane: lane (n, g) holds keys 4 g + {0..3} of each 16-key tile; running maximum per query
//            (two cross-row exchanges per 32 keys), sum kept per lane and folded once at the end; log2 domain
//   output   O^T[d, q] += V^T[d, key] . P[key, q]  A = V^T from the tile-major transposed value cache vv_kv.vt (one 16-B load per 16 d x 32 keys,
//                                                  a whole key tile is 8 KB of contiguous memory), B = P straight from the score registers
//                                                  (key slot 8 g + j of the 32-deep step = the lane's own eight scores), again as bf16 hi + lo
// Each of the NW waves walks 32-key tiles wave, wave + NW, ...; the waves' (m, l, O) meet in LDS once.  The new token's score and value come
// from the projection registers; the workgroup of split 0 appends k (rotated), v and the transposed v at slot pos.
// ---------------------------------------------------------------------------------------------------------------------------------------
typedef __bf16 gq_bf16x8 __attribute__((ext_vector_type(8)));
typedef float gq_f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int gq_u32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ unsigned gq_pk(float a, float b) { return bf16_bits(a) | (bf16_bits(b) << 16); }
__device__ __forceinline__ float gq_bf_round(float a) { return __uint_as_float(bf16_bits(a) << 16); }
__device__ __forceinline__ gq_bf16x8 gq_frag(att_raw v) { return __builtin_bit_cast(gq_bf16x8, v); }
// split 8 floats into bf16 hi and lo fragments: hi + lo reproduces the fp32 value to 2^-17 relative
__device__ __forceinline__ void gq_split(const float (&x)[8], att_raw& hi, att_raw& lo) {
  float h[8], l[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) { h[j] = gq_bf_round(x[j]); l[j] = x[j] - h[j]; }
  hi.x = gq_pk(h[0], h[1]); hi.y = gq_pk(h[2], h[3]); hi.z = gq_pk(h[4], h[5]); hi.w = gq_pk(h[6], h[7]);
  lo.x = gq_pk(l[0], l[1]); lo.y = gq_pk(l[2], l[3]); lo.z = gq_pk(l[4], l[5]); lo.w = gq_pk(l[6], l[7]);
}

constexpr int GQ_MAXG = 8;      // q heads per KV head covered (1.5B: 6, 7B: 7)
constexpr int GQ_PITCH = 132;   // floats per (wave, query) row of the merge buffer

template <int NW>
__global__ __launch_bounds__(NW * 64) void attn_decode_gqa_kernel(const float* qkv, int64_t ld, int heads, vv_kv kv, int layer, const float2* rope, const int* lens,
                                                                 float* out, int64_t ldo, float* part, int* tickets) {
  constexpr int d = 128;
  __shared__ __attribute__((aligned(16))) float so[NW][GQ_MAXG][GQ_PITCH];     // per wave and query: O[128], then m, l
  __shared__ float s_new[GQ_MAXG];
  __shared__ int s_last;
#ifdef VV_CF_TIMING
  long long tprev_ = wall_clock64();
#endif
  const int kvh = blockIdx.x, r = blockIdx.y, split = blockIdx.z, nsplit = gridDim.z;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n = lane & 15, g = lane >> 4;
  const int G = heads / kv.kv_heads;
  const int nq = n < G ? n : G - 1;                      // lanes of the unused query columns read head G - 1 (zeroed below)
  const float* row = qkv + (int64_t)r * ld;
  const int64_t base = ((((int64_t)layer * kv.rows + r) * kv.kv_heads + kvh) * kv.s_max) * d;
  bf16_t* kc = reinterpret_cast<bf16_t*>(kv.k) + base;
  bf16_t* vc = reinterpret_cast<bf16_t*>(kv.v) + base;
  bf16_t* vt = reinterpret_cast<bf16_t*>(kv.vt) + base;   // [s_max / 32][d][32]
  // ---- requests: q chunks of this lane's head, the new k chunks, RoPE table, and (unsplit contexts) the first two key tiles -------------
  const float* qp = row + (int64_t)(kvh * G + nq) * d + 8 * g;
  const float* kp = row + (int64_t)(heads + kvh) * d + 8 * g;
  float4 qr[4][2], kr[4][2], rr[2][4];
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    qr[s][0] = *reinterpret_cast<const float4*>(qp + 32 * s); qr[s][1] = *reinterpret_cast<const float4*>(qp + 32 * s + 4);
    kr[s][0] = *reinterpret_cast<const float4*>(kp + 32 * s); kr[s][1] = *reinterpret_cast<const float4*>(kp + 32 * s + 4);
  }
  const float2* rp = rope + (int64_t)r * 64 + 8 * g;
#pragma unroll
  for (int s = 0; s < 2; ++s)
#pragma unroll
    for (int i = 0; i < 4; ++i) rr[s][i] = *reinterpret_cast<const float4*>(rp + 32 * s + 2 * i);     // {cos, sin} x 2 per float4
  att_raw kraw[2][2][4];            // [buffer][16-row score tile][k-step]
  att_raw vraw[2][8];               // [buffer][d tile]: keys 8 g .. 8 g + 7 of the 32-key tile for d row 16 dt + n
  // Row i of score tile t holds key 8 (i >> 2) + 4 t + (i & 3) of the 32-key tile: lane (n, g)'s eight scores are then keys 8 g .. 8 g + 7 in
  // order - the k slots 8 g + j of the P.V step - and its V^T fragment is ONE 16-byte load from the tile-major transposed cache.
  const int krow = 8 * (n >> 2) + (n & 3);
  auto issue = [&](int buf, int key0, int kmax_valid) {
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const int key = min(key0 + krow + 4 * t, kmax_valid);
#pragma unroll
      for (int s = 0; s < 4; ++s) kraw[buf][t][s] = *reinterpret_cast<const att_raw*>(kc + (int64_t)key * d + 32 * s + 8 * g);
    }
    const bf16_t* vp = vt + (int64_t)(min(key0, kv.s_max - 32) >> 5) * (32 * d) + n * 32 + 8 * g;
#pragma unroll
    for (int dt = 0; dt < 8; ++dt) vraw[buf][dt] = *reinterpret_cast<const att_raw*>(vp + dt * 16 * 32);
  };
  const bool spec = nsplit == 1;     // unsplit: the first tiles are requested before the position is known (masked / scrubbed when consumed)
  if (spec) issue(0, 32 * wave, kv.s_max - 1);
  const int pos = lens[r];
  const int per = ((pos + nsplit - 1) / nsplit + 31) & ~31;          // keys per split, whole tiles
  const int ks = split * per, ke = min(pos, ks + per);
  if (!spec) issue(0, ks + 32 * wave, kv.s_max - 1);
  ASTAMP(0);                                       // requests out, position known
  // ---- RoPE on q (scaled into the log2 domain) and on the new key; Q^T fragments as bf16 hi + lo --------------------------------------------
  const float qsc = rsqrtf((float)d) * 1.4426950408889634f;
  float qv[4][8], kn[4][8];
  {
    float qa[4][8], ka[4][8], cs[2][8], sn[2][8];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      qa[s][0] = qr[s][0].x; qa[s][1] = qr[s][0].y; qa[s][2] = qr[s][0].z; qa[s][3] = qr[s][0].w; qa[s][4] = qr[s][1].x; qa[s][5] = qr[s][1].y; qa[s][6] = qr[s][1].z; qa[s][7] = qr[s][1].w;
      ka[s][0] = kr[s][0].x; ka[s][1] = kr[s][0].y; ka[s][2] = kr[s][0].z; ka[s][3] = kr[s][0].w; ka[s][4] = kr[s][1].x; ka[s][5] = kr[s][1].y; ka[s][6] = kr[s][1].z; ka[s][7] = kr[s][1].w;
    }
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int i = 0; i < 4; ++i) { cs[s][2 * i] = rr[s][i].x; sn[s][2 * i] = rr[s][i].y; cs[s][2 * i + 1] = rr[s][i].z; sn[s][2 * i + 1] = rr[s][i].w; }
    const float live = n < G ? qsc : 0.f;
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int j = 0; j < 8; ++j) {        // x * cos + rotate_half(x) * sin: element e < 64 pairs with e + 64
        qv[s][j] = (qa[s][j] * cs[s][j] - qa[s + 2][j] * sn[s][j]) * live;
        qv[s + 2][j] = (qa[s + 2][j] * cs[s][j] + qa[s][j] * sn[s][j]) * live;
        kn[s][j] = ka[s][j] * cs[s][j] - ka[s + 2][j] * sn[s][j];
        kn[s + 2][j] = ka[s + 2][j] * cs[s][j] + ka[s][j] * sn[s][j];
      }
  }
  att_raw qhi[4], qlo[4];
#pragma unroll
  for (int s = 0; s < 4; ++s) gq_split(qv[s], qhi[s], qlo[s]);
  // the new token is dealt with HERE so that the fp32 q / k registers are dead before the key loop (its cache slot is masked there: kidx < pos)
  if (split == 0 && wave == NW - 1) {
    // the new token's score per q head, from the projection registers (its cache slot is written below, by this workgroup only)
    float dd = 0.f;
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
      for (int j = 0; j < 8; ++j) dd = fmaf(qv[s][j], kn[s][j], dd);
    dd += __shfl_xor(dd, 16);
    dd += __shfl_xor(dd, 32);
    if (g == 0 && n < G) s_new[n] = dd;
  }
  if (split == 0 && wave == 0 && n == 0) {          // one writer per (row, kv head): append the rotated key at slot pos
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      att_raw pk;
      pk.x = gq_pk(kn[s][0], kn[s][1]); pk.y = gq_pk(kn[s][2], kn[s][3]); pk.z = gq_pk(kn[s][4], kn[s][5]); pk.w = gq_pk(kn[s][6], kn[s][7]);
      *reinterpret_cast<att_raw*>(kc + (int64_t)pos * d + 32 * s + 8 * g) = pk;
    }
  }
  ASTAMP(1);                                       // q / k / rope landed, RoPE, split, new token
  // ---- key tiles ----------------------------------------------------------------------------------------------------------------------------
  issue(1, ks + 32 * (wave + NW), kv.s_max - 1);      // second tile of this wave: requested once the fp32 q / k registers are free
  gq_f32x4 oacc[8];
#pragma unroll
  for (int dt = 0; dt < 8; ++dt) oacc[dt] = gq_f32x4{0.f, 0.f, 0.f, 0.f};
  float m_run = -INFINITY, l_part = 0.f;
  auto consume = [&](int buf, int key0) {
    gq_f32x4 sc[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      sc[t] = gq_f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        sc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(gq_frag(kraw[buf][t][s]), gq_frag(qhi[s]), sc[t], 0, 0, 0);
        sc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(gq_frag(kraw[buf][t][s]), gq_frag(qlo[s]), sc[t], 0, 0, 0);
      }
    }
    float p[8];
    float tmax = -INFINITY;
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int kidx = key0 + 8 * g + 4 * t + j;
        p[4 * t + j] = kidx < ke ? sc[t][j] : -INFINITY;
        tmax = fmaxf(tmax, p[4 * t + j]);
      }
    tmax = fmaxf(tmax, __shfl_xor(tmax, 16));
    tmax = fmaxf(tmax, __shfl_xor(tmax, 32));
    const float m_new = fmaxf(m_run, tmax);            // finite: the caller only passes tiles with key0 < ke
    const float corr = ex2(m_run - m_new);
    float ps = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) { p[j] = ex2(p[j] - m_new); ps += p[j]; }
    l_part = l_part * corr + ps;
    m_run = m_new;
    att_raw phi, plo;
    gq_split(p, phi, plo);
    // value fragments: keys past the position may hold anything (0 x NaN): scrub them
    const int left = ke - (key0 + 8 * g);
    auto keep = [&](int lo) { return left >= lo + 2 ? 0xffffffffu : (left == lo + 1 ? 0x0000ffffu : 0u); };
    const unsigned a0 = keep(0), a1 = keep(2), a2 = keep(4), a3 = keep(6);
#pragma unroll
    for (int dt = 0; dt < 8; ++dt) {
      att_raw a;
      a.x = vraw[buf][dt].x & a0; a.y = vraw[buf][dt].y & a1; a.z = vraw[buf][dt].z & a2; a.w = vraw[buf][dt].w & a3;
      gq_f32x4 o = oacc[dt];
      o[0] *= corr; o[1] *= corr; o[2] *= corr; o[3] *= corr;
      o = __builtin_amdgcn_mfma_f32_16x16x32_bf16(gq_frag(a), gq_frag(phi), o, 0, 0, 0);
      o = __builtin_amdgcn_mfma_f32_16x16x32_bf16(gq_frag(a), gq_frag(plo), o, 0, 0, 0);
      oacc[dt] = o;
    }
  };
  for (int key0 = ks + 32 * wave; key0 < ke; key0 += 64 * NW) {       // two tiles in flight per wave, buffers with fixed roles
    consume(0, key0);
    if (key0 + 64 * NW < ke) issue(0, key0 + 64 * NW, kv.s_max - 1);
    if (key0 + 32 * NW >= ke) break;
    consume(1, key0 + 32 * NW);
    if (key0 + 96 * NW < ke) issue(1, key0 + 96 * NW, kv.s_max - 1);
  }
  ASTAMP(2);                                       // key tiles
  // ---- this wave's (m, l, O) to LDS ------------------------------------------------------------------------------------------------------------
  float l_tot = l_part + __shfl_xor(l_part, 16);
  l_tot += __shfl_xor(l_tot, 32);
  if (n < G) {
#pragma unroll
    for (int dt = 0; dt < 8; ++dt)
      *reinterpret_cast<float4*>(&so[wave][n][dt * 16 + 4 * g]) = make_float4(oacc[dt][0], oacc[dt][1], oacc[dt][2], oacc[dt][3]);
    if (g == 0) { so[wave][n][128] = m_run; so[wave][n][129] = l_tot; }
  }
  __syncthreads();
  ASTAMP(3);                                       // partials in LDS, barrier (waits for the slowest wave)
  // ---- merge: thread (q head, d) folds the waves' partials and the new token -----------------------------------------------------------------
  const float* vnew = row + (int64_t)(heads + kv.kv_heads + kvh) * d;
  float* pp = nullptr;
  for (int o = tid; o < G * d; o += NW * 64) {
    const int qh = o >> 7, dd = o & 127;
    float M = -INFINITY;
#pragma unroll
    for (int w = 0; w < NW; ++w) M = fmaxf(M, so[w][qh][128]);
    float sn = -INFINITY, vn = 0.f;
    if (split == 0) { sn = s_new[qh]; vn = vnew[dd]; M = fmaxf(M, sn); }
    float num = 0.f, den = 0.f;
    if (M > -INFINITY) {
#pragma unroll
      for (int w = 0; w < NW; ++w) {
        const float wgt = ex2(so[w][qh][128] - M);          // a wave without keys has m = -inf: weight 0
        den = fmaf(wgt, so[w][qh][129], den);
        num = fmaf(wgt, so[w][qh][dd], num);
      }
      if (split == 0) { const float wgt = ex2(sn - M); den += wgt; num = fmaf(wgt, vn, num); }
    }
    if (nsplit == 1) {
      out[(int64_t)r * ldo + (int64_t)(kvh * G + qh) * d + dd] = num / den;
    } else {
      pp = part + (((int64_t)r * heads + kvh * G + qh) * nsplit + split) * (d + 2);
      pp[2 + dd] = num;
      if (dd == 0) { pp[0] = M; pp[1] = den; }
    }
    if (split == 0 && qh == 0) {                       // append v (key-major and transposed copies) at slot pos
      const bf16_t vb = (bf16_t)bf16_bits(vn);
      vc[(int64_t)pos * d + dd] = vb;
      vt[(int64_t)(pos >> 5) * (32 * d) + dd * 32 + (pos & 31)] = vb;
    }
  }
  ASTAMP(4);                                       // merge + store
  if (nsplit == 1) return;
  // split keys: the workgroup that draws the last ticket of its (row, kv head) folds the partials of all G heads
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (tid == 0) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const int tk = __hip_atomic_fetch_add(&tickets[r * heads + kvh * G], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    s_last = (tk == nsplit - 1);
    if (s_last) {
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __hip_atomic_store(&tickets[r * heads + kvh * G], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);     // ready for the next launch
    }
  }
  __syncthreads();
  ASTAMP(5);                                       // partials out, release fence, ticket
  if (!s_last) return;
  // the last workgroup folds nsplit x G partials (up to 200 KB): (m, den) of every (head, split) go to LDS first, then thread (split quarter, d)
  // sums its quarter of the splits with 8 loads in flight and the four quarters meet in LDS - a serial loop of dependent loads over 64
  // splits cost more than the attention itself (1.5B, S = 16 000: 2.07 -> 3.56 ms per LLM step)
  float* sm = &so[0][0][0];                               // [2][G * nsplit] (m, den), then [4][G * d] quarter sums behind them
  const int gn = G * nsplit;
  for (int i = tid; i < gn; i += NW * 64) {
    const int qh = i / nsplit, sg = i - qh * nsplit;
    const float* pq = part + ((((int64_t)r * heads + kvh * G + qh) * nsplit) + sg) * (d + 2);
    sm[i] = pq[0];
    sm[gn + i] = pq[1];
  }
  __syncthreads();
  float* qsum = sm + 2 * gn;                              // [4][G][d]
  const int dd = tid & 127, sq = tid >> 7;                // NW * 64 = 512 threads: 4 split quarters x 128 d
  // ALL of a thread's partial loads (G heads x its quarter of the splits) are requested before any is used: folding head by head put one
  // round trip to memory per head in a row (8 us at S = 440, tools/convffn_phase.py attn_split).  J = splits per quarter, wave-uniform.
  auto fold = [&](auto jc) {
    constexpr int J = decltype(jc)::value;
    float v[GQ_MAXG][J];
#pragma unroll
    for (int qh = 0; qh < GQ_MAXG; ++qh) {
      const float* p0 = part + (((int64_t)r * heads + kvh * G + min(qh, G - 1)) * nsplit) * (d + 2) + 2 + dd;
#pragma unroll
      for (int j = 0; j < J; ++j) v[qh][j] = p0[(int64_t)min(sq + 4 * j, nsplit - 1) * (d + 2)];
    }
#pragma unroll
    for (int qh = 0; qh < GQ_MAXG; ++qh) {
      if (qh < G) {
        float MM = -INFINITY;
        for (int sg = 0; sg < nsplit; ++sg) MM = fmaxf(MM, sm[qh * nsplit + sg]);
        float nn = 0.f;
#pragma unroll
        for (int j = 0; j < J; ++j) {
          const int sg = sq + 4 * j;
          const float mv = sg < nsplit ? sm[qh * nsplit + sg] : -INFINITY;
          nn = fmaf(mv == -INFINITY ? 0.f : ex2(mv - MM), v[qh][j], nn);      // a split without keys (or past nsplit): weight 0
        }
        qsum[(sq * G + qh) * d + dd] = nn;
      }
    }
  };
  const int per_q = (nsplit + 3) >> 2;
  if (per_q <= 2) fold(std::integral_constant<int, 2>{});
  else if (per_q <= 4) fold(std::integral_constant<int, 4>{});
  else if (per_q <= 8) fold(std::integral_constant<int, 8>{});
  else fold(std::integral_constant<int, 16>{});            // nsplit <= 64 (launcher)
  __syncthreads();
  for (int o = tid; o < G * d; o += NW * 64) {
    const int qh = o >> 7, d2 = o & 127;
    float MM = -INFINITY;
    for (int sg = 0; sg < nsplit; ++sg) MM = fmaxf(MM, sm[qh * nsplit + sg]);
    float dn = 0.f;
    for (int sg = 0; sg < nsplit; ++sg) {
      const float mv = sm[qh * nsplit + sg];
      dn = fmaf(mv == -INFINITY ? 0.f : ex2(mv - MM), sm[gn + qh * nsplit + sg], dn);
    }
    const float nn = (qsum[(0 * G + qh) * d + d2] + qsum[(1 * G + qh) * d + d2]) + (qsum[(2 * G + qh) * d + d2] + qsum[(3 * G + qh) * d + d2]);
    out[(int64_t)r * ldo + (int64_t)(kvh * G + qh) * d + d2] = nn / dn;
  }
  ASTAMP(6);                                       // final fold (last workgroup only)
}

int g_gqa_keys = 1024;   // tuning hook "attn_gqa_keys": cached keys per split of the grouped kernel at long contexts
int g_gqa = 1;     // tuning hook "attn_gqa"

}  // namespace

// 1 launched, 0 not covered (caller falls back to the generic kernel), < 0 error.  part / tickets: split-key workspace or null (nsplit = 1)
int vv_launch_attn_decode(const float* qkv, int64_t ld_qkv, int R, int heads, const vv_kv* kv, int layer, const float2* rope, const int* lens, float* out,
                          int64_t ldo, float* part, int* tickets, int nsplit, int part_cap, hipStream_t s) {
  if (kv->kvdt != VV_BF16 || kv->head_dim != 128) return 0;
  if (((uintptr_t)qkv % 16) || (ld_qkv % 4) || ((uintptr_t)rope % 16) || ((uintptr_t)kv->k % 16) || ((uintptr_t)kv->v % 16)) return 0;
  if (!part || !tickets || nsplit < 1) nsplit = 1;
  if (nsplit > 16) nsplit = 16;
  const int G = heads / kv->kv_heads;
  // The grouped kernel pays where K / V bytes set the time - from ~6K cached keys on at 1.5B (2 KV heads), ~3K at 7B (4 KV heads), and most in the
  // long-form regime (the reference's 45 - 90 minute dialogues: 32K - 64K contexts): one K / V pass per KV head instead of one per q head.
  // Measured LLM step, per-head -> grouped (tools/mb_attn_long.py, MI355X, profiles/r03_attn_long_context.txt): 7B S = 3 600: 3.67 -> 3.44 ms,
  // S = 7 200: 4.37 -> 3.59, S = 32 000: 6.21 -> 4.23 ms; 1.5B S = 7 200: 1.68 -> 1.60, S = 16 000: 2.04 -> 1.77, S = 64 000: 3.91 -> 2.65 ms.
  // Below that its few workgroups pull a (row, KV head)'s whole cache through one CU's address path (12 us per layer at S = 370 against
  // 7.4 us for the per-head kernel's 24 workgroups; phase timing: tools/convffn_phase.py attn 2 / attn_split), so short contexts keep the
  // per-head kernel.  g_gqa (tuning hook "attn_gqa"): 1 = by that rule, 2 = always, 0 = never.
  const long gqa_min = 12288;                              // cached key rows per (layer, cache row): kv_heads x s_max
  if (g_gqa && ((nsplit > 1 && (long)kv->kv_heads * kv->s_max >= gqa_min) || g_gqa == 2) && kv->vt && G <= GQ_MAXG && kv->s_max % 32 == 0 && kv->s_max >= 64 && ((uintptr_t)kv->vt % 8) == 0) {
    // very long contexts: more splits than the per-head kernel's 16, so that a workgroup pulls ~0.5 MB through its CU instead of megabytes
    // (4 - 8 (row, KV head) pairs x 16 splits leave three quarters of the chip idle: 68 us per layer at S = 64 000)
    int ng = nsplit;
    if (nsplit > 1) {                                   // ~g_gqa_keys keys per split once the per-head kernel's 16 splits exceed that
      if (kv->s_max / g_gqa_keys > ng) ng = kv->s_max / g_gqa_keys;
      if (ng > part_cap) ng = part_cap;
      if (ng > 64) ng = 64;                             // the last workgroup folds <= 16 splits per thread quarter with all loads in flight
    }
    hipLaunchKernelGGL((attn_decode_gqa_kernel<8>), dim3(kv->kv_heads, R, ng), dim3(512), 0, s, qkv, ld_qkv, heads, *kv, layer, rope, lens, out, ldo, part, tickets);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return vv_set_error(VV_E_HIP, "vv_attn_decode (gqa): %s", hipGetErrorString(e));
    return 1;
  }
  hipLaunchKernelGGL((attn_decode_kernel<8>), dim3(heads, R, nsplit), dim3(512), 0, s, qkv, ld_qkv, heads, *kv, layer, rope, lens, out, ldo, part, tickets);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return vv_set_error(VV_E_HIP, "vv_attn_decode: %s", hipGetErrorString(e));
  return 1;
}

void vv_attn_decode_set_gqa(int on) { g_gqa = on; }
void vv_attn_decode_set_gqa_keys(int k) { if (k >= 256) g_gqa_keys = k; }

#ifdef VV_CF_TIMING
extern "C" int vv_attn_debug_times(unsigned long long* out8, int reset) {
  if (hipMemcpyFromSymbol(out8, HIP_SYMBOL(g_at_t), 8 * sizeof(unsigned long long)) != hipSuccess) return -1;
  if (reset) { unsigned long long z[8] = {0}; if (hipMemcpyToSymbol(HIP_SYMBOL(g_at_t), z, sizeof(z)) != hipSuccess) return -1; }
  return 0;
}
#endif
