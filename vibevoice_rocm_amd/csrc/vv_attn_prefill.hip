// vv_attn_prefill.hip — prompt (prefill) attention on the matrix cores: causal GQA attention of R prompt rows against one cache row,
// bf16 KV cache, head_dim 128 (Qwen2 attention under modeling_vibevoice.py:187-199 at step 0 of generate(),
// modeling_vibevoice_inference.py:478; the reference picks FlashAttention-2 / SDPA for it, demo/inference_from_file.py:23-38).
//
// One wave per (32-query tile, q head), flash style, both products on v_mfma_f32_32x32x16_bf16 with NO LDS and no transposes:
//   S^T[key, q]  = K[key, :] . Q[q, :]      A = K rows straight from the cache (16 B per lane per 16 d), B = Q^T kept in registers
//                                           for the whole tile (fp32 -> bf16, pre-scaled by 1/sqrt(d) * log2 e)
//   the accumulator has the QUERY on the lane and the keys in its 16 registers: the online softmax is lane-local (one exchange with
//   the partner lane l ^ 32 per key tile for the maximum), running sum kept per lane half
//   O^T[d, q]   += V^T[d, key] . P[key, q]  B = P straight from the S^T accumulator registers (cdna_hip_programming.md, "An accumulator
//                                           tile as the next MFMA's operand": element j of lane half h is key 16s + 8(j>>2) + 4h + (j&3)),
//                                           A = V^T rows in that key order: two 8-byte loads per fragment from the TRANSPOSED value
//                                           cache vv_kv.vt, kept in 32-key tiles [..][s_max / 32][head_dim][32] (a key tile is 8 KB of
//                                           contiguous memory: one load instruction of the wave covers a 2 KB run instead of 32 rows
//                                           s_max elements apart) by vv_rope_store and vv_attn_decode
// The 6 / 7 q heads of a GQA group read the same K / V tiles (L2 hits); the K fragments of tile t + 1 are requested before tile t is
// scored.  Heaviest query tiles (most key tiles under the causal mask) are dispatched first.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <math.h>

#include "vv_hip.h"
#include "vv_common.h"

namespace {

typedef unsigned short bf16_t;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ unsigned pk2(float a, float b) {      // two floats -> packed bf16 pair, round to nearest even
  const unsigned ua = __float_as_uint(a), ub = __float_as_uint(b);
  return ((ua + 0x7fffu + ((ua >> 16) & 1u)) >> 16) | (((ub + 0x7fffu + ((ub >> 16) & 1u)) >> 16) << 16);
}
__device__ __forceinline__ bf16x8 as_frag(u32x4 v) { return __builtin_bit_cast(bf16x8, v); }

__global__ __launch_bounds__(64) void attn_prefill_kernel(const float* qkv, int64_t ld, int heads, vv_kv kv, int layer, const int* lens, const int* cache_rows,
                                                          int R, float* out, int64_t ldo) {
  constexpr int d = 128;
  const int lane = threadIdx.x, r = lane & 31, h = lane >> 5;
  const int nqt = gridDim.x;
  const int qt = nqt - 1 - blockIdx.x;                     // heaviest (last) query tiles first
  const int head = blockIdx.y;
  const int kvh = head / (heads / kv.kv_heads);
  const int q0 = qt * 32;
  const int qrow = min(q0 + r, R - 1);                    // rows past R repeat the last row (never stored)
  const int len_q = lens[qrow];
  // cache row of THIS lane's query (vv_hip.h: cache_rows[r], NULL = r).  A tile's K / V operands come from one cache row, so a tile whose
  // queries name several rows runs once per distinct row: the lanes of the other rows ride along (finite garbage on their own accumulator
  // columns only: an MFMA output column depends on its own B column) and store nothing.  A prompt prefill has one row per call: one pass.
  const int my_crow = cache_rows ? cache_rows[qrow] : qrow;
  // Q^T fragments of the tile: B operand of k-step s holds Q[q0 + r][16 s + 8 h + j], scaled into the log2 domain
  const float qsc = rsqrtf((float)d) * 1.4426950408889634f;
  const float* qp = qkv + (int64_t)qrow * ld + head * d + 8 * h;
  bf16x8 qf[8];
#pragma unroll
  for (int s = 0; s < 8; ++s) {
    const float4 a = *reinterpret_cast<const float4*>(qp + 16 * s), b = *reinterpret_cast<const float4*>(qp + 16 * s + 4);
    u32x4 p;
    p.x = pk2(a.x * qsc, a.y * qsc); p.y = pk2(a.z * qsc, a.w * qsc); p.z = pk2(b.x * qsc, b.y * qsc); p.w = pk2(b.z * qsc, b.w * qsc);
    qf[s] = as_frag(p);
  }
  unsigned long long todo = __ballot(1);
  while (todo) {
  const int crow = __builtin_amdgcn_readlane(my_crow, __ffsll((long long)todo) - 1);
  const bool mine = my_crow == crow;
  todo &= ~__ballot(mine);
  // number of keys any query of this pass may see: max(len) + 1 over its lanes
  int kmax = mine ? len_q + 1 : 1;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) kmax = max(kmax, __shfl_xor(kmax, o));
  const int n_tiles = (kmax + 31) >> 5;
  const int64_t base = ((((int64_t)layer * kv.rows + crow) * kv.kv_heads + kvh) * kv.s_max) * d;
  const bf16_t* kc = reinterpret_cast<const bf16_t*>(kv.k) + base;          // [s_max][d]
  const bf16_t* vt = reinterpret_cast<const bf16_t*>(kv.vt) + base;         // [s_max / 32][d][32]
  f32x16 oacc[4];
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int i = 0; i < 16; ++i) oacc[t][i] = 0.f;
  float m_run = -INFINITY, l_half = 0.f;
  // K fragments of key tile 0: A operand of k-step s = K[key0 + r][16 s + 8 h .. + 8]
  u32x4 kcur[8], knxt[8];
  const bf16_t* kp = kc + (int64_t)r * d + 8 * h;
#pragma unroll
  for (int s = 0; s < 8; ++s) kcur[s] = *reinterpret_cast<const u32x4*>(kp + 16 * s);
  for (int t = 0; t < n_tiles; ++t) {
    const int key0 = t * 32;
    // V^T fragments of this tile (consumed after the softmax): for d tile dt, k-step s: keys key0 + 16 s + 4 h + {0..3} and + 8
    u32x2 vf[4][2][2];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) {
      const bf16_t* vp = vt + (int64_t)t * (32 * d) + (dt * 32 + r) * 32 + 4 * h;       // 32-key tiles: [tile][d][32]
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        vf[dt][s][0] = *reinterpret_cast<const u32x2*>(vp + 16 * s);
        vf[dt][s][1] = *reinterpret_cast<const u32x2*>(vp + 16 * s + 8);
      }
    }
    if (t + 1 == n_tiles) {
      // slots past the last visible key carry P = 0, but 0 x NaN is NaN: whatever bit patterns the cache holds behind the position
      // (vv_kv.vt need not be zero-initialised) are scrubbed from the value fragments of the last tile
#pragma unroll
      for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int x = 0; x < 2; ++x) {
          const int left = kmax - (key0 + 16 * s + 8 * x + 4 * h);       // visible keys among this fragment's 4
          const unsigned m0 = left >= 2 ? 0xffffffffu : (left == 1 ? 0x0000ffffu : 0u);
          const unsigned m1 = left >= 4 ? 0xffffffffu : (left == 3 ? 0x0000ffffu : 0u);
#pragma unroll
          for (int dt = 0; dt < 4; ++dt) { vf[dt][s][x].x &= m0; vf[dt][s][x].y &= m1; }
        }
    }
    if (t + 1 < n_tiles) {
#pragma unroll
      for (int s = 0; s < 8; ++s) knxt[s] = *reinterpret_cast<const u32x4*>(kp + (int64_t)(key0 + 32) * d + 16 * s);
    }
    // S^T = K Q^T over d = 128
    f32x16 sacc;
#pragma unroll
    for (int i = 0; i < 16; ++i) sacc[i] = 0.f;
#pragma unroll
    for (int s = 0; s < 8; ++s) sacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_frag(kcur[s]), qf[s], sacc, 0, 0, 0);
    // causal mask + online softmax for query (lane & 31): this lane holds keys key0 + (i & 3) + 8 (i >> 2) + 4 h
    float tmax = -INFINITY;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int kidx = key0 + (i & 3) + 8 * (i >> 2) + 4 * h;
      sacc[i] = kidx <= len_q ? sacc[i] : -INFINITY;
      tmax = fmaxf(tmax, sacc[i]);
    }
    tmax = fmaxf(tmax, __shfl_xor(tmax, 32));             // both lane halves of a query must scale P by the same maximum
    const float m_new = fmaxf(m_run, tmax);                // finite from tile 0 on: key 0 is visible to every query
    const float corr = __builtin_amdgcn_exp2f(m_run - m_new);
    float psum = 0.f;
    float p[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) { p[i] = __builtin_amdgcn_exp2f(sacc[i] - m_new); psum += p[i]; }
    l_half = l_half * corr + psum;
    m_run = m_new;
#pragma unroll
    for (int dt = 0; dt < 4; ++dt)
#pragma unroll
      for (int i = 0; i < 16; ++i) oacc[dt][i] *= corr;
    // P as the B operand, straight from the accumulator registers: k-step s takes registers 8 s .. 8 s + 7
    bf16x8 pf[2];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      u32x4 w;
      w.x = pk2(p[8 * s + 0], p[8 * s + 1]); w.y = pk2(p[8 * s + 2], p[8 * s + 3]); w.z = pk2(p[8 * s + 4], p[8 * s + 5]); w.w = pk2(p[8 * s + 6], p[8 * s + 7]);
      pf[s] = as_frag(w);
    }
    // O^T += V^T P
#pragma unroll
    for (int dt = 0; dt < 4; ++dt)
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        u32x4 a;
        a.x = vf[dt][s][0].x; a.y = vf[dt][s][0].y; a.z = vf[dt][s][1].x; a.w = vf[dt][s][1].y;
        oacc[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_frag(a), pf[s], oacc[dt], 0, 0, 0);
      }
#pragma unroll
    for (int s = 0; s < 8; ++s) kcur[s] = knxt[s];
  }
  const float l_tot = l_half + __shfl_xor(l_half, 32);
  const float inv = 1.0f / l_tot;
  if (mine && q0 + r < R) {
    float* op = out + (int64_t)(q0 + r) * ldo + head * d + 4 * h;
#pragma unroll
    for (int dt = 0; dt < 4; ++dt)
#pragma unroll
      for (int g = 0; g < 4; ++g) {       // registers 4 g .. 4 g + 3 = d rows dt * 32 + 8 g + 4 h + {0..3}
        *reinterpret_cast<float4*>(op + dt * 32 + 8 * g) =
            make_float4(oacc[dt][4 * g] * inv, oacc[dt][4 * g + 1] * inv, oacc[dt][4 * g + 2] * inv, oacc[dt][4 * g + 3] * inv);
      }
  }
  }
}

}  // namespace

// 1 launched, 0 not covered (caller falls back), < 0 error
int vv_launch_attn_prefill(const float* qkv, int64_t ld_qkv, int R, int heads, const vv_kv* kv, int layer, const int* lens, const int* cache_rows,
                           float* out, int64_t ldo, hipStream_t s) {
  if (kv->kvdt != VV_BF16 || kv->head_dim != 128 || !kv->vt || R < 16) return 0;
  if ((kv->s_max % 32) || ((uintptr_t)qkv % 16) || (ld_qkv % 4) || ((uintptr_t)out % 16) || (ldo % 4) || ((uintptr_t)kv->k % 16) || ((uintptr_t)kv->vt % 8)) return 0;
  const int nqt = (R + 31) / 32;
  hipLaunchKernelGGL(attn_prefill_kernel, dim3(nqt, heads), dim3(64), 0, s, qkv, ld_qkv, heads, *kv, layer, lens, cache_rows, R, out, ldo);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return vv_set_error(VV_E_HIP, "vv_attn (prefill): %s", hipGetErrorString(e));
  return 1;
}
