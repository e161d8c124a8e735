// vv_gemv_rows.hip — the weight-streaming GEMV for 3 .. 8 activation rows (dialogues batched into the row dimension: rows = {positive,
// negative} x 2 .. 4 dialogues), on the matrix cores.
//
// Roofline: HBM.  The VALU GEMV (vv_gemv_stream.hip) keeps M x K/wave activations in registers and spends M FMAs per weight: its time grows
// 1.2 - 1.6 us per row and M = 8 does not fit its registers at all for K > 1024.  Here one v_mfma_f32_16x16x32_bf16 consumes a wave's 1 KB
// weight load (16 weight rows x 32 k, 16 bytes per lane, straight to registers: nothing is shared between waves) against a 16-row activation
// fragment whose rows 0 .. 7 are the bf16 HIGH parts and rows 8 .. 15 the bf16 LOW parts of the 8 fp32 activation rows (hi + lo reproduces
// the fp32 value to 2^-17; weights are bf16, products and sums fp32): the cost per weight byte does not depend on the row count.
//
//   weights             row-major [N, K], or - VV_LIN_W_FRAG - the fragment-major copy [N / 16][K / 32][64 lanes][8]: lane (n = lane & 15,
//                       c = lane >> 4) of k step j of row group g finds W[16 g + n][32 j + 8 c ..+8] at ((g * K/32 + j) * 64 + lane) * 8, so
//                       one wave instruction reads 1 KB of contiguous memory and a wave's steps are one contiguous run.  From the row-major
//                       matrix the same instruction gathers 16 rows x 64 B (measured: 16.9 vs 14.0 us on the head's SwiGLU GEMV, 24.1 vs
//                       18.4 us on the LLM's, tools/mb_rows8.py).
//   workgroup (g, s)    = 16 weight rows (row group g)  x  K slice s of `ksplit`;  its NW waves split the slice into `spw` 32-wide k steps each
//   prologue            the block's x slice [8, K / ksplit] is fetched once (fp32), RMSNorm statistics over the block's columns, norm weight and
//                       adaLN shift / scale applied, split into hi / lo and laid out in LDS in fragment order; every wave then holds its A
//                       fragments in registers.  All of it is REQUESTED before the weights (loads return in order) and computed while the
//                       weights are in flight.
//   ksplit == 1         (K <= 2048) a block owns whole rows; with more row groups than resident blocks (the SwiGLU shapes) the blocks are
//                       persistent: they keep their fragments and walk row groups g, g + grid, ... with the next group's weights in flight.
//   ksplit > 1          long K with few rows (N = 1536: 96 row groups) cannot fill 256 CUs by row groups alone: the K slices of one row group
//                       meet through write-through partial tiles and a ticket; the last-arriving block folds them in a fixed order
//                       (deterministic) with all partial loads in flight at once, multiplies by rstd (RMSNorm is a per-row scalar: the
//                       slices accumulate the un-normalised x * norm_w and each contributes its partial sum of squares) and runs the epilogue.
//                       No release / acquire fences: the partials are sc1 stores and sc1 loads (cdna_hip_programming.md, in-launch split-K
//                       reduction); an agent-scope release per block (an L2 write-back each) cost 5 - 20 us per launch with 400 - 800 blocks.
//   epilogue            bias / GELU / SwiGLU / per-row or per-channel gate / residual, one output per thread, operands requested up front.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "vv_hip.h"
#include "vv_common.h"

namespace {

typedef unsigned short bf16_t;
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ unsigned bf_bits(float f) {     // round to nearest even (finite activations)
  const unsigned u = __float_as_uint(f);
  return (u + 0x7fffu + ((u >> 16) & 1u)) >> 16;
}
__device__ __forceinline__ float silu1(float v) { return v / (1.0f + expf(-v)); }
__device__ __forceinline__ float gelu1(float v) { return 0.5f * v * (1.0f + erff(v * 0.70710678118654752440f)); }

struct RowsAux {
  int atomic;        // ksplit > 1 with a linear epilogue whose residual already sits in `out` (res == out): every K slice adds gate * (partial [+ bias])
                     // to out with fp32 atomics - no partial tiles, no ticket, nothing to wait for (the summation ORDER of the slices then varies from
                     // run to run: results reproduce to fp32 rounding, ~1e-7, not bit for bit; vv_tune("gemv_rows_atomic", 0) keeps the ticket)
  int dbg;           // timing experiments (wrong results): 2 = no ticket merge, 4 = no activation loads
  float* part;       // [n_groups][ksplit][PST] partial tiles (ksplit > 1)
  int* tickets;      // [n_groups], zero on entry, left zero
  int ksplit, spw;   // K slices per row group; 32-wide k steps per wave
  int n_groups;
};

constexpr int MAXSPLIT = 16;

struct EpiOps { float b, g, r; };

template <bool DUAL, int NW, int KS, bool PERS>
__global__ __launch_bounds__(NW * 64) void gemv_rows_kernel(const vv_lin_args a, const RowsAux x) {
  constexpr int T = NW * 64;
  constexpr int NCH = (KS + 7) / 8;                 // 4-column chunks per thread per activation row
  constexpr int NM = DUAL ? 2 : 1;
  constexpr int PST = 128 * NM + 8;                 // floats per partial tile: [NM][8][16] sums + 8 partial sums of squares
  constexpr bool MOD_OK = NCH == 1;                 // adaLN shift / scale rows: only the instantiations whose slice is one chunk per thread (registers)
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  u32x4* xa = reinterpret_cast<u32x4*>(smem);       // [NW][KS][64] A fragments (16 B per lane)
  __shared__ float red[PERS ? 2 : 1][NW][NM][256];  // per wave: the 16 x 16 accumulator tile(s); ping-pong when the block walks row groups
  __shared__ float ssr[NW][8];
  __shared__ float s_tot[8];
  __shared__ int s_last;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n = lane & 15, c = lane >> 4;
  const int ks = blockIdx.y;
  const int K = a.k, N = a.n, mr = a.m;
  const int spw = x.spw, ksplit = x.ksplit, n_groups = x.n_groups;
  const int steps_total = K >> 5;
  const int step0 = ks * NW * spw;                  // first k step of this block
  const int k0 = step0 << 5;
  const bool rms = a.pro == VV_PRO_RMSNORM;
  const bool has_nw = rms && a.norm_w != nullptr, has_mod = MOD_OK && a.mod_scale != nullptr;
  const bool frag = (a.flags & VV_LIN_W_FRAG) != 0;
  const bool reused = (a.flags & VV_LIN_W_REUSED) != 0;

  // ---- requests, oldest first: activations, norm weight, modulation, epilogue operands, then the weights ----------------------------------
  float4 xv[8][NCH], nv[NCH], sv[8][MOD_OK ? NCH : 1], cv[8][MOD_OK ? NCH : 1];
  bool cval[NCH];
#pragma unroll
  for (int cc = 0; cc < NCH; ++cc) {
    const int q = tid + cc * T;
    cval[cc] = q < NW * spw * 8 && (k0 + 4 * q) < K;
    const int kk = (cval[cc] && !(x.dbg & 4)) ? k0 + 4 * q : 4 * (tid & 7);
#pragma unroll
    for (int m = 0; m < 8; ++m) xv[m][cc] = *reinterpret_cast<const float4*>(a.x + (int64_t)(m < mr ? m : mr - 1) * a.ldx + kk);
    if (has_nw) nv[cc] = *reinterpret_cast<const float4*>(a.norm_w + kk);
    if constexpr (MOD_OK) {
      if (has_mod) {
#pragma unroll
        for (int m = 0; m < 8; ++m) {
          const int64_t mo = (int64_t)(m < mr ? m : mr - 1) * a.ld_mod + kk;
          sv[m][cc] = *reinterpret_cast<const float4*>(a.mod_shift + mo);
          cv[m][cc] = *reinterpret_cast<const float4*>(a.mod_scale + mo);
        }
      }
    }
  }
  // epilogue operands of output (em, en) of a row group = thread tid < 128; absent operands read x[0] (a valid address), ignored at use
  const int em = (tid >> 4) & 7, en = tid & 15;
  const int emr = em < mr ? em : mr - 1;
  auto load_eo = [&](int grp) {
    EpiOps e;
    const int egn = min(min(grp, n_groups - 1) * 16 + en, N - 1);
    const float* pb = a.bias ? a.bias + egn : a.x;
    const float* pg = a.gate ? a.gate + (a.gate_ld ? (int64_t)emr * a.gate_ld + egn : (int64_t)egn) : a.x;
    const float* pr = a.res ? a.res + (int64_t)emr * a.ldres + egn : a.x;
    e.b = *pb; e.g = *pg; e.r = *pr;
    return e;
  };
  const int sb = step0 + wave * spw;                // this wave's first k step
  const bf16_t* __restrict__ W = reinterpret_cast<const bf16_t*>(a.w);
  const bf16_t* __restrict__ W2 = reinterpret_cast<const bf16_t*>(a.w2);
  auto issue = [&](u32x4 (&w)[KS], u32x4 (&w2)[DUAL ? KS : 1], int grp) {
    const bool glive = grp < n_groups;
    int64_t base;
    if (frag) base = ((int64_t)grp * steps_total) * 512 + lane * 8;
    else base = (int64_t)min(grp * 16 + n, N - 1) * K + c * 8;
    const int64_t sstep = frag ? 512 : 32;
#pragma unroll
    for (int j = 0; j < KS; ++j) {
      const bool live = glive && j < spw && sb + j < steps_total;
      const int64_t off = live ? base + (int64_t)(sb + j) * sstep : 0;
      const u32x4* p1 = reinterpret_cast<const u32x4*>(W + off);
      const u32x4* p2 = reinterpret_cast<const u32x4*>(W2 + off);
      if (reused) {
        w[j] = *p1;
        if (DUAL) w2[j] = *p2;
      } else {
        w[j] = __builtin_nontemporal_load(p1);
        if (DUAL) w2[j] = __builtin_nontemporal_load(p2);
      }
    }
  };
  const int gstride = gridDim.x;
  int g = blockIdx.x;
  u32x4 wc[KS], wc2[DUAL ? KS : 1];
  u32x4 wn[PERS ? KS : 1], wn2[(PERS && DUAL) ? KS : 1];
  EpiOps eo = load_eo(g), eo_n = eo;
  issue(wc, wc2, g);
  if constexpr (PERS) {
    eo_n = load_eo(g + gstride);
    issue(wn, wn2, g + gstride);
  }
  __builtin_amdgcn_sched_barrier(0);
#define VV_FENCE4(v) asm volatile("" : "+v"((v).x), "+v"((v).y), "+v"((v).z), "+v"((v).w))
#pragma unroll
  for (int cc = 0; cc < NCH; ++cc) {
#pragma unroll
    for (int m = 0; m < 8; ++m) VV_FENCE4(xv[m][cc]);
    if (has_nw) VV_FENCE4(nv[cc]);
    if constexpr (MOD_OK) {
      if (has_mod) {
#pragma unroll
        for (int m = 0; m < 8; ++m) { VV_FENCE4(sv[m][cc]); VV_FENCE4(cv[m][cc]); }
      }
    }
  }
#undef VV_FENCE4

  // ---- activation prologue -------------------------------------------------------------------------------------------------------------
  float rstd[8];
#pragma unroll
  for (int m = 0; m < 8; ++m) rstd[m] = 1.0f;
  if (rms) {
#pragma unroll
    for (int m = 0; m < 8; ++m) {
      float s1 = 0.f;
#pragma unroll
      for (int cc = 0; cc < NCH; ++cc) {
        const float4 v = xv[m][cc];
        s1 += cval[cc] ? (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w) : 0.f;
      }
      s1 = vv_wave_sum(s1);
      if (lane == 0) ssr[wave][m] = s1;
    }
    __syncthreads();
    if (tid < 8) {
      float tot = 0.f;
#pragma unroll
      for (int w4 = 0; w4 < NW; ++w4) tot += ssr[w4][tid];          // fixed order: deterministic
      s_tot[tid] = tot;
    }
    __syncthreads();
    if (ksplit == 1) {
#pragma unroll
      for (int m = 0; m < 8; ++m) rstd[m] = rsqrtf(s_tot[m] / (float)K + a.eps);
    }
  }
#pragma unroll
  for (int cc = 0; cc < NCH; ++cc) {
    const int q = tid + cc * T;
    if (q >= NW * KS * 8) continue;
    const int js = q >> 3, wv = js / spw, j = js - wv * spw;        // block-relative k step -> (wave, step of the wave)
    if (wv >= NW) continue;
    const int cq = (q & 7) >> 1, half = q & 1;
    unsigned char* base = smem + ((size_t)((wv * KS + j) * 64 + cq * 16) * 16 + half * 8);
#pragma unroll
    for (int m = 0; m < 8; ++m) {
      float4 v = xv[m][cc];
      if (rms) {
        const float r = rstd[m];
        v.x *= r; v.y *= r; v.z *= r; v.w *= r;
        if (has_nw) { v.x *= nv[cc].x; v.y *= nv[cc].y; v.z *= nv[cc].z; v.w *= nv[cc].w; }
        if constexpr (MOD_OK) {
          if (has_mod) {
            v.x = v.x * (1.0f + cv[m][cc].x) + sv[m][cc].x; v.y = v.y * (1.0f + cv[m][cc].y) + sv[m][cc].y;
            v.z = v.z * (1.0f + cv[m][cc].z) + sv[m][cc].z; v.w = v.w * (1.0f + cv[m][cc].w) + sv[m][cc].w;
          }
        }
      }
      if (!cval[cc] || m >= mr) v = make_float4(0.f, 0.f, 0.f, 0.f);
      const unsigned h0 = bf_bits(v.x), h1 = bf_bits(v.y), h2 = bf_bits(v.z), h3 = bf_bits(v.w);
      const float l0 = v.x - __uint_as_float(h0 << 16), l1 = v.y - __uint_as_float(h1 << 16);
      const float l2 = v.z - __uint_as_float(h2 << 16), l3 = v.w - __uint_as_float(h3 << 16);
      u32x2 hi, lo;
      hi.x = h0 | (h1 << 16); hi.y = h2 | (h3 << 16);
      lo.x = bf_bits(l0) | (bf_bits(l1) << 16); lo.y = bf_bits(l2) | (bf_bits(l3) << 16);
      *reinterpret_cast<u32x2*>(base + (size_t)m * 16) = hi;               // fragment row m: high parts
      *reinterpret_cast<u32x2*>(base + (size_t)(m + 8) * 16) = lo;         // fragment row m + 8: low parts
    }
  }
  __syncthreads();
  u32x4 af[KS];
#pragma unroll
  for (int j = 0; j < KS; ++j) af[j] = xa[(wave * KS + j) * 64 + lane];

  // ---- the weights meet the fragments: one row group per pass -------------------------------------------------------------------------
  int pp_ = 0;
  while (g < n_groups) {
    f32x4 acc = {0.f, 0.f, 0.f, 0.f}, acc2 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < KS; ++j) {
      if (j < spw && sb + j < steps_total) {
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, af[j]), __builtin_bit_cast(bf16x8, wc[j]), acc, 0, 0, 0);
        if (DUAL) acc2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, af[j]), __builtin_bit_cast(bf16x8, wc2[j]), acc2, 0, 0, 0);
      }
    }
    // accumulator: lane (n, c) holds fragment rows 4 c + i of weight row n
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      red[pp_][wave][0][(4 * c + i) * 16 + n] = acc[i];
      if (DUAL) red[pp_][wave][NM - 1][(4 * c + i) * 16 + n] = acc2[i];
    }
    __syncthreads();
    float s = 0.f, s2 = 0.f;
    if (tid < 128) {
#pragma unroll
      for (int w4 = 0; w4 < NW; ++w4) {
        s += red[pp_][w4][0][tid] + red[pp_][w4][0][tid + 128];               // high-part row + low-part row
        if (DUAL) s2 += red[pp_][w4][NM - 1][tid] + red[pp_][w4][NM - 1][tid + 128];
      }
    }
    float rs = 1.0f;
    if (!PERS && ksplit > 1 && x.atomic) {
      if (tid < 128) {
        const int gn = g * 16 + en;
        if (em < mr && gn < N) {
          float v = s;
          if (a.bias && ks == 0) v += eo.b;
          if (a.gate) v *= eo.g;
          atomicAdd(a.out + (int64_t)em * a.ldo + gn, v);
        }
      }
      return;
    }
    if (!PERS && ksplit > 1 && !(x.dbg & 2)) {
      // partial tile out (write-through), ticket, the last block of the row group folds
      float* pp = x.part + ((int64_t)g * ksplit + ks) * PST;
      if (tid < 128) {
        __hip_atomic_store(pp + tid, s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (DUAL) __hip_atomic_store(pp + 128 + tid, s2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      if (tid < 8) __hip_atomic_store(pp + 128 * NM + tid, rms ? s_tot[tid] : 0.f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (tid == 0) {
        const int tk = __hip_atomic_fetch_add(&x.tickets[g], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s_last = (tk == ksplit - 1);
        if (s_last) __hip_atomic_store(&x.tickets[g], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);     // ready for the next launch
      }
      __syncthreads();
      if (!s_last) return;
      if (tid >= 128) return;
      const float* p0 = x.part + (int64_t)g * ksplit * PST;
      float pv[MAXSPLIT], pv2[DUAL ? MAXSPLIT : 1], pq[MAXSPLIT];
#pragma unroll
      for (int i = 0; i < MAXSPLIT; ++i) {                            // every partial requested before the first is used
        const int ii = i < ksplit ? i : 0;
        pv[i] = __hip_atomic_load(p0 + (int64_t)ii * PST + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (DUAL) pv2[i] = __hip_atomic_load(p0 + (int64_t)ii * PST + 128 + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        pq[i] = __hip_atomic_load(p0 + (int64_t)ii * PST + 128 * NM + em, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      s = 0.f; s2 = 0.f;
      float q2 = 0.f;
#pragma unroll
      for (int i = 0; i < MAXSPLIT; ++i) {
        if (i < ksplit) { s += pv[i]; if (DUAL) s2 += pv2[i]; q2 += pq[i]; }
      }
      if (rms) rs = rsqrtf(q2 / (float)K + a.eps);
    }
    if (tid < 128) {
      const int gn = g * 16 + en;
      if (em < mr && gn < N) {
        float v = s * rs, v2 = s2 * rs;
        if (a.bias) v += eo.b;
        if (a.act == VV_ACT_GELU) v = gelu1(v);
        else if (a.act == VV_ACT_SWIGLU) v = silu1(v) * v2;
        if (a.gate) v *= eo.g;
        if (a.res) v += eo.r;
        a.out[(int64_t)em * a.ldo + gn] = v;
      }
    }
    if constexpr (!PERS) {
      break;
    } else {
      g += gstride;
      pp_ ^= 1;                                        // ping-pong: one barrier per row group is enough
      eo = eo_n;
#pragma unroll
      for (int j = 0; j < KS; ++j) { wc[j] = wn[j]; if (DUAL) wc2[j] = wn2[j]; }
      if (g + gstride < n_groups) {
        eo_n = load_eo(g + gstride);
        issue(wn, wn2, g + gstride);
      }
    }
  }
}

int g_rows_on = 1;          // tuning hook "gemv_rows": 0 = never take this path
int g_rows_blocks = 448;    // row groups x K slices aimed for before the K split stops growing
int g_rows_pers = 192;      // persistent blocks of the whole-row SwiGLU kernels: each block's activation prologue is 150 KB of L2 reads, so FEWER blocks
                            // than CUs win (head: 11.05 us at 144 blocks, 11.65 at 256, 15.2 at 288 one-shot blocks; LLM: 15.7 at 187, 16.9 at 256)
int g_rows_dbg = 0;
int g_rows_atomic = 1;      // tuning hook "gemv_rows_atomic": 0 = always fold K slices through the ticket (deterministic summation order)

template <bool DUAL, int NW, int KS, bool PERS>
int launch_cfg(const vv_lin_args& a, RowsAux x, int n_groups, hipStream_t s) {
  const size_t lds = (size_t)NW * KS * 64 * 16;        // the LDS limit of every instantiation is raised in vv_gemv_rows_init (not capturable)
  int gx = n_groups;
  if (PERS && gx > g_rows_pers) {                       // the same number of row groups for every block
    const int per = (n_groups + g_rows_pers - 1) / g_rows_pers;
    gx = (n_groups + per - 1) / per;
  }
  x.n_groups = n_groups;
  hipLaunchKernelGGL((gemv_rows_kernel<DUAL, NW, KS, PERS>), dim3(gx, x.ksplit), dim3(NW * 64), lds, s, a, x);
  return 1;
}

}  // namespace

void vv_gemv_rows_set_dbg(int d) { g_rows_dbg = d; }
void vv_gemv_rows_set_atomic(int on) { g_rows_atomic = on; }
void vv_gemv_rows_set(int on, int blocks, int pers) {
  if (on >= 0) g_rows_on = on;
  if (blocks > 0) g_rows_blocks = blocks;
  if (pers > 0) g_rows_pers = pers;
}

// floats of partials workspace a launch on (n, k, dual) may need (upper bound over the launcher's rules), and ticket ints
size_t vv_gemv_rows_part_floats(int n, int dual) { return (size_t)((n + 15) / 16) * MAXSPLIT * (dual ? 264 : 136); }
size_t vv_gemv_rows_tickets(int n) { return (size_t)((n + 15) / 16); }

// every kernel's LDS attribute is set before any graph capture (hipFuncSetAttribute is not capturable)
int vv_gemv_rows_init() {
#define VV_ROWS_ATTR(D, NW, KS, P)                                                                                                        \
  if (hipFuncSetAttribute(reinterpret_cast<const void*>(&gemv_rows_kernel<D, NW, KS, P>), hipFuncAttributeMaxDynamicSharedMemorySize,     \
                          NW * KS * 64 * 16) != hipSuccess)                                                                               \
    return vv_set_error(VV_E_HIP, "gemv_rows: cannot raise the dynamic LDS limit");
  VV_ROWS_ATTR(false, 4, 3, false) VV_ROWS_ATTR(false, 4, 6, false) VV_ROWS_ATTR(false, 4, 9, false) VV_ROWS_ATTR(false, 4, 12, false)
  VV_ROWS_ATTR(false, 8, 4, false) VV_ROWS_ATTR(false, 8, 6, false) VV_ROWS_ATTR(false, 8, 8, false)
  VV_ROWS_ATTR(true, 8, 4, false) VV_ROWS_ATTR(true, 8, 6, false) VV_ROWS_ATTR(true, 8, 8, false)
  VV_ROWS_ATTR(true, 8, 4, true) VV_ROWS_ATTR(true, 8, 6, true)
#undef VV_ROWS_ATTR
  return 0;
}

// 1 launched, 0 not covered (the caller falls back), < 0 error.  part / tickets: split-K workspace (tickets zeroed by the caller once; every
// launch leaves them zero) or null (then only shapes that need no K split are taken).
int vv_launch_gemv_rows(const vv_lin_args& a, float* part, size_t part_floats, int* tickets, size_t n_tickets, hipStream_t s) {
  if (!g_rows_on || a.wdt != VV_BF16 || a.m < 3 || a.m > 8 || a.k % 32 || a.k < 32) return 0;
  if (a.pro == VV_PRO_SILU || (a.flags & (VV_LIN_X_BF16 | VV_LIN_OUT_BF16)) || a.ldx == 0) return 0;
  if ((uintptr_t)a.w % 16 || (a.w2 && (uintptr_t)a.w2 % 16) || (uintptr_t)a.x % 16 || a.ldx % 4) return 0;
  if (a.norm_w && (uintptr_t)a.norm_w % 16) return 0;
  if (a.mod_scale && ((uintptr_t)a.mod_scale % 16 || (uintptr_t)a.mod_shift % 16 || a.ld_mod % 4)) return 0;
  if ((a.flags & VV_LIN_W_FRAG) && a.n % 16) return 0;
  const bool dual = a.w2 != nullptr;
  const int steps = a.k / 32, n_groups = (a.n + 15) / 16;
  RowsAux x;
  x.part = part; x.tickets = tickets; x.dbg = g_rows_dbg; x.n_groups = n_groups; x.atomic = 0;
  // whole rows per block when K <= 2048 (8 waves x <= 8 steps): no cross-block reduction at all
  if (steps <= 64) {
    x.ksplit = 1;
    x.spw = (steps + 7) / 8;
    const bool pers = dual && n_groups > g_rows_pers && x.spw <= 6;      // 8 steps x 2 matrices x 2 buffers do not fit the registers
    if (dual) {
      if (pers) {
        if (x.spw <= 4) return launch_cfg<true, 8, 4, true>(a, x, n_groups, s);
        return launch_cfg<true, 8, 6, true>(a, x, n_groups, s);
      }
      if (x.spw <= 4) return launch_cfg<true, 8, 4, false>(a, x, n_groups, s);
      if (x.spw <= 6) return launch_cfg<true, 8, 6, false>(a, x, n_groups, s);
      return launch_cfg<true, 8, 8, false>(a, x, n_groups, s);
    }
    if (x.spw <= 4) return launch_cfg<false, 8, 4, false>(a, x, n_groups, s);
    if (x.spw <= 6) return launch_cfg<false, 8, 6, false>(a, x, n_groups, s);
    return launch_cfg<false, 8, 8, false>(a, x, n_groups, s);
  }
  // long rows: K slices across blocks, folded by the last arriver
  if (a.mod_scale) return 0;                          // the modulated prologue needs the whole row's statistic up front
  const int NW = dual ? 8 : 4, kscap = dual ? 8 : 9, ksmax = dual ? 8 : 12;
  int ksplit = 0, spw = 0;
  for (int sp = 2; sp <= MAXSPLIT && !ksplit; ++sp) {
    const int w = (steps + sp * NW - 1) / (sp * NW);
    if (w <= kscap && ((long)n_groups * sp >= g_rows_blocks || w <= 3)) { ksplit = sp; spw = w; }
  }
  for (int sp = 2; sp <= MAXSPLIT && !ksplit; ++sp) {
    const int w = (steps + sp * NW - 1) / (sp * NW);
    if (w <= ksmax) { ksplit = sp; spw = w; }
  }
  if (!ksplit) return 0;
  while (ksplit > 1 && (ksplit - 1) * NW * spw >= steps) --ksplit;     // drop K slices that would start past the end
  const size_t pst = dual ? 264 : 136;
  x.atomic = g_rows_atomic && !dual && a.pro == VV_PRO_NONE && a.act == VV_ACT_NONE && a.res && a.res == a.out && a.ldres == a.ldo;
  if (!x.atomic && (!part || !tickets || (size_t)n_groups * ksplit * pst > part_floats || (size_t)n_groups > n_tickets)) return 0;
  x.ksplit = ksplit; x.spw = spw;
  if (dual) {
    if (spw <= 4) return launch_cfg<true, 8, 4, false>(a, x, n_groups, s);
    if (spw <= 6) return launch_cfg<true, 8, 6, false>(a, x, n_groups, s);
    return launch_cfg<true, 8, 8, false>(a, x, n_groups, s);
  }
  if (spw <= 3) return launch_cfg<false, 4, 3, false>(a, x, n_groups, s);
  if (spw <= 6) return launch_cfg<false, 4, 6, false>(a, x, n_groups, s);
  if (spw <= 9) return launch_cfg<false, 4, 9, false>(a, x, n_groups, s);
  return launch_cfg<false, 4, 12, false>(a, x, n_groups, s);
}
