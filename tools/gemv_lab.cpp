// gemv_lab — where does a dependent M = 2 SwiGLU GEMV (the head's gate/up: N 4608, K 1536, RMSNorm + adaLN modulate prologue) spend
// its time?  A standalone replica of gemv_stream_kernel<2, true, 1, 3, 1> with s_memtime stamps, and structural variants, run as a
// dependent chain inside a hipGraph like the real frame (tools/mb_chain.cpp gives the synthetic floor: 6.8 us at 28.3 MB).
//   hipcc --offload-arch=gfx950 -O3 tools/gemv_lab.cpp -o tools/bin/gemv_lab
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned short bf16_t;

__device__ __forceinline__ float wsum(float v) {
#define DPP_ADD(ctrl) v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), ctrl, 0xF, 0xF, true))
  DPP_ADD(0xB1); DPP_ADD(0x4E); DPP_ADD(0x141); DPP_ADD(0x140);
#undef DPP_ADD
  const float r0 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 0));
  const float r1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 16));
  const float r2 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 32));
  const float r3 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 48));
  return (r0 + r1) + (r2 + r3);
}
__device__ __forceinline__ void unpack8(const u32x4 v, float (&o)[8]) {
  o[0] = __uint_as_float(v.x << 16); o[1] = __uint_as_float(v.x & 0xffff0000u);
  o[2] = __uint_as_float(v.y << 16); o[3] = __uint_as_float(v.y & 0xffff0000u);
  o[4] = __uint_as_float(v.z << 16); o[5] = __uint_as_float(v.z & 0xffff0000u);
  o[6] = __uint_as_float(v.w << 16); o[7] = __uint_as_float(v.w & 0xffff0000u);
}
__device__ __forceinline__ unsigned long long stamp() {     // constant 100 MHz counter, the same on every XCD: 10 ns resolution
  unsigned long long t;
  asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory");
  return t;
}

struct Args {
  const float* x; long ldx;            // [2, K] fp32
  const float* norm_w; const float* shift; const float* scale; long ld_mod;
  const bf16_t* w; const bf16_t* w2;   // [N, K]
  float* out; long ldo;                // [2, N]
  int N, K; float eps;
  unsigned long long* stamps;          // [blocks][8] (wave 0 lane 0), or null
};

#define FENCE4(v) asm volatile("" : "+v"((v).x), "+v"((v).y), "+v"((v).z), "+v"((v).w))

// VARIANT 0: replica of the library kernel (batched prologue).  1: prologue computed once per block through LDS.
template <int VARIANT, int WAVES>
__global__ __launch_bounds__(WAVES * 64) void dual_kernel(const Args a, const int n_groups) {
  constexpr int M = 2, KU = 3;
  __shared__ float xs[M][KU * 512];
  __shared__ float red[WAVES][M];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int K = a.K, N = a.N;
  const bool st = a.stamps && wave == 0;
  unsigned long long t[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  if (st) t[0] = stamp();
  int koff[KU];
#pragma unroll
  for (int u = 0; u < KU; ++u) koff[u] = u * 512 + lane * 8;
  const int gstride = gridDim.x * WAVES;
  int g = blockIdx.x * WAVES + wave;
  u32x4 cur[KU], cur2[KU], nxt[KU], nxt2[KU];
  auto issue = [&](u32x4 (&b)[KU], u32x4 (&b2)[KU], int grp) {
    const bool live = grp < n_groups;
    const int n = min(grp, N - 1);
#pragma unroll
    for (int u = 0; u < KU; ++u) {
      const long off = live ? (long)n * K + koff[u] : 0;
      b[u] = *reinterpret_cast<const u32x4*>(a.w + off);
      b2[u] = *reinterpret_cast<const u32x4*>(a.w2 + off);
    }
  };
  float xr[M][KU][8];
  if (VARIANT == 0) {
    float4 xa[M][KU], xb[M][KU], na[KU], nb[KU], sa[M][KU], sb[M][KU], ca[M][KU], cb[M][KU];
#pragma unroll
    for (int m = 0; m < M; ++m)
#pragma unroll
      for (int u = 0; u < KU; ++u) {
        xa[m][u] = *reinterpret_cast<const float4*>(a.x + m * a.ldx + koff[u]);
        xb[m][u] = *reinterpret_cast<const float4*>(a.x + m * a.ldx + koff[u] + 4);
      }
#pragma unroll
    for (int u = 0; u < KU; ++u) { na[u] = *reinterpret_cast<const float4*>(a.norm_w + koff[u]); nb[u] = *reinterpret_cast<const float4*>(a.norm_w + koff[u] + 4); }
#pragma unroll
    for (int m = 0; m < M; ++m)
#pragma unroll
      for (int u = 0; u < KU; ++u) {
        sa[m][u] = *reinterpret_cast<const float4*>(a.shift + m * a.ld_mod + koff[u]); sb[m][u] = *reinterpret_cast<const float4*>(a.shift + m * a.ld_mod + koff[u] + 4);
        ca[m][u] = *reinterpret_cast<const float4*>(a.scale + m * a.ld_mod + koff[u]); cb[m][u] = *reinterpret_cast<const float4*>(a.scale + m * a.ld_mod + koff[u] + 4);
      }
    issue(cur, cur2, g);
    issue(nxt, nxt2, g + gstride);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int u = 0; u < KU; ++u) {
#pragma unroll
      for (int m = 0; m < M; ++m) { FENCE4(xa[m][u]); FENCE4(xb[m][u]); FENCE4(sa[m][u]); FENCE4(sb[m][u]); FENCE4(ca[m][u]); FENCE4(cb[m][u]); }
      FENCE4(na[u]); FENCE4(nb[u]);
    }
    if (st) t[1] = stamp();
#pragma unroll
    for (int m = 0; m < M; ++m) {
      float ss = 0.f;
#pragma unroll
      for (int u = 0; u < KU; ++u) {
        const float xv[8] = {xa[m][u].x, xa[m][u].y, xa[m][u].z, xa[m][u].w, xb[m][u].x, xb[m][u].y, xb[m][u].z, xb[m][u].w};
#pragma unroll
        for (int j = 0; j < 8; ++j) { xr[m][u][j] = xv[j]; ss = fmaf(xv[j], xv[j], ss); }
      }
      ss = wsum(ss);
      const float rstd = rsqrtf(ss / (float)K + a.eps);
#pragma unroll
      for (int u = 0; u < KU; ++u) {
        const float nw[8] = {na[u].x, na[u].y, na[u].z, na[u].w, nb[u].x, nb[u].y, nb[u].z, nb[u].w};
        const float sh[8] = {sa[m][u].x, sa[m][u].y, sa[m][u].z, sa[m][u].w, sb[m][u].x, sb[m][u].y, sb[m][u].z, sb[m][u].w};
        const float sc[8] = {ca[m][u].x, ca[m][u].y, ca[m][u].z, ca[m][u].w, cb[m][u].x, cb[m][u].y, cb[m][u].z, cb[m][u].w};
#pragma unroll
        for (int j = 0; j < 8; ++j) xr[m][u][j] = xr[m][u][j] * rstd * nw[j] * (1.0f + sc[j]) + sh[j];
      }
    }
  } else if (VARIANT >= 2) {
    // prologue once per BLOCK, activation-side loads FIRST: thread t owns 4-element chunks t, t + T, ... of each row
    constexpr int T = WAVES * 64;
    constexpr int NCH = (KU * 512 / 4 + T - 1) / T;       // chunks per thread per row
    float4 xv[M][NCH], nv[NCH], sv[M][NCH], cv[M][NCH];
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      const int k = (tid + c * T) * 4;
      const int kk = k < K ? k : 0;
#pragma unroll
      for (int m = 0; m < M; ++m) xv[m][c] = *reinterpret_cast<const float4*>(a.x + m * a.ldx + kk);
      nv[c] = *reinterpret_cast<const float4*>(a.norm_w + kk);
#pragma unroll
      for (int m = 0; m < M; ++m) { sv[m][c] = *reinterpret_cast<const float4*>(a.shift + m * a.ld_mod + kk); cv[m][c] = *reinterpret_cast<const float4*>(a.scale + m * a.ld_mod + kk); }
    }
    if (VARIANT == 2 || VARIANT == 4) issue(cur, cur2, g);
    if (VARIANT == 2) issue(nxt, nxt2, g + gstride);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      FENCE4(nv[c]);
#pragma unroll
      for (int m = 0; m < M; ++m) { FENCE4(xv[m][c]); FENCE4(sv[m][c]); FENCE4(cv[m][c]); }
    }
    float ssm[M];
#pragma unroll
    for (int m = 0; m < M; ++m) {
      float ss = 0.f;
#pragma unroll
      for (int c = 0; c < NCH; ++c) {
        const bool ok = (tid + c * T) * 4 < K;
        const float4 v = xv[m][c];
        ss += ok ? v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w : 0.f;
      }
      ssm[m] = wsum(ss);
    }
    if (VARIANT == 3) issue(cur, cur2, g);
    if (VARIANT == 3 || VARIANT == 4) issue(nxt, nxt2, g + gstride);
    __builtin_amdgcn_sched_barrier(0);
    if (lane == 0) { red[wave][0] = ssm[0]; red[wave][1] = ssm[1]; }
    __syncthreads();
    if (st) t[1] = stamp();
#pragma unroll
    for (int m = 0; m < M; ++m) {
      float tot = 0.f;
#pragma unroll
      for (int w = 0; w < WAVES; ++w) tot += red[w][m];
      const float rstd = rsqrtf(tot / (float)K + a.eps);
#pragma unroll
      for (int c = 0; c < NCH; ++c) {
        const int k = (tid + c * T) * 4;
        if (k < K) {
          float4 v = xv[m][c];
          const float4 nw = nv[c], sh = sv[m][c], sc = cv[m][c];
          v.x = v.x * rstd * nw.x * (1.f + sc.x) + sh.x; v.y = v.y * rstd * nw.y * (1.f + sc.y) + sh.y;
          v.z = v.z * rstd * nw.z * (1.f + sc.z) + sh.z; v.w = v.w * rstd * nw.w * (1.f + sc.w) + sh.w;
          *reinterpret_cast<float4*>(&xs[m][k]) = v;
        }
      }
    }
    __syncthreads();
#pragma unroll
    for (int m = 0; m < M; ++m)
#pragma unroll
      for (int u = 0; u < KU; ++u) {
        const float4 p = *reinterpret_cast<const float4*>(&xs[m][koff[u]]), q = *reinterpret_cast<const float4*>(&xs[m][koff[u] + 4]);
        xr[m][u][0] = p.x; xr[m][u][1] = p.y; xr[m][u][2] = p.z; xr[m][u][3] = p.w; xr[m][u][4] = q.x; xr[m][u][5] = q.y; xr[m][u][6] = q.z; xr[m][u][7] = q.w;
      }
  } else {
    // prologue once per block: thread t owns elements [t*EPT, (t+1)*EPT) of each row
    issue(cur, cur2, g);
    issue(nxt, nxt2, g + gstride);
    constexpr int T = WAVES * 64;
    float ssm[M];
#pragma unroll
    for (int m = 0; m < M; ++m) {
      float ss = 0.f;
      for (int k = tid * 4; k < K; k += T * 4) {
        const float4 v = *reinterpret_cast<const float4*>(a.x + m * a.ldx + k);
        *reinterpret_cast<float4*>(&xs[m][k]) = v;
        ss += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
      }
      ssm[m] = wsum(ss);
    }
    if (lane == 0) { red[wave][0] = ssm[0]; red[wave][1] = ssm[1]; }
    __syncthreads();
    if (st) t[1] = stamp();
#pragma unroll
    for (int m = 0; m < M; ++m) {
      float tot = 0.f;
#pragma unroll
      for (int w = 0; w < WAVES; ++w) tot += red[w][m];
      const float rstd = rsqrtf(tot / (float)K + a.eps);
      for (int k = tid * 4; k < K; k += T * 4) {
        float4 v = *reinterpret_cast<float4*>(&xs[m][k]);
        const float4 nw = *reinterpret_cast<const float4*>(a.norm_w + k);
        const float4 sh = *reinterpret_cast<const float4*>(a.shift + m * a.ld_mod + k);
        const float4 sc = *reinterpret_cast<const float4*>(a.scale + m * a.ld_mod + k);
        v.x = v.x * rstd * nw.x * (1.f + sc.x) + sh.x; v.y = v.y * rstd * nw.y * (1.f + sc.y) + sh.y;
        v.z = v.z * rstd * nw.z * (1.f + sc.z) + sh.z; v.w = v.w * rstd * nw.w * (1.f + sc.w) + sh.w;
        *reinterpret_cast<float4*>(&xs[m][k]) = v;
      }
    }
    __syncthreads();
#pragma unroll
    for (int m = 0; m < M; ++m)
#pragma unroll
      for (int u = 0; u < KU; ++u) {
        const float4 p = *reinterpret_cast<const float4*>(&xs[m][koff[u]]), q = *reinterpret_cast<const float4*>(&xs[m][koff[u] + 4]);
        xr[m][u][0] = p.x; xr[m][u][1] = p.y; xr[m][u][2] = p.z; xr[m][u][3] = p.w; xr[m][u][4] = q.x; xr[m][u][5] = q.y; xr[m][u][6] = q.z; xr[m][u][7] = q.w;
      }
  }
  if (st) t[2] = stamp();
  int round = 0;
  while (g < n_groups) {
    const int gn = g + gstride;
    float acc[M] = {0.f, 0.f}, acc2[M] = {0.f, 0.f};
#pragma unroll
    for (int u = 0; u < KU; ++u) {
      float w[8], w2[8];
      unpack8(cur[u], w); unpack8(cur2[u], w2);
      if (st && u == 0 && round < 3) t[3 + round] = stamp();
#pragma unroll
      for (int m = 0; m < M; ++m)
#pragma unroll
        for (int j = 0; j < 8; ++j) { acc[m] = fmaf(w[j], xr[m][u][j], acc[m]); acc2[m] = fmaf(w2[j], xr[m][u][j], acc2[m]); }
    }
#pragma unroll
    for (int m = 0; m < M; ++m) { acc[m] = wsum(acc[m]); acc2[m] = wsum(acc2[m]); }
    if (lane < M) {
      const float v = lane == 0 ? acc[0] : acc[1], v2 = lane == 0 ? acc2[0] : acc2[1];
      a.out[(long)lane * a.ldo + g] = v / (1.0f + expf(-v)) * v2;
    }
#pragma unroll
    for (int u = 0; u < KU; ++u) { cur[u] = nxt[u]; cur2[u] = nxt2[u]; }
    g = gn;
    if (g + gstride < n_groups) issue(nxt, nxt2, g + gstride);
    ++round;
  }
  if (st) {
    t[6] = stamp();
    if (lane == 0) for (int i = 0; i < 8; ++i) a.stamps[(long)blockIdx.x * 8 + i] = t[i];
  }
}

static float bf2f(bf16_t v) { unsigned u = (unsigned)v << 16; float f; memcpy(&f, &u, 4); return f; }
static bf16_t f2bf(float f) { unsigned u; memcpy(&u, &f, 4); u += 0x7fff + ((u >> 16) & 1); return (bf16_t)(u >> 16); }

template <class F> static double chain(int N, hipStream_t s, F enqueue) {
  hipGraph_t g; hipGraphExec_t ge;
  CK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
  for (int i = 0; i < N; ++i) enqueue(i);
  CK(hipStreamEndCapture(s, &g));
  CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
  for (int i = 0; i < 3; ++i) CK(hipGraphLaunch(ge, s));
  CK(hipStreamSynchronize(s));
  double best = 1e9;
  for (int rep = 0; rep < 5; ++rep) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0, s));
    for (int i = 0; i < 5; ++i) CK(hipGraphLaunch(ge, s));
    CK(hipEventRecord(e1, s)); CK(hipStreamSynchronize(s));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    best = std::min(best, (double)ms * 1e3 / 5 / N);
  }
  CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
  return best;
}

int main(int argc, char** argv) {
  const int N = argc > 1 ? atoi(argv[1]) : 4608, K = 1536, COPIES = argc > 2 ? atoi(argv[2]) : 4, L = 64;
  hipStream_t s; CK(hipStreamCreate(&s));
  std::vector<bf16_t> hw((size_t)N * K), hw2((size_t)N * K);
  srand(1);
  for (auto& v : hw) v = f2bf((rand() / (float)RAND_MAX - 0.5f) * 0.05f);
  for (auto& v : hw2) v = f2bf((rand() / (float)RAND_MAX - 0.5f) * 0.05f);
  std::vector<bf16_t*> W(COPIES), W2(COPIES);
  for (int c = 0; c < COPIES; ++c) {
    CK(hipMalloc(&W[c], hw.size() * 2)); CK(hipMalloc(&W2[c], hw.size() * 2));
    CK(hipMemcpy(W[c], hw.data(), hw.size() * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(W2[c], hw2.data(), hw.size() * 2, hipMemcpyHostToDevice));
  }
  const long ld = N > K ? N : K;
  float *b0, *b1, *nw, *sh, *sc;
  CK(hipMalloc(&b0, 2 * ld * 4)); CK(hipMalloc(&b1, 2 * ld * 4)); CK(hipMalloc(&nw, K * 4)); CK(hipMalloc(&sh, 2 * K * 4)); CK(hipMalloc(&sc, 2 * K * 4));
  std::vector<float> hx(2 * ld), hnw(K), hsh(2 * K), hsc(2 * K);
  for (auto& v : hx) v = rand() / (float)RAND_MAX - 0.5f;
  for (auto& v : hnw) v = 1.f + 0.1f * (rand() / (float)RAND_MAX - 0.5f);
  for (auto& v : hsh) v = 0.1f * (rand() / (float)RAND_MAX - 0.5f);
  for (auto& v : hsc) v = 0.1f * (rand() / (float)RAND_MAX - 0.5f);
  CK(hipMemcpy(b0, hx.data(), hx.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(b1, hx.data(), hx.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(nw, hnw.data(), K * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(sh, hsh.data(), 2 * K * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(sc, hsc.data(), 2 * K * 4, hipMemcpyHostToDevice));
  unsigned long long* stamps; CK(hipMalloc(&stamps, 4096 * 8 * 8));
  // host reference of one launch (x = hx rows)
  std::vector<float> ref(2 * (size_t)N);
  for (int m = 0; m < 2; ++m) {
    double ss = 0; for (int k = 0; k < K; ++k) ss += (double)hx[m * ld + k] * hx[m * ld + k];
    const float rstd = 1.0f / sqrtf((float)(ss / K) + 1e-5f);
    std::vector<float> y(K);
    for (int k = 0; k < K; ++k) y[k] = hx[m * ld + k] * rstd * hnw[k] * (1.f + hsc[m * K + k]) + hsh[m * K + k];
    for (int n = 0; n < N; ++n) {
      double g = 0, u = 0;
      for (int k = 0; k < K; ++k) { g += (double)bf2f(hw[(size_t)n * K + k]) * y[k]; u += (double)bf2f(hw2[(size_t)n * K + k]) * y[k]; }
      ref[(size_t)m * N + n] = (float)(g / (1.0 + exp(-g)) * u);
    }
  }
  auto make = [&](int i, bool st) {
    Args a;
    a.x = (i & 1) ? b1 : b0; a.ldx = ld; a.norm_w = nw; a.shift = sh; a.scale = sc; a.ld_mod = K;
    a.w = W[i % COPIES]; a.w2 = W2[i % COPIES]; a.out = (i & 1) ? b0 : b1; a.ldo = ld; a.N = N; a.K = K; a.eps = 1e-5f;
    a.stamps = st ? stamps : nullptr;
    return a;
  };
  auto check = [&](const char* name) {
    std::vector<float> got(2 * ld);
    CK(hipMemcpy(b0, hx.data(), hx.size() * 4, hipMemcpyHostToDevice));
    return got;
  };
  (void)check;
#define RUN(VAR, WAVES, BLOCKS, NAME)                                                                                     \
  do {                                                                                                                     \
    CK(hipMemcpy(b0, hx.data(), hx.size() * 4, hipMemcpyHostToDevice));                                                    \
    { Args a = make(0, false); hipLaunchKernelGGL((dual_kernel<VAR, WAVES>), dim3(BLOCKS), dim3(WAVES * 64), 0, s, a, N); } \
    CK(hipStreamSynchronize(s));                                                                                           \
    std::vector<float> got(2 * ld); CK(hipMemcpy(got.data(), b1, got.size() * 4, hipMemcpyDeviceToHost));                  \
    double num = 0, den = 0;                                                                                               \
    for (int m = 0; m < 2; ++m) for (int n = 0; n < N; ++n) { double d = got[m * ld + n] - ref[(size_t)m * N + n]; num += d * d; den += (double)ref[(size_t)m * N + n] * ref[(size_t)m * N + n]; } \
    double us = chain(L, s, [&](int i) { Args a = make(i, false); hipLaunchKernelGGL((dual_kernel<VAR, WAVES>), dim3(BLOCKS), dim3(WAVES * 64), 0, s, a, N); }); \
    printf("%-40s blocks %4d x %d waves: %6.2f us/kernel   rel err %.2e\n", NAME, BLOCKS, WAVES, us, sqrt(num / den));     \
  } while (0)
  for (int rep = 0; rep < 2; ++rep) {
    RUN(0, 4, 512, "v0 replica (batched prologue, regs)");
    RUN(0, 4, 384, "v0");
    RUN(0, 4, 768, "v0");
    RUN(0, 4, 1152, "v0");
    RUN(0, 2, 1024, "v0 2 waves");
    RUN(0, 8, 256, "v0 8 waves");
    RUN(2, 4, 512, "v2 x-first block prologue via LDS");
    RUN(4, 3, 512, "v4 balanced 3 waves x 3 rows");
    RUN(4, 3, 768, "v4 balanced 3 waves x 2 rows");
    RUN(4, 6, 256, "v4 balanced 6 waves x 3 rows");
    RUN(4, 4, 384, "v4 balanced 4 waves x 3 rows");
    RUN(4, 4, 576, "v4 balanced 4 waves x 2 rows");
    RUN(4, 9, 256, "v4 balanced 9 waves x 2 rows");
    RUN(2, 3, 512, "v2 balanced 3 waves x 3 rows");
    RUN(2, 3, 768, "v2 balanced 3 waves x 2 rows");
    RUN(2, 6, 256, "v2 balanced 6 waves x 3 rows");
    RUN(2, 9, 256, "v2 balanced 9 waves x 2 rows");
    RUN(0, 3, 512, "v0 balanced 3 waves x 3 rows");
    RUN(0, 3, 768, "v0 balanced 3 waves x 2 rows");
    RUN(3, 4, 512, "v3 weights after x landed");
    RUN(3, 8, 256, "v3 8 waves");
    RUN(3, 4, 768, "v3");
    RUN(4, 4, 512, "v4 group 0 before, group 1 after x");
    RUN(4, 8, 256, "v4 8 waves");
    RUN(4, 4, 768, "v4");
    RUN(2, 4, 384, "v2");
    RUN(2, 4, 768, "v2");
    RUN(2, 8, 256, "v2 8 waves");
    RUN(2, 8, 512, "v2 8 waves");
    RUN(2, 16, 256, "v2 16 waves");
    RUN(2, 2, 1024, "v2 2 waves");
    RUN(1, 4, 512, "v1 prologue once per block via LDS");
    RUN(1, 4, 768, "v1");
    RUN(1, 4, 1152, "v1");
    RUN(1, 8, 256, "v1 8 waves");
    RUN(1, 8, 288, "v1 8 waves");
    RUN(1, 8, 576, "v1 8 waves");
    RUN(1, 16, 256, "v1 16 waves");
  }
  // stamps of one launch in a chain (last launch stamped)
  for (int var = 0; var < 5; ++var) {
    CK(hipMemset(stamps, 0, 4096 * 64));
    for (int i = 0; i < 20; ++i) {
      Args a = make(i, i == 19);
      if (var == 0) hipLaunchKernelGGL((dual_kernel<0, 4>), dim3(512), dim3(256), 0, s, a, N);
      else if (var == 1) hipLaunchKernelGGL((dual_kernel<1, 4>), dim3(512), dim3(256), 0, s, a, N);
      else if (var == 2) hipLaunchKernelGGL((dual_kernel<2, 4>), dim3(512), dim3(256), 0, s, a, N);
      else if (var == 3) hipLaunchKernelGGL((dual_kernel<3, 4>), dim3(512), dim3(256), 0, s, a, N);
      else hipLaunchKernelGGL((dual_kernel<4, 4>), dim3(512), dim3(256), 0, s, a, N);
    }
    CK(hipStreamSynchronize(s));
    std::vector<unsigned long long> hs(512 * 8);
    CK(hipMemcpy(hs.data(), stamps, hs.size() * 8, hipMemcpyDeviceToHost));
    unsigned long long tmin = ~0ull, tmax = 0;
    for (int b = 0; b < 512; ++b) { tmin = std::min(tmin, hs[b * 8]); tmax = std::max(tmax, hs[b * 8 + 6]); }
    printf("variant %d: stamps in us after the first block's entry (s_memrealtime, 10 ns ticks); kernel span %.2f us\n", var, (tmax - tmin) * 0.01);
    const char* names[7] = {"entry", "x landed / stats", "prologue done", "grp0 weights", "grp1 weights", "grp2 weights", "end"};
    for (int i = 0; i < 7; ++i) {
      std::vector<long long> d;
      for (int b = 0; b < 512; ++b) if (hs[b * 8 + i]) d.push_back((long long)(hs[b * 8 + i] - tmin));
      if (d.empty()) continue;
      std::sort(d.begin(), d.end());
      printf("  %-18s min %6.2f  p10 %6.2f  median %6.2f  p90 %6.2f  max %6.2f   (n=%zu)\n", names[i], d.front() * 0.01, d[d.size() / 10] * 0.01, d[d.size() / 2] * 0.01, d[d.size() * 9 / 10] * 0.01, d.back() * 0.01, d.size());
    }
  }
  return 0;
}
