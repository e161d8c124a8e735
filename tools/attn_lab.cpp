// attn_lab — timeline of the fused decode attention (RoPE + KV append + GQA attention, vv_kernels.hip::attn_fused_kernel) inside a
// dependent chain [qkv-like producer -> attention], with s_memrealtime stamps.  hipcc --offload-arch=gfx950 -O3 tools/attn_lab.cpp -o tools/bin/attn_lab
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
typedef unsigned short bf16_t;
typedef unsigned int att_raw __attribute__((ext_vector_type(4)));
__device__ __forceinline__ unsigned long long stamp() { unsigned long long t; asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory"); return t; }
__device__ __forceinline__ void unpack8(const att_raw v, float (&o)[8]) {
  o[0] = __uint_as_float(v.x << 16); o[1] = __uint_as_float(v.x & 0xffff0000u); o[2] = __uint_as_float(v.y << 16); o[3] = __uint_as_float(v.y & 0xffff0000u);
  o[4] = __uint_as_float(v.z << 16); o[5] = __uint_as_float(v.z & 0xffff0000u); o[6] = __uint_as_float(v.w << 16); o[7] = __uint_as_float(v.w & 0xffff0000u);
}
__device__ __forceinline__ float gsum16(float v) {
#define DPP_ADD(ctrl) v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), ctrl, 0xF, 0xF, true))
  DPP_ADD(0xB1); DPP_ADD(0x4E); DPP_ADD(0x141); DPP_ADD(0x140);
#undef DPP_ADD
  return v;
}
// producer: writes qkv rows (dependent input of the attention), like the qkv GEMV's epilogue
__global__ void producer(const float* in, float* qkv, int n) { int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) qkv[i] = in[i] * 0.999f + 0.001f; }

#define UNR 4
// replica of attn_fused_kernel<bf16, 8, 16> (head_dim 128), NW waves per block
template <int NW>
__global__ __launch_bounds__(NW * 64) void attn_fused(const float* qkv, long ld, int heads, int kv_heads, int s_max, const bf16_t* kcache, const bf16_t* vcache,
                                                     const float2* rope, const int* lens, float* out, long ldo, unsigned long long* stamps) {
  extern __shared__ float sm[];
  constexpr int d = 128, half = 64, G = 16, KPW = 4, EPL = 8;
  const int r = blockIdx.y, h = blockIdx.x;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const bool st = stamps && tid == 0;
  unsigned long long t[8] = {0};
  if (st) t[0] = stamp();
  const int gl = lane % G, gi = lane / G;
  const int pos = lens[r];
  const int gsz = heads / kv_heads, kvh = h / gsz;
  const long base = (((long)r * kv_heads + kvh) * s_max) * d;
  const bf16_t* kc = kcache + base; const bf16_t* vc = vcache + base;
  const float scale = rsqrtf((float)d);
  const float* row = qkv + (long)r * ld;
  const int e0 = gl * EPL;
  const int stride = NW * KPW * UNR;
  att_raw kraw[2][UNR], vraw[2][UNR];
  auto issue_kv = [&](int buf, int s0) {
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      const int sidx = s0 + u * KPW + gi; const int sc = sidx < pos ? sidx : 0;
      kraw[buf][u] = *reinterpret_cast<const att_raw*>(kc + (long)sc * d + e0);
      vraw[buf][u] = *reinterpret_cast<const att_raw*>(vc + (long)sc * d + e0);
    }
  };
  if (st) t[1] = stamp();          // pos known
  const int s_first = (wave * UNR) * KPW;
  if (s_first < pos) issue_kv(0, s_first);
  const bool lo = e0 < half; const int pe0 = lo ? e0 + half : e0 - half;
  float q[EPL], kn[EPL], vn[EPL];
  {
    const float* qp = row + h * d; const float* kp = row + (heads + kvh) * d; const float* vp = row + (heads + kv_heads + kvh) * d;
#pragma unroll
    for (int j = 0; j < EPL; ++j) {
      const int fi = (lo ? e0 : pe0) + j; const float2 cs = rope[(long)r * half + fi];
      const float qa = qp[e0 + j], qb = qp[pe0 + j], ka = kp[e0 + j], kb = kp[pe0 + j];
      q[j] = (lo ? qa * cs.x - qb * cs.y : qa * cs.x + qb * cs.y) * scale;
      kn[j] = lo ? ka * cs.x - kb * cs.y : ka * cs.x + kb * cs.y;
      vn[j] = vp[e0 + j];
    }
  }
  if (st) { asm volatile("" :: "v"(q[0]), "v"(kn[0])); t[2] = stamp(); }     // q / rope landed
  float mmax = -INFINITY, lsum = 0.f, acc[EPL];
#pragma unroll
  for (int j = 0; j < EPL; ++j) acc[j] = 0.f;
  auto update = [&](float dot, const float (&vx)[EPL]) {
    const float mn = fmaxf(mmax, dot); const float corr = expf(mmax - mn); const float p = expf(dot - mn);
    lsum = lsum * corr + p;
#pragma unroll
    for (int j = 0; j < EPL; ++j) acc[j] = fmaf(p, vx[j], acc[j] * corr);
    mmax = mn;
  };
  auto consume = [&](int buf, int s0) {
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      float kx[EPL], vx[EPL]; unpack8(kraw[buf][u], kx); unpack8(vraw[buf][u], vx);
      float dot = 0.f;
#pragma unroll
      for (int j = 0; j < EPL; ++j) dot = fmaf(q[j], kx[j], dot);
      dot = gsum16(dot);
      if (s0 + u * KPW + gi < pos) update(dot, vx);
    }
  };
  for (int s0 = s_first; s0 < pos; s0 += 2 * stride) {
    if (s0 + stride < pos) issue_kv(1, s0 + stride);
    consume(0, s0);
    if (s0 + stride >= pos) break;
    if (s0 + 2 * stride < pos) issue_kv(0, s0 + 2 * stride);
    consume(1, s0 + stride);
  }
  { float dot = 0.f;
#pragma unroll
    for (int j = 0; j < EPL; ++j) dot = fmaf(q[j], kn[j], dot);
    dot = gsum16(dot);
    if (wave == 0 && gi == 0) update(dot, vn); }
  if (st) { asm volatile("" :: "v"(acc[0])); t[3] = stamp(); }               // keys consumed
  for (int o = G; o < 64; o <<= 1) {
    const float m2 = __shfl_xor(mmax, o), l2 = __shfl_xor(lsum, o);
    const float mn = fmaxf(mmax, m2);
    const float c1 = (mmax == -INFINITY) ? 0.f : expf(mmax - mn), c2 = (m2 == -INFINITY) ? 0.f : expf(m2 - mn);
    lsum = lsum * c1 + l2 * c2;
#pragma unroll
    for (int j = 0; j < EPL; ++j) { const float a2 = __shfl_xor(acc[j], o); acc[j] = acc[j] * c1 + a2 * c2; }
    mmax = mn;
  }
  float* rec = sm + (long)wave * (d + 2);
  if (gi == 0) { if (gl == 0) { rec[0] = mmax; rec[1] = lsum; }
#pragma unroll
    for (int j = 0; j < EPL; ++j) rec[2 + e0 + j] = acc[j]; }
  __syncthreads();
  if (st) t[4] = stamp();                                                     // wave merge + barrier
  for (int i = tid; i < d; i += blockDim.x) {
    float M = -INFINITY;
    for (int gg = 0; gg < NW; ++gg) M = fmaxf(M, sm[(long)gg * (d + 2)]);
    float num = 0.f, den = 0.f;
    for (int gg = 0; gg < NW; ++gg) { const float* rr = sm + (long)gg * (d + 2); const float wgt = (rr[0] == -INFINITY) ? 0.f : expf(rr[0] - M); den = fmaf(rr[1], wgt, den); num = fmaf(rr[2 + i], wgt, num); }
    out[(long)r * ldo + h * d + i] = num / den;
  }
  if (st) { t[5] = stamp(); for (int i = 0; i < 8; ++i) stamps[((long)blockIdx.y * gridDim.x + blockIdx.x) * 8 + i] = t[i]; }
}


// ---- v2: batched online softmax in the log2 domain, vectorised q / k / v / RoPE loads, one LDS merge for all lane groups, optional split over keys
__device__ __forceinline__ float ex2(float x) { return __builtin_amdgcn_exp2f(x); }
template <int NW>
__global__ __launch_bounds__(NW * 64) void attn2(const float* qkv, long ld, int heads, int kv_heads, int s_max, bf16_t* kcache, bf16_t* vcache,
                                                const float2* rope, const int* lens, float* out, long ldo, float* part, int* tickets, unsigned long long* stamps) {
  constexpr int d = 128, half = 64, G = 16, NG = NW * 4, EPL = 8;
  __shared__ __attribute__((aligned(16))) float sacc[NG][d];
  __shared__ float sm_[NG], sl_[NG], sw_[NG];
  __shared__ float sML[2];
  __shared__ int s_last;
  const int r = blockIdx.y, h = blockIdx.x, split = blockIdx.z, nsplit = gridDim.z;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const bool st = stamps && tid == 0 && split == 0;
  unsigned long long t[8] = {0};
  if (st) t[0] = stamp();
  const int gl = lane & 15, gi = lane >> 4, grp = wave * 4 + gi;
  const int e0 = gl * EPL;
  const bool lo = e0 < half; const int pe0 = lo ? e0 + half : e0 - half;
  // q / new k / new v / RoPE: 16-byte loads, issued before anything depends on lens
  const float* row = qkv + (long)r * ld;
  const int gsz = heads / kv_heads, kvh = h / gsz;
  const float* qp = row + h * d; const float* kp = row + (heads + kvh) * d; const float* vp = row + (heads + kv_heads + kvh) * d;
  const float4 qa0 = *reinterpret_cast<const float4*>(qp + e0), qa1 = *reinterpret_cast<const float4*>(qp + e0 + 4);
  const float4 qb0 = *reinterpret_cast<const float4*>(qp + pe0), qb1 = *reinterpret_cast<const float4*>(qp + pe0 + 4);
  const float4 ka0 = *reinterpret_cast<const float4*>(kp + e0), ka1 = *reinterpret_cast<const float4*>(kp + e0 + 4);
  const float4 kb0 = *reinterpret_cast<const float4*>(kp + pe0), kb1 = *reinterpret_cast<const float4*>(kp + pe0 + 4);
  const float4 vn0 = *reinterpret_cast<const float4*>(vp + e0), vn1 = *reinterpret_cast<const float4*>(vp + e0 + 4);
  const float2* rp = rope + (long)r * half + (lo ? e0 : pe0);
  const float4 r0 = *reinterpret_cast<const float4*>(rp), r1 = *reinterpret_cast<const float4*>(rp + 2), r2 = *reinterpret_cast<const float4*>(rp + 4), r3 = *reinterpret_cast<const float4*>(rp + 6);
  const int pos = lens[r];
  const long base = (((long)r * kv_heads + kvh) * s_max) * d;
  bf16_t* kc = kcache + base; bf16_t* vc = vcache + base;
  // this block's key range [ks, ke) of the pos cached keys
  const int per = (pos + nsplit - 1) / nsplit;
  const int ks = split * per, ke = min(pos, ks + per);
  att_raw kraw[2][UNR], vraw[2][UNR];
  auto issue_kv = [&](int buf, int s0) {
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      const int sidx = s0 + u * NG; const int sc = sidx < ke ? sidx : 0;
      kraw[buf][u] = *reinterpret_cast<const att_raw*>(kc + (long)sc * d + e0);
      vraw[buf][u] = *reinterpret_cast<const att_raw*>(vc + (long)sc * d + e0);
    }
  };
  const int s_first = ks + grp;
  if (s_first < ke) issue_kv(0, s_first);
  if (st) t[1] = stamp();
  const float qsc = rsqrtf((float)d) * 1.4426950408889634f;       // scores in the log2 domain
  const float cs[8] = {r0.x, r0.z, r1.x, r1.z, r2.x, r2.z, r3.x, r3.z}, sn[8] = {r0.y, r0.w, r1.y, r1.w, r2.y, r2.w, r3.y, r3.w};
  const float qa[8] = {qa0.x, qa0.y, qa0.z, qa0.w, qa1.x, qa1.y, qa1.z, qa1.w}, qb[8] = {qb0.x, qb0.y, qb0.z, qb0.w, qb1.x, qb1.y, qb1.z, qb1.w};
  const float ka[8] = {ka0.x, ka0.y, ka0.z, ka0.w, ka1.x, ka1.y, ka1.z, ka1.w}, kb[8] = {kb0.x, kb0.y, kb0.z, kb0.w, kb1.x, kb1.y, kb1.z, kb1.w};
  const float vn[8] = {vn0.x, vn0.y, vn0.z, vn0.w, vn1.x, vn1.y, vn1.z, vn1.w};
  float q[EPL], kn[EPL];
#pragma unroll
  for (int j = 0; j < EPL; ++j) {
    q[j] = (lo ? qa[j] * cs[j] - qb[j] * sn[j] : qa[j] * cs[j] + qb[j] * sn[j]) * qsc;
    kn[j] = lo ? ka[j] * cs[j] - kb[j] * sn[j] : ka[j] * cs[j] + kb[j] * sn[j];
  }
  if (st) { asm volatile("" :: "v"(q[0]), "v"(kn[0])); t[2] = stamp(); }
  float mmax = -INFINITY, lsum = 0.f, acc[EPL];
#pragma unroll
  for (int j = 0; j < EPL; ++j) acc[j] = 0.f;
  auto consume = [&](int buf, int s0) {
    float dot[UNR], vx[UNR][EPL];
    float bm = -INFINITY;
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      float kx[EPL]; unpack8(kraw[buf][u], kx); unpack8(vraw[buf][u], vx[u]);
      float dd = 0.f;
#pragma unroll
      for (int j = 0; j < EPL; ++j) dd = fmaf(q[j], kx[j], dd);
      dd = gsum16(dd);
      dot[u] = (s0 + u * NG < ke) ? dd : -INFINITY;
      bm = fmaxf(bm, dot[u]);
    }
    if (bm == -INFINITY) return;          // group-uniform: the batch holds no key
    const float mn = fmaxf(mmax, bm);
    const float corr = ex2(mmax - mn);    // exp2(-inf) = 0 on the first batch
    float ps = 0.f;
#pragma unroll
    for (int j = 0; j < EPL; ++j) acc[j] *= corr;
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      const float p = ex2(dot[u] - mn);
      ps += p;
#pragma unroll
      for (int j = 0; j < EPL; ++j) acc[j] = fmaf(p, vx[u][j], acc[j]);
    }
    lsum = lsum * corr + ps;
    mmax = mn;
  };
  const int stride = NG * UNR;
  for (int s0 = s_first; s0 < ke; s0 += 2 * stride) {
    if (s0 + stride < ke) issue_kv(1, s0 + stride);
    consume(0, s0);
    if (s0 + stride >= ke) break;
    if (s0 + 2 * stride < ke) issue_kv(0, s0 + 2 * stride);
    consume(1, s0 + stride);
  }
  if (split == 0 && grp == 0) {            // the new token (its cache slot may not be written yet by the owner block)
    float dd = 0.f;
#pragma unroll
    for (int j = 0; j < EPL; ++j) dd = fmaf(q[j], kn[j], dd);
    dd = gsum16(dd);
    const float mn = fmaxf(mmax, dd), corr = ex2(mmax - mn), p = ex2(dd - mn);
    lsum = lsum * corr + p;
#pragma unroll
    for (int j = 0; j < EPL; ++j) acc[j] = fmaf(p, vn[j], acc[j] * corr);
    mmax = mn;
    if (h % gsz == 0) {                    // one writer per (row, kv head): append k, v at slot pos
      unsigned pk[4], pv[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        auto bf = [](float f) { unsigned u = __float_as_uint(f); return (u + 0x7fffu + ((u >> 16) & 1u)) >> 16; };
        pk[j] = bf(kn[2 * j]) | (bf(kn[2 * j + 1]) << 16); pv[j] = bf(vn[2 * j]) | (bf(vn[2 * j + 1]) << 16);
      }
      *reinterpret_cast<att_raw*>(kc + (long)pos * d + e0) = att_raw{pk[0], pk[1], pk[2], pk[3]};
      *reinterpret_cast<att_raw*>(vc + (long)pos * d + e0) = att_raw{pv[0], pv[1], pv[2], pv[3]};
    }
  }
  if (st) { asm volatile("" :: "v"(acc[0])); t[3] = stamp(); }
  // one merge for all lane groups of the block
  if (gl == 0) { sm_[grp] = mmax; sl_[grp] = lsum; }
  *reinterpret_cast<float4*>(&sacc[grp][e0]) = make_float4(acc[0], acc[1], acc[2], acc[3]);
  *reinterpret_cast<float4*>(&sacc[grp][e0 + 4]) = make_float4(acc[4], acc[5], acc[6], acc[7]);
  __syncthreads();
  if (wave == 0) {
    const float mg = lane < NG ? sm_[lane] : -INFINITY;
    float M = mg;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) M = fmaxf(M, __shfl_xor(M, o));
    const float w = (mg == -INFINITY) ? 0.f : ex2(mg - M);
    float Ls = lane < NG ? w * sl_[lane] : 0.f;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) Ls += __shfl_xor(Ls, o);
    if (lane < NG) sw_[lane] = w;
    if (lane == 0) { sML[0] = M; sML[1] = Ls; }
  }
  __syncthreads();
  if (st) t[4] = stamp();
  float num = 0.f;
  if (tid < d) {
#pragma unroll 8
    for (int gg = 0; gg < NG; ++gg) num = fmaf(sw_[gg], sacc[gg][tid], num);
  }
  if (nsplit == 1) {
    if (tid < d) out[(long)r * ldo + h * d + tid] = num / sML[1];
  } else {
    float* pp = part + (((long)r * heads + h) * nsplit + split) * (d + 2);
    if (tid < d) pp[2 + tid] = num;
    if (tid == 0) { pp[0] = sML[0]; pp[1] = sML[1]; }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      const int tk = __hip_atomic_fetch_add(&tickets[r * heads + h], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      s_last = (tk == nsplit - 1);
      if (s_last) { __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent"); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); tickets[r * heads + h] = 0; }
    }
    __syncthreads();
    if (s_last && tid < d) {
      const float* p0 = part + (((long)r * heads + h) * nsplit) * (d + 2);
      float M = -INFINITY;
      for (int sgi = 0; sgi < nsplit; ++sgi) M = fmaxf(M, p0[sgi * (d + 2)]);
      float nn = 0.f, dn = 0.f;
      for (int sgi = 0; sgi < nsplit; ++sgi) {
        const float* ps = p0 + sgi * (d + 2);
        const float w = (ps[0] == -INFINITY) ? 0.f : ex2(ps[0] - M);
        dn = fmaf(w, ps[1], dn); nn = fmaf(w, ps[2 + tid], nn);
      }
      out[(long)r * ldo + h * d + tid] = nn / dn;
    }
  }
  if (st) { t[5] = stamp(); for (int i = 0; i < 8; ++i) stamps[((long)blockIdx.y * gridDim.x + blockIdx.x) * 8 + i] = t[i]; }
}

int main(int argc, char** argv) {
  const int S = argc > 1 ? atoi(argv[1]) : 440, heads = 12, kvh = 2, d = 128, R = 2, s_max = 4096, L = 28;
  hipStream_t s; CK(hipStreamCreate(&s));
  const long ld = (heads + 2 * kvh) * d;
  float *in, *qkv, *out, *rope; int* lens; bf16_t *kc, *vc; unsigned long long* stamps;
  CK(hipMalloc(&in, R * ld * 4)); CK(hipMalloc(&qkv, R * ld * 4)); CK(hipMalloc(&out, R * heads * d * 4)); CK(hipMalloc(&rope, R * 64 * 8)); CK(hipMalloc(&lens, 8));
  const size_t kvn = (size_t)L * R * kvh * s_max * d;
  CK(hipMalloc(&kc, kvn * 2)); CK(hipMalloc(&vc, kvn * 2)); CK(hipMalloc(&stamps, 64 * 8 * 8));
  std::vector<float> h(R * ld); for (auto& v : h) v = rand() / (float)RAND_MAX - 0.5f;
  CK(hipMemcpy(in, h.data(), h.size() * 4, hipMemcpyHostToDevice));
  std::vector<unsigned short> hk(kvn); for (auto& v : hk) v = 0x3c00 + (rand() & 0xff);
  CK(hipMemcpy(kc, hk.data(), kvn * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(vc, hk.data(), kvn * 2, hipMemcpyHostToDevice));
  std::vector<float> hr(R * 128, 0.7f); CK(hipMemcpy(rope, hr.data(), hr.size() * 4, hipMemcpyHostToDevice));
  int hl[2] = {S, S / 4}; CK(hipMemcpy(lens, hl, 8, hipMemcpyHostToDevice));
  float* part; int* tickets; CK(hipMalloc(&part, R * heads * 16 * 130 * 4)); CK(hipMalloc(&tickets, R * heads * 4)); CK(hipMemset(tickets, 0, R * heads * 4));
  float* out2; CK(hipMalloc(&out2, R * heads * d * 4));
  auto run2 = [&](int nw, int nsplit, bool stamped, float* o) {
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
    for (int l = 0; l < L; ++l) {
      hipLaunchKernelGGL(producer, dim3((R * ld + 255) / 256), dim3(256), 0, s, l ? o : in, qkv, (int)(R * ld));
      bf16_t* k = kc + (size_t)l * R * kvh * s_max * d; bf16_t* v = vc + (size_t)l * R * kvh * s_max * d;
      unsigned long long* sp = (stamped && l == L - 1) ? stamps : nullptr;
      if (nw == 8) hipLaunchKernelGGL((attn2<8>), dim3(heads, R, nsplit), dim3(512), 0, s, qkv, ld, heads, kvh, s_max, k, v, (const float2*)rope, lens, o, (long)heads * d, part, tickets, sp);
      else if (nw == 4) hipLaunchKernelGGL((attn2<4>), dim3(heads, R, nsplit), dim3(256), 0, s, qkv, ld, heads, kvh, s_max, k, v, (const float2*)rope, lens, o, (long)heads * d, part, tickets, sp);
      else hipLaunchKernelGGL((attn2<16>), dim3(heads, R, nsplit), dim3(1024), 0, s, qkv, ld, heads, kvh, s_max, k, v, (const float2*)rope, lens, o, (long)heads * d, part, tickets, sp);
    }
    CK(hipStreamEndCapture(s, &g)); CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    for (int i = 0; i < 3; ++i) CK(hipGraphLaunch(ge, s));
    CK(hipStreamSynchronize(s));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    double best = 1e9;
    for (int rep = 0; rep < 5; ++rep) { CK(hipEventRecord(e0, s)); for (int i = 0; i < 5; ++i) CK(hipGraphLaunch(ge, s)); CK(hipEventRecord(e1, s)); CK(hipStreamSynchronize(s));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1)); best = std::min(best, (double)ms * 1e3 / 5 / L); }
    CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
    return best;
  };
  auto run = [&](int nw, bool with_attn, bool stamped) {
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
    for (int l = 0; l < L; ++l) {
      hipLaunchKernelGGL(producer, dim3((R * ld + 255) / 256), dim3(256), 0, s, l ? out : in, qkv, (int)(R * ld));
      if (!with_attn) continue;
      const bf16_t* k = kc + (size_t)l * R * kvh * s_max * d; const bf16_t* v = vc + (size_t)l * R * kvh * s_max * d;
      unsigned long long* sp = (stamped && l == L - 1) ? stamps : nullptr;
      const size_t lds = (size_t)nw * (d + 2) * 4;
      if (nw == 16) hipLaunchKernelGGL((attn_fused<16>), dim3(heads, R), dim3(1024), lds, s, qkv, ld, heads, kvh, s_max, k, v, (const float2*)rope, lens, out, (long)heads * d, sp);
      else if (nw == 8) hipLaunchKernelGGL((attn_fused<8>), dim3(heads, R), dim3(512), lds, s, qkv, ld, heads, kvh, s_max, k, v, (const float2*)rope, lens, out, (long)heads * d, sp);
      else hipLaunchKernelGGL((attn_fused<4>), dim3(heads, R), dim3(256), lds, s, qkv, ld, heads, kvh, s_max, k, v, (const float2*)rope, lens, out, (long)heads * d, sp);
    }
    CK(hipStreamEndCapture(s, &g)); CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    for (int i = 0; i < 3; ++i) CK(hipGraphLaunch(ge, s));
    CK(hipStreamSynchronize(s));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    double best = 1e9;
    for (int rep = 0; rep < 5; ++rep) { CK(hipEventRecord(e0, s)); for (int i = 0; i < 5; ++i) CK(hipGraphLaunch(ge, s)); CK(hipEventRecord(e1, s)); CK(hipStreamSynchronize(s));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1)); best = std::min(best, (double)ms * 1e3 / 5 / L); }
    CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
    return best;
  };
  const double base = run(16, false, false);
  printf("S=%d: producer alone %.2f us per layer\n", S, base);
  for (int nw : {16, 8, 4}) printf("  producer + attention (%2d waves): %.2f us per layer -> attention %.2f us\n", nw, run(nw, true, false), run(nw, true, false) - base);
  // correctness of v2 against the replica on the same (single-layer) inputs
  {
    const size_t lds = (size_t)8 * (d + 2) * 4;
    CK(hipMemcpy(qkv, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    hipLaunchKernelGGL((attn_fused<8>), dim3(heads, R), dim3(512), lds, s, qkv, ld, heads, kvh, s_max, kc, vc, (const float2*)rope, lens, out, (long)heads * d, (unsigned long long*)nullptr);
    CK(hipStreamSynchronize(s));
    std::vector<float> a(R * heads * d), b(R * heads * d);
    CK(hipMemcpy(a.data(), out, a.size() * 4, hipMemcpyDeviceToHost));
    for (int nsplit : {1, 3}) for (int nw : {4, 8}) {
      if (nw == 8) hipLaunchKernelGGL((attn2<8>), dim3(heads, R, nsplit), dim3(512), 0, s, qkv, ld, heads, kvh, s_max, kc, vc, (const float2*)rope, lens, out2, (long)heads * d, part, tickets, (unsigned long long*)nullptr);
      else hipLaunchKernelGGL((attn2<4>), dim3(heads, R, nsplit), dim3(256), 0, s, qkv, ld, heads, kvh, s_max, kc, vc, (const float2*)rope, lens, out2, (long)heads * d, part, tickets, (unsigned long long*)nullptr);
      CK(hipStreamSynchronize(s));
      CK(hipMemcpy(b.data(), out2, b.size() * 4, hipMemcpyDeviceToHost));
      double num = 0, den = 0; for (size_t i = 0; i < a.size(); ++i) { num += (double)(a[i] - b[i]) * (a[i] - b[i]); den += (double)a[i] * a[i]; }
      printf("  v2 (%d waves, %d splits) vs replica: rel err %.2e\n", nw, nsplit, sqrt(num / den));
    }
  }
  for (int nsplit : {1, 2, 4, 8}) for (int nw : {4, 8, 16}) printf("  v2 %2d waves x %d splits: attention %.2f us\n", nw, nsplit, run2(nw, nsplit, false, out2) - base);
  for (int var = 0; var < 2; ++var) {
  if (var == 0) run(16, true, true); else run2(8, 1, true, out2);
  std::vector<unsigned long long> hs(24 * 8); CK(hipMemcpy(hs.data(), stamps, hs.size() * 8, hipMemcpyDeviceToHost));
  unsigned long long t0 = ~0ull; for (int b = 0; b < 24; ++b) t0 = std::min(t0, hs[b * 8]);
  const char* nm[6] = {"entry", "pos known", "q / rope landed", "keys consumed", "merged + barrier", "end"};
  for (int i = 0; i < 6; ++i) { std::vector<double> dd; for (int b = 0; b < 24; ++b) dd.push_back((hs[b * 8 + i] - t0) * 0.01); std::sort(dd.begin(), dd.end());
    printf("  %-18s min %5.2f  median %5.2f  max %5.2f us\n", nm[i], dd.front(), dd[12], dd.back()); }
  }
  return 0;
}
