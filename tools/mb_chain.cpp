// Dependent-launch floor on MI355X: chains of graph-captured kernels whose inputs are the previous kernel's outputs.
// hipcc --offload-arch=gfx950 -O3 tools/mb_chain.cpp -o tools/bin/mb_chain
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void k_indep(const float* x, float* y, int n) { int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) y[i] = 2.f * x[i]; }
__global__ void k_dep(const float* in, float* out, int n) { int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) out[i] = in[i] + 1.f; }
// every block reads the whole previous vector (n floats), reduces it, and writes `per` outputs: the activation side of a GEMV
__global__ void k_allread(const float* in, float* out, int n, int per) {
  float s = 0.f;
  for (int i = threadIdx.x * 4; i < n; i += blockDim.x * 4) { float4 v = *(const float4*)(in + i); s += v.x + v.y + v.z + v.w; }
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
  __shared__ float red[16];
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  float t = 0.f; for (int w = 0; w < (int)blockDim.x / 64; ++w) t += red[w];
  if ((int)threadIdx.x < per) out[blockIdx.x * per + threadIdx.x] = t * 1e-6f + 1.f;
}
// same plus a weight stream: block b reads `wbytes` contiguous bytes of W (16 B per lane) and folds them in
__global__ void k_gemvlike(const float* in, float* out, int n, int per, const u32x4* W, size_t w_per_block) {
  const u32x4* wp = W + (size_t)blockIdx.x * w_per_block;
  unsigned acc = 0;
  u32x4 buf[8];
  for (size_t i = threadIdx.x; i < w_per_block; i += blockDim.x * 8) {
#pragma unroll
    for (int u = 0; u < 8; ++u) { size_t j = i + (size_t)u * blockDim.x; buf[u] = j < w_per_block ? __builtin_nontemporal_load(wp + j) : u32x4{0, 0, 0, 0}; }
#pragma unroll
    for (int u = 0; u < 8; ++u) acc += buf[u].x ^ buf[u].y ^ buf[u].z ^ buf[u].w;
  }
  float s = (float)(acc & 1);
  for (int i = threadIdx.x * 4; i < n; i += blockDim.x * 4) { float4 v = *(const float4*)(in + i); s += v.x + v.y + v.z + v.w; }
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
  __shared__ float red[16];
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  float t = 0.f; for (int w = 0; w < (int)blockDim.x / 64; ++w) t += red[w];
  if ((int)threadIdx.x < per) out[blockIdx.x * per + threadIdx.x] = t * 1e-9f + 1.f;
}

template <class F> static int chain(const char* name, int N, hipStream_t s, F enqueue) {
  hipGraph_t g; hipGraphExec_t ge;
  CK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
  for (int i = 0; i < N; ++i) enqueue(i);
  CK(hipStreamEndCapture(s, &g));
  CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
  for (int i = 0; i < 3; ++i) CK(hipGraphLaunch(ge, s));
  CK(hipStreamSynchronize(s));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  CK(hipEventRecord(e0, s));
  const int reps = 10;
  for (int i = 0; i < reps; ++i) CK(hipGraphLaunch(ge, s));
  CK(hipEventRecord(e1, s)); CK(hipStreamSynchronize(s));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  printf("%-44s %7.2f us per kernel\n", name, ms * 1e3 / reps / N);
  return 0;
}

int main() {
  hipStream_t s; CK(hipStreamCreate(&s));
  float *a, *b; CK(hipMalloc(&a, 1 << 20)); CK(hipMalloc(&b, 1 << 20));
  CK(hipMemset(a, 0, 1 << 20)); CK(hipMemset(b, 0, 1 << 20));
  u32x4* W; size_t wbytes = (size_t)512 << 20; CK(hipMalloc(&W, wbytes)); CK(hipMemset(W, 1, wbytes));
  const int N = 400;
  chain("independent, 1 block x 64", N, s, [&](int) { hipLaunchKernelGGL(k_indep, dim3(1), dim3(64), 0, s, a, b, 64); });
  chain("dependent ping-pong, 1 block x 64", N, s, [&](int i) { hipLaunchKernelGGL(k_dep, dim3(1), dim3(64), 0, s, (i & 1) ? b : a, (i & 1) ? a : b, 64); });
  chain("dependent ping-pong, 12 blocks x 256 (3072 el)", N, s, [&](int i) { hipLaunchKernelGGL(k_dep, dim3(12), dim3(256), 0, s, (i & 1) ? b : a, (i & 1) ? a : b, 3072); });
  for (int blocks : {64, 256, 512, 1024})
    for (int wg : {256, 1024}) {
      char nm[96]; snprintf(nm, sizeof nm, "all-read 3072 floats, %d blocks x %d", blocks, wg);
      chain(nm, N, s, [&](int i) { hipLaunchKernelGGL(k_allread, dim3(blocks), dim3(wg), 0, s, (i & 1) ? b : a, (i & 1) ? a : b, 3072, 3072 / blocks > 0 ? 3072 / blocks : 1); });
    }
  // weight-streaming: total MB per kernel from a 512 MB pool walked cyclically by launch index (so ~HBM/MALL mix like real weights)
  for (double mb : {4.7, 14.0, 28.3, 55.0}) {
    for (int blocks : {256, 512, 1024}) {
      size_t per_block = (size_t)(mb * 1e6 / 16 / blocks);
      char nm[96]; snprintf(nm, sizeof nm, "gemv-like %.1f MB (HBM), %d blocks x 256", mb, blocks);
      size_t span = per_block * blocks;
      size_t slots = (wbytes / 16) / span;
      chain(nm, 200, s, [&](int i) { hipLaunchKernelGGL(k_gemvlike, dim3(blocks), dim3(256), 0, s, (i & 1) ? b : a, (i & 1) ? a : b, 3072, 3072 / blocks > 0 ? 3072 / blocks : 1, W + (i % slots) * span, per_block); });
      snprintf(nm, sizeof nm, "gemv-like %.1f MB (cache-resident), %d blocks x 256", mb, blocks);
      size_t rs = (size_t)(160e6 / 16) / span; if (rs < 1) rs = 1;
      chain(nm, 200, s, [&](int i) { hipLaunchKernelGGL(k_gemvlike, dim3(blocks), dim3(256), 0, s, (i & 1) ? b : a, (i & 1) ? a : b, 3072, 3072 / blocks > 0 ? 3072 / blocks : 1, W + (i % rs) * span, per_block); });
    }
  }
  return 0;
}
