// Stand-alone check + timing of the chained diffusion-head kernel against the per-GEMV launch sequence (no torch):
//   hipcc --offload-arch=gfx950 -O2 -Iinclude tools/chain_head_test.cpp -Lvibevoice_rocm_amd -lvv_hip -o gpurun_out/chain_head_test
//   LD_LIBRARY_PATH=vibevoice_rocm_amd gpurun_out/chain_head_test [blocks ...]
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

#include "vv_hip.h"

#ifdef VV_CHAIN_TIMING
extern "C" int vv_chain_debug_times(unsigned long long* out8, int reset);
#endif
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(2); } } while (0)
#define VV(x) do { int r_ = (x); if (r_) { printf("vv error %d (%s) at line %d\n", r_, vv_last_error(), __LINE__); exit(3); } } while (0)

static uint32_t rng = 12345;
static float frand() { rng = rng * 1664525u + 1013904223u; return ((rng >> 8) & 0xffff) / 32768.0f - 1.0f; }
static uint16_t bf16(float f) { uint32_t u; memcpy(&u, &f, 4); u += 0x7fff + ((u >> 16) & 1); return (uint16_t)(u >> 16); }

static void* dev_bf16(size_t n, float scale) {
  std::vector<uint16_t> h(n);
  for (size_t i = 0; i < n; ++i) h[i] = bf16(frand() * scale);
  void* d; CK(hipMalloc(&d, n * 2)); CK(hipMemcpy(d, h.data(), n * 2, hipMemcpyHostToDevice));
  return d;
}
static float* dev_f32(size_t n, float scale, float offset = 0.f) {
  std::vector<float> h(n);
  for (size_t i = 0; i < n; ++i) h[i] = offset + frand() * scale;
  float* d; CK(hipMalloc(&d, n * 4)); CK(hipMemcpy(d, h.data(), n * 4, hipMemcpyHostToDevice));
  return d;
}

int main(int argc, char** argv) {
  const int D = 1536, ffn = 4608, layers = 4, latent = 64, cond_dim = 1536, n_steps = 20;
  VV(vv_init());
  hipStream_t s; CK(hipStreamCreate(&s));
  vv_head_layer L[4];
  for (int l = 0; l < layers; ++l) {
    L[l].norm_w = dev_f32(D, 0.1f, 1.0f);
    L[l].wgate = dev_bf16((size_t)ffn * D, 1.0f / sqrtf((float)D));
    L[l].wup = dev_bf16((size_t)ffn * D, 1.0f / sqrtf((float)D));
    L[l].wdown = dev_bf16((size_t)D * ffn, 1.0f / sqrtf((float)ffn));
    L[l].adaln = dev_bf16((size_t)3 * D * D, 0.5f / sqrtf((float)D));
  }
  vv_head h;
  memset(&h, 0, sizeof(h));
  h.wdt = VV_BF16; h.D = D; h.ffn = ffn; h.layers = layers; h.latent = latent; h.cond_dim = cond_dim; h.eps = 1e-5f;
  h.noisy_proj = dev_bf16((size_t)D * latent, 1.0f / sqrtf((float)latent));
  h.cond_proj = dev_bf16((size_t)D * cond_dim, 1.0f / sqrtf((float)cond_dim));
  h.final_adaln = dev_bf16((size_t)2 * D * D, 0.5f / sqrtf((float)D));
  h.final_linear = dev_bf16((size_t)latent * D, 1.0f / sqrtf((float)D));
  h.layer = L;
  float* cond2 = dev_f32(2 * cond_dim, 1.0f);
  float* noise = dev_f32(latent, 1.0f);
  float* temb = dev_f32((size_t)n_steps * D, 0.5f);
  vv_dpm_coef coef[32];
  for (int i = 0; i < n_steps; ++i) {
    const float t = (i + 1.0f) / (n_steps + 1.0f);
    coef[i].alpha_s = cosf(1.4f * (1 - t)); coef[i].sigma_s = sinf(1.4f * (1 - t));
    coef[i].cx = 0.9f + 0.05f * t; coef[i].cd = -0.2f - 0.1f * t; coef[i].rinv = 1.0f + 0.3f * t; coef[i].order = (i == 0 || i == n_steps - 1) ? 1 : 2;
  }
  const size_t wsb = vv_head_ws_bytes(&h, n_steps);
  void* ws; CK(hipMalloc(&ws, wsb)); CK(hipMemset(ws, 0, wsb));
  float *outA, *outB; CK(hipMalloc(&outA, latent * 4)); CK(hipMalloc(&outB, latent * 4));

  auto run = [&](int flags, float* out, const char* name, int iters) {
    h.flags = flags;
    VV(vv_head_sample(&h, cond2, cond_dim, noise, temb, coef, n_steps, 1.3f, out, ws, nullptr, s));   // eager once
    CK(hipStreamSynchronize(s));
    void* g = nullptr;
    VV(vv_graph_begin(s));
    VV(vv_head_sample(&h, cond2, cond_dim, noise, temb, coef, n_steps, 1.3f, out, ws, nullptr, s));
    VV(vv_graph_end(s, &g));
    for (int i = 0; i < 3; ++i) VV(vv_graph_launch(g, s));
    CK(hipStreamSynchronize(s));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0, s));
    for (int i = 0; i < iters; ++i) VV(vv_graph_launch(g, s));
    CK(hipEventRecord(e1, s));
    CK(hipStreamSynchronize(s));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("%-28s %8.1f us per vv_head_sample (%d solver steps)\n", name, ms * 1e3f / iters, n_steps);
    fflush(stdout);
    VV(vv_graph_destroy(g));
  };
  run(0, outA, "per-GEMV launches (graph)", 20);
  std::vector<float> a(latent), b(latent);
  CK(hipMemcpy(a.data(), outA, latent * 4, hipMemcpyDeviceToHost));
  auto compare = [&](const char* tag) {
    CK(hipMemcpy(b.data(), outB, latent * 4, hipMemcpyDeviceToHost));
    double num = 0, den = 0; int nan = 0;
    for (int i = 0; i < latent; ++i) { if (!(b[i] == b[i])) nan++; num += (double)(a[i] - b[i]) * (a[i] - b[i]); den += (double)a[i] * a[i]; }
    printf("  %s: rel rms vs per-GEMV path %.3e  (nan %d)  a[0..3] = %g %g %g %g | b = %g %g %g %g\n", tag, sqrt(num / (den + 1e-30)), nan,
           a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]);
    fflush(stdout);
  };
  if (argc <= 1) {
    run(VV_HEAD_CHAIN, outB, "chained kernel (default grid)", 20);
    compare("default");
  }
#ifdef VV_CHAIN_TIMING
  for (int mode : {3, 1, 2, 0})
  for (int which : {64 * 3, 64 * 6, 64 * 7, (100 << 16) | 64}) {
    VV(vv_tune("chain_dbg_mode", mode));
    if (which == 64 * 3) printf("dbg_mode %d\n", mode);
    VV(vv_tune("chain_dbg", which));
    unsigned long long t[16];
    vv_chain_debug_times(t, 1);
    h.flags = VV_HEAD_CHAIN;
    const int reps = 10;
    for (int i = 0; i < reps; ++i) VV(vv_head_sample(&h, cond2, cond_dim, noise, temb, coef, n_steps, 1.3f, outB, ws, nullptr, s));
    CK(hipStreamSynchronize(s));
    vv_chain_debug_times(t, 1);
    const double ph = 201.0 * reps;
    if ((which & 0xffff) < 64 * 6)
      printf("block %3d thread %3d stream wave, per phase [us]: describe+issue %.2f  wait-rows %.2f  stream %.2f  store-wait %.2f\n", which >> 16, which & 0xffff,
             t[0] / ph / 100.0, t[1] / ph / 100.0, t[2] / ph / 100.0, t[3] / ph / 100.0);
    else
      printf("block %3d thread %3d helper wave, per phase [us]: fetch(poll) %.2f  flag-publish %.2f  rows->LDS %.2f  wait-stream-waves %.2f\n", which >> 16, which & 0xffff,
             t[4] / ph / 100.0, t[5] / ph / 100.0, t[6] / ph / 100.0, t[7] / ph / 100.0);
  }
#endif
  for (int mode : {1, 2, 3}) {
    VV(vv_tune("chain_dbg_mode", mode));
    char name[64]; snprintf(name, sizeof(name), "chain, dbg_mode %d (garbage)", mode);
    run(VV_HEAD_CHAIN, outB, name, 20);
  }
  VV(vv_tune("chain_dbg_mode", 0));
  for (int i = 1; i < argc; ++i) {
    const int blocks = atoi(argv[i]);
    VV(vv_tune("chain_blocks", blocks));
    char name[64]; snprintf(name, sizeof(name), "chained kernel, %d blocks", blocks);
    CK(hipMemset(outB, 0, latent * 4));
    run(VV_HEAD_CHAIN, outB, name, 20);
    compare(name);
  }
  return 0;
}
