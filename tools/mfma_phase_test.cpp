// Phase timing of mfma_linear_kernel on the conv-stage shapes (debug build with -DVV_MFMA_TIMING):
//   hipcc ... -DVV_MFMA_TIMING csrc/*.hip -o tools/bin/libvv_hip.so ; hipcc -DVV_MFMA_TIMING tools/mfma_phase_test.cpp -Ltools/bin -lvv_hip
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include "vv_hip.h"
extern "C" int vv_mfma_debug_times(unsigned long long* out8, int reset);
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(2); } } while (0)
#define VV(x) do { int r_ = (x); if (r_) { printf("vv error %d (%s) line %d\n", r_, vv_last_error(), __LINE__); exit(3); } } while (0)
static uint32_t rng = 7;
static float frand() { rng = rng * 1664525u + 1013904223u; return ((rng >> 8) & 0xffff) / 32768.0f - 1.0f; }
static uint16_t bf16(float f) { uint32_t u; memcpy(&u, &f, 4); u += 0x7fff + ((u >> 16) & 1); return (uint16_t)(u >> 16); }
int main() {
  VV(vv_init());
  hipStream_t s; CK(hipStreamCreate(&s));
  struct Shape { int m, n, k, pro, act, obf, xbf; } shapes[] = {{40, 2048, 512, 1, 1, 1, 0}, {40, 512, 2048, 0, 0, 0, 1}, {200, 1024, 256, 1, 1, 1, 0}, {200, 256, 1024, 0, 0, 0, 1},
                                                              {40, 1280, 1024, 0, 0, 0, 0}, {40, 4608, 1536, 0, 0, 0, 0}, {3200, 32, 128, 0, 0, 0, 0}};
  for (auto sh : shapes) {
    const int nb = 8;
    std::vector<void*> ws(nb);
    for (int i = 0; i < nb; ++i) { std::vector<uint16_t> h((size_t)sh.n * sh.k); for (auto& v : h) v = bf16(frand() / sqrtf((float)sh.k)); CK(hipMalloc(&ws[i], h.size() * 2)); CK(hipMemcpy(ws[i], h.data(), h.size() * 2, hipMemcpyHostToDevice)); }
    std::vector<float> hx((size_t)sh.m * sh.k); for (auto& v : hx) v = frand();
    void* x; 
    if (sh.xbf) { std::vector<uint16_t> hb(hx.size()); for (size_t i = 0; i < hx.size(); ++i) hb[i] = bf16(hx[i]); CK(hipMalloc(&x, hb.size() * 2)); CK(hipMemcpy(x, hb.data(), hb.size() * 2, hipMemcpyHostToDevice)); }
    else { CK(hipMalloc(&x, hx.size() * 4)); CK(hipMemcpy(x, hx.data(), hx.size() * 4, hipMemcpyHostToDevice)); }
    float *nw, *bias, *out; CK(hipMalloc(&nw, sh.k * 4)); CK(hipMalloc(&bias, sh.n * 4)); CK(hipMalloc(&out, (size_t)sh.m * sh.n * 4));
    CK(hipMemset(bias, 0, sh.n * 4)); { std::vector<float> o(sh.k, 1.f); CK(hipMemcpy(nw, o.data(), sh.k * 4, hipMemcpyHostToDevice)); }
    vv_lin_args a; memset(&a, 0, sizeof(a));
    a.x = (const float*)x; a.ldx = sh.k; a.m = sh.m; a.n = sh.n; a.k = sh.k; a.wdt = VV_BF16; a.out = out; a.ldo = sh.n; a.bias = bias;
    if (sh.pro) { a.pro = VV_PRO_RMSNORM; a.norm_w = nw; a.eps = 1e-5f; }
    if (sh.act) a.act = VV_ACT_GELU;
    a.flags = (sh.obf ? VV_LIN_OUT_BF16 : 0) | (sh.xbf ? VV_LIN_X_BF16 : 0);
    for (int i = 0; i < 5; ++i) { a.w = ws[i % nb]; VV(vv_linear(&a, s)); }
    CK(hipStreamSynchronize(s));
    unsigned long long t[8]; vv_mfma_debug_times(t, 1);
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int it = 200;
    CK(hipEventRecord(e0, s));
    for (int i = 0; i < it; ++i) { a.w = ws[i % nb]; VV(vv_linear(&a, s)); }
    CK(hipEventRecord(e1, s)); CK(hipStreamSynchronize(s));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    vv_mfma_debug_times(t, 1);
    printf("m=%4d n=%4d k=%4d pro=%d gelu=%d out_bf16=%d x_bf16=%d: %6.2f us/launch | block(0,0) thread 0 [us]: stats %.2f  stage %.2f  weights+mfma %.2f  combine+epilogue %.2f\n",
           sh.m, sh.n, sh.k, sh.pro, sh.act, sh.obf, sh.xbf, ms * 1e3 / it, t[0] / (double)it / 100, t[1] / (double)it / 100, t[2] / (double)it / 100, t[3] / (double)it / 100);
    fflush(stdout);
  }
  return 0;
}
