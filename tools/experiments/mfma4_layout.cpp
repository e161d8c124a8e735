// Operand layout of v_mfma_f32_4x4x4_16b_bf16 (16 independent 4x4x4 blocks per wave), checked against a host loop:
//   lane l = 4 b + q:  A operand = A_b[i = q][k = 0..3],  B operand = B_b[k = 0..3][j = q],  D register r = D_b[i = r][j = q]
// hipcc --offload-arch=gfx950 -O2 tools/mfma4_layout.cpp -o tools/bin/mfma4_layout && tools/bin/mfma4_layout
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <string.h>
typedef short s4 __attribute__((ext_vector_type(4)));
typedef float f4 __attribute__((ext_vector_type(4)));
__global__ void k(const s4* a, const s4* b, f4* c) {
  f4 acc = {0, 0, 0, 0};
  acc = __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(a[threadIdx.x], b[threadIdx.x], acc, 0, 0, 0);
  c[threadIdx.x] = acc;
}
static uint16_t bf(float f) { uint32_t u; memcpy(&u, &f, 4); return (uint16_t)(u >> 16); }
int main() {
  float A[16][4][4], B[16][4][4];
  uint16_t ha[64][4], hb[64][4];
  for (int b = 0; b < 16; ++b) for (int i = 0; i < 4; ++i) for (int kk = 0; kk < 4; ++kk) { A[b][i][kk] = (float)((b * 7 + i * 3 + kk) % 13 - 6); B[b][kk][i] = (float)((b * 5 + i * 2 + kk * 3) % 11 - 5); }
  for (int l = 0; l < 64; ++l) for (int kk = 0; kk < 4; ++kk) { ha[l][kk] = bf(A[l / 4][l % 4][kk]); hb[l][kk] = bf(B[l / 4][kk][l % 4]); }
  s4 *da, *db; f4* dc;
  hipMalloc(&da, sizeof(ha)); hipMalloc(&db, sizeof(hb)); hipMalloc(&dc, 64 * sizeof(f4));
  hipMemcpy(da, ha, sizeof(ha), hipMemcpyHostToDevice); hipMemcpy(db, hb, sizeof(hb), hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, da, db, dc);
  float hc[64][4];
  if (hipMemcpy(hc, dc, sizeof(hc), hipMemcpyDeviceToHost) != hipSuccess) { printf("HIP error\n"); return 2; }
  int bad = 0;
  for (int l = 0; l < 64; ++l) for (int r = 0; r < 4; ++r) {
    float want = 0; for (int kk = 0; kk < 4; ++kk) want += A[l / 4][r][kk] * B[l / 4][kk][l % 4];
    if (want != hc[l][r]) { if (bad < 8) printf("lane %d reg %d: got %g want %g\n", l, r, hc[l][r], want); ++bad; }
  }
  printf(bad ? "LAYOUT MISMATCH (%d)\n" : "layout ok: lane 4b+q holds A_b[q][k], B_b[k][q], D reg r = D_b[r][q]\n", bad);
  return bad ? 1 : 0;
}
