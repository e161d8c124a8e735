// internal helpers shared by the .hip translation units (not part of the C ABI)
#ifndef VV_COMMON_H
#define VV_COMMON_H
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdint.h>
#include "vv_hip.h"

int vv_set_error(int code, const char* fmt, ...);

#define VV_CHECK_LAUNCH(name)                                                                    \
  do {                                                                                           \
    hipError_t e_ = hipGetLastError();                                                           \
    if (e_ != hipSuccess) return vv_set_error(VV_E_HIP, "%s: %s", name, hipGetErrorString(e_));  \
  } while (0)

#define VV_TRY(expr)            \
  do {                          \
    int rc_ = (expr);           \
    if (rc_) return rc_;        \
  } while (0)

int vv_launch_gemv_stream(const vv_lin_args& a, hipStream_t s);   // vv_gemv_stream.hip: 1 = launched, 0 = not covered
int vv_launch_mfma_gemm(const vv_lin_args& a, hipStream_t s);     // vv_mfma_gemm.hip: 1 launched, 0 not covered, <0 error
int vv_mfma_gemm_init();
int vv_rmsnorm_rows(const float* x, int64_t ldx, const float* w, float eps, int rows, int n, float* out, int64_t ldo, hipStream_t s);

#endif
