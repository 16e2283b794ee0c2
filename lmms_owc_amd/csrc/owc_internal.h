// Internal (non-ABI) declarations shared by the translation units of libowc_hip.so.
#pragma once
#include "owc_common.h"
#include <string>

struct owc_ctx {
  int device = 0;
  void* zeros = nullptr;  // 256-byte device zero page (K-tail source of the LDS-DMA GEMMs)
  std::string err;
};

#define OWC_CHECK_HIP(ctx, expr)                                                        \
  do {                                                                                  \
    hipError_t e__ = (expr);                                                            \
    if (e__ != hipSuccess) {                                                            \
      if (ctx) (ctx)->err = std::string(#expr) + ": " + hipGetErrorString(e__);         \
      return OWC_ERR_HIP;                                                               \
    }                                                                                   \
  } while (0)

#define OWC_FAIL(ctx, code, msg) \
  do {                           \
    if (ctx) (ctx)->err = (msg); \
    return (code);               \
  } while (0)

#define OWC_TRY(expr)            \
  do {                           \
    int rc__ = (expr);           \
    if (rc__ != OWC_OK) return rc__; \
  } while (0)

// ---- kernel launchers (one per .hip file) ----
int owc_launch_gemm_bf16(const void* A, long lda, const void* W, long ldw, const void* bias,
                         const void* R, long ldr, void* C, long ldc, int M, int N, int K, int epi,
                         const void* zeros, hipStream_t s);
