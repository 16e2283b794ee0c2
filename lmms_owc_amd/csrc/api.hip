// C ABI of libowc_hip.so (declared in include/owc.h).
#include "../../include/owc.h"
#include "owc_internal.h"

extern "C" {

int owc_abi_version(void) { return 1; }

int owc_init(int device, owc_ctx** out) {
  if (out == nullptr) return OWC_ERR_ARG;
  *out = nullptr;
  if (hipSetDevice(device) != hipSuccess) return OWC_ERR_HIP;
  owc_ctx* ctx = new owc_ctx();
  ctx->device = device;
  if (hipMalloc(&ctx->zeros, 256) != hipSuccess || hipMemset(ctx->zeros, 0, 256) != hipSuccess) {
    delete ctx;
    return OWC_ERR_HIP;
  }
  *out = ctx;
  return OWC_OK;
}

int owc_destroy(owc_ctx* ctx) {
  if (ctx == nullptr) return OWC_ERR_ARG;
  if (ctx->zeros) (void)hipFree(ctx->zeros);
  delete ctx;
  return OWC_OK;
}

const char* owc_last_error(const owc_ctx* ctx) { return ctx ? ctx->err.c_str() : "null ctx"; }

int owc_gemm_bf16(owc_ctx* ctx, const void* A, int64_t lda, const void* W, int64_t ldw,
                  const void* bias, const void* residual, int64_t ldr, void* C, int64_t ldc, int M,
                  int N, int K, int epilogue, void* stream) {
  if (!ctx || !A || !W || !C) return OWC_ERR_ARG;
  int rc = owc_launch_gemm_bf16(A, lda, W, ldw, bias, residual, ldr, C, ldc, M, N, K, epilogue,
                                ctx->zeros, (hipStream_t)stream);
  if (rc != OWC_OK) ctx->err = "owc_gemm_bf16: bad shape/argument or launch failure";
  return rc;
}

}  // extern "C"
