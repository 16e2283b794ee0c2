/*
 * owc.h — C ABI of libowc_hip.so, the MI355X (gfx950) hot path of open-world LMM classification.
 *
 * The reference (altndrr/lmms-owc) is pure Python and has no FFI: its plug points are the
 * `src/models` registry + `Model` ABC and the `src/data/metrics` registry.  This
 * header is the boundary UNDER those plug points: every entry replaces arithmetic the reference
 * reaches through HF transformers / torch, and is what a ctypes stub in the reference's
 * `src/models/_qwen2_vl.py` / `src/data/pipelines/text/_text.py` would bind.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless the name ends in `_host`;
 *   - the caller owns every buffer (inputs, outputs, weights, workspaces); the library owns only a
 *     small zero page inside `owc_ctx`;
 *   - all work is enqueued on `stream` (a hipStream_t passed as void*; NULL = default stream);
 *     no hidden synchronisation, no internal threads;
 *   - return value: 0 (OWC_OK) or a negative owc_status; `owc_last_error(ctx)` gives the text;
 *   - bf16 tensors are raw 16-bit brain-float, row-major, leading dimensions in ELEMENTS.
 */
#ifndef OWC_H_
#define OWC_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct owc_ctx owc_ctx;

enum owc_status {
  OWC_STATUS_OK = 0,
  OWC_STATUS_ERR_ARG = -1,
  OWC_STATUS_ERR_HIP = -2,
  OWC_STATUS_ERR_SHAPE = -3,
  OWC_STATUS_ERR_WORKSPACE = -4
};

/* GEMM epilogues (what a bf16 torch module would do after the matmul, same rounding points). */
enum owc_epilogue {
  OWC_EPILOGUE_NONE = 0,       /* C = bf16(acc + bias) */
  OWC_EPILOGUE_QUICK_GELU = 1, /* HF ACT2FN["quick_gelu"], vision MLP (modeling_qwen2_vl.py:293-302) */
  OWC_EPILOGUE_GELU_ERF = 2,   /* nn.GELU(), PatchMerger (modeling_qwen2_vl.py:281-286) */
  OWC_EPILOGUE_RESIDUAL = 3,   /* C = bf16(residual + bf16(acc + bias)) */
  OWC_EPILOGUE_SWIGLU = 4,     /* gate/up rows interleaved in 16-row groups; C[M, N/2] */
  OWC_EPILOGUE_F32 = 5         /* fp32 output, no rounding */
};

/* ---- lifetime ------------------------------------------------------------------------------ */
int owc_init(int device, owc_ctx** out);
int owc_destroy(owc_ctx* ctx);
const char* owc_last_error(const owc_ctx* ctx);
/* ABI version of this header; bumped whenever a signature changes. */
int owc_abi_version(void);

/* ---- op level ------------------------------------------------------------------------------ */
/* C[M,N] = A[M,K] . W[N,K]^T  (+bias[N]) with a fused epilogue.  Replaces torch.nn.Linear.forward
 * as called from HF modeling_qwen2_vl.py (:349-350, :296-301, :501-504, :460-466).
 * Requirements: K % 8 == 0, lda % 8 == 0, ldw % 8 == 0, N % 4 == 0, ldc % 4 == 0. */
int owc_gemm_bf16(owc_ctx* ctx, const void* A, int64_t lda, const void* W, int64_t ldw,
                  const void* bias, const void* residual, int64_t ldr, void* C, int64_t ldc,
                  int M, int N, int K, int epilogue, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* OWC_H_ */
