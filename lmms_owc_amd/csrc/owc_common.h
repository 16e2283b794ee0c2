// Shared device/host helpers for the owc HIP kernels (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef __bf16 bf16_t;
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

#define OWC_WAVE 64

// Timing-only experiment knobs (parts of a kernel switched OFF to price them: results are garbage) exist only in the second
// library `python -m lmms_owc_amd.build --timing` builds for tools/ (libowc_hip_timing.so, -DOWC_TIMING_KNOBS).  In the product
// library the branches below are dead code and `owc_tuning_set("gemm_dbg" | "attn_dbg")` is an unknown knob.
#ifdef OWC_TIMING_KNOBS
#define OWC_TK(cond) (cond)
#else
#define OWC_TK(cond) false
#endif

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

// 16-byte async global -> LDS copy (LDS dest = wave-uniform base + lane*16).
__device__ __forceinline__ void glds16(const void* g, void* l) {
  __builtin_amdgcn_global_load_lds((gptr_t)g, (lptr_t)l, 16, 0, 0);
}

__device__ __forceinline__ float bf2f(bf16_t x) { return (float)x; }
__device__ __forceinline__ bf16_t f2bf(float x) { return (bf16_t)x; }
// round-trip through bf16 (the rounding point of a bf16 torch op)
__device__ __forceinline__ float rbf(float x) { return (float)(bf16_t)x; }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// RMSNorm statistics of one row held by ONE WAVE, the arithmetic every kernel that normalises a row must share bit for bit
// (norm_kernel in elementwise.hip; the decode GEMM that normalises its own activations, gemm_bf16.hip): lane l accumulates the
// 8-element chunks l, l + 64, ... in ascending order with ONE fma per element (pinned: never a separate multiply and add), then the
// xor-butterfly of wave_sum; rstd = rsq(fma(sq, 1/d, eps)).
__device__ __forceinline__ float owc_rms_rstd(const bf16_t* __restrict__ x, int d, float eps, int l) {
  const int nch = d >> 3;
  float sq = 0.f;
  for (int ch = l; ch < nch; ch += 64) {
    const bf16x8 c = *(const bf16x8*)(x + ch * 8);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float v = bf2f(c[e]);
      sq = __builtin_fmaf(v, v, sq);
    }
  }
  sq = wave_sum(sq);
  return rsqrtf(__builtin_fmaf(sq, 1.0f / (float)d, eps));
}
// y = bf16(w * bf16(x * rstd))  (Qwen2RMSNorm: the normalised value is cast to the input dtype before the weight multiply)
__device__ __forceinline__ bf16_t owc_rms_apply(bf16_t x, bf16_t w, float rstd) { return f2bf(bf2f(w) * rbf(bf2f(x) * rstd)); }

// Epilogue selectors shared by the GEMM kernels and the C ABI (include/owc.h).
enum {
  OWC_EPI_NONE = 0,        // C = bf16(acc + bias)
  OWC_EPI_QUICK_GELU = 1,  // C = bf16(x * sigmoid(1.702 x)), x = bf16(acc + bias)
  OWC_EPI_GELU_ERF = 2,    // C = bf16(gelu_erf(x))
  OWC_EPI_RESIDUAL = 3,    // C = bf16(res + bf16(acc + bias))
  OWC_EPI_SWIGLU = 4,      // interleaved gate/up rows: C[m][f] = bf16(bf16(silu(g)) * u)
  OWC_EPI_F32 = 5,         // C (fp32) = acc + bias   (no rounding)
  OWC_EPI_VROPE = 6,       // vision qkv: bf16(acc + bias), then 2-D RoPE on pair-interleaved q/k columns
};

// Extra operands of the fused-RoPE epilogue (vision qkv projection).
struct owc_gemm_aux {
  const int* pos_hw;     // [M][2] patch (h, w)
  const float* cos_t;    // [positions][head_dim/4]
  const float* sin_t;
  int rope_cols;         // columns < rope_cols (q and k blocks) are rotated
  int head_dim;
  // the ring kernel's RMSNorm-fused form (decode at <= 8 rows): A is the RAW residual rows, normalised on their way into LDS
  const void* gamma = nullptr;
  float eps = 0.f;
};

#define OWC_OK 0
#define OWC_ERR_ARG -1
#define OWC_ERR_HIP -2
#define OWC_ERR_SHAPE -3
#define OWC_ERR_WORKSPACE -4
