// fp32 "NT" GEMM on the f32-input MFMA (v_mfma_f32_16x16x4_f32, exact fp32 products/accumulation):
//   C[M,N] = A[M,K] . W[N,K]^T (+bias) with optional erf-GELU or residual epilogue, all fp32.
// Used by the sentence encoder (MiniLM/BERT linears, BERT: modeling_bert.py BertSelfAttention /
// BertSelfOutput / BertIntermediate / BertOutput) where the reference's CPU path is fp32 and the
// cosine tolerance is 1e-4 (src/data/pipelines/text/_text.py:165-170 uses fp32 on CPU).
//
// Same CDNA4 structure as gemm_bf16.hip: 128x128 block tile, K-step of 32 floats (128-byte rows),
// 16-byte LDS-DMA staging with the (row>>1)&7 XOR swizzle on source + read, double-buffered LDS,
// swapped operands so a lane owns 4 consecutive output columns (16-byte stores).  A 16-byte
// fragment read supplies 4 consecutive k of one row; lane group g reads chunk 4s+g, and the 4 floats
// feed 4 successive 16x16x4 MFMAs — a fixed permutation of k applied identically to both operands.
#include "owc_common.h"
#include "owc_internal.h"

namespace {

constexpr int BM = 128, BN = 128, BKF = 32;
constexpr int TILE_BYTES = BM * BKF * 4;  // 16 KiB
constexpr int GROUP_M = 8;

__device__ __forceinline__ float gelu_erf_f(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }

template <int EPI>
__global__ __launch_bounds__(256) void gemm_f32_nt_kernel(
    const float* __restrict__ A, long lda, const float* __restrict__ W, long ldw,
    const float* __restrict__ bias, const float* R, long ldr, float* C, long ldc, int M, int N, int K,
    const void* __restrict__ zeros, int tiles_m, int tiles_n) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int tid = threadIdx.x;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l = tid & 63;

  const int nblk = tiles_m * tiles_n;
  const int bid = blockIdx.x;
  const int q = nblk >> 3, r = nblk & 7;
  const int xcd = bid & 7, idx = bid >> 3;
  const int lid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  const int width = GROUP_M * tiles_n;
  const int group = lid / width;
  const int first_m = group * GROUP_M;
  const int gsize = min(tiles_m - first_m, GROUP_M);
  const int tm = first_m + (lid % width) % gsize;
  const int tn = (lid % width) / gsize;
  const int m0 = tm * BM, n0 = tn * BN;

  const char* asrc[4];
  const char* wsrc[4];
  int kchunk[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int row = 32 * w + 8 * j + (l >> 3);
    const int c = (l & 7) ^ ((row >> 1) & 7);
    kchunk[j] = c * 4;
    const int am = min(m0 + row, M - 1);
    const int wn_ = min(n0 + row, N - 1);
    asrc[j] = (const char*)(A + (long)am * lda + c * 4);
    wsrc[j] = (const char*)(W + (long)wn_ * ldw + c * 4);
  }
  const int nk = (K + BKF - 1) / BKF;

  auto stage = [&](int buf, int kt) {
    char* la = lds + buf * (2 * TILE_BYTES) + w * 4096;
    char* lw = la + TILE_BYTES;
    const int k0 = kt * BKF;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const bool ok = (k0 + kchunk[j]) < K;
      const void* ga = ok ? (const void*)(asrc[j] + (long)k0 * 4) : zeros;
      const void* gw = ok ? (const void*)(wsrc[j] + (long)k0 * 4) : zeros;
      glds16(ga, la + j * 1024);
      glds16(gw, lw + j * 1024);
    }
  };

  const int wm = w >> 1, wn = w & 1;
  const int fr = l & 15, fq = l >> 4;
  const int swz = (fr >> 1) & 7;
  const int offA0 = (wm * 64 + fr) * 128 + (((0 + fq) ^ swz) << 4);
  const int offA1 = (wm * 64 + fr) * 128 + (((4 + fq) ^ swz) << 4);
  const int offW0 = (wn * 64 + fr) * 128 + (((0 + fq) ^ swz) << 4);
  const int offW1 = (wn * 64 + fr) * 128 + (((4 + fq) ^ swz) << 4);

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  stage(0, 0);
  __syncthreads();

  for (int kt = 0; kt < nk; ++kt) {
    const int cur = kt & 1;
    if (kt + 1 < nk) stage(cur ^ 1, kt + 1);
    const char* la = lds + cur * (2 * TILE_BYTES);
    const char* lw = la + TILE_BYTES;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      f32x4 fa[4], fw[4];
      const int oa = ks ? offA1 : offA0;
      const int ow = ks ? offW1 : offW0;
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        fa[t] = *(const f32x4*)(la + oa + t * 16 * 128);
        fw[t] = *(const f32x4*)(lw + ow + t * 16 * 128);
      }
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
#pragma unroll
          for (int mt = 0; mt < 4; ++mt)
            acc[nt][mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(fw[nt][e], fa[mt][e], acc[nt][mt], 0, 0, 0);
    }
    __syncthreads();
  }

#pragma unroll
  for (int mt = 0; mt < 4; ++mt) {
    const int m = m0 + wm * 64 + mt * 16 + fr;
    if (m >= M) continue;
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
      const int n = n0 + wn * 64 + nt * 16 + fq * 4;
      if (n >= N) continue;
      f32x4 v = acc[nt][mt];
      if (bias != nullptr) v += *(const f32x4*)(bias + n);
      if constexpr (EPI == OWC_EPI_GELU_ERF) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = gelu_erf_f(v[e]);
      } else if constexpr (EPI == OWC_EPI_RESIDUAL) {
        v += *(const f32x4*)(R + (long)m * ldr + n);
      }
      *(f32x4*)(C + (long)m * ldc + n) = v;
    }
  }
}

// ------------------------------------------------------------------------------------------------
// fp32 GEMM on the bf16 MFMA ("bf16x3"): every fp32 operand is split exactly into three bf16 pieces
// (hi = bf16(a), mid = bf16(a - hi), lo = bf16(a - hi - mid); the two subtractions are exact in fp32) and the product is the six
// largest of the nine piece products, accumulated in fp32 by v_mfma_f32_16x16x32_bf16:
//   a.b ~= hi.hi + hi.mid + mid.hi + hi.lo + lo.hi + mid.mid        (dropped: mid.lo, lo.mid, lo.lo <= 2^-26 |a.b|)
// bf16 x bf16 products are exact in fp32, so the result carries fp32-level error (below the fp32 rounding of the sum itself)
// at 6 bf16 MFMAs (96 cycles) per 16x16x32 block instead of 8 f32-input MFMAs (256 cycles).  Same tiles, staging and epilogues as
// the exact kernel above; the sentence encoder uses this one (owc_tuning_set("bert_bf16x3", 0) selects the exact kernel).
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void split3(const f32x4& u, const f32x4& v, bf16x8& h, bf16x8& m, bf16x8& lo) {
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const float a = e < 4 ? u[e] : v[e - 4];
    const bf16_t ah = f2bf(a);
    const float r1 = a - bf2f(ah);
    const bf16_t am = f2bf(r1);
    const float r2 = r1 - bf2f(am);
    h[e] = ah;
    m[e] = am;
    lo[e] = f2bf(r2);
  }
}

template <int EPI>
__global__ __launch_bounds__(256) void gemm_f32x3_nt_kernel(
    const float* __restrict__ A, long lda, const float* __restrict__ W, long ldw,
    const float* __restrict__ bias, const float* R, long ldr, float* C, long ldc, int M, int N, int K,
    const void* __restrict__ zeros, int tiles_m, int tiles_n) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int tid = threadIdx.x;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l = tid & 63;

  const int nblk = tiles_m * tiles_n;
  const int bid = blockIdx.x;
  const int q = nblk >> 3, r = nblk & 7;
  const int xcd = bid & 7, idx = bid >> 3;
  const int lid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  const int width = GROUP_M * tiles_n;
  const int group = lid / width;
  const int first_m = group * GROUP_M;
  const int gsize = min(tiles_m - first_m, GROUP_M);
  const int tm = first_m + (lid % width) % gsize;
  const int tn = (lid % width) / gsize;
  const int m0 = tm * BM, n0 = tn * BN;

  const char* asrc[4];
  const char* wsrc[4];
  int kchunk[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int row = 32 * w + 8 * j + (l >> 3);
    const int c = (l & 7) ^ ((row >> 1) & 7);
    kchunk[j] = c * 4;
    const int am = min(m0 + row, M - 1);
    const int wn_ = min(n0 + row, N - 1);
    asrc[j] = (const char*)(A + (long)am * lda + c * 4);
    wsrc[j] = (const char*)(W + (long)wn_ * ldw + c * 4);
  }
  const int nk = (K + BKF - 1) / BKF;

  auto stage = [&](int buf, int kt) {
    char* la = lds + buf * (2 * TILE_BYTES) + w * 4096;
    char* lw = la + TILE_BYTES;
    const int k0 = kt * BKF;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const bool ok = (k0 + kchunk[j]) < K;
      const void* ga = ok ? (const void*)(asrc[j] + (long)k0 * 4) : zeros;
      const void* gw = ok ? (const void*)(wsrc[j] + (long)k0 * 4) : zeros;
      glds16(ga, la + j * 1024);
      glds16(gw, lw + j * 1024);
    }
  };

  const int wm = w >> 1, wn = w & 1;
  const int fr = l & 15, fq = l >> 4;
  const int swz = (fr >> 1) & 7;
  const int offA0 = (wm * 64 + fr) * 128 + (((0 + fq) ^ swz) << 4);
  const int offA1 = (wm * 64 + fr) * 128 + (((4 + fq) ^ swz) << 4);
  const int offW0 = (wn * 64 + fr) * 128 + (((0 + fq) ^ swz) << 4);
  const int offW1 = (wn * 64 + fr) * 128 + (((4 + fq) ^ swz) << 4);

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  stage(0, 0);
  __syncthreads();

  for (int kt = 0; kt < nk; ++kt) {
    const int cur = kt & 1;
    if (kt + 1 < nk) stage(cur ^ 1, kt + 1);
    const char* la = lds + cur * (2 * TILE_BYTES);
    const char* lw = la + TILE_BYTES;
    // a lane's 8 k values of the 32-deep tile: floats 4fq .. 4fq+3 and 16+4fq .. 16+4fq+3 of its row -- the same
    // (arbitrary) k permutation on both operands
    // the A fragments are split up front; each W fragment is split right before its 24 MFMAs, so the splitting VALU work of
    // n tile nt+1 can issue under the matrix pipe of n tile nt (one basic block: the scheduler interleaves them)
    bf16x8 ah[4], am[4], al[4];
    f32x4 w0[4], w1[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      w0[t] = *(const f32x4*)(lw + offW0 + t * 16 * 128);
      w1[t] = *(const f32x4*)(lw + offW1 + t * 16 * 128);
    }
#pragma unroll
    for (int t = 0; t < 4; ++t)
      split3(*(const f32x4*)(la + offA0 + t * 16 * 128), *(const f32x4*)(la + offA1 + t * 16 * 128), ah[t], am[t], al[t]);
    bf16x8 wh[2], wmd[2], wl[2];
    split3(w0[0], w1[0], wh[0], wmd[0], wl[0]);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
      const int c_ = nt & 1, n_ = c_ ^ 1;
      if (nt < 3) split3(w0[nt + 1], w1[nt + 1], wh[n_], wmd[n_], wl[n_]);  // rides under this n tile's MFMAs
#pragma unroll
      for (int mt = 0; mt < 4; ++mt) {
        f32x4 c = acc[nt][mt];
        // small terms first
        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wmd[c_], am[mt], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wl[c_], ah[mt], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh[c_], al[mt], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wmd[c_], ah[mt], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh[c_], am[mt], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh[c_], ah[mt], c, 0, 0, 0);
        acc[nt][mt] = c;
      }
      // one MFMA, then two of the next fragment's splitting instructions, ...
#pragma unroll
      for (int i = 0; i < 24; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    __syncthreads();
  }

#pragma unroll
  for (int mt = 0; mt < 4; ++mt) {
    const int m = m0 + wm * 64 + mt * 16 + fr;
    if (m >= M) continue;
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
      const int n = n0 + wn * 64 + nt * 16 + fq * 4;
      if (n >= N) continue;
      f32x4 v = acc[nt][mt];
      if (bias != nullptr) v += *(const f32x4*)(bias + n);
      if constexpr (EPI == OWC_EPI_GELU_ERF) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = gelu_erf_f(v[e]);
      } else if constexpr (EPI == OWC_EPI_RESIDUAL) {
        v += *(const f32x4*)(R + (long)m * ldr + n);
      }
      *(f32x4*)(C + (long)m * ldc + n) = v;
    }
  }
}

template <int EPI>
int launch(const float* A, long lda, const float* W, long ldw, const float* bias, const float* R,
           long ldr, float* C, long ldc, int M, int N, int K, const void* zeros, bool x3, hipStream_t s) {
  const int tiles_m = (M + BM - 1) / BM, tiles_n = (N + BN - 1) / BN;
  static bool attr_set = false;
  if (!attr_set) {
    if (hipFuncSetAttribute((const void*)gemm_f32_nt_kernel<EPI>,
                            hipFuncAttributeMaxDynamicSharedMemorySize, 4 * TILE_BYTES) != hipSuccess ||
        hipFuncSetAttribute((const void*)gemm_f32x3_nt_kernel<EPI>,
                            hipFuncAttributeMaxDynamicSharedMemorySize, 4 * TILE_BYTES) != hipSuccess)
      return OWC_ERR_HIP;
    attr_set = true;
  }
  const int prof = owc_gemm_profile_begin(2.0 * (double)M * (double)N * (double)K, OWC_PROF_SCORER_GEMM, s);
  if (x3)
    hipLaunchKernelGGL(gemm_f32x3_nt_kernel<EPI>, dim3(tiles_m * tiles_n), dim3(256), 4 * TILE_BYTES, s, A,
                       lda, W, ldw, bias, R, ldr, C, ldc, M, N, K, zeros, tiles_m, tiles_n);
  else
    hipLaunchKernelGGL(gemm_f32_nt_kernel<EPI>, dim3(tiles_m * tiles_n), dim3(256), 4 * TILE_BYTES, s, A,
                       lda, W, ldw, bias, R, ldr, C, ldc, M, N, K, zeros, tiles_m, tiles_n);
  owc_gemm_profile_end(prof, s);
  return hipGetLastError() == hipSuccess ? OWC_OK : OWC_ERR_HIP;
}

int g_bert_x3 = 1;  // owc_tuning_set("bert_bf16x3", 0): the sentence encoder's linears on the exact f32-input MFMA kernel

}  // namespace

static int gemm_f32_any(const float* A, long lda, const float* W, long ldw, const float* bias,
                        const float* R, long ldr, float* C, long ldc, int M, int N, int K, int epi,
                        const void* zeros, bool x3, hipStream_t s) {
  if (M <= 0 || N <= 0 || K <= 0) return OWC_ERR_SHAPE;
  if ((K & 3) || (lda & 3) || (ldw & 3) || (N & 3) || (ldc & 3)) return OWC_ERR_SHAPE;
  if (epi == OWC_EPI_RESIDUAL && (R == nullptr || (ldr & 3))) return OWC_ERR_ARG;
  switch (epi) {
    case OWC_EPI_NONE: return launch<OWC_EPI_NONE>(A, lda, W, ldw, bias, R, ldr, C, ldc, M, N, K, zeros, x3, s);
    case OWC_EPI_GELU_ERF: return launch<OWC_EPI_GELU_ERF>(A, lda, W, ldw, bias, R, ldr, C, ldc, M, N, K, zeros, x3, s);
    case OWC_EPI_RESIDUAL: return launch<OWC_EPI_RESIDUAL>(A, lda, W, ldw, bias, R, ldr, C, ldc, M, N, K, zeros, x3, s);
    default: return OWC_ERR_ARG;
  }
}

// exact f32-input MFMA (the C ABI's owc_gemm_f32)
int owc_launch_gemm_f32(const float* A, long lda, const float* W, long ldw, const float* bias,
                        const float* R, long ldr, float* C, long ldc, int M, int N, int K, int epi,
                        const void* zeros, hipStream_t s) {
  return gemm_f32_any(A, lda, W, ldw, bias, R, ldr, C, ldc, M, N, K, epi, zeros, false, s);
}

// the sentence encoder's linears: bf16x3 unless switched off
int owc_launch_gemm_f32_bert(const float* A, long lda, const float* W, long ldw, const float* bias,
                             const float* R, long ldr, float* C, long ldc, int M, int N, int K, int epi,
                             const void* zeros, hipStream_t s) {
  return gemm_f32_any(A, lda, W, ldw, bias, R, ldr, C, ldc, M, N, K, epi, zeros, g_bert_x3 != 0, s);
}

void owc_bert_set_x3(int v) { g_bert_x3 = v; }
