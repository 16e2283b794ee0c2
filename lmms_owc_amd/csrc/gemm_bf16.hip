// bf16 "NT" GEMM for nn.Linear-shaped work:  C[M,N] = A[M,K] . W[N,K]^T (+bias, fused epilogue)
//
// Replaces every torch.nn.Linear on the Qwen2-VL path that the reference reaches through
// HF transformers (HF:models/qwen2_vl/modeling_qwen2_vl.py:268-275 patch-embed conv-as-GEMM,
// :349-350 vision qkv/proj, :296-301 vision MLP, :281-291 merger MLP, :501-504 decoder q/k/v/o,
// :460-466 decoder MLP, lm_head).  fp32 accumulation on the MFMA pipe, one rounding to bf16 at the
// same points where a bf16 torch module rounds.
//
// CDNA4 design (v_mfma_f32_16x16x32_bf16, 64-lane waves):
//  * 128x128x64 block tile, 256 threads = 4 waves (2 along M x 2 along N), 64x64 per wave,
//    4x4 accumulator tiles of 16x16 per wave (64 acc VGPRs).
//  * both operands are K-contiguous, so both tiles are staged with 16-byte LDS-DMA
//    (global_load_lds_dwordx4): one wave-instruction = 8 rows x 128 B, LDS image linear.  The XOR
//    swizzle chunk ^= (row>>1)&7 is applied on the per-lane SOURCE address and again on the
//    ds_read_b128 address, which makes every 16-lane ds_read_b128 group hit 16 distinct 16-B slots.
//  * double-buffered LDS (64 KiB / block -> 2 blocks per CU), next K-step's DMA is issued before
//    the current step's MFMAs.
//  * the MFMA operands are swapped (D^T = W . A^T) so every lane owns 4 consecutive output
//    columns of one row: 8-byte packed bf16 stores, bias/activation/residual fused.
//  * ragged edges: rows >= M / >= N are clamped on the source (results discarded at the store),
//    K tails (K % 64 != 0, K % 8 == 0) read from a 16-byte zero page.
//  * XCD-aware block->tile map: blocks that share an XCD (bid % 8) walk a contiguous range of
//    tiles in GROUP_M-major order so A/W panels are reused out of that XCD's L2.
#include "owc_common.h"
#include <vector>

namespace {

// ---- optional live profiling of THIS kernel (bench.py roofline leg): one HIP-event pair per launch on
// the launch stream, summed by owc_gemm_profile_read().  Off by default (zero cost).
struct GemmProfile {
  bool on = false;
  std::vector<hipEvent_t> ev;  // start/stop pairs
  std::vector<double> flops;
  size_t used = 0;
};
GemmProfile g_prof;

constexpr int BM = 128, BN = 128, BK = 64;
constexpr int TILE_BYTES = BM * BK * 2;  // 16 KiB per operand tile
constexpr int GROUP_M = 8;

__device__ __forceinline__ float act_quick_gelu(float x) { return x / (1.0f + __expf(-1.702f * x)); }
__device__ __forceinline__ float act_gelu_erf(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }
__device__ __forceinline__ float act_silu(float x) { return x / (1.0f + __expf(-x)); }

template <int EPI>
__global__ __launch_bounds__(256) void gemm_bf16_nt_kernel(
    const bf16_t* __restrict__ A, long lda, const bf16_t* __restrict__ W, long ldw,
    const bf16_t* __restrict__ bias, const bf16_t* R, long ldr, void* Cv,
    long ldc, int M, int N, int K, const void* __restrict__ zeros, int tiles_m, int tiles_n) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  // [buf][A|W][128 rows][128 B]
  const int tid = threadIdx.x;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l = tid & 63;

  // ---- XCD-aware tile id ----
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

  // ---- staging addresses: wave w stages rows [32w, 32w+32) of both tiles, 4 pieces of 8 rows ----
  const char* asrc[4];
  const char* wsrc[4];
  int kchunk[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int row = 32 * w + 8 * j + (l >> 3);
    const int c = (l & 7) ^ ((row >> 1) & 7);
    kchunk[j] = c * 8;
    const int am = min(m0 + row, M - 1);
    const int wn_ = min(n0 + row, N - 1);
    asrc[j] = (const char*)(A + (long)am * lda + c * 8);
    wsrc[j] = (const char*)(W + (long)wn_ * ldw + c * 8);
  }
  const int nk = (K + BK - 1) / BK;

  auto stage = [&](int buf, int kt) {
    char* la = lds + buf * (2 * TILE_BYTES) + w * 4096;
    char* lw = la + TILE_BYTES;
    const int k0 = kt * BK;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const bool ok = (k0 + kchunk[j]) < K;
      const void* ga = ok ? (const void*)(asrc[j] + (long)k0 * 2) : zeros;
      const void* gw = ok ? (const void*)(wsrc[j] + (long)k0 * 2) : zeros;
      glds16(ga, la + j * 1024);
      glds16(gw, lw + j * 1024);
    }
  };

  // ---- fragment read offsets ----
  const int wm = w >> 1, wn = w & 1;
  const int fr = l & 15, fq = l >> 4;
  const int swz = (fr >> 1) & 7;
  // byte offset inside a tile for k-step ks (0/1): row*128 + ((ks*4+fq) ^ swz)*16
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
      bf16x8 fa[4], fw[4];
      const int oa = ks ? offA1 : offA0;
      const int ow = ks ? offW1 : offW0;
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        fa[t] = *(const bf16x8*)(la + oa + t * 16 * 128);
        fw[t] = *(const bf16x8*)(lw + ow + t * 16 * 128);
      }
#pragma unroll
      for (int nt = 0; nt < 4; ++nt)
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
          acc[nt][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fw[nt], fa[mt], acc[nt][mt], 0, 0, 0);
    }
    __syncthreads();
  }

  // ---- epilogue: lane owns row m = ..+fr, 4 consecutive columns n = ..+fq*4+{0..3} per tile ----
#pragma unroll
  for (int mt = 0; mt < 4; ++mt) {
    const int m = m0 + wm * 64 + mt * 16 + fr;
    if (m >= M) continue;
    if constexpr (EPI == OWC_EPI_SWIGLU) {
      bf16_t* C = (bf16_t*)Cv;
#pragma unroll
      for (int nt = 0; nt < 4; nt += 2) {
        const int nb = n0 + wn * 64 + nt * 16;  // gate rows nb.., up rows nb+16..
        if (nb + 16 + fq * 4 >= N) continue;
        const int f = (nb >> 1) + fq * 4;
        bf16x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float g = rbf(acc[nt][mt][e]);
          const float u = rbf(acc[nt + 1][mt][e]);
          o[e] = f2bf(rbf(act_silu(g)) * u);
        }
        *(bf16x4*)(C + (long)m * ldc + f) = o;
      }
    } else {
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) {
        const int n = n0 + wn * 64 + nt * 16 + fq * 4;
        if (n >= N) continue;
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = acc[nt][mt][e];
        if (bias != nullptr) {
          const bf16x4 b = *(const bf16x4*)(bias + n);
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] += bf2f(b[e]);
        }
        if constexpr (EPI == OWC_EPI_F32) {
          float* C = (float*)Cv;
          *(f32x4*)(C + (long)m * ldc + n) = (f32x4){v[0], v[1], v[2], v[3]};
        } else {
          bf16_t* C = (bf16_t*)Cv;
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = rbf(v[e]);
          if constexpr (EPI == OWC_EPI_QUICK_GELU) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = act_quick_gelu(v[e]);
          } else if constexpr (EPI == OWC_EPI_GELU_ERF) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = act_gelu_erf(v[e]);
          } else if constexpr (EPI == OWC_EPI_RESIDUAL) {
            const bf16x4 rr = *(const bf16x4*)(R + (long)m * ldr + n);
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] += bf2f(rr[e]);
          }
          bf16x4 o;
#pragma unroll
          for (int e = 0; e < 4; ++e) o[e] = f2bf(v[e]);
          *(bf16x4*)(C + (long)m * ldc + n) = o;
        }
      }
    }
  }
}

template <int EPI>
int launch(const void* A, long lda, const void* W, long ldw, const void* bias, const void* R,
           long ldr, void* C, long ldc, int M, int N, int K, const void* zeros, hipStream_t s) {
  const int tiles_m = (M + BM - 1) / BM, tiles_n = (N + BN - 1) / BN;
  static bool attr_set = false;
  if (!attr_set) {
    if (hipFuncSetAttribute((const void*)gemm_bf16_nt_kernel<EPI>,
                            hipFuncAttributeMaxDynamicSharedMemorySize, 4 * TILE_BYTES) != hipSuccess)
      return OWC_ERR_HIP;
    attr_set = true;
  }
  hipEvent_t e0 = nullptr, e1 = nullptr;
  if (g_prof.on) {
    if (g_prof.used + 2 > g_prof.ev.size()) {
      hipEvent_t a, b;
      if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) return OWC_ERR_HIP;
      g_prof.ev.push_back(a);
      g_prof.ev.push_back(b);
    }
    e0 = g_prof.ev[g_prof.used];
    e1 = g_prof.ev[g_prof.used + 1];
    g_prof.used += 2;
    g_prof.flops.push_back(2.0 * (double)M * (double)N * (double)K);
    (void)hipEventRecord(e0, s);
  }
  hipLaunchKernelGGL(gemm_bf16_nt_kernel<EPI>, dim3(tiles_m * tiles_n), dim3(256), 4 * TILE_BYTES, s,
                     (const bf16_t*)A, lda, (const bf16_t*)W, ldw, (const bf16_t*)bias,
                     (const bf16_t*)R, ldr, C, ldc, M, N, K, zeros, tiles_m, tiles_n);
  if (e1) (void)hipEventRecord(e1, s);
  return hipGetLastError() == hipSuccess ? OWC_OK : OWC_ERR_HIP;
}

}  // namespace

// Internal entry used by the C ABI (api.cpp) and by the model drivers.
int owc_launch_gemm_bf16(const void* A, long lda, const void* W, long ldw, const void* bias,
                         const void* R, long ldr, void* C, long ldc, int M, int N, int K, int epi,
                         const void* zeros, hipStream_t s) {
  if (M <= 0 || N <= 0 || K <= 0) return OWC_ERR_SHAPE;
  if ((K & 7) || (lda & 7) || (ldw & 7) || (N & 3) || (ldc & 3)) return OWC_ERR_SHAPE;
  if (epi == OWC_EPI_SWIGLU && (N & 31)) return OWC_ERR_SHAPE;
  if (epi == OWC_EPI_RESIDUAL && (R == nullptr || (ldr & 3))) return OWC_ERR_ARG;
  switch (epi) {
    case OWC_EPI_NONE: return launch<OWC_EPI_NONE>(A, lda, W, ldw, bias, R, ldr, C, ldc, M, N, K, zeros, s);
    case OWC_EPI_QUICK_GELU: return launch<OWC_EPI_QUICK_GELU>(A, lda, W, ldw, bias, R, ldr, C, ldc, M, N, K, zeros, s);
    case OWC_EPI_GELU_ERF: return launch<OWC_EPI_GELU_ERF>(A, lda, W, ldw, bias, R, ldr, C, ldc, M, N, K, zeros, s);
    case OWC_EPI_RESIDUAL: return launch<OWC_EPI_RESIDUAL>(A, lda, W, ldw, bias, R, ldr, C, ldc, M, N, K, zeros, s);
    case OWC_EPI_SWIGLU: return launch<OWC_EPI_SWIGLU>(A, lda, W, ldw, bias, R, ldr, C, ldc, M, N, K, zeros, s);
    case OWC_EPI_F32: return launch<OWC_EPI_F32>(A, lda, W, ldw, bias, R, ldr, C, ldc, M, N, K, zeros, s);
    default: return OWC_ERR_ARG;
  }
}

// ---- profiling hooks (C ABI: owc_gemm_profile_enable / owc_gemm_profile_read in api.hip) ----
void owc_gemm_profile_set(int on) {
  g_prof.on = on != 0;
  g_prof.used = 0;
  g_prof.flops.clear();
}

// Sums the recorded launches (the caller has synchronised the stream): total kernel milliseconds,
// total algorithmic FLOPs (2*M*N*K per launch) and the launch count; then resets the recording.
int owc_gemm_profile_collect(double* total_ms, double* total_flops, long* launches) {
  double ms = 0.0, fl = 0.0;
  const size_t n = g_prof.used / 2;
  for (size_t i = 0; i < n; ++i) {
    float t = 0.f;
    if (hipEventElapsedTime(&t, g_prof.ev[2 * i], g_prof.ev[2 * i + 1]) != hipSuccess) return OWC_ERR_HIP;
    ms += t;
    fl += g_prof.flops[i];
  }
  *total_ms = ms;
  *total_flops = fl;
  *launches = (long)n;
  g_prof.used = 0;
  g_prof.flops.clear();
  return OWC_OK;
}
