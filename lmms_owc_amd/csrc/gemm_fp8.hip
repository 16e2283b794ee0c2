// fp8 (OCP e4m3fn) path of the decoder projections (BASELINE.json config #5: Qwen2-VL-72B fp8 MFMA decode; no reference
// counterpart - the reference's only reduced-precision loader is bitsandbytes, src/models/_base.py:116-121).
//
//   quant_rows_fp8_kernel : q[r][c] = rne_e4m3(x[r][c] / s[r]),  s[r] = max|x[r]| / 448  (per token / per output channel)
//   gemm_fp8_nt_256_kernel: C[M,N] = epilogue((A8[M,K] . W8[N,K]^T) * sa[m] * sw[n] + bias)   (bf16 out)
//
// The GEMM is the 256x256 bf16 kernel (gemm_bf16.hip) with one change of arithmetic: a K-tile is 128 fp8 elements, i.e. the
// same 128-byte rows, the same LDS image, swizzle and LDS-DMA schedule, but each lane's operand is 32 consecutive K bytes
// (two swizzled 16-byte chunks) and the product is ONE v_mfma_scale_f32_16x16x128_f8f6f4 per 16x16 tile with unit block
// scales (e8m0 0x7f): 4x the K of the bf16 instruction in 2x its cycles = twice the bf16 rate per byte staged.
// Operand map checked with exact integer data (tools/probes/probe_fp8_mfma.hip): lane l holds row l & 15,
// k = 32 * (l >> 4) + j, j = 0..31; C/D is the dtype-independent 16x16 map.
#include "gemm_epilogue.h"
#include "owc_internal.h"
#include <type_traits>

namespace {

typedef int i32x8 __attribute__((ext_vector_type(8)));

constexpr int BT = 256, BKB = 128;                 // tile edge, K-tile in BYTES (= fp8 elements)
constexpr int OP_BYTES = BT * BKB;                 // 32 KiB per operand per stage
constexpr int STAGE_BYTES = 2 * OP_BYTES;
constexpr int GROUP_M2 = 4;
int g_fp8_pingpong = 1;       // "gemm_pingpong" knob: 2 / default = on, 0 or 1 = the lock-step fp8 kernel.  Round 2 had it off (1.74 vs 2.44 PF on 7b.gateup);
                              // round 3 found why - LLVM sank the last K-tile's MFMAs into the epilogue's store blocks (scratch) - and gave it the
                              // bf16 kernel's two-set W layout: 7b.down 2573 -> 3001, 72b.gateup 2658 -> 3008, sq8192 2736 -> 3059 TFLOP/s
int g_fp8_skinny_max_m = 0;   // follows the "gemm_skinny_max_m" knob.  Off by default since round 3: the ring kernel's narrow / wide tiles beat it at every M (72B fp8 decode step at batch 1 / 8 / 16 / 32 / 64: 20.8 / 21.9 / 23.7 / 28.0 / 41.2 -> 19.9 / 20.1 / 20.0 / 20.9 / 22.9 ms)
int g_fp8_shapes = 1;          // the ring kernel's other tile shapes / K grouping (follows "gemm_small_tiles")
int g_fp8_ring_128 = 1;        // 128x64 tiles of the ring kernel where 64x64 tiles need more than one round of two blocks per CU (follows "gemm_ring_128")
int g_fp8_mid_max_tiles = 128;  // fewer 256x256 tiles than this -> 64x64 tiles (follows "gemm_mid_max_tiles": 0 disables)
constexpr int LDS_BYTES = 2 * STAGE_BYTES;

// ---- row quantiser: one wave per row, the row cached in registers (bf16x8 chunks) ----
template <int NC>
__global__ __launch_bounds__(256) void quant_rows_fp8_kernel(const bf16_t* __restrict__ X, long ldx,
                                                             uint8_t* __restrict__ Q, long ldq,
                                                             float* __restrict__ S, int rows, int cols) {
  const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + w;
  if (row >= rows) return;
  const bf16_t* x = X + (long)row * ldx;
  const int nch = cols >> 3;
  bf16x8 c[NC];
  float amax = 0.f;
#pragma unroll
  for (int i = 0; i < NC; ++i) {
    const int ch = i * 64 + l;
    if (ch < nch) {
      c[i] = *(const bf16x8*)(x + ch * 8);
#pragma unroll
      for (int e = 0; e < 8; ++e) amax = fmaxf(amax, fabsf(bf2f(c[i][e])));
    }
  }
  amax = wave_max(amax);
  const float scale = amax > 0.f ? amax / 448.0f : 1.0f;
  if (l == 0) S[row] = scale;
  uint8_t* q = Q + (long)row * ldq;
#pragma unroll
  for (int i = 0; i < NC; ++i) {
    const int ch = i * 64 + l;
    if (ch < nch) {
      float v[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = fminf(fmaxf(bf2f(c[i][e]) / scale, -448.0f), 448.0f);
      int lo = 0, hi = 0;
      lo = __builtin_amdgcn_cvt_pk_fp8_f32(v[0], v[1], lo, false);
      lo = __builtin_amdgcn_cvt_pk_fp8_f32(v[2], v[3], lo, true);
      hi = __builtin_amdgcn_cvt_pk_fp8_f32(v[4], v[5], hi, false);
      hi = __builtin_amdgcn_cvt_pk_fp8_f32(v[6], v[7], hi, true);
      *(int2*)(q + ch * 8) = make_int2(lo, hi);
    }
  }
}

// wide rows (cols > 4096): one 256-thread block per row, up to 16 chunks of 8 per thread, amax through LDS
__global__ __launch_bounds__(256) void quant_rows_fp8_wide_kernel(const bf16_t* __restrict__ X, long ldx,
                                                                  uint8_t* __restrict__ Q, long ldq,
                                                                  float* __restrict__ S, int rows, int cols) {
  __shared__ float red[4];
  const int t = threadIdx.x, row = blockIdx.x;
  const bf16_t* x = X + (long)row * ldx;
  const int nch = cols >> 3;
  bf16x8 c[16];
  float amax = 0.f;
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int ch = i * 256 + t;
    if (ch < nch) {
      c[i] = *(const bf16x8*)(x + ch * 8);
#pragma unroll
      for (int e = 0; e < 8; ++e) amax = fmaxf(amax, fabsf(bf2f(c[i][e])));
    }
  }
  amax = wave_max(amax);
  if ((t & 63) == 0) red[t >> 6] = amax;
  __syncthreads();
  amax = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  const float scale = amax > 0.f ? amax / 448.0f : 1.0f;
  if (t == 0) S[row] = scale;
  uint8_t* q = Q + (long)row * ldq;
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int ch = i * 256 + t;
    if (ch < nch) {
      float v[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = fminf(fmaxf(bf2f(c[i][e]) / scale, -448.0f), 448.0f);
      int lo = 0, hi = 0;
      lo = __builtin_amdgcn_cvt_pk_fp8_f32(v[0], v[1], lo, false);
      lo = __builtin_amdgcn_cvt_pk_fp8_f32(v[2], v[3], lo, true);
      hi = __builtin_amdgcn_cvt_pk_fp8_f32(v[4], v[5], hi, false);
      hi = __builtin_amdgcn_cvt_pk_fp8_f32(v[6], v[7], hi, true);
      *(int2*)(q + ch * 8) = make_int2(lo, hi);
    }
  }
}

// ---- GEMM ----
template <int EPI>
__global__ __launch_bounds__(512) void gemm_fp8_nt_256_kernel(
    const uint8_t* __restrict__ A, long lda, const float* __restrict__ SA, const uint8_t* __restrict__ W, long ldw,
    const float* __restrict__ SW, const bf16_t* __restrict__ bias, const bf16_t* R, long ldr, void* Cv, long ldc,
    int M, int N, int K, int tiles_m, int tiles_n, owc_gemm_aux aux) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int tid = threadIdx.x;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l = tid & 63;
  const int nblk = tiles_m * tiles_n;
  const int nk = K / BKB;

  // tile id -> (m0, n0): the blocks of one XCD (id & 7) walk a contiguous, GROUP_M-major range of tiles
  int m0, n0;
  {
    const int bid = blockIdx.x;
    const int q = nblk >> 3, r = nblk & 7;
    const int xcd = bid & 7, idx = bid >> 3;
    const int lid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    const int width = GROUP_M2 * tiles_n;
    const int group = lid / width;
    const int first_m = group * GROUP_M2;
    const int gsize = min(tiles_m - first_m, GROUP_M2);
    m0 = (first_m + (lid % width) % gsize) * BT;
    n0 = ((lid % width) / gsize) * BT;
  }
  // DMA sources: wave w stages rows [32w, 32w+32) of both operand tiles (4 pieces of 8 rows x 128 B)
  const char* abase = (const char*)(A + (long)m0 * lda);
  const char* wbase = (const char*)(W + (long)n0 * ldw);
  unsigned aoff[4], woff[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int row = 32 * w + 8 * j + (l >> 3);
    const int c = (l & 7) ^ ((row >> 1) & 7);
    aoff[j] = (unsigned)((long)min(row, M - 1 - m0) * lda + c * 16);
    woff[j] = (unsigned)((long)min(row, N - 1 - n0) * ldw + c * 16);
  }
  // K-tile bases pinned in SGPRs and the lane offsets kept 32-bit at the point of use: the DMA takes the scalar-base + lane-offset form
  // instead of a v_lshl_add_u64 per piece (vector work beside MFMAs is paid in full, see gemm_bf16.hip)
  auto stage = [&](int buf, int kt) {
    char* la = lds + buf * STAGE_BYTES + w * 4096;
    const char* ab = abase + (long)kt * BKB;
    const char* wb = wbase + (long)kt * BKB;
    asm volatile("" : "+s"(ab), "+s"(wb));
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      asm volatile("" : "+v"(aoff[j]), "+v"(woff[j]));
      glds16(ab + aoff[j], la + j * 1024);
      glds16(wb + woff[j], la + OP_BYTES + j * 1024);
    }
  };
  auto stage_piece = [&](int buf, int kt, int j) {
    char* la = lds + buf * STAGE_BYTES + w * 4096;
    const char* ab = abase + (long)kt * BKB;
    const char* wb = wbase + (long)kt * BKB;
    asm volatile("" : "+s"(ab), "+s"(wb), "+v"(aoff[j]), "+v"(woff[j]));
    glds16(ab + aoff[j], la + j * 1024);
    glds16(wb + woff[j], la + OP_BYTES + j * 1024);
  };

  const int wr = w >> 2, wc = w & 3;
  const int fr = l & 15, fq = l >> 4;
  const int swz = (fr >> 1) & 7;
  const int rowA = (wr * 128 + fr) * 128;
  const int rowW = OP_BYTES + (wc * 64 + fr) * 128;
  // the lane's 32 operand bytes = 16-B chunks fq and 4 + fq of the 128-byte K-tile row (any fixed pairing of bytes with operand
  // slots works as long as A and W use the same one; this one makes a wave-instruction of the skinny kernel below read 64
  // contiguous bytes per row, and both kernels must agree to stay bit-identical)
  const int chlo = (fq ^ swz) << 4, chhi = ((4 + fq) ^ swz) << 4;

  f32x4 acc[4][8];  // [nt][mt]
  i32x8 xa[2], ya[2], wk[4];  // A fragments of two m tiles (double-buffered), W fragments of the K-tile

  typedef int i32x4 __attribute__((ext_vector_type(4)));
  auto rd = [&](i32x8& dst, const char* p) {
    const i32x4 lo = *(const i32x4*)(p + chlo), hi = *(const i32x4*)(p + chhi);
    dst = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
  };
  auto read_a = [&](i32x8 (&dst)[2], const char* sbase, int quarter) {
#pragma unroll
    for (int t = 0; t < 2; ++t) rd(dst[t], sbase + rowA + (quarter * 2 + t) * 2048);
  };
  auto mfma = [&](const i32x8& wf, const i32x8& af, f32x4& c) {
    c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(wf, af, c, 0, 0, 0, 0x7f, 0, 0x7f);
  };
  // 8 MFMAs: m tiles (2q, 2q+1) x the 4 n tiles
  auto phase = [&](const i32x8 (&af)[2], int q) {
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int n = 0; n < 4; ++n)
#pragma unroll
      for (int m = 0; m < 2; ++m) mfma(wk[n], af[m], acc[n][q * 2 + m]);
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
  };

  stage(0, 0);
  if (nk > 1) {
    stage(1, 1);
    asm volatile("s_waitcnt vmcnt(8)\n\ts_barrier" ::: "memory");
  } else {
    asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
  }
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  read_a(xa, lds, 0);
#pragma unroll
  for (int t = 0; t < 4; ++t) rd(wk[t], lds + rowW + t * 2048);

  // One K-tile; ISSUE is a compile-time flag (a run-time "if (issue)" around the DMA splits the body into basic blocks and the
  // compiler then sinks every MFMA behind the last of them - measured 4x slower), so the last two K-tiles are peeled.
  auto ktile = [&](int kt, auto issue_tag) {
    constexpr bool ISSUE = decltype(issue_tag)::value;
    const char* cur = lds + (kt & 1) * STAGE_BYTES;
    const char* nxt = lds + ((kt + 1) & 1) * STAGE_BYTES;
    read_a(ya, cur, 1);
    phase(xa, 0);
    read_a(xa, cur, 2);
    phase(ya, 1);
    read_a(ya, cur, 3);
    phase(xa, 2);
    // all LDS reads of stage kt by this wave are complete and its DMA pieces of stage kt+1 have landed: the barrier publishes
    // stage kt+1 and frees stage kt's buffer, into which the DMA of stage kt+2 goes during the last phase
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    read_a(xa, nxt, 0);  // (after the last K-tile this reads stale LDS that nothing uses)
    // phase 3: m tiles 6, 7; after its last use each W fragment is refilled from stage kt+1, DMA pieces in between
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int n = 0; n < 4; ++n) {
      __builtin_amdgcn_s_setprio(1);
      mfma(wk[n], ya[0], acc[n][6]);
      mfma(wk[n], ya[1], acc[n][7]);
      __builtin_amdgcn_s_setprio(0);
      __builtin_amdgcn_sched_barrier(0);
      rd(wk[n], nxt + rowW + n * 2048);
      if constexpr (ISSUE) stage_piece(kt & 1, kt + 2, n);
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  int kt = 0;
  for (; kt + 2 < nk; ++kt) ktile(kt, std::true_type{});
  for (; kt < nk; ++kt) ktile(kt, std::false_type{});

  // dequantise: acc * sa[m] * sw[n], then the shared bf16 epilogue (bias / activation / residual / SwiGLU, LDS-staged rows)
  {
    f32x4 swv[4];
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) swv[nt] = *(const f32x4*)(SW + min(n0 + wc * 64 + nt * 16 + fq * 4, N - 4));
#pragma unroll
    for (int mt = 0; mt < 8; ++mt) {
      const float sa = SA[min(m0 + wr * 128 + mt * 16 + fr, M - 1)];
#pragma unroll
      for (int nt = 0; nt < 4; ++nt)
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[nt][mt][e] = acc[nt][mt][e] * sa * swv[nt][e];
    }
  }
  constexpr int CCOLS = EPI == OWC_EPI_SWIGLU ? BT / 2 : BT;
  gemm_epilogue<EPI, 8>(acc, m0 + wr * 128, n0 + wc * 64, fr, fq, bias, R, ldr, Cv, ldc, M, N, aux, lds, CCOLS * 2, wr * 128,
                        wc * 64);
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
  store_ctile<8>(lds, BT, CCOLS, (bf16_t*)Cv, ldc, m0, EPI == OWC_EPI_SWIGLU ? (n0 >> 1) : n0, M,
                 EPI == OWC_EPI_SWIGLU ? (N >> 1) : N, w, l);
}


// ---- ping-pong variant of the 256x256x128 kernel: the schedule of gemm_bf16_nt_256pp_kernel (gemm_bf16.hip: two waves per SIMD
// alternate MFMA / load roles, counted vmcnt(6), half-tile LDS-DMA) with the arithmetic swapped: a K-tile is 128 fp8 elements =
// the same 128-byte rows and LDS image, a lane's operand is 32 consecutive K bytes (two swizzled ds_read_b128), one scaled MFMA
// per 16x16 tile per K-tile, so a C quadrant (4 m tiles x 2 n tiles) is 8 MFMAs of 32 cycles = the bf16 quadrant's 256 cycles.
// Same ascending chain per output element: bit-identical to gemm_fp8_nt_256_kernel.  Requires at least two K-tiles.
template <int EPI>
__global__ __launch_bounds__(512) void gemm_fp8_nt_256pp_kernel(
    const uint8_t* __restrict__ A, long lda, const float* __restrict__ SA, const uint8_t* __restrict__ W, long ldw,
    const float* __restrict__ SW, const bf16_t* __restrict__ bias, const bf16_t* R, long ldr, void* Cv, long ldc,
    int M, int N, int K, int tiles_m, int tiles_n, owc_gemm_aux aux, int nt_store) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int tid = threadIdx.x;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l = tid & 63;
  const int nblk = tiles_m * tiles_n;
  const int nk = K / BKB;
  int m0, n0;
  {
    const int bid = blockIdx.x;
    const int q = nblk >> 3, r = nblk & 7;
    const int xcd = bid & 7, idx = bid >> 3;
    const int lid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    const int width = GROUP_M2 * tiles_n;
    const int group = lid / width;
    const int first_m = group * GROUP_M2;
    const int gsize = min(tiles_m - first_m, GROUP_M2);
    m0 = (first_m + (lid % width) % gsize) * BT;
    n0 = ((lid % width) / gsize) * BT;
  }
  const char* abase = (const char*)(A + (long)m0 * lda);
  const char* wbase = (const char*)(W + (long)n0 * ldw);
  unsigned aoff[2][2], woff[2][2];   // half h, piece j -> rows 128h + 16w + 8j .. +8
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int row = 128 * h + 16 * w + 8 * j + (l >> 3);
      const int c = (l & 7) ^ ((row >> 1) & 7);
      aoff[h][j] = (unsigned)((long)min(row, M - 1 - m0) * lda + c * 16);
      woff[h][j] = (unsigned)((long)min(row, N - 1 - n0) * ldw + c * 16);
    }
  // scalar K-tile base + 32-bit lane offsets, pinned (as in gemm_bf16_nt_256pp_kernel): without it the compiler keeps a 64-bit
  // per-lane pointer per piece alive across the loop - this kernel then needs 256 VGPRs + 470-680 bytes of scratch per lane
  auto issue_a = [&](int kt, int h) {
    char* dst = lds + (kt & 1) * STAGE_BYTES + (128 * h + 16 * w) * 128;
    const char* ab = abase + (long)kt * BKB;
    asm volatile("" : "+s"(ab), "+v"(aoff[h][0]), "+v"(aoff[h][1]));
    glds16(ab + aoff[h][0], dst);
    glds16(ab + aoff[h][1], dst + 1024);
  };
  auto issue_w = [&](int kt, int h) {
    char* dst = lds + (kt & 1) * STAGE_BYTES + OP_BYTES + (128 * h + 16 * w) * 128;
    const char* wb = wbase + (long)kt * BKB;
    asm volatile("" : "+s"(wb), "+v"(woff[h][0]), "+v"(woff[h][1]));
    glds16(wb + woff[h][0], dst);
    glds16(wb + woff[h][1], dst + 1024);
  };
  const int wr = w >> 2, wc = w & 3;
  const int fr = l & 15, fq = l >> 4;
  const int swz = (fr >> 1) & 7;
  const int rowA = (wr * 128 + fr) * 128;
  const int rowW = OP_BYTES + (wc * 64 + fr) * 128;
  const int chlo = (fq ^ swz) << 4, chhi = ((4 + fq) ^ swz) << 4;   // the operand byte pairing of gemm_fp8_nt_256_kernel

  f32x4 acc[4][8];  // [nt][mt]
  i32x8 fa[4], wy[2], wx0[2], wx1[2];   // A m-half, W column half 1, and TWO sets of W column half 0 (this K-tile's / the next one's)
  typedef int i32x4 __attribute__((ext_vector_type(4)));
  auto rd = [&](i32x8& dst, const char* p) {
    const i32x4 lo = *(const i32x4*)(p + chlo), hi = *(const i32x4*)(p + chhi);
    dst = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
  };
  auto read_a = [&](const char* sbase, int mh) {
#pragma unroll
    for (int t = 0; t < 4; ++t) rd(fa[t], sbase + rowA + (mh * 4 + t) * 2048);
  };
  auto read_w = [&](i32x8 (&dst)[2], const char* sbase, int nh) {
#pragma unroll
    for (int t = 0; t < 2; ++t) rd(dst[t], sbase + rowW + (nh * 2 + t) * 2048);
  };
#define OWC_PP_SYNC_L(VM)                                                                         \
  do {                                                                                            \
    if constexpr ((VM) >= 0) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((VM) < 0 ? 0 : (VM)) : "memory"); \
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");                               \
    __builtin_amdgcn_sched_barrier(0);                                                            \
  } while (0)
  auto quadrant = [&](const i32x8 (&wf)[2], int mh, int nh) {
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int n = 0; n < 2; ++n)
#pragma unroll
      for (int m = 0; m < 4; ++m)
        acc[nh * 2 + n][mh * 4 + m] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(wf[n], fa[m], acc[nh * 2 + n][mh * 4 + m],
                                                                                         0, 0, 0, 0x7f, 0, 0x7f);
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_barrier" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
  };
  // the schedule of gemm_bf16_nt_256pp_kernel phase for phase: p0 reads A (8 ds_read_b128), p1 W half 1 (4), p2 A (8), p3 the NEXT
  // tile's W half 0 into the other register set (4)
  auto ktile = [&](auto mode_c, int u, i32x8 (&wcur)[2], i32x8 (&wnxt)[2]) {
    constexpr int MODE = decltype(mode_c)::value;   // 0 steady, 1 second to last, 2 last K-tile
    const char* cur = lds + (u & 1) * STAGE_BYTES;
    const char* nxt = lds + ((u + 1) & 1) * STAGE_BYTES;
    read_a(cur, 0);
    OWC_PP_SYNC_L(-1);
    quadrant(wcur, 0, 0);
    read_w(wy, cur, 1);
    if constexpr (MODE <= 1) issue_a(u + 1, 1);
    OWC_PP_SYNC_L(-1);
    quadrant(wy, 0, 1);
    read_a(cur, 1);
    if constexpr (MODE == 0) {
      issue_w(u + 2, 0);
      OWC_PP_SYNC_L(6);
    } else if constexpr (MODE == 1) {
      OWC_PP_SYNC_L(4);
    } else {
      OWC_PP_SYNC_L(-1);
    }
    quadrant(wy, 1, 1);
    if constexpr (MODE <= 1) read_w(wnxt, nxt, 0);
    if constexpr (MODE == 0) {
      issue_w(u + 2, 1);
      issue_a(u + 2, 0);
      OWC_PP_SYNC_L(6);
    } else if constexpr (MODE == 1) {
      OWC_PP_SYNC_L(0);
    } else {
      OWC_PP_SYNC_L(-1);
    }
    quadrant(wcur, 1, 0);
  };
  issue_a(0, 0);
  issue_a(0, 1);
  issue_w(0, 0);
  issue_w(0, 1);
  issue_w(1, 0);
  issue_w(1, 1);
  issue_a(1, 0);
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  asm volatile("s_waitcnt vmcnt(6)\n\ts_barrier" ::: "memory");
  __builtin_amdgcn_sched_barrier(0);
  read_w(wx0, lds, 0);
  if (wr) {
    asm volatile("s_barrier" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
  }
  using I0 = std::integral_constant<int, 0>;
  using I1 = std::integral_constant<int, 1>;
  using I2 = std::integral_constant<int, 2>;
  int u = 0;   // K-tiles in pairs (the two W half-0 sets alternate): an EVEN number of K-tiles, checked by the launcher.  (A second
               // peeled tail for odd counts makes the register allocator park the accumulators in scratch: 660-710 bytes per lane.
               // The one odd case on the path, the 72B down projection's K = 29568 = 231 tiles, is zero-padded to 232 at load time.)
  for (; u + 2 < nk; u += 2) {
    ktile(I0{}, u, wx0, wx1);
    ktile(I0{}, u + 1, wx1, wx0);
  }
  ktile(I1{}, u, wx0, wx1);
  ktile(I2{}, u + 1, wx1, wx0);
  if (!wr) {
    asm volatile("s_barrier" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
  }
#undef OWC_PP_SYNC_L
  // Pin the accumulators HERE.  MFMAs have no side effects the barriers order, so LLVM's sinking pass moved the 32 MFMAs of the
  // last K-tile down into the 16 conditional store blocks of the epilogue (two MFMAs per block, their operands parked in scratch:
  // 470-680 bytes per lane, and the last two of K / 128 K-tiles ran at a fraction of the loop's speed) - the reason this kernel
  // lost to the lock-step one in round 2.  With the values demanded at this point the tail stays a K-tile.
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) asm volatile("" : "+v"(acc[i][j]));
  {
    f32x4 swv[4];
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) swv[nt] = *(const f32x4*)(SW + min(n0 + wc * 64 + nt * 16 + fq * 4, N - 4));
#pragma unroll
    for (int mt = 0; mt < 8; ++mt) {
      const float sa = SA[min(m0 + wr * 128 + mt * 16 + fr, M - 1)];
#pragma unroll
      for (int nt = 0; nt < 4; ++nt)
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[nt][mt][e] = acc[nt][mt][e] * sa * swv[nt][e];
    }
  }
  constexpr int CCOLS = EPI == OWC_EPI_SWIGLU ? BT / 2 : BT;
  gemm_epilogue<EPI, 8>(acc, m0 + wr * 128, n0 + wc * 64, fr, fq, bias, R, ldr, Cv, ldc, M, N, aux, lds, CCOLS * 2, wr * 128,
                        wc * 64);
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
  if (nt_store)   // streaming (non-temporal) C stores for outputs far larger than the caches (as in the bf16 kernel: `gemm_nt_min_mb`)
    store_ctile<8, true>(lds, BT, CCOLS, (bf16_t*)Cv, ldc, m0, EPI == OWC_EPI_SWIGLU ? (n0 >> 1) : n0, M,
                         EPI == OWC_EPI_SWIGLU ? (N >> 1) : N, w, l);
  else
    store_ctile<8>(lds, BT, CCOLS, (bf16_t*)Cv, ldc, m0, EPI == OWC_EPI_SWIGLU ? (n0 >> 1) : n0, M,
                   EPI == OWC_EPI_SWIGLU ? (N >> 1) : N, w, l);
}


// ---- 64x64x128 variant for the in-between shapes (M above the skinny kernel's 64 rows, too few 256x256 tiles for the 256 CUs:
// fp8 decode at batch 65 .. ~1000, single-prompt prefill).  The fp8 twin of gemm_bf16_nt_64_kernel: 4 waves, wave w owns rows
// [16w, 16w+16) x 64 columns, one scaled MFMA per n tile per K-tile, a four-stage LDS-DMA ring (64 KiB), same ascending chain (bit-identical).
constexpr int B64 = 64;
constexpr int TILE64_BYTES = B64 * BKB;  // 8 KiB per operand tile

// Round 3: tile shape TM x TN and KPS K-tiles per stage as in gemm_bf16_nt_64_kernel (the measurements are in DESIGN.md section 4, "What
// one CU can pull"): a launch that cannot fill the chip with 64x64 tiles takes 32x32 / 64x32 tiles, <= 128 rows x tens of thousands of
// columns (gate/up) take 32/64/128 x 256 tiles (one block per CU, A re-read once per CU), and long K takes several K-tiles behind one
// wait + barrier.  TM / 16 waves own a 16-row m tile x all TN columns (TM = 128: two m tiles per wave).  Same chain: bit-identical.
template <int EPI, int NS8 = 4, int TM = 64, int TN = 64, int KPS = 1>
__global__ __launch_bounds__(256) void gemm_fp8_nt_64_kernel(
    const uint8_t* __restrict__ A, long lda, const float* __restrict__ SA, const uint8_t* __restrict__ W, long ldw,
    const float* __restrict__ SW, const bf16_t* __restrict__ bias, const bf16_t* R, long ldr, void* Cv, long ldc,
    int M, int N, int K, int tiles_m, int tiles_n, owc_gemm_aux aux) {
  extern __shared__ __attribute__((aligned(16))) char lds[];  // [stage][K-tile][A: TM rows | W: TN rows][128 B]
  constexpr int SUB = (TM + TN) * BKB, STAGE = KPS * SUB;
  constexpr int NPA = TM / 8, NPT = (TM + TN) / 8, NPW = NPT / 4;   // 1-KiB DMA pieces of a K-tile: of A, in all, per wave
  constexpr int NTW = TN / 16, MTW = TM > 64 ? TM / 64 : 1;
  static_assert(TN % 32 == 0 && (TM + TN) % 32 == 0 && NPA % 4 == 0, "whole tile pairs; the DMA pieces divide over four waves");
  const int tid = threadIdx.x;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l = tid & 63;
  const int nblk = tiles_m * tiles_n;
  const int bid = blockIdx.x;
  const int q = nblk >> 3, r = nblk & 7;
  const int xcd = bid & 7, idx = bid >> 3;
  const int lid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  const int width = 8 * tiles_n;
  const int group = lid / width;
  const int first_m = group * 8;
  const int gsize = min(tiles_m - first_m, 8);
  const int m0 = (first_m + (lid % width) % gsize) * TM, n0 = ((lid % width) / gsize) * TN;
  // staging: piece p = w + 4 i (i < NPW) of a K-tile; pieces [0, NPA) are 8-row slabs of the A tile, the rest of the W tile; scalar
  // tile bases + 32-bit lane offsets (the DMA's scalar-base form: no vector address arithmetic in the loop)
  const char* abase = (const char*)(A + (long)m0 * lda);
  const char* wbase = (const char*)(W + (long)n0 * ldw);
  unsigned off[NPW];
#pragma unroll
  for (int i = 0; i < NPW; ++i) {
    const bool is_a = 4 * i < NPA;
    const int row = 8 * (w + 4 * i - (is_a ? 0 : NPA)) + (l >> 3);
    const int c = (l & 7) ^ ((row >> 1) & 7);
    off[i] = is_a ? (unsigned)((long)min(row, M - 1 - m0) * lda + c * 16) : (unsigned)((long)min(row, N - 1 - n0) * ldw + c * 16);
  }
  const int nk = K / BKB;
  const int nst = (nk + KPS - 1) / KPS;
  // ring of NS8 stages: NS8 - 1 stages in flight behind a counted vmcnt, one raw barrier per stage; the pieces issued past the end of K
  // re-read the last K-tile (nobody reads them)
  auto stage = [&](int buf, int st) {
    char* dst = lds + buf * STAGE + w * 1024;
#pragma unroll
    for (int u = 0; u < KPS; ++u) {
      const long kb = (long)min(st * KPS + u, nk - 1) * BKB;
      const char* ab = abase + kb;
      const char* wb = wbase + kb;
      asm volatile("" : "+s"(ab), "+s"(wb));
#pragma unroll
      for (int i = 0; i < NPW; ++i) {
        asm volatile("" : "+v"(off[i]));
        glds16((4 * i < NPA ? ab : wb) + off[i], dst + u * SUB + i * 4096);
      }
    }
  };
  const int fr = l & 15, fq = l >> 4;
  const int swz = (fr >> 1) & 7;
  const int chlo = (fq ^ swz) << 4, chhi = ((4 + fq) ^ swz) << 4;  // the operand byte pairing of the tiled kernel
  typedef int i32x4 __attribute__((ext_vector_type(4)));
  auto rd = [&](const char* p) -> i32x8 {
    const i32x4 lo = *(const i32x4*)(p + chlo), hi = *(const i32x4*)(p + chhi);
    return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
  };
  const bool active = w * 16 * MTW < TM;
  f32x4 acc[NTW < 4 ? 4 : NTW][MTW];
#pragma unroll
  for (int i = 0; i < (NTW < 4 ? 4 : NTW); ++i)
#pragma unroll
    for (int j = 0; j < MTW; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int i = 0; i < NS8 - 1; ++i) stage(i, i);
  for (int st = 0; st < nst; ++st) {
    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(KPS * NPW * (NS8 - 2)) : "memory");   // lgkmcnt(0): WAR on the slot restaged next, in the source
    __builtin_amdgcn_sched_barrier(0);
    stage((st + NS8 - 1) % NS8, st + NS8 - 1);
    if (active) {
#pragma unroll
      for (int u = 0; u < KPS; ++u) {
        if (KPS > 1 && st * KPS + u >= nk) break;
        const char* la = lds + (st % NS8) * STAGE + u * SUB;
        i32x8 fa[MTW];
#pragma unroll
        for (int mt = 0; mt < MTW; ++mt) fa[mt] = rd(la + (w * 16 * MTW + mt * 16 + fr) * 128);
#pragma unroll
        for (int nt = 0; nt < NTW; ++nt) {
          const i32x8 fw = rd(la + TM * BKB + (nt * 16 + fr) * 128);
#pragma unroll
          for (int mt = 0; mt < MTW; ++mt)
            acc[nt][mt] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(fw, fa[mt], acc[nt][mt], 0, 0, 0, 0x7f, 0, 0x7f);
        }
      }
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (!active) return;
#pragma unroll
  for (int mt = 0; mt < MTW; ++mt) {
    const float sa = SA[min(m0 + w * 16 * MTW + mt * 16 + fr, M - 1)];
#pragma unroll
    for (int nt = 0; nt < NTW; ++nt) {
      const f32x4 swv = *(const f32x4*)(SW + min(n0 + nt * 16 + fq * 4, N - 4));
#pragma unroll
      for (int e = 0; e < 4; ++e) acc[nt][mt][e] = acc[nt][mt][e] * sa * swv[e];
    }
  }
  if constexpr (NTW <= 4) {
    gemm_epilogue<EPI, MTW, false, NTW / 2>(*(const f32x4(*)[4][MTW])&acc[0], m0 + w * 16 * MTW, n0, fr, fq, bias, R, ldr, Cv, ldc, M, N, aux);
  } else {
#pragma unroll
    for (int g = 0; g < NTW / 4; ++g)
      gemm_epilogue<EPI, MTW, false, 2>(*(const f32x4(*)[4][MTW])&acc[4 * g], m0 + w * 16 * MTW, n0 + 64 * g, fr, fq, bias, R, ldr, Cv, ldc, M,
                                        N, aux);
    if constexpr (NTW % 4 != 0)
      gemm_epilogue<EPI, MTW, false, 1>(*(const f32x4(*)[4][MTW])&acc[NTW - 2], m0 + w * 16 * MTW, n0 + 16 * (NTW - 2), fr, fq, bias, R, ldr, Cv,
                                        ldc, M, N, aux);
  }
}

// ---- skinny-M variant (M <= 64, decode at small batch): the fp8 twin of gemm_bf16_skinny_kernel.  One wave owns 16 rows of W8
// (32 for SwiGLU) over the whole K; a super-step is 128 fp8 elements = 32 bytes per lane and ONE scaled MFMA per (m tile, n
// tile); weights and activation codes ride a fully unrolled register ring.  One ascending accumulation chain per output, same
// operand roles as the tiled kernel: bit-identical to it.
template <int EPI, int MT, int DEPTH>
__global__ __launch_bounds__(64) void gemm_fp8_skinny_kernel(
    const uint8_t* __restrict__ A, long lda, const float* __restrict__ SA, const uint8_t* __restrict__ W, long ldw,
    const float* __restrict__ SW, const bf16_t* __restrict__ bias, const bf16_t* R, long ldr, bf16_t* C, long ldc, int M,
    int N, int K) {
  constexpr int NT = EPI == OWC_EPI_SWIGLU ? 2 : 1;
  const int l = threadIdx.x;
  const int fr = l & 15, fq = l >> 4;
  const int n0 = blockIdx.x * (16 * NT);
  const int nss = K >> 7;
  const uint8_t* wrow[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) wrow[nt] = W + (long)(n0 + nt * 16 + fr) * ldw + fq * 16;
  const uint8_t* arow[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) arow[mt] = A + (long)min(mt * 16 + fr, M - 1) * lda + fq * 16;
  f32x4 acc[MT][NT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
  typedef int i32x4 __attribute__((ext_vector_type(4)));
  i32x8 ring[DEPTH][NT], xring[DEPTH][MT];
  auto ld32 = [&](const uint8_t* p) -> i32x8 {
    const i32x4 lo = *(const i32x4*)p, hi = *(const i32x4*)(p + 64);  // chunks fq and 4 + fq, as in the tiled kernel
    return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
  };
  auto load = [&](int i, int ss) {
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) ring[i][nt] = ld32(wrow[nt] + ss * 128);
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) xring[i][mt] = ld32(arow[mt] + ss * 128);
  };
  auto compute = [&](int i) {
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
        acc[mt][nt] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(ring[i][nt], xring[i][mt], acc[mt][nt], 0, 0, 0, 0x7f, 0, 0x7f);
  };
  // Branch-free steady state (see gemm_bf16_skinny_kernel): loads past the end re-read the last super-step, whole rounds run
  // unconditionally, the last partial round only computes.
#pragma unroll
  for (int i = 0; i < DEPTH; ++i) load(i, min(i, nss - 1));
  int ss = 0;
  for (; ss + DEPTH <= nss; ss += DEPTH) {
#pragma unroll
    for (int i = 0; i < DEPTH; ++i) {
      compute(i);
      load(i, min(ss + i + DEPTH, nss - 1));
    }
  }
  const int rem = nss - ss;
#pragma unroll
  for (int i = 0; i < DEPTH; ++i)
    if (i < rem) compute(i);
  // dequantise + epilogue: lane holds row m = 16 mt + fr, columns n0 + 16 nt + 4 fq .. +3
  f32x4 swv[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) swv[nt] = *(const f32x4*)(SW + n0 + nt * 16 + fq * 4);
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    const int m = mt * 16 + fr;
    const float sa = SA[min(m, M - 1)];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int e = 0; e < 4; ++e) acc[mt][nt][e] = acc[mt][nt][e] * sa * swv[nt][e];
    bf16x4 o;
    if constexpr (EPI == OWC_EPI_SWIGLU) {
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = f2bf(rbf(act_silu(rbf(acc[mt][0][e]))) * rbf(acc[mt][1][e]));
      if (m < M) *(bf16x4*)(C + (long)m * ldc + (n0 >> 1) + fq * 4) = o;
    } else {
      const int n = n0 + fq * 4;
      float bv[4] = {0.f, 0.f, 0.f, 0.f};
      if (bias != nullptr) {
        const bf16x4 b = *(const bf16x4*)(bias + n);
#pragma unroll
        for (int e = 0; e < 4; ++e) bv[e] = bf2f(b[e]);
      }
      if constexpr (EPI == OWC_EPI_RESIDUAL) {
        const bf16x4 r = *(const bf16x4*)(R + (long)min(m, M - 1) * ldr + n);
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = f2bf(rbf(acc[mt][0][e] + bv[e]) + bf2f(r[e]));
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = f2bf(acc[mt][0][e] + bv[e]);
      }
      if (m < M) *(bf16x4*)(C + (long)m * ldc + n) = o;
    }
  }
}

template <int EPI>
bool launch_fp8_skinny(const void* A, long lda, const float* sa, const void* W, long ldw, const float* sw, const void* bias,
                       const void* R, long ldr, void* C, long ldc, int M, int N, int K, hipStream_t s) {
  constexpr int ROWS = EPI == OWC_EPI_SWIGLU ? 32 : 16;
  if (M > 64 || (N % ROWS) || g_fp8_skinny_max_m < M) return false;
  const dim3 grid(N / ROWS), block(64);
#define OWC_SK8(MT_, D_)                                                                                               \
  hipLaunchKernelGGL((gemm_fp8_skinny_kernel<EPI, MT_, D_>), grid, block, 0, s, (const uint8_t*)A, lda, sa,              \
                     (const uint8_t*)W, ldw, sw, (const bf16_t*)bias, (const bf16_t*)R, ldr, (bf16_t*)C, ldc, M, N, K)
  constexpr int NT_ = EPI == OWC_EPI_SWIGLU ? 2 : 1;
  // 8 VGPRs per (n tile + m tile) per super-step in flight, about 192 VGPRs of ring
  if (M <= 16) OWC_SK8(1, 24 / (NT_ + 1)); else if (M <= 32) OWC_SK8(2, 24 / (NT_ + 2)); else if (M <= 48) OWC_SK8(3, 24 / (NT_ + 3));
  else OWC_SK8(4, 24 / (NT_ + 4));
#undef OWC_SK8
  return true;
}

template <int EPI>
int launch_fp8(const void* A, long lda, const float* sa, const void* W, long ldw, const float* sw, const void* bias,
               const void* R, long ldr, void* C, long ldc, int M, int N, int K, hipStream_t s) {
  static bool attr_set = false;
  if (!attr_set) {
    if (hipFuncSetAttribute((const void*)gemm_fp8_nt_256_kernel<EPI>, hipFuncAttributeMaxDynamicSharedMemorySize,
                            LDS_BYTES) != hipSuccess ||
        hipFuncSetAttribute((const void*)gemm_fp8_nt_256pp_kernel<EPI>, hipFuncAttributeMaxDynamicSharedMemorySize,
                            LDS_BYTES) != hipSuccess ||
        false)
      return OWC_ERR_HIP;
    attr_set = true;
  }
  const int tiles_m = (M + BT - 1) / BT, tiles_n = (N + BT - 1) / BT;
  const owc_gemm_aux aux = {nullptr, nullptr, nullptr, 0, 8};
  const int prof = owc_gemm_profile_begin(2.0 * (double)M * (double)N * (double)K, 1, s);
  if (launch_fp8_skinny<EPI>(A, lda, sa, W, ldw, sw, bias, R, ldr, C, ldc, M, N, K, s)) {
    owc_gemm_profile_end(prof, s);
    return hipGetLastError() == hipSuccess ? OWC_OK : OWC_ERR_HIP;
  }
#define OWC_L8(NS_, TM_, TN_, KPS_)                                                                                            \
  do {                                                                                                                         \
    constexpr int lds_ = NS_ * KPS_ * (TM_ + TN_) * BKB;                                                                       \
    static bool set_ = false;                                                                                                  \
    if (!set_) {                                                                                                               \
      if (hipFuncSetAttribute((const void*)gemm_fp8_nt_64_kernel<EPI, NS_, TM_, TN_, KPS_>,                                    \
                              hipFuncAttributeMaxDynamicSharedMemorySize, lds_) != hipSuccess) return OWC_ERR_HIP;             \
      set_ = true;                                                                                                             \
    }                                                                                                                          \
    const int tm_ = (M + TM_ - 1) / TM_, tn_ = (N + TN_ - 1) / TN_;                                                            \
    hipLaunchKernelGGL((gemm_fp8_nt_64_kernel<EPI, NS_, TM_, TN_, KPS_>), dim3(tm_ * tn_), dim3(256), lds_, s, (const uint8_t*)A, \
                       lda, sa, (const uint8_t*)W, ldw, sw, (const bf16_t*)bias, (const bf16_t*)R, ldr, C, ldc, M, N, K, tm_,  \
                       tn_, aux);                                                                                              \
  } while (0)
  // <= 128 rows x tens of thousands of columns (the 72B gate/up projection: N = 59 392): 256-column tiles, one block per CU
  const int wide_blocks = (N + 255) / 256;
  if (g_fp8_shapes && M <= 128 && wide_blocks <= 256 && wide_blocks >= 128) {
    if (M <= 32) OWC_L8(4, 32, 256, 1); else if (M <= 64) OWC_L8(3, 64, 256, 1); else OWC_L8(3, 128, 256, 1);
    owc_gemm_profile_end(prof, s);
    return hipGetLastError() == hipSuccess ? OWC_OK : OWC_ERR_HIP;
  }
  if (g_fp8_mid_max_tiles > 0 && tiles_m * tiles_n < g_fp8_mid_max_tiles) {  // too few 256x256 tiles for the 256 CUs
    auto blocks = [&](int tm, int tn) { return (long)((M + tm - 1) / tm) * ((N + tn - 1) / tn); };
    const bool longk = g_fp8_shapes && K >= 2048;
    if (g_fp8_shapes && EPI != OWC_EPI_SWIGLU && blocks(64, 64) <= 256 && blocks(32, 32) <= 256) {
      if (longk) OWC_L8(4, 32, 32, 4); else OWC_L8(8, 32, 32, 1);
    } else if (g_fp8_shapes && EPI != OWC_EPI_SWIGLU && blocks(64, 64) <= 256 && blocks(64, 32) <= 256) {
      if (longk) OWC_L8(3, 64, 32, 4); else OWC_L8(6, 64, 32, 1);
    } else if (g_fp8_ring_128 && g_fp8_shapes && EPI != OWC_EPI_SWIGLU && blocks(64, 64) > 512 && M >= 129) {
      OWC_L8(3, 128, 64, 1);   // (round 4: 72 KiB of LDS, two blocks per CU: 512 blocks in one round)
    } else if (longk && blocks(64, 64) <= 512) {
      OWC_L8(3, 64, 64, 2);
    } else {
      OWC_L8(4, 64, 64, 1);
    }
    owc_gemm_profile_end(prof, s);
    return hipGetLastError() == hipSuccess ? OWC_OK : OWC_ERR_HIP;
  }
#undef OWC_L8
  if (g_fp8_pingpong && K >= 2 * BKB && (K % (2 * BKB)) == 0)
    hipLaunchKernelGGL(gemm_fp8_nt_256pp_kernel<EPI>, dim3(tiles_m * tiles_n), dim3(512), LDS_BYTES, s,
                       (const uint8_t*)A, lda, sa, (const uint8_t*)W, ldw, sw, (const bf16_t*)bias, (const bf16_t*)R, ldr, C,
                       ldc, M, N, K, tiles_m, tiles_n, aux,
                       (size_t)M * (size_t)(EPI == OWC_EPI_SWIGLU ? N / 2 : N) * 2 > ((size_t)owc_gemm_nt_min_mb() << 20) ? 1 : 0);
  else
    hipLaunchKernelGGL(gemm_fp8_nt_256_kernel<EPI>, dim3(tiles_m * tiles_n), dim3(512), LDS_BYTES, s,
                       (const uint8_t*)A, lda, sa, (const uint8_t*)W, ldw, sw, (const bf16_t*)bias, (const bf16_t*)R, ldr, C,
                       ldc, M, N, K, tiles_m, tiles_n, aux);
  owc_gemm_profile_end(prof, s);
  return hipGetLastError() == hipSuccess ? OWC_OK : OWC_ERR_HIP;
}

}  // namespace

void owc_gemm_fp8_set_skinny_max_m(int v) { g_fp8_skinny_max_m = v < 0 ? 0 : v; }  // negative: back to the default
void owc_gemm_fp8_set_pingpong(int v) { g_fp8_pingpong = v; }
void owc_gemm_fp8_set_shapes(int v) { g_fp8_shapes = v; }
void owc_gemm_fp8_set_ring_128(int v) { g_fp8_ring_128 = v < 0 ? 1 : v != 0; }
void owc_gemm_fp8_set_mid_max_tiles(int v) { g_fp8_mid_max_tiles = v ? 128 : 0; }

int owc_launch_quant_rows_fp8(const void* X, long ldx, void* Q, long ldq, float* S, int rows, int cols, hipStream_t st) {
  if (rows <= 0 || cols <= 0 || (cols & 7) || (ldx & 7) || (ldq & 7) || cols > 16 * 256 * 8) return OWC_ERR_SHAPE;
#define OWC_Q(C_)                                                                                                  \
  hipLaunchKernelGGL((quant_rows_fp8_kernel<C_>), dim3((rows + 3) / 4), dim3(256), 0, st, (const bf16_t*)X, ldx,     \
                     (uint8_t*)Q, ldq, S, rows, cols)
  if (cols <= 2048) OWC_Q(4);
  else if (cols <= 4096) OWC_Q(8);
  else
    hipLaunchKernelGGL(quant_rows_fp8_wide_kernel, dim3(rows), dim3(256), 0, st, (const bf16_t*)X, ldx, (uint8_t*)Q, ldq, S,
                       rows, cols);
#undef OWC_Q
  return hipGetLastError() == hipSuccess ? OWC_OK : OWC_ERR_HIP;
}

int owc_launch_gemm_fp8(const void* A, long lda, const float* sa, const void* W, long ldw, const float* sw,
                        const void* bias, const void* R, long ldr, void* C, long ldc, int M, int N, int K, int epi,
                        hipStream_t s) {
  if (M <= 0 || N <= 0 || K <= 0) return OWC_ERR_SHAPE;
  if ((K % BKB) || (lda & 15) || (ldw & 15) || (N & 7) || (ldc & 7)) return OWC_ERR_SHAPE;  // whole 128-byte K-tiles, 16-B rows
  if (epi == OWC_EPI_SWIGLU && (N & 31)) return OWC_ERR_SHAPE;
  if (epi == OWC_EPI_RESIDUAL && (R == nullptr || (ldr & 7))) return OWC_ERR_ARG;
  if (!sa || !sw) return OWC_ERR_ARG;
  switch (epi) {
    case OWC_EPI_NONE: return launch_fp8<OWC_EPI_NONE>(A, lda, sa, W, ldw, sw, bias, R, ldr, C, ldc, M, N, K, s);
    case OWC_EPI_RESIDUAL: return launch_fp8<OWC_EPI_RESIDUAL>(A, lda, sa, W, ldw, sw, bias, R, ldr, C, ldc, M, N, K, s);
    case OWC_EPI_SWIGLU: return launch_fp8<OWC_EPI_SWIGLU>(A, lda, sa, W, ldw, sw, bias, R, ldr, C, ldc, M, N, K, s);
    default: return OWC_ERR_ARG;  // the decoder needs no other epilogue on this path
  }
}
