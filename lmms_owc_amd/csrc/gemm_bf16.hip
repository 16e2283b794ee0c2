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
#include "gemm_epilogue.h"
#include <algorithm>
#include <type_traits>
#include <vector>

int owc_gemm_profile_begin(double flops, int kind, hipStream_t s);
void owc_gemm_profile_end(int handle, hipStream_t s);
static int gemm_profile_begin_shape(int M, int N, int K, int epi, hipStream_t s);

namespace {

// ---- optional live profiling of THIS kernel (bench.py roofline leg): one HIP-event pair per launch on
// the launch stream, summed by owc_gemm_profile_read().  Off by default (zero cost).
struct ShapeKey {
  int m = 0, n = 0, k = 0, epi = -1;
};
struct ShapeStat {
  ShapeKey key;
  long launches = 0;
  double total_ms = 0.0, min_ms = 0.0, max_ms = 0.0;
};
struct GemmProfile {
  bool on = false;
  std::vector<hipEvent_t> ev;  // start/stop pairs
  std::vector<double> flops;
  std::vector<int> kind;  // OWC_PROF_* launch class (include/owc.h)
  std::vector<ShapeKey> shape;   // (M, N, K, epilogue) of a bf16 GEMM launch; epi = -1: not recorded for this launch class
  size_t used = 0;
  std::vector<ShapeStat> last_shapes;   // the bf16 GEMM launches of the last collect, grouped by shape (owc_profile_shapes)
};
GemmProfile g_prof;
int g_gemm_dbg = 0;       // timing-experiment knob (owc_tuning_set "gemm_dbg" of the -DOWC_TIMING_KNOBS build only), results are garbage unless 0 or 512:
                          // 1 no DMA, 2 DMA re-reads K-tiles 0/1 (all L2 hits), 4 no epilogue, 512 direct (un-staged) epilogue stores, 1024 streaming C stores (results unchanged),
                          // 2048 no per-K-tile barrier, 4096 no DMA wait at the barrier
int g_skinny_max_m = 24;   // M at and below which the weight-streaming skinny kernel runs (0 disables: A-B knob "gemm_skinny_max_m"; up to 64 is legal):
                           // measured on the 7B decode step, ms: M = 33 6.4 skinny / 6.9 64x64 tiles, 40 7.0 / 6.7, 48 7.5 / 6.9, 64 9.8 / 6.6
int g_mid_max_tiles = 256;  // fewer 128x128 tiles than this -> 64x64 tiles (0 disables: A-B knob "gemm_mid_max_tiles")
int g_big_min_tiles = 144;  // fewer 256x256 tiles than this -> use the 128x128 kernel (measured: 160-162 tiles 256x256 +24...40 %, 126 tiles -3...8 %; knob "gemm_big_min_tiles")
int g_small_tiles = 1;    // 64x64-tile kernel: also 32x64 / 64x32 / 32x32 tiles where they shorten the launch (knob "gemm_small_tiles": 0 off, 2 / 3 / 4 force 64x64 / 64x32 / 32x32)
int g_ring_128 = 1;       // 128x64 tiles of the ring kernel where 64x64 tiles need more than one round of 256 CUs (knob "gemm_ring_128", round 4:
                          // the 7B down projection at M = 512 / 640 / 1024 160 -> 138 / 199 -> 157 / 278 -> 225 us, o 33 -> 31 / 41 -> 33 / 59 -> 46)
int g_tall_tiles = 1;     // 64x160 / 128x160 tiles of the ring kernel for launches of <= 128 rows x many columns (knob "gemm_wide_tiles")
int g_k_pairs = 1;        // ring kernel, long K: four (knob value 2: two) K-tiles per stage (knob "gemm_k_pairs"; 0 off)
int g_k_pairs_min_k = 1024;   // (knob "gemm_k_pairs_min_k"; 2B widths, K = 1536: step at batch 32 / 64 2.35 / 2.43 -> 2.27 / 2.35 ms)
int g_norm_fuse_ring = 8;  // rows up to which the ring kernel normalises its own activations (knob "decode_norm_fuse_ring", 0 off, at most 8)
int g_wide_min_blocks = 128;   // fewest 160-column blocks for which the wide ring tiles run (knob "gemm_wide_tiles" = n > 1 sets it)
int g_skinny_deep = 1;    // the skinny kernel's 9-deep ring for long-K launches of at most one wave per CU (knob "gemm_skinny_deep")
int g_norm_fuse_max_m = 2; // rows up to which the decoder's RMSNorm is fused into the qkv / gate-up skinny GEMM (knob "decode_norm_fuse", 0 = off, at most 4).
                           // Measured, 7B decode step in ms, separate / fused: 1 row 3.78 / 3.44, 2 rows 3.78 / 3.65, 4 rows 3.88 / 4.11 - every wave
                           // normalises every row itself, so beyond 2 rows the redundant work outweighs the launch it saves
int g_pingpong = 1;       // 256x256 launches with K % 128 == 0 use the ping-pong kernel (A-B knob "gemm_pingpong", 0 = lock-step kernel)
int g_persist = 0;        // 1: 256x256 ping-pong launches with more tiles than CUs and a bias / GELU / rotary epilogue run the persistent form (knob
                          // "gemm_persist").  OFF: measured +3-5 % on bias-free K = 1280 launches, +-0 with a bias, -2-4 % with a residual, -0.4 % on
                          // the 7B bench end to end (profiles/r05_gemm_persistent_*.txt; the kernel's header says what bounds it)
int g_nt_min_mb = 64;     // outputs larger than this many MiB leave the 256x256 ping-pong kernels as streaming (non-temporal) stores (knob "gemm_nt_min_mb";
                          // round 2-4: 512.  Round 5, per launch at 250-335 MB of C: +0.7 ... +3.2 %, never slower; the 7B bench +0.3 / +0.9 % in two
                          // interleaved pairs: profiles/r05_gemm_persistent_*.txt)
int g_pp128_min_tiles = 128;  // the 256x128 ping-pong kernel runs from this many of its tiles, up to 256 = one round (knob "gemm_pp128"; 0 off)
int g_big_min_m = 129;   // M at and above which the 256x256 kernels may run (knob "gemm_big_min_m"; round 1: 1024)
int g_tail_split = 1;    // a short last round of 256x256 tiles runs as 256x128 tiles in a second launch (knob "gemm_tail_split", 0 = off): see `launch`
int g_walk = 1;          // tile order of the 256x256 ping-pong kernels (knob "gemm_walk"): 0 = rows-of-4 walk of rounds 1-5 for every shape; 1 = column
                         // groups (`tile_origin`, below) where they measured faster: outputs with at least as many tile rows as tile columns
                         // and K >= 4.5 N (the down projections); 2 = column groups for every shape (A-B)

constexpr int BM = 128, BN = 128, BK = 64;
constexpr int TILE_BYTES = BM * BK * 2;  // 16 KiB per operand tile
constexpr int GROUP_M = 8;
constexpr int FLAG_COLWALK = 1 << 16;   // kernel `dbg` / `flags` bit: column-group tile order (set by launch(), see g_walk)

// Block id -> output tile of the 256x256 kernels.  The 32 CUs of an XCD hold 32 CONSECUTIVE tiles of the XCD's contiguous share of the
// walk (blocks are dealt to the XCDs round-robin and start in id order) and run their K loops in lock-step, so per K-step the XCD's L2
// fetches one operand panel per distinct tile ROW and per distinct tile COLUMN among them: 4 x 8 tiles = 12 panels for 64 panel reads
// (the 80 % L2 hit rate of profiles/*_pmc_gemm_traffic.json), and whatever the window holds beyond 12 is fabric traffic - energy,
// on a chip that runs this kernel power-limited.
//   rows-of-4 walk (rounds 1-5): 4 tile rows x all tiles_n columns, m fastest.  A window is 4 x 8 only while 4 * tiles_n is a multiple
//   of 32: with 14 columns (N = 3584: the o / down projections) 56 tiles per row group make every other window straddle two groups
//   (4 x 6 + 4 x 2 of the next: 8 + 8 = 16 panels), 18 columns (qkv) two windows in five, 5 columns (the vision proj / fc2) all.
//   column-group walk (round 6): the tile columns are cut into ceil(tiles_n / 8) groups of (nearly) equal width w <= 8, a group is
//   walked down ALL tiles_m rows with n fastest: a window is 32 / w rows x w columns wherever it starts (4 x 8, 4.6 x 7, 5.3 x 6,
//   6.4 x 5: 12-12.6 panels), and it leaves its group only at the group's end - once per tiles_m * w tiles.
//   MEASURED (profiles/r06_gemm_tile_walk_ab.txt, interleaved A/B per launch at the model's launch groups): fewer panels per
//   window is not the whole story.  The rows-of-4 walk reads A ONCE per launch (a 4-row band stays in the Infinity Cache while its
//   XCD sweeps the columns; the eight XCDs sweep the same W columns together), the column walk re-reads all of A once per column
//   GROUP from HBM: 7B down (K = 18944, 2 groups) +2.6 %, vision fc2 (K = 5120, 1 group) +0.9 %, vit25.down +0.5 %, proj +-0 -
//   (fc2 -0.7 % on a second box) but o (2 groups) -2.5 %, qkv (3) -4.5 %, vision qkv / fc1 (2-3) -1...-6 %, gate/up (19 groups)
//   -5.7 %.  So launch() selects it only for K >= 4.5 N with tiles_m >= tiles_n (the down projections: +2.6 / +2.6 % on two boxes).
// The map is a bijection of [0, tiles_m * tiles_n) either way; results do not depend on it (each tile is computed the same way).
//   Also measured and NOT kept (same file): the rows-of-4 walk inside N chunks of 48-192 MiB of W rows, so that a W larger than the
//   256 MB Infinity Cache (7B gate/up: 271 MB) stays cache-resident under all row bands: -1 ... -6 % at every chunk size.
__device__ __forceinline__ void tile_origin(int bid, int tiles_m, int tiles_n, int flags, int group_m, int bt, int& m0, int& n0) {
  const bool colwalk = (flags & FLAG_COLWALK) != 0;
  const int nblk = tiles_m * tiles_n;
  const int q = nblk >> 3, r = nblk & 7;
  const int xcd = bid & 7, idx = bid >> 3;
  const int lid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  if (colwalk) {
    const int ng = (tiles_n + 7) >> 3;
    const int base = tiles_n / ng, rem = tiles_n - base * ng;   // `rem` groups of base + 1 columns, then ng - rem of base
    const int wide = rem * (base + 1) * tiles_m;
    int w, col0, within;
    if (lid < wide) {
      w = base + 1;
      const int g = lid / (w * tiles_m);
      col0 = g * w;
      within = lid - g * w * tiles_m;
    } else {
      w = base;
      const int l2 = lid - wide;
      const int g = l2 / (w * tiles_m);
      col0 = rem * (base + 1) + g * w;
      within = l2 - g * w * tiles_m;
    }
    const int row = within / w;
    m0 = row * bt;
    n0 = (col0 + within - row * w) * bt;
  } else {
    const int width = group_m * tiles_n;
    const int group = lid / width;
    const int first_m = group * group_m;
    const int gsize = min(tiles_m - first_m, group_m);
    m0 = (first_m + (lid % width) % gsize) * bt;
    n0 = ((lid % width) / gsize) * bt;
  }
}

// KTAIL = false (K % 64 == 0): no per-piece predicate / select, and the LDS-DMA takes the scalar-base + 32-bit lane-offset form (the loop
// then holds no vector instruction but the MFMAs: vector work beside MFMAs is paid in full, tools/probes/probe_mfma_valu_overlap.hip).
template <int EPI, bool KTAIL>
__global__ __launch_bounds__(256) void gemm_bf16_nt_kernel(
    const bf16_t* __restrict__ A, long lda, const bf16_t* __restrict__ W, long ldw,
    const bf16_t* __restrict__ bias, const bf16_t* R, long ldr, void* Cv,
    long ldc, int M, int N, int K, const void* __restrict__ zeros, int tiles_m, int tiles_n, owc_gemm_aux aux) {
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

  const char* abase = (const char*)(A + (long)m0 * lda);
  const char* wbase = (const char*)(W + (long)n0 * ldw);
  unsigned aoff[4], woff[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    aoff[j] = (unsigned)(asrc[j] - abase);
    woff[j] = (unsigned)(wsrc[j] - wbase);
  }
  auto stage = [&](int buf, int kt) {
    char* la = lds + buf * (2 * TILE_BYTES) + w * 4096;
    char* lw = la + TILE_BYTES;
    const int k0 = kt * BK;
    if constexpr (KTAIL) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const bool ok = (k0 + kchunk[j]) < K;
        const void* ga = ok ? (const void*)(asrc[j] + (long)k0 * 2) : zeros;
        const void* gw = ok ? (const void*)(wsrc[j] + (long)k0 * 2) : zeros;
        glds16(ga, la + j * 1024);
        glds16(gw, lw + j * 1024);
      }
    } else {
      const char* ab = abase + (long)k0 * 2;
      const char* wb = wbase + (long)k0 * 2;
      asm volatile("" : "+s"(ab), "+s"(wb));
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        asm volatile("" : "+v"(aoff[j]), "+v"(woff[j]));
        glds16(ab + aoff[j], la + j * 1024);
        glds16(wb + woff[j], lw + j * 1024);
      }
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

  gemm_epilogue<EPI, 4>(acc, m0 + wm * 64, n0 + wn * 64, fr, fq, bias, R, ldr, Cv, ldc, M, N, aux);
}


// ------------------------------------------------------------------------------------------------
// 64x64x64 variant for the in-between shapes (M above the skinny kernel's rows but too few 128x128 tiles to occupy the 256 CUs:
// decode at batch 33-512, the prefill of a single prompt).  4 waves, wave w owns rows [16w, 16w+16) x all 64 columns (one m tile x
// four n tiles, so the shared epilogue applies with MT = 1).  These launches are weight streams with ONE block per CU or less (the
// o / down projections of the 7B decoder are 56 x M/64 tiles), so what matters is bytes in flight per block: a ring of NS64 = 4
// stages of 16 KiB (LDS-DMA), three of them in flight behind a counted vmcnt, ONE raw barrier per K-tile (round 1: two stages and a
// __syncthreads, i.e. vmcnt(0), per K-tile: 16 KiB in flight, 0.65 TB/s on the 7B down projection at batch 256).  Same ascending K
// accumulation chain as every other kernel here (bit-identical results), K tails from the zero page.
// ------------------------------------------------------------------------------------------------
constexpr int B64 = 64;
constexpr int TILE64_BYTES = B64 * BK * 2;  // 8 KiB per operand tile
// stages (A + W tile each): 4 (64 KiB, two blocks per CU) unless the launch has more blocks than 2 per CU and at most 3 per CU, where
// 3 stages (48 KiB) keep it to one round (the 7B gate/up projection at M <= 64 is 592 blocks)

// Tile shape TM x TN in {64, 32} x {64, 32} (round 3).  What bounds these launches is neither this loop nor its bytes in flight but the
// per-CU ingest: with HBM misses in the mix a CU pulls ~56-60 GB/s whatever the ring depth (L2-hit re-reads of A queue behind the
// misses: tools/probes/probe_cu_ingest.hip `gemm` reproduces this kernel's 0.29 us per K-tile with the LDS-DMA alone), so the time is
// the block's bytes over that rate, the W (miss) bytes weighing ~10x the A (L2-hit) bytes - and the right tile for a launch that
// cannot fill 256 CUs with 64x64 tiles is a narrower one (launch(): 32x32 for M <= 64, 64x32 for M <= 256 on the N = 3584
// projections: the 7B down projection 122 -> 72 / 84 / 115 us at M = 64 / 128 / 256).  TM / 16 waves own one 16-row m tile x all TN
// columns; with TM = 32 waves 2-3 only stage.  Same accumulation chain: bit-identical.
template <int EPI, int NS64, bool KTAIL, int TM = 64, int TN = 64, int KPS = 1, bool NORMA = false>
__global__ __launch_bounds__(256) void gemm_bf16_nt_64_kernel(
    const bf16_t* __restrict__ A, long lda, const bf16_t* __restrict__ W, long ldw,
    const bf16_t* __restrict__ bias, const bf16_t* R, long ldr, void* Cv,
    long ldc, int M, int N, int K, const void* __restrict__ zeros, int tiles_m, int tiles_n, owc_gemm_aux aux) {
  extern __shared__ __attribute__((aligned(16))) char lds[];  // [stage][A: TM rows | W: TN rows][128 B]
  // NORMA (decode at <= 8 rows): A is not staged - the M raw residual rows are RMS-normalised once into LDS (row pitch K * 2 + 16)
  // with the arithmetic of norm_kernel, bit for bit, and the ring carries W only; aux.gamma / aux.eps, K % 512 == 0.
  constexpr int TMS = NORMA ? 0 : TM;         // A rows in a stage
  constexpr int SUB = (TMS + TN) * 128;       // bytes of one K-tile's A and W tiles
  constexpr int STAGE = KPS * SUB;            // bytes per stage: KPS consecutive K-tiles behind ONE wait + barrier (long-K launches)
  static_assert(!KTAIL || KPS == 1, "the zero-page tail form stages one K-tile");
  static_assert(!NORMA || (!KTAIL && TM == 32), "the norm-fused form: one 32-row tile (<= 8 of them real)");
  constexpr int NPA = TMS / 8, NPT = (TMS + TN) / 8, NPW = NPT / 4;   // 1-KiB DMA pieces: of A, in all, per wave
  constexpr int NTW = TN / 16;                // n tiles of an active wave (4 or 2)
  constexpr int MTW = TM > 64 ? TM / 64 : 1;  // m tiles of a wave: TM = 128 gives every wave 32 rows (two tiles)
  static_assert(TN % 32 == 0 && (TMS + TN) % 32 == 0, "whole tile pairs; the DMA pieces of a stage divide over four waves");
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
  const int m0 = (first_m + (lid % width) % gsize) * TM, n0 = ((lid % width) / gsize) * TN;

  // staging: piece p = w + 4 i (i < NPW) of a stage; pieces [0, NPA) are 8-row slabs of the A tile, the rest of the W tile.  Whether
  // piece i of a wave is an A or a W slab is the same for all four waves (NPA is a multiple of 4): a compile-time fact.
  const char* src[NPW];
  int kchunk[NPW];
  const char* abase = (const char*)(A + (long)m0 * lda);
  const char* wbase = (const char*)(W + (long)n0 * ldw);
  unsigned off[NPW];
#pragma unroll
  for (int i = 0; i < NPW; ++i) {
    const bool is_a = 4 * i < NPA;
    const int row = 8 * (w + 4 * i - (is_a ? 0 : NPA)) + (l >> 3);   // row inside its tile
    const int c = (l & 7) ^ ((row >> 1) & 7);
    kchunk[i] = c * 8;
    src[i] = is_a ? (const char*)(A + (long)min(m0 + row, M - 1) * lda + c * 8) : (const char*)(W + (long)min(n0 + row, N - 1) * ldw + c * 8);
    off[i] = (unsigned)(src[i] - (is_a ? abase : wbase));
  }
  const int nk = (K + BK - 1) / BK;
  // always NPW LDS-DMA instructions per wave and stage, also past the end of K (zero page: an L2 hit nobody reads), so that the
  // counted waits below hold on every iteration
  const int pitchA = K * 2 + 16;
  char* ring = NORMA ? lds + 8 * pitchA + K * 2 : lds;   // NORMA: [8 rows][pitchA] | gamma [K * 2] | ring
  auto stage = [&](int buf, int kt) {   // kt: index of the STAGE (KPS K-tiles)
    char* dst = ring + buf * STAGE + w * 1024;
    if constexpr (KTAIL) {
      const int k0 = kt * BK;
#pragma unroll
      for (int i = 0; i < NPW; ++i) {
        const bool ok = (k0 + kchunk[i]) < K;
        glds16(ok ? (const void*)(src[i] + (long)k0 * 2) : zeros, dst + i * 4096);
      }
    } else {   // K % 64 == 0; the pieces issued past the end of K (the ring's over-issue) re-read the last K-tile: nobody reads them
#pragma unroll
      for (int u = 0; u < KPS; ++u) {
        const long kk = (long)min((kt * KPS + u) * BK, K - BK) * 2;
        const char* ab = abase + kk;
        const char* wb = wbase + kk;
        asm volatile("" : "+s"(ab), "+s"(wb));
#pragma unroll
        for (int i = 0; i < NPW; ++i) {
          asm volatile("" : "+v"(off[i]));
          glds16((4 * i < NPA ? ab : wb) + off[i], dst + u * SUB + i * 4096);
        }
      }
    }
  };
  const int fr = l & 15, fq = l >> 4;
  const int swz = (fr >> 1) & 7;
  const bool active = w * 16 * MTW < (NORMA ? min(M, TM) : TM);   // wave w: rows [16 MTW w, +16 MTW) (waves 2-3 of a 32-row tile only stage)
  const int offA0 = (w * 16 * MTW + fr) * 128 + (((0 + fq) ^ swz) << 4), offA1 = (w * 16 * MTW + fr) * 128 + (((4 + fq) ^ swz) << 4);
  const int offW0 = TMS * 128 + fr * 128 + (((0 + fq) ^ swz) << 4), offW1 = TMS * 128 + fr * 128 + (((4 + fq) ^ swz) << 4);
  const char* arowN = lds + min(w * 16 + fr, M - 1) * pitchA + fq * 16;   // NORMA: this lane's normalised row (clamped: garbage rows are never stored)

  f32x4 acc[NTW < 4 ? 4 : NTW][MTW];
#pragma unroll
  for (int i = 0; i < (NTW < 4 ? 4 : NTW); ++i)
#pragma unroll
    for (int j = 0; j < MTW; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  // (Round 3 measured a software-pipelined form of this loop - next k-step's fragments read under the current MFMAs, hand-counted
  // lgkmcnt, one tile more look-ahead - and an 8-stage ring: both +-0 on every decode shape.  The simple form stays.)
  const int nst = (nk + KPS - 1) / KPS;   // stages
  if constexpr (NORMA) {
    // raw rows and gamma first (LDS-DMA, 1-KiB pieces): loads return in order, so issued behind the ring they would wait for it
    const bf16_t* gam = (const bf16_t*)aux.gamma;
    const int nj = K >> 9;
    for (int j = w; j < nj; j += 4) glds16(gam + j * 512 + l * 8, lds + 8 * pitchA + j * 1024);
    for (int m = w; m < M; m += 4)
      for (int j = 0; j < nj; ++j) glds16(A + (long)m * lda + j * 512 + l * 8, lds + m * pitchA + j * 1024);
    __builtin_amdgcn_sched_barrier(0);
  }
#pragma unroll
  for (int i = 0; i < NS64 - 1; ++i) stage(i, i);
  if constexpr (NORMA) {
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(KPS * NPW * (NS64 - 1)) : "memory");   // this wave's rows / gamma pieces landed; the ring may fly
    __syncthreads();                                                               // ... and everybody's gamma pieces
    const char* gl = lds + 8 * pitchA;
    const int nch = K >> 3;
    for (int m = w; m < M; m += 4) {   // owc_rms_rstd / owc_rms_apply on the staged row, in place (lane l: chunks l, l + 64, ... ascending)
      char* xrow = lds + m * pitchA;
      float sq = 0.f;
      for (int ch = l; ch < nch; ch += 64) {
        const bf16x8 c = *(const bf16x8*)(xrow + ch * 16);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float v = bf2f(c[e]);
          sq = __builtin_fmaf(v, v, sq);
        }
      }
      sq = wave_sum(sq);
      const float rstd = rsqrtf(__builtin_fmaf(sq, 1.0f / (float)K, aux.eps));
      for (int ch = l; ch < nch; ch += 64) {
        const bf16x8 c = *(const bf16x8*)(xrow + ch * 16), g = *(const bf16x8*)(gl + ch * 16);
        bf16x8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = owc_rms_apply(c[e], g[e], rstd);
        *(bf16x8*)(xrow + ch * 16) = o;
      }
    }
    // (the first loop barrier below - s_waitcnt lgkmcnt(0) + s_barrier - publishes the normalised rows)
  }
  for (int st = 0; st < nst; ++st) {
    // this wave's pieces of stage st have landed (the KPS * NPW * (NS64 - 2) newer ones may fly); the barrier publishes everybody's
    // and tells that every wave is done reading stage st - 1, whose slot the next DMA overwrites
    // (lgkmcnt(0): this wave's own ds_reads of stage st - 1 have retired too - the MFMAs consumed them long ago, so it costs
    // nothing, and the WAR guarantee then holds in the source instead of resting on the compiler's placement of its waits)
    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(KPS * NPW * (NS64 - 2)) : "memory");
    __builtin_amdgcn_sched_barrier(0);
    stage((st + NS64 - 1) % NS64, st + NS64 - 1);
    if (active) {
#pragma unroll
      for (int u = 0; u < KPS; ++u) {
        if (KPS > 1 && st * KPS + u >= nk) break;   // an odd number of K-tiles: the last stage is half used
        const char* la = ring + (st % NS64) * STAGE + u * SUB;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
          bf16x8 fa[MTW], fw[NTW];
#pragma unroll
          for (int t = 0; t < MTW; ++t) {
            if constexpr (NORMA) fa[t] = *(const bf16x8*)(arowN + ((st * KPS + u) * 64 + ks * 32) * 2);
            else fa[t] = *(const bf16x8*)(la + (ks ? offA1 : offA0) + t * 16 * 128);
          }
#pragma unroll
          for (int t = 0; t < NTW; ++t) fw[t] = *(const bf16x8*)(la + (ks ? offW1 : offW0) + t * 16 * 128);
#pragma unroll
          for (int nt = 0; nt < NTW; ++nt)
#pragma unroll
            for (int mt = 0; mt < MTW; ++mt) acc[nt][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fw[nt], fa[mt], acc[nt][mt], 0, 0, 0);
        }
      }
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the over-issued zero-page pieces, before the block's LDS goes away
  if (active) {
    if constexpr (NTW <= 4) {
      gemm_epilogue<EPI, MTW, false, NTW / 2>(*(const f32x4(*)[4][MTW])&acc[0], m0 + w * 16 * MTW, n0, fr, fq, bias, R, ldr, Cv, ldc, M, N, aux);
    } else {   // a wide tile: the shared epilogue on groups of four n tiles (two tile pairs), a single pair at the end
#pragma unroll
      for (int g = 0; g < NTW / 4; ++g)
        gemm_epilogue<EPI, MTW, false, 2>(*(const f32x4(*)[4][MTW])&acc[4 * g], m0 + w * 16 * MTW, n0 + 64 * g, fr, fq, bias, R, ldr, Cv, ldc,
                                          M, N, aux);
      if constexpr (NTW % 4 != 0)
        gemm_epilogue<EPI, MTW, false, 1>(*(const f32x4(*)[4][MTW])&acc[NTW - 2], m0 + w * 16 * MTW, n0 + 16 * (NTW - 2), fr, fq, bias, R, ldr,
                                          Cv, ldc, M, N, aux);
    }
  }
}

// ------------------------------------------------------------------------------------------------
// Large-M variant: 256x256x64 block tile, 512 threads = 8 waves (2 along M x 4 along N), 128x64 per wave
// (8x4 accumulator tiles = 128 VGPRs).  Why: with 128-wide tiles the kernel is bound by the L2 -> LDS
// DMA rate of a CU (measured: the DMA stream alone takes as long as the MFMAs); a 256x256 tile halves
// the operand bytes per FLOP.  Structure per K-tile (64 MFMAs per wave) — 4 phases of 16 MFMAs, each
// (64 rows of the wave tile) x (all 64 columns) x (one 32-deep k-step), walked (rows, k) = (0,0) (1,0)
// (1,1) (0,1) so that every phase consumes fragment blocks that were fetched from LDS during an EARLIER
// phase (software pipeline over 4 register blocks of 4 fragments, 64 VGPRs).  One raw s_barrier per
// K-tile, after phase 3: it publishes stage kt+1 (whose first fragments are prefetched under phase 4) and
// frees stage kt's LDS buffer, into which the DMA of stage kt+2 is issued right away (a full K-tile of
// lead time).  LDS: 2 stages x 64 KiB, one block of 8 waves per CU.
// Requires K % 64 == 0 (no K tail; other shapes use the 128x128 kernel above).
// ------------------------------------------------------------------------------------------------
constexpr int BT = 256;
constexpr int OP_BYTES = BT * BK * 2;        // 32 KiB per operand per stage
constexpr int STAGE_BYTES = 2 * OP_BYTES;    // 64 KiB
constexpr int GROUP_M2 = 4;

template <int EPI>
__global__ __launch_bounds__(512) void gemm_bf16_nt_256_kernel(
    const bf16_t* __restrict__ A, long lda, const bf16_t* __restrict__ W, long ldw,
    const bf16_t* __restrict__ bias, const bf16_t* R, long ldr, void* Cv, long ldc, int M, int N, int K,
    int tiles_m, int tiles_n, int dbg, owc_gemm_aux aux) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int tid = threadIdx.x;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l = tid & 63;

  const int nblk = tiles_m * tiles_n;
  const int nk = K / BK;

  // tile id -> (m0, n0): the blocks of one XCD (id & 7) walk a contiguous, GROUP_M-major range of tiles.
  int m0 = 0, n0 = 0;
  const char *abase = nullptr, *wbase = nullptr;
  unsigned aoff[4], woff[4];
  auto setup = [&](int bid) {
    const int q = nblk >> 3, r = nblk & 7;
    const int xcd = bid & 7, idx = bid >> 3;
    const int lid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    const int width = GROUP_M2 * tiles_n;
    const int group = lid / width;
    const int first_m = group * GROUP_M2;
    const int gsize = min(tiles_m - first_m, GROUP_M2);
    m0 = (first_m + (lid % width) % gsize) * BT;
    n0 = ((lid % width) / gsize) * BT;
    // DMA sources: wave w stages rows [32w, 32w+32) of both operand tiles (4 pieces of 8 rows each);
    // per-lane 32-bit byte offsets from the tile's (uniform) base pointer, rows clamped at the ragged edge.
    abase = (const char*)(A + (long)m0 * lda);
    wbase = (const char*)(W + (long)n0 * ldw);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int row = 32 * w + 8 * j + (l >> 3);
      const int c = (l & 7) ^ ((row >> 1) & 7);
      aoff[j] = (unsigned)((long)min(row, M - 1 - m0) * lda * 2 + c * 16);
      woff[j] = (unsigned)((long)min(row, N - 1 - n0) * ldw * 2 + c * 16);
    }
  };

  auto stage = [&](int buf, int kt) {
    char* la = lds + buf * STAGE_BYTES + w * 4096;
    char* lw = la + OP_BYTES;
    const long kb = (long)kt * (BK * 2);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      glds16(abase + kb + aoff[j], la + j * 1024);
      glds16(wbase + kb + woff[j], lw + j * 1024);
    }
  };
  // one A piece + one W piece (j = 0..3) of a stage: issued BETWEEN the MFMAs of a phase, where their issue cost
  // (~60 cycles per LDS-DMA instruction) hides under the matrix pipe instead of in front of it
  auto stage_piece = [&](int buf, int kt, int j) {
    char* la = lds + buf * STAGE_BYTES + w * 4096;
    const long kb = (long)kt * (BK * 2);
    glds16(abase + kb + aoff[j], la + j * 1024);
    glds16(wbase + kb + woff[j], la + OP_BYTES + j * 1024);
  };

  const int wr = w >> 2, wc = w & 3;
  const int fr = l & 15, fq = l >> 4;
  const int swz = (fr >> 1) & 7;
  // byte offsets (inside a stage) of this lane's fragment rows; k-step ks adds the chunk term
  const int rowA = (wr * 128 + fr) * 128;
  const int rowW = OP_BYTES + (wc * 64 + fr) * 128;
  const int ch0 = ((0 + fq) ^ swz) << 4, ch1 = ((4 + fq) ^ swz) << 4;

  f32x4 acc[4][8];  // [nt][mt]

  // Fragment registers hold 4 blocks of 4 fragments: xa/ya = A blocks (4 m tiles of one half, one k-step),
  // wk0/wk1 = the 4 W fragments (n tiles) of k-step 0 / 1.  64 VGPRs, each block is refilled from LDS
  // while a phase that does not use it runs.
  bf16x8 xa[4], ya[4], wk0[4], wk1[4];

  auto read_a = [&](bf16x8 (&dst)[4], const char* sbase, int half, int ch) {
#pragma unroll
    for (int t = 0; t < 4; ++t) dst[t] = *(const bf16x8*)(sbase + rowA + (half * 4 + t) * 2048 + ch);
  };
  auto read_w = [&](bf16x8 (&dst)[4], const char* sbase, int ch) {
#pragma unroll
    for (int t = 0; t < 4; ++t) dst[t] = *(const bf16x8*)(sbase + rowW + t * 2048 + ch);
  };
  // 16 MFMAs: rows half mh of the wave tile, all 4 n tiles, one 32-deep k-step
  auto phase = [&](const bf16x8 (&af)[4], const bf16x8 (&wf)[4], int mh) {
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int n = 0; n < 4; ++n)
#pragma unroll
      for (int m = 0; m < 4; ++m)
        acc[n][mh * 4 + m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[n], af[m], acc[n][mh * 4 + m], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
  };

  // phase 4 variant: the same 16 MFMAs in 4 groups of 4 with one DMA piece pair after each group: the issue cost of
  // an LDS-DMA instruction (~60 cycles) hides under the matrix pipe instead of idling both waves of the SIMD
  auto phase_dma = [&](const bf16x8 (&af)[4], const bf16x8 (&wf)[4], int mh, int buf, int kt2, bool issue) {
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int n = 0; n < 4; ++n) {
  #pragma unroll
      for (int m = 0; m < 4; ++m)
        acc[n][mh * 4 + m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[n], af[m], acc[n][mh * 4 + m], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      if (issue) stage_piece(buf, kt2, n);
      __builtin_amdgcn_sched_barrier(0);
    }
  };

  setup(blockIdx.x);
  // static priority for the second-dispatched half (the arbitration loser of every SIMD pair), no per-phase flips: +1 % on the
  // decoder shapes against per-phase s_setprio around the MFMA groups (interleaved A/B of two builds on one device)
  if (w >= 4) __builtin_amdgcn_s_setprio(1);
  stage(0, 0);
  if (nk > 1) {
    stage(1, 1);
    asm volatile("s_waitcnt vmcnt(8)\n\ts_barrier" ::: "memory");  // stage 0 landed (stage 1's 8 pieces may fly)
  } else {
    asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
  }
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  read_a(xa, lds, 0, ch0);
  read_w(wk0, lds, ch0);

  // Per K-tile, phases walk (m half, k-step) = (0,0) (1,0) (1,1) (0,1): every phase consumes fragment blocks
  // that were fetched from LDS one phase earlier.
  for (int kt = 0; kt < nk; ++kt) {
    const char* cur = lds + (kt & 1) * STAGE_BYTES;
    const char* nxt = lds + ((kt + 1) & 1) * STAGE_BYTES;
    read_a(ya, cur, 1, ch0);   // for phase 2
    phase(xa, wk0, 0);         // phase 1: rows 0-63,  k-step 0
    read_a(xa, cur, 1, ch1);   // for phase 3
    read_w(wk1, cur, ch1);
    phase(ya, wk0, 1);         // phase 2: rows 64-127, k-step 0
    read_a(ya, cur, 0, ch1);   // for phase 4
    phase(xa, wk1, 1);         // phase 3: rows 64-127, k-step 1
    // every LDS read of stage kt by this wave has completed (lgkmcnt(0), issued >= one phase ago) and its DMA
    // pieces of stage kt+1 have landed (issued a whole K-tile ago); the barrier publishes stage kt+1 and proves
    // stage kt's buffer is no longer read by anyone.
    if (OWC_TK(dbg & 2048)) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");          // timing experiment: no barrier
    else if (OWC_TK(dbg & 4096)) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");  // timing experiment: no DMA wait
    else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    if (kt + 1 < nk) {
      read_a(xa, nxt, 0, ch0);  // next K-tile, phase 1
      read_w(wk0, nxt, ch0);
    }
    // phase 4: rows 0-63, k-step 1 — with the DMA of stage kt+2 (into stage kt's buffer, free since the barrier)
    // interleaved between its MFMA groups
    phase_dma(ya, wk1, 0, kt & 1, OWC_TK(dbg & 2) ? (kt & 1) : kt + 2, kt + 2 < nk && !OWC_TK(dbg & 1));  // dbg 2: re-read K-tiles 0/1 (all L2 hits)
  }

  if (OWC_TK(dbg & 4)) {  // timing experiment: no epilogue
    if (acc[0][0][0] == 123.456f) ((float*)Cv)[0] = 1.f;
    return;
  }
  if constexpr (EPI == OWC_EPI_F32) {
    gemm_epilogue<EPI, 8>(acc, m0 + wr * 128, n0 + wc * 64, fr, fq, bias, R, ldr, Cv, ldc, M, N, aux);
  } else {
    // LDS is quiescent here (every wave passed the last K-tile's barrier with lgkmcnt(0); no DMA is pending): the two
    // stage buffers become the 256 x 256 (SWIGLU: 256 x 128) bf16 image of the output tile
    if (OWC_TK(dbg & 512)) {
      gemm_epilogue<EPI, 8>(acc, m0 + wr * 128, n0 + wc * 64, fr, fq, bias, R, ldr, Cv, ldc, M, N, aux);
      return;
    }
    constexpr int CCOLS = EPI == OWC_EPI_SWIGLU ? BT / 2 : BT;
    gemm_epilogue<EPI, 8>(acc, m0 + wr * 128, n0 + wc * 64, fr, fq, bias, R, ldr, Cv, ldc, M, N, aux, lds, CCOLS * 2,
                          wr * 128, wc * 64);
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    store_ctile<8>(lds, BT, CCOLS, (bf16_t*)Cv, ldc, m0, EPI == OWC_EPI_SWIGLU ? (n0 >> 1) : n0, M,
                   EPI == OWC_EPI_SWIGLU ? (N >> 1) : N, w, l);
  }
}



// ------------------------------------------------------------------------------------------------
// Ping-pong variant of the 256x256x64 kernel (same tile, same LDS image, same epilogue, bit-identical results): the two
// waves of every SIMD alternate ROLES instead of running in lock-step.  Waves 0-3 (wr = 0, one per SIMD) and waves 4-7
// (wr = 1, their SIMD partners) run the same program shifted by one barrier interval: while one group issues the 16 MFMAs
// of a C quadrant (64 rows x 32 columns x K = 64), the other group reads its next fragments from LDS and issues the
// LDS-DMA of a later half-tile; at the next barrier they swap.  A K-tile is 4 phases (quadrants (0,0) (0,1) (1,1) (1,0) of
// the wave's 128 x 64 tile), a phase is [load section] s_barrier [MFMA section] s_barrier.  Counters on the lock-step
// kernel (profiles/): MFMA pipe busy 60 %, waves parked 32 % of their cycles - all eight waves reach their
// LDS reads, their DMA waits and the per-K-tile barrier together, so a SIMD's matrix pipe idles whenever its two waves do.
// Here a SIMD always has one wave in its MFMA section (MI355X_MICROARCH.md "Two waves per SIMD"; cdna_hip_programming.md
// "The 256^2 8-phase template").
//
// Staging: an operand K-tile (256 rows x 128 B) is two half-tiles of 128 rows; wave w stages rows 16w..16w+15 of each
// half (2 LDS-DMA instructions).  Schedule for K-tile u (phases p0..p3), buffers = u & 1:
//   p0  read A(rows half 0 of the wave tile, u)  [8 ds_read_b128]
//   p1  read W(cols half 1, u)                   [4]                issue A half 1 of u+1
//   p2  read A(rows half 1, u)                   [8]                issue W half 0 of u+2      wait: W(u+1) landed
//   p3  read W(cols half 0, u+1) -> other set    [4]                issue W half 1, A half 0 of u+2   wait: A(u+1) landed
// Each wait is a counted vmcnt(6) (three half-tiles stay in flight) at the END of a load section; the data is read one
// phase later (RAW: counted wait -> barrier -> read, for both groups); a region is restaged at the earliest one phase
// after its last read, whose lgkmcnt(0) also sits before the barrier (WAR).  The tail (last two K-tiles) is peeled with
// exact counts.  Requires K % 128 == 0 (an even number of K-tiles: the W fragment sets alternate between two register
// blocks so that p3 can fetch the next tile's set while this tile's is still in use).
// ------------------------------------------------------------------------------------------------
template <int EPI>
__global__ __launch_bounds__(512) void gemm_bf16_nt_256pp_kernel(
    const bf16_t* __restrict__ A, long lda, const bf16_t* __restrict__ W, long ldw,
    const bf16_t* __restrict__ bias, const bf16_t* R, long ldr, void* Cv, long ldc, int M, int N, int K,
    int tiles_m, int tiles_n, int dbg, owc_gemm_aux aux) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int tid = threadIdx.x;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l = tid & 63;
  const int nk = K / BK;

  // tile id -> (m0, n0): gemm_bf16_nt_256_kernel's rows-of-4 walk, or the column-group walk (tile_origin)
  int m0, n0;
  tile_origin(blockIdx.x, tiles_m, tiles_n, dbg, GROUP_M2, BT, m0, n0);
  const char* abase = (const char*)(A + (long)m0 * lda);
  const char* wbase = (const char*)(W + (long)n0 * ldw);
  // DMA sources of this wave: half h, piece j -> rows 128h + 16w + 8j .. +8 (lane: row + (l >> 3), chunk l & 7, swizzled)
  unsigned aoff[2][2], woff[2][2];
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int row = 128 * h + 16 * w + 8 * j + (l >> 3);
      const int c = (l & 7) ^ ((row >> 1) & 7);
      aoff[h][j] = (unsigned)((long)min(row, M - 1 - m0) * lda * 2 + c * 16);
      woff[h][j] = (unsigned)((long)min(row, N - 1 - n0) * ldw * 2 + c * 16);
    }
  // The K-tile's base is pinned in SGPRs (opaque to the compiler, which otherwise builds a 64-bit per-lane address with a
  // v_lshl_add_u64 in front of every LDS-DMA instruction: 24 of them per 128 MFMAs, and vector work beside MFMAs is paid in full -
  // tools/probes/probe_mfma_valu_overlap.hip): the DMA then takes the scalar-base + 32-bit lane-offset form, 0 VALU per piece.
  auto issue_a = [&](int kt, int h) {  // A half-tile h of K-tile kt -> buffer kt & 1
    char* dst = lds + (kt & 1) * STAGE_BYTES + (128 * h + 16 * w) * 128;
    const char* ab = abase + (long)kt * (BK * 2);
    asm volatile("" : "+s"(ab), "+v"(aoff[h][0]), "+v"(aoff[h][1]));   // the zero-extension must not be hoisted either
    glds16(ab + aoff[h][0], dst);
    glds16(ab + aoff[h][1], dst + 1024);
  };
  auto issue_w = [&](int kt, int h) {
    char* dst = lds + (kt & 1) * STAGE_BYTES + OP_BYTES + (128 * h + 16 * w) * 128;
    const char* wb = wbase + (long)kt * (BK * 2);
    asm volatile("" : "+s"(wb), "+v"(woff[h][0]), "+v"(woff[h][1]));
    glds16(wb + woff[h][0], dst);
    glds16(wb + woff[h][1], dst + 1024);
  };

  const int wr = w >> 2, wc = w & 3;
  const int fr = l & 15, fq = l >> 4;
  const int swz = (fr >> 1) & 7;
  const int rowA = (wr * 128 + fr) * 128;
  const int rowW = OP_BYTES + (wc * 64 + fr) * 128;
  const int ch0 = ((0 + fq) ^ swz) << 4, ch1 = ((4 + fq) ^ swz) << 4;

  f32x4 acc[4][8];  // [nt][mt]
  // fragment registers: A half (4 m tiles x 2 k-steps), W column half 1 (2 n tiles x 2 k-steps) and TWO sets of W column
  // half 0 (this tile's / the next tile's): 32 + 16 + 32 VGPRs
  bf16x8 fa[2][4], wy[2][2], wx0[2][2], wx1[2][2];

  auto read_a = [&](const char* sbase, int mh) {
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      fa[0][t] = *(const bf16x8*)(sbase + rowA + (mh * 4 + t) * 2048 + ch0);
      fa[1][t] = *(const bf16x8*)(sbase + rowA + (mh * 4 + t) * 2048 + ch1);
    }
  };
  auto read_w = [&](bf16x8 (&dst)[2][2], const char* sbase, int nh) {
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      dst[0][t] = *(const bf16x8*)(sbase + rowW + (nh * 2 + t) * 2048 + ch0);
      dst[1][t] = *(const bf16x8*)(sbase + rowW + (nh * 2 + t) * 2048 + ch1);
    }
  };
  // end of a load section: counted DMA wait (VM >= 0), all of this wave's LDS reads retired, then the barrier
#define OWC_PP_SYNC_L(VM)                                                                         \
  do {                                                                                            \
    if constexpr ((VM) >= 0) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((VM) < 0 ? 0 : (VM)) : "memory"); \
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");                               \
    __builtin_amdgcn_sched_barrier(0);                                                            \
  } while (0)
  // MFMA section: one C quadrant over the whole K-tile (k-step 0 then 1: every output element stays ONE ascending chain)
  auto quadrant = [&](const bf16x8 (&wf)[2][2], int mh, int nh) {
    // (no s_setprio around the MFMA group: the partner wave is in its load section and needs the issue slots - measured
    // +0.5..1.5 % without, tools/bench_gemm.py A-B)
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int n = 0; n < 2; ++n)
#pragma unroll
        for (int m = 0; m < 4; ++m)
          acc[nh * 2 + n][mh * 4 + m] =
              __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[ks][n], fa[ks][m], acc[nh * 2 + n][mh * 4 + m], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_barrier" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
  };
  // one K-tile; MODE 0 steady (u + 2 < nk), 1 second to last, 2 last.  wc/wn: this tile's / the next tile's W column-half-0 set
  auto ktile = [&](auto mode_c, int u, bf16x8 (&wcur)[2][2], bf16x8 (&wnxt)[2][2]) {
    constexpr int MODE = decltype(mode_c)::value;
    const char* cur = lds + (u & 1) * STAGE_BYTES;
    const char* nxt = lds + ((u + 1) & 1) * STAGE_BYTES;
    // p0
    read_a(cur, 0);
    OWC_PP_SYNC_L(-1);
    quadrant(wcur, 0, 0);
    // p1
    read_w(wy, cur, 1);
    if constexpr (MODE <= 1) issue_a(u + 1, 1);
    OWC_PP_SYNC_L(-1);
    quadrant(wy, 0, 1);
    // p2
    read_a(cur, 1);
    if constexpr (MODE == 0) {
      issue_w(u + 2, 0);
      OWC_PP_SYNC_L(6);   // W halves of u+1 landed; in flight: A0(u+1), A1(u+1), W0(u+2)
    } else if constexpr (MODE == 1) {
      OWC_PP_SYNC_L(4);   // in flight: A0(u+1), A1(u+1)
    } else {
      OWC_PP_SYNC_L(-1);
    }
    quadrant(wy, 1, 1);
    // p3
    if constexpr (MODE <= 1) read_w(wnxt, nxt, 0);
    if constexpr (MODE == 0) {
      issue_w(u + 2, 1);
      issue_a(u + 2, 0);
      OWC_PP_SYNC_L(6);   // A halves of u+1 landed; in flight: W0(u+2), W1(u+2), A0(u+2)
    } else if constexpr (MODE == 1) {
      OWC_PP_SYNC_L(0);
    } else {
      OWC_PP_SYNC_L(-1);
    }
    quadrant(wcur, 1, 0);
  };

  // ---- prologue: K-tile 0 entirely, then what p2 / p3 of "tile -1" would have issued for tile 1
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
  asm volatile("s_waitcnt vmcnt(6)\n\ts_barrier" ::: "memory");   // K-tile 0 landed and published
  __builtin_amdgcn_sched_barrier(0);
  read_w(wx0, lds, 0);
  if (wr) {   // group 1 runs one barrier interval behind group 0
    asm volatile("s_barrier" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
  }
  using I0 = std::integral_constant<int, 0>;
  using I1 = std::integral_constant<int, 1>;
  using I2 = std::integral_constant<int, 2>;
  int u = 0;
  for (; u + 2 < nk; u += 2) {
    ktile(I0{}, u, wx0, wx1);
    ktile(I0{}, u + 1, wx1, wx0);
  }
  ktile(I1{}, u, wx0, wx1);
  ktile(I2{}, u + 1, wx1, wx0);
  if (!wr) {  // group 0 waits for group 1's last MFMA section: every wave has executed the same number of barriers
    asm volatile("s_barrier" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
  }
#undef OWC_PP_SYNC_L

  if (OWC_TK(dbg & 4)) {  // timing experiment: no epilogue
    if (acc[0][0][0] == 123.456f) ((float*)Cv)[0] = 1.f;
    return;
  }
  if constexpr (EPI == OWC_EPI_F32) {
    gemm_epilogue<EPI, 8>(acc, m0 + wr * 128, n0 + wc * 64, fr, fq, bias, R, ldr, Cv, ldc, M, N, aux);
  } else {
    // LDS is quiescent: every wave has passed the final barrier with its reads retired and no DMA pending
    constexpr int CCOLS = EPI == OWC_EPI_SWIGLU ? BT / 2 : BT;
    const int cn0 = EPI == OWC_EPI_SWIGLU ? (n0 >> 1) : n0, cN = EPI == OWC_EPI_SWIGLU ? (N >> 1) : N;
    // (Round 6 measured the epilogue in two row halves - the first half's stores issued under the second half's arithmetic, one more
    //  barrier: per launch +0.1 ... +1 % with bias / GELU / residual / SwiGLU epilogues, +3 % on one shape, and +-0 on the 7B bench
    //  (GEMM total 1386.8 vs 1386.7 TFLOP/s, profiles/r06_gemm_split_epilogue_ab.txt): the store tail is not a background drain that
    //  arithmetic can hide behind.  Removed again.)
    gemm_epilogue<EPI, 8>(acc, m0 + wr * 128, n0 + wc * 64, fr, fq, bias, R, ldr, Cv, ldc, M, N, aux, lds, CCOLS * 2,
                          wr * 128, wc * 64);
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    if (dbg & 1024)   // streaming (non-temporal) C stores: set by launch() for outputs far larger than the caches
      store_ctile<8, true>(lds, BT, CCOLS, (bf16_t*)Cv, ldc, m0, cn0, M, cN, w, l);
    else
      store_ctile<8>(lds, BT, CCOLS, (bf16_t*)Cv, ldc, m0, cn0, M, cN, w, l);
  }
}

// ------------------------------------------------------------------------------------------------
// (Round 6, tried and removed: the same kernel with TWO phases per K-tile - MFMA sections of 32 instead of 16, half as many barrier
// pairs, an A ring of three 32-KiB slots beside the two W stages so that ONE counted wait per K-tile suffices, 64 instead of 80
// fragment registers, 218 VGPRs, bit-identical.  Interleaved A-B at M = 65 536 (profiles/r06_gemm_two_phase_pingpong_ab.txt):
// gate/up 1466 -> 1478 TFLOP/s, down 1553 -> 1539, qkv 1474 = 1474, o 1475 -> 1470, vit.fc1 1338 -> 1335, vit.qkv 1325 -> 1307.
// The barrier count is not what holds this kernel at 0.81 of the clock-limited matrix peak.)
// ------------------------------------------------------------------------------------------------
// PERSISTENT form of the ping-pong kernel (round 5; built, bit-identical, measured - and OFF by default, `g_persist`): one block per
// CU walks its output tiles (tile ids blockIdx.x, + gridDim.x, ...) and the operand ring never drains - the LDS-DMA slots that the last
// two K-tiles of an output tile leave empty in gemm_bf16_nt_256pp_kernel carry K-tiles 0 and 1 of the NEXT output tile, so its first
// MFMA section starts one barrier after this tile's last.
// Why it was built: per output tile the one-tile-per-block kernel pays ~6.4 us that are not MFMAs (fit over K = 1280 ... 18944 on one
// box: t = 6.4 us + 1.35 us per K-tile): 19 % of a K = 1280 launch (the vision tower's qkv / proj / fc1), 8 % at K = 3584.
// What the timing build then showed (profiles/r05_gemm_persistent_*.txt, K = 1280, per tile): no epilogue at all 27.5 us; the
// epilogue's arithmetic without its stores +2.3 us; the STORES ALONE (raw accumulator bits, no arithmetic) +4.4 us; everything 33.0
// (the one-tile-per-block kernel: 29.9 without / 36.2 with its epilogue).  So most of the 6.4 us is neither the block's dispatch nor its
// first loads (this kernel removes those: ~1.5 us) but the 128 KB of C a CU writes per tile, which retire at ~28 GB/s per CU next to
// the operand stream - and vmcnt is IN ORDER on gfx9: a counted wait for a load issued after a store cannot pass before the store has
// retired, so the K loop of the next tile stalls on the previous tile's stores however early they are issued.  Tried against that:
// the tile stored in two batches 1.5 K-tiles apart, every wait right behind a batch needing only loads issued before it (below:
// same 4.7 us); a start-up skew between the blocks of an XCD (+-0: the blocks share operand panels through the L2 and re-align);
// streaming (nt) stores (+1 ... +6 % per launch, for both kernels: `g_nt_min_mb` now 64).  Net, interleaved A/B per launch: bias-free
// K = 1280 +3 ... +5 %, with a bias +-0 ... +3 % (the bias / rotary-table loads of the epilogue wait - in order - for the ring loads
// issued just before them), with a residual -2 ... -4 %, long K +-0 ... +1 %; the 7B bench end to end -0.3 ... -0.5 %.
// What a tile boundary does here:
//   * the C tile does NOT go through LDS (the ring owns it): gemm_epilogue_lines() trades 16-byte pieces between lanes fr and
//     fr ^ 8 so that a store instruction writes 8 rows x 128 contiguous bytes - full lines straight from registers, no barrier;
//   * rows 0-63 of a wave tile are final after the last K-tile's second phase and are stored in its third, rows 64-127 at the boundary;
//   * the accumulators are zeroed by 128 moves (the compiler folds them into zero-C MFMAs where it peels).
// Schedule, waits and fragment sets inside a K-tile are gemm_bf16_nt_256pp_kernel's (K % 128 == 0 keeps the buffer / fragment-set
// parity across tile boundaries); each output element is the same ascending K chain: bit-identical results
// (test_gemm_persistent_pingpong_race_screen, test_vision_tower_bits_with_and_without_the_persistent_gemm).  The pointers the DMA
// reads from move to the next output tile in the middle of K-tile nk - 2 (after its A-half issue for K-tile nk - 1, the last issue
// that belongs to this tile).  gridDim.x is a multiple of 8, so a block's tiles stay on its XCD's part of the XCD-aware walk.
// Requires K % 128 == 0, K >= 384; bf16 outputs of the tile's own width (no F32 / SwiGLU epilogue).
// ------------------------------------------------------------------------------------------------
template <int EPI, bool NT>   // NT: streaming (non-temporal) C stores, for outputs far larger than the caches
__global__ __launch_bounds__(512) void gemm_bf16_nt_256pp_persist_kernel(
    const bf16_t* __restrict__ A, long lda, const bf16_t* __restrict__ W, long ldw,
    const bf16_t* __restrict__ bias, const bf16_t* R, long ldr, void* Cv, long ldc, int M, int N, int K,
    int tiles_m, int tiles_n, int flags, owc_gemm_aux aux) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int tid = threadIdx.x;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l = tid & 63;
  const int nblk = tiles_m * tiles_n;
  const int nk = K / BK;

  auto tile_origin = [&](int bid, int& m0, int& n0) {   // as gemm_bf16_nt_256pp_kernel
    ::tile_origin(bid, tiles_m, tiles_n, flags, GROUP_M2, BT, m0, n0);
  };
  // ---- what the DMA reads: the output tile whose K-tiles are being issued (one or two K-tiles ahead of the MFMAs)
  const char *abase, *wbase;
  unsigned aoff[2][2], woff[2][2];
  // (32-bit arithmetic from an opaque copy of the lane id, one offset after the other: this runs in the middle of a K-tile with every
  //  fragment register live - nothing of it may be hoisted into long-lived registers or interleaved into a wide front of temporaries)
  const unsigned lda2 = (unsigned)(lda * 2), ldw2 = (unsigned)(ldw * 2);
  auto point_loads_at = [&](int m0, int n0) {
    abase = (const char*)(A + (long)m0 * lda);
    wbase = (const char*)(W + (long)n0 * ldw);
    const int mlim = M - 1 - m0, nlim = N - 1 - n0;
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        int lx = l;
        asm volatile("" : "+v"(lx));
        const int row = 128 * h + 16 * w + 8 * j + (lx >> 3);
        const unsigned c16 = (unsigned)(((lx & 7) ^ ((row >> 1) & 7)) << 4);
        aoff[h][j] = (unsigned)min(row, mlim) * lda2 + c16;
        woff[h][j] = (unsigned)min(row, nlim) * ldw2 + c16;
        asm volatile("" : "+v"(aoff[h][j]), "+v"(woff[h][j]));
      }
  };
  auto issue_a = [&](int kt, int h) {  // A half-tile h of K-tile kt (of the tile the DMA points at) -> buffer kt & 1
    char* dst = lds + (kt & 1) * STAGE_BYTES + (128 * h + 16 * w) * 128;
    const char* ab = abase + (long)kt * (BK * 2);
    asm volatile("" : "+s"(ab), "+v"(aoff[h][0]), "+v"(aoff[h][1]));
    glds16(ab + aoff[h][0], dst);
    glds16(ab + aoff[h][1], dst + 1024);
  };
  auto issue_w = [&](int kt, int h) {
    char* dst = lds + (kt & 1) * STAGE_BYTES + OP_BYTES + (128 * h + 16 * w) * 128;
    const char* wb = wbase + (long)kt * (BK * 2);
    asm volatile("" : "+s"(wb), "+v"(woff[h][0]), "+v"(woff[h][1]));
    glds16(wb + woff[h][0], dst);
    glds16(wb + woff[h][1], dst + 1024);
  };

  const int wr = w >> 2, wc = w & 3;
  const int fr = l & 15, fq = l >> 4;
  const int swz = (fr >> 1) & 7;
  const int rowA = (wr * 128 + fr) * 128;
  const int rowW = OP_BYTES + (wc * 64 + fr) * 128;
  const int ch0 = ((0 + fq) ^ swz) << 4, ch1 = ((4 + fq) ^ swz) << 4;

  f32x4 acc[4][8];  // [nt][mt]
  bf16x8 fa[2][4], wy[2][2], wx0[2][2], wx1[2][2];

  auto read_a = [&](const char* sbase, int mh) {
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      fa[0][t] = *(const bf16x8*)(sbase + rowA + (mh * 4 + t) * 2048 + ch0);
      fa[1][t] = *(const bf16x8*)(sbase + rowA + (mh * 4 + t) * 2048 + ch1);
    }
  };
  auto read_w = [&](bf16x8 (&dst)[2][2], const char* sbase, int nh) {
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      dst[0][t] = *(const bf16x8*)(sbase + rowW + (nh * 2 + t) * 2048 + ch0);
      dst[1][t] = *(const bf16x8*)(sbase + rowW + (nh * 2 + t) * 2048 + ch1);
    }
  };
#define OWC_PS_SYNC_L(VM)                                                                         \
  do {                                                                                            \
    if constexpr ((VM) >= 0) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((VM) < 0 ? 0 : (VM)) : "memory"); \
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");                               \
    __builtin_amdgcn_sched_barrier(0);                                                            \
  } while (0)
  // MFMA section: one C quadrant over the whole K-tile (k-step 0 then 1: every output element stays ONE ascending chain)
  auto quadrant = [&](const bf16x8 (&wf)[2][2], int mh, int nh) {
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int n = 0; n < 2; ++n)
#pragma unroll
        for (int m = 0; m < 4; ++m)
          acc[nh * 2 + n][mh * 4 + m] =
              __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[ks][n], fa[ks][m], acc[nh * 2 + n][mh * 4 + m], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_barrier" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
  };
  // vmcnt is IN ORDER on gfx9 (loads and stores share the counter and retire in issue order), so the C stores of an output tile sit
  // in the counter between the loads issued before and after them - and they are slow to retire: a CU's stores of one tile (128 KB)
  // take ~4.4 us next to the operand stream (timing build, stores of raw accumulator bits only: profiles/).  A counted wait that
  // only needs loads issued BEFORE a batch of stores may leave that batch in flight: its steady-state count + the batch.  The
  // schedule is arranged so that every wait right behind a batch is of that kind, and the tile is stored in two batches 1.5 K-tiles
  // apart, each with ~2 us until a wait needs a load that is younger than it:
  //   K-tile nk-1  p1  issue A half 1 of K-tile 0'                              (0', 1': K-tiles of the NEXT output tile)
  //                p2  rows 0-63 of the wave tile are final (quadrants (0,0), (0,1)): batch S0 = 8 stores; issue W half 0 of 1'
  //                    wait W(0') [older than S0]: in flight A(0') 4 + S0 8 + 2 = 14
  //                p3  issue W half 1, A halves 0 AND 1 of 1' (the A half 1 would be K-tile 0's p1 issue: it is brought forward so
  //                    that K-tile 0's waits need nothing younger than the stores)    wait A(0'): S0 8 + 8 = 16
  //   boundary         rows 64-127 final: batch S1 = 8 stores
  //   K-tile 0     p2  issue W half 0 of 2     wait W(1') [younger than S0, older than S1]: A(1') 4 + S1 8 + 2 = 14
  //                p3  issue W half 1, A half 0 of 2     wait A(1'): S1 8 + 6 = 14
  //   K-tile 1     p2  wait W(2): the steady-state 6 - the first wait that needs S1 retired
  // The store counts hold for INTERIOR tiles (every wave issues all 16 stores); for a tile on the ragged edge, and for the block's
  // first tile, the steady-state counts are used - merely stricter.
  auto wait_sync = [&](int n) {   // n in {6, 8, 14, 16}: block-uniform
    if (n == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    else if (n == 14) asm volatile("s_waitcnt vmcnt(14)" ::: "memory");
    else if (n == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
  };
  int m0, n0;        // the output tile the MFMAs are working on
  bool interior = false, lax = false;   // this tile / the previous one issued every store
  // One K-tile of the stream: K-tiles kt1 = "u + 1" and kt2 = "u + 2" of whatever output tile the DMA points at are issued.
  // `first`: K-tile 0 of an output tile (no p1 issue; waits that leave the previous tile's second store batch in flight).
  // `last`: K-tile nk - 1 (stores the first half of the tile; also issues A half 1 of kt2 = K-tile 1').  `turn`: called after p1's
  // issue (the point where K-tile nk - 2 hands the DMA over to the next output tile).  All of them are block-uniform: scalar branches
  // around a wait.  (The copies of the K-tile pair follow each other in a straight line; with copies behind an if / else - a peeled
  // "last tile of the block" tail - the register allocator moved the accumulators between them: MFMAs with vDst != srcC, 300 spills.)
  auto ktile = [&](auto first_c, auto last_c, int u, int kt1, int kt2, bf16x8 (&wcur)[2][2], bf16x8 (&wnxt)[2][2], auto&& turn) {
    constexpr bool first = decltype(first_c)::value, last = decltype(last_c)::value;
    const char* cur = lds + (u & 1) * STAGE_BYTES;
    const char* nxt = lds + ((u + 1) & 1) * STAGE_BYTES;
    // p0
    read_a(cur, 0);
    OWC_PS_SYNC_L(-1);
    quadrant(wcur, 0, 0);
    // p1
    read_w(wy, cur, 1);
    if constexpr (!first) issue_a(kt1, 1);
    turn();
    OWC_PS_SYNC_L(-1);
    quadrant(wy, 0, 1);
    // p2
    if (last && !OWC_TK(flags & 4)) {   // rows 0-63 of the wave tile (the A fragments of those rows are dead, the next W set not yet read)
      int lx = l;                       // (opaque lane id: nothing of the epilogue's addressing is hoisted out of the K loop)
      asm volatile("" : "+v"(lx));
      gemm_epilogue_lines<EPI, 8, NT, 0, 4>(acc, m0 + wr * 128, n0 + wc * 64, lx & 15, lx >> 4, bias, R, ldr, Cv, ldc, M, N, aux, flags);
    }
    read_a(cur, 1);
    issue_w(kt2, 0);
    wait_sync(((last && interior) || (first && lax)) ? 14 : 6);
    quadrant(wy, 1, 1);
    // p3
    read_w(wnxt, nxt, 0);
    issue_w(kt2, 1);
    issue_a(kt2, 0);
    if constexpr (last) issue_a(kt2, 1);              // (its LDS rows were last read one phase ago, in this K-tile's p2)
    wait_sync(last ? (interior ? 16 : 8) : ((first && lax) ? 14 : 6));
    quadrant(wcur, 1, 0);
  };
  auto nothing = [] {};
  using Yes = std::true_type;
  using No = std::false_type;

  int tile = blockIdx.x;
  tile_origin(tile, m0, n0);
  point_loads_at(m0, n0);
  // ---- prologue of the block's first output tile: K-tile 0 entirely, then what the previous tile's last K-tile would have issued for K-tile 1
  issue_a(0, 0);
  issue_a(0, 1);
  issue_w(0, 0);
  issue_w(0, 1);
  issue_w(1, 0);
  issue_w(1, 1);
  issue_a(1, 0);
  issue_a(1, 1);
  asm volatile("s_waitcnt vmcnt(8)\n\ts_barrier" ::: "memory");   // K-tile 0 landed and published
  __builtin_amdgcn_sched_barrier(0);
  read_w(wx0, lds, 0);
  if (wr) {   // group 1 runs one barrier interval behind group 0
    asm volatile("s_barrier" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
  }
  // The block's last output tile runs the same code as every other: its spare DMA slots re-read K-tiles 0 / 1 of the tile itself
  // (never consumed; drained before the block ends).
  for (;;) {
    const int next = tile + (int)gridDim.x;
    const bool more = next < nblk;          // block-uniform
    int m1 = m0, n1 = n0;
    if (more) tile_origin(next, m1, n1);
    interior = m0 + BT <= M && n0 + BT <= N && !OWC_TK(flags & (4 | 8));
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    // three copies of the K-tile pair in a straight line - K-tiles 0 / 1, the steady-state loop (no branch in it), K-tiles nk - 2 /
    // nk - 1 whose "u + 2" slots carry K-tiles 0 / 1 of the next output tile (same buffer parity: nk is even)
    ktile(Yes{}, No{}, 0, 1, 2, wx0, wx1, nothing);
    ktile(No{}, No{}, 1, 2, 3, wx1, wx0, nothing);
    int u = 2;
#pragma unroll 1
    for (; u + 2 < nk; u += 2) {
      ktile(No{}, No{}, u, u + 1, u + 2, wx0, wx1, nothing);
      ktile(No{}, No{}, u + 1, u + 2, u + 3, wx1, wx0, nothing);
    }
    ktile(No{}, No{}, u, u + 1, 0, wx0, wx1, [&] { point_loads_at(m1, n1); });
    ktile(No{}, Yes{}, u + 1, 0, 1, wx1, wx0, nothing);
    if (!OWC_TK(flags & 4)) {   // rows 64-127 of the wave tile
      int lx = l;
      asm volatile("" : "+v"(lx));
      gemm_epilogue_lines<EPI, 8, NT, 4, 8>(acc, m0 + wr * 128, n0 + wc * 64, lx & 15, lx >> 4, bias, R, ldr, Cv, ldc, M, N, aux, flags);
    } else if (acc[0][0][0] == 123.456f) {
      ((float*)Cv)[0] = 1.f;
    }
    lax = interior;
    if (!more) break;
    tile = next;
    m0 = m1;
    n0 = n1;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the spare slots' DMA has landed before the block gives its LDS back
  if (!wr) {  // every wave has executed the same number of barriers
    asm volatile("s_barrier" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
  }
#undef OWC_PS_SYNC_L
}

// ------------------------------------------------------------------------------------------------
// Ping-pong kernel on a 256 x 128 tile (round 5): the role-alternating schedule of gemm_bf16_nt_256pp_kernel for launches whose
// 256 x 256 tiles would leave more than half of the CUs idle - the o / down projections (N = 3584) of a decode step at
// 1024-2048 rows (a full pass's decode batch: 8 x 14 = 112 tiles of 256^2, 224 of 256 x 128), which ran on the 128 x 128
// lock-step kernel at 0.36 of the MFMA peak.  8 waves = 4 (rows) x 2 (columns) wave tiles of 64 x 64 (64 accumulator VGPRs);
// waves 0-3 (rows 0-127) and waves 4-7 (rows 128-255) are the two role groups, one barrier interval apart.  A K-tile is TWO
// phases - the wave tile's column half 0, then half 1, each 4 m tiles x 2 n tiles x 2 k-steps = 16 MFMAs - and a phase is
// [load section] s_barrier [MFMA section] s_barrier as there.  A K-tile lasts half as long as a 256 x 256 one, so the DMA
// runs TWO K-tiles ahead through a ring of three 48-KiB stages (A 256 rows + W 128 rows of 128 B; 144 KiB):
//   L0(u)  read A(u) [8 ds_read_b128]                                 issue W(u+2), A rows 0-127 of u+2    wait: W(u+1) landed
//   L1(u)  read W(cols half 1, u) [4], W(cols half 0, u+1) [4]        issue A rows 128-255 of u+2          wait: A(u+1) landed
// with counted vmcnt(8) / vmcnt(6) (four / three half-tiles stay in flight), the wait at the END of a load section and the
// read one phase later (RAW), a stage restaged a whole K-tile after its last read (WAR; every load section ends with
// lgkmcnt(0) before its barrier).  The column-half-0 fragments are dead after the first MFMA section, so the next tile's
// go into the same registers (the 256 x 256 kernel needs two sets).  The loop walks three K-tiles per iteration, so every
// ring slot is a compile-time constant (no vector instruction in the loop but the MFMAs); the 2-4 last K-tiles are peeled
// with exact counts.  Requires K % 64 == 0, K >= 128.  Every output element is ONE ascending K chain (k-step 0 then 1 of
// every K-tile): bit-identical to every other bf16 GEMM kernel here.
// ------------------------------------------------------------------------------------------------
constexpr int P128_A_BYTES = 256 * BK * 2;                 // 32 KiB
constexpr int P128_STAGE = P128_A_BYTES + 128 * BK * 2;    // 48 KiB
constexpr int P128_LDS = 3 * P128_STAGE;                   // 144 KiB

template <int EPI>
__global__ __launch_bounds__(512) void gemm_bf16_nt_256x128pp_kernel(
    const bf16_t* __restrict__ A, long lda, const bf16_t* __restrict__ W, long ldw,
    const bf16_t* __restrict__ bias, const bf16_t* R, long ldr, void* Cv, long ldc, int M, int N, int K,
    int tiles_m, int tiles_n, int dbg, owc_gemm_aux aux) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int tid = threadIdx.x;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l = tid & 63;
  const int nblk = tiles_m * tiles_n;
  const int nk = K / BK;

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
    m0 = (first_m + (lid % width) % gsize) * 256;
    n0 = ((lid % width) / gsize) * 128;
  }
  const char* abase = (const char*)(A + (long)m0 * lda);
  const char* wbase = (const char*)(W + (long)n0 * ldw);
  // DMA sources of this wave: A half h, piece j -> rows 128h + 16w + 8j .. +8; W piece j -> rows 16w + 8j .. +8
  unsigned aoff[2][2], woff[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int row = 128 * h + 16 * w + 8 * j + (l >> 3);
      aoff[h][j] = (unsigned)((long)min(row, M - 1 - m0) * lda * 2 + ((l & 7) ^ ((row >> 1) & 7)) * 16);
    }
    const int row = 16 * w + 8 * j + (l >> 3);
    woff[j] = (unsigned)((long)min(row, N - 1 - n0) * ldw * 2 + ((l & 7) ^ ((row >> 1) & 7)) * 16);
  }
  // (kt: the K-tile, sl: its ring slot = kt % 3, a compile-time constant at every call)
  auto issue_a = [&](int kt, int sl, int h) {   // scalar base + 32-bit lane offset form (see the 256 x 256 kernel)
    char* dst = lds + sl * P128_STAGE + (128 * h + 16 * w) * 128;
    const char* ab = abase + (long)kt * (BK * 2);
    asm volatile("" : "+s"(ab), "+v"(aoff[h][0]), "+v"(aoff[h][1]));
    glds16(ab + aoff[h][0], dst);
    glds16(ab + aoff[h][1], dst + 1024);
  };
  auto issue_w = [&](int kt, int sl) {
    char* dst = lds + sl * P128_STAGE + P128_A_BYTES + 16 * w * 128;
    const char* wb = wbase + (long)kt * (BK * 2);
    asm volatile("" : "+s"(wb), "+v"(woff[0]), "+v"(woff[1]));
    glds16(wb + woff[0], dst);
    glds16(wb + woff[1], dst + 1024);
  };

  const int grp = w >> 2, wr = 2 * grp + ((w >> 1) & 1), wc = w & 1;   // wave tile: rows 64 wr .., columns 64 wc ..
  const int fr = l & 15, fq = l >> 4;
  const int swz = (fr >> 1) & 7;
  const int rowA = (wr * 64 + fr) * 128;
  const int rowW = P128_A_BYTES + (wc * 64 + fr) * 128;
  const int ch0 = ((0 + fq) ^ swz) << 4, ch1 = ((4 + fq) ^ swz) << 4;

  f32x4 acc[4][4];  // [nt][mt]
  bf16x8 fa[2][4], wx[2][2], wy[2][2];   // A (4 m tiles x 2 k-steps), W column half 0, W column half 1

  auto read_a = [&](const char* sbase) {
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      fa[0][t] = *(const bf16x8*)(sbase + rowA + t * 2048 + ch0);
      fa[1][t] = *(const bf16x8*)(sbase + rowA + t * 2048 + ch1);
    }
  };
  auto read_w = [&](bf16x8 (&dst)[2][2], const char* sbase, int nh) {
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      dst[0][t] = *(const bf16x8*)(sbase + rowW + (nh * 2 + t) * 2048 + ch0);
      dst[1][t] = *(const bf16x8*)(sbase + rowW + (nh * 2 + t) * 2048 + ch1);
    }
  };
#define OWC_PP_SYNC_L(VM)                                                                         \
  do {                                                                                            \
    if constexpr ((VM) >= 0) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((VM) < 0 ? 0 : (VM)) : "memory"); \
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");                               \
    __builtin_amdgcn_sched_barrier(0);                                                            \
  } while (0)
  auto half = [&](const bf16x8 (&wf)[2][2], int nh) {   // MFMA section: the wave tile's column half nh over the whole K-tile
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int n = 0; n < 2; ++n)
#pragma unroll
        for (int m = 0; m < 4; ++m)
          acc[nh * 2 + n][m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[ks][n], fa[ks][m], acc[nh * 2 + n][m], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_barrier" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
  };
  // one K-tile u in ring slot SL; MODE 0 steady (u + 2 < nk), 1 second to last, 2 last
  auto ktile = [&](auto mode_c, auto slot_c, int u) {
    constexpr int MODE = decltype(mode_c)::value, SL = decltype(slot_c)::value;
    constexpr int SL1 = (SL + 1) % 3, SL2 = (SL + 2) % 3;
    const char* cur = lds + SL * P128_STAGE;
    // L0
    read_a(cur);
    if constexpr (MODE == 0) {
      issue_w(u + 2, SL2);
      issue_a(u + 2, SL2, 0);
      OWC_PP_SYNC_L(8);   // W(u+1) landed; in flight: A0(u+1), A1(u+1), W(u+2), A0(u+2)
    } else if constexpr (MODE == 1) {
      OWC_PP_SYNC_L(4);   // in flight: A0(u+1), A1(u+1)
    } else {
      OWC_PP_SYNC_L(-1);
    }
    half(wx, 0);
    // L1
    read_w(wy, cur, 1);
    if constexpr (MODE <= 1) read_w(wx, lds + SL1 * P128_STAGE, 0);
    if constexpr (MODE == 0) {
      issue_a(u + 2, SL2, 1);
      OWC_PP_SYNC_L(6);   // A(u+1) landed; in flight: W(u+2), A0(u+2), A1(u+2)
    } else if constexpr (MODE == 1) {
      OWC_PP_SYNC_L(0);
    } else {
      OWC_PP_SYNC_L(-1);
    }
    half(wy, 1);
  };

  // ---- prologue: K-tiles 0 and 1 entirely (issue order = the steady state's)
  issue_a(0, 0, 0);
  issue_a(0, 0, 1);
  issue_w(0, 0);
  issue_w(1, 1);
  issue_a(1, 1, 0);
  issue_a(1, 1, 1);
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  asm volatile("s_waitcnt vmcnt(6)\n\ts_barrier" ::: "memory");   // K-tile 0 landed and published
  __builtin_amdgcn_sched_barrier(0);
  read_w(wx, lds, 0);
  if (grp) {   // group 1 runs one barrier interval behind group 0
    asm volatile("s_barrier" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
  }
  using M0_ = std::integral_constant<int, 0>;
  using M1_ = std::integral_constant<int, 1>;
  using M2_ = std::integral_constant<int, 2>;
  using S0_ = std::integral_constant<int, 0>;
  using S1_ = std::integral_constant<int, 1>;
  using S2_ = std::integral_constant<int, 2>;
  int u = 0;
  for (; u + 4 < nk; u += 3) {   // u % 3 == 0 throughout: the slots are constants
    ktile(M0_{}, S0_{}, u);
    ktile(M0_{}, S1_{}, u + 1);
    ktile(M0_{}, S2_{}, u + 2);
  }
  // The 2-4 last K-tiles run through ONE more copy of the K-tile whose ring slot and mode are run-time (block-uniform) values:
  // scalar branches around an issue or a wait, the slot offset a scalar add.  (Round 5 first peeled them as three if / else chains of
  // compile-time copies: the register allocator then moved the accumulators between the copies - MFMAs with vDst != srcC, 4-36
  // spilled registers per instantiation, each reload a full vmcnt drain in front of the epilogue.)
  auto ktile_rt = [&](int kt, int sl, int rem) {   // rem: K-tiles after this one (>= 2: the steady state)
    const int sl1 = sl == 2 ? 0 : sl + 1, sl2 = sl == 0 ? 2 : sl - 1;
    const char* cur = lds + sl * P128_STAGE;
    // L0
    read_a(cur);
    if (rem >= 2) {
      issue_w(kt + 2, sl2);
      issue_a(kt + 2, sl2, 0);
      asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    } else if (rem == 1) {
      asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    }
    OWC_PP_SYNC_L(-1);
    half(wx, 0);
    // L1
    read_w(wy, cur, 1);
    if (rem >= 1) read_w(wx, lds + sl1 * P128_STAGE, 0);
    if (rem >= 2) {
      issue_a(kt + 2, sl2, 1);
      asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    } else if (rem == 1) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    OWC_PP_SYNC_L(-1);
    half(wy, 1);
  };
  const int left = nk - u;   // 2, 3 or 4 (u % 3 == 0: the tail starts in slot 0)
#pragma unroll 1
  for (int j = 0; j < left; ++j) ktile_rt(u + j, j >= 3 ? 0 : j, left - 1 - j);
  if (!grp) {  // group 0 waits for group 1's last MFMA section: every wave has executed the same number of barriers
    asm volatile("s_barrier" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
  }
#undef OWC_PP_SYNC_L
  // (the accumulators are pinned here: MFMAs have no side effect an s_barrier orders, and LLVM's sinking pass otherwise moves the
  // last K-tile's MFMAs next to their only use, into the epilogue's conditional store blocks - round 3, fp8 kernel)
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) asm volatile("" : "+v"(acc[i][j]));

  if (OWC_TK(dbg & 4)) {  // timing experiment: no epilogue
    if (acc[0][0][0] == 123.456f) ((float*)Cv)[0] = 1.f;
    return;
  }
  if constexpr (EPI == OWC_EPI_F32) {
    gemm_epilogue<EPI, 4>(acc, m0 + wr * 64, n0 + wc * 64, fr, fq, bias, R, ldr, Cv, ldc, M, N, aux);
  } else {
    // LDS is quiescent: every wave has passed the final barrier with its reads retired and no DMA pending
    constexpr int CCOLS = EPI == OWC_EPI_SWIGLU ? 64 : 128;
    gemm_epilogue<EPI, 4>(acc, m0 + wr * 64, n0 + wc * 64, fr, fq, bias, R, ldr, Cv, ldc, M, N, aux, lds, CCOLS * 2, wr * 64, wc * 64);
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    store_ctile<8>(lds, 256, CCOLS, (bf16_t*)Cv, ldc, m0, EPI == OWC_EPI_SWIGLU ? (n0 >> 1) : n0, M,
                   EPI == OWC_EPI_SWIGLU ? (N >> 1) : N, w, l);
  }
}

// ------------------------------------------------------------------------------------------------
// Skinny-M variant (M <= 64: greedy decode at the reference's own batch sizes, 1 ... a few dozen sequences).  There the
// GEMM is a weight STREAM: every byte of W is read once per step, the arithmetic is nothing, and the tiled kernels above
// leave most CUs idle (the o / down projections of the 7B decoder are 28 tiles wide).  Here ONE WAVE owns 16 rows of W
// (32 for SwiGLU: the gate and the up rows of the same 16 features) over the whole K: W fragments go global -> registers
// (16 B per lane, no LDS, no barrier) through a ring of DEPTH 128-wide super-steps kept in flight, the activations (a few
// KB, L1 / L2 resident) are fetched per super-step.  Every output element is ONE MFMA accumulation chain over K in
// ascending order with the same operand roles as the tiled kernels, so the result is bit-identical to theirs: which kernel
// ran never shows in the output (batch invariance across decode batch sizes and against the prefill).
// Requires K % 128 == 0, N % 16 == 0 (32 for SwiGLU).
// ------------------------------------------------------------------------------------------------
template <int EPI, int MT, int DEPTH>
__global__ __launch_bounds__(64) void gemm_bf16_skinny_kernel(
    const bf16_t* __restrict__ A, long lda, const bf16_t* __restrict__ W, long ldw, const bf16_t* __restrict__ bias,
    const bf16_t* R, long ldr, bf16_t* C, long ldc, int M, int N, int K) {
  constexpr int NT = EPI == OWC_EPI_SWIGLU ? 2 : 1;
  const int l = threadIdx.x;
  const int fr = l & 15, fq = l >> 4;
  const int n0 = blockIdx.x * (16 * NT);
  const int nss = K >> 7;  // 128-wide super-steps

  const bf16_t* wrow[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) wrow[nt] = W + (long)(n0 + nt * 16 + fr) * ldw + fq * 8;
  const bf16_t* arow[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) arow[mt] = A + (long)min(mt * 16 + fr, M - 1) * lda + fq * 8;

  f32x4 acc[MT][NT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // rings only ever indexed with compile-time constants (the ring loop below is fully unrolled).  A super-step is 4 * MT * NT
  // MFMAs, far shorter than an L2 hit, so the activations ride the same ring as the weights.
  bf16x8 ring[DEPTH][NT][4];
  bf16x8 xring[DEPTH][MT][4];
  auto load_w = [&](int i, int ss) {
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) ring[i][nt][ks] = *(const bf16x8*)(wrow[nt] + ss * 128 + ks * 32);
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) xring[i][mt][ks] = *(const bf16x8*)(arow[mt] + ss * 128 + ks * 32);
  };
  auto compute = [&](int i, int ss) {
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
          acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ring[i][nt][ks], xring[i][mt][ks], acc[mt][nt], 0, 0, 0);
  };
  // Branch-free steady state (with conditionals around the loads the compiler falls back to vmcnt(0) after every refill and
  // the ring degenerates to depth 1): loads past the end re-read the last super-step (an L2 hit nobody consumes), whole
  // rounds run unconditionally, the last partial round only computes.
#pragma unroll
  for (int i = 0; i < DEPTH; ++i) load_w(i, min(i, nss - 1));
  int ss = 0;
  for (; ss + DEPTH <= nss; ss += DEPTH) {
#pragma unroll
    for (int i = 0; i < DEPTH; ++i) {
      compute(i, ss + i);
      load_w(i, min(ss + i + DEPTH, nss - 1));
    }
  }
  const int rem = nss - ss;
#pragma unroll
  for (int i = 0; i < DEPTH; ++i)
    if (i < rem) compute(i, ss + i);
  // lane holds row m = 16 mt + fr, columns n0 + 16 nt + 4 fq .. +3
  if constexpr (EPI == OWC_EPI_SWIGLU) {
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      const int m = mt * 16 + fr;
      bf16x4 o;
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = f2bf(rbf(act_silu(rbf(acc[mt][0][e]))) * rbf(acc[mt][1][e]));
      if (m < M) *(bf16x4*)(C + (long)m * ldc + (n0 >> 1) + fq * 4) = o;
    }
  } else {
    const int n = n0 + fq * 4;
    float bv[4] = {0.f, 0.f, 0.f, 0.f};
    if (bias != nullptr) {
      const bf16x4 b = *(const bf16x4*)(bias + n);
#pragma unroll
      for (int e = 0; e < 4; ++e) bv[e] = bf2f(b[e]);
    }
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      const int m = mt * 16 + fr;
      bf16x4 o;
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


// ------------------------------------------------------------------------------------------------
// Skinny-M GEMM that RMS-normalises its own activations (round 3; M <= 2 by default, at most 4: greedy decode at the reference's batch size).
// At batch 1 a decoder layer was eight launches and the two RMSNorms - ONE row of 7 KB each - cost 6.9 us apiece (a launch
// cannot be shorter than that on this GPU): 9 % of the step.  Here every wave (= block) of the qkv / gate-up projection first
// issues its W ring, then - while those loads fly - normalises the M raw residual rows itself, with the arithmetic of
// norm_kernel bit for bit (owc_rms_rstd / owc_rms_apply), into LDS (M x K bf16, rows padded by 16 B against bank conflicts), and
// takes its activation fragments from there: the activations no longer ride the register ring, so the same registers hold a
// deeper W ring.  The work is redundant across the N / 16 waves but M <= 4 rows cost ~0.5 us each, under the W latency.
// Results are those of rmsnorm followed by gemm_bf16_skinny_kernel: bit-identical (tested), so batch invariance holds
// across the switch to the separate kernels.
// ------------------------------------------------------------------------------------------------
template <int EPI, int DEPTH>
__global__ __launch_bounds__(64) void gemm_bf16_skinny_norm_kernel(
    const bf16_t* __restrict__ X, long ldx, const bf16_t* __restrict__ gamma, float eps, const bf16_t* __restrict__ W, long ldw,
    const bf16_t* __restrict__ bias, bf16_t* C, long ldc, int M, int N, int K) {
  constexpr int NT = EPI == OWC_EPI_SWIGLU ? 2 : 1;
  extern __shared__ __attribute__((aligned(16))) char lds[];   // [M][K * 2 + 16] activations (raw, then normalised in place) | [K * 2] gamma
  const int l = threadIdx.x;
  const int fr = l & 15, fq = l >> 4;
  const int n0 = blockIdx.x * (16 * NT);
  const int nss = K >> 7;
  const int pitch = K * 2 + 16;

  const bf16_t* wrow[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) wrow[nt] = W + (long)(n0 + nt * 16 + fr) * ldw + fq * 8;
  bf16x8 ring[DEPTH][NT][4];
  auto load_w = [&](int i, int ss) {
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) ring[i][nt][ks] = *(const bf16x8*)(wrow[nt] + ss * 128 + ks * 32);
  };
  // The raw rows and gamma go into LDS (1-KiB LDS-DMA pieces) BEFORE the ring is issued: loads return in order, so issued behind the
  // ring they - and with them the norm phase and the first MFMA - waited for DEPTH super-steps of HBM latency (round 3: qkv at
  // M = 1 15.2 us; the same bytes without any arithmetic 9.7 us).  Needs K % 512 == 0 (whole pieces); otherwise the old order.
  const int nch = K >> 3;
  const bool pre = (nch & 63) == 0;
  char* gl = lds + M * pitch;   // gamma, staged (pre only)
  if (pre) {
    for (int j = 0; j < (nch >> 6); ++j) {
      glds16(gamma + (j * 64 + l) * 8, gl + j * 1024);
      for (int m = 0; m < M; ++m) glds16(X + (long)m * ldx + (j * 64 + l) * 8, lds + m * pitch + j * 1024);
    }
  }
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int i = 0; i < DEPTH; ++i) load_w(i, min(i, nss - 1));
  __builtin_amdgcn_sched_barrier(0);

  // ---- normalise the M rows in LDS while the ring fills (the arithmetic of owc_rms_rstd / owc_rms_apply, bit for bit)
  if (pre) {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(DEPTH * NT * 4) : "memory");   // the staged rows have landed; the ring may fly
    for (int m = 0; m < M; ++m) {
      char* xrow = lds + m * pitch;
      float sq = 0.f;
      for (int ch = l; ch < nch; ch += 64) {
        const bf16x8 c = *(const bf16x8*)(xrow + ch * 16);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float v = bf2f(c[e]);
          sq = __builtin_fmaf(v, v, sq);
        }
      }
      sq = wave_sum(sq);
      const float rstd = rsqrtf(__builtin_fmaf(sq, 1.0f / (float)K, eps));
      for (int ch = l; ch < nch; ch += 64) {   // in place: a lane rewrites the chunks it read
        const bf16x8 c = *(const bf16x8*)(xrow + ch * 16), g = *(const bf16x8*)(gl + ch * 16);
        bf16x8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = owc_rms_apply(c[e], g[e], rstd);
        *(bf16x8*)(xrow + ch * 16) = o;
      }
    }
  } else {
    for (int m = 0; m < M; ++m) {
      const bf16_t* x = X + (long)m * ldx;
      const float rstd = owc_rms_rstd(x, K, eps, l);
      for (int ch = l; ch < nch; ch += 64) {
        const bf16x8 c = *(const bf16x8*)(x + ch * 8), g = *(const bf16x8*)(gamma + ch * 8);
        bf16x8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = owc_rms_apply(c[e], g[e], rstd);
        *(bf16x8*)(lds + m * pitch + ch * 16) = o;
      }
    }
  }
  const char* arow = lds + min(fr, M - 1) * pitch + fq * 16;

  f32x4 acc[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) acc[nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
  auto compute = [&](int i, int ss) {
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const bf16x8 xa = *(const bf16x8*)(arow + ss * 256 + ks * 64);
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) acc[nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ring[i][nt][ks], xa, acc[nt], 0, 0, 0);
    }
  };
  int ss = 0;
  for (; ss + DEPTH <= nss; ss += DEPTH) {
#pragma unroll
    for (int i = 0; i < DEPTH; ++i) {
      compute(i, ss + i);
      load_w(i, min(ss + i + DEPTH, nss - 1));
    }
  }
  const int rem = nss - ss;
#pragma unroll
  for (int i = 0; i < DEPTH; ++i)
    if (i < rem) compute(i, ss + i);
  // epilogue: identical to gemm_bf16_skinny_kernel's (lane holds row fr, columns n0 + 16 nt + 4 fq .. +3)
  const int m = fr;
  if constexpr (EPI == OWC_EPI_SWIGLU) {
    bf16x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = f2bf(rbf(act_silu(rbf(acc[0][e]))) * rbf(acc[1][e]));
    if (m < M) *(bf16x4*)(C + (long)m * ldc + (n0 >> 1) + fq * 4) = o;
  } else {
    const int n = n0 + fq * 4;
    float bv[4] = {0.f, 0.f, 0.f, 0.f};
    if (bias != nullptr) {
      const bf16x4 b = *(const bf16x4*)(bias + n);
#pragma unroll
      for (int e = 0; e < 4; ++e) bv[e] = bf2f(b[e]);
    }
    bf16x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = f2bf(acc[0][e] + bv[e]);
    if (m < M) *(bf16x4*)(C + (long)m * ldc + n) = o;
  }
}

template <int EPI>
bool launch_skinny(const void* A, long lda, const void* W, long ldw, const void* bias, const void* R, long ldr, void* C,
                   long ldc, int M, int N, int K, hipStream_t s) {
  if constexpr (EPI == OWC_EPI_NONE || EPI == OWC_EPI_RESIDUAL || EPI == OWC_EPI_SWIGLU) {  // what a decoder step needs
    constexpr int ROWS = EPI == OWC_EPI_SWIGLU ? 32 : 16;
    if (M > 64 || (K & 127) || (N % ROWS) || g_skinny_max_m < M) return false;
    if (EPI == OWC_EPI_SWIGLU && bias != nullptr) return false;   // the gated vision MLP's biases: the tiled kernels' epilogue adds them
    // more 16-row waves than CUs (the 7B qkv projection, N = 4608): the ring kernel's 32x32 tiles with four K-tiles per stage are
    // faster from M = 3 (M = 8 / 16: 13.8 / 15.7 -> 11.7 / 11.9 us; below that the RMSNorm-fused form of this kernel runs)
    if (EPI != OWC_EPI_SWIGLU && M > 2 && N / ROWS > 256 && N / 32 <= 256 && (K % BK) == 0 && g_small_tiles && g_k_pairs && K <= 4096 &&
        K >= g_k_pairs_min_k)
      return false;   // (N / 32 <= 256: one round of 32x32 tiles - not lm_head)
    const dim3 grid(N / ROWS), block(64);
#define OWC_SK(MT_, D_)                                                                                               \
  hipLaunchKernelGGL((gemm_bf16_skinny_kernel<EPI, MT_, D_>), grid, block, 0, s, (const bf16_t*)A, lda, (const bf16_t*)W, \
                     ldw, (const bf16_t*)bias, (const bf16_t*)R, ldr, (bf16_t*)C, ldc, M, N, K)
    constexpr int NT_ = EPI == OWC_EPI_SWIGLU ? 2 : 1;
    // super-steps in flight per wave: 16 VGPRs each per (n tile + m tile), about 192 VGPRs of ring in all
    // One wave per CU or less and a long K (the 7B down projection: 224 waves x 606 KB): occupancy is no concern, bytes in flight per
    // wave are - a ring of 9 super-steps (36 KiB of W in flight per wave, accumulators spill into AGPRs, no scratch) instead of 6:
    // M = 1 / 8 / 16: 32.1 -> 25.4 / 35.4 -> 30.7 / 38.8 -> 36.8 us (5.3 TB/s at M = 1).  Everywhere else the deeper ring LOSES
    // (gate/up 53 -> 70 us: its 1184 waves want two per SIMD; K = 3584 launches are over before the ring pays), so it stays off there.
    if (g_skinny_deep && M <= 16 && NT_ == 1 && grid.x <= 256 && K >= 8192) {
      OWC_SK(1, 9);
    } else {
      if (M <= 16) OWC_SK(1, 12 / (NT_ + 1)); else if (M <= 32) OWC_SK(2, 12 / (NT_ + 2)); else if (M <= 48) OWC_SK(3, 12 / (NT_ + 3));
      else OWC_SK(4, 2);
    }
#undef OWC_SK
    return true;
  }
  return false;
}

int cu_count() {   // compute units of the current device (the persistent kernel's grid)
  static int n = 0;
  if (!n) {
    int dev = 0;
    hipDeviceProp_t p;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&p, dev) == hipSuccess) n = p.multiProcessorCount;
    if (n <= 0) n = -1;
  }
  return n;
}

template <int EPI>
int launch(const void* A, long lda, const void* W, long ldw, const void* bias, const void* R,
           long ldr, void* C, long ldc, int M, int N, int K, const void* zeros, hipStream_t s,
           const owc_gemm_aux& aux) {
  // 256x256 tiles need enough of them to fill the 256 CUs (one block per CU); otherwise 128x128 (2 per CU)
  // ... and, below 1024 rows, not when M is three 128-row tiles (the second 256-row tile half empty).  Measured on the 7B gate/up
  // projection, 256x256 against 128x128 tiles: M = 192 / 256 / 512 / 768 86 <- 120 / 81 <- 135 / 155 <- 198 / 163 <- 255 us, but
  // M = 384 149 <- 138 us.  (Round 2's rule - padding to 256 wastes at most an eighth of the rows - left M = 129..223 and
  // 385..455 on the 128x128 kernel: the B = 192 decode step was slower than the B = 256 one.)
  const int mt128 = (M + BM - 1) / BM;
  const bool big = M >= g_big_min_m && (M >= 1024 || mt128 != 3) && N >= BT && (K % BK) == 0 &&
                   (long)((M + BT - 1) / BT) * ((N + BT - 1) / BT) >= g_big_min_tiles;
  const int tiles_m = big ? (M + BT - 1) / BT : (M + BM - 1) / BM;
  const int tiles_n = big ? (N + BT - 1) / BT : (N + BN - 1) / BN;
  static bool attr_set = false;
  if (!attr_set) {
    if (hipFuncSetAttribute((const void*)gemm_bf16_nt_kernel<EPI, true>,
                            hipFuncAttributeMaxDynamicSharedMemorySize, 4 * TILE_BYTES) != hipSuccess ||
        hipFuncSetAttribute((const void*)gemm_bf16_nt_kernel<EPI, false>,
                            hipFuncAttributeMaxDynamicSharedMemorySize, 4 * TILE_BYTES) != hipSuccess ||
        hipFuncSetAttribute((const void*)gemm_bf16_nt_64_kernel<EPI, 4, true>,
                            hipFuncAttributeMaxDynamicSharedMemorySize, 4 * 2 * TILE64_BYTES) != hipSuccess ||
        hipFuncSetAttribute((const void*)gemm_bf16_nt_64_kernel<EPI, 4, false>,
                            hipFuncAttributeMaxDynamicSharedMemorySize, 4 * 2 * TILE64_BYTES) != hipSuccess ||
        hipFuncSetAttribute((const void*)gemm_bf16_nt_256_kernel<EPI>,
                            hipFuncAttributeMaxDynamicSharedMemorySize, 2 * STAGE_BYTES) != hipSuccess ||
        hipFuncSetAttribute((const void*)gemm_bf16_nt_256pp_kernel<EPI>,
                            hipFuncAttributeMaxDynamicSharedMemorySize, 2 * STAGE_BYTES) != hipSuccess ||
        hipFuncSetAttribute((const void*)gemm_bf16_nt_256x128pp_kernel<EPI>,
                            hipFuncAttributeMaxDynamicSharedMemorySize, P128_LDS) != hipSuccess)
      return OWC_ERR_HIP;
    attr_set = true;
  }
  const int prof = gemm_profile_begin_shape(M, N, K, EPI, s);
  // (the wide ring tiles below beat the skinny kernel on their shapes at every M: 48.5 us against 53.9 ... 70.1 us at M = 1 ... 32)
  const bool wide = g_tall_tiles && (K % BK) == 0 && (EPI == OWC_EPI_NONE || EPI == OWC_EPI_SWIGLU) && M <= 128 &&
                    (N + 159) / 160 <= 256 && (N + 159) / 160 >= g_wide_min_blocks;
  if (!wide && launch_skinny<EPI>(A, lda, W, ldw, bias, R, ldr, C, ldc, M, N, K, s)) {
    owc_gemm_profile_end(prof, s);
    return hipGetLastError() == hipSuccess ? OWC_OK : OWC_ERR_HIP;
  }
  // too few 256x256 tiles to fill the chip, but one round of 256x128 tiles does (decode o / down projections at 1024-2048 rows)
  const long t128 = (long)((M + 255) / 256) * ((N + 127) / 128);
  const bool pp128 = !big && !wide && g_pp128_min_tiles > 0 && g_pingpong && M >= g_big_min_m && N >= 128 && (K % BK) == 0 && K >= 2 * BK &&
                     t128 >= g_pp128_min_tiles && t128 <= 256;
  if (pp128) {
    hipLaunchKernelGGL(gemm_bf16_nt_256x128pp_kernel<EPI>, dim3((int)t128), dim3(512), P128_LDS, s, (const bf16_t*)A, lda, (const bf16_t*)W,
                       ldw, (const bf16_t*)bias, (const bf16_t*)R, ldr, C, ldc, M, N, K, (M + 255) / 256, (N + 127) / 128, g_gemm_dbg, aux);
  }
  else if (big && g_pingpong && (K % (2 * BK)) == 0) {
    // C far larger than L2 + Infinity Cache (256 MB): stream it out (`global_store ... nt`) instead of evicting the operand panels the
    // XCD's L2 is sharing: +0.3...3 % per launch on the path's shapes (vit.proj 1154 -> 1192 TFLOP/s, gate/up 1459 -> 1480), never
    // slower; a small C (decode steps) stays cacheable for the kernel that reads it next
    const int flags = g_gemm_dbg | ((size_t)M * (size_t)N * 2 > ((size_t)g_nt_min_mb << 20) ? 1024 : 0) |
                      ((g_walk == 2 || (g_walk == 1 && tiles_m >= tiles_n && 2 * (long)K >= 9 * (long)N)) ? FLAG_COLWALK : 0);
    if constexpr (EPI != OWC_EPI_F32 && EPI != OWC_EPI_SWIGLU) {   // (the persistent form: off by default, see g_persist)
      const int cus = cu_count();
      if (g_persist && K >= 6 * BK && tiles_m * tiles_n > cus && cus >= 8) {
        static bool set_ = false;
        if (!set_) {
          if (hipFuncSetAttribute((const void*)gemm_bf16_nt_256pp_persist_kernel<EPI, false>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                  2 * STAGE_BYTES) != hipSuccess ||
              hipFuncSetAttribute((const void*)gemm_bf16_nt_256pp_persist_kernel<EPI, true>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                  2 * STAGE_BYTES) != hipSuccess) return OWC_ERR_HIP;
          set_ = true;
        }
        if (flags & 1024)
          hipLaunchKernelGGL((gemm_bf16_nt_256pp_persist_kernel<EPI, true>), dim3(cus & ~7), dim3(512), 2 * STAGE_BYTES, s, (const bf16_t*)A, lda,
                             (const bf16_t*)W, ldw, (const bf16_t*)bias, (const bf16_t*)R, ldr, C, ldc, M, N, K, tiles_m, tiles_n, flags, aux);
        else
          hipLaunchKernelGGL((gemm_bf16_nt_256pp_persist_kernel<EPI, false>), dim3(cus & ~7), dim3(512), 2 * STAGE_BYTES, s, (const bf16_t*)A, lda,
                             (const bf16_t*)W, ldw, (const bf16_t*)bias, (const bf16_t*)R, ldr, C, ldc, M, N, K, tiles_m, tiles_n, flags, aux);
        owc_gemm_profile_end(prof, s);
        return hipGetLastError() == hipSuccess ? OWC_OK : OWC_ERR_HIP;
      }
    }
    // (round 6) A SHORT LAST ROUND: T tiles on 256 CUs take ceil(T / 256) rounds, and the decode steps' gate/up projection sits on
    // the worst counts (M = 512: 296 tiles = 1.16 rounds for the price of 2; M = 1024: 592 = 2.3 for 3; M = 1792: 1036 = 4.05 for 5).
    // When the tiles beyond the whole rounds are at most half a round, the tile COLUMNS of the whole rounds go to this kernel and the
    // rest of N to one round of 256 x 128 tiles (`gemm_bf16_nt_256x128pp_kernel`, <= 256 blocks, each ~0.6 of a 256 x 256 tile's
    // time): 2 -> ~1.6 rounds at M = 512.  A column split changes no output element's K chain: bit-identical
    // (tests/test_gemm_gpu.py::test_gemm_short_last_round_split_does_not_change_a_bit).  Plain and SwiGLU epilogues only (no
    // per-column side inputs to offset); up to 8 rounds (beyond that the last round is < 3 % of the launch).
    if constexpr (EPI == OWC_EPI_NONE || EPI == OWC_EPI_SWIGLU) {
      const int cus = cu_count();
      const long T = (long)tiles_m * tiles_n;
      if (g_tail_split && bias == nullptr && cus >= 64 && T > cus && T <= 8L * cus && K >= 2 * BK) {
        const int cols1 = (int)((T / cus) * cus / tiles_m);            // tile columns that fill the whole rounds
        const long rest = T - (long)cols1 * tiles_m;
        const int n1 = cols1 * BT, n2 = N - n1;
        const int tn2 = (n2 + 127) / 128;
        if (cols1 > 0 && rest > 0 && 2 * rest <= cus && (long)tiles_m * tn2 <= cus) {
          const long c_off = EPI == OWC_EPI_SWIGLU ? (long)(n1 / 2) * 2 : (long)n1 * 2;   // bytes: both epilogues write bf16
          hipLaunchKernelGGL(gemm_bf16_nt_256pp_kernel<EPI>, dim3(tiles_m * cols1), dim3(512), 2 * STAGE_BYTES, s,
                             (const bf16_t*)A, lda, (const bf16_t*)W, ldw, (const bf16_t*)nullptr,
                             (const bf16_t*)R, ldr, C, ldc, M, n1, K, tiles_m, cols1, flags, aux);
          hipLaunchKernelGGL(gemm_bf16_nt_256x128pp_kernel<EPI>, dim3(tiles_m * tn2), dim3(512), P128_LDS, s, (const bf16_t*)A, lda,
                             (const bf16_t*)((const char*)W + (long)n1 * ldw * 2), ldw, (const bf16_t*)nullptr, (const bf16_t*)R, ldr,
                             (void*)((char*)C + c_off), ldc, M, n2, K, tiles_m, tn2, g_gemm_dbg, aux);
          owc_gemm_profile_end(prof, s);
          return hipGetLastError() == hipSuccess ? OWC_OK : OWC_ERR_HIP;
        }
      }
    }
    hipLaunchKernelGGL(gemm_bf16_nt_256pp_kernel<EPI>, dim3(tiles_m * tiles_n), dim3(512), 2 * STAGE_BYTES, s,
                       (const bf16_t*)A, lda, (const bf16_t*)W, ldw, (const bf16_t*)bias,
                       (const bf16_t*)R, ldr, C, ldc, M, N, K, tiles_m, tiles_n, flags, aux);
  }
  else if (big)
    hipLaunchKernelGGL(gemm_bf16_nt_256_kernel<EPI>, dim3(tiles_m * tiles_n), dim3(512), 2 * STAGE_BYTES, s,
                       (const bf16_t*)A, lda, (const bf16_t*)W, ldw, (const bf16_t*)bias,
                       (const bf16_t*)R, ldr, C, ldc, M, N, K, tiles_m, tiles_n, g_gemm_dbg, aux);
  else if (!wide && g_mid_max_tiles > 0 && tiles_m * tiles_n < g_mid_max_tiles) {  // too few 128x128 tiles for 256 CUs: 64x64 tiles
    const int tm64 = (M + B64 - 1) / B64, tn64 = (N + B64 - 1) / B64;
#define OWC_L64(NS_, KT_, TM_, TN_, ...)                                                                                   \
  do {                                                                                                                     \
    constexpr int kps_ = (0, ##__VA_ARGS__) ? (0, ##__VA_ARGS__) : 1;                                                      \
    constexpr int lds_ = NS_ * kps_ * (TM_ + TN_) * 128;                                                                   \
    static bool set_ = false;                                                                                              \
    if (!set_ && lds_ > 65536) {                                                                                           \
      if (hipFuncSetAttribute((const void*)gemm_bf16_nt_64_kernel<EPI, NS_, KT_, TM_, TN_, kps_>,                          \
                              hipFuncAttributeMaxDynamicSharedMemorySize, lds_) != hipSuccess) return OWC_ERR_HIP;         \
      set_ = true;                                                                                                         \
    }                                                                                                                      \
    const int tm_ = (M + TM_ - 1) / TM_, tn_ = (N + TN_ - 1) / TN_;                                                        \
    hipLaunchKernelGGL((gemm_bf16_nt_64_kernel<EPI, NS_, KT_, TM_, TN_, kps_>), dim3(tm_ * tn_), dim3(256), lds_, s,       \
                       (const bf16_t*)A, lda, (const bf16_t*)W, ldw, (const bf16_t*)bias, (const bf16_t*)R, ldr, C, ldc,   \
                       M, N, K, zeros, tm_, tn_, aux);                                                                     \
  } while (0)
    const bool ktail = (K % BK) != 0;
    // Tile shape (measured, profiles/r03_gemm_small_tiles_ab.txt): a launch that leaves CUs idle is bound by what ONE CU pulls from HBM
    // (~40-55 GB/s of misses, see the kernel's header), so its time goes with the W bytes per block - halving TN nearly halves it,
    // halving TM (L2 hits) barely matters on its own but doubles the blocks.  Hence: 64x64 once it fills the chip; else 32x32 when
    // that still fits one round of 256 CUs; else 64x32 up to two rounds.  (32x64 measured -3 %: not instantiated; SwiGLU pairs 64
    // columns and its shapes have many n tiles anyway.)
    int shape = 0;   // 0: 64x64, 1: 64x32, 2: 32x32
    if constexpr (EPI == OWC_EPI_NONE || EPI == OWC_EPI_RESIDUAL) {
      if (!ktail && g_small_tiles) {
        auto blocks = [&](int tm, int tn) { return (long)((M + tm - 1) / tm) * ((N + tn - 1) / tn); };
        if (blocks(64, 64) > 256) shape = 0;
        else if (blocks(32, 32) <= 256) shape = 2;
        else if (blocks(64, 32) <= 512) shape = 1;
        if (g_small_tiles > 1) shape = (g_small_tiles - 2) % 3;   // A-B: force a shape
      }
    }
    bool done128 = false;
    if constexpr (EPI == OWC_EPI_NONE || EPI == OWC_EPI_RESIDUAL) {
      // more than one round of 64x64 tiles (M >= 257 on the N = 3584 projections; where a long-answer task's decode loop lives): 128x64
      // tiles halve the blocks and cut the L2 -> LDS bytes by a quarter (A 128 + W 64 rows per K-tile for twice the output); up to 256
      // blocks one per CU with two K-tiles per stage, above that 72 KiB of LDS per block so that two share a CU (one round to 512)
      if (g_ring_128 && shape == 0 && !ktail && tm64 * tn64 > 384 && M >= 129) {   // (at 257-384 blocks of 64x64 it is +-2 %: they stay)
        const long nb128 = (long)((M + 127) / 128) * tn64;
        if (nb128 > 256) OWC_L64(3, false, 128, 64);          // 72 KiB of LDS: two blocks per CU, one round up to 512 blocks
        else if (g_k_pairs && K >= g_k_pairs_min_k) OWC_L64(3, false, 128, 64, 2);
        else OWC_L64(4, false, 128, 64);
        done128 = true;
      }
    }
    if (shape == 0 && !done128) {
      if (g_k_pairs && !ktail && K >= g_k_pairs_min_k && tm64 * tn64 <= 512) OWC_L64(2, false, 64, 64, 2);   // (qkv at M = 300-400: 33 -> 29 us)
      else if (tm64 * tn64 > 512 && tm64 * tn64 <= 768) { if (ktail) OWC_L64(3, true, 64, 64); else OWC_L64(3, false, 64, 64); }
      else { if (ktail) OWC_L64(4, true, 64, 64); else OWC_L64(4, false, 64, 64); }
    }
    if constexpr (EPI == OWC_EPI_NONE || EPI == OWC_EPI_RESIDUAL) {
      // long K, one block per CU: two K-tiles per stage (one wait + barrier per 128 of K)
      const long nb = shape == 1 ? (long)((M + 63) / 64) * ((N + 31) / 32) : (long)((M + 31) / 32) * ((N + 31) / 32);
      const bool longk = g_k_pairs && K >= g_k_pairs_min_k;
      const int kps = (longk && nb <= 256) ? (g_k_pairs == 2 ? 2 : 4) : 1;
      if (shape == 1) {
        if (kps == 4) OWC_L64(3, false, 64, 32, 4);
        else if (kps == 2) OWC_L64(4, false, 64, 32, 2);
        else if (longk && (long)tm64 * tn64 <= 256) OWC_L64(3, false, 64, 64, 2);   // two rounds of 64x32: one of 64x64 (M = 129-256: 115 -> 89 us)
        else OWC_L64(6, false, 64, 32);
      }
      if (shape == 2) {
        if (kps == 4) OWC_L64(4, false, 32, 32, 4);
        else if (kps == 2) OWC_L64(5, false, 32, 32, 2);
        else OWC_L64(8, false, 32, 32);
      }
    }
#undef OWC_L64
  } else if (wide) {
    // At most 128 rows x tens of thousands of columns (the 7B gate/up projection of a decode step up to batch 128): a weight stream in which
    // every block ALSO re-reads the whole A (M x K) out of L2, and what a CU can ingest (A + W, ~40-58 GB/s with misses in the
    // mix) is the bound - so the tile is as WIDE as it can be while the launch still fills the chip in one round: 160 columns
    // (N = 37888: 237 blocks; per CU 0.9 MB of A + 1.15 MB of W instead of 2.3 x (0.9 + 0.46) with 128x64 / 128x128 tiles),
    // four 28 / 36-KiB stages.  (Measured first: 128x64 tiles with three stages - more bytes in flight, same bytes per CU - were
    // 4 % SLOWER than the 128x128 kernel.)
    constexpr int NS_ = 4, TN_ = 160;
#define OWC_LWIDE(TM_)                                                                                                      \
  do {                                                                                                                      \
    constexpr int lds_ = NS_ * (TM_ + TN_) * 128;                                                                           \
    static bool set_ = false;                                                                                               \
    if (!set_) {                                                                                                            \
      if (hipFuncSetAttribute((const void*)gemm_bf16_nt_64_kernel<EPI, NS_, false, TM_, TN_>,                              \
                              hipFuncAttributeMaxDynamicSharedMemorySize, lds_) != hipSuccess) return OWC_ERR_HIP;          \
      set_ = true;                                                                                                          \
    }                                                                                                                       \
    const int tm_ = (M + TM_ - 1) / TM_, tn_ = (N + TN_ - 1) / TN_;                                                         \
    hipLaunchKernelGGL((gemm_bf16_nt_64_kernel<EPI, NS_, false, TM_, TN_>), dim3(tm_ * tn_), dim3(256), lds_, s,            \
                       (const bf16_t*)A, lda, (const bf16_t*)W, ldw, (const bf16_t*)bias, (const bf16_t*)R, ldr, C, ldc, M, \
                       N, K, zeros, tm_, tn_, aux);                                                                         \
  } while (0)
    if constexpr (EPI == OWC_EPI_NONE || EPI == OWC_EPI_SWIGLU) {
      // (two K-tiles per stage, measured: +-0 ... -12 % here - these launches run at the HBM rate, not at the per-K-tile cost)
      if (M <= 32) OWC_LWIDE(32); else if (M <= 64) OWC_LWIDE(64); else OWC_LWIDE(128);
    }
#undef OWC_LWIDE
  } else
  {
    if ((K % BK) != 0)
      hipLaunchKernelGGL((gemm_bf16_nt_kernel<EPI, true>), dim3(tiles_m * tiles_n), dim3(256), 4 * TILE_BYTES, s,
                         (const bf16_t*)A, lda, (const bf16_t*)W, ldw, (const bf16_t*)bias,
                         (const bf16_t*)R, ldr, C, ldc, M, N, K, zeros, tiles_m, tiles_n, aux);
    else
      hipLaunchKernelGGL((gemm_bf16_nt_kernel<EPI, false>), dim3(tiles_m * tiles_n), dim3(256), 4 * TILE_BYTES, s,
                         (const bf16_t*)A, lda, (const bf16_t*)W, ldw, (const bf16_t*)bias,
                         (const bf16_t*)R, ldr, C, ldc, M, N, K, zeros, tiles_m, tiles_n, aux);
  }
  owc_gemm_profile_end(prof, s);
  return hipGetLastError() == hipSuccess ? OWC_OK : OWC_ERR_HIP;
}

}  // namespace

// Internal entry used by the C ABI (api.cpp) and by the model drivers.
int owc_launch_gemm_bf16_aux(const void* A, long lda, const void* W, long ldw, const void* bias,
                             const void* R, long ldr, void* C, long ldc, int M, int N, int K, int epi,
                             const void* zeros, hipStream_t s, const owc_gemm_aux* auxp) {
  if (M <= 0 || N <= 0 || K <= 0) return OWC_ERR_SHAPE;
  if ((K & 7) || (lda & 7) || (ldw & 7) || (N & 3) || (ldc & 3)) return OWC_ERR_SHAPE;
  if (epi != OWC_EPI_F32 && epi != OWC_EPI_SWIGLU && ((N & 7) || (ldc & 7))) return OWC_ERR_SHAPE;  // 16-byte stores
  if (epi == OWC_EPI_SWIGLU && (ldc & 7)) return OWC_ERR_SHAPE;
  if (epi == OWC_EPI_SWIGLU && (N & 31)) return OWC_ERR_SHAPE;
  if (epi == OWC_EPI_RESIDUAL && (R == nullptr || (ldr & 7))) return OWC_ERR_ARG;
  owc_gemm_aux aux = {nullptr, nullptr, nullptr, 0, 8};
  if (epi == OWC_EPI_VROPE) {
    if (!auxp || !auxp->pos_hw || !auxp->cos_t || !auxp->sin_t || (auxp->head_dim % 16) || (auxp->rope_cols % 8))
      return OWC_ERR_ARG;
    aux = *auxp;
  }

  switch (epi) {
    case OWC_EPI_NONE: return launch<OWC_EPI_NONE>(A, lda, W, ldw, bias, R, ldr, C, ldc, M, N, K, zeros, s, aux);
    case OWC_EPI_QUICK_GELU: return launch<OWC_EPI_QUICK_GELU>(A, lda, W, ldw, bias, R, ldr, C, ldc, M, N, K, zeros, s, aux);
    case OWC_EPI_GELU_ERF: return launch<OWC_EPI_GELU_ERF>(A, lda, W, ldw, bias, R, ldr, C, ldc, M, N, K, zeros, s, aux);
    case OWC_EPI_RESIDUAL: return launch<OWC_EPI_RESIDUAL>(A, lda, W, ldw, bias, R, ldr, C, ldc, M, N, K, zeros, s, aux);
    case OWC_EPI_SWIGLU: return launch<OWC_EPI_SWIGLU>(A, lda, W, ldw, bias, R, ldr, C, ldc, M, N, K, zeros, s, aux);
    case OWC_EPI_F32: return launch<OWC_EPI_F32>(A, lda, W, ldw, bias, R, ldr, C, ldc, M, N, K, zeros, s, aux);
    case OWC_EPI_VROPE: return launch<OWC_EPI_VROPE>(A, lda, W, ldw, bias, R, ldr, C, ldc, M, N, K, zeros, s, aux);
    default: return OWC_ERR_ARG;
  }
}

// rmsnorm(X) . W^T (+bias | SwiGLU) for M <= OWC_NORM_FUSE_MAX_M rows in one launch; OWC_ERR_SHAPE when the shape is outside what the fused
// kernel takes (the caller then runs the two separate kernels: same bits).
// ring kernel, norm-fused form (M <= 8): W-only ring, the normalised rows resident in LDS
template <int EPI, int NS_, int TN_, int KPS_>
static int launch_ring_norma(const void* X, long ldx, const void* gamma, float eps, const void* W, long ldw, const void* bias, void* C,
                             long ldc, int M, int N, int K, hipStream_t s) {
  const int lds_bytes = 8 * (K * 2 + 16) + K * 2 + NS_ * KPS_ * TN_ * 128;
  if (lds_bytes > 160 * 1024) return OWC_ERR_SHAPE;
  static bool set_ = false;
  if (!set_) {
    if (hipFuncSetAttribute((const void*)gemm_bf16_nt_64_kernel<EPI, NS_, false, 32, TN_, KPS_, true>,
                            hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
      return OWC_ERR_HIP;
    set_ = true;
  }
  owc_gemm_aux aux = {nullptr, nullptr, nullptr, 0, 8};
  aux.gamma = gamma;
  aux.eps = eps;
  const int tn_ = (N + TN_ - 1) / TN_;
  const int prof = gemm_profile_begin_shape(M, N, K, EPI, s);
  hipLaunchKernelGGL((gemm_bf16_nt_64_kernel<EPI, NS_, false, 32, TN_, KPS_, true>), dim3(tn_), dim3(256), lds_bytes, s, (const bf16_t*)X,
                     ldx, (const bf16_t*)W, ldw, (const bf16_t*)bias, (const bf16_t*)nullptr, 0, C, ldc, M, N, K, nullptr, 1, tn_, aux);
  owc_gemm_profile_end(prof, s);
  return hipGetLastError() == hipSuccess ? OWC_OK : OWC_ERR_HIP;
}

int owc_launch_gemm_bf16_rmsnorm(const void* X, long ldx, const void* gamma, float eps, const void* W, long ldw, const void* bias,
                                 void* C, long ldc, int M, int N, int K, int epi, hipStream_t s) {
  // M <= 8 on the ring kernel (round 3): the wide gate/up tiles and the 32-column qkv tiles with the RMSNorm folded in
  if (g_norm_fuse_ring && M > 0 && M <= g_norm_fuse_ring && M <= 8 && (K & 511) == 0 && !(ldx & 7) && !(ldw & 7) && g_norm_fuse_max_m > 0) {
    if (epi == OWC_EPI_SWIGLU && bias == nullptr && g_tall_tiles && (N & 31) == 0 && (ldc & 3) == 0 && (N + 159) / 160 <= 256 &&
        (N + 159) / 160 >= g_wide_min_blocks) {
      const int rc = launch_ring_norma<OWC_EPI_SWIGLU, 4, 160, 1>(X, ldx, gamma, eps, W, ldw, bias, C, ldc, M, N, K, s);
      if (rc != OWC_ERR_SHAPE) return rc;
    }
    if (epi == OWC_EPI_NONE && g_small_tiles && g_k_pairs && (N & 7) == 0 && (ldc & 7) == 0 && N / 16 > 256 && N / 32 <= 256 &&
        K >= g_k_pairs_min_k) {
      const int rc = launch_ring_norma<OWC_EPI_NONE, 4, 32, 4>(X, ldx, gamma, eps, W, ldw, bias, C, ldc, M, N, K, s);
      if (rc != OWC_ERR_SHAPE) return rc;
    }
  }
  if (M <= 0 || M > g_norm_fuse_max_m || (K & 127) || K > 8192 || (ldx & 7) || (ldw & 7) || g_skinny_max_m < M) return OWC_ERR_SHAPE;
  const bool swiglu = epi == OWC_EPI_SWIGLU && (N & 31) == 0 && (ldc & 3) == 0;
  if (!swiglu && !(epi == OWC_EPI_NONE && (N & 15) == 0 && (ldc & 3) == 0)) return OWC_ERR_SHAPE;
  const int lds_bytes = M * (K * 2 + 16) + K * 2;   // the rows (normalised in place) + gamma
  static bool attr_set = false;
  if (!attr_set) {   // 4 rows of K = 8192 (the 72B decoder) are 65.6 KB: above the default dynamic-LDS limit
    if (hipFuncSetAttribute((const void*)gemm_bf16_skinny_norm_kernel<OWC_EPI_SWIGLU, 5>, hipFuncAttributeMaxDynamicSharedMemorySize,
                            4 * (8192 * 2 + 16) + 8192 * 2) != hipSuccess ||
        hipFuncSetAttribute((const void*)gemm_bf16_skinny_norm_kernel<OWC_EPI_NONE, 8>, hipFuncAttributeMaxDynamicSharedMemorySize,
                            4 * (8192 * 2 + 16) + 8192 * 2) != hipSuccess)
      return OWC_ERR_HIP;
    attr_set = true;
  }
  const int prof = gemm_profile_begin_shape(M, N, K, epi, s);
  if (swiglu) {
    hipLaunchKernelGGL((gemm_bf16_skinny_norm_kernel<OWC_EPI_SWIGLU, 5>), dim3(N / 32), dim3(64), lds_bytes, s, (const bf16_t*)X, ldx,
                       (const bf16_t*)gamma, eps, (const bf16_t*)W, ldw, (const bf16_t*)bias, (bf16_t*)C, ldc, M, N, K);
  } else {
    hipLaunchKernelGGL((gemm_bf16_skinny_norm_kernel<OWC_EPI_NONE, 8>), dim3(N / 16), dim3(64), lds_bytes, s, (const bf16_t*)X, ldx,
                       (const bf16_t*)gamma, eps, (const bf16_t*)W, ldw, (const bf16_t*)bias, (bf16_t*)C, ldc, M, N, K);
  }
  owc_gemm_profile_end(prof, s);
  return hipGetLastError() == hipSuccess ? OWC_OK : OWC_ERR_HIP;
}
void owc_gemm_set_norm_fuse_max_m(int v) { g_norm_fuse_max_m = v < 0 ? 2 : (v > 4 ? 4 : v); }

int owc_launch_gemm_bf16(const void* A, long lda, const void* W, long ldw, const void* bias,
                         const void* R, long ldr, void* C, long ldc, int M, int N, int K, int epi,
                         const void* zeros, hipStream_t s) {
  if (epi == OWC_EPI_VROPE) return OWC_ERR_ARG;  // needs the aux operands
  return owc_launch_gemm_bf16_aux(A, lda, W, ldw, bias, R, ldr, C, ldc, M, N, K, epi, zeros, s, nullptr);
}

// ---- profiling hooks (C ABI: owc_gemm_profile_enable / owc_gemm_profile_read in api.hip) ----
void owc_gemm_profile_set(int on) {
  g_prof.on = on != 0;
  g_prof.used = 0;
  g_prof.flops.clear();
  g_prof.kind.clear();
}

// Brackets one GEMM launch (bf16 or fp8) with an event pair when profiling is on; returns a handle for _end (-1: off).
int owc_gemm_profile_begin(double flops, int kind, hipStream_t s) {
  if (!g_prof.on) return -1;
  if (g_prof.used + 2 > g_prof.ev.size()) {
    hipEvent_t a, b;
    if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) return -1;
    g_prof.ev.push_back(a);
    g_prof.ev.push_back(b);
  }
  const int idx = (int)g_prof.used;
  g_prof.used += 2;
  g_prof.flops.push_back(flops);
  g_prof.kind.push_back(kind);
  g_prof.shape.push_back(ShapeKey());
  (void)hipEventRecord(g_prof.ev[idx], s);
  return idx;
}

// ... the same for a bf16 GEMM launch, which also records its shape: the per-shape table of the timed region (owc_profile_shapes)
static int gemm_profile_begin_shape(int M, int N, int K, int epi, hipStream_t s) {
  const int h = owc_gemm_profile_begin(2.0 * (double)M * (double)N * (double)K, 0, s);
  if (h >= 0) {
    ShapeKey& key = g_prof.shape.back();
    key.m = M;
    key.n = N;
    key.k = K;
    key.epi = epi;
  }
  return h;
}

void owc_gemm_profile_end(int handle, hipStream_t s) {
  if (handle >= 0) (void)hipEventRecord(g_prof.ev[handle + 1], s);
}

// Sums the recorded launches per launch class (the caller has synchronised the stream): kernel milliseconds, algorithmic work
// (FLOPs as given to _begin) and launch count for classes [0, n_kinds); then resets the recording.
int owc_profile_collect(int n_kinds, double* total_ms, double* total_work, long* launches) {
  for (int k = 0; k < n_kinds; ++k) {
    total_ms[k] = 0.0;
    total_work[k] = 0.0;
    launches[k] = 0;
  }
  const size_t n = g_prof.used / 2;
  g_prof.last_shapes.clear();
  for (size_t i = 0; i < n; ++i) {
    float t = 0.f;
    if (hipEventElapsedTime(&t, g_prof.ev[2 * i], g_prof.ev[2 * i + 1]) != hipSuccess) return OWC_ERR_HIP;
    const int k = g_prof.kind[i];
    if (k < 0 || k >= n_kinds) continue;
    total_ms[k] += t;
    total_work[k] += g_prof.flops[i];
    ++launches[k];
    const ShapeKey& key = g_prof.shape[i];
    if (key.epi >= 0) {   // a bf16 GEMM launch: group by (M, N, K, epilogue) - a few dozen distinct shapes per model, linear search
      ShapeStat* st = nullptr;
      for (auto& c : g_prof.last_shapes)
        if (c.key.m == key.m && c.key.n == key.n && c.key.k == key.k && c.key.epi == key.epi) {
          st = &c;
          break;
        }
      if (!st) {
        g_prof.last_shapes.push_back(ShapeStat());
        st = &g_prof.last_shapes.back();
        st->key = key;
        st->min_ms = st->max_ms = t;
      }
      ++st->launches;
      st->total_ms += t;
      st->min_ms = t < st->min_ms ? t : st->min_ms;
      st->max_ms = t > st->max_ms ? t : st->max_ms;
    }
  }
  g_prof.used = 0;
  g_prof.flops.clear();
  g_prof.kind.clear();
  g_prof.shape.clear();
  return OWC_OK;
}

// The bf16 GEMM launches of the LAST owc_profile_collect grouped by shape: entry i = shape[4 i ..] = (M, N, K, epilogue),
// stats[4 i ..] = (launches, total ms, min ms, max ms).  Returns the number of distinct shapes (which may exceed max_n: call again).
int owc_profile_shapes_collect(int max_n, int* shape, double* stats) {
  const int n = (int)g_prof.last_shapes.size();
  for (int i = 0; i < n && i < max_n; ++i) {
    const ShapeStat& c = g_prof.last_shapes[i];
    shape[4 * i] = c.key.m;
    shape[4 * i + 1] = c.key.n;
    shape[4 * i + 2] = c.key.k;
    shape[4 * i + 3] = c.key.epi;
    stats[4 * i] = (double)c.launches;
    stats[4 * i + 1] = c.total_ms;
    stats[4 * i + 2] = c.min_ms;
    stats[4 * i + 3] = c.max_ms;
  }
  return n;
}

int owc_gemm_profile_collect(double* total_ms, double* total_flops, long* launches) {  // classes 0 (bf16 GEMM), 1 (fp8 GEMM)
  return owc_profile_collect(2, total_ms, total_flops, launches);
}

void owc_gemm_set_big_min_m(int m) { g_big_min_m = m; }
void owc_gemm_set_dbg(int v) { g_gemm_dbg = OWC_TK(true) ? v : 0; }
void owc_gemm_set_mid_max_tiles(int v) { g_mid_max_tiles = v; }
void owc_gemm_set_big_min_tiles(int v) { g_big_min_tiles = v < 0 ? 144 : v; }
void owc_gemm_set_skinny_max_m(int v) { g_skinny_max_m = v < 0 ? 24 : v; }  // negative: back to the default
void owc_gemm_set_pingpong(int v) { g_pingpong = v; }
void owc_gemm_set_skinny_deep(int v) { g_skinny_deep = v; }
void owc_gemm_set_norm_fuse_ring(int v) { g_norm_fuse_ring = v < 0 ? 8 : (v > 8 ? 8 : v); }
void owc_gemm_set_k_pairs(int v) { g_k_pairs = v; }
void owc_gemm_set_k_pairs_min_k(int v) { g_k_pairs_min_k = v < 0 ? 1024 : v; }
void owc_gemm_set_tall_tiles(int v) { g_tall_tiles = v != 0; g_wide_min_blocks = v > 1 ? v : 128; }
void owc_gemm_set_small_tiles(int v) { g_small_tiles = v < 0 ? 1 : v; }
void owc_gemm_set_ring_128(int v) { g_ring_128 = v < 0 ? 1 : v != 0; }
void owc_gemm_set_pp128(int v) { g_pp128_min_tiles = v < 0 ? 128 : v; }
void owc_gemm_set_persist(int v) { g_persist = v < 0 ? 0 : v; }
void owc_gemm_set_walk(int v) { g_walk = v < 0 ? 1 : v; }
void owc_gemm_set_tail_split(int v) { g_tail_split = v < 0 ? 1 : v; }
void owc_gemm_set_nt_min_mb(int v) { g_nt_min_mb = v < 0 ? 64 : v; }
int owc_gemm_nt_min_mb() { return g_nt_min_mb; }   // (the fp8 kernels use the same threshold)
