// How many bytes per second can ONE CU pull from L2 / Infinity Cache / HBM, and does the path matter?
//   mode 0: LDS-DMA      (global_load_lds_dwordx4: 64 lanes x 16 B -> LDS), `depth` instructions in flight per wave
//   mode 1: register load (global_load_dwordx4 -> VGPRs), `depth` instructions in flight per wave
// Every wave streams its own contiguous slice of a buffer of `mb` MiB over and over (mb = 24: L2-resident, 192: Infinity
// Cache, 4096: HBM), `waves` waves per block, `blocks` blocks (1 per CU up to 256).  Prints GB/s per CU and TB/s in all.
// The 64x64 ring GEMM at one block per CU moves 16 KiB per K-tile in 0.30 us = 53 GB/s per CU whatever the ring depth and the
// software pipelining (tools/bench_decode_gemms.py): is that a ceiling of the LDS-DMA path, of the CU, or of the kernel?
//   hipcc -O3 --offload-arch=gfx950 tools/probes/probe_cu_ingest.hip -o /tmp/probe_ingest && /tmp/probe_ingest
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

template <int MODE, int DEPTH>
__global__ __launch_bounds__(1024) void k(const char* __restrict__ buf, size_t slice, int iters, unsigned* out) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
  const int nw = blockDim.x >> 6;
  const char* p = buf + ((size_t)blockIdx.x * nw + w) * slice + l * 16;
  const size_t steps = slice / 1024;   // 1 KiB per wave-instruction
  u32x4 acc = {0, 0, 0, 0};
  for (int it = 0; it < iters; ++it) {
    for (size_t s = 0; s + DEPTH <= steps; s += DEPTH) {
      if (MODE == 0) {
#pragma unroll
        for (int d = 0; d < DEPTH; ++d)
          __builtin_amdgcn_global_load_lds((gptr_t)(p + (s + d) * 1024), (lptr_t)(lds + (w * DEPTH + d) * 1024), 16, 0, 0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      } else {
        u32x4 v[DEPTH];
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) v[d] = *(const u32x4*)(p + (s + d) * 1024);
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) acc ^= v[d];
      }
    }
  }
  if (acc[0] == 0x12345678u && acc[1] == 1u) out[0] = acc[2];
}

// mode 2: the access pattern of a K-contiguous GEMM operand: one wave-instruction covers 1024 / SEG rows x SEG contiguous bytes
// (row stride `stride` bytes), consecutive instructions of a wave move along the rows (k direction), 8 instructions in flight.
// G: row groups a wave walks TOGETHER along k (G x RPI rows stay "open" per wave: G = 1 is a wave on 8 rows, G = 4 with 4 waves is the
// 128 rows a 64x64 GEMM block keeps open - does the number of distinct rows (pages) in flight per CU matter?)
template <int SEG, int G>
__global__ __launch_bounds__(1024) void ks(const char* __restrict__ buf, size_t stride, int rows_total, int iters, unsigned* out) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  constexpr int RPI = 1024 / SEG;                 // rows per instruction
  const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
  const int nw = blockDim.x >> 6;
  const int gw = blockIdx.x * nw + w, ngw = gridDim.x * nw;
  const int steps = (int)(stride / SEG);
  for (int it = 0; it < iters; ++it)
    for (int r0 = gw * RPI * G; r0 + RPI * G <= rows_total; r0 += ngw * RPI * G) {
      const char* p = buf + (size_t)(r0 + l / (SEG / 16)) * stride + (l % (SEG / 16)) * 16;
      for (int s = 0; s + 8 / G <= steps; s += 8 / G) {
#pragma unroll
        for (int d = 0; d < 8; ++d)
          __builtin_amdgcn_global_load_lds((gptr_t)(p + (size_t)((d % G) * RPI) * stride + (size_t)(s + d / G) * SEG),
                                           (lptr_t)(lds + (w * 8 + d) * 1024), 16, 0, 0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
    }
  if (out == nullptr) lds[0] = 1;
}

// mode 3: STORES.  Every wave writes its own contiguous slice with 16-byte stores (1 KiB per wave-instruction), `burst` instructions
// back to back, then (optionally) a wait: what can one CU push out, alone and with all 256 CUs writing at once?  (The 256x256 GEMM's
// epilogue writes 128 KiB per CU in ~6 us = 20 GB/s per CU with every CU in its epilogue at the same time.)
__global__ __launch_bounds__(1024) void kst(char* __restrict__ buf, size_t slice, int iters, int burst, int xcd_mask) {
  if (!((xcd_mask >> (blockIdx.x & 7)) & 1)) return;   // blocks are dealt to the 8 XCDs round-robin: only the XCDs of the mask write
  const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
  const int nw = blockDim.x >> 6;
  char* p = buf + ((size_t)blockIdx.x * nw + w) * slice + l * 16;
  const u32x4 v = {1u, 2u, 3u, (unsigned)l};
  const size_t steps = slice / 1024;
  for (int it = 0; it < iters; ++it)
    for (size_t s = 0; s + burst <= steps; s += burst) {
      for (int d = 0; d < burst; ++d) *(u32x4*)(p + (s + d) * 1024) = v;
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
}

void run_store(char* buf, size_t total, int blocks, int waves, int burst, int xcd_mask = 0xff) {
  const size_t slice = (total / ((size_t)blocks * waves)) / (1024 * burst) * (1024 * burst);
  if (slice == 0) return;
  const int iters = (int)(((size_t)4 << 30) / (slice * (size_t)blocks * waves)) + 1;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(kst, dim3(blocks), dim3(64 * waves), 0, 0, buf, slice, 1, burst, xcd_mask);
  hipEventRecord(e0);
  hipLaunchKernelGGL(kst, dim3(blocks), dim3(64 * waves), 0, 0, buf, slice, iters, burst, xcd_mask);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const int active = blocks * __builtin_popcount(xcd_mask & 0xff) / 8;
  const double bytes = (double)slice * active * waves * iters;
  printf("store burst %2d  blocks %3d (XCD mask %02x: %3d active) x %2d waves: %7.1f GB/s per active CU  %6.2f TB/s\n", burst, blocks,
         xcd_mask & 0xff, active, waves, bytes / ms / 1e6 / active, bytes / ms / 1e9);
}

// mode 4: the MEMORY side of the 64x64 ring GEMM without its arithmetic.  A block of 4 waves walks K in tiles of 128 bytes: per
// tile every wave issues 2 LDS-DMA for its 16 rows of the W tile (rows private to the block) and 2 for its 16 rows of the A tile
// (`a_shared`: the SAME 64 rows for every block with the same m tile - the GEMM; 0: private rows), NS - 1 tiles in flight behind a
// counted vmcnt, then (`barrier`) one s_barrier per tile.  7B down projection at M = 128: 112 blocks, K = 18944 (296 tiles).
template <int NS>
__global__ __launch_bounds__(256) void kgemm(const char* __restrict__ W, const char* __restrict__ A, size_t stride, int tiles_m, int nk,
                                             int a_shared, int barrier, int iters, int w_copies, size_t w_bytes) {
  extern __shared__ __attribute__((aligned(16))) char lds[];   // [NS][A 8 KiB | W 8 KiB]
  const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
  const int tn = blockIdx.x / tiles_m, tm = blockIdx.x % tiles_m;
  const char* wp0 = W + (size_t)(tn * 64 + 16 * w + (l >> 3)) * stride + (l & 7) * 16;
  const char* ap = A + (size_t)((a_shared ? tm : blockIdx.x) * 64 + 16 * w + (l >> 3)) * stride + (l & 7) * 16;
  auto stage = [&](int slot, int kt, const char* wp) {
    char* la = lds + slot * 16384 + w * 2048;
    const size_t k = (size_t)(kt < nk ? kt : nk - 1) * 128;
    __builtin_amdgcn_global_load_lds((gptr_t)(ap + k), (lptr_t)la, 16, 0, 0);
    __builtin_amdgcn_global_load_lds((gptr_t)(ap + 8 * stride + k), (lptr_t)(la + 1024), 16, 0, 0);
    __builtin_amdgcn_global_load_lds((gptr_t)(wp + k), (lptr_t)(la + 8192), 16, 0, 0);
    __builtin_amdgcn_global_load_lds((gptr_t)(wp + 8 * stride + k), (lptr_t)(la + 8192 + 1024), 16, 0, 0);
  };
  for (int it = 0; it < iters; ++it) {
    const char* wp = wp0 + (size_t)(it % w_copies) * w_bytes;   // another copy of W every pass: an HBM stream, like the 28 layers of a step
#pragma unroll
    for (int i = 0; i < NS - 1; ++i) stage(i, i, wp);
    for (int kt = 0; kt < nk; ++kt) {
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 * (NS - 2)) : "memory");
      if (barrier) asm volatile("s_barrier" ::: "memory");
      stage((kt + NS - 1) % NS, kt + NS - 1, wp);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
}

template <int NS>
void run_gemm_mem(const char* W, const char* A, int tiles_m, int tiles_n, int K, int a_shared, int barrier) {
  const size_t stride = (size_t)K * 2;
  const int nk = K / 64, blocks = tiles_m * tiles_n, iters = 20;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipFuncSetAttribute((const void*)kgemm<NS>, hipFuncAttributeMaxDynamicSharedMemorySize, NS * 16384);
  hipLaunchKernelGGL((kgemm<NS>), dim3(blocks), dim3(256), NS * 16384, 0, W, A, stride, tiles_m, nk, a_shared, barrier, 1, 8, 3584 * stride);
  hipEventRecord(e0);
  hipLaunchKernelGGL((kgemm<NS>), dim3(blocks), dim3(256), NS * 16384, 0, W, A, stride, tiles_m, nk, a_shared, barrier, iters, 8, 3584 * stride);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double us = ms * 1e3 / iters;
  printf("gemm-mem M=%3d N=%4d K=%5d  ring %d  A %s  %s: %7.1f us per pass = %5.3f us per K-tile, W at %5.2f TB/s\n", tiles_m * 64,
         tiles_n * 64, K, NS, a_shared ? "shared " : "private", barrier ? "barrier   " : "no barrier", us, us / nk,
         (double)tiles_n * 64 * stride / us / 1e6);
}

template <int SEG, int G = 1>
void run_strided(const char* buf, size_t total, size_t stride, int blocks, int waves, unsigned* out) {
  const int rows = (int)(total / stride);
  const int iters = (int)(((size_t)4 << 30) / total) + 1;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int ldsb = waves * 8 * 1024;
  hipFuncSetAttribute((const void*)ks<SEG, G>, hipFuncAttributeMaxDynamicSharedMemorySize, ldsb);
  hipLaunchKernelGGL((ks<SEG, G>), dim3(blocks), dim3(64 * waves), ldsb, 0, buf, stride, rows, 1, out);
  hipEventRecord(e0);
  hipLaunchKernelGGL((ks<SEG, G>), dim3(blocks), dim3(64 * waves), ldsb, 0, buf, stride, rows, iters, out);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double bytes = (double)rows * stride * iters;
  printf("strided seg %4d B  row stride %6zu B  %3d rows open per CU  blocks %3d x %2d waves: %7.1f GB/s per CU  %6.2f TB/s\n", SEG, stride,
         (1024 / SEG) * G * waves, blocks, waves, bytes / ms / 1e6 / blocks, bytes / ms / 1e9);
}

template <int MODE, int DEPTH>
void run(const char* buf, size_t total, int blocks, int waves, unsigned* out, const char* tag) {
  const size_t slice = (total / ((size_t)blocks * waves)) / (1024 * DEPTH) * (1024 * DEPTH);
  if (slice == 0) return;
  const int iters = (int)(((size_t)6 << 30) / (slice * (size_t)blocks * waves)) + 1;   // ~6 GiB moved per launch
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int ldsb = MODE == 0 ? waves * DEPTH * 1024 : 0;
  hipFuncSetAttribute((const void*)k<MODE, DEPTH>, hipFuncAttributeMaxDynamicSharedMemorySize, ldsb);
  hipLaunchKernelGGL((k<MODE, DEPTH>), dim3(blocks), dim3(64 * waves), ldsb, 0, buf, slice, 1, out);
  hipEventRecord(e0);
  hipLaunchKernelGGL((k<MODE, DEPTH>), dim3(blocks), dim3(64 * waves), ldsb, 0, buf, slice, iters, out);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double bytes = (double)slice * blocks * waves * iters;
  printf("%-4s depth %2d  blocks %3d x %2d waves  slice %8zu B: %7.1f GB/s per CU  %6.2f TB/s\n", tag, DEPTH, blocks, waves, slice,
         bytes / ms / 1e6 / blocks, bytes / ms / 1e9);
}

int main(int argc, char** argv) {
  unsigned* out; hipMalloc(&out, 16);
  if (argc > 1 && argv[1][0] == 'w') {   // `probe_ingest write`: store throughput (128 KiB per CU per round = the GEMM's C tile, and a stream)
    char* buf; const size_t total = (size_t)2 << 30;
    if (hipMalloc(&buf, total) != hipSuccess) return 1;
    for (int blocks : {8, 256})
      for (int waves : {4, 8})
        for (int burst : {4, 16}) run_store(buf, total, blocks, waves, burst);
    // is the 5.8 TB/s of all 256 CUs a chip limit (HBM) or eight per-XCD limits?  all 32 CUs of ONE / TWO / FOUR XCDs writing
    for (int mask : {0x01, 0x03, 0x0f, 0xff}) run_store(buf, total, 256, 8, 16, mask);
    return 0;
  }
  if (argc > 1 && argv[1][0] == 'g') {   // `probe_ingest gemm`: the ring GEMM's memory side alone
    char *W, *A; const int K = 18944; const size_t stride = (size_t)K * 2;
    // W: 8 copies of 3584 rows, one per pass in turn (1.09 GB: an HBM stream, not an Infinity-Cache hit); A: up to 256 x 64 private rows
    if (hipMalloc(&W, 8 * 3584 * stride) != hipSuccess || hipMalloc(&A, (size_t)256 * 64 * stride) != hipSuccess) return 1;
    hipMemset(W, 1, 8 * 3584 * stride); hipMemset(A, 1, (size_t)256 * 64 * stride);
    for (int tiles_m : {1, 2, 4})
      for (int a_shared : {1, 0})
        for (int barrier : {1, 0}) {
          run_gemm_mem<4>(W, A, tiles_m, 56, K, a_shared, barrier);
          run_gemm_mem<8>(W, A, tiles_m, 56, K, a_shared, barrier);
        }
    return 0;
  }
  if (argc > 1 && argv[1][0] == 'r') {   // `probe_ingest rows`: how many distinct rows (pages) a CU keeps open
    char* buf; const size_t total = (size_t)2 << 30;
    if (hipMalloc(&buf, total) != hipSuccess) return 1;
    hipMemset(buf, 1, total);
    for (size_t stride : {(size_t)37888, (size_t)7168})
      for (int blocks : {112, 256}) {
        run_strided<128, 1>(buf, total, stride, blocks, 4, out);
        run_strided<128, 2>(buf, total, stride, blocks, 4, out);
        run_strided<128, 4>(buf, total, stride, blocks, 4, out);
        run_strided<128, 8>(buf, total, stride, blocks, 4, out);
        run_strided<128, 8>(buf, total, stride, blocks, 8, out);
      }
    return 0;
  }
  if (argc > 1) {   // strided part only: `probe_ingest strided`
    char* buf; const size_t total = (size_t)2 << 30;
    if (hipMalloc(&buf, total) != hipSuccess) return 1;
    hipMemset(buf, 1, total);
    for (size_t stride : {(size_t)37888, (size_t)7168, (size_t)2560})
      for (int blocks : {112, 256})
        for (int waves : {4, 8}) {
          run_strided<64>(buf, total, stride, blocks, waves, out);
          run_strided<128>(buf, total, stride, blocks, waves, out);
          run_strided<256>(buf, total, stride, blocks, waves, out);
          run_strided<512>(buf, total, stride, blocks, waves, out);
        }
    return 0;
  }
  for (size_t mb : {24ul, 192ul, 4096ul}) {
    char* buf; if (hipMalloc(&buf, mb << 20) != hipSuccess) return 1;
    hipMemset(buf, 1, mb << 20);
    printf("== buffer %zu MiB\n", mb);
    for (int blocks : {8, 256}) {          // 8 blocks: one CU per XCD has the memory system to itself; 256: every CU at once
      for (int waves : {4, 8, 16}) {
        run<0, 4>(buf, mb << 20, blocks, waves, out, "dma");
        run<0, 8>(buf, mb << 20, blocks, waves, out, "dma");
        run<1, 4>(buf, mb << 20, blocks, waves, out, "vgpr");
        run<1, 8>(buf, mb << 20, blocks, waves, out, "vgpr");
        run<1, 16>(buf, mb << 20, blocks, waves, out, "vgpr");
      }
    }
    hipFree(buf);
  }
  return 0;
}
