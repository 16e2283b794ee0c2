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
