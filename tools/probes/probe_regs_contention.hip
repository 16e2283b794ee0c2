// Are a wave's registers preserved when several PROCESSES share one GPU?  Each lane parks NREG known values in VGPRs (and an MFMA
// accumulator, which the compiler keeps in AGPRs), waits on a dependent chain of global loads (where a wave sits when the
// scheduler takes the CU away), and checks every value afterwards.  Run two or more instances at once:
//   hipcc -O2 --offload-arch=gfx950 tools/probes/probe_regs_contention.hip -o /tmp/probe_regs
//   /tmp/probe_regs & /tmp/probe_regs & wait
// A non-zero "corrupted" count with 2+ instances and zero with one is a platform (wave save / restore) problem, not a kernel race:
// the kernel has no LDS, no barrier, no cross-lane traffic and every lane only ever reads its own registers.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
constexpr int NREG = 40;

__global__ __launch_bounds__(256) void probe(const int* __restrict__ chain, int mask, int steps, int iters, unsigned* bad,
                                             unsigned* detail) {
  const unsigned id = blockIdx.x * blockDim.x + threadIdx.x;
  for (int it = 0; it < iters; ++it) {
    float r[NREG];
#pragma unroll
    for (int i = 0; i < NREG; ++i) {
      r[i] = (float)((id * 131u + (unsigned)i * 7u + (unsigned)it) & 0xfffffu) * 0.5f + 1.0f;
      asm volatile("" : "+v"(r[i]));
    }
    f32x4 acc = {1.0f + it, 2.0f, 3.0f, 4.0f};
    bf16x8 z = {0, 0, 0, 0, 0, 0, 0, 0};
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(z, z, acc, 0, 0, 0);   // acc + 0: lives in AGPRs across the wait
    int p = (int)((id * 2654435761u + (unsigned)it * 40503u) & (unsigned)mask);
    for (int s = 0; s < steps; ++s) p = chain[p];                         // dependent loads: the wave waits here
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(z, z, acc, 0, 0, 0);
    unsigned nb = 0, first = 0xffffffffu;
#pragma unroll
    for (int i = 0; i < NREG; ++i) {
      asm volatile("" : "+v"(r[i]));
      const float e = (float)((id * 131u + (unsigned)i * 7u + (unsigned)it) & 0xfffffu) * 0.5f + 1.0f;
      if (r[i] != e) { ++nb; if (first == 0xffffffffu) first = (unsigned)i | (__float_as_uint(r[i]) == 0 ? 0x100u : 0); }
    }
    if (acc[0] != 1.0f + it || acc[1] != 2.0f || acc[2] != 3.0f || acc[3] != 4.0f) { ++nb; if (first == 0xffffffffu) first = 0x200u; }
    if (nb) {
      atomicAdd(bad, nb);
      const unsigned slot = atomicAdd(bad + 1, 1u);
      if (slot < 64) { detail[slot * 4] = id; detail[slot * 4 + 1] = first; detail[slot * 4 + 2] = nb; detail[slot * 4 + 3] = (unsigned)it; }
    }
    if (p == -12345) bad[2] = 1;
  }
}

int main(int argc, char** argv) {
  const int blocks = argc > 1 ? atoi(argv[1]) : 4096, steps = argc > 2 ? atoi(argv[2]) : 24, iters = argc > 3 ? atoi(argv[3]) : 40,
            launches = argc > 4 ? atoi(argv[4]) : 60;
  const int n = 1 << 24;
  std::vector<int> h(n);
  unsigned x = 12345;
  for (int i = 0; i < n; ++i) { x = x * 1664525u + 1013904223u; h[i] = (int)((x >> 4) & (n - 1)); }
  int* chain; unsigned *bad, *detail;
  hipMalloc(&chain, n * sizeof(int)); hipMalloc(&bad, 16); hipMalloc(&detail, 64 * 16);
  hipMemcpy(chain, h.data(), n * sizeof(int), hipMemcpyHostToDevice);
  hipMemset(bad, 0, 16); hipMemset(detail, 0xff, 64 * 16);
  for (int l = 0; l < launches; ++l) hipLaunchKernelGGL(probe, dim3(blocks), dim3(256), 0, 0, chain, n - 1, steps, iters, bad, detail);
  hipDeviceSynchronize();
  unsigned hb[4], hd[256];
  hipMemcpy(hb, bad, 16, hipMemcpyDeviceToHost); hipMemcpy(hd, detail, 64 * 16, hipMemcpyDeviceToHost);
  printf("corrupted register values: %u in %u (lane, iteration) events; checks: %.3g\n", hb[0], hb[1],
         (double)blocks * 256 * iters * launches * (NREG + 1));
  for (unsigned i = 0; i < hb[1] && i < 12; ++i)
    printf("  thread %u (lane %u, wave %u of block %u): first bad register %s%u%s, %u bad, iteration %u\n", hd[i * 4], hd[i * 4] & 63,
           (hd[i * 4] & 255) >> 6, hd[i * 4] >> 8, (hd[i * 4 + 1] & 0x200) ? "acc " : "v", hd[i * 4 + 1] & 0xff,
           (hd[i * 4 + 1] & 0x100) ? " (reads 0)" : "", hd[i * 4 + 2], hd[i * 4 + 3]);
  return hb[0] ? 1 : 0;
}
