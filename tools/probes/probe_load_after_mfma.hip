// After a burst of MFMAs: global_load_dwordx4 -> s_waitcnt vmcnt(0) -> use the first loaded register AT ONCE (v_pk_mul_f32, like the
// RoPE epilogue) and again 64+ cycles later.  Counts, per 16-lane quarter of the wave, how often (a) the immediate packed product's LOW
// half is wrong, (b) its HIGH half is wrong, (c) a plain copy taken at once differs from the value in memory, (d) a late copy differs.
//   hipcc -O3 --offload-arch=gfx950 tools/probes/probe_load_after_mfma.hip -o /tmp/probe_load && /tmp/probe_load [spin] [lds bytes]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;

#ifndef FIRST_USE
#define FIRST_USE "v_pk_mul_f32 %1, %0, %4 op_sel:[0,1] op_sel_hi:[0,0]"   // lo = d0 * x2, hi = d0 * x1 (the epilogue's instruction)
#endif
#ifndef WANT_LO
#define WANT_LO (want * 5.0f)
#define WANT_HI (want * 3.0f)
#endif
#ifndef GAP
#define GAP ""      // e.g. "s_nop 0\n\t" between the wait and the first use
#endif
__global__ __launch_bounds__(256) void probe(const float* __restrict__ table, int n4, int spin, int rounds, unsigned* cnt) {
  extern __shared__ char lds[];
  const int l = threadIdx.x & 63;
  const unsigned id = blockIdx.x * blockDim.x + threadIdx.x;
  bf16x8 z = {0, 0, 0, 0, 0, 0, 0, 0};
  f32x4 acc[4] = {{1, 2, 3, 4}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
  for (int r = 0; r < rounds; ++r) {
    for (int s = 0; s < spin; ++s)
#pragma unroll
      for (int t = 0; t < 4; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(z, z, acc[t], 0, 0, 0);
    const unsigned idx = (id * 2654435761u + (unsigned)r * 7919u) % (unsigned)n4;   // table[4 idx + e] = idx + 0.25 e + 1 (exact in fp32)
    const float* p = table + 4 * (size_t)idx;
    float x1 = 3.0f, x2 = 5.0f;
    asm volatile("" : "+v"(x1), "+v"(x2));
    f32x2 xs = {x1, x2};
    f32x2 d, prod, early2, zero2 = {0.0f, 0.0f};
    asm volatile("" : "+v"(zero2));
    asm volatile("global_load_dwordx2 %0, %3, off\n\t"
                 "s_waitcnt vmcnt(0)\n\t" GAP FIRST_USE "\n\t"
                 "v_pk_add_f32 %2, %0, %5"
                 : "=&v"(d), "=&v"(prod), "=&v"(early2) : "v"(p), "v"(xs), "v"(zero2) : "memory");
    asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15" ::: "memory");
    float late = d[0], early = early2[0];
    asm volatile("" : "+v"(late));
    const float want = (float)idx + 1.0f;
    const int q = l >> 4;
    if (prod[0] != WANT_LO) atomicAdd(cnt + q, 1u);
    if (prod[1] != WANT_HI) atomicAdd(cnt + 4 + q, 1u);
    if (early != want) atomicAdd(cnt + 8 + q, 1u);
    if (late != want) atomicAdd(cnt + 12 + q, 1u);
  }
  if (acc[0][0] + acc[1][1] + acc[2][2] + acc[3][3] == -1.0f) cnt[16] = 1;
}

int main(int argc, char** argv) {
  const int spin = argc > 1 ? atoi(argv[1]) : 40, lds = argc > 2 ? atoi(argv[2]) : 0, rounds = 50, blocks = 4096, launches = 40, n4 = 1 << 20;
  std::vector<float> h((size_t)n4 * 4);
  for (int i = 0; i < n4; ++i)
    for (int e = 0; e < 4; ++e) h[(size_t)i * 4 + e] = (float)i + 0.25f * e + 1.0f;
  float* t; unsigned *cnt, hc[17];
  hipMalloc(&t, h.size() * 4); hipMalloc(&cnt, 68); hipMemset(cnt, 0, 68);
  hipMemcpy(t, h.data(), h.size() * 4, hipMemcpyHostToDevice);
  if (lds) hipFuncSetAttribute((const void*)probe, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  for (int i = 0; i < launches; ++i) hipLaunchKernelGGL(probe, dim3(blocks), dim3(256), lds, 0, t, n4, spin, rounds, cnt);
  hipDeviceSynchronize();
  hipMemcpy(hc, cnt, 68, hipMemcpyDeviceToHost);
  printf("spin %d lds %d: of %.3g loads, wrong by lane quarter 0..3 | packed product LOW %u %u %u %u | HIGH %u %u %u %u | copy at once %u %u %u %u | copy 64+ cycles later %u %u %u %u\n",
         spin, lds, (double)blocks * 256 * rounds * launches, hc[0], hc[1], hc[2], hc[3], hc[4], hc[5], hc[6], hc[7], hc[8], hc[9], hc[10], hc[11],
         hc[12], hc[13], hc[14], hc[15]);
  return 0;
}
