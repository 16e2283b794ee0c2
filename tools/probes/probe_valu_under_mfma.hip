// Do plain vector-ALU results stay correct while OTHER waves of the same SIMD run dense MFMA bursts?  Each wave alternates a burst of
// dependent MFMAs (zero or non-zero operands) with a small packed / scalar fp32 computation whose exact result is known, and counts
// wrong lanes.  No LDS, no barrier, no memory traffic inside the loop.
//   hipcc -O3 --offload-arch=gfx950 tools/probes/probe_valu_under_mfma.hip -o /tmp/probe_valu && /tmp/probe_valu [spin] [nonzero]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;

__global__ __launch_bounds__(256) void probe(int spin, int rounds, int nonzero, unsigned* bad, unsigned* byq) {
  const int l = threadIdx.x & 63;
  const unsigned id = blockIdx.x * blockDim.x + threadIdx.x;
  bf16x8 a, b;
#pragma unroll
  for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(nonzero ? (float)((id + i) % 7) * 0.125f : 0.0f); b[i] = (__bf16)(nonzero ? 0.25f : 0.0f); }
  f32x4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
  unsigned nbad = 0;
  for (int r = 0; r < rounds; ++r) {
    for (int s = 0; s < spin; ++s)
#pragma unroll
      for (int t = 0; t < 4; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[t], 0, 0, 0);
    // the rotation of the RoPE epilogue on exactly representable values: (x1, x2) by (c, s) -> (x1 c - x2 s, x2 c + x1 s)
    float x1 = (float)((id + r) % 13) + 1.0f, x2 = (float)((id * 3 + r) % 11) + 2.0f, c = 0.5f, s_ = 0.25f;
    asm volatile("" : "+v"(x1), "+v"(x2), "+v"(c), "+v"(s_));
    f32x2 xs = {x1, x2}, cc = {c, c}, ss = {s_, s_};
    f32x2 p = {xs[1] * ss[0], xs[0] * ss[1]};                 // packed multiply with a swizzled operand (v_pk_mul_f32 op_sel)
    const float v0 = xs[0] * cc[0] - p[0], v1 = xs[1] * cc[1] + p[1];
    float w0 = v0, w1 = v1;
    asm volatile("" : "+v"(w0), "+v"(w1));
    const float e0 = ((float)((id + r) % 13) + 1.0f) * 0.5f - ((float)((id * 3 + r) % 11) + 2.0f) * 0.25f;
    const float e1 = ((float)((id * 3 + r) % 11) + 2.0f) * 0.5f + ((float)((id + r) % 13) + 1.0f) * 0.25f;
    if (w0 != e0 || w1 != e1) { ++nbad; atomicAdd(byq + (l >> 4), 1u); }
  }
  if (nbad) atomicAdd(bad, nbad);
  if (acc[0][0] + acc[1][1] + acc[2][2] + acc[3][3] == -1.0f) bad[1] = 1;
}

int main(int argc, char** argv) {
  const int spin = argc > 1 ? atoi(argv[1]) : 40, nonzero = argc > 2 ? atoi(argv[2]) : 0, rounds = 200, blocks = 4096, launches = 20;
  unsigned *bad, *byq, hb[2], hq[4];
  hipMalloc(&bad, 8); hipMalloc(&byq, 16); hipMemset(bad, 0, 8); hipMemset(byq, 0, 16);
  for (int i = 0; i < launches; ++i) hipLaunchKernelGGL(probe, dim3(blocks), dim3(256), 0, 0, spin, rounds, nonzero, bad, byq);
  hipDeviceSynchronize();
  hipMemcpy(hb, bad, 8, hipMemcpyDeviceToHost); hipMemcpy(hq, byq, 16, hipMemcpyDeviceToHost);
  printf("spin %d, %s operands: %u wrong results of %.3g; by 16-lane quarter of the wave: %u %u %u %u\n", spin, nonzero ? "non-zero" : "zero", hb[0],
         (double)blocks * 256 * rounds * launches, hq[0], hq[1], hq[2], hq[3]);
  return hb[0] ? 1 : 0;
}
