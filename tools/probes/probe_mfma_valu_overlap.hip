// Do MFMA work and plain vector-ALU work of CO-RESIDENT waves overlap on a gfx950 SIMD, or add up?  Each wave runs `rounds` of
// [nm dependent-free MFMAs (4 independent accumulators)] + [nv independent v_fma_f32 / v_exp_f32]; with several waves per SIMD the
// phases of different waves interleave freely.  Compare the time of (MFMA only), (VALU only) and (both): both ~ max => they overlap,
// both ~ sum => one issue stream.
//   hipcc -O3 --offload-arch=gfx950 tools/probes/probe_mfma_valu_overlap.hip -o /tmp/probe_ov && /tmp/probe_ov
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

template <int NM, int NV, bool EXP>
__global__ __launch_bounds__(256) void k(int rounds, float* out) {
  bf16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(0.001f * (threadIdx.x % 7)); b[i] = (__bf16)0.5f; }
  f32x4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
  float v[8];
  for (int i = 0; i < 8; ++i) v[i] = 0.001f * (threadIdx.x + i);
  for (int r = 0; r < rounds; ++r) {
#pragma unroll
    for (int i = 0; i < NM; ++i) acc[i & 3] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[i & 3], 0, 0, 0);
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      if (EXP) v[i & 7] = __builtin_amdgcn_exp2f(v[i & 7]) - 1.0f;
      else v[i & 7] = __builtin_fmaf(v[i & 7], 0.999f, 0.001f);
    }
  }
  float s = 0;
  for (int i = 0; i < 8; ++i) s += v[i];
  for (int i = 0; i < 4; ++i) s += acc[i][0];
  if (s == 12345.678f) out[0] = s;
}

template <int NM, int NV, bool EXP>
float run(int blocks, int rounds, float* out) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((k<NM, NV, EXP>), dim3(blocks), dim3(256), 0, 0, rounds, out);
  hipEventRecord(e0);
  hipLaunchKernelGGL((k<NM, NV, EXP>), dim3(blocks), dim3(256), 0, 0, rounds, out);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  return ms;
}

int main() {
  float* out; hipMalloc(&out, 4);
  const int rounds = 2000;
  for (int wps : {1, 2, 3, 4}) {   // waves per SIMD = blocks per CU (a block is 4 waves, one per SIMD)
    const int blocks = 256 * wps;
    const float m = run<44, 0, false>(blocks, rounds, out), f = run<0, 176, false>(blocks, rounds, out), mf = run<44, 176, false>(blocks, rounds, out);
    const float e = run<0, 32, true>(blocks, rounds, out), me = run<44, 32, true>(blocks, rounds, out);
    printf("%d wave(s) per SIMD: 44 MFMA %.2f ms | 176 v_fma %.2f ms | both %.2f ms (sum %.2f, max %.2f) || 32 v_exp %.2f ms | MFMA + exp %.2f ms (sum %.2f)\n", wps, m, f, mf,
           m + f, m > f ? m : f, e, me, m + e);
  }
  return 0;
}
