// Do transcendental ops (v_exp_f32, quarter rate) and plain vector ops (v_fma_f32 / v_pk_fma_f32) of one SIMD overlap on gfx950, or add
// up?  If they overlapped, a softmax could evaluate part of its exponentials as a polynomial on the FMA lanes while v_exp handles
// the rest (the trick FlashAttention-4 plays on Blackwell's MUFU).  Each wave runs `rounds` of [NE independent v_exp_f32] +
// [NF independent v_fma_f32], either back to back or finely interleaved; 1-4 waves per SIMD.
//   hipcc -O3 --offload-arch=gfx950 tools/probes/probe_trans_valu_overlap.hip -o /tmp/probe_tv && /tmp/probe_tv
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int NE, int NF, bool PK>
__global__ __launch_bounds__(256) void k(int rounds, float* out) {
  float e[8], f[8];
  f32x2 p[8];
  for (int i = 0; i < 8; ++i) {
    e[i] = 0.001f * (threadIdx.x + i);
    f[i] = 0.002f * (threadIdx.x + i);
    p[i] = (f32x2){f[i], f[i] + 1.f};
  }
  for (int r = 0; r < rounds; ++r) {
    constexpr int N = NE > NF ? NE : NF;
#pragma unroll
    for (int i = 0; i < N; ++i) {   // interleaved: one exp, then NF / NE fmas (or just one kind)
      if (i < NE) e[i & 7] = __builtin_amdgcn_exp2f(e[i & 7]);
      if (NE == 0 || i * NF / (NE ? NE : 1) < NF) {
        constexpr int per = NE ? (NF + NE - 1) / NE : 1;
#pragma unroll
        for (int j = 0; j < (NE ? per : 1); ++j) {
          const int q = (i * (NE ? per : 1) + j);
          if (q < NF) {
            if (PK) p[q & 7] = p[q & 7] * (f32x2){0.999f, 0.998f} + (f32x2){0.001f, 0.002f};
            else f[q & 7] = __builtin_fmaf(f[q & 7], 0.999f, 0.001f);
          }
        }
      }
    }
  }
  float s = 0;
  for (int i = 0; i < 8; ++i) s += e[i] + f[i] + p[i][0] + p[i][1];
  if (s == 12345.678f) out[0] = s;
}

template <int NE, int NF, bool PK>
float run(int blocks, int rounds, float* out) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((k<NE, NF, PK>), dim3(blocks), dim3(256), 0, 0, rounds, out);
  hipEventRecord(e0);
  hipLaunchKernelGGL((k<NE, NF, PK>), dim3(blocks), dim3(256), 0, 0, rounds, out);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  return ms;
}

int main() {
  float* out; hipMalloc(&out, 4);
  const int rounds = 4000;
  for (int wps : {1, 2, 4}) {
    const int blocks = 256 * wps;
    const float e = run<32, 0, false>(blocks, rounds, out), f = run<0, 128, false>(blocks, rounds, out), ef = run<32, 128, false>(blocks, rounds, out);
    const float pk = run<0, 128, true>(blocks, rounds, out), epk = run<32, 128, true>(blocks, rounds, out);
    printf("%d wave(s) per SIMD: 32 v_exp %.3f ms | 128 v_fma %.3f ms | interleaved %.3f ms (sum %.3f, max %.3f) || 128 v_pk_fma %.3f ms | exp + pk_fma %.3f ms (sum %.3f)\n",
           wps, e, f, ef, e + f, e > f ? e : f, pk, epk, e + pk);
  }
  return 0;
}
