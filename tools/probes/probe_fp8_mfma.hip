// Operand-layout probe for v_mfma_scale_f32_16x16x128_f8f6f4 with e4m3 operands and unit block scales.
// Hypothesis under test (exact small-integer data, asymmetric A and B):
//   lane l holds A[row = l & 15][k = 32 * (l >> 4) + j], j = 0..31, in its 8 operand VGPRs (byte j of the 32);
//   B likewise with col = l & 15;  D: col = l & 15, row = 4 * (l >> 4) + reg  (the dtype-independent C/D map).
// Build: hipcc --offload-arch=gfx950 -O2 tools/probes/probe_fp8_mfma.hip -o gpurun_out/probe_fp8
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v4f __attribute__((ext_vector_type(4)));

static uint8_t enc_e4m3(int v) {  // exact for |v| <= 16
  if (v == 0) return 0;
  uint8_t s = v < 0 ? 0x80 : 0;
  int a = abs(v), e = 0;
  while ((a >> (e + 1)) != 0) ++e;               // floor(log2 a)
  int mant = ((a << 3) >> e) & 7;                // 3 fraction bits (exact for small a)
  return s | (uint8_t)((e + 7) << 3) | (uint8_t)mant;
}

__global__ void probe(const uint8_t* A, const uint8_t* B, float* D, int scale) {
  const int l = threadIdx.x;
  v8i a, b;
  const int* ap = (const int*)(A + (l & 15) * 128 + 32 * (l >> 4));
  const int* bp = (const int*)(B + (l & 15) * 128 + 32 * (l >> 4));
  for (int i = 0; i < 8; ++i) {
    a[i] = ap[i];
    b[i] = bp[i];
  }
  v4f c = {0.f, 0.f, 0.f, 0.f};
  c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c, 0, 0, 0, scale, 0, scale);
  for (int r = 0; r < 4; ++r) D[(4 * (l >> 4) + r) * 16 + (l & 15)] = c[r];
}

int main() {
  std::vector<uint8_t> A(16 * 128), B(16 * 128);
  std::vector<int> Ai(16 * 128), Bi(16 * 128);
  srand(7);
  for (int i = 0; i < 16 * 128; ++i) {
    Ai[i] = rand() % 9 - 4;
    Bi[i] = rand() % 7 - 3;
    A[i] = enc_e4m3(Ai[i]);
    B[i] = enc_e4m3(Bi[i]);
  }
  uint8_t *dA, *dB;
  float* dD;
  hipMalloc(&dA, A.size());
  hipMalloc(&dB, B.size());
  hipMalloc(&dD, 256 * 4);
  hipMemcpy(dA, A.data(), A.size(), hipMemcpyHostToDevice);
  hipMemcpy(dB, B.data(), B.size(), hipMemcpyHostToDevice);
  for (int scale : {0x7F7F7F7F, 0x7F, (int)0x80808080u}) {
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, dA, dB, dD, scale);
    std::vector<float> D(256);
    hipMemcpy(D.data(), dD, 1024, hipMemcpyDeviceToHost);
    int bad = 0, badT = 0;
    double ratio = 0;
    for (int i = 0; i < 16; ++i)
      for (int j = 0; j < 16; ++j) {
        long ref = 0;
        for (int k = 0; k < 128; ++k) ref += (long)Ai[i * 128 + k] * Bi[j * 128 + k];
        // our use: D = mfma(a = W frag (cols), b = A frag (rows)) -> test both orientations
        if (D[i * 16 + j] != (float)ref) ++bad;
        if (D[j * 16 + i] != (float)ref) ++badT;
        if (ref != 0) ratio = D[i * 16 + j] / (double)ref;
      }
    printf("scale=0x%08x  mismatches: D[row=a-row][col=b-col] %d, transposed %d   (sample D/ref = %g)\n", scale, bad, badT, ratio);
  }
  return 0;
}
