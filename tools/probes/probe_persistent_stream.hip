// What would a PERSISTENT decode-step kernel buy at the reference's batch size?  A 7B decode step at 1-8 rows is five dependent
// launches per layer (qkv 33 MB, attention, o 26 MB, gate/up 272 MB, down 136 MB of weights) and every launch costs a fixed
// ~5 us (dispatch + first-byte latency) during which HBM idles: 3.44 ms per step where the weight stream alone is 2.4 ms.
// This probe moves the same bytes in the same phase structure with NO arithmetic, three ways:
//   mode 0: one launch per phase (256 blocks x 4 waves, each wave streams its slice through a private LDS ring; the attention
//           phase is a 4-block launch that chases 6 dependent loads), like the product today;
//   mode 1: ONE resident grid, a grid barrier between phases (atomic counter, bounded spin), the next phase's stream starts
//           after the barrier;
//   mode 2: as 1, but a wave's ring runs ahead ACROSS the barrier: the next phase's first RING KiB per wave are in flight
//           while the grid waits (weights do not depend on activations).
// After every barrier / at the start of every launch each wave first reads a line another CU wrote in the previous phase
// (the activations), as the real kernels must.
//   hipcc -O3 --offload-arch=gfx950 tools/probes/probe_persistent_stream.hip -o /tmp/probe_persist && /tmp/probe_persist
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

constexpr int NPH = 5;
struct Plan {
  size_t off[NPH];     // byte offset of the phase's region inside one layer's weights
  size_t per_wave[NPH];  // bytes each of the 1024 waves streams in this phase (multiple of 1 KiB); 0 = the latency (attention) phase
  size_t layer_bytes;
  int layers;
};

__device__ __forceinline__ void chase(const unsigned* __restrict__ ring, unsigned& idx) {
#pragma unroll 1
  for (int h = 0; h < 6; ++h) idx = ring[idx];
}

template <int RING>
__device__ __forceinline__ void stream_slice(const char* p, size_t bytes, char* slot0, int l, u32x4& acc) {
  const int n = (int)(bytes >> 10);
  for (int i = 0; i < RING && i < n; ++i) __builtin_amdgcn_global_load_lds((gptr_t)(p + (size_t)i * 1024 + l * 16), (lptr_t)(slot0 + (i % RING) * 1024), 16, 0, 0);
  for (int i = 0; i < n; ++i) {
    if (i + RING <= n) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(RING - 1) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    acc ^= *(const u32x4*)(slot0 + (i % RING) * 1024 + l * 16);
    if (i + RING < n) __builtin_amdgcn_global_load_lds((gptr_t)(p + (size_t)(i + RING) * 1024 + l * 16), (lptr_t)(slot0 + (i % RING) * 1024), 16, 0, 0);
  }
}

// mode 0: one phase per launch
template <int RING>
__global__ __launch_bounds__(256) void phase_kernel(const char* __restrict__ wbase, size_t per_wave, unsigned* act, unsigned* out) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
  const int gw = blockIdx.x * 4 + w;
  u32x4 acc = {0, 0, 0, 0};
  acc[0] = __hip_atomic_load(act + ((blockIdx.x * 37 + 11) & 255) * 32, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // the activations another CU wrote
  stream_slice<RING>(wbase + (size_t)gw * per_wave + (acc[0] & 0), per_wave, lds + w * RING * 1024, l, acc);
  if (threadIdx.x == 0) act[blockIdx.x * 32] = acc[1] | 1u;
  if (acc[0] == 0x12345678u && acc[1] == 1u) out[0] = acc[2];
}
__global__ void chase_kernel(const unsigned* ring, unsigned* act, unsigned* out) {
  unsigned idx = act[blockIdx.x * 32] & 1023u;
  chase(ring, idx);
  if (threadIdx.x == 0) act[blockIdx.x * 32] = idx | 1u;
}

// FENCED: release / acquire at agent scope (the compiler adds the L2 write-back + invalidate a coarse-grained buffer needs to be
// seen by another XCD).  !FENCED: relaxed counter; the data that crosses the barrier is itself moved with agent-scope
// (cache-bypassing) loads / stores and the stores are waited for before the counter moves.
// (relaxed flavour, two levels: the 32 blocks of an XCD - block b runs on XCD b % 8 - count on their own line, the last of them
// counts on the global line all blocks poll: 32 + 8 serialised atomics instead of 256.)
template <bool FENCED>
__device__ __forceinline__ bool grid_barrier(unsigned* cnt, unsigned target, unsigned* fail) {
  __syncthreads();
  if (threadIdx.x == 0) {
    if (FENCED) __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    else {
      const unsigned per = gridDim.x / 8;
      const unsigned old = __hip_atomic_fetch_add(cnt + 32 * (1 + (blockIdx.x & 7)), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if ((old + 1) % per == 0) __hip_atomic_fetch_add(cnt, per, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    long spins = 0;
    while ((FENCED ? __hip_atomic_load(cnt, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT)
                   : __hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) < target) {
      __builtin_amdgcn_s_sleep(1);
      if (++spins > 20000000L) { *fail = 1; break; }   // bounded: a lost block must not hang the GPU
    }
  }
  __syncthreads();
  return true;
}

// modes 1 / 2: the whole step in one resident grid
template <int RING, bool AHEAD, bool FENCED>
__global__ __launch_bounds__(256) void step_kernel(const char* __restrict__ wbase, Plan plan, const unsigned* __restrict__ ring, unsigned* act,
                                                   unsigned* cnt, unsigned* fail, unsigned* out) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
  const int gw = blockIdx.x * 4 + w;
  char* slot0 = lds + w * RING * 1024;
  u32x4 acc = {0, 0, 0, 0};
  const int nph = plan.layers * NPH;
  unsigned bar = 0;
  auto base_of = [&](int ph) {
    const int L = ph / NPH, q = ph % NPH;
    return wbase + (size_t)L * plan.layer_bytes + plan.off[q] + (size_t)gw * plan.per_wave[q] + l * 16;
  };
  // a wave's ring: RING slots; `primed` pieces of the CURRENT phase are already in flight when the phase starts (AHEAD)
  int primed = 0, sb = 0;   // sb: ring slot of the current phase's piece 0
  for (int ph = 0; ph < nph; ++ph) {
    const int q = ph % NPH;
    const int np = (int)(plan.per_wave[q] >> 10);
    if (ph > 0) {
      grid_barrier<FENCED>(cnt, ++bar * gridDim.x, fail);
      acc[0] ^= __hip_atomic_load(act + ((blockIdx.x * 37 + 11) & 255) * 32, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (np == 0) {   // attention: 4 blocks chase, the rest go straight to the barrier
      if (blockIdx.x < 4 && threadIdx.x == 0) {
        unsigned idx = acc[0] & 1023u;
        chase(ring, idx);
        acc[1] ^= idx;
      }
    } else {
      const char* p = base_of(ph);
      for (int i = primed; i < RING && i < np; ++i) __builtin_amdgcn_global_load_lds((gptr_t)(p + (size_t)i * 1024), (lptr_t)(slot0 + ((sb + i) % RING) * 1024), 16, 0, 0);
      primed = 0;
      // the next streaming phase (its ring is primed under this phase's tail when AHEAD)
      int nx = ph + 1;
      while (nx < nph && plan.per_wave[nx % NPH] == 0) ++nx;
      const char* pn = nx < nph ? base_of(nx) : p;
      const int npn = nx < nph ? (int)(plan.per_wave[nx % NPH] >> 10) : 0;
      for (int i = 0; i < np; ++i) {
        // RING - 1 newer pieces stay in flight throughout: past the end of this phase they are the next phase's (AHEAD) - or
        // nothing, and the wait drains
        if (AHEAD || i + RING <= np) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(RING - 1) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        char* slot = slot0 + ((sb + i) % RING) * 1024;
        acc ^= *(const u32x4*)(slot + l * 16);
        const int j = i + RING;
        if (j < np) __builtin_amdgcn_global_load_lds((gptr_t)(p + (size_t)j * 1024), (lptr_t)slot, 16, 0, 0);
        else if (AHEAD) {   // always issue, so that the counted wait above stays exact (past the very end: a re-read nobody uses)
          const int jn = min(j - np, max(npn - 1, 0));
          __builtin_amdgcn_global_load_lds((gptr_t)(pn + (size_t)jn * 1024), (lptr_t)slot, 16, 0, 0);
        }
      }
      if (AHEAD) primed = min(RING, npn);
      sb = (sb + np) % RING;   // next-phase piece k went to slot (sb + np + k) % RING
    }
    if (threadIdx.x == 0) __hip_atomic_store(act + blockIdx.x * 32, acc[1] | 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (acc[0] == 0x12345678u && acc[1] == 1u) out[0] = acc[2];
}

int main(int argc, char** argv) {
  const int steps = argc > 1 ? atoi(argv[1]) : 3;
  // Qwen2-VL-7B decoder layer: qkv [4608, 3584], o [3584, 3584], gate/up [37888, 3584], down [3584, 18944] bf16
  const size_t sz[NPH] = {(size_t)4608 * 3584 * 2, 0, (size_t)3584 * 3584 * 2, (size_t)37888 * 3584 * 2, (size_t)3584 * 18944 * 2};
  Plan plan;
  plan.layers = 28;
  size_t off = 0;
  for (int q = 0; q < NPH; ++q) {
    plan.off[q] = off;
    plan.per_wave[q] = sz[q] ? ((sz[q] / 1024 / 1024) < 32 ? 32 : (sz[q] / 1024 / 1024)) * 1024 : 0;   // bytes / 1024 waves, whole KiB, at least one ring (the o projection: 24 -> 32 KiB)
    off += plan.per_wave[q] * 1024;
  }
  plan.layer_bytes = off;
  const size_t total = plan.layer_bytes * plan.layers;
  printf("per layer %.1f MB, per step %.2f GB (lm_head not included)\n", plan.layer_bytes / 1e6, total / 1e9);
  char* w;
  CK(hipMalloc(&w, total));
  CK(hipMemset(w, 1, total));
  unsigned *ring, *act, *cnt, *fail, *out;
  CK(hipMalloc(&ring, 1024 * 4 * 64));
  std::vector<unsigned> h(1024 * 64);
  for (int i = 0; i < 1024 * 64; ++i) h[i] = ((i / 64 * 389 + 7) % 1024) * 64 % (1024 * 64);   // hops land in different lines
  CK(hipMemcpy(ring, h.data(), h.size() * 4, hipMemcpyHostToDevice));
  CK(hipMalloc(&act, 256 * 32 * 4));
  CK(hipMemset(act, 0, 256 * 32 * 4));
  CK(hipMalloc(&cnt, 4 * 32 * 9));
  CK(hipMalloc(&fail, 4));
  CK(hipMalloc(&out, 4));
  CK(hipMemset(fail, 0, 4));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  constexpr int RING = 32;
  const int lds = 4 * RING * 1024;
  CK(hipFuncSetAttribute((const void*)phase_kernel<RING>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
  const void* kern[4] = {(const void*)step_kernel<RING, false, true>, (const void*)step_kernel<RING, true, true>,
                         (const void*)step_kernel<RING, false, false>, (const void*)step_kernel<RING, true, false>};
  for (int i = 0; i < 4; ++i) CK(hipFuncSetAttribute(kern[i], hipFuncAttributeMaxDynamicSharedMemorySize, lds));
  int nb = 0;
  CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, kern[1], 256, lds));
  printf("resident blocks per CU of the step kernel: %d (need >= 1 x 256 CUs)\n", nb);
  // per phase, 28 launches back to back (each on another layer's weights): what one launch of the phase costs by itself
  for (int q = 0; q < NPH; ++q) {
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
      CK(hipEventRecord(e0));
      for (int L = 0; L < plan.layers; ++L) {
        if (plan.per_wave[q] == 0) hipLaunchKernelGGL(chase_kernel, dim3(4), dim3(64), 0, 0, ring, act, out);
        else hipLaunchKernelGGL(phase_kernel<RING>, dim3(256), dim3(256), lds, 0, w + (size_t)L * plan.layer_bytes + plan.off[q], plan.per_wave[q], act, out);
      }
      CK(hipEventRecord(e1));
      CK(hipEventSynchronize(e1));
      float ms;
      CK(hipEventElapsedTime(&ms, e0, e1));
      if (rep > 0 && ms < best) best = ms;
    }
    printf("phase %d alone: %6.1f MB  %6.1f us per launch  %.2f TB/s\n", q, plan.per_wave[q] * 1024 / 1e6, best * 1e3 / plan.layers,
           plan.per_wave[q] * 1024.0 * plan.layers / best / 1e9);
  }
  const bool empty = argc > 2;   // any second argument: no bytes at all, the phases are barriers only
  Plan run = plan;
  if (empty) for (int q = 0; q < NPH; ++q) run.per_wave[q] = 0;
  for (int mode = 0; mode < 5; ++mode) {
    if (empty && mode == 0) continue;
    float best = 1e30f;
    for (int rep = 0; rep < steps + 1; ++rep) {
      CK(hipMemset(cnt, 0, 4 * 32 * 9));
      CK(hipEventRecord(e0));
      if (mode == 0) {
        for (int L = 0; L < plan.layers; ++L)
          for (int q = 0; q < NPH; ++q) {
            if (plan.per_wave[q] == 0) hipLaunchKernelGGL(chase_kernel, dim3(4), dim3(64), 0, 0, ring, act, out);
            else hipLaunchKernelGGL(phase_kernel<RING>, dim3(256), dim3(256), lds, 0, w + (size_t)L * plan.layer_bytes + plan.off[q], plan.per_wave[q], act, out);
          }
      } else {
        void* args[] = {(void*)&w, (void*)&run, (void*)&ring, (void*)&act, (void*)&cnt, (void*)&fail, (void*)&out};
        CK(hipLaunchCooperativeKernel(kern[mode - 1], dim3(256), dim3(256), args, lds, 0));
      }
      CK(hipEventRecord(e1));
      CK(hipEventSynchronize(e1));
      float ms;
      CK(hipEventElapsedTime(&ms, e0, e1));
      if (rep > 0 && ms < best) best = ms;
    }
    unsigned f = 0;
    CK(hipMemcpy(&f, fail, 4, hipMemcpyDeviceToHost));
    const char* names[5] = {"one launch per phase                          ", "resident grid, fenced barrier per phase       ",
                            "resident grid, fenced barrier, ring runs ahead", "resident grid, 2-level relaxed barrier         ",
                            "resident grid, 2-level relaxed, ring runs ahead"};
    printf("mode %d %s: %.3f ms per step = %.2f TB/s of weights, %.1f us per layer%s\n", mode, names[mode], best, total / best / 1e9,
           best * 1e3 / plan.layers, f ? "  [a barrier TIMED OUT]" : "");
  }
  return 0;
}
