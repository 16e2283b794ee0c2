// Fused GEMM epilogues shared by the bf16 and the fp8 kernels (gemm_bf16.hip, gemm_fp8.hip): the accumulator layout is
// the dtype-independent 16x16 C/D map of gfx950 with SWAPPED operands (D^T = W . A^T): per 16x16 tile a lane owns row
// fr = lane & 15 and the 4 consecutive columns 4 * (lane >> 4) .. +3.
#pragma once
#include "owc_common.h"

namespace {

// v_rcp_f32 (1 ulp) instead of an IEEE division: the result is rounded to bf16 right after
__device__ __forceinline__ float act_quick_gelu(float x) { return x * __builtin_amdgcn_rcpf(1.0f + __expf(-1.702f * x)); }
__device__ __forceinline__ float act_gelu_erf(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }
__device__ __forceinline__ float act_silu(float x) { return x * __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }

// ---- epilogue ----
// After the MFMAs a lane owns, per 16x16 tile, 4 consecutive columns of one row (8 bytes of bf16).  The
// store tail is instruction-issue bound, so adjacent tile pairs are first exchanged across 16-lane rows
// with v_permlane16_swap (odd rows of the first <-> even rows of the second): every lane then owns 8
// consecutive columns and bias / residual / output move as 16-byte accesses (half the instructions).
// Loads are issued unconditionally from clamped addresses so they batch; only stores are predicated.
__device__ __forceinline__ void swap16(float& x, float& y) {
  const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(y), false, false);
  x = __uint_as_float(r[0]);
  y = __uint_as_float(r[1]);
}

// `ctile` != NULL: instead of going to memory, the finished bf16 values are parked in a block-wide LDS image of the output
// tile (row pitch `cpitch` bytes, 16-byte chunks XOR-swizzled by row & 7, block-local origin (lrow0, lcol0)) and
// store_ctile() writes them out as whole rows.  Why: a lane group of the MFMA layout only covers 64 contiguous bytes of
// a row, and half-line writes measurably slow the whole kernel (ablation: full-line pattern +2..8 %).
// NTP: pairs of 16-column tiles the wave holds in acc[0 .. 2 NTP) (2 = a 64-column wave tile, 1 = a 32-column one)
// MT0 / MT1: only the m tiles [MT0, MT1) of the wave's MT are processed (a caller may run the epilogue in row pieces).
template <int EPI, int MT, bool FULL = false, int NTP = 2, int MT0 = 0, int MT1 = MT>
__device__ __forceinline__ void gemm_epilogue(const f32x4 (&acc)[4][MT], int mrow0, int ncol0, int fr, int fq,
                                              const bf16_t* __restrict__ bias, const bf16_t* R, long ldr,
                                              void* Cv, long ldc, int M, int N, const owc_gemm_aux& aux,
                                              char* ctile = nullptr, int cpitch = 0, int lrow0 = 0, int lcol0 = 0) {
  if constexpr (EPI == OWC_EPI_F32) {
    float* C = (float*)Cv;
#pragma unroll
    for (int mt = MT0; mt < MT1; ++mt) {
      const int m = mrow0 + mt * 16 + fr;
#pragma unroll
      for (int nt = 0; nt < 2 * NTP; ++nt) {
        const int n = ncol0 + nt * 16 + fq * 4;
        f32x4 v = acc[nt][mt];
        if (bias != nullptr && n < N) {
          const bf16x4 b = *(const bf16x4*)(bias + n);
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] += bf2f(b[e]);
        }
        if (FULL || (m < M && n < N)) *(f32x4*)(C + (long)m * ldc + n) = v;
      }
    }
  } else if constexpr (EPI == OWC_EPI_SWIGLU && NTP == 1) {
    // one gate / up tile pair (acc[0], acc[1]): a lane owns 4 consecutive features of its row - 8-byte stores, no lane exchange
    bf16_t* C = (bf16_t*)Cv;
    const int f = (ncol0 >> 1) + fq * 4;
    const int nout = N >> 1;
    float bgv[2][4];
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
      for (int e = 0; e < 4; ++e) bgv[nt][e] = 0.f;
    if (bias != nullptr) {
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) {
        const bf16x4 b = *(const bf16x4*)(bias + min(ncol0 + nt * 16 + fq * 4, N - 4));
#pragma unroll
        for (int e = 0; e < 4; ++e) bgv[nt][e] = bf2f(b[e]);
      }
    }
#pragma unroll
    for (int mt = MT0; mt < MT1; ++mt) {
      const int m = mrow0 + mt * 16 + fr;
      bf16x4 o;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float g0 = acc[0][mt][e], u0 = acc[1][mt][e];
        if (bias != nullptr) {
          g0 += bgv[0][e];
          u0 += bgv[1][e];
        }
        o[e] = f2bf(rbf(rbf(act_silu(rbf(g0))) * rbf(u0)));
      }
      if (FULL || (m < M && f < nout)) *(bf16x4*)(C + (long)m * ldc + f) = o;
    }
  } else if constexpr (EPI == OWC_EPI_SWIGLU) {
    bf16_t* C = (bf16_t*)Cv;
    const int odd = fq & 1;
    const int f = (ncol0 >> 1) + odd * 16 + (fq >> 1) * 8;  // first of this lane's 8 output features
    const int nout = N >> 1;
    // bias (the gated MLP of the Qwen2.5-VL vision blocks, HF qwen2_5_vl:85-96 with bias=True; the decoders have none): rows of
    // the weight are [gate 16 | up 16 | ...], so the bias of accumulator tile nt sits at column ncol0 + 16 nt + 4 fq + e
    float bg[4][4];
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
      if (bias != nullptr) {
        const bf16x4 b = *(const bf16x4*)(bias + min(ncol0 + nt * 16 + fq * 4, N - 4));
#pragma unroll
        for (int e = 0; e < 4; ++e) bg[nt][e] = bf2f(b[e]);
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) bg[nt][e] = 0.f;
      }
    }
#pragma unroll
    for (int mt = MT0; mt < MT1; ++mt) {
      const int m = mrow0 + mt * 16 + fr;
      float o0[4], o1[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float g0 = acc[0][mt][e], u0 = acc[1][mt][e], g1 = acc[2][mt][e], u1 = acc[3][mt][e];
        if (bias != nullptr) {   // (no `+ 0.f` on the bias-free path: the decoders' bits stay exactly what they were)
          g0 += bg[0][e];
          u0 += bg[1][e];
          g1 += bg[2][e];
          u1 += bg[3][e];
        }
        o0[e] = rbf(rbf(act_silu(rbf(g0))) * rbf(u0));
        o1[e] = rbf(rbf(act_silu(rbf(g1))) * rbf(u1));
        swap16(o0[e], o1[e]);
      }
      bf16x8 o;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        o[e] = f2bf(o0[e]);
        o[4 + e] = f2bf(o1[e]);
      }
      if (ctile) {
        const int lr = lrow0 + mt * 16 + fr, lc = (lcol0 >> 1) + odd * 16 + (fq >> 1) * 8;
        *(bf16x8*)(ctile + lr * cpitch + (((lc >> 3) ^ (lr & 7)) << 4)) = o;
      } else if (FULL || (m < M && f < nout)) {
        *(bf16x8*)(C + (long)m * ldc + f) = o;
      }
    }
  } else {
    bf16_t* C = (bf16_t*)Cv;
    const int odd = fq & 1;
    // VROPE: the (h, w) position of each of the lane's MT rows, fetched ONCE (round 6: it was re-read for every column pair, and the
    // table loads of a pair waited behind it - 4 x MT dependent load chains per lane; profiles/r06_gemm_by_shape_timed_region.csv
    // has the vision qkv launch at 1125 TFLOP/s against 1290 for the same shape with a plain epilogue)
    int2 hwv[EPI == OWC_EPI_VROPE ? MT : 1];
    if constexpr (EPI == OWC_EPI_VROPE) {
#pragma unroll
      for (int mt = MT0; mt < MT1; ++mt) hwv[mt] = *(const int2*)(aux.pos_hw + 2 * (long)min(mrow0 + mt * 16 + fr, M - 1));
    }
#pragma unroll
    for (int p = 0; p < NTP; ++p) {
      const int n = ncol0 + (2 * p + odd) * 16 + (fq >> 1) * 8;  // first of this lane's 8 columns
      const int nc = min(n, N - 8);
      float bv[8];
      if (bias != nullptr) {
        const bf16x8 b = *(const bf16x8*)(bias + nc);
#pragma unroll
        for (int e = 0; e < 8; ++e) bv[e] = bf2f(b[e]);
      } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) bv[e] = 0.f;
      }
      // all residual rows of this column block are read before the first store (R may alias C)
      bf16x8 rr[MT];
      if constexpr (EPI == OWC_EPI_RESIDUAL) {
#pragma unroll
        for (int mt = MT0; mt < MT1; ++mt)
          rr[mt] = *(const bf16x8*)(R + (long)min(mrow0 + mt * 16 + fr, M - 1) * ldr + nc);
      }
      // VROPE: the cos / sin quadruples of ALL the m tiles of this column block are requested before the first is used (round 6: one
      // dependent table load per m tile in front of its arithmetic was a chain of L2 round trips, ~4.6 us per 256x256 tile)
      f32x4 c4v[EPI == OWC_EPI_VROPE ? MT : 1], s4v[EPI == OWC_EPI_VROPE ? MT : 1];
      if constexpr (EPI == OWC_EPI_VROPE) {
        if (n < aux.rope_cols) {
          const int quarter = aux.head_dim >> 2;
          const int j0 = (n % aux.head_dim) >> 1;  // multiple of 4, never straddles `quarter`
#pragma unroll
          for (int mt = MT0; mt < MT1; ++mt) {
            const int2 hw = hwv[mt];
            const int ti = (j0 < quarter) ? hw.x * quarter + j0 : hw.y * quarter + (j0 - quarter);
            c4v[mt] = *(const f32x4*)(aux.cos_t + ti);
            s4v[mt] = *(const f32x4*)(aux.sin_t + ti);
          }
        }
      }
#pragma unroll
      for (int mt = MT0; mt < MT1; ++mt) {
        const int m = mrow0 + mt * 16 + fr;
        float v[8];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          v[e] = acc[2 * p][mt][e];
          v[4 + e] = acc[2 * p + 1][mt][e];
          swap16(v[e], v[4 + e]);
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = rbf(v[e] + bv[e]);
        if constexpr (EPI == OWC_EPI_VROPE) {
          // apply_rotary_pos_emb_vision (HF:225-236) on pair-interleaved columns: (2j, 2j+1) hold the
          // original (j, j + head_dim/2); angle j uses the row's h position for j < head_dim/4, else w.
          if (n < aux.rope_cols) {
            const f32x4 c4 = c4v[mt];
            const f32x4 s4 = s4v[mt];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              // Scalar form on purpose, each factor pinned in its own register: the SLP vectoriser otherwise packs the pair into
              // v_pk_mul_f32 ... op_sel:[0,1] (low result = src0.lo * src1.HI), and that instruction form returns a wrong low result in
              // lanes 48-63 on gfx950 while another wave of the SIMD runs MFMAs (tools/probes/probe_load_after_mfma.hip; build.py
              // rejects any object that contains it).
              float x1 = v[2 * e], x2 = v[2 * e + 1], c = c4[e], sn = s4[e];
              asm volatile("" : "+v"(x1), "+v"(x2), "+v"(c), "+v"(sn));
              float t0 = x2 * sn, t1 = x1 * sn;
              asm volatile("" : "+v"(t0), "+v"(t1));
              v[2 * e] = __builtin_fmaf(x1, c, -t0);
              v[2 * e + 1] = __builtin_fmaf(x2, c, t1);
            }
          }
        }
        if constexpr (EPI == OWC_EPI_QUICK_GELU) {
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] = act_quick_gelu(v[e]);
        } else if constexpr (EPI == OWC_EPI_GELU_ERF) {
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] = act_gelu_erf(v[e]);
        } else if constexpr (EPI == OWC_EPI_RESIDUAL) {
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] += bf2f(rr[mt][e]);
        }
        bf16x8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = f2bf(v[e]);
        if (ctile) {
          const int lr = lrow0 + mt * 16 + fr, lc = lcol0 + (2 * p + odd) * 16 + (fq >> 1) * 8;
          *(bf16x8*)(ctile + lr * cpitch + (((lc >> 3) ^ (lr & 7)) << 4)) = o;
        } else if (FULL || (m < M && n < N)) {
          *(bf16x8*)(C + (long)m * ldc + n) = o;
        }
      }
    }
  }
}

// Epilogue WITHOUT the LDS image, for a kernel whose LDS stays busy (the persistent ping-pong GEMM, whose operand ring already carries the
// NEXT output tile's K-tiles while this one is written): full 128-byte lines straight from registers.  After the permlane16 swap a lane
// owns, for tile pair p of the wave's 64 columns, 16 bytes of row fr - four lanes cover 64 bytes of a row, a half line per store
// instruction and row.  One more exchange fixes that: lanes fr and fr ^ 8 of a 16-lane row trade one of their two 16-byte pieces
// (DPP row_ror:8), after which lanes 0-7 hold rows 0-7 / 8-15 of pair 0's columns and lanes 8-15 the same rows of pair 1's: a store
// instruction then writes 8 rows x 128 contiguous bytes.  Arithmetic, rounding points and their order are gemm_epilogue's generic
// branch, value for value.  (64-column wave tiles only; F32 / SwiGLU outputs are not taken.)
__device__ __forceinline__ unsigned dpp_ror8(unsigned x) {
  return (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, 0x128 /* row_ror:8 */, 0xf, 0xf, false);
}

// MH0, MH1: the m tiles [MH0, MH1) of the wave tile (multiples of 4) - the caller may write the tile in pieces
template <int EPI, int MT, bool NT, int MH0 = 0, int MH1 = MT>
__device__ __forceinline__ void gemm_epilogue_lines(const f32x4 (&acc)[4][MT], int mrow0, int ncol0, int fr, int fq,
                                                    const bf16_t* __restrict__ bias, const bf16_t* R, long ldr, void* Cv, long ldc,
                                                    int M, int N, const owc_gemm_aux& aux, int tk = 0) {
  // tk (timing-knob build only): 8 = all the arithmetic, no stores; 16 = stores of the raw accumulator bits, no arithmetic
  static_assert(EPI != OWC_EPI_F32 && EPI != OWC_EPI_SWIGLU, "bf16 outputs of the wave tile's own width only");
  if (OWC_TK(tk & 16)) {
    bf16_t* C = (bf16_t*)Cv;
    const int nst = ncol0 + ((fr < 8) ? 0 : 32) + fq * 8;
#pragma unroll
    for (int mt = MH0; mt < MH1; ++mt) {
      const int ra = mrow0 + mt * 16 + (fr & 7);
      *(u32x4*)(C + (long)ra * ldc + nst) = __builtin_bit_cast(u32x4, acc[0][mt]);
      *(u32x4*)(C + (long)(ra + 8) * ldc + nst) = __builtin_bit_cast(u32x4, acc[1][mt]);
    }
    return;
  }
  bf16_t* C = (bf16_t*)Cv;
  const int odd = fq & 1;
  const bool lo = fr < 8;
  int ncol[2], nclamp[2];
  float bv[2][8];
#pragma unroll
  for (int p = 0; p < 2; ++p) {
    ncol[p] = ncol0 + (2 * p + odd) * 16 + (fq >> 1) * 8;   // first of the lane's 8 columns of pair p (before the row exchange)
    nclamp[p] = min(ncol[p], N - 8);
    if (bias != nullptr) {
      const bf16x8 b = *(const bf16x8*)(bias + nclamp[p]);
#pragma unroll
      for (int e = 0; e < 8; ++e) bv[p][e] = bf2f(b[e]);
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e) bv[p][e] = 0.f;
    }
  }
  const int nst = lo ? ncol[0] : ncol[1];   // the lane's columns in the stores
#pragma unroll
  for (int mh = MH0; mh < MH1; mh += 4) {
    // the residual rows of four m tiles are read in one batch (R may alias C: an element is read before the instruction that stores it)
    bf16x8 rr[2][4];
    if constexpr (EPI == OWC_EPI_RESIDUAL) {
#pragma unroll
      for (int p = 0; p < 2; ++p)
#pragma unroll
        for (int t = 0; t < 4; ++t)
          rr[p][t] = *(const bf16x8*)(R + (long)min(mrow0 + (mh + t) * 16 + fr, M - 1) * ldr + nclamp[p]);
    }
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int mt = mh + t;
      const int m = mrow0 + mt * 16 + fr;
      u32x4 pk[2];
#pragma unroll
      for (int p = 0; p < 2; ++p) {
        float v[8];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          v[e] = acc[2 * p][mt][e];
          v[4 + e] = acc[2 * p + 1][mt][e];
          swap16(v[e], v[4 + e]);
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = rbf(v[e] + bv[p][e]);
        if constexpr (EPI == OWC_EPI_VROPE) {
          if (ncol[p] < aux.rope_cols) {   // (see gemm_epilogue: scalar form, every factor pinned - the packed-multiply form is refused)
            const int quarter = aux.head_dim >> 2;
            const int j0 = (ncol[p] % aux.head_dim) >> 1;
            const int2 hw = *(const int2*)(aux.pos_hw + 2 * (long)min(m, M - 1));
            const int ti = (j0 < quarter) ? hw.x * quarter + j0 : hw.y * quarter + (j0 - quarter);
            const f32x4 c4 = *(const f32x4*)(aux.cos_t + ti);
            const f32x4 s4 = *(const f32x4*)(aux.sin_t + ti);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              float x1 = v[2 * e], x2 = v[2 * e + 1], c = c4[e], sn = s4[e];
              asm volatile("" : "+v"(x1), "+v"(x2), "+v"(c), "+v"(sn));
              float t0 = x2 * sn, t1 = x1 * sn;
              asm volatile("" : "+v"(t0), "+v"(t1));
              v[2 * e] = __builtin_fmaf(x1, c, -t0);
              v[2 * e + 1] = __builtin_fmaf(x2, c, t1);
            }
          }
        }
        if constexpr (EPI == OWC_EPI_QUICK_GELU) {
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] = act_quick_gelu(v[e]);
        } else if constexpr (EPI == OWC_EPI_GELU_ERF) {
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] = act_gelu_erf(v[e]);
        } else if constexpr (EPI == OWC_EPI_RESIDUAL) {
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] += bf2f(rr[p][t][e]);
        }
        bf16x8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = f2bf(v[e]);
        pk[p] = __builtin_bit_cast(u32x4, o);
      }
      // lanes fr < 8 give pair 1's piece of their row and get pair 0's piece of row fr + 8; lanes fr >= 8 the other way round
      u32x4 first, second;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const unsigned got = dpp_ror8(lo ? pk[1][i] : pk[0][i]);
        first[i] = lo ? pk[0][i] : got;     // row mt * 16 + (fr & 7)
        second[i] = lo ? got : pk[1][i];    // row mt * 16 + (fr & 7) + 8
      }
      const int ra = mrow0 + mt * 16 + (fr & 7), rb = ra + 8;
      if (OWC_TK(tk & 8)) {
        if ((first[0] ^ second[1]) == 0x12345678u && first[2] == 0x9abcdef0u) *(u32x4*)(C + (long)ra * ldc + nst) = first;
        continue;
      }
      if (nst < N) {
        if (ra < M) {
          if constexpr (NT) __builtin_nontemporal_store(first, (u32x4*)(C + (long)ra * ldc + nst));
          else *(u32x4*)(C + (long)ra * ldc + nst) = first;
        }
        if (rb < M) {
          if constexpr (NT) __builtin_nontemporal_store(second, (u32x4*)(C + (long)rb * ldc + nst));
          else *(u32x4*)(C + (long)rb * ldc + nst) = second;
        }
      }
    }
  }
}

// Second half of the LDS-staged epilogue: the block's output tile (rows x cols bf16, pitch = cols * 2 bytes) leaves LDS as
// whole rows - a wave-instruction covers 64 lanes x 16 B = 1 KiB of consecutive row segments.
template <int WAVES, bool NT = false>
__device__ __forceinline__ void store_ctile(const char* ctile, int rows, int cols, bf16_t* C, long ldc, int m0, int n0,
                                            int M, int Ncols, int w, int l) {
  const int lpr = cols >> 3;            // lanes per row (16-byte chunks)
  const int rpi = 64 / lpr;             // rows per wave-instruction
  const int rows_per_wave = rows / WAVES;
  const int c = l % lpr;
  const int r_in = l / lpr;
  const int n = n0 + c * 8;
#pragma unroll 4
  for (int it = 0; it < rows_per_wave / rpi; ++it) {
    const int r = w * rows_per_wave + it * rpi + r_in;
    const bf16x8 v = *(const bf16x8*)(ctile + r * (cols * 2) + ((c ^ (r & 7)) << 4));
    if (m0 + r < M && n < Ncols) {
      if constexpr (NT) __builtin_nontemporal_store(v, (bf16x8*)(C + (long)(m0 + r) * ldc + n));
      else *(bf16x8*)(C + (long)(m0 + r) * ldc + n) = v;
    }
  }
}

}  // namespace
