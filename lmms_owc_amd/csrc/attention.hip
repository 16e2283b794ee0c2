// Flash-style attention forward on the MFMA pipe, for both attention shapes of the Qwen2-VL path:
//   * vision tower: non-causal, per-image (cu_seqlens) attention, 16 heads x head_dim 80
//     (HF:models/qwen2_vl/modeling_qwen2_vl.py:356-422, eager reference :317-339)
//   * decoder prefill: causal GQA attention, head_dim 128 (HF: :508-556)
//
// CDNA4 design (one 256-thread block = 4 waves = 128 query rows of one (sequence, head)):
//  * S^T = K . Q^T with v_mfma_f32_16x16x32_bf16 (keys on the MFMA rows, queries on the lanes), so a
//    lane owns ONE query column: the running max / rescale factor of the online softmax is lane
//    local and a row reduction is 15 VALU max/add + 2 cross-lane steps (xor 16, 32).
//  * O^T = V^T . P^T: the S^T accumulator registers, converted pairwise to bf16, ARE the B operand of
//    the second product (key order permuted identically on the V side), so P never touches LDS.
//  * V^T fragments come from row-major V tiles through ds_read_b64_tr_b16 (hardware transpose).
//  * K/V tiles (64 keys) are staged by 16-byte LDS-DMA into double-buffered LDS; each wave computes
//    two 16-query tiles against every K/V fragment it reads (halves LDS bytes per FLOP).
//  * LDS images: head_dim 80 -> natural 160-B rows (conflict-free for both read kinds);
//    head_dim 128 -> 256-B rows with chunk ^= (key & 7) << 1 applied on the DMA source and on both reads.
//  * softmax in fp32 with exp2; masked/ragged keys handled on the last tile only.
//  * round 5 (the loop was vector-ALU bound: 37 % matrix pipe, 56 % VALU, and the two do not overlap on this SIMD): the
//    per-score vector work is down to fma + exp2 + add + half a pack - p = exp2(s * c - m_ref * c) with c = scale * log2(e), as
//    before, but m_ref is a LAZY reference maximum: the row's running maximum as of the last time it was updated.  A tile
//    computes its exponentials straight away (no row maximum, no cross-lane exchange, no rescale); only when a lane's sum of 16
//    fresh exponentials exceeds 2^10 - some score of the row beat m_ref by more than 6 in log2 units - does the wave enter the
//    slow path, where every such row (the decision is per ROW: OR over its four lanes, a function of the row's own scores, so
//    a row's bits do not depend on which rows share its wave) takes the tile's true maximum, rescales O and l and recomputes
//    its exponentials; rows that did not ask recompute the same bits.  P is a bf16 FLOATING-point operand and O / l are fp32,
//    so a reference up to 2^10 below the true maximum costs no precision (values <= 2^10 instead of <= 1; tools/attn_accuracy.py:
//    error against a float64 attention unchanged).  The first tile always takes the slow path (m_ref starts at 0).
//    Measured and NOT kept: Q pre-multiplied by c in bf16 with the score chains started from -m_ref (the accumulators are then
//    the exp2 arguments: another 8 VALU per 16 scores gone, +5 % over this form) - the extra rounding of Q costs 2-10 x the
//    error at logit std 3-12 (rms 0.0018 -> 0.004-0.006 of rms |O|, profiles/r05_attn_lazy_max_ab3.txt).
#include "owc_internal.h"

namespace {

int g_attn_dbg = 0;  // timing experiment (owc_tuning_set "attn_dbg", -DOWC_TIMING_KNOBS build only): 1 = no K/V DMA in the loop

int g_decode_nbuf1_min_blocks = 256;  // fused decode attention: launches with more blocks than this use one V buffer per wave (knob "decode_attn_nbuf1")
int g_attn_gqa_pack = 1;       // causal launches with kv_group > 1: (position, head) rows of a kv group packed into the blocks (knob "attn_gqa_pack", 0 = one head per block)
int g_attn_class_prefill = 0;  // set by owc_llm_prefill around its non-causal last-token launch (profile class only)

constexpr int QB = 128;  // query rows per block
constexpr int KB = 64;   // keys per tile

template <int HD>
struct Cfg {
  static constexpr int CPR = HD / 8;              // 16-byte chunks per row
  static constexpr int ROWB = HD * 2;             // bytes per row
  static constexpr int TILE = KB * ROWB;          // bytes per K (or V) tile
  static constexpr int NP = TILE / 1024;          // LDS-DMA pieces per tile
  static constexpr int KS = (HD + 31) / 32;       // 32-wide k-steps of Q.K^T
  static constexpr int DT = HD / 16;              // 16-wide d tiles of O
  static constexpr bool SWZ = (HD % 128) == 0;    // 256-B rows need the XOR swizzle
};

// max over the four 16-lane groups of a wave (lanes l, l^16, l^32, l^48), result in every lane: two register swaps
// (v_permlane16_swap / v_permlane32_swap, VALU) instead of two ds_bpermute round trips through the LDS
__device__ __forceinline__ float max_over_groups(float x) {
  const unsigned u = __float_as_uint(x);
  const auto a = __builtin_amdgcn_permlane16_swap(u, u, false, false);  // {r0,r0,r2,r2}, {r1,r1,r3,r3}
  const float m = fmaxf(__uint_as_float(a[0]), __uint_as_float(a[1]));
  const unsigned v = __float_as_uint(m);
  const auto b = __builtin_amdgcn_permlane32_swap(v, v, false, false);  // {lo,lo}, {hi,hi}
  return fmaxf(__uint_as_float(b[0]), __uint_as_float(b[1]));
}

template <int HD, bool CAUSAL>
__global__ __launch_bounds__(256, 2) void attn_fwd_kernel(
    const bf16_t* __restrict__ Q, long q_ts, long q_hs, const bf16_t* __restrict__ K, long k_ts,
    long k_hs, const bf16_t* __restrict__ V, long v_ts, long v_hs, bf16_t* __restrict__ O, long o_ts,
    long o_hs, const int* __restrict__ q_start, const int* __restrict__ o_start,
    const int* __restrict__ k_start, const int* __restrict__ seq_len, const int* __restrict__ q_len, int n_heads, int kv_group, int nqb,
    int n_pairs, float scale_log2e, int dbg, int qrows, int pack) {
  // `qrows` (<= QB, a multiple of 32): query rows per block.  128 except for causal launches, where the rows of the longest
  // sequence are spread EVENLY over its blocks (272 prompt rows: 3 x 96 instead of 128 + 128 + 16 - the block of 16 rows streamed
  // five key tiles for one wave); a row's result does not depend on the block or wave that holds it (same key tiles, same order).
  using C = Cfg<HD>;
  extern __shared__ __attribute__((aligned(16))) char lds[];  // [2 buf][K tile | V tile]
  const int tid = threadIdx.x;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l = tid & 63;
  const int fr = l & 15, g = l >> 4;

  // block -> (pair = sequence*head, q block); the q blocks of one pair share an XCD when possible
  const int bid = blockIdx.x;
  int qb, pair;
  if ((n_pairs & 7) == 0) {
    qb = (bid >> 3) % nqb;
    pair = (bid / (8 * nqb)) * 8 + (bid & 7);
  } else {
    qb = bid % nqb;
    pair = bid / nqb;
  }
  // `pack` (round 5; causal GQA launches: pack = kv_group = G > 1): a pair is (sequence, KV HEAD) and the block's rows are the G x Lq
  // (position, head) pairs of the group in position-major order - packed row R = position * G + head.  The G heads of a group
  // read the same K / V rows, so a 128-row block then holds ~18 positions x 7 heads instead of 96-128 positions of one head: it
  // needs the key tiles of 18 positions (S = 286: 46 block-tiles per sequence and kv head instead of 70, 16 blocks instead of 21,
  // and all four waves of a block own rows).  A row's tiles, their order and its arithmetic are unchanged: the same bits.
  const int G = pack > 0 ? pack : 1;
  const int b = pack > 0 ? pair / (n_heads / G) : pair / n_heads;
  const int h0 = pack > 0 ? (pair % (n_heads / G)) * G : pair % n_heads;   // the pair's (first) head
  const int hk = h0 / kv_group;
  const int L = seq_len[b];                  // keys
  const int Lq = q_len ? q_len[b] : L;       // query positions (== L except for the decode mapping)
  const int nrows = G * Lq;                  // rows of the pair
  if (qb * qrows >= nrows) return;
  const bool active = w * 32 < qrows && (qb * qrows + w * 32) < nrows;  // wave-uniform: waves without rows only stage tiles
  const long qs = q_start[b], ks0 = k_start[b];
  const long os = o_start ? (long)o_start[b] : qs;

  // causal: query position i sits at absolute position coff + i (coff = L - Lq > 0 when a shared prefix's keys
  // precede the rows handled here) and sees keys <= coff + i
  const int coff = L - Lq;
  const int blk_last_pos = min(Lq - 1, (qb * qrows + qrows - 1) / G);        // last position of the block's rows
  const int wave_first_pos = (qb * qrows + w * 32) / G, wave_last_pos = (qb * qrows + w * 32 + 31) / G;
  const int kmax = CAUSAL ? min(L, blk_last_pos + 1 + coff) : L;
  const int ntiles = (kmax + KB - 1) / KB;

  // ---- Q fragments (B operand): lane = query column fr, 8 consecutive d per k-step ----
  bf16x8 qf[2][C::KS];
  int qrow[2], qpos[2], qh[2];   // packed row, its position, its head
#pragma unroll
  for (int qt = 0; qt < 2; ++qt) {
    qrow[qt] = qb * qrows + w * 32 + qt * 16 + fr;
    qpos[qt] = qrow[qt] / G;
    qh[qt] = h0 + (qrow[qt] - qpos[qt] * G);
    const bf16_t* qp = Q + (qs + min(qpos[qt], Lq - 1)) * q_ts + (long)qh[qt] * q_hs;
#pragma unroll
    for (int ks = 0; ks < C::KS; ++ks) {
      const int c = ks * 4 + g;
      if (c < C::CPR) {
        qf[qt][ks] = *(const bf16x8*)(qp + c * 8);
      } else {
        qf[qt][ks] = (bf16x8){0, 0, 0, 0, 0, 0, 0, 0};
      }
    }
  }

  const bf16_t* Kb = K + (long)hk * k_hs;
  const bf16_t* Vb = V + (long)hk * v_hs;

  // per-piece DMA source offsets (bytes from the sequence's first K/V row), hoisted out of the tile loop.
  // piece p = w + 4i: K pieces first, then V pieces (whether piece i is a V piece is a compile-time fact when NP % 4 == 0)
  constexpr int NPW = (2 * C::NP + 3) / 4;  // pieces per wave
  constexpr bool STATIC_KV = (C::NP % 4) == 0;
  const char* Kseq = (const char*)(Kb + ks0 * k_ts);
  const char* Vseq = (const char*)(Vb + ks0 * v_ts);
  auto piece_isv = [&](int i) { return STATIC_KV ? (4 * i >= C::NP) : (w + 4 * i >= C::NP); };
  auto piece_key = [&](int i) {  // tile row this lane's chunk of piece i belongs to
    const int pp = w + 4 * i - (piece_isv(i) ? C::NP : 0);
    return (pp * 64 + l) / C::CPR;
  };
  unsigned poff[NPW];
#pragma unroll
  for (int i = 0; i < NPW; ++i) {
    const int pp = w + 4 * i - (piece_isv(i) ? C::NP : 0);
    const int ci = pp * 64 + l;
    const int key = ci / C::CPR;
    const int pos = ci - key * C::CPR;
    const int c = C::SWZ ? (pos ^ ((key & 7) << 1)) : pos;
    poff[i] = (unsigned)((long)key * (piece_isv(i) ? v_ts : k_ts) * 2 + c * 16);
  }

  // a full tile costs no vector ALU work beyond a pointer bump: uniform base (sequence + tile) + the hoisted 32-bit lane
  // offset; only the ragged last tile clamps rows per lane (kept a separate branch so the selects stay out of the common path)
  auto stage = [&](int buf, int t) {
    char* base = lds + buf * (2 * C::TILE);
    if ((t * KB + KB) <= L) {
      // the tile bases are pinned in SGPRs (opaque to the loop-strength-reduction pass, which otherwise keeps one 64-bit
      // VGPR pointer per piece): the DMA then uses the scalar-base + 32-bit lane-offset form
      const char* kt0 = Kseq + (long)t * KB * k_ts * 2;
      const char* vt0 = Vseq + (long)t * KB * v_ts * 2;
      asm volatile("" : "+s"(kt0), "+s"(vt0));
#pragma unroll
      for (int i = 0; i < NPW; ++i)
        if (w + 4 * i < 2 * C::NP) {
          const bool isv = piece_isv(i);
          glds16((isv ? vt0 : kt0) + poff[i], base + (isv ? C::TILE - C::NP * 1024 : 0) + (w + 4 * i) * 1024);
        }
    } else {
      asm volatile("" ::: "memory");
#pragma unroll
      for (int i = 0; i < NPW; ++i)
        if (w + 4 * i < 2 * C::NP) {
          const bool isv = piece_isv(i);
          const long ts2 = (isv ? v_ts : k_ts) * 2;
          const int key = piece_key(i);
          const char* src = (isv ? Vseq : Kseq) + (long)min(t * KB + key, L - 1) * ts2 + (poff[i] - (unsigned)((long)key * ts2));
          glds16(src, base + (isv ? C::TILE - C::NP * 1024 : 0) + (w + 4 * i) * 1024);
        }
    }
  };

  f32x4 o[C::DT][2];
#pragma unroll
  for (int d = 0; d < C::DT; ++d) {
    o[d][0] = (f32x4){0.f, 0.f, 0.f, 0.f};
    o[d][1] = (f32x4){0.f, 0.f, 0.f, 0.f};
  }
  float nmc[2] = {0.f, 0.f};       // -m_ref * c of the lane's two query rows (log2 units), uniform over a row's four lanes
  float lrun[2] = {0.f, 0.f};
  constexpr float LAZY_BIG = 1024.f;   // a lane's 16 exponentials sum to <= 16 while no score exceeds m_ref

  // fragment byte offsets inside a tile
  // K (A operand of S^T): row key = 16*kt + fr, chunk ks*4+g
  // V (A operand of O^T via transposed read): rows 32s+4g+q' (and +16), q' = fr>>2, p = fr&3
  const int kswz = C::SWZ ? ((fr & 7) << 1) : 0;
  const int vq = fr >> 2, vp = fr & 3;

  stage(0, 0);
  __syncthreads();

  for (int t = 0; t < ntiles; ++t) {
    const int cur = t & 1;
    if (t + 1 < ntiles && !OWC_TK(dbg & 1)) stage(cur ^ 1, t + 1);
    const char* kt_ = lds + cur * (2 * C::TILE);
    const char* vt_ = kt_ + C::TILE;
    // causal: a tile whose first key lies beyond the LAST row of this wave is masked for all 32 rows - it would add exp2(-inf) = 0
    // to every sum and leave the running maximum alone, so skipping it changes no bit (the wave still stages and meets the
    // barriers).  S = 286 prompts in 128-row blocks: 29 instead of 37 wave-tiles per prompt and head.
    if (active && (!CAUSAL || t * KB <= wave_last_pos + coff)) {

    // ---- S^T = K . Q^T ----
    f32x4 s[4][2];
#pragma unroll
    for (int kt = 0; kt < 4; ++kt) {
      s[kt][0] = (f32x4){0.f, 0.f, 0.f, 0.f};
      s[kt][1] = (f32x4){0.f, 0.f, 0.f, 0.f};
      const char* krow = kt_ + (kt * 16 + fr) * C::ROWB;
#pragma unroll
      for (int ks = 0; ks < C::KS; ++ks) {
        const int c = ks * 4 + g;
        bf16x8 kf;
        if (C::CPR % 4 == 0 || ks < C::KS - 1) {
          kf = *(const bf16x8*)(krow + ((c ^ kswz) << 4));
        } else {
          // head_dim 80: chunks 10, 11 of the last k-step do not exist; the Q fragment is zero there, so any finite K data (the
          // row's last real chunk) contributes exactly 0.  (Round 5 measured ONE v_mfma_f32_16x16x16_bf16 on dims 64-79 for that
          // half step instead: the same bits and the same time - the 16-deep MFMA takes as long as the 32-deep one on gfx950.)
          const int cc = min(c, C::CPR - 1);
          kf = *(const bf16x8*)(krow + ((cc ^ kswz) << 4));
        }
        if (OWC_TK(dbg & 8)) {   // (timing build: no QK^T MFMAs)
          s[kt][0][0] += bf2f(kf[0]);
          continue;
        }
        s[kt][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qf[0][ks], s[kt][0], 0, 0, 0);
        s[kt][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qf[1][ks], s[kt][1], 0, 0, 0);
      }
    }

    // ---- online softmax (lane = query column; keys 16kt + 4g + r) ----
    // ragged / causal masking is needed on the last tiles only: keep it a real (wave-uniform) branch -- written
    // as selects it costs ~70 VALU per tile on every tile, and the loop is VALU-bound
    const bool edge = (t * KB + KB > L) || (CAUSAL && (t * KB + KB - 1 > wave_first_pos + coff));
    if (edge) {
      asm volatile("" ::: "memory");  // keeps the compiler from if-converting the block into selects
#pragma unroll
      for (int qt = 0; qt < 2; ++qt)
#pragma unroll
        for (int kt = 0; kt < 4; ++kt)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int key = t * KB + kt * 16 + g * 4 + r;
            if (key >= L || (CAUSAL && key > qpos[qt] + coff)) s[kt][qt][r] = -1e30f;
          }
    }
    // per 16-query tile: fast path p = exp2(s c - m_ref c), one add per value for the row sum; slow path (wave-uniform branch:
    // the first tile, and whenever some row of the wave outgrew its reference maximum) as described on top
    bf16x8 pf[2][2];
#pragma unroll
    for (int qt = 0; qt < 2; ++qt) {
      float x[4][4];
      float s0 = 0.f, s1 = 0.f;   // two chains: half the dependent-add latency
      const f32x2 sc2 = (f32x2){scale_log2e, scale_log2e};
      f32x2 ng2 = (f32x2){nmc[qt], nmc[qt]};
      asm volatile("" : "+v"(ng2));   // a real register pair holding the value twice: the packed fma must not pick src2's high half for its low
                                      // result (`op_sel:[0,0,1]`, the form build.py refuses)
#pragma unroll
      for (int kt = 0; kt < 4; ++kt)
#pragma unroll
        for (int r = 0; r < 4; r += 2) {
          const f32x2 a = (f32x2){s[kt][qt][r], s[kt][qt][r + 1]} * sc2 + ng2;  // v_pk_fma_f32
          const f32x2 p = OWC_TK(dbg & 2) ? a : (f32x2){__builtin_amdgcn_exp2f(a[0]), __builtin_amdgcn_exp2f(a[1])};   // (timing build: no exp)
          x[kt][r] = p[0];
          x[kt][r + 1] = p[1];
          s0 += p[0];
          s1 += p[1];
        }
      // (the two halves are pinned in registers of their own before they are added: the SLP vectoriser keeps (s0, s1) as one packed
      // pair, and its horizontal add is `v_pk_add_f32 ... op_sel:[0,1]`, the form build.py refuses - gemm_epilogue.h has the story)
      asm volatile("" : "+v"(s0), "+v"(s1));
      float sum = s0 + s1;
      if (t == 0 || __builtin_amdgcn_ballot_w64(sum > LAZY_BIG) != 0) {
        asm volatile("" ::: "memory");
        float mx = -1e30f;
#pragma unroll
        for (int kt = 0; kt < 4; ++kt)
#pragma unroll
          for (int r = 0; r < 4; ++r) mx = fmaxf(mx, s[kt][qt][r]);
        mx = max_over_groups(mx);                                          // the tile's row maximum (raw score units)
        const float ask = max_over_groups(sum > LAZY_BIG ? 1.f : 0.f);   // the ROW asks (any of its four lanes)
        const float over = __builtin_fmaf(mx, scale_log2e, nmc[qt]);       // by how much it exceeds m_ref, in log2 units
        const float d = (t == 0) ? over : (ask > 0.f ? over : 0.f);        // rows that did not ask keep m_ref: same bits below
        if (t != 0) {
          const float alpha = __builtin_amdgcn_exp2f(-d);
          lrun[qt] *= alpha;
#pragma unroll
          for (int dd = 0; dd < C::DT; ++dd) {
            o[dd][qt][0] *= alpha;
            o[dd][qt][1] *= alpha;
            o[dd][qt][2] *= alpha;
            o[dd][qt][3] *= alpha;
          }
        }
        nmc[qt] -= d;
        ng2 = (f32x2){nmc[qt], nmc[qt]};
        asm volatile("" : "+v"(ng2));
        s0 = 0.f;
        s1 = 0.f;
#pragma unroll
        for (int kt = 0; kt < 4; ++kt)
#pragma unroll
          for (int r = 0; r < 4; r += 2) {
            const f32x2 a = (f32x2){s[kt][qt][r], s[kt][qt][r + 1]} * sc2 + ng2;
            const float pa = __builtin_amdgcn_exp2f(a[0]), pb = __builtin_amdgcn_exp2f(a[1]);
            x[kt][r] = pa;
            x[kt][r + 1] = pb;
            s0 += pa;
            s1 += pb;
          }
        asm volatile("" : "+v"(s0), "+v"(s1));
        sum = s0 + s1;
      }
      lrun[qt] += sum;
#pragma unroll
      for (int sx = 0; sx < 2; ++sx) {
        bf16x8 f;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          f[r] = f2bf(x[2 * sx][r]);
          f[4 + r] = f2bf(x[2 * sx + 1][r]);
        }
        pf[qt][sx] = f;
      }
    }

    // ---- O^T += V^T . P^T ----
#pragma unroll
    for (int sx = 0; sx < 2; ++sx) {
      const int key0 = sx * 32 + g * 4 + vq;
      const int key1 = key0 + 16;
#pragma unroll
      for (int d = 0; d < C::DT; ++d) {
        int a0, a1;
        if (C::SWZ) {
          const int ch = 2 * d + (vp >> 1);
          a0 = key0 * C::ROWB + ((ch ^ ((key0 & 7) << 1)) << 4) + (vp & 1) * 8;
          a1 = key1 * C::ROWB + ((ch ^ ((key1 & 7) << 1)) << 4) + (vp & 1) * 8;
        } else {
          a0 = key0 * C::ROWB + d * 32 + vp * 8;
          a1 = key1 * C::ROWB + d * 32 + vp * 8;
        }
        const bf16x4 v0 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)(vt_ + a0));
        const bf16x4 v1 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)(vt_ + a1));
        const bf16x8 vf = (bf16x8){v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
        if (OWC_TK(dbg & 4)) {   // (timing build: no P.V MFMAs)
          o[d][0][0] += bf2f(vf[0]) + bf2f(pf[0][sx][0]) + bf2f(pf[1][sx][0]);
          continue;
        }
        o[d][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, pf[0][sx], o[d][0], 0, 0, 0);
        o[d][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, pf[1][sx], o[d][1], 0, 0, 0);
      }
    }
    }  // active
    if (!OWC_TK(dbg & 16)) __syncthreads();   // (timing build, bit 16: no block barrier per tile - racy, timing only)
  }

  // ---- epilogue: O[q][16d + 4g + r] = o[d][qt][r] / l ----
#pragma unroll
  for (int qt = 0; qt < 2; ++qt) {
    float lsum = lrun[qt];
    lsum += __shfl_xor(lsum, 16, 64);
    lsum += __shfl_xor(lsum, 32, 64);
    const float inv = 1.0f / lsum;
    if (active && qrow[qt] < nrows) {   // (a wave beyond `qrows` holds rows of the NEXT block: it must not write them)
      bf16_t* op = O + (os + qpos[qt]) * o_ts + (long)qh[qt] * o_hs + g * 4;
#pragma unroll
      for (int d = 0; d < C::DT; ++d) {
        bf16x4 ov;
#pragma unroll
        for (int r = 0; r < 4; ++r) ov[r] = f2bf(o[d][qt][r] * inv);
        *(bf16x4*)(op + d * 16) = ov;
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------------------------------
// Round 6: the vision towers' attention (non-causal, head_dim 80 / 64) on v_mfma_f32_32x32x16_bf16.
//
// Why another MFMA shape.  Round 5's counters on attn_fwd_kernel<80>: matrix pipe busy 37 % of the cycles, vector ALU 56 % - and the
// two ADD UP: a 16x16x32 MFMA occupies its SIMD's matrix pipe for 16 cycles and holds the SIMD's vector issue for 8 of them
// (MI355X_MICROARCH.md, "vector-instruction ISSUE cost"), so at best half of an MFMA's time can carry other instructions.  The
// 32x32x16 MFMA does the same work per cycle (32 cycles for 4 x the output) but holds the vector issue for 8 of its 32: three
// quarters of the matrix time are open to the softmax's vector work and the LDS reads.  Per wave and 64-key tile (32 query rows,
// head_dim 80): matrix time 704 cycles either way (QK^T 10 x 32 + PV 12 x 32 against 24 x 16 + 20 x 16), open issue cycles 528
// against 352, for ~620 cycles of exp2 / fma / add / cvt and ~130 of LDS issue.  head_dim 80 = 5 k-steps of 16: no padded half step
// in QK^T (the 32-deep MFMA ran 3 steps for 2.5); PV pays instead (3 d blocks of 32 for 2.5).
//
// Layout (one 256-thread block = 4 waves = 128 query rows of one (image, head); wave = 32 rows; lane = (q = l & 31, hi = l >> 5)):
//  * S^T = K . Q^T in two 32-key blocks: A = K rows (lane: S^T row i = l & 31 -> key pi(i) of the block, dims 16 ks + 8 hi ..),
//    B = Q (registers, loaded once).  The accumulator of lane (q, hi) holds S^T rows i = (r & 3) + 8 (r >> 2) + 4 hi, r = 0..15: one
//    QUERY per lane, 16 of the block's 32 keys; the other 16 live in lane l ^ 32 (a row maximum is one permlane32 swap).
//  * P never leaves the registers: registers 8h .. 8h + 7 of key block kb, rounded to bf16, ARE the B operand of PV k-step (kb, h)
//    (16 keys each); the A operand V^T takes the same keys in the same slots through two transposing LDS reads (4 keys each).
//  * pi: WHICH key a S^T row stands for is free (it permutes S^T rows and P slots consistently); it is chosen - with a per-row
//    rotation of the 16-byte chunks of the LDS image - so that both the K row reads (ds_read_b128, 32 rows at one chunk) and the
//    V transposing reads (4 keys x 64 bytes per half wave) are bank-conflict free on natural 160-byte / 128-byte rows
//    (exhaustive search over bit permutations of i and rotations against the bank model of MI355X_MICROARCH.md "LDS"):
//      head_dim 80:  pi(i) = bits (i0 -> 1, i1 -> 2, i2 -> 0, i3 -> 3, i4 -> 4), chunk' = (chunk + ((row >> 3) & 1)) mod 10
//      head_dim 64:  pi(i) = bits (i0 -> 0, i1 -> 3, i2 -> 1, i3 -> 2, i4 -> 4), chunk' = (chunk + ((row >> 1) & 7)) mod 8
//  * online softmax with the LAZY reference maximum of attn_fwd_kernel (same rule, same threshold; a row is two lanes here).
// Results differ from attn_fwd_kernel in the last bits only (another summation order inside the MFMAs); accuracy against a float64
// attention: tests/test_attention_accuracy_gpu.py (same bounds, all kernels).  NOT the default: see g_attn_mfma32 for what it measured.
// ------------------------------------------------------------------------------------------------------------------------
int g_attn_mfma32 = 0;   // 0 (default): attn_fwd_kernel; 1: attn_fwd32_kernel; 2: attn_fwd32p_kernel (knob "attn_mfma32").  MEASURED, one box, interleaved
                         // (profiles/r06_attn_mfma32_ab.txt): 64 x 1024 patches 781 / 683 / 694 TFLOP/s, 16 x 4096: 937 / 830 / 845 - the
                         // 32x32x16 kernels 10-12 % SLOWER at 2 waves per SIMD (170 / 214 VGPRs), the software pipelining worth +1.5 %.
                         // With attn_fwd32_kernel squeezed to 168 VGPRs = 3 waves per SIMD like attn_fwd_kernel (161): 795 / 765 and
                         // 961 / 935 - 3-4 % slower.  Resident waves are what this loop runs on (PMC, round 5: a wave issues 30 % of its
                         // cycles, three of them keep the SIMD's issue port ~90 % busy), not the matrix / vector overlap inside one
                         // wave.  They stay in the tree as the tested alternative, not on the path.

template <int HD>
struct Cfg32 {
  static constexpr int CPR = HD / 8;
  static constexpr int ROWB = HD * 2;
  static constexpr int TILE = KB * ROWB;
  static constexpr int NP = TILE / 1024;
  static constexpr int KS16 = HD / 16;          // 16-deep k-steps of QK^T
  static constexpr int DB = (HD + 31) / 32;     // 32-wide d blocks of O (head_dim 80: the third one half used)
  __device__ static __forceinline__ int pi(int i) {   // S^T row of a 32-key block -> key of the block
    if constexpr (HD == 80) return ((i & 1) << 1) | ((i & 2) << 1) | ((i & 4) >> 2) | (i & 24);
    else return (i & 1) | ((i & 2) << 2) | ((i & 4) >> 1) | ((i & 8) >> 1) | (i & 16);
  }
  __device__ static __forceinline__ int rot(int row) {   // chunk rotation of a tile row
    if constexpr (HD == 80) return (row >> 3) & 1;
    else return (row >> 1) & 7;
  }
  __device__ static __forceinline__ int chunk(int row, int c) {   // position of source chunk c in the LDS row
    const int x = c + rot(row);
    return x >= CPR ? x - CPR : x;
  }
};

template <int HD>
__global__ __launch_bounds__(256, 3) void attn_fwd32_kernel(
    const bf16_t* __restrict__ Q, long q_ts, long q_hs, const bf16_t* __restrict__ K, long k_ts,
    long k_hs, const bf16_t* __restrict__ V, long v_ts, long v_hs, bf16_t* __restrict__ O, long o_ts,
    long o_hs, const int* __restrict__ q_start, const int* __restrict__ o_start,
    const int* __restrict__ k_start, const int* __restrict__ seq_len, int n_heads, int kv_group, int nqb,
    int n_pairs, float scale_log2e) {
  using C = Cfg32<HD>;
  static_assert(HD == 80 || HD == 64, "layouts searched for head_dim 80 and 64");
  extern __shared__ __attribute__((aligned(16))) char lds[];  // [2 buf][K tile | V tile]
  const int tid = threadIdx.x;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l = tid & 63;
  const int ql = l & 31, hi = l >> 5;

  const int bid = blockIdx.x;
  int qb, pair;
  if ((n_pairs & 7) == 0) {
    qb = (bid >> 3) % nqb;
    pair = (bid / (8 * nqb)) * 8 + (bid & 7);
  } else {
    qb = bid % nqb;
    pair = bid / nqb;
  }
  const int b = pair / n_heads, h0 = pair % n_heads;
  const int hk = h0 / kv_group;
  const int L = seq_len[b];
  if (qb * QB >= L) return;
  const bool active = (qb * QB + w * 32) < L;   // wave-uniform: waves without rows only stage tiles
  const long qs = q_start[b], ks0 = k_start[b];
  const long os = o_start ? (long)o_start[b] : qs;
  const int ntiles = (L + KB - 1) / KB;

  // ---- Q fragments (B operand): lane = query column ql, dims 16 ks + 8 hi .. + 7
  const int qrow = qb * QB + w * 32 + ql;
  bf16x8 qf[C::KS16];
  {
    const bf16_t* qp = Q + (qs + min(qrow, L - 1)) * q_ts + (long)h0 * q_hs;
#pragma unroll
    for (int ks = 0; ks < C::KS16; ++ks) qf[ks] = *(const bf16x8*)(qp + ks * 16 + hi * 8);
  }

  // ---- staging: piece p = w + 4 i of a tile's 2 NP pieces (K first, then V); LDS position (row, c') <- source chunk c' - rot(row)
  const bf16_t* Kb = K + (long)hk * k_hs;
  const bf16_t* Vb = V + (long)hk * v_hs;
  constexpr int NPW = (2 * C::NP + 3) / 4;
  constexpr bool STATIC_KV = (C::NP % 4) == 0;
  const char* Kseq = (const char*)(Kb + ks0 * k_ts);
  const char* Vseq = (const char*)(Vb + ks0 * v_ts);
  auto piece_isv = [&](int i) { return STATIC_KV ? (4 * i >= C::NP) : (w + 4 * i >= C::NP); };
  auto piece_row = [&](int i) {
    const int pp = w + 4 * i - (piece_isv(i) ? C::NP : 0);
    return (pp * 64 + l) / C::CPR;
  };
  unsigned poff[NPW];
#pragma unroll
  for (int i = 0; i < NPW; ++i) {
    const int pp = w + 4 * i - (piece_isv(i) ? C::NP : 0);
    const int ci = pp * 64 + l;
    const int row = ci / C::CPR;
    const int cp = ci - row * C::CPR;                 // position in the LDS row
    int c = cp - C::rot(row);                         // the source chunk that belongs there
    c = c < 0 ? c + C::CPR : c;
    poff[i] = (unsigned)((long)row * (piece_isv(i) ? v_ts : k_ts) * 2 + c * 16);
  }
  auto stage = [&](int buf, int t) {
    char* base = lds + buf * (2 * C::TILE);
    if ((t * KB + KB) <= L) {
      const char* kt0 = Kseq + (long)t * KB * k_ts * 2;
      const char* vt0 = Vseq + (long)t * KB * v_ts * 2;
      asm volatile("" : "+s"(kt0), "+s"(vt0));
#pragma unroll
      for (int i = 0; i < NPW; ++i)
        if (w + 4 * i < 2 * C::NP) {
          const bool isv = piece_isv(i);
          glds16((isv ? vt0 : kt0) + poff[i], base + (isv ? C::TILE - C::NP * 1024 : 0) + (w + 4 * i) * 1024);
        }
    } else {   // the ragged last tile: rows beyond the sequence re-read its last row (their scores are masked below)
      asm volatile("" ::: "memory");
#pragma unroll
      for (int i = 0; i < NPW; ++i)
        if (w + 4 * i < 2 * C::NP) {
          const bool isv = piece_isv(i);
          const long ts2 = (isv ? v_ts : k_ts) * 2;
          const int row = piece_row(i);
          const char* src = (isv ? Vseq : Kseq) + (long)min(t * KB + row, L - 1) * ts2 + (poff[i] - (unsigned)((long)row * ts2));
          glds16(src, base + (isv ? C::TILE - C::NP * 1024 : 0) + (w + 4 * i) * 1024);
        }
    }
  };

  f32x16 o[C::DB];
#pragma unroll
  for (int d = 0; d < C::DB; ++d)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[d][r] = 0.f;
  float nmc = 0.f, lrun = 0.f;   // -m_ref * c of the lane's query row (uniform over the row's two lanes), its running sum
  constexpr float LAZY_BIG = 1024.f;

  // fragment addresses inside a tile (bytes): K row reads and the two V transposing reads of a k-step
  const int krow_b = C::pi(ql);                                   // key of a block this lane's S^T row stands for
  int koff[C::KS16];                                              // K: row krow_b, source chunk 2 ks + hi
#pragma unroll
  for (int ks = 0; ks < C::KS16; ++ks) koff[ks] = krow_b * C::ROWB + C::chunk(krow_b, 2 * ks + hi) * 16;
  // V: lane (16-lane group gp = l >> 4, its row qq = (l & 15) >> 2, column quad p4 = l & 3) supplies the address of key
  //    pi(qq + 4 hi + 8 rd + 16 h) of block kb, columns 32 db + 16 (gp & 1) + 4 p4 ..
  const int vq = (l & 15) >> 2, vp4 = l & 3, vg = (l >> 4) & 1;

  stage(0, 0);
  __syncthreads();

  for (int t = 0; t < ntiles; ++t) {
    const int cur = t & 1;
    if (t + 1 < ntiles) stage(cur ^ 1, t + 1);
    const char* kt_ = lds + cur * (2 * C::TILE);
    const char* vt_ = kt_ + C::TILE;
    if (active) {
      // ---- S^T = K . Q^T, two 32-key blocks
      f32x16 s[2];
#pragma unroll
      for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
        for (int r = 0; r < 16; ++r) s[kb][r] = 0.f;
#pragma unroll
        for (int ks = 0; ks < C::KS16; ++ks) {
          const bf16x8 kf = *(const bf16x8*)(kt_ + kb * 32 * C::ROWB + koff[ks]);
          s[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[ks], s[kb], 0, 0, 0);
        }
      }
      // ---- ragged last tile: keys beyond the sequence (a real, wave-uniform branch)
      if (t * KB + KB > L) {
        asm volatile("" ::: "memory");
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int key = t * KB + kb * 32 + C::pi((r & 3) + 8 * (r >> 2) + 4 * hi);
            if (key >= L) s[kb][r] = -1e30f;
          }
      }
      // ---- online softmax, lazy reference maximum (attn_fwd_kernel's rule; the row's other half is lane l ^ 32)
      float x[2][16];
      float s0 = 0.f, s1 = 0.f;
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int r = 0; r < 16; r += 2) {
          const float pa = __builtin_amdgcn_exp2f(__builtin_fmaf(s[kb][r], scale_log2e, nmc));
          const float pb = __builtin_amdgcn_exp2f(__builtin_fmaf(s[kb][r + 1], scale_log2e, nmc));
          x[kb][r] = pa;
          x[kb][r + 1] = pb;
          s0 += pa;
          s1 += pb;
        }
      asm volatile("" : "+v"(s0), "+v"(s1));
      float sum = s0 + s1;
      if (t == 0 || __builtin_amdgcn_ballot_w64(sum > LAZY_BIG) != 0) {
        asm volatile("" ::: "memory");
        float mx = -1e30f;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
          for (int r = 0; r < 16; ++r) mx = fmaxf(mx, s[kb][r]);
        {   // the row's maximum / whether the row asks: over its two lanes
          const unsigned u = __float_as_uint(mx);
          const auto sw = __builtin_amdgcn_permlane32_swap(u, u, false, false);
          mx = fmaxf(__uint_as_float(sw[0]), __uint_as_float(sw[1]));
        }
        float ask = sum > LAZY_BIG ? 1.f : 0.f;
        {
          const unsigned u = __float_as_uint(ask);
          const auto sw = __builtin_amdgcn_permlane32_swap(u, u, false, false);
          ask = fmaxf(__uint_as_float(sw[0]), __uint_as_float(sw[1]));
        }
        const float over = __builtin_fmaf(mx, scale_log2e, nmc);
        const float d = (t == 0) ? over : (ask > 0.f ? over : 0.f);
        if (t != 0) {
          const float alpha = __builtin_amdgcn_exp2f(-d);
          lrun *= alpha;
#pragma unroll
          for (int db = 0; db < C::DB; ++db)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[db][r] *= alpha;
        }
        nmc -= d;
        s0 = 0.f;
        s1 = 0.f;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
          for (int r = 0; r < 16; r += 2) {
            const float pa = __builtin_amdgcn_exp2f(__builtin_fmaf(s[kb][r], scale_log2e, nmc));
            const float pb = __builtin_amdgcn_exp2f(__builtin_fmaf(s[kb][r + 1], scale_log2e, nmc));
            x[kb][r] = pa;
            x[kb][r + 1] = pb;
            s0 += pa;
            s1 += pb;
          }
        asm volatile("" : "+v"(s0), "+v"(s1));
        sum = s0 + s1;
      }
      lrun += sum;
      bf16x8 pf[2][2];   // [kb][h]: registers 8 h .. 8 h + 7 of block kb = the 8 key slots of this lane in PV k-step (kb, h)
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
          for (int e = 0; e < 8; ++e) pf[kb][h][e] = f2bf(x[kb][8 * h + e]);

      // ---- O^T += V^T . P^T: 4 k-steps of 16 keys x DB d blocks of 32
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const int key0 = kb * 32 + C::pi(vq + 4 * hi + 16 * h);        // rd = 0
          const int key1 = kb * 32 + C::pi(vq + 4 * hi + 8 + 16 * h);    // rd = 1
#pragma unroll
          for (int db = 0; db < C::DB; ++db) {
            const int col = db * 32 + 16 * vg + 4 * vp4;
            const int ch = min(col >> 3, C::CPR - 1);   // (head_dim 80, third block: columns 80-95 do not exist - any finite data, rows discarded)
            const int a0 = key0 * C::ROWB + C::chunk(key0, ch) * 16 + (col & 7) * 2;
            const int a1 = key1 * C::ROWB + C::chunk(key1, ch) * 16 + (col & 7) * 2;
            const bf16x4 v0 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)(vt_ + a0));
            const bf16x4 v1 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)(vt_ + a1));
            const bf16x8 vf = (bf16x8){v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
            o[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf[kb][h], o[db], 0, 0, 0);
          }
        }
    }  // active
    __syncthreads();
  }

  // ---- epilogue: O[q][32 db + 8 rg + 4 hi + j] = o[db][4 rg + j] / l
  float lsum = lrun;
  lsum += __shfl_xor(lsum, 32, 64);
  const float inv = 1.0f / lsum;
  if (active && qrow < L) {
    bf16_t* op = O + (os + qrow) * o_ts + (long)h0 * o_hs;
#pragma unroll
    for (int db = 0; db < C::DB; ++db)
#pragma unroll
      for (int rg = 0; rg < 4; ++rg) {
        const int d0 = db * 32 + 8 * rg + 4 * hi;
        if (d0 < HD) {
          bf16x4 ov;
#pragma unroll
          for (int j = 0; j < 4; ++j) ov[j] = f2bf(o[db][4 * rg + j] * inv);
          *(bf16x4*)(op + d0) = ov;
        }
      }
  }
}

// ------------------------------------------------------------------------------------------------------------------------
// The same kernel SOFTWARE-PIPELINED (round 6, knob "attn_mfma32" = 2): a wave's matrix and vector work only overlap when they sit
// next to each other in its instruction stream (in-order issue: a block of 10 MFMAs followed by a block of 100 vector instructions
// runs them one after the other, whatever the other waves of the SIMD do - the round-5 counters).  So the loop is skewed by one
// tile: iteration t computes S(t+1) = K(t+1) . Q^T while it exponentiates S(t) - the two are independent - with the vector work cut
// into slices placed BETWEEN the MFMAs (`sched_barrier` pins the order written here), and then O += V(t)^T . P(t) with the
// transposing reads and the bf16 conversion of the later k-steps between those MFMAs.  K tiles therefore run two ahead (a ring
// of three), V tiles one ahead (two); one block barrier per tile as before.
// Same arithmetic as attn_fwd32_kernel in the same order per element: the two give the same bits (tested).
// ------------------------------------------------------------------------------------------------------------------------
template <int HD>
__global__ __launch_bounds__(256, 2) void attn_fwd32p_kernel(
    const bf16_t* __restrict__ Q, long q_ts, long q_hs, const bf16_t* __restrict__ K, long k_ts,
    long k_hs, const bf16_t* __restrict__ V, long v_ts, long v_hs, bf16_t* __restrict__ O, long o_ts,
    long o_hs, const int* __restrict__ q_start, const int* __restrict__ o_start,
    const int* __restrict__ k_start, const int* __restrict__ seq_len, int n_heads, int kv_group, int nqb,
    int n_pairs, float scale_log2e) {
  using C = Cfg32<HD>;
  extern __shared__ __attribute__((aligned(16))) char lds[];  // [3 K tiles][2 V tiles]
  const int tid = threadIdx.x;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l = tid & 63;
  const int ql = l & 31, hi = l >> 5;

  const int bid = blockIdx.x;
  int qb, pair;
  if ((n_pairs & 7) == 0) {
    qb = (bid >> 3) % nqb;
    pair = (bid / (8 * nqb)) * 8 + (bid & 7);
  } else {
    qb = bid % nqb;
    pair = bid / nqb;
  }
  const int b = pair / n_heads, h0 = pair % n_heads;
  const int hk = h0 / kv_group;
  const int L = seq_len[b];
  if (qb * QB >= L) return;
  const bool active = (qb * QB + w * 32) < L;
  const long qs = q_start[b], ks0 = k_start[b];
  const long os = o_start ? (long)o_start[b] : qs;
  const int ntiles = (L + KB - 1) / KB;

  const int qrow = qb * QB + w * 32 + ql;
  bf16x8 qf[C::KS16];
  {
    const bf16_t* qp = Q + (qs + min(qrow, L - 1)) * q_ts + (long)h0 * q_hs;
#pragma unroll
    for (int ks = 0; ks < C::KS16; ++ks) qf[ks] = *(const bf16x8*)(qp + ks * 16 + hi * 8);
  }

  // ---- staging: a K (or V) tile is NP pieces of 1 KiB; wave w takes pieces w, w + 4, ...
  const bf16_t* Kb = K + (long)hk * k_hs;
  const bf16_t* Vb = V + (long)hk * v_hs;
  constexpr int NPW1 = (C::NP + 3) / 4;
  const char* Kseq = (const char*)(Kb + ks0 * k_ts);
  const char* Vseq = (const char*)(Vb + ks0 * v_ts);
  unsigned koffs[NPW1], voffs[NPW1];
  int prow[NPW1];
#pragma unroll
  for (int i = 0; i < NPW1; ++i) {
    const int ci = (w + 4 * i) * 64 + l;
    const int row = ci / C::CPR;
    const int cp = ci - row * C::CPR;
    int c = cp - C::rot(row);
    c = c < 0 ? c + C::CPR : c;
    prow[i] = row;
    koffs[i] = (unsigned)((long)row * k_ts * 2 + c * 16);
    voffs[i] = (unsigned)((long)row * v_ts * 2 + c * 16);
  }
  char* const kring = lds;
  char* const vring = lds + 3 * C::TILE;
  auto stage_tile = [&](char* dst, const char* seq, long ts, const unsigned (&offs)[NPW1], int t) {
    if ((t * KB + KB) <= L) {
      const char* t0 = seq + (long)t * KB * ts * 2;
      asm volatile("" : "+s"(t0));
#pragma unroll
      for (int i = 0; i < NPW1; ++i)
        if (w + 4 * i < C::NP) glds16(t0 + offs[i], dst + (w + 4 * i) * 1024);
    } else {
      asm volatile("" ::: "memory");
#pragma unroll
      for (int i = 0; i < NPW1; ++i)
        if (w + 4 * i < C::NP) {
          const long ts2 = ts * 2;
          const char* src = seq + (long)min(t * KB + prow[i], L - 1) * ts2 + (offs[i] - (unsigned)((long)prow[i] * ts2));
          glds16(src, dst + (w + 4 * i) * 1024);
        }
    }
  };

  f32x16 o[C::DB];
#pragma unroll
  for (int d = 0; d < C::DB; ++d)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[d][r] = 0.f;
  float nmc = 0.f, lrun = 0.f;
  constexpr float LAZY_BIG = 1024.f;

  const int krow_b = C::pi(ql);
  int koff[C::KS16];
#pragma unroll
  for (int ks = 0; ks < C::KS16; ++ks) koff[ks] = krow_b * C::ROWB + C::chunk(krow_b, 2 * ks + hi) * 16;
  const int vq = (l & 15) >> 2, vp4 = l & 3, vg = (l >> 4) & 1;
  // V read offsets of the 4 k-steps x DB d blocks x 2 reads are too many to keep: the per-k-step row parts (8) and the per-block
  // column parts are kept and combined by adds
  int vrow[2][2][2];   // [kb][h][rd] -> key row byte offset
  int vrot[2][2][2];   // rot of that row (chunks)
#pragma unroll
  for (int kb = 0; kb < 2; ++kb)
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int rd = 0; rd < 2; ++rd) {
        const int key = kb * 32 + C::pi(vq + 4 * hi + 8 * rd + 16 * h);
        vrow[kb][h][rd] = key * C::ROWB;
        vrot[kb][h][rd] = C::rot(key);
      }
  auto v_addr = [&](int kb, int h, int rd, int db) {
    const int col = db * 32 + 16 * vg + 4 * vp4;
    const int ch = min(col >> 3, C::CPR - 1);
    int x = ch + vrot[kb][h][rd];
    x = x >= C::CPR ? x - C::CPR : x;
    return vrow[kb][h][rd] + x * 16 + (col & 7) * 2;
  };

  auto qk_plain = [&](f32x16 (&s)[2], const char* kt_) {
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
      for (int r = 0; r < 16; ++r) s[kb][r] = 0.f;
#pragma unroll
      for (int ks = 0; ks < C::KS16; ++ks) {
        const bf16x8 kf = *(const bf16x8*)(kt_ + kb * 32 * C::ROWB + koff[ks]);
        s[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[ks], s[kb], 0, 0, 0);
      }
    }
  };

  // one tile: S(t) in `sc`; computes S(t+1) into `sn` (from K ring slot kn) while it exponentiates sc, then O += V(t)^T P(t)
  auto tile = [&](f32x16 (&sc)[2], f32x16 (&sn)[2], int t, const char* kn_, const char* vt_) {
    if (t * KB + KB > L) {   // ragged last tile
      asm volatile("" ::: "memory");
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int key = t * KB + kb * 32 + C::pi((r & 3) + 8 * (r >> 2) + 4 * hi);
          if (key >= L) sc[kb][r] = -1e30f;
        }
    }
    // ---- phase A: S(t+1) MFMAs with the fast-path exponentials of S(t) between them
    float x[2][16];
    float s0 = 0.f, s1 = 0.f;
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int r = 0; r < 16; ++r) sn[kb][r] = 0.f;
    constexpr int NM = 2 * C::KS16;            // MFMAs of phase A
    constexpr int PER = (32 + NM - 3) / (NM - 2);   // exponentials per slice: everything done two MFMAs before the end
    bf16x8 kf_next = *(const bf16x8*)(kn_ + koff[0]);
    int done = 0;
#pragma unroll
    for (int m = 0; m < NM; ++m) {
      const int kb = m / C::KS16, ks = m % C::KS16;
      const bf16x8 kf = kf_next;
      if (m + 1 < NM) kf_next = *(const bf16x8*)(kn_ + ((m + 1) / C::KS16) * 32 * C::ROWB + koff[(m + 1) % C::KS16]);
      __builtin_amdgcn_sched_barrier(0);
      sn[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[ks], sn[kb], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int j = 0; j < PER; j += 2) {
        if (done < 32) {
          const int kb2 = done >> 4, r = done & 15;
          const float pa = __builtin_amdgcn_exp2f(__builtin_fmaf(sc[kb2][r], scale_log2e, nmc));
          const float pb = __builtin_amdgcn_exp2f(__builtin_fmaf(sc[kb2][r + 1], scale_log2e, nmc));
          x[kb2][r] = pa;
          x[kb2][r + 1] = pb;
          s0 += pa;
          s1 += pb;
          done += 2;
        }
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("" : "+v"(s0), "+v"(s1));
    float sum = s0 + s1;
    if (t == 0 || __builtin_amdgcn_ballot_w64(sum > LAZY_BIG) != 0) {
      asm volatile("" ::: "memory");
      float mx = -1e30f;
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int r = 0; r < 16; ++r) mx = fmaxf(mx, sc[kb][r]);
      {
        const unsigned u = __float_as_uint(mx);
        const auto sw = __builtin_amdgcn_permlane32_swap(u, u, false, false);
        mx = fmaxf(__uint_as_float(sw[0]), __uint_as_float(sw[1]));
      }
      float ask = sum > LAZY_BIG ? 1.f : 0.f;
      {
        const unsigned u = __float_as_uint(ask);
        const auto sw = __builtin_amdgcn_permlane32_swap(u, u, false, false);
        ask = fmaxf(__uint_as_float(sw[0]), __uint_as_float(sw[1]));
      }
      const float over = __builtin_fmaf(mx, scale_log2e, nmc);
      const float d = (t == 0) ? over : (ask > 0.f ? over : 0.f);
      if (t != 0) {
        const float alpha = __builtin_amdgcn_exp2f(-d);
        lrun *= alpha;
#pragma unroll
        for (int db = 0; db < C::DB; ++db)
#pragma unroll
          for (int r = 0; r < 16; ++r) o[db][r] *= alpha;
      }
      nmc -= d;
      s0 = 0.f;
      s1 = 0.f;
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int r = 0; r < 16; r += 2) {
          const float pa = __builtin_amdgcn_exp2f(__builtin_fmaf(sc[kb][r], scale_log2e, nmc));
          const float pb = __builtin_amdgcn_exp2f(__builtin_fmaf(sc[kb][r + 1], scale_log2e, nmc));
          x[kb][r] = pa;
          x[kb][r + 1] = pb;
          s0 += pa;
          s1 += pb;
        }
      asm volatile("" : "+v"(s0), "+v"(s1));
      sum = s0 + s1;
    }
    lrun += sum;
    // ---- phase B: O += V(t)^T P(t); the transposing reads run one MFMA ahead, the bf16 conversion of the NEXT k-step's P sits
    //      between this k-step's MFMAs
    bf16x8 pf;
#pragma unroll
    for (int e = 0; e < 8; ++e) pf[e] = f2bf(x[0][e]);
    bf16x4 va = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)(vt_ + v_addr(0, 0, 0, 0)));
    bf16x4 vb = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)(vt_ + v_addr(0, 0, 1, 0)));
#pragma unroll
    for (int kstep = 0; kstep < 4; ++kstep) {
      const int kb = kstep >> 1, h = kstep & 1;
      bf16x8 pf_next = pf;
#pragma unroll
      for (int db = 0; db < C::DB; ++db) {
        const bf16x8 vf = (bf16x8){va[0], va[1], va[2], va[3], vb[0], vb[1], vb[2], vb[3]};
        const int nstep = db + 1 < C::DB ? kstep : kstep + 1, ndb = db + 1 < C::DB ? db + 1 : 0;
        if (nstep < 4) {
          va = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)(vt_ + v_addr(nstep >> 1, nstep & 1, 0, ndb)));
          vb = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)(vt_ + v_addr(nstep >> 1, nstep & 1, 1, ndb)));
        }
        __builtin_amdgcn_sched_barrier(0);
        o[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf, o[db], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        if (db == 0 && kstep + 1 < 4) {
          const int kb2 = (kstep + 1) >> 1, h2 = (kstep + 1) & 1;
#pragma unroll
          for (int e = 0; e < 8; ++e) pf_next[e] = f2bf(x[kb2][8 * h2 + e]);
        }
      }
      pf = pf_next;
      (void)kb;
      (void)h;
    }
  };

  // ---- prologue: K(0), K(1), V(0) in flight; S(0)
  stage_tile(kring, Kseq, k_ts, koffs, 0);
  stage_tile(vring, Vseq, v_ts, voffs, 0);
  if (ntiles > 1) stage_tile(kring + C::TILE, Kseq, k_ts, koffs, 1);
  __syncthreads();
  f32x16 sa[2], sb[2];
  if (active) qk_plain(sa, kring);
  // iteration t: K(t+1) in ring slot (t+1) % 3 (clamped to the last tile: the extra S of the last iteration is never used)
  for (int t = 0; t < ntiles; t += 2) {
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      const int tt = t + half;
      if (tt < ntiles) {
        if (tt > 0) __syncthreads();   // K(tt+1), V(tt) landed (every wave waited for its own pieces: the barrier's vmcnt(0)); K(tt-1) / V(tt-1) free
        if (tt + 2 < ntiles) stage_tile(kring + ((tt + 2) % 3) * C::TILE, Kseq, k_ts, koffs, tt + 2);
        if (tt + 1 < ntiles) stage_tile(vring + ((tt + 1) & 1) * C::TILE, Vseq, v_ts, voffs, tt + 1);
        if (active) {
          const char* kn_ = kring + (min(tt + 1, ntiles - 1) % 3) * C::TILE;
          const char* vt_ = vring + (tt & 1) * C::TILE;
          if (half == 0) tile(sa, sb, tt, kn_, vt_);
          else tile(sb, sa, tt, kn_, vt_);
        }
      }
    }
  }

  float lsum = lrun;
  lsum += __shfl_xor(lsum, 32, 64);
  const float inv = 1.0f / lsum;
  if (active && qrow < L) {
    bf16_t* op = O + (os + qrow) * o_ts + (long)h0 * o_hs;
#pragma unroll
    for (int db = 0; db < C::DB; ++db)
#pragma unroll
      for (int rg = 0; rg < 4; ++rg) {
        const int d0 = db * 32 + 8 * rg + 4 * hi;
        if (d0 < HD) {
          bf16x4 ov;
#pragma unroll
          for (int j = 0; j < 4; ++j) ov[j] = f2bf(o[db][4 * rg + j] * inv);
          *(bf16x4*)(op + d0) = ov;
        }
      }
  }
}

template <int HD>
int launch32(const void* Q, long q_ts, long q_hs, const void* K, long k_ts, long k_hs, const void* V,
             long v_ts, long v_hs, void* O, long o_ts, long o_hs, const int* q_start,
             const int* o_start, const int* k_start, const int* seq_len, int n_seq, int n_heads,
             int kv_group, int max_len, float scale, hipStream_t st) {
  using C = Cfg32<HD>;
  const int nqb = (max_len + QB - 1) / QB;
  const int n_pairs = n_seq * n_heads;
  static bool attr_set = false;
  if (!attr_set) {
    if (hipFuncSetAttribute((const void*)attn_fwd32_kernel<HD>, hipFuncAttributeMaxDynamicSharedMemorySize, 4 * C::TILE) != hipSuccess ||
        hipFuncSetAttribute((const void*)attn_fwd32p_kernel<HD>, hipFuncAttributeMaxDynamicSharedMemorySize, 5 * C::TILE) != hipSuccess)
      return OWC_ERR_HIP;
    attr_set = true;
  }
  const int prof = owc_gemm_profile_begin(0.0, OWC_PROF_ATTN_VISION, st);
  if (g_attn_mfma32 == 2)
    hipLaunchKernelGGL((attn_fwd32p_kernel<HD>), dim3(n_pairs * nqb), dim3(256), 5 * C::TILE, st, (const bf16_t*)Q, q_ts, q_hs,
                       (const bf16_t*)K, k_ts, k_hs, (const bf16_t*)V, v_ts, v_hs, (bf16_t*)O, o_ts, o_hs, q_start, o_start, k_start,
                       seq_len, n_heads, kv_group, nqb, n_pairs, scale * 1.4426950408889634f);
  else
    hipLaunchKernelGGL((attn_fwd32_kernel<HD>), dim3(n_pairs * nqb), dim3(256), 4 * C::TILE, st, (const bf16_t*)Q, q_ts, q_hs,
                       (const bf16_t*)K, k_ts, k_hs, (const bf16_t*)V, v_ts, v_hs, (bf16_t*)O, o_ts, o_hs, q_start, o_start, k_start,
                       seq_len, n_heads, kv_group, nqb, n_pairs, scale * 1.4426950408889634f);
  owc_gemm_profile_end(prof, st);
  return hipGetLastError() == hipSuccess ? OWC_OK : OWC_ERR_HIP;
}

template <int HD, bool CAUSAL>
int launch(const void* Q, long q_ts, long q_hs, const void* K, long k_ts, long k_hs, const void* V,
           long v_ts, long v_hs, void* O, long o_ts, long o_hs, const int* q_start,
           const int* o_start, const int* k_start, const int* seq_len, const int* q_len, int n_seq, int n_heads,
           int kv_group, int max_len, float scale, hipStream_t st) {
  using C = Cfg<HD>;
  // causal GQA: the heads of a kv group are packed into the rows of a block (see the kernel): a pair is (sequence, kv head)
  const int pack = (CAUSAL && kv_group > 1 && g_attn_gqa_pack && n_heads % kv_group == 0) ? kv_group : 0;
  const int rows = (pack ? pack : 1) * max_len;      // rows of the longest pair
  const int nqb = (rows + QB - 1) / QB;
  // causal: even split of the longest pair's rows over its nqb blocks, in whole 32-row wave tiles
  const int qrows = CAUSAL ? min(QB, ((rows + nqb - 1) / nqb + 31) / 32 * 32) : QB;
  const int n_pairs = pack ? n_seq * (n_heads / pack) : n_seq * n_heads;
  const int lds_bytes = 4 * C::TILE;
  static bool attr_set = false;
  if (!attr_set) {
    if (hipFuncSetAttribute((const void*)attn_fwd_kernel<HD, CAUSAL>,
                            hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes) != hipSuccess)
      return OWC_ERR_HIP;
    attr_set = true;
  }
  // sequence lengths live on the device: the launch is recorded with work 0 and priced by the caller (include/owc.h)
  // class: head_dim 128 causal = decoder prefill, head_dim 128 non-causal = the decode-step mapping (a KV-cache stream) unless the
  // prefill driver says otherwise (its last layer's last-token launch), 80 / 64 / 32 = the vision towers
  const int cls = HD != 128 ? OWC_PROF_ATTN_VISION : (CAUSAL || g_attn_class_prefill) ? OWC_PROF_ATTN_PREFILL : OWC_PROF_ATTN_DECODE;
  const int prof = owc_gemm_profile_begin(0.0, cls, st);
  hipLaunchKernelGGL((attn_fwd_kernel<HD, CAUSAL>), dim3(n_pairs * nqb), dim3(256), lds_bytes, st,
                     (const bf16_t*)Q, q_ts, q_hs, (const bf16_t*)K, k_ts, k_hs, (const bf16_t*)V,
                     v_ts, v_hs, (bf16_t*)O, o_ts, o_hs, q_start, o_start, k_start, seq_len, q_len, n_heads,
                     kv_group, nqb, n_pairs, scale * 1.4426950408889634f, g_attn_dbg, qrows, pack);
  owc_gemm_profile_end(prof, st);
  return hipGetLastError() == hipSuccess ? OWC_OK : OWC_ERR_HIP;
}


// ------------------------------------------------------------------------------------------------------------------------
// Decode step (round 3): M-RoPE of the new token + KV-cache write + attention over the sequence's cache in ONE launch.
//
// A decode step used to be  mrope_kv_kernel (rotate q and k of the fed token, write k / v into the cache: ~6 us of launch and
// latency per layer at small batch) followed by attn_fwd_kernel in its "decode mapping" (the G q heads of a kv group are the
// query rows of one block) - where only ONE of the block's four waves had rows and walked the key tiles one after the other
// behind two block barriers per tile (12.6 us per layer at batch 1 for 286 keys: with the launch before it, 13 % of the step).
// Here a block is still one (sequence, kv head) pair, but
//   * the four waves SPLIT THE KEYS: wave w owns the 64-key tiles w, w + 4, ... and runs its own online softmax over them, with
//     no block barrier in the loop; the partial (max, sum, O) of the waves are merged in wave order through LDS - a fixed
//     order, so the result of a sequence does not depend on what else is in the batch;
//   * K fragments go global -> registers (every K byte is used by exactly one wave: LDS staging would only add a hop), V tiles
//     go through wave-private LDS (16 KiB, LDS-DMA) because the O^T = V^T . P^T product needs the transposing LDS read;
//   * the wave that owns the LAST tile rotates the fed token's k with the rope table, writes that k row and the v row into
//     the cache and waits for the stores before it loads its tiles, so the new key is simply part of its last tile;
//   * every wave rotates the G query rows in registers (a rotary pair (i, i + 64) lives in the same lane: k-steps ks and ks + 2).
// Arithmetic and rounding points are those of mrope_kv_kernel + attn_fwd_kernel (rope: bf16(bf16(x cos) + bf16(rot sin)) with
// the bf16-rounded table; scores fp32, P rounded to bf16 before P.V); only the order in which partial softmax results are
// combined differs.  HF: apply_multimodal_rotary_pos_emb (:180-222; three equal position streams for a generated token, i.e.
// plain 1-D rope), Qwen2VLAttention.forward (:508-556).
// ------------------------------------------------------------------------------------------------------------------------
constexpr int DEC_VTILE = KB * 256;  // one V tile: 64 keys x 256 B

template <int NBUF>   // V tiles per wave in LDS: 2 = the next tile is fetched under the current one (128 KiB per block: one block per CU, the
                      // small-batch form), 1 = 64 KiB per block, two blocks = eight streaming waves per CU (large batches: HBM-bound)
__global__ __launch_bounds__(256) void attn_decode_fused_kernel(
    const bf16_t* __restrict__ qkv, long ld, const int* __restrict__ pos, const float* __restrict__ cos_t,
    const float* __restrict__ sin_t, bf16_t* __restrict__ kc, bf16_t* __restrict__ vc, const int* __restrict__ slot,
    const int* __restrict__ write_idx, const int* __restrict__ k_len, bf16_t* __restrict__ O, long ldo, int n_q, int n_kv,
    int s_max, float scale_log2e) {
  // The two instantiations (NBUF 1 / 2) must give the SAME BITS: a decode batch that shrinks (EOS-aware row compaction) crosses
  // from one to the other in the middle of a sequence.  Under -ffp-contract=fast the compiler decided per instantiation which
  // mul + add pairs become an fma (round 4: the disassemblies differed by one v_fma_f32 / v_sub_f32, and ~1 % of the row-steps of
  // a 7B decode at >= 5 key tiles flipped a near-tie argmax), so contraction is OFF here and every fused multiply-add below is spelled out.
#pragma clang fp contract(off)
  using C = Cfg<128>;
  extern __shared__ __attribute__((aligned(16))) char lds[];  // [4 waves][2 buffers][V tile 16 KiB]; reused for the merge
  const int tid = threadIdx.x;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l = tid & 63;
  const int fr = l & 15, g = l >> 4;
  const int b = blockIdx.x / n_kv, hk = blockIdx.x % n_kv;
  const int G = n_q / n_kv;
  const int L = k_len[b];                       // keys, the fed token's included
  const int widx = write_idx[b];
  const long row0 = ((long)slot[b] * n_kv + hk) * s_max;   // first cache row of this (sequence, kv head)
  bf16_t* Kseq = kc + row0 * 128;
  bf16_t* Vseq = vc + row0 * 128;
  const int ntiles = (L + KB - 1) / KB;
  const bf16_t* tok = qkv + (long)b * ld;
  const long trow = (long)pos[b] * 64;          // rope table row of the fed token

  // ---- the owner of the last tile: rope(k), cache write of k and v (lanes 0-7: one rotary pair of 8-dim chunks each; lanes 8-23: v)
  // (widx < s_max: a finished row that the caller has not dropped yet keeps advancing its write index; it must never write into
  //  the next slot - the host compacts such rows away before that, this is the second line of defence)
  if (w == ((ntiles - 1) & 3) && widx < s_max) {
    if (l < 8) {
      const bf16_t* kp = tok + (long)(n_q + hk) * 128;
      const bf16x8 a = *(const bf16x8*)(kp + l * 8), bb = *(const bf16x8*)(kp + l * 8 + 64);
      const f32x4 c0 = *(const f32x4*)(cos_t + trow + l * 8), c1 = *(const f32x4*)(cos_t + trow + l * 8 + 4);
      const f32x4 s0 = *(const f32x4*)(sin_t + trow + l * 8), s1 = *(const f32x4*)(sin_t + trow + l * 8 + 4);
      bf16x8 oa, ob;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float c = e < 4 ? c0[e & 3] : c1[e & 3], sn = e < 4 ? s0[e & 3] : s1[e & 3];
        const float x1 = bf2f(a[e]), x2 = bf2f(bb[e]);
        oa[e] = f2bf(rbf(x1 * c) + rbf(-x2 * sn));
        ob[e] = f2bf(rbf(x2 * c) + rbf(x1 * sn));
      }
      bf16_t* dst = Kseq + (long)widx * 128;
      *(bf16x8*)(dst + l * 8) = oa;
      *(bf16x8*)(dst + l * 8 + 64) = ob;
    } else if (l < 24) {
      const bf16_t* vp = tok + (long)(n_q + n_kv + hk) * 128;
      *(bf16x8*)(Vseq + (long)widx * 128 + (l - 8) * 8) = *(const bf16x8*)(vp + (l - 8) * 8);
    }
    __builtin_amdgcn_s_waitcnt(0);   // the two rows are in L2 before this wave's tile loads are issued (gfx9: vmcnt counts stores)
  }

  // ---- Q fragments of the G query heads (lane = query column fr, clamped), rotated in registers
  bf16x8 qf[4];
  {
    const bf16_t* qp = tok + (long)(hk * G + min(fr, G - 1)) * 128;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) qf[ks] = *(const bf16x8*)(qp + (ks * 4 + g) * 8);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {   // chunk c = 4 ks + g < 8 pairs with chunk c + 8 = k-step ks + 2, same lane
      const int i0 = (ks * 4 + g) * 8;
      const f32x4 c0 = *(const f32x4*)(cos_t + trow + i0), c1 = *(const f32x4*)(cos_t + trow + i0 + 4);
      const f32x4 s0 = *(const f32x4*)(sin_t + trow + i0), s1 = *(const f32x4*)(sin_t + trow + i0 + 4);
      bf16x8 oa, ob;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float c = e < 4 ? c0[e & 3] : c1[e & 3], sn = e < 4 ? s0[e & 3] : s1[e & 3];
        const float x1 = bf2f(qf[ks][e]), x2 = bf2f(qf[ks + 2][e]);
        oa[e] = f2bf(rbf(x1 * c) + rbf(-x2 * sn));
        ob[e] = f2bf(rbf(x2 * c) + rbf(x1 * sn));
      }
      qf[ks] = oa;
      qf[ks + 2] = ob;
    }
  }

  f32x4 o[C::DT];
#pragma unroll
  for (int d = 0; d < C::DT; ++d) o[d] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float mrun = -1e30f, lrun = 0.f;

  char* vbuf = lds + w * (NBUF * DEC_VTILE);
  // V tile t -> this wave's LDS buffer: 16 pieces of 1 KiB (4 keys x 256 B each), rows clamped at the ragged end, the 16-byte
  // chunk order of a row XOR-swizzled with (key & 7) << 1 like the block-staged kernel above
  auto stage_v = [&](int buf, int t) {
    char* base = vbuf + buf * DEC_VTILE;
#pragma unroll
    for (int p = 0; p < 16; ++p) {
      const int key = p * 4 + (l >> 4), posc = l & 15;
      const int c = posc ^ ((key & 7) << 1);
      const bf16_t* src = Vseq + (long)min(t * KB + key, L - 1) * 128 + c * 8;
      glds16(src, base + p * 1024);
    }
  };
  bf16x8 kf[4][4];
  auto load_k = [&](int t) {   // K fragments of tile t: lane (fr, g) <- key 16 kt + fr, chunk 4 ks + g
#pragma unroll
    for (int kt = 0; kt < 4; ++kt) {
      const bf16_t* kr = Kseq + (long)min(t * KB + kt * 16 + fr, L - 1) * 128 + g * 8;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) kf[kt][ks] = *(const bf16x8*)(kr + ks * 32);
    }
  };
  const int vq = fr >> 2, vp = fr & 3;

  int buf = 0;
  if (NBUF == 2 && w < ntiles) stage_v(0, w);
  for (int t = w; t < ntiles; t += 4) {
    load_k(t);
    // WAR guard in the SOURCE (as in the ring GEMMs): the buffer restaged below was last read by the previous tile's transposing
    // LDS reads; they were consumed by its MFMAs a whole softmax ago, so the wait is free - but nothing else stops a scheduler or
    // unroll change from hoisting the DMA above them
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    if (NBUF == 1) stage_v(0, t);
    else if (t + 4 < ntiles) stage_v(buf ^ 1, t + 4);
    // ---- S^T = K . Q^T (waits for the K fragments only: the V tiles may still fly)
    f32x4 s[4];
#pragma unroll
    for (int kt = 0; kt < 4; ++kt) {
      s[kt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) s[kt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf[kt][ks], qf[ks], s[kt], 0, 0, 0);
    }
    if (t * KB + KB > L) {   // ragged last tile (wave-uniform branch)
      asm volatile("" ::: "memory");
#pragma unroll
      for (int kt = 0; kt < 4; ++kt)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (t * KB + kt * 16 + g * 4 + r >= L) s[kt][r] = -1e30f;
    }
    // ---- online softmax over this wave's keys (lane = query column)
    float mx = -1e30f;
#pragma unroll
    for (int kt = 0; kt < 4; ++kt)
#pragma unroll
      for (int r = 0; r < 4; ++r) mx = fmaxf(mx, s[kt][r]);
    mx = max_over_groups(mx);
    const float mnew = fmaxf(mrun, mx);
    const float alpha = __builtin_amdgcn_exp2f((mrun - mnew) * scale_log2e);
    const float neg = -mnew * scale_log2e;
    mrun = mnew;
    float sum = 0.f;
    float x[4][4];
#pragma unroll
    for (int kt = 0; kt < 4; ++kt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        x[kt][r] = __builtin_amdgcn_exp2f(__builtin_fmaf(s[kt][r], scale_log2e, neg));
        sum += x[kt][r];
      }
    lrun = __builtin_fmaf(lrun, alpha, sum);
#pragma unroll
    for (int d = 0; d < C::DT; ++d) {
      o[d][0] *= alpha;
      o[d][1] *= alpha;
      o[d][2] *= alpha;
      o[d][3] *= alpha;
    }
    bf16x8 pf[2];
#pragma unroll
    for (int sx = 0; sx < 2; ++sx)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        pf[sx][r] = f2bf(x[2 * sx][r]);
        pf[sx][4 + r] = f2bf(x[2 * sx + 1][r]);
      }
    // ---- O^T += V^T . P^T : this tile's V pieces have landed (the next tile's 16 may fly)
    if (NBUF == 2 && t + 4 < ntiles) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const char* vt_ = vbuf + buf * DEC_VTILE;
#pragma unroll
    for (int sx = 0; sx < 2; ++sx) {
      const int key0 = sx * 32 + g * 4 + vq;
      const int key1 = key0 + 16;
#pragma unroll
      for (int d = 0; d < C::DT; ++d) {
        const int ch = 2 * d + (vp >> 1);
        const int a0 = key0 * 256 + ((ch ^ ((key0 & 7) << 1)) << 4) + (vp & 1) * 8;
        const int a1 = key1 * 256 + ((ch ^ ((key1 & 7) << 1)) << 4) + (vp & 1) * 8;
        const bf16x4 v0 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)(vt_ + a0));
        const bf16x4 v1 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)(vt_ + a1));
        const bf16x8 vf = (bf16x8){v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
        o[d] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, pf[sx], o[d], 0, 0, 0);
      }
    }
    if (NBUF == 2) buf ^= 1;
  }

  // ---- merge the four partial results in wave order (LDS reused: every wave is past its last V read after the barrier)
  float lsum = lrun;
  lsum += __shfl_xor(lsum, 16, 64);
  lsum += __shfl_xor(lsum, 32, 64);
  __syncthreads();
  float* mg = (float*)lds;                 // [4 waves][34 values][64 lanes]: m, l, o[8][4]
  {
    float* mine = mg + (size_t)w * 34 * 64 + l;
    mine[0] = mrun;
    mine[64] = lsum;
#pragma unroll
    for (int d = 0; d < C::DT; ++d)
#pragma unroll
      for (int r = 0; r < 4; ++r) mine[(2 + d * 4 + r) * 64] = o[d][r];
  }
  __syncthreads();
  if (w == 0 && fr < G) {
    float M = -1e30f;
#pragma unroll
    for (int k = 0; k < 4; ++k) M = fmaxf(M, mg[(size_t)k * 34 * 64 + l]);
    float f[4], Lt = 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      f[k] = __builtin_amdgcn_exp2f((mg[(size_t)k * 34 * 64 + l] - M) * scale_log2e);   // 0 for a wave without tiles
      Lt = __builtin_fmaf(mg[(size_t)k * 34 * 64 + 64 + l], f[k], Lt);
    }
    const float inv = 1.0f / Lt;
    bf16_t* op = O + (long)b * ldo + (long)(hk * G + fr) * 128 + g * 4;
#pragma unroll
    for (int d = 0; d < C::DT; ++d) {
      bf16x4 ov;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float acc = 0.f;
#pragma unroll
        for (int k = 0; k < 4; ++k) acc = __builtin_fmaf(mg[(size_t)k * 34 * 64 + (2 + d * 4 + r) * 64 + l], f[k], acc);
        ov[r] = f2bf(acc * inv);
      }
      *(bf16x4*)(op + d * 16) = ov;
    }
  }
}

}  // namespace

int owc_launch_attention(const void* Q, long q_ts, long q_hs, const void* K, long k_ts, long k_hs,
                         const void* V, long v_ts, long v_hs, void* O, long o_ts, long o_hs,
                         const int* q_start, const int* o_start, const int* k_start,
                         const int* seq_len, const int* q_len, int n_seq, int n_heads, int kv_group, int head_dim,
                         int max_q_len, int causal, float scale, hipStream_t st) {
  const int max_len = max_q_len;
  if (n_seq <= 0 || n_heads <= 0 || kv_group <= 0 || max_len <= 0) return OWC_ERR_SHAPE;
  if ((q_ts & 7) || (q_hs & 7) || (k_ts & 7) || (k_hs & 7) || (v_ts & 7) || (v_hs & 7) || (o_ts & 3) || (o_hs & 3))
    return OWC_ERR_SHAPE;
  // the vision towers (non-causal, every query row of a sequence, head_dim 80 / 64): the 32x32x16 kernel
  if (!causal && !q_len && g_attn_mfma32 && (head_dim == 80 || head_dim == 64)) {
    if (head_dim == 80)
      return launch32<80>(Q, q_ts, q_hs, K, k_ts, k_hs, V, v_ts, v_hs, O, o_ts, o_hs, q_start, o_start, k_start, seq_len, n_seq, n_heads,
                          kv_group, max_len, scale, st);
    return launch32<64>(Q, q_ts, q_hs, K, k_ts, k_hs, V, v_ts, v_hs, O, o_ts, o_hs, q_start, o_start, k_start, seq_len, n_seq, n_heads,
                        kv_group, max_len, scale, st);
  }
#define OWC_ATTN_CASE(HD_)                                                                         \
  if (head_dim == HD_)                                                                             \
    return causal ? launch<HD_, true>(Q, q_ts, q_hs, K, k_ts, k_hs, V, v_ts, v_hs, O, o_ts, o_hs, \
                                      q_start, o_start, k_start, seq_len, q_len, n_seq, n_heads, kv_group, \
                                      max_len, scale, st)                                          \
                  : launch<HD_, false>(Q, q_ts, q_hs, K, k_ts, k_hs, V, v_ts, v_hs, O, o_ts, o_hs,\
                                       q_start, o_start, k_start, seq_len, q_len, n_seq, n_heads, kv_group,\
                                       max_len, scale, st);
  OWC_ATTN_CASE(80)
  OWC_ATTN_CASE(128)
  OWC_ATTN_CASE(32)
  OWC_ATTN_CASE(64)
#undef OWC_ATTN_CASE
  return OWC_ERR_SHAPE;
}

// Fused decode step: rope(q, k) of the fed token + KV-cache write + attention (head_dim 128).  `O` rows are [B][n_q * 128].
int owc_launch_attn_decode_fused(const void* qkv, long ld, const int* pos, const float* cos_t, const float* sin_t, void* kc,
                                 void* vc, const int* slot, const int* write_idx, const int* k_len, void* O, long ldo, int B,
                                 int n_q, int n_kv, int s_max, float scale, hipStream_t st) {
  if (B <= 0 || n_q <= 0 || n_kv <= 0 || (n_q % n_kv) || n_q / n_kv > 16 || (ld & 7) || (ldo & 3)) return OWC_ERR_SHAPE;
  static bool attr_set = false;
  if (!attr_set) {
    if (hipFuncSetAttribute((const void*)attn_decode_fused_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 8 * DEC_VTILE) != hipSuccess ||
        hipFuncSetAttribute((const void*)attn_decode_fused_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 4 * DEC_VTILE) != hipSuccess)
      return OWC_ERR_HIP;
    attr_set = true;
  }
  const int prof = owc_gemm_profile_begin(0.0, OWC_PROF_ATTN_DECODE, st);
  // more blocks than CUs: the launch is a KV-cache stream - two blocks (eight loading waves) per CU beat the look-ahead buffer.
  // The two forms differ only in WHEN a V tile is fetched: same arithmetic, same order, same bits.
  if (B * n_kv > g_decode_nbuf1_min_blocks)
    hipLaunchKernelGGL(attn_decode_fused_kernel<1>, dim3(B * n_kv), dim3(256), 4 * DEC_VTILE, st, (const bf16_t*)qkv, ld, pos, cos_t,
                       sin_t, (bf16_t*)kc, (bf16_t*)vc, slot, write_idx, k_len, (bf16_t*)O, ldo, n_q, n_kv, s_max,
                       scale * 1.4426950408889634f);
  else
    hipLaunchKernelGGL(attn_decode_fused_kernel<2>, dim3(B * n_kv), dim3(256), 8 * DEC_VTILE, st, (const bf16_t*)qkv, ld, pos, cos_t,
                       sin_t, (bf16_t*)kc, (bf16_t*)vc, slot, write_idx, k_len, (bf16_t*)O, ldo, n_q, n_kv, s_max,
                       scale * 1.4426950408889634f);
  owc_gemm_profile_end(prof, st);
  return hipGetLastError() == hipSuccess ? OWC_OK : OWC_ERR_HIP;
}

void owc_attn_set_decode_nbuf1(int v) { g_decode_nbuf1_min_blocks = v < 0 ? 256 : v; }
void owc_attn_set_gqa_pack(int v) { g_attn_gqa_pack = v != 0; }
void owc_attn_set_mfma32(int v) { g_attn_mfma32 = v < 0 ? 0 : v; }
void owc_attn_class_prefill(int on) { g_attn_class_prefill = on; }
void owc_attn_set_dbg(int v) { g_attn_dbg = OWC_TK(true) ? v : 0; }
