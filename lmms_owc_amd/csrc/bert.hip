// Sentence encoder (all-MiniLM-L6-v2 == 6-layer BERT, hidden 384; round 5: MPNet - all-mpnet-base-v2, BASELINE.json configs[0] - through
// the same kernels: no token-type embedding, position ids = column + 2, one relative-position bias table added to the scores) + cosine scorer, fp32.
// Replaces the arithmetic of encode_sentence_bert (src/data/pipelines/text/_text.py:193-202:
// BertModel forward, mask-weighted mean pooling with clamp(min=1e-9), L2 normalisation) and of
// semantic_similarity's paired torch.bmm (src/data/metrics/_group.py:537-544), plus the
// all-classes cosine top-k the north star asks for.  The reference's CPU path is fp32, so every
// GEMM here runs as three-piece bf16 splits on the bf16 MFMA (gemm_f32.hip; the exact f32-input MFMA kernel stays selectable); the small kernels below are HBM/latency
// bound and use one wave per row.
#include "../../include/owc.h"
#include "owc_internal.h"

namespace {

inline size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

// out[t] = LayerNorm(word[ids[t]] + pos[t % L] + type[0])      (BERT: BertEmbeddings.forward)
__global__ __launch_bounds__(256) void bert_embed_ln_kernel(
    const int* __restrict__ ids, const int* __restrict__ tok_pos, const float* __restrict__ word, const float* __restrict__ pos,
    const float* __restrict__ type0, const float* __restrict__ g, const float* __restrict__ b,
    float* __restrict__ out, int T, int L, int H, float eps, int pos_offset) {
  const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
  const int t = blockIdx.x * 4 + w;
  if (t >= T) return;
  const float* we = word + (long)ids[t] * H;
  // packed rows carry their own column; MPNet's position ids are column + 2 (pos_offset; padding_idx + 1), BERT's the column
  const float* pe = pos + (long)((tok_pos ? tok_pos[t] : t % L) + pos_offset) * H;
  float v[16];
  float sum = 0.f;
  const int per = (H + 63) / 64;
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int c = i * 64 + l;
    v[i] = 0.f;
    if (i < per && c < H) {
      v[i] = type0 ? we[c] + type0[c] + pe[c] : we[c] + pe[c];   // (MPNet has no token-type embedding)
      sum += v[i];
    }
  }
  const float mean = wave_sum(sum) / (float)H;
  float var = 0.f;
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int c = i * 64 + l;
    if (i < per && c < H) {
      const float d = v[i] - mean;
      var += d * d;
    }
  }
  const float rstd = 1.0f / sqrtf(wave_sum(var) / (float)H + eps);
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int c = i * 64 + l;
    if (i < per && c < H) out[(long)t * H + c] = (v[i] - mean) * rstd * g[c] + b[c];
  }
}

// in-place fp32 LayerNorm, one wave per row (BERT: BertSelfOutput / BertOutput LayerNorm, eps 1e-12)
__global__ __launch_bounds__(256) void ln_f32_kernel(float* __restrict__ x, const float* __restrict__ g,
                                                     const float* __restrict__ b, int T, int H, float eps) {
  const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
  const int t = blockIdx.x * 4 + w;
  if (t >= T) return;
  float* row = x + (long)t * H;
  float v[16];
  float sum = 0.f;
  const int per = (H + 63) / 64;
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int c = i * 64 + l;
    v[i] = 0.f;
    if (i < per && c < H) {
      v[i] = row[c];
      sum += v[i];
    }
  }
  const float mean = wave_sum(sum) / (float)H;
  float var = 0.f;
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int c = i * 64 + l;
    if (i < per && c < H) {
      const float d = v[i] - mean;
      var += d * d;
    }
  }
  const float rstd = 1.0f / sqrtf(wave_sum(var) / (float)H + eps);
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int c = i * 64 + l;
    if (i < per && c < H) row[c] = (v[i] - mean) * rstd * g[c] + b[c];
  }
}

// Self-attention for short sequences, head_dim 32: one block per (sequence, head), K/V of the head in
// LDS, one thread per query row with an online softmax (BERT: BertSdpaSelfAttention.forward; padded
// keys get finfo.min added, i.e. weight exactly 0).
template <int HD>
__global__ __launch_bounds__(64) void bert_attn_kernel(const float* __restrict__ qkv,
                                                       const int* __restrict__ mask,
                                                       float* __restrict__ ctx, int L, int H,
                                                       int n_heads, float scale, const float* __restrict__ relb, int rel_span, int chunk) {
  // `chunk` keys of K / V are LDS-resident at a time (round 6: the whole sequence had to fit - head_dim 64 stopped at 317 tokens,
  // all-mpnet-base-v2 takes 512).  The online softmax walks the keys in ascending order whatever the chunking, and a thread keeps
  // its query's (q, acc, m, l) in registers across chunks: the result does not depend on `chunk`, bit for bit.
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* ks = (float*)smem;               // [chunk][HD]
  float* vs = ks + (size_t)chunk * HD;    // [chunk][HD]
  int* ms = (int*)(vs + (size_t)chunk * HD);
  const int s = blockIdx.x / n_heads, h = blockIdx.x % n_heads;
  const long base = (long)s * L;
  // MPNet: one learned bias per (head, key column - query column), added to the scaled score (`rel_bias`, include/owc.h)
  const float* rb = relb ? relb + (long)h * (2 * rel_span - 1) + (rel_span - 1) : nullptr;
  for (int i0 = 0; i0 < L; i0 += blockDim.x) {       // 64 query rows at a time (block-uniform trip count: the barriers below are safe)
    const int i = i0 + threadIdx.x;
    const bool live = i < L;
    float q[HD], acc[HD];
    const float* qr = qkv + (base + (live ? i : 0)) * 3 * H + h * HD;
#pragma unroll
    for (int d = 0; d < HD; ++d) {
      q[d] = qr[d];
      acc[d] = 0.f;
    }
    float m = -INFINITY, lsum = 0.f;
    for (int j0 = 0; j0 < L; j0 += chunk) {
      const int nj = min(chunk, L - j0);
      if (i0 > 0 || j0 > 0) __syncthreads();          // the previous chunk has been read by everybody
      if (i0 == 0 || chunk < L) {                     // (one chunk = the whole sequence: staged once, kept for every query group)
        for (int x = threadIdx.x; x < nj * (HD / 4); x += blockDim.x) {
          const int j = x / (HD / 4), c = x % (HD / 4);
          const float* row = qkv + (base + j0 + j) * 3 * H + h * HD + c * 4;
          *(f32x4*)(ks + j * HD + c * 4) = *(const f32x4*)(row + H);
          *(f32x4*)(vs + j * HD + c * 4) = *(const f32x4*)(row + 2 * H);
        }
        for (int j = threadIdx.x; j < nj; j += blockDim.x) ms[j] = mask[base + j0 + j];
      }
      __syncthreads();
      if (live) {
        for (int j = 0; j < nj; ++j) {
          if (!ms[j]) continue;
          float sc = 0.f;
#pragma unroll
          for (int d = 0; d < HD; ++d) sc += q[d] * ks[j * HD + d];
          sc *= scale;
          if (rb) sc += rb[j0 + j - i];
          const float mn = fmaxf(m, sc);
          const float a = expf(m - mn), p = expf(sc - mn);
          lsum = lsum * a + p;
#pragma unroll
          for (int d = 0; d < HD; ++d) acc[d] = acc[d] * a + p * vs[j * HD + d];
          m = mn;
        }
      }
    }
    if (live) {
      const float inv = 1.0f / lsum;
      float* o = ctx + (base + i) * H + h * HD;
#pragma unroll
      for (int d = 0; d < HD; ++d) o[d] = acc[d] * inv;
    }
  }
}

// sum(h * mask) / clamp(sum(mask), 1e-9), then / L2 norm   (_text.py:175-189, :202)
__global__ __launch_bounds__(256) void pool_norm_kernel(const float* __restrict__ x,
                                                        const int* __restrict__ mask,
                                                        float* __restrict__ out, int n, int L, int H) {
  const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
  const int s = blockIdx.x * 4 + w;
  if (s >= n) return;
  float v[16];
  const int per = (H + 63) / 64;
#pragma unroll
  for (int i = 0; i < 16; ++i) v[i] = 0.f;
  float cnt = 0.f;
  for (int j = 0; j < L; ++j) {
    const float mk = (float)mask[(long)s * L + j];
    cnt += mk;
    const float* row = x + ((long)s * L + j) * H;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int c = i * 64 + l;
      if (i < per && c < H) v[i] += row[c] * mk;
    }
  }
  const float den = fmaxf(cnt, 1e-9f);
  float sq = 0.f;
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    v[i] /= den;
    sq += v[i] * v[i];
  }
  const float nrm = sqrtf(wave_sum(sq));
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int c = i * 64 + l;
    if (i < per && c < H) out[(long)s * H + c] = v[i] / nrm;
  }
}

// Packed (padding-free) variants: rows are the REAL tokens only, sequence s owns rows [seq_start[s], seq_start[s+1]).
// Padded positions never reach the output in the reference either: they are masked out as keys (finfo.min) and get weight 0
// in the pooling, so dropping their rows changes no result.
template <int HD>
__global__ __launch_bounds__(64) void bert_attn_packed_kernel(const float* __restrict__ qkv, const int* __restrict__ seq_start,
                                                              float* __restrict__ ctx, int H, int n_heads, float scale,
                                                              const float* __restrict__ relb, int rel_span, int chunk) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int s = blockIdx.x / n_heads, h = blockIdx.x % n_heads;
  const long base = seq_start[s];
  const int L = seq_start[s + 1] - (int)base;
  const float* rb = relb ? relb + (long)h * (2 * rel_span - 1) + (rel_span - 1) : nullptr;   // (right-padded rows: packed index = column)
  float* ks = (float*)smem;               // [chunk][HD]
  float* vs = ks + (size_t)chunk * HD;    // [chunk][HD]
  for (int i0 = 0; i0 < L; i0 += blockDim.x) {   // (chunked like bert_attn_kernel: same key order, same bits)
    const int i = i0 + threadIdx.x;
    const bool live = i < L;
    float q[HD], acc[HD];
    const float* qr = qkv + (base + (live ? i : 0)) * 3 * H + h * HD;
#pragma unroll
    for (int d = 0; d < HD; ++d) {
      q[d] = qr[d];
      acc[d] = 0.f;
    }
    float m = -INFINITY, lsum = 0.f;
    for (int j0 = 0; j0 < L; j0 += chunk) {
      const int nj = min(chunk, L - j0);
      if (i0 > 0 || j0 > 0) __syncthreads();
      if (i0 == 0 || chunk < L) {
        for (int x = threadIdx.x; x < nj * (HD / 4); x += blockDim.x) {
          const int j = x / (HD / 4), c = x % (HD / 4);
          const float* row = qkv + (base + j0 + j) * 3 * H + h * HD + c * 4;
          *(f32x4*)(ks + j * HD + c * 4) = *(const f32x4*)(row + H);
          *(f32x4*)(vs + j * HD + c * 4) = *(const f32x4*)(row + 2 * H);
        }
      }
      __syncthreads();
      if (live) {
        for (int j = 0; j < nj; ++j) {   // same key order and online-softmax arithmetic as bert_attn_kernel over the unmasked keys
          float sc = 0.f;
#pragma unroll
          for (int d = 0; d < HD; ++d) sc += q[d] * ks[j * HD + d];
          sc *= scale;
          if (rb) sc += rb[j0 + j - i];
          const float mn = fmaxf(m, sc);
          const float a = expf(m - mn), p = expf(sc - mn);
          lsum = lsum * a + p;
#pragma unroll
          for (int d = 0; d < HD; ++d) acc[d] = acc[d] * a + p * vs[j * HD + d];
          m = mn;
        }
      }
    }
    if (live) {
      const float inv = 1.0f / lsum;
      float* o = ctx + (base + i) * H + h * HD;
#pragma unroll
      for (int d = 0; d < HD; ++d) o[d] = acc[d] * inv;
    }
  }
}

__global__ __launch_bounds__(256) void pool_norm_packed_kernel(const float* __restrict__ x, const int* __restrict__ seq_start,
                                                               float* __restrict__ out, int n, int H) {
  const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
  const int s = blockIdx.x * 4 + w;
  if (s >= n) return;
  float v[16];
  const int per = (H + 63) / 64;
#pragma unroll
  for (int i = 0; i < 16; ++i) v[i] = 0.f;
  const int t0 = seq_start[s], t1 = seq_start[s + 1];
  for (int t = t0; t < t1; ++t) {
    const float* row = x + (long)t * H;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int c = i * 64 + l;
      if (i < per && c < H) v[i] += row[c];
    }
  }
  const float den = fmaxf((float)(t1 - t0), 1e-9f);
  float sq = 0.f;
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    v[i] /= den;
    sq += v[i] * v[i];
  }
  const float nrm = sqrtf(wave_sum(sq));
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int c = i * 64 + l;
    if (i < per && c < H) out[(long)s * H + c] = v[i] / nrm;
  }
}

// paired[i] = <a[i], b[i]>
__global__ __launch_bounds__(256) void paired_dot_kernel(const float* __restrict__ a,
                                                         const float* __restrict__ b,
                                                         const int* __restrict__ label,
                                                         float* __restrict__ out, int N, int D) {
  const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
  const int i = blockIdx.x * 4 + w;
  if (i >= N) return;
  const float* x = a + (long)i * D;
  const float* y = b + (long)(label ? label[i] : i) * D;
  float s = 0.f;
  for (int c = l; c < D; c += 64) s += x[c] * y[c];
  s = wave_sum(s);
  if (l == 0) out[i] = s;
}

// Fused cosine top-k on the f32-input MFMA: sim^T tile [16 classes x 16 preds] per wave, the N x C matrix is
// never written.  Lane = one prediction column; it sees classes 4g..4g+3 of every 16-class tile and keeps
// a sorted top-k list in registers; the 4 lists of a prediction are merged through LDS at the end.
constexpr int TOPK_MAX = 16;
constexpr int COS_KMAX = 96;  // D <= 1536 -> chunks of 16 floats per lane... (D/16 k-groups of 4 MFMAs)

template <int KG>  // KG = D / 16: number of 16-float groups per row
__global__ __launch_bounds__(256) void cosine_topk_kernel(const float* __restrict__ P,
                                                          const float* __restrict__ Cl, int N, int C,
                                                          int k, float* __restrict__ top_val,
                                                          int* __restrict__ top_idx) {
  __shared__ float lv[4][16][4][TOPK_MAX];
  __shared__ int li[4][16][4][TOPK_MAX];
  constexpr int D = KG * 16;
  const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
  const int fr = l & 15, g = l >> 4;
  const int p0 = blockIdx.x * 64 + w * 16;
  const int prow = min(p0 + fr, N - 1);
  // B operand: lane (pred fr, group g) holds chunk 4s+g of its row: 4 floats per s
  f32x4 pf[KG];
#pragma unroll
  for (int s = 0; s < KG; ++s) pf[s] = *(const f32x4*)(P + (long)prow * D + (4 * s + g) * 4);

  float bv[TOPK_MAX];
  int bi[TOPK_MAX];
#pragma unroll
  for (int j = 0; j < TOPK_MAX; ++j) {
    bv[j] = -INFINITY;
    bi[j] = 0x7fffffff;
  }

  for (int c0 = 0; c0 < C; c0 += 16) {
    const int crow = min(c0 + fr, C - 1);
    const float* cr = Cl + (long)crow * D;
    f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < KG; ++s) {
      const f32x4 cf = *(const f32x4*)(cr + (4 * s + g) * 4);
#pragma unroll
      for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(cf[e], pf[s][e], acc, 0, 0, 0);
    }
    // acc[r] = sim[class c0 + 4g + r][pred fr]
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int ci = c0 + 4 * g + r;
      float v = acc[r];
      if (ci >= C) continue;
      if (v > bv[TOPK_MAX - 1] || (v == bv[TOPK_MAX - 1] && ci < bi[TOPK_MAX - 1])) {
        // insertion into the sorted list (descending value, ascending index on ties)
        int cidx = ci;
#pragma unroll
        for (int j = 0; j < TOPK_MAX; ++j) {
          const bool better = (v > bv[j]) || (v == bv[j] && cidx < bi[j]);
          const float tv = bv[j];
          const int ti = bi[j];
          if (better) {
            bv[j] = v;
            bi[j] = cidx;
            v = tv;
            cidx = ti;
          }
        }
      }
    }
  }
#pragma unroll
  for (int j = 0; j < TOPK_MAX; ++j) {
    lv[w][fr][g][j] = bv[j];
    li[w][fr][g][j] = bi[j];
  }
  __syncthreads();
  if (l < 16 && p0 + l < N) {
    int head[4] = {0, 0, 0, 0};
    for (int j = 0; j < k; ++j) {
      int best = 0;
      float v = -INFINITY;
      int ix = 0x7fffffff;
      bool found = false;
      for (int q = 0; q < 4; ++q) {
        if (head[q] >= TOPK_MAX) continue;
        const float cv = lv[w][l][q][head[q]];
        const int cx = li[w][l][q][head[q]];
        if (!found || cv > v || (cv == v && cx < ix)) {
          v = cv;
          ix = cx;
          best = q;
          found = true;
        }
      }
      head[best]++;
      top_val[(long)(p0 + l) * k + j] = v;
      top_idx[(long)(p0 + l) * k + j] = (ix == 0x7fffffff) ? -1 : ix;
    }
  }
}

}  // namespace

extern "C" {

size_t owc_bert_workspace_bytes(const owc_bert_weights* w, int n, int L) {
  if (!w || n <= 0 || L <= 0) return 0;
  const size_t T = (size_t)n * L, H = (size_t)w->hidden;
  return align256(T * H * 4) * 2 + align256(T * 3 * H * 4) + align256(T * (size_t)w->inter * 4) + 1024;
}

int owc_bert_embed(owc_ctx* ctx, const owc_bert_weights* w, const int32_t* ids, const int32_t* mask,
                   int n, int L, float* out, void* workspace, size_t ws_bytes, void* stream) {
  if (!ctx || !w || !ids || !mask || !out || !workspace) return OWC_ERR_ARG;
  const int H = w->hidden, NH = w->n_heads, I = w->inter;
  const int HD = NH > 0 ? H / NH : 0;
  if (n <= 0 || L <= 0 || L + w->pos_offset > w->max_pos || H > 1024 || (H % NH) != 0 || (HD != 32 && HD != 64))
    OWC_FAIL(ctx, OWC_ERR_SHAPE, "owc_bert_embed: unsupported shape (head_dim must be 32 or 64, hidden <= 1024)");
  if (w->rel_bias && L > w->rel_span) OWC_FAIL(ctx, OWC_ERR_SHAPE, "owc_bert_embed: sequence longer than the relative-bias table");
  if (ws_bytes < owc_bert_workspace_bytes(w, n, L)) OWC_FAIL(ctx, OWC_ERR_WORKSPACE, "owc_bert_embed: workspace too small");
  hipStream_t st = (hipStream_t)stream;
  const int T = n * L;
  char* p = (char*)workspace;
  float* x = (float*)p;
  p += align256((size_t)T * H * 4);
  float* cx = (float*)p;
  p += align256((size_t)T * H * 4);
  float* qkv = (float*)p;
  p += align256((size_t)T * 3 * H * 4);
  float* ff = (float*)p;
  // keys resident in LDS at a time: the whole sequence when it fits 128 KiB, else chunks of 256 (head_dim 32: 512) keys
  const int chunk_cap = (128 * 1024) / (HD * 4 * 2 + 4);
  const int chunk = L <= chunk_cap ? L : (HD > 32 ? 256 : 512);
  const size_t attn_lds = (size_t)chunk * HD * 4 * 2 + (size_t)chunk * 4;
  static bool attr_set = false;
  if (!attr_set) {
    if (hipFuncSetAttribute((const void*)bert_attn_kernel<32>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess ||
        hipFuncSetAttribute((const void*)bert_attn_kernel<64>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
      return OWC_ERR_HIP;
    attr_set = true;
  }
  hipLaunchKernelGGL(bert_embed_ln_kernel, dim3((T + 3) / 4), dim3(256), 0, st, ids, (const int*)nullptr, w->word_emb, w->pos_emb,
                     w->type_emb, w->emb_ln_w, w->emb_ln_b, x, T, L, H, w->ln_eps, w->pos_offset);
  const float scale = 1.0f / sqrtf((float)HD);
  for (int i = 0; i < w->n_layers; ++i) {
    const owc_bert_layer& Ly = w->layers[i];
    OWC_TRY(owc_launch_gemm_f32_bert(x, H, Ly.qkv_w, H, Ly.qkv_b, nullptr, 0, qkv, 3 * H, T, 3 * H, H,
                                OWC_EPI_NONE, ctx->zeros, st));
    if (HD == 32)
      hipLaunchKernelGGL(bert_attn_kernel<32>, dim3(n * NH), dim3(64), attn_lds, st, qkv, mask, cx, L, H, NH, scale, w->rel_bias, w->rel_span, chunk);
    else
      hipLaunchKernelGGL(bert_attn_kernel<64>, dim3(n * NH), dim3(64), attn_lds, st, qkv, mask, cx, L, H, NH, scale, w->rel_bias, w->rel_span, chunk);
    // x = LN(dense(ctx) + x)
    OWC_TRY(owc_launch_gemm_f32_bert(cx, H, Ly.o_w, H, Ly.o_b, x, H, x, H, T, H, H, OWC_EPI_RESIDUAL, ctx->zeros, st));
    hipLaunchKernelGGL(ln_f32_kernel, dim3((T + 3) / 4), dim3(256), 0, st, x, Ly.ln1_w, Ly.ln1_b, T, H, w->ln_eps);
    // x = LN(dense(gelu(dense(x))) + x)
    OWC_TRY(owc_launch_gemm_f32_bert(x, H, Ly.fc1_w, H, Ly.fc1_b, nullptr, 0, ff, I, T, I, H, OWC_EPI_GELU_ERF,
                                ctx->zeros, st));
    OWC_TRY(owc_launch_gemm_f32_bert(ff, I, Ly.fc2_w, I, Ly.fc2_b, x, H, x, H, T, H, I, OWC_EPI_RESIDUAL, ctx->zeros, st));
    hipLaunchKernelGGL(ln_f32_kernel, dim3((T + 3) / 4), dim3(256), 0, st, x, Ly.ln2_w, Ly.ln2_b, T, H, w->ln_eps);
  }
  hipLaunchKernelGGL(pool_norm_kernel, dim3((n + 3) / 4), dim3(256), 0, st, x, mask, out, n, L, H);
  if (hipGetLastError() != hipSuccess) OWC_FAIL(ctx, OWC_ERR_HIP, "owc_bert_embed: launch failure");
  return OWC_OK;
}

size_t owc_bert_packed_workspace_bytes(const owc_bert_weights* w, int T) {
  if (!w || T <= 0) return 0;
  const size_t t = (size_t)T, H = (size_t)w->hidden;
  return align256(t * H * 4) * 2 + align256(t * 3 * H * 4) + align256(t * (size_t)w->inter * 4) + 1024;
}

int owc_bert_embed_packed(owc_ctx* ctx, const owc_bert_weights* w, const int32_t* tok_ids, const int32_t* tok_pos,
                          const int32_t* seq_start, int n, int T, int max_len, float* out, void* workspace, size_t ws_bytes,
                          void* stream) {
  if (!ctx || !w || !tok_ids || !tok_pos || !seq_start || !out || !workspace) return OWC_ERR_ARG;
  const int H = w->hidden, NH = w->n_heads, I = w->inter;
  const int HD = NH > 0 ? H / NH : 0;
  if (n <= 0 || T <= 0 || max_len <= 0 || max_len + w->pos_offset > w->max_pos || H > 1024 || (H % NH) != 0 || (HD != 32 && HD != 64))
    OWC_FAIL(ctx, OWC_ERR_SHAPE, "owc_bert_embed_packed: unsupported shape (head_dim must be 32 or 64, hidden <= 1024)");
  if (w->rel_bias && max_len > w->rel_span) OWC_FAIL(ctx, OWC_ERR_SHAPE, "owc_bert_embed_packed: sequence longer than the relative-bias table");
  if (ws_bytes < owc_bert_packed_workspace_bytes(w, T)) OWC_FAIL(ctx, OWC_ERR_WORKSPACE, "owc_bert_embed_packed: workspace too small");
  hipStream_t st = (hipStream_t)stream;
  char* p = (char*)workspace;
  float* x = (float*)p;
  p += align256((size_t)T * H * 4);
  float* cx = (float*)p;
  p += align256((size_t)T * H * 4);
  float* qkv = (float*)p;
  p += align256((size_t)T * 3 * H * 4);
  float* ff = (float*)p;
  const int chunk_cap = (128 * 1024) / (HD * 4 * 2);
  const int chunk = max_len <= chunk_cap ? max_len : (HD > 32 ? 256 : 512);
  const size_t attn_lds = (size_t)chunk * HD * 4 * 2;
  static bool attr_set = false;
  if (!attr_set) {
    if (hipFuncSetAttribute((const void*)bert_attn_packed_kernel<32>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess ||
        hipFuncSetAttribute((const void*)bert_attn_packed_kernel<64>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
      return OWC_ERR_HIP;
    attr_set = true;
  }
  hipLaunchKernelGGL(bert_embed_ln_kernel, dim3((T + 3) / 4), dim3(256), 0, st, tok_ids, tok_pos, w->word_emb, w->pos_emb,
                     w->type_emb, w->emb_ln_w, w->emb_ln_b, x, T, 1, H, w->ln_eps, w->pos_offset);
  const float scale = 1.0f / sqrtf((float)HD);
  for (int i = 0; i < w->n_layers; ++i) {
    const owc_bert_layer& Ly = w->layers[i];
    OWC_TRY(owc_launch_gemm_f32_bert(x, H, Ly.qkv_w, H, Ly.qkv_b, nullptr, 0, qkv, 3 * H, T, 3 * H, H,
                                OWC_EPI_NONE, ctx->zeros, st));
    if (HD == 32)
      hipLaunchKernelGGL(bert_attn_packed_kernel<32>, dim3(n * NH), dim3(64), attn_lds, st, qkv, seq_start, cx, H, NH, scale, w->rel_bias, w->rel_span, chunk);
    else
      hipLaunchKernelGGL(bert_attn_packed_kernel<64>, dim3(n * NH), dim3(64), attn_lds, st, qkv, seq_start, cx, H, NH, scale, w->rel_bias, w->rel_span, chunk);
    OWC_TRY(owc_launch_gemm_f32_bert(cx, H, Ly.o_w, H, Ly.o_b, x, H, x, H, T, H, H, OWC_EPI_RESIDUAL, ctx->zeros, st));
    hipLaunchKernelGGL(ln_f32_kernel, dim3((T + 3) / 4), dim3(256), 0, st, x, Ly.ln1_w, Ly.ln1_b, T, H, w->ln_eps);
    OWC_TRY(owc_launch_gemm_f32_bert(x, H, Ly.fc1_w, H, Ly.fc1_b, nullptr, 0, ff, I, T, I, H, OWC_EPI_GELU_ERF,
                                ctx->zeros, st));
    OWC_TRY(owc_launch_gemm_f32_bert(ff, I, Ly.fc2_w, I, Ly.fc2_b, x, H, x, H, T, H, I, OWC_EPI_RESIDUAL, ctx->zeros, st));
    hipLaunchKernelGGL(ln_f32_kernel, dim3((T + 3) / 4), dim3(256), 0, st, x, Ly.ln2_w, Ly.ln2_b, T, H, w->ln_eps);
  }
  hipLaunchKernelGGL(pool_norm_packed_kernel, dim3((n + 3) / 4), dim3(256), 0, st, x, seq_start, out, n, H);
  if (hipGetLastError() != hipSuccess) OWC_FAIL(ctx, OWC_ERR_HIP, "owc_bert_embed_packed: launch failure");
  return OWC_OK;
}

int owc_paired_dot(owc_ctx* ctx, const float* a, const float* b, int N, int D, float* out, void* stream) {
  if (!ctx || !a || !b || !out || N <= 0 || D <= 0) return OWC_ERR_ARG;
  hipLaunchKernelGGL(paired_dot_kernel, dim3((N + 3) / 4), dim3(256), 0, (hipStream_t)stream, a, b,
                     (const int*)nullptr, out, N, D);
  return hipGetLastError() == hipSuccess ? OWC_OK : OWC_ERR_HIP;
}

int owc_cosine_topk(owc_ctx* ctx, const float* preds, const float* classes, const int32_t* label, int N,
                    int C, int D, int k, float* top_val, int32_t* top_idx, float* paired, void* stream) {
  if (!ctx || !preds || !classes || N <= 0 || C <= 0) return OWC_ERR_ARG;
  hipStream_t st = (hipStream_t)stream;
  if (paired) {
    if (!label) OWC_FAIL(ctx, OWC_ERR_ARG, "owc_cosine_topk: paired output needs labels");
    hipLaunchKernelGGL(paired_dot_kernel, dim3((N + 3) / 4), dim3(256), 0, st, preds, classes, label, paired, N, D);
  }
  if (top_val || top_idx) {
    if (!top_val || !top_idx || k <= 0 || k > TOPK_MAX) OWC_FAIL(ctx, OWC_ERR_ARG, "owc_cosine_topk: 1 <= k <= 16 with both outputs");
    if (D != 384 && D != 768 && D != 64) OWC_FAIL(ctx, OWC_ERR_SHAPE, "owc_cosine_topk: D must be 384, 768 or 64");
    const dim3 grid((N + 63) / 64);
    const int prof = owc_gemm_profile_begin(2.0 * (double)N * (double)C * (double)D, OWC_PROF_COSINE_TOPK, st);  // after validation: never unmatched
    switch (D) {
      case 384: hipLaunchKernelGGL(cosine_topk_kernel<24>, grid, dim3(256), 0, st, preds, classes, N, C, k, top_val, top_idx); break;
      case 768: hipLaunchKernelGGL(cosine_topk_kernel<48>, grid, dim3(256), 0, st, preds, classes, N, C, k, top_val, top_idx); break;
      case 64: hipLaunchKernelGGL(cosine_topk_kernel<4>, grid, dim3(256), 0, st, preds, classes, N, C, k, top_val, top_idx); break;
      default: break;   // unreachable (validated above)
    }
    owc_gemm_profile_end(prof, st);
  }
  if (hipGetLastError() != hipSuccess) OWC_FAIL(ctx, OWC_ERR_HIP, "owc_cosine_topk: launch failure");
  return OWC_OK;
}

}  // extern "C"
