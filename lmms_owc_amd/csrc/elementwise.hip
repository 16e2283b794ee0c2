// HBM-bound glue kernels of the Qwen2-VL path: norms, rotary embeddings, KV-cache write, token
// embedding gather, greedy argmax, pixel patchify.  All loads/stores are 8/16-byte vectors; every
// kernel rounds to bf16 exactly where the bf16 torch module it replaces rounds.
#include "owc_internal.h"

namespace {

// ------------------------------------------------------------------------------------------
// LayerNorm / RMSNorm: one wave per row, row cached in registers (bf16x8 chunks), fp32 stats.
//   LayerNorm  = torch.nn.LayerNorm (vision blocks, merger ln_q: HF modeling_qwen2_vl.py:428-429, :281)
//   RMSNorm    = Qwen2VLRMSNorm (HF modeling_qwen2_vl.py:105-110): x*rsqrt(var+eps) -> bf16 -> *w -> bf16
// ------------------------------------------------------------------------------------------
constexpr int NORM_MAXC = 16;  // chunks of 8 per lane -> rows up to 8192

// QUANT (fp8 decoder, RMS only): the normalised bf16 row never goes to memory; it is quantised per token exactly as
// owc_quantize_rows_fp8 would quantise it (s = max|y| / 448, q = rne_e4m3(y / s)) and Y / ldy address the e4m3 codes.
template <bool RMS, int NORM_C, bool QUANT = false>  // NORM_C = chunks per lane this instantiation covers (d <= 512 * NORM_C)
__global__ __launch_bounds__(256) void norm_kernel(const bf16_t* __restrict__ X, long ldx,
                                                   const bf16_t* __restrict__ Wt,
                                                   const bf16_t* __restrict__ Bs,
                                                   bf16_t* __restrict__ Y, long ldy, int rows, int d,
                                                   float eps, const int* __restrict__ row_index,
                                                   float* __restrict__ qscale = nullptr) {
  const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + w;
  if (row >= rows) return;
  const long src_row = row_index ? (long)row_index[row] : (long)row;
  const bf16_t* x = X + src_row * ldx;
  const int nch = d >> 3;
  bf16x8 c[NORM_C];
  float sum = 0.f, sq = 0.f;
#pragma unroll
  for (int i = 0; i < NORM_C; ++i) {
    const int ch = i * 64 + l;
    if (ch < nch) {
      c[i] = *(const bf16x8*)(x + ch * 8);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float v = bf2f(c[i][e]);
        sum += v;
        sq = __builtin_fmaf(v, v, sq);   // pinned: the same fma chain as owc_rms_rstd (owc_common.h)
      }
    }
  }
  sum = wave_sum(sum);
  sq = wave_sum(sq);
  const float inv_d = 1.0f / (float)d;
  float mean = 0.f, rstd;
  if (RMS) {
    rstd = rsqrtf(__builtin_fmaf(sq, inv_d, eps));
  } else {
    mean = sum * inv_d;
    float var = 0.f;
#pragma unroll
    for (int i = 0; i < NORM_C; ++i) {
      const int ch = i * 64 + l;
      if (ch < nch) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float v = bf2f(c[i][e]) - mean;
          var += v * v;
        }
      }
    }
    var = wave_sum(var) * inv_d;
    rstd = rsqrtf(var + eps);
  }
  if constexpr (QUANT) {
    float amax = 0.f;
#pragma unroll
    for (int i = 0; i < NORM_C; ++i) {
      const int ch = i * 64 + l;
      if (ch < nch) {
        const bf16x8 wv = *(const bf16x8*)(Wt + ch * 8);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          c[i][e] = f2bf(bf2f(wv[e]) * rbf(bf2f(c[i][e]) * rstd));  // the bf16 value the un-fused kernel would store
          amax = fmaxf(amax, fabsf(bf2f(c[i][e])));
        }
      }
    }
    amax = wave_max(amax);
    const float scale = amax > 0.f ? amax / 448.0f : 1.0f;
    if (l == 0) qscale[row] = scale;
    uint8_t* q = (uint8_t*)Y + (long)row * ldy;
#pragma unroll
    for (int i = 0; i < NORM_C; ++i) {
      const int ch = i * 64 + l;
      if (ch < nch) {
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = fminf(fmaxf(bf2f(c[i][e]) / scale, -448.0f), 448.0f);
        int lo = 0, hi = 0;
        lo = __builtin_amdgcn_cvt_pk_fp8_f32(v[0], v[1], lo, false);
        lo = __builtin_amdgcn_cvt_pk_fp8_f32(v[2], v[3], lo, true);
        hi = __builtin_amdgcn_cvt_pk_fp8_f32(v[4], v[5], hi, false);
        hi = __builtin_amdgcn_cvt_pk_fp8_f32(v[6], v[7], hi, true);
        *(int2*)(q + ch * 8) = make_int2(lo, hi);
      }
    }
    return;
  }
  bf16_t* y = Y + (long)row * ldy;
#pragma unroll
  for (int i = 0; i < NORM_C; ++i) {
    const int ch = i * 64 + l;
    if (ch < nch) {
      const bf16x8 wv = *(const bf16x8*)(Wt + ch * 8);
      bf16x8 o;
      if (RMS) {
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = f2bf(bf2f(wv[e]) * rbf(bf2f(c[i][e]) * rstd));
      } else {
        const bf16x8 bv = *(const bf16x8*)(Bs + ch * 8);
#pragma unroll
        for (int e = 0; e < 8; ++e)
          o[e] = f2bf((bf2f(c[i][e]) - mean) * rstd * bf2f(wv[e]) + bf2f(bv[e]));
      }
      *(bf16x8*)(y + ch * 8) = o;
    }
  }
}

// ------------------------------------------------------------------------------------------
// Rotary tables (built once per model on the device).
//   vision : freq[j] = 10000^(-2j/40), j < 20        (HF :239-248 VisionRotaryEmbedding(head_dim/2))
//   decoder: freq[i] = theta^(-2i/128), i < 64, table entries rounded to bf16 (HF :156-170 casts
//            cos/sin to the activation dtype before apply_multimodal_rotary_pos_emb)
// ------------------------------------------------------------------------------------------
__global__ void rope_table_kernel(float* __restrict__ cos_t, float* __restrict__ sin_t, int n_pos,
                                  int n_freq, int dim, float theta, int round_bf16) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_pos * n_freq) return;
  const int p = i / n_freq, j = i % n_freq;
  const float inv = (float)(1.0 / pow((double)theta, (double)(2 * j) / (double)dim));
  const float ang = (float)p * inv;
  float c = cosf(ang), s = sinf(ang);
  if (round_bf16) {
    c = rbf(c);
    s = rbf(s);
  }
  cos_t[i] = c;
  sin_t[i] = s;
}

// Vision 2-D RoPE, in place on the fused qkv buffer [T, 3, H, 80] (q and k parts only).
// HF apply_rotary_pos_emb_vision (:225-236): fp32 math, one rounding.  pos_hw[t] = {h, w}.
__global__ __launch_bounds__(256) void vision_rope_kernel(bf16_t* __restrict__ qkv, long ld,
                                                          const int* __restrict__ pos_hw,
                                                          const float* __restrict__ cos_t,
                                                          const float* __restrict__ sin_t, int T,
                                                          int n_heads, int hd) {
  // one thread handles 4 rotation pairs (i..i+3, i+hd/2..) of one (token, q|k, head)
  const int half = hd >> 1, quads = half >> 2, quarter = half >> 1;
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long total = (long)T * 2 * n_heads * quads;
  if (idx >= total) return;
  const int qd = idx % quads;
  long r = idx / quads;
  const int h = r % n_heads;
  r /= n_heads;
  const int which = r & 1;
  const int t = r >> 1;
  const int i0 = qd * 4;
  bf16_t* p = qkv + (long)t * ld + (long)which * n_heads * hd + (long)h * hd;
  const bf16x4 a = *(const bf16x4*)(p + i0);
  const bf16x4 b = *(const bf16x4*)(p + i0 + half);
  const int ph = pos_hw[2 * t], pw = pos_hw[2 * t + 1];
  bf16x4 oa, ob;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const int i = i0 + e;
    const int ti = (i < quarter) ? (ph * quarter + i) : (pw * quarter + (i - quarter));
    const float c = cos_t[ti], s = sin_t[ti];
    const float x1 = bf2f(a[e]), x2 = bf2f(b[e]);
    oa[e] = f2bf(x1 * c - x2 * s);
    ob[e] = f2bf(x2 * c + x1 * s);
  }
  *(bf16x4*)(p + i0) = oa;
  *(bf16x4*)(p + i0 + half) = ob;
}

// Decoder M-RoPE (HF apply_multimodal_rotary_pos_emb :180-222) on the fused qkv buffer
// [T, (H + 2 KV) * 128]: q rotated in place, rotated k and v written into the KV cache
// cache[(slot*KV + kvh) * s_max + idx][128].  bf16 arithmetic with torch's rounding sequence:
// bf16(bf16(x*cos) + bf16(rot*sin)), cos/sin already bf16-rounded in the table.
__global__ __launch_bounds__(256) void mrope_kv_kernel(
    bf16_t* __restrict__ qkv, long ld, const int* __restrict__ pos3, long pos_stride,
    const float* __restrict__ cos_t, const float* __restrict__ sin_t, bf16_t* __restrict__ kc,
    bf16_t* __restrict__ vc, const int* __restrict__ tok_slot, const int* __restrict__ tok_idx,
    int T, int n_q, int n_kv, int s_max, int sec0, int sec1, int bcast_first, int bcast_n) {
  // thread = (token, head among q+2kv, group of 8 dims in [0,64)): two 16-byte accesses per half
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const int n_h = n_q + 2 * n_kv;
  const long total = (long)T * n_h * 8;
  if (idx >= total) return;
  const int od = idx & 7;
  long r = idx >> 3;
  const int h = r % n_h;
  const int t = r / n_h;
  const int i0 = od * 8;
  bf16_t* p = qkv + (long)t * ld + (long)h * 128;
  const bf16x8 a = *(const bf16x8*)(p + i0);
  const bf16x8 b = *(const bf16x8*)(p + i0 + 64);
  // tok_slot < 0: a shared-prefix token, its K/V row goes to every slot of [bcast_first, bcast_first + bcast_n)
  const int s0 = tok_slot[t] < 0 ? bcast_first : tok_slot[t];
  const int ns = tok_slot[t] < 0 ? bcast_n : 1;
  if (h >= n_q + n_kv) {  // v: straight copy into the cache
    const int kvh = h - n_q - n_kv;
    for (int sl = s0; sl < s0 + ns; ++sl) {
      bf16_t* dst = vc + (((long)sl * n_kv + kvh) * s_max + tok_idx[t]) * 128;
      *(bf16x8*)(dst + i0) = a;
      *(bf16x8*)(dst + i0 + 64) = b;
    }
    return;
  }
  bf16x8 oa, ob;
  if (((sec0 | sec1) & 7) == 0) {
    // the section boundaries are multiples of 8 (Qwen2-VL: [16, 24, 24]): the thread's 8 dims share one position stream, so the
    // table rows come in as four 16-byte loads instead of sixteen scalar gathers
    const int stream = (i0 < sec0) ? 0 : ((i0 < sec0 + sec1) ? 1 : 2);
    const long row = (long)pos3[stream * pos_stride + t] * 64 + i0;
    const f32x4 c0 = *(const f32x4*)(cos_t + row), c1 = *(const f32x4*)(cos_t + row + 4);
    const f32x4 s0v = *(const f32x4*)(sin_t + row), s1v = *(const f32x4*)(sin_t + row + 4);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float c = e < 4 ? c0[e & 3] : c1[e & 3], s = e < 4 ? s0v[e & 3] : s1v[e & 3];
      const float x1 = bf2f(a[e]), x2 = bf2f(b[e]);
      oa[e] = f2bf(rbf(x1 * c) + rbf(-x2 * s));
      ob[e] = f2bf(rbf(x2 * c) + rbf(x1 * s));
    }
  } else {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int i = i0 + e;
      const int stream = (i < sec0) ? 0 : ((i < sec0 + sec1) ? 1 : 2);
      const int pos = pos3[stream * pos_stride + t];
      const float c = cos_t[pos * 64 + i], s = sin_t[pos * 64 + i];
      const float x1 = bf2f(a[e]), x2 = bf2f(b[e]);
      oa[e] = f2bf(rbf(x1 * c) + rbf(-x2 * s));
      ob[e] = f2bf(rbf(x2 * c) + rbf(x1 * s));
    }
  }
  if (h < n_q) {
    *(bf16x8*)(p + i0) = oa;
    *(bf16x8*)(p + i0 + 64) = ob;
  } else {
    const int kvh = h - n_q;
    for (int sl = s0; sl < s0 + ns; ++sl) {
      bf16_t* dst = kc + (((long)sl * n_kv + kvh) * s_max + tok_idx[t]) * 128;
      *(bf16x8*)(dst + i0) = oa;
      *(bf16x8*)(dst + i0 + 64) = ob;
    }
  }
}

// Token embedding gather with image-embedding scatter (HF Qwen2VLModel.forward :1160-1168):
// out[t] = ids[t] == image_token ? img[img_index[t]] : table[ids[t]]
__global__ __launch_bounds__(256) void embed_kernel(const int* __restrict__ ids,
                                                    const int* __restrict__ img_index,
                                                    const bf16_t* __restrict__ table,
                                                    const bf16_t* __restrict__ img,
                                                    bf16_t* __restrict__ out, int T, int d) {
  const int nch = d >> 3;
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long)T * nch) return;
  const int t = idx / nch, ch = idx % nch;
  const int ii = img_index ? img_index[t] : -1;
  const bf16_t* src = (ii >= 0) ? (img + (long)ii * d) : (table + (long)ids[t] * d);
  *(bf16x8*)(out + (long)t * d + ch * 8) = *(const bf16x8*)(src + ch * 8);
}

// Greedy argmax over bf16 logits (HF generation: logits.float().argmax(-1); lowest index on ties).
// One block of 1024 threads per row, four 16-byte loads in flight per thread: a decode step at the reference's batch size is ONE
// row of 152 064 logits, and with 256 threads and one load in flight that row took 55 us (1.4 % of the step) for 300 KB.
constexpr int ARGMAX_T = 1024;
__global__ __launch_bounds__(ARGMAX_T) void argmax_kernel(const bf16_t* __restrict__ logits, long ld,
                                                          int V, int* __restrict__ out) {
  __shared__ float sv[ARGMAX_T / 64];
  __shared__ int si[ARGMAX_T / 64];
  const int row = blockIdx.x;
  const bf16_t* x = logits + (long)row * ld;
  float best = -INFINITY;
  int bi = 0x7fffffff;
  const int nch = V >> 3;
  auto take = [&](const bf16x8& v, int ch) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float f = bf2f(v[e]);
      const int i = ch * 8 + e;
      if (f > best || (f == best && i < bi)) {
        best = f;
        bi = i;
      }
    }
  };
  int ch = threadIdx.x;
  for (; ch + 3 * ARGMAX_T < nch; ch += 4 * ARGMAX_T) {
    const bf16x8 v0 = *(const bf16x8*)(x + (long)ch * 8), v1 = *(const bf16x8*)(x + (long)(ch + ARGMAX_T) * 8);
    const bf16x8 v2 = *(const bf16x8*)(x + (long)(ch + 2 * ARGMAX_T) * 8), v3 = *(const bf16x8*)(x + (long)(ch + 3 * ARGMAX_T) * 8);
    take(v0, ch);
    take(v1, ch + ARGMAX_T);
    take(v2, ch + 2 * ARGMAX_T);
    take(v3, ch + 3 * ARGMAX_T);
  }
  for (; ch < nch; ch += ARGMAX_T) take(*(const bf16x8*)(x + (long)ch * 8), ch);
  for (int i = nch * 8 + threadIdx.x; i < V; i += ARGMAX_T) {
    const float f = bf2f(x[i]);
    if (f > best || (f == best && i < bi)) {
      best = f;
      bi = i;
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float ov = __shfl_xor(best, o, 64);
    const int oi = __shfl_xor(bi, o, 64);
    if (ov > best || (ov == best && oi < bi)) {
      best = ov;
      bi = oi;
    }
  }
  const int w = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) {
    sv[w] = best;
    si[w] = bi;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int k = 1; k < ARGMAX_T / 64; ++k)
      if (sv[k] > best || (sv[k] == best && si[k] < bi)) {
        best = sv[k];
        bi = si[k];
      }
    out[row] = bi;
  }
}

// ------------------------------------------------------------------------------------------------------------------------
// Repetition penalty (round 6).  HF's RepetitionPenaltyLogitsProcessor is in force in the REFERENCE whenever the checkpoint's
// generation_config.json carries `repetition_penalty` != 1 (Qwen2-VL / Qwen2.5-VL instruct checkpoints ship 1.05): the reference
// calls `model.generate(..., do_sample = temperature > 0, temperature, top_p, num_beams, max_new_tokens)`
// (/root/reference/src/models/_qwen2_vl.py:319-329) and HF merges every generation_config field the call does not override, so
// GREEDY decoding is penalised too.  The processor (transformers generation/logits_process.py): the next-token logits go to
// fp32; for every token id that occurs in input_ids - the PROMPT (image placeholders included) and everything generated so far -
// score = score < 0 ? score * p : score / p, once per distinct id; then the warpers / argmax.
// Here: a bitmap of the ids a sequence has seen (one row of `wpr` 32-bit words per KV-cache SLOT, caller-owned, zeroed by the
// caller before the prefill), marked by seen_mark_kernel from the prompt rows at prefill and from the token FED at every decode
// step (under teacher forcing that is the forced token: what HF's input_ids would hold), and an argmax that applies the penalty
// in fp32 to the marked ids on the fly (argmax_penalized_kernel: argmax_kernel + one bitmap word per 32 logits).  For sampled
// requests penalize_rows_kernel writes the penalised values back as bf16 in front of sample_kernel (one rounding HF does not do;
// the greedy path has none).
// ------------------------------------------------------------------------------------------------------------------------
__global__ void seen_mark_kernel(const int* __restrict__ ids, const int* __restrict__ slot, int n, int V, unsigned* __restrict__ seen,
                                 int wpr, int bcast_first, int bcast_n) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n) return;
  const int id = ids[t];
  if (id < 0 || id >= V) return;
  const int sl = slot ? slot[t] : t;
  const unsigned bit = 1u << (id & 31);
  if (sl >= 0) {
    atomicOr(&seen[(long)sl * wpr + (id >> 5)], bit);
  } else {   // a shared-prefix row of the prefill (tok_slot = -1): it belongs to every sequence of the launch
    for (int s2 = bcast_first; s2 < bcast_first + bcast_n; ++s2) atomicOr(&seen[(long)s2 * wpr + (id >> 5)], bit);
  }
}

__device__ __forceinline__ float rep_penalize(float v, float p) { return v < 0.f ? v * p : v / p; }

__global__ __launch_bounds__(ARGMAX_T) void argmax_penalized_kernel(const bf16_t* __restrict__ logits, long ld, int V,
                                                                    const unsigned* __restrict__ seen, int wpr,
                                                                    const int* __restrict__ row_slot, const int* __restrict__ row_index,
                                                                    float penalty, int* __restrict__ out) {
  __shared__ float sv[ARGMAX_T / 64];
  __shared__ int si[ARGMAX_T / 64];
  const int row = blockIdx.x;
  const bf16_t* x = logits + (long)row * ld;
  // the sequence's bitmap row: slot = row_slot[row_index[row]] (prefill: tok_slot of the prompt's last row), row_slot[row] (decode), or row
  const int ri = row_index ? row_index[row] : row;
  const unsigned char* bits = (const unsigned char*)(seen + (long)(row_slot ? row_slot[ri] : ri) * wpr);   // little endian: byte c = ids 8c .. 8c + 7
  float best = -INFINITY;
  int bi = 0x7fffffff;
  const int nch = V >> 3;
  for (int ch = threadIdx.x; ch < nch; ch += ARGMAX_T) {
    const bf16x8 v = *(const bf16x8*)(x + (long)ch * 8);
    const unsigned m = bits[ch];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      float f = bf2f(v[e]);
      if ((m >> e) & 1u) f = rep_penalize(f, penalty);
      const int i = ch * 8 + e;
      if (f > best || (f == best && i < bi)) {
        best = f;
        bi = i;
      }
    }
  }
  for (int i = nch * 8 + threadIdx.x; i < V; i += ARGMAX_T) {
    float f = bf2f(x[i]);
    if ((bits[i >> 3] >> (i & 7)) & 1u) f = rep_penalize(f, penalty);
    if (f > best || (f == best && i < bi)) {
      best = f;
      bi = i;
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float ov = __shfl_xor(best, o, 64);
    const int oi = __shfl_xor(bi, o, 64);
    if (ov > best || (ov == best && oi < bi)) {
      best = ov;
      bi = oi;
    }
  }
  const int w = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) {
    sv[w] = best;
    si[w] = bi;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int k = 1; k < ARGMAX_T / 64; ++k)
      if (sv[k] > best || (sv[k] == best && si[k] < bi)) {
        best = sv[k];
        bi = si[k];
      }
    out[row] = bi;
  }
}

__global__ __launch_bounds__(1024) void penalize_rows_kernel(bf16_t* __restrict__ logits, long ld, int V, const unsigned* __restrict__ seen,
                                                             int wpr, const int* __restrict__ row_slot, const int* __restrict__ row_index,
                                                             float penalty) {
  const int row = blockIdx.x;
  bf16_t* x = logits + (long)row * ld;
  const int ri = row_index ? row_index[row] : row;
  const unsigned* bits = seen + (long)(row_slot ? row_slot[ri] : ri) * wpr;
  for (int wd = threadIdx.x; wd * 32 < V; wd += blockDim.x) {
    unsigned m = bits[wd];
    while (m) {
      const int e = __builtin_ctz(m);
      m &= m - 1;
      const int i = wd * 32 + e;
      if (i < V) x[i] = f2bf(rep_penalize(bf2f(x[i]), penalty));
    }
  }
}

// ------------------------------------------------------------------------------------------------------------------------
// Temperature / top-k / top-p sampling of one token per row (HF GenerationMixin._sample with TemperatureLogitsWarper ->
// TopKLogitsWarper -> TopPLogitsWarper -> softmax -> multinomial; reached from the reference at src/models/_qwen2_vl.py:319-329
// with do_sample = temperature > 0, and _llava_hf.py:365-376).  One block of 1024 threads per row, everything that decides the
// result in INTEGER arithmetic, so a row's token depends on (its logits, seed, stream id, step) only - not on its neighbours, not
// on the order in which threads arrive:
//   * weight of token i: q_i = floor(2^40 * exp2((l_i - max) * log2(e) / T)), a 64-bit integer (sum over 152 k tokens < 2^58);
//   * the kept set is cut by VALUE: bf16 logits have 65 536 possible values, an order-preserving 16-bit key is histogrammed by its
//     high byte (counts + integer mass, LDS atomics on integers are order-independent), then by its low byte inside the bin that
//     holds the cut: top-k keeps every token whose value is >= the k-th largest value (HF's TopKLogitsWarper removes `scores <
//     k-th value`: a run of equal logits at the cut stays WHOLE there too), top-p (on the top-k survivors, as HF applies it) the
//     tokens in front of which the descending cumulative mass is still below top_p.  Where that cut falls inside a run of EQUAL
//     logits HF keeps a prefix of the run in its (unspecified) sort order; this kernel keeps the same NUMBER of them -
//     ceil((top_p Z - mass above the run) / weight) - and takes the LOWEST token ids (round 6: before, the whole run stayed).  So
//     with Qwen2-VL's generation_config (top_k 1, top_p 0.001) exactly one token survives, the lowest id among the maxima: the
//     greedy argmax, bit for bit, ties included;
//   * the draw: Philox4x32-10 keyed by the 64-bit seed, counter (stream id, step, 0, 0) -> r in [0, 2^64);
//     target = floor(r * Z_kept / 2^64); the token is the first one, in INDEX order, whose running kept mass exceeds target.
// ------------------------------------------------------------------------------------------------------------------------
__device__ inline void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1, uint32_t* out) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
    const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1, n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
    c0 = n0;
    c1 = n1;
    c2 = n2;
    c3 = n3;
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  out[0] = c0;
  out[1] = c1;
  out[2] = c2;
  out[3] = c3;
}

constexpr int SAMPLE_T = 1024;
__device__ inline uint32_t sample_key(bf16_t v) {   // order-preserving 16-bit key of a bf16 value (ascending)
  const uint32_t b = (uint32_t)__builtin_bit_cast(unsigned short, v);
  return (b & 0x8000u) ? (0xFFFFu ^ b) : (b | 0x8000u);
}

__global__ __launch_bounds__(SAMPLE_T) void sample_kernel(const bf16_t* __restrict__ logits, long ld, int V, float inv_temp_log2e,
                                                          int top_k, float top_p, uint32_t seed_lo, uint32_t seed_hi,
                                                          const int* __restrict__ row_map, const int* __restrict__ stream_id,
                                                          const int* __restrict__ step_offset, int step,
                                                          const int* __restrict__ step_state, int* __restrict__ out) {
  __shared__ unsigned int cnt[256];
  __shared__ unsigned long long mass[256];
  __shared__ float red[SAMPLE_T / 64];
  __shared__ unsigned long long part[SAMPLE_T];
  __shared__ unsigned long long sh_u64[2];
  __shared__ int sh_i[4];
  __shared__ int part_c[SAMPLE_T];
  const int row = blockIdx.x, tid = threadIdx.x;
  const bf16_t* x = logits + (long)row * ld;
  if (step_state) step = step_state[0];
  // ---- row maximum
  float mx = -INFINITY;
  for (int i = tid; i < V; i += SAMPLE_T) mx = fmaxf(mx, bf2f(x[i]));
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
  if ((tid & 63) == 0) red[tid >> 6] = mx;
  __syncthreads();
  mx = red[0];
  for (int k = 1; k < SAMPLE_T / 64; ++k) mx = fmaxf(mx, red[k]);
  auto weight = [&](bf16_t v) -> unsigned long long {
    const float e = __builtin_amdgcn_exp2f((bf2f(v) - mx) * inv_temp_log2e);   // in [0, 1]
    return (unsigned long long)(e * 1099511627776.0f);                          // 2^40
  };
  // histogram of the tokens whose key lies in [lo_key, 0xFFFF] by byte `shift` (8: high byte, 0: low byte inside high byte `hb`)
  auto histogram = [&](int shift, uint32_t hb, uint32_t lo_key) {
    __syncthreads();
    if (tid < 256) {
      cnt[tid] = 0;
      mass[tid] = 0;
    }
    __syncthreads();
    for (int i = tid; i < V; i += SAMPLE_T) {
      const bf16_t v = x[i];
      const uint32_t k = sample_key(v);
      if (k < lo_key || (shift == 0 && (k >> 8) != hb)) continue;
      const uint32_t b = shift ? (k >> 8) : (k & 0xFF);
      atomicAdd(&cnt[b], 1u);
      atomicAdd(&mass[b], weight(v));
    }
    __syncthreads();
  };
  // ---- top-k: the smallest key such that at least k tokens have a key >= it (top_k <= 0 or >= V: everything)
  uint32_t key_cut = 0;
  if (top_k > 0 && top_k < V) {
    histogram(8, 0, 0);
    if (tid == 0) {
      unsigned int c = 0;
      int b = 255;
      for (; b > 0 && c + cnt[b] < (unsigned)top_k; --b) c += cnt[b];
      sh_i[0] = b;
      sh_i[1] = (int)c;   // tokens in the bins above b
    }
    __syncthreads();
    const uint32_t hb = sh_i[0];
    const unsigned int above = sh_i[1];
    histogram(0, hb, 0);
    if (tid == 0) {
      unsigned int c = above;
      int b = 255;
      for (; b > 0 && c + cnt[b] < (unsigned)top_k; --b) c += cnt[b];
      sh_i[2] = (int)((hb << 8) | (uint32_t)b);
    }
    __syncthreads();
    key_cut = (uint32_t)sh_i[2];
  }
  // ---- top-p on the survivors: descending cumulative mass reaches top_p * Z at which value?
  histogram(8, 0, key_cut);
  if (tid == 0) {
    unsigned long long z = 0;
    for (int b = 0; b < 256; ++b) z += mass[b];
    sh_u64[0] = z;
  }
  __syncthreads();
  unsigned long long Z = sh_u64[0];
  int n_keep_cut = -1;   // how many of the tokens whose value IS the cut value stay (lowest ids first); -1: all of them
  if (top_p > 0.f && top_p < 1.f) {
    const unsigned long long need = (unsigned long long)((double)Z * (double)top_p);
    if (tid == 0) {
      unsigned long long c = 0;
      int b = 255;
      for (; b > 0 && c + mass[b] < need; --b) c += mass[b];
      sh_i[0] = b;
      sh_u64[1] = c;   // mass in the bins above b
    }
    __syncthreads();
    const uint32_t hb = sh_i[0];
    const unsigned long long above = sh_u64[1];
    histogram(0, hb, key_cut);
    if (tid == 0) {
      unsigned long long c = above;
      int b = 255;
      for (; b > 0 && c + mass[b] < need; --b) c += mass[b];
      const uint32_t kp = (hb << 8) | (uint32_t)b;
      sh_i[2] = (int)(kp > key_cut ? kp : key_cut);
      // inside the run of tokens that share the cut value (equal weight q each): token j of the run (j = 0, 1, ...) has c + j q in
      // front of it and stays while that is < need - HF's `cumulative_probs <= 1 - top_p` removal read from the top
      const unsigned long long q = cnt[b] ? mass[b] / cnt[b] : 0;
      unsigned long long nk = cnt[b];
      if (kp >= key_cut && q > 0 && need > c) {
        const unsigned long long want = (need - c + q - 1) / q;
        if (want < nk) nk = want;
      }
      if (nk < 1) nk = 1;
      sh_i[3] = nk < cnt[b] ? (int)nk : -1;
      sh_u64[0] = c + (nk < cnt[b] ? nk * q : mass[b]);   // kept mass (bins above + the kept part of the cut value's run)
    }
    __syncthreads();
    key_cut = (uint32_t)sh_i[2];
    n_keep_cut = sh_i[3];
    Z = sh_u64[0];
    __syncthreads();
  }
  // ---- the draw
  uint32_t rnd[4];
  const int orig = row_map ? row_map[row] : row;   // the sequence's ORIGINAL batch row (row compaction moves it)
  philox4x32_10((uint32_t)(stream_id ? stream_id[orig] : orig), (uint32_t)(step + (step_offset ? step_offset[orig] : 0)), 0u, 0u, seed_lo,
                seed_hi, rnd);
  const unsigned long long r = ((unsigned long long)rnd[0] << 32) | rnd[1];
  const unsigned long long target = (unsigned long long)(((unsigned __int128)r * (unsigned __int128)Z) >> 64);   // < Z
  // ---- first token, in index order, whose running kept mass exceeds target: contiguous chunk per thread, block scan, rescan
  const int chunk = (V + SAMPLE_T - 1) / SAMPLE_T;
  const int i0 = tid * chunk, i1 = min(V, i0 + chunk);
  int rank0 = 0;   // tokens of the cut value's run in front of this thread's chunk (only when the run is cut: n_keep_cut >= 0)
  if (n_keep_cut >= 0) {
    int c = 0;
    for (int i = i0; i < i1; ++i) c += sample_key(x[i]) == key_cut;
    part_c[tid] = c;
    __syncthreads();
    if (tid == 0) {
      int run = 0;
      for (int t = 0; t < SAMPLE_T; ++t) {
        const int c_t = part_c[t];
        part_c[t] = run;
        run += c_t;
      }
    }
    __syncthreads();
    rank0 = part_c[tid];
  }
  unsigned long long mine = 0;
  {
    int rank = rank0;
    for (int i = i0; i < i1; ++i) {
      const bf16_t v = x[i];
      const uint32_t k = sample_key(v);
      if (k > key_cut || (k == key_cut && (n_keep_cut < 0 || rank++ < n_keep_cut))) mine += weight(v);
    }
  }
  part[tid] = mine;
  __syncthreads();
  if (tid == 0) {
    unsigned long long c = 0;
    int t = 0;
    for (; t < SAMPLE_T - 1 && c + part[t] <= target; ++t) c += part[t];
    sh_i[0] = t;
    sh_u64[1] = c;
  }
  __syncthreads();
  if (tid == sh_i[0]) {
    unsigned long long c = sh_u64[1];
    int pick = -1, last_kept = -1, rank = rank0;
    for (int i = i0; i < i1; ++i) {
      const bf16_t v = x[i];
      const uint32_t k = sample_key(v);
      if (k < key_cut || (k == key_cut && n_keep_cut >= 0 && rank++ >= n_keep_cut)) continue;
      last_kept = i;
      c += weight(v);
      if (c > target) {
        pick = i;
        break;
      }
    }
    out[row] = pick >= 0 ? pick : last_kept;   // (pick < 0 cannot happen: target < Z = the sum of exactly these weights)
  }
}

// Beam search, the arithmetic of one step (HF GenerationMixin._beam_search: `log_softmax(logits.float())`, then `torch.topk` over
// beams x vocabulary - of which a beam's own best K suffice, K = 2 x num_beams): per row of bf16 logits its log-sum-exp (fp32:
// max + log(sum exp(x - max)), fixed summation order) and its K largest logits with their token ids, descending, the LOWEST id
// first among equal bf16 values.  One block of 1024 threads per row; the K picks are K sweeps of the row (300 KB, L2-resident
// after the first): "the best (value, id) strictly after the previous pick".  A path the reference's task configs never take
// (all `num_beams: 1`): simple and deterministic beats fast here.
constexpr int BEAM_T = 1024;
__global__ __launch_bounds__(BEAM_T) void beam_candidates_kernel(const bf16_t* __restrict__ logits, long ld, int V, int K,
                                                                 float* __restrict__ logz, float* __restrict__ top_val,
                                                                 int* __restrict__ top_idx) {
  __shared__ float sv[BEAM_T / 64];
  __shared__ int si[BEAM_T / 64];
  __shared__ float bcast_v;
  __shared__ int bcast_i;
  const int row = blockIdx.x, tid = threadIdx.x, w = tid >> 6;
  const bf16_t* x = logits + (long)row * ld;
  auto block_best = [&](float v, int i) {   // (max value, min id among equals) over the block -> bcast_v / bcast_i
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const float ov = __shfl_xor(v, o, 64);
      const int oi = __shfl_xor(i, o, 64);
      if (ov > v || (ov == v && oi < i)) {
        v = ov;
        i = oi;
      }
    }
    __syncthreads();   // (the previous round's broadcast has been read by everybody)
    if ((tid & 63) == 0) {
      sv[w] = v;
      si[w] = i;
    }
    __syncthreads();
    if (tid == 0) {
      for (int k = 1; k < BEAM_T / 64; ++k)
        if (sv[k] > v || (sv[k] == v && si[k] < i)) {
          v = sv[k];
          i = si[k];
        }
      bcast_v = v;
      bcast_i = i;
    }
    __syncthreads();
  };
  // the K largest: sweep k finds the best (value, id) lexicographically after the previous pick
  float pv = INFINITY;
  int pi = -1;
  float row_max = 0.f;
  for (int k = 0; k < K; ++k) {
    float best = -INFINITY;
    int bi = 0x7fffffff;
    for (int i = tid; i < V; i += BEAM_T) {
      const float f = bf2f(x[i]);
      if ((f < pv || (f == pv && i > pi)) && (f > best || (f == best && i < bi))) {
        best = f;
        bi = i;
      }
    }
    block_best(best, bi);
    pv = bcast_v;
    pi = bcast_i;
    if (k == 0) row_max = pv;
    if (tid == 0) {
      top_val[(long)row * K + k] = pv;
      top_idx[(long)row * K + k] = pi;
    }
  }
  // log-sum-exp: per-thread strided sums, then a fixed tree
  float s = 0.f;
  for (int i = tid; i < V; i += BEAM_T) s += __expf(bf2f(x[i]) - row_max);
  s = wave_sum(s);
  __syncthreads();
  if ((tid & 63) == 0) sv[w] = s;
  __syncthreads();
  if (tid == 0) {
    float t = 0.f;
    for (int k = 0; k < BEAM_T / 64; ++k) t += sv[k];
    logz[row] = row_max + __logf(t);
  }
}

// Decode bookkeeping (HF GenerationMixin greedy loop): finished sequences emit pad, EOS marks done.
//   out_row (optional): row of out_tokens (and index into `forced`) that compact row b belongs to - after an EOS-aware
//     row compaction the B live rows are a subset of the original batch (NULL: identity);
//   forced (optional, indexed by ORIGINAL row): teacher forcing - the token FED to the next step (and the one whose EOS ends
//     the sequence) is forced[row] instead of the step's own argmax, which is still what out_tokens receives.
__global__ void decode_update_kernel(int* __restrict__ next_tok, uint8_t* __restrict__ done,
                                     int* __restrict__ out_tokens, int out_stride, int step,
                                     const int* __restrict__ step_state, int B, int eos0, int eos1, int pad,
                                     const int* __restrict__ out_row, const int* __restrict__ forced) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  if (step_state) step = step_state[0];  // graph-replayed decode: the column lives on the device
  const int row = out_row ? out_row[b] : b;
  int t = next_tok[b];
  int feed = forced ? forced[row] : t;
  if (done[b]) t = feed = pad;
  out_tokens[(long)row * out_stride + step] = t;
  if (feed == eos0 || feed == eos1) done[b] = 1;
  next_tok[b] = feed;
}

// EOS-aware row compaction of the decode batch: the per-row state of the `n` surviving rows `live[i]` (ascending indices into
// the CURRENT rows) moves to rows 0..n-1 of a second set of buffers (a gather cannot run in place).  Seven int32 vectors (fed
// token, rope position, cache write index, key count, cache slot, key start, output row) and the done flags.
struct compact_ptrs {
  const int* src[7];
  int* dst[7];
};
__global__ void decode_compact_kernel(compact_ptrs p, const uint8_t* __restrict__ done_src, uint8_t* __restrict__ done_dst,
                                      const int* __restrict__ live, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int r = live[i];
#pragma unroll
  for (int a = 0; a < 7; ++a) p.dst[a][i] = p.src[a][r];
  done_dst[i] = done_src[r];
}

// Graph-replayed decode: advance the per-sequence rope position / cache write index / key count and the output column on
// the device, so the next step is the same launch sequence with the same arguments.
__global__ void decode_advance_kernel(int* __restrict__ pos, int* __restrict__ widx, int* __restrict__ klen,
                                      int* __restrict__ step_state, int B) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b < B) {
    pos[b] += 1;
    widx[b] += 1;
    klen[b] += 1;
  }
  if (b == 0) step_state[0] += 1;
}

// uint8 CHW image -> normalised, patchified pixel_values row (HF Qwen2VLImageProcessor._preprocess,
// image_processing_qwen2_vl.py:164-246: x/255, (x-mean)/std, temporal frame duplicated,
// rows ordered [gh/2, gw/2, 2, 2], columns [C, T=2, 14, 14]).  The image is already H,W % 28 == 0.
__global__ __launch_bounds__(256) void patchify_kernel(const uint8_t* __restrict__ img,
                                                       bf16_t* __restrict__ out, long ldo, int n_img,
                                                       int H, int Wd, float m0, float m1, float m2,
                                                       float s0, float s1, float s2) {
  const int gh = H / 14, gw = Wd / 14;
  const int P = gh * gw;
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;  // (img, patch, c, py) -> 14 px
  const long total = (long)n_img * P * 3 * 14;
  if (idx >= total) return;
  const int py = idx % 14;
  long r = idx / 14;
  const int c = r % 3;
  r /= 3;
  const int pi = r % P;
  const int n = r / P;
  // patch index -> (block row, block col, in-block row, in-block col)
  const int bw = gw / 2;
  const int blk = pi >> 2, in = pi & 3;
  const int gy = (blk / bw) * 2 + (in >> 1), gx = (blk % bw) * 2 + (in & 1);
  const uint8_t* src = img + (((long)n * 3 + c) * H + (gy * 14 + py)) * Wd + gx * 14;
  const float mean = c == 0 ? m0 : (c == 1 ? m1 : m2);
  const float stdv = c == 0 ? s0 : (c == 1 ? s1 : s2);
  bf16_t* dst = out + ((long)n * P + pi) * ldo + c * 392 + py * 14;
#pragma unroll
  for (int px = 0; px < 14; ++px) {
    const float v = ((float)src[px] * (1.0f / 255.0f) - mean) / stdv;
    const bf16_t o = f2bf(v);
    dst[px] = o;
    dst[196 + px] = o;
  }
}

// CLIP ViT front end (HF modeling_clip.py CLIPVisionEmbeddings / image_processing_clip.py):
//   uint8 [n,3,S,S] -> rows [n*(S/14)^2, ldo] bf16, row = flattened Conv2d patch (c, py, px), columns
//   [588, kpad) zeroed so the patch GEMM can run on a K padded to the MFMA tile.
__global__ __launch_bounds__(256) void clip_patchify_kernel(const uint8_t* __restrict__ img,
                                                            bf16_t* __restrict__ out, long ldo, int kpad,
                                                            int n_img, int S, float m0, float m1, float m2,
                                                            float s0, float s1, float s2) {
  const int g = S / 14, P = g * g;
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;  // (img, patch, c, py) -> 14 px
  const long total = (long)n_img * P * 3 * 14;
  if (idx >= total) return;
  const int py = idx % 14;
  long r = idx / 14;
  const int c = r % 3;
  r /= 3;
  const int pi = r % P;
  const int n = r / P;
  const int gy = pi / g, gx = pi % g;
  const uint8_t* src = img + (((long)n * 3 + c) * S + (gy * 14 + py)) * S + gx * 14;
  const float mean = c == 0 ? m0 : (c == 1 ? m1 : m2);
  const float stdv = c == 0 ? s0 : (c == 1 ? s1 : s2);
  bf16_t* row = out + ((long)n * P + pi) * ldo;
  bf16_t* dst = row + c * 196 + py * 14;
#pragma unroll
  for (int px = 0; px < 14; ++px) dst[px] = f2bf(((float)src[px] * (1.0f / 255.0f) - mean) / stdv);
  if (c == 2 && py == 13)
    for (int k = 588; k < kpad; ++k) row[k] = f2bf(0.f);
}

// embeddings = cat([class_embedding, patch_embeds]) + position_embedding  (modeling_clip.py CLIPVisionEmbeddings.forward)
// pos_cls row 0 already holds bf16(class_embedding + position_embedding[0]).
__global__ __launch_bounds__(256) void clip_embed_kernel(const bf16_t* __restrict__ pe,
                                                         const bf16_t* __restrict__ pos_cls,
                                                         bf16_t* __restrict__ x, int n_img, int tokens, int E) {
  const int nch = E >> 3;
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long)n_img * tokens * nch) return;
  const int ch = idx % nch;
  const long row = idx / nch;
  const int t = row % tokens;
  const long n = row / tokens;
  bf16x8 p = *(const bf16x8*)(pos_cls + (long)t * E + ch * 8);
  if (t > 0) {
    const bf16x8 v = *(const bf16x8*)(pe + (n * (tokens - 1) + (t - 1)) * E + ch * 8);
#pragma unroll
    for (int e = 0; e < 8; ++e) p[e] = f2bf(bf2f(v[e]) + bf2f(p[e]));
  }
  *(bf16x8*)(x + row * E + ch * 8) = p;
}

// Last-layer pruning of the prefill (qwen2vl.hip): out[j] = X[idx[j]] (16-byte chunks), and the index triple that maps the
// last token of prompt j onto the decode-style attention launch (rows = the G q heads of a kv group).
__global__ void gather_rows_kernel(const bf16_t* __restrict__ X, long ldx, const int* __restrict__ idx, bf16_t* __restrict__ Y,
                                   long ldy, int n, int chunks) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long)n * chunks) return;
  const int j = (int)(i / chunks), c = (int)(i - (long)j * chunks);
  *(bf16x8*)(Y + (long)j * ldy + c * 8) = *(const bf16x8*)(X + (long)idx[j] * ldx + c * 8);
}

__global__ void last_rows_prep_kernel(const int* __restrict__ last_index, int* __restrict__ q_start, int* __restrict__ o_start,
                                      int* __restrict__ q_len, int n, int qkv_heads, int n_q_heads, int group) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j < n) {
    q_start[j] = last_index[j] * qkv_heads;
    o_start[j] = j * n_q_heads;
    q_len[j] = group;
  }
}

__global__ void seq_iota_kernel(int* __restrict__ start, int* __restrict__ len, int n, int L) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) {
    start[i] = i * L;
    len[i] = L;
  }
}

// log p(target | row) = logit[target] - logsumexp(row) in fp32 over bf16 logits (HF: CrossEntropyLoss on logits.float(), the
// loss of LLaVA.loglikelihood, reference _llava_hf.py:243-245).  One block per row: max, then sum of exp(x - max) (ascending
// chunks per thread, fixed tree across threads: deterministic).  target < 0 (an ignored label): 0.
__global__ __launch_bounds__(256) void token_logprob_kernel(const bf16_t* __restrict__ logits, long ld, const int* __restrict__ target,
                                                             int V, float* __restrict__ out) {
  __shared__ float red[4];
  const int row = blockIdx.x;
  const bf16_t* x = logits + (long)row * ld;
  const int t = target[row];
  if (t < 0 || t >= V) {
    if (threadIdx.x == 0) out[row] = 0.f;
    return;
  }
  float mx = -INFINITY;
  for (int i = threadIdx.x; i < V; i += 256) mx = fmaxf(mx, bf2f(x[i]));
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = mx;
  __syncthreads();
  mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  __syncthreads();
  float sum = 0.f;
  for (int i = threadIdx.x; i < V; i += 256) sum += __expf(bf2f(x[i]) - mx);
  sum = wave_sum(sum);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = sum;
  __syncthreads();
  if (threadIdx.x == 0) out[row] = bf2f(x[t]) - mx - __logf((red[0] + red[1]) + (red[2] + red[3]));
}

}  // namespace

int owc_launch_clip_patchify(const uint8_t* img, void* out, long ldo, int kpad, int n_img, int S,
                             const float* mean, const float* stdv, hipStream_t st) {
  if (n_img <= 0 || S % 14 || kpad < 588 || ldo < kpad) return OWC_ERR_SHAPE;
  const long total = (long)n_img * (S / 14) * (S / 14) * 3 * 14;
  hipLaunchKernelGGL(clip_patchify_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, img,
                     (bf16_t*)out, ldo, kpad, n_img, S, mean[0], mean[1], mean[2], stdv[0], stdv[1], stdv[2]);
  return hipGetLastError() == hipSuccess ? OWC_OK : OWC_ERR_HIP;
}

int owc_launch_clip_embed(const void* pe, const void* pos_cls, void* x, int n_img, int tokens, int E,
                          hipStream_t st) {
  if (n_img <= 0 || tokens < 2 || (E & 7)) return OWC_ERR_SHAPE;
  const long total = (long)n_img * tokens * (E >> 3);
  hipLaunchKernelGGL(clip_embed_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st,
                     (const bf16_t*)pe, (const bf16_t*)pos_cls, (bf16_t*)x, n_img, tokens, E);
  return hipGetLastError() == hipSuccess ? OWC_OK : OWC_ERR_HIP;
}

int owc_launch_gather_rows(const void* X, long ldx, const int* idx, void* Y, long ldy, int n, int width, hipStream_t st) {
  if (n <= 0 || (width & 7) || (ldx & 7) || (ldy & 7)) return OWC_ERR_SHAPE;
  const long total = (long)n * (width / 8);
  hipLaunchKernelGGL(gather_rows_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, (const bf16_t*)X, ldx, idx,
                     (bf16_t*)Y, ldy, n, width / 8);
  return hipGetLastError() == hipSuccess ? OWC_OK : OWC_ERR_HIP;
}

int owc_launch_last_rows_prep(const int* last_index, int* q_start, int* o_start, int* q_len, int n, int qkv_heads,
                              int n_q_heads, int group, hipStream_t st) {
  hipLaunchKernelGGL(last_rows_prep_kernel, dim3((n + 255) / 256), dim3(256), 0, st, last_index, q_start, o_start, q_len, n,
                     qkv_heads, n_q_heads, group);
  return hipGetLastError() == hipSuccess ? OWC_OK : OWC_ERR_HIP;
}

int owc_launch_seq_iota(int* start, int* len, int n, int L, hipStream_t st) {
  hipLaunchKernelGGL(seq_iota_kernel, dim3((n + 255) / 256), dim3(256), 0, st, start, len, n, L);
  return hipGetLastError() == hipSuccess ? OWC_OK : OWC_ERR_HIP;
}

int owc_launch_layernorm(const void* X, long ldx, const void* W, const void* B, void* Y, long ldy,
                         int rows, int d, float eps, hipStream_t st) {
  if ((d & 7) || d > NORM_MAXC * 512 || (ldx & 7) || (ldy & 7) || rows <= 0) return OWC_ERR_SHAPE;
#define OWC_LN(C_)                                                                                         \
  hipLaunchKernelGGL((norm_kernel<false, C_>), dim3((rows + 3) / 4), dim3(256), 0, st, (const bf16_t*)X, ldx, \
                     (const bf16_t*)W, (const bf16_t*)B, (bf16_t*)Y, ldy, rows, d, eps, (const int*)nullptr)
  if (d <= 1536) OWC_LN(3); else if (d <= 4096) OWC_LN(8); else OWC_LN(16);
#undef OWC_LN
  return hipGetLastError() == hipSuccess ? OWC_OK : OWC_ERR_HIP;
}

int owc_launch_rmsnorm(const void* X, long ldx, const void* W, void* Y, long ldy, int rows, int d,
                       float eps, const int* row_index, hipStream_t st) {
  if ((d & 7) || d > NORM_MAXC * 512 || (ldx & 7) || (ldy & 7) || rows <= 0) return OWC_ERR_SHAPE;
#define OWC_RMS(C_)                                                                                       \
  hipLaunchKernelGGL((norm_kernel<true, C_>), dim3((rows + 3) / 4), dim3(256), 0, st, (const bf16_t*)X, ldx, \
                     (const bf16_t*)W, (const bf16_t*)nullptr, (bf16_t*)Y, ldy, rows, d, eps, row_index)
  if (d <= 1536) OWC_RMS(3); else if (d <= 4096) OWC_RMS(8); else OWC_RMS(16);
#undef OWC_RMS
  return hipGetLastError() == hipSuccess ? OWC_OK : OWC_ERR_HIP;
}

// RMSNorm fused with the per-token e4m3fn quantisation of its output (fp8 decoder): Q [rows, d] codes, S [rows] scales.
int owc_launch_rmsnorm_quant_fp8(const void* X, long ldx, const void* W, void* Q, long ldq, float* S, int rows, int d,
                                 float eps, hipStream_t st) {
  if ((d & 7) || d > NORM_MAXC * 512 || (ldx & 7) || (ldq & 7) || rows <= 0) return OWC_ERR_SHAPE;
#define OWC_RMSQ(C_)                                                                                               \
  hipLaunchKernelGGL((norm_kernel<true, C_, true>), dim3((rows + 3) / 4), dim3(256), 0, st, (const bf16_t*)X, ldx,   \
                     (const bf16_t*)W, (const bf16_t*)nullptr, (bf16_t*)Q, ldq, rows, d, eps, (const int*)nullptr, S)
  if (d <= 1536) OWC_RMSQ(3); else if (d <= 4096) OWC_RMSQ(8); else OWC_RMSQ(16);
#undef OWC_RMSQ
  return hipGetLastError() == hipSuccess ? OWC_OK : OWC_ERR_HIP;
}

int owc_launch_rope_table(float* cos_t, float* sin_t, int n_pos, int n_freq, int dim, float theta,
                          int round_bf16, hipStream_t st) {
  const int n = n_pos * n_freq;
  if (n <= 0) return OWC_ERR_SHAPE;
  hipLaunchKernelGGL(rope_table_kernel, dim3((n + 255) / 256), dim3(256), 0, st, cos_t, sin_t, n_pos,
                     n_freq, dim, theta, round_bf16);
  return hipGetLastError() == hipSuccess ? OWC_OK : OWC_ERR_HIP;
}

int owc_launch_vision_rope(void* qkv, long ld, const int* pos_hw, const float* cos_t,
                           const float* sin_t, int T, int n_heads, int hd, hipStream_t st) {
  if ((hd & 15) || (ld & 3) || T <= 0) return OWC_ERR_SHAPE;
  const long total = (long)T * 2 * n_heads * (hd / 8);
  hipLaunchKernelGGL(vision_rope_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st,
                     (bf16_t*)qkv, ld, pos_hw, cos_t, sin_t, T, n_heads, hd);
  return hipGetLastError() == hipSuccess ? OWC_OK : OWC_ERR_HIP;
}

int owc_launch_mrope_kv(void* qkv, long ld, const int* pos3, long pos_stride, const float* cos_t,
                        const float* sin_t, void* kc, void* vc, const int* tok_slot,
                        const int* tok_idx, int T, int n_q, int n_kv, int s_max, int sec0, int sec1,
                        int bcast_first, int bcast_n, hipStream_t st) {
  if (T <= 0 || (ld & 7)) return OWC_ERR_SHAPE;
  const long total = (long)T * (n_q + 2 * n_kv) * 8;
  hipLaunchKernelGGL(mrope_kv_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st,
                     (bf16_t*)qkv, ld, pos3, pos_stride, cos_t, sin_t, (bf16_t*)kc, (bf16_t*)vc,
                     tok_slot, tok_idx, T, n_q, n_kv, s_max, sec0, sec1, bcast_first, bcast_n);
  return hipGetLastError() == hipSuccess ? OWC_OK : OWC_ERR_HIP;
}

int owc_launch_embed(const int* ids, const int* img_index, const void* table, const void* img,
                     void* out, int T, int d, hipStream_t st) {
  if ((d & 7) || T <= 0) return OWC_ERR_SHAPE;
  const long total = (long)T * (d / 8);
  hipLaunchKernelGGL(embed_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, ids,
                     img_index, (const bf16_t*)table, (const bf16_t*)img, (bf16_t*)out, T, d);
  return hipGetLastError() == hipSuccess ? OWC_OK : OWC_ERR_HIP;
}

int owc_launch_token_logprob(const void* logits, long ld, const int* target, int rows, int V, float* out, hipStream_t st) {
  if (rows <= 0 || V <= 0) return OWC_ERR_SHAPE;
  hipLaunchKernelGGL(token_logprob_kernel, dim3(rows), dim3(256), 0, st, (const bf16_t*)logits, ld, target, V, out);
  return hipGetLastError() == hipSuccess ? OWC_OK : OWC_ERR_HIP;
}

int owc_launch_beam_candidates(const void* logits, long ld, int rows, int V, int K, float* logz, float* top_val, int* top_idx,
                               hipStream_t st) {
  hipLaunchKernelGGL(beam_candidates_kernel, dim3(rows), dim3(BEAM_T), 0, st, (const bf16_t*)logits, ld, V, K, logz, top_val, top_idx);
  return hipGetLastError() == hipSuccess ? OWC_OK : OWC_ERR_HIP;
}

int owc_launch_argmax(const void* logits, long ld, int rows, int V, int* out, hipStream_t st) {
  if (rows <= 0 || V <= 0 || (ld & 7)) return OWC_ERR_SHAPE;
  hipLaunchKernelGGL(argmax_kernel, dim3(rows), dim3(ARGMAX_T), 0, st, (const bf16_t*)logits, ld, V, out);
  return hipGetLastError() == hipSuccess ? OWC_OK : OWC_ERR_HIP;
}

int owc_launch_seen_mark(const int* ids, const int* slot, int n, int V, unsigned* seen, int wpr, int bcast_first, int bcast_n, hipStream_t st) {
  if (n <= 0) return OWC_OK;
  if (!ids || !seen || wpr * 32 < V) return OWC_ERR_SHAPE;
  hipLaunchKernelGGL(seen_mark_kernel, dim3((n + 255) / 256), dim3(256), 0, st, ids, slot, n, V, seen, wpr, bcast_first, bcast_n);
  return hipGetLastError() == hipSuccess ? OWC_OK : OWC_ERR_HIP;
}

int owc_launch_argmax_penalized(const void* logits, long ld, int rows, int V, const unsigned* seen, int wpr, const int* row_slot,
                                const int* row_index, float penalty, int* out, hipStream_t st) {
  if (rows <= 0 || V <= 0 || (ld & 7) || !seen || wpr * 32 < V || !(penalty > 0.f)) return OWC_ERR_SHAPE;
  hipLaunchKernelGGL(argmax_penalized_kernel, dim3(rows), dim3(ARGMAX_T), 0, st, (const bf16_t*)logits, ld, V, seen, wpr, row_slot, row_index, penalty, out);
  return hipGetLastError() == hipSuccess ? OWC_OK : OWC_ERR_HIP;
}

int owc_launch_penalize_rows(void* logits, long ld, int rows, int V, const unsigned* seen, int wpr, const int* row_slot,
                             const int* row_index, float penalty, hipStream_t st) {
  if (rows <= 0 || V <= 0 || !seen || wpr * 32 < V || !(penalty > 0.f)) return OWC_ERR_SHAPE;
  hipLaunchKernelGGL(penalize_rows_kernel, dim3(rows), dim3(1024), 0, st, (bf16_t*)logits, ld, V, seen, wpr, row_slot, row_index, penalty);
  return hipGetLastError() == hipSuccess ? OWC_OK : OWC_ERR_HIP;
}

int owc_launch_sample(const void* logits, long ld, int rows, int V, const owc_sampling* sp, const int* row_map, int step,
                      const int* step_state, int* out, hipStream_t st) {
  if (rows <= 0 || V <= 0 || !sp || !(sp->temperature > 0.f)) return OWC_ERR_SHAPE;
  hipLaunchKernelGGL(sample_kernel, dim3(rows), dim3(SAMPLE_T), 0, st, (const bf16_t*)logits, ld, V, 1.4426950408889634f / sp->temperature,
                     sp->top_k, sp->top_p, (uint32_t)sp->seed, (uint32_t)(sp->seed >> 32), row_map, sp->stream_id, sp->step_offset, step, step_state,
                     out);
  return hipGetLastError() == hipSuccess ? OWC_OK : OWC_ERR_HIP;
}

int owc_launch_decode_update(int* next_tok, uint8_t* done, int* out_tokens, int out_stride, int step,
                             const int* step_state, int B, int eos0, int eos1, int pad, const int* out_row, const int* forced,
                             hipStream_t st) {
  hipLaunchKernelGGL(decode_update_kernel, dim3((B + 255) / 256), dim3(256), 0, st, next_tok, done,
                     out_tokens, out_stride, step, step_state, B, eos0, eos1, pad, out_row, forced);
  return hipGetLastError() == hipSuccess ? OWC_OK : OWC_ERR_HIP;
}

int owc_launch_decode_compact(const int* const* src, int* const* dst, const uint8_t* done_src, uint8_t* done_dst,
                              const int* live, int n, hipStream_t st) {
  if (n <= 0) return OWC_ERR_SHAPE;
  compact_ptrs p;
  for (int a = 0; a < 7; ++a) {
    p.src[a] = src[a];
    p.dst[a] = dst[a];
  }
  hipLaunchKernelGGL(decode_compact_kernel, dim3((n + 255) / 256), dim3(256), 0, st, p, done_src, done_dst, live, n);
  return hipGetLastError() == hipSuccess ? OWC_OK : OWC_ERR_HIP;
}

int owc_launch_decode_advance(int* pos, int* widx, int* klen, int* step_state, int B, hipStream_t st) {
  hipLaunchKernelGGL(decode_advance_kernel, dim3((B + 255) / 256), dim3(256), 0, st, pos, widx, klen, step_state, B);
  return hipGetLastError() == hipSuccess ? OWC_OK : OWC_ERR_HIP;
}

int owc_launch_patchify(const uint8_t* img, void* out, long ldo, int n_img, int H, int W,
                        const float* mean, const float* stdv, hipStream_t st) {
  if (n_img <= 0 || H % 28 || W % 28 || ldo < 1176) return OWC_ERR_SHAPE;
  const long total = (long)n_img * (H / 14) * (W / 14) * 3 * 14;
  hipLaunchKernelGGL(patchify_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, img,
                     (bf16_t*)out, ldo, n_img, H, W, mean[0], mean[1], mean[2], stdv[0], stdv[1], stdv[2]);
  return hipGetLastError() == hipSuccess ? OWC_OK : OWC_ERR_HIP;
}
