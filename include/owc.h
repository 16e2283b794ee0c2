/*
 * owc.h — C ABI of libowc_hip.so, the MI355X (gfx950) hot path of open-world LMM classification.
 *
 * The reference (altndrr/lmms-owc) is pure Python and has no FFI: its plug points are the
 * `src/models` registry + `Model` ABC and the `src/data/metrics` registry.  This
 * header is the boundary UNDER those plug points: every entry replaces arithmetic the reference
 * reaches through HF transformers / torch, and is what a ctypes stub in the reference's
 * `src/models/_qwen2_vl.py` / `src/data/pipelines/text/_text.py` would bind.
 * `HF:` = transformers/models/qwen2_vl/modeling_qwen2_vl.py (the third-party code the reference
 * calls at src/models/_qwen2_vl.py:319-329); `BERT:` = transformers/models/bert/modeling_bert.py
 * (called at src/data/pipelines/text/_text.py:197-198).
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless the name ends in `_host` or the comment says host;
 *   - the caller owns every buffer (inputs, outputs, weights, workspaces); the library owns only a
 *     small zero page inside `owc_ctx`;
 *   - all work is enqueued on `stream` (a hipStream_t passed as void*; NULL = default stream);
 *     no hidden synchronisation, no internal threads, safe to capture into a hipGraph;
 *   - return value: 0 (OWC_STATUS_OK) or a negative owc_status; `owc_last_error(ctx)` has the text;
 *   - bf16 tensors are raw 16-bit brain-float, row-major, leading dimensions in ELEMENTS.
 */
#ifndef OWC_H_
#define OWC_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct owc_ctx owc_ctx;

enum owc_status {
  OWC_STATUS_OK = 0,
  OWC_STATUS_ERR_ARG = -1,
  OWC_STATUS_ERR_HIP = -2,
  OWC_STATUS_ERR_SHAPE = -3,
  OWC_STATUS_ERR_WORKSPACE = -4
};

/* GEMM epilogues (what a bf16 torch module does after the matmul, same rounding points). */
enum owc_epilogue {
  OWC_EPILOGUE_NONE = 0,       /* C = bf16(acc + bias) */
  OWC_EPILOGUE_QUICK_GELU = 1, /* HF ACT2FN["quick_gelu"], vision MLP (HF:293-302) */
  OWC_EPILOGUE_GELU_ERF = 2,   /* nn.GELU(), PatchMerger (HF:281-286); BERT intermediate */
  OWC_EPILOGUE_RESIDUAL = 3,   /* C = bf16(residual + bf16(acc + bias)) */
  OWC_EPILOGUE_SWIGLU = 4,     /* gate/up rows interleaved in 16-row groups; C[M, N/2] (HF:464-466) */
  OWC_EPILOGUE_F32 = 5         /* fp32 output, no rounding */
};

/* ---- lifetime ------------------------------------------------------------------------------ */
/* One context per PROCESS: the design is one process per GPU, and kernel attributes, tuning knobs and
 * the profile recording are process-wide.  A second owc_init on a different device returns OWC_STATUS_ERR_ARG. */
int owc_init(int device, owc_ctx** out);
int owc_destroy(owc_ctx* ctx);
const char* owc_last_error(const owc_ctx* ctx);
/* A-B / tuning knobs (process-wide; the library reads NO environment variable - every switch goes through this call):
 * "gemm_big_min_m", "gemm_big_min_tiles" (fewest 256x256 tiles for which the 256x256 kernels run; negative = default 144),
 * "gemm_skinny_max_m" (0 disables the weight-streaming small-M kernel, a negative value restores the defaults),
 * "gemm_mid_max_tiles" (0 disables the 64x64-tile kernel), "gemm_pingpong" (0: the lock-step 256x256 kernels, 1: bf16
 * ping-pong only, 2 / negative = default: the bf16 and the fp8 ping-pong kernels), "prefill_prune_last" (0: owc_llm_prefill runs the last layer's attention /
 * o-proj / MLP on every row instead of the last-token rows only -- same logits bit for bit, tested), "bert_bf16x3" (0:
 * owc_bert_embed runs its linears on the exact f32-input MFMA instead of the three-piece bf16 split), "decode_fuse" (0: a
 * decode step runs mrope_kv + the generic attention kernel instead of owc_decode_attention), "decode_attn_nbuf1" (block count above
 * which the fused decode attention single-buffers V), "decode_norm_fuse" (rows up to which the decoder's RMSNorm is fused into the
 * qkv / gate-up skinny GEMM: 0 off, default 2, at most 4 -- same bits, tested), "gemm_skinny_deep" (0: no 9-deep ring for the long-K
 * skinny launches), "gemm_small_tiles" (the 64x64-tile kernel's 64x32 / 32x32 shapes for launches that cannot fill the chip: 0 off,
 * 1 = default: chosen by block count, 2 / 3 / 4 force 64x64 / 64x32 / 32x32), "decode_norm_fuse_ring" (rows, at most 8, up to which the ring kernel
 * folds the decoder's RMSNorm into the qkv / gate-up projection; 0 off), "gemm_k_pairs" / "gemm_k_pairs_min_k" (K-tiles per ring stage: 0 one,
 * default four from K >= 1024), "gemm_wide_tiles" (0: no 64x160 / 128x160 tiles of that
 * kernel for launches of at most 128 rows x tens of thousands of columns), "gemm_ring_128" (0: no 128x64 tiles of that kernel - bf16
 * and fp8 - for a few hundred rows x a few thousand columns, where 64x64 tiles need more than a round and a half of the chip),
 * "gemm_pp128" (the 256x128-tile ping-pong kernel for launches with too few 256x256 tiles to fill the chip - the o / down projections
 * of a decode step at 1024-2048 rows: 0 off, n > 0: from n tiles of 256x128, negative: the default), "gemm_walk" (block id -> output
 * tile of the 256x256 ping-pong kernels: 0 the rows-of-4 walk of rounds 1-5, 2 column groups of <= 8 tile columns walked down all
 * tile rows, 1 / negative (default) column groups where they measured faster - K >= 4.5 N with at least as many tile rows as columns),
 * "gemm_tail_split" (1 / negative, the default: when the 256x256 tiles beyond the whole rounds of one tile per CU are at most half a
 * round, that rest of N runs as one round of 256x128 tiles in a second launch - the gate/up projection of 512- / 1024- / 1792-row
 * decode steps; 0: one launch).
 * Every knob above selects between kernels that return the SAME results.  The timing-only experiment knobs "gemm_dbg" /
 * "attn_dbg" (parts of a kernel switched off to price them; outputs are garbage) exist only in libowc_hip_timing.so, which
 * `python -m lmms_owc_amd.build --timing` builds with -DOWC_TIMING_KNOBS for tools/; the product library does not know them.
 * Returns OWC_ERR_ARG for an unknown name.  Measurement aid only: no reference counterpart. */
int owc_tuning_set(const char* name, int value);
int owc_abi_version(void); /* bumped whenever a signature in this header changes */
int owc_has_timing_knobs(void); /* 0 in the product library; 1 in the -DOWC_TIMING_KNOBS build that tools/ load */

/* ---- measurement hooks (bench.py roofline leg) ------------------------------------------------ */
/* When enabled, every owc_gemm_bf16 / owc_gemm_fp8 launch (also inside the model drivers) is bracketed by a
 * HIP-event pair on its own stream.  owc_gemm_profile_read (call after synchronising) returns the summed
 * kernel time, the summed algorithmic FLOPs (2*M*N*K per launch) and the launch count, then resets.
 * Every output is an ARRAY OF TWO: [0] the bf16 GEMMs, [1] the fp8 GEMMs. */
int owc_gemm_profile_enable(owc_ctx* ctx, int on);
int owc_gemm_profile_read(owc_ctx* ctx, double* total_ms, double* total_flops, int64_t* launches);
/* The same recording, every launch class: arrays of n_kinds (<= OWC_PROF_KINDS) entries indexed by OWC_PROF_*.  `total_work` is
 * 2*M*N*K for the GEMM classes and 2*N*C*D for the cosine top-k; the attention launcher only sees device-side sequence
 * lengths, so its entry is 0 and the caller prices the launches from the shapes it passed (bench.py). */
enum owc_prof_kind {
  OWC_PROF_GEMM_BF16 = 0,     /* owc_gemm_bf16 and every bf16 Linear of the model drivers */
  OWC_PROF_GEMM_FP8 = 1,      /* owc_gemm_fp8 (fp8 decoder projections) */
  OWC_PROF_ATTN_VISION = 2,   /* owc_attention_bf16 with head_dim != 128 (Qwen2-VL vision tower 80, CLIP 64) */
  OWC_PROF_ATTN_PREFILL = 3,  /* owc_attention_bf16 with head_dim 128, causal (decoder prefill; MFMA-bound) + the last prefill layer's
                                 last-token launch */
  OWC_PROF_SCORER_GEMM = 4,   /* the sentence encoder's linears (owc_bert_embed) */
  OWC_PROF_COSINE_TOPK = 5,   /* cosine_topk_kernel */
  OWC_PROF_ATTN_DECODE = 6,   /* owc_attention_bf16 with head_dim 128, non-causal: the decode-step mapping (one query row per q head
                                 over the whole KV cache of the sequence; HBM-bound: it streams the cache) */
  OWC_PROF_KINDS = 7
};
int owc_profile_read(owc_ctx* ctx, int n_kinds, double* total_ms, double* total_work, int64_t* launches);
/* The bf16 GEMM launches of the last owc_profile_read / owc_gemm_profile_read, grouped by shape (round 6: the per-SHAPE table a
 * roofline fraction can be recomputed from).  Entry i: shape[4 i ..] = (M, N, K, epilogue), stats[4 i ..] = (launches, total ms,
 * min ms, max ms) of its launches.  Returns the number of distinct shapes (entries beyond max_n are not written), -1 on bad arguments. */
int owc_profile_shapes(owc_ctx* ctx, int max_n, int32_t* shape, double* stats);

/* ---- op level (each is one kernel launch; used by the model drivers below and by tests) ------ */

/* C[M,N] = A[M,K] . W[N,K]^T (+bias[N]) with a fused epilogue.  Replaces torch.nn.Linear.forward as
 * called from HF:268-275 (patch-embed conv == GEMM), :349-350, :296-301, :281-291, :501-504, :460-466.
 * K % 8 == 0, lda % 8 == 0, ldw % 8 == 0; bf16 outputs: N % 8 == 0, ldc % 8 == 0 (16-byte stores). */
int owc_gemm_bf16(owc_ctx* ctx, const void* A, int64_t lda, const void* W, int64_t ldw,
                  const void* bias, const void* residual, int64_t ldr, void* C, int64_t ldc,
                  int M, int N, int K, int epilogue, void* stream);

/* fp32 variant on the f32-input MFMA (exact fp32 products) for the sentence encoder
 * (BERT:  BertSelfAttention / BertSelfOutput / BertIntermediate / BertOutput linears).
 * epilogue: NONE, GELU_ERF or RESIDUAL (all fp32).  K % 4 == 0, N % 4 == 0. */
int owc_gemm_f32(owc_ctx* ctx, const float* A, int64_t lda, const float* W, int64_t ldw,
                 const float* bias, const float* residual, int64_t ldr, float* C, int64_t ldc,
                 int M, int N, int K, int epilogue, void* stream);

/* torch.nn.LayerNorm over rows of a bf16 matrix (HF:428-429 norm1/norm2, HF:281 ln_q). d % 8 == 0. */
int owc_layernorm_bf16(owc_ctx* ctx, const void* X, int64_t ldx, const void* weight,
                       const void* bias, void* Y, int64_t ldy, int rows, int d, float eps,
                       void* stream);

/* Qwen2VLRMSNorm (HF:105-110).  `row_index` (int32[rows], may be NULL) gathers source rows. */
int owc_rmsnorm_bf16(owc_ctx* ctx, const void* X, int64_t ldx, const void* weight, void* Y,
                     int64_t ldy, int rows, int d, float eps, const int32_t* row_index,
                     void* stream);

/* cos/sin tables: entry [p][j] = cos|sin(p * theta^(-2j/dim)), fp32, optionally rounded to bf16
 * (HF:156-170 casts the decoder's cos/sin to the activation dtype; HF:239-248 vision keeps fp32). */
int owc_rope_table(owc_ctx* ctx, float* cos_t, float* sin_t, int n_pos, int n_freq, int dim,
                   float theta, int round_bf16, void* stream);

/* apply_rotary_pos_emb_vision (HF:225-236) in place on the q and k parts of qkv[T, 3*H*hd];
 * pos_hw[T][2] = (h, w) patch coordinates in merge-block order (vision_utils.get_vision_position_ids). */
int owc_vision_rope(owc_ctx* ctx, void* qkv, int64_t ld, const int32_t* pos_hw, const float* cos_t,
                    const float* sin_t, int T, int n_heads, int head_dim, void* stream);

/* apply_multimodal_rotary_pos_emb (HF:180-222) + DynamicCache.update: rotates q in place inside
 * qkv[T, (Hq + 2 Hkv) * 128] and writes rotated k / plain v to cache rows
 * [(tok_slot[t] * Hkv + kvh) * s_max + tok_idx[t]].  pos3 = int32 [3][pos_stride]. */
int owc_mrope_kv_write(owc_ctx* ctx, void* qkv, int64_t ld, const int32_t* pos3, int64_t pos_stride,
                       const float* cos_t, const float* sin_t, void* k_cache, void* v_cache,
                       const int32_t* tok_slot, const int32_t* tok_idx, int T, int n_q_heads,
                       int n_kv_heads, int s_max, int mrope_sec0, int mrope_sec1, void* stream);

/* One DECODE step's attention for B sequences in a single launch (head_dim 128): rotate q and k of the fed token with the rope
 * table row pos[b] (a generated token's three M-RoPE streams are equal: plain 1-D rope), write the rotated k row and the v row
 * into the KV cache at row write_idx[b] of slot[b], then attend the G = n_q / n_kv query heads of every kv group over the
 * k_len[b] cached rows (the new one included).  qkv: [B][(n_q + 2 n_kv) * 128] as the fused projection leaves it (NOT modified);
 * caches: [slot][n_kv][s_max][128]; out: [B][n_q * 128].  Equals owc_mrope_kv_write + owc_attention_bf16 in the decode
 * mapping: identical cache rows, attention output within the rounding of P (the four waves of a block split the keys and merge
 * their partial softmax results in a fixed order).  Replaces HF apply_multimodal_rotary_pos_emb (:180-222), the cache
 * update and Qwen2VLAttention's sdpa (:508-556) for a generated token. */
int owc_decode_attention(owc_ctx* ctx, const void* qkv, int64_t ld, const int32_t* pos, const float* cos_t, const float* sin_t,
                         void* k_cache, void* v_cache, const int32_t* slot, const int32_t* write_idx, const int32_t* k_len,
                         void* out, int64_t ldo, int B, int n_q_heads, int n_kv_heads, int s_max, float scale, void* stream);

/* softmax(Q K^T * scale [+ causal mask]) V for packed variable-length sequences (flash-style).
 * Element (seq b, head h, row i, dim d) lives at
 *   Q: Q + (q_start[b] + i) * q_ts + h * q_hs + d        K/V: K + (k_start[b] + j) * k_ts + (h / kv_group) * k_hs + d
 *   O: O + (o_start[b] + i) * o_ts + h * o_hs + d        (o_start == NULL -> q_start)
 * seq_len[b] = number of keys; q_len[b] = number of query rows (NULL -> seq_len).
 * head_dim in {80, 128, 64, 32}.  Replaces HF:317-339 / sdpa / flash-attn (vision HF:381-419,
 * decoder HF:537-552). */
int owc_attention_bf16(owc_ctx* ctx, const void* Q, int64_t q_ts, int64_t q_hs, const void* K,
                       int64_t k_ts, int64_t k_hs, const void* V, int64_t v_ts, int64_t v_hs,
                       void* O, int64_t o_ts, int64_t o_hs, const int32_t* q_start,
                       const int32_t* o_start, const int32_t* k_start, const int32_t* seq_len,
                       const int32_t* q_len, int n_seq, int n_heads, int kv_group, int head_dim,
                       int max_q_len, int causal, float scale, void* stream);

/* ---- fp8 (OCP e4m3fn) decoder projections: BASELINE.json config #5 (Qwen2-VL-72B fp8 MFMA decode).  No reference
 * counterpart (the reference's reduced-precision loader is bitsandbytes, src/models/_base.py:116-121); the arithmetic is
 * defined by oracle/fp8_np.py.
 * q[r][c] = rne_e4m3(x[r][c] / scale[r]), scale[r] = max|x[r]| / 448 (1 for a zero row); x bf16 [rows, cols], cols % 8 == 0.
 * Per token for activations, per output channel for weights. */
int owc_quantize_rows_fp8(owc_ctx* ctx, const void* x, int64_t ldx, void* q, int64_t ldq, float* scale, int rows,
                          int cols, void* stream);

/* RMSNorm (owc_rmsnorm_bf16 arithmetic) fused with owc_quantize_rows_fp8 of its output: bit-identical to running the two, without
 * the bf16 round trip through memory.  Used in front of the qkv and gate/up projections of the fp8 decoder. */
int owc_rmsnorm_quant_fp8(owc_ctx* ctx, const void* x, int64_t ldx, const void* weight, void* q, int64_t ldq, float* scale,
                          int rows, int d, float eps, void* stream);

/* C[M,N] bf16 = epilogue((A8[M,K] . W8[N,K]^T) * a_scale[m] * w_scale[n] + bias) on v_mfma_scale_f32_16x16x128_f8f6f4.
 * K % 128 == 0, lda / ldw in bytes and % 16 == 0; epilogue in {OWC_EPI_NONE, OWC_EPI_RESIDUAL, OWC_EPI_SWIGLU}
 * (SWIGLU: W rows and w_scale interleaved gate/up per 16 like owc_gemm_bf16). */
int owc_gemm_fp8(owc_ctx* ctx, const void* A, int64_t lda, const float* a_scale, const void* W, int64_t ldw,
                 const float* w_scale, const void* bias, const void* R, int64_t ldr, void* C, int64_t ldc, int M, int N,
                 int K, int epilogue, void* stream);

/* inputs_embeds = embed_tokens(ids) with image rows scattered in (HF:1160-1168):
 * out[t] = img_index[t] >= 0 ? img_embeds[img_index[t]] : table[ids[t]]   (img_index may be NULL). */
int owc_embed_tokens(owc_ctx* ctx, const int32_t* ids, const int32_t* img_index, const void* table,
                     const void* img_embeds, void* out, int T, int d, void* stream);

/* Sampling parameters of a generation (NULL wherever one is taken: greedy argmax).  HF GenerationMixin._sample, reached from the
 * reference with do_sample = temperature > 0 (src/models/_qwen2_vl.py:319-329, _llava_hf.py:365-376): logits / temperature ->
 * top-k (0: off) -> top-p on the survivors (<= 0 or >= 1: off) -> softmax -> one multinomial draw.
 *   Random stream (documented, NOT torch's): Philox4x32-10, key = seed, counter = (stream id of the sequence, step + its step
 *   offset, 0, 0);
 *   stream_id (optional, int32 per ORIGINAL batch row; NULL: the row index): a sequence's draws depend on (seed, its stream id,
 *   step) only - not on the batch it runs in, not on row compaction, not on the rank count when the caller passes document ids.
 *   Weights are exact integers floor(2^40 exp((l - max) / T)).  Ties at a cut: top-k keeps every token whose value equals the k-th
 *   largest (exactly HF's TopKLogitsWarper, which removes `scores < k-th value`); where the top-p cut falls inside a run of equal
 *   bf16 logits, HF keeps a prefix of the run in its unspecified sort order - this library keeps the same NUMBER of them and takes
 *   the lowest token ids.  Hence top_k = 1 with a top_p < 1 (Qwen2-VL's generation_config.json: top_k 1, top_p 0.001) leaves exactly
 *   one token, the lowest id among the maxima: the draw equals owc_argmax_bf16 bit for bit, ties included. */
typedef struct owc_sampling {
  float temperature; /* > 0 */
  int32_t top_k;
  float top_p;
  uint64_t seed;
  const int32_t* stream_id;
  const int32_t* step_offset; /* optional, per ORIGINAL row: added to the step of the Philox counter (a sequence that continues in a
                                 later pass keeps counting its own steps: owc_llm_decode_step's `step` is pass-local) */
} owc_sampling;

/* one token per row of bf16 logits by owc_sampling (row_map: optional ORIGINAL row of each row, see owc_llm_decode_step). */
int owc_sample_bf16(owc_ctx* ctx, const void* logits, int64_t ld, int rows, int vocab, const owc_sampling* sampling,
                    const int32_t* row_map, int step, int32_t* out, void* stream);

/* greedy argmax over bf16 logits rows (lowest index on ties). */
int owc_argmax_bf16(owc_ctx* ctx, const void* logits, int64_t ld, int rows, int vocab,
                    int32_t* out, void* stream);
/* Beam search, the arithmetic of one step (HF GenerationMixin._beam_search, reached through `num_beams=gen_kwargs["num_beams"]`,
 * /root/reference/src/models/_qwen2_vl.py:308-329, _llava_hf.py:365-376): per row of bf16 logits [rows, vocab] (row stride ld)
 *   logz[r]               = log sum_v exp(logits[r][v])      (fp32; log_softmax(x)[v] = x[v] - logz)
 *   top_val / top_idx[r][] = the k largest logits and their token ids, descending, the lowest id first among equal values
 * k = 2 x num_beams is what a step needs from every running hypothesis (k <= 64).  The hypothesis bookkeeping that consumes them
 * (accumulated scores, finished slots with the length penalty, parents for the KV cache) is host integer / scalar work:
 * lmms_owc_amd/engine/beam.py. */
int owc_beam_candidates(owc_ctx* ctx, const void* logits, int64_t ld, int rows, int vocab, int k, float* logz, float* top_val,
                        int32_t* top_idx, void* stream);
/* out[r] = log softmax(logits[r])[target[r]] in fp32 (0 where target[r] < 0): the per-token terms of HF's causal-LM loss,
 * which LLaVA.loglikelihood averages (reference src/models/_llava_hf.py:243-245, outputs["loss"]). */
int owc_token_logprob_bf16(owc_ctx* ctx, const void* logits, int64_t ld, const int32_t* target, int rows, int vocab, float* out,
                           void* stream);

/* uint8 [n,3,H,W] -> pixel_values rows [n * (H/14)*(W/14), 1176] bf16
 * (HF image_processing_qwen2_vl.py:164-246: rescale, normalise, duplicate frame, patchify). */
int owc_patchify_u8(owc_ctx* ctx, const uint8_t* images, void* pixel_values, int64_t ld, int n,
                    int H, int W, const float* mean_host, const float* std_host, void* stream);

/* ---- model level: Qwen2-VL vision tower ----------------------------------------------------- */
typedef struct owc_vit_layer {
  /* qkv_w / qkv_b: inside every q and k head the rows are PAIR-INTERLEAVED, new[2j] = old[j],
   * new[2j+1] = old[j + head_dim/2] (q.k is invariant to it), so the rotary pairs of HF:225-236 are
   * adjacent output columns and the rotation is fused into the projection's epilogue. */
  const void *ln1_w, *ln1_b, *qkv_w, *qkv_b, *proj_w, *proj_b;
  const void *ln2_w, *ln2_b, *fc1_w, *fc1_b, *fc2_w, *fc2_b;
} owc_vit_layer;

typedef struct owc_vit_weights {
  int32_t depth, embed_dim, num_heads, mlp_hidden, patch_k, out_dim, merge_unit; /* merge_unit = 4 */
  float ln_eps;
  const void* patch_w;            /* [embed_dim, patch_k] (Conv3d weight flattened, HF:266) */
  const owc_vit_layer* layers;    /* HOST array of `depth` entries */
  const void *merger_ln_w, *merger_ln_b, *merger_fc1_w, *merger_fc1_b, *merger_fc2_w, *merger_fc2_b;
  const float *rope_cos, *rope_sin; /* [rope_positions][head_dim/4] from owc_rope_table(dim = head_dim/2) */
  int32_t rope_positions;
  /* variant 0: Qwen2-VL (LayerNorm blocks, fc1 -> quick_gelu -> fc2).
   * variant 1: Qwen2.5-VL (HF modeling_qwen2_5_vl.py:294-323, :137-150): RMSNorm blocks and merger norm (the *_b norm pointers are
   *   NULL), gated MLP down(silu(gate(x)) * up(x)) with biases - fc1_w / fc1_b hold the gate and up rows INTERLEAVED in groups of 16
   *   ([g0..g15, u0..u15, g16.., ...], the layout of OWC_EPILOGUE_SWIGLU) with `mlp_hidden` = the intermediate size zero-padded to a
   *   multiple of 128 (3420 -> 3456: zero rows give silu(0) * 0 = 0 and meet zero columns of fc2_w = down_proj, exactly), and
   *   WINDOW attention: layer i attends inside the windows unless bit i of `fullatt_mask` is set (owc_vit25_forward). */
  int32_t variant;
  uint64_t fullatt_mask;
} owc_vit_weights;

size_t owc_vit_workspace_bytes(const owc_vit_weights* w, int T);

/* Qwen2VisionTransformerPretrainedModel.forward (HF:700-731): pixel_values[T, patch_k] ->
 * merged image embeddings out[T / merge_unit, out_dim].  seq_start/seq_len (int32[n_img]) are the
 * cu_seqlens of HF:711; pos_hw as in owc_vision_rope.  max_pos_hw = the largest patch coordinate + 1 in pos_hw (the host
 * builds pos_hw from the grids, so it knows): OWC_STATUS_ERR_SHAPE when it exceeds w->rope_positions - the rotary table
 * is indexed by these coordinates (smart_resize admits aspect ratios up to 200, i.e. grids far from square). */
int owc_vit_forward(owc_ctx* ctx, const owc_vit_weights* w, const void* pixel_values, int64_t ld_pix,
                    const int32_t* pos_hw, const int32_t* seq_start, const int32_t* seq_len,
                    int n_img, int T, int max_len, int max_pos_hw, void* out, void* workspace, size_t ws_bytes,
                    void* stream);

/* Qwen2_5_VisionTransformerPretrainedModel.forward (HF modeling_qwen2_5_vl.py:408-470), as the reference loads it for the
 * `qwen2.5-vl-*` registry names (src/models/_qwen2_vl.py:106-115): patch embed -> rows re-ordered so that the 2x2 token groups of
 * one 112-pixel window are contiguous (`tok_index[T]`: the source row of every window-ordered row; integer bookkeeping of
 * transformers/vision_utils.get_vision_window_index, done by the host) -> blocks with window / full attention -> RMSNorm merger
 * -> `out` rows back in the ORIGINAL order (`out_index[T / merge_unit]`: the window-ordered merged row of every output row, i.e.
 * argsort(window_index)).  pos_hw is given in window order; seq_start / seq_len describe the images (an image's rows stay
 * contiguous), win_start / win_len the windows (at most 64 patches each). */
int owc_vit25_forward(owc_ctx* ctx, const owc_vit_weights* w, const void* pixel_values, int64_t ld_pix,
                      const int32_t* pos_hw, const int32_t* tok_index, const int32_t* out_index, const int32_t* seq_start,
                      const int32_t* seq_len, int n_img, int max_len, const int32_t* win_start, const int32_t* win_len, int n_win,
                      int max_win_len, int T, int max_pos_hw, void* out, void* workspace, size_t ws_bytes, void* stream);

/* ---- model level: LLaVA image branch (CLIP ViT + projector) -------------------------------- */
/* Replaces LlavaModel.get_image_features (HF modeling_llava.py:144-189) as reached from the reference's
 * src/models/_llava_hf.py:365-376 (`self.model.generate(**inputs)`), config #4 of BASELINE.json. */
typedef struct owc_clip_weights {
  int32_t n_layers;   /* encoder layers to RUN = num_hidden_layers + 1 + vision_feature_layer (-2 -> L-1) */
  int32_t embed_dim, num_heads, mlp_hidden;
  int32_t patch_k;    /* 3*14*14 = 588 zero-padded to a multiple of 8 (640 keeps K % 64 == 0) */
  int32_t tokens;     /* 1 + (image_size/14)^2 per image (CLS first) */
  int32_t out_dim;    /* projector width = decoder d_model */
  float ln_eps;
  const void* patch_w;            /* [embed_dim, patch_k], zero in the padded columns */
  const void* pos_cls;            /* [tokens, embed_dim]: position_embedding, row 0 += class_embedding */
  const void *pre_ln_w, *pre_ln_b;
  const owc_vit_layer* layers;    /* HOST array; qkv_w = cat(q_proj, k_proj, v_proj) rows, NOT interleaved */
  const void *proj1_w, *proj1_b, *proj2_w, *proj2_b; /* multi_modal_projector.linear_1 / linear_2 */
} owc_clip_weights;

size_t owc_clip_workspace_bytes(const owc_clip_weights* w, int n_img);

/* patches [n_img * (tokens-1), patch_k] bf16 -> out [n_img * tokens, out_dim] bf16: the projected
 * hidden_states[vision_feature_layer] of every token, CLS row included (row n*tokens); the "default"
 * feature-select strategy is the caller skipping that row when it builds owc_llm_prefill's img_index. */
int owc_clip_forward(owc_ctx* ctx, const owc_clip_weights* w, const void* patches, int64_t ld_patches,
                     int n_img, void* out, void* workspace, size_t ws_bytes, void* stream);

/* uint8 [n,3,S,S] (resized + cropped on the host) -> patches [n*(S/14)^2, ld] bf16, columns (c, py, px),
 * [588, patch_k) zeroed (HF image_processing_clip.py: rescale 1/255, normalise; Conv2d patch order). */
int owc_clip_patchify_u8(owc_ctx* ctx, const uint8_t* images, void* patches, int64_t ld, int patch_k, int n,
                         int S, const float* mean_host, const float* std_host, void* stream);

/* ---- model level: Qwen2-VL decoder ---------------------------------------------------------- */
typedef struct owc_llm_layer {
  const void *ln1_w, *qkv_w, *qkv_b, *o_w, *ln2_w, *gateup_w, *down_w;
  /* qkv_w = cat(q_proj, k_proj, v_proj) rows; gateup_w = gate/up rows interleaved per 16 */
  /* owc_llm_weights.weight_dtype == OWC_WEIGHTS_FP8: the four projection weights above are e4m3fn codes [N, K] (same row
   * order) from owc_quantize_rows_fp8 and these are their per-row scales; NULL / unused for bf16 weights */
  const float *qkv_s, *o_s, *gateup_s, *down_s;
} owc_llm_layer;

#define OWC_WEIGHTS_BF16 0
#define OWC_WEIGHTS_FP8 1

typedef struct owc_llm_weights {
  int32_t n_layers, d_model, n_q_heads, n_kv_heads, head_dim, d_ff, vocab;
  int32_t mrope_sec0, mrope_sec1; /* mrope_section[0], [1] (HF:214) */
  float rms_eps;
  const void* embed;             /* [vocab, d_model] */
  const owc_llm_layer* layers;   /* HOST array of n_layers entries */
  const void* final_norm_w;
  const void* lm_head_w;         /* [vocab, d_model] (== embed when tied) */
  const float *rope_cos, *rope_sin; /* [rope_positions][head_dim/2], bf16-rounded values */
  int32_t rope_positions;
  int32_t weight_dtype;          /* OWC_WEIGHTS_BF16 | OWC_WEIGHTS_FP8 (decoder projections only: embedding, norms, biases and
                                    lm_head stay bf16; activations are quantised per token in front of every fp8 projection) */
} owc_llm_weights;

typedef struct owc_kv_cache {
  void* k; /* [n_layers][n_slots][n_kv_heads][s_max][head_dim] bf16 */
  void* v;
  int32_t n_slots, s_max;
} owc_kv_cache;

size_t owc_llm_workspace_bytes(const owc_llm_weights* w, int T, int n_seq);

/* Prefill of packed prompts (T token rows in total), HF:1144-1205 + :762-846 + lm_head.
 *   ids/img_index/tok_slot/tok_idx: int32[T]; pos3: int32[3][T] from get_rope_index (HF:914-1019);
 *   the token rows form `n_seq` attention segments: segment s covers rows [seq_start[s], +q_len[s]) and attends,
 *   causally, the keys [0, seq_len[s]) of the cache slot at k_start[s] (= slot * n_kv_heads * s_max); its rows
 *   are the LAST q_len[s] positions of that key range (q_len == NULL -> q_len = seq_len, the plain case).
 *   Shared-prefix mode (every prompt of a classification task starts with the same system/question tokens): the
 *   prefix is computed ONCE as segment n_seq-1 (tok_slot = -1 on its rows: the K/V rows are written to all slots
 *   [bcast_first_slot, +bcast_n_slots)), the other segments hold only the per-image suffixes.
 *   last_index: int32[n_out] = row of each prompt's last token (n_out = number of real prompts <= n_seq).
 *   score_mode: OWC_PREFILL_LAST_TOKENS - last_index[j] MUST be the last row of sequence j (n_out <= n_seq); the last layer
 *   is then pruned to those rows (they attend every key of their sequence: the wrong thing for any other row, hence the explicit
 *   mode instead of an inference from the counts).  OWC_PREFILL_SCORE_ROWS - last_index names ANY n_out <= T packed rows, the
 *   positions whose logits a loglikelihood request needs (reference src/models/_llava_hf.py:243-252 reads outputs["logits"] of
 *   every position); the last layer then runs on every row and the workspace must be sized with owc_llm_workspace_bytes(w, T, n_out).
 * Writes the KV cache and next_tok[n_out] = argmax of those rows' logits - or, with `sampling`, one draw per row at step 0
 * (row j of this launch is original batch row sampling_row0 + j for the random stream).
 * `logits_out` (optional, [n_out, vocab] bf16) receives the logits. */
enum { OWC_PREFILL_LAST_TOKENS = 0, OWC_PREFILL_SCORE_ROWS = 1 };
int owc_llm_prefill(owc_ctx* ctx, const owc_llm_weights* w, const owc_kv_cache* cache,
                    const int32_t* ids, const int32_t* img_index, const void* img_embeds,
                    const int32_t* pos3, const int32_t* tok_slot, const int32_t* tok_idx,
                    const int32_t* seq_start, const int32_t* seq_len, const int32_t* q_len,
                    const int32_t* k_start, const int32_t* last_index, int n_seq, int n_out, int T,
                    int max_len, int bcast_first_slot, int bcast_n_slots, int score_mode, const owc_sampling* sampling,
                    int sampling_row0, int32_t* next_tok, void* logits_out, void* workspace, size_t ws_bytes, void* stream);

/* One greedy decode step for B sequences (HF GenerationMixin loop body, one token each):
 *   tok_io: int32[B] in = tokens to feed, out = next tokens (pad once a sequence is done);
 *   pos: int32[B] rope position of the fed token (same for the 3 mrope streams);
 *   slot/write_idx/k_start/k_len/q_start/o_start/q_len: int32[B] cache addressing
 *     (k_len = write_idx + 1; q_start[b] = b * (Hq + 2 Hkv); o_start[b] = b * Hq; q_len[b] = Hq / Hkv);
 *   done: uint8[B]; out_tokens[row * out_stride + step] receives the emitted token, row = out_row ? out_row[b] : b.
 *   out_row (optional, int32[B]): the ORIGINAL batch row of compact row b.  In the reference every image is its own
 *     `generate` call and stops at its own EOS (src/models/_qwen2_vl.py:319-337); here the batch decodes together, and
 *     once sequences have finished the caller may drop their rows (owc_decode_compact) so that a step's GEMMs and
 *     attention grid shrink to the live rows.  Every kernel of the step computes a row independently of its
 *     neighbours, so the tokens of the surviving rows are bit-identical to the uncompacted run (tested).
 *   forced_tok (optional, int32 indexed by ORIGINAL row): teacher forcing - the token fed to the next step, and the
 *     one whose EOS finishes the sequence, is forced_tok[row]; out_tokens still receives the step's own argmax.
 *   sampling (optional): the next token is a draw (owc_sampling) at step = the output column instead of the argmax.
 *   step_state (optional, int32[1] on the device): when non-NULL the output column is read from step_state[0] (`step` is
 *     ignored) and, after the step, pos[b], write_idx[b], k_len[b] and step_state[0] are incremented IN PLACE, so that
 *     consecutive decode steps are byte-identical launch sequences: capture one in a hipGraph and replay it. */
int owc_llm_decode_step(owc_ctx* ctx, const owc_llm_weights* w, const owc_kv_cache* cache,
                        int32_t* tok_io, int32_t* pos, const int32_t* slot,
                        int32_t* write_idx, const int32_t* k_start, int32_t* k_len,
                        const int32_t* q_start, const int32_t* o_start, const int32_t* q_len,
                        uint8_t* done, int32_t* out_tokens, int out_stride, int step, int32_t* step_state, int B,
                        int eos_id0, int eos_id1, int pad_id, const int32_t* out_row, const int32_t* forced_tok,
                        const owc_sampling* sampling, void* logits_out, void* workspace, size_t ws_bytes, void* stream);

/* Repetition penalty of the generation(s) that follow on this context (HF RepetitionPenaltyLogitsProcessor; in force in the
 * reference whenever the checkpoint's generation_config.json carries `repetition_penalty` != 1 - HF merges that file into every
 * `generate` call, greedy ones included: /root/reference/src/models/_qwen2_vl.py:319-329 does not override it; Qwen2-VL /
 * Qwen2.5-VL instruct checkpoints ship 1.05).  Semantics: before the argmax (or the sampling warpers) of a sequence's next-token
 * logits, in fp32, every token id that occurs in its prompt (image placeholders included) or was fed to it since gets
 * score = score < 0 ? score * penalty : score / penalty, once per distinct id.
 *   seen: caller-owned device bitmap [n_slots][words_per_row] of uint32 (words_per_row * 32 >= vocab), one row per KV-cache SLOT,
 *   ZEROED by the caller before a sequence's prefill; owc_llm_prefill marks the prompt ids (tok_slot; shared-prefix rows in every
 *   slot of the launch), owc_llm_decode_step marks the token it is fed (slot[b]) - under teacher forcing the forced token, as HF's
 *   input_ids would hold - and both apply the penalty through the bitmap row of the sequence.  Greedy: exact fp32 arithmetic on
 *   the fly inside the argmax.  Sampled (`sampling` != NULL): the penalised values are written back as bf16 in front of the draw.
 *   seen == NULL switches it off (the default).  Not applied in OWC_PREFILL_SCORE_ROWS mode (loglikelihood reads raw logits). */
int owc_llm_set_repetition_penalty(owc_ctx* ctx, float penalty, uint32_t* seen, int words_per_row);
/* the two pieces on their own (op-level tests): mark ids[t] in bitmap row slot[t] (slot NULL: row t); penalised greedy argmax of
 * bf16 logits rows through bitmap rows row_slot[r] (NULL: row r), lowest index on ties. */
int owc_seen_mark(owc_ctx* ctx, const int32_t* ids, const int32_t* slot, int n, int vocab, uint32_t* seen, int words_per_row, void* stream);
int owc_argmax_penalized_bf16(owc_ctx* ctx, const void* logits, int64_t ld, int rows, int vocab, const uint32_t* seen, int words_per_row,
                              const int32_t* row_slot, float penalty, int32_t* out, void* stream);

/* first-token bookkeeping after prefill: same done/pad/out_tokens update as a decode step. */
int owc_decode_update(owc_ctx* ctx, int32_t* next_tok, uint8_t* done, int32_t* out_tokens,
                      int out_stride, int step, int B, int eos_id0, int eos_id1, int pad_id,
                      const int32_t* out_row, const int32_t* forced_tok, void* stream);

/* EOS-aware row compaction between decode steps: the per-row decode state of the n_live surviving rows
 * (live: int32[n_live], ascending indices into the CURRENT rows) is gathered into rows 0..n_live-1 of a second
 * set of buffers (`*_c`; a gather cannot run in place, the caller ping-pongs two sets).  q_start / o_start / q_len
 * depend on the compact row index only, so their first n_live entries stay valid; the KV cache is not moved
 * (slot / k_start indirection). */
int owc_decode_compact(owc_ctx* ctx, const int32_t* live, int n_live, const int32_t* tok, const int32_t* pos,
                       const int32_t* write_idx, const int32_t* k_len, const int32_t* slot, const int32_t* k_start,
                       const int32_t* out_row, const uint8_t* done, int32_t* tok_c, int32_t* pos_c, int32_t* write_idx_c,
                       int32_t* k_len_c, int32_t* slot_c, int32_t* k_start_c, int32_t* out_row_c, uint8_t* done_c, void* stream);

/* ---- model level: sentence encoder + cosine scorer (fp32) ------------------------------------- */
typedef struct owc_bert_layer {
  const float *qkv_w, *qkv_b; /* cat(query, key, value) [3h, h] */
  const float *o_w, *o_b, *ln1_w, *ln1_b, *fc1_w, *fc1_b, *fc2_w, *fc2_b, *ln2_w, *ln2_b;
} owc_bert_layer;

typedef struct owc_bert_weights {
  int32_t n_layers, hidden, n_heads, inter, vocab, max_pos;
  float ln_eps;
  const float *word_emb, *pos_emb, *type_emb, *emb_ln_w, *emb_ln_b; /* type_emb: row 0 is added; NULL = none (MPNet) */
  const owc_bert_layer* layers; /* HOST array */
  /* MPNet (HF modeling_mpnet.py: MPNetEmbeddings / MPNetEncoder.compute_position_bias; all-mpnet-base-v2 is BASELINE.json configs[0]'s
   * encoder): a token in column c of a right-padded row takes position embedding c + pos_offset (padding_idx + 1 = 2; BERT: 0), and
   * rel_bias[head][(key column - query column) + rel_span - 1] - the learned [32 buckets][heads] table expanded over the offsets
   * -(rel_span - 1) .. rel_span - 1 by the caller (the bucket function is integer / log bookkeeping) - is added to every scaled score,
   * the same table in every layer.  NULL / 0 / 0 for BERT.  head_dim 32 or 64. */
  const float* rel_bias;
  int32_t rel_span, pos_offset;
} owc_bert_weights;

size_t owc_bert_workspace_bytes(const owc_bert_weights* w, int n, int L);

/* encode_sentence_bert's arithmetic (src/data/pipelines/text/_text.py:197-202): BertModel forward,
 * mask-weighted mean pool (clamp 1e-9), L2 normalisation.  ids/mask: int32[n][L]; out fp32 [n][hidden]. */
int owc_bert_embed(owc_ctx* ctx, const owc_bert_weights* w, const int32_t* ids, const int32_t* mask,
                   int n, int L, float* out, void* workspace, size_t ws_bytes, void* stream);

/* The same encoder over PACKED rows: only the tokens with attention_mask == 1 are rows (tok_ids / tok_pos: int32[T], tok_pos =
 * the token's column in the padded matrix = its BERT position id; sequence s owns rows [seq_start[s], seq_start[s+1]),
 * seq_start: int32[n + 1]; max_len = longest sequence).  Same results as owc_bert_embed (to fp32 rounding, tested <= 1e-6): in the reference
 * (`padding=True`, _text.py:193-196) padded positions are masked out as keys and weigh 0 in the pooling, so their rows feed
 * nothing; here they are simply not computed (labels of 2..16 tokens padded to 16 are ~45 % padding). */
size_t owc_bert_packed_workspace_bytes(const owc_bert_weights* w, int T);
int owc_bert_embed_packed(owc_ctx* ctx, const owc_bert_weights* w, const int32_t* tok_ids, const int32_t* tok_pos,
                          const int32_t* seq_start, int n, int T, int max_len, float* out, void* workspace, size_t ws_bytes,
                          void* stream);

/* Cosine scorer: sim = preds[N,D] . classes[C,D]^T (rows already L2-normalised), never materialised:
 * top_val/top_idx[N][k] = k best classes per prediction (descending, lowest index on ties),
 * paired[N] = sim[i, label[i]] — the reference's torch.bmm pairing
 * (src/data/metrics/_group.py:537-544).  label / paired / top_* may be NULL. k <= 16. */
int owc_cosine_topk(owc_ctx* ctx, const float* preds, const float* classes, const int32_t* label,
                    int N, int C, int D, int k, float* top_val, int32_t* top_idx, float* paired,
                    void* stream);

/* paired[i] = <a[i], b[i]> for two [N,D] fp32 matrices (torch.bmm(refs[N,1,D], preds[N,D,1])). */
int owc_paired_dot(owc_ctx* ctx, const float* a, const float* b, int N, int D, float* out,
                   void* stream);

#ifdef __cplusplus
}
#endif
#endif /* OWC_H_ */
