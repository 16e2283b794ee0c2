// C ABI of libowc_hip.so (declared in include/owc.h): lifetime + op-level entry points.
// The model-level entries live in qwen2vl.hip / bert.hip.
#include "../../include/owc.h"
#include "owc_internal.h"
#include <string.h>
#include <cstdio>
#include <cstdlib>

#define ST(s) ((hipStream_t)(s))
#define RET(ctx, name, rc)                                                        \
  do {                                                                            \
    int rc__ = (rc);                                                              \
    if (rc__ != OWC_OK) (ctx)->err = name ": bad shape/argument or launch failure"; \
    return rc__;                                                                  \
  } while (0)

extern "C" {

int owc_abi_version(void) { return 19; }

int owc_has_timing_knobs(void) {   // 1 only in libowc_hip_timing.so (tools/); the product library answers 0
#ifdef OWC_TIMING_KNOBS
  return 1;
#else
  return 0;
#endif
}

int owc_tuning_set(const char* name, int value) {
  if (!name) return OWC_ERR_ARG;
  if (!strcmp(name, "gemm_mid_max_tiles")) {
    owc_gemm_set_mid_max_tiles(value);
    owc_gemm_fp8_set_mid_max_tiles(value);
  }
  else if (!strcmp(name, "gemm_skinny_max_m")) {
    owc_gemm_set_skinny_max_m(value);
    owc_gemm_fp8_set_skinny_max_m(value);
  }
  else if (!strcmp(name, "gemm_big_min_m")) owc_gemm_set_big_min_m(value);
  else if (!strcmp(name, "gemm_big_min_tiles")) owc_gemm_set_big_min_tiles(value);
  else if (!strcmp(name, "gemm_pingpong")) {  // 0: lock-step kernels, 1: bf16 ping-pong only, 2 or negative (default): bf16 and fp8 ping-pong
    owc_gemm_set_pingpong(value != 0);
    owc_gemm_fp8_set_pingpong(value >= 2 || value < 0);
  }
#ifdef OWC_TIMING_KNOBS   // timing-only experiments: the product library does not know these names (OWC_ERR_ARG)
  else if (!strcmp(name, "gemm_dbg")) owc_gemm_set_dbg(value);
  else if (!strcmp(name, "attn_dbg")) owc_attn_set_dbg(value);
#endif
  else if (!strcmp(name, "gemm_skinny_deep")) owc_gemm_set_skinny_deep(value);
  else if (!strcmp(name, "decode_norm_fuse_ring")) owc_gemm_set_norm_fuse_ring(value);   // rows (<= 8) up to which the ring kernel folds the RMSNorm in; 0 = off
  else if (!strcmp(name, "gemm_k_pairs")) owc_gemm_set_k_pairs(value);   // 0: one K-tile per stage also for long K; 2: two instead of four
  else if (!strcmp(name, "gemm_k_pairs_min_k")) owc_gemm_set_k_pairs_min_k(value);
  else if (!strcmp(name, "gemm_wide_tiles")) owc_gemm_set_tall_tiles(value);   // 0: no 64x160 / 128x160 ring tiles for <= 128 rows x many columns
  else if (!strcmp(name, "gemm_small_tiles")) {
    owc_gemm_set_small_tiles(value);
    owc_gemm_fp8_set_shapes(value != 0 && value != 2);   // (2 forces 64x64 for both dtypes)
  }   // 0: 64x64 tiles only; 1 (default): by block count; 2..4: force 64x64 / 64x32 / 32x32
  else if (!strcmp(name, "gemm_ring_128")) {   // 128x64 ring tiles for several hundred rows x few thousand columns (bf16 and fp8)
    owc_gemm_set_ring_128(value);
    owc_gemm_fp8_set_ring_128(value);
  }
  else if (!strcmp(name, "gemm_nt_min_mb")) owc_gemm_set_nt_min_mb(value);   // streaming C stores for outputs above this many MiB (negative: default)
  else if (!strcmp(name, "gemm_walk")) owc_gemm_set_walk(value);   // 256x256 ping-pong tile order: 0 rows-of-4 walk, 1 (default, negative) column groups when tiles_m >= tiles_n
  else if (!strcmp(name, "gemm_tail_split")) owc_gemm_set_tail_split(value);   // short last round of 256x256 tiles as 256x128 tiles: 0 off, 1 (default, negative) on
  else if (!strcmp(name, "gemm_persist")) owc_gemm_set_persist(value);   // persistent 256x256 ping-pong kernel: 0 off, negative: default
  else if (!strcmp(name, "gemm_pp128")) owc_gemm_set_pp128(value);   // 256x128 ping-pong tiles: 0 off, n > 0: from n tiles, negative: default
  else if (!strcmp(name, "decode_fuse")) owc_llm_set_decode_fuse(value);
  else if (!strcmp(name, "decode_norm_fuse")) owc_gemm_set_norm_fuse_max_m(value);   // max rows (<= 4) for the RMSNorm-fused skinny GEMM; 0 = off
  else if (!strcmp(name, "attn_mfma32")) owc_attn_set_mfma32(value);   // vision attention (head_dim 80 / 64, non-causal): 0 / negative (default) = attn_fwd_kernel (16x16x32 MFMA), 1 = the 32x32x16 kernel, 2 = its software-pipelined form
  else if (!strcmp(name, "attn_gqa_pack")) owc_attn_set_gqa_pack(value);   // causal GQA attention: 0 = one head per block (rounds 1-4), else the heads of a kv group packed into the rows
  else if (!strcmp(name, "decode_attn_nbuf1")) owc_attn_set_decode_nbuf1(value);   // block count above which the fused decode attention single-buffers V
  else if (!strcmp(name, "prefill_prune_last")) owc_llm_set_prune_last(value);
  else if (!strcmp(name, "bert_bf16x3")) owc_bert_set_x3(value);
  else return OWC_ERR_ARG;
  return OWC_OK;
}

static int g_ctx_device = -1;  // one context per process (one process per GPU): kernel attributes, knobs and the profile recording
                               // are process-wide, so a second device in the same process is refused instead of half-working

int owc_init(int device, owc_ctx** out) {
  if (out == nullptr) return OWC_ERR_ARG;
  *out = nullptr;
  if (g_ctx_device >= 0 && g_ctx_device != device) return OWC_ERR_ARG;
  if (hipSetDevice(device) != hipSuccess) return OWC_ERR_HIP;
  owc_ctx* ctx = new owc_ctx();
  ctx->device = device;
  if (hipMalloc(&ctx->zeros, 256) != hipSuccess || hipMemset(ctx->zeros, 0, 256) != hipSuccess) {
    delete ctx;
    return OWC_ERR_HIP;
  }
  // (no environment variable is read here: every knob goes through owc_tuning_set, i.e. through the caller's own code)
#ifdef OWC_TIMING_KNOBS
  fprintf(stderr, "[libowc_hip] this is the TIMING-KNOB build (tools/ only): `gemm_dbg` / `attn_dbg` can switch parts of a kernel off\n");
#endif
  g_ctx_device = device;
  *out = ctx;
  return OWC_OK;
}

int owc_destroy(owc_ctx* ctx) {
  if (ctx == nullptr) return OWC_ERR_ARG;
  g_ctx_device = -1;
  if (ctx->zeros) (void)hipFree(ctx->zeros);
  delete ctx;
  return OWC_OK;
}

const char* owc_last_error(const owc_ctx* ctx) { return ctx ? ctx->err.c_str() : "null ctx"; }

int owc_gemm_bf16(owc_ctx* ctx, const void* A, int64_t lda, const void* W, int64_t ldw,
                  const void* bias, const void* residual, int64_t ldr, void* C, int64_t ldc, int M,
                  int N, int K, int epilogue, void* stream) {
  if (!ctx || !A || !W || !C) return OWC_ERR_ARG;
  RET(ctx, "owc_gemm_bf16",
      owc_launch_gemm_bf16(A, lda, W, ldw, bias, residual, ldr, C, ldc, M, N, K, epilogue, ctx->zeros, ST(stream)));
}

int owc_gemm_f32(owc_ctx* ctx, const float* A, int64_t lda, const float* W, int64_t ldw,
                 const float* bias, const float* residual, int64_t ldr, float* C, int64_t ldc, int M,
                 int N, int K, int epilogue, void* stream) {
  if (!ctx || !A || !W || !C) return OWC_ERR_ARG;
  RET(ctx, "owc_gemm_f32",
      owc_launch_gemm_f32(A, lda, W, ldw, bias, residual, ldr, C, ldc, M, N, K, epilogue, ctx->zeros, ST(stream)));
}

int owc_layernorm_bf16(owc_ctx* ctx, const void* X, int64_t ldx, const void* weight, const void* bias,
                       void* Y, int64_t ldy, int rows, int d, float eps, void* stream) {
  if (!ctx || !X || !weight || !bias || !Y) return OWC_ERR_ARG;
  RET(ctx, "owc_layernorm_bf16", owc_launch_layernorm(X, ldx, weight, bias, Y, ldy, rows, d, eps, ST(stream)));
}

int owc_rmsnorm_bf16(owc_ctx* ctx, const void* X, int64_t ldx, const void* weight, void* Y,
                     int64_t ldy, int rows, int d, float eps, const int32_t* row_index, void* stream) {
  if (!ctx || !X || !weight || !Y) return OWC_ERR_ARG;
  RET(ctx, "owc_rmsnorm_bf16", owc_launch_rmsnorm(X, ldx, weight, Y, ldy, rows, d, eps, row_index, ST(stream)));
}

int owc_rope_table(owc_ctx* ctx, float* cos_t, float* sin_t, int n_pos, int n_freq, int dim,
                   float theta, int round_bf16, void* stream) {
  if (!ctx || !cos_t || !sin_t) return OWC_ERR_ARG;
  RET(ctx, "owc_rope_table", owc_launch_rope_table(cos_t, sin_t, n_pos, n_freq, dim, theta, round_bf16, ST(stream)));
}

int owc_vision_rope(owc_ctx* ctx, void* qkv, int64_t ld, const int32_t* pos_hw, const float* cos_t,
                    const float* sin_t, int T, int n_heads, int head_dim, void* stream) {
  if (!ctx || !qkv || !pos_hw || !cos_t || !sin_t) return OWC_ERR_ARG;
  RET(ctx, "owc_vision_rope", owc_launch_vision_rope(qkv, ld, pos_hw, cos_t, sin_t, T, n_heads, head_dim, ST(stream)));
}

int owc_mrope_kv_write(owc_ctx* ctx, void* qkv, int64_t ld, const int32_t* pos3, int64_t pos_stride,
                       const float* cos_t, const float* sin_t, void* k_cache, void* v_cache,
                       const int32_t* tok_slot, const int32_t* tok_idx, int T, int n_q_heads,
                       int n_kv_heads, int s_max, int mrope_sec0, int mrope_sec1, void* stream) {
  if (!ctx || !qkv || !pos3 || !cos_t || !sin_t || !k_cache || !v_cache || !tok_slot || !tok_idx)
    return OWC_ERR_ARG;
  RET(ctx, "owc_mrope_kv_write",
      owc_launch_mrope_kv(qkv, ld, pos3, pos_stride, cos_t, sin_t, k_cache, v_cache, tok_slot, tok_idx, T,
                          n_q_heads, n_kv_heads, s_max, mrope_sec0, mrope_sec1, 0, 0, ST(stream)));
}

int owc_attention_bf16(owc_ctx* ctx, const void* Q, int64_t q_ts, int64_t q_hs, const void* K,
                       int64_t k_ts, int64_t k_hs, const void* V, int64_t v_ts, int64_t v_hs, void* O,
                       int64_t o_ts, int64_t o_hs, const int32_t* q_start, const int32_t* o_start,
                       const int32_t* k_start, const int32_t* seq_len, const int32_t* q_len, int n_seq,
                       int n_heads, int kv_group, int head_dim, int max_q_len, int causal, float scale,
                       void* stream) {
  if (!ctx || !Q || !K || !V || !O || !q_start || !k_start || !seq_len) return OWC_ERR_ARG;
  RET(ctx, "owc_attention_bf16",
      owc_launch_attention(Q, q_ts, q_hs, K, k_ts, k_hs, V, v_ts, v_hs, O, o_ts, o_hs, q_start, o_start,
                           k_start, seq_len, q_len, n_seq, n_heads, kv_group, head_dim, max_q_len, causal,
                           scale, ST(stream)));
}

int owc_decode_attention(owc_ctx* ctx, const void* qkv, int64_t ld, const int32_t* pos, const float* cos_t, const float* sin_t,
                         void* k_cache, void* v_cache, const int32_t* slot, const int32_t* write_idx, const int32_t* k_len,
                         void* out, int64_t ldo, int B, int n_q_heads, int n_kv_heads, int s_max, float scale, void* stream) {
  if (!ctx || !qkv || !pos || !cos_t || !sin_t || !k_cache || !v_cache || !slot || !write_idx || !k_len || !out) return OWC_ERR_ARG;
  RET(ctx, "owc_decode_attention",
      owc_launch_attn_decode_fused(qkv, ld, pos, cos_t, sin_t, k_cache, v_cache, slot, write_idx, k_len, out, ldo, B, n_q_heads,
                                   n_kv_heads, s_max, scale, ST(stream)));
}

int owc_embed_tokens(owc_ctx* ctx, const int32_t* ids, const int32_t* img_index, const void* table,
                     const void* img_embeds, void* out, int T, int d, void* stream) {
  if (!ctx || !ids || !table || !out) return OWC_ERR_ARG;
  RET(ctx, "owc_embed_tokens", owc_launch_embed(ids, img_index, table, img_embeds, out, T, d, ST(stream)));
}

int owc_argmax_bf16(owc_ctx* ctx, const void* logits, int64_t ld, int rows, int vocab, int32_t* out,
                    void* stream) {
  if (!ctx || !logits || !out) return OWC_ERR_ARG;
  RET(ctx, "owc_argmax_bf16", owc_launch_argmax(logits, ld, rows, vocab, out, ST(stream)));
}

int owc_beam_candidates(owc_ctx* ctx, const void* logits, int64_t ld, int rows, int vocab, int k, float* logz, float* top_val,
                        int32_t* top_idx, void* stream) {
  if (!ctx || !logits || !logz || !top_val || !top_idx) return OWC_ERR_ARG;
  if (rows <= 0 || vocab <= 0 || k <= 0 || k > vocab || k > 64) OWC_FAIL(ctx, OWC_ERR_SHAPE, "owc_beam_candidates: need 1 <= k <= min(64, vocab)");
  RET(ctx, "owc_beam_candidates", owc_launch_beam_candidates(logits, ld, rows, vocab, k, logz, top_val, top_idx, ST(stream)));
}

int owc_sample_bf16(owc_ctx* ctx, const void* logits, int64_t ld, int rows, int vocab, const owc_sampling* sampling,
                    const int32_t* row_map, int step, int32_t* out, void* stream) {
  if (!ctx || !logits || !out || !sampling) return OWC_ERR_ARG;
  if (!(sampling->temperature > 0.f)) OWC_FAIL(ctx, OWC_ERR_ARG, "owc_sample_bf16: temperature must be > 0 (0 = greedy: owc_argmax_bf16)");
  RET(ctx, "owc_sample_bf16", owc_launch_sample(logits, ld, rows, vocab, sampling, row_map, step, nullptr, out, ST(stream)));
}

int owc_token_logprob_bf16(owc_ctx* ctx, const void* logits, int64_t ld, const int32_t* target, int rows, int vocab, float* out,
                           void* stream) {
  if (!ctx || !logits || !target || !out) return OWC_ERR_ARG;
  RET(ctx, "owc_token_logprob_bf16", owc_launch_token_logprob(logits, ld, target, rows, vocab, out, ST(stream)));
}

int owc_patchify_u8(owc_ctx* ctx, const uint8_t* images, void* pixel_values, int64_t ld, int n, int H,
                    int W, const float* mean_host, const float* std_host, void* stream) {
  if (!ctx || !images || !pixel_values || !mean_host || !std_host) return OWC_ERR_ARG;
  RET(ctx, "owc_patchify_u8", owc_launch_patchify(images, pixel_values, ld, n, H, W, mean_host, std_host, ST(stream)));
}

int owc_quantize_rows_fp8(owc_ctx* ctx, const void* x, int64_t ldx, void* q, int64_t ldq, float* scale, int rows,
                          int cols, void* stream) {
  if (!ctx || !x || !q || !scale) return OWC_ERR_ARG;
  RET(ctx, "owc_quantize_rows_fp8", owc_launch_quant_rows_fp8(x, ldx, q, ldq, scale, rows, cols, ST(stream)));
}

int owc_rmsnorm_quant_fp8(owc_ctx* ctx, const void* x, int64_t ldx, const void* weight, void* q, int64_t ldq, float* scale,
                          int rows, int d, float eps, void* stream) {
  if (!ctx || !x || !weight || !q || !scale) return OWC_ERR_ARG;
  RET(ctx, "owc_rmsnorm_quant_fp8", owc_launch_rmsnorm_quant_fp8(x, ldx, weight, q, ldq, scale, rows, d, eps, ST(stream)));
}

int owc_gemm_fp8(owc_ctx* ctx, const void* A, int64_t lda, const float* a_scale, const void* W, int64_t ldw,
                 const float* w_scale, const void* bias, const void* R, int64_t ldr, void* C, int64_t ldc, int M, int N,
                 int K, int epilogue, void* stream) {
  if (!ctx || !A || !W || !C || !a_scale || !w_scale) return OWC_ERR_ARG;
  RET(ctx, "owc_gemm_fp8",
      owc_launch_gemm_fp8(A, lda, a_scale, W, ldw, w_scale, bias, R, ldr, C, ldc, M, N, K, epilogue, ST(stream)));
}

int owc_decode_update(owc_ctx* ctx, int32_t* next_tok, uint8_t* done, int32_t* out_tokens,
                      int out_stride, int step, int B, int eos_id0, int eos_id1, int pad_id,
                      const int32_t* out_row, const int32_t* forced_tok, void* stream) {
  if (!ctx || !next_tok || !done || !out_tokens) return OWC_ERR_ARG;
  RET(ctx, "owc_decode_update",
      owc_launch_decode_update(next_tok, done, out_tokens, out_stride, step, nullptr, B, eos_id0, eos_id1, pad_id, out_row,
                               forced_tok, ST(stream)));
}

int owc_decode_compact(owc_ctx* ctx, const int32_t* live, int n_live, const int32_t* tok, const int32_t* pos,
                       const int32_t* write_idx, const int32_t* k_len, const int32_t* slot, const int32_t* k_start,
                       const int32_t* out_row, const uint8_t* done, int32_t* tok_c, int32_t* pos_c, int32_t* write_idx_c,
                       int32_t* k_len_c, int32_t* slot_c, int32_t* k_start_c, int32_t* out_row_c, uint8_t* done_c, void* stream) {
  if (!ctx || !live || !tok || !pos || !write_idx || !k_len || !slot || !k_start || !out_row || !done || !tok_c || !pos_c ||
      !write_idx_c || !k_len_c || !slot_c || !k_start_c || !out_row_c || !done_c)
    return OWC_ERR_ARG;
  const int32_t* src[7] = {tok, pos, write_idx, k_len, slot, k_start, out_row};
  int32_t* dst[7] = {tok_c, pos_c, write_idx_c, k_len_c, slot_c, k_start_c, out_row_c};
  for (int a = 0; a < 7; ++a)
    if (src[a] == dst[a]) OWC_FAIL(ctx, OWC_ERR_ARG, "owc_decode_compact: a gather cannot run in place");
  if ((const uint8_t*)done_c == done) OWC_FAIL(ctx, OWC_ERR_ARG, "owc_decode_compact: a gather cannot run in place");
  RET(ctx, "owc_decode_compact", owc_launch_decode_compact(src, dst, done, done_c, live, n_live, ST(stream)));
}

int owc_gemm_profile_enable(owc_ctx* ctx, int on) {
  if (!ctx) return OWC_ERR_ARG;
  owc_gemm_profile_set(on);
  return OWC_OK;
}

int owc_gemm_profile_read(owc_ctx* ctx, double* total_ms, double* total_flops, int64_t* launches) {
  if (!ctx || !total_ms || !total_flops || !launches) return OWC_ERR_ARG;
  double ms[2], fl[2];
  long n[2] = {0, 0};
  int rc = owc_gemm_profile_collect(ms, fl, n);
  for (int k = 0; k < 2; ++k) {
    total_ms[k] = ms[k];
    total_flops[k] = fl[k];
    launches[k] = n[k];
  }
  RET(ctx, "owc_gemm_profile_read", rc);
}

int owc_profile_read(owc_ctx* ctx, int n_kinds, double* total_ms, double* total_work, int64_t* launches) {
  if (!ctx || !total_ms || !total_work || !launches || n_kinds <= 0 || n_kinds > OWC_PROF_KINDS) return OWC_ERR_ARG;
  long n[OWC_PROF_KINDS];
  const int rc = owc_profile_collect(n_kinds, total_ms, total_work, n);
  for (int k = 0; k < n_kinds; ++k) launches[k] = n[k];
  RET(ctx, "owc_profile_read", rc);
}

int owc_profile_shapes(owc_ctx* ctx, int max_n, int32_t* shape, double* stats) {
  if (!ctx || !shape || !stats || max_n <= 0) return -1;
  return owc_profile_shapes_collect(max_n, shape, stats);
}

}  // extern "C"
