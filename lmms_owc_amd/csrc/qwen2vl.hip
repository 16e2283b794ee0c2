// Host-side drivers of the Qwen2-VL forward pass: they only enqueue the hand-written kernels of this
// library on the caller's stream, in the order HF transformers runs the corresponding modules.
//   owc_vit_forward     <- Qwen2VisionTransformerPretrainedModel.forward (HF:700-731)
//   owc_llm_prefill     <- Qwen2VLModel.forward + Qwen2VLTextModel.forward + lm_head (HF:1144-1205, :762-846)
//   owc_llm_decode_step <- one iteration of GenerationMixin's greedy loop with the KV cache
// No allocation, no synchronisation: graph-capturable.
#include "../../include/owc.h"
#include "owc_internal.h"

// Pre-LN transformer encoder layers shared by the Qwen2-VL vision tower (aux != NULL: 2-D RoPE fused into
// the qkv epilogue, HF:442-450) and the CLIP tower of LLaVA (aux == NULL; modeling_clip.py CLIPEncoderLayer).
int owc_vit_layers(owc_ctx* ctx, const owc_vit_layer* layers, int n_layers, void* x, void* h, void* attn,
                   void* qkv, void* mlp, int T, int E, int H, int F, float eps, const int32_t* seq_start,
                   const int32_t* seq_len, int n_img, int max_len, const owc_gemm_aux* aux, hipStream_t st,
                   const owc_vit_windows* win) {
  const int hd = E / H;
  const float scale = 1.0f / sqrtf((float)hd);
  const bool v25 = win != nullptr;   // Qwen2.5-VL blocks: RMSNorm, gated MLP, window attention
  auto norm = [&](const void* gamma, const void* beta) -> int {
    if (v25) return owc_launch_rmsnorm(x, E, gamma, h, E, T, E, eps, nullptr, st);
    return owc_launch_layernorm(x, E, gamma, beta, h, E, T, E, eps, st);
  };
  for (int i = 0; i < n_layers; ++i) {
    const owc_vit_layer& L = layers[i];
    // x = x + proj(attn(rope(qkv(norm1(x)))))
    OWC_TRY(norm(L.ln1_w, L.ln1_b));
    if (aux) {  // q/k rows of qkv_w are pair-interleaved, rotation in the epilogue
      OWC_TRY(owc_launch_gemm_bf16_aux(h, E, L.qkv_w, E, L.qkv_b, nullptr, 0, qkv, 3 * E, T, 3 * E, E,
                                       OWC_EPI_VROPE, ctx->zeros, st, aux));
    } else {
      OWC_TRY(owc_launch_gemm_bf16(h, E, L.qkv_w, E, L.qkv_b, nullptr, 0, qkv, 3 * E, T, 3 * E, E,
                                   OWC_EPI_NONE, ctx->zeros, st));
    }
    const bf16_t* q = (const bf16_t*)qkv;
    if (v25 && !((win->fullatt_mask >> i) & 1)) {   // HF qwen2_5_vl:448-454: a window layer attends inside each window only
      OWC_TRY(owc_launch_attention(q, 3 * E, hd, q + E, 3 * E, hd, q + 2 * E, 3 * E, hd, attn, E, hd,
                                   win->start, nullptr, win->start, win->len, nullptr, win->n, H, 1, hd,
                                   win->max_len, 0, scale, st));
    } else {
      OWC_TRY(owc_launch_attention(q, 3 * E, hd, q + E, 3 * E, hd, q + 2 * E, 3 * E, hd, attn, E, hd,
                                   seq_start, nullptr, seq_start, seq_len, nullptr, n_img, H, 1, hd,
                                   max_len, 0, scale, st));
    }
    OWC_TRY(owc_launch_gemm_bf16(attn, E, L.proj_w, E, L.proj_b, x, E, x, E, T, E, E, OWC_EPI_RESIDUAL,
                                 ctx->zeros, st));
    // x = x + fc2(quick_gelu(fc1(norm2(x))))      |  Qwen2.5: x + down(silu(gate(norm2(x))) * up(norm2(x)))
    OWC_TRY(norm(L.ln2_w, L.ln2_b));
    if (v25)
      OWC_TRY(owc_launch_gemm_bf16(h, E, L.fc1_w, E, L.fc1_b, nullptr, 0, mlp, F, T, 2 * F, E, OWC_EPI_SWIGLU, ctx->zeros, st));
    else
      OWC_TRY(owc_launch_gemm_bf16(h, E, L.fc1_w, E, L.fc1_b, nullptr, 0, mlp, F, T, F, E,
                                   OWC_EPI_QUICK_GELU, ctx->zeros, st));
    OWC_TRY(owc_launch_gemm_bf16(mlp, F, L.fc2_w, F, L.fc2_b, x, E, x, E, T, E, F, OWC_EPI_RESIDUAL,
                                 ctx->zeros, st));
  }
  return OWC_OK;
}

extern "C" {

size_t owc_vit_workspace_bytes(const owc_vit_weights* w, int T) {
  if (!w || T <= 0) return 0;
  const size_t e = (size_t)w->embed_dim, t = (size_t)T;
  size_t b = 0;
  b += owc_align256(t * e * 2) * 3;                 // x, h, attn
  b += owc_align256(t * e * 3 * 2);                 // qkv
  b += owc_align256(t * (size_t)w->mlp_hidden * 2); // mlp hidden (also merger hidden)
  return b + 1024;
}

int owc_vit_forward(owc_ctx* ctx, const owc_vit_weights* w, const void* pixel_values, int64_t ld_pix,
                    const int32_t* pos_hw, const int32_t* seq_start, const int32_t* seq_len,
                    int n_img, int T, int max_len, int max_pos_hw, void* out, void* workspace, size_t ws_bytes,
                    void* stream) {
  if (!ctx || !w || !pixel_values || !pos_hw || !seq_start || !seq_len || !out || !workspace)
    return OWC_ERR_ARG;
  if (T <= 0 || n_img <= 0 || (T % w->merge_unit) != 0) OWC_FAIL(ctx, OWC_ERR_SHAPE, "owc_vit_forward: bad T");
  if (max_pos_hw <= 0 || max_pos_hw > w->rope_positions)
    OWC_FAIL(ctx, OWC_ERR_SHAPE, "owc_vit_forward: a grid side exceeds the vision rotary table (rope_positions)");
  if (ws_bytes < owc_vit_workspace_bytes(w, T)) OWC_FAIL(ctx, OWC_ERR_WORKSPACE, "owc_vit_forward: workspace too small");
  hipStream_t st = (hipStream_t)stream;
  const int E = w->embed_dim, H = w->num_heads, hd = E / H, F = w->mlp_hidden;
  Carver cv(workspace, ws_bytes);
  void* x = cv.take((size_t)T * E * 2);
  void* h = cv.take((size_t)T * E * 2);
  void* attn = cv.take((size_t)T * E * 2);
  void* qkv = cv.take((size_t)T * E * 3 * 2);
  void* mlp = cv.take((size_t)T * F * 2);
  const owc_gemm_aux aux = {pos_hw, w->rope_cos, w->rope_sin, 2 * E, hd};

  // patch embed: Conv3d(kernel == stride) == GEMM [T, patch_k] x [E, patch_k]^T, no bias (HF:268-275)
  OWC_TRY(owc_launch_gemm_bf16(pixel_values, ld_pix, w->patch_w, w->patch_k, nullptr, nullptr, 0, x, E,
                               T, E, w->patch_k, OWC_EPI_NONE, ctx->zeros, st));
  OWC_TRY(owc_vit_layers(ctx, w->layers, w->depth, x, h, attn, qkv, mlp, T, E, H, F, w->ln_eps, seq_start,
                         seq_len, n_img, max_len, &aux, st, nullptr));
  // PatchMerger (HF:288-291): ln_q, view [T/4, 4E], Linear+GELU, Linear
  const int M = T / w->merge_unit, E4 = E * w->merge_unit;
  OWC_TRY(owc_launch_layernorm(x, E, w->merger_ln_w, w->merger_ln_b, h, E, T, E, w->ln_eps, st));
  OWC_TRY(owc_launch_gemm_bf16(h, E4, w->merger_fc1_w, E4, w->merger_fc1_b, nullptr, 0, mlp, E4, M, E4,
                               E4, OWC_EPI_GELU_ERF, ctx->zeros, st));
  OWC_TRY(owc_launch_gemm_bf16(mlp, E4, w->merger_fc2_w, E4, w->merger_fc2_b, nullptr, 0, out,
                               w->out_dim, M, w->out_dim, E4, OWC_EPI_NONE, ctx->zeros, st));
  return OWC_OK;
}

int owc_vit25_forward(owc_ctx* ctx, const owc_vit_weights* w, const void* pixel_values, int64_t ld_pix,
                      const int32_t* pos_hw, const int32_t* tok_index, const int32_t* out_index, const int32_t* seq_start,
                      const int32_t* seq_len, int n_img, int max_len, const int32_t* win_start, const int32_t* win_len, int n_win,
                      int max_win_len, int T, int max_pos_hw, void* out, void* workspace, size_t ws_bytes, void* stream) {
  if (!ctx || !w || !pixel_values || !pos_hw || !tok_index || !out_index || !seq_start || !seq_len || !win_start || !win_len ||
      !out || !workspace)
    return OWC_ERR_ARG;
  if (w->variant != 1) OWC_FAIL(ctx, OWC_ERR_ARG, "owc_vit25_forward: weights are not the Qwen2.5-VL variant");
  if (T <= 0 || n_img <= 0 || n_win <= 0 || (T % w->merge_unit) != 0) OWC_FAIL(ctx, OWC_ERR_SHAPE, "owc_vit25_forward: bad T");
  if ((w->mlp_hidden % 128) != 0) OWC_FAIL(ctx, OWC_ERR_SHAPE, "owc_vit25_forward: mlp_hidden must be padded to a multiple of 128");
  if (max_pos_hw <= 0 || max_pos_hw > w->rope_positions)
    OWC_FAIL(ctx, OWC_ERR_SHAPE, "owc_vit25_forward: a grid side exceeds the vision rotary table (rope_positions)");
  if (ws_bytes < owc_vit_workspace_bytes(w, T)) OWC_FAIL(ctx, OWC_ERR_WORKSPACE, "owc_vit25_forward: workspace too small");
  hipStream_t st = (hipStream_t)stream;
  const int E = w->embed_dim, H = w->num_heads, hd = E / H, F = w->mlp_hidden;
  Carver cv(workspace, ws_bytes);
  void* x = cv.take((size_t)T * E * 2);
  void* h = cv.take((size_t)T * E * 2);
  void* attn = cv.take((size_t)T * E * 2);
  void* qkv = cv.take((size_t)T * E * 3 * 2);
  void* mlp = cv.take((size_t)T * F * 2);
  if (F < E || w->out_dim > 4 * E) OWC_FAIL(ctx, OWC_ERR_SHAPE, "owc_vit25_forward: needs mlp_hidden >= embed_dim and out_dim <= 4 embed_dim (buffers are reused)");
  const owc_gemm_aux aux = {pos_hw, w->rope_cos, w->rope_sin, 2 * E, hd};
  const owc_vit_windows win = {win_start, win_len, n_win, max_win_len, w->fullatt_mask};

  // patch embed in the ORIGINAL order (into h), then the window re-order (HF qwen2_5_vl:437-440)
  OWC_TRY(owc_launch_gemm_bf16(pixel_values, ld_pix, w->patch_w, w->patch_k, nullptr, nullptr, 0, h, E, T, E, w->patch_k,
                               OWC_EPI_NONE, ctx->zeros, st));
  OWC_TRY(owc_launch_gather_rows(h, E, tok_index, x, E, T, E, st));
  OWC_TRY(owc_vit_layers(ctx, w->layers, w->depth, x, h, attn, qkv, mlp, T, E, H, F, w->ln_eps, seq_start, seq_len, n_img,
                         max_len, &aux, st, &win));
  // Qwen2_5_VLPatchMerger (HF qwen2_5_vl:137-150): RMSNorm, view [T/4, 4E], Linear + GELU, Linear; rows back in the original order
  const int M = T / w->merge_unit, E4 = E * w->merge_unit;
  OWC_TRY(owc_launch_rmsnorm(x, E, w->merger_ln_w, h, E, T, E, w->ln_eps, nullptr, st));
  OWC_TRY(owc_launch_gemm_bf16(h, E4, w->merger_fc1_w, E4, w->merger_fc1_b, nullptr, 0, mlp, E4, M, E4, E4, OWC_EPI_GELU_ERF,
                               ctx->zeros, st));
  OWC_TRY(owc_launch_gemm_bf16(mlp, E4, w->merger_fc2_w, E4, w->merger_fc2_b, nullptr, 0, x, w->out_dim, M, w->out_dim, E4,
                               OWC_EPI_NONE, ctx->zeros, st));
  OWC_TRY(owc_launch_gather_rows(x, w->out_dim, out_index, out, w->out_dim, M, w->out_dim, st));
  return OWC_OK;
}

size_t owc_llm_workspace_bytes(const owc_llm_weights* w, int T, int n_seq) {
  if (!w || T <= 0 || n_seq <= 0) return 0;
  const size_t t = (size_t)T, d = (size_t)w->d_model;
  const size_t qkv = (size_t)(w->n_q_heads + 2 * w->n_kv_heads) * w->head_dim;
  size_t b = 0;
  b += owc_align256(t * d * 2) * 2;                                   // x, h
  b += owc_align256(t * qkv * 2);                                     // qkv
  b += owc_align256(t * (size_t)w->n_q_heads * w->head_dim * 2);      // attn
  b += owc_align256(t * (size_t)w->d_ff * 2);                         // mlp
  b += owc_align256((size_t)n_seq * d * 2) * 2;                       // last-token residual rows, last-token hidden
  b += owc_align256((size_t)n_seq * 4) * 3;                           // last-token attention index triple
  b += owc_align256((size_t)n_seq * (size_t)w->vocab * 2);            // logits
  if (w->weight_dtype == OWC_WEIGHTS_FP8) {
    size_t widest = d > (size_t)w->d_ff ? d : (size_t)w->d_ff;
    if ((size_t)w->n_q_heads * w->head_dim > widest) widest = (size_t)w->n_q_heads * w->head_dim;
    b += owc_align256(t * widest) + owc_align256(t * 4);            // per-token e4m3 codes + scales
  }
  return b + 1024;
}

static int g_decode_fuse = 1;  // owc_tuning_set("decode_fuse", 0): decode steps run mrope_kv_kernel + the generic attention kernel (A-B / parity)
static int g_prune_last = 1;  // owc_tuning_set("prefill_prune_last", 0): full last layer (the A side of the bit-identity test)

// `last_index` != NULL (prefill): only the last token of the first `n_out` sequences feeds the lm_head, so the LAST layer
// computes K/V (and the KV-cache write) for every row but attention, o-proj and the MLP for those n_out rows only, into the
// compact residual rows `xl` -- the same values the full computation leaves in x[last_index[j]] (every kernel accumulates a
// row independently of its neighbours), minus 1/n_layers of the prefill's o-proj + MLP work.
static int llm_layers(owc_ctx* ctx, const owc_llm_weights* w, const owc_kv_cache* cache, void* x,
                      void* h, void* qkv, void* attn, void* mlp, void* q8, float* qs, const int32_t* pos3,
                      int64_t pos_stride, const int32_t* tok_slot, const int32_t* tok_idx,
                      const int32_t* q_start, const int32_t* o_start, const int32_t* k_start,
                      const int32_t* k_len, const int32_t* q_len, int n_seq, int T, int max_q_len,
                      bool decode, int bcast_first, int bcast_n, const int32_t* last_index, int n_out, void* xl,
                      int32_t* idx3, hipStream_t st) {
  const int d = w->d_model, Hq = w->n_q_heads, Hkv = w->n_kv_heads, hd = w->head_dim, F = w->d_ff;
  const int NQKV = (Hq + 2 * Hkv) * hd;
  const int G = Hq / Hkv;
  const float scale = 1.0f / sqrtf((float)hd);
  const size_t layer_elems = (size_t)cache->n_slots * Hkv * cache->s_max * hd;
  const bool fp8 = w->weight_dtype == OWC_WEIGHTS_FP8;
  // one decoder projection over M rows: bf16 weights -> owc_gemm_bf16; fp8 weights -> per-token quantisation of the input
  // rows, then the scaled-fp8-MFMA GEMM with the same fused epilogue
  // (`in` == NULL: the rows are already in q8 / qs, put there by the fused RMSNorm + quantise kernel)
  auto linear = [&](int M, const void* in, long ld_in, const void* wt, const float* ws, const void* bias, const void* res,
                    void* out, long ld_out, int N, int K, int epi) -> int {
    if (!fp8) return owc_launch_gemm_bf16(in, ld_in, wt, K, bias, res, ld_out, out, ld_out, M, N, K, epi, ctx->zeros, st);
    if (in) OWC_TRY(owc_launch_quant_rows_fp8(in, ld_in, q8, K, qs, M, K, st));
    return owc_launch_gemm_fp8(q8, K, qs, wt, K, ws, bias, res, ld_out, out, ld_out, M, N, K, epi, st);
  };
  auto norm = [&](int M, const void* xin, const void* gamma) -> int {  // h = rmsnorm(x) (bf16), or straight to e4m3 codes
    if (fp8) return owc_launch_rmsnorm_quant_fp8(xin, d, gamma, q8, d, qs, M, d, w->rms_eps, st);
    return owc_launch_rmsnorm(xin, d, gamma, h, d, M, d, w->rms_eps, nullptr, st);
  };
  for (int i = 0; i < w->n_layers; ++i) {
    const owc_llm_layer& L = w->layers[i];
    bf16_t* kc = (bf16_t*)cache->k + (size_t)i * layer_elems;
    bf16_t* vc = (bf16_t*)cache->v + (size_t)i * layer_elems;
    const bool tail = last_index && g_prune_last && i == w->n_layers - 1;  // last prefill layer: only the n_out last-token rows go on
    // self-attention block (HF:601-614); a handful of rows (decode at the reference's batch size): the projection normalises its own
    // activations (gemm_bf16_skinny_norm_kernel) - same bits as the two launches, one launch less
    // (OWC_ERR_SHAPE = "not a shape the fused form takes": the two launches below; any other failure is a real one)
    int rc_f = fp8 ? OWC_ERR_SHAPE
                   : owc_launch_gemm_bf16_rmsnorm(x, d, L.ln1_w, w->rms_eps, L.qkv_w, d, L.qkv_b, qkv, NQKV, T, NQKV, d, OWC_EPI_NONE, st);
    if (rc_f != OWC_OK && rc_f != OWC_ERR_SHAPE) return rc_f;
    if (rc_f == OWC_ERR_SHAPE) {
      OWC_TRY(norm(T, x, L.ln1_w));
      OWC_TRY(linear(T, fp8 ? nullptr : h, d, L.qkv_w, L.qkv_s, L.qkv_b, nullptr, qkv, NQKV, NQKV, d, OWC_EPI_NONE));
    }
    const bool fused_decode = decode && g_decode_fuse && hd == 128 && G <= 16;
    if (!fused_decode)
      OWC_TRY(owc_launch_mrope_kv(qkv, NQKV, pos3, pos_stride, w->rope_cos, w->rope_sin, kc, vc, tok_slot,
                                  tok_idx, T, Hq, Hkv, cache->s_max, w->mrope_sec0, w->mrope_sec1, bcast_first, bcast_n, st));
    int M = T;
    void* xr = x;
    if (tail) {
      // the last token sees every key of its sequence: the decode mapping below, one "sequence" per output
      int32_t *qs_l = idx3, *os_l = idx3 + owc_align256((size_t)n_seq * 4) / 4, *ql_l = os_l + owc_align256((size_t)n_seq * 4) / 4;
      OWC_TRY(owc_launch_last_rows_prep(last_index, qs_l, os_l, ql_l, n_out, Hq + 2 * Hkv, Hq, G, st));
      owc_attn_class_prefill(1);
      const int rc_tail = owc_launch_attention(qkv, hd, (long)G * hd, kc, hd, (long)cache->s_max * hd, vc, hd,
                                               (long)cache->s_max * hd, attn, hd, (long)G * hd, qs_l, os_l,
                                               k_start, k_len, ql_l, n_out, Hkv, 1, hd, G, 0, scale, st);
      owc_attn_class_prefill(0);
      OWC_TRY(rc_tail);
      OWC_TRY(owc_launch_gather_rows(x, d, last_index, xl, d, n_out, d, st));
      M = n_out;
      xr = xl;
    } else if (!decode) {
      OWC_TRY(owc_launch_attention(qkv, NQKV, hd, kc, hd, (long)cache->s_max * hd, vc, hd,
                                   (long)cache->s_max * hd, attn, (long)Hq * hd, hd, q_start, nullptr,
                                   k_start, k_len, q_len, n_seq, Hq, G, hd, max_q_len, 1, scale, st));
    } else if (fused_decode) {
      // rope of the fed token + KV-cache write + attention over the cache in one launch (attention.hip, round 3)
      OWC_TRY(owc_launch_attn_decode_fused(qkv, NQKV, pos3, w->rope_cos, w->rope_sin, kc, vc, tok_slot, tok_idx, k_len, attn,
                                           (long)Hq * hd, T, Hq, Hkv, cache->s_max, scale, st));
    } else {
      // one query row per q head: map the G heads of a kv group onto the "rows" of the kernel
      OWC_TRY(owc_launch_attention(qkv, hd, (long)G * hd, kc, hd, (long)cache->s_max * hd, vc, hd,
                                   (long)cache->s_max * hd, attn, hd, (long)G * hd, q_start, o_start,
                                   k_start, k_len, q_len, n_seq, Hkv, 1, hd, G, 0, scale, st));
    }
    OWC_TRY(linear(M, attn, (long)Hq * hd, L.o_w, L.o_s, nullptr, xr, xr, d, d, Hq * hd, OWC_EPI_RESIDUAL));
    // MLP block (HF:617-620, :464-466)
    rc_f = fp8 ? OWC_ERR_SHAPE
               : owc_launch_gemm_bf16_rmsnorm(xr, d, L.ln2_w, w->rms_eps, L.gateup_w, d, nullptr, mlp, F, M, 2 * F, d, OWC_EPI_SWIGLU, st);
    if (rc_f != OWC_OK && rc_f != OWC_ERR_SHAPE) return rc_f;
    if (rc_f == OWC_ERR_SHAPE) {
      OWC_TRY(norm(M, xr, L.ln2_w));
      OWC_TRY(linear(M, fp8 ? nullptr : h, d, L.gateup_w, L.gateup_s, nullptr, nullptr, mlp, F, 2 * F, d, OWC_EPI_SWIGLU));
    }
    OWC_TRY(linear(M, mlp, F, L.down_w, L.down_s, nullptr, xr, xr, d, d, F, OWC_EPI_RESIDUAL));
  }
  return OWC_OK;
}

int owc_llm_prefill(owc_ctx* ctx, const owc_llm_weights* w, const owc_kv_cache* cache,
                    const int32_t* ids, const int32_t* img_index, const void* img_embeds,
                    const int32_t* pos3, const int32_t* tok_slot, const int32_t* tok_idx,
                    const int32_t* seq_start, const int32_t* seq_len, const int32_t* q_len,
                    const int32_t* k_start, const int32_t* last_index, int n_seq, int n_out, int T,
                    int max_len, int bcast_first_slot, int bcast_n_slots, int score_mode, const owc_sampling* sampling,
                    int sampling_row0, int32_t* next_tok, void* logits_out, void* workspace, size_t ws_bytes, void* stream) {
  if (!ctx || !w || !cache || !ids || !pos3 || !tok_slot || !tok_idx || !seq_start || !seq_len ||
      !k_start || !last_index || !next_tok || !workspace)
    return OWC_ERR_ARG;
  if (n_out <= 0 || n_out > T) OWC_FAIL(ctx, OWC_ERR_ARG, "owc_llm_prefill: 0 < n_out <= T");
  if (score_mode != OWC_PREFILL_LAST_TOKENS && score_mode != OWC_PREFILL_SCORE_ROWS)
    OWC_FAIL(ctx, OWC_ERR_ARG, "owc_llm_prefill: score_mode must be OWC_PREFILL_LAST_TOKENS or OWC_PREFILL_SCORE_ROWS");
  if (score_mode == OWC_PREFILL_LAST_TOKENS && n_out > n_seq)
    OWC_FAIL(ctx, OWC_ERR_ARG, "owc_llm_prefill: more output rows than sequences needs OWC_PREFILL_SCORE_ROWS");
  // scoring: logits of ARBITRARY rows - the last layer runs on every row (a pruned last layer lets its rows attend like last tokens)
  const bool scoring = score_mode == OWC_PREFILL_SCORE_ROWS;
  const int n_rows = scoring ? n_out : n_seq;
  if (w->head_dim != 128) OWC_FAIL(ctx, OWC_ERR_SHAPE, "owc_llm_prefill: head_dim must be 128");
  if (w->weight_dtype == OWC_WEIGHTS_FP8 && ((w->d_model % 128) || (w->d_ff % 128)))
    OWC_FAIL(ctx, OWC_ERR_SHAPE, "owc_llm_prefill: fp8 weights need d_model and d_ff to be multiples of 128");
  if (ws_bytes < owc_llm_workspace_bytes(w, T, n_rows)) OWC_FAIL(ctx, OWC_ERR_WORKSPACE, "owc_llm_prefill: workspace too small");
  hipStream_t st = (hipStream_t)stream;
  const int d = w->d_model;
  Carver cv(workspace, ws_bytes);
  void* x = cv.take((size_t)T * d * 2);
  void* h = cv.take((size_t)T * d * 2);
  void* qkv = cv.take((size_t)T * (w->n_q_heads + 2 * w->n_kv_heads) * w->head_dim * 2);
  void* attn = cv.take((size_t)T * w->n_q_heads * w->head_dim * 2);
  void* mlp = cv.take((size_t)T * w->d_ff * 2);
  void* xl = cv.take((size_t)n_rows * d * 2);
  void* last = cv.take((size_t)n_rows * d * 2);
  int32_t* idx3 = (int32_t*)cv.take((size_t)n_rows * 4);
  cv.take((size_t)n_rows * 4);
  cv.take((size_t)n_rows * 4);
  void* logits = cv.take((size_t)n_rows * w->vocab * 2);
  if (logits_out) logits = logits_out;
  void* q8 = nullptr;
  float* qs = nullptr;
  if (w->weight_dtype == OWC_WEIGHTS_FP8) {
    size_t widest = (size_t)(d > w->d_ff ? d : w->d_ff);
    if ((size_t)w->n_q_heads * w->head_dim > widest) widest = (size_t)w->n_q_heads * w->head_dim;
    q8 = cv.take((size_t)T * widest);
    qs = (float*)cv.take((size_t)T * 4);
  }

  const bool rep = ctx->rep_seen != nullptr && !scoring;   // repetition penalty in force (owc_llm_set_repetition_penalty)
  if (rep) {
    if (ctx->rep_wpr * 32 < w->vocab) OWC_FAIL(ctx, OWC_ERR_SHAPE, "owc_llm_prefill: repetition-penalty bitmap narrower than the vocabulary");
    // every prompt row marks its id in its sequence's bitmap row (shared-prefix rows: in every sequence of the launch)
    OWC_TRY(owc_launch_seen_mark(ids, tok_slot, T, w->vocab, ctx->rep_seen, ctx->rep_wpr, bcast_first_slot, bcast_n_slots, st));
  }
  OWC_TRY(owc_launch_embed(ids, img_index, w->embed, img_embeds, x, T, d, st));
  OWC_TRY(llm_layers(ctx, w, cache, x, h, qkv, attn, mlp, q8, qs, pos3, T, tok_slot, tok_idx, seq_start, nullptr,
                     k_start, seq_len, q_len, n_seq, T, max_len, false, bcast_first_slot, bcast_n_slots, scoring ? nullptr : last_index,
                     n_out, xl, idx3, st));
  // final norm on the last token of every prompt only (the rows the last layer left in xl), then lm_head + greedy argmax
  if (!g_prune_last || scoring) OWC_TRY(owc_launch_gather_rows(x, d, last_index, xl, d, n_out, d, st));
  OWC_TRY(owc_launch_rmsnorm(xl, d, w->final_norm_w, last, d, n_out, d, w->rms_eps, nullptr, st));
  OWC_TRY(owc_launch_gemm_bf16(last, d, w->lm_head_w, d, nullptr, nullptr, 0, logits, w->vocab, n_out,
                               w->vocab, d, OWC_EPI_NONE, ctx->zeros, st));
  if (sampling && !scoring) {   // first new token by a draw at step 0; the stream ids are indexed by original batch row
    if (!sampling->stream_id && sampling_row0) OWC_FAIL(ctx, OWC_ERR_ARG, "owc_llm_prefill: sampling_row0 != 0 needs explicit stream ids");
    owc_sampling sp = *sampling;
    if (sp.stream_id) sp.stream_id += sampling_row0;
    if (sp.step_offset) sp.step_offset += sampling_row0;
    if (rep) OWC_TRY(owc_launch_penalize_rows(logits, w->vocab, n_out, w->vocab, ctx->rep_seen, ctx->rep_wpr, tok_slot, last_index, ctx->rep_penalty, st));
    OWC_TRY(owc_launch_sample(logits, w->vocab, n_out, w->vocab, &sp, nullptr, 0, nullptr, next_tok, st));
  } else if (rep) {
    OWC_TRY(owc_launch_argmax_penalized(logits, w->vocab, n_out, w->vocab, ctx->rep_seen, ctx->rep_wpr, tok_slot, last_index, ctx->rep_penalty,
                                        next_tok, st));
  } else {
    OWC_TRY(owc_launch_argmax(logits, w->vocab, n_out, w->vocab, next_tok, st));
  }
  return OWC_OK;
}

int owc_llm_decode_step(owc_ctx* ctx, const owc_llm_weights* w, const owc_kv_cache* cache,
                        int32_t* tok_io, int32_t* pos, const int32_t* slot,
                        int32_t* write_idx, const int32_t* k_start, int32_t* k_len,
                        const int32_t* q_start, const int32_t* o_start, const int32_t* q_len,
                        uint8_t* done, int32_t* out_tokens, int out_stride, int step, int32_t* step_state, int B,
                        int eos_id0, int eos_id1, int pad_id, const int32_t* out_row, const int32_t* forced_tok,
                        const owc_sampling* sampling, void* logits_out, void* workspace, size_t ws_bytes, void* stream) {
  if (!ctx || !w || !cache || !tok_io || !pos || !slot || !write_idx || !k_start || !k_len ||
      !q_start || !o_start || !q_len || !done || !out_tokens || !workspace)
    return OWC_ERR_ARG;
  if (ws_bytes < owc_llm_workspace_bytes(w, B, B)) OWC_FAIL(ctx, OWC_ERR_WORKSPACE, "owc_llm_decode_step: workspace too small");
  hipStream_t st = (hipStream_t)stream;
  const int d = w->d_model;
  Carver cv(workspace, ws_bytes);
  void* x = cv.take((size_t)B * d * 2);
  void* h = cv.take((size_t)B * d * 2);
  void* qkv = cv.take((size_t)B * (w->n_q_heads + 2 * w->n_kv_heads) * w->head_dim * 2);
  void* attn = cv.take((size_t)B * w->n_q_heads * w->head_dim * 2);
  void* mlp = cv.take((size_t)B * w->d_ff * 2);
  void* last = cv.take((size_t)B * d * 2);
  void* logits = cv.take((size_t)B * w->vocab * 2);
  if (logits_out) logits = logits_out;
  void* q8 = nullptr;
  float* qs = nullptr;
  if (w->weight_dtype == OWC_WEIGHTS_FP8) {
    size_t widest = (size_t)(d > w->d_ff ? d : w->d_ff);
    if ((size_t)w->n_q_heads * w->head_dim > widest) widest = (size_t)w->n_q_heads * w->head_dim;
    q8 = cv.take((size_t)B * widest);
    qs = (float*)cv.take((size_t)B * 4);
  }

  const bool rep = ctx->rep_seen != nullptr;
  if (rep) {   // the token fed now is part of HF's input_ids when this step's logits are processed
    if (ctx->rep_wpr * 32 < w->vocab) OWC_FAIL(ctx, OWC_ERR_SHAPE, "owc_llm_decode_step: repetition-penalty bitmap narrower than the vocabulary");
    OWC_TRY(owc_launch_seen_mark(tok_io, slot, B, w->vocab, ctx->rep_seen, ctx->rep_wpr, 0, 0, st));
  }
  OWC_TRY(owc_launch_embed(tok_io, nullptr, w->embed, nullptr, x, B, d, st));
  // the three mrope streams of a generated token are identical: pos_stride 0 re-reads `pos`
  OWC_TRY(llm_layers(ctx, w, cache, x, h, qkv, attn, mlp, q8, qs, pos, 0, slot, write_idx, q_start, o_start,
                     k_start, k_len, q_len, B, B, w->n_q_heads / w->n_kv_heads, true, 0, 0, nullptr, 0, nullptr, nullptr, st));
  OWC_TRY(owc_launch_rmsnorm(x, d, w->final_norm_w, last, d, B, d, w->rms_eps, nullptr, st));
  OWC_TRY(owc_launch_gemm_bf16(last, d, w->lm_head_w, d, nullptr, nullptr, 0, logits, w->vocab, B,
                               w->vocab, d, OWC_EPI_NONE, ctx->zeros, st));
  if (sampling) {
    if (rep) OWC_TRY(owc_launch_penalize_rows(logits, w->vocab, B, w->vocab, ctx->rep_seen, ctx->rep_wpr, slot, nullptr, ctx->rep_penalty, st));
    OWC_TRY(owc_launch_sample(logits, w->vocab, B, w->vocab, sampling, out_row, step, step_state, tok_io, st));
  } else if (rep) {
    OWC_TRY(owc_launch_argmax_penalized(logits, w->vocab, B, w->vocab, ctx->rep_seen, ctx->rep_wpr, slot, nullptr, ctx->rep_penalty, tok_io, st));
  } else {
    OWC_TRY(owc_launch_argmax(logits, w->vocab, B, w->vocab, tok_io, st));
  }
  OWC_TRY(owc_launch_decode_update(tok_io, done, out_tokens, out_stride, step, step_state, B, eos_id0, eos_id1,
                                   pad_id, out_row, forced_tok, st));
  if (step_state) OWC_TRY(owc_launch_decode_advance(pos, write_idx, k_len, step_state, B, st));
  return OWC_OK;
}

int owc_llm_set_repetition_penalty(owc_ctx* ctx, float penalty, uint32_t* seen, int words_per_row) {
  if (!ctx) return OWC_ERR_ARG;
  if (!seen) {
    ctx->rep_seen = nullptr;
    ctx->rep_penalty = 1.f;
    ctx->rep_wpr = 0;
    return OWC_OK;
  }
  if (!(penalty > 0.f) || words_per_row <= 0) OWC_FAIL(ctx, OWC_ERR_ARG, "owc_llm_set_repetition_penalty: need penalty > 0 and words_per_row > 0");
  ctx->rep_seen = seen;
  ctx->rep_penalty = penalty;
  ctx->rep_wpr = words_per_row;
  return OWC_OK;
}

int owc_argmax_penalized_bf16(owc_ctx* ctx, const void* logits, int64_t ld, int rows, int vocab, const uint32_t* seen, int words_per_row,
                              const int32_t* row_slot, float penalty, int32_t* out, void* stream) {
  if (!ctx || !logits || !seen || !out) return OWC_ERR_ARG;
  int rc = owc_launch_argmax_penalized(logits, ld, rows, vocab, seen, words_per_row, row_slot, nullptr, penalty, out, (hipStream_t)stream);
  if (rc != OWC_OK) ctx->err = "owc_argmax_penalized_bf16: bad shape or launch failure";
  return rc;
}

int owc_seen_mark(owc_ctx* ctx, const int32_t* ids, const int32_t* slot, int n, int vocab, uint32_t* seen, int words_per_row, void* stream) {
  if (!ctx || !ids || !seen) return OWC_ERR_ARG;
  int rc = owc_launch_seen_mark(ids, slot, n, vocab, seen, words_per_row, 0, 0, (hipStream_t)stream);
  if (rc != OWC_OK) ctx->err = "owc_seen_mark: bad shape or launch failure";
  return rc;
}

}  // extern "C"

void owc_llm_set_prune_last(int v) { g_prune_last = v != 0; }
void owc_llm_set_decode_fuse(int v) { g_decode_fuse = v != 0; }
