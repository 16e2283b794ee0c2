// Host-side driver of the LLaVA image branch: CLIP ViT tower up to `vision_feature_layer` followed by the
// 2-layer GELU projector.  Enqueues this library's kernels in the order HF transformers runs
//   LlavaModel.get_image_features (modeling_llava.py:144-189) -> CLIPVisionTransformer.forward
//   (modeling_clip.py: embeddings, pre_layrnorm, encoder layers) -> LlavaMultiModalProjector (:87-106)
// which is what the reference reaches from src/models/_llava_hf.py:365-376.  No allocation, no sync.
#include "owc_internal.h"

extern "C" {

size_t owc_clip_workspace_bytes(const owc_clip_weights* w, int n_img) {
  if (!w || n_img <= 0) return 0;
  const size_t t = (size_t)n_img * w->tokens, e = (size_t)w->embed_dim;
  const size_t wide = (size_t)(w->mlp_hidden > w->out_dim ? w->mlp_hidden : w->out_dim);
  size_t b = 0;
  b += owc_align256(t * e * 2) * 3;      // x, h, attn (attn doubles as the patch-embed output)
  b += owc_align256(t * e * 3 * 2);      // qkv
  b += owc_align256(t * wide * 2);       // mlp hidden / projector hidden
  b += owc_align256((size_t)n_img * 4) * 2;  // seq_start, seq_len
  return b + 1024;
}

int owc_clip_forward(owc_ctx* ctx, const owc_clip_weights* w, const void* patches, int64_t ld_patches,
                     int n_img, void* out, void* workspace, size_t ws_bytes, void* stream) {
  if (!ctx || !w || !patches || !out || !workspace) return OWC_ERR_ARG;
  if (n_img <= 0 || w->tokens < 2 || (w->patch_k & 7) || ld_patches < w->patch_k)
    OWC_FAIL(ctx, OWC_ERR_SHAPE, "owc_clip_forward: bad shape (patch_k % 8, ld_patches >= patch_k)");
  if (ws_bytes < owc_clip_workspace_bytes(w, n_img)) OWC_FAIL(ctx, OWC_ERR_WORKSPACE, "owc_clip_forward: workspace too small");
  hipStream_t st = (hipStream_t)stream;
  const int E = w->embed_dim, H = w->num_heads, F = w->mlp_hidden, D = w->out_dim;
  const int T = n_img * w->tokens, TP = n_img * (w->tokens - 1);
  const size_t wide = (size_t)(F > D ? F : D);
  Carver cv(workspace, ws_bytes);
  void* x = cv.take((size_t)T * E * 2);
  void* h = cv.take((size_t)T * E * 2);
  void* attn = cv.take((size_t)T * E * 2);
  void* qkv = cv.take((size_t)T * E * 3 * 2);
  void* mlp = cv.take((size_t)T * wide * 2);
  int32_t* seq_start = (int32_t*)cv.take((size_t)n_img * 4);
  int32_t* seq_len = (int32_t*)cv.take((size_t)n_img * 4);

  OWC_TRY(owc_launch_seq_iota(seq_start, seq_len, n_img, w->tokens, st));
  // patch_embedding: Conv2d(kernel == stride, bias=False) == GEMM over flattened patches (K zero-padded)
  OWC_TRY(owc_launch_gemm_bf16(patches, ld_patches, w->patch_w, w->patch_k, nullptr, nullptr, 0, attn, E, TP,
                               E, w->patch_k, OWC_EPI_NONE, ctx->zeros, st));
  // cat(class token, patches) + position embedding, then pre_layrnorm
  OWC_TRY(owc_launch_clip_embed(attn, w->pos_cls, h, n_img, w->tokens, E, st));
  OWC_TRY(owc_launch_layernorm(h, E, w->pre_ln_w, w->pre_ln_b, x, E, T, E, w->ln_eps, st));
  // hidden_states[vision_feature_layer]: the first n_layers encoder layers (no post_layernorm)
  OWC_TRY(owc_vit_layers(ctx, w->layers, w->n_layers, x, h, attn, qkv, mlp, T, E, H, F, w->ln_eps, seq_start,
                         seq_len, n_img, w->tokens, nullptr, st));
  // projector on every row (row-wise op; the caller's gather skips the CLS rows = "default" select strategy)
  OWC_TRY(owc_launch_gemm_bf16(x, E, w->proj1_w, E, w->proj1_b, nullptr, 0, mlp, D, T, D, E,
                               OWC_EPI_GELU_ERF, ctx->zeros, st));
  OWC_TRY(owc_launch_gemm_bf16(mlp, D, w->proj2_w, D, w->proj2_b, nullptr, 0, out, D, T, D, D,
                               OWC_EPI_NONE, ctx->zeros, st));
  return OWC_OK;
}

int owc_clip_patchify_u8(owc_ctx* ctx, const uint8_t* images, void* patches, int64_t ld, int patch_k, int n,
                         int S, const float* mean_host, const float* std_host, void* stream) {
  if (!ctx || !images || !patches || !mean_host || !std_host) return OWC_ERR_ARG;
  int rc = owc_launch_clip_patchify(images, patches, ld, patch_k, n, S, mean_host, std_host, (hipStream_t)stream);
  if (rc != OWC_OK) OWC_FAIL(ctx, rc, "owc_clip_patchify_u8: bad shape (S % 14, 588 <= patch_k <= ld)");
  return OWC_OK;
}

}  // extern "C"
