// Internal (non-ABI) declarations shared by the translation units of libowc_hip.so.
#pragma once
#include "owc_common.h"
#include "../../include/owc.h"
#include <string>

struct owc_ctx {
  int device = 0;
  void* zeros = nullptr;  // 256-byte device zero page (K-tail source of the LDS-DMA GEMMs)
  std::string err;
  // repetition penalty of the generation in progress (owc_llm_set_repetition_penalty; rep_seen == nullptr: off)
  float rep_penalty = 1.f;
  unsigned* rep_seen = nullptr;   // [slots][rep_wpr] bitmap of the token ids each sequence has seen, caller-owned
  int rep_wpr = 0;
};

#define OWC_CHECK_HIP(ctx, expr)                                                        \
  do {                                                                                  \
    hipError_t e__ = (expr);                                                            \
    if (e__ != hipSuccess) {                                                            \
      if (ctx) (ctx)->err = std::string(#expr) + ": " + hipGetErrorString(e__);         \
      return OWC_ERR_HIP;                                                               \
    }                                                                                   \
  } while (0)

#define OWC_FAIL(ctx, code, msg) \
  do {                           \
    if (ctx) (ctx)->err = (msg); \
    return (code);               \
  } while (0)

#define OWC_TRY(expr)            \
  do {                           \
    int rc__ = (expr);           \
    if (rc__ != OWC_OK) return rc__; \
  } while (0)

// ---- kernel launchers (one per .hip file) ----
int owc_launch_gemm_bf16(const void* A, long lda, const void* W, long ldw, const void* bias,
                         const void* R, long ldr, void* C, long ldc, int M, int N, int K, int epi,
                         const void* zeros, hipStream_t s);
int owc_launch_attention(const void* Q, long q_ts, long q_hs, const void* K, long k_ts, long k_hs,
                         const void* V, long v_ts, long v_hs, void* O, long o_ts, long o_hs,
                         const int* q_start, const int* o_start, const int* k_start,
                         const int* seq_len, const int* q_len, int n_seq, int n_heads, int kv_group, int head_dim,
                         int max_q_len, int causal, float scale, hipStream_t st);
int owc_launch_layernorm(const void* X, long ldx, const void* W, const void* B, void* Y, long ldy,
                         int rows, int d, float eps, hipStream_t st);
int owc_launch_rmsnorm(const void* X, long ldx, const void* W, void* Y, long ldy, int rows, int d,
                       float eps, const int* row_index, hipStream_t st);
int owc_launch_rope_table(float* cos_t, float* sin_t, int n_pos, int n_freq, int dim, float theta,
                          int round_bf16, hipStream_t st);
int owc_launch_vision_rope(void* qkv, long ld, const int* pos_hw, const float* cos_t,
                           const float* sin_t, int T, int n_heads, int hd, hipStream_t st);
int owc_launch_mrope_kv(void* qkv, long ld, const int* pos3, long pos_stride, const float* cos_t,
                        const float* sin_t, void* kc, void* vc, const int* tok_slot,
                        const int* tok_idx, int T, int n_q, int n_kv, int s_max, int sec0, int sec1,
                        int bcast_first, int bcast_n, hipStream_t st);
int owc_launch_embed(const int* ids, const int* img_index, const void* table, const void* img,
                     void* out, int T, int d, hipStream_t st);
int owc_launch_argmax(const void* logits, long ld, int rows, int V, int* out, hipStream_t st);
int owc_launch_seen_mark(const int* ids, const int* slot, int n, int V, unsigned* seen, int wpr, int bcast_first, int bcast_n, hipStream_t st);
int owc_launch_argmax_penalized(const void* logits, long ld, int rows, int V, const unsigned* seen, int wpr, const int* row_slot,
                                const int* row_index, float penalty, int* out, hipStream_t st);
int owc_launch_penalize_rows(void* logits, long ld, int rows, int V, const unsigned* seen, int wpr, const int* row_slot,
                             const int* row_index, float penalty, hipStream_t st);
int owc_launch_beam_candidates(const void* logits, long ld, int rows, int V, int K, float* logz, float* top_val, int* top_idx,
                               hipStream_t st);
int owc_launch_sample(const void* logits, long ld, int rows, int V, const owc_sampling* sp, const int* row_map, int step,
                      const int* step_state, int* out, hipStream_t st);
int owc_launch_token_logprob(const void* logits, long ld, const int* target, int rows, int V, float* out, hipStream_t st);
int owc_launch_decode_update(int* next_tok, uint8_t* done, int* out_tokens, int out_stride, int step,
                             const int* step_state, int B, int eos0, int eos1, int pad, const int* out_row, const int* forced,
                             hipStream_t st);
int owc_launch_decode_compact(const int* const* src, int* const* dst, const uint8_t* done_src, uint8_t* done_dst,
                              const int* live, int n, hipStream_t st);
int owc_launch_decode_advance(int* pos, int* widx, int* klen, int* step_state, int B, hipStream_t st);
int owc_launch_patchify(const uint8_t* img, void* out, long ldo, int n_img, int H, int W,
                        const float* mean, const float* stdv, hipStream_t st);
int owc_launch_gemm_f32(const float* A, long lda, const float* W, long ldw, const float* bias,
                        const float* R, long ldr, float* C, long ldc, int M, int N, int K, int epi,
                        const void* zeros, hipStream_t s);
int owc_launch_gemm_f32_bert(const float* A, long lda, const float* W, long ldw, const float* bias,
                        const float* R, long ldr, float* C, long ldc, int M, int N, int K, int epi,
                        const void* zeros, hipStream_t s);
void owc_bert_set_x3(int v);
void owc_gemm_profile_set(int on);
int owc_gemm_profile_collect(double* total_ms, double* total_flops, long* launches);  // arrays of 2: [bf16, fp8]
int owc_profile_collect(int n_kinds, double* total_ms, double* total_work, long* launches);  // OWC_PROF_* classes [0, n_kinds)
int owc_gemm_profile_begin(double flops, int kind, hipStream_t s);
int owc_profile_shapes_collect(int max_n, int* shape, double* stats);
void owc_gemm_profile_end(int handle, hipStream_t s);
void owc_gemm_set_big_min_m(int m);
void owc_gemm_set_dbg(int v);
void owc_gemm_set_mid_max_tiles(int v);
void owc_gemm_set_skinny_max_m(int v);
void owc_gemm_set_big_min_tiles(int v);
void owc_gemm_set_pingpong(int v);
void owc_gemm_set_norm_fuse_max_m(int v);
int owc_launch_gemm_bf16_rmsnorm(const void* X, long ldx, const void* gamma, float eps, const void* W, long ldw, const void* bias,
                                 void* C, long ldc, int M, int N, int K, int epi, hipStream_t s);
void owc_gemm_set_skinny_deep(int v);
void owc_gemm_set_small_tiles(int v);
void owc_gemm_set_ring_128(int v);
void owc_gemm_set_pp128(int v);
void owc_gemm_set_persist(int v);
void owc_gemm_set_walk(int v);
void owc_gemm_set_tail_split(int v);
void owc_gemm_set_nt_min_mb(int v);
int owc_gemm_nt_min_mb();
void owc_gemm_fp8_set_ring_128(int v);
void owc_gemm_fp8_set_shapes(int v);
void owc_gemm_set_k_pairs(int v);
void owc_gemm_set_norm_fuse_ring(int v);
void owc_gemm_set_k_pairs_min_k(int v);
void owc_gemm_set_tall_tiles(int v);
void owc_gemm_fp8_set_pingpong(int v);
void owc_gemm_fp8_set_skinny_max_m(int v);
void owc_gemm_fp8_set_mid_max_tiles(int v);
int owc_launch_gemm_bf16_aux(const void* A, long lda, const void* W, long ldw, const void* bias,
                             const void* R, long ldr, void* C, long ldc, int M, int N, int K, int epi,
                             const void* zeros, hipStream_t s, const owc_gemm_aux* aux);
void owc_attn_set_dbg(int v);
void owc_attn_class_prefill(int on);
void owc_attn_set_gqa_pack(int v);
void owc_attn_set_mfma32(int v);
void owc_attn_set_decode_nbuf1(int v);  // profile class of the next non-causal head_dim-128 launches
void owc_llm_set_prune_last(int v);
void owc_llm_set_decode_fuse(int v);
int owc_launch_attn_decode_fused(const void* qkv, long ld, const int* pos, const float* cos_t, const float* sin_t, void* kc,
                                 void* vc, const int* slot, const int* write_idx, const int* k_len, void* O, long ldo, int B,
                                 int n_q, int n_kv, int s_max, float scale, hipStream_t st);
int owc_launch_clip_patchify(const uint8_t* img, void* out, long ldo, int kpad, int n_img, int S,
                             const float* mean, const float* stdv, hipStream_t st);
int owc_launch_clip_embed(const void* pe, const void* pos_cls, void* x, int n_img, int tokens, int E,
                          hipStream_t st);
int owc_launch_seq_iota(int* start, int* len, int n, int L, hipStream_t st);
int owc_launch_gather_rows(const void* X, long ldx, const int* idx, void* Y, long ldy, int n, int width, hipStream_t st);
int owc_launch_last_rows_prep(const int* last_index, int* q_start, int* o_start, int* q_len, int n, int qkv_heads,
                              int n_q_heads, int group, hipStream_t st);
int owc_launch_rmsnorm_quant_fp8(const void* X, long ldx, const void* W, void* Q, long ldq, float* S, int rows, int d,
                                 float eps, hipStream_t st);
int owc_launch_quant_rows_fp8(const void* X, long ldx, void* Q, long ldq, float* S, int rows, int cols, hipStream_t st);
int owc_launch_gemm_fp8(const void* A, long lda, const float* sa, const void* W, long ldw, const float* sw,
                        const void* bias, const void* R, long ldr, void* C, long ldc, int M, int N, int K, int epi,
                        hipStream_t s);

// ---- host-side driver helpers ----
#include "../../include/owc.h"
inline size_t owc_align256(size_t x) { return (x + 255) & ~(size_t)255; }
struct Carver {  // bump allocator over a caller-provided workspace
  char* base;
  size_t off = 0, cap;
  Carver(void* p, size_t c) : base((char*)p), cap(c) {}
  void* take(size_t bytes) {
    void* r = base + off;
    off += owc_align256(bytes);
    return r;
  }
};
struct owc_vit_windows {   // Qwen2.5-VL window attention (NULL: Qwen2-VL / CLIP blocks)
  const int32_t *start, *len;
  int n, max_len;
  uint64_t fullatt_mask;
};
int owc_vit_layers(owc_ctx* ctx, const owc_vit_layer* layers, int n_layers, void* x, void* h, void* attn,
                   void* qkv, void* mlp, int T, int E, int H, int F, float eps, const int32_t* seq_start,
                   const int32_t* seq_len, int n_img, int max_len, const owc_gemm_aux* aux, hipStream_t st,
                   const owc_vit_windows* win = nullptr);
