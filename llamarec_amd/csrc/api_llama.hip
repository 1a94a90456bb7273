// api_llama.hip -- C ABI entry points for stage 2 (declared in include/llamarec_mi355x.h):
// one Llama prefill over packed prompts + verbalizer gather at each prompt's last token.
#include <stdlib.h>
#include <string.h>

#include <vector>

#include "llama_kernels.h"

typedef unsigned short u16;

extern "C" int lr_llama_create(const LrLlamaConfig* cfg, const LrLlamaWeightsDesc* w, lr_llama_t** out) {
  if (!cfg || !w || !out || !w->layers) LR_FAIL(LR_EINVAL, "lr_llama_create: null argument");
  if (cfg->num_layers < 1 || cfg->num_heads < 1 || cfg->num_kv_heads < 1 || cfg->vocab_size < 1)
    LR_FAIL(LR_EINVAL, "lr_llama_create: bad config");
  if (cfg->hidden_size % 8 != 0 || cfg->intermediate_size % 16 != 0)
    LR_FAIL(LR_EUNSUPPORTED, "lr_llama_create: hidden_size %% 8 and intermediate_size %% 16 must be 0");
  if (cfg->num_heads % cfg->num_kv_heads != 0 || cfg->head_dim < 4 || cfg->head_dim % 4 != 0 ||
      cfg->head_dim > 256)
    LR_FAIL(LR_EUNSUPPORTED, "lr_llama_create: heads=%d kv_heads=%d head_dim=%d", cfg->num_heads,
            cfg->num_kv_heads, cfg->head_dim);
  if (cfg->max_positions < 1) LR_FAIL(LR_EINVAL, "lr_llama_create: max_positions");
  if (!w->embed || !w->final_norm || !w->lm_head) LR_FAIL(LR_EINVAL, "lr_llama_create: null weight");
  for (int i = 0; i < cfg->num_layers; ++i) {
    const LrLlamaLayerWeights& l = w->layers[i];
    if (!l.input_norm || !l.wqkv || !l.wo || !l.post_norm || !l.wgu || !l.wdown)
      LR_FAIL(LR_EINVAL, "lr_llama_create: layer %d has a null weight", i);
  }
  lr_llama* h = (lr_llama*)calloc(1, sizeof(lr_llama));
  if (!h) LR_FAIL(LR_EINVAL, "lr_llama_create: out of host memory");
  h->cfg = *cfg;
  h->embed = w->embed;
  h->final_norm = w->final_norm;
  h->lm_head = w->lm_head;
  h->prune_last = 1;
  h->layers = (LrLlamaLayerWeights*)malloc(sizeof(LrLlamaLayerWeights) * cfg->num_layers);
  memcpy(h->layers, w->layers, sizeof(LrLlamaLayerWeights) * cfg->num_layers);
  LR_CHECK_HIP(hipGetDevice(&h->device));
  *out = h;
  return LR_OK;
}

extern "C" int lr_llama_set_variants(lr_llama_t* h, int32_t gemm_variant, int32_t attention_variant) {
  if (!h || (gemm_variant != 0 && gemm_variant != 1 && gemm_variant != 4 && gemm_variant != 5) || attention_variant < 0 ||
      attention_variant > 3)
    LR_FAIL(LR_EINVAL, "lr_llama_set_variants: gemm in {0, 1, 4, 5}, attention in {0, 1, 2, 3}");
  h->gemm_variant = gemm_variant;
  h->attn_variant = attention_variant;
  return LR_OK;
}

extern "C" int lr_llama_set_last_layer_pruning(lr_llama_t* h, int32_t enable) {
  if (!h) LR_FAIL(LR_EINVAL, "lr_llama_set_last_layer_pruning: null handle");
  h->prune_last = enable ? 1 : 0;
  return LR_OK;
}

extern "C" void lr_llama_destroy(lr_llama_t* h) {
  if (!h) return;
  free(h->layers);
  free(h->wqkv_folded);
  free(h->wgu_folded);
  free(h);
}

extern "C" int lr_fold_norm_bf16(const uint16_t* w, const uint16_t* norm_w, int32_t rows, int32_t cols, uint16_t* out,
                                 void* hip_stream) {
  if (!w || !norm_w || !out || rows < 1 || cols < 8) LR_FAIL(LR_EINVAL, "lr_fold_norm_bf16: bad argument");
  return lr_launch_fold_norm(w, norm_w, out, (size_t)rows, cols, (hipStream_t)hip_stream);
}

extern "C" int lr_llama_set_folded_norms(lr_llama_t* h, const uint16_t* const* wqkv_folded,
                                         const uint16_t* const* wgu_folded) {
  if (!h) LR_FAIL(LR_EINVAL, "lr_llama_set_folded_norms: null handle");
  free(h->wqkv_folded);
  free(h->wgu_folded);
  h->wqkv_folded = h->wgu_folded = nullptr;
  if (!wqkv_folded && !wgu_folded) return LR_OK;  // back to the separate RMSNorm pass
  if (!wqkv_folded || !wgu_folded) LR_FAIL(LR_EINVAL, "lr_llama_set_folded_norms: give both arrays or neither");
  const int L = h->cfg.num_layers;
  for (int l = 0; l < L; ++l)
    if (!wqkv_folded[l] || !wgu_folded[l]) LR_FAIL(LR_EINVAL, "lr_llama_set_folded_norms: layer %d has a null matrix", l);
  h->wqkv_folded = (const uint16_t**)malloc(sizeof(void*) * L);
  h->wgu_folded = (const uint16_t**)malloc(sizeof(void*) * L);
  memcpy(h->wqkv_folded, wqkv_folded, sizeof(void*) * L);
  memcpy(h->wgu_folded, wgu_folded, sizeof(void*) * L);
  return LR_OK;
}

struct LlamaWs {
  int32_t *tok_pos, *tok_src, *last_rows, *last_pos, *seg_start;
  float *rope, *rstd;
  u16 *x, *xn, *qkv, *att, *hmid;
  u16 *x_last, *xn_last, *att_last, *h_last, *q_last;  // compact [B][.] buffers of the pruned last layer
  float* splitk;                              // fp32 partial planes of the split-K GEMMs (gemm variant 5)
  unsigned* rope16;                           // the rope table as packed bf16 (cos | sin << 16) pairs
  int32_t* prefix_bad;                        // device word: a prompt broke the shared-prefix promise (token_meta_kernel)
  void* attn_items;                           // work-item list of the 256-row attention kernel (llama_attn256.hip)
  size_t attn_items_bytes;
  bool compact;                               // ws.x_last (not ws.x) holds the final residual rows
  size_t total;
};

static LlamaWs carve(const LrLlamaConfig& c, int max_tokens, int max_seqs, char* base) {
  LlamaWs w;
  w.compact = false;
  size_t o = 0;
  auto take = [&](size_t bytes) {
    size_t at = o;
    o += lr_align_up(bytes, 256);
    return base + at;
  };
  const size_t n = (size_t)max_tokens;
  const size_t qkv_w = (size_t)(c.num_heads + 2 * c.num_kv_heads) * c.head_dim;
  w.tok_pos = (int32_t*)take(n * 4);
  w.tok_src = (int32_t*)take(n * 4);
  w.rope = (float*)take((size_t)c.max_positions * (c.head_dim / 2) * 2 * sizeof(float));
  w.rope16 = (unsigned*)take((size_t)c.max_positions * (c.head_dim / 2) * sizeof(unsigned));
  w.rstd = (float*)take(n * sizeof(float));
  w.x = (u16*)take(n * c.hidden_size * 2);
  w.xn = (u16*)take(n * c.hidden_size * 2);
  w.qkv = (u16*)take(n * qkv_w * 2);
  w.att = (u16*)take(n * (size_t)c.num_heads * c.head_dim * 2);
  w.hmid = (u16*)take(n * c.intermediate_size * 2);
  const size_t nb = (size_t)(max_seqs > 0 ? max_seqs : 1);
  w.last_rows = (int32_t*)take(nb * 4);
  w.last_pos = (int32_t*)take(nb * 4);
  w.seg_start = (int32_t*)take((nb + 2) * 4);
  w.x_last = (u16*)take(nb * c.hidden_size * 2);
  w.xn_last = (u16*)take(nb * c.hidden_size * 2);
  w.att_last = (u16*)take(nb * (size_t)c.num_heads * c.head_dim * 2);
  w.h_last = (u16*)take(nb * c.intermediate_size * 2);
  w.q_last = (u16*)take(nb * (size_t)c.num_heads * c.head_dim * 2);
  w.splitk = (float*)take(LR_SPLITK_WS_BYTES);
  w.prefix_bad = (int32_t*)take(sizeof(int32_t));
  w.attn_items_bytes = lr_attn256_ws_bytes(max_tokens, (int)nb + 1, c.num_heads);
  w.attn_items = take(w.attn_items_bytes);
  w.total = o;
  return w;
}

extern "C" size_t lr_llama_workspace_bytes(const lr_llama_t* h, int32_t max_tokens, int32_t max_seqs) {
  if (!h || max_tokens < 1) return 0;
  if (max_seqs < 1) max_seqs = 1;
  if (max_seqs > max_tokens) max_seqs = max_tokens;
  return carve(h->cfg, max_tokens, max_seqs, nullptr).total;
}

// Runs the transformer body; leaves the residual stream (before the final norm) in ws.x (all internal rows) or,
// after a pruned last layer, in ws.x_last (one row per prompt). prefix_len = P > 0: the first P tokens of every
// prompt are the same (the caller's promise) and are run ONCE as segment 0 of the internal layout (llama_elem.hip,
// token_meta_kernel); the other segments attend to its K/V rows. Every row then sees exactly the operands of the
// unshared run, so the scores are bit-identical to prefix_len = 0.
static int run_body(lr_llama_t* h, const int32_t* ids, const int32_t* cu, const int32_t* cu_host, int B, int P,
                    void* workspace, size_t workspace_bytes, hipStream_t st, LlamaWs* out_ws) {
  if (!h || !ids || !cu || !cu_host || !workspace) LR_FAIL(LR_EINVAL, "llama prefill: null argument");
  if (B < 1) LR_FAIL(LR_EINVAL, "llama prefill: B=%d", B);
  const LrLlamaConfig& c = h->cfg;
  if (cu_host[0] != 0) LR_FAIL(LR_EINVAL, "llama prefill: cu_seqlens[0] must be 0");
  int maxT = 0, minT = 0x7fffffff;
  for (int b = 0; b < B; ++b) {
    int t = cu_host[b + 1] - cu_host[b];
    if (t < 1) LR_FAIL(LR_EINVAL, "llama prefill: prompt %d is empty", b);
    if (t > maxT) maxT = t;
    if (t < minT) minT = t;
  }
  if (maxT > c.max_positions)
    LR_FAIL(LR_EINVAL, "llama prefill: prompt of %d tokens exceeds max_positions %d", maxT, c.max_positions);
  if (P < 0 || (P > 0 && P >= minT))
    LR_FAIL(LR_EINVAL, "llama prefill: shared prefix of %d tokens, shortest prompt has %d (every prompt keeps >= 1 own token)",
            P, minT);
  if (P > 0 && (c.head_dim != 128 || h->attn_variant == 1)) P = 0;  // only the MFMA attention kernels read a shared prefix
  if (B == 1) P = 0;
  const int n_in = cu_host[B];
  const int n = P > 0 ? n_in - (B - 1) * P : n_in;  // internal rows
  const int S = P > 0 ? B + 1 : B;                   // segments
  LlamaWs ws = carve(c, n_in, B, (char*)workspace);
  if (ws.total > workspace_bytes)
    LR_FAIL(LR_EWORKSPACE, "llama prefill: workspace needs %zu bytes for %d tokens, have %zu", ws.total, n_in,
            workspace_bytes);
  // host copy of the segment starts (launch geometry of the attention kernel)
  std::vector<int32_t> seg_host_v((size_t)S + 1);
  int32_t* seg_host = seg_host_v.data();
  if (P > 0) {
    seg_host[0] = 0;
    for (int b = 0; b <= B; ++b) seg_host[b + 1] = P + cu_host[b] - b * P;
  } else {
    for (int b = 0; b <= B; ++b) seg_host[b] = cu_host[b];
  }
  const int d = c.hidden_size, f = c.intermediate_size, nh = c.num_heads, nkv = c.num_kv_heads,
            hd = c.head_dim;
  const int qkv_w = (nh + 2 * nkv) * hd;
  int rc;
#define RUN(x)              \
  do {                      \
    rc = (x);               \
    if (rc) return rc;      \
  } while (0)
  RUN(lr_launch_token_meta(cu, B, P, ws.seg_start, ws.tok_pos, ws.tok_src, ws.last_rows, st, ws.last_pos, ids,
                           ws.prefix_bad));
  RUN(lr_launch_rope_table(ws.rope, maxT, hd, c.rope_theta, st, ws.rope16));
  // attention: 3 = 256-row tiles, one wave per SIMD (llama_attn256.hip), its item list built once per call. Auto keeps variant 2:
  // on the prompts of this path (460 .. 1 125 tokens) the 128-row kernel with two workgroups per CU is still ahead
  const bool attn256 = h->attn_variant == 3 && lr_attention256_takes(hd, P);
  if (h->attn_variant == 3 && !attn256 && hd != 128)
    LR_FAIL(LR_EUNSUPPORTED, "llama prefill: attention variant 3 needs head_dim 128 (got %d)", hd);
  const int attn_var = attn256 ? 3 : (h->attn_variant == 3 ? 2 : h->attn_variant);
  if (attn256) RUN(lr_launch_attn256_items(ws.seg_start, S, n, nh, P, ws.attn_items, ws.attn_items_bytes, st));
  RUN(lr_launch_embed(ids, ws.tok_src, h->embed, c.vocab_size, d, ws.x, n, st));
  bool input_normed = false;   // ws.xn already holds RMSNorm(ws.x) with this layer's input_norm
#ifdef LR_EXPERIMENTS   // timing-only arm (never in the product library): the row statistics of layer 0 serve every layer
  static const bool exp_rstd_once = getenv("LR_EXP_RSTD_ONCE") && getenv("LR_EXP_RSTD_ONCE")[0] == '1';
#define EXP_SKIP_SWEEP (exp_rstd_once && l > 0)
#else
#define EXP_SKIP_SWEEP false
#endif
  for (int l = 0; l < c.num_layers; ++l) {
    const LrLlamaLayerWeights& w = h->layers[l];
    // RMSNorm: either its own pass (read + write every row), or -- folded -- only the row statistic, with the norm
    // weight already inside the projection matrix and rstd applied to the accumulator rows in the GEMM epilogue
    const bool folded = h->wqkv_folded != nullptr;
    const bool last_pruned = l == c.num_layers - 1 && h->prune_last;
    // Pruned last layer, head_dim 128: only each prompt's last token is consumed after it (model/llm.py:131), so Q is
    // needed for B rows only -- K and V for all of them. The projection runs on the K | V rows of wqkv (2/3 of the
    // product) for every row and on its Q rows for the B last rows; the attention kernel then evaluates ONE query row
    // per (prompt, head) over the prompt's keys instead of every tile.
    const bool last_q_only = last_pruned && !folded && hd == 128 && h->attn_variant != 1 && h->gemm_variant != 1;
    if (last_q_only) {
      const int q_w = nh * hd, kv_w = 2 * nkv * hd;
      const int pv = (h->gemm_variant == 5 || (h->gemm_variant == 0 && B <= 256)) ? 5 : 1;
      if (!input_normed) RUN(lr_launch_rmsnorm(ws.x, w.input_norm, ws.xn, n, d, c.rms_eps, nullptr, st));
      RUN(lr_launch_gemm(ws.xn, w.wqkv + (size_t)q_w * d, ws.qkv, nullptr, n, kv_w, d, LR_EPI_ROPE, h->gemm_variant, st,
                         ws.tok_pos, ws.rope, hd, nkv * hd, ws.splitk, LR_SPLITK_WS_BYTES, nullptr, ws.rope16));
      RUN(lr_launch_gather_rows(ws.xn, ws.last_rows, B, d, ws.xn_last, st));
      RUN(lr_launch_gemm(ws.xn_last, w.wqkv, ws.q_last, nullptr, B, q_w, d, LR_EPI_ROPE, pv, st, ws.last_pos, ws.rope, hd,
                         q_w, ws.splitk, LR_SPLITK_WS_BYTES, nullptr, ws.rope16));
      RUN(lr_launch_attention_last(ws.qkv, ws.q_last, ws.att_last, ws.seg_start, seg_host, S, n, nh, nkv, hd, st, P));
    } else if (folded) {
      if (!EXP_SKIP_SWEEP) RUN(lr_launch_rms_rstd(ws.x, ws.rstd, n, d, c.rms_eps, st));
      RUN(lr_launch_gemm(ws.x, h->wqkv_folded[l], ws.qkv, nullptr, n, qkv_w, d, LR_EPI_ROPE, h->gemm_variant, st, ws.tok_pos,
                         ws.rope, hd, (nh + nkv) * hd, ws.splitk, LR_SPLITK_WS_BYTES, ws.rstd, ws.rope16));
    } else {
      if (!input_normed) RUN(lr_launch_rmsnorm(ws.x, w.input_norm, ws.xn, n, d, c.rms_eps, nullptr, st));
      RUN(lr_launch_gemm(ws.xn, w.wqkv, ws.qkv, nullptr, n, qkv_w, d, LR_EPI_ROPE, h->gemm_variant, st, ws.tok_pos,
                         ws.rope, hd, (nh + nkv) * hd, ws.splitk, LR_SPLITK_WS_BYTES, nullptr, ws.rope16));
    }
    if (last_pruned) {
      // Only each prompt's LAST token is consumed after the final layer (model/llm.py:131), so the
      // last layer needs K/V for every token but attention output, o_proj, and the MLP for B rows only.
      if (last_q_only) {
        // ws.att_last already holds the attention rows of the last tokens
      } else if (hd == 128 && h->attn_variant != 1) {
        // the MFMA kernel over ALL rows (188 us for 14.8 k tokens, 16 us for one prompt) beats the scalar kernel over the
        // B last rows (459 / 295 us): attend everything, keep the last rows
        if (attn256)
          RUN(lr_launch_attention256(ws.qkv, ws.att, ws.seg_start, seg_host, S, n, nh, nkv, hd, nullptr, ws.attn_items, st, P));
        else
          RUN(lr_launch_attention(ws.qkv, ws.att, ws.seg_start, seg_host, ws.tok_pos, nullptr, S, n, nh, nkv, hd,
                                  attn_var, nullptr, st, P));
        RUN(lr_launch_gather_rows(ws.att, ws.last_rows, B, nh * hd, ws.att_last, st));
      } else {
        RUN(lr_launch_attention_rows(ws.qkv, ws.att_last, cu, B, ws.last_rows, B, nh, nkv, hd, st));
      }
      RUN(lr_launch_gather_rows(ws.x, ws.last_rows, B, d, ws.x_last, st));
      // B-row products: split-K over the 256-column tiles (weight streaming spread over 64-128 CUs instead of N / 64
      // workgroups of the small-tile kernel: 23 rows x 4096 x 4096 took 134 us there). With B <= 256 rows there is one
      // row tile, so the split count depends on the weight's shape only and a prompt's arithmetic stays the same whatever
      // else is in the batch; more prompts than that take the small-tile kernel as before. (Shapes the 256-tile kernel
      // does not take fall back to it inside lr_launch_gemm.)
      const int pv = (h->gemm_variant == 5 || (h->gemm_variant == 0 && B <= 256)) ? 5 : 1;
      bool normed = false;   // a split-K product's reduce pass also writes the RMSNorm that follows (same bits)
      RUN(lr_launch_gemm(ws.att_last, w.wo, ws.x_last, ws.x_last, B, d, nh * hd, LR_EPI_RESIDUAL, pv, st, nullptr, nullptr,
                         0, 0, ws.splitk, LR_SPLITK_WS_BYTES, nullptr, nullptr, w.post_norm, ws.xn_last, c.rms_eps, &normed));
      if (!normed) RUN(lr_launch_rmsnorm(ws.x_last, w.post_norm, ws.xn_last, B, d, c.rms_eps, nullptr, st));
      RUN(lr_launch_gemm(ws.xn_last, w.wgu, ws.h_last, nullptr, B, 2 * f, d, LR_EPI_SWIGLU, pv, st, nullptr, nullptr, 0, 0,
                         ws.splitk, LR_SPLITK_WS_BYTES));
      RUN(lr_launch_gemm(ws.h_last, w.wdown, ws.x_last, ws.x_last, B, d, f, LR_EPI_RESIDUAL, pv, st, nullptr, nullptr, 0, 0,
                         ws.splitk, LR_SPLITK_WS_BYTES));
      ws.compact = true;
      break;
    }
    if (attn256)
      RUN(lr_launch_attention256(ws.qkv, ws.att, ws.seg_start, seg_host, S, n, nh, nkv, hd, nullptr, ws.attn_items, st, P));
    else
      RUN(lr_launch_attention(ws.qkv, ws.att, ws.seg_start, seg_host, ws.tok_pos, nullptr, S, n, nh, nkv, hd, attn_var,
                              nullptr, st, P));
    bool post_normed = false;
    RUN(lr_launch_gemm(ws.att, w.wo, ws.x, ws.x, n, d, nh * hd, LR_EPI_RESIDUAL, h->gemm_variant, st, nullptr, nullptr, 0,
                       0, ws.splitk, LR_SPLITK_WS_BYTES, nullptr, nullptr, folded ? nullptr : w.post_norm, ws.xn, c.rms_eps,
                       &post_normed));
    if (folded) {
      if (!EXP_SKIP_SWEEP) RUN(lr_launch_rms_rstd(ws.x, ws.rstd, n, d, c.rms_eps, st));
      RUN(lr_launch_gemm(ws.x, h->wgu_folded[l], ws.hmid, nullptr, n, 2 * f, d, LR_EPI_SWIGLU, h->gemm_variant, st, nullptr,
                         nullptr, 0, 0, ws.splitk, LR_SPLITK_WS_BYTES, ws.rstd));
    } else {
      if (!post_normed) RUN(lr_launch_rmsnorm(ws.x, w.post_norm, ws.xn, n, d, c.rms_eps, nullptr, st));
      RUN(lr_launch_gemm(ws.xn, w.wgu, ws.hmid, nullptr, n, 2 * f, d, LR_EPI_SWIGLU, h->gemm_variant, st, nullptr, nullptr,
                         0, 0, ws.splitk, LR_SPLITK_WS_BYTES));
    }
    // down_proj; its reduce pass (latency mode) also writes the NEXT layer's input RMSNorm when that layer reads ws.xn
    const bool next_reads_xn = l + 1 < c.num_layers && !folded;
    RUN(lr_launch_gemm(ws.hmid, w.wdown, ws.x, ws.x, n, d, f, LR_EPI_RESIDUAL, h->gemm_variant, st, nullptr, nullptr, 0, 0,
                       ws.splitk, LR_SPLITK_WS_BYTES, nullptr, nullptr, next_reads_xn ? h->layers[l + 1].input_norm : nullptr,
                       ws.xn, c.rms_eps, &input_normed));
  }
#undef RUN
#undef EXP_SKIP_SWEEP
  *out_ws = ws;
  return LR_OK;
}

static int prefill_head(lr_llama_t* h, const int32_t* packed_ids, const int32_t* cu_seqlens, const int32_t* cu_seqlens_host,
                        int32_t B, int32_t prefix_len, const int32_t* class_ids, int32_t C, float* out, void* workspace,
                        size_t workspace_bytes, hipStream_t st) {
  LlamaWs ws;
  int rc = run_body(h, packed_ids, cu_seqlens, cu_seqlens_host, B, prefix_len, workspace, workspace_bytes, st, &ws);
  if (rc) return rc;
  return lr_launch_head(ws.compact ? ws.x_last : ws.x, ws.compact ? nullptr : ws.last_rows, h->final_norm, h->lm_head,
                        class_ids, B, C, h->cfg.hidden_size, h->cfg.rms_eps, out, h->cfg.vocab_size, st, ws.prefix_bad);
}

extern "C" int lr_llama_prefill_verbalize(lr_llama_t* h, const int32_t* packed_ids, const int32_t* cu_seqlens,
                                          const int32_t* cu_seqlens_host, int32_t B,
                                          const int32_t* label_token_ids, int32_t C, float* out_scores,
                                          void* workspace, size_t workspace_bytes, void* hip_stream) {
  if (!label_token_ids || !out_scores || C < 1) LR_FAIL(LR_EINVAL, "lr_llama_prefill_verbalize: bad label ids / output");
  return prefill_head(h, packed_ids, cu_seqlens, cu_seqlens_host, B, 0, label_token_ids, C, out_scores, workspace,
                      workspace_bytes, (hipStream_t)hip_stream);
}

extern "C" int lr_llama_prefill_verbalize_prefix(lr_llama_t* h, const int32_t* packed_ids, const int32_t* cu_seqlens,
                                                 const int32_t* cu_seqlens_host, int32_t B, int32_t prefix_len,
                                                 const int32_t* label_token_ids, int32_t C, float* out_scores,
                                                 void* workspace, size_t workspace_bytes, void* hip_stream) {
  if (!label_token_ids || !out_scores || C < 1)
    LR_FAIL(LR_EINVAL, "lr_llama_prefill_verbalize_prefix: bad label ids / output");
  return prefill_head(h, packed_ids, cu_seqlens, cu_seqlens_host, B, prefix_len, label_token_ids, C, out_scores, workspace,
                      workspace_bytes, (hipStream_t)hip_stream);
}

extern "C" int lr_llama_last_logits(lr_llama_t* h, const int32_t* packed_ids, const int32_t* cu_seqlens,
                                    const int32_t* cu_seqlens_host, int32_t B, float* out_logits,
                                    void* workspace, size_t workspace_bytes, void* hip_stream) {
  if (!out_logits) LR_FAIL(LR_EINVAL, "lr_llama_last_logits: null output");
  return prefill_head(h, packed_ids, cu_seqlens, cu_seqlens_host, B, 0, nullptr, h ? h->cfg.vocab_size : 0, out_logits,
                      workspace, workspace_bytes, (hipStream_t)hip_stream);
}

extern "C" int32_t lr_common_prefix_len(const int32_t* packed_ids_host, const int32_t* cu_seqlens_host, int32_t B) {
  if (!packed_ids_host || !cu_seqlens_host || B < 2) return 0;
  int n = 0x7fffffff;
  for (int b = 0; b < B; ++b) {
    const int t = cu_seqlens_host[b + 1] - cu_seqlens_host[b];
    if (t - 1 < n) n = t - 1;
  }
  if (n <= 0) return 0;
  const int32_t* head = packed_ids_host + cu_seqlens_host[0];
  for (int b = 1; b < B && n > 0; ++b) {
    const int32_t* p = packed_ids_host + cu_seqlens_host[b];
    int i = 0;
    while (i < n && p[i] == head[i]) ++i;
    n = i;
  }
  return n;
}

extern "C" int lr_llama_pack_gate_up(const uint16_t* gate, const uint16_t* up, int32_t inter, int32_t hidden,
                                     uint16_t* out) {
  if (!gate || !up || !out || inter < 16 || inter % 16 != 0 || hidden < 1)
    LR_FAIL(LR_EINVAL, "lr_llama_pack_gate_up: inter=%d (must be a multiple of 16) hidden=%d", inter, hidden);
  for (int t = 0; t < inter / 16; ++t) {
    memcpy(out + (size_t)(32 * t) * hidden, gate + (size_t)(16 * t) * hidden, (size_t)16 * hidden * 2);
    memcpy(out + (size_t)(32 * t + 16) * hidden, up + (size_t)(16 * t) * hidden, (size_t)16 * hidden * 2);
  }
  return LR_OK;
}

extern "C" int lr_llama_pack_qkv(const uint16_t* q, const uint16_t* k, const uint16_t* v, int32_t num_heads,
                                 int32_t num_kv_heads, int32_t head_dim, int32_t hidden, uint16_t* out) {
  if (!q || !k || !v || !out || num_heads < 1 || num_kv_heads < 1 || head_dim < 2 || head_dim % 2 || hidden < 1)
    LR_FAIL(LR_EINVAL, "lr_llama_pack_qkv: bad arguments");
  const int half = head_dim / 2;
  const size_t rb = (size_t)hidden * 2;
  size_t o = 0;
  for (int part = 0; part < 2; ++part) {
    const uint16_t* src = part == 0 ? q : k;
    const int heads = part == 0 ? num_heads : num_kv_heads;
    for (int hh = 0; hh < heads; ++hh)
      for (int i = 0; i < half; ++i) {
        memcpy(out + (o++) * hidden, src + ((size_t)hh * head_dim + i) * hidden, rb);
        memcpy(out + (o++) * hidden, src + ((size_t)hh * head_dim + half + i) * hidden, rb);
      }
  }
  memcpy(out + o * hidden, v, (size_t)num_kv_heads * head_dim * rb);
  return LR_OK;
}

extern "C" int lr_gemm_bf16_nt(const uint16_t* A, const uint16_t* B, uint16_t* C, int32_t M, int32_t N,
                               int32_t K, int32_t variant, void* hip_stream) {
  if (!A || !B || !C) LR_FAIL(LR_EINVAL, "lr_gemm_bf16_nt: null pointer");
  return lr_launch_gemm(A, B, C, nullptr, M, N, K, LR_EPI_STORE, variant, (hipStream_t)hip_stream);
}

extern "C" int lr_gemm_bf16_nt_ws(const uint16_t* A, const uint16_t* B, uint16_t* C, int32_t M, int32_t N,
                                  int32_t K, int32_t variant, void* workspace, size_t workspace_bytes,
                                  void* hip_stream) {
  if (!A || !B || !C) LR_FAIL(LR_EINVAL, "lr_gemm_bf16_nt_ws: null pointer");
  return lr_launch_gemm(A, B, C, nullptr, M, N, K, LR_EPI_STORE, variant, (hipStream_t)hip_stream, nullptr, nullptr, 0,
                        0, (float*)workspace, workspace_bytes);
}

extern "C" int lr_gemm_bf16_nt_epi(const uint16_t* A, const uint16_t* B, uint16_t* C, const uint16_t* R, int32_t M,
                                   int32_t N, int32_t K, int32_t epilogue, int32_t variant, const int32_t* tok_pos,
                                   const float* rope_cs, int32_t rope_positions, int32_t head_dim, int32_t rot_cols,
                                   void* workspace, size_t workspace_bytes, void* hip_stream) {
  if (!A || !B || !C) LR_FAIL(LR_EINVAL, "lr_gemm_bf16_nt_epi: null pointer");
  if (epilogue < LR_EPI_STORE || epilogue > LR_EPI_ROPE) LR_FAIL(LR_EINVAL, "lr_gemm_bf16_nt_epi: epilogue %d", epilogue);
  // the packed half of lr_rope_table's buffer sits behind the fp32 half
  const unsigned* cs16 = (rope_cs && rope_positions > 0 && head_dim >= 2)
                             ? reinterpret_cast<const unsigned*>(rope_cs + (size_t)rope_positions * head_dim) : nullptr;
  return lr_launch_gemm(A, B, C, R, M, N, K, epilogue, variant, (hipStream_t)hip_stream, tok_pos, rope_cs, head_dim,
                        rot_cols, (float*)workspace, workspace_bytes, nullptr, cs16);
}

extern "C" int lr_gemm_bf16_nt_residual_rmsnorm(const uint16_t* A, const uint16_t* B, uint16_t* C, const uint16_t* R,
                                                int32_t M, int32_t N, int32_t K, int32_t variant, const uint16_t* norm_w,
                                                uint16_t* norm_out, float eps, int32_t fuse, int32_t* was_fused,
                                                void* workspace, size_t workspace_bytes, void* hip_stream) {
  if (!A || !B || !C || !R || !norm_w || !norm_out) LR_FAIL(LR_EINVAL, "lr_gemm_bf16_nt_residual_rmsnorm: null pointer");
  hipStream_t st = (hipStream_t)hip_stream;
  bool done = false;
  int rc = lr_launch_gemm(A, B, C, R, M, N, K, LR_EPI_RESIDUAL, variant, st, nullptr, nullptr, 0, 0, (float*)workspace,
                          workspace_bytes, nullptr, nullptr, fuse ? norm_w : nullptr, norm_out, eps, &done);
  if (rc) return rc;
  if (was_fused) *was_fused = done ? 1 : 0;
  if (done) return LR_OK;
  return lr_launch_rmsnorm(C, norm_w, norm_out, M, N, eps, nullptr, st);
}

extern "C" size_t lr_rope_table_bytes(int32_t max_positions, int32_t head_dim) {
  if (max_positions < 1 || head_dim < 2) return 0;
  return (size_t)max_positions * (head_dim / 2) * (2 * sizeof(float) + sizeof(unsigned));
}

extern "C" int lr_rope_table(float* cs, int32_t max_positions, int32_t head_dim, float theta, void* hip_stream) {
  if (!cs || max_positions < 1 || head_dim < 2) LR_FAIL(LR_EINVAL, "lr_rope_table: bad argument");
  return lr_launch_rope_table(cs, max_positions, head_dim, theta, (hipStream_t)hip_stream,
                              reinterpret_cast<unsigned*>(cs + (size_t)max_positions * head_dim));
}

extern "C" int lr_attention_varlen(const uint16_t* qkv, uint16_t* out, const int32_t* cu_seqlens,
                                   const int32_t* cu_seqlens_host, int32_t B, int32_t num_heads,
                                   int32_t num_kv_heads, int32_t head_dim, int32_t variant, void* hip_stream) {
  if (!qkv || !out || !cu_seqlens || !cu_seqlens_host || B < 1) LR_FAIL(LR_EINVAL, "lr_attention_varlen: bad argument");
  return lr_launch_attention(qkv, out, cu_seqlens, cu_seqlens_host, nullptr, nullptr, B, cu_seqlens_host[B],
                             num_heads, num_kv_heads, head_dim, variant, nullptr, (hipStream_t)hip_stream);
}

extern "C" size_t lr_attention_workspace_bytes(int32_t total_tokens, int32_t B, int32_t num_heads) {
  if (total_tokens < 1 || B < 1 || num_heads < 1) return 0;
  return lr_attn256_ws_bytes(total_tokens, B, num_heads);
}

extern "C" int lr_attention_varlen_ws(const uint16_t* qkv, uint16_t* out, float* lse, const int32_t* cu_seqlens,
                                      const int32_t* cu_seqlens_host, int32_t B, int32_t num_heads, int32_t num_kv_heads,
                                      int32_t head_dim, int32_t variant, void* workspace, size_t workspace_bytes,
                                      void* hip_stream) {
  if (!qkv || !out || !cu_seqlens || !cu_seqlens_host || B < 1) LR_FAIL(LR_EINVAL, "lr_attention_varlen_ws: bad argument");
  hipStream_t st = (hipStream_t)hip_stream;
  const int n = cu_seqlens_host[B];
  if (variant == 3 || (variant == 0 && head_dim == 128 && workspace)) {
    if (head_dim != 128) LR_FAIL(LR_EUNSUPPORTED, "lr_attention_varlen_ws: variant 3 needs head_dim 128 (got %d)", head_dim);
    if (int rc = lr_launch_attn256_items(cu_seqlens, B, n, num_heads, 0, workspace, workspace_bytes, st)) return rc;
    return lr_launch_attention256(qkv, out, cu_seqlens, cu_seqlens_host, B, n, num_heads, num_kv_heads, head_dim, lse,
                                  workspace, st, 0);
  }
  return lr_launch_attention(qkv, out, cu_seqlens, cu_seqlens_host, nullptr, nullptr, B, n, num_heads, num_kv_heads, head_dim,
                             variant, lse, st);
}
