// api_llama_train.hip -- C ABI of the ranker's LoRA training step (declared in include/llamarec_mi355x.h,
// SURVEY.md 8(f) #4): forward with the adapters live (not merged), shifted cross-entropy on the labelled rows,
// backward through the frozen bf16 base to the LoRA matrices, clipping + AdamW.
//
// Replaces trainer/llm.py:103-136 (HF Trainer.train over the patched LlamaForCausalLM, model/llm.py:89-127, with
// peft LoRA on q_proj / v_proj, train_ranker.py:71-79). Data-gradient GEMMs reuse the forward's NT kernel on
// TRANSPOSED copies of the frozen weights (caller-owned, made once with lr_transpose_bf16): 13.5 GB more for
// Llama-2-7b, nothing against 288 GB, and the backward then runs at the forward's GEMM rate. Every activation the
// backward needs is kept (~100 KB per token and layer): no recomputation (the reference checkpoints, config.py:269,
// because its cards are 24-80 GB).
#include <stdlib.h>
#include <string.h>

#include "llama_train.h"

typedef unsigned short u16;

struct lr_llama_lora {
  lr_llama* base;
  LrLoraTrainConfig cfg;
  LrLlamaLayerWeightsT* layers_t;  // host array
  const u16* lm_head_t;
  float *params, *grads, *m, *v;   // flat fp32 device buffers, n_params each
  size_t n_params, per_layer;
  u16* work;                       // bf16 working copies, work_per_layer elements per layer
  size_t work_per_layer;
  float* scratch;                  // [8]: 0 loss sum, 1 bad targets, 2 sumsq, 3 lr, 4 limit
  int* ctr;                        // [4]: 0 optimizer steps
  uint32_t pass;                   // host-side pass counter (dropout streams)
  int qcols, kcols, vcols;
  // The adapter-only kernels (rank-r products, dA / dB reductions: ~5 % of a step, a few hundred workgroups each) run
  // on a low-priority side stream next to the big GEMM they are independent of, and fill the CUs its last, partly
  // empty round of tiles leaves idle (7 k tokens: 448 tiles on 256 CUs). LR_LORA_OVERLAP=0 keeps everything in order.
  hipStream_t side;
  hipEvent_t ev_fork, ev_join;
};

static int fork_side(lr_llama_lora* h, hipStream_t main, hipStream_t* work) {
  *work = main;
  if (!h->side) return LR_OK;
  LR_CHECK_HIP(hipEventRecord(h->ev_fork, main));
  LR_CHECK_HIP(hipStreamWaitEvent(h->side, h->ev_fork, 0));
  *work = h->side;
  return LR_OK;
}
static int join_side(lr_llama_lora* h, hipStream_t main) {
  if (!h->side) return LR_OK;
  LR_CHECK_HIP(hipEventRecord(h->ev_join, h->side));
  LR_CHECK_HIP(hipStreamWaitEvent(main, h->ev_join, 0));
  return LR_OK;
}

struct LoraStateLayout {
  size_t params, grads, m, v, work, scratch, ctr, total;
};
static LoraStateLayout state_layout(const LrLlamaConfig& c, int r) {
  LoraStateLayout s;
  const size_t qcols = (size_t)c.num_heads * c.head_dim, vcols = (size_t)c.num_kv_heads * c.head_dim;
  const size_t n = (size_t)c.num_layers * (2 * (size_t)r * c.hidden_size + (size_t)r * (qcols + vcols));
  const size_t wl = 2 * (size_t)LT_RP * c.hidden_size + (size_t)LT_RP * (qcols + vcols);
  size_t o = 0;
  auto take = [&](size_t bytes) {
    size_t at = o;
    o += lr_align_up(bytes, 256);
    return at;
  };
  s.params = take(n * 4);
  s.grads = take(n * 4);
  s.m = take(n * 4);
  s.v = take(n * 4);
  s.work = take((size_t)c.num_layers * wl * 2);
  s.scratch = take(8 * 4);
  s.ctr = take(4 * 4);
  s.total = o;
  return s;
}

static int check_cfg(const lr_llama_t* base, const LrLoraTrainConfig* cfg, const char* who) {
  if (!base || !cfg) LR_FAIL(LR_EINVAL, "%s: null argument", who);
  if (cfg->r < 1 || cfg->r > LT_RP) LR_FAIL(LR_EUNSUPPORTED, "%s: LoRA rank %d outside [1, %d]", who, cfg->r, LT_RP);
  if (cfg->dropout < 0.f || cfg->dropout >= 1.f) LR_FAIL(LR_EINVAL, "%s: dropout outside [0, 1)", who);
  const LrLlamaConfig& c = base->cfg;
  if (c.hidden_size % 64 != 0 || (c.num_heads * c.head_dim) % 64 != 0 || (c.num_kv_heads * c.head_dim) % 64 != 0)
    LR_FAIL(LR_EUNSUPPORTED, "%s: hidden size and q / v widths must be multiples of 64", who);
  if (c.head_dim % 8 != 0) LR_FAIL(LR_EUNSUPPORTED, "%s: head_dim %d (must be a multiple of 8)", who, c.head_dim);
  return LR_OK;
}

extern "C" size_t lr_llama_lora_state_bytes(const lr_llama_t* base, const LrLoraTrainConfig* cfg) {
  if (check_cfg(base, cfg, "lr_llama_lora_state_bytes")) return 0;
  return state_layout(base->cfg, cfg->r).total;
}

extern "C" int lr_llama_lora_create(lr_llama_t* base, const LrLlamaWeightsTDesc* wt, const LrLoraTrainConfig* cfg,
                                    void* state, size_t state_bytes, void* hip_stream, lr_llama_lora_t** out) {
  int rc = check_cfg(base, cfg, "lr_llama_lora_create");
  if (rc) return rc;
  if (!wt || !wt->layers || !wt->lm_head_t || !state || !out) LR_FAIL(LR_EINVAL, "lr_llama_lora_create: null argument");
  const LrLlamaConfig& c = base->cfg;
  for (int l = 0; l < c.num_layers; ++l) {
    const LrLlamaLayerWeightsT& t = wt->layers[l];
    if (!t.wqkv_t || !t.wo_t || !t.wgu_t || !t.wdown_t)
      LR_FAIL(LR_EINVAL, "lr_llama_lora_create: layer %d has a null transposed weight", l);
  }
  const LoraStateLayout s = state_layout(c, cfg->r);
  if (state_bytes < s.total)
    LR_FAIL(LR_EWORKSPACE, "lr_llama_lora_create: state needs %zu bytes, have %zu", s.total, state_bytes);
  lr_llama_lora* h = (lr_llama_lora*)calloc(1, sizeof(lr_llama_lora));
  if (!h) LR_FAIL(LR_EINVAL, "lr_llama_lora_create: out of host memory");
  h->base = base;
  h->cfg = *cfg;
  h->layers_t = (LrLlamaLayerWeightsT*)malloc(sizeof(LrLlamaLayerWeightsT) * c.num_layers);
  memcpy(h->layers_t, wt->layers, sizeof(LrLlamaLayerWeightsT) * c.num_layers);
  h->lm_head_t = wt->lm_head_t;
  char* b = (char*)state;
  h->params = (float*)(b + s.params);
  h->grads = (float*)(b + s.grads);
  h->m = (float*)(b + s.m);
  h->v = (float*)(b + s.v);
  h->work = (u16*)(b + s.work);
  h->scratch = (float*)(b + s.scratch);
  h->ctr = (int*)(b + s.ctr);
  h->qcols = c.num_heads * c.head_dim;
  h->kcols = h->vcols = c.num_kv_heads * c.head_dim;
  h->per_layer = 2 * (size_t)cfg->r * c.hidden_size + (size_t)cfg->r * (h->qcols + h->vcols);
  h->n_params = (size_t)c.num_layers * h->per_layer;
  h->work_per_layer = 2 * (size_t)LT_RP * c.hidden_size + (size_t)LT_RP * (h->qcols + h->vcols);
  const char* ov = getenv("LR_LORA_OVERLAP");
  if (!ov || ov[0] != '0') {
    // NORMAL priority (round 5). The side stream used to be created with the device's lowest priority, so that the rank-r products
    // never took a CU from the GEMM they run beside. But a lowest-priority stream -- alive or destroyed -- makes this HIP runtime
    // map some LATER normal-priority streams of the process onto its hardware queue: every third or fourth torch stream created
    // afterwards ran the same captured graph 2-3 x slower (tools/diag/stream_index_probe.py: 0.50 -> 0.94-1.56 ms; a normal-priority
    // stream has no such effect). LR_LORA_SIDE_PRIORITY=low restores the old behaviour for A/B runs.
    int lo = 0, hi = 0, prio = 0;
    const char* sp = getenv("LR_LORA_SIDE_PRIORITY");
    if (hipDeviceGetStreamPriorityRange(&lo, &hi) != hipSuccess ||
        hipStreamCreateWithPriority(&h->side, hipStreamNonBlocking, (sp && sp[0] == 'l') ? lo : prio) != hipSuccess ||
        hipEventCreateWithFlags(&h->ev_fork, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&h->ev_join, hipEventDisableTiming) != hipSuccess) {
      free(h->layers_t);
      free(h);
      LR_FAIL(LR_EHIP, "lr_llama_lora_create: side stream / events");
    }
  }
  // parameters are the caller's to fill (lr_llama_lora_buffers); gradients, moments and counters start at zero
  hipStream_t st = (hipStream_t)hip_stream;
  if (hipMemsetAsync(b + s.grads, 0, s.total - s.grads, st) != hipSuccess) {
    free(h->layers_t);
    free(h);
    LR_FAIL(LR_EHIP, "lr_llama_lora_create: hipMemsetAsync failed");
  }
  *out = h;
  return LR_OK;
}

extern "C" void lr_llama_lora_destroy(lr_llama_lora_t* h) {
  if (!h) return;
  if (h->side) {
    (void)hipStreamSynchronize(h->side);
    (void)hipEventDestroy(h->ev_fork);
    (void)hipEventDestroy(h->ev_join);
    (void)hipStreamDestroy(h->side);
  }
  free(h->layers_t);
  free(h);
}

extern "C" int lr_llama_lora_buffers(lr_llama_lora_t* h, float** params, float** grads, float** m, float** v,
                                     size_t* n) {
  if (!h) LR_FAIL(LR_EINVAL, "lr_llama_lora_buffers: null handle");
  if (params) *params = h->params;
  if (grads) *grads = h->grads;
  if (m) *m = h->m;
  if (v) *v = h->v;
  if (n) *n = h->n_params;
  return LR_OK;
}

// which: 0 q_proj, 1 v_proj; ab: 0 lora_A [r][hidden], 1 lora_B [out][r] (peft's layouts)
extern "C" int lr_llama_lora_param_range(const lr_llama_lora_t* h, int32_t layer, int32_t which, int32_t ab,
                                         size_t* offset, size_t* count) {
  if (!h || !offset || !count) LR_FAIL(LR_EINVAL, "lr_llama_lora_param_range: null argument");
  const LrLlamaConfig& c = h->base->cfg;
  if (layer < 0 || layer >= c.num_layers || which < 0 || which > 1 || ab < 0 || ab > 1)
    LR_FAIL(LR_EINVAL, "lr_llama_lora_param_range: layer=%d which=%d ab=%d", layer, which, ab);
  const size_t r = h->cfg.r, d = c.hidden_size;
  const size_t aq = 0, bq = r * d, av = bq + r * h->qcols, bv = av + r * d;
  const size_t off[2][2] = {{aq, bq}, {av, bv}};
  const size_t cnt[2][2] = {{r * d, r * (size_t)h->qcols}, {r * d, r * (size_t)h->vcols}};
  *offset = (size_t)layer * h->per_layer + off[which][ab];
  *count = cnt[which][ab];
  return LR_OK;
}

// ---- workspace -----------------------------------------------------------------------------------------------
struct LoraLayerSave {
  u16 *x, *xn, *qkv, *att, *xmid, *gu, *t;
  float* lse;
};
struct LoraWs {
  int32_t *tok_pos, *tok_seq, *last_rows;
  float* rope;
  u16 *x_final, *xn2, *hmid;              // forward transients (xn2/hmid alias backward transients)
  u16 *dx, *dh, *dxn, *datt, *dqkv, *dt;  // backward
  float *dsum, *dkv32;
  u16 *xg, *hn, *logits, *dhn;            // loss head, m rows
  size_t save0, save_stride;              // per-layer saved activations: base + l * save_stride
  size_t o_x, o_xn, o_qkv, o_att, o_xmid, o_gu, o_t, o_lse;
  char* base;
  size_t total;
};
static LoraWs carve(const LrLlamaConfig& c, int n_tok, int B, int m, int slots, bool training, char* base) {
  LoraWs w;
  memset(&w, 0, sizeof(w));
  w.base = base;
  size_t o = 0;
  auto take = [&](size_t bytes) {
    size_t at = o;
    o += lr_align_up(bytes, 256);
    return at;
  };
  const size_t n = (size_t)n_tok, d = c.hidden_size, f = c.intermediate_size;
  const size_t qcols = (size_t)c.num_heads * c.head_dim, kv = (size_t)c.num_kv_heads * c.head_dim;
  const size_t qw = qcols + 2 * kv;
  w.tok_pos = (int32_t*)(base + take(n * 4));
  w.tok_seq = (int32_t*)(base + take(n * 4));
  w.last_rows = (int32_t*)(base + take((size_t)(B > 0 ? B : 1) * 4));
  w.rope = (float*)(base + take((size_t)c.max_positions * (c.head_dim / 2) * 2 * sizeof(float)));
  w.x_final = (u16*)(base + take(n * d * 2));
  // one slot of saved activations
  size_t so = 0;
  auto stake = [&](size_t bytes) {
    size_t at = so;
    so += lr_align_up(bytes, 256);
    return at;
  };
  w.o_x = stake(n * d * 2);
  w.o_xn = stake(n * d * 2);
  w.o_qkv = stake(n * qw * 2);
  w.o_att = stake(n * qcols * 2);
  w.o_xmid = stake(n * d * 2);
  w.o_gu = stake(n * 2 * f * 2);
  w.o_t = stake(n * 2 * LT_RP * 2);
  w.o_lse = stake(n * c.num_heads * 4);
  w.save_stride = so;
  w.save0 = take(so * (size_t)slots);
  // transients: forward's (xn2, hmid) share memory with backward's (dxn, dh)
  w.xn2 = w.dxn = (u16*)(base + take(n * d * 2));
  w.hmid = w.dh = (u16*)(base + take(n * f * 2));
  if (training) {
    w.dx = (u16*)(base + take(n * d * 2));
    w.datt = (u16*)(base + take(n * qcols * 2));
    w.dqkv = (u16*)(base + take(n * qw * 2));
    w.dt = (u16*)(base + take(n * 2 * LT_RP * 2));
    w.dsum = (float*)(base + take(n * c.num_heads * 4));
    w.dkv32 = c.head_dim == 128 ? nullptr : (float*)(base + take(n * 2 * kv * 4));
    const size_t mm = (size_t)(m > 0 ? m : 1);
    w.xg = (u16*)(base + take(mm * d * 2));
    w.hn = (u16*)(base + take(mm * d * 2));
    w.logits = (u16*)(base + take(mm * c.vocab_size * 2));
    w.dhn = (u16*)(base + take(mm * d * 2));
  }
  w.total = o;
  return w;
}
static LoraLayerSave slot(const LoraWs& w, int l) {
  char* b = w.base + w.save0 + (size_t)l * w.save_stride;
  LoraLayerSave s;
  s.x = (u16*)(b + w.o_x);
  s.xn = (u16*)(b + w.o_xn);
  s.qkv = (u16*)(b + w.o_qkv);
  s.att = (u16*)(b + w.o_att);
  s.xmid = (u16*)(b + w.o_xmid);
  s.gu = (u16*)(b + w.o_gu);
  s.t = (u16*)(b + w.o_t);
  s.lse = (float*)(b + w.o_lse);
  return s;
}

extern "C" size_t lr_llama_lora_workspace_bytes(const lr_llama_lora_t* h, int32_t max_tokens, int32_t max_seqs,
                                                int32_t max_loss_rows) {
  if (!h || max_tokens < 1) return 0;
  if (max_seqs < 1) max_seqs = 1;
  return carve(h->base->cfg, max_tokens, max_seqs, max_loss_rows, h->base->cfg.num_layers, true, nullptr).total;
}
extern "C" size_t lr_llama_lora_eval_workspace_bytes(const lr_llama_lora_t* h, int32_t max_tokens, int32_t max_seqs) {
  if (!h || max_tokens < 1) return 0;
  if (max_seqs < 1) max_seqs = 1;
  return carve(h->base->cfg, max_tokens, max_seqs, 0, 1, false, nullptr).total;
}

#define RUN(x)         \
  do {                 \
    rc = (x);          \
    if (rc) return rc; \
  } while (0)

static u16* work_of(const lr_llama_lora* h, int l) { return h->work + (size_t)l * h->work_per_layer; }

// bf16 working copies of every layer's adapters from the fp32 masters
static int prep_adapters(lr_llama_lora* h, hipStream_t st) {
  const LrLlamaConfig& c = h->base->cfg;
  const size_t r = h->cfg.r, d = c.hidden_size;
  int rc;
  for (int l = 0; l < c.num_layers; ++l) {
    const float* p = h->params + (size_t)l * h->per_layer;
    u16* w = work_of(h, l);
    RUN(lr_launch_prep_lora(p, p + r * d, p + r * d + r * h->qcols, p + 2 * r * d + r * h->qcols, (int)r, (int)d,
                            h->qcols, h->vcols, c.head_dim, w, w + 2 * LT_RP * d, w + 2 * LT_RP * d + LT_RP * h->qcols,
                            st));
  }
  return LR_OK;
}

static int validate_batch(const lr_llama_lora* h, const int32_t* cu_host, int B, int* n_out, int* maxT_out) {
  const LrLlamaConfig& c = h->base->cfg;
  if (B < 1) LR_FAIL(LR_EINVAL, "llama lora: B=%d", B);
  if (cu_host[0] != 0) LR_FAIL(LR_EINVAL, "llama lora: cu_seqlens[0] must be 0");
  int maxT = 0;
  for (int b = 0; b < B; ++b) {
    const int t = cu_host[b + 1] - cu_host[b];
    if (t < 1) LR_FAIL(LR_EINVAL, "llama lora: prompt %d is empty", b);
    if (t > maxT) maxT = t;
  }
  if (maxT > c.max_positions)
    LR_FAIL(LR_EINVAL, "llama lora: prompt of %d tokens exceeds max_positions %d", maxT, c.max_positions);
  *n_out = cu_host[B];
  *maxT_out = maxT;
  return LR_OK;
}

// forward with live adapters; layer l's activations go to slot (save ? l : 0); the final residual lands in ws.x_final
static int forward(lr_llama_lora* h, const int32_t* ids, const int32_t* cu, const int32_t* cu_host, int B, int n,
                   int maxT, const LoraWs& ws, bool save, float drop_p, hipStream_t st) {
  const LrLlamaConfig& c = h->base->cfg;
  const int d = c.hidden_size, f = c.intermediate_size, nh = c.num_heads, nkv = c.num_kv_heads, hd = c.head_dim;
  const int qw = (nh + 2 * nkv) * hd;
  const float scaling = h->cfg.alpha / (float)h->cfg.r;
  const int gv = h->base->gemm_variant == 5 ? 0 : h->base->gemm_variant;
  int rc;
  RUN(lr_launch_token_meta(cu, B, 0, nullptr, ws.tok_pos, nullptr, ws.last_rows, st));
  RUN(lr_launch_rope_table(ws.rope, maxT, hd, c.rope_theta, st));
  RUN(lr_launch_embed(ids, nullptr, h->base->embed, c.vocab_size, d, slot(ws, 0).x, n, st));
  for (int l = 0; l < c.num_layers; ++l) {
    const LrLlamaLayerWeights& w = h->base->layers[l];
    const LoraLayerSave s = slot(ws, save ? l : 0);
    u16* x_next = l + 1 < c.num_layers ? (save ? slot(ws, l + 1).x : s.x) : ws.x_final;
    const u16* wk = work_of(h, l);
    const u16 *a_cat = wk, *bq_t = wk + 2 * LT_RP * (size_t)d, *bv_t = bq_t + LT_RP * (size_t)h->qcols;
    const uint32_t stream = lr_lora_drop_stream(h->cfg.seed, h->pass, (uint32_t)l);
    RUN(lr_launch_rmsnorm(s.x, w.input_norm, s.xn, n, d, c.rms_eps, nullptr, st));
    hipStream_t sd;
    RUN(fork_side(h, st, &sd));  // t = drop(xn) A^T next to the QKV GEMM: both only read xn
    RUN(lr_launch_skinny(s.xn, d, n, d, a_cat, 2, s.t, 2 * LT_RP, 0, 1.0f, stream, drop_p, sd));
    RUN(lr_launch_gemm(s.xn, w.wqkv, s.qkv, nullptr, n, qw, d, LR_EPI_STORE, gv, st));
    RUN(join_side(h, st));
    RUN(lr_launch_lora_rope_fwd(s.qkv, n, qw, h->qcols, h->kcols, hd, s.t, bq_t, bv_t, h->cfg.r, scaling, ws.tok_pos,
                                ws.rope, st));
    RUN(lr_launch_attention_lse(s.qkv, s.att, s.lse, cu, cu_host, B, n, nh, nkv, hd, h->base->attn_variant, st));
    RUN(lr_launch_gemm(s.att, w.wo, s.xmid, s.x, n, d, nh * hd, LR_EPI_RESIDUAL, gv, st));
    RUN(lr_launch_rmsnorm(s.xmid, w.post_norm, ws.xn2, n, d, c.rms_eps, nullptr, st));
    RUN(lr_launch_gemm(ws.xn2, w.wgu, s.gu, nullptr, n, 2 * f, d, LR_EPI_STORE, gv, st));
    RUN(lr_launch_swiglu_fwd(s.gu, ws.hmid, n, f, st));
    RUN(lr_launch_gemm(ws.hmid, w.wdown, x_next, s.xmid, n, d, f, LR_EPI_RESIDUAL, gv, st));
  }
  return LR_OK;
}

extern "C" int lr_llama_lora_loss_grad(lr_llama_lora_t* h, const int32_t* packed_ids, const int32_t* cu_seqlens,
                                       const int32_t* cu_seqlens_host, int32_t B, const int32_t* loss_rows,
                                       const int32_t* loss_targets, int32_t m, float grad_scale, int32_t accumulate,
                                       float* out, void* workspace, size_t workspace_bytes, void* hip_stream) {
  if (!h || !packed_ids || !cu_seqlens || !cu_seqlens_host || !loss_rows || !loss_targets || !out || !workspace)
    LR_FAIL(LR_EINVAL, "lr_llama_lora_loss_grad: null argument");
  if (m < 1) LR_FAIL(LR_EINVAL, "lr_llama_lora_loss_grad: no labelled rows (m=%d)", m);
  hipStream_t st = (hipStream_t)hip_stream;
  const LrLlamaConfig& c = h->base->cfg;
  int n, maxT, rc;
  RUN(validate_batch(h, cu_seqlens_host, B, &n, &maxT));
  if (m > n) LR_FAIL(LR_EINVAL, "lr_llama_lora_loss_grad: %d labelled rows for %d tokens", m, n);
  const LoraWs ws = carve(c, n, B, m, c.num_layers, true, (char*)workspace);
  if (ws.total > workspace_bytes)
    LR_FAIL(LR_EWORKSPACE, "lr_llama_lora_loss_grad: workspace needs %zu bytes for %d tokens, have %zu", ws.total, n,
            workspace_bytes);
  const int d = c.hidden_size, f = c.intermediate_size, nh = c.num_heads, nkv = c.num_kv_heads, hd = c.head_dim;
  const int qw = (nh + 2 * nkv) * hd, r = h->cfg.r;
  const float scaling = h->cfg.alpha / (float)r, drop_p = h->cfg.dropout;
  const int gv = h->base->gemm_variant == 5 ? 0 : h->base->gemm_variant;
  h->pass += 1;
  if (!accumulate) LR_CHECK_HIP(hipMemsetAsync(h->grads, 0, h->n_params * sizeof(float), st));
  LR_CHECK_HIP(hipMemsetAsync(h->scratch, 0, 2 * sizeof(float), st));
  RUN(prep_adapters(h, st));
  RUN(forward(h, packed_ids, cu_seqlens, cu_seqlens_host, B, n, maxT, ws, true, drop_p, st));

  // ---- loss head on the labelled rows only (model/llm.py:113-126)
  RUN(lr_launch_gather_rows(ws.x_final, loss_rows, m, d, ws.xg, st));
  RUN(lr_launch_rmsnorm(ws.xg, h->base->final_norm, ws.hn, m, d, c.rms_eps, nullptr, st));
  RUN(lr_launch_gemm(ws.hn, h->base->lm_head, ws.logits, nullptr, m, c.vocab_size, d, LR_EPI_STORE, gv, st));
  RUN(lr_launch_ce_bf16(ws.logits, m, c.vocab_size, loss_targets, grad_scale / (float)m, h->scratch, st));
  RUN(lr_launch_finish_loss(h->scratch, m, out, st));
  RUN(lr_launch_gemm(ws.logits, h->lm_head_t, ws.dhn, nullptr, m, d, c.vocab_size, LR_EPI_STORE, gv, st));
  LR_CHECK_HIP(hipMemsetAsync(ws.dx, 0, (size_t)n * d * 2, st));
  RUN(lr_launch_rmsnorm_bwd(ws.dhn, ws.xg, h->base->final_norm, nullptr, ws.dx, m, d, c.rms_eps, loss_rows, nullptr,
                            nullptr, 0, 0, 0.f, st));

  // ---- backward through the layers; ws.dx = gradient of the residual stream
  for (int l = c.num_layers - 1; l >= 0; --l) {
    const LrLlamaLayerWeights& w = h->base->layers[l];
    const LrLlamaLayerWeightsT& wt = h->layers_t[l];
    const LoraLayerSave s = slot(ws, l);
    const u16* wk = work_of(h, l);
    const u16 *a_cat = wk, *bq_t = wk + 2 * LT_RP * (size_t)d, *bv_t = bq_t + LT_RP * (size_t)h->qcols;
    float* g = h->grads + (size_t)l * h->per_layer;
    float *daq = g, *dbq = g + (size_t)r * d, *dav = dbq + (size_t)r * h->qcols, *dbv = dav + (size_t)r * d;
    const uint32_t stream = lr_lora_drop_stream(h->cfg.seed, h->pass, (uint32_t)l);
    // MLP: x_out = xmid + down(silu(gate) * up)
    RUN(lr_launch_gemm(ws.dx, wt.wdown_t, ws.dh, nullptr, n, f, d, LR_EPI_STORE, gv, st));
    RUN(lr_launch_swiglu_bwd(s.gu, ws.dh, n, f, st));
    RUN(lr_launch_gemm(s.gu, wt.wgu_t, ws.dxn, nullptr, n, d, 2 * f, LR_EPI_STORE, gv, st));
    RUN(lr_launch_rmsnorm_bwd(ws.dxn, s.xmid, w.post_norm, ws.dx, ws.dx, n, d, c.rms_eps, nullptr, nullptr, nullptr, 0,
                              0, 0.f, st));
    // attention block: xmid = x + o_proj(attention(q, k, v))
    RUN(lr_launch_gemm(ws.dx, wt.wo_t, ws.datt, nullptr, n, nh * hd, d, LR_EPI_STORE, gv, st));
    // ... down to the gradient of the UNROTATED q, k, v (the inverse rotation rides in the attention passes)
    RUN(lr_launch_attention_bwd(s.qkv, s.att, ws.datt, s.lse, ws.dqkv, ws.dsum, ws.dkv32, cu_seqlens, cu_seqlens_host, B,
                                n, nh, nkv, hd, h->base->attn_variant, st, ws.tok_pos, ws.rope));
    // adapters (side stream, next to the qkv data-gradient GEMM; both only read dqkv):
    // d B, d t = scaling * (d q B_q | d v B_v), d A
    hipStream_t sd;
    RUN(fork_side(h, st, &sd));
    RUN(lr_launch_lora_db(ws.dqkv, n, qw, h->qcols, h->kcols, hd, s.t, r, scaling, dbq, dbv, sd));
    RUN(lr_launch_skinny(ws.dqkv, qw, n, h->qcols, bq_t, 1, ws.dt, 2 * LT_RP, 0, scaling, 0, 0.f, sd));
    RUN(lr_launch_skinny(ws.dqkv + h->qcols + h->kcols, qw, n, h->vcols, bv_t, 1, ws.dt, 2 * LT_RP, LT_RP, scaling, 0,
                         0.f, sd));
    RUN(lr_launch_lora_da(s.xn, n, d, ws.dt, r, stream, drop_p, daq, dav, sd));
    if (l > 0)  // below layer 0 only the frozen embedding is left: its input gradient has no reader
      RUN(lr_launch_gemm(ws.dqkv, wt.wqkv_t, ws.dxn, nullptr, n, d, qw, LR_EPI_STORE, gv, st));
    RUN(join_side(h, st));
    if (l > 0)
      RUN(lr_launch_rmsnorm_bwd(ws.dxn, s.x, w.input_norm, ws.dx, ws.dx, n, d, c.rms_eps, nullptr, ws.dt, a_cat, r,
                                stream, drop_p, st));
  }
  return LR_OK;
}

extern "C" int lr_llama_lora_apply(lr_llama_lora_t* h, float lr, float max_grad_norm, float* out_norm,
                                   void* hip_stream) {
  if (!h) LR_FAIL(LR_EINVAL, "lr_llama_lora_apply: null handle");
  return lr_launch_lora_adamw(h->params, h->grads, h->m, h->v, h->n_params, h->scratch + 2, h->ctr, lr, max_grad_norm,
                              h->cfg.beta1, h->cfg.beta2, h->cfg.eps, h->cfg.weight_decay, out_norm,
                              (hipStream_t)hip_stream);
}

// scoring with the adapters as they are now (validation during training, trainer/llm.py:123-126): the forward
// above without dropout, then the inference head (final norm + verbalizer rows of lm_head at each prompt's end)
extern "C" int lr_llama_lora_prefill_verbalize(lr_llama_lora_t* h, const int32_t* packed_ids,
                                               const int32_t* cu_seqlens, const int32_t* cu_seqlens_host, int32_t B,
                                               const int32_t* label_token_ids, int32_t C, float* out_scores,
                                               void* workspace, size_t workspace_bytes, void* hip_stream) {
  if (!h || !packed_ids || !cu_seqlens || !cu_seqlens_host || !label_token_ids || !out_scores || !workspace || C < 1)
    LR_FAIL(LR_EINVAL, "lr_llama_lora_prefill_verbalize: bad argument");
  hipStream_t st = (hipStream_t)hip_stream;
  const LrLlamaConfig& c = h->base->cfg;
  int n, maxT, rc;
  RUN(validate_batch(h, cu_seqlens_host, B, &n, &maxT));
  const LoraWs ws = carve(c, n, B, 0, 1, false, (char*)workspace);
  if (ws.total > workspace_bytes)
    LR_FAIL(LR_EWORKSPACE, "lr_llama_lora_prefill_verbalize: workspace needs %zu bytes for %d tokens, have %zu",
            ws.total, n, workspace_bytes);
  RUN(prep_adapters(h, st));
  RUN(forward(h, packed_ids, cu_seqlens, cu_seqlens_host, B, n, maxT, ws, false, 0.f, st));
  return lr_launch_head(ws.x_final, ws.last_rows, h->base->final_norm, h->base->lm_head, label_token_ids, B, C,
                        c.hidden_size, c.rms_eps, out_scores, c.vocab_size, st);
}

extern "C" int lr_transpose_bf16(const uint16_t* src, int32_t rows, int32_t cols, uint16_t* dst, void* hip_stream) {
  if (!src || !dst) LR_FAIL(LR_EINVAL, "lr_transpose_bf16: null pointer");
  return lr_launch_transpose_bf16(src, rows, cols, dst, (hipStream_t)hip_stream);
}

extern "C" int lr_attention_varlen_lse(const uint16_t* qkv, uint16_t* out, float* lse, const int32_t* cu_seqlens,
                                       const int32_t* cu_seqlens_host, int32_t B, int32_t num_heads,
                                       int32_t num_kv_heads, int32_t head_dim, int32_t variant, void* hip_stream) {
  if (!qkv || !out || !lse || !cu_seqlens || !cu_seqlens_host || B < 1)
    LR_FAIL(LR_EINVAL, "lr_attention_varlen_lse: bad argument");
  return lr_launch_attention_lse(qkv, out, lse, cu_seqlens, cu_seqlens_host, B, cu_seqlens_host[B], num_heads,
                                 num_kv_heads, head_dim, variant, (hipStream_t)hip_stream);
}

extern "C" size_t lr_attention_bwd_scratch_bytes(int32_t total, int32_t num_heads, int32_t num_kv_heads,
                                                 int32_t head_dim) {
  if (total < 1) return 0;
  return lr_align_up((size_t)total * num_heads * 4, 256) + (size_t)total * 2 * num_kv_heads * head_dim * 4;
}

extern "C" int lr_attention_varlen_bwd(const uint16_t* qkv, const uint16_t* out, const uint16_t* d_out,
                                       const float* lse, uint16_t* dqkv, const int32_t* cu_seqlens,
                                       const int32_t* cu_seqlens_host, int32_t B, int32_t num_heads,
                                       int32_t num_kv_heads, int32_t head_dim, int32_t variant, void* scratch,
                                       size_t scratch_bytes, void* hip_stream) {
  if (!qkv || !out || !d_out || !lse || !dqkv || !cu_seqlens || !cu_seqlens_host || !scratch || B < 1)
    LR_FAIL(LR_EINVAL, "lr_attention_varlen_bwd: bad argument");
  const int n = cu_seqlens_host[B];
  if (scratch_bytes < lr_attention_bwd_scratch_bytes(n, num_heads, num_kv_heads, head_dim))
    LR_FAIL(LR_EWORKSPACE, "lr_attention_varlen_bwd: scratch too small");
  float* dsum = (float*)scratch;
  float* dkv32 = (float*)((char*)scratch + lr_align_up((size_t)n * num_heads * 4, 256));
  return lr_launch_attention_bwd(qkv, out, d_out, lse, dqkv, dsum, dkv32, cu_seqlens, cu_seqlens_host, B, n, num_heads,
                                 num_kv_heads, head_dim, variant, (hipStream_t)hip_stream);
}
