/* llamarec_mi355x.h -- C ABI of libllamarec_mi355x.so (MI355X / gfx950).
 *
 * Drop-in boundary for the two-stage retrieve-then-rerank scoring path of GarciaLnk/LlamaRec.
 * The reference has no FFI: its seams are Python nn.Module.forward calls. Each entry point below
 * names the reference call site it replaces (paths are into the reference tree).
 *
 * Conventions
 *  - plain pointers and sizes only; no torch / C++ types cross the boundary.
 *  - the CALLER owns every buffer (weights, workspaces, inputs, outputs; normally torch tensors);
 *    the library owns nothing but small host-side handles. No entry point allocates device
 *    memory or synchronises the device, so every call is hipGraph-capturable.
 *  - all device work is enqueued on the caller's hipStream_t (passed as void*); the caller
 *    synchronises.
 *  - return value: 0 = LR_OK, negative = LR_E*; lr_last_error() gives a thread-local message.
 *  - re-entrant across handles; one in-flight call per handle+workspace.
 */
#ifndef LLAMAREC_MI355X_H
#define LLAMAREC_MI355X_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LR_OK 0
#define LR_EINVAL (-1)   /* bad argument (null pointer, shape out of range) */
#define LR_EUNSUPPORTED (-2) /* configuration outside what the kernels implement */
#define LR_EHIP (-3)     /* HIP runtime error (launch failure, wrong device, ...) */
#define LR_EWORKSPACE (-4) /* caller-provided workspace too small */

#define LR_MAX_LRU_BLOCKS 4
#define LR_MAX_TOPK 64

const char* lr_last_error(void);
/* "llamarec_mi355x <semver> gfx950" */
const char* lr_version(void);

/* ------------------------------------------------------------------------------------------
 * Stage 1: LRURec retriever  (model/lru.py:8-175, trainer/lru.py:30-42,44-138)
 * ------------------------------------------------------------------------------------------ */

/* One LRUBlock's tensors exactly as in the reference state_dict (model/lru.py:89-175);
 * complex64 tensors are passed as interleaved (re,im) float pairs (torch.view_as_real). */
typedef struct LrLruBlockWeights {
  const float* params_log;  /* [3][128]      lru_layer.params_log (nu_log, theta_log, gamma_log) */
  const float* in_proj_w;   /* [128][64][2]  lru_layer.in_proj.weight  (complex64) */
  const float* in_proj_b;   /* [128][2]      lru_layer.in_proj.bias    (complex64) */
  const float* out_proj_w;  /* [64][128][2]  lru_layer.out_proj.weight (complex64) */
  const float* out_proj_b;  /* [64][2]       lru_layer.out_proj.bias   (complex64) */
  const float* ln1_w;       /* [64]          lru_layer.layer_norm.weight */
  const float* ln1_b;       /* [64]          lru_layer.layer_norm.bias */
  const float* ffn_w1;      /* [256][64]     feed_forward.w_1.weight */
  const float* ffn_b1;      /* [256]         feed_forward.w_1.bias */
  const float* ffn_w2;      /* [64][256]     feed_forward.w_2.weight */
  const float* ffn_b2;      /* [64]          feed_forward.w_2.bias */
  const float* ln2_w;       /* [64]          feed_forward.layer_norm.weight */
  const float* ln2_b;       /* [64]          feed_forward.layer_norm.bias */
} LrLruBlockWeights;

/* LRURec state_dict (SURVEY.md 8(a) a-W). All pointers are HOST pointers, fp32. */
typedef struct LrLruWeightsDesc {
  int32_t num_items;        /* V; the item table has V+1 rows (row 0 = pad id, a learned row) */
  int32_t hidden;           /* bert_hidden_units; must be 64 */
  int32_t num_blocks;       /* bert_num_blocks; 1..LR_MAX_LRU_BLOCKS */
  int32_t reserved;
  const float* item_emb;    /* [V+1][64]  embedding.token.weight (tied output table) */
  const float* item_bias;   /* [V+1]      model.bias */
  const float* emb_ln_w;    /* [64]       embedding.layer_norm.weight */
  const float* emb_ln_b;    /* [64]       embedding.layer_norm.bias */
  LrLruBlockWeights blocks[LR_MAX_LRU_BLOCKS];
} LrLruWeightsDesc;

typedef struct lr_lru lr_lru_t;

/* Bytes of the packed device image for a model with num_items = V, num_blocks blocks. */
size_t lr_lru_packed_bytes(int32_t num_items, int32_t num_blocks);

/* Host-side packer: state_dict layout -> the kernels' device layout (transposed projection
 * matrices, split re/im, recurrence coefficients lambda/gamma derived once). Pure CPU; writes
 * lr_lru_packed_bytes() bytes to host_out. The caller then copies the image to the GPU. */
int lr_lru_pack(const LrLruWeightsDesc* desc, void* host_out, size_t host_out_bytes);

/* Bind a handle to a packed image resident in DEVICE memory (kept alive by the caller).
 * Replaces LRURec.__init__ + load_state_dict (model/lru.py:8-14, trainer/base.py:158-161). */
int lr_lru_create(const void* packed_dev, size_t packed_bytes, int32_t num_items, int32_t num_blocks,
                  lr_lru_t** out);
void lr_lru_destroy(lr_lru_t* h);

/* Diagnostic switch: 1 (default) runs the encoder's LRU layer on the software-pipelined kernel (two 64-row tiles in
 * flight, role-split waves), 0 on the one-tile-at-a-time kernel. Both produce the same bits (tests/test_gpu_lru.py
 * compares them); the switch exists for that test and for A/B timing. No counterpart in the reference. */
int lr_lru_set_encoder_pipeline(lr_lru_t* h, int32_t enable);

/* Workspace (device) bytes needed by lr_lru_retrieve_topk / lr_lru_scores_last for up to
 * max_users histories of up to max_len ids per call (q rows, per-chunk partial top-K lists,
 * sorted history ids for the mask, the bound pre-pass's scratch for catalogs it serves). h must be a live handle
 * (the size depends on the catalog; NULL returns 0 and sets lr_last_error). The size is monotone in every argument:
 * a workspace sized for (max_users, max_k, max_len) serves every call with B <= max_users, K <= max_k, L <= max_len. */
size_t lr_lru_workspace_bytes(const lr_lru_t* h, int32_t max_users, int32_t max_k, int32_t max_len);

/* Encode B histories and return the hidden state of the LAST position, q[B][64] (fp32).
 * Replaces LRURec.forward up to (not including) the item GEMM, last position only
 * (model/lru.py:38-41,57-60,73-83; consumers slice [:, -1, :]: trainer/lru.py:33,67,105,
 * demo/inference.py:48).
 *   ids: DEVICE int64 [B][L] row-major, 0 = pad (left padded by the reference's datasets,
 *        dataloader/lru.py:142-151; zeros anywhere are honoured: mask = ids > 0).
 *   workspace: with at least lr_lru_workspace_bytes(h, B, 1, L) bytes the batched MFMA encoder runs (live
 *        tokens of all users packed into one row matrix); with less (or NULL) a one-workgroup-per-user
 *        kernel that needs no scratch -- same bits either way. */
int lr_lru_encode_last(lr_lru_t* h, const int64_t* ids, int32_t B, int32_t L, float* out_q,
                       void* workspace, size_t workspace_bytes, void* hip_stream);

/* Fused retrieve: encode -> item GEMM on the last position -> optional history/pad mask
 * (-1e9, trainer/lru.py:35-38,72-74,110-112) -> ordered top-K.
 * Replaces `model(seqs)[:, -1, :]` + masking loop + torch.topk / argsort
 * (trainer/lru.py:33-38,67-84,105-126).
 *   out_idx  : DEVICE int32 [B][K], score descending, ties -> lower item id first
 *   out_score: DEVICE fp32  [B][K] (may be NULL)
 *   K <= LR_MAX_TOPK. The [B][V+1] score matrix is never written to memory. */
int lr_lru_retrieve_topk(lr_lru_t* h, const int64_t* ids, int32_t B, int32_t L, int32_t K,
                         int32_t exclude_history, int32_t* out_idx, float* out_score,
                         void* workspace, size_t workspace_bytes, void* hip_stream);

/* Diagnostic: which path the LAST lr_lru_retrieve_topk call with exactly these (B, L, K, exclude_history) on this
 * workspace took -- *out_path = 0: the exact f32 pass over every item (catalogs and history lengths the bound does not
 * serve), 1: bf16 bound -> candidates -> exact rescoring, 2: that path overflowed a candidate list and the exact pass
 * redid the call (results are correct either way; 2 is a performance cliff the tests watch for). Synchronises the
 * stream. No counterpart in the reference (it materialises every score, model/lru.py:85). */
int lr_lru_topk_path(const lr_lru_t* h, int32_t B, int32_t L, int32_t K, int32_t exclude_history,
                     const void* workspace, size_t workspace_bytes, int32_t* out_path, void* hip_stream);

/* Compatibility path: materialise last-position scores [B][V+1] (fp32), optionally masked.
 * Replaces `self.model(seqs)[:, -1, :]` for callers that need the full score row
 * (trainer/lru.py:33, demo/inference.py:48). */
int lr_lru_scores_last(lr_lru_t* h, const int64_t* ids, int32_t B, int32_t L,
                       int32_t exclude_history, float* out_scores, void* workspace,
                       size_t workspace_bytes, void* hip_stream);

/* ------------------------------------------------------------------------------------------
 * Ranking metrics  (trainer/utils.py:43-90 with preprocessed ranks)
 * ------------------------------------------------------------------------------------------ */

/* Rank histogram (integer work on the GPU, exact and order-independent):
 *   ranked: DEVICE int32 [B][Kmax] ids best-first (item ids for stage 1, class ids 0..C-1 for
 *           stage 2); labels: DEVICE int64 [B]; hist: DEVICE int64 [Kmax+1].
 *   hist[p] += number of rows whose label sits at rank p (0-based); hist[Kmax] += rows whose
 *   label is not among the Kmax ranked ids. The caller zeroes hist first. Data-parallel ranks
 *   all-reduce (sum) the histogram: every metric below is a function of it. */
int lr_rank_histogram(const int32_t* ranked, int32_t Kmax, const int64_t* labels, int32_t B,
                      int64_t* hist, void* hip_stream);

/* Full descending ranking of C <= 64 class scores per row (ties -> lower class id first).
 * Replaces `(-scores).argsort(dim=1)` (trainer/utils.py:55) for the reranker's [N][20]
 * verbalizer scores (trainer/llm.py:63-72). scores: DEVICE fp32 [B][C]; out: DEVICE int32 [B][C].
 * items (DEVICE int32 [B][C], may be NULL): when given, out[b][rank] = items[b][class] -- the
 * candidate item ids in reranked order -- instead of the class id. */
int lr_rank_classes(const float* scores, int32_t B, int32_t C, const int32_t* items,
                    int32_t* out_ranked, void* hip_stream);

/* Pure CPU: Recall@k / MRR@k / NDCG@k NUMERATORS (sums over users, float64) from a HOST
 * histogram: sums[3*j+0..2] for k = ks[j]. One relevant item per user, so Recall's denominator
 * min(k, 1) and the ideal DCG are 1 (trainer/utils.py:63-88). Divide by the user count for the
 * reference's batch-mean (trainer/utils.py:68,76,87) / per-user average (trainer/lru.py:134-137). */
int lr_metrics_from_histogram(const int64_t* hist, int32_t Kmax, const int32_t* ks, int32_t nk,
                              double* sums);

/* ------------------------------------------------------------------------------------------
 * Stage 2: Llama-2 ranker -- single prefill + verbalizer gather
 * (model/llm.py:35-145, HF LlamaModel, trainer/verb.py:524-544, trainer/llm.py:63-72)
 * ------------------------------------------------------------------------------------------ */

typedef struct LrLlamaConfig {
  int32_t vocab_size;
  int32_t hidden_size;
  int32_t intermediate_size;
  int32_t num_layers;
  int32_t num_heads;
  int32_t num_kv_heads;          /* == num_heads for Llama-2-7b; GQA is supported */
  int32_t head_dim;              /* hidden_size / num_heads */
  int32_t max_positions;         /* RoPE table length (>= llm_max_text_len = 1536) */
  float rms_eps;
  float rope_theta;
} LrLlamaConfig;

/* All pointers are DEVICE pointers to bf16 (uint16) row-major [out][in] matrices, as stored
 * by HF (nn.Linear.weight). LoRA adapters must be merged into q_proj/v_proj beforehand
 * (W + (alpha/r) B A; config.py:257-260, train_ranker.py:71-79). */
typedef struct LrLlamaLayerWeights {
  const uint16_t* input_norm;    /* [hidden]                input_layernorm.weight */
  const uint16_t* wqkv;          /* [(nh+2*nkv)*hd][hidden] q_proj;k_proj;v_proj stacked; inside every q and
                                    k head the rows are pair-interleaved for the fused rotary epilogue:
                                    row 2i = HF row i, row 2i+1 = HF row i + hd/2 (lr_llama_pack_qkv).
                                    q.k dot products are unchanged by this permutation. */
  const uint16_t* wo;            /* [hidden][nh*hd]         o_proj */
  const uint16_t* post_norm;     /* [hidden]                post_attention_layernorm.weight */
  const uint16_t* wgu;           /* [2*inter][hidden]       gate_proj / up_proj, interleaved in
                                    blocks of 16 rows: rows [32t,32t+16) = gate[16t..16t+16),
                                    rows [32t+16,32t+32) = up[16t..16t+16)  (lr_llama_pack_gate_up) */
  const uint16_t* wdown;         /* [hidden][inter]         down_proj */
} LrLlamaLayerWeights;

typedef struct LrLlamaWeightsDesc {
  const uint16_t* embed;         /* [vocab][hidden]  model.embed_tokens.weight */
  const uint16_t* final_norm;    /* [hidden]         model.norm.weight */
  const uint16_t* lm_head;       /* [vocab][hidden]  lm_head.weight */
  const LrLlamaLayerWeights* layers; /* HOST array [num_layers] of device pointers */
} LrLlamaWeightsDesc;

typedef struct lr_llama lr_llama_t;

int lr_llama_create(const LrLlamaConfig* cfg, const LrLlamaWeightsDesc* w, lr_llama_t** out);
void lr_llama_destroy(lr_llama_t* h);

/* Kernel selection: 0 = auto (default), 1 = generic kernels (any shape; the in-library cross-check of the fast
 * ones), attention 2 = head_dim-128 MFMA flash attention (K/V by LDS-DMA, 128 query rows per workgroup),
 * attention 3 = head_dim-128 flash attention on 256-row tiles, one wave per SIMD, persistent workgroups (shared prefix of at
 * most 64 tokens, falls back to 2 otherwise; ahead of 2 on long prompts -- thousands of tokens -- and behind it on this
 * path's 460 .. 1 125-token prompts, so auto keeps 2),
 * gemm 4 = the ping-pong pipelined 256x256x64 MFMA GEMM (an error if a shape does not fit).
 * gemm 5 = LATENCY MODE for the online single-user path (demo/inference.py:56-76):
 * variant 4 plus split-K wherever the output tiles alone would leave most CUs idle (a 460-token prompt
 * gives o_proj / down_proj 32 tiles for 256 CUs); shapes that do not fit fall back as in auto. The
 * split-K summation order depends on the token count, so unlike the default a prompt's scores are
 * then reproducible only for the same packed batch size (bf16-level differences otherwise). */
int lr_llama_set_variants(lr_llama_t* h, int32_t gemm_variant, int32_t attention_variant);

/* Folded RMSNorm (optional, scoring path only). HF's layer computes  proj(norm_w * bf16(x * rstd))  with its own
 * read-and-write pass over the residual stream in front of q/k/v_proj and of gate/up_proj (LlamaRMSNorm, reached from
 * model/llm.py:89-100). Here the norm weight can be multiplied into the projection matrices once at load
 * (lr_fold_norm_bf16: out[j][k] = bf16(w[j][k] * norm_w[k]), DEVICE pointers) and handed over with
 * lr_llama_set_folded_norms (HOST arrays [num_layers] of DEVICE pointers: wqkv * diag(input_norm), wgu *
 * diag(post_norm), layouts as in LrLlamaLayerWeights; the caller keeps them alive). The prefill then reads each row
 * once for rstd = 1 / sqrt(mean(x^2) + eps) and the GEMM epilogue scales its fp32 accumulator rows by it:
 * rstd * (x . (W_j * w)) == (x * rstd * w) . W_j up to bf16 rounding points (two activation roundings fewer, one weight
 * rounding more; within the parity tolerance of tests/test_gpu_llama.py). The original matrices stay in use for the
 * last layer's B pruned rows and for LoRA fine-tuning. Passing two NULLs restores the separate RMSNorm pass. */
int lr_fold_norm_bf16(const uint16_t* w, const uint16_t* norm_w, int32_t rows, int32_t cols, uint16_t* out,
                      void* hip_stream);
int lr_llama_set_folded_norms(lr_llama_t* h, const uint16_t* const* wqkv_folded, const uint16_t* const* wgu_folded);

/* Last-layer pruning (default ON): after the final layer only each prompt's last token is consumed
 * (model/llm.py:131), so that layer computes K/V for all tokens but attention output, o_proj and the
 * MLP for B rows only. Results are unchanged up to bf16 summation order; disable for A/B tests. */
int lr_llama_set_last_layer_pruning(lr_llama_t* h, int32_t enable);

/* Device workspace bytes for up to max_tokens packed tokens and max_seqs sequences per call. */
size_t lr_llama_workspace_bytes(const lr_llama_t* h, int32_t max_tokens, int32_t max_seqs);

/* One prefill over B packed (unpadded) prompts + verbalizer gather at each prompt's last token.
 * Replaces LlamaForCausalLM.forward (patched, model/llm.py:35-145: logits[:, -1] in fp32) followed
 * by ManualVerbalizer.process_logits (trainer/verb.py:546-586 == logits[:, label_token_ids],
 * SURVEY.md 8(a) a17) -- only the C verbalizer rows of lm_head are evaluated.
 *   packed_ids     : DEVICE int32 [total_tokens]; prompt b occupies [cu_seqlens[b], cu_seqlens[b+1])
 *   cu_seqlens     : DEVICE int32 [B+1] AND the same values in host memory (cu_seqlens_host),
 *                    used only for launch geometry
 *   label_token_ids: DEVICE int32 [C]  (tokenizer.encode(chr(65+c)), trainer/verb.py:494)
 *   out_scores     : DEVICE fp32 [B][C]
 */
int lr_llama_prefill_verbalize(lr_llama_t* h, const int32_t* packed_ids, const int32_t* cu_seqlens,
                               const int32_t* cu_seqlens_host, int32_t B,
                               const int32_t* label_token_ids, int32_t C, float* out_scores,
                               void* workspace, size_t workspace_bytes, void* hip_stream);

/* The same call when every prompt of the batch starts with the same prefix_len tokens -- the template text in front
 * of the first history item (dataloader/utils.py:24-40, templates/alpaca_short.json:3; about 36 Llama-2 tokens) --
 * which is then evaluated ONCE: internally the batch is laid out as [prefix][rest of prompt 0]...[rest of prompt B-1]
 * and the attention kernel reads keys/values of positions < prefix_len from the prefix rows. Inputs are the caller's
 * ordinary packed prompts (prefix included in each). Scores are BIT-IDENTICAL to lr_llama_prefill_verbalize.
 * Preconditions: 0 <= prefix_len < the shortest prompt; ids[cu[b] + i] == ids[cu[0] + i] for all b, i < prefix_len
 * (lr_common_prefix_len computes the largest such value from host copies). The second precondition is VERIFIED on the
 * device: if any prompt's first prefix_len ids differ from prompt 0's, every score of the call is NaN (prefix rows, keys
 * and values come from prompt 0, so a stale prefix_len would otherwise score the other prompts silently wrong).
 * prefix_len = 0, B = 1, head_dim != 128 or the generic attention variant run the plain path. */
int lr_llama_prefill_verbalize_prefix(lr_llama_t* h, const int32_t* packed_ids, const int32_t* cu_seqlens,
                                      const int32_t* cu_seqlens_host, int32_t B, int32_t prefix_len,
                                      const int32_t* label_token_ids, int32_t C, float* out_scores,
                                      void* workspace, size_t workspace_bytes, void* hip_stream);
/* Host helper (pure CPU): longest token prefix shared by all B prompts, capped at (shortest prompt - 1). */
int32_t lr_common_prefix_len(const int32_t* packed_ids_host, const int32_t* cu_seqlens_host, int32_t B);

/* Compatibility: full last-position logits fp32 [B][vocab] (model/llm.py:131). */
int lr_llama_last_logits(lr_llama_t* h, const int32_t* packed_ids, const int32_t* cu_seqlens,
                         const int32_t* cu_seqlens_host, int32_t B, float* out_logits,
                         void* workspace, size_t workspace_bytes, void* hip_stream);

/* Host helper: stack HF q_proj/k_proj/v_proj ([nh*hd][hidden], [nkv*hd][hidden] x2, bf16) into the
 * wqkv layout above (pair-interleaved q/k head rows). Pure CPU. */
int lr_llama_pack_qkv(const uint16_t* q, const uint16_t* k, const uint16_t* v, int32_t num_heads,
                      int32_t num_kv_heads, int32_t head_dim, int32_t hidden, uint16_t* out);

/* Host helper: interleave gate_proj / up_proj rows ([inter][hidden] each, bf16) into the wgu
 * layout above. Pure CPU. */
int lr_llama_pack_gate_up(const uint16_t* gate, const uint16_t* up, int32_t inter, int32_t hidden,
                          uint16_t* out);

/* Load-time NF4 round trip  W -> dequant(quant(W))  of a frozen bf16 weight matrix (DEVICE pointers, n elements,
 * row-major blocks of 64): what the reference's forward effectively multiplies by, since it quantises every Linear
 * of the base model with BitsAndBytesConfig(load_in_4bit, nf4, double quant, bf16 compute) (train_ranker.py:49-56,
 * setup_demo.py:67-74; bitsandbytes 0.43.1 is absent here, its published algorithm is restated -- parity with it is
 * unpinned). double_quant != 0 also quantises the per-block absmax (8-bit dynamic code, blocks of 256, mean offset).
 * out may alias w. Synchronises the stream once when double_quant is set (load-time call).
 * lr_nf4_dynamic_map: the 256 sorted levels of that 8-bit code (HOST float[256]). */
size_t lr_nf4_scratch_bytes(size_t n);
int lr_nf4_roundtrip_bf16(const uint16_t* w, size_t n, int32_t double_quant, uint16_t* out, void* scratch,
                          size_t scratch_bytes, void* hip_stream);
int lr_nf4_dynamic_map(float* out256);

/* Stand-alone bf16 GEMM used by the prefill (exposed for parity tests and roofline runs):
 * C[M][N] = A[M][K] * B[N][K]^T, bf16 in, fp32 accumulate, bf16 out; all DEVICE pointers,
 * row-major, leading dimensions = K, K, N. variant: 0 = auto, 1 = generic (any shape),
 * 4 = 256x256x64 MFMA tile, ping-pong pipeline with balanced fragment reads and region-recycling DMA prefetch
 * (default for N%256==0, K%64==0, M>=128; any M: rows are bounds-checked). */
int lr_gemm_bf16_nt(const uint16_t* A, const uint16_t* B, uint16_t* C, int32_t M, int32_t N,
                    int32_t K, int32_t variant, void* hip_stream);
/* The same with a device workspace for variant 5 (split-K, see lr_llama_set_variants): fp32 partial
 * planes, at most 64 MiB (8 splits x M x N x 4 bytes, splits x tiles <= 256). */
int lr_gemm_bf16_nt_ws(const uint16_t* A, const uint16_t* B, uint16_t* C, int32_t M, int32_t N,
                       int32_t K, int32_t variant, void* workspace, size_t workspace_bytes, void* hip_stream);

/* The projection GEMM with each of the prefill's fused epilogues (exposed for parity tests and per-shape timing):
 *   epilogue 0 store | 1 residual: C = bf16(bf16(acc) + R), R bf16 [M][N], may alias C | 2 SwiGLU over gate/up rows
 *   interleaved in groups of 16 (lr_llama_pack_gate_up), C is [M][N/2] | 3 rotary embedding on columns
 *   [0, rot_cols) of pair-interleaved q/k rows (lr_llama_pack_qkv), tok_pos int32 [M] (every entry < rope_positions),
 *   rope_cs / rope_positions = the buffer lr_rope_table filled and its max_positions.
 * These are the epilogues of HF LlamaAttention / LlamaMLP around the Linears of model/llm.py:89-100 (reference). */
int lr_gemm_bf16_nt_epi(const uint16_t* A, const uint16_t* B, uint16_t* C, const uint16_t* R, int32_t M, int32_t N,
                        int32_t K, int32_t epilogue, int32_t variant, const int32_t* tok_pos, const float* rope_cs,
                        int32_t rope_positions, int32_t head_dim, int32_t rot_cols, void* workspace, size_t workspace_bytes,
                        void* hip_stream);
/* o_proj / down_proj together with the RMSNorm that reads their result (exposed for parity tests):
 *   C = bf16(bf16(A B^T) + R),   norm_out = bf16(norm_w * bf16(C * rsqrt(mean(C^2) + eps)))     (C may alias R)
 * fuse != 0 lets a product that variant 5 splits over K write norm_out in its reduce pass (one launch less per product in
 * the online path); *was_fused (optional) reports whether that happened. Both outputs carry the same bits either way. */
int lr_gemm_bf16_nt_residual_rmsnorm(const uint16_t* A, const uint16_t* B, uint16_t* C, const uint16_t* R, int32_t M,
                                     int32_t N, int32_t K, int32_t variant, const uint16_t* norm_w, uint16_t* norm_out,
                                     float eps, int32_t fuse, int32_t* was_fused, void* workspace, size_t workspace_bytes,
                                     void* hip_stream);
/* cs: DEVICE buffer of lr_rope_table_bytes(max_positions, head_dim) bytes: fp32 [max_positions][head_dim/2][2] = (cos, sin)
 * of position * theta^(-2i/head_dim), rounded to bf16 values (HF LlamaRotaryEmbedding casts cos/sin to the activations'
 * dtype), followed by the same values packed as bf16 pairs, uint32 [max_positions][head_dim/2] = cos | sin << 16. */
size_t lr_rope_table_bytes(int32_t max_positions, int32_t head_dim);
int lr_rope_table(float* cs, int32_t max_positions, int32_t head_dim, float theta, void* hip_stream);

/* Stand-alone varlen causal attention (exposed for parity tests):
 * qkv: DEVICE bf16 [total][(nh+2*nkv)*hd] (RoPE already applied; any consistent permutation of the
 * dims inside q and k heads), out: bf16 [total][nh*hd]. */
int lr_attention_varlen(const uint16_t* qkv, uint16_t* out, const int32_t* cu_seqlens,
                        const int32_t* cu_seqlens_host, int32_t B, int32_t num_heads,
                        int32_t num_kv_heads, int32_t head_dim, int32_t variant, void* hip_stream);
/* The same with a DEVICE workspace of lr_attention_workspace_bytes(total, B, num_heads) bytes, which variant 3 (and
 * auto, for head_dim 128) needs for its work-item list; lse: optional DEVICE fp32 [total][num_heads] log-sum-exp of
 * the scaled scores (NULL: not written). */
size_t lr_attention_workspace_bytes(int32_t total_tokens, int32_t B, int32_t num_heads);
int lr_attention_varlen_ws(const uint16_t* qkv, uint16_t* out, float* lse, const int32_t* cu_seqlens,
                           const int32_t* cu_seqlens_host, int32_t B, int32_t num_heads, int32_t num_kv_heads,
                           int32_t head_dim, int32_t variant, void* workspace, size_t workspace_bytes, void* hip_stream);

/* ------------------------------------------------------------------------------------------
 * Optional in-library kernel timing (HIP events on the caller's stream). Not part of the
 * reference's surface: it exists so a benchmark can report the measured duration and the
 * algorithmic work of each kernel family over its timed region.
 *   kinds: 0 gemm 256x256x64 (work = flops), 1 generic gemm (flops), 2 MFMA attention (flops),
 *          3 generic attention (flops), 4 LRU encoder (users), 5 item GEMM + top-K (flops)
 * lr_profile_start/stop bracket the region (start allocates events: call it outside graph
 * capture); lr_profile_collect is valid after the stream has been synchronised.
 * ------------------------------------------------------------------------------------------ */
int lr_profile_start(int32_t max_records);
int lr_profile_stop(void);
int lr_profile_collect(int32_t kind, double* total_ms, double* total_work, int64_t* launches);
/* Per-launch records of one kind, in launch order: ms[i], work[i], tag[i] for i < min(return value, max_records);
 * returns the number of records of that kind (-1 on error). GEMM tags: (epilogue << 56) | (N << 28) | K, so a caller
 * can report each projection shape separately. Pass max_records = 0 (arrays may be NULL) to count. */
int64_t lr_profile_records(int32_t kind, double* ms, double* work, int64_t* tag, int64_t max_records);

/* ===========================================================================================
 * Retriever training step
 *   replaces  LRUTrainer.calculate_loss                      trainer/lru.py:20-28
 *             loss.backward() through model/lru.py:38-175     (torch autograd)
 *             clip_gradients + AdamW.step                     trainer/base.py:106-112,201-246
 * One flat fp32 buffer holds every parameter in the reference's state_dict layouts (complex tensors as
 * interleaved (re, im) pairs = torch.view_as_real); gradients, and Adam's two moments have the same
 * layout, so a data-parallel job all-reduces ONE buffer between lr_lru_train_loss_grad and
 * lr_lru_train_apply. The caller owns the state buffer and the workspace (device memory).
 * =========================================================================================== */
typedef struct {
  float weight_decay;   /* 1e-2  config.py:123-124 (applied to names without "bias"/"layer_norm", trainer/base.py:222) */
  float beta1, beta2;   /* 0.9, 0.999 (torch.optim.AdamW defaults) */
  float eps;            /* 1e-9  config.py:181 */
  float max_grad_norm;  /* 5.0   config.py:184 */
  float dropout;        /* bert_dropout      0.2 config.py:216 (embedding, FFN activation, FFN output) */
  float attn_dropout;   /* bert_attn_dropout 0.2 config.py:217 (LRU layer output) */
  uint64_t seed;        /* dropout stream (own counter-based generator: torch's masks cannot be reproduced) */
  int32_t ce_mode;      /* item GEMM + cross-entropy: 0 = auto, 1 = store the [B*L][V+1] logits (<= 256 MB problems),
                           2 = fused, logits never stored (large catalogs) */
} LrLruTrainConfig;

typedef struct lr_lru_train lr_lru_train_t;

/* Bytes of DEVICE memory for parameters + gradients + Adam moments (+ a decay mask and scalars). */
size_t lr_lru_train_state_bytes(int32_t num_items, int32_t num_blocks);
/* init: HOST fp32 arrays in the reference's layouts (the same descriptor lr_lru_pack takes). */
int lr_lru_train_create(const LrLruWeightsDesc* init, const LrLruTrainConfig* cfg, void* state_dev,
                        size_t state_bytes, lr_lru_train_t** out);
void lr_lru_train_destroy(lr_lru_train_t* h);
size_t lr_lru_train_workspace_bytes(const lr_lru_train_t* h, int32_t B, int32_t L);
/* Forward + backward of one batch: tokens/labels DEVICE int64 [B][L], left-padded with 0, label 0 =
 * ignored (dataloader/lru.py:119-131). Fills the gradient buffer (zeroed first) and
 * out_loss[0] = mean CE over labelled positions, out_loss[1] = their count, out_loss[2] = number of labels
 * outside [0, num_items] (ignored like label 0; torch would raise) -- DEVICE float[3]. */
int lr_lru_train_loss_grad(lr_lru_train_t* h, const int64_t* tokens, const int64_t* labels, int32_t B,
                           int32_t L, float* out_loss, void* workspace, size_t workspace_bytes,
                           void* hip_stream);
/* clip_grad_norm_(max_grad_norm) + one AdamW step at learning rate lr (schedulers live with the caller;
 * max_grad_norm <= 0 = the configured limit); out_grad_norm (DEVICE float, may be null) receives the
 * pre-clipping global norm. */
int lr_lru_train_apply(lr_lru_train_t* h, float lr, float max_grad_norm, float* out_grad_norm,
                       void* hip_stream);
/* hipGraph replay: with enable != 0, lr_lru_train_loss_grad and lr_lru_train_apply capture their launch
 * sequence once per (pointers, B, L) on the caller's stream -- which must not be the default stream -- and
 * replay it afterwards (one graph launch instead of ~120 kernel launches per step). Pass the same device
 * buffers every step to stay on the replay path; any change re-captures. */
int lr_lru_train_set_graph(lr_lru_train_t* h, int32_t enable);
/* The LRU blocks of the step run as row-panel kernels (a 16-row panel of activations in LDS through every product of a
 * block: 14 launches for two blocks instead of 42). enable = 0 selects the first form, one generic GEMM launch per
 * product -- same mathematics, other summation order (the cross-check of tests/test_gpu_lru_train.py). Default 1. */
int lr_lru_train_set_fused(lr_lru_train_t* h, int32_t enable);
/* Deterministic mode (replaces nothing in the reference: torch's own scatter / index_add backward is atomic too,
 * trainer/lru.py:20-28; asked for so that two runs of a step can be compared bit for bit). With enable != 0 every fp32 atomic of
 * the pass -- the loss sum, d x behind the item GEMM, every parameter gradient summed over rows -- adds a 64-bit fixed-point
 * number into a shadow buffer instead (integer addition commutes; csrc/lr_det.h) and the pass folds the shadows back at fixed
 * points; the gradient norm of lr_lru_train_apply is summed by one workgroup. Two passes over the same batch then give the same
 * bits. Needs the row-panel kernels (lr_lru_train_set_fused(h, 1), the default) and a larger workspace: ask
 * lr_lru_train_workspace_bytes AFTER enabling it. One deterministic engine per process at a time. Default 0. */
int lr_lru_train_set_deterministic(lr_lru_train_t* h, int32_t enable);
/* Device pointers of the flat parameter / gradient buffers and their length in floats. */
int lr_lru_train_buffers(lr_lru_train_t* h, float** params, float** grads, size_t* count);
/* Offset and length (floats) of a parameter inside those buffers, by its reference state_dict name,
 * e.g. "model.lru_blocks.1.lru_layer.in_proj.weight" (complex: 2 floats per element). */
int lr_lru_train_param_range(const lr_lru_train_t* h, const char* name, size_t* offset, size_t* count);

/* ------------------------------------------------------------------------------------------
 * Ranker LoRA fine-tuning step.
 * Replaces the HF Trainer loop of trainer/llm.py:103-136 over the patched LlamaForCausalLM
 * (model/llm.py:89-127: shifted CrossEntropyLoss on the tokens whose label is not -100; the
 * reference labels only the answer letter and EOS, dataloader/llm.py:55-58) with peft LoRA on
 * q_proj / v_proj (train_ranker.py:71-79, config.py:257-260: r 8, alpha 32, dropout 0.05).
 * Deviations, both documented in DESIGN.md: the frozen base is bf16 (the reference's is NF4 through
 * bitsandbytes, absent here), and the optimizer is plain fp32 AdamW with HF's defaults (the
 * reference's "adamw_bnb_8bit" keeps block-quantised moments).
 *
 * The base handle must hold the UNMERGED base weights. The data-gradient GEMMs run on transposed
 * copies of the frozen matrices (same packed row / column orders), owned by the caller and made
 * once with lr_transpose_bf16.
 * ------------------------------------------------------------------------------------------ */
typedef struct LrLlamaLayerWeightsT {
  const uint16_t* wqkv_t;   /* [hidden][(nh+2*nkv)*hd]  = wqkv^T  */
  const uint16_t* wo_t;     /* [nh*hd][hidden]          = wo^T    */
  const uint16_t* wgu_t;    /* [hidden][2*inter]        = wgu^T   */
  const uint16_t* wdown_t;  /* [inter][hidden]          = wdown^T */
} LrLlamaLayerWeightsT;

typedef struct LrLlamaWeightsTDesc {
  const LrLlamaLayerWeightsT* layers; /* HOST array [num_layers] of device pointers */
  const uint16_t* lm_head_t;          /* [hidden][vocab] = lm_head^T */
} LrLlamaWeightsTDesc;

typedef struct LrLoraTrainConfig {
  int32_t r;            /* config.py:257, 1..16 */
  float alpha;          /* config.py:258; scaling = alpha / r */
  float dropout;        /* config.py:259, on the adapters' input only */
  float beta1, beta2, eps, weight_decay; /* HF TrainingArguments defaults: 0.9, 0.999, 1e-8, 0 */
  uint64_t seed;        /* dropout stream (own counter-based generator) */
} LrLoraTrainConfig;

typedef struct lr_llama_lora lr_llama_lora_t;

/* dst[c][r] = src[r][c], bf16, DEVICE pointers. */
int lr_transpose_bf16(const uint16_t* src, int32_t rows, int32_t cols, uint16_t* dst, void* hip_stream);

/* DEVICE bytes for the adapters: fp32 parameters, gradients, Adam moments, bf16 working copies. */
size_t lr_llama_lora_state_bytes(const lr_llama_t* base, const LrLoraTrainConfig* cfg);
/* Gradients, moments and the step counter start at zero; the PARAMETERS are the caller's to fill
 * through lr_llama_lora_buffers (peft: A ~ kaiming-uniform, B = 0). `base` must outlive the handle. */
int lr_llama_lora_create(lr_llama_t* base, const LrLlamaWeightsTDesc* wt, const LrLoraTrainConfig* cfg,
                         void* state_dev, size_t state_bytes, void* hip_stream, lr_llama_lora_t** out);
void lr_llama_lora_destroy(lr_llama_lora_t* h);
/* Flat fp32 DEVICE buffers of n floats each. Per layer: q_proj.lora_A [r][hidden], q_proj.lora_B
 * [nh*hd][r], v_proj.lora_A [r][hidden], v_proj.lora_B [nkv*hd][r] -- peft's layouts and HF's
 * (unpermuted) row order. Data-parallel training all-reduces `grads` between loss_grad and apply. */
int lr_llama_lora_buffers(lr_llama_lora_t* h, float** params, float** grads, float** m, float** v, size_t* n);
/* which: 0 q_proj, 1 v_proj; ab: 0 lora_A, 1 lora_B. */
int lr_llama_lora_param_range(const lr_llama_lora_t* h, int32_t layer, int32_t which, int32_t ab,
                              size_t* offset, size_t* count);
size_t lr_llama_lora_workspace_bytes(const lr_llama_lora_t* h, int32_t max_tokens, int32_t max_seqs,
                                     int32_t max_loss_rows);
/* One micro-batch: forward over the packed prompts (as lr_llama_prefill_verbalize packs them), loss
 * = mean over the m labelled rows of -log softmax(logits[loss_rows[i]])[loss_targets[i]] (row p of a
 * prompt predicts token p+1: the caller passes the rows whose NEXT token is labelled, each once),
 * backward, gradients * grad_scale ADDED to the gradient buffer (accumulate != 0) or written over it.
 * grad_scale = 1 / gradient_accumulation_steps (HF Trainer). out: DEVICE float[3] = {loss, m,
 * targets outside the vocabulary (ignored, should be 0)}. All pointers DEVICE except cu_seqlens_host. */
int lr_llama_lora_loss_grad(lr_llama_lora_t* h, const int32_t* packed_ids, const int32_t* cu_seqlens,
                            const int32_t* cu_seqlens_host, int32_t B, const int32_t* loss_rows,
                            const int32_t* loss_targets, int32_t m, float grad_scale, int32_t accumulate,
                            float* out, void* workspace, size_t workspace_bytes, void* hip_stream);
/* clip_grad_norm_(max_grad_norm; <= 0: none) + one AdamW step at learning rate lr. out_norm
 * (DEVICE float, optional) = gradient norm before clipping. */
int lr_llama_lora_apply(lr_llama_lora_t* h, float lr, float max_grad_norm, float* out_norm, void* hip_stream);
/* lr_llama_prefill_verbalize with the adapters as they are now (validation during training). */
size_t lr_llama_lora_eval_workspace_bytes(const lr_llama_lora_t* h, int32_t max_tokens, int32_t max_seqs);
int lr_llama_lora_prefill_verbalize(lr_llama_lora_t* h, const int32_t* packed_ids, const int32_t* cu_seqlens,
                                    const int32_t* cu_seqlens_host, int32_t B, const int32_t* label_token_ids,
                                    int32_t C, float* out_scores, void* workspace, size_t workspace_bytes,
                                    void* hip_stream);
/* Varlen causal attention backward (exposed for parity tests): qkv as lr_attention_varlen; out / d_out
 * bf16 [total][nh*hd]; lse fp32 [total][nh] from lr_attention_varlen_lse; dqkv bf16 like qkv.
 * scratch: DEVICE, lr_attention_bwd_scratch_bytes. */
int lr_attention_varlen_lse(const uint16_t* qkv, uint16_t* out, float* lse, const int32_t* cu_seqlens,
                            const int32_t* cu_seqlens_host, int32_t B, int32_t num_heads, int32_t num_kv_heads,
                            int32_t head_dim, int32_t variant, void* hip_stream);
size_t lr_attention_bwd_scratch_bytes(int32_t total, int32_t num_heads, int32_t num_kv_heads, int32_t head_dim);
int lr_attention_varlen_bwd(const uint16_t* qkv, const uint16_t* out, const uint16_t* d_out, const float* lse,
                            uint16_t* dqkv, const int32_t* cu_seqlens, const int32_t* cu_seqlens_host, int32_t B,
                            int32_t num_heads, int32_t num_kv_heads, int32_t head_dim, int32_t variant,
                            void* scratch, size_t scratch_bytes, void* hip_stream);

#ifdef __cplusplus
}
#endif
#endif /* LLAMAREC_MI355X_H */
