// lr_common.h -- internal declarations shared by the translation units of libllamarec_mi355x.so.
#ifndef LR_COMMON_H
#define LR_COMMON_H

#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include "../../include/llamarec_mi355x.h"
#include "lr_math.h"

// ---- error plumbing (no C++ exception crosses the ABI) -------------------------------------
void lr_set_error(const char* fmt, ...);
#define LR_FAIL(code, ...)      \
  do {                          \
    lr_set_error(__VA_ARGS__);  \
    return (code);              \
  } while (0)
#define LR_CHECK_HIP(expr)                                                             \
  do {                                                                                 \
    hipError_t e_ = (expr);                                                            \
    if (e_ != hipSuccess) LR_FAIL(LR_EHIP, "%s: %s", #expr, hipGetErrorString(e_));    \
  } while (0)
#define LR_CHECK_LAUNCH(name)                                                          \
  do {                                                                                 \
    hipError_t e_ = hipGetLastError();                                                 \
    if (e_ != hipSuccess) LR_FAIL(LR_EHIP, "launch %s: %s", name, hipGetErrorString(e_)); \
  } while (0)

static inline size_t lr_align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

// hipFuncAttributeMaxDynamicSharedMemorySize is a per-DEVICE attribute of a kernel: a launcher keeps one `done`
// array per kernel and calls this before every launch (one hipGetDevice on the fast path). A racing first call from
// two host threads sets the attribute twice, which is harmless.
#define LR_MAX_DEVICES 64
static inline int lr_ensure_dynamic_lds(const void* kernel, int bytes, bool* done /*[LR_MAX_DEVICES]*/) {
  int dev = 0;
  LR_CHECK_HIP(hipGetDevice(&dev));
  const bool tracked = dev >= 0 && dev < LR_MAX_DEVICES;
  if (tracked && done[dev]) return LR_OK;
  LR_CHECK_HIP(hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
  if (tracked) done[dev] = true;
  return LR_OK;
}

// ---- packed LRURec device image (all float32; offsets in floats) ----------------------------
// Produced by lr_lru_pack() on the host, consumed by the kernels in lru_encoder.hip / lru_topk.hip.
#define LR_ITEM_TILE 32  // items per MFMA tile; the table is zero-padded to a multiple of this
#define LR_LRU_IMAGE_FORMAT 5.0f   // written to item_stats[2] by lr_lru_pack; bump when an array of the image changes meaning

struct LrLruBlockLayout {
  size_t lam_re, lam_im, gamma;  // [128] each
  size_t in_wt;                  // [64][256]  in_wt[k][j]: j<128 Re W_in[j][k], j>=128 Im W_in[j-128][k]
  size_t in_b;                   // [256]
  size_t out_wt;                 // [256][64]  k<128: Re W_out[o][k]; k>=128: -Im W_out[o][k-128]
  size_t out_b;                  // [64]       Re b_out
  size_t ln1_w, ln1_b;           // [64]
  size_t w1t;                    // [64][256]  W1[j][k] transposed
  size_t b1;                     // [256]
  size_t w2t;                    // [256][64]  W2[o][k] transposed
  size_t b2;                     // [64]
  size_t ln2_w, ln2_b;           // [64]
};

struct LrLruLayout {
  int32_t num_items;     // V
  int32_t rows_padded;   // ceil((V+1)/32)*32
  int32_t num_blocks;
  size_t item_emb;       // [rows_padded][64]
  size_t item_bias;      // [rows_padded]
  size_t item_emb_bf16;  // [rows_padded][64] bfloat16 (round-to-nearest-even copy of item_emb): the top-K bound pre-pass
  size_t item_stats;     // [0] >= max_i ||item_emb[i]||_2, [1] >= max_i |item_bias[i]|  (rounded up); [2] LR_LRU_IMAGE_FORMAT; [32..63] the biases of the
                         // table's LAST 32-row tile with -inf on its padding rows (item_bias has NaN there): the bf16 passes'
                         // accumulator start values, so that a padding row's approximate score is -inf and needs no row test
  size_t emb_ln_w, emb_ln_b;
  LrLruBlockLayout blk[LR_MAX_LRU_BLOCKS];
  size_t total_floats;
};

static inline LrLruLayout lr_lru_layout(int32_t num_items, int32_t num_blocks) {
  LrLruLayout L;
  L.num_items = num_items;
  L.rows_padded = (int32_t)lr_align_up((size_t)num_items + 1, LR_ITEM_TILE);
  L.num_blocks = num_blocks;
  size_t o = 0;
  auto take = [&](size_t n) {
    size_t at = o;
    o += lr_align_up(n, 64);  // keep every array 256-byte aligned
    return at;
  };
  L.item_emb = take((size_t)L.rows_padded * 64);
  L.item_bias = take((size_t)L.rows_padded);
  L.item_emb_bf16 = take((size_t)L.rows_padded * 32);
  L.item_stats = take(64);
  L.emb_ln_w = take(64);
  L.emb_ln_b = take(64);
  for (int b = 0; b < LR_MAX_LRU_BLOCKS; ++b) {
    LrLruBlockLayout& B = L.blk[b];
    if (b >= num_blocks) {
      B = LrLruBlockLayout{};
      continue;
    }
    B.lam_re = take(128);
    B.lam_im = take(128);
    B.gamma = take(128);
    B.in_wt = take(64 * 256);
    B.in_b = take(256);
    B.out_wt = take(256 * 64);
    B.out_b = take(64);
    B.ln1_w = take(64);
    B.ln1_b = take(64);
    B.w1t = take(64 * 256);
    B.b1 = take(256);
    B.w2t = take(256 * 64);
    B.b2 = take(64);
    B.ln2_w = take(64);
    B.ln2_b = take(64);
  }
  L.total_floats = o;
  return L;
}

struct lr_lru {
  const float* img;  // device image
  LrLruLayout lay;
  int device;
  int encoder_pipeline;  // 1 (default): the LRU layer on em_pipe_kernel; 0: on em_layer_kernel (lr_lru_set_encoder_pipeline)
};

// ---- stage-1 launchers (lru_encoder.hip, lru_topk.hip, metrics.hip) -------------------------
int lr_launch_lru_encode(const lr_lru* h, const int64_t* ids, int B, int L, float* out_q,
                         hipStream_t st);
// batched MFMA encoder (lru_encoder_mfma.hip): same result as lr_launch_lru_encode, needs a workspace
size_t lr_encoder_mfma_workspace_bytes(int B, int L);
int lr_launch_lru_encode_mfma(const lr_lru* h, const int64_t* ids, int B, int L, float* out_q, void* ws,
                              size_t ws_bytes, hipStream_t st);
size_t lr_topk_workspace_bytes(int B, int K, int L, int n_tiles);
int lr_topk_path(const lr_lru* h, int B, int K, int L, int exclude_history, const void* ws, size_t ws_bytes, int* out_path,
                 hipStream_t st);
int lr_launch_item_topk(const lr_lru* h, const float* q, const int64_t* ids, int B, int L, int K,
                        int exclude_history, int32_t* out_idx, float* out_score, void* ws,
                        size_t ws_bytes, hipStream_t st);
int lr_launch_item_scores(const lr_lru* h, const float* q, const int64_t* ids, int B, int L,
                          int exclude_history, float* out_scores, hipStream_t st);

// ---- stage-2 handle -------------------------------------------------------------------------
struct lr_llama {
  LrLlamaConfig cfg;
  const uint16_t* embed;
  const uint16_t* final_norm;
  const uint16_t* lm_head;
  LrLlamaLayerWeights* layers;  // host array
  int device;
  int gemm_variant;  // 0 auto, 1 generic, 4 256x256x64 MFMA tile, 5 = 4 + split-K (latency mode)
  int attn_variant;  // 0 auto, 1 generic, 2 MFMA head_dim 128
  int prune_last;    // last layer: attention output / o_proj / MLP only for each prompt's last token
  // folded RMSNorm (lr_llama_set_folded_norms): per layer wqkv * diag(input_norm) and wgu * diag(post_norm), or null
  const uint16_t** wqkv_folded;
  const uint16_t** wgu_folded;
};

#endif  // LR_COMMON_H
