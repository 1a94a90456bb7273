// api_lru.hip -- C ABI entry points for stage 1 (declared in include/llamarec_mi355x.h) plus the
// library-wide error plumbing.
#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "lr_common.h"

static thread_local char g_err[512] = "";

void lr_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof g_err, fmt, ap);
  va_end(ap);
}

extern "C" const char* lr_last_error(void) { return g_err; }
extern "C" const char* lr_version(void) { return "llamarec_mi355x 0.1.0 gfx950"; }

extern "C" size_t lr_lru_packed_bytes(int32_t num_items, int32_t num_blocks) {
  if (num_items < 0 || num_blocks < 1 || num_blocks > LR_MAX_LRU_BLOCKS) return 0;
  return lr_lru_layout(num_items, num_blocks).total_floats * sizeof(float);
}

extern "C" int lr_lru_pack(const LrLruWeightsDesc* d, void* host_out, size_t host_out_bytes) {
  if (!d || !host_out) LR_FAIL(LR_EINVAL, "lr_lru_pack: null argument");
  if (d->hidden != LR_D) LR_FAIL(LR_EUNSUPPORTED, "lr_lru_pack: hidden=%d, only 64 is implemented", d->hidden);
  if (d->num_blocks < 1 || d->num_blocks > LR_MAX_LRU_BLOCKS || d->num_items < 0)
    LR_FAIL(LR_EINVAL, "lr_lru_pack: num_blocks=%d num_items=%d", d->num_blocks, d->num_items);
  LrLruLayout L = lr_lru_layout(d->num_items, d->num_blocks);
  if (host_out_bytes < L.total_floats * sizeof(float))
    LR_FAIL(LR_EINVAL, "lr_lru_pack: output buffer %zu < %zu bytes", host_out_bytes, L.total_floats * sizeof(float));
  float* o = (float*)host_out;
  memset(o, 0, L.total_floats * sizeof(float));
  const size_t rows = (size_t)d->num_items + 1;
  memcpy(o + L.item_emb, d->item_emb, rows * 64 * sizeof(float));
  memcpy(o + L.item_bias, d->item_bias, rows * sizeof(float));
  // The table is padded to whole 32-row tiles. A padding row is not an item: its bias is NaN, so its "score" fails every
  // `score >= threshold` test of the top-K kernels without a row-bound check per element (item_scores_mfma_kernel
  // never stores rows >= V + 1; fmaxf in the bound pre-pass ignores NaN).
  for (size_t i = rows; i < (size_t)L.rows_padded; ++i) o[L.item_bias + i] = NAN;
  {  // bf16 copy of the table (RNE) in MFMA A-FRAGMENT order and the two norms the top-K bound pre-pass needs (lru_topk.hip,
     // item_bound_kernel): [tile][step s][lane][j] = bf16(E[32 tile + (lane & 31)][32 (lane >> 5) + 8 s + j]), so that a
     // wave's fragment load of one MFMA step is 1 KiB of contiguous memory
    uint16_t* e16 = reinterpret_cast<uint16_t*>(o + L.item_emb_bf16);
    double emax = 0.0, bmax = 0.0;
    for (size_t i = 0; i < rows; ++i) {
      double ss = 0.0;
      for (int k = 0; k < 64; ++k) {
        const float v = d->item_emb[i * 64 + k];
        ss += (double)v * (double)v;
        uint32_t u;
        memcpy(&u, &v, 4);
        const uint16_t b = (v != v) ? (uint16_t)0x7fc0 : (uint16_t)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
        const size_t tile = i / 32, lane = (i % 32) + 32 * (size_t)(k / 32), step = (size_t)(k % 32) / 8, j = (size_t)k % 8;
        e16[((tile * 4 + step) * 64 + lane) * 8 + j] = b;
      }
      const double nrm = sqrt(ss), ab = fabs((double)d->item_bias[i]);
      if (nrm > emax || nrm != nrm) emax = nrm;
      if (ab > bmax || ab != ab) bmax = ab;
    }
    o[L.item_stats + 0] = nextafterf((float)emax, INFINITY);
    o[L.item_stats + 1] = nextafterf((float)bmax, INFINITY);
    o[L.item_stats + 2] = LR_LRU_IMAGE_FORMAT;   // what item_stats[32..63] mean depends on it: lr_lru_create rejects another value
    for (size_t r = 0; r < 32; ++r) {   // the last tile's accumulator start values for the bf16 passes (lru_topk.hip, tk_stage_issue)
      const size_t i = (size_t)L.rows_padded - 32 + r;
      o[L.item_stats + 32 + r] = i < rows ? d->item_bias[i] : -INFINITY;
    }
  }
  memcpy(o + L.emb_ln_w, d->emb_ln_w, 64 * sizeof(float));
  memcpy(o + L.emb_ln_b, d->emb_ln_b, 64 * sizeof(float));
  for (int b = 0; b < d->num_blocks; ++b) {
    const LrLruBlockWeights& w = d->blocks[b];
    const LrLruBlockLayout& B = L.blk[b];
    lr_lru_derive(w.params_log, o + B.lam_re, o + B.lam_im, o + B.gamma);
    for (int c = 0; c < LR_H; ++c) {
      for (int k = 0; k < 64; ++k) {
        o[B.in_wt + (size_t)k * 256 + c] = w.in_proj_w[((size_t)c * 64 + k) * 2];
        o[B.in_wt + (size_t)k * 256 + 128 + c] = w.in_proj_w[((size_t)c * 64 + k) * 2 + 1];
      }
      o[B.in_b + c] = w.in_proj_b[2 * c];
      o[B.in_b + 128 + c] = w.in_proj_b[2 * c + 1];
    }
    for (int oo = 0; oo < 64; ++oo) {
      for (int k = 0; k < LR_H; ++k) {
        o[B.out_wt + (size_t)k * 64 + oo] = w.out_proj_w[((size_t)oo * LR_H + k) * 2];
        o[B.out_wt + (size_t)(128 + k) * 64 + oo] = -w.out_proj_w[((size_t)oo * LR_H + k) * 2 + 1];
      }
      o[B.out_b + oo] = w.out_proj_b[2 * oo];
    }
    memcpy(o + B.ln1_w, w.ln1_w, 64 * sizeof(float));
    memcpy(o + B.ln1_b, w.ln1_b, 64 * sizeof(float));
    for (int j = 0; j < LR_FF; ++j)
      for (int k = 0; k < 64; ++k) o[B.w1t + (size_t)k * 256 + j] = w.ffn_w1[(size_t)j * 64 + k];
    memcpy(o + B.b1, w.ffn_b1, LR_FF * sizeof(float));
    for (int oo = 0; oo < 64; ++oo)
      for (int k = 0; k < LR_FF; ++k) o[B.w2t + (size_t)k * 64 + oo] = w.ffn_w2[(size_t)oo * LR_FF + k];
    memcpy(o + B.b2, w.ffn_b2, 64 * sizeof(float));
    memcpy(o + B.ln2_w, w.ln2_w, 64 * sizeof(float));
    memcpy(o + B.ln2_b, w.ln2_b, 64 * sizeof(float));
  }
  return LR_OK;
}

extern "C" int lr_lru_create(const void* packed_dev, size_t packed_bytes, int32_t num_items,
                             int32_t num_blocks, lr_lru_t** out) {
  if (!packed_dev || !out) LR_FAIL(LR_EINVAL, "lr_lru_create: null argument");
  size_t need = lr_lru_packed_bytes(num_items, num_blocks);
  if (need == 0 || packed_bytes < need)
    LR_FAIL(LR_EINVAL, "lr_lru_create: image is %zu bytes, layout needs %zu", packed_bytes, need);
  hipPointerAttribute_t attr;
  LR_CHECK_HIP(hipPointerGetAttributes(&attr, packed_dev));
  if (attr.type != hipMemoryTypeDevice)
    LR_FAIL(LR_EINVAL, "lr_lru_create: packed image must live in device memory");
  // The image's meaning changed without its size changing (round 4: item_stats[32..63] became the last tile's accumulator start
  // values, -inf on padding rows; an older image has zeros there and would score padding rows 0 -- silently wrong top-K). A
  // caller that caches images across library versions is told so instead.
  const LrLruLayout lay = lr_lru_layout(num_items, num_blocks);
  float tag = 0.f;
  LR_CHECK_HIP(hipMemcpy(&tag, (const float*)packed_dev + lay.item_stats + 2, sizeof(float), hipMemcpyDeviceToHost));
  if (tag != LR_LRU_IMAGE_FORMAT)
    LR_FAIL(LR_EINVAL, "lr_lru_create: packed image has format tag %g, this library reads %g: repack it with lr_lru_pack", (double)tag,
            (double)LR_LRU_IMAGE_FORMAT);
  lr_lru* h = (lr_lru*)calloc(1, sizeof(lr_lru));
  if (!h) LR_FAIL(LR_EINVAL, "lr_lru_create: out of host memory");
  h->img = (const float*)packed_dev;
  h->lay = lay;
  h->device = attr.device;
  h->encoder_pipeline = 1;
  *out = h;
  return LR_OK;
}

extern "C" void lr_lru_destroy(lr_lru_t* h) { free(h); }

extern "C" int lr_lru_set_encoder_pipeline(lr_lru_t* h, int32_t enable) {
  if (!h) LR_FAIL(LR_EINVAL, "lr_lru_set_encoder_pipeline: null handle");
  h->encoder_pipeline = enable ? 1 : 0;
  return LR_OK;
}

static size_t q_bytes(int B) { return lr_align_up((size_t)B * 64 * sizeof(float), 256); }

// Encoder dispatch: the batched MFMA encoder when the caller's workspace has room for it (what
// lr_lru_workspace_bytes asks for), else the one-workgroup-per-user kernel (no workspace); LR_ENCODER=1 forces
// the latter for A/B runs. Both produce the same bits.
static int encode(lr_lru_t* h, const int64_t* ids, int B, int L, float* q, void* ws, size_t ws_bytes, hipStream_t st) {
  static int forced = -1;
  if (forced < 0) {
    const char* e = getenv("LR_ENCODER");
    forced = (e && e[0] == '1') ? 1 : 0;
  }
  if (!forced && ws && ws_bytes >= lr_encoder_mfma_workspace_bytes(B, L))
    return lr_launch_lru_encode_mfma(h, ids, B, L, q, ws, ws_bytes, st);
  return lr_launch_lru_encode(h, ids, B, L, q, st);
}

extern "C" size_t lr_lru_workspace_bytes(const lr_lru_t* h, int32_t max_users, int32_t max_k, int32_t max_len) {
  if (!h) {  // the top-K scratch depends on the catalog's tile count: no handle, no size (0 = error, like the other *_bytes)
    lr_set_error("lr_lru_workspace_bytes: the model handle is null");
    return 0;
  }
  if (max_users < 1) max_users = 1;
  if (max_k < 1) max_k = 1;
  if (max_len < 1) max_len = 1;
  // [q | encoder scratch or top-K scratch (never live together)]
  const size_t enc = lr_encoder_mfma_workspace_bytes(max_users, max_len),
               tk = lr_topk_workspace_bytes(max_users, max_k, max_len, h->lay.rows_padded / LR_ITEM_TILE);
  return q_bytes(max_users) + (enc > tk ? enc : tk);
}

static int check_ids(const char* fn, const lr_lru_t* h, const void* ids, int B, int L) {
  if (!h || !ids) LR_FAIL(LR_EINVAL, "%s: null argument", fn);
  if (B < 0 || L < 1) LR_FAIL(LR_EINVAL, "%s: B=%d L=%d", fn, B, L);
  return LR_OK;
}

extern "C" int lr_lru_encode_last(lr_lru_t* h, const int64_t* ids, int32_t B, int32_t L, float* out_q,
                                  void* workspace, size_t workspace_bytes, void* hip_stream) {
  int rc = check_ids("lr_lru_encode_last", h, ids, B, L);
  if (rc) return rc;
  if (!out_q) LR_FAIL(LR_EINVAL, "lr_lru_encode_last: out_q is null");
  return encode(h, ids, B, L, out_q, workspace, workspace_bytes, (hipStream_t)hip_stream);
}

extern "C" int lr_lru_retrieve_topk(lr_lru_t* h, const int64_t* ids, int32_t B, int32_t L, int32_t K,
                                    int32_t exclude_history, int32_t* out_idx, float* out_score,
                                    void* workspace, size_t workspace_bytes, void* hip_stream) {
  int rc = check_ids("lr_lru_retrieve_topk", h, ids, B, L);
  if (rc) return rc;
  if (!out_idx || !workspace) LR_FAIL(LR_EINVAL, "lr_lru_retrieve_topk: null output/workspace");
  if (K < 1 || K > LR_MAX_TOPK) LR_FAIL(LR_EINVAL, "lr_lru_retrieve_topk: K=%d outside 1..%d", K, LR_MAX_TOPK);
  if (workspace_bytes < q_bytes(B)) LR_FAIL(LR_EWORKSPACE, "lr_lru_retrieve_topk: workspace too small");
  float* q = (float*)workspace;
  hipStream_t st = (hipStream_t)hip_stream;
  rc = encode(h, ids, B, L, q, (char*)workspace + q_bytes(B), workspace_bytes - q_bytes(B), st);
  if (rc) return rc;
  return lr_launch_item_topk(h, q, ids, B, L, K, exclude_history, out_idx, out_score,
                             (char*)workspace + q_bytes(B), workspace_bytes - q_bytes(B), st);
}

extern "C" int lr_lru_topk_path(const lr_lru_t* h, int32_t B, int32_t L, int32_t K, int32_t exclude_history,
                                const void* workspace, size_t workspace_bytes, int32_t* out_path, void* hip_stream) {
  if (!h || !workspace || !out_path) LR_FAIL(LR_EINVAL, "lr_lru_topk_path: null argument");
  if (workspace_bytes < q_bytes(B)) LR_FAIL(LR_EWORKSPACE, "lr_lru_topk_path: workspace too small");
  int path = 0;
  const int rc = lr_topk_path(h, B, K, L, exclude_history, (const char*)workspace + q_bytes(B), workspace_bytes - q_bytes(B), &path,
                              (hipStream_t)hip_stream);
  if (rc) return rc;
  *out_path = path;
  return LR_OK;
}

extern "C" int lr_lru_scores_last(lr_lru_t* h, const int64_t* ids, int32_t B, int32_t L,
                                  int32_t exclude_history, float* out_scores, void* workspace,
                                  size_t workspace_bytes, void* hip_stream) {
  int rc = check_ids("lr_lru_scores_last", h, ids, B, L);
  if (rc) return rc;
  if (!out_scores || !workspace) LR_FAIL(LR_EINVAL, "lr_lru_scores_last: null output/workspace");
  if (workspace_bytes < q_bytes(B)) LR_FAIL(LR_EWORKSPACE, "lr_lru_scores_last: workspace too small");
  float* q = (float*)workspace;
  hipStream_t st = (hipStream_t)hip_stream;
  rc = encode(h, ids, B, L, q, (char*)workspace + q_bytes(B), workspace_bytes - q_bytes(B), st);
  if (rc) return rc;
  return lr_launch_item_scores(h, q, ids, B, L, exclude_history, out_scores, st);
}
