// llama_kernels.h -- internal launcher declarations and bf16 helpers for the stage-2 kernels.
#ifndef LLAMA_KERNELS_H
#define LLAMA_KERNELS_H

#include "lr_common.h"

__device__ __forceinline__ float bf2f(unsigned short b) {
  return __builtin_bit_cast(float, (unsigned int)b << 16);
}
// round-to-nearest-even (v_cvt_pk_bf16_f32 on gfx950; NaN stays NaN)
__device__ __forceinline__ unsigned short f2bf(float f) {
  return __builtin_bit_cast(unsigned short, (__bf16)f);
}
// HF: down_proj(act_fn(gate) * up) with bf16 tensors: gate/up already bf16-valued floats here;
// silu output is rounded to bf16, then the product is rounded to bf16.
// The quotient is g * v_rcp_f32(1 + exp(-g)) (1 ulp, like the v_exp_f32 under __expf): the IEEE division sequence costs
// ten vector instructions per value, 128 values per lane in a gate-up tile's epilogue.
__device__ __forceinline__ unsigned short swiglu_bf16(float g, float u) {
  float s = bf2f(f2bf(g * __builtin_amdgcn_rcpf(1.0f + __expf(-g))));
  return f2bf(s * u);
}

// epilogue modes of the GEMMs
#define LR_EPI_STORE 0     // C = bf16(acc)
#define LR_EPI_RESIDUAL 1  // C = bf16( bf16(acc) + R )      (R may alias C)
#define LR_EPI_SWIGLU 2    // C[M][N/2] = swiglu over interleaved gate/up 16-column groups
#define LR_EPI_ROPE 3      // C = bf16(acc) with rotary embedding on the pair-interleaved q/k columns
#define LR_EPI_PARTIAL 4   // internal (split-K): fp32 partial sums to the workspace, real epilogue in the reduce pass
#define LR_SPLITK_WS_BYTES ((size_t)64 << 20)  // splits x tiles <= 256 tiles of 256 x 256 fp32

// packed-layout metadata (llama_elem.hip): prefix_len = P > 0 lays the batch out as [P shared prefix rows][rest of
// prompt 0]...[rest of prompt B-1]; seg_start (optional) receives the B + 1 (P = 0: B) + 1 segment starts, tok_src
// (optional) the index of every internal row in the caller's packed ids, last_rows (optional) each prompt's last row,
// last_pos (optional) the position of that token inside its prompt
int lr_launch_token_meta(const int32_t* cu, int B, int prefix_len, int32_t* seg_start, int32_t* tok_pos,
                         int32_t* tok_src, int32_t* last_rows, hipStream_t st, int32_t* last_pos = nullptr,
                         const int32_t* ids = nullptr /* with prefix_bad: verify the shared-prefix promise */,
                         int32_t* prefix_bad = nullptr /* device word: zeroed, then set if a prompt's first P ids differ */);
int lr_launch_gather_rows(const unsigned short* x, const int32_t* rows, int n_rows, int d, unsigned short* out,
                          hipStream_t st);
int lr_launch_attention_rows(const unsigned short* qkv, unsigned short* out, const int32_t* cu, int B,
                             const int32_t* q_rows, int n_rows, int nh, int nkv, int hd, hipStream_t st);
int lr_launch_embed(const int32_t* ids, const int32_t* tok_src /*nullptr: identity*/, const unsigned short* table,
                    int vocab, int d, unsigned short* out, int n, hipStream_t st);
int lr_launch_rmsnorm(const unsigned short* x, const unsigned short* w, unsigned short* out, int rows, int d,
                      float eps, const int32_t* row_map, hipStream_t st);
int lr_launch_rope_table(float* cs, int T, int hd, float theta, hipStream_t st,
                         unsigned* cs16 = nullptr /* [T][hd/2] (cos | sin << 16) as bf16 pairs, optional */);
int lr_launch_head(const unsigned short* x, const int32_t* rows /*[B]; nullptr: x holds one row per prompt*/,
                   const unsigned short* norm_w,
                   const unsigned short* lm_head, const int32_t* class_ids, int B, int C, int d, float eps,
                   float* out, int vocab, hipStream_t st,
                   const int32_t* poison = nullptr /* device word: non-zero -> every score of the call is NaN */);

// C[M][N] (+epilogue) = A[M][K] * B[N][K]^T. variant: 0 auto, 1 generic, 4 = 256x256x64 MFMA tile,
// 5 = variant 4 with split-K when the tiles alone would leave most CUs idle (needs splitk_ws).
int lr_launch_gemm(const unsigned short* A, const unsigned short* B, unsigned short* C,
                   const unsigned short* R, int M, int N, int K, int epi, int variant, hipStream_t st,
                   const int32_t* tok_pos = nullptr, const float* rope_cs = nullptr, int head_dim = 0,
                   int rot_cols = 0, float* splitk_ws = nullptr, size_t splitk_ws_bytes = 0,
                   const float* row_scale = nullptr /* rope / swiglu epilogues: accumulator row m times row_scale[m] */,
                   const unsigned* rope_cs16 = nullptr /* the rope table as packed bf16 pairs (lr_launch_rope_table): lets
                   the 256-tile kernel stage a tile's (cos, sin) rows through LDS instead of 262 KB of half-line loads */,
                   const unsigned short* then_norm_w = nullptr, unsigned short* then_norm_out = nullptr,
                   float then_norm_eps = 0.f, bool* then_norm_done = nullptr /* residual epilogue only: the caller runs
                   RMSNorm(C) with this weight into then_norm_out next. If the product is split over K, its reduce pass
                   does that too (same bits) and *then_norm_done is set; otherwise it is left false and the caller launches
                   lr_launch_rmsnorm itself */);
// split-K reduce (S fp32 planes of M x N) + residual + RMSNorm of the result in one pass (llama_elem.hip)
bool lr_reduce_residual_rmsnorm_fits(int N);
int lr_launch_reduce_residual_rmsnorm(const float* part, int S, unsigned short* C, const unsigned short* R, int M, int N,
                                      const unsigned short* norm_w, unsigned short* norm_out, float eps, hipStream_t st);
// rstd[m] = 1 / sqrt(mean(x[m][:]^2) + eps), fp32 (the statistic of HF's LlamaRMSNorm)
int lr_launch_rms_rstd(const unsigned short* x, float* rstd, int rows, int d, float eps, hipStream_t st);
// out[j][k] = bf16(w[j][k] * norm_w[k]): an RMSNorm weight folded into the following projection's [out][in] matrix
int lr_launch_fold_norm(const unsigned short* w, const unsigned short* norm_w, unsigned short* out, size_t rows, int cols,
                        hipStream_t st);

// varlen causal attention over packed qkv (RoPE applied). variant: 0 auto, 1 generic, 2 MFMA hd=128.
// cu / cu_host = segment starts; prefix_len > 0: segment 0 is the prefix the other segments continue (MFMA kernel only)
int lr_launch_attention_last(const unsigned short* kv, const unsigned short* q_last, unsigned short* out_last,
                             const int32_t* cu, const int32_t* cu_host, int S, int n_tok, int nh, int nkv, int hd,
                             hipStream_t st, int prefix_len);
int lr_launch_attention(const unsigned short* qkv, unsigned short* out, const int32_t* cu,
                        const int32_t* cu_host, const int32_t* tok_pos, const int32_t* tok_seq, int B,
                        int n_tok, int nh, int nkv, int hd, int variant, void* scratch, hipStream_t st,
                        int prefix_len = 0);

// attention variant 3 (llama_attn256.hip): 256-row query tiles, one wave per SIMD, persistent workgroups over a
// device-built item list. lr_launch_attn256_items builds the list for (cu, S, nh, prefix_len) into items_ws
// (lr_attn256_ws_bytes); any number of lr_launch_attention256 calls over the same segments may follow (one per layer).
// prefix_len must be <= 64 (lr_attention256_takes): only a tile's block 0 may hold shared-prefix keys.
size_t lr_attn256_ws_bytes(int n_tok, int S, int nh);
static inline bool lr_attention256_takes(int hd, int prefix_len) { return hd == 128 && prefix_len <= 64; }
int lr_launch_attn256_items(const int32_t* cu, int S, int n_tok, int nh, int prefix_len, void* items_ws, size_t ws_bytes,
                            hipStream_t st);
int lr_launch_attention256(const unsigned short* qkv, unsigned short* out, const int32_t* cu, const int32_t* cu_host, int S,
                           int n_tok, int nh, int nkv, int hd, float* lse /* optional [n_tok][nh] */, void* items_ws,
                           hipStream_t st, int prefix_len);

#endif
