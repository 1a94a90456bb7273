// lru_train_blocks.h -- the retriever training step's LRU blocks as row-panel kernels (lru_train_blocks.hip): every
// product of a block whose weight matrix is one of the four 64 x 256 matrices runs inside a kernel that keeps a 16-row
// panel of activations in LDS, instead of one generic GEMM launch per product with an elementwise launch between
// each pair (lru_train.hip's first form, kept behind lr_lru_train_set_fused(h, 0) as the cross-check).
//   forward of a block  (model/lru.py:117-175):  in_proj | recurrence | out_proj + dropout + residual + LN + FFN + LN
//   backward of a block:                         the row-local chain back to d h | recurrence | in_proj + weight gradients
#pragma once
#include "lr_common.h"

// dropout: counter-based hash -> keep mask, identical in forward and backward (site = which dropout, idx = element)
__device__ __forceinline__ float tr_drop_scale(unsigned long long seed, unsigned site, unsigned long long idx, float p) {
  if (p <= 0.f) return 1.f;
  unsigned long long x = seed ^ (0x9E3779B97F4A7C15ull * (site + 1)) ^ (idx * 0xD6E8FEB86659FD93ull);
  x ^= x >> 32;
  x *= 0xD6E8FEB86659FD93ull;
  x ^= x >> 32;
  x *= 0xD6E8FEB86659FD93ull;
  x ^= x >> 32;
  const float u = (float)(x >> 40) * (1.0f / 16777216.0f);
  return u < p ? 0.f : 1.0f / (1.0f - p);
}

// A block's four 64 x 256 weight matrices and their transposes in MFMA FRAGMENT ORDER ([n / 16][k / 16][lane][4], see
// tb_load_w), written once per pass by tr_prep_kernel (lru_train.hip) together with the derived weights. The forward
// products read wiF [256][64], woF [64][256], w1F [256][64], w2F [64][256]; the data-gradient products read the
// transposes wiT [64][256], woT [256][64], w1T [64][256], w2T [256][64] (they read W[n][k] along n).
struct TbTransposed {
  float *wiF, *woF, *w1F, *w2F;
  float *wiT, *woT, *w1T, *w2T;
};
// position (floats) of element (n, k) of a matrix with K columns in fragment order
__host__ __device__ __forceinline__ int tb_frag_pos(int n, int k, int K) {
  return (((n >> 4) * (K >> 4) + (k >> 4)) * 64 + ((k >> 2) & 3) * 16 + (n & 15)) * 4 + (k & 3);
}

// x = LN(dropout(E[id])) (saves xhat, rstd) and block 0's in_proj u = x wi^T + bi, one launch
struct TbEmbedInProj {
  const long long* ids;
  const float *E, *ln_w, *ln_b, *wi, *bi;   // wi: fragment order (TbTransposed::wiF)
  float *x, *xhat, *rstd, *u;
  int R, V;
  const unsigned long long* seed;
  float p_drop;
};
int tb_launch_embed_in_proj(const TbEmbedInProj& p, hipStream_t st);

struct TbBlockFwd {
  const float *h, *x;   // [R][256] the recurrence's output (Re | Im), [R][64] the block's input
  const float *wo, *bo, *ln1_w, *ln1_b, *w1, *b1, *w2, *b2, *ln2_w, *ln2_b;   // wo, w1, w2: fragment order (woF, w1F, w2F)
  float *y, *xhat1, *rstd1, *a, *g, *xout, *xhat2, *rstd2;   // saved for the backward pass
  const float *next_wi, *next_bi;   // the NEXT block's in_proj, wiF (null for the last block): next_u = xout next_wi^T + next_bi
  float* next_u;
  int R;
  const unsigned long long* seed;
  unsigned site0;        // dropout sites: site0 (after out_proj), site0 + 1 (after GELU), site0 + 2 (after W2)
  float p_attn, p_drop;
};
int tb_launch_block_fwd(const TbBlockFwd& p, hipStream_t st);

struct TbBlockBwd {
  float* dx;                        // [R][64] in: gradient of the block's output; out: gradient reaching the block's input
                                    //         through the residual path (tb_launch_bwd_tail adds the recurrence's share)
  const float *xhat2, *rstd2, *ln2_w, *a, *xhat1, *rstd1, *ln1_w;
  const float *w2T, *w1T, *woT;
  float *dz0, *da, *dy0, *dh;       // [R][64], [R][256], [R][64], [R][256]: operands of the weight gradients / the recurrence
  float *dln2_w, *dln2_b, *dln1_w, *dln1_b;   // += (atomics; pre-zeroed)
  int R;
  const unsigned long long* seed;
  unsigned site0;
  float p_attn, p_drop;
};
int tb_launch_block_bwd(const TbBlockBwd& p, hipStream_t st);


// The four weight gradients of a block in one launch: dW[i][N][K] += P[i]^T Q[i], db[i][N] += column sums of P[i]
// (i = 0, 2: N = 64, K = 256; i = 1, 3: N = 256, K = 64), atomics onto pre-zeroed buffers.
struct TbWeightGrads {
  const float* P[4];
  const float* Q[4];
  float* dW[4];
  float* db[4];
  int R;
};
// ... and, in the same launch, the data gradient of in_proj: dx[R][64] += du[R][256] wi[256][64] (wiT = wi transposed)
int tb_launch_bwd_tail(const TbWeightGrads& p, const float* du, const float* wiT, float* dx, hipStream_t st);
