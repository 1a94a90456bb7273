// lru_encoder.hip -- LRURec history encoder for gfx950: embedding + LayerNorm, then per block
// {in_proj, diagonal complex linear-recurrence scan, out_proj.real + residual + LN, FFN + LN},
// returning the hidden state of the LAST position only.
//
// Replaces (reference): LRUEmbedding.forward model/lru.py:54-60, LRUModel.forward :73-83,
// LRULayer.forward / lru_parallel :135-161, PositionwiseFeedForward.forward :173-175.
//
// Design (MI355X-first, not a translation of the log2(L) recursive-doubling tensor program):
//  * one 256-thread workgroup (4 wave64) per user; only the user's LIVE tokens are touched
//    (positions after the last pad id; padding cannot reach the last position because the
//    mask severs the recurrence, model/lru.py:145), so Beauty/Games users cost ~8 tokens, not 64.
//  * the recurrence is a sequential scan h_t = lambda*h_{t-1} + bu_t carried in registers by
//    128 threads (one complex channel each) across 16-token tiles; projections are register-tiled
//    (each weight element loaded once per 16 tokens, coalesced from the L2-resident image).
//  * every reduction order is fixed and mirrored by oracle/lr_oracle.c -> bit-identical results:
//    k-ascending fmaf chains, butterfly LayerNorm sums over the 64 lanes of a wave.
//  * the last block's out_proj/FFN run only for the tile that holds the last token.
#include "lr_common.h"
#include "lr_profile.h"

#define TT 16  // tokens per tile

__device__ const float d_erf_tab[LR_ERF_NINT * (LR_ERF_DEG + 1)] = LR_ERF_TABLE;

struct EncParams {
  const float* img;
  LrLruLayout lay;
  const int64_t* ids;
  int B, L;
  float* out_q;
};

__device__ __forceinline__ float wave_sum64(float v) {
#pragma unroll
  for (int s = 32; s >= 1; s >>= 1) v = v + __shfl_xor(v, s, 64);
  return v;
}

// LayerNorm over the 64 features held one per lane.
__device__ __forceinline__ float wave_layer_norm(float x, float w, float b) {
  float mean = wave_sum64(x) * 0.015625f;
  float d = x - mean;
  float var = wave_sum64(d * d) * 0.015625f;
  float rstd = 1.0f / sqrtf(var + LR_LN_EPS);
  return lr_fma(d * rstd, w, b);
}

// out[t][tid] = bias[tid] + sum_k wt[k][tid] * in[t][k]   for all TT tokens (K = 64, 256 outputs)
__device__ __forceinline__ void proj_64_to_256(const float* __restrict__ wt, float bias,
                                               const float (*in)[64], int tid, float* acc) {
#pragma unroll
  for (int t = 0; t < TT; ++t) acc[t] = bias;
#pragma unroll 4
  for (int k = 0; k < 64; k += 4) {
    float w0 = wt[(k + 0) * 256 + tid], w1 = wt[(k + 1) * 256 + tid];
    float w2 = wt[(k + 2) * 256 + tid], w3 = wt[(k + 3) * 256 + tid];
#pragma unroll
    for (int t = 0; t < TT; ++t) {
      float4 x4 = *reinterpret_cast<const float4*>(&in[t][k]);
      acc[t] = lr_fma(w0, x4.x, acc[t]);
      acc[t] = lr_fma(w1, x4.y, acc[t]);
      acc[t] = lr_fma(w2, x4.z, acc[t]);
      acc[t] = lr_fma(w3, x4.w, acc[t]);
    }
  }
}

// acc[tt] = bias[lane] + sum_k wt[k][lane] * in[4*wave+tt][k]   (K = 256, 64 outputs per token)
__device__ __forceinline__ void proj_256_to_64(const float* __restrict__ wt, float bias,
                                               const float (*in)[256], int wave, int lane,
                                               float* acc) {
#pragma unroll
  for (int tt = 0; tt < 4; ++tt) acc[tt] = bias;
#pragma unroll 8
  for (int k = 0; k < 256; k += 4) {
    float w0 = wt[(k + 0) * 64 + lane], w1 = wt[(k + 1) * 64 + lane];
    float w2 = wt[(k + 2) * 64 + lane], w3 = wt[(k + 3) * 64 + lane];
#pragma unroll
    for (int tt = 0; tt < 4; ++tt) {
      float4 h4 = *reinterpret_cast<const float4*>(&in[4 * wave + tt][k]);
      acc[tt] = lr_fma(w0, h4.x, acc[tt]);
      acc[tt] = lr_fma(w1, h4.y, acc[tt]);
      acc[tt] = lr_fma(w2, h4.z, acc[tt]);
      acc[tt] = lr_fma(w3, h4.w, acc[tt]);
    }
  }
}

__global__ __launch_bounds__(256) void lru_encode_kernel(EncParams p) {
  __shared__ __attribute__((aligned(16))) float xs[TT][64];   // block input x (per token)
  __shared__ __attribute__((aligned(16))) float bu[TT][256];  // gamma*(W_in x+b) -> h -> FFN hidden
  __shared__ __attribute__((aligned(16))) float ys[TT][64];   // after the LRU layer's LayerNorm
  __shared__ int s_start;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int u = blockIdx.x;
  const int nb = p.lay.num_blocks;
  const int64_t* ids = p.ids + (size_t)u * p.L;
  const float* img = p.img;

  if (tid == 0) s_start = 0;
  __syncthreads();
  {
    int loc = 0;
    for (int t = tid; t < p.L - 1; t += 256)
      if (ids[t] <= 0) loc = t + 1;
    if (loc) atomicMax(&s_start, loc);
  }
  __syncthreads();
  const int start = s_start;
  const int n = p.L - start;  // >= 1

  float hr[LR_MAX_LRU_BLOCKS], hi[LR_MAX_LRU_BLOCKS];
#pragma unroll
  for (int b = 0; b < LR_MAX_LRU_BLOCKS; ++b) hr[b] = hi[b] = 0.0f;

  const float eln_w = img[p.lay.emb_ln_w + lane], eln_b = img[p.lay.emb_ln_b + lane];

  for (int tile0 = 0; tile0 < n; tile0 += TT) {
    const int nt = min(TT, n - tile0);
    const bool last_tile = (tile0 + TT >= n);
    // ---- embedding gather + LayerNorm: wave w owns tokens 4w..4w+3, lane = feature
#pragma unroll
    for (int tt = 0; tt < 4; ++tt) {
      int tok = 4 * wave + tt;
      float v = 0.0f;
      if (tok < nt) {
        long long id = ids[start + tile0 + tok];
        if (id < 0 || id > p.lay.num_items) id = 0;
        float e = img[p.lay.item_emb + (size_t)id * 64 + lane];
        v = wave_layer_norm(e, eln_w, eln_b);
      }
      xs[tok][lane] = v;
    }
    __syncthreads();

#pragma unroll
    for (int b = 0; b < LR_MAX_LRU_BLOCKS; ++b) {
      if (b < nb) {
        const LrLruBlockLayout& BL = p.lay.blk[b];
        float acc[TT];
        // ---- in_proj (complex, input imag == 0) and gamma
        proj_64_to_256(img + BL.in_wt, img[BL.in_b + tid], xs, tid, acc);
        {
          float g = img[BL.gamma + (tid & 127)];
#pragma unroll
          for (int t = 0; t < TT; ++t) bu[t][tid] = acc[t] * g;
        }
        __syncthreads();
        // ---- diagonal complex recurrence, one channel per thread, state carried across tiles
        if (tid < 128) {
          float lr_ = img[BL.lam_re + tid], li = img[BL.lam_im + tid];
          float h_r = hr[b], h_i = hi[b];
          for (int t = 0; t < nt; ++t) {
            float br = bu[t][tid], bi = bu[t][128 + tid];
            if (tile0 + t == 0) {
              h_r = br;
              h_i = bi;
            } else {
              float nr = lr_fma(lr_, h_r, lr_fma(-li, h_i, br));
              float ni = lr_fma(lr_, h_i, lr_fma(li, h_r, bi));
              h_r = nr;
              h_i = ni;
            }
            bu[t][tid] = h_r;
            bu[t][128 + tid] = h_i;
          }
          hr[b] = h_r;
          hi[b] = h_i;
        }
        __syncthreads();
        // the last block's tail is only consumed at the last position
        if (b < nb - 1 || last_tile) {
          float a4[4];
          // ---- Re(W_out h + b_out) + x, LayerNorm
          proj_256_to_64(img + BL.out_wt, img[BL.out_b + lane], bu, wave, lane, a4);
          {
            float w = img[BL.ln1_w + lane], bb = img[BL.ln1_b + lane];
#pragma unroll
            for (int tt = 0; tt < 4; ++tt) {
              float y0 = a4[tt] + xs[4 * wave + tt][lane];
              ys[4 * wave + tt][lane] = wave_layer_norm(y0, w, bb);
            }
          }
          __syncthreads();
          // ---- FFN: GELU(W1 y + b1)
          proj_64_to_256(img + BL.w1t, img[BL.b1 + tid], ys, tid, acc);
#pragma unroll
          for (int t = 0; t < TT; ++t) bu[t][tid] = lr_gelu_tab(acc[t], d_erf_tab);
          __syncthreads();
          // ---- W2 a + b2 + y, LayerNorm -> next block's input
          proj_256_to_64(img + BL.w2t, img[BL.b2 + lane], bu, wave, lane, a4);
          {
            float w = img[BL.ln2_w + lane], bb = img[BL.ln2_b + lane];
#pragma unroll
            for (int tt = 0; tt < 4; ++tt) {
              float z0 = a4[tt] + ys[4 * wave + tt][lane];
              xs[4 * wave + tt][lane] = wave_layer_norm(z0, w, bb);
            }
          }
          __syncthreads();
        }
      }
    }
  }
  const int tl = (n - 1) % TT;
  if (tid < 64) p.out_q[(size_t)u * 64 + tid] = xs[tl][tid];
}

int lr_launch_lru_encode(const lr_lru* h, const int64_t* ids, int B, int L, float* out_q,
                         hipStream_t st) {
  if (B <= 0) return LR_OK;
  EncParams p;
  p.img = h->img;
  p.lay = h->lay;
  p.ids = ids;
  p.B = B;
  p.L = L;
  p.out_q = out_q;
  LrProfScope prof(LR_PROF_LRU_ENCODE, (double)B, st);
  hipLaunchKernelGGL(lru_encode_kernel, dim3(B), dim3(256), 0, st, p);
  LR_CHECK_LAUNCH("lru_encode_kernel");
  return LR_OK;
}
