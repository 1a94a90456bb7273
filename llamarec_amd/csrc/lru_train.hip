// lru_train.hip -- LRURec retriever TRAINING step on gfx950.
//
// Replaces (reference): LRUTrainer.calculate_loss trainer/lru.py:20-28 (teacher-forced cross-entropy over
// all positions, ignore_index = 0) with torch autograd through model/lru.py:38-175, and
// BaseTrainer's clip_gradients + AdamW step (trainer/base.py:106-112,201-246).
//
// Shape of the computation (R = B*L token rows, row-major fp32 everywhere):
//   forward   embed+LN -> per block { in_proj (gamma folded into the weights) -> diagonal complex scan over time
//             -> Re(out_proj)+residual+LN -> W1+GELU -> W2+residual+LN } -> item GEMM -> softmax CE
//   backward  the same chain reversed; all dense products (including the three V-sized ones of the tied
//             item table: logits, d hidden, d table) run on one strided f32 MFMA GEMM
//             (v_mfma_f32_32x32x2_f32); the recurrence is a reverse-time scan per (sequence, channel):
//                 G_t = g_t + m_t conj(lambda) G_{t+1},   d lambda = sum_t m_{t-1} conj(h_{t-1}) G_t
//   update    global-norm clipping + AdamW over ONE flat parameter buffer (decoupled decay only on the
//             reference's "decay" group: names without "bias" / "layer_norm").
// Complex parameters are stored interleaved (re, im) like torch.view_as_real and carry the gradient
// dL/dRe + i dL/dIm -- exactly what torch autograd leaves in .grad and what AdamW consumes.
// The recursive-doubling scan of the reference equals the sequential recurrence on left-padded batches
// (dataloader/lru.py:119-131), pads included: h_t = u_t + m_{t-1} lambda h_{t-1}.
//
// Correctness-first except for the V-sized part: one generic GEMM and simple row kernels for the blocks, fp32
// atomics for the cross-row reductions (parameter-gradient sums); the item GEMM + cross-entropy is fused and
// never materialises the logits (lru_train_ce.hip).
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "lr_common.h"
#include "lr_profile.h"
#include "lru_train_blocks.h"
#include "lru_train_scores.h"
#include "lr_det.h"
LR_DET_DEFINE(train)

typedef float floatx16 __attribute__((ext_vector_type(16)));

// fused item GEMM + cross-entropy (lru_train_ce.hip)
size_t lr_train_ce_part_floats(int R, int C);
int lr_launch_train_ce(const float* X, const float* E, const float* bias, const long long* labels, int R, int C,
                       float* ws_part, float* scal, float* dX, float* dE, float* dbias, hipStream_t st);

// =============================================================================================
// flat parameter layout (floats)
// =============================================================================================
struct TrBlockOff {
  size_t plog, in_w, in_b, out_w, out_b, ln1_w, ln1_b, w1, b1, w2, b2, ln2_w, ln2_b;
};
struct TrLayout {
  int V, nb;
  size_t emb, eln_w, eln_b, bias;
  TrBlockOff blk[LR_MAX_LRU_BLOCKS];
  size_t total;
};
struct TrSeg {
  const char* suffix;  // name inside a block, or full name when blk < 0
  size_t off, count;
  int decay;
};

static TrLayout tr_layout(int V, int nb) {
  TrLayout L;
  L.V = V;
  L.nb = nb;
  size_t o = 0;
  auto take = [&](size_t n) {
    size_t at = o;
    o += lr_align_up(n, 64);
    return at;
  };
  L.emb = take((size_t)(V + 1) * 64);
  L.eln_w = take(64);
  L.eln_b = take(64);
  L.bias = take((size_t)V + 1);
  for (int b = 0; b < LR_MAX_LRU_BLOCKS; ++b) {
    TrBlockOff& B = L.blk[b];
    if (b >= nb) {
      B = TrBlockOff{};
      continue;
    }
    B.plog = take(3 * 128);
    B.in_w = take(128 * 64 * 2);
    B.in_b = take(128 * 2);
    B.out_w = take(64 * 128 * 2);
    B.out_b = take(64 * 2);
    B.ln1_w = take(64);
    B.ln1_b = take(64);
    B.w1 = take(256 * 64);
    B.b1 = take(256);
    B.w2 = take(64 * 256);
    B.b2 = take(64);
    B.ln2_w = take(64);
    B.ln2_b = take(64);
  }
  L.total = o;
  return L;
}

// segments with the reference's state_dict names (trainer/base.py:222: decay unless "bias"/"layer_norm" in name)
static int tr_segments(const TrLayout& L, int blk, TrSeg* out) {
  int n = 0;
  if (blk < 0) {
    out[n++] = {"embedding.token.weight", L.emb, (size_t)(L.V + 1) * 64, 1};
    out[n++] = {"embedding.layer_norm.weight", L.eln_w, 64, 0};
    out[n++] = {"embedding.layer_norm.bias", L.eln_b, 64, 0};
    out[n++] = {"model.bias", L.bias, (size_t)L.V + 1, 0};
    return n;
  }
  const TrBlockOff& B = L.blk[blk];
  out[n++] = {"lru_layer.params_log", B.plog, 3 * 128, 1};
  out[n++] = {"lru_layer.in_proj.weight", B.in_w, 128 * 64 * 2, 1};
  out[n++] = {"lru_layer.in_proj.bias", B.in_b, 128 * 2, 0};
  out[n++] = {"lru_layer.out_proj.weight", B.out_w, 64 * 128 * 2, 1};
  out[n++] = {"lru_layer.out_proj.bias", B.out_b, 64 * 2, 0};
  out[n++] = {"lru_layer.layer_norm.weight", B.ln1_w, 64, 0};
  out[n++] = {"lru_layer.layer_norm.bias", B.ln1_b, 64, 0};
  out[n++] = {"feed_forward.w_1.weight", B.w1, 256 * 64, 1};
  out[n++] = {"feed_forward.w_1.bias", B.b1, 256, 0};
  out[n++] = {"feed_forward.w_2.weight", B.w2, 64 * 256, 1};
  out[n++] = {"feed_forward.w_2.bias", B.b2, 64, 0};
  out[n++] = {"feed_forward.layer_norm.weight", B.ln2_w, 64, 0};
  out[n++] = {"feed_forward.layer_norm.bias", B.ln2_b, 64, 0};
  return n;
}

struct lr_lru_train {
  TrLayout lay;
  LrLruTrainConfig cfg;
  float *p, *g, *m, *v;  // flat buffers [total]
  unsigned char* decay;  // [total] 1 = decoupled weight decay applies
  float* scal;              // [8] device scalars: 0 loss sum, 1 n_valid, 2 grad norm^2, 4 lr, 5 clip limit
  unsigned long long* ctr;  // [4] device counters: optimizer steps, forward passes, current dropout seed
  int device;
  // optional hipGraph replay of the two launch sequences (keyed by every pointer and shape baked into them)
  int use_graph;
  int fused;   // 1: the blocks run as row-panel kernels (lru_train_blocks.hip); 0: one generic GEMM launch per product
  int deterministic;   // lr_lru_train_set_deterministic: atomic adds go to fixed-point shadows (lr_det.h)
  hipGraphExec_t g_fb, g_opt;
  const void *k_tok, *k_lab, *k_out, *k_ws, *k_norm;
  int k_B, k_L;
};

// =============================================================================================
// generic strided f32 GEMM on v_mfma_f32_32x32x2_f32
//   C[m][n] (ldc) = sum_k A(m,k) * B(k,n) (+ bias[n]) (+ C[m][n] if accumulate)
//   A(m,k) = A[m*sam + k*sak], B(k,n) = B[k*sbk + n*sbn]
// 64 x 64 tile, BK = 32, 4 waves (2 x 2) of one 32 x 32 MFMA block each.
// =============================================================================================
#define TG_BK 32

// gridDim.z > 1 splits K: every split ADDS its partial product with fp32 atomics (C must hold the value to
// accumulate onto, e.g. zero). rowsum (optional): rowsum[m] += sum_k A(m,k), taken from the A tiles by the
// workgroups of column block 0 -- the bias gradient that goes with a weight gradient dW = dY^T X.
// Tile BM x BN (64 or 128 each), 4 waves (2 x 2) of (BM/64) x (BN/64) MFMA blocks of 32 x 32: the three V-sized products
// of a step (scores, d hidden, d table: 4.95 GFLOP each on Beauty = 31 us at the f32 MFMA peak) run on 128-row tiles --
// on 64 x 64 a K step is 16 MFMAs per wave between two barriers and 16 LDS stores per thread, and they took 82-110 us.
template <int BM, int BN, bool SWAP, int BK = TG_BK>   // BK: K depth of a staged tile (32; 64 for the few-tile K >= 128 products)
__global__ __launch_bounds__(256) void train_gemm_kernel(const float* __restrict__ A, long long sam, long long sak,
                                                         const float* __restrict__ B, long long sbk, long long sbn,
                                                         float* C, long long ldc, const float* bias, int M, int N,
                                                         int K, int accumulate, float* rowsum) {
  constexpr int MI = BM / 64, NI = BN / 64, LDA = BM + 4, LDB = BN + 4;
  constexpr int NA = BM * BK / 256, NB = BN * BK / 256;   // tile elements per thread
  __shared__ float As[BK][LDA];
  __shared__ float Bs[BK][LDB];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
  floatx16 acc[MI][NI];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NI; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  const bool a_kfast = (sak == 1), b_kfast = (sbk == 1);
  const int ksteps = (K + BK - 1) / BK;
  const int ks0 = (int)((long long)blockIdx.z * ksteps / gridDim.z), ks1 = (int)((long long)(blockIdx.z + 1) * ksteps / gridDim.z);
  const bool want_rowsum = rowsum != nullptr && blockIdx.x == 0;
  float rs[MI];
#pragma unroll
  for (int i = 0; i < MI; ++i) rs[i] = 0.f;
  // register double buffer: the global loads of K step k+1 are issued before the MFMAs of step k
  float va[NA], vb[NB];
  auto load_tile = [&](int k0) {
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      const int e = tid + 256 * i;
      const int kk = a_kfast ? (e % BK) : (e / BM), mm = a_kfast ? (e / BK) : (e % BM);
      const int gm = m0 + mm, gk = k0 + kk;
      va[i] = (gm < M && gk < K) ? A[gm * sam + gk * sak] : 0.f;
    }
#pragma unroll
    for (int i = 0; i < NB; ++i) {
      const int e = tid + 256 * i;
      const int kk = b_kfast ? (e % BK) : (e / BN), nn = b_kfast ? (e / BK) : (e % BN);
      const int gn = n0 + nn, gk = k0 + kk;
      vb[i] = (gn < N && gk < K) ? B[gk * sbk + gn * sbn] : 0.f;
    }
  };
  if (ks0 < ks1) load_tile(ks0 * BK);
  for (int k0 = ks0 * BK; k0 < ks1 * BK; k0 += BK) {
    __syncthreads();  // previous tile consumed
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      const int e = tid + 256 * i;
      As[a_kfast ? (e % BK) : (e / BM)][a_kfast ? (e / BK) : (e % BM)] = va[i];
    }
#pragma unroll
    for (int i = 0; i < NB; ++i) {
      const int e = tid + 256 * i;
      Bs[b_kfast ? (e % BK) : (e / BN)][b_kfast ? (e / BK) : (e % BN)] = vb[i];
    }
    __syncthreads();
    if (k0 + BK < ks1 * BK) load_tile(k0 + BK);
    if (want_rowsum) {   // thread = (row tid & 63 (+ 64 i), k quarter tid >> 6): 8 independent LDS reads per K step and row.
                         // (One thread per row walking all 32 k of the step was a chain of 32 dependent reads, ~2 us per K
                         // step, with the other three waves parked at the next barrier: the eight weight-gradient
                         // products of a step took 22 us each for 105 MFLOP.)
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int kk = 0; kk < BK / 4; ++kk) rs[i] += As[(tid >> 6) * (BK / 4) + kk][(tid & 63) + 64 * i];
    }
#pragma unroll
    for (int s = 0; s < BK / 2; ++s) {
      float a[MI], b[NI];
#pragma unroll
      for (int i = 0; i < MI; ++i) a[i] = As[2 * s + (lane >> 5)][wm * (BM / 2) + i * 32 + (lane & 31)];
#pragma unroll
      for (int j = 0; j < NI; ++j) b[j] = Bs[2 * s + (lane >> 5)][wn * (BN / 2) + j * 32 + (lane & 31)];
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j)
          acc[i][j] = SWAP ? __builtin_amdgcn_mfma_f32_32x32x2f32(b[j], a[i], acc[i][j], 0, 0, 0)
                           : __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
    }
  }
  if (want_rowsum) {
#pragma unroll
    for (int i = 0; i < MI; ++i)
      if (m0 + (tid & 63) + 64 * i < M) atomicAdd(rowsum + m0 + (tid & 63) + 64 * i, rs[i]);
  }
  // Operands are swapped in the MFMA (D = B_frag x A_frag; the same products summed in the same k order, so the same
  // bits): a lane then owns ONE row (lane & 31) and, per group g, 4 CONSECUTIVE columns 8 g + 4 (lane >> 5) .. + 3 of each
  // 32 x 32 block -- 16-byte stores. With one column per lane the score product's 155 MB of logits left through 64
  // four-byte stores per lane (110-127 us for 4.95 GFLOP).
  // SWAP only without a K split: the split's atomic adds want a wave-instruction to cover whole 128-byte row segments
  // (a lane per column), which is the un-swapped layout -- 64 lanes in 64 different rows is the slow shape for atomics
  // (MI355X_MICROARCH.md, global float atomics; measured: 82-103 -> 264 us on the split products with the swap).
  if (!SWAP) {
#pragma unroll
    for (int j = 0; j < NI; ++j) {
      const int col = n0 + wn * (BN / 2) + j * 32 + (lane & 31);
      if (col >= N) continue;
      const float bv = (bias && blockIdx.z == 0) ? bias[col] : 0.f;
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = m0 + wm * (BM / 2) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
          if (row < M) {
            float v = acc[i][j][r] + bv;
            float* c = C + row * ldc + col;
            if (gridDim.z > 1) atomicAdd(c, v);
            else *c = accumulate ? (*c + v) : v;
          }
        }
    }
    return;
  }
  const bool vec_ok = (ldc & 3) == 0 && (reinterpret_cast<size_t>(C) & 15) == 0;
#pragma unroll
  for (int i = 0; i < MI; ++i) {
    const int row = m0 + wm * (BM / 2) + i * 32 + (lane & 31);
    if (row >= M) continue;
#pragma unroll
    for (int j = 0; j < NI; ++j)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int col = n0 + wn * (BN / 2) + j * 32 + 8 * g + 4 * (lane >> 5);
        if (col >= N) continue;
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = acc[i][j][4 * g + e] + ((bias && blockIdx.z == 0 && col + e < N) ? bias[col + e] : 0.f);
        float* c = C + row * ldc + col;
        if (gridDim.z > 1) {
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (col + e < N) atomicAdd(c + e, v[e]);
        } else if (vec_ok && col + 3 < N) {
          float4 o = make_float4(v[0], v[1], v[2], v[3]);
          if (accumulate) {
            const float4 old = *reinterpret_cast<const float4*>(c);
            o = make_float4(old.x + o.x, old.y + o.y, old.z + o.z, old.w + o.w);
          }
          *reinterpret_cast<float4*>(c) = o;
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (col + e < N) c[e] = accumulate ? (c[e] + v[e]) : v[e];
        }
      }
  }
}

// accumulate: 0 = overwrite, 1 = add to C. split_ok: the caller guarantees that C already holds the value to add to
// (zero for a fresh result), so K may be split over workgroups with atomic adds when the tiles alone are too few.
static int tr_gemm(const float* A, long long sam, long long sak, const float* B, long long sbk, long long sbn, float* C,
                   long long ldc, const float* bias, int M, int N, int K, int accumulate, hipStream_t st,
                   bool split_ok = false, float* rowsum = nullptr) {
  if (M <= 0 || N <= 0) return LR_OK;
  dim3 grid((N + 63) / 64, (M + 63) / 64, 1);
  const int tiles = grid.x * grid.y;
  // Few tiles and a long K: the launch is a chain of K steps, each one exposed global latency (one step of register
  // prefetch) -- [3 200 x 256] x [256 -> 64] is 50 workgroups walking 8 steps in 17 us for 0.1 GFLOP, the weight
  // gradients (K = 3 200 over 12 splits) 8 steps in 21-23 us (tools/gpu_train_trace.sh). Deep tiles (BK = 64: half of
  // the steps, twice the loads in flight per step; 128 would need 70 KB of static LDS) for those; the V-sized products keep BK = 32 (they live on occupancy).
  const bool deep = tiles <= 128 && K >= 128 && K <= 4096;   // (the V-sized d hidden product, K = V + 1 over 20 splits, loses: 86 -> 101 us)
  const int bk = deep ? 64 : TG_BK;
  if (split_ok) {
    const int ksteps = (K + bk - 1) / bk;
    int S = 1024 / tiles;                          // ~4 workgroups per CU
    const int min_steps = deep ? 2 : 8;            // K steps per split
    if (S > ksteps / min_steps) S = ksteps / min_steps;
    if (S > 1) grid.z = S;
  }
  // 64 x 64 tiles for every product: 128-row tiles (the template still takes them) were measured on the three V-sized
  // products of the Beauty step and lost -- scores 110 -> 127 us, the split products 82-88 -> 103 us (3 instead of 8
  // workgroups per CU, and these launches live on occupancy: K = 64 is two K steps between a cold start and the store)
  if (deep) {
    if (grid.z > 1)
      hipLaunchKernelGGL((train_gemm_kernel<64, 64, false, 64>), grid, dim3(256), 0, st, A, sam, sak, B, sbk, sbn, C, ldc, bias, M,
                         N, K, accumulate, rowsum);
    else
      hipLaunchKernelGGL((train_gemm_kernel<64, 64, true, 64>), grid, dim3(256), 0, st, A, sam, sak, B, sbk, sbn, C, ldc, bias, M,
                         N, K, accumulate, rowsum);
  } else if (grid.z > 1)
    hipLaunchKernelGGL((train_gemm_kernel<64, 64, false>), grid, dim3(256), 0, st, A, sam, sak, B, sbk, sbn, C, ldc, bias, M, N,
                       K, accumulate, rowsum);
  else
    hipLaunchKernelGGL((train_gemm_kernel<64, 64, true>), grid, dim3(256), 0, st, A, sam, sak, B, sbk, sbn, C, ldc, bias, M, N,
                       K, accumulate, rowsum);
  LR_CHECK_LAUNCH("train_gemm_kernel");
  return LR_OK;
}
// Y[R][N] = X[R][K] W[N][K]^T + b
static int tr_linear_fwd(const float* X, const float* W, const float* b, float* Y, int R, int N, int K, hipStream_t st) {
  return tr_gemm(X, K, 1, W, 1, K, Y, N, b, R, N, K, 0, st);
}
// dX[R][K] = dY[R][N] W[N][K]      (accumulate optional)
static int tr_linear_bwd_data(const float* dY, const float* W, float* dX, int R, int N, int K, int acc, hipStream_t st) {
  return tr_gemm(dY, N, 1, W, K, 1, dX, K, nullptr, R, K, N, acc, st, /*split_ok=*/acc != 0);   // adding onto dX: K may be split
}
// dW[N][K] += dY[R][N]^T X[R][K] and db[N] += column sums of dY   (both buffers pre-zeroed: K = R is split)
static int tr_linear_bwd_weight(const float* dY, const float* X, float* dW, float* db, int R, int N, int K,
                                hipStream_t st) {
  return tr_gemm(dY, 1, N, X, K, 1, dW, K, nullptr, N, K, R, 1, st, true, db);
}

// =============================================================================================
// row kernels (one wave per 64-feature row)
// =============================================================================================
__device__ __forceinline__ float tr_wave_sum(float v) {
#pragma unroll
  for (int s = 32; s >= 1; s >>= 1) v += __shfl_xor(v, s, 64);
  return v;
}

// x = LN(dropout(E[id])): saves xhat and rstd
__global__ __launch_bounds__(256) void tr_embed_ln_fwd(const long long* ids, const float* E, int V, const float* w,
                                                       const float* b, float* x, float* xhat, float* rstd, int R,
                                                       const unsigned long long* seedp, float p) {
  const unsigned long long seed = *seedp;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= R) return;
  long long id = ids[row];
  if (id < 0 || id > V) id = 0;
  float e = E[id * 64 + lane] * tr_drop_scale(seed, 0, (unsigned long long)row * 64 + lane, p);
  const float mu = tr_wave_sum(e) * (1.0f / 64);
  const float d = e - mu;
  const float rs = 1.0f / sqrtf(tr_wave_sum(d * d) * (1.0f / 64) + LR_LN_EPS);
  const float xh = d * rs;
  xhat[row * 64 + lane] = xh;
  x[row * 64 + lane] = xh * w[lane] + b[lane];
  if (lane == 0) rstd[row] = rs;
}

// y = LN(dropout(o) + res)
__global__ __launch_bounds__(256) void tr_res_ln_fwd(const float* o, const float* res, const float* w, const float* b,
                                                     float* y, float* xhat, float* rstd, int R,
                                                     const unsigned long long* seedp, unsigned site, float p) {
  const unsigned long long seed = *seedp;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= R) return;
  const size_t i = (size_t)row * 64 + lane;
  const float e = o[i] * tr_drop_scale(seed, site, i, p) + res[i];
  const float mu = tr_wave_sum(e) * (1.0f / 64);
  const float d = e - mu;
  const float rs = 1.0f / sqrtf(tr_wave_sum(d * d) * (1.0f / 64) + LR_LN_EPS);
  const float xh = d * rs;
  xhat[i] = xh;
  y[i] = xh * w[lane] + b[lane];
  if (lane == 0) rstd[row] = rs;
}

// LN backward for y = LN(e): de (pre-LN gradient); dw/db: each wave sums its 8 rows in registers, the workgroup
// adds once per column (32 rows per workgroup keeps the atomics on the 64 shared addresses few)
#define TR_LNB_ROWS 32
// res (optional) / seedp / site / p: the two consumers of a block's pre-LN gradient -- the residual path takes it as it is
// (res, may alias dy: a lane reads dy[i] before it writes res[i]), the branch behind it through its dropout mask (de);
// one kernel instead of LN backward + copy + dropout backward.
__global__ __launch_bounds__(256) void tr_ln_bwd(const float* dy, const float* xhat, const float* rstd, const float* w,
                                                 float* de, float* dw, float* db, int R, float* res = nullptr,
                                                 const unsigned long long* seedp = nullptr, unsigned site = 0, float p = 0.f,
                                                 const long long* ids = nullptr, int V = 0, float* dE = nullptr,
                                                 const float* scal = nullptr, float* out_loss = nullptr) {
  if (out_loss && blockIdx.x == 0 && threadIdx.x == 0) {   // the loss the pass reports (its last launch carries it)
    out_loss[0] = scal[0] / fmaxf(scal[1], 1.0f);
    out_loss[1] = scal[1];
    out_loss[2] = scal[3];
  }
  __shared__ float sw[4][64], sb[4][64];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const float wl = w[lane];
  float gw = 0.f, gb = 0.f;
#pragma unroll 2
  for (int j = 0; j < TR_LNB_ROWS / 4; ++j) {
    const int row = blockIdx.x * TR_LNB_ROWS + j * 4 + wave;
    if (row < R) {
      const size_t i = (size_t)row * 64 + lane;
      const float g = dy[i], xh = xhat[i];
      gw += g * xh;
      gb += g;
      const float dxh = g * wl;
      const float m1 = tr_wave_sum(dxh) * (1.0f / 64);
      const float m2 = tr_wave_sum(dxh * xh) * (1.0f / 64);
      const float d = rstd[row] * (dxh - m1 - xh * m2);
      if (dE) {   // the embedding LayerNorm: the row's gradient goes straight into its table row (through the embedding
                  // dropout's mask; unlabelled pad positions have an exactly zero gradient and are skipped)
        long long id = ids[row];
        if (id < 0 || id > V) id = 0;
        const float v = d * tr_drop_scale(*seedp, site, i, p);
        if (v != 0.f) lr_det_add(dE + id * 64 + lane, v);
      } else if (res) {
        res[i] = d;
        de[i] = d * tr_drop_scale(*seedp, site, i, p);
      } else {
        de[i] = d;
      }
    }
  }
  sw[wave][lane] = gw;
  sb[wave][lane] = gb;
  __syncthreads();
  if (wave == 0) {
    lr_det_add(dw + lane, sw[0][lane] + sw[1][lane] + sw[2][lane] + sw[3][lane]);
    lr_det_add(db + lane, sb[0][lane] + sb[1][lane] + sb[2][lane] + sb[3][lane]);
  }
}

// g = dropout(gelu(a)) elementwise; backward: da = dg * mask * gelu'(a)
__global__ void tr_gelu_fwd(const float* a, float* g, size_t n, const unsigned long long* seedp, unsigned site, float p) {
  const unsigned long long seed = *seedp;
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float x = a[i];
  g[i] = 0.5f * x * (1.0f + erff(x * 0.70710678118654752f)) * tr_drop_scale(seed, site, i, p);
}
__global__ void tr_gelu_bwd(const float* a, float* dg_to_da, size_t n, const unsigned long long* seedp, unsigned site, float p) {
  const unsigned long long seed = *seedp;
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float x = a[i];
  const float d = 0.5f * (1.0f + erff(x * 0.70710678118654752f)) + x * 0.3989422804014327f * expf(-0.5f * x * x);
  dg_to_da[i] = dg_to_da[i] * tr_drop_scale(seed, site, i, p) * d;
}
// x *= dropout mask (backward of a dropout whose forward was fused elsewhere)
__global__ void tr_drop_bwd(float* x, size_t n, const unsigned long long* seedp, unsigned site, float p) {
  const unsigned long long seed = *seedp;
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) x[i] *= tr_drop_scale(seed, site, i, p);
}
// =============================================================================================
// derived per-step weights of a block and their gradients
// =============================================================================================
struct TrDerived {  // offsets (floats) inside the workspace's derived region, per block
  size_t wi, bi, wo, bo, lam;  // [256][64], [256], [64][256], [64], [256] (re | im)
  size_t dwi, dbi, dwo, dbo, dlam;
};

struct TrPrepBlock {
  const float *plog, *in_w, *in_b, *out_w, *out_b, *w1, *w2;
  float *wi, *bi, *wo, *bo, *lam;
  TbTransposed t;
};
struct TrPrepArgs {
  TrPrepBlock blk[LR_MAX_LRU_BLOCKS];
};
// grid (64, blocks): the derived weights of every block in one launch; with_t: also the four 64 x 256 matrices and their
// transposes in the fragment order the row-panel kernels read (TbTransposed)
__global__ __launch_bounds__(256) void tr_prep_kernel(TrPrepArgs a, int with_t) {
  const TrPrepBlock& q = a.blk[blockIdx.y];
  const float *plog = q.plog, *in_w = q.in_w, *in_b = q.in_b, *out_w = q.out_w, *out_b = q.out_b;
  float *wi = q.wi, *bi = q.bi, *wo = q.wo, *bo = q.bo, *lam = q.lam;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;  // 0 .. 16383
  if (i < 128) {
    const float nu = expf(plog[i]), th = expf(plog[128 + i]);
    const float mag = expf(-nu);
    lam[i] = mag * cosf(th);
    lam[128 + i] = mag * sinf(th);
    const float ga = expf(plog[256 + i]);
    bi[i] = ga * in_b[2 * i];
    bi[128 + i] = ga * in_b[2 * i + 1];
  }
  if (i < 64) bo[i] = out_b[2 * i];
  if (i < 128 * 64) {  // in_w[c][k] complex
    const int c = i >> 6, k = i & 63;
    const float ga = expf(plog[256 + c]);
    const float re = ga * in_w[2 * i], im = ga * in_w[2 * i + 1];
    wi[c * 64 + k] = re;
    wi[(128 + c) * 64 + k] = im;
    if (with_t) {   // wi [256][64] and wi^T [64][256], fragment order
      q.t.wiF[tb_frag_pos(c, k, 64)] = re;
      q.t.wiF[tb_frag_pos(128 + c, k, 64)] = im;
      q.t.wiT[tb_frag_pos(k, c, 256)] = re;
      q.t.wiT[tb_frag_pos(k, 128 + c, 256)] = im;
    }
  }
  if (i < 64 * 128) {  // out_w[o][c] complex
    const int o = i >> 7, c = i & 127;
    const float re = out_w[2 * i], im = -out_w[2 * i + 1];
    wo[o * 256 + c] = re;
    wo[o * 256 + 128 + c] = im;
    if (with_t) {   // wo [64][256] and wo^T [256][64]
      q.t.woF[tb_frag_pos(o, c, 256)] = re;
      q.t.woF[tb_frag_pos(o, 128 + c, 256)] = im;
      q.t.woT[tb_frag_pos(c, o, 64)] = re;
      q.t.woT[tb_frag_pos(128 + c, o, 64)] = im;
    }
  }
  if (with_t) {   // w1 [256][64], w1^T [64][256]; w2 [64][256], w2^T [256][64]
    const float a1 = q.w1[i], a2 = q.w2[i];
    q.t.w1F[tb_frag_pos(i >> 6, i & 63, 64)] = a1;
    q.t.w1T[tb_frag_pos(i & 63, i >> 6, 256)] = a1;
    q.t.w2F[tb_frag_pos(i >> 8, i & 255, 256)] = a2;
    q.t.w2T[tb_frag_pos(i & 255, i >> 8, 64)] = a2;
  }
}

// gradients of the stored parameters from those of the derived ones: workgroup = complex channel c, thread = k / o
struct TrUnprepBlock {
  const float *plog, *in_w, *in_b, *lam, *dwi, *dbi, *dwo, *dbo, *dlam;
  float *g_plog, *g_in_w, *g_in_b, *g_out_w, *g_out_b;
};
struct TrUnprepArgs {
  TrUnprepBlock blk[LR_MAX_LRU_BLOCKS];
};
__global__ __launch_bounds__(64) void tr_unprep_kernel(TrUnprepArgs a) {   // grid (128 channels, blocks)
  const TrUnprepBlock& q = a.blk[blockIdx.y];
  const float *plog = q.plog, *in_w = q.in_w, *in_b = q.in_b, *lam = q.lam, *dwi = q.dwi, *dbi = q.dbi, *dwo = q.dwo, *dbo = q.dbo,
              *dlam = q.dlam;
  float *g_plog = q.g_plog, *g_in_w = q.g_in_w, *g_in_b = q.g_in_b, *g_out_w = q.g_out_w, *g_out_b = q.g_out_b;
  const int c = blockIdx.x, k = threadIdx.x;
  const float nu = expf(plog[c]), th = expf(plog[128 + c]), ga = expf(plog[256 + c]);
  const float dre = dwi[c * 64 + k], dim = dwi[(128 + c) * 64 + k];
  float dga = in_w[2 * (c * 64 + k)] * dre + in_w[2 * (c * 64 + k) + 1] * dim;
  g_in_w[2 * (c * 64 + k)] = ga * dre;
  g_in_w[2 * (c * 64 + k) + 1] = ga * dim;
#pragma unroll
  for (int s = 32; s >= 1; s >>= 1) dga += __shfl_xor(dga, s, 64);
  const int o = k;  // out_w[o][c]
  g_out_w[2 * (o * 128 + c)] = dwo[o * 256 + c];
  g_out_w[2 * (o * 128 + c) + 1] = -dwo[o * 256 + 128 + c];
  if (k == 0) {
    dga += in_b[2 * c] * dbi[c] + in_b[2 * c + 1] * dbi[128 + c];
    g_in_b[2 * c] = ga * dbi[c];
    g_in_b[2 * c + 1] = ga * dbi[128 + c];
    const float lr_ = lam[c], li = lam[128 + c], dr = dlam[c], di = dlam[128 + c];
    // d/d nu = -Re(dlam conj(lam)); d/d theta = Re(dlam conj(i lam)) = dr*(-li) + di*lr
    g_plog[c] = -(dr * lr_ + di * li) * nu;
    g_plog[128 + c] = (-dr * li + di * lr_) * th;
    g_plog[256 + c] = dga * ga;
    if (c < 64) {
      g_out_b[2 * c] = dbo[c];
      g_out_b[2 * c + 1] = 0.f;
    }
  }
}

// =============================================================================================
// the recurrence: forward in place (u -> h), backward in place (g -> du) + d lambda
// =============================================================================================
// grid: B blocks of 128 threads (thread = complex channel); rows b*L .. b*L+L-1, columns c (re) and 128+c (im).
// The recurrence is a dependent chain of L steps, but its INPUTS are not: written load -> step -> store per t, every
// step waited for its own loads (one memory latency per step: 15 us forward, 32 us backward for L = 50). A chunk of
// TR_SCAN_T steps now has all its loads in flight before the first step runs (one exposed latency per chunk).
#define TR_SCAN_T 32
__global__ __launch_bounds__(128) void tr_scan_fwd(float* uh, const long long* ids, const float* lam, int L) {
  extern __shared__ unsigned char live[];  // live[t] = ids[b][t] > 0 (kept out of the dependent chain)
  const int b = blockIdx.x, c = threadIdx.x;
  for (int t = c; t < L; t += 128) live[t] = ids[(size_t)b * L + t] > 0;
  __syncthreads();
  const float lr_ = lam[c], li = lam[128 + c];
  float hr = 0.f, hi = 0.f;
  float* base = uh + (size_t)b * L * 256;
  for (int t0 = 0; t0 < L; t0 += TR_SCAN_T) {
    float ur[TR_SCAN_T], ui[TR_SCAN_T];
#pragma unroll
    for (int i = 0; i < TR_SCAN_T; ++i) {
      const int t = min(t0 + i, L - 1);
      ur[i] = base[t * 256 + c];
      ui[i] = base[t * 256 + 128 + c];
    }
#pragma unroll
    for (int i = 0; i < TR_SCAN_T; ++i) {
      const int t = t0 + i;
      if (t < L) {
        const bool carry = t > 0 && live[t - 1];
        const float nr = carry ? ur[i] + (lr_ * hr - li * hi) : ur[i];
        const float ni = carry ? ui[i] + (lr_ * hi + li * hr) : ui[i];
        hr = nr;
        hi = ni;
        base[t * 256 + c] = hr;
        base[t * 256 + 128 + c] = hi;
      }
    }
  }
}
__global__ __launch_bounds__(128) void tr_scan_bwd(float* g, const float* h, const long long* ids, const float* lam,
                                                   float* dlam, int L) {
  extern __shared__ unsigned char live[];
  const int b = blockIdx.x, c = threadIdx.x;
  for (int t = c; t < L; t += 128) live[t] = ids[(size_t)b * L + t] > 0;
  __syncthreads();
  const float lr_ = lam[c], li = lam[128 + c];
  float Gr = 0.f, Gi = 0.f, dr = 0.f, di = 0.f;
  float* gb = g + (size_t)b * L * 256;
  const float* hb = h + (size_t)b * L * 256;
  for (int t1 = L - 1; t1 >= 0; t1 -= TR_SCAN_T) {   // chunk t1, t1 - 1, .. (descending)
    float vr[TR_SCAN_T], vi[TR_SCAN_T], pr[TR_SCAN_T], pi[TR_SCAN_T];
#pragma unroll
    for (int i = 0; i < TR_SCAN_T; ++i) {
      const int t = max(t1 - i, 0), tp = max(t1 - i - 1, 0);
      vr[i] = gb[t * 256 + c];
      vi[i] = gb[t * 256 + 128 + c];
      pr[i] = hb[tp * 256 + c];
      pi[i] = hb[tp * 256 + 128 + c];
    }
#pragma unroll
    for (int i = 0; i < TR_SCAN_T; ++i) {
      const int t = t1 - i;
      if (t >= 0) {
        float gr = vr[i], gi = vi[i];
        if (t < L - 1 && live[t]) {  // h_{t+1} = u_{t+1} + lambda h_t : G_t += conj(lambda) G_{t+1}
          gr += lr_ * Gr + li * Gi;
          gi += lr_ * Gi - li * Gr;
        }
        Gr = gr;
        Gi = gi;
        gb[t * 256 + c] = Gr;
        gb[t * 256 + 128 + c] = Gi;
        if (t > 0 && live[t - 1]) {  // d lambda += conj(h_{t-1}) G_t
          dr += pr[i] * Gr + pi[i] * Gi;
          di += pr[i] * Gi - pi[i] * Gr;
        }
      }
    }
  }
  lr_det_add(dlam + c, dr);
  lr_det_add(dlam + 128 + c, di);
}

// =============================================================================================
// softmax cross-entropy over a chunk of logit rows, in place: logits -> d logits; loss sum accumulated
// =============================================================================================
// labels: 0 = ignored (CrossEntropyLoss(ignore_index=0)); a label outside [0, V] would index past the logit row --
// torch raises there, here such rows are ignored too and counted in scal[3] (reported as out_loss[2])
__device__ __forceinline__ float tr_block_reduce(float v, bool is_max, float* sh) {
#pragma unroll
  for (int s = 32; s >= 1; s >>= 1) {
    const float o = __shfl_xor(v, s, 64);
    v = is_max ? fmaxf(v, o) : v + o;
  }
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  __syncthreads();
  if (lane == 0) sh[wave] = v;
  __syncthreads();
  float r = sh[0];
  for (int w = 1; w < 4; ++w) r = is_max ? fmaxf(r, sh[w]) : r + sh[w];
  return r;
}
// Materialised variant of the cross-entropy (small problems, see TR_MATERIALISE_ELEMS): softmax over one stored logit
// row, rewritten in place into d logits; loss sum accumulated.
__global__ __launch_bounds__(256) void tr_ce_kernel(float* logits, long long ld, int C, const long long* labels,
                                                    float* scal) {
  __shared__ float sh[4];
  const int row = blockIdx.x;
  float* x = logits + row * ld;
  const long long lab = labels[row];
  if (lab <= 0 || lab >= C) {  // ignore_index (or out of range: counted in scal[3]): no loss, no gradient
    for (int j = threadIdx.x; j < C; j += 256) x[j] = 0.f;
    return;
  }
  float mx = -__builtin_inff();
  for (int j = threadIdx.x; j < C; j += 256) mx = fmaxf(mx, x[j]);
  mx = tr_block_reduce(mx, true, sh);
  float se = 0.f;
  for (int j = threadIdx.x; j < C; j += 256) se += expf(x[j] - mx);
  se = tr_block_reduce(se, false, sh);
  const float inv_n = 1.0f / scal[1], inv_se = 1.0f / se;
  const float picked = x[lab];
  __syncthreads();
  for (int j = threadIdx.x; j < C; j += 256) {
    const float p = expf(x[j] - mx) * inv_se;
    x[j] = (p - (j == lab ? 1.0f : 0.0f)) * inv_n;
  }
  if (threadIdx.x == 0) atomicAdd(scal, logf(se) + mx - picked);
}
// device-side counters (so that a captured graph can be replayed): ctr[0] optimizer steps, ctr[1] forward passes,
// ctr[2] dropout seed of the current pass (advanced by tr_pass_init_kernel)
// zero / copy as kernels, not hipMemsetAsync / hipMemcpyAsync. Round 1 saw wrong gradients "from the second step on"
// with memset / memcpy nodes in the captured step and blamed their ordering. tools/diag/graph_memset_order.cpp
// (profiles/r02_graph_memset_order.txt) shows the ordering is fine on this stack -- a captured kernel -> memset ->
// kernel -> memcpy chain gets the linear dependency edges and replays like the eager stream. What does differ is what
// a captured memcpy node REMEMBERS: the host POINTER, not the bytes. The step used to copy its per-call scalars from a
// stack variable; the first launch (same call) read it alive, every later replay read a dead stack slot. The scalars
// now travel as launch arguments of a 1-thread kernel (tr_set_step_scalars, outside the captured part), and zeroing /
// copying stays in kernels so that nothing in the replayed graph refers to host memory at all.
// Start of a pass in ONE launch (they were six: the seed, three zero fills, the label count, the zero fill of d x): every
// workgroup zeroes its share of the gradient buffer, of the derived-weight gradients and of d x; workgroup 0 also advances
// the dropout seed, zeroes the step scalars and counts the labelled rows (scal[1]) and the out-of-range labels (scal[3])
// by itself -- no atomics onto words another workgroup of this launch might still be zeroing.
__global__ __launch_bounds__(256) void tr_pass_init_kernel(float* G, size_t n_g, float* dgrad, size_t n_d, float* dx, size_t n_x,
                                                           float* scal, unsigned long long* ctr, unsigned long long seedbase,
                                                           const long long* labels, int R, int V) {
  const size_t stride = (size_t)gridDim.x * 256, t = (size_t)blockIdx.x * 256 + threadIdx.x;
  for (size_t i = t; i < n_g; i += stride) G[i] = 0.f;
  for (size_t i = t; i < n_d; i += stride) dgrad[i] = 0.f;
  for (size_t i = t; i < n_x; i += stride) dx[i] = 0.f;
  if (blockIdx.x != 0) return;
  __shared__ int s_ok[4], s_bad[4];
  int c = 0, b = 0;
  for (int i = threadIdx.x; i < R; i += 256) {
    const long long l = labels[i];
    c += l > 0 && l <= V;
    b += l < 0 || l > V;
  }
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) {
    c += __shfl_xor(c, o, 64);
    b += __shfl_xor(b, o, 64);
  }
  if ((threadIdx.x & 63) == 0) {
    s_ok[threadIdx.x >> 6] = c;
    s_bad[threadIdx.x >> 6] = b;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    ctr[2] = seedbase + ctr[1] * 0xA24BAED4963EE407ull;   // this pass's dropout seed
    ctr[1] += 1;
    scal[0] = 0.f;
    scal[1] = (float)(s_ok[0] + s_ok[1] + s_ok[2] + s_ok[3]);
    scal[2] = 0.f;
    scal[3] = (float)(s_bad[0] + s_bad[1] + s_bad[2] + s_bad[3]);
  }
}
__global__ void tr_zero_kernel(float* p, size_t n) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = 0.f;
}
__global__ void tr_copy_kernel(float* dst, const float* src, size_t n) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) dst[i] = src[i];
}
__global__ void tr_set_step_scalars(float* scal, float lr, float max_norm) {
  scal[4] = lr;
  scal[5] = max_norm;
}

// =============================================================================================
// optimizer
// =============================================================================================
__global__ __launch_bounds__(256) void tr_sumsq_kernel(const float* g, size_t n, float* out, unsigned long long* ctr) {
  __shared__ float sh[4];
  if (blockIdx.x == 0 && threadIdx.x == 0) ctr[0] += 1;  // this optimizer step's number, read by tr_adamw_kernel
  float s = 0.f;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) s += g[i] * g[i];
  s = tr_block_reduce(s, false, sh);
  if (threadIdx.x == 0) atomicAdd(out, s);
}
// deterministic mode: TR_SUMSQ_PARTS workgroups store their partial sums (plain stores into the spare floats behind the state
// buffer's scalars and counters), one thread adds them in index order
#define TR_SUMSQ_PARTS 48
__global__ __launch_bounds__(256) void tr_sumsq_part_kernel(const float* g, size_t n, float* part, unsigned long long* ctr) {
  __shared__ float sh[4];
  if (blockIdx.x == 0 && threadIdx.x == 0) ctr[0] += 1;
  float s = 0.f;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) s += g[i] * g[i];
  s = tr_block_reduce(s, false, sh);
  if (threadIdx.x == 0) part[blockIdx.x] = s;
}
__global__ void tr_sumsq_final_kernel(const float* part, float* out) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    float s = 0.f;
    for (int i = 0; i < TR_SUMSQ_PARTS; ++i) s += part[i];
    *out = s;
  }
}
// scal[2] = sum of squared gradients, scal[4] = learning rate, scal[5] = clipping limit of this step
__global__ void tr_adamw_kernel(float* p, const float* g, float* m, float* v, const unsigned char* decay, size_t n,
                                const float* scal, const unsigned long long* ctr, float wd, float b1, float b2,
                                float eps, float* out_norm) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const float norm = sqrtf(scal[2]), lr = scal[4], max_norm = scal[5];
  const float t = (float)ctr[0];
  const float bc1 = 1.0f - powf(b1, t), bc2_sqrt = sqrtf(1.0f - powf(b2, t));
  if (i == 0 && out_norm) *out_norm = norm;
  if (i >= n) return;
  const float coef = fminf(1.0f, max_norm / (norm + 1e-6f));  // torch.nn.utils.clip_grad_norm_
  const float gi = g[i] * coef;
  float pi = p[i];
  if (decay[i]) pi *= 1.0f - lr * wd;
  const float mi = b1 * m[i] + (1.0f - b1) * gi;
  const float vi = b2 * v[i] + (1.0f - b2) * gi * gi;
  m[i] = mi;
  v[i] = vi;
  p[i] = pi - (lr / bc1) * mi / (sqrtf(vi) / bc2_sqrt + eps);
}

// =============================================================================================
// workspace
// =============================================================================================
struct TrBlockWs {
  TrDerived d;
  float *h, *xhat1, *rstd1, *y, *a, *g, *xhat2, *rstd2, *xout;  // saved activations
  TbTransposed t;                                                // transposed weights (row-panel path)
};
struct TrWs {
  float* derived;  // derived weights + their gradients, all blocks
  float *x0, *xhat0, *rstd0;
  TrBlockWs blk[LR_MAX_LRU_BLOCKS];
  float *d64a, *d64b, *d256;  // gradient scratch [R][64] x2, [R][256]
  float *d64c, *d256b;        // row-panel path: dy0 and da stay live until the block's weight-gradient launch
  float* ce;                  // cross-entropy scratch: the fused path's partials + lse, or the stored logits
  bool materialise;           // small problem: store the [R][V+1] logits (<= 256 MB), three plain GEMM passes over them
  // deterministic mode (lr_det.h): 64-bit fixed-point shadows of [gradient buffer | derived region | d x | 8 scalars]
  long long* shadow;
  size_t derived_floats, shadow_n;
  size_t total;
};

// The fused cross-entropy recomputes every 32 x 32 score tile in each of its three passes (5 GEMM-equivalents
// instead of 3) but never stores logits; while the logits fit the Infinity Cache (Beauty: 155 MB) storing them is
// ~10 % faster per step, at V = 10^6 the fused path is 3.7 x faster (290 -> 79 ms) and needs no 1 GiB buffer.
#define TR_MATERIALISE_ELEMS ((size_t)64 << 20)

static bool tr_use_materialised(const LrLruTrainConfig& cfg, int R, int C) {
  if (cfg.ce_mode == 1) return true;
  if (cfg.ce_mode == 2) return false;
  return (size_t)R * C <= TR_MATERIALISE_ELEMS;
}

static TrWs tr_carve(const TrLayout& lay, const LrLruTrainConfig& cfg, int R, char* base, bool det = false) {
  TrWs w;
  size_t o = 0;
  auto take = [&](size_t floats) {
    size_t at = o;
    o += lr_align_up(floats * sizeof(float), 256);
    return (float*)(base + at);
  };
  size_t doff = 0;
  auto dtake = [&](size_t n) {
    size_t at = doff;
    doff += lr_align_up(n, 64);
    return at;
  };
  for (int b = 0; b < lay.nb; ++b) {
    TrDerived& d = w.blk[b].d;
    d.wi = dtake(256 * 64);
    d.bi = dtake(256);
    d.wo = dtake(64 * 256);
    d.bo = dtake(64);
    d.lam = dtake(256);
    d.dwi = dtake(256 * 64);
    d.dbi = dtake(256);
    d.dwo = dtake(64 * 256);
    d.dbo = dtake(64);
    d.dlam = dtake(256);
  }
  w.derived = take(doff);
  w.derived_floats = doff;
  const size_t r = (size_t)R;
  w.x0 = take(r * 64);
  w.xhat0 = take(r * 64);
  w.rstd0 = take(r);
  for (int b = 0; b < lay.nb; ++b) {
    TrBlockWs& B = w.blk[b];
    B.h = take(r * 256);
    B.xhat1 = take(r * 64);
    B.rstd1 = take(r);
    B.y = take(r * 64);
    B.a = take(r * 256);
    B.g = take(r * 256);
    B.xhat2 = take(r * 64);
    B.rstd2 = take(r);
    B.xout = take(r * 64);
  }
  w.d64a = take(r * 64);
  w.d64b = take(r * 64);
  w.d256 = take(r * 256);
  w.d64c = take(r * 64);
  w.d256b = take(r * 256);
  for (int b = 0; b < lay.nb; ++b) {
    TbTransposed& t = w.blk[b].t;
    t.wiF = take(16384);
    t.woF = take(16384);
    t.w1F = take(16384);
    t.w2F = take(16384);
    t.wiT = take(16384);
    t.woT = take(16384);
    t.w1T = take(16384);
    t.w2T = take(16384);
  }
  w.materialise = tr_use_materialised(cfg, R, lay.V + 1);
  // stored logits: rows padded to a multiple of 4 floats, so that the score product's 16-byte stores are aligned
  {
    const size_t generic = (size_t)R * (((size_t)lay.V + 1 + 3) & ~(size_t)3), panels = lr_train_scores_ws_floats(R, lay.V + 1);
    w.ce = take(w.materialise ? (generic > panels ? generic : panels) : lr_train_ce_part_floats(R, lay.V + 1));
  }
  w.shadow = nullptr;
  w.shadow_n = det ? lay.total + doff + (size_t)R * 64 + 8 : 0;
  if (det) w.shadow = (long long*)take(2 * w.shadow_n);
  w.total = o;
  return w;
}

// turns a shadow back into its float region: f[i] += shadow[i] * 2^-k (regions whose plain writes and whose atomic adds
// never meet in one element: the float side holds zero, or the plain value, where the shadow holds the sum, or zero)
__global__ void lr_det_fold_kernel(float* f, const long long* shadow, size_t n, double inv_scale) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const long long v = shadow[i];
    if (v) f[i] += (float)((double)v * inv_scale);
  }
}
__global__ void lr_det_zero_kernel(long long* shadow, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) shadow[i] = 0;
}
#define LR_DET_GRAD_SCALE 281474976710656.0   // 2^48
#define LR_DET_SCAL_SCALE 4294967296.0        // 2^32

// =============================================================================================
// C ABI
// =============================================================================================
extern "C" size_t lr_lru_train_state_bytes(int32_t num_items, int32_t num_blocks) {
  if (num_items < 1 || num_blocks < 1 || num_blocks > LR_MAX_LRU_BLOCKS) return 0;
  const size_t n = tr_layout(num_items, num_blocks).total;
  return 4 * n * sizeof(float) + lr_align_up(n, 256) + 256;  // + decay mask + 8 float scalars + 4 counters
}

static int tr_find(const TrLayout& L, const char* name, TrSeg* out) {
  TrSeg segs[16];
  int n = tr_segments(L, -1, segs);
  for (int i = 0; i < n; ++i)
    if (!strcmp(name, segs[i].suffix)) {
      *out = segs[i];
      return 1;
    }
  const char* pre = "model.lru_blocks.";
  if (strncmp(name, pre, strlen(pre))) return 0;
  const char* q = name + strlen(pre);
  int b = 0;
  while (*q >= '0' && *q <= '9') b = b * 10 + (*q++ - '0');
  if (*q++ != '.' || b >= L.nb) return 0;
  n = tr_segments(L, b, segs);
  for (int i = 0; i < n; ++i)
    if (!strcmp(q, segs[i].suffix)) {
      *out = segs[i];
      return 1;
    }
  return 0;
}

extern "C" int lr_lru_train_create(const LrLruWeightsDesc* init, const LrLruTrainConfig* cfg, void* state_dev,
                                   size_t state_bytes, lr_lru_train_t** out) {
  if (!init || !cfg || !state_dev || !out) LR_FAIL(LR_EINVAL, "lr_lru_train_create: null argument");
  if (init->hidden != 64) LR_FAIL(LR_EUNSUPPORTED, "lr_lru_train_create: hidden=%d, only 64 is implemented", init->hidden);
  const size_t need = lr_lru_train_state_bytes(init->num_items, init->num_blocks);
  if (!need || state_bytes < need) LR_FAIL(LR_EINVAL, "lr_lru_train_create: state buffer %zu < %zu bytes", state_bytes, need);
  if (cfg->dropout < 0.f || cfg->dropout >= 1.f || cfg->attn_dropout < 0.f || cfg->attn_dropout >= 1.f)
    LR_FAIL(LR_EINVAL, "lr_lru_train_create: dropout outside [0, 1)");
  if (cfg->ce_mode < 0 || cfg->ce_mode > 2) LR_FAIL(LR_EINVAL, "lr_lru_train_create: ce_mode %d", cfg->ce_mode);
  lr_lru_train* h = (lr_lru_train*)calloc(1, sizeof(lr_lru_train));
  if (!h) LR_FAIL(LR_EINVAL, "lr_lru_train_create: out of host memory");
  h->lay = tr_layout(init->num_items, init->num_blocks);
  h->cfg = *cfg;
  h->fused = 1;
  h->deterministic = 0;
  if (const char* e = getenv("LR_TRAIN_FUSED")) h->fused = atoi(e) != 0;   // A/B knob; lr_lru_train_set_fused is the API
  const size_t n = h->lay.total;
  h->p = (float*)state_dev;
  h->g = h->p + n;
  h->m = h->g + n;
  h->v = h->m + n;
  h->decay = (unsigned char*)(h->v + n);
  h->scal = (float*)(h->decay + lr_align_up(n, 256));
  h->ctr = (unsigned long long*)(h->scal + 8);
  LR_CHECK_HIP(hipGetDevice(&h->device));
  // host image of the parameters and the decay mask, one upload each
  float* img = (float*)calloc(n, sizeof(float));
  unsigned char* dec = (unsigned char*)calloc(n, 1);
  if (!img || !dec) {
    free(img);
    free(dec);
    free(h);
    LR_FAIL(LR_EINVAL, "lr_lru_train_create: out of host memory");
  }
  const TrLayout& L = h->lay;
  const size_t rows = (size_t)L.V + 1;
  memcpy(img + L.emb, init->item_emb, rows * 64 * sizeof(float));
  memcpy(img + L.eln_w, init->emb_ln_w, 64 * sizeof(float));
  memcpy(img + L.eln_b, init->emb_ln_b, 64 * sizeof(float));
  memcpy(img + L.bias, init->item_bias, rows * sizeof(float));
  for (int b = 0; b < L.nb; ++b) {
    const LrLruBlockWeights& w = init->blocks[b];
    const TrBlockOff& B = L.blk[b];
    memcpy(img + B.plog, w.params_log, 3 * 128 * sizeof(float));
    memcpy(img + B.in_w, w.in_proj_w, 128 * 64 * 2 * sizeof(float));
    memcpy(img + B.in_b, w.in_proj_b, 128 * 2 * sizeof(float));
    memcpy(img + B.out_w, w.out_proj_w, 64 * 128 * 2 * sizeof(float));
    memcpy(img + B.out_b, w.out_proj_b, 64 * 2 * sizeof(float));
    memcpy(img + B.ln1_w, w.ln1_w, 64 * sizeof(float));
    memcpy(img + B.ln1_b, w.ln1_b, 64 * sizeof(float));
    memcpy(img + B.w1, w.ffn_w1, 256 * 64 * sizeof(float));
    memcpy(img + B.b1, w.ffn_b1, 256 * sizeof(float));
    memcpy(img + B.w2, w.ffn_w2, 64 * 256 * sizeof(float));
    memcpy(img + B.b2, w.ffn_b2, 64 * sizeof(float));
    memcpy(img + B.ln2_w, w.ln2_w, 64 * sizeof(float));
    memcpy(img + B.ln2_b, w.ln2_b, 64 * sizeof(float));
  }
  TrSeg segs[16];
  for (int b = -1; b < L.nb; ++b) {
    const int ns = tr_segments(L, b, segs);
    for (int i = 0; i < ns; ++i)
      if (segs[i].decay) memset(dec + segs[i].off, 1, segs[i].count);
  }
  hipError_t e1 = hipMemcpy(h->p, img, n * sizeof(float), hipMemcpyHostToDevice);
  hipError_t e2 = hipMemcpy(h->decay, dec, n, hipMemcpyHostToDevice);
  hipError_t e3 = hipMemset(h->g, 0, 3 * n * sizeof(float));
  hipError_t e4 = hipMemset(h->scal, 0, 8 * sizeof(float) + 4 * sizeof(unsigned long long));
  free(img);
  free(dec);
  if (e1 != hipSuccess || e2 != hipSuccess || e3 != hipSuccess || e4 != hipSuccess) {
    free(h);
    LR_FAIL(LR_EHIP, "lr_lru_train_create: upload failed");
  }
  *out = h;
  return LR_OK;
}

static void tr_drop_graphs(lr_lru_train* h) {
  if (h->g_fb) (void)hipGraphExecDestroy(h->g_fb);
  if (h->g_opt) (void)hipGraphExecDestroy(h->g_opt);
  h->g_fb = h->g_opt = nullptr;
}

extern "C" void lr_lru_train_destroy(lr_lru_train_t* h) {
  if (!h) return;
  tr_drop_graphs(h);
  free(h);
}

extern "C" int lr_lru_train_set_graph(lr_lru_train_t* h, int32_t enable) {
  if (!h) LR_FAIL(LR_EINVAL, "lr_lru_train_set_graph: null handle");
  h->use_graph = enable ? 1 : 0;
  if (!enable) tr_drop_graphs(h);
  return LR_OK;
}

extern "C" int lr_lru_train_set_fused(lr_lru_train_t* h, int32_t enable) {
  if (!h) LR_FAIL(LR_EINVAL, "lr_lru_train_set_fused: null handle");
  if ((enable != 0) != (h->fused != 0)) tr_drop_graphs(h);   // a captured step holds the other launch sequence
  h->fused = enable ? 1 : 0;
  return LR_OK;
}

static LrDetMap g_det_current = {};   // what the four translation units' constant maps hold (host copy)
static int tr_det_publish(const LrDetMap& m, hipStream_t st) {
  bool same = m.on == g_det_current.on;   // (field by field: the struct has padding that a memcmp would see)
  for (int k = 0; k < LR_DET_REGIONS && same; ++k)
    same = m.base[k] == g_det_current.base[k] && m.bytes[k] == g_det_current.bytes[k] && m.shadow[k] == g_det_current.shadow[k] &&
           m.scale[k] == g_det_current.scale[k];
  if (same) return LR_OK;
  LR_CHECK_HIP(hipStreamSynchronize(st));   // nothing in flight reads the old map
  if (lr_det_set_train(&m) || lr_det_set_blocks(&m) || lr_det_set_ce(&m) || lr_det_set_scores(&m))
    LR_FAIL(LR_EHIP, "deterministic mode: hipMemcpyToSymbol failed");
  g_det_current = m;
  return LR_OK;
}

extern "C" int lr_lru_train_set_deterministic(lr_lru_train_t* h, int32_t enable) {
  if (!h) LR_FAIL(LR_EINVAL, "lr_lru_train_set_deterministic: null handle");
  if ((enable != 0) != (h->deterministic != 0)) tr_drop_graphs(h);   // a captured step holds the other launch sequence
  h->deterministic = enable ? 1 : 0;
  if (!enable) {
    LrDetMap off = {};
    LR_CHECK_HIP(hipDeviceSynchronize());
    return tr_det_publish(off, nullptr);
  }
  return LR_OK;
}

extern "C" size_t lr_lru_train_workspace_bytes(const lr_lru_train_t* h, int32_t B, int32_t L) {
  if (!h || B < 1 || L < 1) return 0;
  return tr_carve(h->lay, h->cfg, B * L, nullptr, h->deterministic != 0).total;
}

extern "C" int lr_lru_train_buffers(lr_lru_train_t* h, float** params, float** grads, size_t* count) {
  if (!h) LR_FAIL(LR_EINVAL, "lr_lru_train_buffers: null handle");
  if (params) *params = h->p;
  if (grads) *grads = h->g;
  if (count) *count = h->lay.total;
  return LR_OK;
}

extern "C" int lr_lru_train_param_range(const lr_lru_train_t* h, const char* name, size_t* offset, size_t* count) {
  if (!h || !name || !offset || !count) LR_FAIL(LR_EINVAL, "lr_lru_train_param_range: null argument");
  TrSeg s;
  if (!tr_find(h->lay, name, &s)) LR_FAIL(LR_EINVAL, "lr_lru_train_param_range: unknown parameter '%s'", name);
  *offset = s.off;
  *count = s.count;
  return LR_OK;
}

#define TR_RUN(x)            \
  do {                       \
    int rc_ = (x);           \
    if (rc_) return rc_;     \
  } while (0)
#define TR_EW(kernel, n, ...)                                                                         \
  do {                                                                                                \
    hipLaunchKernelGGL(kernel, dim3((unsigned)(((n) + 255) / 256)), dim3(256), 0, st, __VA_ARGS__);   \
    LR_CHECK_LAUNCH(#kernel);                                                                         \
  } while (0)

static int tr_enqueue_loss_grad(lr_lru_train_t* h, const int64_t* tokens, const int64_t* labels, int32_t B, int32_t L,
                                float* out_loss, void* workspace, size_t workspace_bytes, hipStream_t st) {
  const TrLayout& lay = h->lay;
  const int R = B * L, V = lay.V, C = V + 1;
  TrWs ws = tr_carve(lay, h->cfg, R, (char*)workspace, h->deterministic != 0);
  if (ws.total > workspace_bytes)
    LR_FAIL(LR_EWORKSPACE, "lr_lru_train_loss_grad: workspace needs %zu bytes, have %zu", ws.total, workspace_bytes);
  const long long* ids = (const long long*)tokens;
  const long long* lab = (const long long*)labels;
  float *P = h->p, *G = h->g;
  const unsigned long long* seed = h->ctr + 2;
  const float pd = h->cfg.dropout, pa = h->cfg.attn_dropout;
  const unsigned grid_rows = (unsigned)((R + 3) / 4);
  float* dx = ws.d64b;   // gradient of the blocks' output, then of each block's input
  const bool det = h->deterministic != 0;
  // shadow layout: [gradient buffer | derived region | d x | scalars]
  long long* const sh_g = ws.shadow;
  long long* const sh_d = ws.shadow ? ws.shadow + lay.total : nullptr;
  long long* const sh_x = ws.shadow ? sh_d + ws.derived_floats : nullptr;
  long long* const sh_s = ws.shadow ? sh_x + (size_t)R * 64 : nullptr;
  auto fold = [&](float* f, const long long* sh, size_t n, double scale) {
    hipLaunchKernelGGL(lr_det_fold_kernel, dim3((unsigned)((n + 1023) / 1024 < 1024 ? (n + 1023) / 1024 : 1024)), dim3(256), 0, st, f, sh, n,
                       1.0 / scale);
  };
  if (det) {
    if (!h->fused)
      LR_FAIL(LR_EUNSUPPORTED, "lr_lru_train_loss_grad: deterministic mode runs the row-panel kernels only (lr_lru_train_set_fused(h, 1)): "
                               "the generic GEMM launches split K with atomics into activation buffers");
    hipLaunchKernelGGL(lr_det_zero_kernel, dim3(1024), dim3(256), 0, st, ws.shadow, ws.shadow_n);
    LR_CHECK_LAUNCH("lr_det_zero_kernel");
  }
  {  // seed, zero fills (gradients, derived-weight gradients, d x), label counts
    const TrDerived& d0 = ws.blk[0].d;
    const TrDerived& dl = ws.blk[lay.nb - 1].d;
    hipLaunchKernelGGL(tr_pass_init_kernel, dim3(512), dim3(256), 0, st, G, lay.total, ws.derived + d0.wi, dl.dlam + 256 - d0.wi, dx,
                       (size_t)R * 64, h->scal, h->ctr, h->cfg.seed * 0x9E3779B97F4A7C15ull, lab, R, V);
    LR_CHECK_LAUNCH("tr_pass_init_kernel");
  }
  // ---- forward
  {  // derived weights of every block (and, for the row-panel kernels, the transposed matrices): one launch
    TrPrepArgs pa_;
    for (int b = 0; b < LR_MAX_LRU_BLOCKS; ++b) {
      const int bb = b < lay.nb ? b : 0;
      const TrBlockOff& o = lay.blk[bb];
      const TrDerived& d = ws.blk[bb].d;
      TrPrepBlock& q = pa_.blk[b];
      q.plog = P + o.plog; q.in_w = P + o.in_w; q.in_b = P + o.in_b; q.out_w = P + o.out_w; q.out_b = P + o.out_b;
      q.w1 = P + o.w1; q.w2 = P + o.w2;
      q.wi = ws.derived + d.wi; q.bi = ws.derived + d.bi; q.wo = ws.derived + d.wo; q.bo = ws.derived + d.bo; q.lam = ws.derived + d.lam;
      q.t = ws.blk[bb].t;
    }
    hipLaunchKernelGGL(tr_prep_kernel, dim3(64, lay.nb), dim3(256), 0, st, pa_, h->fused);
    LR_CHECK_LAUNCH("tr_prep_kernel");
  }
  if (h->fused) {   // embedding + LayerNorm + block 0's in_proj
    TbEmbedInProj e;
    e.ids = ids; e.E = P + lay.emb; e.ln_w = P + lay.eln_w; e.ln_b = P + lay.eln_b;
    e.wi = ws.blk[0].t.wiF; e.bi = ws.derived + ws.blk[0].d.bi;
    e.x = ws.x0; e.xhat = ws.xhat0; e.rstd = ws.rstd0; e.u = ws.blk[0].h;
    e.R = R; e.V = V; e.seed = seed; e.p_drop = pd;
    TR_RUN(tb_launch_embed_in_proj(e, st));
  } else {
    hipLaunchKernelGGL(tr_embed_ln_fwd, dim3(grid_rows), dim3(256), 0, st, ids, P + lay.emb, V, P + lay.eln_w,
                       P + lay.eln_b, ws.x0, ws.xhat0, ws.rstd0, R, seed, pd);
    LR_CHECK_LAUNCH("tr_embed_ln_fwd");
  }
  const float* x = ws.x0;
  for (int b = 0; b < lay.nb; ++b) {
    const TrBlockOff& o = lay.blk[b];
    TrBlockWs& W = ws.blk[b];
    const float* D = ws.derived;
    if (h->fused) {
      // (W.h already holds u = in_proj(x): written by the kernel in front -- the embedding kernel or the previous block's)
      hipLaunchKernelGGL(tr_scan_fwd, dim3(B), dim3(128), (size_t)L, st, W.h, ids, D + W.d.lam, L);
      LR_CHECK_LAUNCH("tr_scan_fwd");
      TbBlockFwd f;
      f.h = W.h; f.x = x;
      f.wo = W.t.woF; f.bo = D + W.d.bo; f.ln1_w = P + o.ln1_w; f.ln1_b = P + o.ln1_b;
      f.w1 = W.t.w1F; f.b1 = P + o.b1; f.w2 = W.t.w2F; f.b2 = P + o.b2; f.ln2_w = P + o.ln2_w; f.ln2_b = P + o.ln2_b;
      f.y = W.y; f.xhat1 = W.xhat1; f.rstd1 = W.rstd1; f.a = W.a; f.g = W.g; f.xout = W.xout; f.xhat2 = W.xhat2; f.rstd2 = W.rstd2;
      f.R = R; f.seed = seed; f.site0 = 10u + 4u * b; f.p_attn = pa; f.p_drop = pd;
      f.next_wi = f.next_bi = nullptr; f.next_u = nullptr;
      if (b + 1 < lay.nb) {
        f.next_wi = ws.blk[b + 1].t.wiF; f.next_bi = D + ws.blk[b + 1].d.bi; f.next_u = ws.blk[b + 1].h;
      }
      TR_RUN(tb_launch_block_fwd(f, st));
      x = W.xout;
      continue;
    }
    TR_RUN(tr_linear_fwd(x, D + W.d.wi, D + W.d.bi, W.h, R, 256, 64, st));
    hipLaunchKernelGGL(tr_scan_fwd, dim3(B), dim3(128), (size_t)L, st, W.h, ids, D + W.d.lam, L);
    LR_CHECK_LAUNCH("tr_scan_fwd");
    TR_RUN(tr_linear_fwd(W.h, D + W.d.wo, D + W.d.bo, ws.d64a, R, 64, 256, st));
    hipLaunchKernelGGL(tr_res_ln_fwd, dim3(grid_rows), dim3(256), 0, st, ws.d64a, x, P + o.ln1_w, P + o.ln1_b, W.y,
                       W.xhat1, W.rstd1, R, seed, 10u + 4u * b, pa);
    LR_CHECK_LAUNCH("tr_res_ln_fwd");
    TR_RUN(tr_linear_fwd(W.y, P + o.w1, P + o.b1, W.a, R, 256, 64, st));
    TR_EW(tr_gelu_fwd, (size_t)R * 256, W.a, W.g, (size_t)R * 256, seed, 11u + 4u * b, pd);
    TR_RUN(tr_linear_fwd(W.g, P + o.w2, P + o.b2, ws.d64a, R, 64, 256, st));
    hipLaunchKernelGGL(tr_res_ln_fwd, dim3(grid_rows), dim3(256), 0, st, ws.d64a, W.y, P + o.ln2_w, P + o.ln2_b, W.xout,
                       W.xhat2, W.rstd2, R, seed, 12u + 4u * b, pd);
    LR_CHECK_LAUNCH("tr_res_ln_fwd");
    x = W.xout;
  }
  // ---- item GEMM + cross-entropy, fused (lru_train_ce.hip): the [R x (V+1)] logits are never stored.
  // d x_final -> d64b, d table and d bias are added into the gradient buffer
  // (d x was zeroed and the labelled rows were counted by tr_pass_init_kernel: every variant below adds into d x)
  if (ws.materialise && h->fused) {
    TR_RUN(lr_launch_train_scores(x, P + lay.emb, P + lay.bias, lab, R, C, ws.ce, h->scal, dx, G + lay.emb, G + lay.bias, st));
  } else if (ws.materialise) {
    float* logits = ws.ce;
    const long long ldl = ((long long)C + 3) & ~3LL;   // padded row pitch (see the workspace carve)
    TR_RUN(tr_gemm(x, 64, 1, P + lay.emb, 1, 64, logits, ldl, P + lay.bias, R, C, 64, 0, st));  // scores (model/lru.py:85)
    hipLaunchKernelGGL(tr_ce_kernel, dim3(R), dim3(256), 0, st, logits, ldl, C, lab, h->scal);
    LR_CHECK_LAUNCH("tr_ce_kernel");
    TR_RUN(tr_gemm(logits, ldl, 1, P + lay.emb, 64, 1, dx, 64, nullptr, R, 64, C, 1, st, true));  // d x
    // d table += d logits^T x, and d bias += column sums of d logits (the row sums of the A operand)
    TR_RUN(tr_gemm(logits, 1, ldl, x, 64, 1, G + lay.emb, 64, nullptr, C, 64, R, 1, st, true, G + lay.bias));
  } else {
    TR_RUN(lr_launch_train_ce(x, P + lay.emb, P + lay.bias, lab, R, C, ws.ce, h->scal, dx, G + lay.emb, G + lay.bias, st));
  }
  // (loss = scal[0] / scal[1] is written to out_loss by the pass's last launch, the embedding LayerNorm's backward)
  if (det) {   // d x and the loss sum leave their shadows here: the first backward kernel reads d x, the last launch the loss
    fold(dx, sh_x, (size_t)R * 64, LR_DET_GRAD_SCALE);
    fold(h->scal, sh_s, 8, LR_DET_SCAL_SCALE);
    LR_CHECK_LAUNCH("lr_det_fold_kernel");
  }

  // ---- backward through the blocks; dx = gradient of the block's output
  for (int b = lay.nb - 1; b >= 0; --b) {
    const TrBlockOff& o = lay.blk[b];
    TrBlockWs& W = ws.blk[b];
    float* D = ws.derived;
    const float* xin = b ? ws.blk[b - 1].xout : ws.x0;
    if (h->fused) {
      TbBlockBwd q;
      q.dx = dx;
      q.xhat2 = W.xhat2; q.rstd2 = W.rstd2; q.ln2_w = P + o.ln2_w; q.a = W.a; q.xhat1 = W.xhat1; q.rstd1 = W.rstd1; q.ln1_w = P + o.ln1_w;
      q.w2T = W.t.w2T; q.w1T = W.t.w1T; q.woT = W.t.woT;
      q.dz0 = ws.d64a; q.da = ws.d256b; q.dy0 = ws.d64c; q.dh = ws.d256;
      q.dln2_w = G + o.ln2_w; q.dln2_b = G + o.ln2_b; q.dln1_w = G + o.ln1_w; q.dln1_b = G + o.ln1_b;
      q.R = R; q.seed = seed; q.site0 = 10u + 4u * b; q.p_attn = pa; q.p_drop = pd;
      TR_RUN(tb_launch_block_bwd(q, st));
      hipLaunchKernelGGL(tr_scan_bwd, dim3(B), dim3(128), (size_t)L, st, ws.d256, W.h, ids, D + W.d.lam, D + W.d.dlam, L);
      LR_CHECK_LAUNCH("tr_scan_bwd");
      TbWeightGrads wg;
      wg.P[0] = ws.d64a;  wg.Q[0] = W.g;  wg.dW[0] = G + o.w2;      wg.db[0] = G + o.b2;
      wg.P[1] = ws.d256b; wg.Q[1] = W.y;  wg.dW[1] = G + o.w1;      wg.db[1] = G + o.b1;
      wg.P[2] = ws.d64c;  wg.Q[2] = W.h;  wg.dW[2] = D + W.d.dwo;   wg.db[2] = D + W.d.dbo;
      wg.P[3] = ws.d256;  wg.Q[3] = xin;  wg.dW[3] = D + W.d.dwi;   wg.db[3] = D + W.d.dbi;
      wg.R = R;
      TR_RUN(tb_launch_bwd_tail(wg, ws.d256, W.t.wiT, dx, st));   // + dx_in += du Wi
      continue;
    }
    float* dz0 = ws.d64a;
    // LN2: dx -> dz0 (gradient of W2 g + b2 (dropped) + y)
    // dy (residual branch, written back into dx) = the LN gradient; the W2 branch sees dropout(that) in dz0
    hipLaunchKernelGGL(tr_ln_bwd, dim3((unsigned)((R + TR_LNB_ROWS - 1) / TR_LNB_ROWS)), dim3(256), 0, st, dx, W.xhat2, W.rstd2, P + o.ln2_w, dz0, G + o.ln2_w,
                       G + o.ln2_b, R, dx, seed, 12u + 4u * b, pd);
    LR_CHECK_LAUNCH("tr_ln_bwd");
    TR_RUN(tr_linear_bwd_weight(dz0, W.g, G + o.w2, G + o.b2, R, 64, 256, st));
    TR_RUN(tr_linear_bwd_data(dz0, P + o.w2, ws.d256, R, 64, 256, 0, st));  // d g
    TR_EW(tr_gelu_bwd, (size_t)R * 256, W.a, ws.d256, (size_t)R * 256, seed, 11u + 4u * b, pd);  // -> d a
    TR_RUN(tr_linear_bwd_weight(ws.d256, W.y, G + o.w1, G + o.b1, R, 256, 64, st));
    TR_RUN(tr_linear_bwd_data(ws.d256, P + o.w1, dx, R, 256, 64, 1, st));  // dy += da W1
    // LN1: dy -> dy0 (gradient of dropout(o) + x)
    float* dy0 = ws.d64a;
    hipLaunchKernelGGL(tr_ln_bwd, dim3((unsigned)((R + TR_LNB_ROWS - 1) / TR_LNB_ROWS)), dim3(256), 0, st, dx, W.xhat1, W.rstd1, P + o.ln1_w, dy0, G + o.ln1_w,
                       G + o.ln1_b, R, dx, seed, 10u + 4u * b, pa);   // residual: dx_in = the LN gradient + ...; branch: dropout(it)
    LR_CHECK_LAUNCH("tr_ln_bwd");
    // out_proj (derived real form [64][256] over (Re h | Im h))
    TR_RUN(tr_linear_bwd_weight(dy0, W.h, D + W.d.dwo, D + W.d.dbo, R, 64, 256, st));
    TR_RUN(tr_linear_bwd_data(dy0, D + W.d.wo, ws.d256, R, 64, 256, 0, st));  // g_t = direct gradient of h_t
    hipLaunchKernelGGL(tr_scan_bwd, dim3(B), dim3(128), (size_t)L, st, ws.d256, W.h, ids, D + W.d.lam, D + W.d.dlam, L);
    LR_CHECK_LAUNCH("tr_scan_bwd");
    // in_proj (derived, gamma folded): du in d256
    TR_RUN(tr_linear_bwd_weight(ws.d256, xin, D + W.d.dwi, D + W.d.dbi, R, 256, 64, st));
    TR_RUN(tr_linear_bwd_data(ws.d256, D + W.d.wi, dx, R, 256, 64, 1, st));  // dx_in += du Wi
  }
  if (det) {   // the derived weights' gradients (d lambda from the scans, d Wi / d Wo and their biases) in front of their reader
    fold(ws.derived, sh_d, ws.derived_floats, LR_DET_GRAD_SCALE);
    LR_CHECK_LAUNCH("lr_det_fold_kernel");
  }
  {  // gradients of the stored parameters from those of the derived weights, every block in one launch
    TrUnprepArgs ua;
    for (int b = 0; b < LR_MAX_LRU_BLOCKS; ++b) {
      const int bb = b < lay.nb ? b : 0;
      const TrBlockOff& o = lay.blk[bb];
      const TrDerived& d = ws.blk[bb].d;
      float* D = ws.derived;
      TrUnprepBlock& q = ua.blk[b];
      q.plog = P + o.plog; q.in_w = P + o.in_w; q.in_b = P + o.in_b; q.lam = D + d.lam;
      q.dwi = D + d.dwi; q.dbi = D + d.dbi; q.dwo = D + d.dwo; q.dbo = D + d.dbo; q.dlam = D + d.dlam;
      q.g_plog = G + o.plog; q.g_in_w = G + o.in_w; q.g_in_b = G + o.in_b; q.g_out_w = G + o.out_w; q.g_out_b = G + o.out_b;
    }
    hipLaunchKernelGGL(tr_unprep_kernel, dim3(128, lay.nb), dim3(64), 0, st, ua);
    LR_CHECK_LAUNCH("tr_unprep_kernel");
  }
  // ---- embedding LayerNorm and the lookup
  hipLaunchKernelGGL(tr_ln_bwd, dim3((unsigned)((R + TR_LNB_ROWS - 1) / TR_LNB_ROWS)), dim3(256), 0, st, dx, ws.xhat0, ws.rstd0, P + lay.eln_w, ws.d64a,
                     G + lay.eln_w, G + lay.eln_b, R, nullptr, seed, 0u, pd, ids, V, G + lay.emb, h->scal, out_loss);
  LR_CHECK_LAUNCH("tr_ln_bwd");
  if (det) {   // the pass's last launch in this mode: every parameter gradient that was summed over rows
    fold(G, sh_g, lay.total, LR_DET_GRAD_SCALE);
    LR_CHECK_LAUNCH("lr_det_fold_kernel");
  }
  (void)grid_rows;
  return LR_OK;
}

// Runs `enqueue(stream)` directly, or -- with graphs enabled and a capturable (non-default) stream -- captures
// it once into a hipGraph and replays that: the ~120 launches of a step become one graph launch.
template <typename F>
static int tr_run_or_replay(lr_lru_train* h, hipGraphExec_t* exec, bool key_ok, hipStream_t st, F enqueue) {
  if (!h->use_graph || st == nullptr) return enqueue(st);
  if (!*exec || !key_ok) {
    if (*exec) {
      (void)hipGraphExecDestroy(*exec);
      *exec = nullptr;
    }
    LR_CHECK_HIP(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
    const int rc = enqueue(st);
    hipGraph_t graph = nullptr;
    const hipError_t e = hipStreamEndCapture(st, &graph);
    if (rc) {
      if (graph) (void)hipGraphDestroy(graph);
      return rc;
    }
    if (e != hipSuccess || !graph) LR_FAIL(LR_EHIP, "hipStreamEndCapture: %s", hipGetErrorString(e));
    const hipError_t e2 = hipGraphInstantiate(exec, graph, nullptr, nullptr, 0);
    (void)hipGraphDestroy(graph);
    if (e2 != hipSuccess) {
      *exec = nullptr;
      LR_FAIL(LR_EHIP, "hipGraphInstantiate: %s", hipGetErrorString(e2));
    }
  }
  LR_CHECK_HIP(hipGraphLaunch(*exec, st));
  return LR_OK;
}

extern "C" int lr_lru_train_loss_grad(lr_lru_train_t* h, const int64_t* tokens, const int64_t* labels, int32_t B,
                                      int32_t L, float* out_loss, void* workspace, size_t workspace_bytes,
                                      void* hip_stream) {
  if (!h || !tokens || !labels || !out_loss || !workspace) LR_FAIL(LR_EINVAL, "lr_lru_train_loss_grad: null argument");
  if (B < 1 || L < 1) LR_FAIL(LR_EINVAL, "lr_lru_train_loss_grad: B=%d L=%d", B, L);
  if (h->deterministic) {   // the four translation units' maps name THIS pass's buffers (set outside any capture)
    const TrWs w = tr_carve(h->lay, h->cfg, B * L, (char*)workspace, true);
    if (w.total > workspace_bytes)
      LR_FAIL(LR_EWORKSPACE, "lr_lru_train_loss_grad: workspace needs %zu bytes in deterministic mode, have %zu", w.total, workspace_bytes);
    LrDetMap m = {};
    const size_t R_ = (size_t)B * L;
    m.base[0] = h->g;       m.bytes[0] = h->lay.total * 4;       m.shadow[0] = w.shadow;                                     m.scale[0] = (float)LR_DET_GRAD_SCALE;
    m.base[1] = w.derived;  m.bytes[1] = w.derived_floats * 4;   m.shadow[1] = w.shadow + h->lay.total;                      m.scale[1] = (float)LR_DET_GRAD_SCALE;
    m.base[2] = w.d64b;     m.bytes[2] = R_ * 64 * 4;            m.shadow[2] = m.shadow[1] + w.derived_floats;               m.scale[2] = (float)LR_DET_GRAD_SCALE;
    m.base[3] = h->scal;    m.bytes[3] = 8 * 4;                  m.shadow[3] = m.shadow[2] + R_ * 64;                        m.scale[3] = (float)LR_DET_SCAL_SCALE;
    m.on = 1;
    if (int rc = tr_det_publish(m, (hipStream_t)hip_stream)) return rc;
  } else if (g_det_current.on) {   // another engine of this process left its map behind: this pass adds in place
    LrDetMap off = {};
    LR_CHECK_HIP(hipDeviceSynchronize());
    if (int rc = tr_det_publish(off, (hipStream_t)hip_stream)) return rc;
  }
  const bool key_ok = h->k_tok == tokens && h->k_lab == labels && h->k_out == out_loss && h->k_ws == workspace &&
                      h->k_B == B && h->k_L == L;
  const int rc = tr_run_or_replay(h, &h->g_fb, key_ok, (hipStream_t)hip_stream, [&](hipStream_t st) {
    return tr_enqueue_loss_grad(h, tokens, labels, B, L, out_loss, workspace, workspace_bytes, st);
  });
  if (rc == LR_OK) {
    h->k_tok = tokens;
    h->k_lab = labels;
    h->k_out = out_loss;
    h->k_ws = workspace;
    h->k_B = B;
    h->k_L = L;
  }
  return rc;
}

extern "C" int lr_lru_train_apply(lr_lru_train_t* h, float lr, float max_grad_norm, float* out_grad_norm,
                                  void* hip_stream) {
  if (!h) LR_FAIL(LR_EINVAL, "lr_lru_train_apply: null handle");
  if (max_grad_norm <= 0.f) max_grad_norm = h->cfg.max_grad_norm;
  hipStream_t stream = (hipStream_t)hip_stream;
  // this step's learning rate and clipping limit go to device memory (the optimizer kernels may be a replayed
  // graph with frozen arguments); a 1-thread kernel carries them as launch arguments: no host buffer lifetime
  hipLaunchKernelGGL(tr_set_step_scalars, dim3(1), dim3(1), 0, stream, h->scal, lr, max_grad_norm);
  LR_CHECK_LAUNCH("tr_set_step_scalars");
  const bool key_ok = h->k_norm == out_grad_norm;
  const int rc = tr_run_or_replay(h, &h->g_opt, key_ok, stream, [&](hipStream_t st) {
    const size_t n = h->lay.total;
    const LrLruTrainConfig& c = h->cfg;
    hipLaunchKernelGGL(tr_zero_kernel, dim3(1), dim3(64), 0, st, h->scal + 2, (size_t)1);
    LR_CHECK_LAUNCH("tr_zero_kernel");
    if (h->deterministic) {   // partial sums stored and added in index order; the default combines 256 of them through an fp32 atomic
      hipLaunchKernelGGL(tr_sumsq_part_kernel, dim3(TR_SUMSQ_PARTS), dim3(256), 0, st, h->g, n, h->scal + 16, h->ctr);
      hipLaunchKernelGGL(tr_sumsq_final_kernel, dim3(1), dim3(64), 0, st, h->scal + 16, h->scal + 2);
    } else
    hipLaunchKernelGGL(tr_sumsq_kernel, dim3(256), dim3(256), 0, st, h->g, n, h->scal + 2, h->ctr);
    LR_CHECK_LAUNCH("tr_sumsq_kernel");
    hipLaunchKernelGGL(tr_adamw_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, h->p, h->g, h->m, h->v,
                       h->decay, n, h->scal, h->ctr, c.weight_decay, c.beta1, c.beta2, c.eps, out_grad_norm);
    LR_CHECK_LAUNCH("tr_adamw_kernel");
    return (int)LR_OK;
  });
  if (rc == LR_OK) h->k_norm = out_grad_norm;
  return rc;
}
