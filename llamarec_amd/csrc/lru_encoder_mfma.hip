// lru_encoder_mfma.hip -- LRURec history encoder, batched over users as MFMA products (encoder variant 2).
//
// Same arithmetic as lru_encoder.hip / oracle/lr_oracle.c, bit for bit (k-ascending fmaf chains, butterfly
// LayerNorm sums, table GELU, sequential complex recurrence), reorganised for throughput:
//   * the LIVE tokens of a chunk of users are packed into one row matrix (row = off[user] + t); every
//     position-wise layer is then a [rows x K] x [K x N] product on v_mfma_f32_32x32x2_f32, 32 rows per tile;
//   * one MFMA step consumes two k values: lane-half 0 feeds k = 2s, half 1 feeds k = 2s + 1, and the hardware
//     adds them in that order (the chain of lr_item_score), so with operands stored de-interleaved
//     (x'[h*32 + s] = x[2s + h]) the accumulation IS the oracle's k-ascending chain, started from the bias;
//   * K = 64 layers (in_proj, FFN w_1) keep their weights in registers and put outputs on the lanes
//     (coalesced row stores); K = 256 layers (out_proj, FFN w_2) keep weights in LDS and put the 64 features of a
//     token on two lanes' registers, where the 64-lane butterfly of the LayerNorm becomes 35 in-lane adds and
//     one v_permlane32_swap per level-4 exchange -- the same association tree;
//   * the recurrence is one thread per (user, complex channel) over that user's rows;
//   * the last block's out_proj / FFN run on each user's last row only.
// Replaces (reference): model/lru.py:54-60,73-83,135-161,173-175 -- see lru_encoder.hip.
#include "lr_common.h"
#include "lr_profile.h"

typedef float floatx16 __attribute__((ext_vector_type(16)));

__device__ const float em_erf_tab[LR_ERF_NINT * (LR_ERF_DEG + 1)] = LR_ERF_TABLE;  // same table as lru_encoder.hip

#define EM_TILE 32
#define EM_XS 68  // LDS row stride (floats) of a 64-float row: conflict-free ds_read_b128

struct EmChunk {
  const int64_t* ids;  // [users][L] of this chunk
  int users, L, num_items;
  int* n;         // [users] live tokens
  int* off;       // [users + 1] first row of each user; off[users] = live rows of the chunk
  int* last_row;  // [users]
};

// ---- live length per user (positions after the last pad id among the first L-1), one wave per user --------
__global__ __launch_bounds__(256) void em_live_kernel(EmChunk c) {
  const int u = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (u >= c.users) return;
  const int64_t* ids = c.ids + (size_t)u * c.L;
  int loc = 0;
  for (int t = lane; t < c.L - 1; t += 64)
    if (ids[t] <= 0) loc = t + 1;
#pragma unroll
  for (int s = 32; s >= 1; s >>= 1) loc = max(loc, __shfl_xor(loc, s, 64));
  if (lane == 0) c.n[u] = c.L - loc;
}

// ---- exclusive prefix sum of n over the chunk's users (one workgroup; users <= a few thousand) -------------
__global__ __launch_bounds__(1024) void em_offsets_kernel(EmChunk c) {
  __shared__ int part[1024];
  const int tid = threadIdx.x;
  const int per = (c.users + 1023) / 1024;
  const int u0 = tid * per, u1 = min(c.users, u0 + per);
  int s = 0;
  for (int u = u0; u < u1; ++u) s += c.n[u];
  part[tid] = s;
  __syncthreads();
  for (int d = 1; d < 1024; d <<= 1) {
    const int v = tid >= d ? part[tid - d] : 0;
    __syncthreads();
    part[tid] += v;
    __syncthreads();
  }
  int run = part[tid] - s;
  for (int u = u0; u < u1; ++u) {
    c.off[u] = run;
    run += c.n[u];
    c.last_row[u] = run - 1;
  }
  if (tid == 1023) c.off[c.users] = part[1023];
}

// ---- the 64-lane butterfly sum of lru_encoder.hip, for the 64 features of a token held as v[blk][r] on a lane
// pair (feature f = blk*32 + (r&3) + 8*(r>>2) + 4*half): levels xor 32,16,8 and 2,1 are in-lane, xor 4 is the
// lane-half exchange.
__device__ __forceinline__ float em_swap32_add(float v) {
  const unsigned u = __builtin_bit_cast(unsigned, v);
  const auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
  const unsigned r0 = r[0], r1 = r[1];
  return __builtin_bit_cast(float, r0) + __builtin_bit_cast(float, r1);
}
__device__ __forceinline__ float em_butterfly64(const float (&v)[2][16]) {
  float a[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) a[r] = v[0][r] + v[1][r];  // xor 32
  float b[8];
#pragma unroll
  for (int r = 0; r < 8; ++r) b[r] = a[r] + a[r + 8];  // xor 16
  float c[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) c[r] = b[r] + b[r + 4];  // xor 8
#pragma unroll
  for (int r = 0; r < 4; ++r) c[r] = em_swap32_add(c[r]);  // xor 4
  const float e0 = c[0] + c[2], e1 = c[1] + c[3];  // xor 2
  return e0 + e1;                                   // xor 1
}

// ---- embedding gather + LayerNorm: workgroup = user, wave per token (identical to lru_encoder.hip) ---------
__device__ __forceinline__ float em_wave_sum64(float v) {
#pragma unroll
  for (int s = 32; s >= 1; s >>= 1) v = v + __shfl_xor(v, s, 64);
  return v;
}
__global__ __launch_bounds__(256) void em_embed_kernel(EmChunk c, const float* img, LrLruLayout lay, float* X) {
  const int u = blockIdx.x, wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int n = c.n[u], row0 = c.off[u], start = c.L - n;
  const float w = img[lay.emb_ln_w + lane], b = img[lay.emb_ln_b + lane];
  for (int t = wave; t < n; t += 4) {
    long long id = c.ids[(size_t)u * c.L + start + t];
    if (id < 0 || id > c.num_items) id = 0;
    const float e = img[lay.item_emb + (size_t)id * 64 + lane];
    const float mean = em_wave_sum64(e) * 0.015625f;
    const float d = e - mean;
    const float var = em_wave_sum64(d * d) * 0.015625f;
    const float rstd = 1.0f / sqrtf(var + LR_LN_EPS);
    X[(size_t)(row0 + t) * 64 + lane] = lr_fma(d * rstd, w, b);
  }
}

// ---- K = 64 layer: OUT[row][256] = epi( bias + sum_k W[k][.] x[row][k] ); EPI 0: * gamma (in_proj), 1: GELU ----
// Workgroup = 4 waves, wave w owns output blocks 2w, 2w+1 (32 outputs each) for every row tile it walks.
// wt: k-major [64][256] (the packed image). rows: optional gather list (last-row path), n_rows_ptr: live rows.
template <int EPI>
__global__ __launch_bounds__(256) void em_proj64_kernel(const float* __restrict__ wt, const float* __restrict__ bias,
                                                        const float* __restrict__ gamma, const float* __restrict__ X,
                                                        const int* __restrict__ rows, const int* n_rows_ptr, int n_rows_fixed,
                                                        float* OUT, int tiles_cap) {
  __shared__ __attribute__((aligned(16))) float xs[EM_TILE * EM_XS];
  __shared__ float erf_lds[LR_ERF_NINT * (LR_ERF_DEG + 1)];  // lane-divergent coefficient gathers: LDS, not global
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int half = lane >> 5, col = lane & 31;
  const int n_rows = n_rows_ptr ? *n_rows_ptr : n_rows_fixed;
  const int n_tiles = (n_rows + EM_TILE - 1) / EM_TILE;
  if ((int)blockIdx.x >= n_tiles) return;
  if (EPI == 1)
    for (int i = tid; i < LR_ERF_NINT * (LR_ERF_DEG + 1); i += 256) erf_lds[i] = em_erf_tab[i];
  // B operand: weights of my two output blocks, de-interleaved k: step s of half h uses k = 2s + h
  float wq[2][32], bq[2], gq[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int out = (2 * wave + j) * 32 + col;
#pragma unroll
    for (int s = 0; s < 32; ++s) wq[j][s] = wt[(size_t)(2 * s + half) * 256 + out];
    bq[j] = bias[out];
    gq[j] = EPI == 0 ? gamma[out & 127] : 0.f;
  }
  for (int tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
    const int r0 = tile * EM_TILE;
    __syncthreads();  // previous tile consumed
    // stage the tile de-interleaved: xs[row][h*32 + s] = x[row][2s + h]; thread -> (row = tid>>3, 8 floats)
    {
      const int row = tid >> 3, k0 = (tid & 7) * 8;
      const int gr = r0 + row;
      float v[8];
      if (gr < n_rows) {
        const float* src = X + (size_t)(rows ? rows[gr] : gr) * 64 + k0;
        const float4 a = *reinterpret_cast<const float4*>(src), b = *reinterpret_cast<const float4*>(src + 4);
        v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
      } else {
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = 0.f;
      }
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int k = k0 + i;
        xs[row * EM_XS + (k & 1) * 32 + (k >> 1)] = v[i];
      }
    }
    __syncthreads();
    // A operand: token row `col`, 32 de-interleaved values of my half
    float a[32];
    {
      const float* xr = xs + col * EM_XS + 32 * half;
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const float4 v4 = *reinterpret_cast<const float4*>(xr + 4 * q);
        a[4 * q + 0] = v4.x; a[4 * q + 1] = v4.y; a[4 * q + 2] = v4.z; a[4 * q + 3] = v4.w;
      }
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      floatx16 acc;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = bq[j];
#pragma unroll
      for (int s = 0; s < 32; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s], wq[j][s], acc, 0, 0, 0);
      // D[token i][out j]: lane = out column, register r = token (r&3) + 8*(r>>2) + 4*half
      const int out = (2 * wave + j) * 32 + col;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int gr = r0 + (r & 3) + 8 * (r >> 2) + 4 * half;
        if (gr < n_rows) {
          const float v = EPI == 0 ? acc[r] * gq[j] : lr_gelu_tab(acc[r], erf_lds);
          OUT[(size_t)gr * 256 + out] = v;
        }
      }
    }
  }
  (void)tiles_cap;
}

// ---- K = 256 layer: Y[row][64] = LayerNorm( bias + sum_k W[k][.] in[row][k] + RES[row][.] ) -------------------
// wt: k-major [256][64] (packed image). Workgroup = 4 waves, each wave its own 32-row tile; weights in LDS
// de-interleaved per 64-k chunk: ws[out][c*64 + h*32 + s] = W[64c + 2s + h][out]. A = weights (rows = features),
// B = tokens (columns): a lane pair holds the 64 features of one token.
#define EM_WS 260  // LDS row stride (floats) of a 256-float weight row
__global__ __launch_bounds__(256) void em_proj256_ln_kernel(const float* __restrict__ wt, const float* __restrict__ bias,
                                                            const float* __restrict__ lnw, const float* __restrict__ lnb,
                                                            const float* __restrict__ IN, const int* __restrict__ in_rows,
                                                            const float* __restrict__ RES, const int* __restrict__ res_rows,
                                                            const int* n_rows_ptr, int n_rows_fixed, float* Y) {
  extern __shared__ __attribute__((aligned(16))) float smem_f[];
  float* ws = smem_f;                       // [64][EM_WS]
  float* xin = smem_f + 64 * EM_WS;         // [4 waves][32 rows][EM_XS]: one 64-k chunk of the wave's tile
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int half = lane >> 5, col = lane & 31;
  const int n_rows = n_rows_ptr ? *n_rows_ptr : n_rows_fixed;
  const int n_tiles = (n_rows + EM_TILE - 1) / EM_TILE;
  if ((int)blockIdx.x * 4 >= n_tiles) return;
  for (int i = tid; i < 64 * 256; i += 256) {
    const int k = i >> 6, out = i & 63;  // coalesced read of W[k][out]
    ws[out * EM_WS + (k >> 6) * 64 + (k & 1) * 32 + ((k & 63) >> 1)] = wt[i];
  }
  float bq[2][16], wq[2][16], cq[2][16];  // bias, LN weight, LN bias of my 32 features
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int f = j * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
      bq[j][r] = bias[f];
      wq[j][r] = lnw[f];
      cq[j][r] = lnb[f];
    }
  __syncthreads();
  float* xw = xin + wave * (EM_TILE * EM_XS);
  for (int tile = blockIdx.x * 4 + wave; tile < n_tiles; tile += gridDim.x * 4) {
    const int r0 = tile * EM_TILE;
    floatx16 acc[2];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[j][r] = bq[j][r];
#pragma unroll 1
    for (int c = 0; c < 4; ++c) {
      // stage chunk c (k = 64c .. 64c+63) of my 32 input rows, de-interleaved; wave-private: no workgroup barrier
      __builtin_amdgcn_wave_barrier();
      for (int i = lane; i < EM_TILE * 16; i += 64) {
        const int row = i >> 4, k4 = (i & 15) * 4;
        const int gr = r0 + row;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (gr < n_rows) v = *reinterpret_cast<const float4*>(IN + (size_t)(in_rows ? in_rows[gr] : gr) * 256 + c * 64 + k4);
        float* dst = xw + row * EM_XS + (k4 >> 1);
        dst[0] = v.x;   // k4     (even -> half 0, s = k4/2)
        dst[32] = v.y;  // k4 + 1 (odd  -> half 1)
        dst[1] = v.z;   // k4 + 2
        dst[33] = v.w;  // k4 + 3
      }
      __builtin_amdgcn_wave_barrier();
      float b[32];  // token `col`, chunk c, my half
      {
        const float* xr = xw + col * EM_XS + 32 * half;
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          const float4 v4 = *reinterpret_cast<const float4*>(xr + 4 * q);
          b[4 * q + 0] = v4.x; b[4 * q + 1] = v4.y; b[4 * q + 2] = v4.z; b[4 * q + 3] = v4.w;
        }
      }
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const float* wr = ws + (j * 32 + col) * EM_WS + c * 64 + 32 * half;  // feature row j*32 + col
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          const float4 w4 = *reinterpret_cast<const float4*>(wr + 4 * q);
          acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(w4.x, b[4 * q + 0], acc[j], 0, 0, 0);
          acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(w4.y, b[4 * q + 1], acc[j], 0, 0, 0);
          acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(w4.z, b[4 * q + 2], acc[j], 0, 0, 0);
          acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(w4.w, b[4 * q + 3], acc[j], 0, 0, 0);
        }
      }
    }
    // D[feature i][token j]: lane = token `col`, acc[j][r] = feature j*32 + (r&3) + 8*(r>>2) + 4*half
    const int gr = r0 + col;
    const bool live = gr < n_rows;
    float v[2][16];
    {
      const float* res = RES + (size_t)(live ? (res_rows ? res_rows[gr] : gr) : 0) * 64;
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const float4 r4 = live ? *reinterpret_cast<const float4*>(res + j * 32 + 8 * g + 4 * half) : make_float4(0.f, 0.f, 0.f, 0.f);
          v[j][4 * g + 0] = acc[j][4 * g + 0] + r4.x;
          v[j][4 * g + 1] = acc[j][4 * g + 1] + r4.y;
          v[j][4 * g + 2] = acc[j][4 * g + 2] + r4.z;
          v[j][4 * g + 3] = acc[j][4 * g + 3] + r4.w;
        }
    }
    const float mean = em_butterfly64(v) * 0.015625f;
    float d2[2][16];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        v[j][r] = v[j][r] - mean;
        d2[j][r] = v[j][r] * v[j][r];
      }
    const float var = em_butterfly64(d2) * 0.015625f;
    const float rstd = 1.0f / sqrtf(var + LR_LN_EPS);
    if (live) {
      float* y = Y + (size_t)gr * 64;
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          float4 o;
          o.x = lr_fma(v[j][4 * g + 0] * rstd, wq[j][4 * g + 0], cq[j][4 * g + 0]);
          o.y = lr_fma(v[j][4 * g + 1] * rstd, wq[j][4 * g + 1], cq[j][4 * g + 1]);
          o.z = lr_fma(v[j][4 * g + 2] * rstd, wq[j][4 * g + 2], cq[j][4 * g + 2]);
          o.w = lr_fma(v[j][4 * g + 3] * rstd, wq[j][4 * g + 3], cq[j][4 * g + 3]);
          *reinterpret_cast<float4*>(y + j * 32 + 8 * g + 4 * half) = o;
        }
    }
  }
}

// ---- the recurrence over each user's rows, in place on U (re | im) -------------------------------------------
__global__ __launch_bounds__(128) void em_scan_kernel(EmChunk c, const float* lam_re, const float* lam_im, float* U) {
  const int u = blockIdx.x, ch = threadIdx.x;
  const int n = c.n[u];
  float* base = U + (size_t)c.off[u] * 256;
  const float lr_ = lam_re[ch], li = lam_im[ch];
  float h_r = 0.f, h_i = 0.f;
#pragma unroll 4
  for (int t = 0; t < n; ++t) {
    const float br = base[(size_t)t * 256 + ch], bi = base[(size_t)t * 256 + 128 + ch];
    if (t == 0) {
      h_r = br;
      h_i = bi;
    } else {
      const float nr = lr_fma(lr_, h_r, lr_fma(-li, h_i, br));
      const float ni = lr_fma(lr_, h_i, lr_fma(li, h_r, bi));
      h_r = nr;
      h_i = ni;
    }
    base[(size_t)t * 256 + ch] = h_r;
    base[(size_t)t * 256 + 128 + ch] = h_i;
  }
}

// =============================================================================================
#define EM_CHUNK_ROWS (1 << 21)  // rows per chunk of users (workspace: 1.5 KB per row, at most 3.2 GB; proportional to B*L below that)

static size_t em_chunk_users(int L) {
  size_t u = EM_CHUNK_ROWS / (size_t)L;
  return u < 1 ? 1 : u;
}

size_t lr_encoder_mfma_workspace_bytes(int B, int L) {
  const size_t users = (size_t)B < em_chunk_users(L) ? (size_t)B : em_chunk_users(L);
  const size_t rows = users * L;
  size_t o = 0;
  o += lr_align_up((2 * users + (users + 1)) * sizeof(int), 256);
  o += lr_align_up(rows * 64 * sizeof(float), 256) * 2;   // X, Y
  o += lr_align_up(rows * 256 * sizeof(float), 256);      // U / A
  o += lr_align_up(users * 64 * sizeof(float), 256);      // Y of the last rows
  o += lr_align_up(users * 256 * sizeof(float), 256);     // A of the last rows
  return o;
}

int lr_launch_lru_encode_mfma(const lr_lru* h, const int64_t* ids, int B, int L, float* out_q, void* ws,
                              size_t ws_bytes, hipStream_t st) {
  if (B <= 0) return LR_OK;
  if (ws_bytes < lr_encoder_mfma_workspace_bytes(B, L))
    LR_FAIL(LR_EWORKSPACE, "encoder: workspace needs %zu bytes, have %zu", lr_encoder_mfma_workspace_bytes(B, L), ws_bytes);
  const LrLruLayout& lay = h->lay;
  const float* img = h->img;
  const size_t cu = em_chunk_users(L);
  const size_t users_cap = (size_t)B < cu ? (size_t)B : cu;
  const size_t rows_cap = users_cap * L;
  char* p = (char*)ws;
  auto take = [&](size_t bytes) {
    char* at = p;
    p += lr_align_up(bytes, 256);
    return at;
  };
  int* ibuf = (int*)take((2 * users_cap + (users_cap + 1)) * sizeof(int));
  float* X = (float*)take(rows_cap * 64 * sizeof(float));
  float* Y = (float*)take(rows_cap * 64 * sizeof(float));
  float* U = (float*)take(rows_cap * 256 * sizeof(float));
  float* Yl = (float*)take(users_cap * 64 * sizeof(float));
  float* Al = (float*)take(users_cap * 256 * sizeof(float));
  static bool lds_set[LR_MAX_DEVICES] = {};
  const size_t lds256 = (size_t)(64 * EM_WS + 4 * EM_TILE * EM_XS) * sizeof(float);
  if (int rc = lr_ensure_dynamic_lds(reinterpret_cast<const void*>(em_proj256_ln_kernel), (int)lds256, lds_set)) return rc;
  LrProfScope prof(LR_PROF_LRU_ENCODE, (double)B, st);
  const int nb = lay.num_blocks;
  for (size_t u0 = 0; u0 < (size_t)B; u0 += cu) {
    EmChunk c;
    c.users = (int)(((size_t)B - u0) < cu ? ((size_t)B - u0) : cu);
    c.L = L;
    c.num_items = lay.num_items;
    c.ids = ids + u0 * L;
    c.n = ibuf;
    c.off = ibuf + users_cap;
    c.last_row = ibuf + users_cap + (users_cap + 1);
    const int* n_rows_ptr = c.off + c.users;
    const int max_tiles = (int)(((size_t)c.users * L + EM_TILE - 1) / EM_TILE);
    const int grid64 = max_tiles < 1024 ? max_tiles : 1024;
    const int grid256 = (max_tiles + 3) / 4 < 256 ? (max_tiles + 3) / 4 : 256;  // one workgroup per CU stages the 64 KiB of weights once
    const int last_tiles = (c.users + EM_TILE - 1) / EM_TILE;
    hipLaunchKernelGGL(em_live_kernel, dim3((c.users + 3) / 4), dim3(256), 0, st, c);
    LR_CHECK_LAUNCH("em_live_kernel");
    hipLaunchKernelGGL(em_offsets_kernel, dim3(1), dim3(1024), 0, st, c);
    LR_CHECK_LAUNCH("em_offsets_kernel");
    hipLaunchKernelGGL(em_embed_kernel, dim3(c.users), dim3(256), 0, st, c, img, lay, X);
    LR_CHECK_LAUNCH("em_embed_kernel");
    for (int b = 0; b < nb; ++b) {
      const LrLruBlockLayout& BL = lay.blk[b];
      const bool last = b == nb - 1;
      hipLaunchKernelGGL(em_proj64_kernel<0>, dim3(grid64), dim3(256), 0, st, img + BL.in_wt, img + BL.in_b, img + BL.gamma, X,
                         (const int*)nullptr, n_rows_ptr, 0, U, 0);
      LR_CHECK_LAUNCH("em_proj64_kernel<in_proj>");
      hipLaunchKernelGGL(em_scan_kernel, dim3(c.users), dim3(128), 0, st, c, img + BL.lam_re, img + BL.lam_im, U);
      LR_CHECK_LAUNCH("em_scan_kernel");
      if (!last) {
        hipLaunchKernelGGL(em_proj256_ln_kernel, dim3(grid256), dim3(256), lds256, st, img + BL.out_wt, img + BL.out_b,
                           img + BL.ln1_w, img + BL.ln1_b, U, (const int*)nullptr, X, (const int*)nullptr, n_rows_ptr, 0, Y);
        LR_CHECK_LAUNCH("em_proj256_ln_kernel<out_proj>");
        hipLaunchKernelGGL(em_proj64_kernel<1>, dim3(grid64), dim3(256), 0, st, img + BL.w1t, img + BL.b1, (const float*)nullptr,
                           Y, (const int*)nullptr, n_rows_ptr, 0, U, 0);
        LR_CHECK_LAUNCH("em_proj64_kernel<ffn1>");
        hipLaunchKernelGGL(em_proj256_ln_kernel, dim3(grid256), dim3(256), lds256, st, img + BL.w2t, img + BL.b2, img + BL.ln2_w,
                           img + BL.ln2_b, U, (const int*)nullptr, Y, (const int*)nullptr, n_rows_ptr, 0, X);
        LR_CHECK_LAUNCH("em_proj256_ln_kernel<ffn2>");
      } else {  // only each user's last row is consumed after the last block
        hipLaunchKernelGGL(em_proj256_ln_kernel, dim3((last_tiles + 3) / 4), dim3(256), lds256, st, img + BL.out_wt, img + BL.out_b,
                           img + BL.ln1_w, img + BL.ln1_b, U, (const int*)c.last_row, X, (const int*)c.last_row,
                           (const int*)nullptr, c.users, Yl);
        LR_CHECK_LAUNCH("em_proj256_ln_kernel<out_proj,last>");
        hipLaunchKernelGGL(em_proj64_kernel<1>, dim3(last_tiles), dim3(256), 0, st, img + BL.w1t, img + BL.b1,
                           (const float*)nullptr, Yl, (const int*)nullptr, (const int*)nullptr, c.users, Al, 0);
        LR_CHECK_LAUNCH("em_proj64_kernel<ffn1,last>");
        hipLaunchKernelGGL(em_proj256_ln_kernel, dim3((last_tiles + 3) / 4), dim3(256), lds256, st, img + BL.w2t, img + BL.b2,
                           img + BL.ln2_w, img + BL.ln2_b, Al, (const int*)nullptr, Yl, (const int*)nullptr,
                           (const int*)nullptr, c.users, out_q + u0 * 64);
        LR_CHECK_LAUNCH("em_proj256_ln_kernel<ffn2,last>");
      }
    }
  }
  return LR_OK;
}
