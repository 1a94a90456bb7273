// llama_train.hip -- the small kernels of the ranker's LoRA training step: everything between
// the big bf16 GEMMs (llama_gemm.hip) and the attention kernels (llama_attn.hip / llama_attn_bwd.hip).
//
// Replaces, on the reference side: peft 0.11.1's lora.Linear forward/backward on q_proj and v_proj
// (train_ranker.py:71-79; y = W x + (alpha/r) B A dropout(x)), torch autograd through transformers' RMSNorm,
// rotary embedding and SwiGLU (modeling_llama.py, reached from model/llm.py:89-100), the shifted cross-entropy of
// model/llm.py:116-127, clip_grad_norm_ and the optimizer step of HF Trainer (trainer/llm.py:103-136).
//
// All of these are HBM-bound row or column sweeps; none is reshaped into a GEMM except the two rank-r products
// (x A^T and dq B), which are 16-column MFMA tiles reading each activation row exactly once.
#include <stdlib.h>

#include "llama_train.h"

typedef unsigned short u16;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef u16 u16x2 __attribute__((ext_vector_type(2)));
typedef u16 u16x8 __attribute__((ext_vector_type(8)));

// ---- dropout: counter-based, the same mask wherever (stream, row, col) is asked for ------------------------
__device__ __forceinline__ uint32_t lt_mix(uint32_t h) {
  h ^= h >> 16;
  h *= 0x85ebca6bu;
  h ^= h >> 13;
  h *= 0xc2b2ae35u;
  h ^= h >> 16;
  return h;
}
// keep-probability test: true = element survives. `stream` already folds seed, pass counter and layer.
__device__ __forceinline__ bool lt_keep(uint32_t stream, uint32_t row, uint32_t col, uint32_t thresh) {
  return lt_mix(lt_mix(stream ^ (row * 0x9e3779b9u)) + col * 0x7f4a7c15u) >= thresh;
}
static uint32_t lt_drop_thresh(float p) {
  if (p <= 0.f) return 0u;
  double t = (double)p * 4294967296.0;
  return t >= 4294967295.0 ? 0xffffffffu : (uint32_t)t;
}

__device__ __forceinline__ float lt_block_sum(float v, float* sh) {
#pragma unroll
  for (int s = 32; s >= 1; s >>= 1) v += __shfl_xor(v, s, 64);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
  __syncthreads();
  return sh[0] + sh[1] + sh[2] + sh[3];
}
__device__ __forceinline__ float lt_block_max(float v, float* sh) {
#pragma unroll
  for (int s = 32; s >= 1; s >>= 1) v = fmaxf(v, __shfl_xor(v, s, 64));
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
  __syncthreads();
  return fmaxf(fmaxf(sh[0], sh[1]), fmaxf(sh[2], sh[3]));
}

// =============================================================================================
// bf16 transpose (setup): dst[c][r] = src[r][c]
// =============================================================================================
__global__ __launch_bounds__(256) void lt_transpose_kernel(const u16* src, int rows, int cols, u16* dst) {
  __shared__ u16 tile[64][66];
  const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
  for (int i = threadIdx.x; i < 64 * 64; i += 256) {
    const int r = i >> 6, c = i & 63;
    tile[r][c] = (r0 + r < rows && c0 + c < cols) ? src[(size_t)(r0 + r) * cols + c0 + c] : (u16)0;
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 64 * 64; i += 256) {
    const int c = i >> 6, r = i & 63;
    if (r0 + r < rows && c0 + c < cols) dst[(size_t)(c0 + c) * rows + r0 + r] = tile[r][c];
  }
}
int lr_launch_transpose_bf16(const u16* src, int rows, int cols, u16* dst, hipStream_t st) {
  if (rows < 1 || cols < 1) LR_FAIL(LR_EINVAL, "transpose: rows=%d cols=%d", rows, cols);
  hipLaunchKernelGGL(lt_transpose_kernel, dim3((cols + 63) / 64, (rows + 63) / 64), dim3(256), 0, st, src, rows, cols,
                     dst);
  LR_CHECK_LAUNCH("lt_transpose_kernel");
  return LR_OK;
}

// =============================================================================================
// LoRA working copies (bf16, zero-padded to LT_RP rows) of the fp32 masters, once per loss_grad call
// =============================================================================================
// a_cat [2*LT_RP][d]      rows 0..r-1 = A_q, LT_RP..LT_RP+r-1 = A_v
// bq_t  [LT_RP][nh*hd]    bq_t[j][c] = B_q[orig(c)][j], c in the PACKED (pair-interleaved) q column order
// bv_t  [LT_RP][nkv*hd]   bv_t[j][c] = B_v[c][j]
__global__ __launch_bounds__(256) void lt_prep_lora_kernel(const float* aq, const float* bq, const float* av,
                                                           const float* bv, int r, int d, int qcols, int vcols, int hd,
                                                           u16* a_cat, u16* bq_t, u16* bv_t) {
  const int na = 2 * LT_RP * d, nq = LT_RP * qcols, nv = LT_RP * vcols;
  const int half = hd >> 1;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < na + nq + nv; i += gridDim.x * 256) {
    if (i < na) {
      const int row = i / d, c = i % d;
      const int j = row % LT_RP;
      const float* src = row < LT_RP ? aq : av;
      a_cat[i] = j < r ? f2bf(src[(size_t)j * d + c]) : (u16)0;
    } else if (i < na + nq) {
      const int k = i - na, j = k / qcols, c = k % qcols;
      const int head = c / hd, within = c % hd;
      const int orig = head * hd + (within & 1) * half + (within >> 1);
      bq_t[k] = j < r ? f2bf(bq[(size_t)orig * r + j]) : (u16)0;
    } else {
      const int k = i - na - nq, j = k / vcols, c = k % vcols;
      bv_t[k] = j < r ? f2bf(bv[(size_t)c * r + j]) : (u16)0;
    }
  }
}
int lr_launch_prep_lora(const float* aq, const float* bq, const float* av, const float* bv, int r, int d, int qcols,
                        int vcols, int hd, u16* a_cat, u16* bq_t, u16* bv_t, hipStream_t st) {
  const int total = 2 * LT_RP * d + LT_RP * (qcols + vcols);
  hipLaunchKernelGGL(lt_prep_lora_kernel, dim3(min(1024, (total + 255) / 256)), dim3(256), 0, st, aq, bq, av, bv, r, d,
                     qcols, vcols, hd, a_cat, bq_t, bv_t);
  LR_CHECK_LAUNCH("lt_prep_lora_kernel");
  return LR_OK;
}

// =============================================================================================
// rank-r products: out[n][ldo] (NT 16-column tiles from column `ocol`) = scale * drop(X)[n][K] . W[16*NT][K]^T
// =============================================================================================
// A workgroup = 16 rows; its 4 waves take the K steps (64 wide) round-robin and add their partial tiles through LDS,
// so even a 7 k-token batch is ~450 workgroups of short loops. Lane (row li, group g) reads 32 contiguous bytes of its
// row (k0 + 16 g ..): a row's 128-byte line is read by its 4 lanes and every activation byte is read once. The
// k -> (instruction, slot) assignment is the same for X and W, which is all a dot product needs.
template <int NT>
__global__ __launch_bounds__(256) void lt_skinny_kernel(const u16* __restrict__ X, int ldx, int n, int K,
                                                        const u16* __restrict__ W, u16* out, int ldo, int ocol,
                                                        float scale, uint32_t drop_stream, uint32_t drop_thresh,
                                                        float drop_scale) {
  __shared__ floatx4 part[3][NT][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int li = lane & 15, g = lane >> 4;
  const int row = blockIdx.x * 16 + li;
  const int rr = min(row, n - 1);
  const u16* xp = X + (size_t)rr * ldx + g * 16;
  floatx4 acc[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) acc[t] = floatx4{0.f, 0.f, 0.f, 0.f};
  for (int k0 = wave * 64; k0 < K; k0 += 256) {
    u16x8 x0 = *reinterpret_cast<const u16x8*>(xp + k0);
    u16x8 x1 = *reinterpret_cast<const u16x8*>(xp + k0 + 8);
    if (drop_thresh) {
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int c = k0 + g * 16 + e;
        x0[e] = lt_keep(drop_stream, rr, c, drop_thresh) ? f2bf(bf2f(x0[e]) * drop_scale) : (u16)0;
        x1[e] = lt_keep(drop_stream, rr, c + 8, drop_thresh) ? f2bf(bf2f(x1[e]) * drop_scale) : (u16)0;
      }
    }
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const u16* wp = W + (size_t)(t * 16 + li) * K + k0 + g * 16;
      const u16x8 w0 = *reinterpret_cast<const u16x8*>(wp);
      const u16x8 w1 = *reinterpret_cast<const u16x8*>(wp + 8);
      // D[i][j] = sum_k A[i][k] B[k][j]: A row = lane&15 (token row), B column = lane&15 (weight row)
      acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, x0), __builtin_bit_cast(bf16x8, w0),
                                                       acc[t], 0, 0, 0);
      acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, x1), __builtin_bit_cast(bf16x8, w1),
                                                       acc[t], 0, 0, 0);
    }
  }
  if (wave > 0) {
#pragma unroll
    for (int t = 0; t < NT; ++t) part[wave - 1][t][lane] = acc[t];
  }
  __syncthreads();
  if (wave > 0) return;
  // accumulator: column = li (weight row), rows 4g + i
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const floatx4 a = acc[t] + part[0][t][lane] + part[1][t][lane] + part[2][t][lane];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int orow = blockIdx.x * 16 + g * 4 + i;
      if (orow < n) out[(size_t)orow * ldo + ocol + t * 16 + li] = f2bf(a[i] * scale);
    }
  }
}
int lr_launch_skinny(const u16* X, int ldx, int n, int K, const u16* W, int nt, u16* out, int ldo, int ocol, float scale,
                     uint32_t drop_stream, float drop_p, hipStream_t st) {
  if (n < 1) return LR_OK;
  if (K % 64 != 0 || ldx % 8 != 0) LR_FAIL(LR_EUNSUPPORTED, "rank-r product: K=%d must be a multiple of 64", K);
  const uint32_t th = lt_drop_thresh(drop_p);
  const float ds = drop_p > 0.f ? 1.0f / (1.0f - drop_p) : 1.0f;
  const dim3 grid((n + 15) / 16);
  if (nt == 1)
    hipLaunchKernelGGL(lt_skinny_kernel<1>, grid, dim3(256), 0, st, X, ldx, n, K, W, out, ldo, ocol, scale, drop_stream,
                       th, ds);
  else if (nt == 2)
    hipLaunchKernelGGL(lt_skinny_kernel<2>, grid, dim3(256), 0, st, X, ldx, n, K, W, out, ldo, ocol, scale, drop_stream,
                       th, ds);
  else
    LR_FAIL(LR_EINVAL, "rank-r product: %d column tiles", nt);
  LR_CHECK_LAUNCH("lt_skinny_kernel");
  return LR_OK;
}

// =============================================================================================
// forward: qkv (plain GEMM output, packed column order) += LoRA on q and v, then rotary embedding on q and k
// =============================================================================================
// peft: result = base(x) + lora_B(lora_A(drop(x))) * scaling, every Linear output a bf16 tensor under autocast.
// A workgroup sweeps LT_ROWS rows; a thread owns 8 consecutive columns (4 rotation pairs, one 16-byte access) of all of
// them, so the r rows of B^T it needs are fetched (from L2) once per LT_ROWS activation rows.
#define LT_ROWS 4
__global__ __launch_bounds__(256) void lt_lora_rope_fwd_kernel(u16* qkv, int n, int qw, int qcols, int kcols, int hd,
                                                               const u16* t /*[n][2*LT_RP]*/, const u16* bq_t,
                                                               const u16* bv_t, int r, float scaling,
                                                               const int32_t* tok_pos, const float* rope_cs) {
  __shared__ float ts[LT_ROWS][2 * LT_RP];
  __shared__ int pos[LT_ROWS];
  const int row0 = blockIdx.x * LT_ROWS;
  const int nrows = min(LT_ROWS, n - row0);
  if (threadIdx.x < LT_ROWS * 2 * LT_RP) {
    const int rr = threadIdx.x / (2 * LT_RP), j = threadIdx.x % (2 * LT_RP);
    ts[rr][j] = rr < nrows ? bf2f(t[(size_t)(row0 + rr) * 2 * LT_RP + j]) : 0.f;
  }
  if (threadIdx.x < LT_ROWS) pos[threadIdx.x] = threadIdx.x < nrows ? tok_pos[row0 + threadIdx.x] : 0;
  __syncthreads();
  const int half = hd >> 1, vcols = qw - qcols - kcols;
  for (int c = threadIdx.x * 8; c < qw; c += 256 * 8) {
    const bool isq = c < qcols, isv = c >= qcols + kcols;
    float v[LT_ROWS][8];
#pragma unroll
    for (int rr = 0; rr < LT_ROWS; ++rr) {
      const u16x8 in = *reinterpret_cast<const u16x8*>(qkv + (size_t)(row0 + min(rr, nrows - 1)) * qw + c);
#pragma unroll
      for (int e = 0; e < 8; ++e) v[rr][e] = bf2f(in[e]);
    }
    if (isq || isv) {
      const u16* bt = isq ? bq_t + c : bv_t + (c - qcols - kcols);
      const int ld = isq ? qcols : vcols, toff = isq ? 0 : LT_RP;
      float l[LT_ROWS][8];
#pragma unroll
      for (int rr = 0; rr < LT_ROWS; ++rr)
#pragma unroll
        for (int e = 0; e < 8; ++e) l[rr][e] = 0.f;
      for (int j = 0; j < r; ++j) {
        const u16x8 b = *reinterpret_cast<const u16x8*>(bt + (size_t)j * ld);
#pragma unroll
        for (int rr = 0; rr < LT_ROWS; ++rr) {
          const float tj = ts[rr][toff + j];
#pragma unroll
          for (int e = 0; e < 8; ++e) l[rr][e] = __builtin_fmaf(tj, bf2f(b[e]), l[rr][e]);
        }
      }
#pragma unroll
      for (int rr = 0; rr < LT_ROWS; ++rr)
#pragma unroll
        for (int e = 0; e < 8; ++e) v[rr][e] = bf2f(f2bf(v[rr][e] + bf2f(f2bf(bf2f(f2bf(l[rr][e])) * scaling))));
    }
    const int i0 = (c % hd) >> 1;
#pragma unroll
    for (int rr = 0; rr < LT_ROWS; ++rr) {
      if (rr >= nrows) continue;
      if (!isv) {  // pairs (x1_i, x2_i) of one head: rotate-half convention in the packed layout
        const float* cs = rope_cs + ((size_t)pos[rr] * half + i0) * 2;
        const float4 c01 = *reinterpret_cast<const float4*>(cs), c23 = *reinterpret_cast<const float4*>(cs + 4);
        const float co[4] = {c01.x, c01.z, c23.x, c23.z}, si[4] = {c01.y, c01.w, c23.y, c23.w};
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const float a0 = v[rr][2 * q], a1 = v[rr][2 * q + 1];
          v[rr][2 * q] = a0 * co[q] - a1 * si[q];
          v[rr][2 * q + 1] = a1 * co[q] + a0 * si[q];
        }
      }
      u16x8 o;
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] = f2bf(v[rr][e]);
      *reinterpret_cast<u16x8*>(qkv + (size_t)(row0 + rr) * qw + c) = o;
    }
  }
}
int lr_launch_lora_rope_fwd(u16* qkv, int n, int qw, int qcols, int kcols, int hd, const u16* t, const u16* bq_t,
                            const u16* bv_t, int r, float scaling, const int32_t* tok_pos, const float* rope_cs,
                            hipStream_t st) {
  if (n < 1) return LR_OK;
  if (hd % 8 != 0 || qcols % 8 != 0 || kcols % 8 != 0 || qw % 8 != 0)
    LR_FAIL(LR_EUNSUPPORTED, "lora + rotary sweep: head_dim and projection widths must be multiples of 8");
  hipLaunchKernelGGL(lt_lora_rope_fwd_kernel, dim3((n + LT_ROWS - 1) / LT_ROWS), dim3(256), 0, st, qkv, n, qw, qcols,
                     kcols, hd, t, bq_t, bv_t, r, scaling, tok_pos, rope_cs);
  LR_CHECK_LAUNCH("lt_lora_rope_fwd_kernel");
  return LR_OK;
}

// =============================================================================================
// backward of the rotary embedding (generic attention path only; the MFMA passes rotate in their epilogues)
// =============================================================================================
__global__ __launch_bounds__(256) void lt_rope_bwd_kernel(u16* dqkv, int n, int qw, int rot_cols, int hd,
                                                          const int32_t* tok_pos, const float* rope_cs) {
  const int per_row = rot_cols / 8;
  const int half = hd >> 1;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < (size_t)n * per_row; i += (size_t)gridDim.x * 256) {
    const int row = (int)(i / per_row), c = (int)(i % per_row) * 8;
    u16* x = dqkv + (size_t)row * qw + c;
    const u16x8 in = *reinterpret_cast<const u16x8*>(x);
    const float* cs = rope_cs + ((size_t)tok_pos[row] * half + ((c % hd) >> 1)) * 2;
    u16x8 o;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const float v0 = bf2f(in[2 * q]), v1 = bf2f(in[2 * q + 1]), co = cs[2 * q], si = cs[2 * q + 1];
      o[2 * q] = f2bf(v0 * co + v1 * si);  // transpose of the rotation
      o[2 * q + 1] = f2bf(v1 * co - v0 * si);
    }
    *reinterpret_cast<u16x8*>(x) = o;
  }
}
int lr_launch_rope_bwd(u16* dqkv, int n, int qw, int rot_cols, int hd, const int32_t* tok_pos, const float* rope_cs,
                       hipStream_t st) {
  if (n < 1) return LR_OK;
  const size_t total = (size_t)n * (rot_cols / 8);
  hipLaunchKernelGGL(lt_rope_bwd_kernel, dim3((unsigned)min((size_t)8192, (total + 255) / 256)), dim3(256), 0, st, dqkv,
                     n, qw, rot_cols, hd, tok_pos, rope_cs);
  LR_CHECK_LAUNCH("lt_rope_bwd_kernel");
  return LR_OK;
}

// =============================================================================================
// token reductions: out[j][c] += scale * sum_tok T[tok][tcol + j] * drop(X)[tok][c]      (d B, d A)
// =============================================================================================
// M = the adapter's rank (one or two 16-row tiles of T's columns), N = activation columns, K = tokens: both operands
// are indexed k-major in memory, so a lane gathers its 8 tokens with 2-byte loads (16 lanes = 32 contiguous bytes;
// T is a few hundred KB and stays in L2, X is read exactly once). A wave owns 64 columns (4 tiles sharing the T
// fragment), a workgroup 256, and walks one chunk of tokens; partial sums leave with one fp32 atomic per (j, c).
//   layout 0: out_t[j * cols + c]                          (d A: [r][hidden], tile t -> out0 / out1)
//   layout 1: out0[c * r + j]                              (d B_v: [cols][r])
//   layout 2: out0[orig(c) * r + j], packed -> HF column   (d B_q)
template <int NJ>
__global__ __launch_bounds__(256) void lt_tn_kernel(const u16* __restrict__ T, int ldt, int tcol,
                                                    const u16* __restrict__ X, int ldx, int n, int cols, int chunk,
                                                    float scale, float* out0, float* out1, int r, int layout, int hd,
                                                    uint32_t drop_stream, uint32_t drop_thresh, float drop_scale) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int li = lane & 15, g = lane >> 4;
  const int c0 = blockIdx.x * 256 + wave * 64;
  if (c0 >= cols) return;
  const int t0 = blockIdx.y * chunk, t1 = min(n, t0 + chunk);
  floatx4 acc[NJ][4];
#pragma unroll
  for (int a = 0; a < NJ; ++a)
#pragma unroll
    for (int ct = 0; ct < 4; ++ct) acc[a][ct] = floatx4{0.f, 0.f, 0.f, 0.f};
  for (int k0 = t0; k0 < t1; k0 += 32) {
    u16x8 tf[NJ], xf[4];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int tok = k0 + g * 8 + e;
      const bool ok = tok < t1;
      const size_t tr = (size_t)min(tok, n - 1);
#pragma unroll
      for (int a = 0; a < NJ; ++a) tf[a][e] = ok ? T[tr * ldt + tcol + a * 16 + li] : (u16)0;
#pragma unroll
      for (int ct = 0; ct < 4; ++ct) {
        const int c = c0 + ct * 16 + li;
        u16 v = X[tr * ldx + c];
        if (drop_thresh) v = lt_keep(drop_stream, (uint32_t)tr, c, drop_thresh) ? f2bf(bf2f(v) * drop_scale) : (u16)0;
        xf[ct][e] = v;
      }
    }
#pragma unroll
    for (int a = 0; a < NJ; ++a)
#pragma unroll
      for (int ct = 0; ct < 4; ++ct)
        acc[a][ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, tf[a]),
                                                             __builtin_bit_cast(bf16x8, xf[ct]), acc[a][ct], 0, 0, 0);
  }
  // accumulator: column = activation column li of the tile, rows j = 4g + i
  const int half = hd >> 1;
#pragma unroll
  for (int a = 0; a < NJ; ++a) {
    float* out = a == 0 ? out0 : out1;
#pragma unroll
    for (int ct = 0; ct < 4; ++ct) {
      const int c = c0 + ct * 16 + li;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int j = g * 4 + i;
        if (j >= r) continue;
        size_t at;
        if (layout == 0) {
          at = (size_t)j * cols + c;
        } else if (layout == 1) {
          at = (size_t)c * r + j;
        } else {
          const int head = c / hd, within = c % hd;
          at = (size_t)(head * hd + (within & 1) * half + (within >> 1)) * r + j;
        }
        atomicAdd(out + at, acc[a][ct][i] * scale);
      }
    }
  }
}
// The same reduction with both operands staged through LDS: 16-byte global loads in memory order (8 lanes per token row of the
// wave's 64 columns, 2 lanes per row of T's 16 columns), then ds_read_b64_tr_b16 hands every lane its 8 consecutive TOKENS of one
// column -- the k-major fragment the 2-byte gathers above assemble with 32 vector loads per 4 MFMAs (rocprofv3, round 5: 166 us
// per call at 7 k tokens against ~15 us of HBM time for X). Same fragments, same MFMA order: bit-identical sums. The next
// 32-token step's loads are in flight while this one's products run; the LDS tiles are wave-private (no barrier).
// Needs 16-byte-aligned rows of X and T (launch_tn checks and otherwise takes the kernel above).
template <int NJ>
__global__ __launch_bounds__(256) void lt_tn_lds_kernel(const u16* __restrict__ T, int ldt, int tcol,
                                                        const u16* __restrict__ X, int ldx, int n, int cols, int chunk,
                                                        float scale, float* out0, float* out1, int r, int layout, int hd,
                                                        uint32_t drop_stream, uint32_t drop_thresh, float drop_scale) {
  constexpr int XS = 144, TS = NJ * 32 + 16;                    // LDS row strides in bytes (16-byte multiples, not 128: banks)
  __shared__ __attribute__((aligned(16))) char lds[4 * (32 * XS + 32 * TS)];
  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
  typedef short s16x4 __attribute__((ext_vector_type(4)));
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int li = lane & 15, g = lane >> 4;
  const int c0 = blockIdx.x * 256 + wave * 64;
  if (c0 >= cols) return;                                       // whole waves leave: EXEC stays all ones for the transposed reads
  const int t0 = blockIdx.y * chunk, t1 = min(n, t0 + chunk);
  char* const xs = lds + wave * (32 * XS + 32 * TS);
  char* const ts = xs + 32 * XS;
  floatx4 acc[NJ][4];
#pragma unroll
  for (int a = 0; a < NJ; ++a)
#pragma unroll
    for (int ct = 0; ct < 4; ++ct) acc[a][ct] = floatx4{0.f, 0.f, 0.f, 0.f};
  u32x4 xr[4], tr_[NJ];
  auto fetch = [&](int k0) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const size_t row = (size_t)min(k0 + i * 8 + (lane >> 3), n - 1);
      xr[i] = *reinterpret_cast<const u32x4*>(X + row * ldx + c0 + (lane & 7) * 8);
    }
#pragma unroll
    for (int a = 0; a < NJ; ++a) {
      const size_t row = (size_t)min(k0 + (lane >> 1), n - 1);
      tr_[a] = *reinterpret_cast<const u32x4*>(T + row * ldt + tcol + a * 16 + (lane & 1) * 8);
    }
  };
  fetch(t0);
  for (int k0 = t0; k0 < t1; k0 += 32) {
    // this step's rows -> LDS (dropout on X and the zero rows of T past the chunk's end applied on the way)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      u32x4 v = xr[i];
      if (drop_thresh) {
        const uint32_t row = (uint32_t)min(k0 + i * 8 + (lane >> 3), n - 1);
#pragma unroll
        for (int w = 0; w < 4; ++w) {
          const int c = c0 + (lane & 7) * 8 + 2 * w;
          const u16 lo = lt_keep(drop_stream, row, c, drop_thresh) ? f2bf(bf2f((u16)(v[w] & 0xffff)) * drop_scale) : (u16)0;
          const u16 hi = lt_keep(drop_stream, row, c + 1, drop_thresh) ? f2bf(bf2f((u16)(v[w] >> 16)) * drop_scale) : (u16)0;
          v[w] = (unsigned)lo | ((unsigned)hi << 16);
        }
      }
      *reinterpret_cast<u32x4*>(xs + (i * 8 + (lane >> 3)) * XS + (lane & 7) * 16) = v;
    }
#pragma unroll
    for (int a = 0; a < NJ; ++a) {
      const bool ok = k0 + (lane >> 1) < t1;
      *reinterpret_cast<u32x4*>(ts + (lane >> 1) * TS + a * 32 + (lane & 1) * 16) = ok ? tr_[a] : u32x4{0, 0, 0, 0};
    }
    if (k0 + 32 < t1) fetch(k0 + 32);
    // lane 4q + p of a 16-lane group addresses row q, columns 4p .. 4p+3 of a 4 x 16 block; lane i receives column i of the 4 rows
    const int q = li >> 2, p = li & 3;
    u16x8 tf[NJ], xf[4];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int row = 8 * g + 4 * h + q;
#pragma unroll
      for (int a = 0; a < NJ; ++a) {
        const s16x4 t4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
            (__attribute__((address_space(3))) s16x4*)(ts + row * TS + (a * 16 + 4 * p) * 2));
#pragma unroll
        for (int e = 0; e < 4; ++e) tf[a][4 * h + e] = (u16)t4[e];
      }
#pragma unroll
      for (int ct = 0; ct < 4; ++ct) {
        const s16x4 x4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
            (__attribute__((address_space(3))) s16x4*)(xs + row * XS + (ct * 16 + 4 * p) * 2));
#pragma unroll
        for (int e = 0; e < 4; ++e) xf[ct][4 * h + e] = (u16)x4[e];
      }
    }
#pragma unroll
    for (int a = 0; a < NJ; ++a)
#pragma unroll
      for (int ct = 0; ct < 4; ++ct)
        acc[a][ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, tf[a]),
                                                             __builtin_bit_cast(bf16x8, xf[ct]), acc[a][ct], 0, 0, 0);
  }
  const int half = hd >> 1;
#pragma unroll
  for (int a = 0; a < NJ; ++a) {
    float* out = a == 0 ? out0 : out1;
#pragma unroll
    for (int ct = 0; ct < 4; ++ct) {
      const int c = c0 + ct * 16 + li;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int j = g * 4 + i;
        if (j >= r) continue;
        size_t at;
        if (layout == 0) {
          at = (size_t)j * cols + c;
        } else if (layout == 1) {
          at = (size_t)c * r + j;
        } else {
          const int head = c / hd, within = c % hd;
          at = (size_t)(head * hd + (within & 1) * half + (within >> 1)) * r + j;
        }
        atomicAdd(out + at, acc[a][ct][i] * scale);
      }
    }
  }
}
static int lt_tn_chunk(int n) {
  static int forced = -1;
  if (forced < 0) {
    const char* e = getenv("LR_TN_CHUNK");  // tuning knob: tokens per workgroup (multiple of 32)
    forced = e ? atoi(e) / 32 * 32 : 0;
  }
  if (forced > 0) return forced;
  return n >= 4096 ? 256 : (n >= 512 ? 128 : 32);  // 7 k tokens: 128-256 measured best (512: too few workgroups)
}
static int launch_tn(int nj, const u16* T, int ldt, int tcol, const u16* X, int ldx, int n, int cols, float scale,
                     float* out0, float* out1, int r, int layout, int hd, uint32_t drop_stream, float drop_p,
                     hipStream_t st) {
  if (n < 1) return LR_OK;
  if (cols % 64 != 0) LR_FAIL(LR_EUNSUPPORTED, "token reduction: %d columns (must be a multiple of 64)", cols);
  const int chunk = lt_tn_chunk(n);
  const dim3 grid((cols + 255) / 256, (n + chunk - 1) / chunk);
  const uint32_t th = lt_drop_thresh(drop_p);
  const float ds = drop_p > 0.f ? 1.0f / (1.0f - drop_p) : 1.0f;
  static int gather = -1;   // LR_TN_GATHER=1: the 2-byte gather kernel whatever the alignment (the bit-identity test's other side)
  if (gather < 0) {
    const char* e = getenv("LR_TN_GATHER");
    gather = e && e[0] == '1';
  }
  const bool aligned = !gather && ((uintptr_t)X & 15) == 0 && ((uintptr_t)(T + tcol) & 15) == 0 && ldx % 8 == 0 && ldt % 8 == 0;
  if (aligned && nj == 1)
    hipLaunchKernelGGL(lt_tn_lds_kernel<1>, grid, dim3(256), 0, st, T, ldt, tcol, X, ldx, n, cols, chunk, scale, out0, out1, r,
                       layout, hd, drop_stream, th, ds);
  else if (aligned)
    hipLaunchKernelGGL(lt_tn_lds_kernel<2>, grid, dim3(256), 0, st, T, ldt, tcol, X, ldx, n, cols, chunk, scale, out0, out1, r,
                       layout, hd, drop_stream, th, ds);
  else if (nj == 1)
    hipLaunchKernelGGL(lt_tn_kernel<1>, grid, dim3(256), 0, st, T, ldt, tcol, X, ldx, n, cols, chunk, scale, out0, out1, r,
                       layout, hd, drop_stream, th, ds);
  else
    hipLaunchKernelGGL(lt_tn_kernel<2>, grid, dim3(256), 0, st, T, ldt, tcol, X, ldx, n, cols, chunk, scale, out0, out1, r,
                       layout, hd, drop_stream, th, ds);
  LR_CHECK_LAUNCH("lt_tn_kernel");
  return LR_OK;
}
// d B_q += scaling * dq_pre^T t_q,  d B_v += scaling * dv^T t_v      (dqkv: gradient w.r.t. the UNROTATED q, k, v)
int lr_launch_lora_db(const u16* dqkv, int n, int qw, int qcols, int kcols, int hd, const u16* t, int r, float scaling,
                      float* dbq, float* dbv, hipStream_t st) {
  int rc = launch_tn(1, t, 2 * LT_RP, 0, dqkv, qw, n, qcols, scaling, dbq, nullptr, r, 2, hd, 0, 0.f, st);
  if (rc) return rc;
  return launch_tn(1, t, 2 * LT_RP, LT_RP, dqkv + qcols + kcols, qw, n, qw - qcols - kcols, scaling, dbv, nullptr, r, 1,
                   hd, 0, 0.f, st);
}
// d A_q[j][c], d A_v[j][c] += sum_rows dt[row][j (+LT_RP)] * drop(xn)[row][c]   (dt already carries `scaling`)
int lr_launch_lora_da(const u16* xn, int n, int d, const u16* dt, int r, uint32_t drop_stream, float drop_p, float* daq,
                      float* dav, hipStream_t st) {
  return launch_tn(2, dt, 2 * LT_RP, 0, xn, d, n, d, 1.0f, daq, dav, r, 0, 0, drop_stream, drop_p, st);
}

// =============================================================================================
// SwiGLU on the interleaved gate/up layout of the packed wgu GEMM (16 gate columns, 16 up columns, ...)
// =============================================================================================
// a thread owns 8 consecutive h columns = 16 bytes of gate, 16 of up, 16 of h
__global__ __launch_bounds__(256) void lt_swiglu_fwd_kernel(const u16* gu, u16* h, size_t total8 /* n*f/8 */, int f) {
  const int per_row = f >> 3;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total8; i += (size_t)gridDim.x * 256) {
    const size_t row = i / per_row;
    const int c = (int)(i % per_row) * 8;
    const u16* p = gu + row * 2 * f + (c >> 4) * 32 + (c & 15);
    const u16x8 g = *reinterpret_cast<const u16x8*>(p), u = *reinterpret_cast<const u16x8*>(p + 16);
    u16x8 o;
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = swiglu_bf16(bf2f(g[e]), bf2f(u[e]));
    *reinterpret_cast<u16x8*>(h + row * f + c) = o;
  }
}
// in place: (gate, up) -> (d gate, d up) given d h
__global__ __launch_bounds__(256) void lt_swiglu_bwd_kernel(u16* gu, const u16* dh, size_t total8, int f) {
  const int per_row = f >> 3;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total8; i += (size_t)gridDim.x * 256) {
    const size_t row = i / per_row;
    const int c = (int)(i % per_row) * 8;
    u16* p = gu + row * 2 * f + (c >> 4) * 32 + (c & 15);
    const u16x8 gv = *reinterpret_cast<const u16x8*>(p), uv = *reinterpret_cast<const u16x8*>(p + 16);
    const u16x8 dv = *reinterpret_cast<const u16x8*>(dh + row * f + c);
    u16x8 og, ou;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float g = bf2f(gv[e]), u = bf2f(uv[e]), dy = bf2f(dv[e]);
      const float sg = 1.0f / (1.0f + __expf(-g));
      og[e] = f2bf(dy * u * (sg * (1.0f + g * (1.0f - sg))));
      ou[e] = f2bf(dy * (g * sg));
    }
    *reinterpret_cast<u16x8*>(p) = og;
    *reinterpret_cast<u16x8*>(p + 16) = ou;
  }
}
int lr_launch_swiglu_fwd(const u16* gu, u16* h, int n, int f, hipStream_t st) {
  if (n < 1) return LR_OK;
  if (f % 16 != 0) LR_FAIL(LR_EUNSUPPORTED, "swiglu: intermediate size %d", f);
  const size_t total = (size_t)n * f / 8;
  hipLaunchKernelGGL(lt_swiglu_fwd_kernel, dim3((unsigned)min((size_t)16384, (total + 255) / 256)), dim3(256), 0, st, gu,
                     h, total, f);
  LR_CHECK_LAUNCH("lt_swiglu_fwd_kernel");
  return LR_OK;
}
int lr_launch_swiglu_bwd(u16* gu, const u16* dh, int n, int f, hipStream_t st) {
  if (n < 1) return LR_OK;
  if (f % 16 != 0) LR_FAIL(LR_EUNSUPPORTED, "swiglu: intermediate size %d", f);
  const size_t total = (size_t)n * f / 8;
  hipLaunchKernelGGL(lt_swiglu_bwd_kernel, dim3((unsigned)min((size_t)16384, (total + 255) / 256)), dim3(256), 0, st, gu,
                     dh, total, f);
  LR_CHECK_LAUNCH("lt_swiglu_bwd_kernel");
  return LR_OK;
}

// =============================================================================================
// RMSNorm backward (+ residual gradient, + the LoRA path's contribution to d xn)
// =============================================================================================
// y = w * x * rstd. With g = w .* dy:  dx = rstd * g - x * rstd^3 * mean(x .* g).
// out[out_row] = (res ? res[row] : 0) + dx.   dy may get the LoRA term first:
//   dy[c] += keep(row, c) * drop_scale * sum_j dt[row][j] * a_cat[j][c]     (j over both projections)
// A workgroup sweeps ROWS rows; a thread owns 8 consecutive columns (16-byte accesses) of up to LT_NORM_CHUNKS chunks
// of every row, kept in registers between the statistics pass and the output pass. With the LoRA term ROWS = 4
// (hidden <= 4096), so the 2r rows of A_cat a chunk needs come from L2 once per four activation rows.
#define LT_NORM_CHUNKS 4  // hidden <= 256 * 8 * 4 = 8192
template <int ROWS, int CHUNKS, bool LORA>
__global__ __launch_bounds__(256) void lt_rmsnorm_bwd_kernel(const u16* dy, const u16* x, const u16* w, const u16* res,
                                                             u16* out, int n, int d, float eps, const int32_t* out_rows,
                                                             const u16* dt, const u16* a_cat, int r,
                                                             uint32_t drop_stream, uint32_t drop_thresh,
                                                             float drop_scale) {
  __shared__ float sh[4][2 * ROWS];
  __shared__ float ts[ROWS][2 * LT_RP];
  const int row0 = blockIdx.x * ROWS;
  const int nrows = min(ROWS, n - row0);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (LORA) {
    if (threadIdx.x < ROWS * 2 * LT_RP) {
      const int rr = threadIdx.x / (2 * LT_RP), j = threadIdx.x % (2 * LT_RP);
      ts[rr][j] = rr < nrows ? bf2f(dt[(size_t)(row0 + rr) * 2 * LT_RP + j]) : 0.f;
    }
    __syncthreads();
  }
  u16x8 xv[ROWS][CHUNKS];
  float gv[ROWS][CHUNKS][8];
  float ss[ROWS], dot[ROWS];
#pragma unroll
  for (int rr = 0; rr < ROWS; ++rr) ss[rr] = dot[rr] = 0.f;
#pragma unroll
  for (int ch = 0; ch < CHUNKS; ++ch) {
    const int c = (threadIdx.x + ch * 256) * 8;
    if (c >= d) continue;
    const u16x8 wv = *reinterpret_cast<const u16x8*>(w + c);
    float g[ROWS][8];
#pragma unroll
    for (int rr = 0; rr < ROWS; ++rr) {
      const size_t at = (size_t)(row0 + min(rr, nrows - 1)) * d + c;
      xv[rr][ch] = *reinterpret_cast<const u16x8*>(x + at);
      const u16x8 dv = *reinterpret_cast<const u16x8*>(dy + at);
#pragma unroll
      for (int e = 0; e < 8; ++e) g[rr][e] = bf2f(dv[e]);
    }
    if (LORA) {
      float l[ROWS][8];
#pragma unroll
      for (int rr = 0; rr < ROWS; ++rr)
#pragma unroll
        for (int e = 0; e < 8; ++e) l[rr][e] = 0.f;
      for (int j = 0; j < r; ++j) {
        const u16x8 a0 = *reinterpret_cast<const u16x8*>(a_cat + (size_t)j * d + c);
        const u16x8 a1 = *reinterpret_cast<const u16x8*>(a_cat + (size_t)(LT_RP + j) * d + c);
#pragma unroll
        for (int rr = 0; rr < ROWS; ++rr) {
          const float t0 = ts[rr][j], t1 = ts[rr][LT_RP + j];
#pragma unroll
          for (int e = 0; e < 8; ++e)
            l[rr][e] = __builtin_fmaf(t1, bf2f(a1[e]), __builtin_fmaf(t0, bf2f(a0[e]), l[rr][e]));
        }
      }
#pragma unroll
      for (int rr = 0; rr < ROWS; ++rr)
#pragma unroll
        for (int e = 0; e < 8; ++e)
          if (!drop_thresh || lt_keep(drop_stream, row0 + min(rr, nrows - 1), c + e, drop_thresh))
            g[rr][e] = bf2f(f2bf(g[rr][e] + bf2f(f2bf(l[rr][e] * drop_scale))));
    }
#pragma unroll
    for (int rr = 0; rr < ROWS; ++rr)
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float xe = bf2f(xv[rr][ch][e]);
        gv[rr][ch][e] = g[rr][e] * bf2f(wv[e]);
        ss[rr] += xe * xe;
        dot[rr] += xe * gv[rr][ch][e];
      }
  }
#pragma unroll
  for (int rr = 0; rr < ROWS; ++rr) {
#pragma unroll
    for (int s_ = 32; s_ >= 1; s_ >>= 1) {
      ss[rr] += __shfl_xor(ss[rr], s_, 64);
      dot[rr] += __shfl_xor(dot[rr], s_, 64);
    }
    if (lane == 0) {
      sh[wave][2 * rr] = ss[rr];
      sh[wave][2 * rr + 1] = dot[rr];
    }
  }
  __syncthreads();
#pragma unroll
  for (int rr = 0; rr < ROWS; ++rr) {
    if (rr >= nrows) continue;
    const float s2 = sh[0][2 * rr] + sh[1][2 * rr] + sh[2][2 * rr] + sh[3][2 * rr];
    const float dt_ = sh[0][2 * rr + 1] + sh[1][2 * rr + 1] + sh[2][2 * rr + 1] + sh[3][2 * rr + 1];
    const float rstd = 1.0f / sqrtf(s2 / (float)d + eps);
    const float coef = dt_ / (float)d * rstd * rstd * rstd;
    const int orow = out_rows ? out_rows[row0 + rr] : row0 + rr;
#pragma unroll
    for (int ch = 0; ch < CHUNKS; ++ch) {
      const int c = (threadIdx.x + ch * 256) * 8;
      if (c >= d) continue;
      u16x8 rv;
      if (res) rv = *reinterpret_cast<const u16x8*>(res + (size_t)(row0 + rr) * d + c);
      u16x8 o;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        float v = rstd * gv[rr][ch][e] - bf2f(xv[rr][ch][e]) * coef;
        if (res) v = bf2f(f2bf(v)) + bf2f(rv[e]);
        o[e] = f2bf(v);
      }
      *reinterpret_cast<u16x8*>(out + (size_t)orow * d + c) = o;
    }
  }
}
int lr_launch_rmsnorm_bwd(const u16* dy, const u16* x, const u16* w, const u16* res, u16* out, int rows, int d, float eps,
                          const int32_t* out_rows, const u16* dt, const u16* a_cat, int r, uint32_t drop_stream,
                          float drop_p, hipStream_t st) {
  if (rows < 1) return LR_OK;
  if (d % 8 != 0 || d > 256 * 8 * LT_NORM_CHUNKS)
    LR_FAIL(LR_EUNSUPPORTED, "rmsnorm backward: hidden_size %d (multiple of 8, <= %d)", d, 256 * 8 * LT_NORM_CHUNKS);
  const uint32_t th = lt_drop_thresh(drop_p);
  const float ds = drop_p > 0.f ? 1.0f / (1.0f - drop_p) : 1.0f;
#define LT_NORM_LAUNCH(ROWS, CH, LORA)                                                                              \
  hipLaunchKernelGGL((lt_rmsnorm_bwd_kernel<ROWS, CH, LORA>), dim3((rows + ROWS - 1) / ROWS), dim3(256), 0, st, dy, x, w, \
                     res, out, rows, d, eps, out_rows, dt, a_cat, r, drop_stream, th, ds)
  const int chunks = (d + 2047) / 2048;
  if (dt) {
    if (chunks <= 1) LT_NORM_LAUNCH(4, 1, true);
    else if (chunks == 2) LT_NORM_LAUNCH(4, 2, true);
    else LT_NORM_LAUNCH(2, 4, true);
  } else {
    if (chunks <= 1) LT_NORM_LAUNCH(1, 1, false);
    else if (chunks == 2) LT_NORM_LAUNCH(1, 2, false);
    else LT_NORM_LAUNCH(1, 4, false);
  }
#undef LT_NORM_LAUNCH
  LR_CHECK_LAUNCH("lt_rmsnorm_bwd_kernel");
  return LR_OK;
}

// =============================================================================================
// loss head: softmax cross-entropy over bf16 logit rows, rewritten in place into d logits
// =============================================================================================
// model/llm.py:113-126: logits = lm_head(h).float(); CrossEntropyLoss() = mean over the labelled tokens.
// scal[0] += sum of -log p(target); d logits = (softmax - onehot) * gscale, gscale = grad_scale / n_labelled.
__global__ __launch_bounds__(256) void lt_ce_kernel(u16* logits, int V, const int32_t* targets, float gscale,
                                                    float* scal) {
  __shared__ float sh[4];
  const int row = blockIdx.x;
  u16* x = logits + (size_t)row * V;
  const int tgt = targets[row];
  if (tgt < 0 || tgt >= V) {  // not a token id: no loss, no gradient, counted
    for (int j = threadIdx.x; j < V; j += 256) x[j] = 0;
    if (threadIdx.x == 0) atomicAdd(scal + 1, 1.0f);
    return;
  }
  float mx = -__builtin_inff();
  for (int j = threadIdx.x; j < V; j += 256) mx = fmaxf(mx, bf2f(x[j]));
  mx = lt_block_max(mx, sh);
  float se = 0.f;
  for (int j = threadIdx.x; j < V; j += 256) se += __expf(bf2f(x[j]) - mx);
  se = lt_block_sum(se, sh);
  const float picked = bf2f(x[tgt]);
  __syncthreads();
  const float inv = 1.0f / se;
  for (int j = threadIdx.x; j < V; j += 256) {
    const float p = __expf(bf2f(x[j]) - mx) * inv;
    x[j] = f2bf((p - (j == tgt ? 1.0f : 0.0f)) * gscale);
  }
  if (threadIdx.x == 0) atomicAdd(scal, logf(se) + mx - picked);
}
int lr_launch_ce_bf16(u16* logits, int m, int V, const int32_t* targets, float gscale, float* scal, hipStream_t st) {
  if (m < 1) return LR_OK;
  hipLaunchKernelGGL(lt_ce_kernel, dim3(m), dim3(256), 0, st, logits, V, targets, gscale, scal);
  LR_CHECK_LAUNCH("lt_ce_kernel");
  return LR_OK;
}

// out[0] = loss sum / m (mean over the labelled tokens), out[1] = m, out[2] = targets outside the vocabulary
__global__ void lt_finish_loss_kernel(const float* scal, int m, float* out) {
  out[0] = m > 0 ? scal[0] / (float)m : 0.f;
  out[1] = (float)m;
  out[2] = scal[1];
}
int lr_launch_finish_loss(const float* scal, int m, float* out, hipStream_t st) {
  hipLaunchKernelGGL(lt_finish_loss_kernel, dim3(1), dim3(1), 0, st, scal, m, out);
  LR_CHECK_LAUNCH("lt_finish_loss_kernel");
  return LR_OK;
}

// D[row][h] = sum_d dO[row][h][d] * O[row][h][d]   (softmax backward's row term)
__global__ __launch_bounds__(256) void lt_rowdot_kernel(const u16* o, const u16* d_o, int n_items /* n*nh */, int hd,
                                                        float* out) {
  // 16 lanes x 8 elements cover up to 128 dims per step: 4 (token, head) items per wave
  const int lane = threadIdx.x & 63, sub = lane >> 4, l16 = lane & 15;
  const int item = (blockIdx.x * 4 + (threadIdx.x >> 6)) * 4 + sub;
  float s = 0.f;
  if (item < n_items)
    for (int c = l16 * 8; c < hd; c += 128) {
      const u16x8 a = *reinterpret_cast<const u16x8*>(o + (size_t)item * hd + c);
      const u16x8 b = *reinterpret_cast<const u16x8*>(d_o + (size_t)item * hd + c);
#pragma unroll
      for (int e = 0; e < 8; ++e) s = __builtin_fmaf(bf2f(a[e]), bf2f(b[e]), s);
    }
#pragma unroll
  for (int sft = 8; sft >= 1; sft >>= 1) s += __shfl_xor(s, sft, 64);
  if (l16 == 0 && item < n_items) out[item] = s;
}
int lr_launch_rowdot(const u16* o, const u16* d_o, int n, int nh, int hd, float* out, hipStream_t st) {
  if (n < 1) return LR_OK;
  if (hd % 8 != 0) LR_FAIL(LR_EUNSUPPORTED, "rowdot: head_dim %d", hd);
  hipLaunchKernelGGL(lt_rowdot_kernel, dim3((n * nh + 15) / 16), dim3(256), 0, st, o, d_o, n * nh, hd, out);
  LR_CHECK_LAUNCH("lt_rowdot_kernel");
  return LR_OK;
}

// =============================================================================================
// clipping + AdamW over the flat fp32 LoRA buffer (torch.optim.AdamW arithmetic, HF Trainer defaults)
// =============================================================================================
__global__ __launch_bounds__(256) void lt_sumsq_kernel(const float* g, size_t n, float* out) {
  __shared__ float sh[4];
  float s = 0.f;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) s += g[i] * g[i];
  s = lt_block_sum(s, sh);
  if (threadIdx.x == 0) atomicAdd(out, s);
}
// hyper[0] lr, hyper[1] max_grad_norm (<= 0: no clipping); ctr[0] = optimizer steps taken so far
__global__ __launch_bounds__(256) void lt_adamw_kernel(float* p, float* g, float* m, float* v, size_t n,
                                                       const float* sumsq, const float* hyper, const int* ctr,
                                                       float beta1, float beta2, float eps, float wd, float* out_norm) {
  const float norm = sqrtf(sumsq[0]);
  const float lr = hyper[0], limit = hyper[1];
  const float coef = limit > 0.f ? fminf(1.0f, limit / (norm + 1e-6f)) : 1.0f;  // torch clip_grad_norm_
  const int step = ctr[0] + 1;
  const float bc1 = 1.0f - powf(beta1, (float)step), bc2 = 1.0f - powf(beta2, (float)step);
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    const float gi = g[i] * coef;
    float pi = p[i] * (1.0f - lr * wd);
    const float mi = beta1 * m[i] + (1.0f - beta1) * gi;
    const float vi = beta2 * v[i] + (1.0f - beta2) * gi * gi;
    m[i] = mi;
    v[i] = vi;
    pi -= (lr / bc1) * mi / (sqrtf(vi) / sqrtf(bc2) + eps);
    p[i] = pi;
  }
  if (blockIdx.x == 0 && threadIdx.x == 0 && out_norm) out_norm[0] = norm;
}
__global__ void lt_set_hyper_kernel(float* hyper, float lr, float limit, float* sumsq) {
  hyper[0] = lr;
  hyper[1] = limit;
  sumsq[0] = 0.f;
}
__global__ void lt_bump_kernel(int* ctr) { ctr[0] += 1; }
int lr_launch_lora_adamw(float* p, float* g, float* m, float* v, size_t n, float* scratch /*[4]*/, int* ctr, float lr,
                         float max_grad_norm, float beta1, float beta2, float eps, float wd, float* out_norm,
                         hipStream_t st) {
  float* sumsq = scratch;
  float* hyper = scratch + 1;
  hipLaunchKernelGGL(lt_set_hyper_kernel, dim3(1), dim3(1), 0, st, hyper, lr, max_grad_norm, sumsq);
  const unsigned blocks = (unsigned)min((size_t)1024, (n + 255) / 256);
  hipLaunchKernelGGL(lt_sumsq_kernel, dim3(blocks), dim3(256), 0, st, g, n, sumsq);
  hipLaunchKernelGGL(lt_adamw_kernel, dim3(blocks), dim3(256), 0, st, p, g, m, v, n, sumsq, hyper, ctr, beta1, beta2,
                     eps, wd, out_norm);
  hipLaunchKernelGGL(lt_bump_kernel, dim3(1), dim3(1), 0, st, ctr);
  LR_CHECK_LAUNCH("lt_adamw_kernel");
  return LR_OK;
}

uint32_t lr_lora_drop_stream(uint64_t seed, uint32_t pass, uint32_t layer) {
  uint64_t z = seed + 0x9e3779b97f4a7c15ull * ((uint64_t)pass * 131u + layer + 1);
  z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
  z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
  return (uint32_t)(z ^ (z >> 31));
}
