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
// One wave = 16 rows, workgroup = 64 rows; a K step of 64: lane (row li, group g) reads 32 contiguous bytes of its
// row (k0 + 16 g ..), so a row's 128-byte line is read by its 4 lanes and every activation byte is read once. The
// k -> (instruction, slot) assignment is the same for X and W, which is all a dot product needs.
template <int NT>
__global__ __launch_bounds__(256) void lt_skinny_kernel(const u16* __restrict__ X, int ldx, int n, int K,
                                                        const u16* __restrict__ W, u16* out, int ldo, int ocol,
                                                        float scale, uint32_t drop_stream, uint32_t drop_thresh,
                                                        float drop_scale) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int li = lane & 15, g = lane >> 4;
  const int row = blockIdx.x * 64 + wave * 16 + li;
  const int rr = min(row, n - 1);
  const u16* xp = X + (size_t)rr * ldx + g * 16;
  floatx4 acc[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) acc[t] = floatx4{0.f, 0.f, 0.f, 0.f};
  for (int k0 = 0; k0 < K; k0 += 64) {
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
  // accumulator: column = li (weight row), rows 4g + i
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int orow = blockIdx.x * 64 + wave * 16 + g * 4 + i;
      if (orow < n) out[(size_t)orow * ldo + ocol + t * 16 + li] = f2bf(acc[t][i] * scale);
    }
}
int lr_launch_skinny(const u16* X, int ldx, int n, int K, const u16* W, int nt, u16* out, int ldo, int ocol, float scale,
                     uint32_t drop_stream, float drop_p, hipStream_t st) {
  if (n < 1) return LR_OK;
  if (K % 64 != 0 || ldx % 8 != 0) LR_FAIL(LR_EUNSUPPORTED, "rank-r product: K=%d must be a multiple of 64", K);
  const uint32_t th = lt_drop_thresh(drop_p);
  const float ds = drop_p > 0.f ? 1.0f / (1.0f - drop_p) : 1.0f;
  const dim3 grid((n + 63) / 64);
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
__global__ __launch_bounds__(256) void lt_lora_rope_fwd_kernel(u16* qkv, int qw, int qcols, int kcols, int hd,
                                                               const u16* t /*[n][2*LT_RP]*/, const u16* bq_t,
                                                               const u16* bv_t, int r, float scaling,
                                                               const int32_t* tok_pos, const float* rope_cs) {
  __shared__ float ts[2 * LT_RP];
  const int row = blockIdx.x;
  if (threadIdx.x < 2 * LT_RP) ts[threadIdx.x] = bf2f(t[(size_t)row * 2 * LT_RP + threadIdx.x]);
  __syncthreads();
  u16* x = qkv + (size_t)row * qw;
  const int half = hd >> 1, vcols = qw - qcols - kcols;
  const float* cs = rope_cs + (size_t)tok_pos[row] * half * 2;
  for (int p = threadIdx.x; p < qw / 2; p += 256) {
    const int c = 2 * p;
    const u16x2 in = *reinterpret_cast<const u16x2*>(x + c);
    float v0 = bf2f(in[0]), v1 = bf2f(in[1]);
    if (c < qcols || c >= qcols + kcols) {
      const bool isq = c < qcols;
      const u16* bt = isq ? bq_t + c : bv_t + (c - qcols - kcols);
      const int ld = isq ? qcols : vcols;
      const float* tt = isq ? ts : ts + LT_RP;
      float l0 = 0.f, l1 = 0.f;
      for (int j = 0; j < r; ++j) {
        const u16x2 b = *reinterpret_cast<const u16x2*>(bt + (size_t)j * ld);
        l0 = __builtin_fmaf(tt[j], bf2f(b[0]), l0);
        l1 = __builtin_fmaf(tt[j], bf2f(b[1]), l1);
      }
      v0 = bf2f(f2bf(v0 + bf2f(f2bf(bf2f(f2bf(l0)) * scaling))));
      v1 = bf2f(f2bf(v1 + bf2f(f2bf(bf2f(f2bf(l1)) * scaling))));
    }
    if (c < qcols + kcols) {  // pair (x1_i, x2_i) of one head: rotate-half convention in the packed layout
      const int i = (c % hd) >> 1;
      const float co = cs[2 * i], si = cs[2 * i + 1];
      const float o0 = v0 * co - v1 * si, o1 = v1 * co + v0 * si;
      v0 = o0;
      v1 = o1;
    }
    u16x2 o;
    o[0] = f2bf(v0);
    o[1] = f2bf(v1);
    *reinterpret_cast<u16x2*>(x + c) = o;
  }
}
int lr_launch_lora_rope_fwd(u16* qkv, int n, int qw, int qcols, int kcols, int hd, const u16* t, const u16* bq_t,
                            const u16* bv_t, int r, float scaling, const int32_t* tok_pos, const float* rope_cs,
                            hipStream_t st) {
  if (n < 1) return LR_OK;
  hipLaunchKernelGGL(lt_lora_rope_fwd_kernel, dim3(n), dim3(256), 0, st, qkv, qw, qcols, kcols, hd, t, bq_t, bv_t, r,
                     scaling, tok_pos, rope_cs);
  LR_CHECK_LAUNCH("lt_lora_rope_fwd_kernel");
  return LR_OK;
}

// =============================================================================================
// backward of the same: d qkv (post-rotation) -> pre-rotation in place; d B_q, d B_v accumulated
// =============================================================================================
// A thread owns one column pair and walks a chunk of rows: the inverse rotation is pair-local, and
// d B[c][j] = scaling * sum_rows dq_pre[row][c] t[row][j] is a per-column running sum -- one atomic per (c, j)
// and row chunk. t rows are wave-uniform (scalar loads).
__global__ __launch_bounds__(256) void lt_rope_bwd_db_kernel(u16* dqkv, int n, int qw, int qcols, int kcols, int hd,
                                                             const u16* t, int r, float scaling,
                                                             const int32_t* tok_pos, const float* rope_cs, float* dbq,
                                                             float* dbv, int rows_per_wg) {
  const int p = blockIdx.x * 256 + threadIdx.x;
  const int c = 2 * p;
  if (c >= qw) return;
  const int r0 = blockIdx.y * rows_per_wg, r1 = min(n, r0 + rows_per_wg);
  const int half = hd >> 1;
  const bool isq = c < qcols, isk = !isq && c < qcols + kcols;
  const int i = (c % hd) >> 1;
  float a0[LT_RP], a1[LT_RP];
#pragma unroll
  for (int j = 0; j < LT_RP; ++j) a0[j] = a1[j] = 0.f;
  for (int row = r0; row < r1; ++row) {
    u16* x = dqkv + (size_t)row * qw + c;
    const u16x2 in = *reinterpret_cast<const u16x2*>(x);
    float v0 = bf2f(in[0]), v1 = bf2f(in[1]);
    if (isq || isk) {
      const float* cs = rope_cs + ((size_t)tok_pos[row] * half + i) * 2;
      const float co = cs[0], si = cs[1];
      const float o0 = v0 * co + v1 * si, o1 = v1 * co - v0 * si;  // transpose of the rotation
      u16x2 o;
      o[0] = f2bf(o0);
      o[1] = f2bf(o1);
      *reinterpret_cast<u16x2*>(x) = o;
      v0 = bf2f(o[0]);
      v1 = bf2f(o[1]);
    }
    if (!isk) {
      const u16* tr = t + (size_t)row * 2 * LT_RP + (isq ? 0 : LT_RP);
#pragma unroll
      for (int j = 0; j < LT_RP; ++j) {
        const float tj = bf2f(tr[j]);
        a0[j] = __builtin_fmaf(v0, tj, a0[j]);
        a1[j] = __builtin_fmaf(v1, tj, a1[j]);
      }
    }
  }
  if (isk) return;
  if (isq) {
    const int head = c / hd, within = c % hd;  // packed pair (2i, 2i+1) = original columns (i, half + i)
    const int o0 = head * hd + (within >> 1), o1 = o0 + half;
    for (int j = 0; j < r; ++j) {
      atomicAdd(dbq + (size_t)o0 * r + j, a0[j] * scaling);
      atomicAdd(dbq + (size_t)o1 * r + j, a1[j] * scaling);
    }
  } else {
    const int cv = c - qcols - kcols;
    for (int j = 0; j < r; ++j) {
      atomicAdd(dbv + (size_t)cv * r + j, a0[j] * scaling);
      atomicAdd(dbv + (size_t)(cv + 1) * r + j, a1[j] * scaling);
    }
  }
}
int lr_launch_rope_bwd_db(u16* dqkv, int n, int qw, int qcols, int kcols, int hd, const u16* t, int r, float scaling,
                          const int32_t* tok_pos, const float* rope_cs, float* dbq, float* dbv, hipStream_t st) {
  if (n < 1) return LR_OK;
  const int rows_per_wg = n >= 16384 ? 512 : (n >= 2048 ? 128 : 32);
  const dim3 grid((qw / 2 + 255) / 256, (n + rows_per_wg - 1) / rows_per_wg);
  hipLaunchKernelGGL(lt_rope_bwd_db_kernel, grid, dim3(256), 0, st, dqkv, n, qw, qcols, kcols, hd, t, r, scaling,
                     tok_pos, rope_cs, dbq, dbv, rows_per_wg);
  LR_CHECK_LAUNCH("lt_rope_bwd_db_kernel");
  return LR_OK;
}

// d A_q[j][c], d A_v[j][c] += sum_rows dt[row][j (+LT_RP)] * drop(xn)[row][c]   (dt already carries `scaling`)
__global__ __launch_bounds__(256) void lt_da_kernel(const u16* xn, int n, int d, const u16* dt, int r,
                                                    uint32_t drop_stream, uint32_t drop_thresh, float drop_scale,
                                                    float* daq, float* dav, int rows_per_wg) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= d) return;
  const int r0 = blockIdx.y * rows_per_wg, r1 = min(n, r0 + rows_per_wg);
  float a[2 * LT_RP];
#pragma unroll
  for (int j = 0; j < 2 * LT_RP; ++j) a[j] = 0.f;
  for (int row = r0; row < r1; ++row) {
    float x = bf2f(xn[(size_t)row * d + c]);
    if (drop_thresh) x = lt_keep(drop_stream, row, c, drop_thresh) ? bf2f(f2bf(x * drop_scale)) : 0.f;
    const u16* tr = dt + (size_t)row * 2 * LT_RP;
#pragma unroll
    for (int j = 0; j < 2 * LT_RP; ++j) a[j] = __builtin_fmaf(bf2f(tr[j]), x, a[j]);
  }
  for (int j = 0; j < r; ++j) {
    atomicAdd(daq + (size_t)j * d + c, a[j]);
    atomicAdd(dav + (size_t)j * d + c, a[LT_RP + j]);
  }
}
int lr_launch_lora_da(const u16* xn, int n, int d, const u16* dt, int r, uint32_t drop_stream, float drop_p, float* daq,
                      float* dav, hipStream_t st) {
  if (n < 1) return LR_OK;
  const int rows_per_wg = n >= 16384 ? 512 : (n >= 2048 ? 128 : 32);
  const dim3 grid((d + 255) / 256, (n + rows_per_wg - 1) / rows_per_wg);
  hipLaunchKernelGGL(lt_da_kernel, grid, dim3(256), 0, st, xn, n, d, dt, r, drop_stream, lt_drop_thresh(drop_p),
                     drop_p > 0.f ? 1.0f / (1.0f - drop_p) : 1.0f, daq, dav, rows_per_wg);
  LR_CHECK_LAUNCH("lt_da_kernel");
  return LR_OK;
}

// =============================================================================================
// SwiGLU on the interleaved gate/up layout of the packed wgu GEMM (16 gate columns, 16 up columns, ...)
// =============================================================================================
__global__ __launch_bounds__(256) void lt_swiglu_fwd_kernel(const u16* gu, u16* h, size_t total /* n*f */, int f) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const size_t row = i / f;
    const int c = (int)(i % f);
    const u16* p = gu + row * 2 * f + (c >> 4) * 32 + (c & 15);
    h[i] = swiglu_bf16(bf2f(p[0]), bf2f(p[16]));
  }
}
// in place: (gate, up) -> (d gate, d up) given d h
__global__ __launch_bounds__(256) void lt_swiglu_bwd_kernel(u16* gu, const u16* dh, size_t total, int f) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const size_t row = i / f;
    const int c = (int)(i % f);
    u16* p = gu + row * 2 * f + (c >> 4) * 32 + (c & 15);
    const float g = bf2f(p[0]), u = bf2f(p[16]), dy = bf2f(dh[i]);
    const float sg = 1.0f / (1.0f + __expf(-g));
    const float silu = g * sg;
    p[0] = f2bf(dy * u * (sg * (1.0f + g * (1.0f - sg))));
    p[16] = f2bf(dy * silu);
  }
}
int lr_launch_swiglu_fwd(const u16* gu, u16* h, int n, int f, hipStream_t st) {
  if (n < 1) return LR_OK;
  const size_t total = (size_t)n * f;
  hipLaunchKernelGGL(lt_swiglu_fwd_kernel, dim3((unsigned)min((size_t)8192, (total + 255) / 256)), dim3(256), 0, st, gu,
                     h, total, f);
  LR_CHECK_LAUNCH("lt_swiglu_fwd_kernel");
  return LR_OK;
}
int lr_launch_swiglu_bwd(u16* gu, const u16* dh, int n, int f, hipStream_t st) {
  if (n < 1) return LR_OK;
  const size_t total = (size_t)n * f;
  hipLaunchKernelGGL(lt_swiglu_bwd_kernel, dim3((unsigned)min((size_t)8192, (total + 255) / 256)), dim3(256), 0, st, gu,
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
#define LT_NORM_MAX_PER_THREAD 32
__global__ __launch_bounds__(256) void lt_rmsnorm_bwd_kernel(const u16* dy, const u16* x, const u16* w, const u16* res,
                                                             u16* out, int d, float eps, const int32_t* out_rows,
                                                             const u16* dt, const u16* a_cat, int r,
                                                             uint32_t drop_stream, uint32_t drop_thresh,
                                                             float drop_scale) {
  __shared__ float sh[4];
  __shared__ float ts[2 * LT_RP];
  const int row = blockIdx.x;
  const int orow = out_rows ? out_rows[row] : row;
  if (dt) {
    if (threadIdx.x < 2 * LT_RP) ts[threadIdx.x] = bf2f(dt[(size_t)row * 2 * LT_RP + threadIdx.x]);
    __syncthreads();
  }
  float xv[LT_NORM_MAX_PER_THREAD], gv[LT_NORM_MAX_PER_THREAD];
  float ss = 0.f, dot = 0.f;
#pragma unroll
  for (int i = 0; i < LT_NORM_MAX_PER_THREAD; ++i) {
    const int c = threadIdx.x + i * 256;
    xv[i] = gv[i] = 0.f;
    if (c < d) {
      xv[i] = bf2f(x[(size_t)row * d + c]);
      float g = bf2f(dy[(size_t)row * d + c]);
      if (dt && (!drop_thresh || lt_keep(drop_stream, row, c, drop_thresh))) {
        float l = 0.f;
        for (int j = 0; j < r; ++j) {
          l = __builtin_fmaf(ts[j], bf2f(a_cat[(size_t)j * d + c]), l);
          l = __builtin_fmaf(ts[LT_RP + j], bf2f(a_cat[(size_t)(LT_RP + j) * d + c]), l);
        }
        g = bf2f(f2bf(g + bf2f(f2bf(l * drop_scale))));
      }
      gv[i] = g * bf2f(w[c]);
      ss += xv[i] * xv[i];
      dot += xv[i] * gv[i];
    }
  }
  ss = lt_block_sum(ss, sh);
  dot = lt_block_sum(dot, sh);
  const float rstd = 1.0f / sqrtf(ss / (float)d + eps);
  const float coef = dot / (float)d * rstd * rstd * rstd;
#pragma unroll
  for (int i = 0; i < LT_NORM_MAX_PER_THREAD; ++i) {
    const int c = threadIdx.x + i * 256;
    if (c < d) {
      float v = rstd * gv[i] - xv[i] * coef;
      if (res) v = bf2f(f2bf(v)) + bf2f(res[(size_t)row * d + c]);
      out[(size_t)orow * d + c] = f2bf(v);
    }
  }
}
int lr_launch_rmsnorm_bwd(const u16* dy, const u16* x, const u16* w, const u16* res, u16* out, int rows, int d, float eps,
                          const int32_t* out_rows, const u16* dt, const u16* a_cat, int r, uint32_t drop_stream,
                          float drop_p, hipStream_t st) {
  if (rows < 1) return LR_OK;
  if (d > 256 * LT_NORM_MAX_PER_THREAD) LR_FAIL(LR_EUNSUPPORTED, "rmsnorm backward: hidden_size %d > %d", d,
                                                256 * LT_NORM_MAX_PER_THREAD);
  hipLaunchKernelGGL(lt_rmsnorm_bwd_kernel, dim3(rows), dim3(256), 0, st, dy, x, w, res, out, d, eps, out_rows, dt,
                     a_cat, r, drop_stream, lt_drop_thresh(drop_p), drop_p > 0.f ? 1.0f / (1.0f - drop_p) : 1.0f);
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
  const int item = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (item >= n_items) return;
  float s = 0.f;
  for (int c = lane; c < hd; c += 64) s += bf2f(o[(size_t)item * hd + c]) * bf2f(d_o[(size_t)item * hd + c]);
#pragma unroll
  for (int sft = 32; sft >= 1; sft >>= 1) s += __shfl_xor(s, sft, 64);
  if (lane == 0) out[item] = s;
}
int lr_launch_rowdot(const u16* o, const u16* d_o, int n, int nh, int hd, float* out, hipStream_t st) {
  if (n < 1) return LR_OK;
  hipLaunchKernelGGL(lt_rowdot_kernel, dim3((n * nh + 3) / 4), dim3(256), 0, st, o, d_o, n * nh, hd, out);
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
