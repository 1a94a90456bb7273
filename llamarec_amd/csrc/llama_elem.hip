// llama_elem.hip -- memory-bound pieces of the Llama prefill on gfx950: token-embedding gather,
// RMSNorm, the rotary cos/sin table, final norm + verbalizer GEMV. (RoPE itself and SwiGLU are
// fused into GEMM epilogues, llama_gemm.hip.)
//
// Replaces the ATen kernels reached from HF LlamaModel (transformers modeling_llama.py: RMSNorm
// with fp32 statistics, rotate_half RoPE, SiLU*mul) and, for the head, `lm_head` over ALL
// positions + `.float()` + `[:, -1]` (model/llm.py:113-114,131) followed by the verbalizer's
// column gather (trainer/verb.py:524-544): here only the LAST token of each prompt is normalised
// and only the C label-word rows of lm_head are multiplied.
//
// All activations are bf16 with fp32 arithmetic inside a kernel; 16-byte vector accesses.
#include "llama_kernels.h"

typedef unsigned short u16;
typedef u16 u16x8 __attribute__((ext_vector_type(8)));
typedef float floatx4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float block_sum_256(float v, float* red) {
#pragma unroll
  for (int s = 32; s >= 1; s >>= 1) v += __shfl_xor(v, s, 64);
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  __syncthreads();
  if (lane == 0) red[wave] = v;
  __syncthreads();
  return red[0] + red[1] + red[2] + red[3];
}

// ---- packed-layout metadata ---------------------------------------------------------------------
// Without a shared prefix (P = 0) the internal rows are the caller's packed tokens: segment b = prompt b.
// With P > 0 (every prompt starts with the same P tokens, run once): S = B + 1 segments,
//   segment 0      rows [0, P)                                     = the prefix, taken from prompt 0
//   segment b + 1  rows [P + cu[b] - b P, P + cu[b+1] - (b+1) P)   = prompt b's tokens at positions P..T_b-1
// tok_pos = position inside the prompt (rotary embedding), tok_src = index into the caller's packed ids,
// last_rows[b] = internal row of prompt b's last token, seg_start[0..S] = segment starts.
// ids / prefix_bad (shared prefix only): the caller PROMISES that every prompt starts with prompt 0's first P tokens; the
// promise is checked here -- a prompt whose first P ids differ sets *prefix_bad, and head_kernel then writes NaN scores
// for the whole call (prefix rows, K and V come from prompt 0: a stale prefix_len would otherwise score prompts 1..B-1
// silently wrong).
__global__ void token_meta_kernel(const int32_t* cu, int B, int P, int32_t* seg_start, int32_t* tok_pos,
                                  int32_t* tok_src, int32_t* last_rows, int32_t* last_pos, const int32_t* ids,
                                  int32_t* prefix_bad) {
  const int seg = blockIdx.x;
  int start, len, pos0, src0;
  if (P > 0 && seg == 0) {
    start = 0, len = P, pos0 = 0, src0 = cu[0];
  } else {
    const int b = P > 0 ? seg - 1 : seg;
    const int s = cu[b], e = cu[b + 1];
    if (P > 0 && b > 0 && ids && prefix_bad) {
      const int s0 = cu[0];
      bool bad = false;
      for (int i = threadIdx.x; i < P; i += blockDim.x) bad |= ids[s + i] != ids[s0 + i];
      if (bad) atomicOr(prefix_bad, 1);
    }
    start = P > 0 ? P + s - b * P : s;
    len = e - s - P;
    pos0 = P;
    src0 = s + P;
    if (threadIdx.x == 0) {
      if (last_rows) last_rows[b] = start + len - 1;
      if (last_pos) last_pos[b] = pos0 + len - 1;   // position of the prompt's last token (rotary embedding of its query)
      if (seg_start && seg == (int)gridDim.x - 1) seg_start[seg + 1] = start + len;
    }
  }
  if (threadIdx.x == 0 && seg_start) seg_start[seg] = start;
  for (int i = threadIdx.x; i < len; i += blockDim.x) {
    tok_pos[start + i] = pos0 + i;
    if (tok_src) tok_src[start + i] = src0 + i;
  }
}

// ---- embedding gather ------------------------------------------------------------------------
__global__ __launch_bounds__(256) void embed_kernel(const int32_t* ids, const int32_t* tok_src, const u16* table,
                                                    int vocab, int d, u16* out) {
  const int tok = blockIdx.x;
  int id = ids[tok_src ? tok_src[tok] : tok];
  if (id < 0 || id >= vocab) id = 0;
  const u16x8* src = reinterpret_cast<const u16x8*>(table + (size_t)id * d);
  u16x8* dst = reinterpret_cast<u16x8*>(out + (size_t)tok * d);
  for (int i = threadIdx.x; i < d / 8; i += 256) dst[i] = src[i];
}

// ---- RMSNorm: out = bf16( w * bf16( x * rsqrt(mean(x^2) + eps) ) ) --------------------------
// One workgroup per row. Rows of up to 8192 features stay in registers between the square sum and the scaling (each
// thread keeps its <= 4 pieces of 8, and requests the matching weights with them): the kernel is a chain of memory
// latencies per row, not bandwidth, and re-reading the row after the reduction added one more link. Same arithmetic,
// same order, as the two-pass form that longer rows still take.
#define RMS_KEEP 4
__global__ __launch_bounds__(256) void rmsnorm_kernel(const u16* x, const u16* w, u16* out, int d,
                                                      float eps, const int32_t* row_map) {
  __shared__ float red[4];
  const int row = row_map ? row_map[blockIdx.x] : blockIdx.x;
  const u16x8* xr = reinterpret_cast<const u16x8*>(x + (size_t)row * d);
  const u16x8* wr = reinterpret_cast<const u16x8*>(w);
  u16x8* orow = reinterpret_cast<u16x8*>(out + (size_t)blockIdx.x * d);
  const int nv = d / 8;
  const bool keep = nv <= 256 * RMS_KEEP;
  u16x8 vb[RMS_KEEP], wb[RMS_KEEP];
  float ss = 0.f;
  if (keep) {
#pragma unroll
    for (int k = 0; k < RMS_KEEP; ++k) {
      const int i = threadIdx.x + 256 * k;
      if (i < nv) {
        vb[k] = xr[i];
        wb[k] = wr[i];
      }
    }
#pragma unroll
    for (int k = 0; k < RMS_KEEP; ++k) {
      if (threadIdx.x + 256 * k < nv) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float f = bf2f(vb[k][j]);
          ss = __builtin_fmaf(f, f, ss);
        }
      }
    }
  } else {
    for (int i = threadIdx.x; i < nv; i += 256) {
      u16x8 v = xr[i];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        float f = bf2f(v[j]);
        ss = __builtin_fmaf(f, f, ss);
      }
    }
  }
  ss = block_sum_256(ss, red);
  const float rstd = 1.0f / sqrtf(ss / (float)d + eps);
  if (keep) {
#pragma unroll
    for (int k = 0; k < RMS_KEEP; ++k) {
      const int i = threadIdx.x + 256 * k;
      if (i < nv) {
        u16x8 o;
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = f2bf(bf2f(wb[k][j]) * bf2f(f2bf(bf2f(vb[k][j]) * rstd)));
        orow[i] = o;
      }
    }
  } else {
    for (int i = threadIdx.x; i < nv; i += 256) {
      u16x8 v = xr[i], wv = wr[i], o;
#pragma unroll
      for (int j = 0; j < 8; ++j) o[j] = f2bf(bf2f(wv[j]) * bf2f(f2bf(bf2f(v[j]) * rstd)));
      orow[i] = o;
    }
  }
}

// ---- split-K reduce + residual add + the RMSNorm that follows, one pass per row --------------------------------------
// The latency mode's o_proj / down_proj leave S fp32 partial planes (llama_gemm.hip, variant 5); summing them, adding the
// residual and storing the new residual row is one launch, the RMSNorm of that row for the next projection another: two
// dependent ~5-9 us launches per product at B = 1. Here a workgroup owns a row: thread t takes the chunks t + 256 k of 8
// columns, exactly rmsnorm_kernel's assignment, so the square sum is formed from the same values in the same order and the
// two outputs (residual row C, normalised row `out`) carry the same bits as the two-launch form (planes summed in split
// order, then bf16(bf16(sum) + residual) as in the GEMM's residual epilogue).
__global__ __launch_bounds__(256) void reduce_residual_rmsnorm_kernel(const float* __restrict__ part, int S, size_t plane,
                                                                      u16* C, const u16* R, int N, const u16* w, u16* out,
                                                                      float eps) {
  __shared__ float red[4];
  const int row = blockIdx.x;
  const int nv = N / 8;
  u16x8 vb[RMS_KEEP], wb[RMS_KEEP];
  float ss = 0.f;
#pragma unroll
  for (int k = 0; k < RMS_KEEP; ++k) {
    const int i = threadIdx.x + 256 * k;
    if (i < nv) {
      const size_t off = (size_t)row * N + (size_t)i * 8;
      floatx4 a = *reinterpret_cast<const floatx4*>(part + off), b = *reinterpret_cast<const floatx4*>(part + off + 4);
      for (int sp = 1; sp < S; ++sp) {
        a += *reinterpret_cast<const floatx4*>(part + sp * plane + off);
        b += *reinterpret_cast<const floatx4*>(part + sp * plane + off + 4);
      }
      const u16x8 r = *reinterpret_cast<const u16x8*>(R + off);
      u16x8 o;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        o[j] = f2bf(bf2f(f2bf(a[j])) + bf2f(r[j]));
        o[4 + j] = f2bf(bf2f(f2bf(b[j])) + bf2f(r[4 + j]));
      }
      *reinterpret_cast<u16x8*>(C + off) = o;
      vb[k] = o;
      wb[k] = reinterpret_cast<const u16x8*>(w)[i];
    }
  }
#pragma unroll
  for (int k = 0; k < RMS_KEEP; ++k) {
    if (threadIdx.x + 256 * k < nv) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float f = bf2f(vb[k][j]);
        ss = __builtin_fmaf(f, f, ss);
      }
    }
  }
  ss = block_sum_256(ss, red);
  const float rstd = 1.0f / sqrtf(ss / (float)N + eps);
#pragma unroll
  for (int k = 0; k < RMS_KEEP; ++k) {
    const int i = threadIdx.x + 256 * k;
    if (i < nv) {
      u16x8 o;
#pragma unroll
      for (int j = 0; j < 8; ++j) o[j] = f2bf(bf2f(wb[k][j]) * bf2f(f2bf(bf2f(vb[k][j]) * rstd)));
      reinterpret_cast<u16x8*>(out + (size_t)row * N)[i] = o;
    }
  }
}
// false: the row is too long to keep in registers (the caller then runs the two launches)
bool lr_reduce_residual_rmsnorm_fits(int N) { return N % 8 == 0 && N / 8 <= 256 * RMS_KEEP; }
int lr_launch_reduce_residual_rmsnorm(const float* part, int S, u16* C, const u16* R, int M, int N, const u16* norm_w,
                                      u16* norm_out, float eps, hipStream_t st) {
  if (!lr_reduce_residual_rmsnorm_fits(N)) LR_FAIL(LR_EUNSUPPORTED, "fused reduce + RMSNorm: %d columns", N);
  hipLaunchKernelGGL(reduce_residual_rmsnorm_kernel, dim3(M), dim3(256), 0, st, part, S, (size_t)M * N, C, R, N, norm_w,
                     norm_out, eps);
  LR_CHECK_LAUNCH("reduce_residual_rmsnorm_kernel");
  return LR_OK;
}

// ---- RMSNorm statistic only (the normalisation itself is folded into the next GEMM, llama_gemm.hip RopeArgs) ------
// one wave per row: 16-byte loads, fp32 square-sum in a fixed order (lane-strided, then the 6-step butterfly), so a
// row's rstd never depends on what else is in the batch
__global__ __launch_bounds__(256) void rms_rstd_kernel(const u16* x, float* rstd, int rows, int d, float eps) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= rows) return;
  const u16x8* xr = reinterpret_cast<const u16x8*>(x + (size_t)row * d);
  float ss = 0.f;
  for (int i = lane; i < d / 8; i += 64) {
    const u16x8 v = xr[i];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float f = bf2f(v[j]);
      ss = __builtin_fmaf(f, f, ss);
    }
  }
#pragma unroll
  for (int s = 32; s >= 1; s >>= 1) ss += __shfl_xor(ss, s, 64);
  if (lane == 0) rstd[row] = 1.0f / sqrtf(ss / (float)d + eps);
}

__global__ __launch_bounds__(256) void fold_norm_kernel(const u16* w, const u16* nw, u16* out, size_t n_vec, int cols_vec) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n_vec) return;
  const u16x8 a = reinterpret_cast<const u16x8*>(w)[i];
  const u16x8 b = reinterpret_cast<const u16x8*>(nw)[i % cols_vec];
  u16x8 o;
#pragma unroll
  for (int j = 0; j < 8; ++j) o[j] = f2bf(bf2f(a[j]) * bf2f(b[j]));
  reinterpret_cast<u16x8*>(out)[i] = o;
}

int lr_launch_rms_rstd(const u16* x, float* rstd, int rows, int d, float eps, hipStream_t st) {
  if (rows <= 0) return LR_OK;
  if (d % 8 != 0) LR_FAIL(LR_EINVAL, "rms_rstd: hidden size %d is not a multiple of 8", d);
  hipLaunchKernelGGL(rms_rstd_kernel, dim3((rows + 3) / 4), dim3(256), 0, st, x, rstd, rows, d, eps);
  LR_CHECK_LAUNCH("rms_rstd_kernel");
  return LR_OK;
}

int lr_launch_fold_norm(const u16* w, const u16* norm_w, u16* out, size_t rows, int cols, hipStream_t st) {
  if (rows == 0) return LR_OK;
  if (cols % 8 != 0) LR_FAIL(LR_EINVAL, "fold_norm: %d columns are not a multiple of 8", cols);
  const size_t n_vec = rows * (size_t)(cols / 8);
  hipLaunchKernelGGL(fold_norm_kernel, dim3((unsigned)((n_vec + 255) / 256)), dim3(256), 0, st, w, norm_w, out, n_vec,
                     cols / 8);
  LR_CHECK_LAUNCH("fold_norm_kernel");
  return LR_OK;
}

// ---- RoPE table: cos/sin(pos * theta^(-2i/hd)) rounded to bf16 like HF's bf16 rotary ---------
// cs16 (optional): the same values packed as bf16 pairs, cos in the low half-word and sin in the high one -- lossless (the
// fp32 table holds bf16-representable values); the 256-tile GEMM's rotary epilogue stages these 4-byte entries through LDS.
__global__ void rope_table_kernel(float* cs /*[T][hd/2][2]*/, unsigned* cs16 /*[T][hd/2]*/, int T, int hd, float theta) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  int half = hd / 2;
  if (i >= T * half) return;
  int pos = i / half, j = i % half;
  float inv = 1.0f / powf(theta, (float)(2 * j) / (float)hd);
  float ang = (float)pos * inv;
  const u16 c = f2bf(cosf(ang)), sn = f2bf(sinf(ang));
  cs[2 * i + 0] = bf2f(c);
  cs[2 * i + 1] = bf2f(sn);
  if (cs16) cs16[i] = (unsigned)c | ((unsigned)sn << 16);
}

// ---- final RMSNorm on each prompt's last token + dot with selected lm_head rows -------------
// grid (B, ceil(C/32)); out[b][c] = float(bf16(sum_k xn[k] * W[row_c][k])), row_c = ids ? ids[c] : c
__global__ __launch_bounds__(256) void head_kernel(const u16* x, const int32_t* rows, const u16* norm_w,
                                                   const u16* lm_head, const int32_t* class_ids, int C,
                                                   int d, float eps, float* out, int vocab, const int32_t* poison) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  u16* xn = reinterpret_cast<u16*>(smem_raw);  // [d]
  __shared__ float red[4];
  const int b = blockIdx.x;
  if (poison && *poison) {  // the call's shared-prefix promise was broken (token_meta_kernel): no score is trustworthy
    for (int cc = threadIdx.x; cc < 32; cc += 256) {
      const int c = blockIdx.y * 32 + cc;
      if (c < C) out[(size_t)b * C + c] = __builtin_nanf("");
    }
    return;
  }
  const int row = rows ? rows[b] : b;  // rows == nullptr: x is already compact, one row per prompt
  const u16* xr = x + (size_t)row * d;
  float ss = 0.f;
  for (int i = threadIdx.x; i < d; i += 256) {
    float f = bf2f(xr[i]);
    ss = __builtin_fmaf(f, f, ss);
  }
  ss = block_sum_256(ss, red);
  const float rstd = 1.0f / sqrtf(ss / (float)d + eps);
  for (int i = threadIdx.x; i < d; i += 256)
    xn[i] = f2bf(bf2f(norm_w[i]) * bf2f(f2bf(bf2f(xr[i]) * rstd)));
  __syncthreads();
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  for (int cc = wave; cc < 32; cc += 4) {
    int c = blockIdx.y * 32 + cc;
    if (c >= C) break;
    int r = class_ids ? class_ids[c] : c;
    if (r < 0 || r >= vocab) {  // a label word outside the vocabulary: NaN score instead of a wild read
      if (lane == 0) out[(size_t)b * C + c] = __builtin_nanf("");
      continue;
    }
    const u16* wr = lm_head + (size_t)r * d;
    float acc = 0.f;
    for (int k = lane * 8; k < d; k += 512) {
      u16x8 wv = *reinterpret_cast<const u16x8*>(wr + k);
      u16x8 xv = *reinterpret_cast<const u16x8*>(xn + k);
#pragma unroll
      for (int j = 0; j < 8; ++j) acc = __builtin_fmaf(bf2f(wv[j]), bf2f(xv[j]), acc);
    }
#pragma unroll
    for (int s = 32; s >= 1; s >>= 1) acc += __shfl_xor(acc, s, 64);
    if (lane == 0) out[(size_t)b * C + c] = bf2f(f2bf(acc));
  }
}

// ---- gather rows: out[i][:] = x[rows[i]][:] ----------------------------------------------------
__global__ __launch_bounds__(256) void gather_rows_kernel(const u16* x, const int32_t* rows, int d, u16* out) {
  const u16x8* src = reinterpret_cast<const u16x8*>(x + (size_t)rows[blockIdx.x] * d);
  u16x8* dst = reinterpret_cast<u16x8*>(out + (size_t)blockIdx.x * d);
  for (int i = threadIdx.x; i < d / 8; i += 256) dst[i] = src[i];
}

int lr_launch_gather_rows(const u16* x, const int32_t* rows, int n_rows, int d, u16* out, hipStream_t st) {
  if (n_rows <= 0) return LR_OK;
  hipLaunchKernelGGL(gather_rows_kernel, dim3(n_rows), dim3(256), 0, st, x, rows, d, out);
  LR_CHECK_LAUNCH("gather_rows_kernel");
  return LR_OK;
}

// ---------------------------------------------------------------------------------------------
int lr_launch_token_meta(const int32_t* cu, int B, int prefix_len, int32_t* seg_start, int32_t* tok_pos,
                         int32_t* tok_src, int32_t* last_rows, hipStream_t st, int32_t* last_pos, const int32_t* ids,
                         int32_t* prefix_bad) {
  const int S = prefix_len > 0 ? B + 1 : B;
  if (prefix_bad) LR_CHECK_HIP(hipMemsetAsync(prefix_bad, 0, sizeof(int32_t), st));
  hipLaunchKernelGGL(token_meta_kernel, dim3(S), dim3(256), 0, st, cu, B, prefix_len, seg_start, tok_pos, tok_src,
                     last_rows, last_pos, ids, prefix_bad);
  LR_CHECK_LAUNCH("token_meta_kernel");
  return LR_OK;
}

int lr_launch_embed(const int32_t* ids, const int32_t* tok_src, const u16* table, int vocab, int d, u16* out, int n,
                    hipStream_t st) {
  hipLaunchKernelGGL(embed_kernel, dim3(n), dim3(256), 0, st, ids, tok_src, table, vocab, d, out);
  LR_CHECK_LAUNCH("embed_kernel");
  return LR_OK;
}

int lr_launch_rmsnorm(const u16* x, const u16* w, u16* out, int rows, int d, float eps,
                      const int32_t* row_map, hipStream_t st) {
  hipLaunchKernelGGL(rmsnorm_kernel, dim3(rows), dim3(256), 0, st, x, w, out, d, eps, row_map);
  LR_CHECK_LAUNCH("rmsnorm_kernel");
  return LR_OK;
}

int lr_launch_rope_table(float* cs, int T, int hd, float theta, hipStream_t st, unsigned* cs16) {
  int n = T * (hd / 2);
  hipLaunchKernelGGL(rope_table_kernel, dim3((n + 255) / 256), dim3(256), 0, st, cs, cs16, T, hd, theta);
  LR_CHECK_LAUNCH("rope_table_kernel");
  return LR_OK;
}

int lr_launch_head(const u16* x, const int32_t* rows, const u16* norm_w, const u16* lm_head,
                   const int32_t* class_ids, int B, int C, int d, float eps, float* out, int vocab,
                   hipStream_t st, const int32_t* poison) {
  dim3 grid(B, (C + 31) / 32);
  hipLaunchKernelGGL(head_kernel, grid, dim3(256), (size_t)d * sizeof(u16), st, x, rows, norm_w, lm_head,
                     class_ids, C, d, eps, out, vocab, poison);
  LR_CHECK_LAUNCH("head_kernel");
  return LR_OK;
}
