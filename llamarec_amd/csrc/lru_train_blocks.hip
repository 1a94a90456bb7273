// lru_train_blocks.hip -- row-panel kernels of the retriever training step's LRU blocks (see lru_train_blocks.h).
//
// Why panels. A Beauty step is R = B L = 3 200 rows of 64 features; every product of a block multiplies those rows by
// one of four 64 x 256 weight matrices (0.1 GFLOP). As generic GEMM launches each of them was 50 workgroups walking a
// K loop for 17 us, 25 of them per pass with 20 elementwise launches in between: 0.6 of the 1.0 ms step
// (profiles/r04_train_beauty_summary.txt). Here a workgroup owns 16 rows end to end: the panel's activations live in
// LDS, the weights (64 KB per matrix, shared by all workgroups) come from L2 straight into MFMA operand registers, and
// the elementwise steps between two products run on the panel in LDS. 200 workgroups of 4 waves for Beauty.
//
// Arithmetic: v_mfma_f32_16x16x4_f32 (exact fp32 products, fp32 accumulation), operands swapped so that a lane owns ONE
// row (lane & 15) and 4 CONSECUTIVE output columns 4 (lane >> 4) .. + 3 of a 16 x 16 block. The K index is permuted
// inside groups of 16: lane group g = lane >> 4 reads the float4 at k = 16 j + 4 g of BOTH operands and feeds element e
// of it to MFMA (j, e) -- every k meets its own partner, the sum just runs in another order (fp32 sums are compared with
// the float64 oracle and the reference's autograd at 3e-4 / 5e-4 relative, tests/test_gpu_lru_train.py).
#include "lru_train_blocks.h"
#include "lr_det.h"
LR_DET_DEFINE(blocks)

typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef float floatx16 __attribute__((ext_vector_type(16)));

#define TB_ROWS 16
#define TB_LD64 68     // LDS row pitch (floats) of a [16][64] panel: 16 rows x 4 banks = all 64 banks per 16-lane b128 read
#define TB_LD256 260

// wf[nb][j] = W[n0 + 16 nb + li][16 j + 4 g .. + 3] of a matrix W[N][K] stored in FRAGMENT ORDER by tr_prep_kernel
// (lru_train.hip): [n / 16][k / 16][lane = 16 g + li][4] -- the 64 lanes of one operand load read 1 KB contiguous. (Read
// from the row-major matrix the same fragment is 16 rows x 64 B per wave-instruction, the shape a CU moves at a third of
// the rate: lru_train_scores.hip.)
template <int K, int NB>
__device__ __forceinline__ void tb_load_w(const float* __restrict__ Wf, int n0, int lane, float4 (&wf)[NB][K / 16]) {
#pragma unroll
  for (int nb = 0; nb < NB; ++nb)
#pragma unroll
    for (int j = 0; j < K / 16; ++j)
      wf[nb][j] = *reinterpret_cast<const float4*>(Wf + ((size_t)((n0 >> 4) + nb) * (K / 16) + j) * 256 + lane * 4);
}
__device__ __forceinline__ float tb_e(const float4& v, int e) { return e == 0 ? v.x : e == 1 ? v.y : e == 2 ? v.z : v.w; }

// 4 column blocks, K = 64: acc[nb] = D[n = 16 nb + 4 g + r][row = li]; xrow = &X[li][4 g] (LDS)
__device__ __forceinline__ void tb_mma_n4(const float* xrow, const float4 (&wf)[4][4], floatx4 (&acc)[4]) {
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const float4 xa = *reinterpret_cast<const float4*>(xrow + 16 * j);
#pragma unroll
    for (int e = 0; e < 4; ++e)
#pragma unroll
      for (int nb = 0; nb < 4; ++nb)
        acc[nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(tb_e(wf[nb][j], e), tb_e(xa, e), acc[nb], 0, 0, 0);
  }
}
// 1 column block, K = 256: four partial sums (one per element e) keep dependent MFMAs four issues apart
__device__ __forceinline__ floatx4 tb_mma_n1(const float* xrow, const float4 (&wf)[1][16]) {
  floatx4 part[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) part[e] = floatx4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    const float4 xa = *reinterpret_cast<const float4*>(xrow + 16 * j);
#pragma unroll
    for (int e = 0; e < 4; ++e)
      part[e] = __builtin_amdgcn_mfma_f32_16x16x4f32(tb_e(wf[0][j], e), tb_e(xa, e), part[e], 0, 0, 0);
  }
  return (part[0] + part[1]) + (part[2] + part[3]);
}
// sum over the 16 lanes that share a row in the (row = tid >> 4, 4 columns at 4 (tid & 15)) layout
__device__ __forceinline__ float tb_row_sum(float v) {
#pragma unroll
  for (int s = 8; s >= 1; s >>= 1) v += __shfl_xor(v, s, 64);
  return v;
}
__device__ __forceinline__ float4 tb_ld4(const float* p, bool ok) {
  return ok ? *reinterpret_cast<const float4*>(p) : make_float4(0.f, 0.f, 0.f, 0.f);
}

// =============================================================================================
// in_proj forward: u = x wi^T + bi -- inside the kernel that produces x
// =============================================================================================
// Block 0's in_proj with the embedding lookup + dropout + LayerNorm in front of it (one launch instead of two): a wave per
// row for the LayerNorm (the arithmetic of lru_train.hip's tr_embed_ln_fwd), the 16 rows meet in LDS for the MFMAs.
__global__ __launch_bounds__(256) void tb_embed_in_proj_kernel(TbEmbedInProj p) {
  __shared__ __attribute__((aligned(16))) float Xs[TB_ROWS][TB_LD64];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 15, g = lane >> 4;
  const int row0 = blockIdx.x * TB_ROWS;
  const unsigned long long seed = *p.seed;
  float4 wf[4][4];
  tb_load_w<64, 4>(p.wi, 64 * wave, lane, wf);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int r = 4 * i + wave, row = row0 + r;   // wave-uniform
    float xv = 0.f;
    if (row < p.R) {
      long long id = p.ids[row];
      if (id < 0 || id > p.V) id = 0;
      const float e = p.E[id * 64 + lane] * tr_drop_scale(seed, 0, (unsigned long long)row * 64 + lane, p.p_drop);
      float s1 = e;
#pragma unroll
      for (int o = 32; o >= 1; o >>= 1) s1 += __shfl_xor(s1, o, 64);
      const float mu = s1 * (1.0f / 64);
      const float d = e - mu;
      float s2 = d * d;
#pragma unroll
      for (int o = 32; o >= 1; o >>= 1) s2 += __shfl_xor(s2, o, 64);
      const float rs = 1.0f / sqrtf(s2 * (1.0f / 64) + LR_LN_EPS);
      const float xh = d * rs;
      xv = xh * p.ln_w[lane] + p.ln_b[lane];
      p.xhat[(size_t)row * 64 + lane] = xh;
      p.x[(size_t)row * 64 + lane] = xv;
      if (lane == 0) p.rstd[row] = rs;
    }
    Xs[r][lane] = xv;
  }
  __syncthreads();
  floatx4 acc[4];
#pragma unroll
  for (int nb = 0; nb < 4; ++nb) acc[nb] = floatx4{0.f, 0.f, 0.f, 0.f};
  tb_mma_n4(&Xs[li][4 * g], wf, acc);
  const int row = row0 + li;
  if (row >= p.R) return;
#pragma unroll
  for (int nb = 0; nb < 4; ++nb) {
    const int n = 64 * wave + 16 * nb + 4 * g;
    const float4 b = *reinterpret_cast<const float4*>(p.bi + n);
    *reinterpret_cast<float4*>(p.u + (size_t)row * 256 + n) = make_float4(acc[nb][0] + b.x, acc[nb][1] + b.y, acc[nb][2] + b.z, acc[nb][3] + b.w);
  }
}
int tb_launch_embed_in_proj(const TbEmbedInProj& p, hipStream_t st) {
  hipLaunchKernelGGL(tb_embed_in_proj_kernel, dim3((p.R + TB_ROWS - 1) / TB_ROWS), dim3(256), 0, st, p);
  LR_CHECK_LAUNCH("tb_embed_in_proj_kernel");
  return LR_OK;
}

// in_proj backward (data): dx += du wi     (wiT [64][256]: the product has the forward form with N = 64, K = 256); one
// 16-row panel per workgroup; runs as the first workgroups of tb_bwd_tail_kernel
__device__ __forceinline__ void tb_in_proj_bwd_body(const float* __restrict__ du, const float* __restrict__ wiT, float* dx, int R,
                                                    int panel) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, li = lane & 15, g = lane >> 4;
  const int row = panel * TB_ROWS + li;
  const bool ok = row < R;
  float4 wf[1][16];
  tb_load_w<256, 1>(wiT, 16 * wave, lane, wf);
  floatx4 part[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) part[e] = floatx4{0.f, 0.f, 0.f, 0.f};
  float4 xa[16];
#pragma unroll
  for (int j = 0; j < 16; ++j) xa[j] = tb_ld4(du + (size_t)row * 256 + 16 * j + 4 * g, ok);
#pragma unroll
  for (int j = 0; j < 16; ++j)
#pragma unroll
    for (int e = 0; e < 4; ++e)
      part[e] = __builtin_amdgcn_mfma_f32_16x16x4f32(tb_e(wf[0][j], e), tb_e(xa[j], e), part[e], 0, 0, 0);
  if (!ok) return;
  const floatx4 s = (part[0] + part[1]) + (part[2] + part[3]);
  float4* d = reinterpret_cast<float4*>(dx + (size_t)row * 64 + 16 * wave + 4 * g);
  const float4 o = *d;
  *d = make_float4(o.x + s[0], o.y + s[1], o.z + s[2], o.w + s[3]);
}

// =============================================================================================
// forward of a block behind the recurrence:
//   o = h wo^T + bo;  y = LN1(dropout(o) + x);  a = y w1^T + b1;  g = dropout(gelu(a));  z = g w2^T + b2;
//   xout = LN2(dropout(z) + y)                                   (model/lru.py:139-175; dropout sites as in lru_train.hip)
// =============================================================================================
__device__ __forceinline__ float tb_gelu(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752f)); }

// LayerNorm of e (4 values of row r per thread) -> xhat, y, rstd; returns y
__device__ __forceinline__ float4 tb_ln_fwd(float4 e, const float* w, const float* b, int c, float4* xhat_out, float* rstd_out) {
  const float mu = tb_row_sum((e.x + e.y) + (e.z + e.w)) * (1.0f / 64);
  const float4 d = make_float4(e.x - mu, e.y - mu, e.z - mu, e.w - mu);
  const float rs = 1.0f / sqrtf(tb_row_sum((d.x * d.x + d.y * d.y) + (d.z * d.z + d.w * d.w)) * (1.0f / 64) + LR_LN_EPS);
  const float4 xh = make_float4(d.x * rs, d.y * rs, d.z * rs, d.w * rs);
  const float4 w4 = *reinterpret_cast<const float4*>(w + c), b4 = *reinterpret_cast<const float4*>(b + c);
  *xhat_out = xh;
  *rstd_out = rs;
  return make_float4(xh.x * w4.x + b4.x, xh.y * w4.y + b4.y, xh.z * w4.z + b4.z, xh.w * w4.w + b4.w);
}

__global__ __launch_bounds__(256) void tb_block_fwd_kernel(TbBlockFwd p) {
  __shared__ __attribute__((aligned(16))) float Hs[TB_ROWS][TB_LD256];   // h panel, later the g panel
  __shared__ __attribute__((aligned(16))) float Os[TB_ROWS][TB_LD64];    // o, later z
  __shared__ __attribute__((aligned(16))) float Ys[TB_ROWS][TB_LD64];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 15, g = lane >> 4;
  const int row0 = blockIdx.x * TB_ROWS;
  const unsigned long long seed = *p.seed;
  // thread layout of the row-wise steps: row r, columns c .. c + 3
  const int r = tid >> 4, c = 4 * (tid & 15);
  const int rrow = row0 + r;
  const bool rok = rrow < p.R;
  const int mrow = row0 + li;          // row of this lane's MFMA results
  const bool mok = mrow < p.R;

  float4 wfo[1][16];
  tb_load_w<256, 1>(p.wo, 16 * wave, lane, wfo);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int idx = tid + 256 * i, hr = idx >> 6, hc = (idx & 63) * 4;
    *reinterpret_cast<float4*>(&Hs[hr][hc]) = tb_ld4(p.h + (size_t)(row0 + hr) * 256 + hc, row0 + hr < p.R);
  }
  const float4 xres = tb_ld4(p.x + (size_t)rrow * 64 + c, rok);
  __syncthreads();
  {  // out_proj
    const floatx4 s = tb_mma_n1(&Hs[li][4 * g], wfo);
    const int n = 16 * wave + 4 * g;
    const float4 b = *reinterpret_cast<const float4*>(p.bo + n);
    *reinterpret_cast<float4*>(&Os[li][n]) = make_float4(s[0] + b.x, s[1] + b.y, s[2] + b.z, s[3] + b.w);
  }
  float4 wf1[4][4];
  tb_load_w<64, 4>(p.w1, 64 * wave, lane, wf1);
  __syncthreads();
  float4 yv;
  {  // LN1
    const float4 o = *reinterpret_cast<const float4*>(&Os[r][c]);
    const unsigned long long i0 = (unsigned long long)rrow * 64 + c;
    const float4 e = make_float4(o.x * tr_drop_scale(seed, p.site0, i0, p.p_attn) + xres.x,
                                 o.y * tr_drop_scale(seed, p.site0, i0 + 1, p.p_attn) + xres.y,
                                 o.z * tr_drop_scale(seed, p.site0, i0 + 2, p.p_attn) + xres.z,
                                 o.w * tr_drop_scale(seed, p.site0, i0 + 3, p.p_attn) + xres.w);
    float4 xh;
    float rs;
    yv = tb_ln_fwd(e, p.ln1_w, p.ln1_b, c, &xh, &rs);
    *reinterpret_cast<float4*>(&Ys[r][c]) = yv;
    if (rok) {
      *reinterpret_cast<float4*>(p.xhat1 + (size_t)rrow * 64 + c) = xh;
      *reinterpret_cast<float4*>(p.y + (size_t)rrow * 64 + c) = yv;
      if ((tid & 15) == 0) p.rstd1[rrow] = rs;
    }
  }
  __syncthreads();
  {  // W1 + GELU + dropout -> a, g (global) and the g panel (over the h panel: every wave is past its out_proj reads)
    floatx4 acc[4];
#pragma unroll
    for (int nb = 0; nb < 4; ++nb) acc[nb] = floatx4{0.f, 0.f, 0.f, 0.f};
    tb_mma_n4(&Ys[li][4 * g], wf1, acc);
#pragma unroll
    for (int nb = 0; nb < 4; ++nb) {
      const int n = 64 * wave + 16 * nb + 4 * g;
      const float4 b = *reinterpret_cast<const float4*>(p.b1 + n);
      const float4 a = make_float4(acc[nb][0] + b.x, acc[nb][1] + b.y, acc[nb][2] + b.z, acc[nb][3] + b.w);
      const unsigned long long i0 = (unsigned long long)mrow * 256 + n;
      const float4 gg = make_float4(tb_gelu(a.x) * tr_drop_scale(seed, p.site0 + 1, i0, p.p_drop),
                                    tb_gelu(a.y) * tr_drop_scale(seed, p.site0 + 1, i0 + 1, p.p_drop),
                                    tb_gelu(a.z) * tr_drop_scale(seed, p.site0 + 1, i0 + 2, p.p_drop),
                                    tb_gelu(a.w) * tr_drop_scale(seed, p.site0 + 1, i0 + 3, p.p_drop));
      *reinterpret_cast<float4*>(&Hs[li][n]) = gg;
      if (mok) {
        *reinterpret_cast<float4*>(p.a + (size_t)mrow * 256 + n) = a;
        *reinterpret_cast<float4*>(p.g + (size_t)mrow * 256 + n) = gg;
      }
    }
  }
  float4 wf2[1][16];
  tb_load_w<256, 1>(p.w2, 16 * wave, lane, wf2);
  __syncthreads();
  {  // W2
    const floatx4 s = tb_mma_n1(&Hs[li][4 * g], wf2);
    const int n = 16 * wave + 4 * g;
    const float4 b = *reinterpret_cast<const float4*>(p.b2 + n);
    *reinterpret_cast<float4*>(&Os[li][n]) = make_float4(s[0] + b.x, s[1] + b.y, s[2] + b.z, s[3] + b.w);
  }
  __syncthreads();
  {  // LN2
    const float4 z = *reinterpret_cast<const float4*>(&Os[r][c]);
    const unsigned long long i0 = (unsigned long long)rrow * 64 + c;
    const float4 e = make_float4(z.x * tr_drop_scale(seed, p.site0 + 2, i0, p.p_drop) + yv.x,
                                 z.y * tr_drop_scale(seed, p.site0 + 2, i0 + 1, p.p_drop) + yv.y,
                                 z.z * tr_drop_scale(seed, p.site0 + 2, i0 + 2, p.p_drop) + yv.z,
                                 z.w * tr_drop_scale(seed, p.site0 + 2, i0 + 3, p.p_drop) + yv.w);
    float4 xh;
    float rs;
    const float4 out = tb_ln_fwd(e, p.ln2_w, p.ln2_b, c, &xh, &rs);
    if (rok) {
      *reinterpret_cast<float4*>(p.xhat2 + (size_t)rrow * 64 + c) = xh;
      *reinterpret_cast<float4*>(p.xout + (size_t)rrow * 64 + c) = out;
      if ((tid & 15) == 0) p.rstd2[rrow] = rs;
    }
    if (p.next_wi) *reinterpret_cast<float4*>(&Ys[r][c]) = out;   // every LN2 read of Ys (yv) happened before LN1's barrier
  }
  if (!p.next_wi) return;
  // the next block's in_proj on the panel that is already here (one launch and one round trip of x less)
  float4 wfn[4][4];
  tb_load_w<64, 4>(p.next_wi, 64 * wave, lane, wfn);
  __syncthreads();
  floatx4 acc[4];
#pragma unroll
  for (int nb = 0; nb < 4; ++nb) acc[nb] = floatx4{0.f, 0.f, 0.f, 0.f};
  tb_mma_n4(&Ys[li][4 * g], wfn, acc);
  if (mok) {
#pragma unroll
    for (int nb = 0; nb < 4; ++nb) {
      const int n = 64 * wave + 16 * nb + 4 * g;
      const float4 b = *reinterpret_cast<const float4*>(p.next_bi + n);
      *reinterpret_cast<float4*>(p.next_u + (size_t)mrow * 256 + n) = make_float4(acc[nb][0] + b.x, acc[nb][1] + b.y, acc[nb][2] + b.z, acc[nb][3] + b.w);
    }
  }
}
int tb_launch_block_fwd(const TbBlockFwd& p, hipStream_t st) {
  hipLaunchKernelGGL(tb_block_fwd_kernel, dim3((p.R + TB_ROWS - 1) / TB_ROWS), dim3(256), 0, st, p);
  LR_CHECK_LAUNCH("tb_block_fwd_kernel");
  return LR_OK;
}

// =============================================================================================
// backward of a block down to the recurrence's output gradient:
//   d  = LN2'(dx);  dz0 = dropout(d);  dg = dz0 w2;  da = dropout'(gelu'(a)) dg;  dy = d + da w1;
//   d1 = LN1'(dy) -> dx (residual path);  dy0 = dropout(d1);  dh = dy0 wo
// =============================================================================================
// LayerNorm backward of 4 values of a row: returns d (pre-LN gradient); gw/gb = this thread's share of d gamma / d beta
__device__ __forceinline__ float4 tb_ln_bwd(float4 gy, float4 xh, float rs, const float* w, int c, float4* gw, float4* gb) {
  const float4 w4 = *reinterpret_cast<const float4*>(w + c);
  const float4 dxh = make_float4(gy.x * w4.x, gy.y * w4.y, gy.z * w4.z, gy.w * w4.w);
  const float m1 = tb_row_sum((dxh.x + dxh.y) + (dxh.z + dxh.w)) * (1.0f / 64);
  const float m2 = tb_row_sum((dxh.x * xh.x + dxh.y * xh.y) + (dxh.z * xh.z + dxh.w * xh.w)) * (1.0f / 64);
  *gw = make_float4(gy.x * xh.x, gy.y * xh.y, gy.z * xh.z, gy.w * xh.w);
  *gb = gy;
  return make_float4(rs * (dxh.x - m1 - xh.x * m2), rs * (dxh.y - m1 - xh.y * m2), rs * (dxh.z - m1 - xh.z * m2),
                     rs * (dxh.w - m1 - xh.w * m2));
}
__device__ __forceinline__ float tb_gelu_grad(float x) {
  return 0.5f * (1.0f + erff(x * 0.70710678118654752f)) + x * 0.3989422804014327f * expf(-0.5f * x * x);
}

__global__ __launch_bounds__(256) void tb_block_bwd_kernel(TbBlockBwd p) {
  __shared__ __attribute__((aligned(16))) float DAs[TB_ROWS][TB_LD256];
  __shared__ __attribute__((aligned(16))) float Ds[TB_ROWS][TB_LD64];    // d (LN2 gradient, the residual's share of dy)
  __shared__ __attribute__((aligned(16))) float DZs[TB_ROWS][TB_LD64];   // dz0, later dy
  __shared__ __attribute__((aligned(16))) float DYs[TB_ROWS][TB_LD64];   // dy0
  __shared__ __attribute__((aligned(16))) float red[2][TB_ROWS][64];     // d gamma / d beta shares of the 16 rows
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 15, g = lane >> 4;
  const int row0 = blockIdx.x * TB_ROWS;
  const unsigned long long seed = *p.seed;
  const int r = tid >> 4, c = 4 * (tid & 15);
  const int rrow = row0 + r;
  const bool rok = rrow < p.R;
  const int mrow = row0 + li;
  const bool mok = mrow < p.R;

  float4 wf2[4][4];
  tb_load_w<64, 4>(p.w2T, 64 * wave, lane, wf2);
  {  // LN2 backward
    const float4 gy = tb_ld4(p.dx + (size_t)rrow * 64 + c, rok);
    const float4 xh = tb_ld4(p.xhat2 + (size_t)rrow * 64 + c, rok);
    const float rs = rok ? p.rstd2[rrow] : 0.f;
    float4 gw, gb;
    const float4 d = tb_ln_bwd(gy, xh, rs, p.ln2_w, c, &gw, &gb);
    const unsigned long long i0 = (unsigned long long)rrow * 64 + c;
    const float4 dz = make_float4(d.x * tr_drop_scale(seed, p.site0 + 2, i0, p.p_drop), d.y * tr_drop_scale(seed, p.site0 + 2, i0 + 1, p.p_drop),
                                  d.z * tr_drop_scale(seed, p.site0 + 2, i0 + 2, p.p_drop), d.w * tr_drop_scale(seed, p.site0 + 2, i0 + 3, p.p_drop));
    *reinterpret_cast<float4*>(&Ds[r][c]) = d;
    *reinterpret_cast<float4*>(&DZs[r][c]) = dz;
    *reinterpret_cast<float4*>(&red[0][r][c]) = gw;
    *reinterpret_cast<float4*>(&red[1][r][c]) = gb;
    if (rok) *reinterpret_cast<float4*>(p.dz0 + (size_t)rrow * 64 + c) = dz;
  }
  __syncthreads();
  if (tid < 128) {
    const int which = tid >> 6, col = tid & 63;
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < TB_ROWS; ++i) s += red[which][i][col];
    lr_det_add((which ? p.dln2_b : p.dln2_w) + col, s);
  }
  {  // dg = dz0 w2, then through dropout and GELU: da
    floatx4 acc[4];
#pragma unroll
    for (int nb = 0; nb < 4; ++nb) acc[nb] = floatx4{0.f, 0.f, 0.f, 0.f};
    tb_mma_n4(&DZs[li][4 * g], wf2, acc);
#pragma unroll
    for (int nb = 0; nb < 4; ++nb) {
      const int n = 64 * wave + 16 * nb + 4 * g;
      const float4 a = tb_ld4(p.a + (size_t)mrow * 256 + n, mok);
      const unsigned long long i0 = (unsigned long long)mrow * 256 + n;
      const float4 da = make_float4(acc[nb][0] * tr_drop_scale(seed, p.site0 + 1, i0, p.p_drop) * tb_gelu_grad(a.x),
                                    acc[nb][1] * tr_drop_scale(seed, p.site0 + 1, i0 + 1, p.p_drop) * tb_gelu_grad(a.y),
                                    acc[nb][2] * tr_drop_scale(seed, p.site0 + 1, i0 + 2, p.p_drop) * tb_gelu_grad(a.z),
                                    acc[nb][3] * tr_drop_scale(seed, p.site0 + 1, i0 + 3, p.p_drop) * tb_gelu_grad(a.w));
      *reinterpret_cast<float4*>(&DAs[li][n]) = da;
      if (mok) *reinterpret_cast<float4*>(p.da + (size_t)mrow * 256 + n) = da;
    }
  }
  float4 wf1[1][16];
  tb_load_w<256, 1>(p.w1T, 16 * wave, lane, wf1);
  __syncthreads();   // DAs complete; DZs and red are free again
  {  // dy = d + da w1
    const floatx4 s = tb_mma_n1(&DAs[li][4 * g], wf1);
    const int n = 16 * wave + 4 * g;
    const float4 d = *reinterpret_cast<const float4*>(&Ds[li][n]);
    *reinterpret_cast<float4*>(&DZs[li][n]) = make_float4(d.x + s[0], d.y + s[1], d.z + s[2], d.w + s[3]);
  }
  float4 wfo[4][4];
  tb_load_w<64, 4>(p.woT, 64 * wave, lane, wfo);
  __syncthreads();
  {  // LN1 backward
    const float4 gy = *reinterpret_cast<const float4*>(&DZs[r][c]);
    const float4 xh = tb_ld4(p.xhat1 + (size_t)rrow * 64 + c, rok);
    const float rs = rok ? p.rstd1[rrow] : 0.f;
    float4 gw, gb;
    const float4 d1 = tb_ln_bwd(gy, xh, rs, p.ln1_w, c, &gw, &gb);
    const unsigned long long i0 = (unsigned long long)rrow * 64 + c;
    const float4 dy0 = make_float4(d1.x * tr_drop_scale(seed, p.site0, i0, p.p_attn), d1.y * tr_drop_scale(seed, p.site0, i0 + 1, p.p_attn),
                                   d1.z * tr_drop_scale(seed, p.site0, i0 + 2, p.p_attn), d1.w * tr_drop_scale(seed, p.site0, i0 + 3, p.p_attn));
    *reinterpret_cast<float4*>(&DYs[r][c]) = dy0;
    *reinterpret_cast<float4*>(&red[0][r][c]) = gw;
    *reinterpret_cast<float4*>(&red[1][r][c]) = gb;
    if (rok) {
      *reinterpret_cast<float4*>(p.dx + (size_t)rrow * 64 + c) = d1;
      *reinterpret_cast<float4*>(p.dy0 + (size_t)rrow * 64 + c) = dy0;
    }
  }
  __syncthreads();
  if (tid < 128) {
    const int which = tid >> 6, col = tid & 63;
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < TB_ROWS; ++i) s += red[which][i][col];
    lr_det_add((which ? p.dln1_b : p.dln1_w) + col, s);
  }
  {  // dh = dy0 wo
    floatx4 acc[4];
#pragma unroll
    for (int nb = 0; nb < 4; ++nb) acc[nb] = floatx4{0.f, 0.f, 0.f, 0.f};
    tb_mma_n4(&DYs[li][4 * g], wfo, acc);
    if (mok) {
#pragma unroll
      for (int nb = 0; nb < 4; ++nb) {
        const int n = 64 * wave + 16 * nb + 4 * g;
        *reinterpret_cast<float4*>(p.dh + (size_t)mrow * 256 + n) = make_float4(acc[nb][0], acc[nb][1], acc[nb][2], acc[nb][3]);
      }
    }
  }
}
int tb_launch_block_bwd(const TbBlockBwd& p, hipStream_t st) {
  hipLaunchKernelGGL(tb_block_bwd_kernel, dim3((p.R + TB_ROWS - 1) / TB_ROWS), dim3(256), 0, st, p);
  LR_CHECK_LAUNCH("tb_block_bwd_kernel");
  return LR_OK;
}

// =============================================================================================
// the four weight gradients of a block: dW[N][K] += sum_r P[r][N] Q[r][K], db[N] += sum_r P[r][N]
// grid (row slices, 4 products); v_mfma_f32_32x32x2_f32 over 32-row chunks staged in LDS; a wave owns 2 x 2 blocks of
// 32 x 32 outputs; the adds of a wave-instruction cover two 128-byte row segments (the fast shape for float atomics)
// =============================================================================================
#define TB_WG_CHUNK 32
template <int N, int K>
__device__ __forceinline__ void tb_wgrad(const float* __restrict__ P, const float* __restrict__ Q, float* dW, float* db,
                                         int R, int r0, int r1, float* smem) {
  constexpr int LDP = N + 32, LDQ = K + 32;   // pitch = 32 mod 64: the two k rows of an MFMA step read disjoint bank halves
  constexpr int NP4 = TB_WG_CHUNK * N / 4 / 256, NQ4 = TB_WG_CHUNK * K / 4 / 256;   // float4 per thread and chunk
  float* Ps = smem;
  float* Qs = smem + TB_WG_CHUNK * LDP;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // wave's blocks: N = 64: both n blocks x k blocks 2w, 2w + 1;  N = 256: n blocks 2w, 2w + 1 x both k blocks
  const int nb0 = (N == 64) ? 0 : 2 * wave, kb0 = (N == 64) ? 2 * wave : 0;
  floatx16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
  float bsum = 0.f;
  float4 vp[NP4], vq[NQ4];
  auto load = [&](int rb) {
#pragma unroll
    for (int i = 0; i < NP4; ++i) {
      const int idx = tid + 256 * i, rr = idx / (N / 4), cc = (idx % (N / 4)) * 4;
      vp[i] = tb_ld4(P + (size_t)(rb + rr) * N + cc, rb + rr < r1);
    }
#pragma unroll
    for (int i = 0; i < NQ4; ++i) {
      const int idx = tid + 256 * i, rr = idx / (K / 4), cc = (idx % (K / 4)) * 4;
      vq[i] = tb_ld4(Q + (size_t)(rb + rr) * K + cc, rb + rr < r1);
    }
  };
  load(r0);
  for (int rb = r0; rb < r1; rb += TB_WG_CHUNK) {
    __syncthreads();
#pragma unroll
    for (int i = 0; i < NP4; ++i) {
      const int idx = tid + 256 * i, rr = idx / (N / 4), cc = (idx % (N / 4)) * 4;
      *reinterpret_cast<float4*>(Ps + rr * LDP + cc) = vp[i];
    }
#pragma unroll
    for (int i = 0; i < NQ4; ++i) {
      const int idx = tid + 256 * i, rr = idx / (K / 4), cc = (idx % (K / 4)) * 4;
      *reinterpret_cast<float4*>(Qs + rr * LDQ + cc) = vq[i];
    }
    __syncthreads();
    if (rb + TB_WG_CHUNK < r1) load(rb + TB_WG_CHUNK);
    if (tid < N) {
#pragma unroll 8
      for (int rr = 0; rr < TB_WG_CHUNK; ++rr) bsum += Ps[rr * LDP + tid];
    }
#pragma unroll
    for (int s = 0; s < TB_WG_CHUNK / 2; ++s) {
      const int kr = 2 * s + (lane >> 5);
      float a[2], b[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) a[i] = Ps[kr * LDP + (nb0 + i) * 32 + (lane & 31)];
#pragma unroll
      for (int j = 0; j < 2; ++j) b[j] = Qs[kr * LDQ + (kb0 + j) * 32 + (lane & 31)];
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
    }
  }
  if (tid < N) lr_det_add(db + tid, bsum);
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int m = (nb0 + i) * 32 + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
        lr_det_add(dW + (size_t)m * K + (kb0 + j) * 32 + (lane & 31), acc[i][j][e]);
      }
}
// After the recurrence's backward pass two things are ready to run and independent of each other: the data gradient of
// in_proj (n_panels workgroups of 16 rows) and the block's four weight gradients (4 x row slices). One launch: the first
// n_panels workgroups do the former (they are the short ones and finish under the others).
__global__ __launch_bounds__(256) void tb_bwd_tail_kernel(TbWeightGrads p, int rows_per_wg, const float* du, const float* wiT,
                                                          float* dx, int n_panels) {
  __shared__ __attribute__((aligned(16))) float smem[TB_WG_CHUNK * (64 + 32 + 256 + 32)];
  if ((int)blockIdx.x < n_panels) {
    tb_in_proj_bwd_body(du, wiT, dx, p.R, blockIdx.x);
    return;
  }
  const int w = blockIdx.x - n_panels;
  const int i = w & 3;
  const int r0 = (w >> 2) * rows_per_wg, r1 = min(p.R, r0 + rows_per_wg);
  if (r0 >= r1) return;
  if (i & 1) tb_wgrad<256, 64>(p.P[i], p.Q[i], p.dW[i], p.db[i], p.R, r0, r1, smem);
  else tb_wgrad<64, 256>(p.P[i], p.Q[i], p.dW[i], p.db[i], p.R, r0, r1, smem);
}
int tb_launch_bwd_tail(const TbWeightGrads& p, const float* du, const float* wiT, float* dx, hipStream_t st) {
  // 64-row slices while that keeps the launch under ~4 workgroups per CU (Beauty: 50 slices x 4 = 200 workgroups, 13 MB of
  // adds); 128 and 256 rows beyond
  const int rows = p.R <= 16384 ? 64 : p.R <= 65536 ? 128 : 256;
  const int n_panels = (p.R + TB_ROWS - 1) / TB_ROWS, slices = (p.R + rows - 1) / rows;
  hipLaunchKernelGGL(tb_bwd_tail_kernel, dim3(n_panels + 4 * slices), dim3(256), 0, st, p, rows, du, wiT, dx, n_panels);
  LR_CHECK_LAUNCH("tb_bwd_tail_kernel");
  return LR_OK;
}
