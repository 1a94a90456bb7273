// lru_train_ce.hip -- fused item GEMM + softmax cross-entropy of the retriever's training step: the
// [rows x (V+1)] logit matrix of LRUTrainer.calculate_loss (trainer/lru.py:22-27, model/lru.py:85) is never
// written to memory. Three passes recompute 32 x 32 score tiles on v_mfma_f32_32x32x2_f32 (K = 64: 32 MFMAs):
//   A  stats   per row: running max and sum of exp over the items (one partial per item chunk)
//   B  d x     d x[row] = sum_j dl[row][j] e_j              rows stationary, item tiles streamed through LDS
//   C  d E     d e_j = sum_row dl[row][j] x[row], d bias_j   items stationary, row tiles streamed through LDS
// with dl = (exp(s - lse[row]) - [j == label[row]]) / n_labelled. In B and C the tile of dl sits in the MFMA
// accumulator layout, which already IS the B operand of the second product (lane-half 0 / 1 own the items /
// rows that MFMA step t pairs up), so nothing is transposed through LDS.
// Rows whose label is 0 (ignore_index) or out of range get lse = +inf: their dl is exactly 0.
#include <stdlib.h>

#include "lr_common.h"
#include "lr_det.h"
LR_DET_DEFINE(ce)

typedef float floatx16 __attribute__((ext_vector_type(16)));

#define CE_ES 68  // LDS row stride (floats) of a 64-float row

struct CeArgs {
  const float* X;      // [R][64] final hidden states
  const float* E;      // [C][64] item table (tied embedding)
  const float* bias;   // [C]
  const long long* labels;  // [R]
  int R, C;
  int tiles_per_chunk, n_chunks;  // item tiles (pass A, B) or row tiles (pass C) per workgroup along grid.x
  float* part;         // [n_chunks][R][2]  (max, sum) partials of pass A
  float* lse;          // [R]  log-sum-exp, +inf for unlabelled rows
  float* scal;         // training scalars: [0] loss sum, [1] labelled rows
  float* dX;           // [R][64]  (zeroed; chunks add)
  float* dE;           // [C][64]  (gradient buffer; chunks add)
  float* dbias;        // [C]
};

__device__ __forceinline__ float ce_other_half(float v) {  // value held by lane ^ 32
  const unsigned u = __builtin_bit_cast(unsigned, v);
  const auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
  const unsigned r0 = r[0], r1 = r[1];
  return (threadIdx.x & 32) ? __builtin_bit_cast(float, r0) : __builtin_bit_cast(float, r1);
}

// stage a 32-row x 64-float tile (rows row0.., clamped to n_rows-1) into LDS, stride CE_ES; 256 threads
__device__ __forceinline__ void ce_stage(const float* src, int row0, int n_rows, float* dst, int tid) {
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int f = tid + 256 * i;          // float4 index: row = f >> 4, chunk = f & 15
    const int row = min(row0 + (f >> 4), n_rows - 1);
    *reinterpret_cast<float4*>(dst + (f >> 4) * CE_ES + (f & 15) * 4) =
        *reinterpret_cast<const float4*>(src + (size_t)row * 64 + (f & 15) * 4);
  }
}

// ---- pass A: (max, sum exp) of every row over one chunk of item tiles ------------------------------------------
__global__ __launch_bounds__(256) void ce_stats_kernel(CeArgs a) {
  __shared__ __attribute__((aligned(16))) float et[2][32 * CE_ES];
  __shared__ __attribute__((aligned(16))) float bt[2][32];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5, col = lane & 31;
  const int row = blockIdx.y * 128 + wave * 32 + col;
  const int n_tiles = (a.C + 31) / 32;
  const int t0 = blockIdx.x * a.tiles_per_chunk, t1 = min(n_tiles, t0 + a.tiles_per_chunk);
  float xq[32];
  {
    const float4* xp = reinterpret_cast<const float4*>(a.X + (size_t)min(row, a.R - 1) * 64 + 32 * half);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float4 v = xp[j];
      xq[4 * j + 0] = v.x; xq[4 * j + 1] = v.y; xq[4 * j + 2] = v.z; xq[4 * j + 3] = v.w;
    }
  }
  float m = -__builtin_inff(), l = 0.f;
  if (t0 < t1) {
    ce_stage(a.E, t0 * 32, a.C, et[0], tid);
    if (tid < 32) bt[0][tid] = a.bias[min(t0 * 32 + tid, a.C - 1)];
  }
  __syncthreads();
  for (int t = t0; t < t1; ++t) {
    const int cur = (t - t0) & 1;
    if (t + 1 < t1) {
      ce_stage(a.E, (t + 1) * 32, a.C, et[cur ^ 1], tid);
      if (tid < 32) bt[cur ^ 1][tid] = a.bias[min((t + 1) * 32 + tid, a.C - 1)];
    }
    float e[32];
    {
      const float* er = et[cur] + col * CE_ES + 32 * half;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float4 v = *reinterpret_cast<const float4*>(er + 4 * j);
        e[4 * j + 0] = v.x; e[4 * j + 1] = v.y; e[4 * j + 2] = v.z; e[4 * j + 3] = v.w;
      }
    }
    floatx16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
    for (int s = 0; s < 32; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(e[s], xq[s], acc, 0, 0, 0);
    // lane = row `col`; register r = item (r&3) + 8*(r>>2) + 4*half of this tile
    float sc[16], tm = -__builtin_inff();
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const float4 b4 = *reinterpret_cast<const float4*>(&bt[cur][8 * g + 4 * half]);
      const float bb[4] = {b4.x, b4.y, b4.z, b4.w};
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int item = t * 32 + 8 * g + 4 * half + i;
        const float s_ = item < a.C ? acc[4 * g + i] + bb[i] : -__builtin_inff();
        sc[4 * g + i] = s_;
        tm = fmaxf(tm, s_);
      }
    }
    const float mn = fmaxf(m, tm);
    if (mn > -__builtin_inff()) {
      float ps = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) ps += __expf(sc[r] - mn);
      l = l * __expf(m - mn) + ps;
      m = mn;
    }
    __syncthreads();
  }
  // combine the two lane halves (different items of the same row), write the chunk's partial
  const float mo = ce_other_half(m), lo = ce_other_half(l);
  const float mm = fmaxf(m, mo);
  const float ll = (mm > -__builtin_inff()) ? l * __expf(m - mm) + lo * __expf(mo - mm) : 0.f;
  if (half == 0 && row < a.R) {
    float* p = a.part + ((size_t)blockIdx.x * a.R + row) * 2;
    p[0] = mm;
    p[1] = ll;
  }
}

// ---- pass A2: lse per row from the chunk partials, the label logit, the loss sum --------------------------------
__global__ __launch_bounds__(256) void ce_finish_kernel(CeArgs a) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= a.R) return;
  const long long lab = a.labels[row];
  const bool valid = lab > 0 && lab < a.C;
  float m = -__builtin_inff(), l = 0.f;
  if (valid) {
    for (int c = 0; c < a.n_chunks; ++c) {
      const float* p = a.part + ((size_t)c * a.R + row) * 2;
      const float mc = p[0], lc = p[1];
      const float mn = fmaxf(m, mc);
      l = l * __expf(m - mn) + lc * __expf(mc - mn);
      m = mn;
    }
    float d = a.X[(size_t)row * 64 + lane] * a.E[(size_t)lab * 64 + lane];
#pragma unroll
    for (int s = 32; s >= 1; s >>= 1) d += __shfl_xor(d, s, 64);
    if (lane == 0) {
      const float lse = m + logf(l);
      a.lse[row] = lse;
      lr_det_add(a.scal, lse - (d + a.bias[lab]));
    }
  } else if (lane == 0) {
    a.lse[row] = __builtin_inff();
  }
}

// ---- pass B: d x = sum_j dl[.][j] e_j over one chunk of item tiles; rows stationary -------------------------------
__global__ __launch_bounds__(256) void ce_dx_kernel(CeArgs a) {
  __shared__ __attribute__((aligned(16))) float et[2][32 * CE_ES];
  __shared__ __attribute__((aligned(16))) float bt[2][32];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5, col = lane & 31;
  const int row = blockIdx.y * 128 + wave * 32 + col;
  const int n_tiles = (a.C + 31) / 32;
  const int t0 = blockIdx.x * a.tiles_per_chunk, t1 = min(n_tiles, t0 + a.tiles_per_chunk);
  if (t0 >= t1) return;
  float xq[32];
  {
    const float4* xp = reinterpret_cast<const float4*>(a.X + (size_t)min(row, a.R - 1) * 64 + 32 * half);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float4 v = xp[j];
      xq[4 * j + 0] = v.x; xq[4 * j + 1] = v.y; xq[4 * j + 2] = v.z; xq[4 * j + 3] = v.w;
    }
  }
  const float lse = row < a.R ? a.lse[row] : __builtin_inff();
  const long long lab = row < a.R ? a.labels[row] : -1;
  const float inv_n = 1.0f / fmaxf(a.scal[1], 1.0f);
  floatx16 dacc[2];  // d x^T: feature blk*32 + (r&3) + 8*(r>>2) + 4*half of row `col`
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) dacc[j][r] = 0.f;
  ce_stage(a.E, t0 * 32, a.C, et[0], tid);
  if (tid < 32) bt[0][tid] = a.bias[min(t0 * 32 + tid, a.C - 1)];
  __syncthreads();
  for (int t = t0; t < t1; ++t) {
    const int cur = (t - t0) & 1;
    if (t + 1 < t1) {
      ce_stage(a.E, (t + 1) * 32, a.C, et[cur ^ 1], tid);
      if (tid < 32) bt[cur ^ 1][tid] = a.bias[min((t + 1) * 32 + tid, a.C - 1)];
    }
    float e[32];
    {
      const float* er = et[cur] + col * CE_ES + 32 * half;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float4 v = *reinterpret_cast<const float4*>(er + 4 * j);
        e[4 * j + 0] = v.x; e[4 * j + 1] = v.y; e[4 * j + 2] = v.z; e[4 * j + 3] = v.w;
      }
    }
    floatx16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
    for (int s = 0; s < 32; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(e[s], xq[s], acc, 0, 0, 0);
    float dl[16];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const float4 b4 = *reinterpret_cast<const float4*>(&bt[cur][8 * g + 4 * half]);
      const float bb[4] = {b4.x, b4.y, b4.z, b4.w};
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int item = t * 32 + 8 * g + 4 * half + i;
        float p = item < a.C ? __expf(acc[4 * g + i] + bb[i] - lse) : 0.f;
        if ((long long)item == lab) p -= 1.0f;
        dl[4 * g + i] = p * inv_n;
      }
    }
    // d x^T[d][row] += sum_items E[item][d] * dl[row][item]: MFMA step tt pairs item (tt&3)+8*(tt>>2) (half 0)
    // with the one 4 further (half 1) -- exactly the items register tt of the two lane halves holds
#pragma unroll
    for (int tt = 0; tt < 16; ++tt) {
      const int item = (tt & 3) + 8 * (tt >> 2) + 4 * half;
      const float* ep = et[cur] + item * CE_ES + col;
      dacc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(ep[0], dl[tt], dacc[0], 0, 0, 0);
      dacc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(ep[32], dl[tt], dacc[1], 0, 0, 0);
    }
    __syncthreads();
  }
  if (row < a.R && lse < __builtin_inff()) {
    float* dst = a.dX + (size_t)row * 64;
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) lr_det_add(dst + j * 32 + (r & 3) + 8 * (r >> 2) + 4 * half, dacc[j][r]);
  }
}

// ---- pass C: d e_j and d bias_j over one chunk of row tiles; items stationary -------------------------------------
__global__ __launch_bounds__(256) void ce_de_kernel(CeArgs a) {
  __shared__ __attribute__((aligned(16))) float xt[2][32 * CE_ES];
  __shared__ __attribute__((aligned(16))) float lt[2][32];   // lse of the tile's rows
  __shared__ int labt[2][32];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5, col = lane & 31;
  const int item = blockIdx.y * 128 + wave * 32 + col;
  const int n_tiles = (a.R + 31) / 32;
  const int t0 = blockIdx.x * a.tiles_per_chunk, t1 = min(n_tiles, t0 + a.tiles_per_chunk);
  if (t0 >= t1) return;
  float eq[32];
  {
    const float4* ep = reinterpret_cast<const float4*>(a.E + (size_t)min(item, a.C - 1) * 64 + 32 * half);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float4 v = ep[j];
      eq[4 * j + 0] = v.x; eq[4 * j + 1] = v.y; eq[4 * j + 2] = v.z; eq[4 * j + 3] = v.w;
    }
  }
  const float bj = a.bias[min(item, a.C - 1)];
  const float inv_n = 1.0f / fmaxf(a.scal[1], 1.0f);
  floatx16 dacc[2];  // d E^T: feature blk*32 + (r&3) + 8*(r>>2) + 4*half of item `col`
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) dacc[j][r] = 0.f;
  float db = 0.f;
  auto stage = [&](int t, int bufi) {
    ce_stage(a.X, t * 32, a.R, xt[bufi], tid);
    if (tid < 32) {
      const int r_ = t * 32 + tid;
      lt[bufi][tid] = r_ < a.R ? a.lse[r_] : __builtin_inff();
      const long long lb = r_ < a.R ? a.labels[r_] : -1;
      labt[bufi][tid] = (lb > 0 && lb < a.C) ? (int)lb : -1;
    }
  };
  stage(t0, 0);
  __syncthreads();
  for (int t = t0; t < t1; ++t) {
    const int cur = (t - t0) & 1;
    if (t + 1 < t1) stage(t + 1, cur ^ 1);
    float x[32];
    {
      const float* xr = xt[cur] + col * CE_ES + 32 * half;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float4 v = *reinterpret_cast<const float4*>(xr + 4 * j);
        x[4 * j + 0] = v.x; x[4 * j + 1] = v.y; x[4 * j + 2] = v.z; x[4 * j + 3] = v.w;
      }
    }
    floatx16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
    for (int s = 0; s < 32; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(x[s], eq[s], acc, 0, 0, 0);
    // lane = item `col`; register r = row (r&3) + 8*(r>>2) + 4*half of this row tile
    float dl[16];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const float4 l4 = *reinterpret_cast<const float4*>(&lt[cur][8 * g + 4 * half]);
      const float ll[4] = {l4.x, l4.y, l4.z, l4.w};
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        float p = __expf(acc[4 * g + i] + bj - ll[i]);  // lse = +inf -> 0
        if (labt[cur][8 * g + 4 * half + i] == item) p -= 1.0f;
        p *= inv_n;
        dl[4 * g + i] = p;
        db += p;
      }
    }
#pragma unroll
    for (int tt = 0; tt < 16; ++tt) {
      const int r_ = (tt & 3) + 8 * (tt >> 2) + 4 * half;
      const float* xp = xt[cur] + r_ * CE_ES + col;
      dacc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(xp[0], dl[tt], dacc[0], 0, 0, 0);
      dacc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(xp[32], dl[tt], dacc[1], 0, 0, 0);
    }
    __syncthreads();
  }
  db += ce_other_half(db);
  if (item < a.C) {
    float* dst = a.dE + (size_t)item * 64;
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) lr_det_add(dst + j * 32 + (r & 3) + 8 * (r >> 2) + 4 * half, dacc[j][r]);
    if (half == 0) lr_det_add(a.dbias + item, db);
  }
}

// =============================================================================================
// workgroups aimed at per pass (LR_CE_WGS: tuning knob). Every chunk adds its 64-wide partial rows with atomics, so a
// chunk also keeps at least 32 tiles: measured on Beauty 256-512 workgroups are best, on a 10^6-item table 1024.
static int ce_target_wgs() {
  static int t = 0;
  if (!t) {
    const char* e = getenv("LR_CE_WGS");
    t = e ? atoi(e) : 1024;
    if (t < 1) t = 1024;
  }
  return t;
}
static int ce_chunks(int groups, int tiles) {
  int c = (ce_target_wgs() + groups - 1) / groups;
  const int by_len = tiles / 32 > 1 ? tiles / 32 : 1;
  if (c > by_len) c = by_len;
  return c;
}

size_t lr_train_ce_part_floats(int R, int C) {
  const int n_tiles = (C + 31) / 32, rg = (R + 127) / 128;
  return (size_t)ce_chunks(rg, n_tiles) * R * 2 + (size_t)R;
}

// scal[1] (labelled rows) must already be there. dX must be zero; dE / dbias receive added contributions.
int lr_launch_train_ce(const float* X, const float* E, const float* bias, const long long* labels, int R, int C,
                       float* ws_part /* lr_train_ce_part_floats */, float* scal, float* dX, float* dE, float* dbias,
                       hipStream_t st) {
  CeArgs a;
  a.X = X;
  a.E = E;
  a.bias = bias;
  a.labels = labels;
  a.R = R;
  a.C = C;
  a.scal = scal;
  a.dX = dX;
  a.dE = dE;
  a.dbias = dbias;
  const int item_tiles = (C + 31) / 32, row_groups = (R + 127) / 128;
  const int chunks = ce_chunks(row_groups, item_tiles);
  a.tiles_per_chunk = (item_tiles + chunks - 1) / chunks;
  a.n_chunks = (item_tiles + a.tiles_per_chunk - 1) / a.tiles_per_chunk;
  a.part = ws_part;
  a.lse = ws_part + (size_t)chunks * R * 2;
  hipLaunchKernelGGL(ce_stats_kernel, dim3(a.n_chunks, row_groups), dim3(256), 0, st, a);
  LR_CHECK_LAUNCH("ce_stats_kernel");
  hipLaunchKernelGGL(ce_finish_kernel, dim3((R + 3) / 4), dim3(256), 0, st, a);
  LR_CHECK_LAUNCH("ce_finish_kernel");
  hipLaunchKernelGGL(ce_dx_kernel, dim3(a.n_chunks, row_groups), dim3(256), 0, st, a);
  LR_CHECK_LAUNCH("ce_dx_kernel");
  // pass C: items stationary, row tiles chunked
  CeArgs c = a;
  const int row_tiles = (R + 31) / 32, item_groups = (C + 127) / 128;
  const int rchunks = ce_chunks(item_groups, row_tiles);
  c.tiles_per_chunk = (row_tiles + rchunks - 1) / rchunks;
  c.n_chunks = (row_tiles + c.tiles_per_chunk - 1) / c.tiles_per_chunk;
  hipLaunchKernelGGL(ce_de_kernel, dim3(c.n_chunks, item_groups), dim3(256), 0, st, c);
  LR_CHECK_LAUNCH("ce_de_kernel");
  return LR_OK;
}
