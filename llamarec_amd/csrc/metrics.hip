// metrics.hip -- rank histogram and small-C class ranking (integer / index work, exact).
// Replaces the one-hot + argsort + gather formulation of trainer/utils.py:43-90.
#include "lr_common.h"

__global__ __launch_bounds__(256) void rank_hist_kernel(const int32_t* ranked, int Kmax,
                                                        const int64_t* labels, int B,
                                                        unsigned long long* hist) {
  extern __shared__ unsigned int lh[];  // [Kmax+1]
  for (int i = threadIdx.x; i <= Kmax; i += blockDim.x) lh[i] = 0;
  __syncthreads();
  for (int u = blockIdx.x * blockDim.x + threadIdx.x; u < B; u += gridDim.x * blockDim.x) {
    const int64_t lab = labels[u];
    const int32_t* r = ranked + (size_t)u * Kmax;
    int pos = Kmax;
    for (int p = Kmax - 1; p >= 0; --p)
      if ((int64_t)r[p] == lab) pos = p;  // first occurrence wins
    atomicAdd(&lh[pos], 1u);
  }
  __syncthreads();
  for (int i = threadIdx.x; i <= Kmax; i += blockDim.x)
    if (lh[i]) atomicAdd(&hist[i], (unsigned long long)lh[i]);
}

// one wave per row; lane c holds class c's key; rank = number of larger keys
__global__ __launch_bounds__(256) void rank_classes_kernel(const float* scores, int B, int C,
                                                           const int32_t* items, int32_t* out) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= B) return;
  unsigned long long key = lane < C ? lr_rank_key(scores[(size_t)row * C + lane], (uint32_t)lane) : 0ull;
  int rank = 0;
  for (int m = 0; m < C; ++m) {
    unsigned long long km = __shfl(key, m, 64);
    rank += km > key ? 1 : 0;
  }
  if (lane < C) out[(size_t)row * C + rank] = items ? items[(size_t)row * C + lane] : lane;
}

extern "C" int lr_rank_histogram(const int32_t* ranked, int32_t Kmax, const int64_t* labels, int32_t B,
                                 int64_t* hist, void* hip_stream) {
  if (!ranked || !labels || !hist || Kmax < 1 || Kmax > 4096 || B < 0)
    LR_FAIL(LR_EINVAL, "lr_rank_histogram: bad arguments (Kmax=%d B=%d)", Kmax, B);
  if (B == 0) return LR_OK;
  int grid = (B + 255) / 256;
  if (grid > 1024) grid = 1024;
  hipLaunchKernelGGL(rank_hist_kernel, dim3(grid), dim3(256), (Kmax + 1) * sizeof(unsigned int),
                     (hipStream_t)hip_stream, ranked, Kmax, labels, B,
                     reinterpret_cast<unsigned long long*>(hist));
  LR_CHECK_LAUNCH("rank_hist_kernel");
  return LR_OK;
}

extern "C" int lr_rank_classes(const float* scores, int32_t B, int32_t C, const int32_t* items,
                               int32_t* out_ranked, void* hip_stream) {
  if (!scores || !out_ranked || C < 1 || C > 64 || B < 0)
    LR_FAIL(LR_EINVAL, "lr_rank_classes: bad arguments (B=%d C=%d, C must be 1..64)", B, C);
  if (B == 0) return LR_OK;
  hipLaunchKernelGGL(rank_classes_kernel, dim3((B + 3) / 4), dim3(256), 0, (hipStream_t)hip_stream,
                     scores, B, C, items, out_ranked);
  LR_CHECK_LAUNCH("rank_classes_kernel");
  return LR_OK;
}

extern "C" int lr_metrics_from_histogram(const int64_t* hist, int32_t Kmax, const int32_t* ks,
                                         int32_t nk, double* sums) {
  if (!hist || !ks || !sums || Kmax < 1 || nk < 0) LR_FAIL(LR_EINVAL, "lr_metrics_from_histogram: bad arguments");
  for (int j = 0; j < nk; ++j) {
    if (ks[j] < 1 || ks[j] > Kmax) LR_FAIL(LR_EINVAL, "lr_metrics_from_histogram: k=%d outside 1..%d", ks[j], Kmax);
    double rec = 0, mrr = 0, ndcg = 0;
    for (int p = 0; p < ks[j]; ++p) {
      double c = (double)hist[p];
      rec += c;
      mrr += c / (double)(p + 1);
      ndcg += c / log2((double)(p + 2));
    }
    sums[3 * j + 0] = rec;
    sums[3 * j + 1] = mrr;
    sums[3 * j + 2] = ndcg;
  }
  return LR_OK;
}
