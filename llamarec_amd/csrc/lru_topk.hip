// lru_topk.hip -- item-table GEMM (last position only) fused with history masking and an
// LDS-staged wavefront top-K for gfx950. The [B][V+1] score matrix never reaches memory.
//
// Replaces (reference): scores = x @ E^T + bias  model/lru.py:85 (consumed as [:, -1, :] at
// trainer/lru.py:33,67,105), the -1e9 history/pad masking loop trainer/lru.py:35-38,72-74,110-112,
// torch.topk(scores, 20) :82-84,124-126 and (-scores).argsort()[:, :50] :113-115.
//
// Structure
//  * grid = (item chunks, 128-user tiles). A workgroup = 4 wave64; wave w owns 32 users whose
//    q rows live in registers as the MFMA B operand; all 4 waves share each 32-item tile of the
//    table, staged global -> LDS with coalesced 16-byte loads (row stride 68 floats: the
//    ds_read_b128 fragment reads are bank-conflict free).
//  * scores of a 32x32 (item x user) tile = 32 chained v_mfma_f32_32x32x2_f32 (exact f32,
//    k-ordered fma chain: lane-half 0 feeds k = s, lane-half 1 feeds k = 32+s, so the sum order
//    is lr_item_score()'s and the result is bit-identical to the CPU oracle).
//  * each lane then holds 16 item scores of ONE user; a score enters the user's LDS candidate
//    buffer only if its 64-bit rank key beats the user's current K-th key (threshold), so after
//    warm-up almost nothing is inserted. The history / pad-id test (-> -1e9) is evaluated only
//    for scores that pass the threshold. When a buffer could overflow the wave rank-selects it
//    back to its best K (which also tightens the threshold).
//  * per-chunk sorted partial lists are merged by a second tiny kernel (one wave per user).
#include "lr_common.h"
#include "lr_profile.h"

typedef float floatx16 __attribute__((ext_vector_type(16)));

#define TK_WAVES 4
#define TK_USERS 128      // users per workgroup
#define TK_CAP 112        // candidate slots per user (K <= 64 kept + insertion slack)
#define TK_ESTRIDE 68     // floats per staged table row (64 + 4 pad)
#define TK_MAX_CHUNKS 256

struct TopkParams {
  const float* emb;   // [rows_padded][64]
  const float* bias;  // [rows_padded]
  int n_rows;         // V + 1
  int n_tiles;        // rows_padded / 32
  const float* q;     // [B][64]
  const int64_t* ids; // [B][L]
  int B, L, K, exclude;
  int tiles_per_chunk, n_chunks;
  unsigned long long* partial;  // [B][n_chunks][K] rank keys, best first, 0 = empty
};

__device__ __forceinline__ bool in_history(const int64_t* row, int L, int item) {
  bool hit = false;
  for (int t = 0; t < L; ++t) hit |= (row[t] == (int64_t)item);
  return hit;
}

// Wave-cooperative: keep the best min(c,K) keys of b[0..c) sorted best-first in b[0..), and
// return the K-th best key (0 if c < K). All 64 lanes must call it with wave-uniform arguments.
__device__ __forceinline__ unsigned long long compact_user(unsigned long long* b, int c, int K,
                                                           int lane) {
  unsigned long long k0 = lane < c ? b[lane] : 0ull;
  unsigned long long k1 = lane + 64 < c ? b[lane + 64] : 0ull;
  int r0 = 0, r1 = 0;
  for (int m = 0; m < c; ++m) {
    unsigned long long km = b[m];
    r0 += (km > k0) ? 1 : 0;
    r1 += (km > k1) ? 1 : 0;
  }
  __builtin_amdgcn_wave_barrier();
  if (lane < c && r0 < K) b[r0] = k0;
  if (lane + 64 < c && r1 < K) b[r1] = k1;
  __builtin_amdgcn_wave_barrier();
  unsigned long long kth = 0ull;
  if (c >= K) kth = b[K - 1];  // same-wave LDS ops are ordered: this read sees the writes above
  return kth;
}

__global__ __launch_bounds__(256) void item_topk_kernel(TopkParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* etile = reinterpret_cast<float*>(smem);                      // [2][32*68]
  float* btile = etile + 2 * 32 * TK_ESTRIDE;                         // [2][32]
  unsigned long long* buf = reinterpret_cast<unsigned long long*>(btile + 64);  // [128][CAP]
  unsigned long long* thr = buf + TK_USERS * TK_CAP;                  // [128]
  int* cnt = reinterpret_cast<int*>(thr + TK_USERS);                  // [128]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int half = lane >> 5, col = lane & 31;
  const int chunk = blockIdx.x;
  const int u_local = wave * 32 + col;
  const int user = blockIdx.y * TK_USERS + u_local;
  const bool user_ok = user < p.B;
  const int tile_begin = chunk * p.tiles_per_chunk;
  const int tile_end = min(p.n_tiles, tile_begin + p.tiles_per_chunk);
  const int K = p.K;

  // q fragment: B operand of step s is q[user][32*half + s]
  float bq[32];
  {
    const float4* qp = reinterpret_cast<const float4*>(p.q + (size_t)(user_ok ? user : 0) * 64 + 32 * half);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      float4 v = user_ok ? qp[j] : make_float4(0.f, 0.f, 0.f, 0.f);
      bq[4 * j + 0] = v.x;
      bq[4 * j + 1] = v.y;
      bq[4 * j + 2] = v.z;
      bq[4 * j + 3] = v.w;
    }
  }
  if (tid < TK_USERS) {
    cnt[tid] = 0;
    thr[tid] = 0ull;
  }
  const int64_t* hist = p.ids + (size_t)(user_ok ? user : 0) * p.L;

  // staging assignment: 512 float4 per tile, thread handles float4 #tid and #tid+256
  float4 pre0, pre1;
  float preb = 0.f;
  auto load_tile = [&](int tile) {
    const float4* src = reinterpret_cast<const float4*>(p.emb + (size_t)tile * 32 * 64);
    pre0 = src[tid];
    pre1 = src[tid + 256];
    if (tid < 32) preb = p.bias[tile * 32 + tid];
  };
  auto store_tile = [&](int bufi) {
    float* e = etile + bufi * 32 * TK_ESTRIDE;
    int i0 = tid, i1 = tid + 256;
    *reinterpret_cast<float4*>(e + (i0 >> 4) * TK_ESTRIDE + (i0 & 15) * 4) = pre0;
    *reinterpret_cast<float4*>(e + (i1 >> 4) * TK_ESTRIDE + (i1 & 15) * 4) = pre1;
    if (tid < 32) btile[bufi * 32 + tid] = preb;
  };

  if (tile_begin < tile_end) {
    load_tile(tile_begin);
    store_tile(0);
  }
  __syncthreads();

  for (int tile = tile_begin; tile < tile_end; ++tile) {
    const int cur = (tile - tile_begin) & 1;
    if (tile + 1 < tile_end) load_tile(tile + 1);

    // A fragments: item row `col`, k = 32*half + s
    float a[32];
    {
      const float* er = etile + cur * 32 * TK_ESTRIDE + col * TK_ESTRIDE + 32 * half;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        float4 v = *reinterpret_cast<const float4*>(er + 4 * j);
        a[4 * j + 0] = v.x;
        a[4 * j + 1] = v.y;
        a[4 * j + 2] = v.z;
        a[4 * j + 3] = v.w;
      }
    }
    floatx16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
#pragma unroll
    for (int s = 0; s < 32; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s], bq[s], acc, 0, 0, 0);

    // lane now holds, for its user `col`, the scores of items row(r) = (r&3) + 8*(r>>2) + 4*half
    const unsigned long long th = thr[u_local];
    const float* bt = btile + cur * 32;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = (r & 3) + 8 * (r >> 2) + 4 * half;
      const int item = tile * 32 + row;
      const float sc = acc[r] + bt[row];
      unsigned long long key = lr_rank_key(sc, (uint32_t)item);
      bool cand = user_ok && item < p.n_rows && key > th;
      if (cand && p.exclude) {
        if (item == 0 || in_history(hist, p.L, item)) {
          key = lr_rank_key(LR_MASK_SCORE, (uint32_t)item);
          cand = key > th;
        }
      }
      if (cand) {
        int pos = atomicAdd(&cnt[u_local], 1);
        buf[u_local * TK_CAP + pos] = key;
      }
    }
    __builtin_amdgcn_wave_barrier();
    // keep room for the next tile (at most 32 insertions per user per tile)
    {
      int c = cnt[u_local];
      if (__any(c > TK_CAP - 32)) {
        for (int uu = 0; uu < 32; ++uu) {
          int ul = wave * 32 + uu;
          int cu = cnt[ul];  // wave-uniform (broadcast read)
          if (cu > K) {
            unsigned long long kth = compact_user(buf + ul * TK_CAP, cu, K, lane);
            if (lane == 0) {
              cnt[ul] = K;
              thr[ul] = kth;
            }
          }
        }
        __builtin_amdgcn_wave_barrier();
      }
    }
    if (tile + 1 < tile_end) store_tile(cur ^ 1);
    __syncthreads();
  }

  // final: sort every user's buffer and emit the chunk's partial list
  for (int uu = 0; uu < 32; ++uu) {
    int ul = wave * 32 + uu;
    int gu = blockIdx.y * TK_USERS + ul;
    if (gu >= p.B) break;  // wave-uniform
    int cu = cnt[ul];
    compact_user(buf + ul * TK_CAP, cu, K, lane);
    int keep = min(cu, K);
    unsigned long long* dst = p.partial + ((size_t)gu * p.n_chunks + chunk) * K;
    for (int j = lane; j < K; j += 64) dst[j] = j < keep ? buf[ul * TK_CAP + j] : 0ull;
  }
}

// ---- merge: one wave per user, K rounds of "largest head wins" over <= 256 sorted lists -----
struct MergeParams {
  const unsigned long long* partial;
  int B, K, n_chunks;
  int32_t* out_idx;
  float* out_score;
};

__device__ __forceinline__ unsigned long long wave_max_u64(unsigned long long v) {
#pragma unroll
  for (int s = 32; s >= 1; s >>= 1) {
    unsigned long long o = __shfl_xor(v, s, 64);
    v = o > v ? o : v;
  }
  return v;
}

__global__ __launch_bounds__(256) void topk_merge_kernel(MergeParams p) {
  const int lane = threadIdx.x & 63;
  const int user = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (user >= p.B) return;
  const unsigned long long* base = p.partial + (size_t)user * p.n_chunks * p.K;
  int ptr[4];
  unsigned long long hk[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    int l = lane + 64 * i;
    ptr[i] = 0;
    hk[i] = (l < p.n_chunks) ? base[(size_t)l * p.K] : 0ull;
  }
  for (int j = 0; j < p.K; ++j) {
    unsigned long long m = hk[0];
#pragma unroll
    for (int i = 1; i < 4; ++i) m = hk[i] > m ? hk[i] : m;
    m = wave_max_u64(m);
    if (lane == 0) {
      p.out_idx[(size_t)user * p.K + j] = m ? (int32_t)lr_key_item(m) : -1;
      if (p.out_score) p.out_score[(size_t)user * p.K + j] = m ? lr_key_score(m) : -__builtin_inff();
    }
    if (m == 0ull) continue;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      if (hk[i] == m) {  // keys are unique: exactly one lane/list advances
        int l = lane + 64 * i;
        ptr[i] += 1;
        hk[i] = (ptr[i] < p.K) ? base[(size_t)l * p.K + ptr[i]] : 0ull;
      }
    }
  }
}

// ---- compatibility path: materialised last-position scores ----------------------------------
__global__ __launch_bounds__(256) void item_scores_kernel(const float* emb, const float* bias,
                                                          int n_rows, const float* q, int B,
                                                          float* out) {
  __shared__ float qs[64];
  const int user = blockIdx.y;
  if (threadIdx.x < 64) qs[threadIdx.x] = q[(size_t)user * 64 + threadIdx.x];
  __syncthreads();
  int item = blockIdx.x * 256 + threadIdx.x;
  if (item >= n_rows) return;
  out[(size_t)user * n_rows + item] = lr_item_score(emb + (size_t)item * 64, qs, bias[item]);
}

__global__ void mask_history_kernel(float* scores, int n_rows, const int64_t* ids, int B, int L) {
  int user = blockIdx.x;
  for (int t = threadIdx.x; t < L; t += blockDim.x) {
    int64_t id = ids[(size_t)user * L + t];
    if (id >= 0 && id < n_rows) scores[(size_t)user * n_rows + id] = LR_MASK_SCORE;
  }
  if (threadIdx.x == 0) scores[(size_t)user * n_rows] = LR_MASK_SCORE;
}

static void topk_geometry(int n_tiles, int B, int* n_chunks, int* tiles_per_chunk) {
  int n_ut = (B + TK_USERS - 1) / TK_USERS;
  int want = (512 + n_ut - 1) / n_ut;
  if (want > TK_MAX_CHUNKS) want = TK_MAX_CHUNKS;
  if (want > n_tiles) want = n_tiles;
  if (want < 1) want = 1;
  int tpc = (n_tiles + want - 1) / want;
  *tiles_per_chunk = tpc;
  *n_chunks = (n_tiles + tpc - 1) / tpc;
}

size_t lr_topk_workspace_bytes(int B, int K) {
  // worst case over catalog sizes: n_chunks <= min(256, ceil(512 / user tiles))
  int n_ut = (B + TK_USERS - 1) / TK_USERS;
  if (n_ut < 1) n_ut = 1;
  int want = (512 + n_ut - 1) / n_ut;
  if (want > TK_MAX_CHUNKS) want = TK_MAX_CHUNKS;
  return lr_align_up((size_t)B * want * K * sizeof(unsigned long long), 256);
}

int lr_launch_item_topk(const lr_lru* h, const float* q, const int64_t* ids, int B, int L, int K,
                        int exclude_history, int32_t* out_idx, float* out_score, void* ws,
                        size_t ws_bytes, hipStream_t st) {
  if (B <= 0) return LR_OK;
  TopkParams p;
  p.emb = h->img + h->lay.item_emb;
  p.bias = h->img + h->lay.item_bias;
  p.n_rows = h->lay.num_items + 1;
  p.n_tiles = h->lay.rows_padded / LR_ITEM_TILE;
  p.q = q;
  p.ids = ids;
  p.B = B;
  p.L = L;
  p.K = K;
  p.exclude = exclude_history;
  topk_geometry(p.n_tiles, B, &p.n_chunks, &p.tiles_per_chunk);
  size_t need = (size_t)B * p.n_chunks * K * sizeof(unsigned long long);
  if (need > ws_bytes) LR_FAIL(LR_EWORKSPACE, "top-K workspace: need %zu bytes, have %zu", need, ws_bytes);
  p.partial = reinterpret_cast<unsigned long long*>(ws);

  const size_t lds = (2 * 32 * TK_ESTRIDE + 64) * sizeof(float) +
                     (size_t)TK_USERS * TK_CAP * 8 + TK_USERS * 8 + TK_USERS * 4;
  static bool attr_set = false;
  if (!attr_set) {
    LR_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(item_topk_kernel),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    attr_set = true;
  }
  dim3 grid(p.n_chunks, (B + TK_USERS - 1) / TK_USERS);
  LrProfScope prof(LR_PROF_ITEM_TOPK, 2.0 * 64 * (double)p.n_rows * B, st);
  hipLaunchKernelGGL(item_topk_kernel, grid, dim3(256), lds, st, p);
  LR_CHECK_LAUNCH("item_topk_kernel");

  MergeParams m;
  m.partial = p.partial;
  m.B = B;
  m.K = K;
  m.n_chunks = p.n_chunks;
  m.out_idx = out_idx;
  m.out_score = out_score;
  hipLaunchKernelGGL(topk_merge_kernel, dim3((B + 3) / 4), dim3(256), 0, st, m);
  LR_CHECK_LAUNCH("topk_merge_kernel");
  return LR_OK;
}

int lr_launch_item_scores(const lr_lru* h, const float* q, const int64_t* ids, int B, int L,
                          int exclude_history, float* out_scores, hipStream_t st) {
  if (B <= 0) return LR_OK;
  int n_rows = h->lay.num_items + 1;
  dim3 grid((n_rows + 255) / 256, B);
  hipLaunchKernelGGL(item_scores_kernel, grid, dim3(256), 0, st, h->img + h->lay.item_emb,
                     h->img + h->lay.item_bias, n_rows, q, B, out_scores);
  LR_CHECK_LAUNCH("item_scores_kernel");
  if (exclude_history) {
    hipLaunchKernelGGL(mask_history_kernel, dim3(B), dim3(256), 0, st, out_scores, n_rows, ids, B, L);
    LR_CHECK_LAUNCH("mask_history_kernel");
  }
  return LR_OK;
}
