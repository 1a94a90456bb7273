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
//    table, staged global -> registers -> LDS with coalesced 16-byte loads, two tiles ahead (row
//    stride 68 floats: the ds_read_b128 fragment reads are bank-conflict free).
//  * scores of a 32x32 (item x user) tile = 32 chained v_mfma_f32_32x32x2_f32 (exact f32,
//    k-ordered fma chain: lane-half 0 feeds k = s, lane-half 1 feeds k = 32+s, so the sum order
//    is lr_item_score()'s and the result is bit-identical to the CPU oracle).
//  * history / pad masking: a pre-pass sorts every user's history ids (hist_sort_kernel); while
//    the item tiles advance in id order each user walks a pointer through its sorted history and
//    builds a 32-bit mask of the tile's masked items -- two instructions per score, no scans.
//  * each lane holds 16 item scores of ONE user; a score enters the user's LDS candidate buffer
//    only if it beats the user's current K-th best (a float pre-test, then the exact 64-bit rank
//    key), so after warm-up almost nothing is inserted. When a buffer could overflow, the wave
//    finds that user's exact K-th key by counting quickselect (ballots, ~10 rounds) and compacts.
//  * per-chunk sorted partial lists are merged by a second tiny kernel (one wave per user).
#include <limits.h>

#include "lr_common.h"
#include "lr_profile.h"

typedef float floatx16 __attribute__((ext_vector_type(16)));

#define TK_WAVES 4
#define TK_USERS 128      // users per workgroup
#define TK_CAP 112        // candidate slots per user (K <= 64 kept + insertion slack)
#define TK_MAX_CHUNKS 256
#define TK_MAX_WGS 1024   // chunks are added only while user tiles x chunks stays below this

struct TopkParams {
  const float* emb;   // [rows_padded][64]
  const float* bias;  // [rows_padded]
  int n_rows;         // V + 1
  int n_tiles;        // rows_padded / 32
  const float* q;     // [B][64]
  const int32_t* hist_sorted;  // [B][L] ascending (INT_MAX = ignored entry), null if !exclude
  int B, L, K, exclude;
  int tiles_per_chunk, n_chunks;
  unsigned long long* partial;  // [B][n_chunks][K] rank keys, best first, 0 = empty
};

// ---- pre-pass: ascending sort of each user's history ids (LDS bitonic sort, one WG per user) ----
__global__ __launch_bounds__(256) void hist_sort_kernel(const int64_t* ids, int L, int Lp, int n_rows,
                                                        int32_t* out) {
  extern __shared__ int32_t sk[];
  const int u = blockIdx.x;
  for (int i = threadIdx.x; i < Lp; i += 256) {
    long long v = i < L ? ids[(size_t)u * L + i] : -1;
    sk[i] = (v >= 0 && v < n_rows) ? (int32_t)v : INT_MAX;
  }
  __syncthreads();
  for (int k = 2; k <= Lp; k <<= 1) {
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int i = threadIdx.x; i < Lp; i += 256) {
        const int ixj = i ^ j;
        if (ixj > i) {
          const int a = sk[i], b = sk[ixj];
          const bool up = (i & k) == 0;
          if ((a > b) == up) {
            sk[i] = b;
            sk[ixj] = a;
          }
        }
      }
      __syncthreads();
    }
  }
  for (int i = threadIdx.x; i < L; i += 256) out[(size_t)u * L + i] = sk[i];
}

// ---- wave-cooperative helpers on one user's candidate buffer ---------------------------------
// Exact K-th largest key of b[0..c) (c > K, keys unique) by counting quickselect, then keep the K
// keys >= it (unsorted) in b[0..K). Returns the K-th key. Wave-uniform arguments.
__device__ __forceinline__ unsigned long long shrink_user(unsigned long long* b, int c, int K, int lane) {
  const unsigned long long k0 = lane < c ? b[lane] : 0ull;
  const unsigned long long k1 = lane + 64 < c ? b[lane + 64] : 0ull;
  unsigned long long lo = 0ull, hi = ~0ull, kth = 0ull;
  for (int it = 0; it < 2 * TK_CAP; ++it) {
    const unsigned long long m0 = __ballot(k0 > lo && k0 < hi);
    const unsigned long long m1 = __ballot(k1 > lo && k1 < hi);
    unsigned long long pivot;
    if (m0) pivot = __shfl(k0, __ffsll((long long)m0) - 1, 64);
    else pivot = __shfl(k1, __ffsll((long long)m1) - 1, 64);
    const int g = __popcll(__ballot(k0 > pivot)) + __popcll(__ballot(k1 > pivot));
    if (g == K - 1) {
      kth = pivot;
      break;
    }
    if (g > K - 1) lo = pivot;  // the K-th largest is above the pivot
    else hi = pivot;
  }
  const bool keep0 = k0 >= kth && k0 != 0ull, keep1 = k1 >= kth && k1 != 0ull;
  const unsigned long long mk0 = __ballot(keep0), mk1 = __ballot(keep1);
  const unsigned long long lt = (1ull << lane) - 1ull;
  const int pos0 = __popcll(mk0 & lt), pos1 = __popcll(mk0) + __popcll(mk1 & lt);
  __builtin_amdgcn_wave_barrier();
  if (keep0) b[pos0] = k0;
  if (keep1) b[pos1] = k1;
  __builtin_amdgcn_wave_barrier();
  return kth;
}

// Sort b[0..c), c <= 64, best first (rank = number of larger keys).
__device__ __forceinline__ void sort_small(unsigned long long* b, int c, int lane) {
  const unsigned long long k0 = lane < c ? b[lane] : 0ull;
  int r0 = 0;
  for (int m = 0; m < c; ++m) r0 += (b[m] > k0) ? 1 : 0;
  __builtin_amdgcn_wave_barrier();
  if (lane < c) b[r0] = k0;
  __builtin_amdgcn_wave_barrier();
}

#define TK_ESTRIDE 68  // floats per staged table row (64 + 4 pad: conflict-free ds_read_b128 fragments)

// Table tiles are staged global -> VGPR -> LDS (an LDS-DMA ring was tried, but with ds_write / LDS
// atomics in the same loop hipcc drains vmcnt(0) in front of every fragment read).

__global__ __launch_bounds__(256) void item_topk_kernel(TopkParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* etile = reinterpret_cast<float*>(smem);                      // [2][32*68]
  float* btile = etile + 2 * 32 * TK_ESTRIDE;                         // [2][32]
  unsigned long long* buf = reinterpret_cast<unsigned long long*>(btile + 64);  // [128][CAP]
  unsigned long long* thr = buf + TK_USERS * TK_CAP;                           // [128]
  int* cnt = reinterpret_cast<int*>(thr + TK_USERS);                           // [128]

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
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
  // sorted-history cursor of this lane's user (kept by the half-0 lane): first entry >= chunk start
  const bool walker = p.exclude && half == 0 && user_ok;
  const int32_t* hs = p.hist_sorted + (size_t)(user_ok ? user : 0) * p.L;
  int hptr = 0;
  if (walker) {
    int lo = 0, hi = p.L;
    const int first_item = tile_begin * 32;
    while (lo < hi) {
      const int mid = (lo + hi) >> 1;
      if (hs[mid] < first_item) lo = mid + 1;
      else hi = mid;
    }
    hptr = lo;
  }
  // next history id at or after the cursor, kept in a register so that a tile without history
  // items costs one compare (no dependent global load per tile)
  int hnext = (walker && hptr < p.L) ? hs[hptr] : INT_MAX;
  // staging: 512 float4 per tile, thread handles float4 #tid and #tid+256. FOUR tiles are in flight in
  // registers (32 KiB per workgroup: the table streams from HBM / Infinity Cache with ~3 us latency when
  // every CU pulls a different chunk), one is in LDS being multiplied. Tile t lives in register set
  // (t - tile_begin) & 3; the tile loop is unrolled by 4 so the set index is a compile-time constant.
  float4 e0[4], e1[4];
  float bb[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    e0[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    e1[i] = e0[i];
    bb[i] = 0.f;
  }
#define TK_LOAD(set_, tile_)                                                                  \
  do {                                                                                        \
    const float4* src_ = reinterpret_cast<const float4*>(p.emb + (size_t)(tile_)*32 * 64);    \
    e0[set_] = src_[tid];                                                                     \
    e1[set_] = src_[tid + 256];                                                               \
    bb[set_] = (tid < 32) ? p.bias[(tile_)*32 + tid] : 0.f;                                   \
  } while (0)
#define TK_STORE(set_, bufi_)                                                                       \
  do {                                                                                              \
    float* e_ = etile + (bufi_)*32 * TK_ESTRIDE;                                                    \
    *reinterpret_cast<float4*>(e_ + (tid >> 4) * TK_ESTRIDE + (tid & 15) * 4) = e0[set_];           \
    *reinterpret_cast<float4*>(e_ + ((tid + 256) >> 4) * TK_ESTRIDE + (tid & 15) * 4) = e1[set_];   \
    if (tid < 32) btile[(bufi_)*32 + tid] = bb[set_];                                               \
  } while (0)
  if (tile_begin < tile_end) {
    TK_LOAD(0, tile_begin);
    TK_STORE(0, 0);
  }
#pragma unroll
  for (int i = 1; i < 4; ++i)
    if (tile_begin + i < tile_end) TK_LOAD(i, tile_begin + i);
  __syncthreads();

  // body of one tile; `su` = (tile - tile_begin) & 3 is a constant after unrolling
  auto do_tile = [&](int tile, const int su) {
    const int cur = (tile - tile_begin) & 1;
    if (tile + 4 < tile_end) TK_LOAD(su, tile + 4);  // set `su` was emptied into LDS one tile ago

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
    // threshold and bias of my 16 rows (rows 8g + 4*half + 0..3 = one aligned float4 per g), fetched
    // before the MFMA chain so their LDS latency is hidden
    const unsigned long long th = thr[u_local];
    float4 bias4[4];
#pragma unroll
    for (int g = 0; g < 4; ++g)
      bias4[g] = *reinterpret_cast<const float4*>(btile + cur * 32 + 8 * g + 4 * half);
    floatx16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
#pragma unroll
    for (int s = 0; s < 32; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s], bq[s], acc, 0, 0, 0);

    // which of this tile's 32 items are masked for my user (history ids and the pad id 0)
    unsigned hmask = 0u;
    if (p.exclude) {
      if (walker) {
        const int t0 = tile * 32, t1 = t0 + 32;
        while (hnext < t1) {
          hmask |= 1u << (hnext - t0);
          ++hptr;
          hnext = hptr < p.L ? hs[hptr] : INT_MAX;
        }
        if (tile == 0) hmask |= 1u;
      }
      hmask = __shfl(hmask, col, 64);  // the half-0 lane of the user tells its half-1 twin
    }

    // lane holds, for its user `col`, the scores of items row(r) = (r&3) + 8*(r>>2) + 4*half.
    // Two passes: (1) scores + mask + float pre-test for all 16 registers with no LDS traffic inside
    // (bias fetched as 4 x b128 beforehand); (2) the rare survivors take the exact-key path.
    float sc[16];
    unsigned pass = 0u;
    {
      const float thf = th ? lr_key_score(th) : -__builtin_inff();
      const bool tail = (tile + 1) * 32 > p.n_rows;  // only the last tile can hold rows past V
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * half;
        const float4 b4 = bias4[r >> 2];
        float v = acc[r] + ((r & 3) == 0 ? b4.x : (r & 3) == 1 ? b4.y : (r & 3) == 2 ? b4.z : b4.w);
        if ((hmask >> row) & 1u) v = LR_MASK_SCORE;
        sc[r] = v;
        bool ok = v >= thf;
        if (tail) ok = ok && (tile * 32 + row < p.n_rows);
        pass |= ok ? (1u << r) : 0u;
      }
      if (!user_ok) pass = 0u;
    }
    if (__any(pass != 0u)) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        if ((pass >> r) & 1u) {
          const int row = (r & 3) + 8 * (r >> 2) + 4 * half;
          const unsigned long long key = lr_rank_key(sc[r], (uint32_t)(tile * 32 + row));
          if (key > th) {
            const int pos = atomicAdd(&cnt[u_local], 1);
            buf[u_local * TK_CAP + pos] = key;
          }
        }
      }
    }
    __builtin_amdgcn_wave_barrier();
    // keep room for the next tile (at most 32 insertions per user per tile)
    {
      const int c = cnt[u_local];
      if (__any(c > TK_CAP - 32)) {
        for (int uu = 0; uu < 32; ++uu) {
          const int ul = wave * 32 + uu;
          const int cu = cnt[ul];  // wave-uniform (broadcast read)
          if (cu > TK_CAP - 32) {
            const unsigned long long kth = shrink_user(buf + ul * TK_CAP, cu, K, lane);
            if (lane == 0) {
              cnt[ul] = K;
              thr[ul] = kth;
            }
          }
        }
        __builtin_amdgcn_wave_barrier();
      }
    }
    if (tile + 1 < tile_end) TK_STORE((su + 1) & 3, cur ^ 1);
    __syncthreads();
  };
  for (int base = tile_begin; base < tile_end; base += 4) {
#pragma unroll
    for (int u = 0; u < 4; ++u)
      if (base + u < tile_end) do_tile(base + u, u);
  }
#undef TK_LOAD
#undef TK_STORE

  // final: best K of every user's buffer, sorted, as the chunk's partial list
  for (int uu = 0; uu < 32; ++uu) {
    const int ul = wave * 32 + uu;
    const int gu = blockIdx.y * TK_USERS + ul;
    if (gu >= p.B) break;  // wave-uniform
    int cu = cnt[ul];
    if (cu > K) {
      shrink_user(buf + ul * TK_CAP, cu, K, lane);
      cu = K;
    }
    sort_small(buf + ul * TK_CAP, cu, lane);
    unsigned long long* dst = p.partial + ((size_t)gu * p.n_chunks + chunk) * K;
    for (int j = lane; j < K; j += 64) dst[j] = j < cu ? buf[ul * TK_CAP + j] : 0ull;
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
  if (p.n_chunks == 1) {  // already sorted: just unpack
    for (int j = lane; j < p.K; j += 64) {
      const unsigned long long m = base[j];
      p.out_idx[(size_t)user * p.K + j] = m ? (int32_t)lr_key_item(m) : -1;
      if (p.out_score) p.out_score[(size_t)user * p.K + j] = m ? lr_key_score(m) : -__builtin_inff();
    }
    return;
  }
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

// Split the item tiles into chunks so that (user tiles x chunks) fills the 256 CUs in whole rounds:
// minimise rounds x (tiles per chunk + the per-chunk finalisation, ~8 tile-times).
static void topk_geometry(int n_tiles, int B, int* n_chunks, int* tiles_per_chunk) {
  const int n_ut = (B + TK_USERS - 1) / TK_USERS;
  int max_chunks = TK_MAX_WGS / n_ut;
  if (max_chunks > TK_MAX_CHUNKS) max_chunks = TK_MAX_CHUNKS;
  if (max_chunks > n_tiles) max_chunks = n_tiles;
  if (max_chunks < 1) max_chunks = 1;
  long best_cost = -1;
  int best = 1;
  for (int c = 1; c <= max_chunks; ++c) {
    const int tpc = (n_tiles + c - 1) / c;
    const int real = (n_tiles + tpc - 1) / tpc;
    const long rounds = ((long)n_ut * real + 255) / 256;
    const long cost = rounds * (tpc + 8);
    if (best_cost < 0 || cost < best_cost) {
      best_cost = cost;
      best = c;
    }
  }
  *tiles_per_chunk = (n_tiles + best - 1) / best;
  *n_chunks = (n_tiles + *tiles_per_chunk - 1) / *tiles_per_chunk;
}

static size_t partial_bytes_max(int B, int K) {
  // user tiles x chunks <= TK_MAX_WGS (+ one chunk minimum) => B x chunks <= max(B, 128 * TK_MAX_WGS)
  size_t rows = (size_t)B;
  if (rows < (size_t)TK_USERS * TK_MAX_WGS) rows = (size_t)TK_USERS * TK_MAX_WGS;
  return lr_align_up(rows * K * sizeof(unsigned long long), 256);
}

size_t lr_topk_workspace_bytes(int B, int K, int L) {
  return partial_bytes_max(B, K) + lr_align_up((size_t)B * (L > 0 ? L : 1) * sizeof(int32_t), 256);
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
  p.B = B;
  p.L = L;
  p.K = K;
  p.exclude = exclude_history ? 1 : 0;
  topk_geometry(p.n_tiles, B, &p.n_chunks, &p.tiles_per_chunk);
  const size_t need_partial = lr_align_up((size_t)B * p.n_chunks * K * sizeof(unsigned long long), 256);
  const size_t need_hist = p.exclude ? lr_align_up((size_t)B * L * sizeof(int32_t), 256) : 0;
  if (need_partial + need_hist > ws_bytes)
    LR_FAIL(LR_EWORKSPACE, "top-K workspace: need %zu bytes, have %zu", need_partial + need_hist, ws_bytes);
  p.partial = reinterpret_cast<unsigned long long*>(ws);
  int32_t* hist_sorted = reinterpret_cast<int32_t*>(reinterpret_cast<char*>(ws) + need_partial);
  p.hist_sorted = p.exclude ? hist_sorted : nullptr;

  LrProfScope prof(LR_PROF_ITEM_TOPK, 2.0 * 64 * (double)p.n_rows * B, st);
  if (p.exclude) {
    int Lp = 2;
    while (Lp < L) Lp <<= 1;
    if (Lp > 8192) LR_FAIL(LR_EUNSUPPORTED, "history length %d > 8192 is not supported by the mask pre-pass", L);
    hipLaunchKernelGGL(hist_sort_kernel, dim3(B), dim3(256), (size_t)Lp * sizeof(int32_t), st, ids, L, Lp, p.n_rows,
                       hist_sorted);
    LR_CHECK_LAUNCH("hist_sort_kernel");
  }
  const size_t lds = (2 * 32 * TK_ESTRIDE + 64) * sizeof(float) +
                     (size_t)TK_USERS * TK_CAP * 8 + TK_USERS * 8 + TK_USERS * 4;
  static bool attr_set = false;
  if (!attr_set) {
    LR_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(item_topk_kernel),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    attr_set = true;
  }
  dim3 grid(p.n_chunks, (B + TK_USERS - 1) / TK_USERS);
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
