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
#include <stdlib.h>

#include "lr_common.h"
#include "lr_profile.h"

typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef float floatx4 __attribute__((ext_vector_type(4)));

#define TK_WAVES 4
#define TK_USERS 128      // users per workgroup
#define TK_CAP 128        // candidate slots per user (K <= 64 kept + insertion slack)
#define TK_BSTRIDE 129    // slot stride between users (odd: lanes of different users hit different banks)
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
  const float* thresh;          // [B] proven lower bounds of every user's K-th best eligible score (bound pre-pass), or null
  const int* run_flag;          // null: always run. Else the kernel is the FALLBACK of the candidate path and runs only
                                // if *run_flag != 0 (some user's candidate list overflowed)
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

// The same for L <= 64 (every BASELINE shape but ML-100k): one WAVE per user, the 64 ids in one register per lane, a
// bitonic network of 21 compare-exchange steps over __shfl_xor -- no LDS, no barriers, 4 users per workgroup.
__global__ __launch_bounds__(256) void hist_sort_wave_kernel(const int64_t* ids, int B, int L, int n_rows, int32_t* out) {
  const int lane = threadIdx.x & 63;
  const int u = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (u >= B) return;
  const long long v = lane < L ? ids[(size_t)u * L + lane] : -1;
  int x = (v >= 0 && v < n_rows) ? (int)v : INT_MAX;
#pragma unroll
  for (int k = 2; k <= 64; k <<= 1) {
#pragma unroll
    for (int j = k >> 1; j > 0; j >>= 1) {
      const int y = __shfl_xor(x, j, 64);
      const bool up = (lane & k) == 0;           // ascending block
      const bool lower = (lane & j) == 0;        // this lane keeps the smaller of the pair in an ascending block
      x = (lower == up) ? min(x, y) : max(x, y);
    }
  }
  if (lane < L) out[(size_t)u * L + lane] = x;
}

// ---- wave-cooperative helpers on one user's candidate buffer ---------------------------------
// A user's candidates live in two lane-private lists (one per MFMA lane half, appended without
// atomics): b[0..c0) and b[64..64+c1), keys unique and non-zero. Keep the best min(c0+c1, K): the exact
// K-th largest key comes from a counting quickselect over ballots. split = true deals the survivors
// back to the two lists (the first (n+1)/2 to list 0), otherwise they are stored from b[0] on.
// Returns the K-th key (0 while fewer than K candidates exist: no threshold yet); *n_out = survivors.
// Wave-uniform arguments.
__device__ __forceinline__ unsigned long long compact_user(unsigned long long* b, int c0, int c1, int K, int lane,
                                                           bool split, int* n_out) {
  const unsigned long long k0 = lane < c0 ? b[lane] : 0ull;
  const unsigned long long k1 = lane < c1 ? b[64 + lane] : 0ull;
  unsigned long long kth = 0ull;
  if (c0 + c1 >= K) {
    unsigned long long lo = 0ull, hi = ~0ull;
    for (int it = 0; it < 2 * TK_CAP; ++it) {
      const unsigned long long m0 = __ballot(k0 > lo && k0 < hi);
      const unsigned long long m1 = __ballot(k1 > lo && k1 < hi);
      // pivot = first live key (v_readlane with a uniform lane index; no LDS round trip)
      const unsigned long long src = m0 ? k0 : k1;
      const int pl = __ffsll((long long)(m0 ? m0 : m1)) - 1;
      const unsigned long long pivot =
          ((unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)(src >> 32), pl) << 32) |
          (unsigned)__builtin_amdgcn_readlane((int)(unsigned)src, pl);
      const int g = __popcll(__ballot(k0 > pivot)) + __popcll(__ballot(k1 > pivot));
      if (g == K - 1) {
        kth = pivot;
        break;
      }
      if (g > K - 1) lo = pivot;  // the K-th largest is above the pivot
      else hi = pivot;
    }
  }
  const bool keep0 = k0 >= kth && k0 != 0ull, keep1 = k1 >= kth && k1 != 0ull;
  const unsigned long long mk0 = __ballot(keep0), mk1 = __ballot(keep1);
  const unsigned long long lt = (1ull << lane) - 1ull;
  const int n = __popcll(mk0) + __popcll(mk1);
  const int n0 = split ? (n + 1) >> 1 : n;
  const int g0 = __popcll(mk0 & lt), g1 = __popcll(mk0) + __popcll(mk1 & lt);
  __builtin_amdgcn_wave_barrier();
  if (keep0) b[g0 < n0 ? g0 : 64 + g0 - n0] = k0;
  if (keep1) b[g1 < n0 ? g1 : 64 + g1 - n0] = k1;
  __builtin_amdgcn_wave_barrier();
  *n_out = n;
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

// Diagnostic stamps: STAMP = true is instantiated only in a -DLR_EXPERIMENTS build (make EXPERIMENTS=1, then
// LR_TOPK_STAMPS=1 at run time; tools/topk_stamps.py); the product library holds no stamping code. s_memtime
// deltas of the tile loop's segments, summed over the loop, for wave 0 of the first 16 workgroups.
__device__ unsigned long long g_topk_stamps[16 * 8];
#define TK_STAMP(slot)                                                         \
  if (STAMP) {                                                                 \
    unsigned long long t_;                                                     \
    __builtin_amdgcn_sched_barrier(0);                                         \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory"); \
    __builtin_amdgcn_sched_barrier(0);                                         \
    stamp_acc[(slot)] += t_ - t_prev;                                          \
    t_prev = t_;                                                               \
  }

template <bool STAMP>
__global__ __launch_bounds__(256) void item_topk_kernel(TopkParams p) {
  unsigned long long stamp_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, t_prev = 0;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* etile = reinterpret_cast<float*>(smem);                      // [2][32*68]
  float* btile = etile + 2 * 32 * TK_ESTRIDE;                         // [4][32]
  unsigned long long* buf = reinterpret_cast<unsigned long long*>(btile + 128);  // [128][BSTRIDE]

  if (p.run_flag && *p.run_flag == 0) return;  // wave-uniform
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
  // candidate list of this lane (its user's list `half`), its fill count, and the user's threshold score
  // (the K-th best seen at the last compaction; +inf for users past B: nothing ever passes)
  unsigned long long* mine = buf + u_local * TK_BSTRIDE + half * 64;
  int pos = 0;
  float thf = user_ok ? (p.thresh ? p.thresh[user] : -__builtin_inff()) : __builtin_inff();
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
  // the next two history ids at or after the cursor live in registers (h0, h1): a tile without history
  // items costs one compare, and when h0 is consumed the id after h1 is fetched a hit ahead of its use
  int h0 = (walker && hptr < p.L) ? hs[hptr] : INT_MAX;
  int h1 = (walker && hptr + 1 < p.L) ? hs[hptr + 1] : INT_MAX;
  // staging: 512 float4 per tile, thread handles float4 #tid and #tid+256. Tiles t+2 .. t+5 are in flight in
  // registers while tile t is multiplied (32 KiB per workgroup: the table streams from HBM / Infinity
  // Cache with ~3 us latency when every CU pulls a different chunk); tile T lives in register set
  // (T - tile_begin) & 3, LDS table buffer (T - tile_begin) & 1 and bias slot (T - tile_begin) & 3. The
  // tile loop is unrolled by 4 so all of these are compile-time constants.
  // The loads are inline asm with hand-counted s_waitcnt: hipcc's own counter tracking gives up on this
  // loop (the history walker's load sits in a data-dependent inner loop) and would drain vmcnt(0) before
  // every LDS store, i.e. expose the full memory latency once per tile. Every wave issues exactly three
  // loads per tile (table x2, bias), tile indices are clamped instead of guarded, so "all but the newest
  // 9" is the same for every wave and every iteration.
  floatx4 e0[4], e1[4];
  float bb[4];
  const unsigned voff0 = (unsigned)tid * 16u, voff1 = voff0 + 4096u, voffb = (unsigned)(tid & 31) * 4u;
#define TK_LOAD(set_, tile_)                                                                          \
  do {                                                                                                \
    const int t_ = min((tile_), p.n_tiles - 1);                                                       \
    const float* se_ = p.emb + (size_t)t_ * 32 * 64;                                                  \
    const float* sb_ = p.bias + (size_t)t_ * 32;                                                      \
    asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(e0[set_]) : "v"(voff0), "s"(se_) : "memory"); \
    asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(e1[set_]) : "v"(voff1), "s"(se_) : "memory"); \
    asm volatile("global_load_dword %0, %1, %2" : "=v"(bb[set_]) : "v"(voffb), "s"(sb_) : "memory");   \
  } while (0)
#define TK_WAIT_LOADS(n_) asm volatile("s_waitcnt vmcnt(" #n_ ")" ::: "memory")
#define TK_STORE(set_)                                                                              \
  do {                                                                                              \
    float* e_ = etile + ((set_)&1) * 32 * TK_ESTRIDE;                                               \
    *reinterpret_cast<floatx4*>(e_ + (tid >> 4) * TK_ESTRIDE + (tid & 15) * 4) = e0[set_];          \
    *reinterpret_cast<floatx4*>(e_ + ((tid + 256) >> 4) * TK_ESTRIDE + (tid & 15) * 4) = e1[set_];  \
    if (tid < 32) btile[(set_)*32 + tid] = bb[set_];                                                \
  } while (0)
  // A fragments of the tile in table buffer bufi_: item row `col`, k = 32*half + s
#define TK_FRAGS(dst_, bufi_)                                                                       \
  do {                                                                                              \
    const float* er_ = etile + (bufi_)*32 * TK_ESTRIDE + col * TK_ESTRIDE + 32 * half;              \
    _Pragma("unroll") for (int j_ = 0; j_ < 8; ++j_) {                                              \
      const float4 v_ = *reinterpret_cast<const float4*>(er_ + 4 * j_);                             \
      dst_[4 * j_ + 0] = v_.x;                                                                      \
      dst_[4 * j_ + 1] = v_.y;                                                                      \
      dst_[4 * j_ + 2] = v_.z;                                                                      \
      dst_[4 * j_ + 3] = v_.w;                                                                      \
    }                                                                                               \
  } while (0)
#pragma unroll
  for (int i = 0; i < 4; ++i) TK_LOAD(i, tile_begin + i);
  TK_WAIT_LOADS(6);
  TK_STORE(0);
  TK_STORE(1);
  TK_LOAD(0, tile_begin + 4);
  // everything hipcc tracks has landed before the loop, or it re-waits (vmcnt(0)) at the top of every tile
#pragma unroll
  for (int j = 0; j < 32; ++j) asm volatile("" ::"v"(bq[j]));
  asm volatile("" ::"v"(h0), "v"(h1));
  __syncthreads();
  float afr[2][32];
  TK_FRAGS(afr[0], 0);

  // Software pipeline, three tiles deep. While the 32-MFMA chain of tile t runs (each MFMA depends on
  // the one before: 64-cycle shadows), the same wave
  //   * filters tile t-1: one saved score per two MFMAs, interleaved by hand; a score that reaches the
  //     user's threshold is appended to the lane's own candidate list (no atomics);
  //   * has tile t+1's A fragments on their way from LDS into the other fragment register set;
  //   * walks the sorted history for tile t's mask (needed one iteration later).
  // Two accumulator sets and two bias sets alternate between tiles: tile t's chain runs in accs[t & 1] while the filter
  // reads the finished scores of tile t - 1 straight from accs[(t - 1) & 1] (+ that tile's bias), so the next chain's
  // first MFMA issues right behind the previous chain's last one -- no copy-out of 16 scores between two chains.
  floatx16 accs[2];
  float4 bias4s[2][4];
  unsigned hmaskp = 0u;
  int tilep = 0;
  bool prev_ok = false;
#pragma unroll
  for (int r = 0; r < 16; ++r) accs[0][r] = accs[1][r] = 0.f;
#pragma unroll
  for (int g = 0; g < 4; ++g) bias4s[0][g] = bias4s[1][g] = make_float4(0.f, 0.f, 0.f, 0.f);

#define TK_BIAS(B4, r) ((r & 3) == 0 ? B4[(r) >> 2].x : (r & 3) == 1 ? B4[(r) >> 2].y : (r & 3) == 2 ? B4[(r) >> 2].z : B4[(r) >> 2].w)
// One finished score of the previous tile. On gfx950 the f32-input MFMA runs on the f32 vector datapath, so vector ALU
// work between the chained MFMAs is NOT free (timing ablation: the 12-instruction filter per element costs as much
// as 0.7 of the chain). The common case is therefore two instructions -- bias add, threshold compare -- and a
// wave-uniform branch: with a seeded or settled threshold about one element in three has ANY of its 64 lanes pass.
// Only then: history mask (a bit test on hm = the tile's mask shifted to this lane half; a masked item keeps the
// reference's score -1e9, trainer/lru.py:37-38, and is re-tested), rank key, append to the lane's list. Padding rows of
// the last tile carry a NaN bias (lr_lru_pack): NaN >= threshold is false.
#define TK_ELEM(ACC, B4, r, thv)                                                                        \
  {                                                                                                     \
    const float s_ = ACC[r] + TK_BIAS(B4, r);                                                           \
    if (__ballot(s_ >= (thv)) != 0ull) {                                                                \
      const int rowc_ = ((r)&3) + 8 * ((r) >> 2); /* row = rowc_ + 4 * half */                          \
      const float v_ = ((hm >> rowc_) & 1u) ? LR_MASK_SCORE : s_;                                       \
      const bool ok_ = v_ >= (thv);                                                                     \
      if (STAMP) stamp_acc[7] += __popcll(__ballot(ok_));                                               \
      if (ok_) {                                                                                        \
        mine[pos] = lr_rank_key(v_, (uint32_t)(tilep * 32 + rowc_ + 4 * half));                         \
        ++pos;                                                                                          \
      }                                                                                                 \
    }                                                                                                   \
  }
  // after a tile's inserts: compact every user whose list could overflow on the next tile (a lane
  // appends <= 16 keys per tile and a list holds 64), which also refreshes the user's threshold
  auto make_room = [&]() {
    const unsigned long long need = __ballot(pos > 48);
    if (need) {
      unsigned m = (unsigned)need | (unsigned)(need >> 32);
      do {
        const int uu = __ffs(m) - 1;  // wave-uniform
        m &= m - 1u;
        const int c0 = __builtin_amdgcn_readlane(pos, uu), c1 = __builtin_amdgcn_readlane(pos, uu + 32);
        int n;
        if (STAMP) stamp_acc[6] += 1;
        const unsigned long long kth = compact_user(buf + (wave * 32 + uu) * TK_BSTRIDE, c0, c1, K, lane, true, &n);
        if (col == uu) {
          pos = half ? n - ((n + 1) >> 1) : (n + 1) >> 1;
          if (kth) thf = lr_key_score(kth);
        }
      } while (m);
    }
  };
  // which of `tile`'s 32 items are masked for my user (history ids and the pad id 0). Called late in the
  // MFMA chain, see the note on the wait below.
  auto walk = [&](int tile) -> unsigned {
    unsigned hmask = 0u;
    if (p.exclude) {
      if (walker) {
        const int t0 = tile * 32, t1 = t0 + 32;
        while (h0 < t1) {
          hmask |= 1u << (h0 - t0);
          h0 = h1;
          ++hptr;
          h1 = hptr + 1 < p.L ? hs[hptr + 1] : INT_MAX;
        }
        if (tile == 0) hmask |= 1u;
      }
      // pin hipcc's wait for the look-ahead load HERE (late in the tile: the table loads issued at the top of
      // the tile have had ~2000 cycles); left alone it waits where it next copies h1 -- at the top of the next
      // tile, right behind the freshly issued table loads
      asm volatile("" ::"v"(h1));
      hmask = __shfl(hmask, col, 64);  // the half-0 lane of the user tells its half-1 twin
    }
    return hmask;
  };

  // body of one tile; `su` = (tile - tile_begin) & 3 is a constant after unrolling
  auto do_tile = [&](int tile, const int su) {
    TK_LOAD((su + 1) & 3, tile + 5);  // that set went to LDS one tile ago
    // tile + 2 (loaded three tiles ago) goes to the LDS buffer of tile `tile`, whose fragments every wave read during the
    // previous tile (the barrier that closed it separates those reads from this store); the barrier that closes THIS
    // tile publishes it for the fragment reads of the next one. Up here the stores drain in the chain's shadow.
    TK_WAIT_LOADS(9);  // tile + 2 has landed; tiles + 3, + 4, + 5 may still be in flight
    TK_STORE((su + 2) & 3);
    if (tile + 1 < tile_end) TK_FRAGS(afr[(su + 1) & 1], (su + 1) & 1);
    // this tile's bias rows (rows 8g + 4*half + 0..3 = one aligned float4 per g), added when the scores are filtered
#pragma unroll
    for (int g = 0; g < 4; ++g)
      bias4s[su & 1][g] = *reinterpret_cast<const float4*>(btile + su * 32 + 8 * g + 4 * half);
    TK_STAMP(0)

    const float thv = prev_ok ? thf : __builtin_inff();
    const unsigned hm = hmaskp >> (4 * half);
    unsigned hmaskn = 0u;
#pragma unroll
    for (int r = 0; r < 16; ++r) accs[su & 1][r] = 0.0f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      accs[su & 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(afr[su & 1][2 * r], bq[2 * r], accs[su & 1], 0, 0, 0);
      accs[su & 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(afr[su & 1][2 * r + 1], bq[2 * r + 1], accs[su & 1], 0, 0, 0);
      TK_ELEM(accs[(su + 1) & 1], bias4s[(su + 1) & 1], r, thv)
      if (r == 15) hmaskn = walk(tile);
    }
    if (STAMP) asm volatile("" ::"v"(accs[su & 1]));
    TK_STAMP(2)
    make_room();
    TK_STAMP(3)
    hmaskp = hmaskn;
    tilep = tile;
    prev_ok = true;
    TK_STAMP(4)
    __syncthreads();
    TK_STAMP(5)
  };
  if (STAMP) asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_prev)::"memory");
  for (int base = tile_begin; base < tile_end; base += 4) {
#pragma unroll
    for (int u = 0; u < 4; ++u)
      if (base + u < tile_end) do_tile(base + u, u);
  }
  if (prev_ok) {  // drain the pipeline: the last tile's scores sit in the set of its parity
    const unsigned hm = hmaskp >> (4 * half);
    if ((tilep - tile_begin) & 1) {
#pragma unroll
      for (int r = 0; r < 16; ++r) TK_ELEM(accs[1], bias4s[1], r, thf)
    } else {
#pragma unroll
      for (int r = 0; r < 16; ++r) TK_ELEM(accs[0], bias4s[0], r, thf)
    }
  }
#undef TK_ELEM
#undef TK_BIAS
#undef TK_LOAD
#undef TK_STORE
#undef TK_FRAGS
#undef TK_WAIT_LOADS

  // final: best K of every user's two lists, sorted, as the chunk's partial list
  __builtin_amdgcn_wave_barrier();
  for (int uu = 0; uu < 32; ++uu) {
    const int ul = wave * 32 + uu;
    const int gu = blockIdx.y * TK_USERS + ul;
    if (gu >= p.B) break;  // wave-uniform
    const int c0 = __builtin_amdgcn_readlane(pos, uu), c1 = __builtin_amdgcn_readlane(pos, uu + 32);
    int cu;
    compact_user(buf + ul * TK_BSTRIDE, c0, c1, K, lane, false, &cu);
    sort_small(buf + ul * TK_BSTRIDE, cu, lane);
    unsigned long long* dst = p.partial + ((size_t)gu * p.n_chunks + chunk) * K;
    for (int j = lane; j < K; j += 64) dst[j] = j < cu ? buf[ul * TK_BSTRIDE + j] : 0ull;
  }
  if (STAMP) {
    TK_STAMP(6)  // drain + final compaction / sort / write-out
    const int wg = blockIdx.y * gridDim.x + blockIdx.x;
    if (wg < 16 && tid == 0) {
#pragma unroll
      for (int i = 0; i < 8; ++i) g_topk_stamps[wg * 8 + i] = stamp_acc[i];
    }
  }
}

#ifdef LR_EXPERIMENTS
extern "C" int lr_debug_topk_stamps(unsigned long long* out, int n) {
  if (!out || n < 1 || n > 16 * 8) LR_FAIL(LR_EINVAL, "lr_debug_topk_stamps: bad arguments");
  LR_CHECK_HIP(hipMemcpyFromSymbol(out, HIP_SYMBOL(g_topk_stamps), (size_t)n * sizeof(unsigned long long)));
  return LR_OK;
}
#endif

// =============================================================================================
// Bound pre-pass (small catalogs): a proven lower bound of every user's K-th best score, 16x cheaper than the scores
// =============================================================================================
// A user's threshold in item_topk_kernel only reaches the K/n quantile after n items, so a chunk pays ~ K ln(n / K)
// list inserts and a compaction per ~23 of them per lane -- on a catalog that is ONE chunk (Beauty: 378 tiles) that
// overhead is 3x the MFMA chain itself. The pre-pass hands the exact pass a threshold that is already (almost) final:
//   1. item_bound_kernel scores every (user, item) APPROXIMATELY on v_mfma_f32_32x32x16_bf16 (bf16 copies of q and of
//      the table: 4 MFMAs of 32 cycles per 32 x 32 tile instead of 32 of 64 cycles) and keeps only each tile's maximum
//      per user: tmax[user][tile].
//   2. bound_select_kernel takes m = the R-th largest tile maximum, R = K + (number of masked ids of the user): the R
//      tiles hold R DIFFERENT items with approximate score >= m, at most R - K of them masked, so at least K eligible
//      items have exact score >= m - delta, where delta bounds |approximate - exact| for every item of that user:
//          |s~ - s| <= ((2u + u^2) + 2 gamma_65) sum_k |q_k e_k| + 2 gamma_66 (sum_k |q_k e_k| + |b|) + 2^-23 |s|,
//                                                                                  u = 2^-8, gamma_n ~ n * 2^-24
//                   <= 0.00785 ||q||_2 E_max + 1e-5 (||q||_2 E_max + B_max)              (Cauchy-Schwarz; E_max = max row
//      norm of the table, B_max = max |bias|, both rounded up at pack time). Since round 4 the bias is the START value
//      of the MFMA accumulation (its C operand) instead of a vector add behind it: the approximate score is a 66-term
//      fp32 sum in the matrix pipe's own order, hence the gamma_66 term on |b| too (taken twice: the pipe's internal
//      alignment is not IEEE add-by-add); the exact side contributes gamma_64 on the products and one rounding of s. T = m - delta (rounded down) is therefore
//      <= the user's true K-th best eligible score: filtering the exact scores by `>= T` loses nothing, the ranked
//      output is bit-identical, and only ~1.2 K items per user pass instead of ~K ln(V / K).
// A 32-item tile maximum at rank R sits at item quantile ~ 1 - (1 - R / n_tiles)^(1/32): for Beauty (R = 59 of 378
// tiles) 0.53 % of 12 086 items = 64 candidates per user against 275 inserts without the bound.
#include "lru_topk_bf16.h"

#define TK_BOUND_MAX_TILES 2048  // 32 maxima per lane in bound_select_kernel: one per tile for catalogs up to 65 536 items,
                                 // one per group of 2^gshift tiles beyond (bound_group_shift): the R-th largest GROUP maximum
                                 // still certifies R different items at or above it, so the bound's proof is unchanged; a
                                 // group of 512 items at rank R of 1 954 (1 M items, R = 251) sits at the same item quantile
                                 // (2.7e-4, ~270 items + the 2 delta band) as a 32-item tile at the same rank fraction
#define TK_BOUND_MAX_GSHIFT 6

// One wave per user: T[user] = (R-th largest of tmax[user][0..n_tiles)) - delta, R = K + masked ids; -inf if there are
// fewer than R tiles (no bound: the exact pass then starts from -inf as it does without the pre-pass).
template <int NS>  // NS x 64 >= n_tiles key slots per wave
__global__ __launch_bounds__(256) void bound_select_kernel(const float* tmax, int ld, int n_tiles, const float* q,
                                                           const int64_t* ids, int L, int n_rows, int exclude, int B, int K,
                                                           const float* stats /*[0] E_max, [1] B_max*/, float* thresh,
                                                           float* cand_thresh, int* cand_count, int* overflow_flag) {
  const int lane = threadIdx.x & 63;
  const int user = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (user >= B) return;
  int R = K;
  if (exclude) {  // masked: the pad id 0 and every history id inside the catalog (duplicates only loosen the bound)
    int n = 0;
    for (int t = lane; t < L; t += 64) {
      const long long id = ids[(size_t)user * L + t];
      n += (id > 0 && id < n_rows) ? 1 : 0;
    }
#pragma unroll
    for (int s = 32; s >= 1; s >>= 1) n += __shfl_xor(n, s, 64);
    R += n + 1;
  }
  float qq = q[(size_t)user * 64 + lane];
  qq *= qq;
#pragma unroll
  for (int s = 32; s >= 1; s >>= 1) qq += __shfl_xor(qq, s, 64);
  float T = -__builtin_inff(), Tc = -__builtin_inff();
  if (user == 0 && lane == 0) *overflow_flag = 0;
  if (R <= n_tiles) {
    // monotone uint keys of this user's tile maxima, 32 per lane; R-th largest by bisection on the key bits
    uint32_t key[NS];
#pragma unroll
    for (int i = 0; i < NS; ++i) {
      const int t = lane + 64 * i;
      key[i] = t < n_tiles ? lr_float_ord(tmax[(size_t)user * ld + t]) : 0u;   // 0 sorts below every float, even -inf
    }
    uint32_t ans = 0u;   // the largest x with count(key >= x) >= R
    for (int bit = 31; bit >= 0; --bit) {
      const uint32_t x = ans | (1u << bit);
      int c = 0;   // wave-uniform: ballots + scalar popcounts, no cross-lane shuffles
#pragma unroll
      for (int i = 0; i < NS; ++i) c += __popcll(__ballot(key[i] >= x));
      if (c >= R) ans = x;
    }
    const float m = lr_ord_float(ans);
    const float e_max = stats[0], b_max = stats[1];
    const float qn = sqrtf(qq) * 1.00001f;
    const float delta = 0.00785f * qn * e_max + 1.0e-5f * (qn * e_max + b_max);
    const float t = m - delta * 1.001f - fabsf(m) * 1e-6f;   // every rounding of this line moves T down, never up
    // candidate threshold on the APPROXIMATE scores: a true top-K item has exact score >= T, hence approximate score
    // >= T - delta (item_cand_kernel keeps every item at or above it)
    const float tc = m - 2.0f * delta * 1.001f - fabsf(m) * 2e-6f;
    if (ans != 0u && t == t && fabsf(t) != __builtin_inff() && tc == tc) {   // non-finite inputs: no bound
      T = t;
      Tc = tc;
    }
  }
  if (lane == 0) {
    thresh[user] = T;
    cand_thresh[user] = Tc;
    cand_count[user] = 0;
  }
}

// ---------------------------------------------------------------------------------------------
// Candidate path (the default for small catalogs): with the bound in hand the exact f32 pass over ALL items is not
// needed at all. item_cand_kernel repeats the bf16 scoring and keeps every item whose approximate score reaches
// T - delta (~1.5 K per user: a superset of the true top-K, see bound_select_kernel); cand_rescore_kernel scores just
// those EXACTLY -- lr_item_score's fmaf chain, the bits of the oracle and of item_topk_kernel -- drops masked ids and
// ranks them. If any user's list overflows TK_CAND_CAP (degenerate data: thousands of near-equal scores) a device flag
// turns on the exact full pass (item_topk_kernel + merge, launched behind it with run_flag) for the whole call.
#define TK_RESCORE_HIST 256   // history ids a wave of cand_rescore_kernel stages in LDS (4 KiB per workgroup)
// One wave per user: exact scores of the candidates, masked ids dropped, rank by counting, ordered top-K written.
__global__ __launch_bounds__(256) void cand_rescore_kernel(const float* emb, const float* bias, const float* q,
                                                           const int32_t* hist_sorted, int L, int exclude, int B, int K,
                                                           int n_rows, const int* cand_count, const int32_t* cand,
                                                           const float* thresh, int* overflow_flag,
                                                           int32_t* out_idx, float* out_score) {
  __shared__ float qs[4][64];
  __shared__ int32_t hsl[4][TK_RESCORE_HIST];   // the user's sorted history when L <= TK_RESCORE_HIST (else the search reads global
                                              // memory: 8 dependent loads per candidate -- at L = 200 that was most of the kernel's 170 us)
  __shared__ unsigned long long keys[4][TK_CAND_CAP];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int user = blockIdx.x * 4 + wave;
  if (user >= B) return;
  const int n = cand_count[user];
  if (n > TK_CAND_CAP) {   // the list is incomplete (its slots may never have been written): the exact full pass behind
    if (lane == 0) atomicOr(overflow_flag, 1);   // this kernel redoes the whole call, this user included
    return;
  }
  qs[wave][lane] = q[(size_t)user * 64 + lane];
  const bool hist_in_lds = exclude && L <= TK_RESCORE_HIST;
  if (hist_in_lds)
    for (int t = lane; t < TK_RESCORE_HIST; t += 64) hsl[wave][t] = t < L ? hist_sorted[(size_t)user * L + t] : INT_MAX;
  __builtin_amdgcn_wave_barrier();
  __threadfence_block();
  const int32_t* hs = hist_in_lds ? hsl[wave] : hist_sorted + (size_t)user * L;
  // T = the bound's proven lower limit of this user's K-th best eligible EXACT score (bound_select_kernel): a candidate
  // scoring below it cannot be in the top K, so it never enters the ranking. The candidates were admitted on their
  // approximate scores down to T - delta; about a third of them fall below T, and the ranking is quadratic in their number
  // (it was two thirds of this kernel's 170 us at 1 M items). Survivors are written compacted.
  const float T = thresh ? thresh[user] : -__builtin_inff();
  int m = 0;   // survivors so far (wave-uniform)
  for (int c0 = 0; c0 < n; c0 += 64) {
    const int c = c0 + lane;
    bool ok = false;
    unsigned long long key = 0ull;
    if (c < n) {
      const int item = cand[(size_t)user * TK_CAND_CAP + c];
      bool masked = (unsigned)item >= (unsigned)n_rows;   // (a padding row that passed a -inf threshold; never index past the table)
      if (exclude && !masked) {
        masked = item == 0;
        int lo = 0, hi = L;   // sorted ascending, INT_MAX = unused entry
        while (lo < hi) {
          const int mid = (lo + hi) >> 1;
          const int h = hs[mid];
          if (h < item) lo = mid + 1;
          else hi = mid;
        }
        masked = masked || (lo < L && hs[lo] == item);   // (generic address space: LDS or global, same code)
      }
      if (!masked) {   // lr_item_score(): the one summation order stage 1 uses everywhere (lr_math.h)
        const float4* e4 = reinterpret_cast<const float4*>(emb + (size_t)item * 64);
        const float4* q4 = reinterpret_cast<const float4*>(qs[wave]);
        float ev[64];
#pragma unroll
        for (int j = 0; j < 16; ++j) {
          const float4 v = e4[j];
          ev[4 * j + 0] = v.x; ev[4 * j + 1] = v.y; ev[4 * j + 2] = v.z; ev[4 * j + 3] = v.w;
        }
        float acc = 0.0f;
#pragma unroll
        for (int j = 0; j < 8; ++j) {   // s = 4 j .. 4 j + 3: fma(e[32 + s], q[32 + s], fma(e[s], q[s], acc))
          const float4 qa = q4[j], qb = q4[8 + j];
          acc = __builtin_fmaf(ev[4 * j + 0], qa.x, acc);
          acc = __builtin_fmaf(ev[32 + 4 * j + 0], qb.x, acc);
          acc = __builtin_fmaf(ev[4 * j + 1], qa.y, acc);
          acc = __builtin_fmaf(ev[32 + 4 * j + 1], qb.y, acc);
          acc = __builtin_fmaf(ev[4 * j + 2], qa.z, acc);
          acc = __builtin_fmaf(ev[32 + 4 * j + 2], qb.z, acc);
          acc = __builtin_fmaf(ev[4 * j + 3], qa.w, acc);
          acc = __builtin_fmaf(ev[32 + 4 * j + 3], qb.w, acc);
        }
        const float score = acc + bias[item];
        ok = score >= T;
        key = lr_rank_key(score, (uint32_t)item);
      }
    }
    const unsigned long long bal = __ballot(ok);
    if (ok) keys[wave][m + __popcll(bal & ((1ull << lane) - 1ull))] = key;
    m += __popcll(bal);
  }
  __builtin_amdgcn_wave_barrier();
  __threadfence_block();
  for (int j = m + lane; j < K; j += 64) {   // fewer than K eligible candidates (overflow only): the open slots
    out_idx[(size_t)user * K + j] = -1;
    if (out_score) out_score[(size_t)user * K + j] = -__builtin_inff();
  }
  for (int c = lane; c < m; c += 64) {
    const unsigned long long key = keys[wave][c];
    int rank = 0;
    for (int j = 0; j < m; ++j) rank += keys[wave][j] > key ? 1 : 0;
    if (rank < K) {
      out_idx[(size_t)user * K + rank] = (int32_t)lr_key_item(key);
      if (out_score) out_score[(size_t)user * K + rank] = lr_key_score(key);
    }
  }
}

// ---- merge: one wave per user, K rounds of "largest head wins" over <= 256 sorted lists -----
struct MergeParams {
  const unsigned long long* partial;
  int B, K, n_chunks;
  int32_t* out_idx;
  float* out_score;
  const int* run_flag;  // as in TopkParams
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
  if (p.run_flag && *p.run_flag == 0) return;
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

// Few chunks (n_chunks * K <= TK_MERGE_KEYS): every key's final rank is its index in its own list plus, for each other
// list, the number of keys there that beat it -- a binary search per list over LDS (lists are sorted best first, keys are
// unique, 0 = empty and sorts last). One wave per user, no dependent global loads (the round-by-round kernel above pays
// one per output position).
#define TK_MERGE_KEYS 1024
__global__ __launch_bounds__(256) void topk_merge_small_kernel(MergeParams p) {
  __shared__ unsigned long long keys[4][TK_MERGE_KEYS];
  if (p.run_flag && *p.run_flag == 0) return;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int user = blockIdx.x * 4 + wave;
  if (user >= p.B) return;
  unsigned long long* k = keys[wave];
  const int n = p.n_chunks * p.K;
  const unsigned long long* base = p.partial + (size_t)user * n;
  int total = 0;
  for (int i = lane; i < n; i += 64) {
    const unsigned long long v = base[i];
    k[i] = v;
    total += v != 0ull ? 1 : 0;
  }
#pragma unroll
  for (int s = 32; s >= 1; s >>= 1) total += __shfl_xor(total, s, 64);
  for (int j = total + lane; j < p.K; j += 64) {  // fewer than K candidates in all: the slots no key will claim
    p.out_idx[(size_t)user * p.K + j] = -1;
    if (p.out_score) p.out_score[(size_t)user * p.K + j] = -__builtin_inff();
  }
  __builtin_amdgcn_wave_barrier();
  __threadfence_block();
  for (int i = lane; i < n; i += 64) {
    const unsigned long long key = k[i];
    if (key == 0ull) continue;
    const int c = i / p.K;
    int rank = i - c * p.K;
    for (int c2 = 0; c2 < p.n_chunks && rank < p.K; ++c2) {
      if (c2 == c) continue;
      const unsigned long long* l = k + c2 * p.K;
      int lo = 0, hi = p.K;  // first index whose key is < mine (keys in [0, lo) beat mine)
      while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (l[mid] > key) lo = mid + 1;
        else hi = mid;
      }
      rank += lo;
    }
    if (rank < p.K) {
      p.out_idx[(size_t)user * p.K + rank] = (int32_t)lr_key_item(key);
      if (p.out_score) p.out_score[(size_t)user * p.K + rank] = lr_key_score(key);
    }
  }
}

// ---- compatibility path: materialised last-position scores ----------------------------------
// MFMA version of the materialised scores: the same 32-MFMA chain per (32 items x 32 users) as item_topk_kernel
// (bit-identical scores), the tile written out through a per-wave LDS transpose so that every user row gets 128
// contiguous bytes. Workgroup = 128 users x a range of item tiles; grid.x walks the tile ranges.
#define IS_TS 36  // floats per user row of the transpose tile (16-byte aligned rows)
__global__ __launch_bounds__(256) void item_scores_mfma_kernel(const float* emb, const float* bias, int n_rows, int n_tiles,
                                                               const float* q, int B, int tiles_per_wg, float* out) {
  __shared__ __attribute__((aligned(16))) float etile[2][32 * TK_ESTRIDE];
  __shared__ __attribute__((aligned(16))) float btile[2][32];
  __shared__ __attribute__((aligned(16))) float tr[4][32 * IS_TS];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int half = lane >> 5, col = lane & 31;
  const int user = blockIdx.y * TK_USERS + wave * 32 + col;
  const bool user_ok = user < B;
  const int tile_begin = blockIdx.x * tiles_per_wg, tile_end = min(n_tiles, tile_begin + tiles_per_wg);
  if (tile_begin >= tile_end) return;
  float bq[32];
  {
    const float4* qp = reinterpret_cast<const float4*>(q + (size_t)(user_ok ? user : 0) * 64 + 32 * half);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float4 v = user_ok ? qp[j] : make_float4(0.f, 0.f, 0.f, 0.f);
      bq[4 * j + 0] = v.x; bq[4 * j + 1] = v.y; bq[4 * j + 2] = v.z; bq[4 * j + 3] = v.w;
    }
  }
  auto stage = [&](int tile, int bufi) {
    const float4* src = reinterpret_cast<const float4*>(emb + (size_t)tile * 32 * 64);
    const float4 e0 = src[tid], e1 = src[tid + 256];
    float* e_ = etile[bufi];
    *reinterpret_cast<float4*>(e_ + (tid >> 4) * TK_ESTRIDE + (tid & 15) * 4) = e0;
    *reinterpret_cast<float4*>(e_ + ((tid + 256) >> 4) * TK_ESTRIDE + (tid & 15) * 4) = e1;
    if (tid < 32) btile[bufi][tid] = bias[tile * 32 + tid];
  };
  stage(tile_begin, 0);
  __syncthreads();
  for (int tile = tile_begin; tile < tile_end; ++tile) {
    const int cur = (tile - tile_begin) & 1;
    if (tile + 1 < tile_end) stage(tile + 1, cur ^ 1);
    float a[32];
    {
      const float* er = etile[cur] + col * TK_ESTRIDE + 32 * half;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float4 v = *reinterpret_cast<const float4*>(er + 4 * j);
        a[4 * j + 0] = v.x; a[4 * j + 1] = v.y; a[4 * j + 2] = v.z; a[4 * j + 3] = v.w;
      }
    }
    floatx16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
#pragma unroll
    for (int s = 0; s < 32; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s], bq[s], acc, 0, 0, 0);
    // lane = user `col`, register r = item row (r&3) + 8*(r>>2) + 4*half  ->  tr[user][item]
    float* t = tr[wave];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const float4 b4 = *reinterpret_cast<const float4*>(&btile[cur][8 * g + 4 * half]);
      float4 o;
      o.x = acc[4 * g + 0] + b4.x;
      o.y = acc[4 * g + 1] + b4.y;
      o.z = acc[4 * g + 2] + b4.z;
      o.w = acc[4 * g + 3] + b4.w;
      *reinterpret_cast<float4*>(t + col * IS_TS + 8 * g + 4 * half) = o;
    }
    __builtin_amdgcn_wave_barrier();
    {  // lane -> (user = lane>>1, 16 items): 64 contiguous bytes per lane, 128 per user row
      const int ul = lane >> 1, i0 = (lane & 1) * 16;
      const int gu = blockIdx.y * TK_USERS + wave * 32 + ul;
      if (gu < B) {
        float* dst = out + (size_t)gu * n_rows + tile * 32 + i0;
        const float* srcp = t + ul * IS_TS + i0;
        if (tile * 32 + i0 + 16 <= n_rows) {
#pragma unroll
          for (int j = 0; j < 16; ++j) dst[j] = srcp[j];
        } else {
          for (int j = 0; j < 16 && tile * 32 + i0 + j < n_rows; ++j) dst[j] = srcp[j];
        }
      }
    }
    __syncthreads();  // next tile staged; this tile's LDS reads done
  }
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
// minimise rounds x (tiles per chunk + the per-chunk fixed cost). The fixed cost -- the unfiltered first tiles
// while the thresholds settle, plus compaction + sort of 128 users' lists at the end -- measures ~130 tile-times
// (Beauty sweep: 1.41 / 1.97 / 2.27 / 2.00 / 2.28 ms for 1..5 chunks = rounds x (0.36 ms + 2.8 us x tiles)).
// With a seeded threshold (bound pre-pass) a chunk's fixed cost is what is left of that: the prologue and the final
// compaction + sort (~ TK_FIXED_SEEDED tile-times), so more, shorter chunks are worth it where they fill the last round.
#define TK_FIXED_PLAIN 128
#define TK_FIXED_SEEDED 24
static void topk_geometry(int n_tiles, int B, bool seeded, int* n_chunks, int* tiles_per_chunk) {
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
    const long cost = rounds * (tpc + (seeded ? TK_FIXED_SEEDED : TK_FIXED_PLAIN));
    if (best_cost < 0 || cost < best_cost) {
      best_cost = cost;
      best = c;
    }
  }
  static int forced = -1;  // LR_TOPK_CHUNKS=n: tuning knob (0 / unset = the model above)
  if (forced < 0) {
    const char* e = getenv("LR_TOPK_CHUNKS");
    forced = e ? atoi(e) : 0;
  }
  if (forced > 0 && forced <= max_chunks) best = forced;
  *tiles_per_chunk = (n_tiles + best - 1) / best;
  *n_chunks = (n_tiles + *tiles_per_chunk - 1) / *tiles_per_chunk;
}

static size_t partial_bytes_max(int B, int K) {
  // user tiles x chunks <= TK_MAX_WGS (+ one chunk minimum) => B x chunks <= max(B, 128 * TK_MAX_WGS)
  size_t rows = (size_t)B;
  if (rows < (size_t)TK_USERS * TK_MAX_WGS) rows = (size_t)TK_USERS * TK_MAX_WGS;
  return lr_align_up(rows * K * sizeof(unsigned long long), 256);
}

// the bound pre-pass runs for catalogs of TK_BOUND_MIN_TILES .. TK_BOUND_MAX_TILES tiles (LR_TOPK_BOUND=0 switches it off)
#define TK_BOUND_MIN_TILES 64
// ... and only where EVERY user is sure to get a bound: rank R = K + masked ids <= K + L + 1 must not exceed the tile count
// (ML-100k: L = 200 against 115 tiles -- its users keep the exact full pass)
// tiles per maximum = 2^shift: the smallest shift (0, or >= 2 so that a group is whole float4 iterations of the kernel)
// that leaves at most TK_BOUND_MAX_TILES maxima per user; -1: the catalog is beyond 2^TK_BOUND_MAX_GSHIFT x that
static int bound_group_shift(int n_tiles) {
  int gs = 0;
  while (((n_tiles + (1 << gs) - 1) >> gs) > TK_BOUND_MAX_TILES) ++gs;
  if (gs == 1) gs = 2;
  return gs <= TK_BOUND_MAX_GSHIFT ? gs : -1;
}
static int bound_groups(int n_tiles) {
  const int gs = bound_group_shift(n_tiles);
  return gs < 0 ? 0 : (n_tiles + (1 << gs) - 1) >> gs;
}
static bool bound_enabled(int n_tiles, int K, int L) {
  static int env = -1;
  if (env < 0) {
    const char* e = getenv("LR_TOPK_BOUND");
    env = (e && e[0] == '0') ? 0 : 1;
  }
  return env && n_tiles >= TK_BOUND_MIN_TILES && bound_group_shift(n_tiles) >= 0 && K + L + 1 <= bound_groups(n_tiles);
}
// tmax [B][ld] | thresh [B] | cand_thresh [B] | cand_count [B] + overflow flag | cand [B][TK_CAND_CAP]
static size_t bound_bytes(int B, int n_tiles) {
  const size_t ld = lr_align_up((size_t)bound_groups(n_tiles), 4);
  return lr_align_up((size_t)B * ld * sizeof(float), 256) + 2 * lr_align_up((size_t)B * sizeof(float), 256) +
         lr_align_up(((size_t)B + 1) * sizeof(int), 256) + lr_align_up((size_t)B * TK_CAND_CAP * sizeof(int32_t), 256);
}

// Sized for EVERY call of up to (B users, K, L): whether the bound pre-pass runs depends on K + L + 1 <= n_tiles, which a
// later call with a smaller K or L can satisfy when the sizing call did not -- so its scratch is included whenever the
// catalog's tile count is in the pre-pass's range, whatever K and L (the size is monotone in B, K and L).
size_t lr_topk_workspace_bytes(int B, int K, int L, int n_tiles) {
  const bool in_range = n_tiles >= TK_BOUND_MIN_TILES && bound_group_shift(n_tiles) >= 0;
  return partial_bytes_max(B, K) + lr_align_up((size_t)B * (L > 0 ? L : 1) * sizeof(int32_t), 256) +
         (in_range ? bound_bytes(B, n_tiles) : 0);
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
  const bool seeded = bound_enabled(p.n_tiles, K, L);
  topk_geometry(p.n_tiles, B, seeded, &p.n_chunks, &p.tiles_per_chunk);
  const size_t need_partial = lr_align_up((size_t)B * p.n_chunks * K * sizeof(unsigned long long), 256);
  const size_t need_hist = p.exclude ? lr_align_up((size_t)B * L * sizeof(int32_t), 256) : 0;
  const size_t need_bound = seeded ? bound_bytes(B, p.n_tiles) : 0;
  if (need_partial + need_hist + need_bound > ws_bytes)
    LR_FAIL(LR_EWORKSPACE, "top-K workspace: need %zu bytes, have %zu", need_partial + need_hist + need_bound, ws_bytes);
  p.partial = reinterpret_cast<unsigned long long*>(ws);
  int32_t* hist_sorted = reinterpret_cast<int32_t*>(reinterpret_cast<char*>(ws) + need_partial);
  p.hist_sorted = p.exclude ? hist_sorted : nullptr;
  p.thresh = nullptr;
  p.run_flag = nullptr;
  int* overflow_flag = nullptr;

  LrProfScope prof(LR_PROF_ITEM_TOPK, 2.0 * 64 * (double)p.n_rows * B, st);
  if (p.exclude) {
    int Lp = 2;
    while (Lp < L) Lp <<= 1;
    if (Lp > 8192) LR_FAIL(LR_EUNSUPPORTED, "history length %d > 8192 is not supported by the mask pre-pass", L);
    if (L <= 64) {
      hipLaunchKernelGGL(hist_sort_wave_kernel, dim3((B + 3) / 4), dim3(256), 0, st, ids, B, L, p.n_rows, hist_sorted);
      LR_CHECK_LAUNCH("hist_sort_wave_kernel");
    } else {
      hipLaunchKernelGGL(hist_sort_kernel, dim3(B), dim3(256), (size_t)Lp * sizeof(int32_t), st, ids, L, Lp, p.n_rows,
                         hist_sorted);
      LR_CHECK_LAUNCH("hist_sort_kernel");
    }
  }
  if (seeded) {
    BoundParams bp;
    bp.emb16 = reinterpret_cast<const unsigned short*>(h->img + h->lay.item_emb_bf16);
    bp.bias = p.bias;
    bp.bias_tail = h->img + h->lay.item_stats + 32;
    bp.n_rows = p.n_rows;
    bp.n_tiles = p.n_tiles;
    bp.q = q;
    bp.B = B;
    bp.gshift = bound_group_shift(p.n_tiles);
    const int n_groups = bound_groups(p.n_tiles);
    bp.ld = (int)lr_align_up((size_t)n_groups, 4);
    char* bw = reinterpret_cast<char*>(ws) + need_partial + need_hist;
    bp.tmax = reinterpret_cast<float*>(bw);
    bw += lr_align_up((size_t)B * bp.ld * sizeof(float), 256);
    float* thresh = reinterpret_cast<float*>(bw);
    bw += lr_align_up((size_t)B * sizeof(float), 256);
    float* cand_thresh = reinterpret_cast<float*>(bw);
    bw += lr_align_up((size_t)B * sizeof(float), 256);
    int* cand_count = reinterpret_cast<int*>(bw);   // [B], then the overflow flag
    overflow_flag = cand_count + B;
    bw += lr_align_up(((size_t)B + 1) * sizeof(int), 256);
    int32_t* cand = reinterpret_cast<int32_t*>(bw);
    const int upw = lr_bf16_users_per_wg(B);
    const int n_ut = (B + upw - 1) / upw;
    int chunks = (1024 + n_ut - 1) / n_ut;                 // two rounds of the 512 resident workgroups (two per CU)
    // ... but never more than lr_bf16_max_chunk_tiles(B) tiles per chunk, however many user tiles there are: the candidate
    // pass's per-chunk LDS lists hold 20-40 entries per user with 16-bit offsets (16 k users used to get
    // 32 chunks of ~1 000 tiles at 1 M items, ~12 expected candidates per user and chunk against 24 slots)
    const int min_chunks = (p.n_tiles + lr_bf16_max_chunk_tiles(B) - 1) / lr_bf16_max_chunk_tiles(B);
    {
      static int target_wgs = -1;   // LR_BF16_WGS=n: tuning knob (workgroups per launch; 0 / unset = 1024)
      if (target_wgs < 0) {
        const char* e = getenv("LR_BF16_WGS");
        target_wgs = e ? atoi(e) : 0;
      }
      if (target_wgs > 0) chunks = (target_wgs + n_ut - 1) / n_ut;
    }
    if (chunks < min_chunks) chunks = min_chunks;
    const int unit = bp.gshift >= 2 ? (4 << bp.gshift) : 4;  // a chunk is whole float4 iterations / whole quads of tile groups
    const int units = (p.n_tiles + unit - 1) / unit;
    if (chunks > units) chunks = units;
    bp.tiles_per_chunk = unit * ((units + chunks - 1) / chunks);
    chunks = (p.n_tiles + bp.tiles_per_chunk - 1) / bp.tiles_per_chunk;
    if (int rc = lr_launch_item_bound(bp, chunks, st)) return rc;
#define TK_SELECT(NS_)                                                                                               \
  hipLaunchKernelGGL(bound_select_kernel<NS_>, dim3((B + 3) / 4), dim3(256), 0, st, bp.tmax, bp.ld, n_groups, q, ids, L, \
                     p.n_rows, p.exclude, B, K, h->img + h->lay.item_stats, thresh, cand_thresh, cand_count, overflow_flag)
    if (n_groups <= 128) TK_SELECT(2);
    else if (n_groups <= 512) TK_SELECT(8);
    else TK_SELECT(TK_BOUND_MAX_TILES / 64);
#undef TK_SELECT
    LR_CHECK_LAUNCH("bound_select_kernel");
    CandParams cp;
    cp.emb16 = bp.emb16;
    cp.bias = p.bias;
    cp.bias_tail = bp.bias_tail;
    cp.n_tiles = p.n_tiles;
    cp.q = q;
    cp.B = B;
    cp.cand_thresh = cand_thresh;
    cp.cand_count = cand_count;
    cp.cand = cand;
    cp.tiles_per_chunk = bp.tiles_per_chunk;
    if (int rc = lr_launch_item_cand(cp, chunks, st)) return rc;
    hipLaunchKernelGGL(cand_rescore_kernel, dim3((B + 3) / 4), dim3(256), 0, st, p.emb, p.bias, q, p.hist_sorted, L,
                       p.exclude, B, K, p.n_rows, cand_count, cand, thresh, overflow_flag, out_idx, out_score);
    LR_CHECK_LAUNCH("cand_rescore_kernel");
    // behind the candidate path: the exact full pass, which runs only if a candidate list overflowed
    p.thresh = thresh;
    p.run_flag = overflow_flag;
  }
  const size_t lds = (2 * 32 * TK_ESTRIDE + 128) * sizeof(float) +
                     (size_t)TK_USERS * TK_BSTRIDE * 8;
  static bool lds_set[LR_MAX_DEVICES] = {};
  if (int rc = lr_ensure_dynamic_lds(reinterpret_cast<const void*>(item_topk_kernel<false>), (int)lds, lds_set)) return rc;
  dim3 grid(p.n_chunks, (B + TK_USERS - 1) / TK_USERS);
#ifdef LR_EXPERIMENTS
  static bool lds_set_stamp[LR_MAX_DEVICES] = {};
  const char* stamp_env = getenv("LR_TOPK_STAMPS");
  if (stamp_env && stamp_env[0] == '1') {
    if (int rc = lr_ensure_dynamic_lds(reinterpret_cast<const void*>(item_topk_kernel<true>), (int)lds, lds_set_stamp)) return rc;
    hipLaunchKernelGGL(item_topk_kernel<true>, grid, dim3(256), lds, st, p);
  } else
#endif
    hipLaunchKernelGGL(item_topk_kernel<false>, grid, dim3(256), lds, st, p);
  LR_CHECK_LAUNCH("item_topk_kernel");

  MergeParams m;
  m.partial = p.partial;
  m.B = B;
  m.K = K;
  m.n_chunks = p.n_chunks;
  m.out_idx = out_idx;
  m.out_score = out_score;
  m.run_flag = p.run_flag;
  if (p.n_chunks > 1 && p.n_chunks * K <= TK_MERGE_KEYS)
    hipLaunchKernelGGL(topk_merge_small_kernel, dim3((B + 3) / 4), dim3(256), 0, st, m);
  else
    hipLaunchKernelGGL(topk_merge_kernel, dim3((B + 3) / 4), dim3(256), 0, st, m);
  LR_CHECK_LAUNCH("topk_merge_kernel");
  return LR_OK;
}

// Which path the LAST lr_launch_item_topk call with exactly these (B, K, L) on this workspace took: 0 = the exact full
// pass (the bound does not serve this shape), 1 = bound -> candidates -> exact rescoring, 2 = that path overflowed a
// candidate list and the exact full pass redid the call. Reads the call's overflow flag where the call carved it.
int lr_topk_path(const lr_lru* h, int B, int K, int L, int exclude_history, const void* ws, size_t ws_bytes, int* out_path,
                 hipStream_t st) {
  const int n_tiles = h->lay.rows_padded / LR_ITEM_TILE;
  if (!bound_enabled(n_tiles, K, L) || B <= 0) {
    *out_path = 0;
    return LR_OK;
  }
  int n_chunks, tpc;
  topk_geometry(n_tiles, B, true, &n_chunks, &tpc);
  const size_t need_partial = lr_align_up((size_t)B * n_chunks * K * sizeof(unsigned long long), 256);
  const size_t need_hist = exclude_history ? lr_align_up((size_t)B * L * sizeof(int32_t), 256) : 0;
  if (need_partial + need_hist + bound_bytes(B, n_tiles) > ws_bytes) LR_FAIL(LR_EWORKSPACE, "lr_topk_path: workspace too small");
  const size_t ld = lr_align_up((size_t)bound_groups(n_tiles), 4);
  const char* bw = reinterpret_cast<const char*>(ws) + need_partial + need_hist + lr_align_up((size_t)B * ld * sizeof(float), 256) +
                   2 * lr_align_up((size_t)B * sizeof(float), 256);
  int flag = 0;
  LR_CHECK_HIP(hipMemcpyAsync(&flag, reinterpret_cast<const int*>(bw) + B, sizeof(int), hipMemcpyDeviceToHost, st));
  LR_CHECK_HIP(hipStreamSynchronize(st));
  *out_path = flag ? 2 : 1;
  return LR_OK;
}

int lr_launch_item_scores(const lr_lru* h, const float* q, const int64_t* ids, int B, int L,
                          int exclude_history, float* out_scores, hipStream_t st) {
  if (B <= 0) return LR_OK;
  int n_rows = h->lay.num_items + 1;
  {
    const int n_tiles = h->lay.rows_padded / LR_ITEM_TILE, n_ut = (B + TK_USERS - 1) / TK_USERS;
    int chunks = (1024 + n_ut - 1) / n_ut;  // ~4 workgroups per CU in flight
    if (chunks > n_tiles) chunks = n_tiles;
    const int tpw = (n_tiles + chunks - 1) / chunks;
    dim3 grid((n_tiles + tpw - 1) / tpw, n_ut);
    hipLaunchKernelGGL(item_scores_mfma_kernel, grid, dim3(256), 0, st, h->img + h->lay.item_emb, h->img + h->lay.item_bias,
                       n_rows, n_tiles, q, B, tpw, out_scores);
    LR_CHECK_LAUNCH("item_scores_mfma_kernel");
  }
  if (exclude_history) {
    hipLaunchKernelGGL(mask_history_kernel, dim3(B), dim3(256), 0, st, out_scores, n_rows, ids, B, L);
    LR_CHECK_LAUNCH("mask_history_kernel");
  }
  return LR_OK;
}
