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

#define EM_XS 68  // LDS row stride (floats) of a 64-float row: conflict-free ds_read_b128

struct EmChunk {
  const int64_t* ids;  // [users][L] of this chunk
  int users, L, num_items;
  int* n;         // [users] live tokens
  int* off;       // [users + 1] first row of each user; off[users] = live rows of the chunk
  int* last_row;  // [users]
  int* row_tag;   // [rows] (user << 2) | (last row of the user) << 1 | (first row of the user)
  int* row_item;  // [rows] item id of the row (out-of-range ids -> 0, the padding row)
  int* wg_row;    // [G + 1] row ranges of the G layer workgroups: about rows / G each, starting at a user's first row
  int G;
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

// ---- exclusive prefix sum of n over the chunk's users + the layer workgroups' row ranges (one workgroup) --------
// Wave w owns a contiguous block of users: pass 1 sums its block (coalesced, no barrier inside), one barrier publishes
// the 16 block totals, pass 2 walks the block again with a shuffle scan per 64 users and writes the offsets.
__global__ __launch_bounds__(1024) void em_offsets_kernel(EmChunk c) {
  __shared__ int wsum[16];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int per_wave = ((c.users + 16 * 64 - 1) / (16 * 64)) * 64;  // users per wave, a multiple of 64
  const int ua = min(c.users, wave * per_wave), ub = min(c.users, ua + per_wave);
  int s = 0;
  for (int u = ua + lane; u < ub; u += 64) s += c.n[u];
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) s += __shfl_xor(s, d, 64);
  if (lane == 0) wsum[wave] = s;
  __syncthreads();
  int run = 0, carry = 0;
  for (int w = 0; w < 16; ++w) {
    if (w < wave) run += wsum[w];
    carry += wsum[w];
  }
  for (int u0 = ua; u0 < ub; u0 += 64) {
    const int u = u0 + lane;
    const int n = u < ub ? c.n[u] : 0;
    int incl = n;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const int v = __shfl_up(incl, d, 64);
      if (lane >= d) incl += v;
    }
    if (u < ub) {
      c.off[u] = run + incl - n;
      c.last_row[u] = run + incl - 1;
    }
    run += __shfl(incl, 63, 64);
  }
  const int total = carry;
  if (tid == 0) c.off[c.users] = total;
  __syncthreads();  // off[] is complete and visible to this workgroup
  // wg_row[g] = first row of the first user that starts at or after row g * ceil(total / G)  (wg_row[G] = total)
  if (tid <= c.G) {
    const int target = (int)min((long long)total, (long long)tid * ((total + c.G - 1) / c.G));
    int lo = 0, hi = c.users;  // smallest u with off[u] >= target (off[users] = total >= target)
    while (lo < hi) {
      const int mid = (lo + hi) >> 1;
      if (__builtin_nontemporal_load(c.off + mid) >= target) hi = mid; else lo = mid + 1;
    }
    c.wg_row[tid] = lo == c.users ? total : __builtin_nontemporal_load(c.off + lo);
  }
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

// ---- row -> (item id, tag): thread = (user, position) --------------------------------------------------------
__global__ __launch_bounds__(256) void em_rows_kernel(EmChunk c) {
  const long long g = (long long)blockIdx.x * 256 + threadIdx.x;
  if (g >= (long long)c.users * c.L) return;
  const int u = (int)(g / c.L), pos = (int)(g - (long long)u * c.L);
  const int n = c.n[u], t = pos - (c.L - n);
  if (t < 0) return;
  long long id = c.ids[g];
  if (id < 0 || id > c.num_items) id = 0;
  const int row = c.off[u] + t;
  c.row_item[row] = (int)id;
  c.row_tag[row] = (u << 2) | ((t == n - 1) << 1) | (t == 0);
}

// ---- embedding gather + LayerNorm: 16 lanes per row, 4 features per lane ----------------------------------------
// The sums are the 64-lane butterfly of lru_encoder.hip (levels xor 32, 16, 8, 4, 2, 1 over the feature index) with the
// feature index split as 4 * lane + component: levels 32..4 exchange lanes 8, 4, 2, 1 apart, levels 2 and 1 add the
// lane's own components (0,2), (1,3), then the two sums -- the same pairs at every level.
__device__ __forceinline__ float em_sum64_q(const float4 v) {
  float a = v.x, b = v.y, c = v.z, d = v.w;
#pragma unroll
  for (int s = 8; s >= 1; s >>= 1) {
    a = a + __shfl_xor(a, s, 64);
    b = b + __shfl_xor(b, s, 64);
    c = c + __shfl_xor(c, s, 64);
    d = d + __shfl_xor(d, s, 64);
  }
  const float e0 = a + c, e1 = b + d;
  return e0 + e1;
}
__global__ __launch_bounds__(256) void em_embed_kernel(EmChunk c, const float* img, LrLruLayout lay, float* X) {
  const int n_rows = c.off[c.users];
  const int l = threadIdx.x & 15;
  const float4 w = *reinterpret_cast<const float4*>(img + lay.emb_ln_w + 4 * l);
  const float4 b = *reinterpret_cast<const float4*>(img + lay.emb_ln_b + 4 * l);
  const int rows_per_pass = gridDim.x * 16;
  for (int base = blockIdx.x * 16; base < n_rows; base += rows_per_pass) {  // workgroup-uniform trip count
    const int row = base + (threadIdx.x >> 4);
    const bool live = row < n_rows;
    const int id = c.row_item[live ? row : 0];
    const float4 e = *reinterpret_cast<const float4*>(img + lay.item_emb + (size_t)id * 64 + 4 * l);
    const float mean = em_sum64_q(e) * 0.015625f;
    const float4 d = make_float4(e.x - mean, e.y - mean, e.z - mean, e.w - mean);
    const float var = em_sum64_q(make_float4(d.x * d.x, d.y * d.y, d.z * d.z, d.w * d.w)) * 0.015625f;
    const float rstd = 1.0f / sqrtf(var + LR_LN_EPS);
    if (live)
      *reinterpret_cast<float4*>(X + (size_t)row * 64 + 4 * l) =
          make_float4(lr_fma(d.x * rstd, w.x, b.x), lr_fma(d.y * rstd, w.y, b.y), lr_fma(d.z * rstd, w.z, b.z),
                      lr_fma(d.w * rstd, w.w, b.w));
  }
}

// GELU of lr_math.h (lr_gelu_tab / lr_erff_tab: same operations in the same order) reading the erf table from an LDS
// copy with 12-float rows: the nine coefficients of a lane's interval arrive in three 16-byte reads instead of nine
// 4-byte ones (rows of different intervals fall on disjoint banks).
#define EM_ERF_ROW 12
// N values at a time, stage by stage (arguments, coefficient reads, the N Horner chains in lockstep, finish): the table
// reads of one value and its eight dependent fmas would otherwise run back to back with nothing beside them.
template <int N>
__device__ __forceinline__ void em_gelu_tab12(const float (&x)[N], float (&y)[N], const float* tab12) {
  static_assert(LR_ERF_DEG == 8, "em_gelu_tab12 unrolls a degree-8 Horner chain");
  float xe[N], a[N], t[N], r[N];
  bool in[N];
  float4 c0[N], c1[N];
#pragma unroll
  for (int n = 0; n < N; ++n) {
    xe[n] = x[n] * 0.70710678118654752440f;
    a[n] = __builtin_fabsf(xe[n]);
    in[n] = a[n] < 4.0f;                   // false for NaN too
    const float ai = in[n] ? a[n] : 0.0f;  // the polynomial is evaluated unconditionally, on a harmless argument where
    const int i = (int)(ai * 2.0f);        // its result is not used
    t[n] = ai - ((float)i * 0.5f + 0.25f);
    const float* c = tab12 + i * EM_ERF_ROW;
    c0[n] = *reinterpret_cast<const float4*>(c);
    c1[n] = *reinterpret_cast<const float4*>(c + 4);
    r[n] = c[8];
  }
#pragma unroll
  for (int n = 0; n < N; ++n) r[n] = lr_fma(r[n], t[n], c1[n].w);
#pragma unroll
  for (int n = 0; n < N; ++n) r[n] = lr_fma(r[n], t[n], c1[n].z);
#pragma unroll
  for (int n = 0; n < N; ++n) r[n] = lr_fma(r[n], t[n], c1[n].y);
#pragma unroll
  for (int n = 0; n < N; ++n) r[n] = lr_fma(r[n], t[n], c1[n].x);
#pragma unroll
  for (int n = 0; n < N; ++n) r[n] = lr_fma(r[n], t[n], c0[n].w);
#pragma unroll
  for (int n = 0; n < N; ++n) r[n] = lr_fma(r[n], t[n], c0[n].z);
#pragma unroll
  for (int n = 0; n < N; ++n) r[n] = lr_fma(r[n], t[n], c0[n].y);
#pragma unroll
  for (int n = 0; n < N; ++n) r[n] = lr_fma(r[n], t[n], c0[n].x);
#pragma unroll
  for (int n = 0; n < N; ++n) {
    const float rr = in[n] ? r[n] : (a[n] != a[n] ? a[n] : 1.0f);  // erf(4) rounds to 1 in binary32; a NaN stays a NaN
    const float e = __builtin_copysignf(rr, xe[n]);
    y[n] = (0.5f * x[n]) * (1.0f + e);
  }
}

// ---- one fused position-wise layer pair on 64-row super tiles ------------------------------------------------
//   MODE 0  LRU layer      : in_proj (* gamma) -> recurrence -> out_proj + residual + LayerNorm        X -> Y
//   MODE 1  feed-forward   : w_1 + GELU -> w_2 + residual + LayerNorm                                  Y -> X
//   MODE 2  LRU, last block: in_proj -> recurrence, the state of each user's LAST row only             X -> Ul[user][256]
//   MODE 3  out_proj + residual + LayerNorm of rows given as [row][256] (the last rows)                Ul -> Yl
// The 256-wide intermediate of a super tile (in_proj output / FFN hidden) lives only in LDS: the layer reads 256 B and
// writes 256 B per row instead of 1 KiB + 1 KiB (+ 2 KiB for the recurrence) per row through HBM.
//   phase A (K = 64, weights in registers): wave w owns output blocks 2w, 2w+1 for both 32-row tiles of the super
//     tile; D[token][out] goes to LDS at ul[token][pos(out)], pos() = the de-interleaved order phase B reads.
//   recurrence: thread = complex channel, walks the 64 rows in LDS; its carry stays in registers from one super tile
//     to the next -- a workgroup's row range starts at a user's first row (wg_row, em_offsets_kernel), row_tag says
//     where users start / end inside it.
//   phase B (K = 256, weights in LDS): wave (rt, j) = (row tile, 32-feature block), one 128-step chain; block 1 hands
//     its 16 values per lane to the block-0 wave through LDS, which finishes the LayerNorm butterfly (first level =
//     feature f + feature f+32, exactly that hand-over) and stores the row.
// Per row the arithmetic is that of lru_encoder.hip: which rows share a tile never changes a row's chain.
#define EM_ST 64   // rows per super tile
#define EM_THREADS 512  // 8 waves, two per SIMD: the epilogues, the recurrence and the staging are VALU / issue bound, and
                        // one wave alone issues a vector instruction every 4 cycles, two waves one every 2
#define EM_US 260  // LDS row stride (floats) of a 256-float row: 16-byte aligned, conflict-free ds_read_b128
#define EM_WS 260  // same for the K = 256 weight rows

// Workgroup barrier that orders LDS traffic only: __syncthreads() also drains vmcnt, which would expose the latency of
// the next super tile's input rows (requested a whole tile ahead) at every barrier.
#define EM_BARRIER()                                   \
  do {                                                 \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); \
    __builtin_amdgcn_s_barrier();                      \
    asm volatile("" ::: "memory");                     \
  } while (0)

// Diagnostic stamps (-DLR_EXPERIMENTS build only, LR_EM_STAMPS=1): wave 0 of the first 16 workgroups sums s_memtime
// ticks per phase over its super tiles; tools/em_stamps.py prints them. The product build contains none of this.
#ifdef LR_EXPERIMENTS
__device__ unsigned long long g_em_stamps[4][16][12];
#define EM_STAMP(k)                                                      \
  do {                                                                   \
    if (p.stamp) {                                                       \
      const unsigned long long now_ = __builtin_amdgcn_s_memtime();     \
      st_acc[k] += now_ - st_prev;                                       \
      st_prev = now_;                                                    \
    }                                                                    \
  } while (0)
#else
#define EM_STAMP(k) do { } while (0)
#endif

struct EmLayer {
  const float *wa, *ba, *gamma, *lam_re, *lam_im;  // phase A: k-major [64][256], bias[256], gamma[128], lambda[128]
  const float *wb, *bb, *lnw, *lnb;                // phase B: k-major [256][64], bias[64], LayerNorm weight / bias
  const float* IN;        // phase A input rows [.][64] (MODE 3: phase B input rows [.][256])
  const float* RES;       // residual rows [.][64]
  const int* res_rows;    // optional gather list for RES
  float* OUT;             // [.][64]; MODE 2: [users][256]
  const int* row_tag;     // (user << 2) | last << 1 | first, per row (MODE 0, 2)
  const int* wg_row;      // [gridDim.x + 1] user-aligned row ranges; null: super tiles strided over n_rows
  const int* n_rows_ptr;
  int n_rows_fixed;
  int stamp;
};

__device__ __forceinline__ int em_pos(int out) { return (out >> 6) * 64 + (out & 1) * 32 + ((out & 63) >> 1); }

// floats [part*8, part*8 + 8) of row r0 + row of a [.][64] matrix; zeros past rb
__device__ __forceinline__ void em_load_rows8(float4 (&v)[2], const float* M, int r0, int rb, int row, int part) {
  if (r0 + row < rb) {
    const float* src = M + (size_t)(r0 + row) * 64 + part * 8;
    v[0] = *reinterpret_cast<const float4*>(src);
    v[1] = *reinterpret_cast<const float4*>(src + 4);
  } else {
    v[0] = v[1] = make_float4(0.f, 0.f, 0.f, 0.f);
  }
}

template <int MODE>
__global__ __launch_bounds__(EM_THREADS) __attribute__((amdgpu_waves_per_eu(2, 2))) void em_layer_kernel(EmLayer p) {
  constexpr bool HAS_A = MODE != 3, HAS_B = MODE != 2, SCAN = MODE == 0 || MODE == 2;
  extern __shared__ __attribute__((aligned(16))) float smem_f[];
  float* ws = smem_f;                    // [64][EM_WS]   phase B weights, de-interleaved per 64-k chunk
  float* ul = ws + 64 * EM_WS;           // [EM_ST][EM_US] the 256-wide intermediate
  float* xs = ul + EM_ST * EM_US;        // [EM_ST][EM_XS] phase A input, de-interleaved
  float* ex = xs + EM_ST * EM_XS;        // [2][16][64]    feature block 1 -> block 0 hand-over
  float* lnp = ex + 2 * 16 * 64;         // [2][64]        LayerNorm weight, bias
  int* tags = reinterpret_cast<int*>(lnp + 128);      // [EM_ST]
  float* erf_lds = reinterpret_cast<float*>(tags + EM_ST);  // [LR_ERF_NINT][EM_ERF_ROW]
  float* carry = erf_lds + LR_ERF_NINT * EM_ERF_ROW;       // [2][64 lanes][4] recurrence state after row 63, by super tile parity
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int half = lane >> 5, col = lane & 31;

#ifdef LR_EXPERIMENTS
  unsigned long long st_acc[12] = {}, st_prev = __builtin_amdgcn_s_memtime();
#endif
  int ra, rb, step;
  if (p.wg_row) {
    ra = p.wg_row[blockIdx.x];
    rb = p.wg_row[blockIdx.x + 1];
    step = EM_ST;
  } else {
    ra = blockIdx.x * EM_ST;
    rb = p.n_rows_ptr ? *p.n_rows_ptr : p.n_rows_fixed;
    step = gridDim.x * EM_ST;
  }
  if (ra >= rb) return;

  if (HAS_B) {
    for (int i = tid; i < 64 * 256; i += EM_THREADS) {
      const int k = i >> 6, out = i & 63;  // coalesced read of W[k][out]
      ws[out * EM_WS + (k >> 6) * 64 + (k & 1) * 32 + ((k & 63) >> 1)] = p.wb[i];
    }
    if (tid < 64) lnp[tid] = p.lnw[tid];
    else if (tid < 128) lnp[tid] = p.lnb[tid - 64];
  }
  if (MODE == 1)
    for (int i = tid; i < LR_ERF_NINT * EM_ERF_ROW; i += EM_THREADS) {
      const int c = i % EM_ERF_ROW;
      erf_lds[i] = c <= LR_ERF_DEG ? em_erf_tab[(i / EM_ERF_ROW) * (LR_ERF_DEG + 1) + c] : 0.f;
    }
  // phase A operand B: weights of my output block (wave w: outputs 32w .. 32w+31), de-interleaved k: step s of half h
  // uses k = 2s + h
  float wq[32], bq = 0.f, gq = 0.f;
  if (HAS_A) {
    const int out = wave * 32 + col;
#pragma unroll
    for (int s = 0; s < 32; ++s) wq[s] = p.wa[(size_t)(2 * s + half) * 256 + out];
    bq = p.ba[out];
    gq = MODE == 1 ? 0.f : p.gamma[out & 127];
  }
  // phase B (waves 0..3, one per SIMD: the chains are MFMA bound): my chain = feature block jb of row tile rt
  const int rt = (wave >> 1) & 1, jb = wave & 1;
  float bb[16];
  if (HAS_B && wave < 4) {
#pragma unroll
    for (int r = 0; r < 16; ++r) bb[r] = p.bb[jb * 32 + (r & 3) + 8 * (r >> 2) + 4 * half];
  }
  // recurrence: lane = two complex channels k, k + 2 (neighbours in the de-interleaved row), wave = a segment of rows
  const int sc_k = (lane >> 5) * 64 + 4 * (lane & 15) + ((lane >> 4) & 1);
  const int sc_pos = em_pos(sc_k);  // em_pos(sc_k + 2) = sc_pos + 1, em_pos(128 + k) = em_pos(k) + 128
  float lam_r0 = 0.f, lam_i0 = 0.f, lam_r1 = 0.f, lam_i1 = 0.f;
  if (SCAN) {
    lam_r0 = p.lam_re[sc_k];
    lam_i0 = p.lam_im[sc_k];
    lam_r1 = p.lam_re[sc_k + 2];
    lam_i1 = p.lam_im[sc_k + 2];
    *reinterpret_cast<float4*>(carry + 4 * lane) = make_float4(0.f, 0.f, 0.f, 0.f);  // every wave: same zeros
  }

  float4 pre[2];  // my 8 floats of the NEXT super tile's phase A input: their latency hides behind this tile's MFMAs
  if (HAS_A) em_load_rows8(pre, p.IN, ra, rb, tid >> 3, tid & 7);
  int par = 0;
  int pre_tag = 1;  // rows past the range restart the recurrence and are never a user's last row
  if (SCAN && tid < EM_ST && ra + tid < rb) pre_tag = p.row_tag[ra + tid];

  EM_STAMP(0);
  for (int r0 = ra; r0 < rb; r0 += step) {
    const int nv = min(EM_ST, rb - r0);
    EM_BARRIER();  // the previous super tile is consumed
    EM_STAMP(1);
    {
      const int row = tid >> 3, part = tid & 7;
      if (HAS_A) {  // 8 floats of a 64-float row (loaded one super tile ahead), de-interleaved: xs[row][h*32 + s] = x[row][2s + h]
        float* dst = xs + row * EM_XS + part * 4;  // even k -> [part*4, +4), odd k -> 32 + the same: two 16-byte stores
        *reinterpret_cast<float4*>(dst) = make_float4(pre[0].x, pre[0].z, pre[1].x, pre[1].z);
        *reinterpret_cast<float4*>(dst + 32) = make_float4(pre[0].y, pre[0].w, pre[1].y, pre[1].w);
        em_load_rows8(pre, p.IN, r0 + step, rb, row, part);
      } else {  // MODE 3: half of a 64-k chunk of a 256-float row, de-interleaved inside the chunk
        float* dst = ul + row * EM_US + (part >> 1) * 64 + (part & 1) * 16;
        const float* src = p.IN + (size_t)(r0 + (row < nv ? row : 0)) * 256 + part * 32;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          float4 v = *reinterpret_cast<const float4*>(src + 4 * i);
          if (row >= nv) v = make_float4(0.f, 0.f, 0.f, 0.f);
          dst[2 * i] = v.x;
          dst[32 + 2 * i] = v.y;
          dst[2 * i + 1] = v.z;
          dst[32 + 2 * i + 1] = v.w;
        }
      }
      if (SCAN && tid < EM_ST) {
        tags[tid] = pre_tag;
        pre_tag = r0 + step + tid < rb ? p.row_tag[r0 + step + tid] : 1;
      }
    }
    EM_STAMP(2);
    EM_BARRIER();
    EM_STAMP(3);

    if (HAS_A) {
      // The two waves of a SIMD (w and w + 4) would run chain -> epilogue -> chain -> epilogue in lockstep: both chains
      // share the matrix pipe at half rate, then both epilogues (GELU: ~25 vector instructions per value) share the
      // vector ALU while the pipe idles. With the second-dispatched half at priority 1 its chain takes the pipe first and
      // the partner's follows while the first wave is in its epilogue: chain | chain || epilogue | ... instead of
      // (chain + chain) | (epilogue + epilogue).
      if (wave >= 4) __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        float a[32];  // token row t*32 + col, 32 de-interleaved values of my half
        const float* xr = xs + (t * 32 + col) * EM_XS + 32 * half;
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          const float4 v4 = *reinterpret_cast<const float4*>(xr + 4 * q);
          a[4 * q + 0] = v4.x; a[4 * q + 1] = v4.y; a[4 * q + 2] = v4.z; a[4 * q + 3] = v4.w;
        }
        {
          floatx16 acc;
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[r] = bq;
#pragma unroll
          for (int s = 0; s < 32; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s], wq[s], acc, 0, 0, 0);
          // D[token i][out]: lane = out column, register r = token (r&3) + 8*(r>>2) + 4*half
          float* dst = ul + (wave >> 1) * 64 + (col & 1) * 32 + (wave & 1) * 16 + (col >> 1);  // em_pos(wave*32 + col)
#pragma unroll
          for (int r8 = 0; r8 < 16; r8 += 8) {
            float xin[8], val[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) xin[i] = acc[r8 + i];
            if (MODE == 1) {
              em_gelu_tab12<8>(xin, val, erf_lds);
            } else {
#pragma unroll
              for (int i = 0; i < 8; ++i) val[i] = xin[i] * gq;
            }
#pragma unroll
            for (int i = 0; i < 8; ++i) {
              const int r = r8 + i;
              const int tok = t * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
              dst[tok * EM_US] = val[i];
            }
          }
        }
      }
      if (wave >= 4) __builtin_amdgcn_s_setprio(0);
      EM_STAMP(4);
      EM_BARRIER();
      EM_STAMP(5);
    }

    if (SCAN) {
      // The 64 rows are cut at users' first rows into up to eight segments (cuts at the first start at or after row 8,
      // 16, .. 56); wave w walks segment w. Only segment 0 can continue a user of the previous super tile: its carry comes
      // from the wave that walked row 63 there, through LDS.
      const int tg_lane = tags[lane];
      const unsigned long long firsts = __ballot(tg_lane & 1);
      const unsigned long long lasts = __ballot(tg_lane & 2);
      int sa = 0, sb = EM_ST;
      if (wave > 0) {
        const unsigned long long m = firsts & (~0ull << (8 * wave));
        sa = m ? __builtin_ctzll(m) : EM_ST;
      }
      if (wave < 7) {
        const unsigned long long m = firsts & (~0ull << (8 * (wave + 1)));
        sb = m ? __builtin_ctzll(m) : EM_ST;
      }
      sa = __builtin_amdgcn_readfirstlane(sa);
      sb = __builtin_amdgcn_readfirstlane(sb);
      if (sa < sb) {
        float h_r0, h_r1, h_i0, h_i1;
        if (wave == 0) {
          const float4 cv = *reinterpret_cast<const float4*>(carry + par * 256 + 4 * lane);
          h_r0 = cv.x; h_r1 = cv.y; h_i0 = cv.z; h_i1 = cv.w;
        } else {
          h_r0 = h_r1 = h_i0 = h_i1 = 0.f;  // overwritten by the segment's first row
        }
        float* base = ul + sc_pos;
        // one recurrence step on the lane's two complex channels (the oracle's fmaf order), then the row's hand-over:
        // MODE 0 writes the state back over the row, MODE 2 emits it where a user ends
#define EM_SCAN_STEP(t_, bre_, bim_, is_first_)                                                        \
  {                                                                                                    \
    if (is_first_) {                                                                                   \
      h_r0 = (bre_).x; h_r1 = (bre_).y; h_i0 = (bim_).x; h_i1 = (bim_).y;                             \
    } else {                                                                                           \
      const float nr0 = lr_fma(lam_r0, h_r0, lr_fma(-lam_i0, h_i0, (bre_).x));                         \
      const float ni0 = lr_fma(lam_r0, h_i0, lr_fma(lam_i0, h_r0, (bim_).x));                          \
      const float nr1 = lr_fma(lam_r1, h_r1, lr_fma(-lam_i1, h_i1, (bre_).y));                         \
      const float ni1 = lr_fma(lam_r1, h_i1, lr_fma(lam_i1, h_r1, (bim_).y));                          \
      h_r0 = nr0; h_i0 = ni0; h_r1 = nr1; h_i1 = ni1;                                                  \
    }                                                                                                  \
    if (MODE == 0) {                                                                                   \
      *reinterpret_cast<float2*>(base + (t_) * EM_US) = make_float2(h_r0, h_r1);                       \
      *reinterpret_cast<float2*>(base + (t_) * EM_US + 128) = make_float2(h_i0, h_i1);                 \
    } else if ((lasts >> (t_)) & 1) {                                                                  \
      float* o = p.OUT + (size_t)(tags[(t_)] >> 2) * 256;                                              \
      o[sc_k] = h_r0;                                                                                  \
      o[sc_k + 2] = h_r1;                                                                              \
      o[128 + sc_k] = h_i0;                                                                            \
      o[128 + sc_k + 2] = h_i1;                                                                        \
    }                                                                                                  \
  }
        // users' first rows strictly inside the segment (a segment STARTS at a first row or continues the previous tile)
        const unsigned long long upto = sb >= 64 ? ~0ull : ((1ull << sb) - 1ull);
        const unsigned long long inner_firsts = firsts & upto & ~((2ull << sa) - 1ull);
        if (inner_firsts == 0ull) {
          // ONE user from sa to sb (long histories: Synth-1M's 110-row users fill whole tiles, and this wave then walks
          // all 64 rows alone while seven waves and the matrix pipe wait -- 7.8 k of a tile's 31.7 k cycles with the loop
          // below, tools/em_stamps.py). No per-row first test, and the next four rows' operands are requested before the
          // current four are stepped through, so the chain waits for fmas only. Same operations in the same order.
          int t = sa;
          {
            const float2 b_re = *reinterpret_cast<const float2*>(base + t * EM_US);
            const float2 b_im = *reinterpret_cast<const float2*>(base + t * EM_US + 128);
            const bool first = (firsts >> t) & 1;   // wave-uniform
            EM_SCAN_STEP(t, b_re, b_im, first)
            ++t;
          }
          float2 cr[4], ci[4], nr[4], ni[4];
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const int tt = min(t + i, sb - 1);
            cr[i] = *reinterpret_cast<const float2*>(base + tt * EM_US);
            ci[i] = *reinterpret_cast<const float2*>(base + tt * EM_US + 128);
          }
          for (; t + 4 <= sb; t += 4) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {   // rows t + 4 .. t + 7 (clamped: a repeated read of row sb - 1 is never used)
              const int tt = min(t + 4 + i, sb - 1);
              nr[i] = *reinterpret_cast<const float2*>(base + tt * EM_US);
              ni[i] = *reinterpret_cast<const float2*>(base + tt * EM_US + 128);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) EM_SCAN_STEP(t + i, cr[i], ci[i], false)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              cr[i] = nr[i];
              ci[i] = ni[i];
            }
          }
#pragma unroll
          for (int i = 0; i < 3; ++i)
            if (t + i < sb) EM_SCAN_STEP(t + i, cr[i], ci[i], false)   // wave-uniform
        } else {
        for (int t0 = sa; t0 < sb; t0 += 4) {
          float2 br[4], bi[4];
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const int t = min(t0 + i, sb - 1);
            br[i] = *reinterpret_cast<const float2*>(base + t * EM_US);
            bi[i] = *reinterpret_cast<const float2*>(base + t * EM_US + 128);
          }
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const int t = t0 + i;
            if (t < sb) {  // wave-uniform
              const bool first = (firsts >> t) & 1;
              EM_SCAN_STEP(t, br[i], bi[i], first)
            }
          }
        }
        }
#undef EM_SCAN_STEP
        // (the other buffer: wave 0 may not have read this tile's carry-in yet)
        if (sb == EM_ST) *reinterpret_cast<float4*>(carry + (par ^ 1) * 256 + 4 * lane) = make_float4(h_r0, h_r1, h_i0, h_i1);
      }
      par ^= 1;
      EM_STAMP(6);
      if (HAS_B) EM_BARRIER();
      EM_STAMP(7);
    }

    if (HAS_B) {
      // D[feature i][token]: lane = token rt*32 + col, acc[r] = feature jb*32 + (r&3) + 8*(r>>2) + 4*half
      const int lrow = rt * 32 + col;
      const bool live = lrow < nv;
      const int gr = r0 + lrow;
      float v[2][16];
      if (wave < 4) {
      float4 r4[4];  // residual, requested before the chain
      {
        const float* res = p.RES + (size_t)(live ? (p.res_rows ? p.res_rows[gr] : gr) : 0) * 64;
#pragma unroll
        for (int g = 0; g < 4; ++g)
          r4[g] = live ? *reinterpret_cast<const float4*>(res + jb * 32 + 8 * g + 4 * half) : make_float4(0.f, 0.f, 0.f, 0.f);
      }
      floatx16 acc;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = bb[r];
      {
        const float* wr = ws + (jb * 32 + col) * EM_WS + 32 * half;   // feature row jb*32 + col
        const float* tr = ul + (rt * 32 + col) * EM_US + 32 * half;   // token row rt*32 + col
        // operands of the next 8 steps are requested before the 8 MFMAs of the current ones (the partner wave of the
        // SIMD is idle here: nothing else hides the LDS latency)
        float4 wc[2], bc[2], wn[2], bn[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          wc[i] = *reinterpret_cast<const float4*>(wr + 4 * i);
          bc[i] = *reinterpret_cast<const float4*>(tr + 4 * i);
        }
#pragma unroll
        for (int g = 0; g < 16; ++g) {  // 8 k steps each: k chunk g >> 2, float4 pairs 2*(g&3), 2*(g&3) + 1
          if (g < 15) {
            const int o = ((g + 1) >> 2) * 64 + ((g + 1) & 3) * 8;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
              wn[i] = *reinterpret_cast<const float4*>(wr + o + 4 * i);
              bn[i] = *reinterpret_cast<const float4*>(tr + o + 4 * i);
            }
          }
          __builtin_amdgcn_sched_barrier(0);  // keep the reads above the MFMAs (the scheduler sinks them back otherwise)
#pragma unroll
          for (int i = 0; i < 2; ++i) {
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wc[i].x, bc[i].x, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wc[i].y, bc[i].y, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wc[i].z, bc[i].z, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wc[i].w, bc[i].w, acc, 0, 0, 0);
          }
#pragma unroll
          for (int i = 0; i < 2; ++i) {
            wc[i] = wn[i];
            bc[i] = bn[i];
          }
        }
      }
      EM_STAMP(8);
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        v[0][4 * g + 0] = acc[4 * g + 0] + r4[g].x;
        v[0][4 * g + 1] = acc[4 * g + 1] + r4[g].y;
        v[0][4 * g + 2] = acc[4 * g + 2] + r4[g].z;
        v[0][4 * g + 3] = acc[4 * g + 3] + r4[g].w;
      }
      if (jb == 1) {
#pragma unroll
        for (int r = 0; r < 16; ++r) ex[(rt * 16 + r) * 64 + lane] = v[0][r];
      }
      }  // wave < 4
      EM_STAMP(9);
      EM_BARRIER();
      EM_STAMP(10);
      if (jb == 0 && wave < 4) {
#pragma unroll
        for (int r = 0; r < 16; ++r) v[1][r] = ex[(rt * 16 + r) * 64 + lane];
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
          float* y = p.OUT + (size_t)gr * 64;
#pragma unroll
          for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
              const int f = j * 32 + 8 * g + 4 * half;
              const float4 w4 = *reinterpret_cast<const float4*>(lnp + f);
              const float4 c4 = *reinterpret_cast<const float4*>(lnp + 64 + f);
              float4 o;
              o.x = lr_fma(v[j][4 * g + 0] * rstd, w4.x, c4.x);
              o.y = lr_fma(v[j][4 * g + 1] * rstd, w4.y, c4.y);
              o.z = lr_fma(v[j][4 * g + 2] * rstd, w4.z, c4.z);
              o.w = lr_fma(v[j][4 * g + 3] * rstd, w4.w, c4.w);
              *reinterpret_cast<float4*>(y + f) = o;
            }
        }
      }
      EM_STAMP(11);
    }
  }
#ifdef LR_EXPERIMENTS
  if (p.stamp && tid == 0 && blockIdx.x < 16)
    for (int k = 0; k < 12; ++k) g_em_stamps[MODE][blockIdx.x][k] = st_acc[k];
#endif
}


// =============================================================================================
// em_pipe_kernel: the same layer pair, software-pipelined over super tiles (round 4)
// =============================================================================================
// em_layer_kernel runs a tile's stages one after the other with all eight waves in each: phase A (MFMA) -> recurrence (one
// to eight waves, vector ALU + LDS) -> phase B (MFMA on waves 0..3) -> LayerNorm + store; at Synth-1M the matrix pipe works
// for 16.4 k of a tile's ~29 k cycles (tools/gpu_em_stamps.sh). Here the workgroup is split into roles and two tiles are in
// flight:
//     waves 0..3 ("B waves", one per SIMD): phase B chain, residual, LayerNorm, store of tile j
//     waves 4..7 ("A waves", one per SIMD): staging, phase A (two 32-output blocks each) and the recurrence of tile j + 1
// with the 256-wide intermediate double-buffered in LDS. That needs 2 x 65 KiB, so the phase B weights (65 KiB in
// em_layer_kernel's LDS) live in the B waves' REGISTERS instead: 128 per lane, in the register array that holds the A
// waves' 64 phase A weights. One iteration is two intervals separated by workgroup barriers:
//     interval 1:  A waves: phase A of tile j+1 -> ul[(j+1)&1]      | B waves: LayerNorm + store of tile j-1, then the first
//                                                                   |          EP_B1 of the 16 groups of tile j's chain
//     interval 2:  A waves: recurrence of tile j+1, staging of j+2  | B waves: the rest of the chain, + residual, hand-over
// so every SIMD's matrix pipe has the A wave's and the B wave's chains to interleave in interval 1 and the B wave's in
// interval 2, while the recurrence, the staging, the epilogues and the LayerNorm run beside them on the other wave.
// Every chain is the one em_layer_kernel runs (same operands, same k order); which rows share a tile, and which wave
// walks which segment of the recurrence, never changes a row's arithmetic: bit-identical to em_layer_kernel and to the
// oracle. MODE 0 (LRU layer) and MODE 1 (feed-forward); the last block's two kernels (MODE 2, 3) stay on em_layer_kernel.
#define EP_B1 4

template <int MODE>
__global__ __launch_bounds__(EM_THREADS) __attribute__((amdgpu_waves_per_eu(2, 2))) void em_pipe_kernel(EmLayer p) {
  static_assert(MODE == 0 || MODE == 1, "em_pipe_kernel: LRU layer or feed-forward");
  constexpr bool SCAN = MODE == 0;
  extern __shared__ __attribute__((aligned(16))) float smem_f[];
  float* ul0 = smem_f;                       // [2][EM_ST][EM_US] the 256-wide intermediate of two tiles
  float* xs = ul0 + 2 * EM_ST * EM_US;       // [EM_ST][EM_XS]    phase A input, de-interleaved
  float* ex = xs + EM_ST * EM_XS;            // [2][16][64]       feature block 1 -> block 0 hand-over
  float* lnp = ex + 2 * 16 * 64;             // [2][64]           LayerNorm weight, bias
  int* tags0 = reinterpret_cast<int*>(lnp + 128);               // [2][EM_ST]
  float* erf_lds = reinterpret_cast<float*>(tags0 + 2 * EM_ST);  // [LR_ERF_NINT][EM_ERF_ROW]
  float* carry = erf_lds + LR_ERF_NINT * EM_ERF_ROW;            // [2][64 lanes][4]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int half = lane >> 5, col = lane & 31;
  const bool is_b = wave < 4;
#ifdef LR_EXPERIMENTS
  // experiment builds (tools/gpu_em_stamps.sh): s_memtime ticks per stage summed over the tiles, wave 0 (B role: slots
  // 0 LayerNorm | 1 chain part 1 | 2 wait | 3 chain part 2 + hand-over | 4 wait) and wave 4 (A role: 5 phase A | 6 wait |
  // 7 loads + recurrence + staging | 8 wait), first 16 workgroups
  unsigned long long st_acc[12] = {}, st_prev = __builtin_amdgcn_s_memtime();
#define EP_STAMP(k)                                                      \
  do {                                                                   \
    if (p.stamp) {                                                       \
      const unsigned long long now_ = __builtin_amdgcn_s_memtime();     \
      st_acc[k] += now_ - st_prev;                                       \
      st_prev = now_;                                                    \
    }                                                                    \
  } while (0)
#else
#define EP_STAMP(k) do { } while (0)
#endif
  const int aw = wave & 3;   // index inside the role

  int ra, rb, step;
  if (p.wg_row) {
    ra = p.wg_row[blockIdx.x];
    rb = p.wg_row[blockIdx.x + 1];
    step = EM_ST;
  } else {
    ra = blockIdx.x * EM_ST;
    rb = p.n_rows_ptr ? *p.n_rows_ptr : p.n_rows_fixed;
    step = gridDim.x * EM_ST;
  }
  if (ra >= rb) return;
  const int nt = (rb - ra + step - 1) / step;   // tiles of this workgroup: rows ra + j * step ..

  if (tid < 64) lnp[tid] = p.lnw[tid];
  else if (tid < 128) lnp[tid] = p.lnb[tid - 64];
  if (MODE == 1)
    for (int i = tid; i < LR_ERF_NINT * EM_ERF_ROW; i += EM_THREADS) {
      const int c = i % EM_ERF_ROW;
      erf_lds[i] = c <= LR_ERF_DEG ? em_erf_tab[(i / EM_ERF_ROW) * (LR_ERF_DEG + 1) + c] : 0.f;
    }
  // ---- roles. The two roles are two separate loops below (same number of barriers in each): written as one loop with a
  // branch per interval, every role's loop-carried registers (128 + 64 weights, accumulator, residual, staged rows ...)
  // would be live in every wave and hipcc spilled 216 of them.
  const int rt = (wave >> 1) & 1, jb = wave & 1;   // B waves: row tile, feature block
  // recurrence (A waves): lane = two complex channels k, k + 2 (neighbours in the de-interleaved row)
  const int sc_k = (lane >> 5) * 64 + 4 * (lane & 15) + ((lane >> 4) & 1);
  const int sc_pos = em_pos(sc_k);
  float lam_r0 = 0.f, lam_i0 = 0.f, lam_r1 = 0.f, lam_i1 = 0.f;
  if (SCAN) {
    lam_r0 = p.lam_re[sc_k];
    lam_i0 = p.lam_im[sc_k];
    lam_r1 = p.lam_re[sc_k + 2];
    lam_i1 = p.lam_im[sc_k + 2];
    *reinterpret_cast<float4*>(carry + 4 * lane) = make_float4(0.f, 0.f, 0.f, 0.f);  // every wave: same zeros
  }
  int par = 0;   // parity of the carry buffer the next recurrence reads

  if (!is_b) {
    // ================================================= A waves =================================================
    // phase A weights of my two output blocks 2 aw, 2 aw + 1 (step s uses k = 2 s + half)
    float wq[64], bq[2], gq[2];
#pragma unroll
    for (int bi = 0; bi < 2; ++bi) {
      const int out = (2 * aw + bi) * 32 + col;
#pragma unroll
      for (int s2 = 0; s2 < 32; ++s2) wq[bi * 32 + s2] = p.wa[(size_t)(2 * s2 + half) * 256 + out];
      bq[bi] = p.ba[out];
      gq[bi] = MODE == 1 ? 0.f : p.gamma[out & 127];
    }
  // ---- A-wave stages ------------------------------------------------------------------------------------------------
  const int atid = tid & 255;   // thread index inside the role
  // staging of tile `jt`: two (row, part) pairs per A thread -- 8 floats of a 64-float input row each, de-interleaved:
  // xs[row][h*32 + s] = x[row][2s + h] -- and the rows' tags. Loads and LDS writes are separate calls so that the loads can
  // be in flight across the recurrence.
  float4 pre[2][2];
  int pre_tag = 1;
  auto stage_load = [&](int jt) {
    const int r0 = ra + jt * step;
#pragma unroll
    for (int h2 = 0; h2 < 2; ++h2) em_load_rows8(pre[h2], p.IN, r0, rb, (atid >> 3) + 32 * h2, atid & 7);
    if (SCAN) pre_tag = (atid < EM_ST && r0 + atid < rb) ? p.row_tag[r0 + atid] : 1;
  };
  auto stage_store = [&](int jt) {
#pragma unroll
    for (int h2 = 0; h2 < 2; ++h2) {
      float* dst = xs + ((atid >> 3) + 32 * h2) * EM_XS + (atid & 7) * 4;
      *reinterpret_cast<float4*>(dst) = make_float4(pre[h2][0].x, pre[h2][0].z, pre[h2][1].x, pre[h2][1].z);
      *reinterpret_cast<float4*>(dst + 32) = make_float4(pre[h2][0].y, pre[h2][0].w, pre[h2][1].y, pre[h2][1].w);
    }
    if (SCAN && atid < EM_ST) tags0[(jt & 1) * EM_ST + atid] = pre_tag;
  };
  // phase A of the tile staged in xs -> ulb: my two output blocks for both 32-row tiles
  auto phase_a = [&](float* ulb) {
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      float a[32];  // token row t*32 + col, 32 de-interleaved values of my half
      const float* xr = xs + (t * 32 + col) * EM_XS + 32 * half;
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const float4 v4 = *reinterpret_cast<const float4*>(xr + 4 * q);
        a[4 * q + 0] = v4.x; a[4 * q + 1] = v4.y; a[4 * q + 2] = v4.z; a[4 * q + 3] = v4.w;
      }
#pragma unroll
      for (int bi = 0; bi < 2; ++bi) {
        floatx16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = bq[bi];
#pragma unroll
        for (int s2 = 0; s2 < 32; ++s2) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s2], wq[bi * 32 + s2], acc, 0, 0, 0);
        // D[token i][out]: lane = out column, register r = token (r&3) + 8*(r>>2) + 4*half
        const int blk = 2 * aw + bi;
        float* dst = ulb + (blk >> 1) * 64 + (col & 1) * 32 + (blk & 1) * 16 + (col >> 1);  // em_pos(blk*32 + col)
#pragma unroll
        for (int r8 = 0; r8 < 16; r8 += 4) {   // four values at a time (eight, as in em_layer_kernel, spill beside wq[64])
          float xin[4], val[4];
#pragma unroll
          for (int i = 0; i < 4; ++i) xin[i] = acc[r8 + i];
          if (MODE == 1) {
            em_gelu_tab12<4>(xin, val, erf_lds);
          } else {
#pragma unroll
            for (int i = 0; i < 4; ++i) val[i] = xin[i] * gq[bi];
          }
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const int r = r8 + i;
            const int tok = t * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
            dst[tok * EM_US] = val[i];
          }
        }
      }
    }
  };
  // recurrence over the 64 rows of ulb (MODE 0), four A waves: the rows are cut at users' first rows into up to four
  // segments (cuts at the first start at or after row 16, 32, 48); wave aw walks segment aw. Only segment 0 can continue a
  // user of the previous tile: its carry comes from the wave that walked row 63 there, through LDS.
  auto scan = [&](float* ulb, const int* tg) {
    const int tg_lane = tg[lane];
    const unsigned long long firsts = __ballot(tg_lane & 1);
    int sa = 0, sb = EM_ST;
    if (aw > 0) {
      const unsigned long long m = firsts & (~0ull << (16 * aw));
      sa = m ? __builtin_ctzll(m) : EM_ST;
    }
    if (aw < 3) {
      const unsigned long long m = firsts & (~0ull << (16 * (aw + 1)));
      sb = m ? __builtin_ctzll(m) : EM_ST;
    }
    sa = __builtin_amdgcn_readfirstlane(sa);
    sb = __builtin_amdgcn_readfirstlane(sb);
    if (sa < sb) {
      float h_r0, h_r1, h_i0, h_i1;
      if (aw == 0) {
        const float4 cv = *reinterpret_cast<const float4*>(carry + par * 256 + 4 * lane);
        h_r0 = cv.x; h_r1 = cv.y; h_i0 = cv.z; h_i1 = cv.w;
      } else {
        h_r0 = h_r1 = h_i0 = h_i1 = 0.f;  // overwritten by the segment's first row
      }
      float* base = ulb + sc_pos;
#define EP_SCAN_STEP(t_, bre_, bim_, is_first_)                                                        \
  {                                                                                                    \
    if (is_first_) {                                                                                   \
      h_r0 = (bre_).x; h_r1 = (bre_).y; h_i0 = (bim_).x; h_i1 = (bim_).y;                             \
    } else {                                                                                           \
      const float nr0 = lr_fma(lam_r0, h_r0, lr_fma(-lam_i0, h_i0, (bre_).x));                         \
      const float ni0 = lr_fma(lam_r0, h_i0, lr_fma(lam_i0, h_r0, (bim_).x));                          \
      const float nr1 = lr_fma(lam_r1, h_r1, lr_fma(-lam_i1, h_i1, (bre_).y));                         \
      const float ni1 = lr_fma(lam_r1, h_i1, lr_fma(lam_i1, h_r1, (bim_).y));                          \
      h_r0 = nr0; h_i0 = ni0; h_r1 = nr1; h_i1 = ni1;                                                  \
    }                                                                                                  \
    *reinterpret_cast<float2*>(base + (t_) * EM_US) = make_float2(h_r0, h_r1);                         \
    *reinterpret_cast<float2*>(base + (t_) * EM_US + 128) = make_float2(h_i0, h_i1);                   \
  }
      const unsigned long long upto = sb >= 64 ? ~0ull : ((1ull << sb) - 1ull);
      const unsigned long long inner_firsts = firsts & upto & ~((2ull << sa) - 1ull);
      if (inner_firsts == 0ull) {   // one user from sa to sb: no per-row first test, operands requested four rows ahead
        int t = sa;
        {
          const float2 b_re = *reinterpret_cast<const float2*>(base + t * EM_US);
          const float2 b_im = *reinterpret_cast<const float2*>(base + t * EM_US + 128);
          const bool first = (firsts >> t) & 1;   // wave-uniform
          EP_SCAN_STEP(t, b_re, b_im, first)
          ++t;
        }
        float2 cr[4], ci[4], nr[4], ni[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int tt = min(t + i, sb - 1);
          cr[i] = *reinterpret_cast<const float2*>(base + tt * EM_US);
          ci[i] = *reinterpret_cast<const float2*>(base + tt * EM_US + 128);
        }
        for (; t + 4 <= sb; t += 4) {
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const int tt = min(t + 4 + i, sb - 1);
            nr[i] = *reinterpret_cast<const float2*>(base + tt * EM_US);
            ni[i] = *reinterpret_cast<const float2*>(base + tt * EM_US + 128);
          }
#pragma unroll
          for (int i = 0; i < 4; ++i) EP_SCAN_STEP(t + i, cr[i], ci[i], false)
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            cr[i] = nr[i];
            ci[i] = ni[i];
          }
        }
#pragma unroll
        for (int i = 0; i < 3; ++i)
          if (t + i < sb) EP_SCAN_STEP(t + i, cr[i], ci[i], false)   // wave-uniform
      } else {
        for (int t0 = sa; t0 < sb; t0 += 4) {
          float2 br[4], bi2[4];
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const int t = min(t0 + i, sb - 1);
            br[i] = *reinterpret_cast<const float2*>(base + t * EM_US);
            bi2[i] = *reinterpret_cast<const float2*>(base + t * EM_US + 128);
          }
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const int t = t0 + i;
            if (t < sb) {  // wave-uniform
              const bool first = (firsts >> t) & 1;
              EP_SCAN_STEP(t, br[i], bi2[i], first)
            }
          }
        }
      }
#undef EP_SCAN_STEP
      // (the other buffer: wave 0 may not have read this tile's carry-in yet)
      if (sb == EM_ST) *reinterpret_cast<float4*>(carry + (par ^ 1) * 256 + 4 * lane) = make_float4(h_r0, h_r1, h_i0, h_i1);
    }
    par ^= 1;
  };


    // prologue: tile 0 through staging, phase A and the recurrence; tile 1 staged
    stage_load(0);
    stage_store(0);
    EM_BARRIER();
    if (nt > 1) stage_load(1);
    phase_a(ul0);
    EM_BARRIER();
    if (SCAN) scan(ul0, tags0);
    if (nt > 1) stage_store(1);
    EM_BARRIER();
    EP_STAMP(11);
    for (int j = 0; j < nt; ++j) {
      float* const uln = ul0 + ((j + 1) & 1) * (EM_ST * EM_US);
      if (j + 1 < nt) phase_a(uln);                                         // interval 1
      EP_STAMP(5);
      EM_BARRIER();
      EP_STAMP(6);
      if (j + 2 < nt) stage_load(j + 2);                                    // interval 2
      if (SCAN && j + 1 < nt) scan(uln, tags0 + ((j + 1) & 1) * EM_ST);
      if (j + 2 < nt) stage_store(j + 2);
      EP_STAMP(7);
      EM_BARRIER();
      EP_STAMP(8);
    }
#ifdef LR_EXPERIMENTS
    if (p.stamp && tid == 256 && blockIdx.x < 16)
      for (int k = 5; k < 12; ++k) g_em_stamps[MODE][blockIdx.x][k] = st_acc[k];
#endif
  } else {
    // ================================================= B waves =================================================
    // 128 phase B weights of feature jb*32 + col: chain step i uses k = (i >> 5) * 64 + 2 * (i & 31) + half, the order
    // em_layer_kernel reads them from LDS
    float wreg[128];
    {
      const int f = jb * 32 + col;
#pragma unroll
      for (int i = 0; i < 128; ++i) wreg[i] = p.wb[(size_t)((i >> 5) * 64 + 2 * (i & 31) + half) * 64 + f];
    }
  // ---- B-wave stages ------------------------------------------------------------------------------------------------
  floatx16 acc_b;     // the chain's accumulator, alive across the barrier between its two parts
  float4 r4[4];       // residual of my row, requested before the chain
  float v0[16];       // my 16 features (+ residual) of the finished tile, alive until its LayerNorm in the next interval 1
  int ln_gr = 0;      // global row of v0's token, -1: none / not live
  auto chain_begin = [&](int jt) {
    const int r0 = ra + jt * step;
    const int lrow = rt * 32 + col;
    const bool live = r0 + lrow < rb;
    const int gr = r0 + lrow;
    const float* res = p.RES + (size_t)(live ? (p.res_rows ? p.res_rows[gr] : gr) : 0) * 64;
#pragma unroll
    for (int g = 0; g < 4; ++g)
      r4[g] = live ? *reinterpret_cast<const float4*>(res + jb * 32 + 8 * g + 4 * half) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int r = 0; r < 16; ++r) acc_b[r] = p.bb[jb * 32 + (r & 3) + 8 * (r >> 2) + 4 * half];   // (L1-resident; 16 registers less)
  };
  // groups [G0, G1) of the 16-group chain (8 k steps each) on the tile in ulb; the operands of the next group are requested
  // before the current group's 8 MFMAs
#define EP_CHAIN(ulb_, G0, G1)                                                                             \
  {                                                                                                        \
    const float* tr = (ulb_) + (rt * 32 + col) * EM_US + 32 * half;   /* token row rt*32 + col */           \
    float4 bc[2], bn[2];                                                                                   \
    {                                                                                                      \
      const int o0 = ((G0) >> 2) * 64 + ((G0)&3) * 8;                                                      \
      bc[0] = *reinterpret_cast<const float4*>(tr + o0);                                                   \
      bc[1] = *reinterpret_cast<const float4*>(tr + o0 + 4);                                               \
    }                                                                                                      \
    _Pragma("unroll") for (int g = (G0); g < (G1); ++g) {                                                  \
      if (g + 1 < (G1)) {                                                                                  \
        const int o = ((g + 1) >> 2) * 64 + ((g + 1) & 3) * 8;                                             \
        bn[0] = *reinterpret_cast<const float4*>(tr + o);                                                  \
        bn[1] = *reinterpret_cast<const float4*>(tr + o + 4);                                              \
      }                                                                                                    \
      __builtin_amdgcn_sched_barrier(0);                                                                   \
      _Pragma("unroll") for (int i = 0; i < 2; ++i) {                                                      \
        acc_b = __builtin_amdgcn_mfma_f32_32x32x2f32(wreg[g * 8 + 4 * i + 0], bc[i].x, acc_b, 0, 0, 0);    \
        acc_b = __builtin_amdgcn_mfma_f32_32x32x2f32(wreg[g * 8 + 4 * i + 1], bc[i].y, acc_b, 0, 0, 0);    \
        acc_b = __builtin_amdgcn_mfma_f32_32x32x2f32(wreg[g * 8 + 4 * i + 2], bc[i].z, acc_b, 0, 0, 0);    \
        acc_b = __builtin_amdgcn_mfma_f32_32x32x2f32(wreg[g * 8 + 4 * i + 3], bc[i].w, acc_b, 0, 0, 0);    \
      }                                                                                                    \
      bc[0] = bn[0];                                                                                       \
      bc[1] = bn[1];                                                                                       \
    }                                                                                                      \
  }
  auto chain_end = [&](int jt) {   // + residual; block 1 hands its 16 values to the block-0 wave of its row tile
    const int r0 = ra + jt * step;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      v0[4 * g + 0] = acc_b[4 * g + 0] + r4[g].x;
      v0[4 * g + 1] = acc_b[4 * g + 1] + r4[g].y;
      v0[4 * g + 2] = acc_b[4 * g + 2] + r4[g].z;
      v0[4 * g + 3] = acc_b[4 * g + 3] + r4[g].w;
    }
    if (jb == 1) {
#pragma unroll
      for (int r = 0; r < 16; ++r) ex[(rt * 16 + r) * 64 + lane] = v0[r];
    }
    ln_gr = (r0 + rt * 32 + col < rb) ? r0 + rt * 32 + col : -1;
  };
  auto layer_norm_store = [&]() {   // block-0 waves: the LayerNorm butterfly over my 16 + the handed-over 16 features
    float v[2][16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      v[0][r] = v0[r];
      v[1][r] = ex[(rt * 16 + r) * 64 + lane];
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
    if (ln_gr >= 0) {
      float* y = p.OUT + (size_t)ln_gr * 64;
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int f = j * 32 + 8 * g + 4 * half;
          const float4 w4 = *reinterpret_cast<const float4*>(lnp + f);
          const float4 c4 = *reinterpret_cast<const float4*>(lnp + 64 + f);
          float4 o;
          o.x = lr_fma(v[j][4 * g + 0] * rstd, w4.x, c4.x);
          o.y = lr_fma(v[j][4 * g + 1] * rstd, w4.y, c4.y);
          o.z = lr_fma(v[j][4 * g + 2] * rstd, w4.z, c4.z);
          o.w = lr_fma(v[j][4 * g + 3] * rstd, w4.w, c4.w);
          *reinterpret_cast<float4*>(y + f) = o;
        }
    }
  };


    EM_BARRIER();   // the A waves' prologue: three barriers
    EM_BARRIER();
    EM_BARRIER();
    EP_STAMP(10);
    for (int j = 0; j < nt; ++j) {
      float* const ulj = ul0 + (j & 1) * (EM_ST * EM_US);
      if (j > 0 && jb == 0) layer_norm_store();   // interval 1: tile j - 1 (its hand-over was written before the last barrier)
      EP_STAMP(0);
      chain_begin(j);
      EP_CHAIN(ulj, 0, EP_B1)
      EP_STAMP(1);
      EM_BARRIER();
      EP_STAMP(2);
      EP_CHAIN(ulj, EP_B1, 16)                     // interval 2
      chain_end(j);
      EP_STAMP(3);
      EM_BARRIER();
      EP_STAMP(4);
    }
    if (jb == 0) layer_norm_store();   // the last tile
#ifdef LR_EXPERIMENTS
    if (p.stamp && tid == 0 && blockIdx.x < 16) {
      for (int k = 0; k < 5; ++k) g_em_stamps[MODE][blockIdx.x][k] = st_acc[k];
      g_em_stamps[MODE][blockIdx.x][10] = st_acc[10];
    }
#endif
  }
#undef EP_CHAIN
#undef EP_STAMP
}

#ifdef LR_EXPERIMENTS
extern "C" int lr_debug_em_stamps(unsigned long long* out, int n) {
  if (!out || n != 4 * 16 * 12) LR_FAIL(LR_EINVAL, "lr_debug_em_stamps: bad arguments");
  LR_CHECK_HIP(hipMemcpyFromSymbol(out, HIP_SYMBOL(g_em_stamps), (size_t)n * sizeof(unsigned long long)));
  return LR_OK;
}
#endif

// =============================================================================================
#define EM_CHUNK_ROWS (1 << 21)  // rows per chunk of users (workspace: 0.5 KB per row, at most 1.1 GB; proportional to B*L below that)
#define EM_MAX_WGS 256           // layer workgroups: one per CU (156 KiB of LDS each)

static size_t em_chunk_users(int L) {
  size_t u = EM_CHUNK_ROWS / (size_t)L;
  return u < 1 ? 1 : u;
}

size_t lr_encoder_mfma_workspace_bytes(int B, int L) {
  const size_t users = (size_t)B < em_chunk_users(L) ? (size_t)B : em_chunk_users(L);
  const size_t rows = users * L;
  size_t o = 0;
  o += lr_align_up((2 * users + (users + 1) + (EM_MAX_WGS + 1) + 2 * rows) * sizeof(int), 256);
  o += lr_align_up(rows * 64 * sizeof(float), 256) * 2;   // X, Y
  o += lr_align_up(users * 256 * sizeof(float), 256);     // recurrence state of the last rows
  o += lr_align_up(users * 64 * sizeof(float), 256);      // Y of the last rows
  return o;
}

// LR_EM_PIPE=0 in the environment (A/B runs) or lr_lru_set_encoder_pipeline(h, 0) (the bit-identity test of the two kernels,
// tests/test_gpu_lru.py): the LRU layer on em_layer_kernel (one tile at a time) instead of em_pipe_kernel. Same bits.
static bool em_env_pipe() {
  static int v = -1;
  if (v < 0) {
    const char* e = getenv("LR_EM_PIPE");
    v = (e && e[0] == '0') ? 0 : 1;
  }
  return v != 0;
}

template <int MODE>
static int em_launch_layer(int grid, size_t lds, hipStream_t st, const EmLayer& p, bool pipe = true) {
  // The pipelined kernel serves the LRU layer only. Measured at Synth-1M / Beauty (gpurun_out/r4s5): LRU layer 425 -> 395 us /
  // 144 -> 139 us, feed-forward 409 -> 449 us / 152 -> 171 us. The stamps say why the gain is small and the feed-forward
  // loses: v_mfma_f32_32x32x2_f32 runs on the f32 vector datapath, so a SIMD's vector work (recurrence, LayerNorm, GELU) does
  // NOT hide behind its partner wave's chain -- LayerNorm takes 6.6 k cycles per tile beside phase A against 1.5 k alone,
  // the recurrence 12.3 k beside the phase B chain against 5.5 k -- and the roles only move where a SIMD's MFMA + vector
  // cycles are spent: the SIMD whose A wave walks a long user's 64 rows carries 16.4 k + 5.5 k + epilogues while the others
  // wait at the barrier, and four A waves evaluate the GELU that eight waves shared.
  if constexpr (MODE == 0) {
    if (pipe && em_env_pipe()) {
      // [2][64][EM_US] intermediates | xs | hand-over | LayerNorm | 2 x tags | erf table | carry
      const size_t lds_pipe = (size_t)(2 * EM_ST * EM_US + EM_ST * EM_XS + 2 * 16 * 64 + 128 + 2 * EM_ST + LR_ERF_NINT * EM_ERF_ROW + 512) * sizeof(float);
      static bool lds_set_pipe[LR_MAX_DEVICES] = {};
      if (int rc = lr_ensure_dynamic_lds(reinterpret_cast<const void*>(em_pipe_kernel<MODE>), (int)lds_pipe, lds_set_pipe)) return rc;
      hipLaunchKernelGGL(em_pipe_kernel<MODE>, dim3(grid), dim3(EM_THREADS), lds_pipe, st, p);
      LR_CHECK_LAUNCH("em_pipe_kernel");
      return LR_OK;
    }
  }
  static bool lds_set[LR_MAX_DEVICES] = {};
  if (int rc = lr_ensure_dynamic_lds(reinterpret_cast<const void*>(em_layer_kernel<MODE>), (int)lds, lds_set)) return rc;
  hipLaunchKernelGGL(em_layer_kernel<MODE>, dim3(grid), dim3(EM_THREADS), lds, st, p);
  LR_CHECK_LAUNCH("em_layer_kernel");
  return LR_OK;
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
  int* ibuf = (int*)take((2 * users_cap + (users_cap + 1) + (EM_MAX_WGS + 1) + 2 * rows_cap) * sizeof(int));
  float* X = (float*)take(rows_cap * 64 * sizeof(float));
  float* Y = (float*)take(rows_cap * 64 * sizeof(float));
  float* Ul = (float*)take(users_cap * 256 * sizeof(float));
  float* Yl = (float*)take(users_cap * 64 * sizeof(float));
  const size_t lds = (size_t)(64 * EM_WS + EM_ST * EM_US + EM_ST * EM_XS + 2 * 16 * 64 + 128 + EM_ST + LR_ERF_NINT * EM_ERF_ROW + 512) * sizeof(float);
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
    c.last_row = c.off + (users_cap + 1);
    c.wg_row = c.last_row + users_cap;
    c.row_tag = c.wg_row + (EM_MAX_WGS + 1);
    c.row_item = c.row_tag + rows_cap;
    const size_t max_st = ((size_t)c.users * L + EM_ST - 1) / EM_ST;   // super tiles if every position were live
    c.G = (int)(max_st < EM_MAX_WGS ? max_st : EM_MAX_WGS);
    const int last_st = (c.users + EM_ST - 1) / EM_ST;
    const int grid_last = last_st < EM_MAX_WGS ? last_st : EM_MAX_WGS;
    hipLaunchKernelGGL(em_live_kernel, dim3((c.users + 3) / 4), dim3(256), 0, st, c);
    LR_CHECK_LAUNCH("em_live_kernel");
    hipLaunchKernelGGL(em_offsets_kernel, dim3(1), dim3(1024), 0, st, c);
    LR_CHECK_LAUNCH("em_offsets_kernel");
    const long long cells = (long long)c.users * L;
    hipLaunchKernelGGL(em_rows_kernel, dim3((unsigned)((cells + 255) / 256)), dim3(256), 0, st, c);
    LR_CHECK_LAUNCH("em_rows_kernel");
    const long long embed_wgs = (cells + 15) / 16;  // upper bound of the live rows / 16
    hipLaunchKernelGGL(em_embed_kernel, dim3((unsigned)(embed_wgs < 256 * 16 ? embed_wgs : 256 * 16)), dim3(256), 0, st, c, img, lay, X);
    LR_CHECK_LAUNCH("em_embed_kernel");
    for (int b = 0; b < nb; ++b) {
      const LrLruBlockLayout& BL = lay.blk[b];
      EmLayer lru = {};
#ifdef LR_EXPERIMENTS
      static const int stamp = getenv("LR_EM_STAMPS") ? atoi(getenv("LR_EM_STAMPS")) : 0;
      lru.stamp = stamp;
#endif
      lru.wa = img + BL.in_wt; lru.ba = img + BL.in_b; lru.gamma = img + BL.gamma;
      lru.lam_re = img + BL.lam_re; lru.lam_im = img + BL.lam_im;
      lru.wb = img + BL.out_wt; lru.bb = img + BL.out_b; lru.lnw = img + BL.ln1_w; lru.lnb = img + BL.ln1_b;
      lru.IN = X; lru.RES = X; lru.row_tag = c.row_tag; lru.wg_row = c.wg_row;
      EmLayer ffn = {};
#ifdef LR_EXPERIMENTS
      ffn.stamp = stamp;
#endif
      ffn.wa = img + BL.w1t; ffn.ba = img + BL.b1;
      ffn.wb = img + BL.w2t; ffn.bb = img + BL.b2; ffn.lnw = img + BL.ln2_w; ffn.lnb = img + BL.ln2_b;
      if (b < nb - 1) {
        lru.OUT = Y;
        if (int rc = em_launch_layer<0>(c.G, lds, st, lru, h->encoder_pipeline != 0)) return rc;
        ffn.IN = Y; ffn.RES = Y; ffn.OUT = X; ffn.n_rows_ptr = c.off + c.users;  // no recurrence: super tiles strided over the rows
        if (int rc = em_launch_layer<1>(c.G, lds, st, ffn)) return rc;
      } else {  // only each user's last row is consumed after the last block
        lru.OUT = Ul;
        if (int rc = em_launch_layer<2>(c.G, lds, st, lru)) return rc;
        EmLayer outp = lru;
        outp.IN = Ul; outp.res_rows = c.last_row; outp.OUT = Yl; outp.wg_row = nullptr; outp.n_rows_fixed = c.users;
        if (int rc = em_launch_layer<3>(grid_last, lds, st, outp)) return rc;
        ffn.IN = Yl; ffn.RES = Yl; ffn.OUT = out_q + u0 * 64; ffn.n_rows_fixed = c.users;
        if (int rc = em_launch_layer<1>(grid_last, lds, st, ffn)) return rc;
      }
    }
  }
  return LR_OK;
}
