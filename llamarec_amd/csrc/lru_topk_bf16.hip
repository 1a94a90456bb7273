// lru_topk_bf16.hip -- item_bound_kernel / item_cand_kernel: every (user, item) score APPROXIMATELY on
// v_mfma_f32_32x32x16_bf16, once for the maxima of tile groups (-> a proven threshold per user, bound_select_kernel in
// lru_topk.hip) and once more to list every item that reaches it (-> exact rescoring, cand_rescore_kernel).
// Replaces (reference): scores = x @ E^T + bias, model/lru.py:85, as far as finding the top-K needs it
// (trainer/lru.py:82-84,113-115); the exact values come from cand_rescore_kernel.
//
// THIS FILE IS COMPILED WITH -fno-honor-nans (csrc/Makefile): hipcc puts a `v_max_f32 x, x` (sNaN quieting, IEEE mode) in
// front of every fmaxf on an MFMA output, which doubled the vector instructions of the maximum; without NaNs to honour a
// 16-register maximum is 8 v_max / v_max3. Nothing here relies on NaN semantics: the padding rows of the table's last
// tile score -inf (their accumulator start value, `bias_tail`), users past B get threshold +inf, and infinities ARE
// honoured. The exact kernels (NaN bias on padding rows) stay in lru_topk.hip, compiled as before.
//
// Round 4: the table stream. Rounds 2-3 let every wave fetch its A fragments from global memory (4 KiB per tile and
// wave, L1 hits for seven of the eight waves): 32 KiB per tile and workgroup through a 64 B/clk L1 = 512 cycles, exactly
// the 512 cycles the tile's 64 MFMAs take on the four SIMDs -- the passes ran at 0.22 of the bf16 MFMA peak with the
// vector-memory path as busy as the matrix pipe. Now a tile crosses the L1 once per WORKGROUP: the packed tile (4 KiB of
// fragments, already in MFMA A-fragment order, so the LDS image is lane-linear and its ds_read_b128 are conflict-free)
// and its 32 biases are DMA'd into a two-stage LDS ring, ST tiles per stage, and every wave reads fragments and biases
// from LDS (256 B/clk). The bias rides into the product as the MFMA's C operand (accumulator start value = bias of the
// lane's 16 item rows), so no vector add is spent on it: the approximate score is fl(bias + sum) in whatever order the
// matrix pipe adds, covered by delta's gamma term over 66 terms (bound_select_kernel). One barrier per stage; the next
// stage's DMA is issued right behind it and lands while the current one is multiplied.
// The DMA is inline asm on purpose: issued through the builtin, hipcc counts it as a pending LDS write and drains
// vmcnt(0) in front of the next ds_read (llama_attn.hip, "the compiler's hidden wait"); the one wait that is needed is
// written by hand in front of the stage's barrier. M0 is written inside the asm block only (tests/test_isa_checks.py
// verifies that no compiler-emitted instruction of these kernels reads M0).
#include "lru_topk_bf16.h"

typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef __bf16 tk_bf16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ void tk_glds16(const void* gsrc, const void* lds_wave_base) {   // 64 lanes x 16 B -> 1 KiB
  const unsigned m0v = (unsigned)(size_t)((__attribute__((address_space(3))) const char*)lds_wave_base);
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" ::"s"(m0v), "v"(gsrc) : "memory");
}
__device__ __forceinline__ void tk_glds4(const void* gsrc, const void* lds_wave_base) {    // 64 lanes x 4 B -> 256 B
  const unsigned m0v = (unsigned)(size_t)((__attribute__((address_space(3))) const char*)lds_wave_base);
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dword %1, off" ::"s"(m0v), "v"(gsrc) : "memory");
}

template <int ST>
struct TkStage {
  static constexpr int FRAG_BYTES = ST * 4096;           // ST tiles x 4 MFMA steps x 64 lanes x 16 B
  static constexpr int BIAS_BYTES = ST * 128;            // ST tiles x 32 floats
  static constexpr int BYTES = FRAG_BYTES + BIAS_BYTES;  // 16-byte multiple
};

// Stage `tile0 .. tile0 + ST` -> `buf`. Tile indices are clamped to the table (a duplicate of the last tile is never
// CONSUMED by the candidate pass and only repeats a maximum in the bound pass); every wave issues ST / 2 fragment pieces,
// the first ST / 2 waves one bias piece (2 tiles' biases) each. All 64 lanes are active in every DMA instruction.
template <int ST>
__device__ __forceinline__ void tk_stage_issue(const unsigned short* emb16, const float* bias, const float* bias_tail,
                                               int tile0, int n_tiles, char* buf, int wave, int lane) {
  static_assert(ST % 2 == 0 && ST * 4 % 8 == 0, "a stage is whole pieces per wave");
#pragma unroll
  for (int i = 0; i < ST / 2; ++i) {
    const int pc = wave + 8 * i;   // piece = (tile of the stage, MFMA step)
    const int t = min(tile0 + (pc >> 2), n_tiles - 1);
    tk_glds16(reinterpret_cast<const char*>(emb16) + (size_t)t * 4096 + (pc & 3) * 1024 + lane * 16, buf + pc * 1024);
  }
  if (wave < ST / 2) {
    const int row = min((tile0 + 2 * wave) * 32 + lane, n_tiles * 32 - 1);
    const int tail0 = (n_tiles - 1) * 32;
    tk_glds4(row >= tail0 ? bias_tail + (row - tail0) : bias + row, buf + TkStage<ST>::FRAG_BYTES + wave * 256);
  }
}

// the users' q rows as MFMA B operands: q[user][k], k = 32 half + 8 s + j for MFMA step s -- the k order of the packed A
// fragments; any order is as good as another for a sum that only has to be APPROXIMATELY the score. UC column tiles of
// 32 users per wave share every A fragment (a column tile's MFMA sequence does not depend on UC: the same scores).
template <int UC>
__device__ __forceinline__ void tk_load_q_bf16(const float* q, int B, int user0, int col, int half, int (&user)[UC],
                                               tk_bf16x8 (&bq)[UC][4]) {
#pragma unroll
  for (int c = 0; c < UC; ++c) {
    user[c] = user0 + c * 32 + col;
    const float4* qp = reinterpret_cast<const float4*>(q + (size_t)(user[c] < B ? user[c] : 0) * 64 + 32 * half);
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const float4 a = qp[2 * s], b = qp[2 * s + 1];
      bq[c][s][0] = (__bf16)a.x; bq[c][s][1] = (__bf16)a.y; bq[c][s][2] = (__bf16)a.z; bq[c][s][3] = (__bf16)a.w;
      bq[c][s][4] = (__bf16)b.x; bq[c][s][5] = (__bf16)b.y; bq[c][s][6] = (__bf16)b.z; bq[c][s][7] = (__bf16)b.w;
    }
  }
}

// One 32-item tile against one of the wave's two 32-user column tiles, operands from the LDS stage: acc[4 g + e] =
// approximate score of item 32 tile + 8 g + 4 half + e for user c * 32 + col. THE instruction sequence both passes share
// (TkTile::load, then TkTile::scores per column tile).
struct TkTile {
  tk_bf16x8 a[4];
  floatx16 ci;   // the 16 item rows' biases: the accumulator's start value
  __device__ __forceinline__ void load(const char* tb, const char* bb, int lane, int half) {
#pragma unroll
    for (int s = 0; s < 4; ++s) a[s] = *reinterpret_cast<const tk_bf16x8*>(tb + s * 1024 + lane * 16);
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const float4 b = *reinterpret_cast<const float4*>(bb + (8 * g + 4 * half) * 4);
      ci[4 * g] = b.x; ci[4 * g + 1] = b.y; ci[4 * g + 2] = b.z; ci[4 * g + 3] = b.w;
    }
  }
  __device__ __forceinline__ floatx16 scores(const tk_bf16x8 (&bqc)[4]) const {
    floatx16 acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], bqc[0], ci, 0, 0, 0);
#pragma unroll
    for (int s = 1; s < 4; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[s], bqc[s], acc, 0, 0, 0);
    return acc;
  }
};

// max of the 16 accumulator registers (no NaNs to honour in this file: 1 v_max + 7 v_max3, no canonicalisation)
__device__ __forceinline__ float tk_max16(const floatx16& v) {
  float m = fmaxf(v[0], v[1]);
#pragma unroll
  for (int r = 2; r < 16; r += 2) m = fmaxf(fmaxf(m, v[r]), v[r + 1]);
  return m;
}

#define TK_BOUND_ST 8   // tiles per LDS stage of the bound pass: 2 x 33 KiB -> two workgroups per CU

// GROUPED (gshift >= 2, catalogs past 65 536 items): one maximum per group of 2^gshift tiles -- a lane keeps the running
// maximum of ITS 16 item rows and the two lane halves meet once per group. !GROUPED (gshift = 0): one maximum per tile,
// four tiles' maxima stored as one float4.
template <bool GROUPED, int UC>   // UC column tiles of 32 users per wave: 256 UC users per workgroup
__global__ __launch_bounds__(512, 4) void item_bound_kernel(BoundParams p) {
  constexpr int ST = TK_BOUND_ST;
  __shared__ __attribute__((aligned(16))) char smem[2 * TkStage<ST>::BYTES];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave-uniform: the DMA's LDS base goes through M0 (an SGPR)
  const int half = lane >> 5, col = lane & 31;
  const TkWho who = tk_who(p.n_chunks, p.n_user_groups);
  if (who.chunk < 0) return;
  const int tile_begin = who.chunk * p.tiles_per_chunk;
  const int tile_end = min(p.n_tiles, tile_begin + p.tiles_per_chunk);
  if (tile_begin >= tile_end) return;
  const int n_stages = (tile_end - tile_begin + ST - 1) / ST;
  tk_stage_issue<ST>(p.emb16, p.bias, p.bias_tail, tile_begin, p.n_tiles, smem, wave, lane);
  int user[UC];
  tk_bf16x8 bq[UC][4];
  tk_load_q_bf16<UC>(p.q, p.B, (who.group * 8 + wave) * (32 * UC), col, half, user, bq);
  float gm[UC];   // GROUPED: running maximum of the lane's rows over the current group
  float4 g4[UC];  // GROUPED: the maxima of four consecutive groups, stored as one 16-byte piece (a 4-byte store per user and group
                  // touched a line each: 259 MB of write traffic per launch for 32 MB of maxima, measured in round 4)
#pragma unroll
  for (int c = 0; c < UC; ++c) g4[c] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
  for (int c = 0; c < UC; ++c) gm[c] = -__builtin_inff();
  for (int st = 0; st < n_stages; ++st) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's pieces of stage st have landed ...
    __syncthreads();                                   // ... everyone's have, and everyone is done reading stage st - 1
    char* const buf = smem + (st & 1) * TkStage<ST>::BYTES;
    if (st + 1 < n_stages)
      tk_stage_issue<ST>(p.emb16, p.bias, p.bias_tail, tile_begin + (st + 1) * ST, p.n_tiles, smem + ((st + 1) & 1) * TkStage<ST>::BYTES,
                         wave, lane);
    if constexpr (GROUPED) {
      // (an explicit two-register-set prefetch of tile u + 1's fragments ahead of tile u's MFMAs does not fit the 128
      // VGPRs that two workgroups per CU leave a wave: hipcc spilled 85 registers; four waves per SIMD cover the reads)
#pragma unroll 1
      for (int u = 0; u < ST; ++u) {
        const int tile = tile_begin + st * ST + u;
        if (tile >= tile_end) break;   // wave-uniform
        TkTile tl;   // (a tile past the table cannot occur here: tile < tile_end <= n_tiles)
        tl.load(buf + u * 4096, buf + TkStage<ST>::FRAG_BYTES + u * 128, lane, half);
#pragma unroll
        for (int c = 0; c < UC; ++c) {
          gm[c] = fmaxf(gm[c], tk_max16(tl.scores(bq[c])));
        }
        if ((((tile + 1) >> p.gshift) != (tile >> p.gshift)) || tile + 1 >= tile_end) {   // last tile of its group (wave-uniform)
          const int grp = tile >> p.gshift, slot = grp & 3;   // (a chunk starts at a multiple of four groups or is the ragged last one:
                                                              // tmax rows are padded to a multiple of 4, so a whole float4 is always in bounds)
          const bool flush = slot == 3 || tile + 1 >= tile_end;
#pragma unroll
          for (int c = 0; c < UC; ++c) {
            const float m = fmaxf(gm[c], __shfl_xor(gm[c], 32, 64));   // the other lane half holds the tiles' other 16 rows
            g4[c].x = slot == 0 ? m : g4[c].x;
            g4[c].y = slot == 1 ? m : g4[c].y;
            g4[c].z = slot == 2 ? m : g4[c].z;
            g4[c].w = slot == 3 ? m : g4[c].w;
            if (flush && user[c] < p.B && half == 0) *reinterpret_cast<float4*>(p.tmax + (size_t)user[c] * p.ld + (grp & ~3)) = g4[c];
            gm[c] = -__builtin_inff();
          }
        }
      }
    } else {
#pragma unroll 1
      for (int u4 = 0; u4 < ST; u4 += 4) {
        const int t4 = tile_begin + st * ST + u4;
        if (t4 >= tile_end) break;   // wave-uniform (tile_begin and tiles_per_chunk are multiples of 4)
        // four tiles per pass of this loop, one at a time (unrolled, hipcc requests all four tiles' fragments first: 149
        // VGPRs against the 128 that two workgroups per CU leave a wave); their maxima collect in a float4 by selects.
        // Tiles past the table are the DMA's duplicates of the last tile: they repeat its maximum in slots nobody reads
        float4 m4[UC];
#pragma unroll
        for (int c = 0; c < UC; ++c) m4[c] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 1
        for (int u = 0; u < 4; ++u) {
          TkTile tl;
          tl.load(buf + (u4 + u) * 4096, buf + TkStage<ST>::FRAG_BYTES + (u4 + u) * 128, lane, half);
#pragma unroll
          for (int c = 0; c < UC; ++c) {
            float m = tk_max16(tl.scores(bq[c]));   // padding rows of the last tile: -inf (their accumulator start value)
            m = fmaxf(m, __shfl_xor(m, 32, 64));    // the other lane half holds the tile's other 16 items
            m4[c].x = u == 0 ? m : m4[c].x;
            m4[c].y = u == 1 ? m : m4[c].y;
            m4[c].z = u == 2 ? m : m4[c].z;
            m4[c].w = u == 3 ? m : m4[c].w;
          }
        }
#pragma unroll
        for (int c = 0; c < UC; ++c)
          if (user[c] < p.B && half == 0) *reinterpret_cast<float4*>(p.tmax + (size_t)user[c] * p.ld + t4) = m4[c];
      }
    }
  }
}

#define TK_CAND_ST 4      // tiles per LDS stage of the candidate pass (2 x 16.5 KiB)

// Passing items are collected in LDS (one list per user of the workgroup, LDS atomics only) and appended to the user's
// global list once per workgroup: a returning global atomic per passing element inside the tile loop cost 4x the
// scoring itself. Same operands, same instruction sequence (TkTile) as item_bound_kernel: the same approximate scores.
template <int UC>
__global__ __launch_bounds__(512, 4) void item_cand_kernel(CandParams p) {
  constexpr int ST = TK_CAND_ST;
  constexpr int USERS = 256 * UC, LCAP = TK_CAND_LIST_SLOTS / USERS;   // 40 KiB of 16-bit list slots per workgroup
  __shared__ __attribute__((aligned(16))) char smem[2 * TkStage<ST>::BYTES + USERS * 4 + TK_CAND_LIST_SLOTS * 2];
  int* const lcnt = reinterpret_cast<int*>(smem + 2 * TkStage<ST>::BYTES);
  unsigned short* const llist = reinterpret_cast<unsigned short*>(smem + 2 * TkStage<ST>::BYTES + USERS * 4);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave-uniform: the DMA's LDS base goes through M0 (an SGPR)
  const int half = lane >> 5, col = lane & 31;
  const TkWho who = tk_who(p.n_chunks, p.n_user_groups);
  if (who.chunk < 0) return;
  const int tile_begin = who.chunk * p.tiles_per_chunk;
  const int tile_end = min(p.n_tiles, tile_begin + p.tiles_per_chunk);
  if (tile_begin >= tile_end) return;
  const int n_stages = (tile_end - tile_begin + ST - 1) / ST;
  tk_stage_issue<ST>(p.emb16, p.bias, p.bias_tail, tile_begin, p.n_tiles, smem, wave, lane);
  for (int i = tid; i < USERS; i += 512) lcnt[i] = 0;
  int user[UC];
  float thr[UC];
  tk_bf16x8 bq[UC][4];
  tk_load_q_bf16<UC>(p.q, p.B, (who.group * 8 + wave) * (32 * UC), col, half, user, bq);
#pragma unroll
  for (int c = 0; c < UC; ++c) thr[c] = user[c] < p.B ? p.cand_thresh[user[c]] : __builtin_inff();
  for (int st = 0; st < n_stages; ++st) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    char* const buf = smem + (st & 1) * TkStage<ST>::BYTES;
    if (st + 1 < n_stages)
      tk_stage_issue<ST>(p.emb16, p.bias, p.bias_tail, tile_begin + (st + 1) * ST, p.n_tiles, smem + ((st + 1) & 1) * TkStage<ST>::BYTES,
                         wave, lane);
#pragma unroll 1
    for (int u = 0; u < ST; ++u) {
      const int tile = tile_begin + st * ST + u;
      if (tile >= tile_end) break;   // wave-uniform
      TkTile tl;
      tl.load(buf + u * 4096, buf + TkStage<ST>::FRAG_BYTES + u * 128, lane, half);
#pragma unroll
      for (int c = 0; c < UC; ++c) {
        const floatx16 acc = tl.scores(bq[c]);
        // the common path: the 16-register maximum (8 v_max3) against the user's threshold and one wave-uniform branch;
        // only when some lane's maximum reaches it (~ a third of the column tiles at 1 M items) are the 16 registers
        // compared one by one. (Measured and dropped: 16 compares straight into scalar lane masks, OR-ed on the scalar
        // unit, instead of the maximum -- 805 against 566 us at 1 M items.)
        // padding rows: -inf; should a user have no bound (threshold -inf) its list overflows and the exact pass takes over
        if (__ballot(tk_max16(acc) >= thr[c]) != 0ull) {   // wave-uniform
          unsigned mask = 0u;   // bit 4 g + e: the item of accumulator register 4 g + e passed
#pragma unroll
          for (int r = 0; r < 16; ++r) mask |= (acc[r] >= thr[c] ? 1u : 0u) << r;
          if (mask) {
            const int ul = (wave * UC + c) * 32 + col;
            int slot = atomicAdd(&lcnt[ul], __popc(mask));   // the user's two lane halves share the counter
            const int off0 = (tile - tile_begin) * 32 + 4 * half;
            do {
              const int idx = __ffs(mask) - 1;
              mask &= mask - 1u;
              const int off = off0 + 8 * (idx >> 2) + (idx & 3);
              if (slot < LCAP) {
                llist[ul * LCAP + slot] = (unsigned short)off;
              } else {   // the chunk's LDS list is full (a cluster of near-equal items, e.g. 100 identical rows): straight
                         // to the user's global list -- slow, rare, and no longer a reason to redo the whole call
                const int g = atomicAdd(p.cand_count + user[c], 1);
                if (g < TK_CAND_CAP) p.cand[(size_t)user[c] * TK_CAND_CAP + g] = tile_begin * 32 + off;
              }
              ++slot;
            } while (mask);
          }
        }
      }
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (no DMA is pending here: the last stage issues none; tests/test_isa_checks.py)
  __syncthreads();
  for (int ul = tid; ul < USERS; ul += 512) {  // thread = one user of the workgroup: reserve room in the global list once, copy
    const int gu = who.group * USERS + ul;
    const int n = lcnt[ul];
    if (gu < p.B && n > 0) {
      // (entries past LCAP went to the global list when they were found; a global count past TK_CAND_CAP makes
      // cand_rescore_kernel raise the flag that hands the call to the exact pass)
      const int m = min(n, LCAP);
      const int base = atomicAdd(p.cand_count + gu, m);
      for (int i = 0; i < m; ++i)
        if (base + i < TK_CAND_CAP) p.cand[(size_t)gu * TK_CAND_CAP + base + i] = tile_begin * 32 + (int)llist[ul * LCAP + i];
    }
  }
}

// 512 users per workgroup (two column tiles of 32 users per wave). Four column tiles (every fragment and bias read from
// LDS feeding 16 MFMAs instead of 8) were built and measured in round 4: bq alone is then 64 of the 128 VGPRs that two
// workgroups per CU leave a wave, hipcc spills 9-34 registers, and the passes ran 431 / 616 us at 1 M items against
// 449 / 566 us, 83 / 104 us against 53 / 84 us on Beauty. The kernels keep the template parameter.
int lr_bf16_users_per_wg(int B) { (void)B; return 512; }

int lr_launch_item_bound(const BoundParams& p0, int chunks, hipStream_t st) {
  BoundParams p = p0;
  // a chunk is whole float4 iterations (one maximum per tile) or whole QUADS of tile groups (grouped: four groups' maxima leave
  // as one 16-byte store)
  if (p.gshift < 0 || p.gshift == 1 || p.tiles_per_chunk % (p.gshift >= 2 ? (4 << p.gshift) : 4) != 0)
    LR_FAIL(LR_EINVAL, "item_bound_kernel: gshift=%d tiles_per_chunk=%d", p.gshift, p.tiles_per_chunk);
  const int upw = lr_bf16_users_per_wg(p.B);
  p.n_chunks = chunks;
  p.n_user_groups = (p.B + upw - 1) / upw;
  const dim3 grid(tk_grid(p.n_chunks, p.n_user_groups));
  if (p.gshift == 0) hipLaunchKernelGGL((item_bound_kernel<false, 2>), grid, dim3(512), 0, st, p);
  else hipLaunchKernelGGL((item_bound_kernel<true, 2>), grid, dim3(512), 0, st, p);
  LR_CHECK_LAUNCH("item_bound_kernel");
  return LR_OK;
}

int lr_launch_item_cand(const CandParams& p0, int chunks, hipStream_t st) {
  CandParams p = p0;
  const int upw = lr_bf16_users_per_wg(p.B);
  if (p.tiles_per_chunk > lr_bf16_max_chunk_tiles(p.B))
    LR_FAIL(LR_EINVAL, "item_cand_kernel: %d tiles per chunk, at most %d", p.tiles_per_chunk, lr_bf16_max_chunk_tiles(p.B));
  p.n_chunks = chunks;
  p.n_user_groups = (p.B + upw - 1) / upw;
  const dim3 grid(tk_grid(p.n_chunks, p.n_user_groups));
  hipLaunchKernelGGL((item_cand_kernel<2>), grid, dim3(512), 0, st, p);
  LR_CHECK_LAUNCH("item_cand_kernel");
  return LR_OK;
}
