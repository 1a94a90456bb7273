// Prototype: the product GEMM's two-wave ping-pong K loop (llamarec_amd/csrc/llama_gemm.hip,
// gemm256rb_kernel) with the tile's ROW count as a template parameter -- 256 (the product's tile, as the yardstick on the same
// box) and 128 (the tile the review asked for, for the online single-user path: M = 460 -> 4 row tiles instead of 2). Same wave
// grid (2 x 4), same regions and recycling order of the LDS-DMA, same LDS image and MFMA; a 128-row tile has half the A
// pieces (one per wave and region instead of two) and half the MFMAs per phase. Plain bf16 store; s_memtime around the K loop.
//   C[M][N] = A[M][K] B[N][K]^T; M a multiple of BM, N of 256, K of 64.
// A second kernel, gemm_bm_rs_kernel, stages the operand stream through REGISTERS instead of LDS-DMA (global_load_dwordx4 at the
// phase where the product issues a region's DMA, ds_write_b128 one K tile later at the same phase; both as inline asm with
// hand-counted vmcnt: through plain C++ loads hipcc waited for every phase's fresh loads in front of that phase's barrier,
// 11-26 k cycles per K tile). Result at 128 rows: 1 832 cycles per K tile against 1 856 with LDS-DMA -- the form of the request is
// not what the LOAD phases wait for. (At 256 rows the variant spills 34 registers: timings meaningless.)
// hipcc --offload-arch=gfx950 -O3 -o /tmp/gemm_bm tools/diag/gemm_bm.hip && /tmp/gemm_bm
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

typedef unsigned short u16;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float floatx4 __attribute__((ext_vector_type(4)));
#define STAGE_BYTES 65536
#ifndef GB_SADDR
#define GB_SADDR 0   // 1: LDS-DMA sources as scalar base + 32-bit lane offset (first kernel only)
#endif
#ifndef GB_ABL
#define GB_ABL 0   // ablations of the LDS-DMA kernel (wrong results, timings only): 1 = the B operand neither streamed nor re-read after the first K tile, 2 = the same for A, 5 = 1 + the B operand fetched by plain loads into registers (what a direct global -> register B path would cost)
#endif

__device__ __forceinline__ void glds16(const void* gsrc, void* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                   (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}
// the same DMA with the source as scalar base + 32-bit lane offset (hipcc does not select this form for the builtin: it adds the
// uniform part to a per-lane 64-bit pointer on the vector ALU in front of every issue); as inline asm it is invisible to hipcc's
// vmcnt bookkeeping, which this loop does by hand anyway
__device__ __forceinline__ void glds16_s(const void* base, unsigned voff, void* lds_wave_base) {
  const unsigned m0v = __builtin_amdgcn_readfirstlane((unsigned)(size_t)((__attribute__((address_space(3))) const char*)lds_wave_base));
  const unsigned long long b = (unsigned long long)base;
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)b), hi = __builtin_amdgcn_readfirstlane((unsigned)(b >> 32));
  const unsigned long long bs = ((unsigned long long)hi << 32) | lo;
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(m0v), "v"(voff), "s"(bs) : "memory");
}
__device__ __forceinline__ u16 f2bf(float f) { return __builtin_bit_cast(u16, (__bf16)f); }
#define WAIT_VM(n) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(n) : "memory")
#define GB_SYNC() __builtin_amdgcn_s_barrier()
#define BARRIER()                           \
  do {                                      \
    __builtin_amdgcn_sched_barrier(0);      \
    GB_SYNC();                              \
    __builtin_amdgcn_sched_barrier(0);      \
  } while (0)

template <int BM>
__global__ __launch_bounds__(512) void gemm_bm_kernel(const u16* __restrict__ A, const u16* __restrict__ B, u16* __restrict__ C,
                                                      int M, int N, int K, unsigned long long* stamps) {
  constexpr int AP = BM / 128;      // A pieces per wave and region (2 / 1)
  constexpr int MQ = BM / 64;       // 16-row blocks per quadrant (4 / 2)
  constexpr int HALF = BM / 2;      // rows of a wave group
  // "all but my newest n" at phases 0, 2 / 1, 3: the vector-memory operations of the last five phases may be outstanding
  // (A phases issue AP of them, B phases 2; 10, 10 for the product's tile). The ablations keep that set of PHASES:
  // a dropped operand's phases issue nothing, GB_ABL & 4 issues 4 plain loads in each B phase.
  constexpr int OPA = (GB_ABL & 2) ? 0 : AP, OPB = (GB_ABL & 4) ? 4 : ((GB_ABL & 1) ? 0 : 2);
  constexpr int WA = 3 * OPA + 2 * OPB, WB = 3 * OPB + 2 * OPA;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;
  const int nkt = K >> 6;
  const int tilesM = M / BM, tilesN = N >> 8, nwg = tilesM * tilesN;
  int id;
  {
    const int bid = blockIdx.x, q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
    id = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  int tm, tn;
  {
    const int group_m = 8, per_group = group_m * tilesN;
    const int g = id / per_group, rem = id % per_group, first_m = g * group_m;
    const int gsz = min(group_m, tilesM - first_m);
    tm = first_m + rem % gsz;
    tn = rem / gsz;
  }
  const int m0 = tm * BM, n0 = tn << 8;

  // DMA pieces of this wave (8 rows x 128 B): region 0 = A rows of quadrant row-half 0, 1 = B rows nh0, 2 = B rows nh1, 3 = A rows of
  // row-half 1. A: AP pieces per wave (q = AP wave + j: group q / (8 AP / 2) ..), B: 2 per wave
  const int srow = lane >> 3, spos = lane & 7;
  const char* src[4][2];
  unsigned soff[4][2];   // GB_SADDR: the same source as a 32-bit lane offset from the tile's (wave-uniform) A / B base
  int ldsoff[4][2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int qa = AP * wave + (j < AP ? j : 0);                 // 0 .. 8 AP - 1
    const int ra = (qa / (4 * AP)) * HALF + (qa % (4 * AP)) * 8;  // row-half 0 rows of wave group qa / (4 AP)
    const int qb = 2 * wave + j;
    const int rb = (qb >> 2) * 64 + (qb & 3) * 8;
    const int rows[4] = {ra, rb, rb + 32, ra + HALF / 2};
    const bool isA[4] = {true, false, false, true};
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int row = rows[t] + srow;
      const int chunk = spos ^ ((row >> 1) & 7);
      if (isA[t]) {
        src[t][j] = reinterpret_cast<const char*>(A) + ((size_t)(m0 + row) * K) * 2 + chunk * 16;
        soff[t][j] = (unsigned)(row * K * 2 + chunk * 16);
        ldsoff[t][j] = rows[t] * 128;
      } else {
        src[t][j] = reinterpret_cast<const char*>(B) + ((size_t)(n0 + row) * K) * 2 + chunk * 16;
        soff[t][j] = (unsigned)(row * K * 2 + chunk * 16);
        ldsoff[t][j] = 32768 + rows[t] * 128;
      }
    }
  }
  const char* const a_tile = reinterpret_cast<const char*>(A) + (size_t)m0 * K * 2;   // wave-uniform bases (SGPR pairs)
  const char* const b_tile = reinterpret_cast<const char*>(B) + (size_t)n0 * K * 2;
#if GB_SADDR
  // source = uniform base (+ the K tile's 128 bytes, scalar arithmetic) + a 32-bit lane offset: the saddr form of the DMA, no
  // 64-bit vector adds per issue (the per-lane pointer form costs two v_lshl_add_u64 in front of every global_load_lds)
#define RB_DMA(reg, tile)                                                                         \
  do {                                                                                            \
    char* dst_ = smem + ((tile)&1) * STAGE_BYTES;                                                 \
    const char* base_ = (((reg) == 1 || (reg) == 2) ? b_tile : a_tile) + (size_t)(tile)*128;      \
    glds16_s(base_, soff[(reg)][0], dst_ + ldsoff[(reg)][0]);                                     \
    if (((reg) == 1 || (reg) == 2) || AP == 2) glds16_s(base_, soff[(reg)][1], dst_ + ldsoff[(reg)][1]); \
  } while (0)
#else
#define RB_DMA(reg, tile)                                                                         \
  do {                                                                                            \
    char* dst_ = smem + ((tile)&1) * STAGE_BYTES;                                                 \
    glds16(src[(reg)][0] + (size_t)(tile)*128, dst_ + ldsoff[(reg)][0]);                          \
    if (((reg) == 1 || (reg) == 2) || AP == 2) glds16(src[(reg)][1] + (size_t)(tile)*128, dst_ + ldsoff[(reg)][1]); \
  } while (0)
#endif

  const int frow = lane & 15, fsw = frow >> 1;
  const int fo0 = frow * 128 + (((lane >> 4) ^ fsw) << 4);
  const int fo1 = frow * 128 + (((4 + (lane >> 4)) ^ fsw) << 4);
  const int a_base = wm * HALF * 128;
  const int b_base = 32768 + wn * 64 * 128;

  floatx4 acc[2 * MQ][4];
#pragma unroll
  for (int i = 0; i < 2 * MQ; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = floatx4{0.f, 0.f, 0.f, 0.f};
  bf16x8 afr[2 * MQ], b0x[4], b0y[4], b1[4];

#define RB_LOAD_A(buf, mh)                                                                                   \
  _Pragma("unroll") for (int mt = 0; mt < MQ; ++mt) {                                                        \
    afr[mt] = *reinterpret_cast<const bf16x8*>((buf) + a_base + ((mh)*(HALF / 2) + mt * 16) * 128 + fo0);      \
    afr[MQ + mt] = *reinterpret_cast<const bf16x8*>((buf) + a_base + ((mh)*(HALF / 2) + mt * 16) * 128 + fo1); \
  }
#define RB_LOAD_B(dst, buf, nh)                                                                       \
  _Pragma("unroll") for (int nt = 0; nt < 2; ++nt) {                                                  \
    dst[nt] = *reinterpret_cast<const bf16x8*>((buf) + b_base + ((nh)*32 + nt * 16) * 128 + fo0);     \
    dst[2 + nt] = *reinterpret_cast<const bf16x8*>((buf) + b_base + ((nh)*32 + nt * 16) * 128 + fo1); \
  }
#define RB_MFMA(bfrag, mh, nh)                                                                        \
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                  \
  __builtin_amdgcn_sched_barrier(0);                                                                  \
  _Pragma("unroll") for (int ks = 0; ks < 2; ++ks)                                                    \
  _Pragma("unroll") for (int mt = 0; mt < MQ; ++mt)                                                   \
  _Pragma("unroll") for (int nt = 0; nt < 2; ++nt)                                                    \
    acc[(mh)*MQ + mt][(nh)*2 + nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(                         \
        bfrag[ks * 2 + nt], afr[ks * MQ + mt], acc[(mh)*MQ + mt][(nh)*2 + nt], 0, 0, 0);
  // GB_ABL & 4 (with & 1): what a B operand taken straight from global memory would cost the loop -- per K tile and wave 8
  // coalesced 1 KB loads (its 64 columns x 64 k, as if the weights were stored in fragment order; both row halves of the
  // tile read the same bytes), issued where the B regions' DMAs were, into scratch registers that nothing reads.
  const char* bdir = reinterpret_cast<const char*>(B) + (size_t)n0 * K * 2 + wn * 8192 + lane * 16;
  floatx4 bscr[4] = {floatx4{0.f, 0.f, 0.f, 0.f}, floatx4{0.f, 0.f, 0.f, 0.f}, floatx4{0.f, 0.f, 0.f, 0.f}, floatx4{0.f, 0.f, 0.f, 0.f}};   // "+v" below and the use after the loop keep these registers reserved while loads are in flight
#define RB_BDIRECT(tile, half)                                                                        \
  _Pragma("unroll") for (int i_ = 0; i_ < 4; ++i_)                                                    \
    asm volatile("global_load_dwordx4 %0, %1, off" : "+v"(bscr[i_]) : "v"(bdir + (size_t)(tile)*32768 + ((half)*4 + i_) * 1024) : "memory");
#define RB_TILE(kt, b0cur, b0nxt)                                                                     \
  {                                                                                                   \
    const char* cur = smem + ((kt)&1) * STAGE_BYTES;                                                   \
    const char* nxt = smem + (((kt) + 1) & 1) * STAGE_BYTES;                                           \
    const bool more = (kt) + 1 < nkt;                                                                 \
    const bool more2 = (kt) + 2 < nkt;                                                                \
    if (!(GB_ABL & 2) || (kt) == 0) { RB_LOAD_A(cur, 0) }                                             \
    if (more && !(GB_ABL & 2)) RB_DMA(3, (kt) + 1);                                                   \
    if (more2) { WAIT_VM(WA); } else { WAIT_VM(0); }                                                  \
    BARRIER();                                                                                        \
    RB_MFMA(b0cur, 0, 0)                                                                              \
    BARRIER();                                                                                        \
    if (!(GB_ABL & 1) || (kt) == 0) { RB_LOAD_B(b1, cur, 1) }                                         \
    if (more2 && !(GB_ABL & 1)) RB_DMA(1, (kt) + 2);                                                  \
    if (more2 && (GB_ABL & 4)) { RB_BDIRECT((kt) + 2, 0) }                                             \
    if (more2) { WAIT_VM(WB); } else { WAIT_VM(0); }                                                  \
    BARRIER();                                                                                        \
    RB_MFMA(b1, 0, 1)                                                                                 \
    BARRIER();                                                                                        \
    if (!(GB_ABL & 2) || (kt) == 0) { RB_LOAD_A(cur, 1) }                                             \
    if (more2 && !(GB_ABL & 2)) RB_DMA(0, (kt) + 2);                                                  \
    if (more2) { WAIT_VM(WA); } else { WAIT_VM(0); }                                                  \
    BARRIER();                                                                                        \
    RB_MFMA(b1, 1, 1)                                                                                 \
    BARRIER();                                                                                        \
    if (more && !(GB_ABL & 1)) { RB_LOAD_B(b0nxt, nxt, 0) }                                           \
    if (more2 && !(GB_ABL & 1)) RB_DMA(2, (kt) + 2);                                                  \
    if (more2 && (GB_ABL & 4)) { RB_BDIRECT((kt) + 2, 1) }                                             \
    if (more2) { WAIT_VM(WB); } else { WAIT_VM(0); }                                                  \
    BARRIER();                                                                                        \
    RB_MFMA(b0cur, 1, 0)                                                                              \
    BARRIER();                                                                                        \
  }

#pragma unroll
  for (int reg = 0; reg < 4; ++reg) RB_DMA(reg, 0);
  if (nkt > 1) {
    RB_DMA(1, 1);
    RB_DMA(0, 1);
    RB_DMA(2, 1);
    WAIT_VM(4 + AP);
  } else {
    WAIT_VM(0);
  }
  BARRIER();
  RB_LOAD_B(b0x, smem, 0)
  unsigned long long t_begin = 0, t_end = 0;
  if (stamps) asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_begin)::"memory");
  if (wm == 1) {
    __builtin_amdgcn_s_setprio(1);
    BARRIER();   // group 1 runs one barrier behind group 0
  }
  for (int kt = 0; kt < nkt; kt += 2) {
    RB_TILE(kt, b0x, b0y)
    if (kt + 1 < nkt) RB_TILE(kt + 1, b0y, b0x)
  }
  if (GB_ABL & 4) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int i_ = 0; i_ < 4; ++i_) asm volatile("" ::"v"(bscr[i_]));
  }
  if (wm == 0) BARRIER();
  if (stamps) {
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_end)::"memory");
    if (lane == 0) stamps[blockIdx.x * 8 + wave] = t_end - t_begin;
  }
  const int quad = lane >> 4;
#pragma unroll
  for (int mt = 0; mt < 2 * MQ; ++mt) {
    const int row = m0 + wm * HALF + mt * 16 + (lane & 15);
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
      const int col = n0 + wn * 64 + nt * 16 + 4 * quad;
      const unsigned lo = (unsigned)f2bf(acc[mt][nt][0]) | ((unsigned)f2bf(acc[mt][nt][1]) << 16);
      const unsigned hi = (unsigned)f2bf(acc[mt][nt][2]) | ((unsigned)f2bf(acc[mt][nt][3]) << 16);
      *reinterpret_cast<uint2*>(C + (size_t)row * N + col) = make_uint2(lo, hi);
    }
  }
}

template <int BM>
__global__ __launch_bounds__(512) void gemm_bm_rs_kernel(const u16* __restrict__ A, const u16* __restrict__ B, u16* __restrict__ C,
                                                      int M, int N, int K, unsigned long long* stamps) {
  constexpr int AP = BM / 128;      // A pieces per wave and region (2 / 1)
  constexpr int MQ = BM / 64;       // 16-row blocks per quadrant (4 / 2)
  constexpr int HALF = BM / 2;      // rows of a wave group
  constexpr int WA = 3 * AP + 4, WB = 6 + 2 * AP;   // "all but my newest n" at phases 0, 2 / 1, 3 (10, 10 for the product's tile)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;
  const int nkt = K >> 6;
  const int tilesM = M / BM, tilesN = N >> 8, nwg = tilesM * tilesN;
  int id;
  {
    const int bid = blockIdx.x, q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
    id = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  int tm, tn;
  {
    const int group_m = 8, per_group = group_m * tilesN;
    const int g = id / per_group, rem = id % per_group, first_m = g * group_m;
    const int gsz = min(group_m, tilesM - first_m);
    tm = first_m + rem % gsz;
    tn = rem / gsz;
  }
  const int m0 = tm * BM, n0 = tn << 8;

  // DMA pieces of this wave (8 rows x 128 B): region 0 = A rows of quadrant row-half 0, 1 = B rows nh0, 2 = B rows nh1, 3 = A rows of
  // row-half 1. A: AP pieces per wave (q = AP wave + j: group q / (8 AP / 2) ..), B: 2 per wave
  const int srow = lane >> 3, spos = lane & 7;
  const char* src[4][2];
  int ldsoff[4][2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int qa = AP * wave + (j < AP ? j : 0);                 // 0 .. 8 AP - 1
    const int ra = (qa / (4 * AP)) * HALF + (qa % (4 * AP)) * 8;  // row-half 0 rows of wave group qa / (4 AP)
    const int qb = 2 * wave + j;
    const int rb = (qb >> 2) * 64 + (qb & 3) * 8;
    const int rows[4] = {ra, rb, rb + 32, ra + HALF / 2};
    const bool isA[4] = {true, false, false, true};
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int row = rows[t] + srow;
      const int chunk = spos ^ ((row >> 1) & 7);
      if (isA[t]) {
        src[t][j] = reinterpret_cast<const char*>(A) + ((size_t)(m0 + row) * K) * 2 + chunk * 16;
        ldsoff[t][j] = rows[t] * 128;
      } else {
        src[t][j] = reinterpret_cast<const char*>(B) + ((size_t)(n0 + row) * K) * 2 + chunk * 16;
        ldsoff[t][j] = 32768 + rows[t] * 128;
      }
    }
  }
  // register-staged stream: a region's pieces are requested with plain loads at the phase where the product issues their DMA and
  // written to LDS one K tile later at the same phase (two phases before their first reader); hipcc tracks these loads' vmcnt
  floatx4 stg[4][2];   // (the kernel has no static LDS: the dynamic array starts at LDS address 0, which the asm stores use)
#define RS_LOAD(reg, tile)   /* inline asm: hipcc must not know these loads (it would wait for them in front of the next barrier) */ \
  do {                                                                                                \
    asm volatile("global_load_dwordx4 %0, %1, off" : "=&v"(stg[(reg)][0]) : "v"(src[(reg)][0] + (size_t)(tile)*128) : "memory"); \
    if (((reg) == 1 || (reg) == 2) || AP == 2)                                                        \
      asm volatile("global_load_dwordx4 %0, %1, off" : "=&v"(stg[(reg)][1]) : "v"(src[(reg)][1] + (size_t)(tile)*128) : "memory"); \
  } while (0)
// the pieces requested four phases ago: all but the (pieces of the last three phases) newest loads have landed
#define RS_WRITE(reg, tile)                                                                           \
  do {                                                                                                \
    const unsigned dst_ = (unsigned)(((tile)&1) * STAGE_BYTES + lane * 16);                           \
    if ((reg) == 1 || (reg) == 2) { WAIT_VM(2 * AP + 2); } else { WAIT_VM(AP + 4); }                  \
    asm volatile("ds_write_b128 %0, %1" ::"v"(dst_ + (unsigned)ldsoff[(reg)][0]), "v"(stg[(reg)][0]) : "memory"); \
    if (((reg) == 1 || (reg) == 2) || AP == 2)                                                        \
      asm volatile("ds_write_b128 %0, %1" ::"v"(dst_ + (unsigned)ldsoff[(reg)][1]), "v"(stg[(reg)][1]) : "memory"); \
  } while (0)
#define RB_DMA(reg, tile)                                                                         \
  do {                                                                                            \
    char* dst_ = smem + ((tile)&1) * STAGE_BYTES;                                                 \
    glds16(src[(reg)][0] + (size_t)(tile)*128, dst_ + ldsoff[(reg)][0]);                          \
    if (((reg) == 1 || (reg) == 2) || AP == 2) glds16(src[(reg)][1] + (size_t)(tile)*128, dst_ + ldsoff[(reg)][1]); \
  } while (0)

  const int frow = lane & 15, fsw = frow >> 1;
  const int fo0 = frow * 128 + (((lane >> 4) ^ fsw) << 4);
  const int fo1 = frow * 128 + (((4 + (lane >> 4)) ^ fsw) << 4);
  const int a_base = wm * HALF * 128;
  const int b_base = 32768 + wn * 64 * 128;

  floatx4 acc[2 * MQ][4];
#pragma unroll
  for (int i = 0; i < 2 * MQ; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = floatx4{0.f, 0.f, 0.f, 0.f};
  bf16x8 afr[2 * MQ], b0x[4], b0y[4], b1[4];

#define RB_LOAD_A(buf, mh)                                                                                   \
  _Pragma("unroll") for (int mt = 0; mt < MQ; ++mt) {                                                        \
    afr[mt] = *reinterpret_cast<const bf16x8*>((buf) + a_base + ((mh)*(HALF / 2) + mt * 16) * 128 + fo0);      \
    afr[MQ + mt] = *reinterpret_cast<const bf16x8*>((buf) + a_base + ((mh)*(HALF / 2) + mt * 16) * 128 + fo1); \
  }
#define RB_LOAD_B(dst, buf, nh)                                                                       \
  _Pragma("unroll") for (int nt = 0; nt < 2; ++nt) {                                                  \
    dst[nt] = *reinterpret_cast<const bf16x8*>((buf) + b_base + ((nh)*32 + nt * 16) * 128 + fo0);     \
    dst[2 + nt] = *reinterpret_cast<const bf16x8*>((buf) + b_base + ((nh)*32 + nt * 16) * 128 + fo1); \
  }
#define RB_MFMA(bfrag, mh, nh)                                                                        \
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                  \
  __builtin_amdgcn_sched_barrier(0);                                                                  \
  _Pragma("unroll") for (int ks = 0; ks < 2; ++ks)                                                    \
  _Pragma("unroll") for (int mt = 0; mt < MQ; ++mt)                                                   \
  _Pragma("unroll") for (int nt = 0; nt < 2; ++nt)                                                    \
    acc[(mh)*MQ + mt][(nh)*2 + nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(                         \
        bfrag[ks * 2 + nt], afr[ks * MQ + mt], acc[(mh)*MQ + mt][(nh)*2 + nt], 0, 0, 0);
#define RB_TILE(kt, b0cur, b0nxt)                                                                     \
  {                                                                                                   \
    const char* cur = smem + ((kt)&1) * STAGE_BYTES;                                                   \
    const char* nxt = smem + (((kt) + 1) & 1) * STAGE_BYTES;                                           \
    const bool more = (kt) + 1 < nkt;                                                                 \
    const bool more2 = (kt) + 2 < nkt;                                                                \
    const bool first = (kt) == 0;                                                                     \
    RB_LOAD_A(cur, 0)                                                                                 \
    if (!first) RS_WRITE(3, (kt));                                                                    \
    RS_LOAD(3, min((kt) + 1, nkt - 1));   /* unconditional (clamped): a conditional load voids hipcc's vmcnt bookkeeping */ \
    BARRIER();                                                                                        \
    RB_MFMA(b0cur, 0, 0)                                                                              \
    BARRIER();                                                                                        \
    RB_LOAD_B(b1, cur, 1)                                                                             \
    if (!first && more) RS_WRITE(1, (kt) + 1);                                                        \
    RS_LOAD(1, min((kt) + 2, nkt - 1));   /* unconditional (clamped): a conditional load voids hipcc's vmcnt bookkeeping */                                                                  \
    BARRIER();                                                                                        \
    RB_MFMA(b1, 0, 1)                                                                                 \
    BARRIER();                                                                                        \
    RB_LOAD_A(cur, 1)                                                                                 \
    if (!first && more) RS_WRITE(0, (kt) + 1);                                                        \
    RS_LOAD(0, min((kt) + 2, nkt - 1));   /* unconditional (clamped): a conditional load voids hipcc's vmcnt bookkeeping */                                                                  \
    BARRIER();                                                                                        \
    RB_MFMA(b1, 1, 1)                                                                                 \
    BARRIER();                                                                                        \
    if (more) { RB_LOAD_B(b0nxt, nxt, 0) }                                                            \
    if (!first && more) RS_WRITE(2, (kt) + 1);                                                        \
    RS_LOAD(2, min((kt) + 2, nkt - 1));   /* unconditional (clamped): a conditional load voids hipcc's vmcnt bookkeeping */                                                                  \
    BARRIER();                                                                                        \
    RB_MFMA(b0cur, 1, 0)                                                                              \
    BARRIER();                                                                                        \
  }

#pragma unroll
  for (int reg = 0; reg < 4; ++reg) RB_DMA(reg, 0);
  if (nkt > 1) {
    RB_DMA(1, 1);
    RB_DMA(0, 1);
    RB_DMA(2, 1);
    WAIT_VM(4 + AP);
  } else {
    WAIT_VM(0);
  }
  BARRIER();
  RB_LOAD_B(b0x, smem, 0)
  unsigned long long t_begin = 0, t_end = 0;
  if (stamps) asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_begin)::"memory");
  if (wm == 1) {
    __builtin_amdgcn_s_setprio(1);
    BARRIER();   // group 1 runs one barrier behind group 0
  }
  for (int kt = 0; kt < nkt; kt += 2) {
    RB_TILE(kt, b0x, b0y)
    if (kt + 1 < nkt) RB_TILE(kt + 1, b0y, b0x)
  }
  if (wm == 0) BARRIER();
  if (stamps) {
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_end)::"memory");
    if (lane == 0) stamps[blockIdx.x * 8 + wave] = t_end - t_begin;
  }
  const int quad = lane >> 4;
#pragma unroll
  for (int mt = 0; mt < 2 * MQ; ++mt) {
    const int row = m0 + wm * HALF + mt * 16 + (lane & 15);
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
      const int col = n0 + wn * 64 + nt * 16 + 4 * quad;
      const unsigned lo = (unsigned)f2bf(acc[mt][nt][0]) | ((unsigned)f2bf(acc[mt][nt][1]) << 16);
      const unsigned hi = (unsigned)f2bf(acc[mt][nt][2]) | ((unsigned)f2bf(acc[mt][nt][3]) << 16);
      *reinterpret_cast<uint2*>(C + (size_t)row * N + col) = make_uint2(lo, hi);
    }
  }
}

static float bf2f(u16 b) {
  unsigned u = (unsigned)b << 16;
  float f;
  memcpy(&f, &u, 4);
  return f;
}
static u16 f2bf_host(float f) {
  unsigned u;
  memcpy(&u, &f, 4);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (u16)(u >> 16);
}

template <int BM, bool RS>
static void run(int M, int N, int K) {
  auto kernel = RS ? gemm_bm_rs_kernel<BM> : gemm_bm_kernel<BM>;
  std::vector<u16> hA((size_t)M * K), hB((size_t)N * K);
  unsigned st = 12345u;
  auto rnd = [&]() { st = st * 1664525u + 1013904223u; return ((st >> 8) & 0xffff) / 65536.0f - 0.5f; };
  for (auto& v : hA) v = f2bf_host(rnd() * 2.0f);
  for (auto& v : hB) v = f2bf_host(rnd() * 0.04f);
  u16 *dA, *dB, *dC;
  unsigned long long* dS;
  hipMalloc(&dA, hA.size() * 2); hipMalloc(&dB, hB.size() * 2); hipMalloc(&dC, (size_t)M * N * 2);
  const int nwg = (M / BM) * (N / 256);
  hipMalloc(&dS, (size_t)nwg * 8 * 8);
  hipMemcpy(dA, hA.data(), hA.size() * 2, hipMemcpyHostToDevice);
  hipMemcpy(dB, hB.data(), hB.size() * 2, hipMemcpyHostToDevice);
  hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * STAGE_BYTES);
  for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(kernel, dim3(nwg), dim3(512), 2 * STAGE_BYTES, 0, dA, dB, dC, M, N, K, (unsigned long long*)nullptr);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0);
  const int reps = 10;
  for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(kernel, dim3(nwg), dim3(512), 2 * STAGE_BYTES, 0, dA, dB, dC, M, N, K, (unsigned long long*)nullptr);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0.f;
  hipEventElapsedTime(&ms, e0, e1);
  ms /= reps;
  hipLaunchKernelGGL(kernel, dim3(nwg), dim3(512), 2 * STAGE_BYTES, 0, dA, dB, dC, M, N, K, dS);
  hipDeviceSynchronize();
  hipError_t err = hipGetLastError();
  std::vector<unsigned long long> hs((size_t)nwg * 8);
  hipMemcpy(hs.data(), dS, hs.size() * 8, hipMemcpyDeviceToHost);
  std::sort(hs.begin(), hs.end());
  std::vector<u16> hC((size_t)M * N);
  hipMemcpy(hC.data(), dC, hC.size() * 2, hipMemcpyDeviceToHost);
  double worst = 0.0;
  for (int t = 0; t < 64; ++t) {
    const int m = (int)(((long long)t * 7919 + 13) % M), n = (int)(((long long)t * 104729 + 7) % N);
    double ref = 0.0;
    for (int k = 0; k < K; ++k) ref += (double)bf2f(hA[(size_t)m * K + k]) * bf2f(hB[(size_t)n * K + k]);
    worst = std::max(worst, std::fabs((double)bf2f(hC[(size_t)m * N + n]) - ref) / (std::fabs(ref) + 1e-3));
  }
  printf("%s BM=%3d M=%5d N=%5d K=%5d: %4d workgroups, %7.1f us = %5.0f TF/s; K loop %5.0f cycles per K tile (median wave; MFMA-bound = %d); max rel err %.4f; %s\n",
         RS ? "staged" : "LDS-DMA", BM, M, N, K, nwg, ms * 1e3, 2.0 * M * N * K / (ms * 1e-3) / 1e12, (double)hs[hs.size() / 2] / (K / 64), BM * 8, worst, hipGetErrorString(err));
  hipFree(dA); hipFree(dB); hipFree(dC); hipFree(dS);
}

int main(int argc, char** argv) {
  if (argc > 1 && argv[1][0] == 'q') {   // "quick": the product's tile on the two batch shapes only (the -DGB_ABL arms)
    for (int rep = 0; rep < 2; ++rep) {
      run<256, false>(32768, 4096, 4096);
      run<256, false>(32768, 4096, 11008);
    }
    return 0;
  }
  // the batch shape (many rounds of tiles): per-K-tile cost of the two tiles at full occupancy, LDS-DMA against register staging
  run<256, false>(32768, 4096, 4096);
  run<256, true>(32768, 4096, 4096);
  run<128, false>(32768, 4096, 4096);
  run<128, true>(32768, 4096, 4096);
  run<256, false>(32768, 4096, 11008);
  run<256, true>(32768, 4096, 11008);
  run<128, false>(32768, 4096, 11008);
  run<128, true>(32768, 4096, 11008);
  // the online shape, M = 460 padded to 512: gate-up (unsplit in latency mode), qkv, o, down as plain products
  run<256, false>(512, 22016, 4096);
  run<128, false>(512, 22016, 4096);
  run<128, true>(512, 22016, 4096);
  run<256, false>(512, 12288, 4096);
  run<128, false>(512, 12288, 4096);
  run<256, false>(512, 4096, 4096);
  run<128, false>(512, 4096, 4096);
  run<256, false>(512, 4096, 11008);
  run<128, false>(512, 4096, 11008);
  return 0;
}
