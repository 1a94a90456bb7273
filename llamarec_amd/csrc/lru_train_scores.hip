// lru_train_scores.hip -- item GEMM + softmax cross-entropy of the retriever's training step with STORED logits
// (trainer/lru.py:22-27, model/lru.py:85), the form used while [rows x (V+1)] fits the Infinity Cache
// (TR_MATERIALISE_ELEMS in lru_train.hip; beyond that lru_train_ce.hip recomputes tiles and never stores them).
//
// First form: a generic 64 x 64 GEMM launch per product with a row-softmax kernel in between -- 106 + 99 + 90 + 92 us of a
// 0.68 ms Beauty step (profiles/r04_train_beauty_row_panels.txt), the logits streamed five times. The three products are
// 4.95 GFLOP each = 31.5 us at the fp32 MFMA peak. Here:
//   ts_scores_kernel   a wave holds a 16-row panel of x in operand registers, the table streams through as the other MFMA
//                      operand; writes the logits and, per (row, item split), the running (max, sum) -- no softmax pass
//   ts_combine_kernel  per row: lse from the partials, the loss term, lse = +inf / label = -1 for unlabelled rows, and the
//                      one-hot half of dl = softmax / n - onehot / n as rank-1 updates of d x, d E and d bias
//   ts_dx_kernel       d x = dl E: 16-row panels per wave, dl = softmax / n formed in the operand registers from the stored
//                      logits (one v_exp_f32 per element), E^T streamed; item splits add with atomics
//   ts_de_kernel       d E = dl^T x, d bias: 64-item panels x row parts, rows streamed, the four waves of a workgroup split
//                      their rows and meet in LDS; the row parts add with atomics (whole 256-byte rows per instruction)
// v_mfma_f32_16x16x4_f32 with the K index permuted inside groups of 16 (see lru_train_blocks.hip): both operands of a
// product are read as float4 along their contiguous axis. In ts_de_kernel the ITEM index is permuted the same way (a lane
// reads 4 consecutive items of a row; element s of the float4 belongs to sub-panel s).
//
// Memory shapes (the second version of this file ran at 86-105 cycles per MFMA against 37 for the same loop on registers,
// tools/diag/mfma_f32_issue.hip, whatever the streams hit -- L1, L2 or HBM): an MFMA operand fragment is 16 rows x 64 B per
// wave-instruction when read from a row-major matrix, and a CU moves that shape at 13-15 B/clk (a vector-memory
// instruction is priced per cache line it touches, tools/diag/store_rate.hip) -- 80 KB per 64 MFMAs and workgroup. So
//   * the streamed operands (E for the scores, E^T for d x, x for d E) are re-laid once per pass in FRAGMENT ORDER by
//     ts_fragments_kernel: the float4 of the 64 lanes of one operand load are 1 KB contiguous;
//   * the logits stay row-major (d E reads them 4 rows x 256 B per instruction, whole lines) and the two kernels whose
//     lanes own rows move their 16 x 64 tile through a wave-private LDS buffer: 4 rows x 256 B per global instruction.
#include <stdlib.h>

#include "lru_train_scores.h"
#include "lr_det.h"
LR_DET_DEFINE(scores)

typedef float floatx4 __attribute__((ext_vector_type(4)));

#define TS_NS_MAX 32       // most item splits of the score pass (grid.y)
#define TS_NEG (-3.0e38f)  // finite stand-in for -inf in the running maxima (2^(TS_NEG - TS_NEG) = 1, never NaN)
#define TS_LOG2E 1.4426950408889634f
#define TS_LP 68           // LDS pitch (floats) of a wave's [16][64] logits tile: 17 chunks of 16 B, conflict-free both ways



struct TsArgs {
  const float *x, *E, *bias;
  const long long* labels;
  int R, C, ldl, Rpad;
  int ns, ns2, nq;   // item splits of the score pass / of the d x pass; row parts of the d E pass
  float *logits, *Ef, *ETf, *xTf, *biasf, *lse, *part, *scal;   // logits: [Rpad][ldl], in the log2 domain (x log2 e)
  int* lab32;
  float *dX, *dE, *dbias;
};

__device__ __forceinline__ float ts_e(const float4& v, int e) { return e == 0 ? v.x : e == 1 ? v.y : e == 2 ? v.z : v.w; }

// ---- operands in fragment order (lane = 16 g + li) ------------------------------------------------------------------------
//   Ef [sb][nb][j][lane] = E[64 sb + 16 nb + li][16 j + 4 g ..]             (0 for items >= C)
//   ETf[sb][j][nb][lane] = E[64 sb + 16 j + 4 g + e][16 nb + li], e = 0..3   (0 for items >= C)
//   xTf[it][nb][lane]    = x[16 it + 4 g + e][16 nb + li], e = 0..3          (0 for rows >= R)
//   biasf[v]             = bias[v] log2 e                                   (0 for items >= C): the logits are kept in the
//                          log2 domain (x and the bias scaled by log2 e), so every exponential downstream is one v_exp_f32
__global__ __launch_bounds__(256) void ts_fragments_kernel(TsArgs a) {
  const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
  const int nsb = (a.ldl + 63) >> 6, nit = a.Rpad >> 4;
  const long long nE = (long long)nsb * 1024, nX = (long long)nit * 256;
  const int lane = (int)(t & 63), li = lane & 15, g = lane >> 4;
  float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
  if (t < nE) {
    const int sb = (int)(t >> 10), nb = (int)(t >> 8) & 3, j = (int)(t >> 6) & 3;
    const int item = 64 * sb + 16 * nb + li;
    if (item < a.C) v = *reinterpret_cast<const float4*>(a.E + (size_t)item * 64 + 16 * j + 4 * g);
    *reinterpret_cast<float4*>(a.Ef + t * 4) = v;
  } else if (t < 2 * nE) {
    const long long u = t - nE;
    const int sb = (int)(u >> 10), j = (int)(u >> 8) & 3, nb = (int)(u >> 6) & 3;
    const int item = 64 * sb + 16 * j + 4 * g, d = 16 * nb + li;
    v.x = item + 0 < a.C ? a.E[(size_t)(item + 0) * 64 + d] : 0.f;
    v.y = item + 1 < a.C ? a.E[(size_t)(item + 1) * 64 + d] : 0.f;
    v.z = item + 2 < a.C ? a.E[(size_t)(item + 2) * 64 + d] : 0.f;
    v.w = item + 3 < a.C ? a.E[(size_t)(item + 3) * 64 + d] : 0.f;
    *reinterpret_cast<float4*>(a.ETf + u * 4) = v;
  } else if (t >= 2 * nE + nX) {
    const long long u = t - 2 * nE - nX;   // bias x log2 e, 0 on the padding columns
    if (u < a.ldl) a.biasf[u] = u < a.C ? a.bias[u] * TS_LOG2E : 0.f;
  } else {
    const long long u = t - 2 * nE;
    const int it = (int)(u >> 8), nb = (int)(u >> 6) & 3;
    const int row = 16 * it + 4 * g, d = 16 * nb + li;
    v.x = row + 0 < a.R ? a.x[(size_t)(row + 0) * 64 + d] : 0.f;
    v.y = row + 1 < a.R ? a.x[(size_t)(row + 1) * 64 + d] : 0.f;
    v.z = row + 2 < a.R ? a.x[(size_t)(row + 2) * 64 + d] : 0.f;
    v.w = row + 3 < a.R ? a.x[(size_t)(row + 3) * 64 + d] : 0.f;
    *reinterpret_cast<float4*>(a.xTf + u * 4) = v;
  }
}

struct TsScLoads {
  float4 w[4][4];   // Ef[sb][nb][j][lane]
  float4 b[4];      // bias[v0 + 16 nb + 4 g ..]
};
// ---- scores + running (max, sum exp) -------------------------------------------------------------------------------------
// grid (row tiles of 64, a.ns item splits). A WAVE owns a 16-row panel; the four waves of a workgroup walk the same 64-item
// super-blocks of the split (the table fragments come from L2 once per workgroup and from L1 for the other three waves).
#define TS_P 3   // 16-row panels per wave in the score pass: a fragment load of the table feeds TS_P x 64 MFMAs, so the loop's
                 // per-block costs that do not overlap with f32 MFMAs (20 operand loads and their address arithmetic, the waits
                 // in front of the first MFMA) are paid once per TS_P x 64: 85 us with one panel, 73 with two, 69 with three
                 // (319 registers; four: 366 and accumulator copies through AGPRs, not tried on the GPU). The same change in
                 // the d x pass (two panels; its second stream, the logits, grows with the rows): 66 -> 82 us, not kept.
                 // (Not a bandwidth effect: tools/diag/mfma_f32_issue.hip pulls 59 B/clk per CU of coalesced loads under a full
                 // MFMA stream at one wave per SIMD.)
__global__ __launch_bounds__(256) void ts_scores_kernel(TsArgs a) {
  __shared__ __attribute__((aligned(16))) float lt[4][TS_P * 16 * TS_LP];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 15, g = lane >> 4;
  const int row0 = (blockIdx.x * 4 + wave) * (16 * TS_P);
  const int nsb = (a.ldl + 63) >> 6;
  const int sb0 = (int)((long long)blockIdx.y * nsb / a.ns), sb1 = (int)((long long)(blockIdx.y + 1) * nsb / a.ns);
  float* tile = lt[wave];
  float4 xa[TS_P][4];
#pragma unroll
  for (int q = 0; q < TS_P; ++q)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float4 t = *reinterpret_cast<const float4*>(a.x + (size_t)min(row0 + 16 * q + li, a.R - 1) * 64 + 16 * j + 4 * g);
      xa[q][j] = make_float4(t.x * TS_LOG2E, t.y * TS_LOG2E, t.z * TS_LOG2E, t.w * TS_LOG2E);
    }
  float m[TS_P], s[TS_P];   // running maximum and sum of 2^(v - m), v = logit log2 e
#pragma unroll
  for (int q = 0; q < TS_P; ++q) {
    m[q] = TS_NEG;
    s[q] = 0.f;
  }
  auto load = [&](int sb, TsScLoads& L) {
    const float* f = a.Ef + (size_t)sb * 4096 + lane * 4;
#pragma unroll
    for (int nb = 0; nb < 4; ++nb) {
#pragma unroll
      for (int j = 0; j < 4; ++j) L.w[nb][j] = *reinterpret_cast<const float4*>(f + (nb * 4 + j) * 256);
      L.b[nb] = *reinterpret_cast<const float4*>(a.biasf + (sb << 6) + 16 * nb + 4 * g);
    }
  };
  auto compute = [&](int sb, const TsScLoads& L) {
    const int v0 = sb << 6;
#pragma unroll
    for (int q = 0; q < TS_P; ++q) {
      floatx4 acc[4];   // the bias is the MFMA chain's C operand: D[n = 4 g + r][row] starts at biasf[n], the float4 this lane loaded
#pragma unroll
      for (int nb = 0; nb < 4; ++nb) acc[nb] = floatx4{L.b[nb].x, L.b[nb].y, L.b[nb].z, L.b[nb].w};
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
          for (int nb = 0; nb < 4; ++nb)
            acc[nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(ts_e(L.w[nb][j], e), ts_e(xa[q][j], e), acc[nb], 0, 0, 0);
#pragma unroll
      for (int nb = 0; nb < 4; ++nb) {
        const int n = v0 + 16 * nb + 4 * g;   // this lane's 4 items of the block
        float v[4] = {acc[nb][0], acc[nb][1], acc[nb][2], acc[nb][3]};
        *reinterpret_cast<float4*>(tile + (16 * q + li) * TS_LP + 16 * nb + 4 * g) = make_float4(v[0], v[1], v[2], v[3]);
        if (n + 4 > a.C) {   // the row's last items: padding columns do not count
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = (n + e < a.C) ? v[e] : TS_NEG;
        }
        const float mn = fmaxf(m[q], fmaxf(fmaxf(v[0], v[1]), fmaxf(v[2], v[3])));
        s[q] = s[q] * __builtin_amdgcn_exp2f(m[q] - mn) + ((__builtin_amdgcn_exp2f(v[0] - mn) + __builtin_amdgcn_exp2f(v[1] - mn)) +
                                                           (__builtin_amdgcn_exp2f(v[2] - mn) + __builtin_amdgcn_exp2f(v[3] - mn)));
        m[q] = mn;
      }
    }
    // the wave's (16 TS_P) x 64 tile leaves through LDS: 4 rows x 256 B per store instruction (lane = row 4 i + g, chunk li).
    // The tile is wave-private, written by one set of lanes and read straight back by another: the wave barrier + fence pin the
    // order of those may-alias LDS accesses for any future scheduler (no instruction is emitted for them)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    const int col = v0 + 4 * li;
#pragma unroll
    for (int i = 0; i < 4 * TS_P; ++i) {
      const int r = 4 * i + g;
      const float4 o = *reinterpret_cast<const float4*>(tile + r * TS_LP + 4 * li);
      if (row0 + r < a.Rpad) *reinterpret_cast<float4*>(a.logits + (size_t)(row0 + r) * a.ldl + col) = o;   // padding rows: finite values
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    __builtin_amdgcn_wave_barrier();   // ... and the next block's stores stay behind these reads
  };
  // One register set of loads ahead. The sched_barrier keeps hipcc from sinking the next set's loads below the MFMAs (it
  // does, to save registers). The loads are UNCONDITIONAL (index clamped; a set that is not needed is loaded and dropped):
  // behind a conditional load hipcc's vmcnt bookkeeping assumes the shorter queue at the join and every wait then covers
  // the set just issued.
  TsScLoads A, B;
  int sb = sb0;
  const int last = max(sb1 - 1, sb0);
  load(min(sb, last), A);
  for (; sb < sb1; sb += 2) {
    load(min(sb + 1, last), B);
    __builtin_amdgcn_sched_barrier(0);
    compute(sb, A);
    load(min(sb + 2, last), A);
    __builtin_amdgcn_sched_barrier(0);
    if (sb + 1 < sb1) compute(sb + 1, B);
  }
#pragma unroll
  for (int q = 0; q < TS_P; ++q) {
    float mq = m[q], sq = s[q];
#pragma unroll
    for (int off = 16; off <= 32; off <<= 1) {   // the 4 lane groups of a row
      const float m2 = __shfl_xor(mq, off, 64), s2 = __shfl_xor(sq, off, 64);
      const float mn = fmaxf(mq, m2);
      sq = sq * __builtin_amdgcn_exp2f(mq - mn) + s2 * __builtin_amdgcn_exp2f(m2 - mn);
      mq = mn;
    }
    const int row = row0 + 16 * q + li;
    if (g == 0 && row < a.R) {
      float* p = a.part + ((size_t)row * a.ns + blockIdx.y) * 2;
      p[0] = mq;
      p[1] = sq;
    }
  }
}

// ---- per row (a wave each): lse, the loss term, what the gradient kernels want, and the one-hot half of dl -----------------
//   lse2[row] = log2 sum 2^logit2 + log2 n  (so that 2^(logit2 - lse2) = softmax / n), +inf for unlabelled and padding rows
//   dl = softmax / n - onehot / n: the three gradient products use the first term; the second is rank 1 per row and is added
//   here: d x[row] = -E[label] / n (d x is zero before this launch), d E[label] -= x[row] / n, d bias[label] -= 1 / n
__global__ __launch_bounds__(256) void ts_combine_kernel(TsArgs a) {
  __shared__ float sh[4];
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  float loss = 0.f;
  if (row < a.Rpad) {
    float lse2 = __builtin_inff();
    int lab = -1;
    if (row < a.R) {
      const long long l = a.labels[row];
      if (l > 0 && l < a.C) {   // 0 = ignore_index; out of range: ignored (and counted in scal[3])
        const float* p = a.part + (size_t)row * a.ns * 2;
        const float mk = lane < a.ns ? p[2 * lane] : TS_NEG, sk = lane < a.ns ? p[2 * lane + 1] : 0.f;
        float M = mk;
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) M = fmaxf(M, __shfl_xor(M, o, 64));
        float S = sk * exp2f(mk - M);
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) S += __shfl_xor(S, o, 64);
        const float l2 = M + log2f(S);              // log2 of the sum of 2^logit2
        lab = (int)l;
        const float nv = a.scal[1];                 // >= 1: this row is labelled
        loss = (l2 - a.logits[(size_t)row * a.ldl + l]) * 0.6931471805599453f;
        lse2 = l2 + log2f(nv);
        const float inv_n = 1.0f / nv;
        a.dX[(size_t)row * 64 + lane] = -inv_n * a.E[(size_t)lab * 64 + lane];
        lr_det_add(a.dE + (size_t)lab * 64 + lane, -inv_n * a.x[(size_t)row * 64 + lane]);
        if (lane == 0) lr_det_add(a.dbias + lab, -inv_n);
      }
    }
    if (lane == 0) {
      a.lse[row] = lse2;
      a.lab32[row] = lab;
    }
  }
  if (lane == 0) sh[threadIdx.x >> 6] = loss;
  __syncthreads();
  if (threadIdx.x == 0) {
    const float t = (sh[0] + sh[1]) + (sh[2] + sh[3]);
    if (t != 0.f) lr_det_add(a.scal, t);
  }
}

// softmax / n of 4 consecutive items of one row: fma + v_exp_f32 per element (0 for rows with lse2 = +inf)
__device__ __forceinline__ float4 ts_p4(float4 l, float lse2) {
  return make_float4(__builtin_amdgcn_exp2f(l.x - lse2), __builtin_amdgcn_exp2f(l.y - lse2), __builtin_amdgcn_exp2f(l.z - lse2),
                     __builtin_amdgcn_exp2f(l.w - lse2));
}

// ---- d x += dl E -------------------------------------------------------------------------------------------------------------
// grid (row tiles of 64, a.ns2 item splits): a wave owns a 16-row panel, the four waves walk the same super-blocks; the splits
// add into d x (a.ns2 adders per element; ts_combine_kernel left the one-hot term there). Columns past C: the logits there
// are finite, E^T's fragments are 0.
__global__ __launch_bounds__(256) void ts_dx_kernel(TsArgs a) {
  __shared__ __attribute__((aligned(16))) float lt[4][16 * TS_LP];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 15, g = lane >> 4;
  const int row0 = blockIdx.x * 64 + wave * 16;
  const int row = row0 + li;   // < Rpad64
  const float lse2 = row < a.Rpad ? a.lse[row] : __builtin_inff();
  const int nsb = (a.ldl + 63) >> 6;
  const int sb0 = (int)((long long)blockIdx.y * nsb / a.ns2), sb1 = (int)((long long)(blockIdx.y + 1) * nsb / a.ns2);
  float* tile = lt[wave];
  // rows 4 i + g of the panel, i = 0 .. 3: lane offsets (floats) from a per-super-block uniform base
  const int lo0 = min(row0 + g, a.Rpad - 1) * a.ldl + 4 * li, lo1 = min(row0 + 4 + g, a.Rpad - 1) * a.ldl + 4 * li;
  const int lo2 = min(row0 + 8 + g, a.Rpad - 1) * a.ldl + 4 * li, lo3 = min(row0 + 12 + g, a.Rpad - 1) * a.ldl + 4 * li;
  floatx4 acc[4];
#pragma unroll
  for (int nb = 0; nb < 4; ++nb) acc[nb] = floatx4{0.f, 0.f, 0.f, 0.f};
  // (macros, not lambdas over a struct: with the logits' registers inside a struct passed by reference hipcc kept them on the
  // stack -- scratch_store / scratch_load round trips in front of every MFMA block)
#define DX_LOAD(sb_, l_, w_)                                                                                      \
  {                                                                                                               \
    const int sbc_ = min((sb_), nsb - 1);                                                                         \
    const float* lb_ = a.logits + (sbc_ << 6);                                                                    \
    l_##0 = *reinterpret_cast<const float4*>(lb_ + lo0);                                                          \
    l_##1 = *reinterpret_cast<const float4*>(lb_ + lo1);                                                          \
    l_##2 = *reinterpret_cast<const float4*>(lb_ + lo2);                                                          \
    l_##3 = *reinterpret_cast<const float4*>(lb_ + lo3);                                                          \
    const float* f_ = a.ETf + (size_t)sbc_ * 4096 + lane * 4;                                                     \
    _Pragma("unroll") for (int j = 0; j < 4; ++j)                                                                 \
    _Pragma("unroll") for (int nb = 0; nb < 4; ++nb) w_[nb][j] = *reinterpret_cast<const float4*>(f_ + (j * 4 + nb) * 256); \
  }
#define DX_COMPUTE(l_, w_)                                                                                        \
  {                                                                                                               \
    *reinterpret_cast<float4*>(tile + (0 + g) * TS_LP + 4 * li) = l_##0;                                          \
    *reinterpret_cast<float4*>(tile + (4 + g) * TS_LP + 4 * li) = l_##1;                                          \
    *reinterpret_cast<float4*>(tile + (8 + g) * TS_LP + 4 * li) = l_##2;                                          \
    *reinterpret_cast<float4*>(tile + (12 + g) * TS_LP + 4 * li) = l_##3;                                         \
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");   /* the wave-private tile: stores by one set of lanes, reads by another */ \
    __builtin_amdgcn_wave_barrier();                                                                              \
    /* all 16 softmax values first, then the 64 MFMAs back to back: left to itself hipcc forms one value per four MFMAs */ \
    /* in ONE register, the subtract -> v_exp_f32 chain behind every fourth MFMA (67-68 -> 65-66 us; the same regrouping in */ \
    /* ts_de_kernel, whose values feed sixteen MFMAs each, cost 57 -> 77 us: there the interleaved form hides them)      */ \
    float4 dl_[4];                                                                                                \
    _Pragma("unroll") for (int j = 0; j < 4; ++j)                                                                 \
      dl_[j] = ts_p4(*reinterpret_cast<const float4*>(tile + li * TS_LP + 16 * j + 4 * g), lse2);                 \
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");   /* the next block's stores stay behind these reads */ \
    __builtin_amdgcn_wave_barrier();                                                                              \
    __builtin_amdgcn_sched_barrier(0);                                                                            \
    _Pragma("unroll") for (int j = 0; j < 4; ++j)                                                                 \
      _Pragma("unroll") for (int e = 0; e < 4; ++e)                                                               \
      _Pragma("unroll") for (int nb = 0; nb < 4; ++nb)                                                            \
        acc[nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(ts_e(w_[nb][j], e), ts_e(dl_[j], e), acc[nb], 0, 0, 0);    \
    __builtin_amdgcn_sched_barrier(0);                                                                            \
  }
  float4 la0, la1, la2, la3, lb0, lb1, lb2, lb3, wa[4][4], wb[4][4];   // unconditional loads, see ts_scores_kernel (DX_LOAD clamps the addresses itself)
  int sb = sb0;
  DX_LOAD(sb, la, wa)
  for (; sb < sb1; sb += 2) {
    DX_LOAD(sb + 1, lb, wb)
    __builtin_amdgcn_sched_barrier(0);
    DX_COMPUTE(la, wa)
    DX_LOAD(sb + 2, la, wa)
    __builtin_amdgcn_sched_barrier(0);
    if (sb + 1 < sb1) DX_COMPUTE(lb, wb)
  }
#undef DX_LOAD
#undef DX_COMPUTE
  if (row < a.R) {
#pragma unroll
    for (int nb = 0; nb < 4; ++nb)
#pragma unroll
      for (int e = 0; e < 4; ++e) lr_det_add(a.dX + (size_t)row * 64 + 16 * nb + 4 * g + e, acc[nb][e]);
  }
}

// ---- d E += dl^T x, d bias += column sums of dl -------------------------------------------------------------------------------
// grid: item panels of 64; lane li reads items v0 + 4 li .. + 3 of a row (element s -> sub-panel s; 4 rows x 256 B per
// instruction); wave w streams the 16-row groups w, w + 4, ..; within a group lane group g owns rows r0 + 4 g + e, e = the
// MFMA step. Items past C compute garbage that is never stored.
struct TsDeLoads {
  float4 l[4];    // logits[r0 + 4 g + e][v0 + 4 li ..]
  float4 lse;     // lse2[r0 + 4 g ..]
  float4 x[4];    // xTf[it][nb][lane]
};
__global__ __launch_bounds__(256) void ts_de_kernel(TsArgs a) {
  extern __shared__ __attribute__((aligned(16))) float dyn[];   // red[4][64][64] + bred[4][64]
  float* red = dyn;
  float* bred = dyn + 4 * 64 * 64;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 15, g = lane >> 4;
  const int v0 = blockIdx.x * 64, first = v0 + 4 * li;
  int lo[4];   // lane offsets (floats) of rows 4 g + e from a per-step uniform base
#pragma unroll
  for (int e = 0; e < 4; ++e) lo[e] = (4 * g + e) * a.ldl + first;
  const int nit_all = a.Rpad >> 4;   // 16-row groups; this workgroup's part (grid.y = a.nq parts):
  const int it0 = (int)((long long)blockIdx.y * nit_all / a.nq), nit = (int)((long long)(blockIdx.y + 1) * nit_all / a.nq);
  floatx4 acc[4][4];   // [sub-panel s][feature block nb]: D[d = 16 nb + 4 g + r][item v0 + 4 li + s]
#pragma unroll
  for (int s = 0; s < 4; ++s)
#pragma unroll
    for (int nb = 0; nb < 4; ++nb) acc[s][nb] = floatx4{0.f, 0.f, 0.f, 0.f};
  float bs[4] = {0.f, 0.f, 0.f, 0.f};
  auto load = [&](int it, TsDeLoads& L) {
    const int r0 = (it << 4) + 4 * g;
    const float* lb = a.logits + (size_t)(it << 4) * a.ldl;   // [Rpad] rows exist: no clamp
#pragma unroll
    for (int e = 0; e < 4; ++e) L.l[e] = *reinterpret_cast<const float4*>(lb + lo[e]);
    L.lse = *reinterpret_cast<const float4*>(a.lse + r0);
    const float* f = a.xTf + (size_t)it * 1024 + lane * 4;
#pragma unroll
    for (int nb = 0; nb < 4; ++nb) L.x[nb] = *reinterpret_cast<const float4*>(f + nb * 256);
  };
  auto compute = [&](const TsDeLoads& L) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float4 dl = ts_p4(L.l[e], ts_e(L.lse, e));   // softmax / n; the one-hot term: ts_combine_kernel
      bs[0] += dl.x;
      bs[1] += dl.y;
      bs[2] += dl.z;
      bs[3] += dl.w;
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int nb = 0; nb < 4; ++nb)
          acc[s][nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(ts_e(L.x[nb], e), ts_e(dl, s), acc[s][nb], 0, 0, 0);
    }
  };
  TsDeLoads A, B;   // unconditional loads, see ts_scores_kernel
  int it = it0 + wave;
  load(min(it, nit - 1), A);
  for (; it < nit; it += 8) {
    load(min(it + 4, nit - 1), B);
    __builtin_amdgcn_sched_barrier(0);
    compute(A);
    load(min(it + 8, nit - 1), A);
    __builtin_amdgcn_sched_barrier(0);
    if (it + 4 < nit) compute(B);
  }
#pragma unroll
  for (int s = 0; s < 4; ++s) {
#pragma unroll
    for (int nb = 0; nb < 4; ++nb)
      *reinterpret_cast<float4*>(red + ((size_t)wave * 64 + 4 * li + s) * 64 + 16 * nb + 4 * g) =
          make_float4(acc[s][nb][0], acc[s][nb][1], acc[s][nb][2], acc[s][nb][3]);
    float b = bs[s];
    b += __shfl_xor(b, 16, 64);
    b += __shfl_xor(b, 32, 64);
    if (g == 0) bred[wave * 64 + 4 * li + s] = b;
  }
  __syncthreads();
  // the row parts of a panel add into d E / d bias (pre-zeroed; the one-hot term is already there): 64 consecutive lanes
  // cover one item's 256-byte row per atomic instruction
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int el = tid + 256 * i, item = el >> 6, d = el & 63;
    if (v0 + item < a.C) {
      const float o = (red[(size_t)item * 64 + d] + red[(size_t)(64 + item) * 64 + d]) +
                      (red[(size_t)(128 + item) * 64 + d] + red[(size_t)(192 + item) * 64 + d]);
      lr_det_add(a.dE + (size_t)(v0 + item) * 64 + d, o);
    }
  }
  if (tid < 64 && v0 + tid < a.C) lr_det_add(a.dbias + v0 + tid, (bred[tid] + bred[64 + tid]) + (bred[128 + tid] + bred[192 + tid]));
}

// ---- host side --------------------------------------------------------------------------------------------------------------
static inline int ts_ldl(int C) { return (C + 63) & ~63; }   // logits pitch: whole 64-item super-blocks
static inline int ts_rpad(int R) { return (R + 15) & ~15; }
static inline size_t ts_up64(size_t n) { return (n + 63) & ~(size_t)63; }

size_t lr_train_scores_ws_floats(int R, int C) {
  const size_t ldl = ts_ldl(C), rp = ts_rpad(R), nsb = (ldl + 63) / 64;
  return ts_up64(rp * ldl) + 2 * nsb * 4096 + (rp / 16) * 1024 + ldl + ts_up64(rp) + ts_up64(rp) + ts_up64((size_t)R * TS_NS_MAX * 2);
}

int lr_launch_train_scores(const float* x, const float* E, const float* bias, const long long* labels, int R, int C, float* ws,
                           float* scal, float* dX, float* dE, float* dbias, hipStream_t st) {
  static bool lds_done[LR_MAX_DEVICES];
  if (R < 1 || C < 2) LR_FAIL(LR_EINVAL, "lr_launch_train_scores: R=%d C=%d", R, C);
  TsArgs a;
  a.x = x; a.E = E; a.bias = bias; a.labels = labels;
  a.R = R; a.C = C; a.ldl = ts_ldl(C); a.Rpad = ts_rpad(R);
  const int nsb = (a.ldl + 63) / 64;
  float* p = ws;
  a.logits = p; p += ts_up64((size_t)a.Rpad * a.ldl);
  a.Ef = p;     p += (size_t)nsb * 4096;
  a.ETf = p;    p += (size_t)nsb * 4096;
  a.xTf = p;    p += (size_t)(a.Rpad / 16) * 1024;
  a.biasf = p;  p += a.ldl;
  a.lse = p;    p += ts_up64(a.Rpad);
  a.lab32 = reinterpret_cast<int*>(p); p += ts_up64(a.Rpad);
  a.part = p;
  a.scal = scal; a.dX = dX; a.dE = dE; a.dbias = dbias;
  // Splits: a workgroup is 64 TS_P rows (score pass; 64 rows in d x) x (items / splits): one full round of the chip's 256
  // workgroup slots per launch (both kernels hold > 128 registers per lane: one workgroup per CU)
  const int nrt = (R + 64 * TS_P - 1) / (64 * TS_P), n64 = (R + 63) / 64;
  a.ns = 256 / nrt < 1 ? 1 : 256 / nrt > TS_NS_MAX ? TS_NS_MAX : 256 / nrt;
  a.ns2 = 256 / n64 < 1 ? 1 : 256 / n64 > 8 ? 8 : 256 / n64;
  if (a.ns > nsb) a.ns = nsb;
  if (a.ns2 > nsb) a.ns2 = nsb;
  {  // d E: P item panels x nq row parts, nq chosen for the shortest makespan on 256 CUs (a CU's workgroups run one after
     // the other: f32 MFMA and vector work do not overlap across waves), parts of at least 8 row groups
    const int P = (C + 63) / 64, nit = a.Rpad / 16;
    a.nq = 1;
    double best = 1e30;
    for (int q = 1; q <= 8 && nit / q >= 8; ++q) {
      const double span = (double)((P * q + 255) / 256) / q;   // rounds of workgroups x work per workgroup
      if (span < best - 1e-9) { best = span; a.nq = q; }
    }
  }
  const long long nt = 2LL * nsb * 1024 + (long long)(a.Rpad / 16) * 256 + a.ldl;
  hipLaunchKernelGGL(ts_fragments_kernel, dim3((unsigned)((nt + 255) / 256)), dim3(256), 0, st, a);
  LR_CHECK_LAUNCH("ts_fragments_kernel");
  hipLaunchKernelGGL(ts_scores_kernel, dim3(nrt, a.ns), dim3(256), 0, st, a);
  LR_CHECK_LAUNCH("ts_scores_kernel");
  hipLaunchKernelGGL(ts_combine_kernel, dim3((a.Rpad + 3) / 4), dim3(256), 0, st, a);
  LR_CHECK_LAUNCH("ts_combine_kernel");
  hipLaunchKernelGGL(ts_dx_kernel, dim3(n64, a.ns2), dim3(256), 0, st, a);
  LR_CHECK_LAUNCH("ts_dx_kernel");
  const int de_lds = (4 * 64 * 64 + 4 * 64) * (int)sizeof(float);
  int rc = lr_ensure_dynamic_lds((const void*)ts_de_kernel, de_lds, lds_done);
  if (rc) return rc;
  hipLaunchKernelGGL(ts_de_kernel, dim3((C + 63) / 64, a.nq), dim3(256), de_lds, st, a);
  LR_CHECK_LAUNCH("ts_de_kernel");
  return LR_OK;
}
