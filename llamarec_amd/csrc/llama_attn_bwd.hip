// llama_attn_bwd.hip -- backward of the varlen causal self-attention (llama_attn.hip) for the ranker's LoRA
// training step.
//
// Replaces flash-attn 2.5.8's CUDA backward (train_ranker.py:61, reached through torch autograd from
// trainer/llm.py's HF Trainer loop). Inputs: the rotated q/k/v of the forward pass (packed [n][(nh+2nkv)*hd]), the
// forward output O, its gradient dO, and the forward's log-sum-exp per (token, head). Output: d q, d k, d v in the
// same packed layout. With P = exp(S - lse), D = rowsum(dO .* O):
//     dV = P^T dO,   dP = dO V^T,   dS = P .* (dP - D),   dQ = scale dS K,   dK = scale dS^T Q.
//
//  attn_bwd_generic_kernel : any head_dim <= 256, GQA; one wave per (token, head), fp32 atomics for dK/dV. Test models.
//  attn_bwd_dq_kernel / attn_bwd_dkv_kernel : head_dim 128 on v_mfma_f32_16x16x32_bf16, no atomics. Two passes, each
//    recomputing S and dP (7 tile products instead of 5), so each output has exactly one owner:
//      dQ pass  : a workgroup owns 128 query rows (like the forward); everything transposed: S^T = K Q^T and
//                 dP^T = V dO^T put a query row in a lane's accumulator column, so lse / D are one scalar per lane and
//                 dS^T is, as it stands in the accumulators, the B operand of dQ^T += K^T dS^T (K^T through
//                 ds_read_b64_tr_b16 from the same row-major LDS tile).
//      dKdV pass: a workgroup owns 64 keys of one kv head and walks the query blocks at or after them (and the query
//                 heads of its group): S = Q K^T, dP = dO V^T put a key in the accumulator column; P and dS are the B
//                 operands of dV^T += dO^T P and dK^T += Q^T dS (Q^T, dO^T through transposed LDS reads).
#include <stdlib.h>

#include "llama_train.h"
#include "lr_profile.h"

typedef unsigned short u16;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef short short4v __attribute__((ext_vector_type(4)));
typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef u16 u16x4 __attribute__((ext_vector_type(4)));

// =============================================================================================
// generic
// =============================================================================================
__global__ __launch_bounds__(256) void attn_bwd_generic_kernel(const u16* qkv, const u16* d_out, const float* lse,
                                                               const float* dsum, u16* dqkv, float* dkv32,
                                                               const int32_t* cu, int B, int n_tok, int nh, int nkv,
                                                               int hd) {
  __shared__ float qs[4][256], dos[4][256];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int item = blockIdx.x * 4 + wave;  // (token, head)
  if (item >= n_tok * nh) return;
  const int tok = item / nh, h = item % nh;
  const int kvh = h / (nh / nkv);
  const int stride = (nh + 2 * nkv) * hd;
  int lo = 0, hi = B;
  while (hi - lo > 1) {
    int mid = (lo + hi) >> 1;
    if (cu[mid] <= tok) lo = mid; else hi = mid;
  }
  const int s0 = cu[lo];
  const int pos = tok - s0;
  for (int d = lane; d < hd; d += 64) {
    qs[wave][d] = bf2f(qkv[(size_t)tok * stride + h * hd + d]);
    dos[wave][d] = bf2f(d_out[(size_t)tok * nh * hd + h * hd + d]);
  }
  __builtin_amdgcn_wave_barrier();
  const float scale = 1.0f / sqrtf((float)hd);
  const float l = lse[(size_t)tok * nh + h], D = dsum[(size_t)tok * nh + h];
  float dq[4] = {0.f, 0.f, 0.f, 0.f};
  for (int k0 = 0; k0 <= pos; k0 += 64) {
    const int key = k0 + lane;
    float p = 0.f, ds = 0.f;
    if (key <= pos) {
      const u16* kp = qkv + (size_t)(s0 + key) * stride + (nh + kvh) * hd;
      const u16* vp = qkv + (size_t)(s0 + key) * stride + (nh + nkv + kvh) * hd;
      float acc = 0.f, dp = 0.f;
      for (int d = 0; d < hd; ++d) {
        acc = __builtin_fmaf(qs[wave][d], bf2f(kp[d]), acc);
        dp = __builtin_fmaf(dos[wave][d], bf2f(vp[d]), dp);
      }
      p = __expf(acc * scale - l);
      ds = p * (dp - D) * scale;
    }
    const int nk = min(64, pos - k0 + 1);
    for (int j = 0; j < nk; ++j) {
      const float pj = __shfl(p, j, 64), dsj = __shfl(ds, j, 64);
      const size_t krow = (size_t)(s0 + k0 + j);
      const u16* kp = qkv + krow * stride + (nh + kvh) * hd;
      float* dk = dkv32 + krow * (2 * nkv * hd) + kvh * hd;
      float* dv = dk + nkv * hd;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int d = lane + 64 * i;
        if (d < hd) {
          dq[i] = __builtin_fmaf(dsj, bf2f(kp[d]), dq[i]);
          atomicAdd(dk + d, dsj * qs[wave][d]);
          atomicAdd(dv + d, pj * dos[wave][d]);
        }
      }
    }
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int d = lane + 64 * i;
    if (d < hd) dqkv[(size_t)tok * stride + h * hd + d] = f2bf(dq[i]);
  }
}

__global__ __launch_bounds__(256) void attn_bwd_kv_to_bf16_kernel(const float* dkv32, u16* dqkv, int n_tok, int qcols,
                                                                  int kvcols /* 2*nkv*hd */) {
  const size_t total = (size_t)n_tok * kvcols;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const size_t row = i / kvcols;
    const int c = (int)(i % kvcols);
    dqkv[row * (qcols + kvcols) + qcols + c] = f2bf(dkv32[i]);
  }
}

// =============================================================================================
// MFMA, head_dim 128
// =============================================================================================
#define AB_KB 64                         // keys (dQ pass) / queries (dKdV pass) per streamed block
#define AB_TILE_BYTES (AB_KB * 256)      // 64 rows x 128 dims bf16
#define AB_STAGE_BYTES (2 * AB_TILE_BYTES)
#define AB_QROWS 128                     // query rows per workgroup in the dQ pass

// dual-use swizzle (row reads with ds_read_b128 AND transposed reads), 256-byte rows: 16-byte chunk `ch` of `row`
__device__ __forceinline__ int ab_sw(int row) { return ((row & 3) << 2) | ((row >> 2) & 3); }
__device__ __forceinline__ int ab_off(int row, int ch) { return 256 * row + 16 * (ch ^ ab_sw(row)); }

// LDS-DMA as inline asm (see llama_attn.hip: issued through the builtin, hipcc puts `s_waitcnt vmcnt(0)` in front of the
// first transposed LDS read that follows -- here in the middle of the iteration that should hide the fetch). hipcc does not
// count these requests: the one wait they need is written by hand in front of each iteration's barrier.
__device__ __forceinline__ void ab_glds16(const void* gsrc, void* lds_wave_base) {
  const unsigned m0v = (unsigned)(size_t)((__attribute__((address_space(3))) const char*)lds_wave_base);
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" ::"s"(m0v), "v"(gsrc) : "memory");
}
#define AB_DMA_LANDED() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")

// stage one 64 x 128 tile (rows row0.., clamped to [0, T-1]) of a row-major matrix with `stride` elements per row:
// 16 pieces of 4 rows, wave w moves pieces 4w..4w+3 (LDS image lane-linear, swizzle applied to the source chunk)
__device__ __forceinline__ void ab_stage_tile(const u16* base, size_t stride, int row0, int T, char* tile, int wave,
                                              int lane) {
  const int prow = lane >> 4, ppos = lane & 15;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row = (wave * 4 + i) * 4 + prow;
    const int src = min(row0 + row, T - 1);
    ab_glds16(base + (size_t)src * stride + ((ppos ^ ab_sw(row)) << 3), tile + (wave * 4 + i) * 1024);
  }
}

// A operand = X^T fragment for a 16-dim tile dt and a 32-row step ks2 of a row-major LDS tile X[row][d]:
// element e < 4: row ks2*32 + 4*quad + e, e >= 4: row ks2*32 + 16 + 4*quad + (e - 4); M index (lane & 15) = dim.
__device__ __forceinline__ bf16x8 ab_read_tr(const char* tile, int ks2, int dt, int quad, int li) {
  const int qp = li >> 2, p4 = li & 3;
  const int row0 = ks2 * 32 + quad * 4 + qp;
  const int ch = dt * 2 + (p4 >> 1);
  const short4v t0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
      (__attribute__((address_space(3))) short4v*)(tile + ab_off(row0, ch) + 8 * (p4 & 1)));
  const short4v t1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
      (__attribute__((address_space(3))) short4v*)(tile + ab_off(row0 + 16, ch) + 8 * (p4 & 1)));
  const bf16x4 b0 = __builtin_bit_cast(bf16x4, t0), b1 = __builtin_bit_cast(bf16x4, t1);
  bf16x8 f;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    f[r] = b0[r];
    f[4 + r] = b1[r];
  }
  return f;
}

// Optional epilogue: gradient w.r.t. the UNROTATED q / k. A lane holds dims dt*16 + 4*quad + 0..3 = rotation pairs
// i0 = dt*8 + 2*quad and i0 + 1 of its row (packed layout: pair (x1_i, x2_i) adjacent); transpose of the rotation.
__device__ __forceinline__ void ab_unrotate(float (&v)[4], const float* rope_cs, int pos, int dt, int quad) {
  if (!rope_cs) return;
  const float4 cs = *reinterpret_cast<const float4*>(rope_cs + ((size_t)pos * 64 + dt * 8 + 2 * quad) * 2);
  const float a0 = v[0], a1 = v[1], b0 = v[2], b1 = v[3];
  v[0] = a0 * cs.x + a1 * cs.y;
  v[1] = a1 * cs.x - a0 * cs.y;
  v[2] = b0 * cs.z + b1 * cs.w;
  v[3] = b1 * cs.z - b0 * cs.w;
}

// ---------------------------------------------------------------------------------------------
// dQ pass
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256, 2) void attn_bwd_dq_kernel(const u16* __restrict__ qkv, const u16* __restrict__ d_out,
                                                             const float* __restrict__ lse,
                                                             const float* __restrict__ dsum, u16* dqkv,
                                                             const int32_t* cu, int nh, int nkv, int max_qblocks,
                                                             const float* __restrict__ rope_cs) {
  extern __shared__ __attribute__((aligned(16))) char smem[];  // [2 stages][K tile | V tile]
  const int hd = 128;
  const int b = blockIdx.z, h = blockIdx.y;
  const int qb = max_qblocks - 1 - (int)blockIdx.x;
  const int tok0 = cu[b];
  const int T = cu[b + 1] - tok0;
  if (qb * AB_QROWS >= T) return;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int quad = lane >> 4, li = lane & 15;
  const int kvh = h / (nh / nkv);
  const size_t stride = (size_t)(nh + 2 * nkv) * hd;
  const u16* kbase = qkv + (size_t)tok0 * stride + (nh + kvh) * hd;
  const u16* vbase = qkv + (size_t)tok0 * stride + (nh + nkv + kvh) * hd;

  bf16x8 qf[2][4], dof[2][4];
  int qabs[2];
  float lse2[2], dq_row[2];
#pragma unroll
  for (int qt = 0; qt < 2; ++qt) {
    qabs[qt] = qb * AB_QROWS + wave * 32 + qt * 16 + li;
    const int qr = min(qabs[qt], T - 1);
    const u16* qp = qkv + (size_t)(tok0 + qr) * stride + h * hd + quad * 8;
    const u16* dp = d_out + (size_t)(tok0 + qr) * nh * hd + h * hd + quad * 8;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      qf[qt][ks] = *reinterpret_cast<const bf16x8*>(qp + ks * 32);
      dof[qt][ks] = *reinterpret_cast<const bf16x8*>(dp + ks * 32);
    }
    lse2[qt] = lse[(size_t)(tok0 + qr) * nh + h] * 1.4426950408889634f;
    dq_row[qt] = dsum[(size_t)(tok0 + qr) * nh + h];
  }
  floatx4 dqt[2][8];
#pragma unroll
  for (int qt = 0; qt < 2; ++qt)
#pragma unroll
    for (int dt = 0; dt < 8; ++dt) dqt[qt][dt] = floatx4{0.f, 0.f, 0.f, 0.f};

  const int q_last = min(qb * AB_QROWS + AB_QROWS - 1, T - 1);
  const int kb_last = q_last / AB_KB;
  const int wave_q_last = qb * AB_QROWS + wave * 32 + 31;
  const float sl2 = 0.08838834764831845f * 1.4426950408889634f;

  auto stage = [&](int kb, int buf) {
    char* base = smem + buf * AB_STAGE_BYTES;
    ab_stage_tile(kbase, stride, kb * AB_KB, T, base, wave, lane);
    ab_stage_tile(vbase, stride, kb * AB_KB, T, base + AB_TILE_BYTES, wave, lane);
  };
  stage(0, 0);
#pragma unroll
  for (int qt = 0; qt < 2; ++qt) {
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) asm volatile("" ::"v"(qf[qt][ks]), "v"(dof[qt][ks]));
    asm volatile("" ::"v"(lse2[qt]), "v"(dq_row[qt]));
  }
  AB_DMA_LANDED();
  __syncthreads();

  for (int kb = 0; kb <= kb_last; ++kb) {
    const char* Ks = smem + (kb & 1) * AB_STAGE_BYTES;
    const char* Vs = Ks + AB_TILE_BYTES;
    if (kb < kb_last) stage(kb + 1, (kb + 1) & 1);
    if (kb * AB_KB <= wave_q_last) {
      // ---- S^T = K Q^T, then P^T in place
      floatx4 st[2][4];
#pragma unroll
      for (int qt = 0; qt < 2; ++qt)
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) st[qt][nt] = floatx4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < 4; ++ks)
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
          const bf16x8 kf = *reinterpret_cast<const bf16x8*>(Ks + ab_off(nt * 16 + li, ks * 4 + quad));
#pragma unroll
          for (int qt = 0; qt < 2; ++qt)
            st[qt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qf[qt][ks], st[qt][nt], 0, 0, 0);
        }
#pragma unroll
      for (int qt = 0; qt < 2; ++qt)
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int key = kb * AB_KB + nt * 16 + quad * 4 + r;
            const float p = __builtin_amdgcn_exp2f(__builtin_fmaf(st[qt][nt][r], sl2, -lse2[qt]));
            st[qt][nt][r] = (key <= qabs[qt] && qabs[qt] < T) ? p : 0.f;
          }
      // ---- dP^T = V dO^T
      floatx4 dpt[2][4];
#pragma unroll
      for (int qt = 0; qt < 2; ++qt)
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) dpt[qt][nt] = floatx4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < 4; ++ks)
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
          const bf16x8 vf = *reinterpret_cast<const bf16x8*>(Vs + ab_off(nt * 16 + li, ks * 4 + quad));
#pragma unroll
          for (int qt = 0; qt < 2; ++qt)
            dpt[qt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, dof[qt][ks], dpt[qt][nt], 0, 0, 0);
        }
      // ---- dS^T = P^T .* (dP^T - D), packed as the B operand of dQ^T += K^T dS^T
      bf16x8 dsb[2][2];
#pragma unroll
      for (int qt = 0; qt < 2; ++qt)
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
#pragma unroll
          for (int r = 0; r < 4; ++r)
            dsb[qt][nt >> 1][(nt & 1) * 4 + r] = (__bf16)(st[qt][nt][r] * (dpt[qt][nt][r] - dq_row[qt]));
#pragma unroll
      for (int ks2 = 0; ks2 < 2; ++ks2)
#pragma unroll
        for (int dt = 0; dt < 8; ++dt) {
          const bf16x8 kt = ab_read_tr(Ks, ks2, dt, quad, li);
#pragma unroll
          for (int qt = 0; qt < 2; ++qt)
            dqt[qt][dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kt, dsb[qt][ks2], dqt[qt][dt], 0, 0, 0);
        }
    }
    AB_DMA_LANDED();   // this wave's pieces of block kb + 1
    __syncthreads();
  }
  // ---- store scale * dQ: lane owns query row li, d = dt*16 + 4*quad + r
#pragma unroll
  for (int qt = 0; qt < 2; ++qt) {
    if (qabs[qt] < T) {
      u16* op = dqkv + (size_t)(tok0 + qabs[qt]) * stride + h * hd + quad * 4;
#pragma unroll
      for (int dt = 0; dt < 8; ++dt) {
        float v[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = dqt[qt][dt][r] * 0.08838834764831845f;
        ab_unrotate(v, rope_cs, qabs[qt], dt, quad);
        u16x4 o;
#pragma unroll
        for (int r = 0; r < 4; ++r) o[r] = f2bf(v[r]);
        *reinterpret_cast<u16x4*>(op + dt * 16) = o;
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------
// dK / dV pass
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256, 2) void attn_bwd_dkv_kernel(const u16* __restrict__ qkv,
                                                              const u16* __restrict__ d_out,
                                                              const float* __restrict__ lse,
                                                              const float* __restrict__ dsum, u16* dqkv,
                                                              const int32_t* cu, int nh, int nkv,
                                                              const float* __restrict__ rope_cs) {
  extern __shared__ __attribute__((aligned(16))) char smem[];  // [2 stages][Q tile | dO tile], then stats
  float* stats = reinterpret_cast<float*>(smem + 2 * AB_STAGE_BYTES);  // [2 stages][lse2[64] | D[64]]
  const int hd = 128;
  const int b = blockIdx.z, kvh = blockIdx.y, kb = blockIdx.x;
  const int tok0 = cu[b];
  const int T = cu[b + 1] - tok0;
  if (kb * AB_KB >= T) return;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int quad = lane >> 4, li = lane & 15;
  const int rep = nh / nkv;
  const size_t stride = (size_t)(nh + 2 * nkv) * hd;
  const size_t ostride = (size_t)nh * hd;

  // ---- this lane's key: K and V fragments (B operands), d = 32*ks + 8*quad + 0..7
  const int kabs = kb * AB_KB + wave * 16 + li;
  const int kr = min(kabs, T - 1);
  bf16x8 kf[4], vf[4];
  {
    const u16* kp = qkv + (size_t)(tok0 + kr) * stride + (nh + kvh) * hd + quad * 8;
    const u16* vp = qkv + (size_t)(tok0 + kr) * stride + (nh + nkv + kvh) * hd + quad * 8;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      kf[ks] = *reinterpret_cast<const bf16x8*>(kp + ks * 32);
      vf[ks] = *reinterpret_cast<const bf16x8*>(vp + ks * 32);
    }
  }
  floatx4 dkt[8], dvt[8];
#pragma unroll
  for (int dt = 0; dt < 8; ++dt) dkt[dt] = dvt[dt] = floatx4{0.f, 0.f, 0.f, 0.f};
  const float sl2 = 0.08838834764831845f * 1.4426950408889634f;

  const int qb_first = kb, qb_last = (T - 1) / AB_KB;
  const int nqb = qb_last - qb_first + 1;
  const int steps = nqb * rep;  // (query head of the group, query block) pairs, head-major
  // a step's tiles are requested at the top of the step before (LDS-DMA) together with its 64 query rows' statistics
  // (ordinary loads into two registers of wave 0); the statistics are written to LDS at the END of that step, so that the
  // wait hipcc puts in front of the write coincides with the hand-written one for the DMA
  float st_l = 0.f, st_dd = 0.f;
  auto stage = [&](int step, int buf) {
    const int h = kvh * rep + step / nqb, qb = qb_first + step % nqb;
    char* base = smem + buf * AB_STAGE_BYTES;
    ab_stage_tile(qkv + (size_t)tok0 * stride + h * hd, stride, qb * AB_KB, T, base, wave, lane);
    ab_stage_tile(d_out + (size_t)tok0 * ostride + h * hd, ostride, qb * AB_KB, T, base + AB_TILE_BYTES, wave, lane);
    if (tid < 64) {
      const int q = min(qb * AB_KB + tid, T - 1);
      st_l = lse[(size_t)(tok0 + q) * nh + h];   // (first use of either value: commit_stats)
      st_dd = dsum[(size_t)(tok0 + q) * nh + h];
    }
  };
  auto commit_stats = [&](int buf) {
    __builtin_amdgcn_sched_barrier(0);
    if (tid < 64) {
      stats[buf * 128 + tid] = st_l * 1.4426950408889634f;
      stats[buf * 128 + 64 + tid] = st_dd;
    }
  };
  stage(0, 0);
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) asm volatile("" ::"v"(kf[ks]), "v"(vf[ks]));
  commit_stats(0);
  AB_DMA_LANDED();
  __syncthreads();

  for (int step = 0; step < steps; ++step) {
    const int qb = qb_first + step % nqb;
    const char* Qs = smem + (step & 1) * AB_STAGE_BYTES;
    const char* Os = Qs + AB_TILE_BYTES;
    const float* st_lse = stats + (step & 1) * 128;
    const float* st_d = st_lse + 64;
    if (step + 1 < steps) stage(step + 1, (step + 1) & 1);
    // ---- S = Q K^T and dP = dO V^T : rows = queries mt*16 + 4*quad + r, column = this lane's key
    floatx4 s[4], dp[4];
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) s[mt] = dp[mt] = floatx4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
#pragma unroll
      for (int mt = 0; mt < 4; ++mt) {
        const bf16x8 qa = *reinterpret_cast<const bf16x8*>(Qs + ab_off(mt * 16 + li, ks * 4 + quad));
        const bf16x8 oa = *reinterpret_cast<const bf16x8*>(Os + ab_off(mt * 16 + li, ks * 4 + quad));
        s[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qa, kf[ks], s[mt], 0, 0, 0);
        dp[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(oa, vf[ks], dp[mt], 0, 0, 0);
      }
    // ---- P and dS, packed as B operands (K index = query row)
    bf16x8 pb[2], dsb[2];
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
      const floatx4 l4 = *reinterpret_cast<const floatx4*>(st_lse + mt * 16 + quad * 4);
      const floatx4 d4 = *reinterpret_cast<const floatx4*>(st_d + mt * 16 + quad * 4);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int q = qb * AB_KB + mt * 16 + quad * 4 + r;
        float p = __builtin_amdgcn_exp2f(__builtin_fmaf(s[mt][r], sl2, -l4[r]));
        p = (kabs <= q && q < T) ? p : 0.f;
        pb[mt >> 1][(mt & 1) * 4 + r] = (__bf16)p;
        dsb[mt >> 1][(mt & 1) * 4 + r] = (__bf16)(p * (dp[mt][r] - d4[r]));
      }
    }
    // ---- dV^T += dO^T P,  dK^T += Q^T dS
#pragma unroll
    for (int ks2 = 0; ks2 < 2; ++ks2)
#pragma unroll
      for (int dt = 0; dt < 8; ++dt) {
        const bf16x8 ot = ab_read_tr(Os, ks2, dt, quad, li);
        const bf16x8 qt = ab_read_tr(Qs, ks2, dt, quad, li);
        dvt[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ot, pb[ks2], dvt[dt], 0, 0, 0);
        dkt[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qt, dsb[ks2], dkt[dt], 0, 0, 0);
      }
    if (step + 1 < steps) commit_stats((step + 1) & 1);
    AB_DMA_LANDED();   // this wave's pieces of step + 1
    __syncthreads();
  }
  if (kabs < T) {
    u16* kp = dqkv + (size_t)(tok0 + kabs) * stride + (nh + kvh) * hd + quad * 4;
    u16* vp = dqkv + (size_t)(tok0 + kabs) * stride + (nh + nkv + kvh) * hd + quad * 4;
#pragma unroll
    for (int dt = 0; dt < 8; ++dt) {
      u16x4 ok, ov;
      float v[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) v[r] = dkt[dt][r] * 0.08838834764831845f;
      ab_unrotate(v, rope_cs, kabs, dt, quad);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        ok[r] = f2bf(v[r]);
        ov[r] = f2bf(dvt[dt][r]);
      }
      *reinterpret_cast<u16x4*>(kp + dt * 16) = ok;
      *reinterpret_cast<u16x4*>(vp + dt * 16) = ov;
    }
  }
}

// =============================================================================================
int lr_launch_attention_bwd(const u16* qkv, const u16* out, const u16* d_out, const float* lse, u16* dqkv, float* dsum,
                            float* dkv32, const int32_t* cu, const int32_t* cu_host, int B, int n_tok, int nh, int nkv,
                            int hd, int variant, hipStream_t st, const int32_t* tok_pos, const float* rope_cs) {
  if (n_tok <= 0 || B <= 0) return LR_OK;
  if (nh % nkv != 0) LR_FAIL(LR_EINVAL, "attention backward: num_heads %d not a multiple of num_kv_heads %d", nh, nkv);
  if (hd > 256) LR_FAIL(LR_EUNSUPPORTED, "attention backward: head_dim %d > 256", hd);
  if (variant == 0 || variant == 3) variant = (hd == 128) ? 2 : 1;
  int rc = lr_launch_rowdot(out, d_out, n_tok, nh, hd, dsum, st);
  if (rc) return rc;
  double work = 0;  // 5 causal tile products of 2*T^2/2*hd flops each
  int maxT = 0;
  for (int b = 0; b < B; ++b) {
    const double T = cu_host[b + 1] - cu_host[b];
    work += 10.0 * nh * hd * (T * (T + 1) / 2);
    maxT = max(maxT, cu_host[b + 1] - cu_host[b]);
  }
  LrProfScope prof(variant >= 2 ? LR_PROF_ATTN_MFMA : LR_PROF_ATTN_GENERIC, work, st);
  if (variant == 2) {
    if (hd != 128) LR_FAIL(LR_EUNSUPPORTED, "attention backward variant 2 needs head_dim 128 (got %d)", hd);
    static bool lds_set_dq[LR_MAX_DEVICES] = {}, lds_set_dkv[LR_MAX_DEVICES] = {};
    const size_t dkv_lds = 2 * AB_STAGE_BYTES + 2 * 128 * sizeof(float);
    if (int rc = lr_ensure_dynamic_lds(reinterpret_cast<const void*>(attn_bwd_dq_kernel), 2 * AB_STAGE_BYTES, lds_set_dq))
      return rc;
    if (int rc = lr_ensure_dynamic_lds(reinterpret_cast<const void*>(attn_bwd_dkv_kernel), (int)dkv_lds, lds_set_dkv))
      return rc;
    const int mq = (maxT + AB_QROWS - 1) / AB_QROWS, mk = (maxT + AB_KB - 1) / AB_KB;
    hipLaunchKernelGGL(attn_bwd_dq_kernel, dim3(mq, nh, B), dim3(256), 2 * AB_STAGE_BYTES, st, qkv, d_out, lse, dsum,
                       dqkv, cu, nh, nkv, mq, rope_cs);
    LR_CHECK_LAUNCH("attn_bwd_dq_kernel");
    hipLaunchKernelGGL(attn_bwd_dkv_kernel, dim3(mk, nkv, B), dim3(256), dkv_lds, st, qkv, d_out, lse, dsum, dqkv, cu,
                       nh, nkv, rope_cs);
    LR_CHECK_LAUNCH("attn_bwd_dkv_kernel");
  } else if (variant == 1) {
    if (!dkv32) LR_FAIL(LR_EINVAL, "attention backward (generic): null fp32 scratch");
    const int kvcols = 2 * nkv * hd;
    LR_CHECK_HIP(hipMemsetAsync(dkv32, 0, (size_t)n_tok * kvcols * sizeof(float), st));
    hipLaunchKernelGGL(attn_bwd_generic_kernel, dim3((n_tok * nh + 3) / 4), dim3(256), 0, st, qkv, d_out, lse, dsum,
                       dqkv, dkv32, cu, B, n_tok, nh, nkv, hd);
    LR_CHECK_LAUNCH("attn_bwd_generic_kernel");
    const size_t total = (size_t)n_tok * kvcols;
    hipLaunchKernelGGL(attn_bwd_kv_to_bf16_kernel, dim3((unsigned)min((size_t)4096, (total + 255) / 256)), dim3(256), 0,
                       st, dkv32, dqkv, n_tok, nh * hd, kvcols);
    LR_CHECK_LAUNCH("attn_bwd_kv_to_bf16_kernel");
    if (rope_cs) {
      if (!tok_pos) LR_FAIL(LR_EINVAL, "attention backward: rotary table without token positions");
      rc = lr_launch_rope_bwd(dqkv, n_tok, (nh + 2 * nkv) * hd, (nh + nkv) * hd, hd, tok_pos, rope_cs, st);
      if (rc) return rc;
    }
  } else {
    LR_FAIL(LR_EINVAL, "attention backward: unknown variant %d", variant);
  }
  return LR_OK;
}
