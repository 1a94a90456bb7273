// llama_gemm.hip -- bf16 GEMMs of the Llama prefill on gfx950 (CDNA4 MFMA, LDS-tiled, no BLAS).
//   C[M][N] = A[M][K] * B[N][K]^T   (A = packed activations, B = HF nn.Linear weight [out][in])
// with fused epilogues: plain store, residual add (o_proj / down_proj), SwiGLU (gate/up).
//
// Replaces (reference): every Linear of HF LlamaModel reached from model/llm.py:89-100 -- in the
// reference bitsandbytes NF4 dequant-GEMMs with bf16 compute (train_ranker.py:49-62); here bf16
// weights (LoRA merged offline) on v_mfma_f32_16x16x32_bf16.
//
// Two kernels:
//  gemm_generic_kernel : 64x64x32 tile, any M/N/K (bounds-checked) -- tiny test models, odd shapes,
//                        and the in-library reference for the fast kernel.
//  gemm256rb_kernel    : 256x256x64 tile, 8 wave64 (2 x 4), each wave 128x64 = 8x4 MFMA tiles.
//     * both operands stream global -> LDS with global_load_lds_dwordx4 (no VGPR staging): every
//       wave-instruction moves 8 rows x 128 B = full cache lines; the LDS image is lane-linear
//       (the DMA's constraint) and the 16-byte chunk a lane FETCHES is XOR-permuted
//       (chunk ^ ((row>>1)&7)), so the ds_read_b128 fragment reads are bank-conflict free.
//     * two 64 KiB K-tile stages; a ping-pong schedule between the two wave groups of a workgroup
//       (described at the kernel) keeps the DMA queue and the matrix pipe busy together.
//     * MFMA operands are swapped (D = B_frag x A_frag) so each lane ends up with 4 CONSECUTIVE
//       output columns -> 8-byte epilogue accesses, and gate/up of one SwiGLU output meet in a lane.
//     * workgroup -> tile map: bijective XCD remap (each XCD's L2 sees a compact set of tiles)
//       followed by a grouped (8 M-tiles) raster so neighbours share A and B panels.
#include <stdlib.h>

#include "llama_kernels.h"
#include "lr_profile.h"

typedef unsigned short u16;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef u16 u16x4 __attribute__((ext_vector_type(4)));
typedef u16 u16x8 __attribute__((ext_vector_type(8)));

// Rotary embedding fused into the QKV projection's epilogue. q/k head rows of wqkv are stored
// pair-interleaved (row 2i = dim i, row 2i+1 = dim i + head_dim/2), so a lane's 4 consecutive
// output columns are two complete rotation pairs.
struct RopeArgs {
  const int32_t* tok_pos;  // [M] position of each packed token inside its prompt
  const float* cs;         // [max_T][head_dim/2][2] (cos, sin), bf16-valued
  int head_dim;
  int rot_cols;            // columns [0, rot_cols) are q and k heads; the rest (v) is stored as is
  // Folded RMSNorm (ROPE and SWIGLU epilogues): the GEMM ran on the un-normalised residual rows against weights with
  // the norm weight multiplied in (lr_fold_norm_bf16), and the accumulator of row m is scaled by rstd[m] =
  // 1 / sqrt(mean(x_m^2) + eps) here:  (x_m * rstd_m * w) . W_j  ==  rstd_m * (x_m . (W_j * w)).  nullptr: no scaling.
  const float* row_scale;
  // The same table as packed bf16 pairs [max_T][head_dim/2] (cos | sin << 16), or nullptr. With it (head_dim 128, rot_cols a
  // multiple of 256) the 256-tile kernel stages the tile's 256 x 64 entries through LDS after its K loop: 64 KB of
  // whole-row DMA instead of 262 KB of 64-byte-per-row register loads (16 rows x 64 B per wave-instruction cost a CU
  // three times a full-line access: tools/diag/store_rate.hip; the rotary epilogue took 20-24 k cycles of a 186 k-cycle tile).
  const unsigned* cs16;
};

// ---- shared epilogue: lane holds 4 consecutive columns of one row ----------------------------
// pre_rs / pre_pos / pre_cs: the row's folded-norm scale, token position, (cos, sin) pairs when the caller fetched them
// ahead (gemm256rb_kernel batches all reads of its epilogue in front of the stores); nullptr: fetched here. r: the residual's 4 values (RESIDUAL only).
// FOLD: -1 = scale by the folded-norm rstd when rope.row_scale is set (run-time check), 0 = never, 1 = always (the
// 256-tile kernel is instantiated per case: as a run-time select the scaling cost a multiply and a v_cndmask per value
// on the un-folded default path too).
template <int EPI, int FOLD = -1>
__device__ __forceinline__ u16x4 epi_value4(floatx4 v, floatx4 up, u16x4 r, const RopeArgs& rope, int row, int col,
                                            const float* pre_rs, const int* pre_pos, const float4* pre_cs = nullptr) {
  u16x4 o;
  if ((EPI == LR_EPI_ROPE || EPI == LR_EPI_SWIGLU) && (FOLD == 1 || (FOLD == -1 && rope.row_scale))) {
    const float rs = pre_rs ? *pre_rs : rope.row_scale[row];
    v *= rs;
    up *= rs;
  }
  if (EPI == LR_EPI_ROPE) {
    float x[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) x[j] = bf2f(f2bf(v[j]));  // q/k are bf16 tensors before the rotation
    if (col < rope.rot_cols) {
      const int half = rope.head_dim >> 1;
      const int i0 = (col % rope.head_dim) >> 1;
      float4 t;
      if (pre_cs) {
        t = *pre_cs;
      } else {
        const int pos = pre_pos ? *pre_pos : rope.tok_pos[row];
        t = *reinterpret_cast<const float4*>(rope.cs + ((size_t)pos * half + i0) * 2);
      }
      o[0] = f2bf(x[0] * t.x - x[1] * t.y);
      o[1] = f2bf(x[1] * t.x + x[0] * t.y);
      o[2] = f2bf(x[2] * t.z - x[3] * t.w);
      o[3] = f2bf(x[3] * t.z + x[2] * t.w);
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j) o[j] = f2bf(x[j]);
    }
  } else if (EPI == LR_EPI_STORE) {
#pragma unroll
    for (int j = 0; j < 4; ++j) o[j] = f2bf(v[j]);
  } else if (EPI == LR_EPI_RESIDUAL) {
#pragma unroll
    for (int j = 0; j < 4; ++j) o[j] = f2bf(bf2f(f2bf(v[j])) + bf2f(r[j]));
  } else {
#pragma unroll
    for (int j = 0; j < 4; ++j) o[j] = swiglu_bf16(bf2f(f2bf(v[j])), bf2f(f2bf(up[j])));
  }
  return o;
}

template <int EPI>
__device__ __forceinline__ void epi_store4(floatx4 v, floatx4 up, u16* C, const u16* R, size_t off,
                                           const RopeArgs& rope = RopeArgs{nullptr, nullptr, 0, 0, nullptr, nullptr}, int row = 0, int col = 0,
                                           const float* pre_rs = nullptr, const int* pre_pos = nullptr) {
  if (EPI == LR_EPI_PARTIAL) {  // C is the fp32 partial plane of this split
    *reinterpret_cast<floatx4*>(reinterpret_cast<float*>(C) + off) = v;
    return;
  }
  u16x4 r = u16x4{0, 0, 0, 0};
  if (EPI == LR_EPI_RESIDUAL) r = *reinterpret_cast<const u16x4*>(R + off);
  *reinterpret_cast<u16x4*>(C + off) = epi_value4<EPI>(v, up, r, rope, row, col, pre_rs, pre_pos);
}

// =============================================================================================
// generic kernel
// =============================================================================================
#define GG_BM 64
#define GG_BN 64
#define GG_BK 32
#define GG_LD 40  // padded row length (elements): 80-byte rows keep 16-byte alignment

template <int EPI>
__global__ __launch_bounds__(256) void gemm_generic_kernel(const u16* __restrict__ A,
                                                           const u16* __restrict__ B, u16* C,
                                                           const u16* R, int M, int N, int K, RopeArgs rope) {
  __shared__ __attribute__((aligned(16))) u16 As[GG_BM * GG_LD];
  __shared__ __attribute__((aligned(16))) u16 Bs[GG_BN * GG_LD];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int m0 = blockIdx.y * GG_BM, n0 = blockIdx.x * GG_BN;
  const bool vec_ok = (K % 8 == 0);
  floatx4 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = floatx4{0.f, 0.f, 0.f, 0.f};

  const int lr = tid >> 2, lc = (tid & 3) * 8;  // staging: row, first k of an 8-element chunk
  for (int k0 = 0; k0 < K; k0 += GG_BK) {
    u16x8 va, vb;
#pragma unroll
    for (int j = 0; j < 8; ++j) va[j] = vb[j] = 0;
    {
      int gr = m0 + lr, gk = k0 + lc;
      if (gr < M) {
        if (vec_ok && gk + 8 <= K) va = *reinterpret_cast<const u16x8*>(A + (size_t)gr * K + gk);
        else
          for (int j = 0; j < 8; ++j)
            if (gk + j < K) va[j] = A[(size_t)gr * K + gk + j];
      }
      int gn = n0 + lr;
      if (gn < N) {
        if (vec_ok && gk + 8 <= K) vb = *reinterpret_cast<const u16x8*>(B + (size_t)gn * K + gk);
        else
          for (int j = 0; j < 8; ++j)
            if (gk + j < K) vb[j] = B[(size_t)gn * K + gk + j];
      }
    }
    __syncthreads();  // previous tile fully consumed
    *reinterpret_cast<u16x8*>(As + lr * GG_LD + lc) = va;
    *reinterpret_cast<u16x8*>(Bs + lr * GG_LD + lc) = vb;
    __syncthreads();
    bf16x8 af[2], bfr[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      af[t] = *reinterpret_cast<const bf16x8*>(As + (wm * 32 + t * 16 + (lane & 15)) * GG_LD + (lane >> 4) * 8);
      bfr[t] = *reinterpret_cast<const bf16x8*>(Bs + (wn * 32 + t * 16 + (lane & 15)) * GG_LD + (lane >> 4) * 8);
    }
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int nt = 0; nt < 2; ++nt)
        acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[nt], af[mt], acc[mt][nt], 0, 0, 0);
  }
  // epilogue (element-wise, bounds-checked): lane holds row m = lane&15, cols 4*(lane>>4)+j
#pragma unroll
  for (int mt = 0; mt < 2; ++mt) {
    int row = m0 + wm * 32 + mt * 16 + (lane & 15);
    if (row >= M) continue;
    if (EPI == LR_EPI_ROPE) {
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) {
        const int col = n0 + wn * 32 + nt * 16 + (lane >> 4) * 4;
        if (col + 3 < N) {
          epi_store4<EPI>(acc[mt][nt], acc[mt][nt], C, R, (size_t)row * N + col, rope, row, col);
        } else {
          for (int j = 0; j < 4; ++j)  // only hit when N%4 != 0 (no rope cols there)
            if (col + j < N) C[(size_t)row * N + col + j] = f2bf(acc[mt][nt][j] * (rope.row_scale ? rope.row_scale[row] : 1.0f));
        }
      }
    } else if (EPI == LR_EPI_SWIGLU) {
      int ocol = (n0 + wn * 32) / 2 + (lane >> 4) * 4;
      const float rs = rope.row_scale ? rope.row_scale[row] : 1.0f;
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (n0 + wn * 32 + 16 + (lane >> 4) * 4 + j < N)
          C[(size_t)row * (N / 2) + ocol + j] =
              swiglu_bf16(bf2f(f2bf(acc[mt][0][j] * rs)), bf2f(f2bf(acc[mt][1][j] * rs)));
    } else {
#pragma unroll
      for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          int col = n0 + wn * 32 + nt * 16 + (lane >> 4) * 4 + j;
          if (col < N) {
            size_t off = (size_t)row * N + col;
            float v = bf2f(f2bf(acc[mt][nt][j]));
            C[off] = (EPI == LR_EPI_RESIDUAL) ? f2bf(v + bf2f(R[off])) : f2bf(acc[mt][nt][j]);
          }
        }
    }
  }
}

// =============================================================================================
// 256 x 256 x 64 kernel
// =============================================================================================
#define G2_STAGE_BYTES 65536  // A tile 32 KiB + B tile 32 KiB
#define G2_GROUP_M 8

__device__ __forceinline__ void glds16(const void* gsrc, void* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                   (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

#define PP_WAIT_VM(n) asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory")
#define PP_BARRIER()                        \
  do {                                      \
    __builtin_amdgcn_sched_barrier(0);      \
    __builtin_amdgcn_s_barrier();           \
    __builtin_amdgcn_sched_barrier(0);      \
  } while (0)

// =============================================================================================
// 256 x 256 x 64 kernel, ping-pong pipeline, balanced fragment reads + region-recycling DMA (variant 4)
// =============================================================================================
// The K tile is processed in 4 phases (one 64x32 quadrant of the wave's 128x64 output per phase, 16 MFMAs each):
//     LOAD_p : ds_read the quadrant's new fragments + issue 2 global->LDS DMAs + counted s_waitcnt | s_barrier |
//     MFMA_p : 16 x v_mfma_f32_16x16x32_bf16                                                       | s_barrier |
// The two wave groups (wm = 0 / 1; one wave of each per SIMD) run ONE BARRIER APART, so on every SIMD one wave's
// MFMA segment overlaps the other wave's LOAD segment. s_memtime stamps on the first ping-pong version (DESIGN.md
// section 4, experiment log) showed every LOAD segment longer than the partner's 16-MFMA segment -- 12 ds_read_b128
// land in ~380 cycles, a pair of DMAs takes ~190 cycles to issue, DMA issue -> landed ~ 1 us. Hence this schedule
//  * reads (8,4,8,4) fragments per phase instead of (12,4,8,0): the next tile's B(nh0) fragments are
//    read in LOAD_3 of the current tile into a second register set (tiles alternate b0x / b0y);
//  * recycles each LDS region two phases after its last read, so every DMA is issued SIX phases
//    (~1.5 K tiles) before its first reader waits for it:
//        LOAD_0(t): R2 (A rows mh1) of tile t+1      LOAD_1(t): R0B (B rows nh0) of tile t+2
//        LOAD_2(t): R0A (A rows mh0) of tile t+2     LOAD_3(t): R1 (B rows nh1) of tile t+2
//    a reader needs "all but my newest 10" complete (uniform s_waitcnt vmcnt(10)).
// Hazards: a region is overwritten only after both wave groups retired their reads of it (issue at
// LOAD_{p+2} for reads of LOAD_p: two barriers later for either group); every wait is followed by a
// barrier that each reader passes before its read.
// Split-K (EPI == LR_EPI_PARTIAL): blockIdx.y = split s works on K tiles [s*T/S, (s+1)*T/S) and stores its
// fp32 partial plane at ((float*)C)[s][M][N]; splitk_reduce_kernel sums the planes in order and applies the
// real epilogue.
// Diagnostic stamps: STAMP = true is instantiated only in a -DLR_EXPERIMENTS build (make EXPERIMENTS=1, then
// LR_GEMM_STAMPS=1 at run time; tools/gemm_stamps.py); the product library holds no stamping code. Per workgroup (wave 0
// and wave 4 = one wave of each ping-pong group): s_memtime at kernel entry, after the prologue, after the K loop, after
// the epilogue's last store is ISSUED, and after those stores have completed (vmcnt(0)); s_memrealtime (100 MHz) at
// entry and exit for the clock; the XCC id.
#define GEMM_STAMP_SLOTS 8
#ifndef LR_EXPERIMENTS
#define GEMM_STAMP(slot)
#define GEMM_STAMP_REAL(slot)
#define GEMM_STAMP_END()
#else
__device__ unsigned long long g_gemm_stamps[16384 * 2 * GEMM_STAMP_SLOTS];
#define GEMM_STAMP(slot)                                                                \
  if (STAMP && (wave & 3) == 0 && lane == 0 && blockIdx.x < 16384 && blockIdx.y == 0) { \
    unsigned long long t_;                                                              \
    __builtin_amdgcn_sched_barrier(0);                                                  \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");          \
    __builtin_amdgcn_sched_barrier(0);                                                  \
    g_gemm_stamps[((size_t)blockIdx.x * 2 + (wave >> 2)) * GEMM_STAMP_SLOTS + (slot)] = t_; \
  }
#define GEMM_STAMP_REAL(slot)                                                           \
  if (STAMP && (wave & 3) == 0 && lane == 0 && blockIdx.x < 16384 && blockIdx.y == 0) { \
    unsigned long long t_;                                                              \
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");      \
    g_gemm_stamps[((size_t)blockIdx.x * 2 + (wave >> 2)) * GEMM_STAMP_SLOTS + (slot)] = t_; \
  }
#define GEMM_STAMP_END()                                                                       \
  if (STAMP) {                                                                                 \
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                           \
    GEMM_STAMP(4)                                                                              \
    GEMM_STAMP_REAL(6)                                                                         \
    if ((wave & 3) == 0 && lane == 0 && blockIdx.x < 16384 && blockIdx.y == 0) {              \
      unsigned xcc;                                                                            \
      asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));                       \
      g_gemm_stamps[((size_t)blockIdx.x * 2 + (wave >> 2)) * GEMM_STAMP_SLOTS + 7] = xcc & 0xf; \
    }                                                                                          \
  }
#endif

template <int EPI, bool FOLD = false, bool STAMP = false>
__global__ __launch_bounds__(512) void gemm256rb_kernel(const u16* __restrict__ A,
                                                        const u16* __restrict__ B, u16* C,
                                                        const u16* R, int M, int N, int K, int group_m,
                                                        RopeArgs rope) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;
  GEMM_STAMP(0)
  GEMM_STAMP_REAL(5)
  int kt_first = 0, nkt = K >> 6;
  if (EPI == LR_EPI_PARTIAL) {
    const int s = blockIdx.y, S = gridDim.y, T = K >> 6;
    kt_first = (int)((long long)s * T / S);
    nkt = (int)((long long)(s + 1) * T / S) - kt_first;
    C = reinterpret_cast<u16*>(reinterpret_cast<float*>(C) + (size_t)s * M * N);
  }

  const int tilesM = (M + 255) >> 8, tilesN = N >> 8;
  const int nwg = tilesM * tilesN;
  int id;
  {
    const int bid = blockIdx.x;
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
    id = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  int tm, tn;
  {
    const int per_group = group_m * tilesN;
    const int g = id / per_group, rem = id % per_group;
    const int first_m = g * group_m;
    const int gsz = min(group_m, tilesM - first_m);
    tm = first_m + rem % gsz;
    tn = rem / gsz;
  }
  const int m0 = tm << 8, n0 = tn << 8;

  // ---- DMA pieces of this wave (8 rows x 128 B), two per region; q = 2*wave + j in 0..15
  //  region 0 = R0A: A rows (q>>3)*128 + (q&7)*8          region 1 = R0B: B rows (q>>2)*64 + (q&3)*8
  //  region 2 = R1 : B rows (q>>2)*64 + 32 + (q&3)*8      region 3 = R2 : A rows (q>>3)*128 + 64 + (q&7)*8
  const int srow = lane >> 3, spos = lane & 7;
  const char* src[4][2];
  int ldsoff[4][2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int q = 2 * wave + j;
    const int ra = (q >> 3) * 128 + (q & 7) * 8;
    const int rb = (q >> 2) * 64 + (q & 3) * 8;
    const int rows[4] = {ra, rb, rb + 32, ra + 64};
    const bool isA[4] = {true, false, false, true};
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int row = rows[t] + srow;
      const int chunk = spos ^ ((row >> 1) & 7);
      if (isA[t]) {
        const int arow = min(m0 + row, M - 1);
        src[t][j] = reinterpret_cast<const char*>(A) + ((size_t)arow * K) * 2 + chunk * 16 + (size_t)kt_first * 128;
        ldsoff[t][j] = rows[t] * 128;
      } else {
        src[t][j] = reinterpret_cast<const char*>(B) + ((size_t)(n0 + row) * K) * 2 + chunk * 16 + (size_t)kt_first * 128;
        ldsoff[t][j] = 32768 + rows[t] * 128;
      }
    }
  }
  // DMA both pieces of region `reg` of K tile `tile` into stage buffer tile&1
#define RB_DMA(reg, tile)                                                                         \
  do {                                                                                            \
    char* dst_ = smem + ((tile)&1) * G2_STAGE_BYTES;                                              \
    glds16(src[(reg)][0] + (size_t)(tile)*128, dst_ + ldsoff[(reg)][0]);                          \
    glds16(src[(reg)][1] + (size_t)(tile)*128, dst_ + ldsoff[(reg)][1]);                          \
  } while (0)

  const int frow = lane & 15;
  const int fsw = frow >> 1;
  const int fo0 = frow * 128 + (((lane >> 4) ^ fsw) << 4);
  const int fo1 = frow * 128 + (((4 + (lane >> 4)) ^ fsw) << 4);
  const int a_base = wm * 128 * 128;
  const int b_base = 32768 + wn * 64 * 128;

  floatx4 acc[8][4];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = floatx4{0.f, 0.f, 0.f, 0.f};

  bf16x8 afr[8], b0x[4], b0y[4], b1[4];

#define RB_LOAD_A(buf, mh)                                                                            \
  _Pragma("unroll") for (int mt = 0; mt < 4; ++mt) {                                                  \
    afr[mt] = *reinterpret_cast<const bf16x8*>((buf) + a_base + ((mh)*64 + mt * 16) * 128 + fo0);     \
    afr[4 + mt] = *reinterpret_cast<const bf16x8*>((buf) + a_base + ((mh)*64 + mt * 16) * 128 + fo1); \
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
  _Pragma("unroll") for (int mt = 0; mt < 4; ++mt)                                                    \
  _Pragma("unroll") for (int nt = 0; nt < 2; ++nt)                                                    \
    acc[(mh)*4 + mt][(nh)*2 + nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(                          \
        bfrag[ks * 2 + nt], afr[ks * 4 + mt], acc[(mh)*4 + mt][(nh)*2 + nt], 0, 0, 0);
#define RB_WAIT(steady)                               \
  if (steady) { PP_WAIT_VM(10); } else { PP_WAIT_VM(0); }

  // One K tile; b0cur holds this tile's B(nh0) fragments (read during the previous tile's LOAD_3 or the
  // prologue), b0nxt receives the next tile's.
#define RB_TILE(kt, b0cur, b0nxt)                                                                     \
  {                                                                                                   \
    const char* cur = smem + ((kt)&1) * G2_STAGE_BYTES;                                               \
    const char* nxt = smem + (((kt) + 1) & 1) * G2_STAGE_BYTES;                                       \
    const bool more = (kt) + 1 < nkt;                                                                 \
    const bool more2 = (kt) + 2 < nkt;                                                                \
    /* phase 0: quadrant (0,0) */                                                                     \
    RB_LOAD_A(cur, 0)                                                                                 \
    if (more) RB_DMA(3, (kt) + 1);                                                                    \
    RB_WAIT(more2)                                                                                    \
    PP_BARRIER();                                                                                     \
    RB_MFMA(b0cur, 0, 0)                                                                              \
    PP_BARRIER();                                                                                     \
    /* phase 1: quadrant (0,1) */                                                                     \
    RB_LOAD_B(b1, cur, 1)                                                                             \
    if (more2) RB_DMA(1, (kt) + 2);                                                                   \
    RB_WAIT(more2)                                                                                    \
    PP_BARRIER();                                                                                     \
    RB_MFMA(b1, 0, 1)                                                                                 \
    PP_BARRIER();                                                                                     \
    /* phase 2: quadrant (1,1) */                                                                     \
    RB_LOAD_A(cur, 1)                                                                                 \
    if (more2) RB_DMA(0, (kt) + 2);                                                                   \
    RB_WAIT(more2)                                                                                    \
    PP_BARRIER();                                                                                     \
    RB_MFMA(b1, 1, 1)                                                                                 \
    PP_BARRIER();                                                                                     \
    /* phase 3: quadrant (1,0); reads the next tile's B(nh0) */                                       \
    if (more) { RB_LOAD_B(b0nxt, nxt, 0) }                                                            \
    if (more2) RB_DMA(2, (kt) + 2);                                                                   \
    RB_WAIT(more2)                                                                                    \
    PP_BARRIER();                                                                                     \
    RB_MFMA(b0cur, 1, 0)                                                                              \
    PP_BARRIER();                                                                                     \
  }

  // Folded RMSNorm: the tile's 256 row scales go to LDS (behind the two stage buffers) in front of the K loop, one per thread
  // of the first four waves; the epilogue reads its eight from there. Fetched in the epilogue they were a dependent global load
  // that nothing hides at one workgroup per CU (round 3: GEMMs 1 370 -> 1 347 TF/s).
  constexpr bool has_rs = FOLD && (EPI == LR_EPI_ROPE || EPI == LR_EPI_SWIGLU);
  float* const rs_lds = reinterpret_cast<float*>(smem + 2 * G2_STAGE_BYTES);
  if (has_rs && tid < 256) rs_lds[tid] = rope.row_scale[min(m0 + tid, M - 1)];   // (visible behind the K loop's barriers)
  // ---- prologue: all of tile 0, then R0B, R0A, R1 of tile 1 (steady-state issue order)
#pragma unroll
  for (int reg = 0; reg < 4; ++reg) RB_DMA(reg, 0);
  if (nkt > 1) {
    RB_DMA(1, 1);
    RB_DMA(0, 1);
    RB_DMA(2, 1);
    PP_WAIT_VM(6);
  } else {
    PP_WAIT_VM(0);
  }
  PP_BARRIER();
  RB_LOAD_B(b0x, smem, 0)
  GEMM_STAMP(1)
  if (wm == 1) {
    // Static priority for the second-dispatched half (waves 4-7), NO per-segment s_setprio flips around the MFMA runs
    // (round 1-2 raised the priority around every 16-MFMA segment): within a SIMD the two waves are arbitrated by priority,
    // then age, and the younger wave loses every segment start (MI355X_MICROARCH.md, two waves per SIMD, item 4). Same-box
    // A/B, three interleaved rounds (gpurun_out/abprio): 149.1 -> 149.8 users/s, GEMMs 1414 -> 1420 TF/s, qkv 1370 -> 1384.
    __builtin_amdgcn_s_setprio(1);
    PP_BARRIER();  // group 1 runs one barrier behind group 0
  }

  for (int kt = 0; kt < nkt; kt += 2) {
    RB_TILE(kt, b0x, b0y)
    if (kt + 1 < nkt) RB_TILE(kt + 1, b0y, b0x)
  }
  if (wm == 0) PP_BARRIER();  // balance group 1's extra barrier
  GEMM_STAMP(2)
#undef RB_DMA
#undef RB_LOAD_A
#undef RB_LOAD_B
#undef RB_MFMA
#undef RB_WAIT
#undef RB_TILE

  // A lane holds 4 consecutive columns (8 bytes) of its row per 16-column tile. v_permlane16_swap on the packed pairs of
  // neighbouring tiles leaves even lane quads with 8 consecutive columns of the first tile and odd quads with 8 of the
  // second: 16-byte stores (and residual loads: the swap is its own inverse, so the 16 bytes read at the store position
  // swap back into each lane's own columns). Partners share lane & 15, i.e. the row: the row guard never splits a pair.
  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
  const int ldc = (EPI == LR_EPI_SWIGLU) ? (N >> 1) : N;
  const int quad = lane >> 4;
  // Residual: all 16 reads of the lane are requested before the first store. Written load -> use -> store per pair, hipcc
  // keeps that order (R may alias C) and every read then waits, behind vmcnt(0), for the previous pair's store as well.
  // A lane reads exactly the 16 bytes it later writes, so taking the reads first is safe for the in-place call too.
  // Whole cache lines per wave-instruction. After the swap a lane holds, per 16-row tile, two 16-byte pieces of ITS row:
  // bytes k*64 + (quad&1)*32 + (quad>>1)*16 of the wave's 128-byte strip, k = 0, 1 -- a store of piece k writes 16 rows x
  // 64 B, half lines, which a CU moves at 13-15 B/clk against 41-44 B/clk for 8 rows x 128 B (tools/diag/store_rate.hip:
  // 8.8-9.9 k cycles per 128 KB tile against 3.0-3.2 k; loads 8.6 k against 5.8 k). So lanes li and li ^ 8 of a 16-lane row
  // trade one piece (DPP row_ror:8): instruction j then carries rows 8 j + (li & 7) in full -- the lower eight lanes hold
  // bytes 0..63 (k = 0), the upper eight bytes 64..127 (k = 1). The residual is READ in that layout and traded back.
  const bool lo8 = (lane & 8) == 0;
  const int hrow = lane & 7;                                            // row of the 8-row half this lane stores / loads
  const int hcol = (lo8 ? 0 : 32) + (quad & 1) * 16 + (quad >> 1) * 8;  // its first column inside the wave's 64
  auto trade = [&](u32x4& p0, u32x4& p1) {   // (own row: k = 0, k = 1)  <->  (row hrow: my half, row 8 + hrow: my half)
    u32x4 z, w;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      z[i] = lo8 ? p1[i] : p0[i];
      w[i] = (unsigned)__builtin_amdgcn_update_dpp(0, (int)z[i], 0x128 /* row_ror:8 */, 0xf, 0xf, false);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const unsigned a_ = lo8 ? p0[i] : w[i], b_ = lo8 ? w[i] : p1[i];
      p0[i] = a_;
      p1[i] = b_;
    }
  };
  u32x4 rpre[8][2];
  if (EPI == LR_EPI_RESIDUAL) {
#pragma unroll
    for (int mt = 0; mt < 8; ++mt) {
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int rowc = min(m0 + wm * 128 + mt * 16 + j * 8 + hrow, M - 1);
        rpre[mt][j] = *reinterpret_cast<const u32x4*>(R + (size_t)rowc * ldc + n0 + wn * 64 + hcol);
      }
    }
  }
  // Rotation: the 8 token positions first, then the (cos, sin) pairs of four row groups at a time (16 reads in flight,
  // 64 registers) ahead of their stores -- one exposed latency per half instead of one per read.
  int ppre[8];
  // LDS-staged table (see RopeArgs::cs16): wave w fetches rows 32 w .. 32 w + 31 of the tile, 4 rows (1 KiB) per DMA
  // piece, into the stage buffer the last K tile did not use; the 16-byte chunk a lane FETCHES is XOR-permuted by the
  // row (chunk ^ (row & 15)) so that the ds_read_b64 of 16 rows x 2 quads below touch 64 different banks.
  const bool cs_lds = EPI == LR_EPI_ROPE && rope.cs16 != nullptr && rope.head_dim == 128 && (rope.rot_cols & 255) == 0;
  const bool rot_tile = n0 < rope.rot_cols;          // cs_lds: tile-uniform (rot_cols is a multiple of the tile width)
  const char* cs_stage = smem + (nkt & 1) * G2_STAGE_BYTES;
  if (EPI == LR_EPI_ROPE && cs_lds) {
    if (rot_tile) {
      int prow[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) prow[j] = rope.tok_pos[min(m0 + (wave * 8 + j) * 4 + (lane >> 4), M - 1)];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int row = (wave * 8 + j) * 4 + (lane >> 4);
        glds16(reinterpret_cast<const char*>(rope.cs16) + (size_t)prow[j] * 256 + (((lane & 15) ^ (row & 15)) << 4),
               const_cast<char*>(cs_stage) + (wave * 8 + j) * 1024);
      }
      PP_WAIT_VM(0);
    }
    PP_BARRIER();
  } else if (EPI == LR_EPI_ROPE) {
#pragma unroll
    for (int mt = 0; mt < 8; ++mt) ppre[mt] = rope.tok_pos[min(m0 + wm * 128 + mt * 16 + (lane & 15), M - 1)];
  }
  // LDS addresses of this lane's entries: row (wm 128 + mt 16 + li), pairs (wn & 1) 32 + nt 8 + 2 quad, + 1 = 8 bytes at
  // chunk ((wn & 1) 8 + 2 nt + (quad >> 1)) ^ li; the row tile mt is an immediate (4096 mt)
  int cs_off[4];
#pragma unroll
  for (int nt = 0; nt < 4; ++nt)
    cs_off[nt] = (wm * 128 + (lane & 15)) * 256 + (((((wn & 1) * 8 + 2 * nt + (quad >> 1)) ^ (lane & 15))) << 4) + (quad & 1) * 8;
  float rspre[8];  // folded RMSNorm: the scale of my 8 rows (1 when the norm is not folded)
#pragma unroll
  for (int mt = 0; mt < 8; ++mt) rspre[mt] = 1.0f;
  if (has_rs) {
#pragma unroll
    for (int mt = 0; mt < 8; ++mt) rspre[mt] = rs_lds[wm * 128 + mt * 16 + (lane & 15)];
  }
  float4 tpre[4][4];
#pragma unroll
  for (int mt = 0; mt < 8; ++mt) {
    if (EPI == LR_EPI_ROPE && cs_lds) {
      if (rot_tile) {
        typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
          const u32x2 e = *reinterpret_cast<const u32x2*>(cs_stage + cs_off[nt] + mt * 4096);
          tpre[mt & 3][nt] = float4{__builtin_bit_cast(float, e[0] << 16), __builtin_bit_cast(float, e[0] & 0xffff0000u),
                                    __builtin_bit_cast(float, e[1] << 16), __builtin_bit_cast(float, e[1] & 0xffff0000u)};
        }
      }
    } else if (EPI == LR_EPI_ROPE && (mt & 3) == 0) {
#pragma unroll
      for (int m2 = 0; m2 < 4; ++m2)
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
          const int col = n0 + wn * 64 + nt * 16 + quad * 4;
          const int colc = col < rope.rot_cols ? col : 0;   // v columns: any valid entry, not used
          tpre[m2][nt] = *reinterpret_cast<const float4*>(rope.cs + ((size_t)ppre[mt + m2] * (rope.head_dim >> 1) + ((colc % rope.head_dim) >> 1)) * 2);
        }
    }
    const int row = m0 + wm * 128 + mt * 16 + (lane & 15);
    const bool live = row < M;
    const int rowc = live ? row : M - 1;
    const float* prs = has_rs ? &rspre[mt] : nullptr;
    const int* ppos = nullptr;
    if (EPI == LR_EPI_PARTIAL) {
      if (live) {
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
          epi_store4<EPI>(acc[mt][nt], acc[mt][nt], C, R, (size_t)row * ldc + n0 + wn * 64 + nt * 16 + quad * 4);
      }
    } else if (EPI == LR_EPI_SWIGLU) {
      const int cbase = (n0 + wn * 64) >> 1;   // 32 output columns: tiles t = 0, 1 of 16
      unsigned a[2], b[2];
      {
        const u16x4 oa = epi_value4<EPI, has_rs ? 1 : 0>(acc[mt][0], acc[mt][1], u16x4{0, 0, 0, 0}, rope, rowc, cbase + quad * 4, prs, ppos);
        const u16x4 ob = epi_value4<EPI, has_rs ? 1 : 0>(acc[mt][2], acc[mt][3], u16x4{0, 0, 0, 0}, rope, rowc, cbase + 16 + quad * 4, prs, ppos);
        a[0] = (unsigned)oa[0] | ((unsigned)oa[1] << 16); a[1] = (unsigned)oa[2] | ((unsigned)oa[3] << 16);
        b[0] = (unsigned)ob[0] | ((unsigned)ob[1] << 16); b[1] = (unsigned)ob[2] | ((unsigned)ob[3] << 16);
      }
#pragma unroll
      for (int w = 0; w < 2; ++w) {
        const auto sw = __builtin_amdgcn_permlane16_swap(a[w], b[w], false, false);
        a[w] = sw[0];
        b[w] = sw[1];
      }
      if (live)
        *reinterpret_cast<u32x4*>(C + (size_t)row * ldc + cbase + (quad & 1) * 16 + (quad >> 1) * 8) = u32x4{a[0], a[1], b[0], b[1]};
    } else {
      u32x4 outp[2];
      if (EPI == LR_EPI_RESIDUAL) trade(rpre[mt][0], rpre[mt][1]);   // back to (own row: k = 0, k = 1)
#pragma unroll
      for (int k = 0; k < 2; ++k) {   // tiles 2k, 2k + 1
        const int cpair = n0 + wn * 64 + k * 32;
        u16x4 ra = u16x4{0, 0, 0, 0}, rb = ra;
        if (EPI == LR_EPI_RESIDUAL) {
          const u32x4 r16 = rpre[mt][k];
          unsigned x[2] = {r16[0], r16[1]}, y[2] = {r16[2], r16[3]};
#pragma unroll
          for (int w = 0; w < 2; ++w) {
            const auto sw = __builtin_amdgcn_permlane16_swap(x[w], y[w], false, false);
            x[w] = sw[0];
            y[w] = sw[1];
          }
          ra = u16x4{(u16)(x[0] & 0xffff), (u16)(x[0] >> 16), (u16)(x[1] & 0xffff), (u16)(x[1] >> 16)};
          rb = u16x4{(u16)(y[0] & 0xffff), (u16)(y[0] >> 16), (u16)(y[1] & 0xffff), (u16)(y[1] >> 16)};
        }
        const u16x4 oa = epi_value4<EPI, has_rs ? 1 : 0>(acc[mt][2 * k], acc[mt][2 * k], ra, rope, rowc, cpair + quad * 4, prs, ppos,
                                         EPI == LR_EPI_ROPE ? &tpre[mt & 3][2 * k] : nullptr);
        const u16x4 ob = epi_value4<EPI, has_rs ? 1 : 0>(acc[mt][2 * k + 1], acc[mt][2 * k + 1], rb, rope, rowc, cpair + 16 + quad * 4, prs, ppos,
                                         EPI == LR_EPI_ROPE ? &tpre[mt & 3][2 * k + 1] : nullptr);
        unsigned a[2] = {(unsigned)oa[0] | ((unsigned)oa[1] << 16), (unsigned)oa[2] | ((unsigned)oa[3] << 16)};
        unsigned b[2] = {(unsigned)ob[0] | ((unsigned)ob[1] << 16), (unsigned)ob[2] | ((unsigned)ob[3] << 16)};
#pragma unroll
        for (int w = 0; w < 2; ++w) {
          const auto sw = __builtin_amdgcn_permlane16_swap(a[w], b[w], false, false);
          a[w] = sw[0];
          b[w] = sw[1];
        }
        outp[k] = u32x4{a[0], a[1], b[0], b[1]};   // my 8 columns of tile pair k after the swap
      }
      trade(outp[0], outp[1]);
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int srow = m0 + wm * 128 + mt * 16 + j * 8 + hrow;
        if (srow < M) *reinterpret_cast<u32x4*>(C + (size_t)srow * ldc + n0 + wn * 64 + hcol) = outp[j];
      }
    }
  }
  GEMM_STAMP(3)
  GEMM_STAMP_END()
}

// =============================================================================================
// split-K reduce: sum the S fp32 partial planes in split order, then the real epilogue
// =============================================================================================
template <int EPI>
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* __restrict__ part, int S, u16* C,
                                                            const u16* R, int M, int N, RopeArgs rope) {
  const size_t plane = (size_t)M * N;
  if (EPI == LR_EPI_SWIGLU) {
    // output column c of N/2: gate column (c/16)*32 + c%16, up column 16 further (16-row interleave of wgu)
    const int n_out = N >> 1;
    const size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x) * 4;
    if (i >= (size_t)M * n_out) return;
    const int row = (int)(i / n_out), c = (int)(i % n_out);
    const size_t g_off = (size_t)row * N + (c >> 4) * 32 + (c & 15);
    floatx4 g = *reinterpret_cast<const floatx4*>(part + g_off);
    floatx4 u = *reinterpret_cast<const floatx4*>(part + g_off + 16);
    for (int s = 1; s < S; ++s) {
      g += *reinterpret_cast<const floatx4*>(part + s * plane + g_off);
      u += *reinterpret_cast<const floatx4*>(part + s * plane + g_off + 16);
    }
    epi_store4<EPI>(g, u, C, R, (size_t)row * n_out + c, rope, row, c);
  } else {
    const size_t off = ((size_t)blockIdx.x * 256 + threadIdx.x) * 4;
    if (off >= plane) return;
    floatx4 v = *reinterpret_cast<const floatx4*>(part + off);
    for (int s = 1; s < S; ++s) v += *reinterpret_cast<const floatx4*>(part + s * plane + off);
    epi_store4<EPI>(v, v, C, R, off, rope, (int)(off / N), (int)(off % N));
  }
}

// number of K splits for an M x N x K product on 256 x 256 tiles (1 = do not split): split only when the
// tiles alone leave at least half of the 256 CUs idle, keep >= 16 K tiles per split, <= 8 splits
static int splitk_factor(int M, int N, int K) {
  const int tiles = ((M + 255) / 256) * (N / 256);
  if (tiles > 128) return 1;
  int S = 256 / tiles;
  if (S > 8) S = 8;
  const int by_k = (K >> 6) / 16;
  if (S > by_k) S = by_k;
  return S < 2 ? 1 : S;
}

// tile raster: M tiles per group (llama_gemm.hip header). An XCD's 32 concurrent workgroups form a group_m x (32 / group_m)
// block of tiles that marches along N: every step of that march re-reads the group's A panels (group_m x 256 rows x K)
// and reads 32 / group_m new B panels, so the bytes an XCD's L2 pulls per tile are (group_m + 32 / group_m) / 32 of a
// tile's operand bytes: minimal at 4 x 8 or 8 x 4. Measured at M = 16384 (docs/EXPERIMENTS.md, round 2): 8 is best for the
// K = 4096 products, 4 (shorter A panels stay closer to the 4 MiB L2) for down_proj's K = 11008 (+3 %).
// LR_GEMM_GROUP_M overrides it for tuning runs only.
static int gemm256_group_m(int K) {
  static int g = -1;
  if (g < 0) {
    const char* e = getenv("LR_GEMM_GROUP_M");
    g = e ? atoi(e) : 0;
    if (g < 0) g = 0;
  }
  return g ? g : (K >= 8192 ? 4 : G2_GROUP_M);
}

template <int EPI, bool FOLD = false>
static int gemm256_prepare() {
  static bool done[LR_MAX_DEVICES] = {};
  return lr_ensure_dynamic_lds(reinterpret_cast<const void*>(gemm256rb_kernel<EPI, FOLD>), 2 * G2_STAGE_BYTES + (FOLD ? 1024 : 0), done);
}

template <int EPI>
static int launch_splitk(const u16* A, const u16* B, u16* C, const u16* R, int M, int N, int K, int S, RopeArgs rope,
                         float* ws, hipStream_t st, const u16* then_norm_w = nullptr, u16* then_norm_out = nullptr,
                         float then_norm_eps = 0.f, bool* then_norm_done = nullptr) {
  LrProfScope prof(LR_PROF_GEMM256, 2.0 * M * (double)N * K, st, LR_PROF_GEMM_TAG(EPI, N, K));
  if (int rc = gemm256_prepare<LR_EPI_PARTIAL>()) return rc;
  const int nwg = ((M + 255) / 256) * (N / 256);
  hipLaunchKernelGGL(gemm256rb_kernel<LR_EPI_PARTIAL>, dim3(nwg, S), dim3(512), 2 * G2_STAGE_BYTES, st, A, B,
                     reinterpret_cast<u16*>(ws), nullptr, M, N, K, gemm256_group_m(K), rope);
  LR_CHECK_LAUNCH("gemm256rb_kernel<partial>");
  if (EPI == LR_EPI_RESIDUAL && then_norm_w && then_norm_out && then_norm_done && lr_reduce_residual_rmsnorm_fits(N)) {
    *then_norm_done = true;   // the reduce pass also writes RMSNorm(C) for the next projection
    return lr_launch_reduce_residual_rmsnorm(ws, S, C, R, M, N, then_norm_w, then_norm_out, then_norm_eps, st);
  }
  const size_t quads = (EPI == LR_EPI_SWIGLU ? (size_t)M * (N >> 1) : (size_t)M * N) / 4;
  hipLaunchKernelGGL(splitk_reduce_kernel<EPI>, dim3((unsigned)((quads + 255) / 256)), dim3(256), 0, st, ws, S, C, R, M,
                     N, rope);
  LR_CHECK_LAUNCH("splitk_reduce_kernel");
  return LR_OK;
}

// =============================================================================================
template <int EPI>
static int launch_epi(const u16* A, const u16* B, u16* C, const u16* R, int M, int N, int K, int variant,
                      RopeArgs rope, hipStream_t st) {
  LrProfScope prof(variant >= 2 ? LR_PROF_GEMM256 : LR_PROF_GEMM_GENERIC, 2.0 * M * (double)N * K, st,
                   LR_PROF_GEMM_TAG(EPI, N, K));
  if (variant == 4) {
    const int nwg = ((M + 255) / 256) * (N / 256);
    bool launched = false;
    if constexpr (EPI == LR_EPI_ROPE || EPI == LR_EPI_SWIGLU) {
      if (rope.row_scale) {  // folded RMSNorm: the instantiation that scales its accumulator rows
        if (int rc = gemm256_prepare<EPI, true>()) return rc;
        hipLaunchKernelGGL((gemm256rb_kernel<EPI, true>), dim3(nwg), dim3(512), 2 * G2_STAGE_BYTES + 1024, st, A, B, C, R, M, N, K,
                           gemm256_group_m(K), rope);
        launched = true;
      }
    }
#ifdef LR_EXPERIMENTS
    if (!launched) {
      const char* stamp_env = getenv("LR_GEMM_STAMPS");
      if (stamp_env && stamp_env[0] == '1') {
        static bool done[LR_MAX_DEVICES] = {};
        if (int rc = lr_ensure_dynamic_lds(reinterpret_cast<const void*>(gemm256rb_kernel<EPI, false, true>),
                                           2 * G2_STAGE_BYTES, done))
          return rc;
        hipLaunchKernelGGL((gemm256rb_kernel<EPI, false, true>), dim3(nwg), dim3(512), 2 * G2_STAGE_BYTES, st, A, B, C, R, M, N,
                           K, gemm256_group_m(K), rope);
        launched = true;
      }
    }
#endif
    if (!launched) {
      if (int rc = gemm256_prepare<EPI>()) return rc;
      hipLaunchKernelGGL(gemm256rb_kernel<EPI>, dim3(nwg), dim3(512), 2 * G2_STAGE_BYTES, st, A, B, C, R, M, N, K,
                         gemm256_group_m(K), rope);
    }
    LR_CHECK_LAUNCH("gemm256rb_kernel");
  } else {
    dim3 grid((N + GG_BN - 1) / GG_BN, (M + GG_BM - 1) / GG_BM);
    hipLaunchKernelGGL(gemm_generic_kernel<EPI>, grid, dim3(256), 0, st, A, B, C, R, M, N, K, rope);
    LR_CHECK_LAUNCH("gemm_generic_kernel");
  }
  return LR_OK;
}

#ifdef LR_EXPERIMENTS
extern "C" int lr_debug_gemm_stamps(unsigned long long* out, int n_workgroups) {
  if (!out || n_workgroups < 1 || n_workgroups > 16384) LR_FAIL(LR_EINVAL, "lr_debug_gemm_stamps: bad arguments");
  LR_CHECK_HIP(hipMemcpyFromSymbol(out, HIP_SYMBOL(g_gemm_stamps),
                                   (size_t)n_workgroups * 2 * GEMM_STAMP_SLOTS * sizeof(unsigned long long)));
  return LR_OK;
}
#endif

int lr_launch_gemm(const u16* A, const u16* B, u16* C, const u16* R, int M, int N, int K, int epi,
                   int variant, hipStream_t st, const int32_t* tok_pos, const float* rope_cs, int head_dim,
                   int rot_cols, float* splitk_ws, size_t splitk_ws_bytes, const float* row_scale,
                   const unsigned* rope_cs16, const u16* then_norm_w, u16* then_norm_out, float then_norm_eps,
                   bool* then_norm_done) {
  if (then_norm_done) *then_norm_done = false;
  if (M <= 0) return LR_OK;
  if (N <= 0 || K <= 0) LR_FAIL(LR_EINVAL, "gemm: N=%d K=%d", N, K);
  const bool fast_ok = (N % 256 == 0) && (K % 64 == 0) && M >= 1;
  if (variant == 0) variant = (fast_ok && M >= 128) ? 4 : 1;
  if (variant == 5 && !fast_ok) variant = 1;  // latency mode: any row count (B last-token rows too), shapes as auto
  if (variant != 1 && variant != 4 && variant != 5)
    LR_FAIL(LR_EINVAL, "gemm: unknown variant %d (0 auto, 1 generic, 4 = 256x256x64 MFMA tile, 5 = 4 + split-K)", variant);
  if (variant == 4 && !fast_ok)
    LR_FAIL(LR_EUNSUPPORTED, "gemm variant 4 needs N%%256==0 and K%%64==0 (N=%d K=%d)", N, K);
  if (epi == LR_EPI_SWIGLU && (N % 32 != 0)) LR_FAIL(LR_EINVAL, "swiglu epilogue needs N%%32==0 (N=%d)", N);
  if (epi == LR_EPI_RESIDUAL && !R) LR_FAIL(LR_EINVAL, "residual epilogue without residual pointer");
  RopeArgs rope{tok_pos, rope_cs, head_dim, rot_cols, row_scale, rope_cs16};
  if (row_scale && epi != LR_EPI_ROPE && epi != LR_EPI_SWIGLU)
    LR_FAIL(LR_EINVAL, "gemm: a row scale (folded RMSNorm) is only applied by the rope and swiglu epilogues");
  if (epi == LR_EPI_ROPE) {
    if (!tok_pos || !rope_cs || head_dim < 2 || head_dim % 4 != 0 || rot_cols % 4 != 0 || rot_cols > N)
      LR_FAIL(LR_EINVAL, "rope epilogue: bad arguments (head_dim=%d rot_cols=%d)", head_dim, rot_cols);
  }
  if (variant == 5) {
    const int S = splitk_factor(M, N, K);
    if (S >= 2) {
      if (!splitk_ws || (size_t)S * M * N * sizeof(float) > splitk_ws_bytes)
        LR_FAIL(LR_EWORKSPACE, "gemm variant 5: split-K x%d of %dx%d needs %zu workspace bytes, have %zu", S, M, N,
                (size_t)S * M * N * sizeof(float), splitk_ws ? splitk_ws_bytes : (size_t)0);
      switch (epi) {
        case LR_EPI_STORE: return launch_splitk<LR_EPI_STORE>(A, B, C, R, M, N, K, S, rope, splitk_ws, st);
        case LR_EPI_RESIDUAL:
          return launch_splitk<LR_EPI_RESIDUAL>(A, B, C, R, M, N, K, S, rope, splitk_ws, st, then_norm_w, then_norm_out,
                                                then_norm_eps, then_norm_done);
        case LR_EPI_SWIGLU: return launch_splitk<LR_EPI_SWIGLU>(A, B, C, R, M, N, K, S, rope, splitk_ws, st);
        case LR_EPI_ROPE: return launch_splitk<LR_EPI_ROPE>(A, B, C, R, M, N, K, S, rope, splitk_ws, st);
      }
      LR_FAIL(LR_EINVAL, "gemm: unknown epilogue %d", epi);
    }
    variant = 4;
  }
  switch (epi) {
    case LR_EPI_STORE: return launch_epi<LR_EPI_STORE>(A, B, C, R, M, N, K, variant, rope, st);
    case LR_EPI_RESIDUAL: return launch_epi<LR_EPI_RESIDUAL>(A, B, C, R, M, N, K, variant, rope, st);
    case LR_EPI_SWIGLU: return launch_epi<LR_EPI_SWIGLU>(A, B, C, R, M, N, K, variant, rope, st);
    case LR_EPI_ROPE: return launch_epi<LR_EPI_ROPE>(A, B, C, R, M, N, K, variant, rope, st);
  }
  LR_FAIL(LR_EINVAL, "gemm: unknown epilogue %d", epi);
}
