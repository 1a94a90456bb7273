// llama_attn.hip -- varlen causal self-attention over packed prompts on gfx950.
//
// Replaces flash-attn 2.5.8's CUDA varlen kernel (train_ranker.py:61, attn_implementation=
// "flash_attention_2") / HF eager attention inside LlamaModel (reached from model/llm.py:89-100).
// Packed (unpadded) execution is legal because padding is on the left and masked
// (SURVEY.md 8(a) a15); positions restart at 0 in every prompt.
//
//  attn_generic_kernel : any head_dim <= 256, GQA, one wave per (token, head). Test models.
//  attn_mfma128_kernel : head_dim = 128, flash-style online softmax on v_mfma_f32_16x16x32_bf16.
//    Everything is computed transposed so that no cross-lane data movement is needed:
//      S^T = K Q^T   (A = K rows from LDS, B = Q rows held in registers)  -> lane owns ONE query
//                     row (lane&15) and 16 of the block's 64 keys: row max/sum = 2 shuffles
//      O^T = V^T P^T (A = V read with ds_read_b64_tr_b16 from the row-major LDS tile,
//                     B = P straight from the S^T accumulators, packed to bf16 in-lane)
//                     -> the O accumulator's column is again the lane's query row, so the
//                     online-softmax rescale is lane-local.
//    A workgroup = 4 wave64 = 128 query rows of one (prompt, head); each wave 32 rows, so every K/V
//    fragment read from LDS feeds two MFMA column tiles. K tile XOR-swizzled by (row&15) for
//    ds_read_b128; V tile by the dual-use swizzle (row reads + transposed reads).
//    Shared prompt prefix (prefix_len = P > 0): segment 0 of the packed batch holds the P tokens every prompt
//    starts with (the template text, dataloader/utils.py:24-40), segment s >= 1 the rest of prompt s-1 at
//    positions P.. . Such a segment's keys [0, P) are read from segment 0's rows; query tiles and key blocks stay
//    aligned to ABSOLUTE positions, so every row executes exactly the instruction sequence of the unshared run.
#include <stdlib.h>

#include <type_traits>

#include "llama_kernels.h"
#include "lr_profile.h"

typedef unsigned short u16;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef short short4v __attribute__((ext_vector_type(4)));
typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef u16 u16x4 __attribute__((ext_vector_type(4)));
typedef u16 u16x8 __attribute__((ext_vector_type(8)));

// =============================================================================================
// generic
// =============================================================================================
__global__ __launch_bounds__(256) void attn_generic_kernel(const u16* qkv, u16* out, const int32_t* cu, int B,
                                                           int n_tok, int nh, int nkv, int hd,
                                                           const int32_t* q_rows, float* lse) {
  __shared__ float qs[4][256];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int item = blockIdx.x * 4 + wave;  // (token, head)
  if (item >= n_tok * nh) return;
  // q_rows (optional): only these query tokens are evaluated and the output is compact [n_tok][nh*hd]
  const int orow = item / nh, h = item % nh;
  const int tok = q_rows ? q_rows[orow] : orow;
  const int kvh = h / (nh / nkv);
  const int stride = (nh + 2 * nkv) * hd;
  int lo = 0, hi = B;  // prompt index: largest b with cu[b] <= tok
  while (hi - lo > 1) {
    int mid = (lo + hi) >> 1;
    if (cu[mid] <= tok) lo = mid; else hi = mid;
  }
  const int s0 = cu[lo];
  const int pos = tok - s0;
  const u16* qp = qkv + (size_t)tok * stride + h * hd;
  for (int d = lane; d < hd; d += 64) qs[wave][d] = bf2f(qp[d]);
  __builtin_amdgcn_wave_barrier();
  const float scale = 1.0f / sqrtf((float)hd);
  float m = -__builtin_inff(), l = 0.f;
  float o[4] = {0.f, 0.f, 0.f, 0.f};  // dims lane, lane+64, lane+128, lane+192
  for (int k0 = 0; k0 <= pos; k0 += 64) {
    const int key = k0 + lane;
    float s = -__builtin_inff();
    if (key <= pos) {
      const u16* kp = qkv + (size_t)(s0 + key) * stride + (nh + kvh) * hd;
      float acc = 0.f;
      for (int d = 0; d < hd; ++d) acc = __builtin_fmaf(qs[wave][d], bf2f(kp[d]), acc);
      s = acc * scale;
    }
    float mx = s;
#pragma unroll
    for (int sft = 32; sft >= 1; sft >>= 1) mx = fmaxf(mx, __shfl_xor(mx, sft, 64));
    const float m_new = fmaxf(m, mx);
    const float alpha = __expf(m - m_new);
    const float p = (key <= pos) ? __expf(s - m_new) : 0.f;
    float ps = p;
#pragma unroll
    for (int sft = 32; sft >= 1; sft >>= 1) ps += __shfl_xor(ps, sft, 64);
    l = l * alpha + ps;
    m = m_new;
    const float pb = bf2f(f2bf(p));  // P enters the PV product in bf16, like the MFMA path
#pragma unroll
    for (int i = 0; i < 4; ++i) o[i] *= alpha;
    const int nk = min(64, pos - k0 + 1);
    for (int j = 0; j < nk; ++j) {
      const float pj = __shfl(pb, j, 64);
      const u16* vp = qkv + (size_t)(s0 + k0 + j) * stride + (nh + nkv + kvh) * hd;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        int d = lane + 64 * i;
        if (d < hd) o[i] = __builtin_fmaf(pj, bf2f(vp[d]), o[i]);
      }
    }
  }
  u16* op = out + (size_t)orow * nh * hd + h * hd;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    int d = lane + 64 * i;
    if (d < hd) op[d] = f2bf(o[i] / l);
  }
  if (lse && lane == 0) lse[(size_t)orow * nh + h] = m + logf(l);  // log-sum-exp of the scaled scores (training)
}

// =============================================================================================
// MFMA, head_dim 128
// =============================================================================================
#define FA_QROWS 128  // query rows per workgroup
#define FA_KB 64      // keys per block
#define FA_DEFER 8.0f  // a row's softmax reference moves only when a score exceeds it by more than this (exp2 domain)

__device__ __forceinline__ int v_off(int row, int ch) {  // dual-use swizzle, 256-byte rows
  return 256 * row + 16 * (ch ^ (((row & 3) << 2) | ((row >> 2) & 3)));
}

// LDS-DMA through a buffer descriptor as INLINE ASM. With the builtin (__builtin_amdgcn_raw_ptr_buffer_load_lds) hipcc
// knows an LDS write is pending on the vector-memory counter and -- unable to prove that a ds_read_b64_tr_b16 (the V^T
// fragment reads) touches another stage buffer -- puts `s_waitcnt vmcnt(0)` in front of the first transposed read of every
// key block: each wave then sat out the landing of the NEXT block's tiles in the middle of the current block (found in
// round 3 in the .s of the product kernel; it is also why requesting the V fragments earlier was slower). The asm form is
// invisible to that pass; the one wait that is needed stands in front of the block's barrier, written by hand.
// (M0 is a reserved register to hipcc: it cannot be named as a clobber, and nothing else in these kernels uses it --
// gfx9+ LDS instructions do not read M0.)
typedef int fa_int4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ fa_int4 fa_make_rsrc(const void* base, int num_records) {
  const unsigned long long b = reinterpret_cast<unsigned long long>(base);
  fa_int4 r;
  r[0] = __builtin_amdgcn_readfirstlane((int)(unsigned)b);
  r[1] = __builtin_amdgcn_readfirstlane((int)((b >> 32) & 0xffffu));   // stride 0
  r[2] = __builtin_amdgcn_readfirstlane(num_records);
  r[3] = 0x00020000;
  return r;
}
__device__ __forceinline__ void fa_glds16(const void* gsrc, const void* lds_wave_base) {   // per-lane source address
  const unsigned m0v = (unsigned)(size_t)((__attribute__((address_space(3))) const char*)lds_wave_base);
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" ::"s"(m0v), "v"(gsrc) : "memory");
}
__device__ __forceinline__ void fa_dma16(fa_int4 rsrc, const void* lds_wave_base, unsigned voff) {
  const unsigned m0v = (unsigned)(size_t)((__attribute__((address_space(3))) const char*)lds_wave_base);
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds" ::"s"(m0v), "v"(voff), "s"(rsrc)
               : "memory");
}

// x[lane] (op) x[lane ^ 16] and x[lane] (op) x[lane ^ 32] without the LDS crossbar: gfx950's row / half swaps
// (v_permlane16_swap, v_permlane32_swap) hand both partners to every lane at VALU speed; __shfl_xor compiles
// to ds_bpermute_b32 (~100+ cycles of dependent latency, four of them on every key block's critical path).
// hipcc pitfall: __builtin_bit_cast(float, r[1]) on the builtin's 2-vector result reads element 0 (the cast
// takes the vector's address) -- copy the elements into scalars first.
__device__ __forceinline__ void fa_swap16(float v, float& a, float& b) {
  const unsigned u = __builtin_bit_cast(unsigned, v);
  const auto r = __builtin_amdgcn_permlane16_swap(u, u, false, false);
  const unsigned r0 = r[0], r1 = r[1];
  a = __builtin_bit_cast(float, r0);
  b = __builtin_bit_cast(float, r1);
}
__device__ __forceinline__ void fa_swap32(float v, float& a, float& b) {
  const unsigned u = __builtin_bit_cast(unsigned, v);
  const auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
  const unsigned r0 = r[0], r1 = r[1];
  a = __builtin_bit_cast(float, r0);
  b = __builtin_bit_cast(float, r1);
}
// v_max3_f32 on raw MFMA outputs: fmaxf() makes hipcc canonicalise every input first (a v_max_f32 x, x per score);
// the scores are finite or -inf here, where max is exact whatever the association
__device__ __forceinline__ float fa_max3(float a, float b, float c) {
  float r;
  asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
  return r;
}
__device__ __forceinline__ float fa_max2(float a, float b) {
  float r;
  asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ __forceinline__ float fa_max_xor16_32(float v) {
  float a, b;
  fa_swap16(v, a, b);
  fa_swap32(fa_max2(a, b), a, b);
  return fa_max2(a, b);
}
__device__ __forceinline__ float fa_sum_xor16_32(float v) {
  float a, b;
  fa_swap16(v, a, b);
  fa_swap32(a + b, a, b);
  return a + b;
}

#define FA_TILE_BYTES (FA_KB * 256)        // one K or V tile: 64 keys x 128 dims bf16
#define FA_STAGE_BYTES (2 * FA_TILE_BYTES)  // K tile + V tile

// Diagnostic stamps: STAMP = true is instantiated only in a -DLR_EXPERIMENTS build (make EXPERIMENTS=1, then
// LR_ATTN_STAMPS=1 at run time; tools/attn_stamps.py); the product library holds no stamping code. s_memtime deltas
// of the key-block loop's segments summed over the loop, wave 0 of the first 64 workgroups.
__device__ unsigned long long g_attn_stamps[64 * 8];
// per-workgroup timeline of a stamped launch (s_memrealtime, 100 MHz; tools/attn_wg_trace.py):
// entry | prologue done | key-block loop done | exit | HW_ID | XCC_ID | key blocks | blockIdx
#define FA_WG_TRACE 8192
#ifdef LR_EXPERIMENTS
__device__ unsigned long long g_attn_wg[FA_WG_TRACE * 8];
#else
__device__ unsigned long long g_attn_wg[8];   // never written: the stamped instantiation exists in experiment builds only
#endif
#define FA_RT(dst)                                                                \
  if (STAMP) {                                                                    \
    __builtin_amdgcn_sched_barrier(0);                                            \
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(dst)::"memory"); \
    __builtin_amdgcn_sched_barrier(0);                                            \
  }
#define FA_STAMP(slot)                                                         \
  if (STAMP) {                                                                 \
    unsigned long long t_;                                                     \
    __builtin_amdgcn_sched_barrier(0);                                         \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory"); \
    __builtin_amdgcn_sched_barrier(0);                                         \
    stamp_acc[(slot)] += t_ - t_prev;                                          \
    t_prev = t_;                                                               \
  }

// LASTQ (the pruned last layer, api_llama.hip): only each prompt's LAST query row is evaluated. `qkv` is then the
// [rows][2 * nkv * hd] K | V projection of every row, `q_last` holds one rotated query row per prompt ([prompt][nh * hd]),
// a workgroup = one (prompt, head) walks all the prompt's key blocks with wave 0 computing (every lane of the wave holds
// the same query row, so the tile arithmetic -- and with it the bits of that row -- is that of the full kernel) and all
// four waves staging, and `out` receives one row per prompt ([prompt][nh * hd]).
template <bool STAMP, bool LASTQ = false>
__global__ __launch_bounds__(256, 2) void attn_mfma128_kernel(const u16* __restrict__ qkv, u16* out,
                                                              const int32_t* cu, int prefix_len, int nh, int nkv,
                                                              int max_qblocks, int n_pairs, float* lse,
                                                              const u16* __restrict__ q_rows_last = nullptr) {
  unsigned long long stamp_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, t_prev = 0;
  unsigned long long rt_entry = 0, rt_pro = 0, rt_loop = 0, rt_exit = 0;
  FA_RT(rt_entry)
  if (STAMP) asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_prev)::"memory");
  // [2 stages][K 16 KiB | V 16 KiB]; filled by LDS-DMA (lane-linear 1 KiB pieces = 4 rows x 256 B),
  // the XOR swizzles are applied to the SOURCE chunk: position p of row r holds chunk p ^ s(r).
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int hd = 128;
  // ---- workgroup -> (segment, head, query tile). A (segment, head) PAIR is bound to one of 8 dispatch streams
  // (blocks b and b + 8 are observed to share an XCD: speed only, never correctness), so the 3..12 query tiles that
  // re-read one pair's K/V hit that XCD's L2; inside a stream the pairs' heavy tiles (qb >= 2, heaviest first) run pair
  // after pair, and the two lightest tiles of every pair are kept for the end, where they fill the tail of the launch
  // (a causal tile costs ~ its key-block count 2 qb + 2: launched last, a heavy tile would leave most CUs idle).
  int seg, h, qb;
  if (LASTQ) {
    const int id = blockIdx.x, stream = id & 7, pl = id >> 3;
    const int pair = pl * 8 + stream;
    if (pair >= n_pairs) return;
    seg = __builtin_amdgcn_readfirstlane(pair / nh);
    h = __builtin_amdgcn_readfirstlane(pair - seg * nh);
    if (prefix_len > 0 && seg == 0) return;   // segment 0 is the shared prefix, not a prompt
    qb = 0;                                    // set below, once T is known
  } else {
    const int id = blockIdx.x, stream = id & 7, j = id >> 3;
    const int ppx = (n_pairs + 7) >> 3;                  // pairs per stream
    const int n_light = min(max_qblocks, 2), n_heavy = max_qblocks - n_light;
    int pl;
    if (j < ppx * n_heavy) {
      pl = j / n_heavy;
      qb = max_qblocks - 1 - j % n_heavy;
    } else {
      const int j2 = j - ppx * n_heavy;
      pl = j2 / n_light;
      qb = n_light - 1 - j2 % n_light;
    }
    const int pair = pl * 8 + stream;
    if (pair >= n_pairs) return;
    // the integer divisions above run on the vector ALU (there is no scalar divide): hand the wave-uniform results
    // back to scalar registers, or every buffer descriptor built from them is wrapped in a waterfall loop
    seg = __builtin_amdgcn_readfirstlane(pair / nh);
    h = __builtin_amdgcn_readfirstlane(pair - seg * nh);
    qb = __builtin_amdgcn_readfirstlane(qb);
  }
  const int tok0 = cu[seg];
  const int P = (prefix_len > 0 && seg > 0) ? prefix_len : 0;  // keys [0, P) live in segment 0's rows [0, P)
  const int T = P + cu[seg + 1] - tok0;                        // sequence length, prefix included
  if (LASTQ) qb = (T - 1) / FA_QROWS;
  if (qb * FA_QROWS >= T || (qb + 1) * FA_QROWS <= P) return;  // no query row of this segment in the tile
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int quad = lane >> 4, li = lane & 15;
  const int kvh = __builtin_amdgcn_readfirstlane(h / (nh / nkv));
  const int stride = LASTQ ? 2 * nkv * hd : (nh + 2 * nkv) * hd;
  const int koff0 = LASTQ ? kvh * hd : (nh + kvh) * hd, voff0 = koff0 + nkv * hd;
  const int vtok0 = tok0 - P;  // the row of position p >= P is vtok0 + p (tok0 >= P: segment 0 precedes it)
  const u16* kbase = qkv + (size_t)vtok0 * stride + koff0;
  const u16* vbase = qkv + (size_t)vtok0 * stride + voff0;
  const u16* pkbase = qkv + koff0;        // prefix rows start at packed row 0
  const u16* pvbase = qkv + voff0;
  const int prompt = prefix_len > 0 ? seg - 1 : seg;   // LASTQ: row of q_last / out

  // ---- Q fragments (B operand of S^T = K Q^T): row q, d = 32*ks + 8*quad + 0..7
  bf16x8 qf[2][4];
  int qabs[2];
#pragma unroll
  for (int qt = 0; qt < 2; ++qt) {
    qabs[qt] = LASTQ ? T - 1 : qb * FA_QROWS + wave * 32 + qt * 16 + li;
    const int qr = min(max(qabs[qt], P), T - 1);
    const u16* qp = LASTQ ? q_rows_last + (size_t)prompt * nh * hd + h * hd + quad * 8
                          : qkv + (size_t)(vtok0 + qr) * stride + h * hd + quad * 8;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) qf[qt][ks] = *reinterpret_cast<const bf16x8*>(qp + ks * 32);
  }

  floatx4 ot[2][8];
#pragma unroll
  for (int qt = 0; qt < 2; ++qt)
#pragma unroll
    for (int dt = 0; dt < 8; ++dt) ot[qt][dt] = floatx4{0.f, 0.f, 0.f, 0.f};
  float m_run[2] = {-__builtin_inff(), -__builtin_inff()};
  float mthr[2] = {-__builtin_inff(), -__builtin_inff()};   // (m + FA_DEFER) / scale of the lane's row, in raw-score units
  // Row sums of P through the matrix pipe (round 5): one more A row of ones beside V^T gives l = sum_k P[k][row] as an MFMA
  // accumulator -- every lane of a row's column holds the complete sum of the bf16 P the numerator uses -- instead of 16
  // v_add_f32 per half and block plus two cross-lane exchanges at the end (the block loop is bound by what a SIMD ISSUES: an
  // MFMA costs it ~8 cycles, 16 adds ~70).
  floatx4 l_acc[2] = {floatx4{0.f, 0.f, 0.f, 0.f}, floatx4{0.f, 0.f, 0.f, 0.f}};
  bf16x8 ones_f;
#pragma unroll
  for (int i = 0; i < 8; ++i) ones_f[i] = (__bf16)1.0f;

  const int q_last = min(qb * FA_QROWS + FA_QROWS - 1, T - 1);   // (LASTQ: qb is the tile of row T - 1, so this is T - 1)
  const int kb_last = q_last / FA_KB;
  const int wave_q_last = LASTQ ? T - 1 : qb * FA_QROWS + wave * 32 + 31;  // last query row this wave owns
  const float sl2 = 0.08838834764831845f * 1.4426950408889634f;  // 1/sqrt(128) * log2(e)
  const float inv_sl2 = 1.0f / sl2;

  // ---- DMA staging: 16 pieces per tile (4 rows each); wave w moves pieces 4w..4w+3 of K and of V
  const int prow = lane >> 4, ppos = lane & 15;
  // per-lane byte offsets of this wave's 4 K pieces and 4 V pieces inside a key block (row and swizzled chunk are
  // block-invariant)
  unsigned koff[4], voff[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row = (wave * 4 + i) * 4 + prow;
    koff[i] = (unsigned)(row * stride + (ppos ^ (row & 15)) * 8) * 2u;
    voff[i] = (unsigned)(row * stride + (ppos ^ (((row & 3) << 2) | ((row >> 2) & 3))) * 8) * 2u;
  }
  // A block that lies inside the segment's own rows is fetched through a per-block buffer descriptor: base = the
  // block's first K (V) row, num_records = the bytes up to the end of the segment's last row, so rows past the sequence
  // end are range-checked to ZERO by the hardware (those keys are causally masked for every stored query row: P = 0
  // exactly, whatever finite bytes K and V hold -- same bits as fetching a clamped row). Descriptor, LDS base (M0) and
  // block offset are scalar work; the per-lane offsets koff / voff are block-invariant: no vector ALU in the staging.
  auto stage = [&](int kb, int buf) {
    char* base = smem + buf * FA_STAGE_BYTES + wave * 4096;
    if (kb * FA_KB >= P) {
      const size_t blk_off = (size_t)kb * FA_KB * stride * 2;
      const int records = ((T - 1 - kb * FA_KB) * stride + hd) * 2;  // bytes from the block's first K (V) element
      const fa_int4 rk = fa_make_rsrc(reinterpret_cast<const char*>(kbase) + blk_off, records);
      const fa_int4 rv = fa_make_rsrc(reinterpret_cast<const char*>(vbase) + blk_off, records);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        fa_dma16(rk, base + i * 1024, koff[i]);
        fa_dma16(rv, base + FA_TILE_BYTES + i * 1024, voff[i]);
      }
    } else {  // the block holds shared-prefix keys: rows < P come from segment 0 (per-lane addresses)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int row = (wave * 4 + i) * 4 + prow;
        const int key = min(kb * FA_KB + row, T - 1);
        const int kchunk = ppos ^ (row & 15);
        const int vchunk = ppos ^ (((row & 3) << 2) | ((row >> 2) & 3));
        const u16* kr = key < P ? pkbase : kbase;
        const u16* vr = key < P ? pvbase : vbase;
        fa_glds16(kr + (size_t)key * stride + kchunk * 8, base + i * 1024);
        fa_glds16(vr + (size_t)key * stride + vchunk * 8, base + FA_TILE_BYTES + i * 1024);
      }
    }
  };

  // LDS read addresses: everything lane-dependent is computed ONCE (4 K bases, 8 V bases); stage buffer, key sub-tile
  // and k-step are compile-time immediates of the ds_read (the key-block loop is unrolled by two so that the stage
  // buffer is one of them): no vector ALU between the MFMAs for addressing (was ~38 v_add / v_add3 per key block).
  typedef __attribute__((address_space(3))) char lds_char;
  lds_char* const lds = (lds_char*)smem;
  lds_char *kb_off[4], *vb_off[8];   // pointers, so that the LDS base is added here and not at every read
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) kb_off[ks] = lds + (li * 256 + (((ks * 4 + quad) ^ li) << 4));   // row nt*16 + li: (row & 15) = li
  {
    const int qp = li >> 2, p4 = li & 3;
#pragma unroll
    for (int dt = 0; dt < 8; ++dt) vb_off[dt] = lds + (v_off(quad * 4 + qp, dt * 2 + (p4 >> 1)) + 8 * (p4 & 1));   // + 8192 ks2 + 4096 half
  }

  stage(0, 0);
  // Q must be resident before the loop: otherwise hipcc carries its pending-load state into the loop
  // and waits for the in-loop DMA prefetch (vmcnt is in-order) in front of the first MFMAs.
#pragma unroll
  for (int qt = 0; qt < 2; ++qt)
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) asm volatile("" ::"v"(qf[qt][ks]));
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the asm DMAs of block 0 (hipcc does not count them)
  __syncthreads();
  FA_STAMP(0)  // prologue: Q fragments + first K/V tile landed
  FA_RT(rt_pro)

  auto block = [&](const int kb, auto buf_c) {
    constexpr int BUF = decltype(buf_c)::value;
    constexpr int KS = BUF * FA_STAGE_BYTES, VS = KS + FA_TILE_BYTES;   // LDS byte offsets of this block's K and V tiles
    if (kb < kb_last) stage(kb + 1, BUF ^ 1);
    FA_STAMP(1)  // DMA issue

    if (kb * FA_KB <= wave_q_last && wave_q_last >= P && (!LASTQ || wave == 0)) {  // otherwise every key of the block is masked for this wave
                                                          // (or all its rows belong to the prefix segment)
      // ---- S^T = K Q^T : st[qt][nt] rows = keys nt*16 + 4*quad + r, col = query li
      floatx4 st[2][4];
#pragma unroll
      for (int qt = 0; qt < 2; ++qt)
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) st[qt][nt] = floatx4{0.f, 0.f, 0.f, 0.f};
      typedef __attribute__((address_space(3))) const bf16x8 lds_bf16x8;
      bf16x8 kf[2][4];
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) kf[0][nt] = *reinterpret_cast<lds_bf16x8*>(kb_off[0] + (KS + nt * 4096));
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        if (ks < 3) {
#pragma unroll
          for (int nt = 0; nt < 4; ++nt)
            kf[(ks + 1) & 1][nt] = *reinterpret_cast<lds_bf16x8*>(kb_off[ks + 1] + (KS + nt * 4096));
        }
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
#pragma unroll
          for (int qt = 0; qt < 2; ++qt)
            st[qt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf[ks & 1][nt], qf[qt][ks], st[qt][nt], 0, 0, 0);
      }

      if (STAMP) asm volatile("" ::"v"(st[0][0]), "v"(st[1][3]));
      FA_STAMP(2)  // S^T = K Q^T
      // ---- online softmax (lane-local row), P packed as the B operand of O^T = V^T P^T
      bf16x8 pa[2][2];
      const bool diag = (kb * FA_KB + FA_KB - 1) > (LASTQ ? T - 1 : qb * FA_QROWS + wave * 32);  // block needs masking
#pragma unroll
      for (int qt = 0; qt < 2; ++qt) {
        // raw-score row maximum (the scale is positive, so max commutes with it)
        float mx = -__builtin_inff();
        if (diag) {  // a real (wave-uniform) branch: written as a select inside the loop below, hipcc predicates
                     // all 32 scores of every block -- ~100 wasted VALU instructions on the off-diagonal blocks
#pragma unroll
          for (int nt = 0; nt < 4; ++nt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const int key = kb * FA_KB + nt * 16 + quad * 4 + r;
              st[qt][nt][r] = (key <= qabs[qt]) ? st[qt][nt][r] : -__builtin_inff();
            }
          __builtin_amdgcn_sched_barrier(0);
        }
        mx = fa_max3(st[qt][0][0], st[qt][0][1], st[qt][0][2]);
        mx = fa_max3(mx, st[qt][0][3], st[qt][1][0]);
#pragma unroll
        for (int nt = 1; nt < 4; ++nt) {
          mx = fa_max3(mx, st[qt][nt][1], st[qt][nt][2]);
          if (nt < 3) mx = fa_max3(mx, st[qt][nt][3], st[qt][nt + 1][0]);
        }
        mx = fa_max2(mx, st[qt][3][3]);      // this lane's 16 keys of the row
        // Deferred maximum (round 5, as in llama_attn256.hip): a row's reference value m moves only when a score exceeds it by
        // more than FA_DEFER in the exp2 domain, so P stays below 2^FA_DEFER (exact in fp32 sums, 8 significant bits in bf16
        // whatever its scale) and O / l are rescaled in a tile's first block and then almost never -- with the running maximum
        // some row of the 16 moved in most blocks of a 600-1 100-token prompt (32 multiplies + an exp2 per half and block).
        // The decision is per row: a row's bits do not depend on its tile mates (alpha is exactly 1 for a row that keeps m).
        // Common path: no lane of this half sees a score above its row's cached RAW threshold (m + FA_DEFER) / scale -- one compare
        // and a wave-uniform branch; the row maximum across the four lanes of a row, the new reference, alpha and the rescale
        // run only behind it (a row's own decision is the same either way: its maximum exceeds the threshold iff one of its
        // lanes' does, and a row that keeps m multiplies by exactly 1).
        if (__any(mx > mthr[qt])) {
          const float rmx = fa_max_xor16_32(mx);
          const bool grew = rmx > mthr[qt];   // the SAME predicate as the lanes' test, on the row maximum: a row moves iff one of
                                              // its own lanes asked for it, whatever the other rows of the wave do
          const float m_new = grew ? rmx * sl2 : m_run[qt];
          const float alpha = __builtin_amdgcn_exp2f(m_run[qt] - m_new);
#pragma unroll
          for (int r = 0; r < 4; ++r) l_acc[qt][r] *= alpha;
#pragma unroll
          for (int dt = 0; dt < 8; ++dt)
#pragma unroll
            for (int r = 0; r < 4; ++r) ot[qt][dt][r] *= alpha;
          m_run[qt] = m_new;
          mthr[qt] = (m_new + FA_DEFER) * inv_sl2;
        }
        const float m_new = m_run[qt];
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            // exp2(s*scale*log2e - m): one fma + one exp per score
            const float p = __builtin_amdgcn_exp2f(__builtin_fmaf(st[qt][nt][r], sl2, -m_new));
            pa[qt][nt >> 1][(nt & 1) * 4 + r] = (__bf16)p;
          }
      }

      if (STAMP) asm volatile("" ::"v"(pa[0][0]), "v"(pa[1][1]));
      FA_STAMP(3)  // softmax
      // ---- O^T += V^T P^T : A = V^T fragment via transposed LDS reads
      // V^T fragment rows: keys ks2*32 + quad*4 + qp (+16); their swizzle term ((row & 3) << 2 | (row >> 2) & 3) does not
      // depend on ks2 or the +16, so row0's address differs from vb_off[dt] by the immediate 8192 ks2 (+ 4096)
#pragma unroll
      for (int ks2 = 0; ks2 < 2; ++ks2) {
#pragma unroll
        for (int qt = 0; qt < 2; ++qt) l_acc[qt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones_f, pa[qt][ks2], l_acc[qt], 0, 0, 0);
#pragma unroll
        for (int dt = 0; dt < 8; ++dt) {
          const short4v t0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
              (__attribute__((address_space(3))) short4v*)(vb_off[dt] + (VS + ks2 * 8192)));
          const short4v t1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
              (__attribute__((address_space(3))) short4v*)(vb_off[dt] + (VS + ks2 * 8192 + 4096)));
          bf16x8 vf;
          const bf16x4 b0 = __builtin_bit_cast(bf16x4, t0), b1 = __builtin_bit_cast(bf16x4, t1);
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            vf[r] = b0[r];
            vf[4 + r] = b1[r];
          }
#pragma unroll
          for (int qt = 0; qt < 2; ++qt)
            ot[qt][dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, pa[qt][ks2], ot[qt][dt], 0, 0, 0);
        }
      }
    }
    if (STAMP) asm volatile("" ::"v"(ot[0][0]), "v"(ot[1][7]));
    FA_STAMP(4)  // O^T += V^T P^T
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's DMA pieces of block kb + 1 have landed
    __syncthreads();  // ... and every wave is done with block kb
    FA_STAMP(5)  // barrier + DMA wait
  };
  for (int kb = 0; kb <= kb_last; kb += 2) {
    block(kb, std::integral_constant<int, 0>{});
    if (kb + 1 <= kb_last) block(kb + 1, std::integral_constant<int, 1>{});
  }

  FA_RT(rt_loop)
  // ---- normalise and store: lane owns query row li, d = dt*16 + 4*quad + r
#pragma unroll
  for (int qt = 0; qt < 2; ++qt) {
    const float l = l_acc[qt][0];   // (every register of the tile, in every lane of the row's column, holds the row's sum)
    const float inv = 1.0f / l;
    if (!LASTQ && qabs[qt] < T && qabs[qt] >= P && lse && quad == 0)  // natural-log log-sum-exp of the scaled scores (backward pass)
      lse[(size_t)(vtok0 + qabs[qt]) * nh + h] = (m_run[qt] + __builtin_amdgcn_logf(l)) * 0.6931471805599453f;
    // A lane holds d = 16 dt + 4 quad + r of its row: 8 bytes per tile. v_permlane16_swap (vdst rows 1 / 3 <-> src rows
    // 0 / 2 of 16 lanes) on the packed tiles (2k, 2k+1) leaves even quads with d = 8 (quad/2) .. +7 of tile 2k and odd
    // quads with the same of tile 2k + 1: 4 x 16-byte stores per row instead of 8 x 8 bytes (the tail is store-ISSUE
    // bound). Every lane takes part in the swaps; only rows of this segment store.
    const bool live = LASTQ ? (wave == 0 && qt == 0 && li == 0) : (qabs[qt] < T && qabs[qt] >= P);
    u16* op = out + (size_t)(LASTQ ? prompt : vtok0 + (live ? qabs[qt] : P)) * nh * hd + h * hd + (quad & 1) * 16 + (quad >> 1) * 8;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      unsigned a[2], b[2];
#pragma unroll
      for (int w = 0; w < 2; ++w) {
        a[w] = (unsigned)f2bf(ot[qt][2 * k][2 * w] * inv) | ((unsigned)f2bf(ot[qt][2 * k][2 * w + 1] * inv) << 16);
        b[w] = (unsigned)f2bf(ot[qt][2 * k + 1][2 * w] * inv) | ((unsigned)f2bf(ot[qt][2 * k + 1][2 * w + 1] * inv) << 16);
        const auto sw = __builtin_amdgcn_permlane16_swap(a[w], b[w], false, false);
        a[w] = sw[0];
        b[w] = sw[1];
      }
      if (live) {
        typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
        *reinterpret_cast<u32x4*>(op + k * 32) = u32x4{a[0], a[1], b[0], b[1]};
      }
    }
  }
  if (STAMP) {
    FA_STAMP(6)  // epilogue
    const int wg = blockIdx.x;
    if (wg < 64 && tid == 0) {
      stamp_acc[7] = kb_last + 1;
#pragma unroll
      for (int i = 0; i < 8; ++i) g_attn_stamps[wg * 8 + i] = stamp_acc[i];
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the output stores have left
    FA_RT(rt_exit)
    if (wg < FA_WG_TRACE && tid == 0) {
      unsigned hw, xcc;
      asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
      asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
      unsigned long long* w = g_attn_wg + (size_t)wg * 8;
      w[0] = rt_entry; w[1] = rt_pro; w[2] = rt_loop; w[3] = rt_exit; w[4] = hw; w[5] = xcc; w[6] = kb_last + 1; w[7] = wg;
    }
  }
}


#ifdef LR_EXPERIMENTS
extern "C" int lr_debug_attn_wg_trace(unsigned long long* out, int n) {
  if (!out || n < 1 || n > FA_WG_TRACE * 8) LR_FAIL(LR_EINVAL, "lr_debug_attn_wg_trace: bad arguments");
  LR_CHECK_HIP(hipMemcpyFromSymbol(out, HIP_SYMBOL(g_attn_wg), (size_t)n * sizeof(unsigned long long)));
  return LR_OK;
}
extern "C" int lr_debug_attn_stamps(unsigned long long* out, int n) {
  if (!out || n < 1 || n > 64 * 8) LR_FAIL(LR_EINVAL, "lr_debug_attn_stamps: bad arguments");
  LR_CHECK_HIP(hipMemcpyFromSymbol(out, HIP_SYMBOL(g_attn_stamps), (size_t)n * sizeof(unsigned long long)));
  return LR_OK;
}
#endif

// attention for a list of query tokens only (the last layer needs just each prompt's last token)
int lr_launch_attention_rows(const u16* qkv, u16* out, const int32_t* cu, int B, const int32_t* q_rows,
                             int n_rows, int nh, int nkv, int hd, hipStream_t st) {
  if (n_rows <= 0) return LR_OK;
  if (hd > 256) LR_FAIL(LR_EUNSUPPORTED, "attention: head_dim %d > 256", hd);
  const int items = n_rows * nh;
  hipLaunchKernelGGL(attn_generic_kernel, dim3((items + 3) / 4), dim3(256), 0, st, qkv, out, cu, B, n_rows,
                     nh, nkv, hd, q_rows, (float*)nullptr);
  LR_CHECK_LAUNCH("attn_generic_kernel(rows)");
  return LR_OK;
}

// =============================================================================================
// cu / cu_host: segment starts [S + 1] in packed rows. prefix_len = P > 0: segment 0 is the shared prefix (P rows)
// and segments 1.. continue it (see the kernel); only the MFMA kernel implements that.
int lr_launch_attention(const u16* qkv, u16* out, const int32_t* cu, const int32_t* cu_host,
                        const int32_t* tok_pos, const int32_t* tok_seq, int B, int n_tok, int nh, int nkv,
                        int hd, int variant, void* scratch, hipStream_t st, int prefix_len) {
  float* lse = (float*)scratch;  // optional [n_tok][nh] log-sum-exp output
  (void)tok_pos;
  (void)tok_seq;
  if (n_tok <= 0 || B <= 0) return LR_OK;
  if (nh % nkv != 0) LR_FAIL(LR_EINVAL, "attention: num_heads %d not a multiple of num_kv_heads %d", nh, nkv);
  if (variant == 0) variant = (hd == 128) ? 2 : 1;
  if (prefix_len < 0 || (prefix_len > 0 && (variant != 2 || cu_host[1] - cu_host[0] != prefix_len)))
    LR_FAIL(LR_EINVAL, "attention: shared prefix of %d tokens needs the head_dim-128 MFMA kernel and segment 0 = the prefix",
            prefix_len);
  double work = 0;  // causal QK^T + PV flops of the rows each segment owns
  int maxT = 0;
  for (int b = 0; b < B; ++b) {
    const double P = (prefix_len > 0 && b > 0) ? prefix_len : 0;
    const double T = P + cu_host[b + 1] - cu_host[b];
    work += 4.0 * nh * hd * (T * (T + 1) / 2 - P * (P + 1) / 2);
    maxT = max(maxT, (int)T);
  }
  LrProfScope prof(variant >= 2 ? LR_PROF_ATTN_MFMA : LR_PROF_ATTN_GENERIC, work, st);
  if (variant == 2) {
    if (hd != 128) LR_FAIL(LR_EUNSUPPORTED, "attention variant 2 needs head_dim 128 (got %d)", hd);
    const int mq = (maxT + FA_QROWS - 1) / FA_QROWS;
    if (mq == 0) return LR_OK;
    const long long n_pairs_ll = (long long)B * nh, grid_ll = 8 * ((n_pairs_ll + 7) / 8) * mq;
    if (grid_ll > 0x7fffffffLL) LR_FAIL(LR_EUNSUPPORTED, "attention: %lld workgroups exceed the grid limit", grid_ll);
    if ((long long)n_tok * (nh + 2 * nkv) * hd * 2 > 0x7fffffffLL * 2)
      LR_FAIL(LR_EUNSUPPORTED, "attention: packed qkv of %d tokens exceeds the 4 GiB a buffer descriptor addresses", n_tok);
    const int n_pairs = (int)n_pairs_ll;
    const unsigned grid = (unsigned)grid_ll;
    static bool lds_set[LR_MAX_DEVICES] = {};
    if (int rc = lr_ensure_dynamic_lds(reinterpret_cast<const void*>(attn_mfma128_kernel<false>), 2 * FA_STAGE_BYTES, lds_set))
      return rc;
#ifdef LR_EXPERIMENTS
    static bool lds_set_stamp[LR_MAX_DEVICES] = {};
    const char* stamp_env = getenv("LR_ATTN_STAMPS");
    if (stamp_env && stamp_env[0] == '1') {
      // LR_ATTN_ONE_PER_CU=1: ask for 104 KiB of LDS so that only ONE workgroup fits a CU (what a block costs a lone wave per SIMD)
      const char* one_env = getenv("LR_ATTN_ONE_PER_CU");
      const int lds_bytes = (one_env && one_env[0] == '1') ? 104 * 1024 : 2 * FA_STAGE_BYTES;
      if (int rc = lr_ensure_dynamic_lds(reinterpret_cast<const void*>(attn_mfma128_kernel<true>), 104 * 1024, lds_set_stamp))
        return rc;
      hipLaunchKernelGGL(attn_mfma128_kernel<true>, dim3(grid), dim3(256), lds_bytes, st, qkv, out, cu,
                         prefix_len, nh, nkv, mq, n_pairs, lse);
    } else
#endif
      hipLaunchKernelGGL(attn_mfma128_kernel<false>, dim3(grid), dim3(256), 2 * FA_STAGE_BYTES, st, qkv, out, cu,
                         prefix_len, nh, nkv, mq, n_pairs, lse);
    LR_CHECK_LAUNCH("attn_mfma128_kernel");
  } else if (variant == 1) {
    if (hd > 256) LR_FAIL(LR_EUNSUPPORTED, "attention: head_dim %d > 256", hd);
    const int items = n_tok * nh;
    hipLaunchKernelGGL(attn_generic_kernel, dim3((items + 3) / 4), dim3(256), 0, st, qkv, out, cu, B,
                       n_tok, nh, nkv, hd, (const int32_t*)nullptr, lse);
    LR_CHECK_LAUNCH("attn_generic_kernel");
  } else {
    LR_FAIL(LR_EINVAL, "attention: unknown variant %d here (0 auto, 1 generic, 2 = head_dim-128 MFMA; 3 needs a workspace: "
            "lr_attention_varlen_ws)", variant);
  }
  return LR_OK;
}

// The pruned last layer: kv = [n_tok][2 * nkv * hd] (K | V of every row), q_last = [prompts][nh * hd] rotated query rows
// of each prompt's last token, out_last = [prompts][nh * hd]. cu / cu_host / prefix_len as lr_launch_attention.
int lr_launch_attention_last(const u16* kv, const u16* q_last, u16* out_last, const int32_t* cu, const int32_t* cu_host,
                             int S, int n_tok, int nh, int nkv, int hd, hipStream_t st, int prefix_len) {
  if (n_tok <= 0 || S <= 0) return LR_OK;
  if (hd != 128 || nh % nkv != 0) LR_FAIL(LR_EUNSUPPORTED, "attention (last rows): head_dim 128 and nh %% nkv == 0 only");
  if (prefix_len < 0 || (prefix_len > 0 && cu_host[1] - cu_host[0] != prefix_len))
    LR_FAIL(LR_EINVAL, "attention (last rows): segment 0 must be the %d-token shared prefix", prefix_len);
  double work = 0;
  for (int b = (prefix_len > 0 ? 1 : 0); b < S; ++b) {
    const double T = (prefix_len > 0 ? prefix_len : 0) + cu_host[b + 1] - cu_host[b];
    work += 4.0 * nh * hd * T;
  }
  LrProfScope prof(LR_PROF_ATTN_MFMA, work, st);
  const long long n_pairs_ll = (long long)S * nh, grid_ll = 8 * ((n_pairs_ll + 7) / 8);
  if (grid_ll > 0x7fffffffLL) LR_FAIL(LR_EUNSUPPORTED, "attention: %lld workgroups exceed the grid limit", grid_ll);
  if ((long long)n_tok * 2 * nkv * hd * 2 > 0x7fffffffLL * 2)
    LR_FAIL(LR_EUNSUPPORTED, "attention: packed kv of %d tokens exceeds the 4 GiB a buffer descriptor addresses", n_tok);
  static bool lds_set[LR_MAX_DEVICES] = {};
  if (int rc = lr_ensure_dynamic_lds(reinterpret_cast<const void*>(attn_mfma128_kernel<false, true>), 2 * FA_STAGE_BYTES, lds_set))
    return rc;
  hipLaunchKernelGGL((attn_mfma128_kernel<false, true>), dim3((unsigned)grid_ll), dim3(256), 2 * FA_STAGE_BYTES, st, kv, out_last,
                     cu, prefix_len, nh, nkv, 0, (int)n_pairs_ll, (float*)nullptr, q_last);
  LR_CHECK_LAUNCH("attn_mfma128_kernel<last>");
  return LR_OK;
}

// forward with the softmax statistics kept for llama_attn_bwd.hip (training): variant 0 auto, 1 generic, 2 MFMA
int lr_launch_attention_lse(const u16* qkv, u16* out, float* lse, const int32_t* cu, const int32_t* cu_host, int B,
                            int n_tok, int nh, int nkv, int hd, int variant, hipStream_t st) {
  if (!lse) LR_FAIL(LR_EINVAL, "attention (training): null statistics buffer");
  return lr_launch_attention(qkv, out, cu, cu_host, nullptr, nullptr, B, n_tok, nh, nkv, hd, variant, lse, st, 0);
}
