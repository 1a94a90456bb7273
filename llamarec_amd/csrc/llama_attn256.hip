// llama_attn256.hip -- varlen causal self-attention, head_dim 128, on the one-wave-per-SIMD structure
// (cdna_hip_programming.md, "Fused attention prefill", 4-wave persistent form): attention variant 3.
//
// Replaces flash-attn 2.5.8's varlen forward (train_ranker.py:61) inside the patched LlamaForCausalLM.forward
// (model/llm.py:89-100) like attn_mfma128_kernel (llama_attn.hip) does; this kernel is the one the prefill runs.
//
//  * A workgroup = 4 wave64 = ONE wave per SIMD with the whole 512-register file: a 256-row query tile of one
//    (segment, head), 64 rows per wave as two 32-row halves. 256 persistent workgroups take tiles from per-XCD ticket
//    counters over a device-built item list (attn256_items_kernel): the tiles of one (segment, head) follow each other on
//    ONE XCD, so that their K/V blocks meet in that XCD's L2; heavy tiles first, every pair's lightest tile last.
//  * Query tiles are anchored at the END of a sequence (tile k = rows [T - 256 (k+1), T - 256 k)): the one partial tile
//    is the cheapest (first rows, fewest keys). Key blocks stay at absolute multiples of 64, and every per-row operation
//    (S^T column, row maximum, P column, O^T column) is independent of the other rows of a tile, so a row's bits depend
//    on its own keys only -- not on the tiling, the batch, or whether the shared prefix is stored once (segment 0).
//  * v_mfma_f32_32x32x16_bf16 for both products, everything transposed: S^T = K Q^T puts a query row on a lane
//    (l & 31) with 16 of a tile's 32 keys in its registers; those accumulators, packed pairwise to bf16, ARE the B operand
//    of O^T += V^T P^T in the k order "16 s + 8 (j >> 2) + 4 (l >> 5) + (j & 3)" (guide section 3, "An accumulator tile as the
//    next MFMA's operand"), and the V^T fragments are read with ds_read_b64_tr_b16 in that same order: no cross-lane
//    movement except one v_permlane32_swap per row maximum.
//  * O (128 registers), the Q fragments (64) and the K fragments of the current block (64) live in a[0:255], named
//    literally in the asm statements -- hipcc, given "a" operands, shuttles them through VGPRs around every block. The
//    softmax state, the V^T fragments and P are ordinary C++ values in the 256 arch VGPRs.
//  * The instruction stream of a key block is GENERATED (tools/gen_attn256.py -> llama_attn256_body.inc): 64 MFMAs, every
//    other instruction assigned to one of the 64 gaps. The two halves run half a block apart -- QK(A) | PV(B, previous
//    block) | QK(B) | PV(A) -- so one half's softmax always has the other half's MFMAs to hide behind.
//  * Deferred maximum (guide T13): a row's reference maximum m moves only when the block's maximum exceeds it by more than
//    2^8; P <= 2^8 otherwise. The decision is per ROW (alpha = 1 exactly for the rows that keep m), so it does not couple rows.
//  * K/V blocks arrive by LDS-DMA, K two blocks ahead, V one, into a ring of two slots each; one barrier per block.
//    Q arrives by LDS-DMA into the wave's own 16 KiB, which the epilogue reuses to turn O^T into whole 256-byte rows.
#include <stdlib.h>

#include <type_traits>

#include "llama_kernels.h"
#include "lr_profile.h"

typedef unsigned short u16;
typedef float a2_f16v __attribute__((ext_vector_type(16)));
typedef int a2_int4 __attribute__((ext_vector_type(4)));
typedef long long a2_i64x2 __attribute__((ext_vector_type(2)));
typedef short a2_short4 __attribute__((ext_vector_type(4)));
typedef unsigned a2_u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned a2_u32x2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) char a2_lds;

#ifndef A2_BODY_INC
#define A2_BODY_INC "llama_attn256_body.inc"   // tools/gpu_attn256_abl.sh builds timing-only ablations from other streams
#endif
#define A2_ROWS 256
#define A2_KB 64
#define A2_RING_BYTES 98304               // 3 slots x (K 16 KiB | V 16 KiB): K is requested three blocks ahead, V two
#define A2_QBUF A2_RING_BYTES             // output staging: 4 waves x 8 KiB (32 rows at a time)
#define A2_CTRL (A2_QBUF + 32768)         // ticket words
#define A2_LDS_BYTES (A2_CTRL + 64)
#define A2_THR 8.0f                       // deferred maximum: log2 units

// ---- item list: header int32[16] = {len[8], cap, ...}, counters int32[16] = {ticket[8], done}, items int2[8][cap]
#define A2_HDR_INTS 16
#define A2_CTR_INTS 16

size_t lr_attn256_ws_bytes(int n_tok, int S, int nh) {
  const size_t tiles = (size_t)n_tok / A2_ROWS + (size_t)S + 1;
  const size_t cap = (size_t)((nh + 7) / 8) * tiles;
  return (A2_HDR_INTS + A2_CTR_INTS) * 4 + 8 * cap * 8;
}

// One workgroup. Segment s (T rows incl. the shared prefix, live rows >= P) has nt = ceil((T - P) / 256) tiles counted from
// the END; tiles 0 .. nt-2 are "heavy", tile nt-1 (the first rows) is "light". Stream x (one per XCD) = heads h % 8 == x:
// heavy section: segments ascending, per segment head after head, per head tile 0, 1, ..; then every pair's light tile.
__global__ __launch_bounds__(256) void attn256_items_kernel(const int32_t* cu, int S, int prefix_len, int nh, int32_t* ws,
                                                            int cap) {
  __shared__ int part[256];
  __shared__ int total;
  int32_t* hdr = ws;
  int32_t* ctr = ws + A2_HDR_INTS;
  int2* items = reinterpret_cast<int2*>(ws + A2_HDR_INTS + A2_CTR_INTS);
  const int t = threadIdx.x;
  const int chunk = (S + 255) / 256;
  const int s0 = min(S, t * chunk), s1 = min(S, s0 + chunk);
  auto ntiles = [&](int s) {
    const int P = (prefix_len > 0 && s > 0) ? prefix_len : 0;
    const int T = P + cu[s + 1] - cu[s];
    return (T - P + A2_ROWS - 1) / A2_ROWS;
  };
  int mine = 0;
  for (int s = s0; s < s1; ++s) mine += ntiles(s) - 1;
  part[t] = mine;
  __syncthreads();
  if (t == 0) {
    int run = 0;
    for (int i = 0; i < 256; ++i) {
      const int v = part[i];
      part[i] = run;
      run += v;
    }
    total = run;
  }
  __syncthreads();
  const int htot = total;
  int hpre = part[t];
  for (int s = s0; s < s1; ++s) {
    const int nt = ntiles(s), heavy = nt - 1;
    for (int h = 0; h < nh; ++h) {
      const int x = h & 7, hl = h >> 3, nhx = (nh - x + 7) >> 3;
      int2* st = items + (size_t)x * cap;
      for (int k = 0; k < heavy; ++k) st[hpre * nhx + hl * heavy + k] = make_int2(s, (h << 16) | k);
      st[htot * nhx + s * nhx + hl] = make_int2(s, (h << 16) | (nt - 1));
    }
    hpre += heavy;
  }
  if (t < 8) {
    const int nhx = (nh - t + 7) >> 3;
    hdr[t] = nhx > 0 ? nhx * (htot + S) : 0;
    ctr[t] = 0;
  }
  if (t == 8) {
    hdr[8] = cap;
    ctr[8] = 0;
  }
}

int lr_launch_attn256_items(const int32_t* cu, int S, int n_tok, int nh, int prefix_len, void* ws, size_t ws_bytes,
                            hipStream_t st) {
  const size_t need = lr_attn256_ws_bytes(n_tok, S, nh);
  if (!ws || ws_bytes < need)
    LR_FAIL(LR_EWORKSPACE, "attention (256-row tiles): item list needs %zu bytes, have %zu", need, ws_bytes);
  const int cap = (int)(((size_t)n_tok / A2_ROWS + (size_t)S + 1) * ((nh + 7) / 8));
  hipLaunchKernelGGL(attn256_items_kernel, dim3(1), dim3(256), 0, st, cu, S, prefix_len, nh, (int32_t*)ws, cap);
  LR_CHECK_LAUNCH("attn256_items_kernel");
  return LR_OK;
}

// Every asm statement of the kernel names the WHOLE accumulator file as clobbered. a[0:255] hold O, Q and K across
// statements without hipcc knowing; given the chance it parks values of its own there (v_accvgpr_write / _read around a
// region of high pressure -- seen in the first build of this file: Q fragments overwritten). With the clobber on every
// statement no value of the compiler's can sit in an accumulator register across any of them; tests/test_isa_checks.py
// asserts that the built kernel holds no v_accvgpr_* outside the asm statements and no scratch access.
#define A2_ALLA \
  "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7", "a8", "a9", "a10", "a11", "a12", "a13", "a14", "a15", \
  "a16", "a17", "a18", "a19", "a20", "a21", "a22", "a23", "a24", "a25", "a26", "a27", "a28", "a29", "a30", "a31", \
  "a32", "a33", "a34", "a35", "a36", "a37", "a38", "a39", "a40", "a41", "a42", "a43", "a44", "a45", "a46", "a47", \
  "a48", "a49", "a50", "a51", "a52", "a53", "a54", "a55", "a56", "a57", "a58", "a59", "a60", "a61", "a62", "a63", \
  "a64", "a65", "a66", "a67", "a68", "a69", "a70", "a71", "a72", "a73", "a74", "a75", "a76", "a77", "a78", "a79", \
  "a80", "a81", "a82", "a83", "a84", "a85", "a86", "a87", "a88", "a89", "a90", "a91", "a92", "a93", "a94", "a95", \
  "a96", "a97", "a98", "a99", "a100", "a101", "a102", "a103", "a104", "a105", "a106", "a107", "a108", "a109", "a110", "a111", \
  "a112", "a113", "a114", "a115", "a116", "a117", "a118", "a119", "a120", "a121", "a122", "a123", "a124", "a125", "a126", "a127", \
  "a128", "a129", "a130", "a131", "a132", "a133", "a134", "a135", "a136", "a137", "a138", "a139", "a140", "a141", "a142", "a143", \
  "a144", "a145", "a146", "a147", "a148", "a149", "a150", "a151", "a152", "a153", "a154", "a155", "a156", "a157", "a158", "a159", \
  "a160", "a161", "a162", "a163", "a164", "a165", "a166", "a167", "a168", "a169", "a170", "a171", "a172", "a173", "a174", "a175", \
  "a176", "a177", "a178", "a179", "a180", "a181", "a182", "a183", "a184", "a185", "a186", "a187", "a188", "a189", "a190", "a191", \
  "a192", "a193", "a194", "a195", "a196", "a197", "a198", "a199", "a200", "a201", "a202", "a203", "a204", "a205", "a206", "a207", \
  "a208", "a209", "a210", "a211", "a212", "a213", "a214", "a215", "a216", "a217", "a218", "a219", "a220", "a221", "a222", "a223", \
  "a224", "a225", "a226", "a227", "a228", "a229", "a230", "a231", "a232", "a233", "a234", "a235", "a236", "a237", "a238", "a239", \
  "a240", "a241", "a242", "a243", "a244", "a245", "a246", "a247", "a248", "a249", "a250", "a251", "a252", "a253", "a254", "a255"

// ---- helpers ---------------------------------------------------------------------------------------------------
__device__ __forceinline__ float a2_max3(float a, float b, float c) {   // raw MFMA outputs: no canonicalising v_max x, x
  float r;
  asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
  return r;
}
__device__ __forceinline__ float a2_max2(float a, float b) {
  float r;
  asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ __forceinline__ void a2_swap32(float v, float& a, float& b) {   // both lane halves' values to every lane
  const unsigned u = __builtin_bit_cast(unsigned, v);
  const auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
  const unsigned r0 = r[0], r1 = r[1];
  a = __builtin_bit_cast(float, r0);
  b = __builtin_bit_cast(float, r1);
}
__device__ __forceinline__ a2_int4 a2_make_rsrc(const void* base, int num_records) {
  const unsigned long long b = reinterpret_cast<unsigned long long>(base);
  a2_int4 r;
  r[0] = __builtin_amdgcn_readfirstlane((int)(unsigned)b);
  r[1] = __builtin_amdgcn_readfirstlane((int)((b >> 32) & 0xffffu));   // stride 0
  r[2] = __builtin_amdgcn_readfirstlane(num_records);
  r[3] = 0x00020000;
  return r;
}
// LDS-DMA as inline asm (hipcc must not count it: see llama_attn.hip); M0 is written in the statement that reads it
// soff: 0 / 64 / 128 / 192, the 64-byte column group of the piece (memory side only; the range check of a raw buffer is on voff)
__device__ __forceinline__ void a2_dma16(a2_int4 rsrc, unsigned lds_dst, unsigned voff, int soff) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" ::"s"(lds_dst), "v"(voff), "s"(rsrc), "s"(soff)
               : "memory", A2_ALLA);
}
__device__ __forceinline__ void a2_glds16(const void* sbase, unsigned voff, unsigned lds_dst) {   // scalar base + per-lane offset
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(lds_dst), "v"(voff), "s"(sbase) : "memory", A2_ALLA);
}

// End of a key block: everything requested BEFORE this block's own 8 pieces has landed (K three blocks ahead, V two: a request has
// two blocks to arrive). A tile's first block: its 8 pieces were requested behind the previous tile (pre_issue) and the wave's 16
// output-row stores of that tile are younger still.
#define A2_BARRIER_8() asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)\n\ts_barrier" ::: "memory", A2_ALLA)
#define A2_BARRIER_24() asm volatile("s_waitcnt vmcnt(24) lgkmcnt(0)\n\ts_barrier" ::: "memory", A2_ALLA)
#define A2_BARRIER() asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory", A2_ALLA)

#ifdef A2_STAMPS   // diagnostic build only (tools/build_attn256_abl.sh stamps): the stamping code lives in tools/diag/attn256_stamps.inc
#ifndef A2_STAMPS_INC
#define A2_STAMPS_INC "../../tools/diag/attn256_stamps.inc"
#endif
#define A2_STAMP_DECL
#include A2_STAMPS_INC
#undef A2_STAMP_DECL
#else
#define A2_PHASE(i)
#endif

// =====================================================================================================================
__global__ __launch_bounds__(256) void attn_mfma256_kernel(const u16* __restrict__ qkv, u16* out, const int32_t* cu,
                                                           int prefix_len, int nh, int nkv, const int32_t* ws_ro,
                                                           int32_t* ctr, float* lse, unsigned qkv_bytes) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int hd = 128;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 31, hi = lane >> 5;
  const int stride = (nh + 2 * nkv) * hd;
  // (the softmax scale 1/sqrt(128) * log2(e) = 0.12754 = 0x3e0293ee is an immediate of the generated stream)
  const unsigned lds0 = (unsigned)(size_t)(a2_lds*)smem;
  const int2* items = reinterpret_cast<const int2*>(ws_ro + A2_HDR_INTS + A2_CTR_INTS);

  // ---- LDS images (no XOR swizzle: every read address is ONE lane constant + an immediate)
  //  K tile (and the wave's Q rows): 16-row groups; inside a group chunk-major: (row, 16-byte chunk c) at
  //    (row >> 4) * 4096 + c * 256 + (row & 15) * 16 -- a ds_read_b128 of 16 consecutive rows' chunk c is 256 contiguous bytes
  //  V tile: 64-byte column groups G (= the 32 dims of O^T tile dt): (key, byte b of the row) at
  //    (b >> 6) * 4096 + key * 64 + (b & 63) -- a ds_read_b64_tr_b16 half-wave (4 keys x 64 bytes) is 256 contiguous bytes
  //  LDS-DMA pieces (1 KiB, lane-linear): K piece (row group g, chunk quad p): lane L <- row 16 g + (L & 15), chunk 4 p + (L >> 4);
  //  V piece (column group G, key group g): lane L <- key 16 g + (L >> 2), bytes 64 G + 16 (L & 3). Wave w moves group w of both,
  //  so the per-lane source offset is one constant per operand and the piece index is the load's scalar offset (64 p).
  const unsigned kaddr = lds0 + (r >> 4) * 4096 + (r & 15) * 16 + hi * 256;           // + 512 ks + 8192 kt + slot
  const unsigned qoff = A2_QBUF + wave * 8192;                                       // the wave's own output staging rows
  unsigned vaddr;                                                                    // + 4096 dt + 1024 s + 512 jj + slot
  {
    const int i = lane & 15, q = i >> 2, p = i & 3, g1 = (lane >> 4) & 1;
    vaddr = lds0 + (4 * hi + q) * 64 + 32 * g1 + 8 * p;
  }
  const unsigned koff = (unsigned)((16 * wave + (lane & 15)) * stride) * 2u + (lane >> 4) * 16;
  const unsigned voff = (unsigned)((16 * wave + (lane >> 2)) * stride + nkv * hd) * 2u + (lane & 3) * 16;   // V sits nkv * hd columns behind K
  const int prow = lane >> 4, ppos = lane & 15;   // epilogue: row-major rows
  unsigned xcc;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  xcc &= 7;

  // ---- softmax state (kernel scope: the stream of a block reaches into the next one)
  a2_f16v S[2][2];
  a2_int4 Pf[2][4];
  a2_i64x2 Vf[4][4];
  float t_[2][2], p_[2][4], bt_[2], rt_[8];          // rotating temporaries of the generated stream
  float m_run[2], lsum[2], negm[2], alpha[2], mx[2], mthr[2];   // lsum: the running row sum l; mthr: (m + 2^THR) / scale
  int thr[2];
  const float ninf = -__builtin_inff();
  const int soff[4] = {0, 64, 128, 192};
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    m_run[h] = lsum[h] = negm[h] = alpha[h] = mx[h] = mthr[h] = 0.f;
    thr[h] = 0;
    t_[h][0] = t_[h][1] = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      Pf[h][i] = a2_int4{0, 0, 0, 0};
      p_[h][i] = 0.f;
    }
  }
  bt_[0] = bt_[1] = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i) rt_[i] = 0.f;
#pragma unroll
  for (int dt = 0; dt < 4; ++dt)
#pragma unroll
    for (int s = 0; s < 4; ++s) Vf[dt][s] = a2_i64x2{0, 0};

#ifdef A2_STAMPS
#define A2_STAMP_STATE
#include A2_STAMPS_INC
#undef A2_STAMP_STATE
#endif
  // ---- tickets. Own XCD's stream first (the tiles of one (segment, head) follow each other there: their K/V blocks meet in
  // that XCD's L2), then the neighbours' leftovers. A workgroup always knows its current AND its next tile (the K / V / Q stream
  // runs across the seam); the ticket of the tile after that is drawn at the start of a tile and read at its end.
  const int cap = ws_ro[8], len_own = ws_ro[xcc];
  auto steal = [&]() {
    for (int a = 1; a < 8; ++a) {
      const int s = ((int)xcc + a) & 7, len = ws_ro[s];
      if (len <= 0) continue;
      const int j = atomicAdd(&ctr[s], 1);
      if (j < len) return s * cap + j;
    }
    return -1;
  };
  auto draw = [&]() {
    const int j = len_own > 0 ? atomicAdd(&ctr[xcc], 1) : 0;
    return j < len_own ? (int)xcc * cap + j : steal();
  };
  volatile int* const ctrl = reinterpret_cast<volatile int*>(smem + A2_CTRL);
  if (tid == 0) ctrl[0] = draw();
  __syncthreads();

  // ---- a tile's scalars
  struct Tile {
    int T, P, row0, kb_wg, vtok0, hcol;   // hcol: the head's first q column (elements)
    const char* kbase;                    // K element 0 of the sequence's position 0 in this head (own rows: position p >= P)
    int rec0;                             // bytes from there to the end of the sequence's last V row
    int kcol;                             // the head's first K column (elements)
    bool valid;
  };
  const int kv_tail = (nkv * hd + hd) * 2;                        // bytes from a row's K element 0 to the end of its V row
  const unsigned blk_bytes = (unsigned)A2_KB * stride * 2;        // one key block of rows
  auto decode = [&](int item) {
    Tile t;
    t.valid = item >= 0;
    // a SCALAR load: hipcc reads the list with a vector load (it cannot prove the words invariant beside the ticket atomics), and
    // a vector load's wait is vmcnt(0) -- placed in the next tile's first block, in front of the first use, where it waited for the
    // previous tile's 16 output stores (the first block took 8.4 k cycles instead of ~3 k: profiles/r05_attn256_stamps.txt)
    int2 it;
    {
      unsigned long long v_;
      asm volatile("s_load_dwordx2 %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(v_) : "s"(items + (t.valid ? item : 0)) : "memory");
      it.x = (int)(unsigned)v_;
      it.y = (int)(unsigned)(v_ >> 32);
    }
    const int seg = __builtin_amdgcn_readfirstlane(it.x);
    const int h = __builtin_amdgcn_readfirstlane(it.y >> 16), tk = __builtin_amdgcn_readfirstlane(it.y & 0xffff);
    const int tok0 = cu[seg];
    t.P = (prefix_len > 0 && seg > 0) ? prefix_len : 0;           // keys [0, P) live in segment 0's rows [0, P)
    t.T = t.P + cu[seg + 1] - tok0;
    t.row0 = t.T - A2_ROWS * (tk + 1);                            // may be negative for a sequence's first tile
    t.kb_wg = (t.row0 + A2_ROWS - 1) >> 6;                        // the tile's last key block
    t.vtok0 = tok0 - t.P;                                         // the row of position p >= P is vtok0 + p
    t.hcol = h * hd;
    t.kcol = (nh + h / (nh / nkv)) * hd;
    t.kbase = reinterpret_cast<const char*>(qkv + (size_t)t.vtok0 * stride + t.kcol);
    t.rec0 = (t.T - 1) * stride * 2 + kv_tail;
    return t;
  };
  // descriptor of key block kbx of a tile's OWN rows: rows past T - 1 are range-checked to zero; kbx > kb_wg: nothing
  auto own_rsrc = [&](const Tile& t, int kbx) {
    const unsigned off = (unsigned)kbx * blk_bytes;
    int rec = t.rec0 - (int)off;
    if (kbx > t.kb_wg || !t.valid) rec = 0;
    return a2_make_rsrc(t.kbase + off, rec);
  };
  // Source of key block kbx of the STREAM that starts at tile `c` and runs on into tile `n` (its blocks 0 .. jmax only): the
  // descriptor, and the per-lane byte offset to ADD to koff / voff (non-zero only for block 0 of a tile that shares a prefix:
  // one descriptor over the whole buffer, own rows shifted by the segment's first row; a key past the sequence end then reads
  // the following rows -- finite, and causally masked for every live query row -- where an own-rows descriptor zero-fills).
  auto stream_src = [&](const Tile& c, const Tile& n, int kbx, int jmax, int key_lane, a2_int4& rs, unsigned& delta) {
    delta = 0;
    if (kbx <= c.kb_wg) {
      rs = own_rsrc(c, kbx);
      return;
    }
    const int j = kbx - c.kb_wg - 1;
    if (!n.valid || j > jmax || j > n.kb_wg) {
      rs = a2_make_rsrc(qkv, 0);
    } else if (j == 0 && n.P > 0) {
      const long long left = (long long)qkv_bytes - (long long)n.kcol * 2;
      rs = a2_make_rsrc(reinterpret_cast<const char*>(qkv + n.kcol), (int)(left > 0x7fffffffLL ? 0x7fffffffLL : left));
      delta = key_lane >= n.P ? (unsigned)n.vtok0 * (unsigned)stride * 2u : 0u;
    } else {
      rs = own_rsrc(n, j);
    }
  };
  const int klane_k = 16 * wave + (lane & 15), klane_v = 16 * wave + (lane >> 2);   // the key (in its block) a lane stages

  // synchronous staging of one block with per-lane source addresses (a workgroup's first tile; block 0 behind a one-block tile):
  // keys < P come from segment 0's rows, keys >= T are clamped to T - 1
  auto stage_sync = [&](const Tile& t, int kbx, int slot, bool do_k, bool do_v) {
    const unsigned dst = lds0 + slot * 32768 + wave * 4096;   // slot 0 .. 2
    const int kk = min(kbx * A2_KB + klane_k, t.T - 1), kv = min(kbx * A2_KB + klane_v, t.T - 1);
    const unsigned ok = (unsigned)(((kk < t.P ? kk : t.vtok0 + kk) * stride + t.kcol) * 2 + (lane >> 4) * 16);
    const unsigned ov = (unsigned)(((kv < t.P ? kv : t.vtok0 + kv) * stride + t.kcol + nkv * hd) * 2 + (lane & 3) * 16);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      if (do_k) a2_glds16(qkv, ok + i * 64, dst + i * 1024);
      if (do_v) a2_glds16(qkv, ov + i * 64, dst + 16384 + i * 4096 - wave * 3072);
    }
  };
  // Q fragments of a tile straight from global memory into a[128:191] (B operand: lane = row 32 half + (l & 31), 16 bytes at
  // 32 ks + 16 (l >> 5)); rows outside [P, T) are clamped (computed, never stored)
  auto q_load = [&](const Tile& t) {
    const int q0 = t.row0 + 64 * wave;
    unsigned qvo[2];
#pragma unroll
    for (int hf = 0; hf < 2; ++hf) {
      const int qr = min(max(q0 + 32 * hf + r, t.P), t.T - 1);
      qvo[hf] = (unsigned)((t.vtok0 + qr) * stride + t.hcol) * 2u + hi * 16;
    }
#define A2_EMIT_QGLOAD
#include A2_BODY_INC
#undef A2_EMIT_QGLOAD
  };
  auto k_load = [&](int slot) {   // K fragments of the block in ring slot `slot` -> a[192:255]
    const unsigned kaddr_n = kaddr + slot * 32768;
#define A2_EMIT_KLOAD
#include A2_BODY_INC
#undef A2_EMIT_KLOAD
  };

  const __amdgpu_buffer_rsrc_t out_rsrc = __builtin_amdgcn_make_buffer_rsrc(
      out, 0, (int)((size_t)qkv_bytes / (unsigned)(nh + 2 * nkv) * (unsigned)nh), 0x00020000);
  // A workgroup draws its SECOND ticket only after its first tile's requests are on their way: two draws in a row would hand it
  // two consecutive items -- two tiles of ONE (segment, head) -- which then run one after the other on this CU instead of side
  // by side on two CUs of the XCD, and their K/V blocks no longer meet in L2 (measured: L2 hit rate 0.58 -> 0.40).
  Tile cur = decode(__builtin_amdgcn_readfirstlane(ctrl[0]));
  int sl0 = 0;             // ring slot (0 .. 2) of the current tile's block 0: the workgroup's blocks take the slots in turn
  int prev_blocks = 0;     // key blocks of the previous tile (0: none): what of the current tile was streamed in behind it
  bool had_epilogue = false;   // this wave stored output rows at the end of the previous tile
  // K(3) and V(2) of a tile -- what its first block would request -- are requested behind the previous tile's last barrier
  // instead: in front of that tile's output stores in the wave's vector-memory queue, so that the first block's wait leaves those
  // stores in flight (they take 6-8 k cycles to drain: measured)
  auto nx = [](int sl) { return sl == 2 ? 0 : sl + 1; };   // ring slot of the following block
  auto pre_issue = [&](const Tile& t, int s0) {             // s0: the ring slot of the tile's block 0
    const a2_int4 rs_k = own_rsrc(t, 3), rs_v = own_rsrc(t, 2);
    const unsigned dst_k = lds0 + s0 * 32768 + wave * 4096, dst_v = lds0 + nx(nx(s0)) * 32768 + 16384 + wave * 1024;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      a2_dma16(rs_k, dst_k + i * 1024, koff, i * 64);
      a2_dma16(rs_v, dst_v + i * 4096, voff, i * 64);
    }
  };
  if (cur.valid && cur.row0 + 64 * wave + 63 >= cur.P) q_load(cur);   // the workgroup's first tile: nothing was streamed in
  if (tid == 0) ctrl[1] = cur.valid ? draw() : -1;
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier\n\ts_waitcnt lgkmcnt(0)" ::: "memory", A2_ALLA);
  Tile nxt = decode(__builtin_amdgcn_readfirstlane(ctrl[1]));
  for (int tile_no = 0; cur.valid; ++tile_no) {
    int item_nn = -1;      // the tile after next (drawn by wave 0 in its slack behind its last key block)
    const int P = cur.P, row0 = cur.row0, kb_wg = cur.kb_wg, vtok0 = cur.vtok0;
    const int q0 = row0 + 64 * wave;
    const bool dead = q0 + 63 < P;                                // no live row in this wave
    const int n_full = q0 >= 0 ? (q0 + 1) >> 6 : 0;               // blocks every row of the wave sees unmasked
    const int kl = dead ? -1 : (q0 + 63) >> 6;                    // the wave's last block
    if (prev_blocks < 4) {
      // Not everything was streamed in behind the previous tile: K(0) needs a previous tile of >= 4 blocks (requested in its block
      // kb_wg - 2, which must not be ITS pre-issued block 0), K(1) and V(0) one of >= 3, K(2) and V(1) one of >= 2. Stage what is
      // missing now, then every wave reads K(0) BEFORE this tile's K(3) goes into the same ring slot.
      stage_sync(cur, 0, sl0, true, prev_blocks < 3);
      if (prev_blocks < 3) stage_sync(cur, 1, nx(sl0), true, prev_blocks < 2);
      if (prev_blocks < 2) stage_sync(cur, 2, nx(nx(sl0)), true, false);
      A2_BARRIER();
      if (!dead) k_load(sl0);
      A2_BARRIER();
      pre_issue(cur, sl0);
      had_epilogue = false;   // (the stores are behind us: the first block leaves only its own requests in flight)
    }
    m_run[0] = m_run[1] = mthr[0] = mthr[1] = -__builtin_inff();   // every row's maximum moves in its first block
    negm[0] = negm[1] = lsum[0] = lsum[1] = 0.f;
    A2_PHASE(0)   // seam

    // ================================================================ one key block
    int sl = sl0;   // ring slot of the block about to run: K(kb), V(kb) and the request K(kb + 3) use it; K(kb + 1) the next one; the
                    // request V(kb + 2) the one after (V(kb - 1)'s)
    auto body = [&](auto first_c, auto diag_c, const int kb) {
      constexpr bool FIRST = decltype(first_c)::value, DIAG = decltype(diag_c)::value;
      a2_int4 rs_k, rs_v;
      unsigned dk, dv;
      stream_src(cur, nxt, kb + 3, 2, klane_k, rs_k, dk);
      stream_src(cur, nxt, kb + 2, 1, klane_v, rs_v, dv);
      const unsigned koff_x = koff + dk, voff_x = voff + dv;
      const int sn = nx(sl), sp = nx(sn);
      const unsigned dst_k = lds0 + sl * 32768 + wave * 4096, dst_v = lds0 + sp * 32768 + 16384 + wave * 1024;
      const unsigned kaddr_n = kaddr + sn * 32768, vaddr_c = vaddr + sl * 32768 + 16384;
      sl = sn;
      (void)rt_; (void)bt_; (void)t_; (void)p_; (void)soff; (void)ninf; (void)kaddr_n; (void)vaddr_c; (void)koff_x; (void)voff_x;
      if constexpr (DIAG) {
        thr[0] = max(q0 + r, 0) - kb * A2_KB - 4 * hi;
        thr[1] = max(q0 + 32 + r, 0) - kb * A2_KB - 4 * hi;
      }
#ifdef A2_STAMPS
#define A2_STAMP_BLOCK_ENTRY
#include A2_STAMPS_INC
#undef A2_STAMP_BLOCK_ENTRY
#endif
      if constexpr (FIRST && DIAG) {
#define A2_EMIT_BODY_11
#include A2_BODY_INC
#undef A2_EMIT_BODY_11
      } else if constexpr (FIRST) {
#define A2_EMIT_BODY_10
#include A2_BODY_INC
#undef A2_EMIT_BODY_10
      } else if constexpr (DIAG) {
#define A2_EMIT_BODY_01
#include A2_BODY_INC
#undef A2_EMIT_BODY_01
      } else {
#define A2_EMIT_BODY_00
#include A2_BODY_INC
#undef A2_EMIT_BODY_00
      }
#ifdef A2_STAMPS
#define A2_STAMP_BLOCK_END   // (ends in a dangling `else` in front of the barrier below)
#include A2_STAMPS_INC
#undef A2_STAMP_BLOCK_END
#endif
      if (FIRST && had_epilogue) A2_BARRIER_24(); else A2_BARRIER_8();
    };
    auto idle = [&](const int kb) {   // a wave with no work in this block still moves its share of K(kb + 3) and V(kb + 2)
      a2_int4 rs_k, rs_v;
      unsigned dk, dv;
      stream_src(cur, nxt, kb + 3, 2, klane_k, rs_k, dk);
      stream_src(cur, nxt, kb + 2, 1, klane_v, rs_v, dv);
      const int sn = nx(sl), sp = nx(sn);
      const unsigned dst_k = lds0 + sl * 32768 + wave * 4096, dst_v = lds0 + sp * 32768 + 16384 + wave * 1024;
      if (kb > 0) {   // (block 0's requests were made behind the previous tile: pre_issue)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          a2_dma16(rs_k, dst_k + i * 1024, koff + dk, i * 64);
          a2_dma16(rs_v, dst_v + i * 4096, voff + dv, i * 64);
        }
      }
      // the tile's last block: K(0) of the next tile (streamed in two blocks ago) goes to the fragment registers before the seam's
      // barrier, like the waves that compute this block do in its last gaps
      if (kb == kb_wg && kb_wg >= 3 && nxt.valid) k_load(sn);
      sl = sn;
      if (kb == 0 && had_epilogue) A2_BARRIER_24(); else A2_BARRIER_8();
    };
    using BF = std::false_type;
    using BT = std::true_type;
    // the ticket of the tile after next: drawn by wave 0 in its slack (it has the fewest key blocks of a tile); the atomic's round trip
    // is waited for on the spot (hipcc's wave-aggregated form), which a wave in front of its key blocks cannot afford
    auto draw_next = [&]() {
      if (wave == 0 && nxt.valid) {
        if (tid == 0) item_nn = draw();
      }
    };
    if (dead) {
      draw_next();
      for (int kb = 0; kb <= kb_wg; ++kb) idle(kb);
      A2_PHASE(2)
    } else {
#ifdef A2_STAMPS
#define A2_STAMP_TIMED
#include A2_STAMPS_INC
#undef A2_STAMP_TIMED
#else
      if (n_full == 0) body(BT{}, BT{}, 0); else body(BT{}, BF{}, 0);
      for (int kb = 1; kb < n_full; ++kb) body(BF{}, BF{}, kb);               // the steady state: ONE instance, a self-loop
      for (int kb = max(n_full, 1); kb <= kl; ++kb) body(BF{}, BT{}, kb);     // the one or two blocks the diagonal crosses
#endif
      A2_PHASE(1)   // the wave's key blocks
      {   // what the wave still owes after its last block: the rest of B's softmax and PV(B, kl)
#define A2_EMIT_DRAIN
#include A2_BODY_INC
#undef A2_EMIT_DRAIN
      }
      draw_next();
    }
    // the next tile's Q fragments: their registers are free behind the wave's last score MFMA; the loads land beside the
    // staging-only blocks and the epilogue
    const bool q_next = nxt.valid && nxt.row0 + 64 * wave + 63 >= nxt.P;
    if (q_next) q_load(nxt);
    if (!dead) {
      for (int kb = kl + 1; kb <= kb_wg; ++kb) idle(kb);
      A2_PHASE(2)   // drain + the blocks the wave only stages for
    }
    // behind the tile's last barrier: the next tile's first requests (its K(0) is in the fragment registers of every wave by now)
    const bool streamed = kb_wg >= 3 && nxt.valid;
    if (streamed) pre_issue(nxt, sl);   // (sl has walked past the tile's last block: the next tile's block 0)
    if (dead && q_next) {
      if (streamed) asm volatile("s_waitcnt vmcnt(8)" ::: "memory", A2_ALLA); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory", A2_ALLA);
    }
    if (!dead) {
      // ============================================================== epilogue: O^T / l -> bf16 rows via the wave's own LDS region -> global
      int z_;   // an opaque zero: keeps this lane arithmetic (loop-invariant across tiles) out of the key-block loop's live ranges
      asm volatile("v_mov_b32 %0, 0" : "=v"(z_));
      asm volatile("s_nop 15\n\ts_nop 15" ::: "memory", A2_ALLA);   // the last asm MFMAs' results (the compiler pads nothing)
      float inv[2];
#pragma unroll
      for (int hf = 0; hf < 2; ++hf) {
        float a_, b_;
        a2_swap32(lsum[hf], a_, b_);
        const float l = a_ + b_;
        inv[hf] = 1.0f / l;
        const int q = q0 + 32 * hf + r;
        if (lse && hi == 0 && q >= P) lse[(size_t)(vtok0 + q) * nh + (cur.hcol >> 7)] = (m_run[hf] + __builtin_amdgcn_logf(l)) * 0.6931471805599453f;
      }
      a2_lds* const obase = (a2_lds*)smem + qoff;   // 32 rows x 256 B, chunk c of row r at position c ^ (r & 15)
      float ov[16];
      // lane: query row 32 hf + r, d = 32 dt + 8 g + 4 hi + 0..3 in registers 4g .. 4g+3: 8 bytes of 16-byte chunk 4 dt + g
#define A2_OSTORE(hf, dt)                                                                                          \
  {                                                                                                                \
    _Pragma("unroll") for (int g = 0; g < 4; ++g) {                                                                \
      a2_u32x2 w_;                                                                                                 \
      w_[0] = (unsigned)f2bf(ov[4 * g] * inv[hf]) | ((unsigned)f2bf(ov[4 * g + 1] * inv[hf]) << 16);               \
      w_[1] = (unsigned)f2bf(ov[4 * g + 2] * inv[hf]) | ((unsigned)f2bf(ov[4 * g + 3] * inv[hf]) << 16);           \
      *reinterpret_cast<__attribute__((address_space(3))) a2_u32x2*>(                                              \
          obase + (r | z_) * 256 + (((4 * dt + g) ^ (r & 15)) << 4) + 8 * hi) = w_;                                \
    }                                                                                                              \
  }
      // 8 stores per half and lane, 16 in all, ALWAYS issued (the next tile's first block counts on them: vmcnt(24)): a row that is
      // not this segment's gets an offset past the buffer's end, which the range check drops. Lane ppos takes the row's LOGICAL
      // chunk ppos (it sits at position ppos ^ (row & 15) of the staged row): the 16 lanes of a row store 256 bytes in lane order.
      auto flush_half = [&](int hf) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          const int row = 4 * k + (prow | z_), q = q0 + 32 * hf + row;
          const a2_u32x4 v = *reinterpret_cast<__attribute__((address_space(3))) a2_u32x4*>(obase + row * 256 + ((ppos ^ (row & 15)) << 4));
          const unsigned off = q >= P ? (unsigned)(((vtok0 + q) * nh * hd + cur.hcol + ppos * 8) * 2) : 0xfffffff0u;
          __builtin_amdgcn_raw_buffer_store_b128(v, out_rsrc, off, 0, 0);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
      };
#define A2_EMIT_OREAD_0
#include A2_BODY_INC
#undef A2_EMIT_OREAD_0
      // the next tile's Q fragments have landed (requested behind the wave's last key block) -- waited for HERE, in front of the
      // output stores, so that nothing behind has to wait for those stores
      if (streamed) asm volatile("s_waitcnt vmcnt(8)" ::: "memory", A2_ALLA); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory", A2_ALLA);
      flush_half(0);
#define A2_EMIT_OREAD_1
#include A2_BODY_INC
#undef A2_EMIT_OREAD_1
      flush_half(1);
    }
    had_epilogue = !dead;
    A2_PHASE(3)   // epilogue
    // ---- the seam: the ticket of the tile after next becomes visible; every wave is done with this tile's ring slots. LDS only: the
    // output rows' stores drain beside the next tile (a __syncthreads() here waits for them: 6-8 k cycles per tile, measured)
    if (tid == 0) ctrl[tile_no & 1] = item_nn;
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier\n\ts_waitcnt lgkmcnt(0)" ::: "memory", A2_ALLA);
    prev_blocks = kb_wg + 1;
    sl0 = sl;
    cur = nxt;
    nxt = decode(__builtin_amdgcn_readfirstlane(ctrl[tile_no & 1]));
    A2_PHASE(4)   // ticket + rendezvous
#ifdef A2_STAMPS
#define A2_STAMP_TILE_END
#include A2_STAMPS_INC
#undef A2_STAMP_TILE_END
#endif
  }
#ifdef A2_STAMPS
#define A2_STAMP_STORE
#include A2_STAMPS_INC
#undef A2_STAMP_STORE
#endif
  // ---- the last workgroup to leave re-arms the counters for the next launch over the same item list
  if (tid == 0) {
    const int d = atomicAdd(&ctr[8], 1);
    if (d == (int)gridDim.x - 1)
      for (int i = 0; i < 9; ++i) atomicExch(&ctr[i], 0);
  }
}

// =====================================================================================================================
// cu / cu_host / prefix_len as lr_launch_attention; items_ws = lr_launch_attn256_items' output for the same cu.
int lr_launch_attention256(const u16* qkv, u16* out, const int32_t* cu, const int32_t* cu_host, int S, int n_tok, int nh,
                           int nkv, int hd, float* lse, void* items_ws, hipStream_t st, int prefix_len) {
  if (n_tok <= 0 || S <= 0) return LR_OK;
  if (hd != 128 || nh % nkv != 0 || nh > 0xffff) LR_FAIL(LR_EUNSUPPORTED, "attention (256-row tiles): head_dim 128, nh %% nkv == 0 only");
  if (prefix_len < 0 || prefix_len > A2_KB || (prefix_len > 0 && cu_host[1] - cu_host[0] != prefix_len))
    LR_FAIL(LR_EINVAL, "attention (256-row tiles): shared prefix of %d tokens (<= 64, = segment 0)", prefix_len);
  if ((long long)n_tok * (nh + 2 * nkv) * hd * 2 > 0x7fffffffLL)
    LR_FAIL(LR_EUNSUPPORTED, "attention (256-row tiles): packed qkv of %d tokens exceeds 2 GiB (32-bit byte offsets)", n_tok);
  if (!items_ws) LR_FAIL(LR_EINVAL, "attention (256-row tiles): no item list");
  double work = 0;
  for (int b = 0; b < S; ++b) {
    const double P = (prefix_len > 0 && b > 0) ? prefix_len : 0;
    const double T = P + cu_host[b + 1] - cu_host[b];
    work += 4.0 * nh * hd * (T * (T + 1) / 2 - P * (P + 1) / 2);
  }
  LrProfScope prof(LR_PROF_ATTN_MFMA, work, st);
  int dev = 0, cus = 0;
  LR_CHECK_HIP(hipGetDevice(&dev));
  static int cu_count[LR_MAX_DEVICES] = {};
  if (dev >= 0 && dev < LR_MAX_DEVICES && cu_count[dev] > 0) {
    cus = cu_count[dev];
  } else {
    LR_CHECK_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
    if (dev >= 0 && dev < LR_MAX_DEVICES) cu_count[dev] = cus;
  }
  int32_t* ws = (int32_t*)items_ws;
  static bool lds_set[LR_MAX_DEVICES] = {};
  if (int rc = lr_ensure_dynamic_lds(reinterpret_cast<const void*>(attn_mfma256_kernel), A2_LDS_BYTES, lds_set)) return rc;
  hipLaunchKernelGGL(attn_mfma256_kernel, dim3(cus), dim3(256), A2_LDS_BYTES, st, qkv, out, cu, prefix_len, nh, nkv,
                     (const int32_t*)ws, ws + A2_HDR_INTS, lse, (unsigned)((size_t)n_tok * (nh + 2 * nkv) * hd * 2));
  LR_CHECK_LAUNCH("attn_mfma256_kernel");
  return LR_OK;
}
