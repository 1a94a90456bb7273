// Prototype: the bf16 NT GEMM's K loop with FOUR waves per workgroup, one per SIMD, each owning a
// 128 x 128 piece of the 256 x 256 tile (64 accumulator tiles of 16 x 16 in AGPRs), against the product's eight waves of
// 128 x 64 in a two-wave ping-pong (llamarec_amd/csrc/llama_gemm.hip). Same 256 x 256 x 64 K tile, same LDS image (rows of
// 128 B, 16-byte chunks XOR-swizzled by (row >> 1) & 7), same v_mfma_f32_16x16x32_bf16 in the same k order, plain bf16
// store. What differs is who overlaps what: here ONE instruction stream per SIMD carries the 128 MFMAs of a K tile with the
// 32 fragment reads and 16 LDS-DMA pieces placed between them by hand (4 MFMAs | 1 ds_read | 1 DMA per group), a third
// fewer LDS bytes per MFMA, one barrier per K tile.
//   C[M][N] = A[M][K] B[N][K]^T, bf16 in, fp32 accumulate, bf16 out; M, N multiples of 256, K of 64.
// hipcc --offload-arch=gfx950 -O3 -o /tmp/gemm4w tools/diag/gemm4w.hip && /tmp/gemm4w [M N K]...
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
#ifndef G4_DMA_EARLY
#define G4_DMA_EARLY 0   // where phase B issues the next-but-one tile's 16 LDS-DMA pieces: 0 = one per 4-MFMA group (round 4), 1 / 2: see the loop
#endif
#ifndef G4_ABL
#define G4_ABL 0   // ablations (wrong results, timings only): 1 = no DMA after the prologue, 2 = no fragment reads after the first
#endif

__device__ __forceinline__ void glds16(const void* gsrc, void* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                   (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}
__device__ __forceinline__ u16 f2bf(float f) { return __builtin_bit_cast(u16, (__bf16)f); }
#define SB() __builtin_amdgcn_sched_barrier(0)

__global__ __launch_bounds__(256) void gemm4w_kernel(const u16* __restrict__ A, const u16* __restrict__ B, u16* __restrict__ C,
                                                     int M, int N, int K, unsigned long long* stamps) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int nkt = K >> 6;
  // tile of this workgroup: XCD-major ids, groups of 8 row tiles (as the product kernel)
  const int tilesM = M >> 8, tilesN = N >> 8, nwg = tilesM * tilesN;
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
  const int m0 = tm << 8, n0 = tn << 8;

  // DMA pieces: 64 per K tile (8 rows x 128 B each); wave w issues pieces w + 4 i: i < 8 -> A rows 8 (w + 4 i) .., i >= 8 -> B rows
  const int srow = lane >> 3, spos = lane & 7;
  // a wave's pieces are 32 rows apart: (row >> 1) & 7 = (4 (wave & 1) + (srow >> 1)) & 7 for all of them, so one lane address per
  // operand and a uniform stride serve the 16 pieces (16 pointers in registers spilled)
  const int chunk = spos ^ ((4 * (wave & 1) + (srow >> 1)) & 7);
  const char* srcA = reinterpret_cast<const char*>(A + (size_t)(m0 + 8 * wave + srow) * K) + chunk * 16;
  const char* srcB = reinterpret_cast<const char*>(B + (size_t)(n0 + 8 * wave + srow) * K) + chunk * 16;
  const size_t piece_stride = (size_t)32 * K * 2;
#define DMA(i, tile)                                                                                               \
  glds16(((i) < 8 ? srcA : srcB) + ((i)&7) * piece_stride + (size_t)(tile)*128,                                    \
         smem + ((tile)&1) * STAGE_BYTES + ((i) < 8 ? 0 : 32768) + (wave + 4 * ((i)&7)) * 1024)

  // fragment addresses: 16-row block at row R0, k half ks: lane (row = lane & 15, k group = lane >> 4) reads chunk (4 ks + group)
  const int frow = lane & 15, fsw = frow >> 1;
  const int fo0 = frow * 128 + (((lane >> 4) ^ fsw) << 4);
  const int fo1 = frow * 128 + (((4 + (lane >> 4)) ^ fsw) << 4);
  const int a_base = wm * 128 * 128, b_base = 32768 + wn * 128 * 128;
#define RD_A(dst, buf, mt, fo) dst[mt] = *reinterpret_cast<const bf16x8*>((buf) + a_base + (mt)*16 * 128 + (fo))
#define RD_B(dst, buf, nt, fo) dst[nt] = *reinterpret_cast<const bf16x8*>((buf) + b_base + (nt)*16 * 128 + (fo))

  floatx4 acc[8][8];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[i][j] = floatx4{0.f, 0.f, 0.f, 0.f};
  bf16x8 a0[8], b0[8], a1[8], b1[8];

  // prologue: tiles 0 and 1 in flight, tile 0's first-half fragments in registers
#pragma unroll
  for (int i = 0; i < 16; ++i) DMA(i, 0);
  if (nkt > 1) {
#pragma unroll
    for (int i = 0; i < 16; ++i) DMA(i, 1);
    asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
  } else {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  SB();
  __builtin_amdgcn_s_barrier();
  SB();
#pragma unroll
  for (int t = 0; t < 8; ++t) {
    RD_A(a0, smem, t, fo0);
    RD_B(b0, smem, t, fo0);
  }
  unsigned long long t_begin = 0, t_end = 0;
  if (stamps) asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_begin)::"memory");

  // 4 MFMAs of row block mt, column blocks 4 h .. 4 h + 3 (acc laid out D'[n][m]: B fragment first, as the product kernel)
// Inline asm with the accumulator tied to an AGPR tuple ("+a"): through the builtin hipcc, with all 256 AGPRs taken by the 64
// accumulators, writes results to spare tuples and copies them back through VGPRs (v_accvgpr_read behind s_nop 4-5 after every
// group: the first version of this file ran 5 040 cycles per K tile). Every accumulator is touched once per phase (64 MFMAs
// apart), so no MFMA reads the result of one still in flight; the epilogue waits out the last ones by hand.
#define MF4(af, bf, mt, h)                                                                                        \
  _Pragma("unroll") for (int nt = 4 * (h); nt < 4 * (h) + 4; ++nt)                                                \
    asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc[mt][nt]) : "v"(bf[nt]), "v"(af[mt]));

  for (int kt = 0; kt < nkt; ++kt) {
    const char* cur = smem + (kt & 1) * STAGE_BYTES;
    const char* nxt = smem + ((kt + 1) & 1) * STAGE_BYTES;
    const bool more = kt + 1 < nkt, more2 = kt + 2 < nkt;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    SB();
    // ---- phase A: k 0..31 of tile kt; the second half's fragments arrive between the MFMAs
#pragma unroll
    for (int c = 0; c < 16; ++c) {
      MF4(a0, b0, c >> 1, c & 1)
      if (!(G4_ABL & 2) || kt == 0) { if (c < 8) { RD_A(a1, cur, c, fo1); } else { RD_B(b1, cur, c - 8, fo1); } }
      SB();
    }
    // everyone's reads of `cur` are issued and waited for; my pieces of tile kt + 1 have landed
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    SB();
    __builtin_amdgcn_s_barrier();
    SB();
    // ---- phase B: k 32..63 of tile kt; tile kt + 2 is requested into `cur`, tile kt + 1's first half is read
#pragma unroll
    for (int c = 0; c < 16; ++c) {
      MF4(a1, b1, c >> 1, c & 1)
#if G4_DMA_EARLY == 1      // the 16 pieces in the first 8 groups (two per group): the last one has 1.5 phases until its reader, not 1
      if (more2 && !(G4_ABL & 1) && c < 8) { DMA(2 * c, kt + 2); DMA(2 * c + 1, kt + 2); }
#elif G4_DMA_EARLY == 2    // one per group in groups 0..7, then the rest two per group: a compromise between spacing and slack
      if (more2 && !(G4_ABL & 1)) { if (c < 8) { DMA(c, kt + 2); } else if (c < 12) { DMA(8 + 2 * (c - 8), kt + 2); DMA(9 + 2 * (c - 8), kt + 2); } }
#else
      if (more2 && !(G4_ABL & 1)) DMA(c, kt + 2);
#endif
      if (more && !(G4_ABL & 2)) {
        if (c < 8) { RD_A(a0, nxt, c, fo0); } else { RD_B(b0, nxt, c - 8, fo0); }
      }
      SB();
    }
  }
  if (stamps) {
    asm volatile("s_nop 7\n\ts_nop 7\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_end)::"memory");
    if (lane == 0) stamps[blockIdx.x * 4 + wave] = t_end - t_begin;
  }
  asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");   // the last MFMAs' results (hipcc does not see the asm MFMAs' latency)
  // plain store: lane holds row (lane & 15), columns 4 (lane >> 4) .. + 3 of each 16 x 16 block
  const int quad = lane >> 4;
#pragma unroll
  for (int mt = 0; mt < 8; ++mt) {
    const int row = m0 + wm * 128 + mt * 16 + (lane & 15);
#pragma unroll
    for (int nt = 0; nt < 8; ++nt) {
      const int col = n0 + wn * 128 + nt * 16 + 4 * quad;
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

int main(int argc, char** argv) {
  std::vector<int> shapes;
  for (int i = 1; i + 2 < argc; i += 3) {
    shapes.push_back(atoi(argv[i]));
    shapes.push_back(atoi(argv[i + 1]));
    shapes.push_back(atoi(argv[i + 2]));
  }
  if (shapes.empty()) shapes = {32768, 4096, 4096, 32768, 12288, 4096, 32768, 22016, 4096, 32768, 4096, 11008};
  hipFuncSetAttribute((const void*)gemm4w_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * STAGE_BYTES);
  for (size_t s = 0; s + 2 < shapes.size(); s += 3) {
    const int M = shapes[s], N = shapes[s + 1], K = shapes[s + 2];
    std::vector<u16> hA((size_t)M * K), hB((size_t)N * K);
    unsigned st = 12345u;
    auto rnd = [&]() { st = st * 1664525u + 1013904223u; return ((st >> 8) & 0xffff) / 65536.0f - 0.5f; };
    for (auto& v : hA) v = f2bf_host(rnd() * 2.0f);
    for (auto& v : hB) v = f2bf_host(rnd() * 0.04f);
    u16 *dA, *dB, *dC;
    unsigned long long* dS;
    hipMalloc(&dA, hA.size() * 2); hipMalloc(&dB, hB.size() * 2); hipMalloc(&dC, (size_t)M * N * 2);
    const int nwg = (M / 256) * (N / 256);
    hipMalloc(&dS, (size_t)nwg * 4 * 8);
    hipMemcpy(dA, hA.data(), hA.size() * 2, hipMemcpyHostToDevice);
    hipMemcpy(dB, hB.data(), hB.size() * 2, hipMemcpyHostToDevice);
    for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(gemm4w_kernel, dim3(nwg), dim3(256), 2 * STAGE_BYTES, 0, dA, dB, dC, M, N, K, (unsigned long long*)nullptr);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    const int reps = 10;
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(gemm4w_kernel, dim3(nwg), dim3(256), 2 * STAGE_BYTES, 0, dA, dB, dC, M, N, K, (unsigned long long*)nullptr);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    ms /= reps;
    hipLaunchKernelGGL(gemm4w_kernel, dim3(nwg), dim3(256), 2 * STAGE_BYTES, 0, dA, dB, dC, M, N, K, dS);
    hipDeviceSynchronize();
    hipError_t err = hipGetLastError();
    std::vector<unsigned long long> hs((size_t)nwg * 4);
    hipMemcpy(hs.data(), dS, hs.size() * 8, hipMemcpyDeviceToHost);
    std::sort(hs.begin(), hs.end());
    std::vector<u16> hC((size_t)M * N);
    hipMemcpy(hC.data(), dC, hC.size() * 2, hipMemcpyDeviceToHost);
    double worst = 0.0;
    for (int t = 0; t < 64; ++t) {
      const int m = (int)(((long long)t * 7919 + 13) % M), n = (int)(((long long)t * 104729 + 7) % N);
      double ref = 0.0;
      for (int k = 0; k < K; ++k) ref += (double)bf2f(hA[(size_t)m * K + k]) * bf2f(hB[(size_t)n * K + k]);
      const double got = bf2f(hC[(size_t)m * N + n]);
      worst = std::max(worst, std::fabs(got - ref) / (std::fabs(ref) + 1e-3));
    }
    printf("M=%d N=%d K=%d: %.1f us = %.0f TF/s; K loop %.0f cycles per K tile (median wave; 2048 = MFMA-bound); max rel err of 64 samples %.4f; %s\n",
           M, N, K, ms * 1e3, 2.0 * M * N * K / (ms * 1e-3) / 1e12, (double)hs[hs.size() / 2] / (K / 64), worst, hipGetErrorString(err));
    hipFree(dA); hipFree(dB); hipFree(dC); hipFree(dS);
  }
  return 0;
}
