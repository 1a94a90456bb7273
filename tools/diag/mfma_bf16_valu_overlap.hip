// Diagnostic: do v_mfma_f32_16x16x32_bf16 and vector-ALU work of the OTHER wave of a SIMD overlap? (The attention key block is
// 64 MFMAs + ~190 vector instructions, 34 of them v_exp_f32, per wave; two waves per SIMD.) One workgroup per CU:
//   0  4 waves, MFMA only: 64 MFMAs per iteration on 8 accumulators
//   1  4 waves, VALU only: 156 v_fma_f32 + 34 v_exp_f32 per iteration (independent chains)
//   2  4 waves, both in sequence in every wave (one wave per SIMD: MFMA time + VALU time is the floor)
//   3  8 waves, waves 0-3 MFMA only, waves 4-7 VALU only  (two waves per SIMD, one of each kind: overlap -> max, none -> sum)
//   4  8 waves, both in sequence in every wave            (the attention kernel's situation)
//   5  8 waves, MFMA only in all                          (two MFMA streams on one pipe: 2 x variant 0)
//   6  8 waves, VALU only in all
// hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_bf16_valu_overlap tools/diag/mfma_bf16_valu_overlap.hip && /tmp/mfma_bf16_valu_overlap
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float floatx4 __attribute__((ext_vector_type(4)));

template <bool MFMA, bool VALU>
__device__ __forceinline__ void body(floatx4 (&acc)[8], bf16x8 a, bf16x8 b, float (&v)[8]) {
  if (MFMA) {
#pragma unroll
    for (int i = 0; i < 64; ++i) acc[i & 7] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[i & 7], 0, 0, 0);
  }
  __builtin_amdgcn_sched_barrier(0);
  if (VALU) {
#pragma unroll
    for (int i = 0; i < 156; ++i) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(v[i & 7]) : "v"(v[(i + 1) & 7]));
#pragma unroll
    for (int i = 0; i < 34; ++i) asm volatile("v_exp_f32 %0, %0" : "+v"(v[i & 7]));
  }
  __builtin_amdgcn_sched_barrier(0);
}

template <int V>
__global__ __launch_bounds__(512) void k(float* out, unsigned long long* cyc, int iters) {
  const int wave = threadIdx.x >> 6;
  floatx4 acc[8];
  float v[8];
  for (int i = 0; i < 8; ++i) {
    acc[i] = floatx4{0.f, 0.f, 0.f, 0.f};
    v[i] = 0.001f * (threadIdx.x + i);
  }
  bf16x8 a, b;
  for (int e = 0; e < 8; ++e) {
    a[e] = (__bf16)(0.01f * (threadIdx.x & 7));
    b[e] = (__bf16)(0.02f * e);
  }
  unsigned long long t0, t1;
  __syncthreads();
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
  for (int it = 0; it < iters; ++it) {
    if (V == 0 || V == 5) body<true, false>(acc, a, b, v);
    if (V == 1 || V == 6) body<false, true>(acc, a, b, v);
    if (V == 2 || V == 4) body<true, true>(acc, a, b, v);
    if (V == 3) {
      if (wave < 4) body<true, false>(acc, a, b, v);
      else body<false, true>(acc, a, b, v);
    }
  }
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
  float s = 0.f;
  for (int i = 0; i < 8; ++i) s += acc[i][0] + v[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 8 + wave] = t1 - t0;
}

template <int V>
static void run(const char* name, int threads, float* out, unsigned long long* cyc) {
  const int iters = 2000, grid = 256;
  hipMemset(cyc, 0, grid * 8 * sizeof(unsigned long long));
  hipLaunchKernelGGL(k<V>, dim3(grid), dim3(threads), 0, 0, out, cyc, iters);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipEventRecord(e0, 0);
  hipLaunchKernelGGL(k<V>, dim3(grid), dim3(threads), 0, 0, out, cyc, iters);
  hipEventRecord(e1, 0);
  hipDeviceSynchronize();
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  std::vector<unsigned long long> h(grid * 8);
  hipMemcpy(h.data(), cyc, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost);
  std::vector<unsigned long long> lo, hi;
  for (int g = 0; g < grid; ++g)
    for (int w = 0; w < threads / 64; ++w) (w < 4 ? lo : hi).push_back(h[g * 8 + w]);
  std::sort(lo.begin(), lo.end());
  std::sort(hi.begin(), hi.end());
  printf("%-78s %8.1f us/iter-block  s_memtime per iteration: waves 0-3 %7.0f", name, ms * 1e3 / iters * 1.0, (double)lo[lo.size() / 2] / iters);
  if (!hi.empty()) printf("  waves 4-7 %7.0f", (double)hi[hi.size() / 2] / iters);
  printf("  (%s)\n", hipGetErrorString(hipGetLastError()));
}

int main() {
  float* out;
  unsigned long long* cyc;
  hipMalloc(&out, 256 * 512 * sizeof(float));
  hipMalloc(&cyc, 256 * 8 * sizeof(unsigned long long));
  run<0>("0 one wave per SIMD, 64 MFMA (16x16x32 bf16)", 256, out, cyc);
  run<1>("1 one wave per SIMD, 156 v_fma + 34 v_exp", 256, out, cyc);
  run<2>("2 one wave per SIMD, both in sequence", 256, out, cyc);
  run<3>("3 two waves per SIMD, one MFMA-only, one VALU-only", 512, out, cyc);
  run<4>("4 two waves per SIMD, both in sequence in each", 512, out, cyc);
  run<5>("5 two waves per SIMD, MFMA only", 512, out, cyc);
  run<6>("6 two waves per SIMD, VALU only", 512, out, cyc);
  return 0;
}
