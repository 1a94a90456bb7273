// What the matrix pipe sustains when NOTHING else runs, by MFMA shape and by what the operands hold (diagnostic, not product):
// every wave keeps its A / B fragments in registers and issues back-to-back MFMAs into independent accumulators for ~1 s.
// The product GEMM is clock-limited on model-like data (profiles/r05_gemm_operand_data.txt): this says how much of that is
// the MFMAs themselves, and whether 32x32x16 or 16x16x32 costs less energy per flop.
// build: hipcc --offload-arch=gfx950 -O2 -o tools/diag/mfma_power tools/diag/mfma_power.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef unsigned short u16x8 __attribute__((ext_vector_type(8)));

__device__ inline unsigned short rnd_bf16(uint32_t& s, int kind) {
  s = s * 1664525u + 1013904223u;
  if (kind == 0) return 0;                       // zeros
  // kind 1: roughly normal: sum of 4 uniforms, scaled; bf16 by truncation
  float u = 0.f;
  uint32_t t = s;
  for (int i = 0; i < 4; ++i) { t = t * 1664525u + 1013904223u; u += (float)(t >> 8) * (1.0f / 16777216.0f) - 0.5f; }
  s = t;
  return (unsigned short)(__builtin_bit_cast(uint32_t, u * 1.7f) >> 16);
}

// ORDER (16x16x32 only): which operand changes between consecutive MFMAs. 0: A every MFMA, B every fourth (a register-blocked tile walked
// along its rows); 1: both every MFMA; 2: both only every fourth MFMA (the same pair issued four times into different accumulators)
template <int SHAPE, int ORDER = 0>   // 0: 16x16x32, 1: 32x32x16
__global__ __launch_bounds__(256) void spin(float* sink, int iters, int kind) {
  uint32_t s = threadIdx.x * 747796405u + blockIdx.x * 2891336453u + 1u;
  u16x8 a_[4], b_[4];
  for (int i = 0; i < 4; ++i)
    for (int e = 0; e < 8; ++e) { a_[i][e] = rnd_bf16(s, kind); b_[i][e] = rnd_bf16(s, kind); }
  if constexpr (SHAPE == 0) {
    floatx4 acc[16];
    for (int i = 0; i < 16; ++i) acc[i] = floatx4{0, 0, 0, 0};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < 16; ++i)
        acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a_[ORDER == 2 ? (i >> 2) : (i & 3)]),
                                                         __builtin_bit_cast(bf16x8, b_[ORDER == 1 ? ((i + (i >> 2)) & 3) : (i >> 2)]), acc[i], 0, 0, 0);
    }
    float t = 0;
    for (int i = 0; i < 16; ++i) t += acc[i][0] + acc[i][3];
    if (t == 12345.678f) sink[0] = t;
  } else {
    floatx16 acc[8];
    for (int i = 0; i < 8; ++i)
      for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < 8; ++i)
        acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a_[i & 3]), __builtin_bit_cast(bf16x8, b_[i >> 1 & 3]), acc[i], 0, 0, 0);
    }
    float t = 0;
    for (int i = 0; i < 8; ++i) t += acc[i][0] + acc[i][7];
    if (t == 12345.678f) sink[0] = t;
  }
}

template <int SHAPE, int ORDER = 0>
void run(const char* name, int waves_per_simd, int kind, float* sink) {
  const int iters = 400000;
  const double flops_per_iter_wave = SHAPE == 0 ? 16.0 * 2 * 16 * 16 * 32 : 8.0 * 2 * 32 * 32 * 16;
  dim3 grid(256 * waves_per_simd), block(256);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int rep = 0; rep < 3; ++rep) {   // the third repetition is the reported one: the clock has settled
    hipEventRecord(e0);
    for (int k = 0; k < 12; ++k) hipLaunchKernelGGL((spin<SHAPE, ORDER>), grid, block, 0, 0, sink, iters, kind);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    if (rep == 2) {
      const double tf = 12.0 * flops_per_iter_wave * iters * grid.x * 4 / (ms * 1e-3) / 1e12;
      printf("%s, %d wave(s) per SIMD, operands %s: %.0f ms, %.0f TF/s (dense peak 2500)\n", name, waves_per_simd, kind ? "normal" : "zeros ", ms, tf);
      fflush(stdout);
    }
  }
}

int main() {
  float* sink; hipMalloc(&sink, 16);
  for (int kind = 0; kind < 2; ++kind)
    for (int w = 1; w <= 2; ++w) {
      run<0>("v_mfma_f32_16x16x32_bf16", w, kind, sink);
      run<1>("v_mfma_f32_32x32x16_bf16", w, kind, sink);
    }
  // operand order, 16x16x32, two waves per SIMD, normal operands
  run<0, 0>("16x16x32, A changes every MFMA, B every fourth", 2, 1, sink);
  run<0, 1>("16x16x32, A and B change every MFMA        ", 2, 1, sink);
  run<0, 2>("16x16x32, A and B change every fourth MFMA  ", 2, 1, sink);
  run<0, 0>("16x16x32, A changes every MFMA, B every fourth", 2, 1, sink);
  return 0;
}
