// llama_nf4.hip -- load-time NF4 round trip of the frozen ranker weights (SURVEY.md 8(f) #1 "dequantise at load").
//
// The reference never runs the bf16 checkpoint: every Linear of the base model is quantised at load to 4-bit NF4 with
// double quantisation (train_ranker.py:49-56, setup_demo.py:67-74: BitsAndBytesConfig(load_in_4bit, nf4,
// bnb_4bit_use_double_quant, compute dtype bf16)) and dequantised inside every forward by bitsandbytes 0.43.1
// (environment.yml:327, CUDA-only, absent here). This file restates the published algorithm (QLoRA, Dettmers et al.
// 2023, sections 3 "4-bit NormalFloat" and "Double Quantization"; bitsandbytes functional.quantize_4bit /
// dequantize_4bit) as ONE offline transform  W -> dequant(quant(W))  in bf16, so that the scoring and training paths
// multiply by the weights the reference's forward effectively uses:
//   1. blocks of 64 consecutive elements (row-major): absmax_b = max |w|; code index = nearest of the 16 NF4 levels to
//      w * (1 / absmax_b) (the decision tree of the published kernel = midpoint thresholds);
//   2. double quantisation of the absmax vector: offset = mean(absmax); blocks of 256 of (absmax - offset) quantised to
//      the 8-bit "dynamic" code (256 levels, bitsandbytes create_dynamic_map(signed)) with their own absmax;
//   3. w' = bf16( NF4[index] * (code8[q_b] * absmax2 + offset) ).
// Parity with bitsandbytes itself is UNPINNED (it cannot run here); oracle/nf4_oracle.py is the numpy restatement this
// kernel is bit-compared with. The mean uses exact fixed-point accumulation (order-independent), not fp32 summation.
#include <math.h>
#include <stdlib.h>

#include "llama_kernels.h"

typedef unsigned short u16;

__constant__ float c_nf4[16] = {-1.0f, -0.6961928009986877f, -0.5250730514526367f, -0.39491748809814453f,
                                -0.28444138169288635f, -0.18477343022823334f, -0.09105003625154495f, 0.0f,
                                0.07958029955625534f, 0.16093020141124725f, 0.24611230194568634f, 0.33791524171829224f,
                                0.44070982933044434f, 0.5626170039176941f, 0.7229568362236023f, 1.0f};
// midpoints between neighbouring levels (the thresholds of the published decision tree)
__constant__ float c_nf4_mid[15] = {-0.8480964004993439f, -0.6106329262256622f, -0.4599952697753906f,
                                    -0.33967943489551544f, -0.23460740596055984f, -0.13791173323988914f,
                                    -0.045525018125772476f, 0.03979014977812767f, 0.1202552504837513f,
                                    0.2035212516784668f, 0.2920137718319893f, 0.3893125355243683f,
                                    0.5016634166240692f, 0.6427869200706482f, 0.8614784181118011f};

#define NF4_BLOCK 64
#define NF4_NESTED 256
#define NF4_FIXED_SHIFT 40  // absmax < 2^23 assumed; 2^-40 resolution

// pass 1: absmax per 64-element block; exact fixed-point sum of all of them
__global__ __launch_bounds__(256) void nf4_absmax_kernel(const u16* w, size_t n, float* absmax, size_t nb,
                                                         unsigned long long* sum_fixed) {
  const size_t b = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (b >= nb) return;
  const size_t i = b * NF4_BLOCK + lane;
  float v = i < n ? fabsf(bf2f(w[i])) : 0.f;
#pragma unroll
  for (int s = 32; s >= 1; s >>= 1) v = fmaxf(v, __shfl_xor(v, s, 64));
  if (lane == 0) {
    absmax[b] = v;
    atomicAdd(sum_fixed, (unsigned long long)((double)v * (double)(1ull << NF4_FIXED_SHIFT)));
  }
}

// pass 2 (double quantisation): absmax_b <- code8[nearest((absmax_b - offset) / absmax2)] * absmax2 + offset
__global__ __launch_bounds__(256) void nf4_nested_kernel(const float* absmax, float* absmax_q, size_t nb,
                                                         const unsigned long long* sum_fixed, const float* code8) {
  __shared__ float sh[4];
  __shared__ float s_code[256];
  s_code[threadIdx.x] = code8[threadIdx.x];
  const float offset = (float)((double)sum_fixed[0] / (double)(1ull << NF4_FIXED_SHIFT) / (double)nb);
  const size_t i = (size_t)blockIdx.x * NF4_NESTED + threadIdx.x;
  const float a = i < nb ? absmax[i] - offset : 0.f;
  float m = fabsf(a);
#pragma unroll
  for (int s = 32; s >= 1; s >>= 1) m = fmaxf(m, __shfl_xor(m, s, 64));
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = m;
  __syncthreads();
  const float absmax2 = fmaxf(fmaxf(sh[0], sh[1]), fmaxf(sh[2], sh[3]));
  if (i >= nb) return;
  float q = 0.f;
  if (absmax2 > 0.f) {
    const float v = a * (1.0f / absmax2);
    int lo = 0, hi = 255;  // nearest level of the sorted code: first index whose midpoint to the next level is >= v
    while (lo < hi) {
      const int mid = (lo + hi) >> 1;
      if (v > 0.5f * (s_code[mid] + s_code[mid + 1])) lo = mid + 1; else hi = mid;
    }
    q = s_code[lo] * absmax2;
  }
  absmax_q[i] = q + offset;
}

// pass 3: index with the exact absmax, value with the (possibly double-quantised) one
__global__ __launch_bounds__(256) void nf4_apply_kernel(const u16* w, u16* out, size_t n, const float* absmax,
                                                        const float* absmax_q) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    const size_t b = i / NF4_BLOCK;
    const float am = absmax[b];
    if (am == 0.f) {  // an all-zero block stays zero
      out[i] = 0;
      continue;
    }
    const float v = bf2f(w[i]) * (1.0f / am);
    int idx = 0;
#pragma unroll
    for (int k = 0; k < 15; ++k) idx += v > c_nf4_mid[k] ? 1 : 0;
    out[i] = f2bf(c_nf4[idx] * absmax_q[b]);
  }
}

// bitsandbytes.functional.create_dynamic_map(signed=True, max_exponent_bits=7, total_bits=8): 127 positive and 127
// negative levels (decades 1e-6 .. 1 with 1, 2, 4, .. 64 linear steps each), 0 and 1, sorted
static void nf4_dynamic_map(float* out) {
  int n = 0;
  for (int i = 0; i < 7; ++i) {
    const int items = (1 << i) + 1;
    const float scale = (float)pow(10.0, -6 + i);
    // torch.linspace(0.1, 1, items) in fp32: fma(step, k, start) for the first half, fma(-step, items - 1 - k, end) after
    const float start = 0.1f, end = 1.0f, step = (end - start) / (float)(items - 1);
    auto lin = [&](int k) { return k < items / 2 ? fmaf(step, (float)k, start) : fmaf(-step, (float)(items - 1 - k), end); };
    for (int k = 0; k + 1 < items; ++k) {
      const float b0 = lin(k), b1 = lin(k + 1);
      const float mean = (b0 + b1) / 2.0f;
      out[n++] = scale * mean;
      out[n++] = -scale * mean;
    }
  }
  out[n++] = 0.f;
  out[n++] = 1.0f;
  for (int i = 1; i < n; ++i) {  // insertion sort, 256 entries
    const float v = out[i];
    int j = i - 1;
    while (j >= 0 && out[j] > v) {
      out[j + 1] = out[j];
      --j;
    }
    out[j + 1] = v;
  }
}

extern "C" int lr_nf4_dynamic_map(float* out256) {
  if (!out256) LR_FAIL(LR_EINVAL, "lr_nf4_dynamic_map: null pointer");
  nf4_dynamic_map(out256);
  return LR_OK;
}

extern "C" size_t lr_nf4_scratch_bytes(size_t n) {
  const size_t nb = (n + NF4_BLOCK - 1) / NF4_BLOCK;
  return lr_align_up(nb * 4, 256) * 2 + 256 * 4 + 256;
}

extern "C" int lr_nf4_roundtrip_bf16(const uint16_t* w, size_t n, int32_t double_quant, uint16_t* out, void* scratch,
                                     size_t scratch_bytes, void* hip_stream) {
  if (!w || !out || !scratch || n == 0) LR_FAIL(LR_EINVAL, "lr_nf4_roundtrip_bf16: bad argument");
  if (scratch_bytes < lr_nf4_scratch_bytes(n)) LR_FAIL(LR_EWORKSPACE, "lr_nf4_roundtrip_bf16: scratch too small");
  hipStream_t st = (hipStream_t)hip_stream;
  const size_t nb = (n + NF4_BLOCK - 1) / NF4_BLOCK;
  char* s = (char*)scratch;
  float* absmax = (float*)s;
  float* absmax_q = (float*)(s + lr_align_up(nb * 4, 256));
  float* code8 = (float*)(s + 2 * lr_align_up(nb * 4, 256));
  unsigned long long* sum_fixed = (unsigned long long*)(code8 + 256);
  LR_CHECK_HIP(hipMemsetAsync(sum_fixed, 0, 8, st));
  hipLaunchKernelGGL(nf4_absmax_kernel, dim3((unsigned)((nb + 3) / 4)), dim3(256), 0, st, w, n, absmax, nb, sum_fixed);
  LR_CHECK_LAUNCH("nf4_absmax_kernel");
  if (double_quant) {
    float host_code[256];
    nf4_dynamic_map(host_code);
    LR_CHECK_HIP(hipMemcpyAsync(code8, host_code, sizeof(host_code), hipMemcpyHostToDevice, st));
    LR_CHECK_HIP(hipStreamSynchronize(st));  // host_code is on this frame (load-time call, not on the hot path)
    hipLaunchKernelGGL(nf4_nested_kernel, dim3((unsigned)((nb + NF4_NESTED - 1) / NF4_NESTED)), dim3(256), 0, st, absmax,
                       absmax_q, nb, sum_fixed, code8);
    LR_CHECK_LAUNCH("nf4_nested_kernel");
  }
  const size_t blocks = (n + 255) / 256;
  hipLaunchKernelGGL(nf4_apply_kernel, dim3((unsigned)(blocks < 65536 ? blocks : 65536)), dim3(256), 0, st, w, out, n,
                     absmax, double_quant ? absmax_q : absmax);
  LR_CHECK_LAUNCH("nf4_apply_kernel");
  return LR_OK;
}
