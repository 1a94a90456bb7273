// Diagnostic: what does a wave pay per v_mfma_f32_16x16x4_f32 in the shapes the training kernels use?
//   0  64 MFMAs per iteration on 4 accumulators, operands in registers                     (pure issue rate)
//   1  + the b operand made by 16 x (v_fma + v_exp_f32) per iteration in front of its MFMAs (VALU -> MFMA)
//   2  + the a operands loaded from an L2-resident buffer one set ahead                     (the d x kernel's loop)
//   3  variant 2 with the b operand from registers (loads only)
//   4  v_mfma_f32_32x32x2_f32, 16 per iteration on 4 accumulators, operands in registers
// hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_f32_issue tools/diag/mfma_f32_issue.hip && /tmp/mfma_f32_issue
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>
typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef float floatx16 __attribute__((ext_vector_type(16)));
__device__ __forceinline__ float el(const float4& v, int e) { return e == 0 ? v.x : e == 1 ? v.y : e == 2 ? v.z : v.w; }

template <int V>
__global__ __launch_bounds__(512) void k(const float* buf, float* out, unsigned long long* stamps, int iters, float lse, int lab, int C) {
  const int lane = threadIdx.x & 63;
  unsigned long long t0, t1;
  floatx4 acc[4];
  for (int i = 0; i < 4; ++i) acc[i] = floatx4{0.f, 0.f, 0.f, 0.f};
  floatx16 big[4];
  for (int i = 0; i < 4; ++i) for (int e = 0; e < 16; ++e) big[i][e] = 0.f;
  float4 w[2][16], l4[4];
  for (int i = 0; i < 16; ++i) w[0][i] = w[1][i] = make_float4(lane * 0.01f, 1.f, 2.f, i);
  for (int j = 0; j < 4; ++j) l4[j] = make_float4(0.1f * lane, 0.2f, 0.3f, j);
  const float* p = buf + lane * 4;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
  if (V == 4) {
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int i = 0; i < 4; ++i) big[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(el(w[0][s], i), el(l4[s], i), big[i], 0, 0, 0);
    }
  } else {
    if (V >= 2) {
#pragma unroll
      for (int i = 0; i < 16; ++i) w[0][i] = *reinterpret_cast<const float4*>(p + i * 256);
    }
    for (int it = 0; it < iters; it += 2) {
#pragma unroll
      for (int half = 0; half < 2; ++half) {
        if (V >= 2) {
#pragma unroll
          for (int i = 0; i < 16; ++i) w[half ^ 1][i] = *reinterpret_cast<const float4*>(p + ((it + half + 1) & 63) * 4096 + i * 256);
          __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          float4 dl = l4[j];
          if (V == 1 || V == 2 || V >= 5) {
            dl.x = __builtin_amdgcn_exp2f(__builtin_fmaf(l4[j].x, 1.44f, -lse));
            dl.y = __builtin_amdgcn_exp2f(__builtin_fmaf(l4[j].y, 1.44f, -lse));
            dl.z = __builtin_amdgcn_exp2f(__builtin_fmaf(l4[j].z, 1.44f, -lse));
            dl.w = __builtin_amdgcn_exp2f(__builtin_fmaf(l4[j].w, 1.44f, -lse));
            l4[j].x += 1e-3f;   // keep the chain alive across iterations
          }
          if (V == 5) {   // the label / padding branches of ts_dl4 (rarely taken, but they rewrite EXEC)
            const unsigned kk = (unsigned)(lab - (it * 64 + 16 * j + (lane >> 4) * 4));
            if (kk < 4u) { dl.x -= kk == 0 ? lse : 0.f; dl.y -= kk == 1 ? lse : 0.f; dl.z -= kk == 2 ? lse : 0.f; dl.w -= kk == 3 ? lse : 0.f; }
            if (it * 64 + 16 * j + 4 > C) { dl.x = 0.f; dl.y = 0.f; }
          }
          if (V == 6) {   // branch-free label
            const unsigned kk = (unsigned)(lab - (it * 64 + 16 * j + (lane >> 4) * 4));
            dl.x -= kk == 0 ? lse : 0.f; dl.y -= kk == 1 ? lse : 0.f; dl.z -= kk == 2 ? lse : 0.f; dl.w -= kk == 3 ? lse : 0.f;
          }
#pragma unroll
          for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int nb = 0; nb < 4; ++nb)
              acc[nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(el(w[half][nb * 4 + j], e), el(dl, e), acc[nb], 0, 0, 0);
        }
      }
    }
  }
  asm volatile("s_nop 7\n\ts_nop 7\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
  float s = 0.f;
  for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3] + big[i][0] + big[i][7];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (lane == 0) stamps[blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] = t1 - t0;
}


// Sweep: NL coalesced 1 KB loads (one register set ahead, unconditional) per 64 MFMAs and wave -> what a CU's vector memory path
// delivers under an MFMA stream. Addresses walk a 1 MB window (L2-resident) or stay inside 16 KB (L1-resident).
template <int NL, bool L1>
__global__ __launch_bounds__(512) void ksweep(const float* buf, float* out, unsigned long long* stamps, int iters) {
  const int lane = threadIdx.x & 63;
  unsigned long long t0, t1;
  floatx4 acc[4];
  for (int i = 0; i < 4; ++i) acc[i] = floatx4{0.f, 0.f, 0.f, 0.f};
  float4 w[2][NL];
  const float* p = buf + lane * 4;
  const int span = L1 ? 1 : 16;   // windows of NL KB
  for (int i = 0; i < NL; ++i) w[0][i] = *reinterpret_cast<const float4*>(p + i * 256);
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
  for (int it = 0; it < iters; it += 2) {
#pragma unroll
    for (int half = 0; half < 2; ++half) {
#pragma unroll
      for (int i = 0; i < NL; ++i) w[half ^ 1][i] = *reinterpret_cast<const float4*>(p + ((it + half + 1) % span) * (NL * 256) + i * 256);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int k = 0; k < 64; ++k)
        acc[k & 3] = __builtin_amdgcn_mfma_f32_16x16x4f32(el(w[half][(k >> 2) % NL], k & 3), 1.0f + lane, acc[k & 3], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  asm volatile("s_nop 7\n\ts_nop 7\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
  float s = 0.f;
  for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (lane == 0) stamps[blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] = t1 - t0;
}
template <int NL, bool L1>
void sweep(int threads, const float* buf, float* out, unsigned long long* st) {
  const int iters = 256, grid = 256;
  hipLaunchKernelGGL((ksweep<NL, L1>), dim3(grid), dim3(threads), 0, 0, buf, out, st, iters);
  hipLaunchKernelGGL((ksweep<NL, L1>), dim3(grid), dim3(threads), 0, 0, buf, out, st, iters);
  hipDeviceSynchronize();
  const int n = grid * threads / 64;
  std::vector<unsigned long long> h(n);
  hipMemcpy(h.data(), st, n * 8, hipMemcpyDeviceToHost);
  std::sort(h.begin(), h.end());
  const double cyc = (double)h[n / 2] / (iters * 64);   // per MFMA and wave
  const double bytes_per_clk = (double)(threads / 64) * NL * 1024 / (cyc * 64);   // per CU
  printf("%2d x 1 KB loads per 64 MFMAs and wave, %s, %d waves/SIMD: %6.1f cycles per MFMA and wave -> %5.1f B/clk per CU\n", NL,
         L1 ? "L1-resident" : "L2-resident", threads / 256, cyc, bytes_per_clk);
}

template <int V>
void run(const char* name, int threads, const float* buf, float* out, unsigned long long* st, int mfma_per_iter) {
  const int iters = 256, grid = 256;
  hipLaunchKernelGGL(k<V>, dim3(grid), dim3(threads), 0, 0, buf, out, st, iters, 3.0f, 777 + (int)(size_t)buf % 3, 1 << 30);
  hipLaunchKernelGGL(k<V>, dim3(grid), dim3(threads), 0, 0, buf, out, st, iters, 3.0f, 777 + (int)(size_t)buf % 3, 1 << 30);
  hipDeviceSynchronize();
  const int n = grid * threads / 64;
  std::vector<unsigned long long> h(n);
  hipMemcpy(h.data(), st, n * 8, hipMemcpyDeviceToHost);
  std::sort(h.begin(), h.end());
  printf("%-58s %d waves/SIMD: median %7.1f cycles per MFMA (p95 %7.1f)\n", name, threads / 256, (double)h[n / 2] / (iters * mfma_per_iter),
         (double)h[n * 95 / 100] / (iters * mfma_per_iter));
}

int main() {
  float *buf, *out;
  unsigned long long* st;
  hipMalloc(&buf, 4 << 20);
  hipMemset(buf, 0, 4 << 20);
  hipMalloc(&out, 256 * 512 * 4);
  hipMalloc(&st, 256 * 8 * 8);
  for (int threads : {256, 512}) {
    run<0>("0 16x16x4, registers", threads, buf, out, st, 64);
    run<1>("1 + 16 (fma, exp2) per 64 MFMAs", threads, buf, out, st, 64);
    run<2>("2 + a operands loaded one set ahead (L2)", threads, buf, out, st, 64);
    run<3>("3 loads only (b from registers)", threads, buf, out, st, 64);
    run<4>("4 32x32x2, registers (16 per iteration)", threads, buf, out, st, 16);
    run<5>("5 variant 2 + label / padding branches (EXEC rewrites)", threads, buf, out, st, 64);
    run<6>("6 variant 2 + branch-free label select", threads, buf, out, st, 64);
  }
  for (int threads : {256, 512}) {
    sweep<4, false>(threads, buf, out, st);
    sweep<8, false>(threads, buf, out, st);
    sweep<16, false>(threads, buf, out, st);
    sweep<32, false>(threads, buf, out, st);
    sweep<16, true>(threads, buf, out, st);
    sweep<32, true>(threads, buf, out, st);
  }
  return 0;
}
