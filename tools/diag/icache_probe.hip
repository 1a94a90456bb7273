// Instruction-cache capacity probe for gfx950 (diagnostic, not part of the product): a loop whose body is N KiB of straight-line
// 8-byte VALU instructions, run by one wave per CU on every CU (the two CUs that share an instruction cache run the same code);
// cycles per iteration rise where the body stops fitting. Measured (round 5): within 1.1 % of the issue floor up to a 64 KiB body, +6 %
// at 80 and 96 KiB -- the cache holds 64 KiB, and straight-line misses that hit L2 are cheap.
// build: hipcc --offload-arch=gfx950 -O2 -o icache_probe tools/diag/icache_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int KIB>
__global__ void probe(unsigned long long* out, int iters, const float4* stream, size_t stream_n, float4* sink) {
  unsigned long long t0, t1;
  float v = threadIdx.x;
  // the streaming side: 1 KiB per iteration and wave (a trickle: it does not turn L2 over)
  const char* p = (const char*)(stream ? stream : (const float4*)out) + (stream ? ((size_t)blockIdx.x * iters * 1024 + threadIdx.x * 16) : 0);
  const unsigned step = stream ? 1024u : 0u;
  float4 x;
  for (int pass = 0; pass < 2; ++pass) {   // one warm pass, then the timed one
    int i = iters;
    asm volatile(
        "s_memtime %0\n\ts_waitcnt lgkmcnt(0)\n"
        "1:\n\t.rept %7\n\tv_add_f32 %2, 0x3f800001, %2\n\t.endr\n\t"
        "global_load_dwordx4 %4, %3, off\n\t"
        "v_lshl_add_u64 %3, %6, 0, %3\n\t"
        "s_waitcnt vmcnt(0)\n\t"
        "s_sub_u32 %5, %5, 1\n\ts_cmp_lg_u32 %5, 0\n\ts_cbranch_scc1 1b\n\t"
        "s_memtime %1\n\ts_waitcnt lgkmcnt(0)"
        : "=&s"(t0), "=&s"(t1), "+v"(v), "+v"(p), "=&v"(x), "+s"(i)
        : "s"((unsigned long long)step), "n"(KIB * 128)
        : "memory", "scc");
  }
  if (threadIdx.x == 0) out[blockIdx.x] = (t1 - t0) / iters;
  if (v == 12345.f || x.x == 12345.f) sink[0] = x;
}

template <int KIB>
void run(unsigned long long* d_out, const float4* stream, size_t n, float4* sink) {
  const int iters = 64;
  hipLaunchKernelGGL(probe<KIB>, dim3(256), dim3(64), 0, 0, d_out, iters, stream, n, sink);
  hipDeviceSynchronize();
  std::vector<unsigned long long> h(256);
  hipMemcpy(h.data(), d_out, 256 * 8, hipMemcpyDeviceToHost);
  unsigned long long mn = ~0ull, mx = 0, sum = 0;
  for (auto x : h) { mn = x < mn ? x : mn; mx = x > mx ? x : mx; sum += x; }
  printf("body %3d KiB%s: cycles/iteration min %llu mean %llu max %llu  (issue floor %d)\n", KIB, stream ? " +stream" : "        ", mn, sum / 256, mx,
         KIB * 128 * 4);
  fflush(stdout);
}

int main() {
  unsigned long long* d_out;
  hipMalloc(&d_out, 256 * 8);
  float4 *stream, *sink;
  const size_t n = (size_t)1 << 26;   // 1 GiB
  hipMalloc(&stream, n * 16);
  hipMemset(stream, 0, n * 16);
  hipMalloc(&sink, 16);
  for (int s = 0; s < 2; ++s) {
    const float4* st = s ? stream : nullptr;
    run<8>(d_out, st, n, sink);
    run<16>(d_out, st, n, sink);
    run<24>(d_out, st, n, sink);
    run<32>(d_out, st, n, sink);
    run<40>(d_out, st, n, sink);
    run<48>(d_out, st, n, sink);
    run<56>(d_out, st, n, sink);
    run<64>(d_out, st, n, sink);
    run<80>(d_out, st, n, sink);
    run<96>(d_out, st, n, sink);
  }
  return 0;
}
