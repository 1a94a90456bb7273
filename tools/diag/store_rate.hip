// Diagnostic: how fast can a CU push a GEMM tile's bf16 output to memory? One 512-thread workgroup per CU (8 waves, like
// the 256x256 GEMM's epilogue); every lane issues 16 global_store_dwordx4 (128 KB per workgroup) in one of four shapes
// per wave-instruction:  0: 64 lanes x 16 B contiguous (1 KiB)      1: 16 rows x 64 B (the GEMM epilogue today)
//                        2: 8 rows x 128 B (whole cache lines)      3: 4 rows x 256 B
// rows are `ld` bytes apart (the C matrix's row pitch). Prints s_memtime cycles from the first store to all stores issued
// and to vmcnt(0), median over workgroups.   hipcc --offload-arch=gfx950 -O3 -o /tmp/store_rate tools/diag/store_rate.hip
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int SHAPE>
__global__ __launch_bounds__(512) void store_kernel(char* C, size_t ld, unsigned long long* stamps, int do_loads) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wm = wave >> 2, wn = wave & 3;
  // tile of 256 rows x 512 bytes at (blockIdx.x) : tiles laid out along columns then rows
  const size_t tiles_per_row = ld / 512;
  char* tile = C + (blockIdx.x / tiles_per_row) * 256 * ld + (blockIdx.x % tiles_per_row) * 512;
  u32x4 v = u32x4{(unsigned)lane, (unsigned)wave, blockIdx.x, 7u};
  unsigned long long t0, t1, t2;
  __syncthreads();
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
  u32x4 acc = v;
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    size_t off;
    if (SHAPE == 0) {         // wave w, store i: 1 KiB contiguous = 2 rows of the wave's 128-B column strip? no: 2 full 512-B tile rows
      const int piece = (wave * 16 + i);            // 128 pieces of 1 KiB = 2 tile rows each
      off = (size_t)(piece * 2 + (lane >> 5)) * ld + (lane & 31) * 16;
    } else if (SHAPE == 1) {  // today's epilogue: mt = i >> 1, k = i & 1: row = wm*128 + mt*16 + (lane & 15), 64 B at wn*128 + k*64
      const int mt = i >> 1, k = i & 1, li = lane & 15, quad = lane >> 4;
      off = (size_t)(wm * 128 + mt * 16 + li) * ld + wn * 128 + k * 64 + (quad & 1) * 32 + (quad >> 1) * 16;
    } else if (SHAPE == 2) {  // 8 rows x 128 B: row = wm*128 + i*8 + (lane >> 3), the wave's 128-B strip
      off = (size_t)(wm * 128 + i * 8 + (lane >> 3)) * ld + wn * 128 + (lane & 7) * 16;
    } else {                  // 4 rows x 256 B: two waves' strips: row = wm*128 + (i*2 + (wn&1))*4 + (lane >> 4), 256 B at (wn>>1)*256
      off = (size_t)(wm * 128 + (i * 2 + (wn & 1)) * 4 + (lane >> 4)) * ld + (wn >> 1) * 256 + (lane & 15) * 16;
    }
    if (do_loads) {
      const u32x4 r = *reinterpret_cast<const u32x4*>(tile + off);
      acc += r;
    } else {
      *reinterpret_cast<u32x4*>(tile + off) = v;
    }
  }
  if (do_loads) asm volatile("" ::"v"(acc));
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t2)::"memory");
  if (do_loads && acc[0] == 0x12345u) tile[0] = 1;
  if (lane == 0) {
    stamps[(blockIdx.x * 8 + wave) * 2 + 0] = t1 - t0;
    stamps[(blockIdx.x * 8 + wave) * 2 + 1] = t2 - t0;
  }
}

template <int SHAPE>
void run(char* C, size_t ld, int nwg, unsigned long long* d_st, int do_loads, const char* name) {
  std::vector<unsigned long long> h(nwg * 16);
  for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL(store_kernel<SHAPE>, dim3(nwg), dim3(512), 0, 0, C, ld, d_st, do_loads);
  hipDeviceSynchronize();
  hipMemcpy(h.data(), d_st, h.size() * 8, hipMemcpyDeviceToHost);
  std::vector<unsigned long long> issue, done;
  for (int w = 0; w < nwg; ++w) {
    unsigned long long mi = 0, md = 0;
    for (int k = 0; k < 8; ++k) { mi = std::max(mi, h[(w * 8 + k) * 2]); md = std::max(md, h[(w * 8 + k) * 2 + 1]); }
    issue.push_back(mi); done.push_back(md);
  }
  std::sort(issue.begin(), issue.end()); std::sort(done.begin(), done.end());
  printf("  %-28s %s ld=%6zu nwg=%4d: last wave issued %6llu cyc, completed %6llu cyc (median over workgroups) = %.1f B/clk/CU\n", name,
         do_loads ? "LOAD " : "STORE", ld, nwg, issue[nwg / 2], done[nwg / 2], 131072.0 / done[nwg / 2]);
}

int main() {
  const size_t bytes = (size_t)32768 * 24576;   // a 32768 x 12288 bf16 matrix
  char* C; unsigned long long* st;
  hipMalloc(&C, bytes); hipMemset(C, 0, bytes);
  hipMalloc(&st, 8192 * 16 * 8);
  for (int nwg : {64, 256, 2048}) {
    for (size_t ld : {(size_t)8192, (size_t)24576}) {
      for (int loads = 0; loads < 2; ++loads) {
        run<0>(C, ld, nwg, st, loads, "1 KiB contiguous (2 rows)");
        run<1>(C, ld, nwg, st, loads, "16 rows x 64 B (today)");
        run<2>(C, ld, nwg, st, loads, "8 rows x 128 B");
        run<3>(C, ld, nwg, st, loads, "4 rows x 256 B");
      }
    }
  }
  return 0;
}
