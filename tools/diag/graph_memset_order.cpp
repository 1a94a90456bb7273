// Diagnostic for DESIGN.md section 4b (round 1: "hipMemsetAsync / hipMemcpyAsync nodes were replayed out of order with
// neighbouring kernels"). Captures  fill(a) -> memset(a) -> add(a) -> memcpy D2D (b <- a) -> scale(b) -> memcpy H2D (c <- host)
// -> add_from(b, c)  on one stream, prints the captured graph's nodes and dependency edges, instantiates it and replays it
// with the HOST source of the H2D copy changed between replays. Build + run on the GPU box:
//   hipcc --offload-arch=gfx950 -O2 tools/diag/graph_memset_order.cpp -o /tmp/gmo && /tmp/gmo
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
__global__ void fill(float* a, int n, float v) { int i = blockIdx.x * 256 + threadIdx.x; if (i < n) a[i] = v; }
__global__ void add(float* a, int n, float v) { int i = blockIdx.x * 256 + threadIdx.x; if (i < n) a[i] += v; }
__global__ void scale(float* a, int n, float v) { int i = blockIdx.x * 256 + threadIdx.x; if (i < n) a[i] *= v; }
__global__ void add_from(float* a, const float* c, int n) { int i = blockIdx.x * 256 + threadIdx.x; if (i < n) a[i] += c[0]; }
int main() {
  const int n = 1 << 20;
  float *a, *b, *c;
  CK(hipMalloc(&a, n * 4)); CK(hipMalloc(&b, n * 4)); CK(hipMalloc(&c, 4));
  hipStream_t st; CK(hipStreamCreate(&st));
  float host_val = 100.f;                      // the H2D source: a host variable whose VALUE changes between replays
  CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
  hipLaunchKernelGGL(fill, dim3(n / 256), dim3(256), 0, st, a, n, 7.f);
  CK(hipMemsetAsync(a, 0, n * 4, st));                                   // a = 0
  hipLaunchKernelGGL(add, dim3(n / 256), dim3(256), 0, st, a, n, 5.f);   // a = 5
  CK(hipMemcpyAsync(b, a, n * 4, hipMemcpyDeviceToDevice, st));          // b = 5
  hipLaunchKernelGGL(scale, dim3(n / 256), dim3(256), 0, st, b, n, 2.f); // b = 10
  CK(hipMemcpyAsync(c, &host_val, 4, hipMemcpyHostToDevice, st));        // c = host_val (pageable host memory)
  hipLaunchKernelGGL(add_from, dim3(n / 256), dim3(256), 0, st, b, c, n); // b = 10 + host_val
  hipGraph_t g; CK(hipStreamEndCapture(st, &g));
  size_t nn = 0; CK(hipGraphGetNodes(g, nullptr, &nn));
  std::vector<hipGraphNode_t> nodes(nn); CK(hipGraphGetNodes(g, nodes.data(), &nn));
  const char* tn[] = {"kernel", "memcpy", "memset", "host", "graph", "empty", "waitEvent", "eventRecord", "extSemSignal", "extSemWait", "memAlloc", "memFree"};
  for (size_t i = 0; i < nn; ++i) { hipGraphNodeType t; CK(hipGraphNodeGetType(nodes[i], &t)); printf("node %zu: %s\n", i, (int)t < 12 ? tn[(int)t] : "?"); }
  size_t ne = 0; CK(hipGraphGetEdges(g, nullptr, nullptr, &ne));
  std::vector<hipGraphNode_t> from(ne), to(ne); CK(hipGraphGetEdges(g, from.data(), to.data(), &ne));
  auto idx = [&](hipGraphNode_t x) { for (size_t i = 0; i < nn; ++i) if (nodes[i] == x) return (int)i; return -1; };
  for (size_t i = 0; i < ne; ++i) printf("edge %d -> %d\n", idx(from[i]), idx(to[i]));
  hipGraphExec_t ex; CK(hipGraphInstantiate(&ex, g, nullptr, nullptr, 0));
  std::vector<float> hb(n);
  int bad_total = 0;
  for (int rep = 0; rep < 4; ++rep) {
    host_val = 100.f * (rep + 1);              // eager semantics would give b = 10 + 100 (rep + 1)
    CK(hipGraphLaunch(ex, st)); CK(hipStreamSynchronize(st));
    CK(hipMemcpy(hb.data(), b, n * 4, hipMemcpyDeviceToHost));
    int bad = 0; for (int i = 0; i < n; ++i) bad += hb[i] != 10.f + host_val;
    printf("replay %d: b[0] = %g (eager semantics: %g), %d mismatching elements\n", rep, hb[0], 10.f + host_val, bad);
    bad_total += bad;
  }
  printf(bad_total ? "RESULT: replay differs from eager order/values\n" : "RESULT: replay == eager semantics\n");
  return 0;
}
