// Deterministic accumulation for the retriever's training pass (lr_lru_train_set_deterministic; trainer/lru.py:20-28 and
// trainer/base.py:106-112 are what the pass replaces). The pass combines per-workgroup partial sums -- the loss, every
// parameter gradient that is a sum over rows, and d x behind the item GEMM -- with fp32 atomics, whose order changes from run to
// run. In deterministic mode every such add goes, as a 64-bit FIXED-POINT number, into a shadow of its target instead:
// integer addition commutes, so the shadow's final value does not depend on the order, and it is turned back into the float
// buffer at fixed points of the launch sequence (lr_det_fold_kernel: behind the cross-entropy for d x and the loss, in front
// of tr_unprep_kernel for the derived weights' gradients, by the pass's last launch for the gradient buffer).
// An addend v becomes round(v * 2^k) exactly (a power-of-two scaling of a float, then an integer that a 64-bit register
// holds): k = 48 for gradients (range +-32768, resolution 3.6e-15), k = 32 for the loss sum / the squared gradient norm. An addend
// that is not finite, or outside that range, takes the plain fp32 atomic instead: a NaN stays a NaN in the buffer.
// One map per translation unit in constant memory (LR_DET_DEFINE), set by the host in front of a pass: a process runs ONE
// deterministic engine at a time.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define LR_DET_REGIONS 4
struct LrDetMap {
  const float* base[LR_DET_REGIONS];        // float regions whose atomic adds are redirected ...
  unsigned long long bytes[LR_DET_REGIONS];
  long long* shadow[LR_DET_REGIONS];        // ... to these fixed-point shadows (same element index)
  float scale[LR_DET_REGIONS];              // 2^k
  int on;
};

#define LR_DET_DEFINE(tag)                                                                                   \
  static __constant__ LrDetMap g_lr_det;                                                                      \
  int lr_det_set_##tag(const LrDetMap* m) {                                                                   \
    return hipMemcpyToSymbol(HIP_SYMBOL(g_lr_det), m, sizeof(LrDetMap)) == hipSuccess ? 0 : 1;                \
  }                                                                                                           \
  __device__ __forceinline__ void lr_det_add(float* p, float v) {                                             \
    if (g_lr_det.on) {                                                                                        \
      _Pragma("unroll") for (int k = 0; k < LR_DET_REGIONS; ++k) {                                           \
        const unsigned long long off = (unsigned long long)((uintptr_t)p - (uintptr_t)g_lr_det.base[k]);      \
        if (off < g_lr_det.bytes[k]) {                                                                        \
          const float sv = v * g_lr_det.scale[k];                                                             \
          if (!(fabsf(sv) < 9.0e18f)) break;   /* NaN, inf or out of the fixed-point range: the float add keeps it visible */ \
          atomicAdd(reinterpret_cast<unsigned long long*>(g_lr_det.shadow[k] + (off >> 2)),                   \
                    (unsigned long long)__float2ll_rn(sv));                                                   \
          return;                                                                                             \
        }                                                                                                     \
      }                                                                                                       \
    }                                                                                                         \
    atomicAdd(p, v);                                                                                          \
  }

int lr_det_set_train(const LrDetMap* m);    // lru_train.hip
int lr_det_set_blocks(const LrDetMap* m);   // lru_train_blocks.hip
int lr_det_set_ce(const LrDetMap* m);       // lru_train_ce.hip
int lr_det_set_scores(const LrDetMap* m);   // lru_train_scores.hip
