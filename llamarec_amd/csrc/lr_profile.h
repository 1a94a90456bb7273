// lr_profile.h -- internal hooks of the optional event profiler (profile.hip).
#ifndef LR_PROFILE_H
#define LR_PROFILE_H
#include <hip/hip_runtime.h>

// kinds (mirrored in include/llamarec_mi355x.h)
#define LR_PROF_GEMM256 0
#define LR_PROF_GEMM_GENERIC 1
#define LR_PROF_ATTN_MFMA 2
#define LR_PROF_ATTN_GENERIC 3
#define LR_PROF_LRU_ENCODE 4
#define LR_PROF_ITEM_TOPK 5
#define LR_PROF_ELEMENTWISE 6

// tag: free-form 64-bit label kept with the record (GEMMs: LR_PROF_GEMM_TAG(epi, N, K)) so a caller can split a
// kind by shape (lr_profile_records)
#define LR_PROF_GEMM_TAG(epi, N, K) (((long long)(epi) << 56) | ((long long)(N) << 28) | (long long)(K))
bool lr_prof_begin(int kind, double work, hipStream_t st, long long tag = 0);
void lr_prof_end(hipStream_t st);

struct LrProfScope {
  hipStream_t st;
  bool on;
  LrProfScope(int kind, double work, hipStream_t s, long long tag = 0) : st(s), on(lr_prof_begin(kind, work, s, tag)) {}
  ~LrProfScope() {
    if (on) lr_prof_end(st);
  }
};
#endif
