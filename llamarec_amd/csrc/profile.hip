// profile.hip -- optional per-kernel timing with HIP events recorded on the caller's stream,
// so bench.py can report the dominant kernel's measured duration and algorithmic work over the
// timed region (roofline.achieved) without an external profiler.
#include <vector>

#include "lr_common.h"
#include "lr_profile.h"

namespace {
struct Rec {
  hipEvent_t a, b;
  int kind;
  double work;
  long long tag;
};
std::vector<Rec> g_pool;
int g_used = 0;
bool g_on = false;
}  // namespace

bool lr_prof_begin(int kind, double work, hipStream_t st, long long tag) {
  if (!g_on || g_used >= (int)g_pool.size()) return false;
  Rec& r = g_pool[g_used];
  r.kind = kind;
  r.work = work;
  r.tag = tag;
  return hipEventRecord(r.a, st) == hipSuccess;
}

void lr_prof_end(hipStream_t st) {
  if (!g_on || g_used >= (int)g_pool.size()) return;
  (void)hipEventRecord(g_pool[g_used].b, st);
  ++g_used;
}

extern "C" int lr_profile_start(int32_t max_records) {
  if (max_records < 1) LR_FAIL(LR_EINVAL, "lr_profile_start: max_records=%d", max_records);
  while ((int)g_pool.size() < max_records) {
    Rec r{};
    LR_CHECK_HIP(hipEventCreate(&r.a));
    LR_CHECK_HIP(hipEventCreate(&r.b));
    g_pool.push_back(r);
  }
  g_used = 0;
  g_on = true;
  return LR_OK;
}

extern "C" int lr_profile_stop(void) {
  g_on = false;
  return LR_OK;
}

extern "C" int lr_profile_collect(int32_t kind, double* total_ms, double* total_work, int64_t* launches) {
  if (!total_ms || !total_work || !launches) LR_FAIL(LR_EINVAL, "lr_profile_collect: null output");
  if (g_on) LR_FAIL(LR_EINVAL, "lr_profile_collect: call lr_profile_stop() and synchronise the stream first");
  double ms = 0, work = 0;
  int64_t n = 0;
  for (int i = 0; i < g_used; ++i) {
    if (g_pool[i].kind != kind) continue;
    float t = 0;
    LR_CHECK_HIP(hipEventElapsedTime(&t, g_pool[i].a, g_pool[i].b));
    ms += t;
    work += g_pool[i].work;
    ++n;
  }
  *total_ms = ms;
  *total_work = work;
  *launches = n;
  return LR_OK;
}

extern "C" int64_t lr_profile_records(int32_t kind, double* ms, double* work, int64_t* tag, int64_t max_records) {
  if (g_on) {
    lr_set_error("lr_profile_records: call lr_profile_stop() and synchronise the stream first");
    return -1;
  }
  int64_t n = 0;
  for (int i = 0; i < g_used; ++i) {
    if (g_pool[i].kind != kind) continue;
    if (n < max_records && ms && work && tag) {
      float t = 0;
      if (hipEventElapsedTime(&t, g_pool[i].a, g_pool[i].b) != hipSuccess) {
        lr_set_error("lr_profile_records: hipEventElapsedTime failed (stream not synchronised?)");
        return -1;
      }
      ms[n] = t;
      work[n] = g_pool[i].work;
      tag[n] = g_pool[i].tag;
    }
    ++n;
  }
  return n;
}
