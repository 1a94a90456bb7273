// TEMPORARY stubs (replaced by the real stage-2 implementation).
#include "lr_common.h"
extern "C" int lr_llama_create(const LrLlamaConfig*, const LrLlamaWeightsDesc*, lr_llama_t**) { LR_FAIL(LR_EUNSUPPORTED, "stub"); }
extern "C" void lr_llama_destroy(lr_llama_t*) {}
extern "C" size_t lr_llama_workspace_bytes(const lr_llama_t*, int32_t, int32_t) { return 0; }
extern "C" int lr_llama_prefill_verbalize(lr_llama_t*, const int32_t*, const int32_t*, const int32_t*, int32_t, const int32_t*, int32_t, float*, void*, size_t, void*) { LR_FAIL(LR_EUNSUPPORTED, "stub"); }
extern "C" int lr_llama_last_logits(lr_llama_t*, const int32_t*, const int32_t*, const int32_t*, int32_t, float*, void*, size_t, void*) { LR_FAIL(LR_EUNSUPPORTED, "stub"); }
extern "C" int lr_llama_pack_gate_up(const uint16_t*, const uint16_t*, int32_t, int32_t, uint16_t*) { LR_FAIL(LR_EUNSUPPORTED, "stub"); }
extern "C" int lr_gemm_bf16_nt(const uint16_t*, const uint16_t*, uint16_t*, int32_t, int32_t, int32_t, int32_t, void*) { LR_FAIL(LR_EUNSUPPORTED, "stub"); }
extern "C" int lr_attention_varlen(const uint16_t*, uint16_t*, const int32_t*, const int32_t*, int32_t, int32_t, int32_t, int32_t, int32_t, void*) { LR_FAIL(LR_EUNSUPPORTED, "stub"); }
