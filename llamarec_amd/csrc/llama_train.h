// llama_train.h -- internal launcher declarations of the ranker's LoRA training step (llama_train.hip,
// llama_attn_bwd.hip, api_llama_train.hip).
#ifndef LLAMA_TRAIN_H
#define LLAMA_TRAIN_H

#include "llama_kernels.h"

#define LT_RP 16  // LoRA rank padded to one MFMA column tile per projection (r <= 16)

int lr_launch_transpose_bf16(const unsigned short* src, int rows, int cols, unsigned short* dst, hipStream_t st);
int lr_launch_prep_lora(const float* aq, const float* bq, const float* av, const float* bv, int r, int d, int qcols,
                        int vcols, int hd, unsigned short* a_cat, unsigned short* bq_t, unsigned short* bv_t,
                        hipStream_t st);
int lr_launch_skinny(const unsigned short* X, int ldx, int n, int K, const unsigned short* W, int nt, unsigned short* out,
                     int ldo, int ocol, float scale, uint32_t drop_stream, float drop_p, hipStream_t st);
int lr_launch_lora_rope_fwd(unsigned short* qkv, int n, int qw, int qcols, int kcols, int hd, const unsigned short* t,
                            const unsigned short* bq_t, const unsigned short* bv_t, int r, float scaling,
                            const int32_t* tok_pos, const float* rope_cs, hipStream_t st);
int lr_launch_rope_bwd(unsigned short* dqkv, int n, int qw, int rot_cols, int hd, const int32_t* tok_pos,
                       const float* rope_cs, hipStream_t st);
int lr_launch_lora_db(const unsigned short* dqkv, int n, int qw, int qcols, int kcols, int hd, const unsigned short* t,
                      int r, float scaling, float* dbq, float* dbv, hipStream_t st);
int lr_launch_lora_da(const unsigned short* xn, int n, int d, const unsigned short* dt, int r, uint32_t drop_stream,
                      float drop_p, float* daq, float* dav, hipStream_t st);
int lr_launch_swiglu_fwd(const unsigned short* gu, unsigned short* h, int n, int f, hipStream_t st);
int lr_launch_swiglu_bwd(unsigned short* gu, const unsigned short* dh, int n, int f, hipStream_t st);
int lr_launch_rmsnorm_bwd(const unsigned short* dy, const unsigned short* x, const unsigned short* w,
                          const unsigned short* res, unsigned short* out, int rows, int d, float eps,
                          const int32_t* out_rows, const unsigned short* dt, const unsigned short* a_cat, int r,
                          uint32_t drop_stream, float drop_p, hipStream_t st);
int lr_launch_ce_bf16(unsigned short* logits, int m, int V, const int32_t* targets, float gscale, float* scal,
                      hipStream_t st);
int lr_launch_finish_loss(const float* scal, int m, float* out, hipStream_t st);
int lr_launch_rowdot(const unsigned short* o, const unsigned short* d_o, int n, int nh, int hd, float* out,
                     hipStream_t st);
int lr_launch_lora_adamw(float* p, float* g, float* m, float* v, size_t n, float* scratch, int* ctr, float lr,
                         float max_grad_norm, float beta1, float beta2, float eps, float wd, float* out_norm,
                         hipStream_t st);
uint32_t lr_lora_drop_stream(uint64_t seed, uint32_t pass, uint32_t layer);

// attention with the softmax statistics kept (lse[n][nh], natural log) and its backward (llama_attn_bwd.hip)
int lr_launch_attention_lse(const unsigned short* qkv, unsigned short* out, float* lse, const int32_t* cu,
                            const int32_t* cu_host, int B, int n_tok, int nh, int nkv, int hd, int variant,
                            hipStream_t st);
// dqkv [n][(nh+2nkv)*hd], fully written: the gradient w.r.t. the rotated q, k and v -- or, when rope_cs / tok_pos are
// given, w.r.t. the UNROTATED ones (the inverse rotation is pair-local in the packed layout and rides in the MFMA
// passes' epilogues). dsum: [n][nh] fp32 scratch, dkv32: [n][2*nkv*hd] fp32 scratch (generic path only).
int lr_launch_attention_bwd(const unsigned short* qkv, const unsigned short* out, const unsigned short* d_out,
                            const float* lse, unsigned short* dqkv, float* dsum, float* dkv32, const int32_t* cu,
                            const int32_t* cu_host, int B, int n_tok, int nh, int nkv, int hd, int variant,
                            hipStream_t st, const int32_t* tok_pos = nullptr, const float* rope_cs = nullptr);

#endif
