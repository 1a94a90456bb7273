// lr_math.h -- exact-arithmetic building blocks shared by the HIP kernels and the C oracle.
//
// Every function here is a fixed sequence of IEEE-754 binary32 operations (add, mul, fma,
// div, sqrt -- all correctly rounded on x86-64 and on gfx950 with the default
// -fhip-fp32-correctly-rounded-divide-sqrt) with NO libm transcendental calls, so the same
// source compiled by gcc (oracle/) and by hipcc (device code) returns bit-identical floats.
// Both sides must be compiled with -ffp-contract=off: every fused multiply-add is written
// explicitly as lr_fma().
//
// Reference formulas restated here (citations into /root/reference):
//   LayerNorm(64)          model/lru.py:51,60,133,161,171,175  (nn.LayerNorm, eps 1e-5)
//   GELU (erf form)        model/lru.py:169,174                 (nn.GELU default)
//   complex diag. recurrence  model/lru.py:135-161 (sequential form)
#ifndef LR_MATH_H
#define LR_MATH_H

#include <stdint.h>
#include <string.h>
#include "lr_erf_table.h"

#if defined(__HIPCC__)
#define LR_HD __host__ __device__ __forceinline__
#else
#define LR_HD static inline
#endif

#define LR_D 64        /* bert_hidden_units (config.py:212) */
#define LR_H 128       /* complex state width = 2*D (model/lru.py:109) */
#define LR_FF 256      /* d_ff = 4*D (model/lru.py:96) */
#define LR_LN_EPS 1e-5f
#define LR_MASK_SCORE (-1e9f) /* trainer/lru.py:37-38 */

LR_HD float lr_fma(float a, float b, float c) { return __builtin_fmaf(a, b, c); }

LR_HD uint32_t lr_f2u(float f) {
  uint32_t u;
#if defined(__HIP_DEVICE_COMPILE__)
  u = __builtin_bit_cast(uint32_t, f);
#else
  memcpy(&u, &f, 4);
#endif
  return u;
}

// Piecewise degree-8 polynomial erf, |abs err| < 1e-7 over R (table: tools/gen_erf_table.py).
LR_HD float lr_erff_tab(float x, const float* tab) {
  float a = __builtin_fabsf(x);
  float r;
  if (!(a < 4.0f)) {
    r = 1.0f;  // erf(4) rounds to 1 in binary32; NaN also lands here (propagated below)
    if (a != a) r = a;
  } else {
    int i = (int)(a * 2.0f);
    float t = a - ((float)i * 0.5f + 0.25f);
    const float* c = tab + i * (LR_ERF_DEG + 1);
    r = c[LR_ERF_DEG];
    for (int j = LR_ERF_DEG - 1; j >= 0; --j) r = lr_fma(r, t, c[j]);
  }
  return __builtin_copysignf(r, x);
}

// GELU(x) = 0.5*x*(1+erf(x/sqrt(2)))
LR_HD float lr_gelu_tab(float x, const float* tab) {
  float e = lr_erff_tab(x * 0.70710678118654752440f, tab);
  return (0.5f * x) * (1.0f + e);
}

// Monotone map float -> uint32 (larger float => larger uint), -0 < +0.
LR_HD uint32_t lr_float_ord(float f) {
  uint32_t u = lr_f2u(f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
LR_HD float lr_ord_float(uint32_t o) {
  uint32_t u = (o & 0x80000000u) ? (o & 0x7fffffffu) : ~o;
  float f;
#if defined(__HIP_DEVICE_COMPILE__)
  f = __builtin_bit_cast(float, u);
#else
  memcpy(&f, &u, 4);
#endif
  return f;
}
// Ranking key: larger key = better candidate. Score descending, then item id ascending
// (the one tie rule used everywhere; torch.topk / argsort leave it unspecified,
// trainer/lru.py:82-84,113-115).
LR_HD uint64_t lr_rank_key(float score, uint32_t item) {
  return ((uint64_t)lr_float_ord(score) << 32) | (uint64_t)(0xffffffffu - item);
}
LR_HD uint32_t lr_key_item(uint64_t key) { return 0xffffffffu - (uint32_t)(key & 0xffffffffu); }
LR_HD float lr_key_score(uint64_t key) { return lr_ord_float((uint32_t)(key >> 32)); }

// Item-score dot product order (matches one accumulation chain of v_mfma_f32_32x32x2_f32
// when lane-half 0 feeds k = s and lane-half 1 feeds k = 32 + s for MFMA step s):
//   acc = 0; for s in 0..31: acc = fma(e[32+s], q[32+s], fma(e[s], q[s], acc));
//   score = acc + bias
LR_HD float lr_item_score(const float* e, const float* q, float bias) {
  float acc = 0.0f;
  for (int s = 0; s < 32; ++s) {
    acc = lr_fma(e[s], q[s], acc);
    acc = lr_fma(e[32 + s], q[32 + s], acc);
  }
  return acc + bias;
}

#include <math.h>
// lambda = exp(-exp(nu_log) + i*exp(theta_log)), gamma = exp(gamma_log)  (model/lru.py:151-152).
// Host-only (double libm, rounded once to binary32); called by BOTH the oracle and the
// library's weight packer so the recurrence coefficients are the same 32-bit values.
static inline void lr_lru_derive(const float* params_log /*[3][128]*/, float* lam_re, float* lam_im,
                                 float* gamma) {
  for (int c = 0; c < LR_H; ++c) {
    double nu = exp((double)params_log[c]);
    double th = exp((double)params_log[LR_H + c]);
    double mag = exp(-nu);
    lam_re[c] = (float)(mag * cos(th));
    lam_im[c] = (float)(mag * sin(th));
    gamma[c] = (float)exp((double)params_log[2 * LR_H + c]);
  }
}

#endif  // LR_MATH_H
