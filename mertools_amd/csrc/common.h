// Shared device/host helpers for the MI355X (gfx950) kernels of mertools_amd.
// Everything here is written for wave64 / CDNA4 only.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include "../../include/mer_hip.h"

namespace mer {

typedef _Float16 f16;
typedef __bf16 bf16;
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef short i16x2 __attribute__((ext_vector_type(2)));
typedef int i32x2 __attribute__((ext_vector_type(2)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef long long i64x4 __attribute__((ext_vector_type(4)));

// 16-bit element traits: conversion + the 16x16x32 MFMA for that type.
template <typename T> struct T16;
template <> struct T16<f16> {
  typedef f16x8 v8; typedef f16x4 v4;
  static __device__ __forceinline__ f16 from_f32(float x) { return (f16)x; }
  static __device__ __forceinline__ float to_f32(f16 x) { return (float)x; }
  static __device__ __forceinline__ f32x4 mfma(v8 a, v8 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
  }
};
template <> struct T16<bf16> {
  typedef bf16x8 v8; typedef bf16x4 v4;
  static __device__ __forceinline__ bf16 from_f32(float x) { return (bf16)x; }
  static __device__ __forceinline__ float to_f32(bf16 x) { return (float)x; }
  static __device__ __forceinline__ f32x4 mfma(v8 a, v8 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
  }
};

// hi/lo split of an fp32 value into two 16-bit planes (hi = rn(x), lo = rn(x - hi)).
// x must be ONE fp32 value for both lines: handed `a * b`, hipcc converts the product to 16 bits twice — v_cvt_pk_f16_f32 of the rounded
// fp32 product for the hi it stores, v_fma_mixlo_f16 (the exact product, rounded once) for the hi it subtracts — and when the fp32 product
// sits on a 16-bit rounding tie the two disagree: lo gets the wrong sign, one element in ~10^4 is off by a full 16-bit ulp (found through
// mer_attention_f32's test; the f16 attention kernels' lo planes had it too).  The empty asm pins the value.
template <typename T>
__device__ __forceinline__ void split16(float x, T& hi, T& lo) {
  asm volatile("" : "+v"(x));
  hi = T16<T>::from_f32(x);
  lo = T16<T>::from_f32(x - T16<T>::to_f32(hi));
}

// erf via Abramowitz-Stegun 7.1.26 (|abs err| <= 1.5e-7): 5 FMAs + rcp + exp. The libm erff
// costs ~3x more VALU issue slots, which is visible next to the MFMA loop in GEMM epilogues.
__device__ __forceinline__ float fast_erf(float x) {
  const float ax = fabsf(x);
  const float t = __builtin_amdgcn_rcpf(1.0f + 0.3275911f * ax);
  float p = 1.061405429f;
  p = p * t - 1.453152027f;
  p = p * t + 1.421413741f;
  p = p * t - 0.284496736f;
  p = p * t + 0.254829592f;
  const float r = 1.0f - p * t * __expf(-ax * ax);
  return copysignf(r, x);
}

// GELU(x) = x Phi(x) = max(x, 0) - |x| T(|x|) with T(a) = erfc(a / sqrt 2) / 2 = exp2(Q(a)): Q is a degree-6 polynomial fitted
// on [0, 5.5] under the weight a T(a) (the error as it appears in the GELU value); past 5.5 the polynomial ARGUMENT is clamped, the
// factor |x| in front is not, so the tail term is |x| T(5.5) = 1.9e-8 |x| instead of ~0.  |result - exact| <= 5e-7 absolute in fp32
// arithmetic for |x| <= 12 (4.8e-7 at x = 4.1, 2.3e-7 at x = -12), growing as 1.9e-8 |x| beyond (1.9e-6 at |x| = 100: below half an
// ulp of any 16-bit output of that magnitude's neighbourhood, and 2e-8 relative to a pre-activation that large).
// Cost: min + 6 FMA + v_exp_f32 + max + FMA, against ~15 VALU + v_rcp_f32 + v_exp_f32 for the erf route — the activation is the
// largest single VALU cost of the fc1 / conv GEMM epilogues.
// Do not "tidy" the last line into fmaf(-a, ...): tried in round 3 — the inlined copies of the epilogues then disagree in the last fp32
// bit (36 of 776 k f16 outputs differ between the packed and the generic epilogue), which also breaks the bit-equality of a row's
// result under a different position in its tile (padding invariance of the batch: tests/test_encoders_gpu.py::test_hubert_ragged_batch).
__device__ __forceinline__ float gelu_exp2poly(float x) {
  const float a = fminf(fabsf(x), 5.5f);
  float q = 3.589585917e-05f;
  q = fmaf(q, a, -7.945234977e-04f);
  q = fmaf(q, a, 8.167289912e-03f);
  q = fmaf(q, a, -5.355345435e-02f);
  q = fmaf(q, a, -4.586574375e-01f);
  q = fmaf(q, a, -1.151242835e+00f);
  q = fmaf(q, a, -9.999880846e-01f);
  return fmaf(-fabsf(x), __builtin_amdgcn_exp2f(q), fmaxf(x, 0.f));
}

__device__ __forceinline__ float act_apply(float x, int act) {
  switch (act) {
    case MER_ACT_GELU: return gelu_exp2poly(x);
    case MER_ACT_QUICK_GELU: return x * __builtin_amdgcn_rcpf(1.0f + __expf(-1.702f * x));
    case MER_ACT_RELU: return x > 0.f ? x : 0.f;
    case MER_ACT_GELU_TANH: {   // tanh(u) = 1 - 2 / (exp(2u) + 1); exp overflow -> inf -> tanh = 1, underflow -> -1
      const float u = 0.7978845608028654f * (x + 0.044715f * x * x * x);
      return 0.5f * x * (2.0f - 2.0f / (__expf(2.0f * u) + 1.0f));
    }
    default: return x;
  }
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
  return v;
}

// ---- host-side error plumbing (thread-local message, int status across the C ABI) ----
void set_error(const char* fmt, ...);
int check_launch(const char* what);

#define MER_REQUIRE(cond, code, ...)        \
  do {                                      \
    if (!(cond)) {                          \
      ::mer::set_error(__VA_ARGS__);        \
      return (code);                        \
    }                                       \
  } while (0)

// Optional per-launch timing (bench.py's roofline leg): when enabled, every instrumented launch is
// bracketed by hipEvents on the launch stream; mer_prof_report() resolves them after a sync.
struct ProfScope {
  void* rec;
  hipStream_t st;
  ProfScope(const char* name, double flops, double bytes, hipStream_t stream);
  ~ProfScope();
};

static inline long long cdiv(long long a, long long b) { return (a + b - 1) / b; }

}  // namespace mer
