// Probe of v_mfma_scale_f32_16x16x128_f8f6f4 operand semantics (fp8 e4m3 x fp8 e4m3, E8M0 block scales).
// out[l] = D fragment of lane l for A/B/scale registers loaded verbatim from memory (8 dwords per lane each).
#include <hip/hip_runtime.h>
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));
extern "C" __global__ void mx_probe(const v8i* a, const v8i* b, const int* sa, const int* sb, f4* out, int fmt) {
  const int l = threadIdx.x;
  f4 acc = {0.f, 0.f, 0.f, 0.f};
  if (fmt == 0) acc = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a[l], b[l], acc, 0, 0, 0, sa[l], 0, sb[l]);
  else if (fmt == 1) acc = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a[l], b[l], acc, 4, 4, 0, sa[l], 0, sb[l]);
  else if (fmt == 2) acc = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a[l], b[l], acc, 0, 4, 0, sa[l], 0, sb[l]);   // A fp8, B fp4
  else acc = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a[l], b[l], acc, 0, 4, 0, sa[l], 2, sb[l]);                 // B scale from byte 2
  out[l] = acc;
}
extern "C" int run_probe(const void* a, const void* b, const void* sa, const void* sb, void* out, int fmt, void* stream) {
  hipLaunchKernelGGL(mx_probe, dim3(1), dim3(64), 0, (hipStream_t)stream, (const v8i*)a, (const v8i*)b, (const int*)sa, (const int*)sb, (f4*)out, fmt);
  return (int)hipGetLastError();
}

// v_cvt_scalef32_pk_fp8_f16 semantics: out[i] = {lo word <- in[i], hi word <- in[i+n]} with the given scale
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
typedef short s2 __attribute__((ext_vector_type(2)));
extern "C" __global__ void cvt_probe(const h2* in, s2* out, float sc, int n) {
  int i = threadIdx.x;
  if (i >= n) return;
  s2 r = {0, 0};
  r = __builtin_amdgcn_cvt_scalef32_pk_fp8_f16(r, in[i], sc, false);
  r = __builtin_amdgcn_cvt_scalef32_pk_fp8_f16(r, in[i + n], sc, true);
  out[i] = r;
}
extern "C" int run_cvt(const void* in, void* out, float sc, int n, void* stream) {
  hipLaunchKernelGGL(cvt_probe, dim3(1), dim3(64), 0, (hipStream_t)stream, (const h2*)in, (s2*)out, sc, n);
  return (int)hipGetLastError();
}
