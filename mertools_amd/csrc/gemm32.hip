// gemm32.hip — exact-fp32 GEMM on v_mfma_f32_16x16x4_f32 (bitwise an fmaf chain per output, 1/16 of
// the f16 MFMA rate).  Used where the reference computes tiny fp32 GEMMs whose results feed
// label-level decisions: the fusion classifier MERBench/toolkit/models/attention.py:36-57 /
// modules/encoder.py:30-41 (forward and backward).  One wave per 16x16 output tile; operands are
// read straight from global memory (these problems are a few hundred KB and live in L2).
//   C[m,n] (+)= act(sum_k A[m,k] * W[n,k] + bias[n])
// trans_a: A stored [K,M]; trans_w: W stored [K,N] (needed for dX = dY W and dW = dY^T X).
#include "common.h"

namespace mer {

template <bool TA, bool TW>
__global__ __launch_bounds__(64) void gemm32_kernel(const float* __restrict__ a, long long lda, const float* __restrict__ w,
                                                    long long ldw, const float* __restrict__ bias, int act, float* c,
                                                    long long ldc, int accumulate, int M, int N, int K) {
  const int lane = threadIdx.x, li = lane & 15, lg = lane >> 4;
  const int m = blockIdx.y * 16 + li;   // A row this lane feeds
  const int n = blockIdx.x * 16 + li;   // W row this lane feeds
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  // contiguous-k operands are fetched as one 16-byte load per lane and k-step when alignment allows
  const bool va = !TA && (lda % 4 == 0) && ((((uintptr_t)a) & 15) == 0) && (K % 16 == 0) && m < M;
  const bool vw = !TW && (ldw % 4 == 0) && ((((uintptr_t)w) & 15) == 0) && (K % 16 == 0) && n < N;
  for (int k0 = 0; k0 < K; k0 += 16) {
    float av[4], wv[4];
    const int kb = k0 + lg * 4;
    if (va) {
      const f32x4 t = *reinterpret_cast<const f32x4*>(a + (long long)m * lda + kb);
      av[0] = t[0]; av[1] = t[1]; av[2] = t[2]; av[3] = t[3];
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int k = kb + j;
        av[j] = (k < K && m < M) ? (TA ? a[(long long)k * lda + m] : a[(long long)m * lda + k]) : 0.f;
      }
    }
    if (vw) {
      const f32x4 t = *reinterpret_cast<const f32x4*>(w + (long long)n * ldw + kb);
      wv[0] = t[0]; wv[1] = t[1]; wv[2] = t[2]; wv[3] = t[3];
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int k = kb + j;
        wv[j] = (k < K && n < N) ? (TW ? w[(long long)k * ldw + n] : w[(long long)n * ldw + k]) : 0.f;
      }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[j], wv[j], acc, 0, 0, 0);
  }
  const int col = blockIdx.x * 16 + li;
  if (col >= N) return;
  const float bv = bias ? bias[col] : 0.f;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int row = blockIdx.y * 16 + lg * 4 + r;
    if (row < M) {
      float v = act_apply(acc[r] + bv, act);
      float* dst = c + (long long)row * ldc + col;
      *dst = accumulate ? *dst + v : v;
    }
  }
}

}  // namespace mer

extern "C" int mer_gemm32(const float* a, long long lda, int trans_a, const float* w, long long ldw, int trans_w,
                          const float* bias, int act, float* c, long long ldc, int accumulate, int M, int N, int K,
                          mer_stream_t stream) {
  using namespace mer;
  MER_REQUIRE(a && w && c, MER_EINVAL, "mer_gemm32: null pointer");
  MER_REQUIRE(M > 0 && N > 0 && K > 0, MER_ESHAPE, "mer_gemm32: bad shape M=%d N=%d K=%d", M, N, K);
  dim3 grid((unsigned)cdiv(N, 16), (unsigned)cdiv(M, 16)), block(64);
  hipStream_t st = (hipStream_t)stream;
  if (trans_a && trans_w) gemm32_kernel<true, true><<<grid, block, 0, st>>>(a, lda, w, ldw, bias, act, c, ldc, accumulate, M, N, K);
  else if (trans_a) gemm32_kernel<true, false><<<grid, block, 0, st>>>(a, lda, w, ldw, bias, act, c, ldc, accumulate, M, N, K);
  else if (trans_w) gemm32_kernel<false, true><<<grid, block, 0, st>>>(a, lda, w, ldw, bias, act, c, ldc, accumulate, M, N, K);
  else gemm32_kernel<false, false><<<grid, block, 0, st>>>(a, lda, w, ldw, bias, act, c, ldc, accumulate, M, N, K);
  return check_launch("gemm32");
}
