// abi_selftest.cpp — standalone check of libmer_hip.so through its C ABI (no Python, no torch: a GPU-box call costs ~15 s).
// Covers the code paths written after round 1's GPU budget was spent:
//   1. mer_image_resize_crop_u8 against bytes produced by Pillow on the host (selftest_vectors.h: frames, expected crop,
//      the host-built window / coefficient tables; bicubic and bilinear);
//   2. mer_seq_bias (the per-sequence weight-residual correction table of the "mean" preset) against the same sums on the host
// Prints one JSON line per check; exit code = number of failed checks.
// Build (scripts/probes/build_probes.sh): hipcc abi_selftest.cpp -I../../include -L../../mertools_amd -lmer_hip -Wl,-rpath,...
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <vector>
#include "mer_hip.h"
#include "selftest_vectors.h"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(100); } } while (0)
#define MER(x) do { int rc_ = (x); if (rc_ != 0) { fprintf(stderr, "%s:%d rc=%d %s\n", __FILE__, __LINE__, rc_, mer_last_error()); exit(101); } } while (0)

template <typename T>
static T* to_dev(const T* h, size_t n) {
  T* d;
  CK(hipMalloc(&d, n * sizeof(T)));
  CK(hipMemcpy(d, h, n * sizeof(T), hipMemcpyHostToDevice));
  return d;
}

static int check_resize() {
  int fails = 0;
  for (int c = 0; c < N_RESIZE_CASES; ++c) {
    const ResizeCase& r = RESIZE_CASES[c];
    const size_t nin = (size_t)r.N * r.H * r.W * 3, nout = (size_t)r.N * r.crop * r.crop * 3;
    unsigned char* frames = to_dev(r.frames, nin);
    int* xb = to_dev(r.xb, (size_t)r.new_w * 2);
    int* xk = to_dev(r.xk, (size_t)r.new_w * r.xks);
    int* yb = to_dev(r.yb, (size_t)r.new_h * 2);
    int* yk = to_dev(r.yk, (size_t)r.new_h * r.yks);
    unsigned char *tmp, *out;
    CK(hipMalloc(&tmp, (size_t)r.N * (r.y1 - r.y0) * r.crop * 3));
    CK(hipMalloc(&out, nout));
    MER(mer_image_resize_crop_u8(frames, r.N, r.H, r.W, r.left, r.top, r.crop, r.crop, xb, xk, r.xks, yb, yk, r.yks, r.y0, r.y1, tmp, out, nullptr));
    CK(hipDeviceSynchronize());
    std::vector<unsigned char> got(nout);
    CK(hipMemcpy(got.data(), out, nout, hipMemcpyDeviceToHost));
    size_t bad = 0;
    for (size_t i = 0; i < nout; ++i) bad += got[i] != r.expect[i];
    printf("{\"check\": \"resize_crop_u8 vs Pillow\", \"case\": %d, \"N\": %d, \"H\": %d, \"W\": %d, \"crop\": %d, \"mismatching_bytes\": %zu, \"ok\": %s}\n",
           c, r.N, r.H, r.W, r.crop, bad, bad ? "false" : "true");
    fails += bad != 0;
  }
  return fails;
}

static unsigned lcg(unsigned& s) { s = s * 1664525u + 1013904223u; return s; }
static _Float16 rnd16(unsigned& s, float scale) { return (_Float16)(((int)(lcg(s) >> 8) % 2001 - 1000) * 0.001f * scale); }

// mer_seq_bias through the C ABI: table[s, n] = bias[n] + mean(sampled rows of sequence s)[k] * w_lo[n, k] (the sample: rows
// st / 2, st / 2 + st, ... of the sequence, st the largest power of two that leaves >= 16 of them) against the same sum on the host
static int check_seq_bias(int T, int nseq) {
  const int K = 768, N = 528, M = T * nseq - T / 3;   // the last sequence is partial
  unsigned seed = 777u + T;
  std::vector<_Float16> a((size_t)M * K), wl((size_t)N * K);
  std::vector<float> bias(N);
  for (auto& v : a) v = rnd16(seed, 1.0f) + (_Float16)0.25f;
  for (auto& v : wl) v = rnd16(seed, 0.001f);
  for (auto& v : bias) v = (float)rnd16(seed, 0.5f);
  _Float16 *da = to_dev(a.data(), a.size()), *dwl = to_dev(wl.data(), wl.size());
  float* dbias = to_dev(bias.data(), bias.size());
  void* scratch;
  float* dout;
  CK(hipMalloc(&scratch, (size_t)mer_seq_bias_scratch_bytes(nseq, K)));
  CK(hipMalloc(&dout, (size_t)nseq * N * 4));
  MER(mer_seq_bias(da, MER_DT_F16, K, 0, 0, M, K, T, nullptr, dwl, K, dbias, N, 0, scratch, dout, N, nullptr));
  CK(hipDeviceSynchronize());
  std::vector<float> out((size_t)nseq * N);
  CK(hipMemcpy(out.data(), dout, out.size() * 4, hipMemcpyDeviceToHost));
  int stride = 1;
  while (stride * 2 * 16 <= T) stride *= 2;
  double worst = 0, scale = 0;
  for (int q = 0; q < nseq; ++q) {
    const int valid = (q + 1) * T <= M ? T : M - q * T;
    std::vector<double> mean(K, 0.0);
    int cnt = 0;
    for (int t = stride / 2; t < valid; t += stride, ++cnt)
      for (int k = 0; k < K; ++k) mean[k] += (double)a[(size_t)(q * T + t) * K + k];
    for (int n = 0; n < N; ++n) {
      double ref = bias[n];
      for (int k = 0; k < K && cnt; ++k) ref += mean[k] / cnt * (double)wl[(size_t)n * K + k];
      worst = fmax(worst, fabs(ref - out[(size_t)q * N + n]));
      scale = fmax(scale, fabs(ref));
    }
  }
  const bool ok = worst <= 3e-4 * scale + 1e-7;   // the mean plane is rounded to 16 bits once
  printf("{\"check\": \"seq_bias == host sum over each sequence's sampled rows\", \"T\": %d, \"sequences\": %d, \"max_abs_err\": %.3g, \"ok\": %s}\n", T, nseq, worst, ok ? "true" : "false");
  return ok ? 0 : 1;
}

int main() {
  printf("{\"library\": \"%s\"}\n", mer_version());
  int fails = check_resize();
  fails += check_seq_bias(197, 24);
  fails += check_seq_bias(64, 70);
  printf("{\"failed_checks\": %d}\n", fails);
  return fails;
}
