// abi_selftest.cpp — standalone check of libmer_hip.so through its C ABI (no Python, no torch: a GPU-box call costs ~15 s).
// Covers the code paths written after round 1's GPU budget was spent:
//   1. mer_image_resize_crop_u8 against bytes produced by Pillow on the host (selftest_vectors.h: frames, expected crop,
//      the host-built window / coefficient tables; bicubic and bilinear);
//   2. blocked activation planes: fc1 with c16_blocked must write exactly the row-major 16-bit output re-laid as
//      [M/256][N/32] LDS images, and fc2 reading that plane with a_blocked must equal fc2 on the row-major plane bit for bit
//      (one- and two-pass), M a multiple of 256 and ragged.
// Prints one JSON line per check; exit code = number of failed checks.
// Build (scripts/probes/build_probes.sh): hipcc abi_selftest.cpp -I../../include -L../../mertools_amd -lmer_hip -Wl,-rpath,...
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
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

static int check_blocked(int M) {
  const int K = 768, F = 3072, D = 768;
  const int Mp = (M + 255) / 256 * 256;
  unsigned seed = 12345u + M;
  std::vector<_Float16> a((size_t)M * K), w1((size_t)F * K), w2h((size_t)D * F), w2l((size_t)D * F);
  std::vector<float> bias(F);
  for (auto& v : a) v = rnd16(seed, 1.0f);
  for (auto& v : w1) v = rnd16(seed, 0.05f);
  for (auto& v : w2h) v = rnd16(seed, 0.05f);
  for (auto& v : w2l) v = rnd16(seed, 0.0001f);
  for (auto& v : bias) v = (float)rnd16(seed, 0.5f);
  _Float16 *da = to_dev(a.data(), a.size()), *dw1 = to_dev(w1.data(), w1.size()), *dw2h = to_dev(w2h.data(), w2h.size()), *dw2l = to_dev(w2l.data(), w2l.size());
  float* dbias = to_dev(bias.data(), bias.size());
  _Float16 *h_row, *h_blk;
  float *y_row, *y_blk;
  CK(hipMalloc(&h_row, (size_t)M * F * 2));
  CK(hipMalloc(&h_blk, (size_t)Mp * F * 2));
  CK(hipMemset(h_blk, 0, (size_t)Mp * F * 2));
  CK(hipMalloc(&y_row, (size_t)M * D * 4));
  CK(hipMalloc(&y_blk, (size_t)M * D * 4));
  mer_gemm16_args g;
  memset(&g, 0, sizeof(g));
  g.M = M; g.N = F; g.K = K; g.dtype = MER_DT_F16;
  g.a_hi = da; g.lda = K; g.w_hi = dw1; g.ldw = K; g.bias = dbias; g.act = MER_ACT_GELU;
  g.c16_hi = h_row; g.ldc16 = F; g.nbatch = 1; g.nb_inner = 1; g.passes = 1;
  MER(mer_gemm16(&g, nullptr));
  g.c16_hi = h_blk; g.c16_blocked = 1;
  MER(mer_gemm16(&g, nullptr));
  CK(hipDeviceSynchronize());
  std::vector<_Float16> row((size_t)M * F), blk((size_t)Mp * F);
  CK(hipMemcpy(row.data(), h_row, row.size() * 2, hipMemcpyDeviceToHost));
  CK(hipMemcpy(blk.data(), h_blk, blk.size() * 2, hipMemcpyDeviceToHost));
  size_t bad = 0;
  for (int r = 0; r < M; ++r)
    for (int c = 0; c < F; ++c) {
      const int rr = r & 255, lc = (c & 31) >> 3;
      const size_t o = ((((size_t)(r >> 8) * (F / 32) + (c >> 5)) * 256 + rr) << 5) + ((lc ^ ((-(rr >> 2)) & 3)) << 3) + (c & 7);
      bad += memcmp(&blk[o], &row[(size_t)r * F + c], 2) != 0;
    }
  printf("{\"check\": \"c16_blocked layout == row-major output\", \"M\": %d, \"N\": %d, \"mismatching_elements\": %zu, \"ok\": %s}\n", M, F, bad, bad ? "false" : "true");
  int fails = bad != 0;
  for (int passes = 1; passes <= 2; ++passes) {
    memset(&g, 0, sizeof(g));
    g.M = M; g.N = D; g.K = F; g.dtype = MER_DT_F16;
    g.a_hi = h_row; g.lda = F; g.w_hi = dw2h; g.w_lo = passes == 2 ? dw2l : nullptr; g.ldw = F;
    g.c32 = y_row; g.ldc32 = D; g.nbatch = 1; g.nb_inner = 1; g.passes = passes; g.tile = 3;
    MER(mer_gemm16(&g, nullptr));
    g.a_hi = h_blk; g.a_blocked = 1; g.c32 = y_blk;
    MER(mer_gemm16(&g, nullptr));
    CK(hipDeviceSynchronize());
    std::vector<float> yr((size_t)M * D), yb((size_t)M * D);
    CK(hipMemcpy(yr.data(), y_row, yr.size() * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(yb.data(), y_blk, yb.size() * 4, hipMemcpyDeviceToHost));
    const bool same = memcmp(yr.data(), yb.data(), yr.size() * 4) == 0;
    double asum = 0;
    for (float v : yr) asum += v < 0 ? -v : v;
    printf("{\"check\": \"a_blocked consumer == row-major consumer\", \"M\": %d, \"passes\": %d, \"bit_identical\": %s, \"mean_abs\": %.4f, \"ok\": %s}\n",
           M, passes, same ? "true" : "false", asum / yr.size(), (same && asum > 0) ? "true" : "false");
    fails += !(same && asum > 0);
  }
  return fails;
}

int main() {
  printf("{\"library\": \"%s\"}\n", mer_version());
  int fails = check_resize();
  fails += check_blocked(2048);
  fails += check_blocked(1500);
  printf("{\"failed_checks\": %d}\n", fails);
  return fails;
}
