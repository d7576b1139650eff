// gemm16_bench.cpp — times the REAL mer_gemm16 kernels through the C ABI of libmer_hip.so, without Python or torch, so that one
// GPU-box call costs ~15 s instead of ~45 s (no interpreter / torch import): the tool for A/B-ing kernel changes next round.
//
//   gemm16_bench.bin [warm] [reps] [set]      set: clip (default) | hubert | roberta | hubert32 | square | all;   MER_TILE=3|4 forces a tile class
//
// For every (shape, epilogue) of the bench's block GEMMs it runs pre-blocked W (mer_w_block_pack) with hot operands (one set of planes,
// re-launched) and with cold ones (four rotating A / output plane sets).  Operands are pseudo-random f16 values: constant-filled planes
// toggle so few bits that the chip clocks 15-20 % higher than on real activations.  One JSON line per configuration:
// microseconds per launch (hipEvents around `reps` back-to-back launches after `warm` warm-up launches: sustained clocks) and
// algorithmic TFLOP/s.  Debug switches can be set for a run with MER_SET="gemm_dbg_skip=1" (mer_set_option).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include "mer_hip.h"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(100); } } while (0)
#define MER(x) do { int rc_ = (x); if (rc_ != 0) { fprintf(stderr, "%s:%d rc=%d %s\n", __FILE__, __LINE__, rc_, mer_last_error()); exit(101); } } while (0)

extern "C" int mer_set_option(const char* name, int value);
extern "C" int mer_set_debug_buffer(void* device_u64_buffer);

struct Shape { const char* name; int M, N, K; int passes; int act; bool residual, out32, out16; };

static void* dev_alloc(size_t bytes, int fill) {
  void* p;
  CK(hipMalloc(&p, bytes));
  CK(hipMemset(p, fill, bytes));
  return p;
}

// pseudo-random f16 values in (-scale, scale): data-dependent power makes constant planes unrepresentative
__global__ void fill_rand16(_Float16* p, size_t n, float scale, unsigned seed) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    unsigned h = (unsigned)i * 2654435761u + seed;
    h ^= h >> 15; h *= 2246822519u; h ^= h >> 13; h *= 3266489917u; h ^= h >> 16;
    p[i] = (_Float16)(((int)(h & 0xffff) - 32768) * (scale / 32768.0f));
  }
}
__global__ void fill_rand32(float* p, size_t n, float scale, unsigned seed) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    unsigned h = (unsigned)i * 2654435761u + seed;
    h ^= h >> 15; h *= 2246822519u; h ^= h >> 13; h *= 3266489917u; h ^= h >> 16;
    p[i] = ((int)(h & 0xffff) - 32768) * (scale / 32768.0f);
  }
}
__global__ void count_diff(const unsigned* a, const unsigned* b, size_t nwords, unsigned long long* nd) {
  unsigned long long c = 0;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < nwords; i += (size_t)gridDim.x * blockDim.x) c += a[i] != b[i];
  if (c) atomicAdd(nd, c);
}
static void* dev_rand32(size_t elems, float scale, unsigned seed) {
  void* p;
  CK(hipMalloc(&p, elems * 4));
  hipLaunchKernelGGL(fill_rand32, dim3(2048), dim3(256), 0, 0, (float*)p, elems, scale, seed);
  CK(hipGetLastError());
  return p;
}
static void* dev_rand16(size_t elems, float scale, unsigned seed) {
  void* p;
  CK(hipMalloc(&p, elems * 2));
  hipLaunchKernelGGL(fill_rand16, dim3(2048), dim3(256), 0, 0, (_Float16*)p, elems, scale, seed);
  CK(hipGetLastError());
  return p;
}

static float time_gemm(const mer_gemm16_args& g, int warm, int reps) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i = 0; i < warm; ++i) MER(mer_gemm16(&g, nullptr));
  CK(hipEventRecord(e0, 0));
  for (int i = 0; i < reps; ++i) MER(mer_gemm16(&g, nullptr));
  CK(hipEventRecord(e1, 0));
  CK(hipEventSynchronize(e1));
  float ms = 0;
  CK(hipEventElapsedTime(&ms, e0, e1));
  CK(hipEventDestroy(e0)); CK(hipEventDestroy(e1));
  return ms * 1e3f / reps;
}

// the same launch cycling through `n` argument sets (different A / output planes): the working set leaves the 256-MB Infinity Cache,
// which is what a GEMM meets inside the encoder (its A plane was just written by the producer, nothing of it is cached from a previous launch)
static float time_gemm_rot(const mer_gemm16_args* g, int n, int warm, int reps) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i = 0; i < warm; ++i) MER(mer_gemm16(&g[i % n], nullptr));
  CK(hipEventRecord(e0, 0));
  for (int i = 0; i < reps; ++i) MER(mer_gemm16(&g[i % n], nullptr));
  CK(hipEventRecord(e1, 0));
  CK(hipEventSynchronize(e1));
  float ms = 0;
  CK(hipEventElapsedTime(&ms, e0, e1));
  CK(hipEventDestroy(e0)); CK(hipEventDestroy(e1));
  return ms * 1e3f / reps;
}

static void run_shape(const Shape& s, int warm, int reps) {
  const int Mp = (s.M + 255) / 256 * 256;
  void* a = dev_rand16((size_t)Mp * s.K, 2.0f, 1u);
  void* w = dev_rand16((size_t)s.N * s.K, 0.05f, 2u);
  void* wlo = s.passes >= 2 && s.passes != 4 ? dev_alloc((size_t)s.N * s.K * 2, 0x11) : nullptr;
  void* wblk = dev_alloc((size_t)mer_w_block_bytes(s.N, s.K), 0);
  void* wlo_blk = wlo ? dev_alloc((size_t)mer_w_block_bytes(s.N, s.K), 0) : nullptr;
  void* wblkp = (s.N % 256 == 0 && s.K % 32 == 0) ? dev_alloc((size_t)s.N * s.K * 2, 0) : nullptr;   // the persistent kernel's plane
  MER(mer_w_block_pack(w, s.K, s.N, s.K, wblk, nullptr));
  if (wblkp) MER(mer_w_block_pack_p(w, s.K, s.N, s.K, s.out16 ? 0 : 1, wblkp, nullptr));
  if (wlo) MER(mer_w_block_pack(wlo, s.K, s.N, s.K, wlo_blk, nullptr));
  void* wmx = nullptr;
  if (s.passes == 4) {
    const long long nb = mer_mx_packed_bytes(s.N, s.K);
    std::vector<float> res((size_t)s.N * s.K);
    unsigned sd = 1;
    for (auto& v : res) { sd = sd * 1664525u + 1013904223u; v = ((int)(sd >> 9) % 2001 - 1000) * 1e-7f; }
    std::vector<unsigned char> packed((size_t)nb);
    MER(mer_mx_pack(res.data(), s.K, s.N, s.K, packed.data()));
    CK(hipMalloc(&wmx, (size_t)nb));
    CK(hipMemcpy(wmx, packed.data(), (size_t)nb, hipMemcpyHostToDevice));
  }
  float* bias = (float*)dev_rand32((size_t)s.N, 1.0f, 3u);
  float* resid = s.residual ? (float*)dev_rand32((size_t)s.M * s.N, 1.0f, 4u) : nullptr;
  float* c32 = s.out32 ? (float*)dev_alloc((size_t)s.M * s.N * 4, 0) : nullptr;
  void* c16 = s.out16 ? dev_alloc((size_t)Mp * s.N * 2, 0) : nullptr;
  mer_gemm16_args g;
  memset(&g, 0, sizeof(g));
  g.M = s.M; g.N = s.N; g.K = s.K; g.dtype = MER_DT_F16;
  g.a_hi = a; g.lda = s.K; g.w_hi = w; g.w_lo = wlo; g.w_mx = wmx; g.ldw = s.K;
  g.bias = bias; g.act = s.act; g.residual = resid; g.ldr = s.N;
  g.c32 = c32; g.ldc32 = s.N; g.c16_hi = c16; g.ldc16 = s.N;
  g.nbatch = 1; g.nb_inner = 1; g.passes = s.passes;
  if (const char* t = getenv("MER_TILE")) g.tile = atoi(t);   // 0 auto, 3 = 256x256, 4 = 128x256 (two workgroups per CU)
  const double flops = 2.0 * s.M * (double)s.N * s.K;
  auto report = [&](const char* variant, float us) {
    printf("{\"shape\": \"%s\", \"M\": %d, \"N\": %d, \"K\": %d, \"passes\": %d, \"variant\": \"%s\", \"us\": %.1f, \"TFLOPs\": %.0f}\n",
           s.name, s.M, s.N, s.K, s.passes, variant, us, flops / us * 1e-6);
    fflush(stdout);
  };
  g.w_hi_blk = wblk; g.w_lo_blk = wlo_blk; g.w_hi_blkp = s.out16 ? wblkp : nullptr; g.w_hi_blkq = s.out16 ? nullptr : wblkp;
  // interleaved A/B in one process (same clocks, same box): the tile kernel (gemm16_kernel) against the persistent one, twice each
  for (int round = 0; round < 2; ++round) {
    MER(mer_set_option("gemm_persist", 0));
    report("tile kernel (gemm16_kernel), pre-blocked W", time_gemm(g, warm, reps));
    MER(mer_set_option("gemm_persist", 1));
    if (wblkp && s.passes == 1) report("persistent kernel (gemm16p_kernel)", time_gemm(g, warm, reps));
    if (wblkp && s.passes == 1 && getenv("MER_TM_AB")) {   // 256-row against 192-row tiles on the same planes (the default picks per shape)
      MER(mer_set_option("gemm_tm", 4)); report("persistent, 256-row tiles", time_gemm(g, warm, reps));
      MER(mer_set_option("gemm_tm", 3)); report("persistent, 192-row tiles", time_gemm(g, warm, reps));
      MER(mer_set_option("gemm_tm", 0));
    }
  }
  if (wblkp && s.passes == 1 && getenv("MER_CHECK")) {   // the persistent kernel's bits against the tile kernel's on this shape (rows < M)
    const size_t obytes = s.out16 ? (size_t)s.M * s.N * 2 : (size_t)s.M * s.N * 4;
    void* outp = s.out16 ? c16 : (void*)c32;
    void* ref = dev_alloc(obytes, 0);
    unsigned long long* nd = (unsigned long long*)dev_alloc(8, 0);
    MER(mer_set_option("gemm_persist", 0));
    MER(mer_gemm16(&g, nullptr));
    CK(hipMemcpy(ref, outp, obytes, hipMemcpyDeviceToDevice));
    for (int rep = 0; rep < 3; ++rep) {
      CK(hipMemset(outp, 0xff, obytes));
      CK(hipMemset(nd, 0, 8));
      MER(mer_set_option("gemm_persist", 1));
      MER(mer_gemm16(&g, nullptr));
      hipLaunchKernelGGL(count_diff, dim3(2048), dim3(256), 0, 0, (const unsigned*)ref, (const unsigned*)outp, obytes / 4, nd);
      unsigned long long h = 0;
      CK(hipMemcpy(&h, nd, 8, hipMemcpyDeviceToHost));
      printf("{\"shape\": \"%s\", \"check\": \"persistent vs tile kernel\", \"rep\": %d, \"differing_words\": %llu}\n", s.name, rep, h);
      fflush(stdout);
    }
    CK(hipFree(ref)); CK(hipFree(nd));
  }
  if (const char* sd = getenv("MER_STAMP")) if (wblkp && s.passes == 1) {   // s_memtime timeline of the persistent kernel (gemm16p_impl.h: stamp slots)
    const size_t nb = 256 * 384 * 8;
    void* dbg = dev_alloc(nb, 0);
    MER(mer_set_option("gemm_persist", 1));
    for (int i = 0; i < 5; ++i) MER(mer_gemm16(&g, nullptr));
    MER(mer_set_debug_buffer(dbg));
    MER(mer_gemm16(&g, nullptr));
    CK(hipDeviceSynchronize());
    MER(mer_set_debug_buffer(nullptr));
    std::vector<char> host(nb);
    CK(hipMemcpy(host.data(), dbg, nb, hipMemcpyDeviceToHost));
    char fn[512];
    static int idx = 0;
    snprintf(fn, sizeof(fn), "%s/stamps_%02d.bin", sd, idx++);
    if (FILE* f = fopen(fn, "wb")) { fwrite(host.data(), 1, nb, f); fclose(f); }
    printf("{\"shape\": \"%s\", \"stamps\": \"%s\"}\n", s.name, fn);
    CK(hipFree(dbg));
  }
  if (getenv("MER_STAGGER_AB") && wblkp && s.passes == 1 && s.residual) {   // the free stagger of the fp32 + residual launches, off / on, interleaved
    for (int round = 0; round < 3; ++round) {
      MER(mer_set_option("gemm_dbg_skip", 8)); report("persistent, free stagger OFF", time_gemm(g, warm, reps));
      MER(mer_set_option("gemm_dbg_skip", 0)); report("persistent, free stagger on", time_gemm(g, warm, reps));
    }
  }
  if (getenv("MER_DECOMP") && wblkp && s.passes == 1) {   // where the persistent kernel's time goes: stores skipped / epilogue skipped
    MER(mer_set_option("gemm_dbg_skip", 1)); report("persistent, stores skipped", time_gemm(g, warm, reps));
    MER(mer_set_option("gemm_dbg_skip", 2)); report("persistent, epilogue skipped", time_gemm(g, warm, reps));
    MER(mer_set_option("gemm_dbg_skip", 0));
  }
  {   // cold operands: 4 rotating sets of A / residual / output planes (weights stay: they are small and shared by all row tiles)
    constexpr int R = 4;
    mer_gemm16_args gr[R];
    std::vector<void*> extra;
    for (int r = 0; r < R; ++r) {
      gr[r] = g;
      if (r == 0) continue;
      void* a2 = dev_rand16((size_t)Mp * s.K, 2.0f, 10u + r); extra.push_back(a2); gr[r].a_hi = a2;
      if (s.residual) { void* p = dev_alloc((size_t)s.M * s.N * 4, 0); extra.push_back(p); gr[r].residual = (float*)p; }
      if (s.out32) { void* p = dev_alloc((size_t)s.M * s.N * 4, 0); extra.push_back(p); gr[r].c32 = (float*)p; }
      if (s.out16) { void* p = dev_alloc((size_t)Mp * s.N * 2, 0); extra.push_back(p); gr[r].c16_hi = p; }
    }
    report("4 rotating A / output planes (cold operands), default kernel selection", time_gemm_rot(gr, R, warm, reps));
    for (void* p : extra) CK(hipFree(p));
  }
  for (void* p : {a, w, wlo, wblk, wlo_blk, wblkp, wmx, (void*)bias, (void*)resid, (void*)c32, c16})
    if (p) CK(hipFree(p));
}

int main(int argc, char** argv) {
  const int warm = argc > 1 ? atoi(argv[1]) : 30, reps = argc > 2 ? atoi(argv[2]) : 30;
  const char* set = argc > 3 ? argv[3] : "clip";
  if (const char* opts = getenv("MER_SET")) {
    char buf[512];
    strncpy(buf, opts, sizeof(buf) - 1); buf[sizeof(buf) - 1] = 0;
    for (char* tok = strtok(buf, ","); tok; tok = strtok(nullptr, ",")) {
      char* eq = strchr(tok, '=');
      const int v = eq ? atoi(eq + 1) : 1;
      if (eq) *eq = 0;
      MER(mer_set_option(tok, v));
      printf("{\"option\": \"%s\", \"value\": %d}\n", tok, v);
    }
  }
  printf("{\"library\": \"%s\", \"warm\": %d, \"reps\": %d}\n", mer_version(), warm, reps);
  // the block GEMMs of bench.py's step under the default "mean" preset (every GEMM one f16 pass; the correction is a bias)
  // CLIP-ViT-B/16: 64 clips x 8 frames x 197 tokens
  const int Mc = 100864;
  const Shape clip[] = {
      {"clip QKV", Mc, 2304, 768, 1, MER_ACT_NONE, false, false, true},
      {"clip out-proj (residual, fp32 out)", Mc, 768, 768, 1, MER_ACT_NONE, true, true, false},
      {"clip fc1 (quick_gelu)", Mc, 3072, 768, 1, MER_ACT_QUICK_GELU, false, false, true},
      {"clip fc2 (residual, fp32 out)", Mc, 768, 3072, 1, MER_ACT_NONE, true, true, false},
  };
  // HuBERT-base: 64 clips x 249 frames; conv1 of the feature extractor (M = 64 x 7999)
  const int Mh = 15936;
  const Shape hubert[] = {
      {"hubert QKV", Mh, 2304, 768, 1, MER_ACT_NONE, false, false, true},
      {"hubert out-proj (residual, fp32 out)", Mh, 768, 768, 1, MER_ACT_NONE, true, true, false},
      {"hubert fc1 (gelu)", Mh, 3072, 768, 1, MER_ACT_GELU, false, false, true},
      {"hubert fc2 (residual, fp32 out)", Mh, 768, 3072, 1, MER_ACT_NONE, true, true, false},
      {"hubert conv1-like (gelu, M = 511936, K = 1536, dense rows)", 511936, 512, 1536, 1, MER_ACT_GELU, false, false, true},
  };
  // RoBERTa-base: 64 clips x 64 tokens
  const int Mr = 4096;
  const Shape roberta[] = {
      {"roberta QKV", Mr, 2304, 768, 1, MER_ACT_NONE, false, false, true},
      {"roberta out-proj (residual, fp32 out)", Mr, 768, 768, 1, MER_ACT_NONE, true, true, false},
      {"roberta fc1 (gelu)", Mr, 3072, 768, 1, MER_ACT_GELU, false, false, true},
      {"roberta fc2 (residual, fp32 out)", Mr, 768, 3072, 1, MER_ACT_NONE, true, true, false},
  };
  // HuBERT-base at BASELINE configs[1]'s batch 32: 32 x 249 rows (32 row tiles of 256)
  const int Mh2 = 7968;
  const Shape hubert32[] = {
      {"hubert b32 QKV", Mh2, 2304, 768, 1, MER_ACT_NONE, false, false, true},
      {"hubert b32 out-proj (residual, fp32 out)", Mh2, 768, 768, 1, MER_ACT_NONE, true, true, false},
      {"hubert b32 fc1 (gelu)", Mh2, 3072, 768, 1, MER_ACT_GELU, false, false, true},
      {"hubert b32 fc2 (residual, fp32 out)", Mh2, 768, 3072, 1, MER_ACT_NONE, true, true, false},
  };
  // the guide's calibration shapes (cdna_hip_programming.md §5: its plain-HIP 256^2 8-phase template reaches ~1320-1340 TF at 4096^3 and
  // ~1470 TF at 8192^3 on uniform random operands): what THIS kernel does on them says how much of the gap to that template is the
  // K loop and how much is the encoder's shapes (K = 768: a tile boundary every 24 slabs, fp32 + residual epilogues, partial rounds)
  const Shape square[] = {
      {"4096^3 (16-bit out)", 4096, 4096, 4096, 1, MER_ACT_NONE, false, false, true},
      {"8192^3 (16-bit out)", 8192, 8192, 8192, 1, MER_ACT_NONE, false, false, true},
      {"M = 100864, N = 2304, K = 4096 (CLIP QKV's plane with a long K)", Mc, 2304, 4096, 1, MER_ACT_NONE, false, false, true},
  };
  const bool all = !strcmp(set, "all");
  if (!strcmp(set, "square")) for (const Shape& s : square) run_shape(s, warm, reps);
  if (all || !strcmp(set, "clip")) for (const Shape& s : clip) run_shape(s, warm, reps);
  if (all || !strcmp(set, "hubert")) for (const Shape& s : hubert) run_shape(s, warm, reps);
  if (all || !strcmp(set, "roberta")) for (const Shape& s : roberta) run_shape(s, warm, reps);
  if (all || !strcmp(set, "hubert32")) for (const Shape& s : hubert32) run_shape(s, warm, reps);
  return 0;
}
