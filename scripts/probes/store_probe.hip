// store_probe.hip — what bounds the GEMM epilogue's global stores?  (profiles/r02_gemm16_bench_epilogue_split.txt: the stores
// alone are 22-24 % of the K = 768 GEMMs, and spreading the CUs' phases does not help.)
// Every workgroup (512 threads, one per CU) writes `tiles` C tiles of 256 x 256 16-bit elements exactly as gemm16's epilogue
// does: a wave owns 128 rows x 64 columns, one instruction = 8 rows x 128 B (8 lanes x 16 B per row), rows ldc apart.
//   store_probe.bin            -> one JSON line per (store flavour, number of active workgroups)
// Flavours: plain global_store_dwordx4, nt, sc1, sc0 sc1 (write-through), and 4-byte-per-lane stores for scale.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(100); } } while (0)
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int MODE>
__device__ __forceinline__ void st16(void* p, u32x4 v) {
  if (MODE == 0) asm volatile("global_store_dwordx4 %0, %1, off" :: "v"(p), "v"(v) : "memory");
  if (MODE == 1) asm volatile("global_store_dwordx4 %0, %1, off nt" :: "v"(p), "v"(v) : "memory");
  if (MODE == 2) asm volatile("global_store_dwordx4 %0, %1, off sc1" :: "v"(p), "v"(v) : "memory");
  if (MODE == 3) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" :: "v"(p), "v"(v) : "memory");
}

// N = row length in elements (ldc), tiles walk down the rows of a [rows, N] plane, column tile = blockIdx.x % (N / 256)
template <int MODE>
__global__ __launch_bounds__(512) void store_kernel(unsigned short* c, int N, int tiles, long long rows_total, unsigned long long* cyc) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 2, wn = wave & 3;
  const int c8 = lane & 7, rsub = lane >> 3;
  const int ntn = N / 256;
  const int tn = blockIdx.x % ntn;
  u32x4 v = {(unsigned)tid, 1u, 2u, 3u};
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int t = 0; t < tiles; ++t) {
    const long long tm = ((long long)(blockIdx.x / ntn) * tiles + t) % (rows_total / 256);
    unsigned short* base = c + (tm * 256 + wm * 128) * (long long)N + tn * 256 + wn * 64 + c8 * 8;
#pragma unroll 4
    for (int it = 0; it < 16; ++it) {
      unsigned short* p = base + (long long)(it * 8 + rsub) * N;
      if (MODE == 4) {   // 4 B per lane: 4x the instructions for the same bytes
        for (int j = 0; j < 4; ++j) asm volatile("global_store_dword %0, %1, off" :: "v"(p + 2 * j), "v"(v[j]) : "memory");
      } else st16<MODE>(p, v);
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (tid == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int MODE>
static void run(const char* name, unsigned short* c, int N, long long rows, unsigned long long* dcyc, int blocks, int tiles) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  hipLaunchKernelGGL((store_kernel<MODE>), dim3(blocks), dim3(512), 0, 0, c, N, tiles, rows, dcyc);   // warm-up
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0, 0));
  hipLaunchKernelGGL((store_kernel<MODE>), dim3(blocks), dim3(512), 0, 0, c, N, tiles, rows, dcyc);
  CK(hipEventRecord(e1, 0));
  CK(hipEventSynchronize(e1));
  float ms = 0;
  CK(hipEventElapsedTime(&ms, e0, e1));
  std::vector<unsigned long long> h(blocks);
  CK(hipMemcpy(h.data(), dcyc, blocks * 8, hipMemcpyDeviceToHost));
  double avg = 0;
  for (auto x : h) avg += (double)x;
  avg /= blocks;
  const double bytes_wg = (double)tiles * 256 * 256 * 2;
  printf("{\"flavour\": \"%s\", \"N\": %d, \"workgroups\": %d, \"tiles_per_wg\": %d, \"cycles_per_tile\": %.0f, \"B_per_clk_per_CU\": %.2f, \"chip_TBps\": %.2f}\n",
         name, N, blocks, tiles, avg / tiles, bytes_wg / avg, bytes_wg * blocks / (ms * 1e-3) / 1e12);
  fflush(stdout);
}

int main() {
  const int N = 1536;                    // CLIP Q|K output plane
  const long long rows = 100864 / 256 * 256;
  unsigned short* c;
  unsigned long long* dcyc;
  CK(hipMalloc(&c, (size_t)rows * N * 2 + (1 << 20)));
  CK(hipMalloc(&dcyc, 4096 * 8));
  for (int blocks : {8, 32, 64, 128, 256}) {
    const int tiles = 8;
    run<0>("dwordx4", c, N, rows, dcyc, blocks, tiles);
    run<1>("dwordx4 nt", c, N, rows, dcyc, blocks, tiles);
    run<2>("dwordx4 sc1", c, N, rows, dcyc, blocks, tiles);
    run<3>("dwordx4 sc0 sc1", c, N, rows, dcyc, blocks, tiles);
    run<4>("dword x4", c, N, rows, dcyc, blocks, tiles);
  }
  return 0;
}
