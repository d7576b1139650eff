// ceiling_probe.hip — standalone (no torch) measurement of the ceilings that bound gemm16's tile on gfx950:
//   ring:   the K loop's data path in isolation — an NS-stage LDS ring of 256x256x32 f16 slabs filled by LDS-DMA
//           (global_load_lds_dwordx4, counted vmcnt, one s_barrier per slab), optionally with the fragment reads
//           (ds_read_b128) and the MFMAs of a register-double-buffered K loop.  NW = 8: 2x4 waves of 128x64 (gemm16's
//           wave grid, 256 registers); NW = 4: 2x2 waves of 128x128 (one wave per SIMD, accumulators in 256 AGPRs).
//           Answers: how many L2->LDS bytes per clock per CU the DMA path sustains with every CU streaming, and what a
//           single-barrier, software-pipelined K loop would cost per slab next to gemm16's two-phase schedule.
//   store:  the epilogue's store path in isolation — every wave writes 16-byte lane stores whose lanes cover row
//           segments of SEG bytes (gemm16 today: 128 B for f16 outputs of a 64-column wave tile).
//   template8: the guide's 256^2 BK = 64 8-phase template, whole (round 6) — checked against the two-group loop, then timed beside it.
// Output: one JSON object per line on stdout.  Numerical results are meaningless for the zero-filled sets (only the instruction
// streams and the memory traffic are real); the phase8 / template8 sets fill the operands pseudo-randomly and template8 compares results.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -o ceiling_probe.bin ceiling_probe.hip   (scripts/probes/build_probes.sh)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <type_traits>
#include <vector>

typedef _Float16 f16;
typedef f16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void lds_void_t;
typedef const __attribute__((address_space(1))) void glb_void_t;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(2); } } while (0)

struct RingParams {
  const f16* a; const f16* w;
  int M, N, K, tiles_m, tiles_n;
  unsigned long long* stamps;   // 2 per workgroup: start, end (s_memtime)
  float* sink;
  float* tile_sum;              // optional [tiles]: every thread adds its accumulators' sum to its tile's word (the template8 cross-check)
};

template <int N>
__device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

__device__ __forceinline__ int swz4(int row) { return (-(row >> 2)) & 3; }   // conflict-free for 64-byte LDS rows

// LDS-DMA with a wave-uniform 64-bit base in SGPRs + a 32-bit per-lane byte offset (the form gemm16's MX kernel uses):
// no 64-bit VGPR address arithmetic per piece
__device__ __forceinline__ void dma16_sbase(const void* sbase_any, unsigned voff, unsigned lds_off) {
  const unsigned long long pv = (unsigned long long)sbase_any;
  const unsigned long long sbase = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(pv >> 32)) << 32) |
                                   (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)pv);
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1"
               :: "v"(voff), "s"(sbase), "s"(__builtin_amdgcn_readfirstlane(lds_off)) : "memory");
}
__device__ __forceinline__ unsigned lds_offset_of(const void* p) {
  return (unsigned)(unsigned long)(__attribute__((address_space(3))) const void*)p;
}

// MODE bit 0: fragment reads, bit 1: MFMAs (implies bit 0), bit 2: no DMA at all (MFMA / ds_read ceiling on stale LDS)
// MF: MFMA shape, 16 = v_mfma_f32_16x16x32_f16 (gemm16's), 32 = v_mfma_f32_32x32x16_f16 (half the instructions per slab)
// LAYOUT: which bytes one 1-KiB DMA piece (one wave-instruction) covers — every layout moves the same 32 KB per slab and
// the same bytes per tile with the same reuse across tiles, only the contiguous run per matrix row differs:
//   0  16 rows x  64 B  (gemm16 today: BK = 32, slab s = k-range [32s, 32s+32) of all 256 rows)
//   1  as 0, but the DMAs of slabs (2j, 2j+1) are issued together, piece by piece, so that the two 64-B halves of a
//      128-B line are requested back to back
//   2   8 rows x 128 B  (slab s = k-range [64 (s/2), +64) of rows 128 (s%2) ..)
//   3   4 rows x 256 B  (slab s = k-range [128 (s/4), +128) of rows 64 (s%4) ..)
//   4   1 KiB contiguous (operands pre-blocked [tile][slab][row][32]: what an offline-packed weight plane could look like)
template <int NW, int NS, int MODE, int MF = 16, int LAYOUT = 0>
__global__ __launch_bounds__(NW * 64) void ring_kernel(const RingParams p) {
  constexpr int BM = 256, BN = 256, BK = 32, RB = BK * 2;
  constexpr int A_PLANE = BM * RB, STAGE = (BM + BN) * RB;      // 32 KB per slab
  constexpr int WM = 2, WN = NW / 2;
  constexpr int SM = BM / WM, SN = BN / WN, TM = SM / 16, TN = SN / 16;
  constexpr int PA = 16 / NW * 1;          // 1-KiB DMA pieces of the A plane per wave per slab (16 pieces per plane)
  constexpr int LPS = 2 * PA;
  constexpr int D = NS - 1;
  constexpr bool READS = (MODE & 3) != 0, MATH = (MODE & 2) != 0, NODMA = (MODE & 4) != 0;
  __shared__ __attribute__((aligned(16))) char smem[NS * STAGE];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN, li = lane & 15, lg = lane >> 4;
  if (tid == 0) p.stamps[blockIdx.x * 2] = __builtin_amdgcn_s_memtime();

  // XCD-aware bijective tile map (same idea as gemm16: blocks b, b+8, .. share an XCD and get neighbouring tiles)
  const int nblk = p.tiles_m * p.tiles_n;
  const int L = blockIdx.x, xcd = L & 7, loc = L >> 3, q = nblk >> 3, r = nblk & 7;
  const int t = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
  const int tn = t % p.tiles_n, tm = t / p.tiles_n;

  // DMA piece j of a plane = tile rows 16j .. 16j+15 (lane l: row 16j + l/4, physical chunk l%4 <- logical chunk ^ swizzle)
  const int nk = p.K / BK;
  unsigned a_src[PA], w_src[PA];   // element offsets (< 2^32: the probe matrices are < 4 G elements)
#pragma unroll
  for (int i = 0; i < PA; ++i) {
    const int j = wave + i * NW;
    if (LAYOUT == 4) {
      a_src[i] = ((unsigned)(tm * nk) * 16 + j) * 512 + lane * 8;
      w_src[i] = ((unsigned)(tn * nk) * 16 + j) * 512 + lane * 8;
    } else {
      constexpr int LPR = LAYOUT == 2 ? 8 : (LAYOUT == 3 ? 16 : 4);   // lanes (16-byte chunks) per row run
      const int row = (64 / LPR) * j + lane / LPR;
      const int ch = LAYOUT >= 2 ? lane % LPR : ((lane & 3) ^ swz4(row));
      int m = tm * BM + row; m = m < p.M ? m : p.M - 1;
      int n = tn * BN + row; n = n < p.N ? n : p.N - 1;
      a_src[i] = (unsigned)m * (unsigned)p.K + ch * 8;
      w_src[i] = (unsigned)n * (unsigned)p.K + ch * 8;
    }
  }
  auto slab_off = [&](int kt) -> unsigned {   // element offset of slab kt relative to a_src / w_src
    if (LAYOUT == 2) return (unsigned)(kt & 1) * 128u * (unsigned)p.K + (unsigned)(kt >> 1) * 64u;
    if (LAYOUT == 3) return (unsigned)(kt & 3) * 64u * (unsigned)p.K + (unsigned)(kt >> 2) * 128u;
    if (LAYOUT == 4) return (unsigned)kt * 16u * 512u;
    return (unsigned)kt * BK;
  };
  auto issue = [&](int kt, int stage) {
    if (NODMA) return;
    char* base = smem + stage * STAGE;
    const unsigned so = slab_off(kt);
#pragma unroll
    for (int i = 0; i < PA; ++i)
      __builtin_amdgcn_global_load_lds((glb_void_t*)(p.a + a_src[i] + so), (lds_void_t*)(base + (wave + i * NW) * 1024), 16, 0, 0);
#pragma unroll
    for (int i = 0; i < PA; ++i)
      __builtin_amdgcn_global_load_lds((glb_void_t*)(p.w + w_src[i] + so), (lds_void_t*)(base + A_PLANE + (wave + i * NW) * 1024), 16, 0, 0);
  };
  auto issue_pair = [&](int kt, int stage0, int stage1) {   // LAYOUT 1: slabs kt, kt+1 piece by piece
    char* b0 = smem + stage0 * STAGE;
    char* b1 = smem + stage1 * STAGE;
#pragma unroll
    for (int i = 0; i < PA; ++i) {
      __builtin_amdgcn_global_load_lds((glb_void_t*)(p.a + a_src[i] + kt * BK), (lds_void_t*)(b0 + (wave + i * NW) * 1024), 16, 0, 0);
      __builtin_amdgcn_global_load_lds((glb_void_t*)(p.a + a_src[i] + (kt + 1) * BK), (lds_void_t*)(b1 + (wave + i * NW) * 1024), 16, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < PA; ++i) {
      __builtin_amdgcn_global_load_lds((glb_void_t*)(p.w + w_src[i] + kt * BK), (lds_void_t*)(b0 + A_PLANE + (wave + i * NW) * 1024), 16, 0, 0);
      __builtin_amdgcn_global_load_lds((glb_void_t*)(p.w + w_src[i] + (kt + 1) * BK), (lds_void_t*)(b1 + A_PLANE + (wave + i * NW) * 1024), 16, 0, 0);
    }
  };

  f32x4 acc[TM][TN];
  f32x16 acc32[TM / 2][TN / 2];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int i = 0; i < TM / 2; ++i)
#pragma unroll
    for (int j = 0; j < TN / 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc32[i][j][e] = 0.f;
  f16x8 fa[2][TM], fw[2][TN];
  u32x4 keep = {0u, 0u, 0u, 0u};
  // MF == 32: entry 2*tile + h holds k-half h of a 32-row tile (lane l: row l % 32, 16-byte chunk 2h + l / 32)
  auto load_frags = [&](const char* base, int buf) {
#pragma unroll
    for (int mt = 0; mt < TM; ++mt) {
      const int row = MF == 16 ? wm * SM + mt * 16 + li : wm * SM + (mt >> 1) * 32 + (lane & 31);
      const int ch = MF == 16 ? lg : 2 * (mt & 1) + (lane >> 5);
      fa[buf][mt] = *reinterpret_cast<const f16x8*>(base + row * RB + ((ch ^ swz4(row)) << 4));
    }
#pragma unroll
    for (int nt = 0; nt < TN; ++nt) {
      const int row = MF == 16 ? wn * SN + nt * 16 + li : wn * SN + (nt >> 1) * 32 + (lane & 31);
      const int ch = MF == 16 ? lg : 2 * (nt & 1) + (lane >> 5);
      fw[buf][nt] = *reinterpret_cast<const f16x8*>(base + A_PLANE + row * RB + ((ch ^ swz4(row)) << 4));
    }
  };
  auto consume = [&](int buf) {
    if (MATH && MF == 32) {
#pragma unroll
      for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int mt = 0; mt < TM / 2; ++mt)
#pragma unroll
          for (int nt = 0; nt < TN / 2; ++nt)
            acc32[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[buf][2 * mt + h], fw[buf][2 * nt + h], acc32[mt][nt], 0, 0, 0);
    } else if (MATH) {
#pragma unroll
      for (int mt = 0; mt < TM; ++mt)
#pragma unroll
        for (int nt = 0; nt < TN; ++nt)
          acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa[buf][mt], fw[buf][nt], acc[mt][nt], 0, 0, 0);
    } else if (READS) {
#pragma unroll
      for (int mt = 0; mt < TM; ++mt) keep ^= __builtin_bit_cast(u32x4, fa[buf][mt]);
#pragma unroll
      for (int nt = 0; nt < TN; ++nt) keep ^= __builtin_bit_cast(u32x4, fw[buf][nt]);
    }
  };

  static_assert(LAYOUT != 1 || NS == 4, "pair issue: 4-stage ring");
  // prologue: D slabs in flight, slab 0 landed, its fragments in buffer 0   (LAYOUT 1: the pair (0, 1), both landed)
  if (LAYOUT == 1) {
    if (!NODMA) issue_pair(0, 0, 1);
    wait_vmcnt<0>();
  } else {
#pragma unroll
    for (int s = 0; s < D; ++s)
      if (s < nk) issue(s, s);
    if (nk >= D) wait_vmcnt<LPS*(D - 1)>(); else wait_vmcnt<0>();
  }
  __builtin_amdgcn_s_barrier();
  if (READS) load_frags(smem, 0);
  int nxt = LAYOUT == 1 ? 2 : D, rd = 1;   // stage to refill with slab kt+D / stage holding slab kt+1
  // software-pipelined loop, two slabs per trip so that the fragment buffer index is static
  auto body = [&](int kt, auto buf_tag) {
    constexpr int buf = decltype(buf_tag)::value;
    if (LAYOUT == 1) {
      // even trips issue the pair (kt+2, kt+3) into the stages of slabs kt-2, kt-1 (K % 64 == 0) and wait for everything
      // older; odd trips wait for the even slab of the pair in flight (all but the very last DMA, which is the odd slab's)
      if (buf == 0) {
        const bool more = kt + 2 < nk;
        if (more && !NODMA) issue_pair(kt + 2, nxt, nxt + 1);
        if (more) wait_vmcnt<2 * LPS>(); else wait_vmcnt<0>();
        nxt ^= 2;
      } else {
        if (kt + 1 < nk) wait_vmcnt<1>(); else wait_vmcnt<0>();
      }
    } else {
    const bool more = kt + D < nk;
    if (more) issue(kt + D, nxt);                       // refills the stage of slab kt-1 (all its reads retired before the last barrier)
    if (more) wait_vmcnt<LPS*(D - 1)>(); else wait_vmcnt<0>();   // my share of slab kt+1 has landed
    nxt = nxt + 1 == NS ? 0 : nxt + 1;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // my reads of slab kt (issued last trip) have retired
    __builtin_amdgcn_s_barrier();
    if (READS) load_frags(smem + rd * STAGE, buf ^ 1);   // slab kt+1 -> the other buffer (last trip: a stale stage, harmless; a branch here
                                                         // would make hipcc drain lgkmcnt at the join), in the shadow of ...
    consume(buf);                                        // ... the MFMAs of slab kt
    rd = rd + 1 == NS ? 0 : rd + 1;
  };
  int kt = 0;
  for (; kt + 1 < nk; kt += 2) { body(kt, std::integral_constant<int, 0>{}); body(kt + 1, std::integral_constant<int, 1>{}); }
  if (kt < nk) body(kt, std::integral_constant<int, 0>{});

  float s = 0.f;
  if (MATH) {
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) s += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
#pragma unroll
    for (int i = 0; i < TM / 2; ++i)
#pragma unroll
      for (int j = 0; j < TN / 2; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) s += acc32[i][j][e];
  }
  s += (float)(keep[0] ^ keep[1] ^ keep[2] ^ keep[3]);
  if (s == 12345.678f) p.sink[tid] = s;   // never true for zero inputs: keeps the work alive without a store
  __syncthreads();
  if (tid == 0) p.stamps[blockIdx.x * 2 + 1] = __builtin_amdgcn_s_memtime();
}

// phase_kernel: gemm16's two-group, two-barrier K loop (8 waves = 2 x 4 of 128x64; group 1 runs one phase behind group 0;
// an iteration is LOAD(u) |bar| MATH(u) |bar| for one 32-deep k-step u) over two LDS organisations of the same 128 KB:
//   BK64 = false: four 32-KB stages of 64-byte rows, one slab issued per iteration, three in flight (gemm16 today)
//   BK64 = true : two 64-KB stages of 128-byte rows (a stage = two k-steps); the next stage's DMAs (8 rows x 128 B per
//                 piece: whole 128-B lines) are issued at the top of even k-steps and confirmed before the mid barrier
//                 of odd ones
// BLK: 0 = row-major operands, 1 = W pre-blocked [n-tile][stage-slab][1-KiB piece] (each DMA piece is 1 KiB contiguous; what an
//      offline weight packer can produce), 2 = A and W pre-blocked (upper bound: needs the activation producers to write blocked planes)
// HALF (round 5): the guide's finer phase grain on this LDS organisation — every 32-deep k-step is TWO barrier phases of 16 MFMAs (one
//      64 x 64 quadrant of the wave's 128 x 64 tile each: m-tiles 0-3, then 4-7), the first reading 4 A + 4 W fragments, the second 4 A
//      fragments into the same registers (12 reads per k-step, as before; 32 fragment registers instead of 48), the slab's A pieces
//      issued in the first half and its W pieces in the second, vmcnt waited once per k-step: the per-phase shape of
//      cdna_hip_programming.md §5 "8 phases per iteration" (2 barriers per 16 MFMAs, two wave groups one barrier apart) on four
//      32-KB stages of 64-byte rows.  BK64 = false only.
template <bool BK64, bool DO_MATH, int BLK = 0, bool SADDR = false, int HALF = 0>
__global__ __launch_bounds__(512) void phase_kernel(const RingParams p) {
  constexpr int RB = BK64 ? 128 : 64, C = RB / 16, STAGE = 512 * RB, NS = BK64 ? 2 : 4, PLANE = 256 * RB;
  constexpr int PA = PLANE / 1024 / 8, LPS = 2 * PA, RPP = 64 / C;   // pieces per wave per plane, DMAs per wave per stage, rows per piece
  constexpr int TM = 8, TN = 4;
  __shared__ __attribute__((aligned(16))) char smem[NS * STAGE];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 2, wn = wave & 3, li = lane & 15, lg = lane >> 4;
  if (tid == 0) p.stamps[blockIdx.x * 2] = __builtin_amdgcn_s_memtime();
  const int nblk = p.tiles_m * p.tiles_n;
  const int L = blockIdx.x, xcd = L & 7, loc = L >> 3, q = nblk >> 3, r = nblk & 7;
  const int t = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
  const int tn = t % p.tiles_n, tm = t / p.tiles_n;
  auto swz = [](int row) { return C == 8 ? ((row >> 1) & 7) : ((-(row >> 2)) & 3); };
  unsigned a_src[PA], w_src[PA];
#pragma unroll
  for (int i = 0; i < PA; ++i) {
    const int row = RPP * (wave + i * 8) + lane / C, ch = (lane % C) ^ swz(row);
    a_src[i] = (unsigned)(tm * 256 + row) * (unsigned)p.K + ch * 8;
    w_src[i] = (unsigned)(tn * 256 + row) * (unsigned)p.K + ch * 8;
    // blocked: plane image of (tile, slab) is PLANE bytes contiguous = the LDS image itself
    if (BLK >= 2) a_src[i] = (unsigned)tm * 256u * (unsigned)p.K + (wave + i * 8) * 512 + lane * 8;
    if (BLK >= 1) w_src[i] = (unsigned)tn * 256u * (unsigned)p.K + (wave + i * 8) * 512 + lane * 8;
  }
  auto issue = [&](int s, int stage, int which = 3) {   // stage-sized slab s (32 or 64 k); which: bit 0 = the A pieces, bit 1 = the W pieces
    char* base = smem + stage * STAGE;
    if (!SADDR && which != 3) {
      if (which & 1) {
#pragma unroll
        for (int i = 0; i < PA; ++i)
          __builtin_amdgcn_global_load_lds((glb_void_t*)(p.a + a_src[i] + s * (BLK >= 2 ? PLANE / 2 : RB / 2)), (lds_void_t*)(base + (wave + i * 8) * 1024), 16, 0, 0);
      }
      if (which & 2) {
#pragma unroll
        for (int i = 0; i < PA; ++i)
          __builtin_amdgcn_global_load_lds((glb_void_t*)(p.w + w_src[i] + s * (BLK >= 1 ? PLANE / 2 : RB / 2)), (lds_void_t*)(base + PLANE + (wave + i * 8) * 1024), 16, 0, 0);
      }
      return;
    }
    if (SADDR) {
      const char* ab = (const char*)(p.a + (long long)s * (BLK >= 2 ? PLANE / 2 : RB / 2));
      const char* wb = (const char*)(p.w + (long long)s * (BLK >= 1 ? PLANE / 2 : RB / 2));
      const unsigned lb = lds_offset_of(base);
#pragma unroll
      for (int i = 0; i < PA; ++i) dma16_sbase(ab, a_src[i] * 2, lb + (wave + i * 8) * 1024);
#pragma unroll
      for (int i = 0; i < PA; ++i) dma16_sbase(wb, w_src[i] * 2, lb + PLANE + (wave + i * 8) * 1024);
      return;
    }
#pragma unroll
    for (int i = 0; i < PA; ++i)
      __builtin_amdgcn_global_load_lds((glb_void_t*)(p.a + a_src[i] + s * (BLK >= 2 ? PLANE / 2 : RB / 2)), (lds_void_t*)(base + (wave + i * 8) * 1024), 16, 0, 0);
#pragma unroll
    for (int i = 0; i < PA; ++i)
      __builtin_amdgcn_global_load_lds((glb_void_t*)(p.w + w_src[i] + s * (BLK >= 1 ? PLANE / 2 : RB / 2)), (lds_void_t*)(base + PLANE + (wave + i * 8) * 1024), 16, 0, 0);
  };
  f32x4 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  f16x8 fa[TM], fw[TN];
  u32x4 keep = {0u, 0u, 0u, 0u};
  auto load_frags = [&](const char* base, int ks) {
    const int chunk = ks * 4 + lg;
#pragma unroll
    for (int mt = 0; mt < TM; ++mt) {
      const int row = wm * 128 + mt * 16 + li;
      fa[mt] = *reinterpret_cast<const f16x8*>(base + row * RB + ((chunk ^ swz(row)) << 4));
    }
#pragma unroll
    for (int nt = 0; nt < TN; ++nt) {
      const int row = wn * 64 + nt * 16 + li;
      fw[nt] = *reinterpret_cast<const f16x8*>(base + PLANE + row * RB + ((chunk ^ swz(row)) << 4));
    }
  };
  const int nk = p.K / 32;
  const bool g1 = __builtin_amdgcn_readfirstlane(wave) >= 4;
  if (HALF >= 2 && !BK64) {
    // HALF = 2 / 3 (round 5): the fragment reads are NOT waited for in front of the mid barrier — a load-phase wave arrives at the
    // barrier as soon as its reads are issued, and waits for them behind it (2: one lgkmcnt(0); 3: the compiler's own per-use waits) —
    // which needs one more stage of slack for the write-after-read hazard (the other group's reads of slab u - 1 may still be in
    // flight when this group's DMA goes out): the DMA runs TWO slabs ahead instead of three (slab u + 2 into slab u - 2's stage).
    issue(0, 0); issue(1, 1);
    wait_vmcnt<LPS>();
    __builtin_amdgcn_s_barrier();
    if (g1) __builtin_amdgcn_s_barrier();
    for (int u = 0; u < nk; ++u) {
      const bool more = u + 2 < nk;
      if (more) issue(u + 2, (u + 2) & 3);
      load_frags(smem + (u & 3) * STAGE, 0);
      if (more) wait_vmcnt<LPS>(); else wait_vmcnt<0>();
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();
      if (HALF == 2) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int mt = 0; mt < TM; ++mt)
#pragma unroll
        for (int nt = 0; nt < TN; ++nt) acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa[mt], fw[nt], acc[mt][nt], 0, 0, 0);
      __builtin_amdgcn_s_setprio(0);
      __builtin_amdgcn_s_barrier();
    }
    if (!g1) __builtin_amdgcn_s_barrier();
    float sum2 = 0.f;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) sum2 += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
    if (sum2 == 12345.678f) p.sink[tid] = sum2;
    __syncthreads();
    if (tid == 0) p.stamps[blockIdx.x * 2 + 1] = __builtin_amdgcn_s_memtime();
    return;
  }
  if (BK64) {
    issue(0, 0);
    wait_vmcnt<0>();
  } else {
#pragma unroll
    for (int s2 = 0; s2 < 3; ++s2) issue(s2, s2);
    wait_vmcnt<LPS * 2>();
  }
  __builtin_amdgcn_s_barrier();
  if (g1) __builtin_amdgcn_s_barrier();
  int cur = 0, nxt = 3;
  if (HALF == 1 && !BK64) {
    for (int u = 0; u < nk; ++u) {
      const bool more = u + 3 < nk;
      const char* base = smem + cur * STAGE;
      // ---- first half: m-tiles 0-3 x n-tiles 0-3
      if (more) issue(u + 3, nxt, 1);
#pragma unroll
      for (int mt = 0; mt < 4; ++mt) {
        const int row = wm * 128 + mt * 16 + li;
        fa[mt] = *reinterpret_cast<const f16x8*>(base + row * RB + ((lg ^ swz(row)) << 4));
      }
#pragma unroll
      for (int nt = 0; nt < TN; ++nt) {
        const int row = wn * 64 + nt * 16 + li;
        fw[nt] = *reinterpret_cast<const f16x8*>(base + PLANE + row * RB + ((lg ^ swz(row)) << 4));
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int nt = 0; nt < TN; ++nt) acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa[mt], fw[nt], acc[mt][nt], 0, 0, 0);
      __builtin_amdgcn_s_setprio(0);
      __builtin_amdgcn_s_barrier();
      // ---- second half: m-tiles 4-7 (the W fragments stay)
      if (more) issue(u + 3, nxt, 2);
#pragma unroll
      for (int mt = 0; mt < 4; ++mt) {
        const int row = wm * 128 + (4 + mt) * 16 + li;
        fa[mt] = *reinterpret_cast<const f16x8*>(base + row * RB + ((lg ^ swz(row)) << 4));
      }
      if (more) wait_vmcnt<LPS * 2>(); else wait_vmcnt<0>();
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int nt = 0; nt < TN; ++nt) acc[4 + mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa[mt], fw[nt], acc[4 + mt][nt], 0, 0, 0);
      __builtin_amdgcn_s_setprio(0);
      __builtin_amdgcn_s_barrier();
      cur = (cur + 1) & 3; nxt = (nxt + 1) & 3;
    }
  } else
  for (int u = 0; u < nk; ++u) {
    if (BK64) {
      const int pr = u >> 1, h = u & 1;
      if (h == 0 && (pr + 1) * 2 < nk) issue(pr + 1, (pr + 1) & 1);
      load_frags(smem + (pr & 1) * STAGE, h);
      if (h == 1) wait_vmcnt<0>();
    } else {
      const bool more = u + 3 < nk;
      if (more) issue(u + 3, nxt);
      load_frags(smem + cur * STAGE, 0);
      if (more) wait_vmcnt<LPS * 2>(); else wait_vmcnt<0>();
      cur = (cur + 1) & 3; nxt = (nxt + 1) & 3;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (DO_MATH) {
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int mt = 0; mt < TM; ++mt)
#pragma unroll
        for (int nt = 0; nt < TN; ++nt) acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa[mt], fw[nt], acc[mt][nt], 0, 0, 0);
      __builtin_amdgcn_s_setprio(0);
    } else {
#pragma unroll
      for (int mt = 0; mt < TM; ++mt) keep ^= __builtin_bit_cast(u32x4, fa[mt]);
#pragma unroll
      for (int nt = 0; nt < TN; ++nt) keep ^= __builtin_bit_cast(u32x4, fw[nt]);
    }
    __builtin_amdgcn_s_barrier();
  }
  if (!g1) __builtin_amdgcn_s_barrier();
  float sum = (float)(keep[0] ^ keep[1] ^ keep[2] ^ keep[3]);
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) sum += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
  if (sum == 12345.678f) p.sink[tid] = sum;
  if (p.tile_sum) atomicAdd(p.tile_sum + t, sum);
  __syncthreads();
  if (tid == 0) p.stamps[blockIdx.x * 2 + 1] = __builtin_amdgcn_s_memtime();
}

// template8_kernel (round 6): the guide's 256^2 BK = 64 8-phase template (cdna_hip_programming.md §5, "The 256^2
// 8-phase template"), built WHOLE from its description — its source (examples/gemm_256sq_8phase_bf16.cpp) is not in this image:
//   geometry   8 waves as 2 (M) x 4 (N), 128 x 64 outputs per wave; LDS = 2 K-tile buffers x 4 half-tiles x 16 KB (128 rows x 64 k,
//              128-BYTE rows) = 128 KB; a half-tile is what ONE C-quadrant phase needs first: A-half h = the h-th 64 rows of both
//              wave rows, B-half h = the h-th 32 columns of all four wave columns
//   K loop     8 phases per iteration = 2 K-tiles; a phase = [ds_read the register sub-tile: 4 (B) or 8 (A) or, when both, 4 x B then
//              8 x A] [stage ONE half-tile: 2 global_load_lds_dwordx4 per lane] [s_waitcnt] s_barrier [16 MFMAs = one C-quadrant x
//              K = 64, s_setprio 1] s_barrier; quadrants (0,0) (0,1) (1,1) (1,0): reads 12 / 4 / 8 / 4 (B-half 0 is read again for the
//              fourth quadrant: no second B register set)
//   staging    one half-tile per phase, each into its buffer ONE phase after that buffer's last read — which the description allows
//              when the reads were retired before the reading phase's first barrier, so lgkmcnt(0) sits in FRONT of it here
//              (round 5 measured where that wait sits at 1-1.5 %): tile t's phases stage B0(t+1), A0(t+2), B1(t+2), A1(t+2);
//              vmcnt ONCE per K-tile, in its fourth phase, counted: vmcnt(6) = the three newest half-tiles stay in flight (the
//              guide's formula), the K-tile read from the next phase on is complete; prologue 4 + 3 half-tiles, vmcnt(6)
//   stagger    the second wave row runs one barrier behind the first (`if (wr == 1) s_barrier`)
//   swizzle    16-byte chunk index ^ ((row >> 1) & 7) on the DMA's SOURCE side and on the ds_read address (this repo's conflict-
//              free choice for 128-byte rows; the guide's st_16x32 is 4-way by its own account)
// BLK: 0 = row-major operands; 1 = W pre-blocked (a half-tile's 16 KB contiguous: [tile n][K-tile][half]); 2 = A as well.
template <int BLK>
__global__ __launch_bounds__(512) void template8_kernel(const RingParams p) {
  constexpr int HT = 16384;
  constexpr int hA0 = 0, hA1 = 1, hB0 = 2, hB1 = 3;
  __shared__ __attribute__((aligned(1024))) char smem[2 * 4 * HT];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 2, wn = wave & 3, li = lane & 15, lg = lane >> 4;
  if (tid == 0) p.stamps[blockIdx.x * 2] = __builtin_amdgcn_s_memtime();
  const int nblk = p.tiles_m * p.tiles_n;
  const int L = blockIdx.x, xcd = L & 7, loc = L >> 3, q = nblk >> 3, r = nblk & 7;
  const int t = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
  const int tn = t % p.tiles_n, tm = t / p.tiles_n;
  const int nkt = p.K / 64;
  unsigned src[4][2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int pi = wave + 8 * i, lr = pi * 8 + (lane >> 3), lc = (lane & 7) ^ ((lr >> 1) & 7);
    const int wmr = lr >> 6, ra = lr & 63, wnr = lr >> 5, cb = lr & 31;
    src[hA0][i] = (unsigned)(tm * 256 + wmr * 128 + ra) * (unsigned)p.K + lc * 8;
    src[hA1][i] = (unsigned)(tm * 256 + wmr * 128 + 64 + ra) * (unsigned)p.K + lc * 8;
    src[hB0][i] = (unsigned)(tn * 256 + wnr * 64 + cb) * (unsigned)p.K + lc * 8;
    src[hB1][i] = (unsigned)(tn * 256 + wnr * 64 + 32 + cb) * (unsigned)p.K + lc * 8;
    if (BLK >= 1) {
      src[hB0][i] = (unsigned)(tn * nkt * 2 + 0) * 8192u + pi * 512 + lane * 8;
      src[hB1][i] = (unsigned)(tn * nkt * 2 + 1) * 8192u + pi * 512 + lane * 8;
    }
    if (BLK >= 2) {
      src[hA0][i] = (unsigned)(tm * nkt * 2 + 0) * 8192u + pi * 512 + lane * 8;
      src[hA1][i] = (unsigned)(tm * nkt * 2 + 1) * 8192u + pi * 512 + lane * 8;
    }
  }
  auto stage = [&](int kt, int d, int ht) {
    const bool isb = ht >= hB0;
    const f16* g = isb ? p.w : p.a;
    const unsigned kstep = (isb ? BLK >= 1 : BLK >= 2) ? 2u * 8192u : 64u;
    char* base = smem + (d * 4 + ht) * HT;
#pragma unroll
    for (int i = 0; i < 2; ++i)
      __builtin_amdgcn_global_load_lds((glb_void_t*)(g + src[ht][i] + (unsigned)kt * kstep), (lds_void_t*)(base + (wave + 8 * i) * 1024), 16, 0, 0);
  };
  f32x4 acc[8][4];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  f16x8 fa[4][2], fb[2][2];
  auto readA = [&](const char* base) {
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
      const int lr = wm * 64 + mt * 16 + li;
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) fa[mt][ks] = *reinterpret_cast<const f16x8*>(base + lr * 128 + (((ks * 4 + lg) ^ ((lr >> 1) & 7)) << 4));
    }
  };
  auto readB = [&](const char* base) {
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
      const int lr = wn * 32 + nt * 16 + li;
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) fb[nt][ks] = *reinterpret_cast<const f16x8*>(base + lr * 128 + (((ks * 4 + lg) ^ ((lr >> 1) & 7)) << 4));
    }
  };
  auto math = [&](int mi, int nj) {
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
          acc[mi * 4 + mt][nj * 2 + nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa[mt][ks], fb[nt][ks], acc[mi * 4 + mt][nj * 2 + nt], 0, 0, 0);
    __builtin_amdgcn_s_setprio(0);
  };
#define T8_MID()  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_sched_barrier(0); __builtin_amdgcn_s_barrier()
#define T8_END()  __builtin_amdgcn_sched_barrier(0); __builtin_amdgcn_s_barrier()
  // ---- prologue: K-tile 0 whole, K-tile 1 but its B-half 0
  stage(0, 0, hA0); stage(0, 0, hB0); stage(0, 0, hB1); stage(0, 0, hA1);
  if (nkt > 1) {
    stage(1, 1, hA0); stage(1, 1, hB1); stage(1, 1, hA1);
    wait_vmcnt<6>();
  } else {
    wait_vmcnt<0>();
  }
  __builtin_amdgcn_s_barrier();
  if (wm == 1) __builtin_amdgcn_s_barrier();
  auto ktile = [&](int kt, auto dc) {
    constexpr int d = decltype(dc)::value;
    const bool n1 = kt + 1 < nkt, n2 = kt + 2 < nkt;
    const char* buf = smem + d * 4 * HT;
    // phase 1: quadrant (0,0)
    readB(buf + hB0 * HT); __builtin_amdgcn_sched_barrier(0); readA(buf + hA0 * HT);
    if (n1) stage(kt + 1, d ^ 1, hB0);
    T8_MID(); math(0, 0); T8_END();
    // phase 2: (0,1)
    readB(buf + hB1 * HT);
    if (n2) stage(kt + 2, d, hA0);
    T8_MID(); math(0, 1); T8_END();
    // phase 3: (1,1)
    readA(buf + hA1 * HT);
    if (n2) stage(kt + 2, d, hB1);
    T8_MID(); math(1, 1); T8_END();
    // phase 4: (1,0); the one counted wait of the K-tile
    readB(buf + hB0 * HT);
    if (n2) { stage(kt + 2, d, hA1); wait_vmcnt<6>(); } else { wait_vmcnt<0>(); }
    T8_MID(); math(1, 0); T8_END();
  };
  for (int kt = 0; kt < nkt; kt += 2) {
    ktile(kt, std::integral_constant<int, 0>{});
    if (kt + 1 < nkt) ktile(kt + 1, std::integral_constant<int, 1>{});
  }
#undef T8_MID
#undef T8_END
  if (wm == 0) __builtin_amdgcn_s_barrier();
  float sum = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) sum += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
  if (sum == 12345.678f) p.sink[tid] = sum;
  if (p.tile_sum) atomicAdd(p.tile_sum + t, sum);
  __syncthreads();
  if (tid == 0) p.stamps[blockIdx.x * 2 + 1] = __builtin_amdgcn_s_memtime();
}

// phase2_kernel: the two-group schedule with TWO slabs per iteration (same four 32-KB stages): LOAD(j) reads the fragments
// of slab 2j, MATH(j) issues the 32 MFMAs of slab 2j, re-reads each A fragment from slab 2j+1 as soon as its row of MFMAs
// has been issued (B fragments of slab 2j+1 are read at the top of MATH into a second buffer), then the 32 MFMAs of slab
// 2j+1 — half as many barriers per k, MATH phases of 64 MFMAs.  DMA: in phase ph every wave issues its share of slab ph+2
// (the stage of slab ph-2, whose last readers ran in phase ph-1) and, before the barrier that ends the phase, confirms
// slab ph+1 (first read in phase ph+1) with a counted vmcnt.  Phases: group 0 LOAD(j) = 2j, MATH(j) = 2j+1; group 1 one later.
template <int BLK>
__global__ __launch_bounds__(512) void phase2_kernel(const RingParams p) {
  constexpr int RB = 64, C = 4, STAGE = 512 * RB, PLANE = 256 * RB, PA = 2, LPS = 4, TM = 8, TN = 4;
  __shared__ __attribute__((aligned(16))) char smem[4 * STAGE];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 2, wn = wave & 3, li = lane & 15, lg = lane >> 4;
  if (tid == 0) p.stamps[blockIdx.x * 2] = __builtin_amdgcn_s_memtime();
  const int nblk = p.tiles_m * p.tiles_n;
  const int L = blockIdx.x, xcd = L & 7, loc = L >> 3, q = nblk >> 3, r = nblk & 7;
  const int t = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
  const int tn = t % p.tiles_n, tm = t / p.tiles_n;
  unsigned a_src[PA], w_src[PA];
#pragma unroll
  for (int i = 0; i < PA; ++i) {
    const int row = 16 * (wave + i * 8) + lane / C, ch = (lane % C) ^ swz4(row);
    a_src[i] = (unsigned)(tm * 256 + row) * (unsigned)p.K + ch * 8;
    w_src[i] = (unsigned)(tn * 256 + row) * (unsigned)p.K + ch * 8;
    if (BLK >= 2) a_src[i] = (unsigned)tm * 256u * (unsigned)p.K + (wave + i * 8) * 512 + lane * 8;
    if (BLK >= 1) w_src[i] = (unsigned)tn * 256u * (unsigned)p.K + (wave + i * 8) * 512 + lane * 8;
  }
  auto issue = [&](int s2) {
    char* base = smem + (s2 & 3) * STAGE;
#pragma unroll
    for (int i = 0; i < PA; ++i)
      __builtin_amdgcn_global_load_lds((glb_void_t*)(p.a + a_src[i] + s2 * (BLK >= 2 ? PLANE / 2 : RB / 2)), (lds_void_t*)(base + (wave + i * 8) * 1024), 16, 0, 0);
#pragma unroll
    for (int i = 0; i < PA; ++i)
      __builtin_amdgcn_global_load_lds((glb_void_t*)(p.w + w_src[i] + s2 * (BLK >= 1 ? PLANE / 2 : RB / 2)), (lds_void_t*)(base + PLANE + (wave + i * 8) * 1024), 16, 0, 0);
  };
  f32x4 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  f16x8 fa[TM], fw0[TN], fw1[TN];
  int a_off[TM], w_off[TN];   // LDS byte offsets of this lane's fragments inside a stage
#pragma unroll
  for (int mt = 0; mt < TM; ++mt) { const int row = wm * 128 + mt * 16 + li; a_off[mt] = row * RB + ((lg ^ swz4(row)) << 4); }
#pragma unroll
  for (int nt = 0; nt < TN; ++nt) { const int row = wn * 64 + nt * 16 + li; w_off[nt] = PLANE + row * RB + ((lg ^ swz4(row)) << 4); }
  const int nk = p.K / 32, nj = nk / 2;
  const int g = __builtin_amdgcn_readfirstlane(wave) >= 4 ? 1 : 0;
  auto confirm = [&](int ph) {   // before the barrier that ends phase ph: my share of slab ph+1 has landed
    if (ph == 0) wait_vmcnt<2 * LPS>();
    else if (ph + 2 < nk) wait_vmcnt<LPS>();
    else wait_vmcnt<0>();
  };
#pragma unroll
  for (int s2 = 0; s2 < 4; ++s2) issue(s2);
  wait_vmcnt<2 * LPS>();
  __builtin_amdgcn_s_barrier();
  if (g) __builtin_amdgcn_s_barrier();
  for (int j = 0; j < nj; ++j) {
    const int phL = 2 * j + g, phM = phL + 1;
    // ---- LOAD(j)
    if (phL >= 2 && phL + 2 < nk) issue(phL + 2);
    const char* s0 = smem + ((2 * j) & 3) * STAGE;
    const char* s1 = smem + ((2 * j + 1) & 3) * STAGE;
#pragma unroll
    for (int mt = 0; mt < TM; ++mt) fa[mt] = *reinterpret_cast<const f16x8*>(s0 + a_off[mt]);
#pragma unroll
    for (int nt = 0; nt < TN; ++nt) fw0[nt] = *reinterpret_cast<const f16x8*>(s0 + w_off[nt]);
    confirm(phL);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    // ---- MATH(j)
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int mt = 0; mt < TM; ++mt) {
#pragma unroll
      for (int nt = 0; nt < TN; ++nt) acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa[mt], fw0[nt], acc[mt][nt], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      if (mt == 0) {
#pragma unroll
        for (int nt = 0; nt < TN; ++nt) fw1[nt] = *reinterpret_cast<const f16x8*>(s1 + w_off[nt]);
      }
      fa[mt] = *reinterpret_cast<const f16x8*>(s1 + a_off[mt]);   // slab 2j+1's fragment of this row, used 28 MFMAs later
      __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int mt = 0; mt < TM; ++mt) {
#pragma unroll
      for (int nt = 0; nt < TN; ++nt) acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa[mt], fw1[nt], acc[mt][nt], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      if (mt == 1 && phM >= 2 && phM + 2 < nk) issue(phM + 2);    // this phase's DMA share, in the MFMA shadow, after the last LDS read
    }
    __builtin_amdgcn_s_setprio(0);
    confirm(phM);
    __builtin_amdgcn_s_barrier();
  }
  if (!g) __builtin_amdgcn_s_barrier();
  float sum = 0.f;
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) sum += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
  if (sum == 12345.678f) p.sink[tid] = sum;
  __syncthreads();
  if (tid == 0) p.stamps[blockIdx.x * 2 + 1] = __builtin_amdgcn_s_memtime();
}

// store probe: a workgroup of NW waves writes a 256 x 256 tile of ELEM-byte outputs (ELEM = 2: 128 KB, 4: 256 KB) with
// 16-byte lane stores; SEG = bytes of one row that consecutive lanes of a wave cover (SEG/16 lanes per row).
struct StoreParams { char* c; long long ldc_bytes; int tiles_m, tiles_n; unsigned long long* stamps; };

template <int NW, int SEG, int ELEM>
__global__ __launch_bounds__(NW * 64) void store_kernel(const StoreParams p) {
  constexpr int ROWB = 256 * ELEM;              // bytes of one tile row
  constexpr int LPR = SEG / 16;                 // lanes per row segment
  constexpr int RPI = 64 / LPR;                 // rows per wave-instruction
  constexpr int SEGS = ROWB / SEG;              // segments per tile row
  static_assert(SEGS <= NW && NW % SEGS == 0, "a wave owns one column segment");
  constexpr int G = NW / SEGS;                  // waves stacked over the rows of one column segment
  constexpr int WROWS = 256 / G;                // rows per wave (a wave tile of WROWS x SEG bytes, like gemm16's wave tile)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (tid == 0) p.stamps[blockIdx.x * 2] = __builtin_amdgcn_s_memtime();
  const int nblk = p.tiles_m * p.tiles_n;
  const int L = blockIdx.x, xcd = L & 7, loc = L >> 3, q = nblk >> 3, r = nblk & 7;
  const int t = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
  const int tn = t % p.tiles_n, tm = t / p.tiles_n;
  char* base = p.c + (long long)tm * 256 * p.ldc_bytes + (long long)tn * ROWB;
  const int seg = wave % SEGS, row0 = (wave / SEGS) * WROWS + lane / LPR;
  const u32x4 v = {(unsigned)tid, 1u, 2u, 3u};
#pragma unroll 8
  for (int i = 0; i < WROWS / RPI; ++i)
    *reinterpret_cast<u32x4*>(base + (long long)(row0 + i * RPI) * p.ldc_bytes + seg * SEG + (lane % LPR) * 16) = v;
  __syncthreads();
  if (tid == 0) p.stamps[blockIdx.x * 2 + 1] = __builtin_amdgcn_s_memtime();
}

static double avg_cycles(const std::vector<unsigned long long>& st, int n) {
  double s = 0;
  for (int i = 0; i < n; ++i) s += (double)(st[2 * i + 1] - st[2 * i]);
  return s / n;
}

template <typename F>
static float time_launches(F&& launch, int warm, int reps) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i = 0; i < warm; ++i) launch();
  CK(hipEventRecord(e0, 0));
  for (int i = 0; i < reps; ++i) launch();
  CK(hipEventRecord(e1, 0));
  CK(hipEventSynchronize(e1));
  float ms = 0;
  CK(hipEventElapsedTime(&ms, e0, e1));
  CK(hipGetLastError());
  return ms * 1e3f / reps;   // us per launch
}

template <int NW, int NS, int MODE, int MF = 16, int LAYOUT = 0>
static void run_ring(const char* name, const f16* a, const f16* w, int M, int N, int K, unsigned long long* d_st, float* sink, int warm, int reps) {
  RingParams p{a, w, M, N, K, (M + 255) / 256, (N + 255) / 256, d_st, sink};
  const int nblk = p.tiles_m * p.tiles_n;
  float us = time_launches([&] { hipLaunchKernelGGL((ring_kernel<NW, NS, MODE, MF, LAYOUT>), dim3(nblk), dim3(NW * 64), 0, 0, p); }, warm, reps);
  std::vector<unsigned long long> st(2 * nblk);
  CK(hipMemcpy(st.data(), d_st, st.size() * 8, hipMemcpyDeviceToHost));
  const double cyc = avg_cycles(st, nblk), nk = K / 32.0;
  const double bytes = (double)nblk * nk * 32768.0, flops = 2.0 * p.tiles_m * 256.0 * p.tiles_n * 256.0 * K;
  printf("{\"probe\": \"ring\", \"name\": \"%s\", \"waves\": %d, \"stages\": %d, \"mode\": %d, \"mfma\": %d, \"layout\": %d, \"M\": %d, \"N\": %d, \"K\": %d, \"tiles\": %d, "
         "\"us\": %.1f, \"cycles_per_tile\": %.0f, \"cycles_per_slab\": %.0f, \"dma_B_per_clk_per_CU\": %.1f, \"dma_TBps_chip\": %.2f, \"mfma_TFLOPs\": %.0f}\n",
         name, NW, NS, MODE, MF, LAYOUT, M, N, K, nblk, us, cyc, cyc / nk, (MODE & 4) ? 0.0 : 32768.0 * nk / cyc, (MODE & 4) ? 0.0 : bytes / us * 1e-6,
         (MODE & 2) ? flops / us * 1e-6 : 0.0);
  fflush(stdout);
}

template <bool BK64, bool DO_MATH, int BLK = 0, bool SADDR = false, int HALF = 0>
static void run_phase(const char* name, const f16* a, const f16* w, int M, int N, int K, unsigned long long* d_st, float* sink, int warm, int reps) {
  RingParams p{a, w, M, N, K, M / 256, N / 256, d_st, sink};
  const int nblk = p.tiles_m * p.tiles_n;
  float us = time_launches([&] { hipLaunchKernelGGL((phase_kernel<BK64, DO_MATH, BLK, SADDR, HALF>), dim3(nblk), dim3(512), 0, 0, p); }, warm, reps);
  std::vector<unsigned long long> st(2 * nblk);
  CK(hipMemcpy(st.data(), d_st, st.size() * 8, hipMemcpyDeviceToHost));
  const double cyc = avg_cycles(st, nblk), nk = K / 32.0;
  const double bytes = (double)nblk * nk * 32768.0, flops = 2.0 * M * (double)N * K;
  printf("{\"probe\": \"phase\", \"name\": \"%s\", \"bk64\": %d, \"math\": %d, \"blocked\": %d, \"M\": %d, \"N\": %d, \"K\": %d, \"tiles\": %d, "
         "\"us\": %.1f, \"cycles_per_tile\": %.0f, \"cycles_per_slab\": %.0f, \"dma_B_per_clk_per_CU\": %.1f, \"dma_TBps_chip\": %.2f, \"mfma_TFLOPs\": %.0f}\n",
         name, (int)BK64, (int)DO_MATH, BLK, M, N, K, nblk, us, cyc, cyc / nk, 32768.0 * nk / cyc, bytes / us * 1e-6, DO_MATH ? flops / us * 1e-6 : 0.0);
  fflush(stdout);
}

template <int BLK>
static void run_phase2(const char* name, const f16* a, const f16* w, int M, int N, int K, unsigned long long* d_st, float* sink, int warm, int reps) {
  RingParams p{a, w, M, N, K, M / 256, N / 256, d_st, sink};
  const int nblk = p.tiles_m * p.tiles_n;
  float us = time_launches([&] { hipLaunchKernelGGL((phase2_kernel<BLK>), dim3(nblk), dim3(512), 0, 0, p); }, warm, reps);
  std::vector<unsigned long long> st(2 * nblk);
  CK(hipMemcpy(st.data(), d_st, st.size() * 8, hipMemcpyDeviceToHost));
  const double cyc = avg_cycles(st, nblk), nk = K / 32.0;
  const double bytes = (double)nblk * nk * 32768.0, flops = 2.0 * M * (double)N * K;
  printf("{\"probe\": \"phase2\", \"name\": \"%s\", \"blocked\": %d, \"M\": %d, \"N\": %d, \"K\": %d, \"tiles\": %d, "
         "\"us\": %.1f, \"cycles_per_tile\": %.0f, \"cycles_per_slab\": %.0f, \"dma_B_per_clk_per_CU\": %.1f, \"dma_TBps_chip\": %.2f, \"mfma_TFLOPs\": %.0f}\n",
         name, BLK, M, N, K, nblk, us, cyc, cyc / nk, 32768.0 * nk / cyc, bytes / us * 1e-6, flops / us * 1e-6);
  fflush(stdout);
}

template <int BLK>
static void run_template8(const char* name, const f16* a, const f16* w, int M, int N, int K, unsigned long long* d_st, float* sink, int warm, int reps) {
  RingParams p{a, w, M, N, K, M / 256, N / 256, d_st, sink, nullptr};
  const int nblk = p.tiles_m * p.tiles_n;
  float us = time_launches([&] { hipLaunchKernelGGL((template8_kernel<BLK>), dim3(nblk), dim3(512), 0, 0, p); }, warm, reps);
  std::vector<unsigned long long> st(2 * nblk);
  CK(hipMemcpy(st.data(), d_st, st.size() * 8, hipMemcpyDeviceToHost));
  const double cyc = avg_cycles(st, nblk), nk = K / 32.0;
  const double bytes = (double)nblk * nk * 32768.0, flops = 2.0 * M * (double)N * K;
  printf("{\"probe\": \"template8\", \"name\": \"%s\", \"blocked\": %d, \"M\": %d, \"N\": %d, \"K\": %d, \"tiles\": %d, "
         "\"us\": %.1f, \"cycles_per_tile\": %.0f, \"cycles_per_slab\": %.0f, \"dma_B_per_clk_per_CU\": %.1f, \"dma_TBps_chip\": %.2f, \"mfma_TFLOPs\": %.0f}\n",
         name, BLK, M, N, K, nblk, us, cyc, cyc / nk, 32768.0 * nk / cyc, bytes / us * 1e-6, flops / us * 1e-6);
  fflush(stdout);
}

// the template's tiles against the two-group loop's on the same row-major random operands: per-tile sums of all accumulators (the k
// order differs: fp32 rounding only).  A staging / swizzle / hazard mistake in the reconstruction shows here, not in the cycle counts.
static int check_template8(const f16* a, const f16* w, int M, int N, int K, unsigned long long* d_st, float* sink, int rounds) {
  const int nblk = (M / 256) * (N / 256);
  float *s_ref, *s_t8;
  CK(hipMalloc(&s_ref, nblk * 4)); CK(hipMalloc(&s_t8, nblk * 4));
  std::vector<float> h_ref(nblk), h_t8(nblk);
  int bad = 0; double worst = 0;
  for (int rd = 0; rd < rounds; ++rd) {
    CK(hipMemset(s_ref, 0, nblk * 4)); CK(hipMemset(s_t8, 0, nblk * 4));
    RingParams pr{a, w, M, N, K, M / 256, N / 256, d_st, sink, s_ref}, pt{a, w, M, N, K, M / 256, N / 256, d_st, sink, s_t8};
    hipLaunchKernelGGL((phase_kernel<false, true, 0, false, 0>), dim3(nblk), dim3(512), 0, 0, pr);
    hipLaunchKernelGGL((template8_kernel<0>), dim3(nblk), dim3(512), 0, 0, pt);
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(h_ref.data(), s_ref, nblk * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(h_t8.data(), s_t8, nblk * 4, hipMemcpyDeviceToHost));
    double scale = 0;
    for (int i = 0; i < nblk; ++i) scale = fmax(scale, fabs((double)h_ref[i]));
    for (int i = 0; i < nblk; ++i) {
      const double e = fabs((double)h_ref[i] - (double)h_t8[i]) / (scale + 1e-30);
      worst = fmax(worst, e);
      if (!(e < 2e-3)) ++bad;
    }
  }
  printf("{\"probe\": \"template8_check\", \"M\": %d, \"N\": %d, \"K\": %d, \"tiles\": %d, \"rounds\": %d, \"worst_tile_sum_rel_diff\": %.3g, \"bad_tiles\": %d}\n", M, N, K, nblk, rounds, worst, bad);
  fflush(stdout);
  CK(hipFree(s_ref)); CK(hipFree(s_t8));
  return bad;
}

template <int NW, int SEG, int ELEM>
static void run_store(char* c, int M, int N, unsigned long long* d_st, int warm, int reps) {
  StoreParams p{c, (long long)N * ELEM, M / 256, N / 256, d_st};
  const int nblk = p.tiles_m * p.tiles_n;
  float us = time_launches([&] { hipLaunchKernelGGL((store_kernel<NW, SEG, ELEM>), dim3(nblk), dim3(NW * 64), 0, 0, p); }, warm, reps);
  std::vector<unsigned long long> st(2 * nblk);
  CK(hipMemcpy(st.data(), d_st, st.size() * 8, hipMemcpyDeviceToHost));
  const double cyc = avg_cycles(st, nblk), tile_bytes = 65536.0 * ELEM;
  printf("{\"probe\": \"store\", \"waves\": %d, \"seg_bytes\": %d, \"elem_bytes\": %d, \"M\": %d, \"N\": %d, \"tiles\": %d, \"us\": %.1f, "
         "\"cycles_per_tile\": %.0f, \"B_per_clk_per_CU\": %.1f, \"TBps_chip\": %.2f}\n",
         NW, SEG, ELEM, M, N, nblk, us, cyc, tile_bytes / cyc, (double)nblk * tile_bytes / us * 1e-6);
  fflush(stdout);
}

// pseudo-random f16 operands (round 5): zero-filled planes toggle no bits and the chip then clocks 15-25 % higher than on real data
__global__ void fill_rand16(f16* p, size_t n, float scale, unsigned seed) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    unsigned h = (unsigned)i * 2654435761u + seed;
    h ^= h >> 15; h *= 2246822519u; h ^= h >> 13; h *= 3266489917u; h ^= h >> 16;
    p[i] = (f16)(((int)(h & 0xffff) - 32768) * (scale / 32768.0f));
  }
}

int main(int argc, char** argv) {
  const int warm = argc > 1 ? atoi(argv[1]) : 20, reps = argc > 2 ? atoi(argv[2]) : 40;
  const int M = 100864;          // CLIP-ViT-B/16 at 64 clips x 8 frames x 197 tokens (bench.py's visual leg)
  const int KMAX = 3072, NMAX = 3072;
  f16 *a, *w; char* c; unsigned long long* st; float* sink;
  CK(hipMalloc(&a, (size_t)M * KMAX * 2)); CK(hipMemset(a, 0, (size_t)M * KMAX * 2));
  CK(hipMalloc(&w, (size_t)NMAX * KMAX * 2)); CK(hipMemset(w, 0, (size_t)NMAX * KMAX * 2));
  CK(hipMalloc(&c, (size_t)M * NMAX * 4));
  CK(hipMalloc(&st, 16 * 8192)); CK(hipMalloc(&sink, 4096));
  hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
  printf("{\"device\": \"%s\", \"cus\": %d, \"clock_khz\": %d}\n", prop.gcnArchName, prop.multiProcessorCount, prop.clockRate);

#define PHASE_SET_B(BK64, MATH, BLK, NAME) \
  run_phase<BK64, MATH, BLK>(NAME, a, w, M, 2304, 768, st, sink, warm, reps); \
  run_phase<BK64, MATH, BLK>(NAME, a, w, M, 3072, 768, st, sink, warm, reps); \
  run_phase<BK64, MATH, BLK>(NAME, a, w, M, 768, 3072, st, sink, warm, reps);
  const char* set = argc > 3 ? argv[3] : "all";
  const bool all = !strcmp(set, "all");
  if (!strcmp(set, "phase8")) {
    // round 5: gemm16's loop (32 MFMAs per barrier phase) against the guide's phase grain (16 per phase) on the same LDS organisation,
    // W pre-blocked, PSEUDO-RANDOM operands (realistic power), interleaved twice; CLIP's shapes and a 4096-row plane with K = 4096
    hipLaunchKernelGGL(fill_rand16, dim3(4096), dim3(256), 0, 0, a, (size_t)M * KMAX, 2.0f, 1u);
    hipLaunchKernelGGL(fill_rand16, dim3(4096), dim3(256), 0, 0, w, (size_t)NMAX * KMAX, 0.05f, 2u);
    CK(hipDeviceSynchronize());
    for (int round = 0; round < 2; ++round) {
#define P8(MM, NN, KK) \
      run_phase<false, true, 1, false, 0>("two-group loop, 32 MFMAs per phase (gemm16 now), random operands", a, w, MM, NN, KK, st, sink, warm, reps); \
      run_phase<false, true, 1, false, 1>("two-group loop, 16 MFMAs per phase (the guide's 8-phase grain), random operands", a, w, MM, NN, KK, st, sink, warm, reps); \
      run_phase<false, true, 1, false, 2>("32 per phase, fragment reads waited BEHIND the mid barrier (lgkmcnt(0)), DMA two slabs ahead", a, w, MM, NN, KK, st, sink, warm, reps); \
      run_phase<false, true, 1, false, 3>("32 per phase, fragment reads waited behind the mid barrier by the compiler's per-use waits, DMA two slabs ahead", a, w, MM, NN, KK, st, sink, warm, reps);
      P8(M, 2304, 768) P8(M, 3072, 768) P8(M, 768, 3072) P8(4096, 3072, 3072) P8(8192, 3072, 3072)
    }
    return 0;
  }
  if (!strcmp(set, "template8")) {
    // round 6: the guide's 8-phase template, whole, against this repo's loop — pseudo-random operands, interleaved,
    // no epilogue in either; cycles per 32-deep slab of a 256^2 tile
    hipLaunchKernelGGL(fill_rand16, dim3(4096), dim3(256), 0, 0, a, (size_t)M * KMAX, 2.0f, 1u);
    hipLaunchKernelGGL(fill_rand16, dim3(4096), dim3(256), 0, 0, w, (size_t)NMAX * KMAX, 0.05f, 2u);
    CK(hipDeviceSynchronize());
    int bad = check_template8(a, w, 8192, 768, 768, st, sink, 3);
    bad += check_template8(a, w, 8192, 768, 3072, st, sink, 3);
    bad += check_template8(a, w, M, 2304, 768, st, sink, 2);
    bad += check_template8(a, w, 2048, 512, 64, st, sink, 1);      // one K-tile, two K-tiles, an odd count: the prologue / drain paths
    bad += check_template8(a, w, 2048, 512, 128, st, sink, 1);
    bad += check_template8(a, w, 2048, 512, 192, st, sink, 1);
    for (int round = 0; round < 2; ++round) {
#define T8(MM, NN, KK) \
      run_phase<false, true, 1, false, 0>("this repo's loop: two groups, 32 MFMAs per phase, 4 x 32-KB stages of 64-B rows, W pre-blocked", a, w, MM, NN, KK, st, sink, warm, reps); \
      run_phase<false, true, 0, false, 0>("this repo's loop, row-major operands", a, w, MM, NN, KK, st, sink, warm, reps); \
      run_template8<0>("the guide's 8-phase template (reconstruction), row-major operands", a, w, MM, NN, KK, st, sink, warm, reps); \
      run_template8<1>("the guide's 8-phase template (reconstruction), W pre-blocked per half-tile", a, w, MM, NN, KK, st, sink, warm, reps); \
      run_template8<2>("the guide's 8-phase template (reconstruction), A and W pre-blocked", a, w, MM, NN, KK, st, sink, warm, reps);
      T8(M, 2304, 768) T8(M, 3072, 768) T8(M, 768, 3072) T8(4096, 3072, 3072) T8(8192, 3072, 3072)
    }
    return bad ? 3 : 0;
  }
  // --- the K loop's data path; shapes: CLIP QKV (N=2304, K=768), fc2 (N=768, K=3072)
#define RING_SET(NW, NS, MODE, NAME) RING_SET_MF(NW, NS, MODE, 16, NAME)
#define RING_SET_MF(NW, NS, MODE, MF, NAME) RING_SET_L(NW, NS, MODE, MF, 0, NAME)
#define RING_SET_L(NW, NS, MODE, MF, LAYOUT, NAME) \
  run_ring<NW, NS, MODE, MF, LAYOUT>(NAME, a, w, M, 2304, 768, st, sink, warm, reps); \
  run_ring<NW, NS, MODE, MF, LAYOUT>(NAME, a, w, M, 768, 3072, st, sink, warm, reps);
  if (all || !strcmp(set, "phase")) {
#define PHASE_SET(BK64, MATH, NAME) \
  run_phase<BK64, MATH>(NAME, a, w, M, 2304, 768, st, sink, warm, reps); \
  run_phase<BK64, MATH>(NAME, a, w, M, 1536, 768, st, sink, warm, reps); \
  run_phase<BK64, MATH>(NAME, a, w, M, 3072, 768, st, sink, warm, reps); \
  run_phase<BK64, MATH>(NAME, a, w, M, 768, 3072, st, sink, warm, reps);
    PHASE_SET(false, false, "two-phase loop, 4 x 32 KB stages of 64-B rows, no mfma")
    PHASE_SET(true, false, "two-phase loop, 2 x 64 KB stages of 128-B rows, no mfma")
    PHASE_SET(false, true, "two-phase loop, 4 x 32 KB stages of 64-B rows (gemm16 today)")
    PHASE_SET(true, true, "two-phase loop, 2 x 64 KB stages of 128-B rows")
  }
  if (all || !strcmp(set, "saddr")) {
    run_phase<false, true, 1, false>("two-phase loop, W pre-blocked, flat 64-bit DMA addresses (gemm16 now)", a, w, M, 2304, 768, st, sink, warm, reps);
    run_phase<false, true, 1, true>("two-phase loop, W pre-blocked, SGPR base + 32-bit offsets", a, w, M, 2304, 768, st, sink, warm, reps);
    run_phase<false, true, 1, false>("two-phase loop, W pre-blocked, flat 64-bit DMA addresses (gemm16 now)", a, w, M, 768, 3072, st, sink, warm, reps);
    run_phase<false, true, 1, true>("two-phase loop, W pre-blocked, SGPR base + 32-bit offsets", a, w, M, 768, 3072, st, sink, warm, reps);
    run_phase<false, false, 1, true>("two-phase loop, W pre-blocked, SGPR base + 32-bit offsets, no mfma", a, w, M, 2304, 768, st, sink, warm, reps);
  }
  if (all || !strcmp(set, "phase2")) {
#define PHASE2_SET(BLK, NAME) \
  run_phase2<BLK>(NAME, a, w, M, 2304, 768, st, sink, warm, reps); \
  run_phase2<BLK>(NAME, a, w, M, 3072, 768, st, sink, warm, reps); \
  run_phase2<BLK>(NAME, a, w, M, 768, 3072, st, sink, warm, reps);
    PHASE_SET_B(false, true, 1, "two-phase loop, W pre-blocked (gemm16 now)")
    PHASE2_SET(0, "two slabs per iteration, row-major")
    PHASE2_SET(1, "two slabs per iteration, W pre-blocked")
    PHASE2_SET(2, "two slabs per iteration, A and W pre-blocked")
  }
  if (all || !strcmp(set, "blocked")) {
    PHASE_SET_B(false, true, 0, "two-phase loop (gemm16 today)")
    PHASE_SET_B(false, true, 1, "two-phase loop, W pre-blocked")
    PHASE_SET_B(false, true, 2, "two-phase loop, A and W pre-blocked")
    PHASE_SET_B(false, false, 1, "two-phase loop, W pre-blocked, no mfma")
    PHASE_SET_B(false, false, 2, "two-phase loop, A and W pre-blocked, no mfma")
    PHASE_SET_B(true, true, 1, "two-phase loop 2 x 64 KB, A 128-B rows, W pre-blocked")
  }
  if (all || !strcmp(set, "layout")) {
    RING_SET_L(8, 4, 0, 16, 0, "dma only, 16 rows x 64 B per piece (today)")
    RING_SET_L(8, 4, 0, 16, 1, "dma only, 64-B runs, slab pairs issued together")
    RING_SET_L(8, 4, 0, 16, 2, "dma only, 8 rows x 128 B per piece")
    RING_SET_L(8, 4, 0, 16, 3, "dma only, 4 rows x 256 B per piece")
    RING_SET_L(8, 4, 0, 16, 4, "dma only, 1 KiB contiguous per piece (pre-blocked operands)")
    RING_SET_L(8, 4, 3, 16, 0, "pipelined K loop 8 waves, 64-B runs (today)")
    RING_SET_L(8, 4, 3, 16, 1, "pipelined K loop 8 waves, slab pairs issued together")
    RING_SET_L(8, 4, 3, 16, 2, "pipelined K loop 8 waves, 128-B runs")
    RING_SET_L(8, 4, 3, 16, 3, "pipelined K loop 8 waves, 256-B runs")
    RING_SET_L(8, 4, 3, 16, 4, "pipelined K loop 8 waves, pre-blocked operands")
  }
  if (all || !strcmp(set, "ring")) {
  RING_SET(8, 4, 0, "dma only, 8 waves")
  RING_SET(4, 4, 0, "dma only, 4 waves")
  RING_SET(8, 4, 1, "dma + fragment reads, 8 waves (128x64 per wave)")
  RING_SET(4, 4, 1, "dma + fragment reads, 4 waves (128x128 per wave)")
  RING_SET(4, 4, 6, "mfma + fragment reads, no dma, 4 waves")
  RING_SET(8, 4, 6, "mfma + fragment reads, no dma, 8 waves")
  RING_SET(4, 4, 3, "pipelined K loop, 4 waves x 128x128 (1 wave/SIMD)")
  RING_SET(4, 3, 3, "pipelined K loop, 4 waves x 128x128, 3 stages")
  RING_SET(8, 4, 3, "pipelined K loop, 8 waves x 128x64 (single barrier)")
  RING_SET_MF(4, 4, 6, 32, "mfma 32x32x16 + fragment reads, no dma, 4 waves")
  RING_SET_MF(4, 4, 3, 32, "pipelined K loop, 4 waves x 128x128, 32x32x16 mfma")
  RING_SET_MF(8, 4, 3, 32, "pipelined K loop, 8 waves x 128x64, 32x32x16 mfma")

  }
  if (all || !strcmp(set, "store")) {
  // --- the epilogue's store path: f16 (QKV / fc1 outputs) and fp32 (residual stream) tiles
  run_store<8, 128, 2>(c, M, 2304, st, warm, reps);
  run_store<8, 256, 2>(c, M, 2304, st, warm, reps);
  run_store<8, 512, 2>(c, M, 2304, st, warm, reps);
  run_store<4, 128, 2>(c, M, 2304, st, warm, reps);
  run_store<4, 512, 2>(c, M, 2304, st, warm, reps);
  run_store<8, 256, 4>(c, M, 768, st, warm, reps);
  run_store<8, 512, 4>(c, M, 768, st, warm, reps);
  run_store<8, 1024, 4>(c, M, 768, st, warm, reps);
  }
  return 0;
}
