// gemm16_impl.h — the gemm16 kernel template and its launcher, shared by the translation units that instantiate it
// (gemm16_t3_f16.hip, gemm16_t3_bf16.hip, gemm16_small_f16.hip, gemm16_small_bf16.hip: compiled in parallel — as one file the
// kernel family takes ~6 minutes of hipcc) and by gemm16.hip (C ABI, packers, options).
#pragma once
// gemm16.hip — C = epilogue(A * W^T) on the gfx950 16x16x32 f16/bf16 MFMA, fp32 accumulate.
//
// This is the kernel >90 % of the encoder FLOPs go through (QKV / out-proj / FFN GEMMs of
// HF:hubert/modeling_hubert.py:262-368, HF:clip/modeling_clip.py:280-350,
// HF:roberta/modeling_roberta.py:186-399, the strided Conv1d stack :106-175 as implicit
// im2col, the ViT patch embedding and the grouped positional conv).
//
// Structure (one workgroup = WM x WN waves, wave tile (BM/WM) x (BN/WN), 16x16 MFMA tiles):
//   * A and W k-slabs are staged global -> VGPR -> LDS, double-buffered, one barrier per slab;
//     global loads for slab t+1 are issued before the MFMAs of slab t and written to the other
//     LDS buffer after them.
//   * LDS rows are BK 16-bit elements; the 16-byte chunk index is XOR-swizzled with
//     (row / rows_per_256B) so that the ds_read_b128 fragment reads (16 lanes = 16 different rows,
//     same k-chunk) hit 16 distinct 16-byte bank slots.
//   * "3-pass" mode keeps hi and lo planes of both operands in LDS and issues
//     acc += a_hi*w_hi + a_hi*w_lo + a_lo*w_hi  (fp32-grade result from 16-bit MFMAs);
//     "2-pass" mode splits only the weights (acc += a_hi*w_hi + a_hi*w_lo): weight rounding is the
//     same perturbation for every token, so it is what survives the utterance mean — removing it
//     costs one extra MFMA pass and no extra activation traffic.
//   * epilogue: the wave's accumulator tile goes through LDS once so that bias / activation /
//     residual / fp32 + 16-bit stores all run on 4 consecutive columns per lane with full-line
//     coalesced global accesses.
//   * workgroup -> tile mapping is XCD-aware (blocks b, b+8, b+16.. share an XCD / L2 and get
//     neighbouring tiles; bijective for any grid size).
#include "common.h"
#include <string.h>
#include <math.h>
#include <type_traits>

namespace mer {

struct Gemm16Params {
  int M, N, K;
  const void* a_hi; const void* a_lo; long long lda; int a_rpb; long long a_bstride;
  const void* w_hi; const void* w_lo; long long ldw;
  const void* w_mx;   // MX kernel: packed fp4 + E8M0 correction plane (mer_mx_pack)
  int w_blk;          // w_hi / w_lo are pre-blocked planes (mer_w_block_pack): [n-tile of 256][32-deep k-slab][the 16 KB LDS image]
  const float* bias; int act;
  int bias_T; long long bias_ld;   // bias_T > 0: `bias` is a table [ceil(M / bias_T), bias_ld], output row m takes row m / bias_T (mer_seq_bias)
  const float* residual; long long ldr;
  float* c32; long long ldc32;
  void* c16_hi; void* c16_lo; long long ldc16;
  int nb_inner; long long a_so, a_si, w_si, bias_si, c_so, c_si;
  int tiles_m, tiles_n;
  int vec_ok;  // N % 8 == 0 and all output/residual strides+offsets aligned for 16-byte accesses
  int epi32;      // fp32-only outputs (+ residual): 4 columns per lane, so that one store instruction covers whole 256-byte row runs
  int pk_epi;     // 16-bit-only outputs: activation on the accumulators, row pairs packed before the LDS transposition (half the LDS traffic)
  int dbg_skip;   // tuning experiments: 1 = skip the epilogue global stores, 2 = skip the whole epilogue
  int hm_T, hm_H;  // > 0: 16-bit output scattered head-major [N/(64*hm_H)][M/hm_T][hm_H][hm_T][64] (QKV for attention)
  unsigned long long* dbg;  // optional: 4 s_memtime stamps per workgroup (start, first slab ready, K loop done, end)
};

typedef __attribute__((address_space(3))) void lds_void_t;
typedef const __attribute__((address_space(1))) void glb_void_t;

// XOR term of the 16-byte chunk index for tile row `row` (C = chunks per LDS row).  Chosen so that every
// ds_read_b128 lane group of gfx950 ({0-3,12-15,20-27}, {4-11,16-19,28-31}, ...; 16 lanes = rows li of two
// neighbouring k-chunks) lands on 16 distinct 16-byte slots of the 256-byte bank row:
//   C == 8 (128-B rows, 2 rows per bank row): (row >> 1) & 7
//   C == 4 ( 64-B rows, 4 rows per bank row): (-(row >> 2)) & 3      [(row >> 2) & 3 is still 2-way]
template <int C>
__device__ __forceinline__ int swz_of(int row) {
  return C == 8 ? ((row >> 1) & 7) : ((-(row >> 2)) & 3);
}

// 16-byte streaming (non-temporal) global store: 16-bit GEMM outputs are not re-read by this launch, and allocating their lines in
// L2 / the Infinity Cache evicts the A / W lines the other CUs' K loops are re-reading (+19-20 % on the QKV / fc1 launches, DESIGN.md §3;
// sc1 / sc0 sc1 flavours measured slower).  fp32 outputs keep plain stores: their lines are re-read by the LayerNorm that follows.
__device__ __forceinline__ void gstore16_nt(void* ptr, u32x4 v) {
  asm volatile("global_store_dwordx4 %0, %1, off nt" :: "v"(ptr), "v"(v) : "memory");
}

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// MX mode (256x256 tile, 4x2 waves of 64x128, f16 only): acc += a_hi*w_hi on the f16 MFMA, and once per 128 k the weight
// residual w - w_hi is added through ONE v_mfma_scale_f32_16x16x128_f8f6f4 per accumulator: A = the f16 fragments the
// wave already holds, rounded to bf8 (e5m2: the f16 exponent range, so no scale search and no overflow) in registers,
// B = the residual as MX-fp4 (e2m1, one E8M0 scale per 32 k) packed offline by mer_mx_pack() in exactly the lane
// order the instruction wants.  The correction costs 1/4 of an f16 pass instead of a whole one (the residual only
// needs ~3 bits), and its plane is 1/4 of the bytes of the f16 lo plane.
//   operand layout (measured, scripts/probes/mx_probe*.py): 8-bit A: lane (i, g) byte p <-> k-slot 16g + p (p < 16),
//   64 + 16g + (p - 16) (p >= 16); fp4 B: lane (n, g) element j <-> k-slot 32g + j, low nibble first; the E8M0 scale
//   of (n, slots 32b..32b+31) is byte `opsel` of lane n + 16b's scale register.  A lane's A bytes 8s..8s+7 are its
//   fragment of slab s of the group (k = 32s + 8g + e), which fixes the k <-> slot permutation the packer applies.
//   Registers are the constraint (8 waves x 256): the bf8 A copy is 8 dwords per 16-row tile per group, so the wave
//   tile is 64 rows (32 dwords) and the B fragments are read from LDS only when they are used.
//   LDS: the fp4 plane of a 128-k group (16 column tiles x 1 KB + 1 KB of scales) lives in a double-buffered group
//   area behind the slab ring; every slab's DMA brings one quarter of it (+ the scales), one instruction per wave.
constexpr int MX_BLOCK = 5120;                 // global bytes per (256-column tile, slab): 4 column tiles + group scales
constexpr int MXG_BYTES = 16384 + 1024;        // LDS bytes of one group buffer
constexpr int MX_LDS = 2 * MXG_BYTES + 1024;   // two groups + a dump KB for the waves with nothing to fetch

// 16-byte-per-lane LDS-DMA with a uniform 64-bit base (SGPR pair) + 32-bit per-lane byte offset: one VGPR per address
// instead of the VGPR pair the builtin's flat form needs.  lds_off (uniform) goes to M0; lane l lands at lds_off + 16 l.
__device__ __forceinline__ void dma16_sbase(const void* sbase_any, unsigned voff, unsigned lds_off) {
  const unsigned long long pv = (unsigned long long)sbase_any;   // wave-uniform by construction: pin it to SGPRs
  const unsigned long long sbase = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(pv >> 32)) << 32) |
                                   (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)pv);
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1"
               :: "v"(voff), "s"(sbase), "s"(__builtin_amdgcn_readfirstlane(lds_off)) : "memory");
}
__device__ __forceinline__ unsigned lds_offset_of(const void* p) {
  return (unsigned)(unsigned long)(__attribute__((address_space(3))) const void*)p;
}

template <int OPS>
__device__ __forceinline__ f32x4 mx_mfma(i32x8 a, i32x8 b, f32x4 c, int sb) {
  return __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c, 1 /*A bf8*/, 4 /*B fp4*/, 0, 0x7f7f7f7f, OPS, sb);
}

// STAMP (tuning builds of the 8-wave kernels, mer_set_option("gemm_stamp", 1)): waves 0 and NW/2 accumulate, per K-loop
// iteration, the cycles spent in LOAD work / waiting at the mid barrier / MATH work / waiting at the end barrier, split
// into the MX-burst slabs and the others, into p.dbg[4 * nblk + (blk * 2 + group) * 8 ..].
template <typename T, int BM, int BN, int BK, int WM, int WN, int AP, int WP, bool GLDS, int NS, bool MX = false, bool STAMP = false>
__global__ __launch_bounds__(WM * WN * 64) void gemm16_kernel(const Gemm16Params p) {
  typedef typename T16<T>::v8 v8;
  constexpr int NT = WM * WN * 64;
  constexpr int C = BK / 8;            // 16-byte chunks per LDS row
  constexpr int RB = BK * 2;           // LDS row bytes
  constexpr int SM = BM / WM, SN = BN / WN;
  constexpr int TM = SM / 16, TN = SN / 16;
  constexpr int KS = BK / 32;          // MFMA k-steps per slab
  constexpr bool STAGGER = (WM * WN == 8) && KS == 1;   // two-group phase-shifted schedule (8-wave tiles)
  constexpr int CA = BM * C / NT;      // 16-byte chunks per thread per A plane
  constexpr int CW = BN * C / NT;
  constexpr int ROWS_PER_IT = NT / C;
  constexpr int A_PLANE = BM * RB, W_PLANE = BN * RB;
  constexpr int STAGE = AP * A_PLANE + WP * W_PLANE;
  static_assert(!MX || (GLDS && STAGGER && AP == 1 && WP == 1 && BN == 256 && WN == 2 && TN == 8 && std::is_same<T, f16>::value),
                "MX correction: 256-wide 8-wave (4x2) f16 tile only");
  constexpr int CLD = SN + 4;          // padded fp32 row of the per-wave C staging tile
  constexpr int EROWS = SN > 64 ? 16 : (SM > 64 ? 32 : SM);   // rows of the wave tile staged per epilogue chunk
  constexpr int CSTAGE = WM * WN * EROWS * CLD * 4;
  constexpr int RING = NS * STAGE + (MX ? MX_LDS : 0);
  constexpr int SMEM = RING > CSTAGE ? RING : CSTAGE;   // the C staging area of the epilogue aliases the (dead) ring
  static_assert(NS >= 2 && (GLDS || NS == 2), "register-staged loader is double-buffered only");
  static_assert(NT % C == 0 && (BM * C) % NT == 0 && (BN * C) % NT == 0, "bad tile/thread split");

  __shared__ __attribute__((aligned(16))) char smem[SMEM];

  const int tid = threadIdx.x;
  if (p.dbg && tid == 0) {
    const long long nb = (long long)gridDim.x * gridDim.y, bi = blockIdx.y * gridDim.x + blockIdx.x;
    p.dbg[bi * 4 + 0] = __builtin_amdgcn_s_memtime();
    // which CU ran this workgroup: HW_REG_XCC_ID[3:0] and the SE / SH / CU fields of HW_REG_HW_ID (timeline analysis, scripts/gemm_timeline.py)
    const unsigned hw = __builtin_amdgcn_s_getreg(4 | (0 << 6) | (31 << 11)), xcc = __builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11));
    p.dbg[nb * 20 + bi] = ((unsigned long long)xcc << 32) | hw;
  }
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int li = lane & 15, lg = lane >> 4;

  // ---- XCD-aware, bijective block -> tile map (L = linear workgroup / tile index) ----
  const int nblk = p.tiles_m * p.tiles_n;
  auto tile_of = [&](int L, int& tm, int& tn) __attribute__((always_inline)) {
    const int xcd = L & 7, loc = L >> 3;
    const int q = nblk >> 3, r = nblk & 7;
    const int swz = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
    tn = swz % p.tiles_n;
    tm = swz / p.tiles_n;
  };
  int tile_m, tile_n;
  tile_of(blockIdx.x, tile_m, tile_n);
  const int m0 = tile_m * BM, n0 = tile_n * BN;   // the tile being computed / written

  // ---- batch offsets ----
  const int z = blockIdx.y;
  const int zo = z / p.nb_inner, zi = z % p.nb_inner;
  const long long a_boff = (long long)zo * p.a_so + (long long)zi * p.a_si;
  const long long w_boff = (long long)zi * p.w_si;
  const long long c_boff = (long long)zo * p.c_so + (long long)zi * p.c_si;

  const T* a_pl[2] = {(const T*)p.a_hi + a_boff, AP == 2 ? (const T*)p.a_lo + a_boff : nullptr};
  const T* w_pl[2] = {(const T*)p.w_hi + w_boff, WP == 2 ? (const T*)p.w_lo + w_boff : nullptr};

  // ---- per-thread global load coordinates ----
  const int ld_ch = tid % C;
  const int ld_row0 = tid / C;
  long long a_off[CA], w_off[CW];
  long long a_src[CA], w_src[CW];   // GLDS: the same with the XOR swizzle folded into the source address (see below)
  auto setup_loads = [&](int m0_, int n0_) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < CA; ++i) {
      int m = m0_ + ld_row0 + i * ROWS_PER_IT;
      m = m < p.M ? m : p.M - 1;
      a_off[i] = (p.a_rpb > 0) ? (long long)(m / p.a_rpb) * p.a_bstride + (long long)(m % p.a_rpb) * p.lda
                               : (long long)m * p.lda;
      a_src[i] = a_off[i] + ((ld_ch ^ swz_of<C>(ld_row0 + i * ROWS_PER_IT)) << 3);
    }
#pragma unroll
    for (int i = 0; i < CW; ++i) {
      int n = n0_ + ld_row0 + i * ROWS_PER_IT;
      n = n < p.N ? n : p.N - 1;
      w_off[i] = (long long)n * p.ldw;
      w_src[i] = w_off[i] + ((ld_ch ^ swz_of<C>(ld_row0 + i * ROWS_PER_IT)) << 3);
      // pre-blocked W: the plane of (n-tile, k-slab) is the LDS image itself (swizzle applied by the packer), so a wave's
      // DMA piece is 1 KiB contiguous in memory — whole 128-byte lines instead of sixteen 64-byte row runs, which the
      // L2 -> LDS path moves ~1.7x faster (scripts/probes/ceiling_probe.hip); slab kt sits BN * BK elements after slab kt-1
      if (p.w_blk) w_src[i] = ((long long)(n0_ / BN) * (p.K / BK)) * (BN * BK) + (long long)(ld_row0 + i * ROWS_PER_IT) * BK + ld_ch * 8;
    }
  };
  setup_loads(m0, n0);

  u32x4 ra[AP][CA], rw[WP][CW];
  const u32x4 zero4 = {0u, 0u, 0u, 0u};

  auto gload = [&](int k0) __attribute__((always_inline)) {
    const int k = k0 + ld_ch * 8;
    const bool kin = k < p.K;
#pragma unroll
    for (int pl = 0; pl < AP; ++pl)
#pragma unroll
      for (int i = 0; i < CA; ++i)
        ra[pl][i] = kin ? *reinterpret_cast<const u32x4*>(a_pl[pl] + a_off[i] + k) : zero4;
#pragma unroll
    for (int pl = 0; pl < WP; ++pl)
#pragma unroll
      for (int i = 0; i < CW; ++i)
        rw[pl][i] = kin ? *reinterpret_cast<const u32x4*>(w_pl[pl] + w_off[i] + k) : zero4;
  };
  auto lds_store = [&](int stage) __attribute__((always_inline)) {
    char* base = smem + stage * STAGE;
#pragma unroll
    for (int pl = 0; pl < AP; ++pl)
#pragma unroll
      for (int i = 0; i < CA; ++i) {
        const int row = ld_row0 + i * ROWS_PER_IT;
        const int off = row * RB + ((ld_ch ^ swz_of<C>(row)) << 4);
        *reinterpret_cast<u32x4*>(base + pl * A_PLANE + off) = ra[pl][i];
      }
#pragma unroll
    for (int pl = 0; pl < WP; ++pl)
#pragma unroll
      for (int i = 0; i < CW; ++i) {
        const int row = ld_row0 + i * ROWS_PER_IT;
        const int off = row * RB + ((ld_ch ^ swz_of<C>(row)) << 4);
        *reinterpret_cast<u32x4*>(base + AP * A_PLANE + pl * W_PLANE + off) = rw[pl][i];
      }
  };

  // GLDS path: global -> LDS DMA (global_load_lds_dwordx4), no VGPR round trip and no ds_write.
  // One wave-instruction fills 1 KiB of LDS linearly (lane l -> base + 16*l), i.e. 64/C whole tile
  // rows; the XOR swizzle therefore moves to the SOURCE side: the lane that owns physical chunk c'
  // of row r fetches logical chunk c' ^ f(r).  The LDS image is identical to lds_store()'s.
  const int wave_row0 = (tid >> 6) * (64 / C);  // first tile row of this wave's 1 KiB piece
  // MX kernel: registers are scarce, so the DMA addresses are a uniform base (SGPRs, advanced by k) + a 32-bit per-lane
  // byte offset (the launcher checks that the planes are < 4 GB) instead of a 64-bit VGPR pair per load.
  unsigned a_o32[CA], w_o32[CW], mx_o32 = 0;
  if (MX) {
#pragma unroll
    for (int i = 0; i < CA; ++i) a_o32[i] = (unsigned)(a_src[i] * 2);
#pragma unroll
    for (int i = 0; i < CW; ++i) w_o32[i] = (unsigned)(w_src[i] * 2);
    mx_o32 = (unsigned)((tid < 320 ? tid : 256 + (tid & 63)) * 16);
  }
  const long long w_kmul = p.w_blk ? BN : 1;   // element distance of consecutive k (row-major) or of consecutive k-slabs / BK (pre-blocked)
  const char* mx_base = MX ? (const char*)p.w_mx + ((long long)tile_n * ((p.K + BK - 1) / BK)) * MX_BLOCK : nullptr;
  auto glds_issue = [&](int k0, int stage) __attribute__((always_inline)) {
    char* base = smem + stage * STAGE;
    if constexpr (MX) {
      const char* ab = (const char*)a_pl[0] + (long long)k0 * 2;
      const char* wb = (const char*)w_pl[0] + (long long)k0 * 2 * w_kmul;
      const unsigned lb = lds_offset_of(base) + wave_row0 * RB;
#pragma unroll
      for (int i = 0; i < CA; ++i) dma16_sbase(ab, a_o32[i], lb + i * ROWS_PER_IT * RB);
#pragma unroll
      for (int i = 0; i < CW; ++i) dma16_sbase(wb, w_o32[i], lb + A_PLANE + i * ROWS_PER_IT * RB);
      // this tile column's block of slab kt: waves 0-3 one column tile each, wave 4 the scales, 5-7 -> dump
      const int kt = k0 / BK, w = __builtin_amdgcn_readfirstlane(tid >> 6);
      const unsigned g = lds_offset_of(smem) + NS * STAGE + ((kt >> 2) & 1) * MXG_BYTES;
      const unsigned dst = w < 4 ? g + ((kt & 3) * 4 + w) * 1024 : (w == 4 ? g + 16384 : lds_offset_of(smem) + NS * STAGE + 2 * MXG_BYTES);
      dma16_sbase(mx_base + (long long)kt * MX_BLOCK, mx_o32, dst);
    } else {
#pragma unroll
      for (int pl = 0; pl < AP; ++pl)
#pragma unroll
        for (int i = 0; i < CA; ++i)
          __builtin_amdgcn_global_load_lds((glb_void_t*)(a_pl[pl] + a_src[i] + k0),
                                           (lds_void_t*)(base + pl * A_PLANE + (wave_row0 + i * ROWS_PER_IT) * RB), 16, 0, 0);
#pragma unroll
      for (int pl = 0; pl < WP; ++pl)
#pragma unroll
        for (int i = 0; i < CW; ++i)
          __builtin_amdgcn_global_load_lds((glb_void_t*)(w_pl[pl] + w_src[i] + k0 * w_kmul),
                                           (lds_void_t*)(base + AP * A_PLANE + pl * W_PLANE + (wave_row0 + i * ROWS_PER_IT) * RB), 16, 0, 0);
    }
  };

  f32x4 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int nk = (p.K + BK - 1) / BK;

  // fragment registers of one 32-deep k-step (ks) and the two halves of a k-step: LDS -> registers, registers -> MFMA
  v8 af[AP][TM], wf[WP][TN];
  auto load_frags = [&](const char* base, int ks) __attribute__((always_inline)) {
    const int chunk = ks * 4 + lg;
#pragma unroll
    for (int mt = 0; mt < TM; ++mt) {
      const int row = wm * SM + mt * 16 + li;
      const int off = row * RB + ((chunk ^ swz_of<C>(row)) << 4);
#pragma unroll
      for (int pl = 0; pl < AP; ++pl) af[pl][mt] = *reinterpret_cast<const v8*>(base + pl * A_PLANE + off);
    }
#pragma unroll
    for (int nt = 0; nt < TN; ++nt) {
      const int row = wn * SN + nt * 16 + li;
      const int off = row * RB + ((chunk ^ swz_of<C>(row)) << 4);
#pragma unroll
      for (int pl = 0; pl < WP; ++pl)
        wf[pl][nt] = *reinterpret_cast<const v8*>(base + AP * A_PLANE + pl * W_PLANE + off);
    }
  };
  auto math = [&]() __attribute__((always_inline)) {
    // one pass at a time over all TM x TN accumulators: back-to-back MFMAs never share an accumulator
    // (a dependent 16x16x32 MFMA would wait ~2 issue slots for its predecessor)
    if (AP == 2) {
#pragma unroll
      for (int mt = 0; mt < TM; ++mt)
#pragma unroll
        for (int nt = 0; nt < TN; ++nt) acc[mt][nt] = T16<T>::mfma(af[AP - 1][mt], wf[0][nt], acc[mt][nt]);
    }
    if (WP == 2) {
#pragma unroll
      for (int mt = 0; mt < TM; ++mt)
#pragma unroll
        for (int nt = 0; nt < TN; ++nt) acc[mt][nt] = T16<T>::mfma(af[0][mt], wf[WP - 1][nt], acc[mt][nt]);
    }
#pragma unroll
    for (int mt = 0; mt < TM; ++mt)
#pragma unroll
      for (int nt = 0; nt < TN; ++nt) acc[mt][nt] = T16<T>::mfma(af[0][mt], wf[0][nt], acc[mt][nt]);
  };
  auto compute = [&](const char* base) __attribute__((always_inline)) {
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      load_frags(base, ks);
      math();
    }
  };

  auto run_epilogue = [&]() __attribute__((always_inline)) {
  // ---- epilogue: accumulators -> per-wave LDS staging (EROWS rows at a time) -> 8 consecutive columns per lane, so
  // that 16-bit outputs leave as one 16-byte store per lane (the store tail is issue-bound: half the instructions,
  // half the time) and fp32 outputs / residuals as two.  The stage buffers are dead here (the K loop ended on a
  // barrier); each wave owns a private EROWS x CLD slice.
  float* ct = reinterpret_cast<float*>(smem) + wave * EROWS * CLD;
  constexpr int CPL = 8;                       // columns per lane
  constexpr int LANES_PER_ROW = SN / CPL;
  constexpr int ROWS_IT = 64 / LANES_PER_ROW;
  constexpr int NIT = EROWS / ROWS_IT;         // read-back iterations per chunk
  const int c8 = lane % LANES_PER_ROW;
  const int rsub = lane / LANES_PER_ROW;
  const int col = n0 + wn * SN + c8 * CPL;
  const bool col_ok = col < p.N;
  const float* bias = p.bias ? p.bias + (long long)zi * p.bias_si : nullptr;
  // the static bias of this lane's 8 columns, as two vector registers (a float[8] that is later re-read as vectors ends up in scratch)
  f32x4 bv0 = {0.f, 0.f, 0.f, 0.f}, bv1 = {0.f, 0.f, 0.f, 0.f};
  if (bias && p.bias_T == 0) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      bv0[j] = col + j < p.N ? bias[col + j] : 0.f;
      bv1[j] = col + 4 + j < p.N ? bias[col + 4 + j] : 0.f;
    }
  }
  const float* res = p.residual ? p.residual + c_boff : nullptr;
  float* c32 = p.c32 ? p.c32 + c_boff : nullptr;
  T* c16h = p.c16_hi ? (T*)p.c16_hi + c_boff : nullptr;
  T* c16l = p.c16_lo ? (T*)p.c16_lo + c_boff : nullptr;
  const bool vec = p.vec_ok && (col + CPL <= p.N);

  auto epilogue = [&](auto act_tag) __attribute__((always_inline)) {
    constexpr int ACT = decltype(act_tag)::value;
#pragma unroll
    for (int ch = 0; ch < SM / EROWS; ++ch) {
      // The staging slice is private to this wave and the LDS executes one wave's accesses in order, so the
      // write -> read-back -> overwrite sequence needs no s_barrier and, above all, no vmcnt drain: a
      // __syncthreads() here waits for the previous chunk's global stores (vmcnt counts stores on gfx950),
      // which serialised 4 store round trips per tile (~30k of ~100k cycles).  Compiler-level fences only.
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int mt = 0; mt < EROWS / 16; ++mt)
#pragma unroll
        for (int nt = 0; nt < TN; ++nt)
#pragma unroll
          for (int r = 0; r < 4; ++r) ct[(mt * 16 + lg * 4 + r) * CLD + nt * 16 + li] = acc[ch * (EROWS / 16) + mt][nt][r];
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      const int row0 = m0 + wm * SM + ch * EROWS + rsub;
      if (vec) {
        // all residual loads of the chunk first (rows of different iterations never overlap, so this is safe even
        // when the residual is updated in place) — otherwise every load would wait behind the previous stores
        f32x4 rr[NIT][2];
        if (res) {
#pragma unroll
          for (int it = 0; it < NIT; ++it) {
            const int row = row0 + it * ROWS_IT;
            const bool ok = row < p.M && col_ok;
            const float* rp = res + (long long)row * p.ldr + col;
            rr[it][0] = ok ? *reinterpret_cast<const f32x4*>(rp) : f32x4{0.f, 0.f, 0.f, 0.f};
            rr[it][1] = ok ? *reinterpret_cast<const f32x4*>(rp + 4) : f32x4{0.f, 0.f, 0.f, 0.f};
          }
        }
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
          const int lr = it * ROWS_IT + rsub;
          const int row = row0 + it * ROWS_IT;
          if (row >= p.M || !col_ok) continue;
          const f32x4 a0 = *reinterpret_cast<const f32x4*>(ct + lr * CLD + c8 * CPL);
          const f32x4 a1 = *reinterpret_cast<const f32x4*>(ct + lr * CLD + c8 * CPL + 4);
          float v[CPL];
          f32x4 b0 = bv0, b1 = bv1;
          if (p.bias_T > 0) {   // per-sequence correction table: this row's row of it (bias_ld % 4 == 0, 16-byte aligned: host-checked)
            const float* br = p.bias + (long long)(row / p.bias_T) * p.bias_ld + col;
            b0 = *reinterpret_cast<const f32x4*>(br);
            b1 = *reinterpret_cast<const f32x4*>(br + 4);
          }
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            v[j] = act_apply(a0[j] + b0[j], ACT);
            v[4 + j] = act_apply(a1[j] + b1[j], ACT);
          }
          if (res) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              v[j] += rr[it][0][j];
              v[4 + j] += rr[it][1][j];
            }
          }
          if (c32) {
            float* cp = c32 + (long long)row * p.ldc32 + col;
            *reinterpret_cast<f32x4*>(cp) = f32x4{v[0], v[1], v[2], v[3]};
            *reinterpret_cast<f32x4*>(cp + 4) = f32x4{v[4], v[5], v[6], v[7]};
          }
          if (c16h) {
            v8 h;
#pragma unroll
            for (int j = 0; j < CPL; ++j) h[j] = T16<T>::from_f32(v[j]);
            long long o16 = (long long)row * p.ldc16 + col;
            if (p.hm_T > 0) {  // head-major scatter: (which, b, h, t, d); 8 columns never straddle a 64-wide head
              const int dd = p.hm_H * 64, which = col / dd, hh2 = (col % dd) >> 6, d0 = col & 63;
              const int bb = row / p.hm_T, tt = row % p.hm_T;
              o16 = ((((long long)which * (p.M / p.hm_T) + bb) * p.hm_H + hh2) * p.hm_T + tt) * 64 + d0;
            }
            *reinterpret_cast<v8*>(c16h + o16) = h;
            if (c16l) {  // lo plane only when a 3-pass consumer needs it (3 extra VALU per element otherwise wasted)
              v8 l;
#pragma unroll
              for (int j = 0; j < CPL; ++j) l[j] = T16<T>::from_f32(v[j] - T16<T>::to_f32(h[j]));
              *reinterpret_cast<v8*>(c16l + o16) = l;
            }
          }
        }
      } else {
        for (int it = 0; it < NIT; ++it) {
          const int lr = it * ROWS_IT + rsub;
          const int row = row0 + it * ROWS_IT;
          if (row >= p.M || !col_ok) continue;
          for (int j = 0; j < CPL; ++j) {
            if (col + j >= p.N) break;
            const float bsc = !bias ? 0.f : (p.bias_T > 0 ? p.bias[(long long)(row / p.bias_T) * p.bias_ld + col + j] : bias[col + j]);
            float x = act_apply(ct[lr * CLD + c8 * CPL + j] + bsc, ACT);
            if (res) x += res[(long long)row * p.ldr + col + j];
            if (c32) c32[(long long)row * p.ldc32 + col + j] = x;
            if (c16h) {
              T hh, ll;
              split16<T>(x, hh, ll);
              c16h[(long long)row * p.ldc16 + col + j] = hh;
              if (c16l) c16l[(long long)row * p.ldc16 + col + j] = ll;
            }
          }
        }
      }
    }
  };
  // 16-bit-only outputs (QKV, fc1, conv GEMMs: the bulk of the tiles): bias + activation run on the accumulators (a lane's
  // column of every 16-wide column tile: TN bias registers), rows (2q, 2q+1) of a column are packed into one dword BEFORE the
  // transposition, so the staging tile holds row PAIRS: half the ds_write_b32 (the LDS pipe's slowest instruction, 64 B/clk/CU:
  // 4096 of its cycles per 256x256 fp32 tile) and half the read-back; a lane then owns 8 columns of a row pair and un-zips
  // them with v_perm_b32 into two 16-byte stores.
  auto epilogue_pk = [&](auto act_tag) __attribute__((always_inline)) {
    constexpr int ACT = decltype(act_tag)::value;
    constexpr int CLDP = SN + 8;                 // dwords per staged row pair; 2 * CLDP % 32 == 16: the two lg halves of a ds_write_b32 group never share a bank
    constexpr int PAIRS = EROWS / 2;
    constexpr int PAIRS_IT = 64 / LANES_PER_ROW;
    constexpr int NITP = PAIRS / PAIRS_IT;
    static_assert(PAIRS * CLDP * 4 <= EROWS * CLD * 4 && NITP >= 1, "packed staging must fit the fp32 staging slice");
    unsigned* cp = reinterpret_cast<unsigned*>(ct);
    float bcol[TN];
#pragma unroll
    for (int nt = 0; nt < TN; ++nt) {
      const int c = n0 + wn * SN + nt * 16 + li;
      bcol[nt] = (bias && c < p.N) ? bias[c] : 0.f;
    }
#pragma unroll
    for (int ch = 0; ch < SM / EROWS; ++ch) {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int mt = 0; mt < EROWS / 16; ++mt)
#pragma unroll
        for (int nt = 0; nt < TN; ++nt) {
          const f32x4 a = acc[ch * (EROWS / 16) + mt][nt];
          typename T16<T>::v4 h;
#pragma unroll
          for (int r = 0; r < 4; ++r) h[r] = T16<T>::from_f32(act_apply(a[r] + bcol[nt], ACT));
          const u32x2 pk = __builtin_bit_cast(u32x2, h);
          cp[(mt * 8 + lg * 2 + 0) * CLDP + nt * 16 + li] = pk[0];   // rows 4 lg + {0, 1}
          cp[(mt * 8 + lg * 2 + 1) * CLDP + nt * 16 + li] = pk[1];   // rows 4 lg + {2, 3}
        }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
      for (int it = 0; it < NITP; ++it) {
        const int pr = it * PAIRS_IT + rsub;
        const int row = m0 + wm * SM + ch * EROWS + 2 * pr;
        if (row >= p.M || !col_ok) continue;
        const u32x4 d0 = *reinterpret_cast<const u32x4*>(cp + pr * CLDP + c8 * CPL);
        const u32x4 d1 = *reinterpret_cast<const u32x4*>(cp + pr * CLDP + c8 * CPL + 4);
        u32x4 ev, od;   // even row: low halves of the 8 dwords, odd row: high halves
        ev[0] = __builtin_amdgcn_perm(d0[1], d0[0], 0x05040100u); od[0] = __builtin_amdgcn_perm(d0[1], d0[0], 0x07060302u);
        ev[1] = __builtin_amdgcn_perm(d0[3], d0[2], 0x05040100u); od[1] = __builtin_amdgcn_perm(d0[3], d0[2], 0x07060302u);
        ev[2] = __builtin_amdgcn_perm(d1[1], d1[0], 0x05040100u); od[2] = __builtin_amdgcn_perm(d1[1], d1[0], 0x07060302u);
        ev[3] = __builtin_amdgcn_perm(d1[3], d1[2], 0x05040100u); od[3] = __builtin_amdgcn_perm(d1[3], d1[2], 0x07060302u);
        if (c16h) {
          gstore16_nt(c16h + (long long)row * p.ldc16 + col, ev);
          if (row + 1 < p.M) gstore16_nt(c16h + (long long)(row + 1) * p.ldc16 + col, od);
        }
      }
    }
  };
  // fp32-only outputs (attention output projection, fc2: + residual): the generic path's 8 columns per lane make every store /
  // residual-load instruction touch 16 bytes out of every 32 (two instructions per 128-byte line).  Here a lane owns 4 columns:
  // 16 lanes cover a wave's 256-byte row run, one instruction = 4 whole rows, loads and stores are whole lines (and may stream).
  auto epilogue32 = [&](auto act_tag) __attribute__((always_inline)) {
    constexpr int ACT = decltype(act_tag)::value;
    constexpr int LPR = SN / 4, RIT = 64 / LPR, NIT4 = EROWS / RIT;
    const int c4 = lane % LPR, rs = lane / LPR;
    const int col4 = n0 + wn * SN + c4 * 4;
    const bool ok4 = col4 < p.N;
    f32x4 b4 = {0.f, 0.f, 0.f, 0.f};
    if (bias && ok4 && p.bias_T == 0) b4 = *reinterpret_cast<const f32x4*>(bias + col4);   // (vec: N % 8 == 0; bias is 16-byte aligned: checked on the host)
#pragma unroll
    for (int ch = 0; ch < SM / EROWS; ++ch) {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int mt = 0; mt < EROWS / 16; ++mt)
#pragma unroll
        for (int nt = 0; nt < TN; ++nt)
#pragma unroll
          for (int r = 0; r < 4; ++r) ct[(mt * 16 + lg * 4 + r) * CLD + nt * 16 + li] = acc[ch * (EROWS / 16) + mt][nt][r];
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      const int row0 = m0 + wm * SM + ch * EROWS + rs;
      f32x4 rr[NIT4];
      if (res) {
#pragma unroll
        for (int it = 0; it < NIT4; ++it) {
          const int row = row0 + it * RIT;
          rr[it] = f32x4{0.f, 0.f, 0.f, 0.f};
          if (row < p.M && ok4)
            rr[it] = *reinterpret_cast<const f32x4*>(res + (long long)row * p.ldr + col4);
        }
      }
#pragma unroll
      for (int it = 0; it < NIT4; ++it) {
        const int row = row0 + it * RIT;
        if (row < p.M && ok4) {
        const f32x4 a = *reinterpret_cast<const f32x4*>(ct + (it * RIT + rs) * CLD + c4 * 4);
        f32x4 v;
        f32x4 bb = b4;
        if (p.bias_T > 0) bb = *reinterpret_cast<const f32x4*>(p.bias + (long long)(row / p.bias_T) * p.bias_ld + col4);
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = act_apply(a[j] + bb[j], ACT);
        if (res) v += rr[it];
        if (c32) *reinterpret_cast<f32x4*>(c32 + (long long)row * p.ldc32 + col4) = v;
        }
      }
    }
  };
  if ((p.dbg_skip & 3) == 2) {
    if (p.dbg && tid == 0) p.dbg[(blockIdx.y * gridDim.x + blockIdx.x) * 4 + 3] = __builtin_amdgcn_s_memtime();
    return;
  }
  if (p.epi32) {   // host-checked: vector accesses, fp32 output only
    if ((p.dbg_skip & 3) == 1) c32 = nullptr;
    // (the host routes other activations to the generic path)
    if (p.act == MER_ACT_GELU) epilogue32(std::integral_constant<int, MER_ACT_GELU>{});
    else epilogue32(std::integral_constant<int, MER_ACT_NONE>{});
    return;
  }
  if ((p.dbg_skip & 3) == 1) { c32 = nullptr; c16h = nullptr; c16l = nullptr; }
  if (p.pk_epi) {   // host-checked: vector stores, 16-bit output only, row-major
    switch (p.act) {
      case MER_ACT_GELU: epilogue_pk(std::integral_constant<int, MER_ACT_GELU>{}); break;
      case MER_ACT_QUICK_GELU: epilogue_pk(std::integral_constant<int, MER_ACT_QUICK_GELU>{}); break;
      case MER_ACT_GELU_TANH: epilogue_pk(std::integral_constant<int, MER_ACT_GELU_TANH>{}); break;
      default: epilogue_pk(std::integral_constant<int, MER_ACT_NONE>{}); break;
    }
    return;
  }
  switch (p.act) {  // one specialised copy of the epilogue per activation: no per-element switch
    case MER_ACT_GELU: epilogue(std::integral_constant<int, MER_ACT_GELU>{}); break;
    case MER_ACT_QUICK_GELU: epilogue(std::integral_constant<int, MER_ACT_QUICK_GELU>{}); break;
    case MER_ACT_RELU: epilogue(std::integral_constant<int, MER_ACT_RELU>{}); break;
    case MER_ACT_GELU_TANH: epilogue(std::integral_constant<int, MER_ACT_GELU_TANH>{}); break;
    default: epilogue(std::integral_constant<int, MER_ACT_NONE>{}); break;
  }
  };

  if (GLDS && STAGGER) {
    // Two wave groups (waves [0, NW/2) and [NW/2, NW): one wave of each per SIMD) run the SAME loop one barrier
    // phase apart (group 1 takes one extra barrier up front, group 0 one at the end).  Each iteration is
    // LOAD(t) |bar| MATH(t) |bar|, so while one group issues its MFMAs the other pulls its fragments out of LDS:
    // the matrix pipe of every SIMD is fed by one wave at a time and never waits for an LDS read burst.
    //   phase:     2t          2t+1        2t+2
    //   group 0:   LOAD(t)     MATH(t)     LOAD(t+1)
    //   group 1:   MATH(t-1)   LOAD(t)     MATH(t)
    // DMA for slab t+D goes to stage (t-1) % NS at the top of a wave's iteration t: both groups' LOAD(t-1) ended
    // (lgkmcnt(0)) before the barrier that precedes it.  Each wave confirms its share of slab t+1 (counted
    // vmcnt, D-1 slabs stay in flight) before its mid-iteration barrier, i.e. at least one barrier before any
    // wave of either group reads that slab.
    constexpr int D = NS - 1;
    constexpr int LPS = AP * CA + WP * CW + (MX ? 1 : 0);
    const bool g1 = __builtin_amdgcn_readfirstlane(wave) >= (WM * WN / 2);
#pragma unroll
    for (int s = 0; s < D; ++s)
      if (s < nk) glds_issue(s * BK, s);
    if (nk >= D) wait_vmcnt<LPS*(D - 1)>();
    else wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();
    if (p.dbg && tid == 0) p.dbg[(blockIdx.y * gridDim.x + blockIdx.x) * 4 + 1] = __builtin_amdgcn_s_memtime();
    if (g1) __builtin_amdgcn_s_barrier();
    int cur = 0, nxt = D;
    // MX: bf8 copies of this wave's A fragments of the current 128-k group
    i64x4 aq[MX ? TM : 1];
#pragma unroll
    for (int i = 0; i < (MX ? TM : 1); ++i) aq[i] = i64x4{0, 0, 0, 0};
    unsigned long long acc_t[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    auto iter = [&](int kt) __attribute__((always_inline)) {
      unsigned long long t0 = 0, tL = 0, tB1 = 0, tM = 0;
      if (STAMP) t0 = __builtin_amdgcn_s_memtime();
      const bool more = kt + D < nk;
      if (more) glds_issue((kt + D) * BK, nxt);
      load_frags(smem + cur * STAGE, 0);
      i32x4 wcur, wnx1;   // deliberately not initialised (10 v_mov per slab): only read on the slabs that load them
      int sc0, sc1;
      const char* mg = smem + NS * STAGE + ((kt >> 2) & 1) * MXG_BYTES;
      const bool mx_slab = MX && (kt & 3) == 3;
      if (mx_slab) {   // last slab of a group: the first two fp4 fragments + the scales come in with the f16 fragments
        wcur = *reinterpret_cast<const i32x4*>(mg + (wn * TN) * 1024 + lane * 16);
        wnx1 = *reinterpret_cast<const i32x4*>(mg + (wn * TN + 1) * 1024 + lane * 16);
        sc0 = *reinterpret_cast<const int*>(mg + 16384 + (wn * 2) * 256 + lane * 4);
        sc1 = *reinterpret_cast<const int*>(mg + 16384 + (wn * 2 + 1) * 256 + lane * 4);
      }
      if (more) wait_vmcnt<LPS*(D - 1)>();
      else wait_vmcnt<0>();
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      if constexpr (MX) {
        // The bf8 copies are made HERE, at the end of the LOAD phase (the fragments have just landed), not in the MATH phase: LOAD is
        // the shorter phase of the MX loop (570 vs 745 cycles per plain slab), and every VALU instruction beside
        // the MFMAs of the MATH phase costs ~5 cycles of matrix-pipe time.
      // 8 f16 -> 8 bf8 (RNE) per 16-row tile; the group's window shifts by one slab (oldest slab in dwords 0-1) so that
      // one loop body serves all four slab positions.  Tried and measured worse or spilling: the 4x-unrolled loop with
      // static indices (21 spills), per-position uniform branches (the conversions get hoisted into temporaries + 16
      // copies), in-place inline-asm conversions (16 copies), a 64-bit window (the whole window gets copied).
      {
#pragma unroll
        for (int mt = 0; mt < TM; ++mt) {
          const v8 a = af[0][mt];
          i16x2 r0, r1;   // both halves get written below (deliberately uninitialised: no false dependency on the window)
          asm volatile("" : "=v"(r0), "=v"(r1));
          r0 = __builtin_amdgcn_cvt_scalef32_pk_bf8_f16(r0, f16x2{a[0], a[1]}, 1.0f, false);
          r0 = __builtin_amdgcn_cvt_scalef32_pk_bf8_f16(r0, f16x2{a[2], a[3]}, 1.0f, true);
          r1 = __builtin_amdgcn_cvt_scalef32_pk_bf8_f16(r1, f16x2{a[4], a[5]}, 1.0f, false);
          r1 = __builtin_amdgcn_cvt_scalef32_pk_bf8_f16(r1, f16x2{a[6], a[7]}, 1.0f, true);
          // 64-bit window elements: the shift is three v_mov_b64 per tile instead of six v_mov_b32
          aq[mt] = i64x4{aq[mt][1], aq[mt][2], aq[mt][3],
                         __builtin_bit_cast(long long, i32x2{__builtin_bit_cast(int, r0), __builtin_bit_cast(int, r1)})};
        }
      }
        __builtin_amdgcn_sched_barrier(0);   // keep the conversions on this side of the barrier
      }
      if (STAMP) tL = __builtin_amdgcn_s_memtime();
      __builtin_amdgcn_s_barrier();
      if (STAMP) tB1 = __builtin_amdgcn_s_memtime();
      __builtin_amdgcn_s_setprio(1);
      math();
      if constexpr (MX) {
        if (mx_slab) {   // the group's fp4 residual fragments straight from the group buffer, one column tile at a time
          // two column tiles ahead (the LDS is busy with the other wave group's fragment reads: one tile ahead stalled),
          // no further: the compiler would otherwise pull all 8 reads to the top (32 live registers the kernel lacks)
#pragma unroll
          for (int nt = 0; nt < TN; ++nt) {
            asm volatile("" ::: "memory");
            i32x4 wnx2 = wnx1;
            if (nt + 2 < TN) wnx2 = *reinterpret_cast<const i32x4*>(mg + (wn * TN + nt + 2) * 1024 + lane * 16);
            __builtin_amdgcn_sched_barrier(0);   // keep the read ahead of this tile's MFMAs (it was sunk below them)
            const i32x8 wb = {wcur[0], wcur[1], wcur[2], wcur[3], 0, 0, 0, 0};
            const int sc = nt < 4 ? sc0 : sc1;
#pragma unroll
            for (int mt = 0; mt < TM; ++mt) {
              switch (nt & 3) {
                case 0: acc[mt][nt] = mx_mfma<0>(__builtin_bit_cast(i32x8, aq[mt]), wb, acc[mt][nt], sc); break;
                case 1: acc[mt][nt] = mx_mfma<1>(__builtin_bit_cast(i32x8, aq[mt]), wb, acc[mt][nt], sc); break;
                case 2: acc[mt][nt] = mx_mfma<2>(__builtin_bit_cast(i32x8, aq[mt]), wb, acc[mt][nt], sc); break;
                default: acc[mt][nt] = mx_mfma<3>(__builtin_bit_cast(i32x8, aq[mt]), wb, acc[mt][nt], sc); break;
              }
            }
            wcur = wnx1;
            wnx1 = wnx2;
          }
        }
      }
      __builtin_amdgcn_s_setprio(0);
      if (STAMP) tM = __builtin_amdgcn_s_memtime();
      __builtin_amdgcn_s_barrier();
      if (STAMP) {
        const unsigned long long tB2 = __builtin_amdgcn_s_memtime();
        const bool mxs = MX && (kt & 3) == 3;   // (static indices only: a dynamically indexed array would live in scratch)
        const unsigned long long d0 = tL - t0, d1 = tB1 - tL, d2 = tM - tB1, d3 = tB2 - tM;
        acc_t[0] += mxs ? 0 : d0; acc_t[1] += mxs ? 0 : d1; acc_t[2] += mxs ? 0 : d2; acc_t[3] += mxs ? 0 : d3;
        acc_t[4] += mxs ? d0 : 0; acc_t[5] += mxs ? d1 : 0; acc_t[6] += mxs ? d2 : 0; acc_t[7] += mxs ? d3 : 0;
      }
      cur = cur + 1 == NS ? 0 : cur + 1;
      nxt = nxt + 1 == NS ? 0 : nxt + 1;
    };
    for (int kt = 0; kt < nk; ++kt) iter(kt);   // MX: K % 128 == 0 (checked by the launcher)
    if (STAMP && p.dbg && lane == 0 && (wave == 0 || wave == WM * WN / 2)) {
      unsigned long long* d = p.dbg + 4ll * gridDim.x * gridDim.y + ((blockIdx.y * gridDim.x + blockIdx.x) * 2 + (wave != 0)) * 8;
#pragma unroll
      for (int i = 0; i < 8; ++i) d[i] = acc_t[i];
    }
    if (!g1) __builtin_amdgcn_s_barrier();
    __syncthreads();
    if (p.dbg && tid == 0) p.dbg[(blockIdx.y * gridDim.x + blockIdx.x) * 4 + 2] = __builtin_amdgcn_s_memtime();
  } else
  if (GLDS) {
    // NS-stage LDS ring, LDS-DMA prefetch distance D = NS-1 slabs, counted vmcnt: at the end of iteration
    // t only slab t+1 has to have landed, the newer D-1 slabs stay in flight ACROSS the barrier (raw
    // s_barrier: __syncthreads() would drain vmcnt(0) because an LDS-DMA is a pending LDS write).
    // WAR: iteration t refills stage (t+D) % NS == (t-1) % NS, whose readers all passed barrier t-1.
    constexpr int D = NS - 1;
    constexpr int LPS = AP * CA + WP * CW;  // LDS-DMA instructions per wave per slab
#pragma unroll
    for (int s = 0; s < D; ++s)
      if (s < nk) glds_issue(s * BK, s);
    if (nk >= D) wait_vmcnt<LPS*(D - 1)>();
    else wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();
    int cur = 0, nxt = D;  // stage holding slab t / stage to refill with slab t+D
    for (int kt = 0; kt < nk; ++kt) {
      const bool more = kt + D < nk;
      if (more) glds_issue((kt + D) * BK, nxt);
      compute(smem + cur * STAGE);
      if (more) wait_vmcnt<LPS*(D - 1)>();
      else wait_vmcnt<0>();
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      cur = cur + 1 == NS ? 0 : cur + 1;
      nxt = nxt + 1 == NS ? 0 : nxt + 1;
    }
  } else {
    gload(0);
    lds_store(0);
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
      const int cur = kt & 1;
      if (kt + 1 < nk) gload((kt + 1) * BK);
      compute(smem + cur * STAGE);
      if (kt + 1 < nk) lds_store(cur ^ 1);
      __syncthreads();
    }
  }

  run_epilogue();
  if (p.dbg && tid == 0) p.dbg[(blockIdx.y * gridDim.x + blockIdx.x) * 4 + 3] = __builtin_amdgcn_s_memtime();
}

constexpr int MX_NS = 3;   // 3 x 32 KB slab stages + 35 KB of MX group buffers
// tuning switches (defined in gemm16.hip, set through mer_set_option)
extern int g_gemm_skip, g_gemm_stamp, g_gemm_glds, g_gemm_generic_epi, g_gemm_persist;
extern unsigned long long* g_gemm_dbg;

template <typename T, int BM, int BN, int BK, int WM, int WN, int AP, int WP, int NS, bool MX = false>
static int launch(const Gemm16Params& p0, int nbatch, hipStream_t st) {
  Gemm16Params p = p0;
  p.tiles_m = (int)cdiv(p.M, BM);
  p.tiles_n = (int)cdiv(p.N, BN);
  dim3 grid(p.tiles_m * p.tiles_n, nbatch, 1), block(WM * WN * 64, 1, 1);
  // algorithmic work of this launch: 2*M*N*K flops (one pass, whatever AP/WP execute), A + W read once,
  // outputs (+ residual) touched once
  const double mn = (double)p.M * p.N * nbatch;
  ProfScope prof(MX ? "gemm16_mx" : (AP == 2 ? "gemm16_x3" : (WP == 2 ? "gemm16_w2" : "gemm16")), 2.0 * mn * p.K,
                 2.0 * AP * nbatch * (double)p.M * p.K + (2.0 * WP + (MX ? 0.25 : 0.0)) * (double)p.N * p.K * (nbatch / p.nb_inner > 0 ? p.nb_inner : 1) +
                     mn * ((p.c32 ? 4 : 0) + (p.c16_hi ? 2 : 0) + (p.c16_lo ? 2 : 0) + (p.residual ? 4 : 0)),
                 st);
  if constexpr (MX) {
    if (g_gemm_stamp) hipLaunchKernelGGL((gemm16_kernel<T, BM, BN, BK, WM, WN, AP, WP, true, NS, true, true>), grid, block, 0, st, p);
    else hipLaunchKernelGGL((gemm16_kernel<T, BM, BN, BK, WM, WN, AP, WP, true, NS, true>), grid, block, 0, st, p);
    return check_launch("gemm16_mx");
  } else {
    if constexpr (WM * WN == 8 && AP == 1 && std::is_same<T, f16>::value) {
      if (g_gemm_stamp && g_gemm_glds == 1 && p.K % BK == 0) {
        hipLaunchKernelGGL((gemm16_kernel<T, BM, BN, BK, WM, WN, AP, WP, true, NS, false, true>), grid, block, 0, st, p);
        return check_launch("gemm16");
      }
    }
    if (g_gemm_glds == 2 && p.K % BK == 0)  // A/B: LDS-DMA loader, plain double buffering
      hipLaunchKernelGGL((gemm16_kernel<T, BM, BN, BK, WM, WN, AP, WP, true, 2>), grid, block, 0, st, p);
    else if (g_gemm_glds && p.K % BK == 0)
      hipLaunchKernelGGL((gemm16_kernel<T, BM, BN, BK, WM, WN, AP, WP, true, NS>), grid, block, 0, st, p);
    else
      hipLaunchKernelGGL((gemm16_kernel<T, BM, BN, BK, WM, WN, AP, WP, false, 2>), grid, block, 0, st, p);
    return check_launch("gemm16");
  }
}

// per-(tile class, dtype) entry points, one translation unit each
template <typename T> int dispatch_t3(const Gemm16Params& p, int nbatch, int passes, hipStream_t st);      // 256x256 8-wave kernels + MX
template <typename T> int dispatch_small(const Gemm16Params& p, int nbatch, int passes, int tile, hipStream_t st);   // 128x128 / 128x64

}  // namespace mer
