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
#include <type_traits>

namespace mer {

struct Gemm16Params {
  int M, N, K;
  const void* a_hi; const void* a_lo; long long lda; int a_rpb; long long a_bstride;
  const void* w_hi; const void* w_lo; long long ldw;
  const float* bias; int act;
  const float* residual; long long ldr;
  float* c32; long long ldc32;
  void* c16_hi; void* c16_lo; long long ldc16;
  int nb_inner; long long a_so, a_si, w_si, bias_si, c_so, c_si;
  int tiles_m, tiles_n;
  int vec_ok;  // N % 8 == 0 and all output/residual strides+offsets aligned for 16-byte accesses
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

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

template <typename T, int BM, int BN, int BK, int WM, int WN, int AP, int WP, bool GLDS, int NS>
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
  constexpr int CLD = SN + 4;          // padded fp32 row of the per-wave C staging tile
  constexpr int EROWS = SM > 64 ? 32 : SM;   // rows of the wave tile staged per epilogue chunk
  constexpr int CSTAGE = WM * WN * EROWS * CLD * 4;
  constexpr int SMEM = (NS * STAGE > CSTAGE) ? NS * STAGE : CSTAGE;
  static_assert(NS >= 2 && (GLDS || NS == 2), "register-staged loader is double-buffered only");
  static_assert(NT % C == 0 && (BM * C) % NT == 0 && (BN * C) % NT == 0, "bad tile/thread split");

  __shared__ __attribute__((aligned(16))) char smem[SMEM];

  const int tid = threadIdx.x;
  if (p.dbg && tid == 0) p.dbg[(blockIdx.y * gridDim.x + blockIdx.x) * 4 + 0] = __builtin_amdgcn_s_memtime();
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int li = lane & 15, lg = lane >> 4;

  // ---- XCD-aware, bijective block -> tile map ----
  const int nblk = p.tiles_m * p.tiles_n;
  int tile_m, tile_n;
  {
    const int bid = blockIdx.x;
    const int xcd = bid & 7, loc = bid >> 3;
    const int q = nblk >> 3, r = nblk & 7;
    const int swz = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
    tile_n = swz % p.tiles_n;
    tile_m = swz / p.tiles_n;
  }
  const int m0 = tile_m * BM, n0 = tile_n * BN;

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
#pragma unroll
  for (int i = 0; i < CA; ++i) {
    int m = m0 + ld_row0 + i * ROWS_PER_IT;
    m = m < p.M ? m : p.M - 1;
    a_off[i] = (p.a_rpb > 0) ? (long long)(m / p.a_rpb) * p.a_bstride + (long long)(m % p.a_rpb) * p.lda
                             : (long long)m * p.lda;
  }
#pragma unroll
  for (int i = 0; i < CW; ++i) {
    int n = n0 + ld_row0 + i * ROWS_PER_IT;
    n = n < p.N ? n : p.N - 1;
    w_off[i] = (long long)n * p.ldw;
  }

  u32x4 ra[AP][CA], rw[WP][CW];
  const u32x4 zero4 = {0u, 0u, 0u, 0u};

  auto gload = [&](int k0) {
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
  auto lds_store = [&](int stage) {
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
  long long a_src[CA], w_src[CW];
  if (GLDS) {
#pragma unroll
    for (int i = 0; i < CA; ++i) {
      const int row = ld_row0 + i * ROWS_PER_IT;
      a_src[i] = a_off[i] + ((ld_ch ^ swz_of<C>(row)) << 3);
    }
#pragma unroll
    for (int i = 0; i < CW; ++i) {
      const int row = ld_row0 + i * ROWS_PER_IT;
      w_src[i] = w_off[i] + ((ld_ch ^ swz_of<C>(row)) << 3);
    }
  }
  const int wave_row0 = (tid >> 6) * (64 / C);  // first tile row of this wave's 1 KiB piece
  auto glds_issue = [&](int k0, int stage) {
    char* base = smem + stage * STAGE;
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
        __builtin_amdgcn_global_load_lds((glb_void_t*)(w_pl[pl] + w_src[i] + k0),
                                         (lds_void_t*)(base + AP * A_PLANE + pl * W_PLANE + (wave_row0 + i * ROWS_PER_IT) * RB), 16, 0, 0);
  };

  f32x4 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int nk = (p.K + BK - 1) / BK;

  // fragment registers of one 32-deep k-step (ks) and the two halves of a k-step: LDS -> registers, registers -> MFMA
  v8 af[AP][TM], wf[WP][TN];
  auto load_frags = [&](const char* base, int ks) {
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
  auto math = [&]() {
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
  auto compute = [&](const char* base) {
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      load_frags(base, ks);
      math();
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
    constexpr int LPS = AP * CA + WP * CW;
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
    for (int kt = 0; kt < nk; ++kt) {
      const bool more = kt + D < nk;
      if (more) glds_issue((kt + D) * BK, nxt);
      load_frags(smem + cur * STAGE, 0);
      if (more) wait_vmcnt<LPS*(D - 1)>();
      else wait_vmcnt<0>();
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_s_setprio(1);
      math();
      __builtin_amdgcn_s_setprio(0);
      __builtin_amdgcn_s_barrier();
      cur = cur + 1 == NS ? 0 : cur + 1;
      nxt = nxt + 1 == NS ? 0 : nxt + 1;
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
  float bv[CPL];
#pragma unroll
  for (int j = 0; j < CPL; ++j) bv[j] = (bias && col + j < p.N) ? bias[col + j] : 0.f;
  const float* res = p.residual ? p.residual + c_boff : nullptr;
  float* c32 = p.c32 ? p.c32 + c_boff : nullptr;
  T* c16h = p.c16_hi ? (T*)p.c16_hi + c_boff : nullptr;
  T* c16l = p.c16_lo ? (T*)p.c16_lo + c_boff : nullptr;
  const bool vec = p.vec_ok && (col + CPL <= p.N);

  auto epilogue = [&](auto act_tag) {
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
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            v[j] = act_apply(a0[j] + bv[j], ACT);
            v[4 + j] = act_apply(a1[j] + bv[4 + j], ACT);
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
            float x = act_apply(ct[lr * CLD + c8 * CPL + j] + bv[j], ACT);
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
  if (p.dbg_skip == 2) {
    if (p.dbg && tid == 0) p.dbg[(blockIdx.y * gridDim.x + blockIdx.x) * 4 + 3] = __builtin_amdgcn_s_memtime();
    return;
  }
  if (p.dbg_skip == 1) { c32 = nullptr; c16h = nullptr; c16l = nullptr; }
  switch (p.act) {  // one specialised copy of the epilogue per activation: no per-element switch
    case MER_ACT_GELU: epilogue(std::integral_constant<int, MER_ACT_GELU>{}); break;
    case MER_ACT_QUICK_GELU: epilogue(std::integral_constant<int, MER_ACT_QUICK_GELU>{}); break;
    case MER_ACT_RELU: epilogue(std::integral_constant<int, MER_ACT_RELU>{}); break;
    default: epilogue(std::integral_constant<int, MER_ACT_NONE>{}); break;
  }
  if (p.dbg && tid == 0) p.dbg[(blockIdx.y * gridDim.x + blockIdx.x) * 4 + 3] = __builtin_amdgcn_s_memtime();
}

int g_gemm_skip = 0;
int g_gemm_glds = 1;
unsigned long long* g_gemm_dbg = nullptr;  // mer_set_debug_buffer(); also stamped by attn_sp_kernel  // mer_set_option("gemm_glds", 0) forces the register-staged loader (A/B testing)

template <typename T, int BM, int BN, int BK, int WM, int WN, int AP, int WP, int NS>
static int launch(const Gemm16Params& p0, int nbatch, hipStream_t st) {
  Gemm16Params p = p0;
  p.tiles_m = (int)cdiv(p.M, BM);
  p.tiles_n = (int)cdiv(p.N, BN);
  dim3 grid(p.tiles_m * p.tiles_n, nbatch, 1), block(WM * WN * 64, 1, 1);
  // algorithmic work of this launch: 2*M*N*K flops (one pass, whatever AP/WP execute), A + W read once,
  // outputs (+ residual) touched once
  const double mn = (double)p.M * p.N * nbatch;
  ProfScope prof(AP == 2 ? "gemm16_x3" : (WP == 2 ? "gemm16_w2" : "gemm16"), 2.0 * mn * p.K,
                 2.0 * AP * nbatch * (double)p.M * p.K + 2.0 * WP * (double)p.N * p.K * (nbatch / p.nb_inner > 0 ? p.nb_inner : 1) +
                     mn * ((p.c32 ? 4 : 0) + (p.c16_hi ? 2 : 0) + (p.c16_lo ? 2 : 0) + (p.residual ? 4 : 0)),
                 st);
  if (g_gemm_glds == 2 && p.K % BK == 0)  // A/B: LDS-DMA loader, plain double buffering
    hipLaunchKernelGGL((gemm16_kernel<T, BM, BN, BK, WM, WN, AP, WP, true, 2>), grid, block, 0, st, p);
  else if (g_gemm_glds && p.K % BK == 0)
    hipLaunchKernelGGL((gemm16_kernel<T, BM, BN, BK, WM, WN, AP, WP, true, NS>), grid, block, 0, st, p);
  else
    hipLaunchKernelGGL((gemm16_kernel<T, BM, BN, BK, WM, WN, AP, WP, false, 2>), grid, block, 0, st, p);
  return check_launch("gemm16");
}

template <typename T>
static int dispatch(const Gemm16Params& p, int nbatch, int passes, int tile, hipStream_t st) {
  if (tile == 2) {
    if (passes == 3) return launch<T, 128, 64, 32, 2, 2, 2, 2, 3>(p, nbatch, st);
    if (passes == 2) return launch<T, 128, 64, 32, 2, 2, 1, 2, 3>(p, nbatch, st);
    return launch<T, 128, 64, 32, 2, 2, 1, 1, 4>(p, nbatch, st);
  }
  if (tile == 3) {  // 256x256, 8 waves (2x4), one workgroup per CU: twice the FLOP per byte pulled into the CU
    if (passes == 3) return launch<T, 256, 256, 32, 2, 4, 2, 2, 2>(p, nbatch, st);
    if (passes == 2) return launch<T, 256, 256, 32, 2, 4, 1, 2, 3>(p, nbatch, st);
    return launch<T, 256, 256, 32, 2, 4, 1, 1, 4>(p, nbatch, st);
  }
  // stage counts keep the LDS footprint at <= 80 KB so two workgroups share a CU (the C staging tile of the
  // epilogue needs 69.6 KB anyway): 1-pass 4 x 16 KB, 2-pass 3 x 24 KB, 3-pass 2 x 32 KB.
  if (passes == 3) return launch<T, 128, 128, 32, 2, 2, 2, 2, 2>(p, nbatch, st);
  if (passes == 2) return launch<T, 128, 128, 32, 2, 2, 1, 2, 3>(p, nbatch, st);
  return launch<T, 128, 128, 32, 2, 2, 1, 1, 4>(p, nbatch, st);
}

}  // namespace mer

extern "C" int mer_set_debug_buffer(void* device_u64_buffer) {
  mer::g_gemm_dbg = (unsigned long long*)device_u64_buffer;
  return MER_OK;
}

namespace mer { extern int g_attn_force_nkt; }
extern "C" int mer_set_option(const char* name, int value) {
  if (name && strcmp(name, "gemm_glds") == 0) { mer::g_gemm_glds = value; return MER_OK; }
  if (name && strcmp(name, "gemm_dbg_skip") == 0) { mer::g_gemm_skip = value; return MER_OK; }
  if (name && strcmp(name, "attn_force_nkt") == 0) { mer::g_attn_force_nkt = value; return MER_OK; }
  mer::set_error("mer_set_option: unknown option '%s'", name ? name : "(null)");
  return MER_EINVAL;
}

extern "C" int mer_gemm16(const mer_gemm16_args* a, mer_stream_t stream) {
  using namespace mer;
  MER_REQUIRE(a != nullptr, MER_EINVAL, "mer_gemm16: null args");
  MER_REQUIRE(a->M > 0 && a->N > 0 && a->K > 0, MER_ESHAPE, "mer_gemm16: bad shape M=%d N=%d K=%d", a->M, a->N, a->K);
  MER_REQUIRE(a->a_hi && a->w_hi, MER_EINVAL, "mer_gemm16: a_hi / w_hi must be non-null");
  MER_REQUIRE(a->passes >= 1 && a->passes <= 3, MER_EINVAL, "mer_gemm16: passes must be 1, 2 or 3 (got %d)", a->passes);
  MER_REQUIRE(a->passes < 2 || a->w_lo, MER_EINVAL, "mer_gemm16: passes>=2 needs w_lo");
  MER_REQUIRE(a->passes < 3 || a->a_lo, MER_EINVAL, "mer_gemm16: passes=3 needs a_lo");
  MER_REQUIRE(a->K % 8 == 0 && a->lda % 8 == 0 && a->ldw % 8 == 0, MER_ESHAPE,
              "mer_gemm16: K, lda, ldw must be multiples of 8 (K=%d lda=%lld ldw=%lld)", a->K, a->lda, a->ldw);
  MER_REQUIRE(a->a_so % 8 == 0 && a->a_si % 8 == 0 && a->w_si % 8 == 0 && a->a_batch_stride % 8 == 0, MER_ESHAPE,
              "mer_gemm16: batch strides of A / W must be multiples of 8 elements");
  MER_REQUIRE(a->dtype == MER_DT_F16 || a->dtype == MER_DT_BF16, MER_EINVAL, "mer_gemm16: bad dtype %d", a->dtype);
  MER_REQUIRE(a->c32 || a->c16_hi, MER_EINVAL, "mer_gemm16: no output given");
  MER_REQUIRE(a->headmajor_T == 0 || (a->headmajor_T > 0 && a->headmajor_H > 0 && a->M % a->headmajor_T == 0 && a->N % (64 * a->headmajor_H) == 0 &&
                                      a->c16_hi && a->N % 8 == 0 && (a->nbatch <= 1)),
              MER_ESHAPE, "mer_gemm16: head-major output needs M %% T == 0, N %% (64*H) == 0, a 16-bit output and no batching");
  MER_REQUIRE(!a->c16_lo || a->c16_hi, MER_EINVAL, "mer_gemm16: c16_lo without c16_hi");
  const int nbatch = a->nbatch > 0 ? a->nbatch : 1;
  Gemm16Params p;
  p.M = a->M; p.N = a->N; p.K = a->K;
  p.a_hi = a->a_hi; p.a_lo = a->a_lo; p.lda = a->lda; p.a_rpb = a->a_rows_per_batch; p.a_bstride = a->a_batch_stride;
  p.w_hi = a->w_hi; p.w_lo = a->w_lo; p.ldw = a->ldw;
  p.bias = a->bias; p.act = a->act;
  p.residual = a->residual; p.ldr = a->ldr;
  p.c32 = a->c32; p.ldc32 = a->ldc32;
  p.c16_hi = a->c16_hi; p.c16_lo = a->c16_lo; p.ldc16 = a->ldc16;
  p.nb_inner = a->nb_inner > 0 ? a->nb_inner : 1;
  p.a_so = a->a_so; p.a_si = a->a_si; p.w_si = a->w_si; p.bias_si = a->bias_si; p.c_so = a->c_so; p.c_si = a->c_si;
  p.tiles_m = p.tiles_n = 0;
  p.dbg = g_gemm_dbg;
  p.dbg_skip = g_gemm_skip;
  p.hm_T = a->headmajor_T; p.hm_H = a->headmajor_H;
  // the vector epilogue moves 8 columns per lane with 16-byte accesses
  bool vec = (a->N % 8 == 0) && (a->c_so % 8 == 0) && (a->c_si % 8 == 0);
  if (a->residual) vec = vec && (a->ldr % 4 == 0) && (((uintptr_t)a->residual & 15) == 0);
  if (a->c32) vec = vec && (a->ldc32 % 4 == 0) && (((uintptr_t)a->c32 & 15) == 0);
  if (a->c16_hi) vec = vec && (a->ldc16 % 8 == 0) && (((uintptr_t)a->c16_hi & 15) == 0);
  if (a->c16_lo) vec = vec && (((uintptr_t)a->c16_lo & 15) == 0);
  p.vec_ok = vec ? 1 : 0;
  MER_REQUIRE((((uintptr_t)a->a_hi | (uintptr_t)a->w_hi | (uintptr_t)a->a_lo | (uintptr_t)a->w_lo) & 15) == 0, MER_EINVAL,
              "mer_gemm16: operand planes must be 16-byte aligned");
  int tile = a->tile;
  // 256x256 (8 waves, staggered schedule) wins whenever there are enough rows; narrow / short problems keep 4-wave tiles
  if (tile == 0) tile = (a->N <= 64) ? 2 : ((a->M >= 1024 && a->N >= 192) ? 3 : 1);
  hipStream_t st = (hipStream_t)stream;
  if (a->dtype == MER_DT_F16) return dispatch<f16>(p, nbatch, a->passes, tile, st);
  return dispatch<bf16>(p, nbatch, a->passes, tile, st);
}
