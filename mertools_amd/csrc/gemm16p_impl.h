// gemm16p_impl.h — the PERSISTENT one-pass 256x256 GEMM: C = epi(A * W^T + bias) on v_mfma_f32_16x16x32_{f16,bf16}.
//
// Same slab arithmetic as gemm16_kernel<T,256,256,32,...,GLDS,4> (gemm16_impl.h: 4-stage LDS ring filled by LDS-DMA, counted vmcnt
// across raw s_barriers, two wave groups one barrier phase apart) and bit-identical results, but what surrounds the K loop is
// rebuilt, because that is where 30-45 % of a K = 768 tile went (profiles/r02_gemm16_bench_epilogue_split.txt: 976 TF as shipped,
// 1398 TF with the epilogue skipped; s_memtime: prologue 3.6 k + K loop 32.1 k + epilogue 9.7-13 k (32 k for fp32 + residual)
// + 1.5-4.2 k hand-over cycles per tile):
//
//   * REGISTER-DIRECT EPILOGUE.  The pre-blocked weight plane is packed with its rows permuted inside every 128-row block
//     (mer_w_block_pack_p), so that the eight accumulators acc[mt][0..7][r] of lane (li, lg) — one output row, eight MFMA column
//     tiles — are CONSECUTIVE output columns: layout A (16-bit outputs) columns 8 li .. 8 li + 7, one 16-byte store per lane and
//     row, sixteen lanes = the wave's whole 128-column run (two 128-byte lines); layout B (fp32 outputs) columns 4 li .. 4 li + 3
//     and 64 + 4 li .. + 3, two 16-byte stores (and residual loads) per lane and row, each covering whole lines.  No LDS
//     transposition, no ds_write_b32, no staging tile that aliases the ring.  What a store costs the CU is its INSTRUCTION
//     (~15-23 cycles of the CU's one store path whatever the width: profiles/r04_gemm16p_timeline.txt), hence the widest form.
//   * PERSISTENT WORKGROUPS, ONE CONTINUOUS RING.  One workgroup per CU walks tiles L, L + grid, ... (the XCD-aware map of
//     gemm16_kernel) and treats them as ONE slab stream: the last three iterations of a tile already issue the next tile's first
//     slabs (+ its bias row), the ring never drains and never bursts, and there is no barrier beyond the K loop's own.  vmcnt
//     retires in order, so the counted waits of a tile's first three iterations add the epilogue's stores to their allowance: the
//     stores drain under those slabs instead of in front of them.
//   * THE STAGGER RUNS ACROSS TILES.  Group 0 (waves 0-3) finishes a tile one phase early and stores it while group 1 (waves
//     4-7, the other wave of every SIMD) issues its last 32 MFMAs; group 1 stores at the top of the next tile's first
//     iteration beside group 0's first MATH phase.
//
// Scope (anything else keeps gemm16_kernel): one pass, nbatch == 1, N % 256 == 0, K % 32 == 0, K >= 256, planes < 4 GB, output
// one 16-bit plane (EPI 0, layout A plane) or fp32 (EPI 1) (+ residual, EPI 2) (layout B plane).
#pragma once
#include "gemm16_impl.h"

namespace mer {

constexpr int P_RING = 4 * 32768;          // 4 stages x (A 16 KB + W 16 KB)
constexpr int P_BIAS = 2 * 8 * 1024;       // bias rows of the tile (256 fp32 each): the vector, or up to 8 rows of a per-sequence table; two tile parities
constexpr int P_STAMP = 2 * 12 * 16 * 8;   // timeline stamps (mer_set_debug_buffer): 2 wave groups x 12 tiles x 16 slots, dumped at exit
constexpr int P_SMEM = P_RING + P_BIAS + P_STAMP;

// counted wait with the epilogue's still-in-flight stores (sx = 0, 4 TM or 8 TM, wave-uniform) added to the allowance
template <int N, int TM>
__device__ __forceinline__ void wait_vmcnt_plus(int sx) {
  if (sx == 8 * TM) wait_vmcnt<N + 8 * TM>();
  else if (sx == 4 * TM) wait_vmcnt<N + 4 * TM>();
  else wait_vmcnt<N>();
}

// hides a value's provenance from the optimiser at this point: loop-invariant code motion otherwise hoists the epilogue's per-row
// offsets (they do not depend on the tile) out of the tile walk and keeps them — in scratch memory — across the K loop
__device__ __forceinline__ unsigned opaque(unsigned v) {
  asm volatile("" : "+v"(v));
  return v;
}

__device__ __forceinline__ unsigned long long uniform64(unsigned long long v) {
  return ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(v >> 32)) << 32) |
         (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)v);
}
// 16-byte stores with a uniform 64-bit base (SGPR pair) + a 32-bit per-lane byte offset (+ immediate).  Deliberately invisible to
// hipcc's waitcnt pass: it would otherwise protect "its" stores with near-zero vmcnt waits in front of the next epilogue, i.e. drain
// the ring; their completion is counted by hand (the `sx` allowance of the K loop's waits).  `s_nop 1`: a 16-byte store's data
// registers must not be overwritten in the next two issue slots, and hipcc does not pad an asm statement.
template <int OFF, bool NT>
__device__ __forceinline__ void gstore16_s(unsigned long long sbase, unsigned voff, u32x4 v) {
  if (NT) asm volatile("global_store_dwordx4 %0, %1, %2 offset:%3 nt\n\ts_nop 1" :: "v"(voff), "v"(v), "s"(sbase), "n"(OFF) : "memory");
  else asm volatile("global_store_dwordx4 %0, %1, %2 offset:%3\n\ts_nop 1" :: "v"(voff), "v"(v), "s"(sbase), "n"(OFF) : "memory");
}

// ---- the lean 16-bit epilogue (round 6).  A wave's epilogue is bound by its OWN instruction issue: ~4.75 cycles per instruction whatever
// the other wave of the SIMD does (profiles/r06_gemm16p_before_lean_epilogue_timeline.txt: 3.7 k cycles for the 16 rows of a plain 16-bit tile = ~45
// instructions per row and store; 7.5 k with quick_gelu), not by VALU throughput or the store path.  So the fast path below spends
// instructions, not cycles: rows go in PAIRS — acc[mt][nt] holds rows r .. r + 3 of one column in adjacent registers, so rows (r, r + 1)
// of a column are a packed-fp32 operand as they sit (v_pk_add / v_pk_mul / v_pk_fma: each half the IEEE result of the scalar instruction,
// hence the same bits) — and every wave-uniform decision (interior tile, bias vector or table, debug skips) is taken once per tile.
typedef float f32x2v __attribute__((ext_vector_type(2)));

// both halves of a pair through exactly act_apply's arithmetic: the same operations, in the same order, per element
template <int ACT>
__device__ __forceinline__ f32x2v act_pair(f32x2v x) {
  if constexpr (ACT == MER_ACT_NONE) {
    return x;
  } else if constexpr (ACT == MER_ACT_QUICK_GELU) {   // x * rcp(1 + __expf(-1.702 x)), __expf(t) = v_exp_f32(t * 0x3fb8aa3b)
    const float l2e = __builtin_bit_cast(float, 0x3fb8aa3bu);
    const f32x2v u = (x * f32x2v{-1.702f, -1.702f}) * f32x2v{l2e, l2e};
    const f32x2v d = f32x2v{1.0f, 1.0f} + f32x2v{__builtin_amdgcn_exp2f(u[0]), __builtin_amdgcn_exp2f(u[1])};
    return x * f32x2v{__builtin_amdgcn_rcpf(d[0]), __builtin_amdgcn_rcpf(d[1])};
  } else if constexpr (ACT == MER_ACT_GELU) {         // gelu_exp2poly: the polynomial as six packed FMAs
    const f32x2v a = {fminf(fabsf(x[0]), 5.5f), fminf(fabsf(x[1]), 5.5f)};
    f32x2v q = {3.589585917e-05f, 3.589585917e-05f};
    q = __builtin_elementwise_fma(q, a, f32x2v{-7.945234977e-04f, -7.945234977e-04f});
    q = __builtin_elementwise_fma(q, a, f32x2v{8.167289912e-03f, 8.167289912e-03f});
    q = __builtin_elementwise_fma(q, a, f32x2v{-5.355345435e-02f, -5.355345435e-02f});
    q = __builtin_elementwise_fma(q, a, f32x2v{-4.586574375e-01f, -4.586574375e-01f});
    q = __builtin_elementwise_fma(q, a, f32x2v{-1.151242835e+00f, -1.151242835e+00f});
    q = __builtin_elementwise_fma(q, a, f32x2v{-9.999880846e-01f, -9.999880846e-01f});
    return f32x2v{fmaf(-fabsf(x[0]), __builtin_amdgcn_exp2f(q[0]), fmaxf(x[0], 0.f)), fmaf(-fabsf(x[1]), __builtin_amdgcn_exp2f(q[1]), fmaxf(x[1], 0.f))};
  } else {
    return f32x2v{act_apply(x[0], ACT), act_apply(x[1], ACT)};
  }
}

// one row block (16 rows of the wave tile: 4 rows per lane, as two row pairs) of an INTERIOR tile's 16-bit epilogue.  TABLE: per-sequence
// bias rows from the LDS slot (bl + slot row * 1024; bq / brem as in the generic epilogue), else the bias vector as pairs {b, b} in bp.
template <typename T, int ACT, int TM, bool TABLE>
__device__ __forceinline__ void epi0_block(const f32x4 (&acc)[TM][8], int mt, const f32x2v (&bp)[8], const char* bl, int& bq, int& brem, int bias_T,
                                           unsigned long long cb, unsigned& vo, unsigned rstep) {
  typedef typename T16<T>::v8 v8;
#pragma unroll
  for (int rp = 0; rp < 2; ++rp) {
    f32x2v x[8];
    if constexpr (TABLE) {
      const f32x4 a0 = *reinterpret_cast<const f32x4*>(bl + bq * 1024), a1 = *reinterpret_cast<const f32x4*>(bl + bq * 1024 + 16);
      brem += 1;
      if (brem >= bias_T) { brem -= bias_T; ++bq; }
      const f32x4 c0 = *reinterpret_cast<const f32x4*>(bl + bq * 1024), c1 = *reinterpret_cast<const f32x4*>(bl + bq * 1024 + 16);
      brem += rp ? 13 : 1;
      if (brem >= bias_T) { brem -= bias_T; ++bq; }
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) {
        x[nt] = f32x2v{acc[mt][nt][2 * rp] + a0[nt], acc[mt][nt][2 * rp + 1] + c0[nt]};
        x[4 + nt] = f32x2v{acc[mt][4 + nt][2 * rp] + a1[nt], acc[mt][4 + nt][2 * rp + 1] + c1[nt]};
      }
    } else {
#pragma unroll
      for (int nt = 0; nt < 8; ++nt) x[nt] = f32x2v{acc[mt][nt][2 * rp], acc[mt][nt][2 * rp + 1]} + bp[nt];
    }
    v8 h0, h1;
#pragma unroll
    for (int nt = 0; nt < 8; ++nt) {
      const f32x2v y = act_pair<ACT>(x[nt]);
      // the fp32 values are pinned: hipcc otherwise fuses an activation's last FMA with the conversion (v_fma_mixlo_f16: ONE rounding to
      // 16 bits where every other epilogue rounds to fp32 first) — 198 of 12.6 M GELU outputs then differ from the tile kernel's
      float y0 = y[0], y1 = y[1];
      asm volatile("" : "+v"(y0), "+v"(y1));
      h0[nt] = T16<T>::from_f32(y0);
      h1[nt] = T16<T>::from_f32(y1);
    }
    gstore16_s<0, true>(cb, vo, __builtin_bit_cast(u32x4, h0));
    vo += rstep;
    gstore16_s<0, true>(cb, vo, __builtin_bit_cast(u32x4, h1));
    vo += rstep;
  }
  vo += 12 * rstep;
}

// plane row (within its 128-row block) that block row r = 16 nt + li must hold so that lane li's accumulators nt = 0 .. 7 are
//   layout 0 (A): columns 8 li + nt;   layout 1 (B): columns 64 (nt >> 2) + 4 li + (nt & 3)
__host__ __device__ inline int p_perm_row(int r, int layout) {
  const int li = r & 15, nt = (r >> 4) & 7;
  return (r & ~127) + (layout == 0 ? 8 * li + nt : 64 * (nt >> 2) + 4 * li + (nt & 3));
}

// TM = 4: 256 x 256 tiles (a wave: 64 x 128).  TM = 3: 192 x 256 tiles (a wave: 48 x 128) for planes whose 256-row tile count fills
// the last round of workgroups badly (63 row tiles x 9 column tiles on 256 CUs = 2.2 rounds, paid as 3): the same slab stream — the
// ring still carries 256 A rows per slab, the last 64 of them (the next row tile's, or a clamped repeat of row M - 1) unread — the
// same k order per output element, hence the same bits.
template <typename T, int EPI, int ACT, int TM>
__global__ __launch_bounds__(512) void gemm16p_kernel(const Gemm16Params p) {
  typedef typename T16<T>::v8 v8;
  constexpr int TN = 8, TROWS = 64 * TM, WROWS = 16 * TM;
  constexpr int STAGE = 32768, A_PLANE = 16384;
  __shared__ __attribute__((aligned(16))) char smem[P_SMEM];

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;      // 4 x 2 waves of (16 TM) x 128
  const int li = lane & 15, lg = lane >> 4;
  const bool g1 = wave >= 4;
  const int nk = p.K >> 5;
  const int nblk = p.tiles_m * p.tiles_n;
  const unsigned lds0 = lds_offset_of(smem);

  auto tile_of = [&](int L, int& tm, int& tn) __attribute__((always_inline)) {
    const int xcd = L & 7, loc = L >> 3;
    const int q = nblk >> 3, r = nblk & 7;
    const int swz = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
    tn = swz % p.tiles_n;
    tm = swz / p.tiles_n;
  };

  // ---- LDS-DMA addressing.  Thread t brings chunk (t & 3) of rows (t >> 2) and (t >> 2) + 128 of each plane; a wave's
  // instruction fills 1 KiB = 16 rows.  W is pre-blocked: its per-lane offset never changes; A's follows the row tile.
  const int ld_ch = tid & 3, ld_row0 = tid >> 2;
  const unsigned w_o32 = (unsigned)((ld_row0 * 32 + ld_ch * 8) * 2);   // second chunk: + 8192 (row + 128), folded into the base
  auto a_offsets = [&](int m0_, unsigned (&o)[2]) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int rr = ld_row0 + i * 128;
      int m = m0_ + rr;
      m = m < p.M ? m : p.M - 1;
      const long long off = (p.a_rpb > 0) ? (long long)(m / p.a_rpb) * p.a_bstride + (long long)(m % p.a_rpb) * p.lda : (long long)m * p.lda;
      o[i] = (unsigned)((off + ((ld_ch ^ swz_of<4>(rr)) << 3)) * 2);
    }
  };
  const char* a_plane = (const char*)p.a_hi;
  const char* w_plane = (const char*)p.w_hi;
  // slab kt of the tile whose A offsets are `ao` and column tile `tn_`, into ring stage `stage`
  auto glds_slab = [&](const unsigned (&ao_)[2], int tn_, int kt, int stage) __attribute__((always_inline)) {
    const char* ab = a_plane + (long long)kt * 64;
    const char* wb = w_plane + ((long long)tn_ * nk + kt) * 16384;
    const unsigned lb = lds0 + stage * STAGE + wave * 1024;
    dma16_sbase(ab, ao_[0], lb);
    dma16_sbase(ab, ao_[1], lb + 8192);
    dma16_sbase(wb, w_o32, lb + A_PLANE);
    dma16_sbase(wb + 8192, w_o32, lb + A_PLANE + 8192);
  };
  // bias rows of tile (tm_, tn_): wave j brings row j of the slot — the bias vector (row 0 only), or, with a per-sequence table
  // (mer_seq_bias, bias_T >= 40 rows per sequence: at most 8 sequences touch a 256-row tile), the rows of sequences
  // (256 tm_) / T .. (256 tm_ + 255) / T.  One shared copy: every reader is many barriers behind the issuing wave's counted waits.
  auto glds_bias = [&](int tm_, int tn_, int parity) __attribute__((always_inline)) {
    if (!p.bias) return;
    int srow = 0, nrow = 1;
    if (p.bias_T > 0) {
      const int last = tm_ * TROWS + TROWS - 1 < p.M ? tm_ * TROWS + TROWS - 1 : p.M - 1;
      srow = (tm_ * TROWS) / p.bias_T;
      nrow = last / p.bias_T - srow + 1;
    }
    if (wave < nrow)
      dma16_sbase((const char*)(p.bias + (long long)(srow + wave) * p.bias_ld + tn_ * 256), (unsigned)(lane * 16),
                  lds0 + P_RING + parity * 8192 + wave * 1024);
  };

  f32x4 acc[TM][TN];

  v8 af[TM], wf[TN];
  // (row >> 2) & 3 == (li >> 2) for every fragment row of this lane: one swizzle term, fragments 1 KiB apart
  const int fsw = ((lg ^ ((-(li >> 2)) & 3)) << 4);
  const int a_f0 = (wm * WROWS + li) * 64 + fsw;
  const int w_f0 = A_PLANE + (wn * 128 + li) * 64 + fsw;
  auto load_frags = [&](int stage) __attribute__((always_inline)) {
    const char* base = smem + stage * STAGE;
#pragma unroll
    for (int mt = 0; mt < TM; ++mt) af[mt] = *reinterpret_cast<const v8*>(base + a_f0 + mt * 1024);
#pragma unroll
    for (int nt = 0; nt < TN; ++nt) wf[nt] = *reinterpret_cast<const v8*>(base + w_f0 + nt * 1024);
  };
  auto math = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int mt = 0; mt < TM; ++mt)
#pragma unroll
      for (int nt = 0; nt < TN; ++nt) acc[mt][nt] = T16<T>::mfma(af[mt], wf[nt], acc[mt][nt]);
  };
  auto math0 = [&]() __attribute__((always_inline)) {   // slab 0 of a tile: C = 0 (0 + x is exact: the bits of a zeroed accumulator)
#pragma unroll
    for (int mt = 0; mt < TM; ++mt)
#pragma unroll
      for (int nt = 0; nt < TN; ++nt) acc[mt][nt] = T16<T>::mfma(af[mt], wf[nt], f32x4{0.f, 0.f, 0.f, 0.f});
  };

  // ---- epilogue of tile (tm_, tn_), bias row in parity slot `par`; returns the store allowance for the next counted waits:
  // the number of (asm, uncounted-by-hipcc) stores every lane-group of the wave has just issued — exact for interior tiles, 0 for
  // the predicated last row tile of a plane whose M is not a multiple of 256 (a skipped store makes the count an upper bound
  // only; 0 is always safe: the waits then simply cover the stores).  Residual loads are plain loads: hipcc counts ITS OWN
  // operations exactly, the asm stores between them only make its waits a little stronger than necessary.
  auto epilogue = [&](int tm_, int tn_, int par) __attribute__((always_inline)) -> int {
    if ((p.dbg_skip & 3) == 2) return 0;
    const int m0 = tm_ * TROWS + wm * WROWS, n0 = tn_ * 256 + wn * 128;
    const bool st = (p.dbg_skip & 3) != 1;
    const bool interior = tm_ * TROWS + TROWS <= p.M;   // (N % 256 == 0: no column edge)
    const char* bs = smem + P_RING + par * 8192 + wn * 512;    // this wave's 128 columns of the slot's row 0
    int row = (int)opaque((unsigned)(m0 + 4 * lg));
    // per-sequence table: slot row of output row `row` = row / T - (256 tm_) / T, tracked incrementally (T >= 40 > a step of <= 13 rows)
    int bq = 0, brem = 0;
    if (p.bias_T > 0) {
      bq = row / p.bias_T;
      brem = row - bq * p.bias_T;
      bq -= (tm_ * TROWS) / p.bias_T;
    }
    auto bias_step = [&](int d) __attribute__((always_inline)) {
      brem += d;
      if (brem >= p.bias_T) { brem -= p.bias_T; ++bq; }
    };
    if constexpr (EPI == 0) {
      f32x4 b0 = {0.f, 0.f, 0.f, 0.f}, b1 = b0;
      const char* bl = bs + li * 32;
      if (p.bias) { b0 = *reinterpret_cast<const f32x4*>(bl); b1 = *reinterpret_cast<const f32x4*>(bl + 16); }
      const unsigned long long cb = uniform64((unsigned long long)((T*)p.c16_hi + (long long)m0 * p.ldc16 + n0));
      asm volatile("s_nop 4" :: "s"(cb));   // v_readfirstlane -> SGPR -> VMEM address: 5 wait states
      const unsigned rstep = (unsigned)p.ldc16 * 2;
      unsigned vo = opaque((unsigned)(4 * lg) * rstep + li * 16);
      if (interior && st) {   // the lean path (epi0_block): every store of the tile is issued
        f32x2v bp[8];
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) { bp[nt] = f32x2v{b0[nt], b0[nt]}; bp[4 + nt] = f32x2v{b1[nt], b1[nt]}; }
        if (p.bias_T > 0) {
#pragma unroll
          for (int mt = 0; mt < TM; ++mt) epi0_block<T, ACT, TM, true>(acc, mt, bp, bl, bq, brem, p.bias_T, cb, vo, rstep);
        } else {
#pragma unroll
          for (int mt = 0; mt < TM; ++mt) epi0_block<T, ACT, TM, false>(acc, mt, bp, bl, bq, brem, p.bias_T, cb, vo, rstep);
        }
        return 4 * TM;
      }
#pragma unroll
      for (int mt = 0; mt < TM; ++mt) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          if (p.bias_T > 0) {
            b0 = *reinterpret_cast<const f32x4*>(bl + bq * 1024);
            b1 = *reinterpret_cast<const f32x4*>(bl + bq * 1024 + 16);
            bias_step(r == 3 ? 13 : 1);
          }
          v8 h;
#pragma unroll
          for (int nt = 0; nt < 4; ++nt) {
            h[nt] = T16<T>::from_f32(act_apply(acc[mt][nt][r] + b0[nt], ACT));
            h[4 + nt] = T16<T>::from_f32(act_apply(acc[mt][4 + nt][r] + b1[nt], ACT));
          }
          if (st && (interior || row + r < p.M)) gstore16_s<0, true>(cb, vo, __builtin_bit_cast(u32x4, h));
          vo += rstep;
        }
        vo += 12 * rstep;
        row += 16;
      }
      return (st && interior) ? 4 * TM : 0;
    } else {
      f32x4 b0 = {0.f, 0.f, 0.f, 0.f}, b1 = b0;
      const char* bl = bs + li * 16;
      if (p.bias) { b0 = *reinterpret_cast<const f32x4*>(bl); b1 = *reinterpret_cast<const f32x4*>(bl + 256); }
      const unsigned long long cb = uniform64((unsigned long long)(p.c32 + (long long)m0 * p.ldc32 + n0));
      asm volatile("s_nop 4" :: "s"(cb));
      const unsigned cstep = (unsigned)p.ldc32 * 4;
      unsigned vo = opaque((unsigned)(4 * lg) * cstep + li * 16);
      if constexpr (EPI == 1) {
#pragma unroll
        for (int mt = 0; mt < TM; ++mt) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            if (p.bias_T > 0) {
              b0 = *reinterpret_cast<const f32x4*>(bl + bq * 1024);
              b1 = *reinterpret_cast<const f32x4*>(bl + bq * 1024 + 256);
              bias_step(r == 3 ? 13 : 1);
            }
            f32x4 v0, v1;
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) {
              v0[nt] = act_apply(acc[mt][nt][r] + b0[nt], ACT);
              v1[nt] = act_apply(acc[mt][4 + nt][r] + b1[nt], ACT);
            }
            if (st && (interior || row + r < p.M)) {
              gstore16_s<0, false>(cb, vo, __builtin_bit_cast(u32x4, v0));
              gstore16_s<256, false>(cb, vo, __builtin_bit_cast(u32x4, v1));
            }
            vo += cstep;
          }
          vo += 12 * cstep;
          row += 16;
        }
        return (st && interior) ? 8 * TM : 0;
      } else {
        // The residual may BE the output (the pre-LN stream is updated in place); source order is kept (asm volatile + "memory")
        // and vmcnt retires in order, so a load issued behind a store waits for that store's acknowledgement too.  The tile's 8
        // (row tile, 64-column half) pieces of 4 rows run as a 4-slot pipeline (64 registers): pieces 0-3 go
        // out before the first store, piece j + 4 behind piece j's stores — a wait never names a store younger than two pieces back.
        const char* rb = (const char*)(p.residual + (long long)m0 * p.ldr + n0);
        const unsigned rstep = (unsigned)p.ldr * 4;
        unsigned ro = opaque((unsigned)(4 * lg) * rstep + li * 16);
        int lrow = row;
        f32x4 rr[4][4];
        auto issue = [&](int j) __attribute__((always_inline)) {   // piece j = (row tile j / 2, half j % 2)
#pragma unroll
          for (int r = 0; r < 4; ++r)
            rr[j & 3][r] = (interior || lrow + r < p.M) ? *reinterpret_cast<const f32x4*>(rb + ro + (unsigned)r * rstep + (j & 1) * 256)
                                                        : f32x4{0.f, 0.f, 0.f, 0.f};
          if (j & 1) { ro += 16 * rstep; lrow += 16; }
        };
        issue(0); issue(1); issue(2); issue(3);
        int bqr[4] = {0, 0, 0, 0};    // table mode: the slot row of each of the row tile's 4 rows
#pragma unroll
        for (int j = 0; j < 2 * TM; ++j) {
          const int mt = j >> 1, half = j & 1;
          if (half == 0 && p.bias_T > 0) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              bqr[r] = bq * 1024;
              bias_step(r == 3 ? 13 : 1);
            }
          }
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const f32x4 bb = p.bias_T > 0 ? *reinterpret_cast<const f32x4*>(bl + bqr[r] + half * 256) : (half ? b1 : b0);
            f32x4 v;
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) v[nt] = act_apply(acc[mt][half * 4 + nt][r] + bb[nt], ACT) + rr[j & 3][r][nt];
            if (st && (interior || row + r < p.M)) {
              if (half) gstore16_s<256, false>(cb, vo + (unsigned)r * cstep, __builtin_bit_cast(u32x4, v));
              else gstore16_s<0, false>(cb, vo + (unsigned)r * cstep, __builtin_bit_cast(u32x4, v));
            }
          }
          if (half) { vo += 16 * cstep; row += 16; }
          if (j + 4 < 2 * TM) issue(j + 4);
        }
        return (st && interior) ? 8 * TM : 0;
      }
    }
  };

  // ---- the free stagger (fp32 + residual launches).  Persistent workgroups start together and walk equal tiles, so the whole chip reaches
  // the epilogue at once: 256 CUs then move their 512 KB each (residual in, fp32 out) at the HBM roofline — 6-7 TB/s for ~100 us of a CLIP
  // attention-output launch — while the K loops in between leave HBM three quarters idle.  Delaying every second workgroup by half a tile
  // costs the delayed ones half a tile at the end (round 4: +1.5 % / -5 %).  But nblk % grid workgroups walk one tile MORE than the others
  // (CLIP fc2 / attention output: 158 of 256 do five tiles, 98 do four): the short walkers can start half a tile late for nothing, and
  // 38 % of the chip then stores while the rest computes.  (gemm_dbg_skip & 8 turns it off, for A/B.)
  if constexpr (EPI == 2) {
    const int rem = nblk % (int)gridDim.x;
    if (rem > 0 && nblk > (int)gridDim.x && (int)blockIdx.x >= rem && (p.dbg_skip & 8) == 0) {
      const unsigned long long t0 = __builtin_amdgcn_s_memtime(), d = (unsigned long long)nk * 700ull + 20000ull;
      // (bounded twice: by the shader-clock counter and by the sleeps themselves — s_sleep 32 parks the wave for ~2 k cycles — should
      //  s_memtime ever tick at another rate than the one this delay was sized on)
      for (int it = (int)(d >> 11) + 8; it > 0 && __builtin_amdgcn_s_memtime() - t0 < d; --it) __builtin_amdgcn_s_sleep(32);
    }
  }

  // ---- tile walk: one continuous slab stream; slab kt of the current tile lives in ring stage (base + kt) & 3 ----
  int L = blockIdx.x;
  int tm, tn;
  tile_of(L, tm, tn);
  unsigned ao[2], aon[2] = {0u, 0u};   // A offsets of the current / the next tile
  a_offsets(tm * TROWS, ao);
  glds_bias(tm, tn, 0);
#pragma unroll
  for (int s = 0; s < 4; ++s) glds_slab(ao, tn, s, s);
  wait_vmcnt<12>();                 // slab 0 (and the bias row, older) of this wave's share ...
  __builtin_amdgcn_s_barrier();     // ... and of every wave's
  int seq = 0;                      // tiles done by this workgroup: parity of the bias slot
  int base = 0;                     // ring stage of the current tile's slab 0
  int sx = 0;                       // stores of the last epilogue that the next counted waits may leave in flight

  // timeline instrumentation (tuning runs only): lane 0 of waves 0 and 4 stamps s_memtime into LDS (no VMEM traffic that would
  // disturb the counted waits), slots: 0 tile start, 1-8 past the mid barrier of slabs 0-7, 9 K loop done, 10 boundary slab issued,
  // 11 epilogue done
  unsigned long long* stl = reinterpret_cast<unsigned long long*>(smem + P_RING + P_BIAS);
  auto stamp = [&](int slot) __attribute__((always_inline)) {
    if (p.dbg && (wave & 3) == 0 && seq < 12 && lane == 0) stl[((wave >> 2) * 12 + seq) * 16 + slot] = __builtin_amdgcn_s_memtime();
  };
  if (p.dbg) {
    for (int i = tid; i < P_STAMP / 8; i += 512) stl[i] = 0;
    __builtin_amdgcn_s_barrier();
  }
  if (g1) __builtin_amdgcn_s_barrier();   // group 1 runs one barrier phase behind, from here to the end of the walk

  for (;;) {
    const int Ln = L + (int)gridDim.x;
    const bool has_next = Ln < nblk;
    int ntm = 0, ntn = 0;
    stamp(0);
#pragma clang loop unroll(disable)   // (also keeps hipcc from peeling the first three iterations: three more copies of the loop body)
    for (int kt = 0; kt < nk; ++kt) {
      // the next tile's coordinates and A offsets (integer divisions, 64-bit address math) are worked out HERE, in the slack of a LOAD
      // phase — not between the epilogue and the next tile's first slab, where every cycle is exposed (first needed at kt = nk - 3 >= 5)
      if (kt == 1 && has_next) {
        tile_of(Ln, ntm, ntn);
        a_offsets(ntm * TROWS, aon);
      }
      // slab kt + 3 of the stream into the stage slab kt - 1 was read from (iteration 0's went out at the tile boundary).  (Issuing
      // these four DMA pieces BEHIND the twelve fragment reads instead of in front of them was A/B'd in round 5: 1-3 % slower on
      // every shape, profiles/r05_gemm16p_dma_order_ab.txt.)
      if (kt >= 1) {
        if (kt + 3 < nk) glds_slab(ao, tn, kt + 3, (base + kt + 3) & 3);
        else if (has_next) {
          if (kt + 3 == nk) glds_bias(ntm, ntn, (seq + 1) & 1);
          glds_slab(aon, ntn, kt + 3 - nk, (base + kt + 3) & 3);
        }
      }
      load_frags((base + kt) & 3);
      // this wave's share of the stream's next slab has landed; two younger slabs — and, early in a tile, the stores — stay in
      // flight (the next tile's bias row rides one slot ahead of its slab 0: two waits per tile are one request stricter than needed)
      if (has_next || kt + 3 < nk) {
        if (kt < 3) wait_vmcnt_plus<8, TM>(sx);
        else wait_vmcnt<8>();
      } else wait_vmcnt<0>();               // last tile's tail: nothing younger is issued any more
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();         // mid
      if (kt < 8) stamp(1 + kt);
      if (kt == nk - 1 && g1 && has_next) glds_slab(aon, ntn, 3, (base + nk + 3) & 3);   // every LDS read of slab nk - 1 is over
      __builtin_amdgcn_s_setprio(1);
      if (kt == 0) math0();   // (the tile's first slab starts the accumulators: no 128-register clear between the epilogue and the next tile)
      else math();
      __builtin_amdgcn_s_setprio(0);
      if (!(kt == nk - 1 && g1)) __builtin_amdgcn_s_barrier();   // end (group 1 takes its last one behind its epilogue)
    }
    // Both groups store right behind their last MATH phase — group 0 while group 1 still issues its last 32 MFMAs, then both at
    // once (the CU's one store path is what takes the time: two epilogues side by side cost what one does) — and meet again at
    // group 0's first mid barrier of the next tile.
    stamp(9);
    if (has_next && !g1) glds_slab(aon, ntn, 3, (base + nk + 3) & 3);
    stamp(10);
    __builtin_amdgcn_sched_barrier(0);
    sx = epilogue(tm, tn, seq & 1);
    __builtin_amdgcn_sched_barrier(0);   // the fresh accumulators must not be live beside the ones being stored
    stamp(11);
    if (has_next) {
      if (g1) __builtin_amdgcn_s_barrier();   // group 1's end barrier of slab nk - 1 == group 0's mid barrier of the next tile's slab 0
    }
    ++seq;
    if (!has_next) break;
    base = (base + nk) & 3;
    L = Ln; tm = ntm; tn = ntn;
    ao[0] = aon[0]; ao[1] = aon[1];
  }
  if (p.dbg) {
    __syncthreads();
    for (int i = tid; i < P_STAMP / 8; i += 512) p.dbg[(long long)blockIdx.x * (P_STAMP / 8) + i] = stl[i];
  }
}

int device_cu_count();

template <typename T, int EPI, int TM>
static int launch_p_act(const Gemm16Params& p, dim3 grid, hipStream_t st) {
  dim3 block(512, 1, 1);
  if constexpr (EPI == 0) {
    switch (p.act) {
      case MER_ACT_GELU: hipLaunchKernelGGL((gemm16p_kernel<T, EPI, MER_ACT_GELU, TM>), grid, block, 0, st, p); break;
      case MER_ACT_QUICK_GELU: hipLaunchKernelGGL((gemm16p_kernel<T, EPI, MER_ACT_QUICK_GELU, TM>), grid, block, 0, st, p); break;
      case MER_ACT_GELU_TANH: hipLaunchKernelGGL((gemm16p_kernel<T, EPI, MER_ACT_GELU_TANH, TM>), grid, block, 0, st, p); break;
      default: hipLaunchKernelGGL((gemm16p_kernel<T, EPI, MER_ACT_NONE, TM>), grid, block, 0, st, p); break;
    }
  } else if constexpr (EPI == 1) {
    if (p.act == MER_ACT_GELU) hipLaunchKernelGGL((gemm16p_kernel<T, EPI, MER_ACT_GELU, TM>), grid, block, 0, st, p);
    else hipLaunchKernelGGL((gemm16p_kernel<T, EPI, MER_ACT_NONE, TM>), grid, block, 0, st, p);
  } else {
    hipLaunchKernelGGL((gemm16p_kernel<T, EPI, MER_ACT_NONE, TM>), grid, block, 0, st, p);   // (residual: no activation — checked by mer_gemm16)
  }
  return check_launch("gemm16p");
}

// eligibility was checked by mer_gemm16 (gemm16.hip); TM by dispatch_p (gemm16.hip: p_pick_tm)
template <typename T, int TM>
int dispatch_p_impl(const Gemm16Params& p0, hipStream_t st) {
  Gemm16Params p = p0;
  p.tiles_m = (int)cdiv(p.M, 64 * TM);
  p.tiles_n = p.N / 256;
  const int nblk = p.tiles_m * p.tiles_n;
  const int cus = device_cu_count();
  dim3 grid(nblk < cus ? nblk : cus, 1, 1);
  const double mn = (double)p.M * p.N;
  ProfScope prof("gemm16p", 2.0 * mn * p.K,   // (its own label: bench.py compares this launch set's algorithmic bytes with the PMC pool of the same kernels)
                 2.0 * (double)p.M * p.K + 2.0 * (double)p.N * p.K + mn * ((p.c32 ? 4 : 0) + (p.c16_hi ? 2 : 0) + (p.residual ? 4 : 0)), st);
  if (p.c16_hi) return launch_p_act<T, 0, TM>(p, grid, st);
  if (p.residual) return launch_p_act<T, 2, TM>(p, grid, st);
  return launch_p_act<T, 1, TM>(p, grid, st);
}

template <typename T, int TM> int dispatch_p_tm(const Gemm16Params& p, hipStream_t st);

// Rows per tile for an M x N plane on `cus` workgroups: a 192-row tile does 3/4 of the work of a 256-row one at a higher cost per
// flop — the load phase of a slab (4 LDS-DMA pieces + 11 fragment reads per wave) does not shrink with the 24 instead of 32 MFMAs
// it hides behind: 0.86-0.89 of the 256-row tile's TFLOP/s on CLIP's full-machine shapes (profiles/r04_gemm16_tile_rows_ab.txt) —
// and pays when it saves a round of workgroups (HuBERT b64: 63 x 9 tiles = 2.2 rounds -> 83 x 9 = 2.9 rounds of 3/4 the length).
inline int p_pick_tm(long long M, int N, int cus) {
  const long long t4 = cdiv(M, 256) * (N / 256), t3 = cdiv(M, 192) * (N / 256);
  const double c4 = (double)cdiv(t4, cus), c3 = (double)cdiv(t3, cus) * 0.75 * 1.15;
  return c3 < 0.97 * c4 ? 3 : 4;
}

extern int g_gemm_tm;   // "gemm_tm": 0 = p_pick_tm, 3 / 4 = forced (tests, A/B)

template <typename T>
int dispatch_p_pick(const Gemm16Params& p, hipStream_t st) {
  const int tm = g_gemm_tm == 3 || g_gemm_tm == 4 ? g_gemm_tm : p_pick_tm(p.M, p.N, device_cu_count());
  return tm == 3 ? dispatch_p_tm<T, 3>(p, st) : dispatch_p_tm<T, 4>(p, st);
}

template <typename T> int dispatch_p(const Gemm16Params& p, hipStream_t st);

}  // namespace mer
