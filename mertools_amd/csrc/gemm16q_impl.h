// gemm16q_impl.h — the persistent one-pass 256x256 GEMM with LAGGED wave groups (round 6): C = epi(A * W^T + bias), bit-identical to
// gemm16p_kernel / gemm16_kernel (same MFMA, same k order per output element).
//
// What it changes against gemm16p_kernel (gemm16p_impl.h), and why.  There the two wave groups (waves 0-3 / 4-7: the two waves of every
// SIMD) run the same slab one barrier phase apart, so both reach the end of a tile together: the two epilogues run side by side on each
// SIMD's VALU and on the CU's one store path while the matrix pipe idles — 6.2 k cycles per K = 768 tile with a plain 16-bit epilogue,
// 13.2 k with quick_gelu (two waves x 128 outputs x ~25 VALU cycles on one SIMD), 31-42 k for fp32 + residual
// (profiles/r04_gemm16p_timeline.txt) against 32.4 k of K loop.  Decoupling the groups onto different tiles would need a
// private W stream per group: 48 KB of LDS-DMA per slab pair instead of 32 = 36 B/clk/CU at today's slab rate, against the ~38 the
// L2 -> LDS path delivers — the same wall the 128 x 256 two-workgroup tile hit in round 3.  Here the groups keep SHARING
// every W slab, but group 1 runs L = 2 S + 1 barrier phases (S slabs and a phase) behind group 0:
//
//   * A ring per group (its own 128 rows of the slab, D + 1 stages of 8 KB), ONE W ring of S + D + 1 stages of 16 KB that both groups
//     read, S slabs apart.  Every slab is still DMA'd once: 32 KB per slab pair, as before.  D = prefetch distance in slabs.
//   * The epilogue is cut into 2 S pieces (row blocks of the wave tile) with an s_barrier behind each, so a group's epilogue occupies
//     2 S phases of the common barrier cadence.  While group 0 converts / stores tile i, group 1 is in LOAD / MATH of tile i's last S
//     slabs; while group 1 stores, group 0 is in the first S slabs of tile i + 1.  An epilogue piece never runs beside the other
//     group's epilogue: its VALU and its stores have the SIMD / the store path to themselves, and half of the boundary's phases
//     carry a MATH phase.  Boundary cost: 2 S phases (when a piece fits a phase) instead of both epilogues + a pipeline refill.
//   * DMA issue stays balanced, 4 pieces per wave and LOAD phase: 2 of the group's own A rows (slab k + D) + 2 of a W slab — group 0
//     brings the first half of W slab k + D, group 1 the second half of W slab k + S + D (both: "my slab index + a constant", in
//     every tile position, across tile boundaries).  A reader finds both halves behind a counted vmcnt of the issuing waves and a
//     barrier: group 1's half of slab G is waited for at its LOAD(G - S - 1), the phase before group 0's LOAD(G).
//   * WAR: W slab G + D goes into the stage of slab G - S - 1, whose last reader (group 1, LOAD(G - S - 1)) finished one phase
//     earlier; A slab k + D into the stage of the group's own slab k - 1.
//
// Scope and arguments: exactly gemm16p_kernel's (Gemm16Params with w_blk = 2: the row-permuted pre-blocked plane).
#pragma once
#include "gemm16p_impl.h"

namespace mer {

template <int D, int S>
struct QCfg {
  static constexpr int NA = D + 1, NW = S + D + 1, LAG = 2 * S + 1, NP = 2 * S;
  static constexpr int A_ST = 8192, W_ST = 16384;
  static constexpr int W0 = 2 * NA * A_ST;             // [A ring group 0][A ring group 1][W ring][bias slots]
  static constexpr int BIAS0 = W0 + NW * W_ST;
  static constexpr bool STAMPS = BIAS0 + P_BIAS + P_STAMP <= 163840;   // timeline stamps (tuning runs) where the rings leave room: D = 2
  static constexpr int SMEM = BIAS0 + P_BIAS + (STAMPS ? P_STAMP : 0);
  static_assert(SMEM <= 163840, "LDS budget: 160 KiB per workgroup");
};

template <typename T, int EPI, int ACT, int TM, int D, int S>
__global__ __launch_bounds__(512) void gemm16q_kernel(const Gemm16Params p) {
  typedef typename T16<T>::v8 v8;
  typedef QCfg<D, S> Q;
  constexpr int TN = 8, TROWS = 64 * TM, WROWS = 16 * TM, GROWS = 32 * TM;
  constexpr int NA = Q::NA, NW = Q::NW, NP = Q::NP;
  __shared__ __attribute__((aligned(16))) char smem[Q::SMEM];

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = wave >> 2, wg = wave & 3;          // group, wave within the group
  const int wmg = wg >> 1, wn = wg & 1;              // 2 x 2 waves of (16 TM) x 128 per group
  const int wm = 2 * grp + wmg;                      // row block of the wave within the tile
  const int li = lane & 15, lg = lane >> 4;
  const bool g1 = grp != 0;
  const int WOFF = D + (g1 ? S : 0);                 // W slab this group's LOAD(k) brings: k + WOFF
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

  // ---- LDS-DMA addressing.  Group thread tg brings chunk (tg & 3) of rows (tg >> 2) and (tg >> 2) + 64 of the group's 128 A rows (a
  // wave's instruction = 16 rows = 1 KiB); of a W slab (pre-blocked: the 16 KB LDS image, contiguous) the group brings KiB pieces
  // 8 grp + 2 wg and + 1.
  const int tg = tid & 255;
  const int ld_ch = tg & 3, ld_row0 = tg >> 2;
  const unsigned w_o32 = (unsigned)(lane * 16 + (grp * 8 + wg * 2) * 1024);
  auto a_offsets = [&](int m0g, unsigned (&o)[2]) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int rr = ld_row0 + i * 64;
      int m = m0g + rr;
      m = m < p.M ? m : p.M - 1;
      const long long off = (p.a_rpb > 0) ? (long long)(m / p.a_rpb) * p.a_bstride + (long long)(m % p.a_rpb) * p.lda : (long long)m * p.lda;
      o[i] = (unsigned)((off + ((ld_ch ^ swz_of<4>(rr)) << 3)) * 2);
    }
  };
  const char* a_plane = (const char*)p.a_hi;
  const char* w_plane = (const char*)p.w_hi;
  const unsigned a_ring = lds0 + grp * (NA * Q::A_ST) + wg * 1024;
  const unsigned w_ring = lds0 + Q::W0 + (grp * 8 + wg * 2) * 1024;
  auto dma_a = [&](const unsigned (&ao_)[2], int kt, int stage) __attribute__((always_inline)) {
    const char* ab = a_plane + (long long)kt * 64;
    const unsigned lb = a_ring + stage * Q::A_ST;
    dma16_sbase(ab, ao_[0], lb);
    dma16_sbase(ab, ao_[1], lb + 4096);
  };
  auto dma_w = [&](int tn_, int kt, int stage) __attribute__((always_inline)) {
    const char* wb = w_plane + ((long long)tn_ * nk + kt) * 16384;
    const unsigned lb = w_ring + stage * Q::W_ST;
    dma16_sbase(wb, w_o32, lb);
    dma16_sbase(wb + 1024, w_o32, lb + 1024);
  };
  // bias rows of tile (tm_, tn_) into parity slot `parity` (gemm16p_kernel::glds_bias): group 0's wave j brings rows j and j + 4
  auto dma_bias = [&](int tm_, int tn_, int parity) __attribute__((always_inline)) {
    if (!p.bias || g1) return;
    int srow = 0, nrow = 1;
    if (p.bias_T > 0) {
      const int last = tm_ * TROWS + TROWS - 1 < p.M ? tm_ * TROWS + TROWS - 1 : p.M - 1;
      srow = (tm_ * TROWS) / p.bias_T;
      nrow = last / p.bias_T - srow + 1;
    }
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int j = wg + 4 * h;
      if (j < nrow)
        dma16_sbase((const char*)(p.bias + (long long)(srow + j) * p.bias_ld + tn_ * 256), (unsigned)(lane * 16),
                    lds0 + Q::BIAS0 + parity * 8192 + j * 1024);
    }
  };

  f32x4 acc[TM][TN];
  auto zero_acc = [&]() __attribute__((always_inline)) {
    float z;   // (an opaque zero: see gemm16p_kernel)
    asm volatile("v_mov_b32 %0, 0" : "=v"(z));
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{z, z, z, z};
  };

  v8 af[TM], wf[TN];
  const int fsw = ((lg ^ ((-(li >> 2)) & 3)) << 4);
  const int a_f0 = grp * (NA * Q::A_ST) + (wmg * WROWS + li) * 64 + fsw;
  const int w_f0 = Q::W0 + (wn * 128 + li) * 64 + fsw;
  auto load_frags = [&](int sa, int sw) __attribute__((always_inline)) {
    const char* ba = smem + a_f0 + sa * Q::A_ST;
    const char* bw = smem + w_f0 + sw * Q::W_ST;
#pragma unroll
    for (int mt = 0; mt < TM; ++mt) af[mt] = *reinterpret_cast<const v8*>(ba + mt * 1024);
#pragma unroll
    for (int nt = 0; nt < TN; ++nt) wf[nt] = *reinterpret_cast<const v8*>(bw + nt * 1024);
  };
  auto math = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int mt = 0; mt < TM; ++mt)
#pragma unroll
      for (int nt = 0; nt < TN; ++nt) acc[mt][nt] = T16<T>::mfma(af[mt], wf[nt], acc[mt][nt]);
  };
  // the barriers that close the epilogue pieces ending with row block mt: NP pieces over TM row blocks
  int stamp_seq = 0;   // (a copy of `seq` the piece barriers can see: they are defined ahead of it)
  auto piece_barriers = [&](int mt) __attribute__((always_inline)) {
    const int nb = ((mt + 1) * NP) / TM - (mt * NP) / TM;
    for (int b = 0; b < nb; ++b) {
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();
      if constexpr (Q::STAMPS) {
        if (p.dbg && wg == 0 && stamp_seq < 12 && lane == 0)
          reinterpret_cast<unsigned long long*>(smem + Q::BIAS0 + P_BIAS)[(grp * 12 + stamp_seq) * 16 + 11 + (mt * NP) / TM + b] = __builtin_amdgcn_s_memtime();
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  };

  // ---- epilogue of tile (tm_, tn_), bias rows in parity slot `par`: gemm16p_kernel's register-direct epilogue, with the piece
  // barriers inside.  Returns the store allowance of the next tile's first D counted waits (gemm16p_kernel: `sx`).
  auto epilogue = [&](int tm_, int tn_, int par) __attribute__((always_inline)) -> int {
    if ((p.dbg_skip & 3) == 2) {
#pragma unroll
      for (int mt = 0; mt < TM; ++mt) piece_barriers(mt);
      return 0;
    }
    const int m0 = tm_ * TROWS + wm * WROWS, n0 = tn_ * 256 + wn * 128;
    const bool st = (p.dbg_skip & 3) != 1;
    const bool interior = tm_ * TROWS + TROWS <= p.M;
    const char* bs = smem + Q::BIAS0 + par * 8192 + wn * 512;
    int row = (int)opaque((unsigned)(m0 + 4 * lg));
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
      asm volatile("s_nop 4" :: "s"(cb));
      const unsigned rstep = (unsigned)p.ldc16 * 2;
      unsigned vo = opaque((unsigned)(4 * lg) * rstep + li * 16);
      if (interior && st) {   // the lean path (gemm16p_impl.h: epi0_block), the piece barriers between its row blocks
        f32x2v bp[8];
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) { bp[nt] = f32x2v{b0[nt], b0[nt]}; bp[4 + nt] = f32x2v{b1[nt], b1[nt]}; }
        if (p.bias_T > 0) {
#pragma unroll
          for (int mt = 0; mt < TM; ++mt) {
            epi0_block<T, ACT, TM, true>(acc, mt, bp, bl, bq, brem, p.bias_T, cb, vo, rstep);
            piece_barriers(mt);
          }
        } else {
#pragma unroll
          for (int mt = 0; mt < TM; ++mt) {
            epi0_block<T, ACT, TM, false>(acc, mt, bp, bl, bq, brem, p.bias_T, cb, vo, rstep);
            piece_barriers(mt);
          }
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
        piece_barriers(mt);
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
          piece_barriers(mt);
        }
        return (st && interior) ? 8 * TM : 0;
      } else {
        // fp32 + residual (the residual may BE the output): gemm16p_kernel's 4-slot piece pipeline, the residual loads of the next
        // pieces in flight across the piece barriers
        const char* rb = (const char*)(p.residual + (long long)m0 * p.ldr + n0);
        const unsigned rstep = (unsigned)p.ldr * 4;
        unsigned ro = opaque((unsigned)(4 * lg) * rstep + li * 16);
        int lrow = row;
        f32x4 rr[4][4];
        auto issue = [&](int j) __attribute__((always_inline)) {
#pragma unroll
          for (int r = 0; r < 4; ++r)
            rr[j & 3][r] = (interior || lrow + r < p.M) ? *reinterpret_cast<const f32x4*>(rb + ro + (unsigned)r * rstep + (j & 1) * 256)
                                                        : f32x4{0.f, 0.f, 0.f, 0.f};
          if (j & 1) { ro += 16 * rstep; lrow += 16; }
        };
        issue(0); issue(1); issue(2); issue(3);
        int bqr[4] = {0, 0, 0, 0};
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
          if (half) piece_barriers(mt);
        }
        return (st && interior) ? 8 * TM : 0;
      }
    }
  };

  // ---- the group's walk over its tiles.  Slab k of the walk's slab stream (global index G) lives in A stage G % NA of the group's
  // ring and W stage G % NW; the counters below are those remainders for the slab the next LOAD reads.
  int L = blockIdx.x;
  int tm, tn;
  tile_of(L, tm, tn);
  unsigned ao[2], aon[2] = {0u, 0u};
  a_offsets(tm * TROWS + grp * GROWS, ao);
  dma_bias(tm, tn, 0);
  // prologue: slabs 0 .. D of the group's A rows, and its W halves of slabs 0 .. WOFF (every stage is free; nk >= 8 > S + D)
#pragma unroll
  for (int s = 0; s <= S + D; ++s) {
    if (s <= D) dma_a(ao, s, s);
    if (s <= WOFF) dma_w(tn, s, s);
  }
  // group 0: slab 0's pieces (and the bias rows, older) have landed, D slabs stay in flight; group 1 (D + 1 A slabs + S + D + 1 W
  // halves, whose first readers are group 0's LOAD(0 ..)) waits for all of them once, here
  if (g1) wait_vmcnt<0>();
  else wait_vmcnt<4 * D>();
  __builtin_amdgcn_s_barrier();
  zero_acc();
  int seq = 0, sx = 0;
  // timeline instrumentation (tuning runs, mer_set_debug_buffer; schedules with LDS to spare): lane 0 of waves 0 and 4 stamps s_memtime into
  // LDS — slots: 0 tile start, 1-8 past the mid barrier of slabs 0-7, 9 K loop done, 10 boundary pieces issued, 11.. past epilogue piece barriers
  unsigned long long* stl = reinterpret_cast<unsigned long long*>(smem + Q::BIAS0 + P_BIAS);
  auto stamp = [&](int slot) __attribute__((always_inline)) {
    if constexpr (Q::STAMPS) {
      if (p.dbg && wg == 0 && seq < 12 && lane == 0) stl[(grp * 12 + seq) * 16 + slot] = __builtin_amdgcn_s_memtime();
    }
  };
  if constexpr (Q::STAMPS) {
    if (p.dbg) {
      for (int i = tid; i < P_STAMP / 8; i += 512) stl[i] = 0;
      __builtin_amdgcn_s_barrier();
    }
  }
  int sa = 0, sw = 0;                    // A / W stage of the slab the next LOAD reads
  int ia = D % NA, iw = WOFF % NW;       // A / W stage the next LOAD's DMA fills (slab + D / slab + WOFF)
  auto bump = [](int& v, int n) __attribute__((always_inline)) { v = v + 1 == n ? 0 : v + 1; };
  if (g1) {
    for (int i = 0; i < Q::LAG; ++i) __builtin_amdgcn_s_barrier();   // group 1 runs LAG phases behind, from here to the end of the walk
  }

  for (;;) {
    const int Ln = L + (int)gridDim.x;
    const bool has_next = Ln < nblk;
    int ntm = 0, ntn = 0;
    if (has_next) {
      tile_of(Ln, ntm, ntn);
      a_offsets(ntm * TROWS + grp * GROWS, aon);
    }
    stamp(0);
#pragma clang loop unroll(disable)
    for (int kt = 0; kt < nk; ++kt) {
      // LOAD(kt): A slab kt + D into the stage of the group's slab kt - 1, this group's half of W slab kt + WOFF into the stage of
      // slab kt + WOFF - NW (iteration 0's pieces went out at the tile boundary, in front of the stores)
      if (kt >= 1) {
        const int ka = kt + D, kw = kt + WOFF;
        if (ka < nk) dma_a(ao, ka, ia);
        else if (has_next) {
          if (ka == nk) dma_bias(ntm, ntn, (seq + 1) & 1);
          dma_a(aon, ka - nk, ia);
        }
        if (kw < nk) dma_w(tn, kw, iw);
        else if (has_next) dma_w(ntn, kw - nk, iw);
      }
      load_frags(sa, sw);
      // this wave's pieces of the group's next slab have landed; D - 1 younger LOADs' pieces — and, in a tile's first D iterations, the
      // stores of the last epilogue — stay in flight.  The last tile's tail issues fewer pieces: wait for everything.
      if (has_next || kt + WOFF < nk) {
        if (kt < D) wait_vmcnt_plus<4 * (D - 1), TM>(sx);
        else wait_vmcnt<4 * (D - 1)>();
      } else wait_vmcnt<0>();
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();         // mid
      if (kt < 8) stamp(1 + kt);
      __builtin_amdgcn_s_setprio(1);
      math();
      __builtin_amdgcn_s_setprio(0);
      __builtin_amdgcn_s_barrier();         // end
      bump(sa, NA); bump(ia, NA); bump(sw, NW); bump(iw, NW);
    }
    // tile boundary: LOAD(0)'s pieces of the next tile first (their stages were released by the barrier above), then the epilogue in
    // NP pieces, each closed by a barrier: the other group is in its LOAD / MATH phases meanwhile
    stamp(9);
    if (has_next) {
      dma_a(aon, D, ia);
      dma_w(ntn, WOFF, iw);
    }
    stamp(10);
    stamp_seq = seq;
    __builtin_amdgcn_sched_barrier(0);
    sx = epilogue(tm, tn, seq & 1);
    __builtin_amdgcn_sched_barrier(0);
    ++seq;
    if (!has_next) break;
    zero_acc();
    __builtin_amdgcn_sched_barrier(0);
    L = Ln; tm = ntm; tn = ntn;
    ao[0] = aon[0]; ao[1] = aon[1];
  }
  if (!g1) {
    for (int i = 0; i < Q::LAG; ++i) __builtin_amdgcn_s_barrier();   // group 1's last LAG phases
  }
  if constexpr (Q::STAMPS) {
    if (p.dbg) {
      __syncthreads();
      for (int i = tid; i < P_STAMP / 8; i += 512) p.dbg[(long long)blockIdx.x * (P_STAMP / 8) + i] = stl[i];
    }
  }
}

template <typename T, int EPI, int TM, int D, int S>
static int launch_q_act(const Gemm16Params& p, dim3 grid, hipStream_t st) {
  dim3 block(512, 1, 1);
  if constexpr (EPI == 0) {
    switch (p.act) {
      case MER_ACT_GELU: hipLaunchKernelGGL((gemm16q_kernel<T, EPI, MER_ACT_GELU, TM, D, S>), grid, block, 0, st, p); break;
      case MER_ACT_QUICK_GELU: hipLaunchKernelGGL((gemm16q_kernel<T, EPI, MER_ACT_QUICK_GELU, TM, D, S>), grid, block, 0, st, p); break;
      case MER_ACT_GELU_TANH: hipLaunchKernelGGL((gemm16q_kernel<T, EPI, MER_ACT_GELU_TANH, TM, D, S>), grid, block, 0, st, p); break;
      default: hipLaunchKernelGGL((gemm16q_kernel<T, EPI, MER_ACT_NONE, TM, D, S>), grid, block, 0, st, p); break;
    }
  } else if constexpr (EPI == 1) {
    if (p.act == MER_ACT_GELU) hipLaunchKernelGGL((gemm16q_kernel<T, EPI, MER_ACT_GELU, TM, D, S>), grid, block, 0, st, p);
    else hipLaunchKernelGGL((gemm16q_kernel<T, EPI, MER_ACT_NONE, TM, D, S>), grid, block, 0, st, p);
  } else {
    hipLaunchKernelGGL((gemm16q_kernel<T, EPI, MER_ACT_NONE, TM, D, S>), grid, block, 0, st, p);
  }
  return check_launch("gemm16q");
}

template <typename T, int TM, int D, int S>
int dispatch_q_impl(const Gemm16Params& p0, hipStream_t st) {
  Gemm16Params p = p0;
  p.tiles_m = (int)cdiv(p.M, 64 * TM);
  p.tiles_n = p.N / 256;
  const int nblk = p.tiles_m * p.tiles_n;
  const int cus = device_cu_count();
  dim3 grid(nblk < cus ? nblk : cus, 1, 1);
  const double mn = (double)p.M * p.N;
  ProfScope prof("gemm16p", 2.0 * mn * p.K,   // (the persistent GEMM's label: one launch set for bench.py whichever schedule runs)
                 2.0 * (double)p.M * p.K + 2.0 * (double)p.N * p.K + mn * ((p.c32 ? 4 : 0) + (p.c16_hi ? 2 : 0) + (p.residual ? 4 : 0)), st);
  if (p.c16_hi) return launch_q_act<T, 0, TM, D, S>(p, grid, st);
  if (p.residual) return launch_q_act<T, 2, TM, D, S>(p, grid, st);
  return launch_q_act<T, 1, TM, D, S>(p, grid, st);
}

// one translation unit per (dtype, rows, schedule): gemm16q_*.hip
template <typename T, int TM, int D, int S> int dispatch_q_cfg(const Gemm16Params& p, hipStream_t st);

extern int g_gemm_q_cfg;   // "gemm_q_cfg": 0 = (D 3, S 1), 1 = (D 2, S 1), 2 = (D 2, S 2)

template <typename T, int TM>
int dispatch_q_tm(const Gemm16Params& p, hipStream_t st) {
  switch (g_gemm_q_cfg) {
    case 1: return dispatch_q_cfg<T, TM, 2, 1>(p, st);
    case 2: return dispatch_q_cfg<T, TM, 2, 2>(p, st);
    default: return dispatch_q_cfg<T, TM, 3, 1>(p, st);
  }
}

template <typename T> int dispatch_q(const Gemm16Params& p, hipStream_t st);

}  // namespace mer
