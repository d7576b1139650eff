// gemm16p_impl.h — the PERSISTENT one-pass 256x256 GEMM: C = epi(A * W^T + bias) on v_mfma_f32_16x16x32_{f16,bf16}.
//
// Same K loop as gemm16_kernel<T,256,256,32,2,4,1,1,GLDS,4> (gemm16_impl.h: 4-stage LDS ring filled by LDS-DMA, counted vmcnt
// across raw s_barriers, two wave groups one barrier phase apart) and bit-identical results, but what surrounds the loop is
// rebuilt, because that is where 30-45 % of a K = 768 tile went (profiles/r02_gemm16_bench_epilogue_split.txt: 976 TF as shipped,
// 1398 TF with the epilogue skipped; s_memtime: prologue 3.6 k + K loop 32.1 k + epilogue 9.7-13 k (32 k for fp32 + residual)
// + 1.5-4.2 k hand-over cycles per tile):
//
//   * REGISTER-DIRECT EPILOGUE.  The pre-blocked weight plane is packed with its rows permuted inside every 64-row block
//     (mer_w_block_pack_p: LDS row 16 nt + li holds W row 4 li + nt), so that the four accumulators acc[mt][0..3][r] of lane
//     (li, lg) are four CONSECUTIVE output columns 4 li .. 4 li + 3 of row 16 mt + 4 lg + r.  Sixteen lanes then cover a wave's
//     whole 64-column run: a 16-bit row run is one 128-byte line (global_store_dwordx2), an fp32 one two lines
//     (global_store_dwordx4 / the residual's global_load_dwordx4).  No LDS transposition, no ds_write_b32, no staging tile that
//     aliases the ring — so the ring is free the moment the K loop ends.
//   * PERSISTENT WORKGROUPS, PROLOGUE BEFORE STORES.  One workgroup per CU walks tiles L, L + grid, ... (the XCD-aware map of
//     gemm16_kernel).  When a tile's K loop ends each wave first issues the NEXT tile's whole ring (4 slabs + the bias row, by
//     LDS-DMA) and only then its epilogue's stores: vmcnt retires in order, so the next K loop's counted waits name exactly how
//     many younger operations (stores included) may stay in flight — the stores drain under the next tile's first three slabs
//     instead of in front of them, and the next tile's prologue latency hides behind the epilogue.
//   * THE STAGGER RUNS ACROSS TILES.  Group 0 (waves 0-3) finishes a tile one phase early and does its epilogue while group 1
//     (waves 4-7, the other wave of every SIMD) issues its last 32 MFMAs; group 1's epilogue runs at the top of the next tile's
//     first iteration beside group 0's first MATH phase.
//
// Scope (anything else keeps gemm16_kernel): one pass, nbatch == 1, pre-blocked permuted W, N % 256 == 0, K % 32 == 0, K >= 256,
// planes < 4 GB, output either one 16-bit plane (EPI 0) or fp32 (EPI 1) (+ residual, EPI 2).
#pragma once
#include "gemm16_impl.h"

namespace mer {

constexpr int P_RING = 4 * 32768;          // 4 stages x (A 16 KB + W 16 KB)
constexpr int P_BIAS = 2 * 8 * 1024;       // bias row of the tile (256 fp32), one private copy per wave, two tile parities
constexpr int P_STAMP = 2 * 12 * 16 * 8;   // timeline stamps (mer_set_debug_buffer): 2 wave groups x 12 tiles x 16 slots, dumped at exit
constexpr int P_SMEM = P_RING + P_BIAS + P_STAMP;

// counted wait with the epilogue's still-in-flight stores (sx = 0 or 32, wave-uniform) added to the allowance
template <int N>
__device__ __forceinline__ void wait_vmcnt_plus(int sx) {
  if (sx == 32) wait_vmcnt<N + 32>();
  else wait_vmcnt<N>();
}

__device__ __forceinline__ unsigned long long uniform64(unsigned long long v) {
  return ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(v >> 32)) << 32) |
         (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)v);
}
// stores / loads with a uniform 64-bit base (SGPR pair) + a 32-bit per-lane byte offset; invisible to hipcc's waitcnt pass
// (that is the point: it must not drain the queue for them) — completion is counted by hand
__device__ __forceinline__ void gstore8_nt_s(unsigned long long sbase, unsigned voff, u32x2 v) {
  asm volatile("global_store_dwordx2 %0, %1, %2 nt" :: "v"(voff), "v"(v), "s"(sbase) : "memory");
}
__device__ __forceinline__ void gstore16_s(unsigned long long sbase, unsigned voff, f32x4 v) {
  asm volatile("global_store_dwordx4 %0, %1, %2\n\ts_nop 1" :: "v"(voff), "v"(v), "s"(sbase) : "memory");
}
__device__ __forceinline__ void gload16_s(f32x4& dst, unsigned long long sbase, unsigned voff) {
  asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(dst) : "v"(voff), "s"(sbase) : "memory");
}

// hides a value's provenance from the optimiser at this point: loop-invariant code motion otherwise hoists the epilogue's 32 per-row
// offsets (they do not depend on the tile) out of the tile walk and keeps them — in scratch memory — across the K loop
__device__ __forceinline__ unsigned opaque(unsigned v) {
  asm volatile("" : "+v"(v));
  return v;
}

template <typename T, int EPI, int ACT>
__global__ __launch_bounds__(512) void gemm16p_kernel(const Gemm16Params p) {
  typedef typename T16<T>::v8 v8;
  constexpr int TM = 8, TN = 4;
  constexpr int STAGE = 32768, A_PLANE = 16384;
  __shared__ __attribute__((aligned(16))) char smem[P_SMEM];

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;
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
  unsigned a_o32[2];
  const unsigned w_o32 = (unsigned)((ld_row0 * 32 + ld_ch * 8) * 2);   // second chunk: + 8192 (row + 128), folded into the base
  int cur_tn = 0;        // column tile whose W blocks the DMA is reading
  auto setup_a = [&](int m0_) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int rr = ld_row0 + i * 128;
      int m = m0_ + rr;
      m = m < p.M ? m : p.M - 1;
      const long long off = (p.a_rpb > 0) ? (long long)(m / p.a_rpb) * p.a_bstride + (long long)(m % p.a_rpb) * p.lda : (long long)m * p.lda;
      a_o32[i] = (unsigned)((off + ((ld_ch ^ swz_of<4>(rr)) << 3)) * 2);
    }
  };
  const char* a_plane = (const char*)p.a_hi;
  const char* w_plane = (const char*)p.w_hi;
  auto glds_slab = [&](int kt, int stage) __attribute__((always_inline)) {
    const char* ab = a_plane + (long long)kt * 64;
    const char* wb = w_plane + ((long long)cur_tn * nk + kt) * 16384;
    const unsigned lb = lds0 + stage * STAGE + wave * 1024;
    dma16_sbase(ab, a_o32[0], lb);
    dma16_sbase(ab, a_o32[1], lb + 8192);
    dma16_sbase(wb, w_o32, lb + A_PLANE);
    dma16_sbase(wb + 8192, w_o32, lb + A_PLANE + 8192);
  };
  // the whole ring of a tile + its bias row (the oldest of the 17 operations: every counted wait below covers it)
  auto issue_ring = [&](int tn_, int m0_, int parity) __attribute__((always_inline)) {
    cur_tn = tn_;
    setup_a(m0_);
    if (p.bias) dma16_sbase((const char*)(p.bias + tn_ * 256), (unsigned)(lane * 16), lds0 + P_RING + parity * 8192 + wave * 1024);
#pragma unroll
    for (int s = 0; s < 4; ++s) glds_slab(s, s);
  };

  f32x4 acc[TM][TN];
  auto zero_acc = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  };

  v8 af[TM], wf[TN];
  // (row >> 2) & 3 == (li >> 2) for every fragment row of this lane: one swizzle term, fragments 1 KiB apart
  const int fsw = ((lg ^ ((-(li >> 2)) & 3)) << 4);
  const int a_f0 = (wm * 128 + li) * 64 + fsw;
  const int w_f0 = A_PLANE + (wn * 64 + li) * 64 + fsw;
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

  // ---- epilogue of tile (tm_, tn_), bias row in parity slot `par`; returns the store allowance for the next counted waits
  auto epilogue = [&](int tm_, int tn_, int par) __attribute__((always_inline)) -> int {
    if ((p.dbg_skip & 3) == 2) return 0;
    const int m0 = tm_ * 256 + wm * 128, n0 = tn_ * 256 + wn * 64;
    f32x4 bq = {0.f, 0.f, 0.f, 0.f};
    if (p.bias) bq = *reinterpret_cast<const f32x4*>(smem + P_RING + par * 8192 + wave * 1024 + (wn * 64 + li * 4) * 4);
    const bool interior = tm_ * 256 + 256 <= p.M;   // (N % 256 == 0: no column edge)
    if (interior) {
      // Plain (compiler-scheduled) loads and stores: hipcc counts its own memory operations exactly (vmcnt(N) names the N youngest
      // operations that may stay in flight, so the LDS-DMA requests it knows nothing about — all older — do not disturb its
      // counts), and hand-written asm loads are not safe here: the register allocator copied their destination registers before
      // the data had landed.  What the K loop needs from this function is only an upper bound on the stores it leaves in flight.
      const bool st = (p.dbg_skip & 3) != 1;
      if constexpr (EPI == 0) {
        char* cb = (char*)((T*)p.c16_hi + (long long)m0 * p.ldc16 + n0);
        const unsigned rstep = (unsigned)p.ldc16 * 2;
        unsigned vo = opaque((unsigned)(4 * lg) * rstep + li * 8);
#pragma unroll
        for (int mt = 0; mt < TM; ++mt) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            typename T16<T>::v4 h;
#pragma unroll
            for (int nt = 0; nt < TN; ++nt) h[nt] = T16<T>::from_f32(act_apply(acc[mt][nt][r] + bq[nt], ACT));
            if (st) __builtin_nontemporal_store(__builtin_bit_cast(u32x2, h), reinterpret_cast<u32x2*>(cb + vo));
            vo += rstep;
          }
          vo += 12 * rstep;
        }
        return st ? 32 : 0;
      } else {
        char* cb = (char*)(p.c32 + (long long)m0 * p.ldc32 + n0);
        const unsigned cstep = (unsigned)p.ldc32 * 4;
        unsigned vo = opaque((unsigned)(4 * lg) * cstep + li * 16);
        if constexpr (EPI == 1) {
#pragma unroll
          for (int mt = 0; mt < TM; ++mt) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              f32x4 v;
#pragma unroll
              for (int nt = 0; nt < TN; ++nt) v[nt] = act_apply(acc[mt][nt][r] + bq[nt], ACT);
              if (st) *reinterpret_cast<f32x4*>(cb + vo) = v;
              vo += cstep;
            }
            vo += 12 * cstep;
          }
          return st ? 32 : 0;
        } else {
          // residual rows two row tiles (8 loads) ahead of the stores; the residual may BE the output (the pre-LN stream is updated
          // in place), so the compiler keeps this source order: loads of row tile mt + 2 are issued before the stores of row tile mt
          const char* rb = (const char*)(p.residual + (long long)m0 * p.ldr + n0);
          const unsigned rstep = (unsigned)p.ldr * 4;
          unsigned ro = opaque((unsigned)(4 * lg) * rstep + li * 16);
          f32x4 rr[3][4];
          auto issue = [&](int slot) __attribute__((always_inline)) {
#pragma unroll
            for (int r = 0; r < 4; ++r) { rr[slot][r] = *reinterpret_cast<const f32x4*>(rb + ro); ro += rstep; }
            ro += 12 * rstep;
          };
          issue(0);
          issue(1);
#pragma unroll
          for (int mt = 0; mt < TM; ++mt) {
            if (mt + 2 < TM) issue((mt + 2) % 3);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              f32x4 v;
#pragma unroll
              for (int nt = 0; nt < TN; ++nt) v[nt] = act_apply(acc[mt][nt][r] + bq[nt], ACT) + rr[mt % 3][r][nt];
              if (st) *reinterpret_cast<f32x4*>(cb + vo) = v;
              vo += cstep;
            }
            vo += 12 * cstep;
          }
          return st ? 32 : 0;
        }
      }
    }
    // last row tile of a plane whose M is not a multiple of 256: predicated, compiler-scheduled accesses, no allowance
    {
      const bool st = (p.dbg_skip & 3) != 1;
      char* cb = EPI == 0 ? (char*)((T*)p.c16_hi + (long long)m0 * p.ldc16 + n0) : (char*)(p.c32 + (long long)m0 * p.ldc32 + n0);
      const unsigned cstep = EPI == 0 ? (unsigned)p.ldc16 * 2 : (unsigned)p.ldc32 * 4;
      unsigned vo = opaque((unsigned)(4 * lg) * cstep + li * (EPI == 0 ? 8 : 16));
      const char* rbp = EPI == 2 ? (const char*)(p.residual + (long long)m0 * p.ldr + n0) : nullptr;
      const unsigned rstep = EPI == 2 ? (unsigned)p.ldr * 4 : 0;
      unsigned ro = opaque((unsigned)(4 * lg) * rstep + li * 16);
      int row = opaque((unsigned)(m0 + 4 * lg));
#pragma unroll
      for (int mt = 0; mt < TM; ++mt) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          if (row + r < p.M && st) {
            f32x4 v;
#pragma unroll
            for (int nt = 0; nt < TN; ++nt) v[nt] = act_apply(acc[mt][nt][r] + bq[nt], ACT);
            if constexpr (EPI == 0) {
              typename T16<T>::v4 h;
#pragma unroll
              for (int nt = 0; nt < TN; ++nt) h[nt] = T16<T>::from_f32(v[nt]);
              *reinterpret_cast<u32x2*>(cb + vo) = __builtin_bit_cast(u32x2, h);
            } else {
              if constexpr (EPI == 2) v += *reinterpret_cast<const f32x4*>(rbp + ro);
              *reinterpret_cast<f32x4*>(cb + vo) = v;
            }
          }
          vo += cstep;
          ro += rstep;
        }
        vo += 12 * cstep;
        ro += 12 * rstep;
        row += 16;
      }
    }
    return 0;
  };

  // ---- tile walk ----
  int L = blockIdx.x;
  int tm, tn;
  tile_of(L, tm, tn);
  issue_ring(tn, tm * 256, 0);
  wait_vmcnt<12>();                 // slab 0 (and the bias row) of this wave's share
  __builtin_amdgcn_s_barrier();     // X: ... of every wave's
  zero_acc();
  int seq = 0;                      // tiles done by this workgroup: parity of the bias slot
  int sx = 0;                       // stores of the last epilogue that the next counted waits may leave in flight
  int ptm = 0, ptn = 0;             // group 1: the tile whose accumulators it still holds
  bool have_prev = false;
  // timeline instrumentation (tuning runs only): lane 0 of waves 0 and 4 stamps s_memtime into LDS (no VMEM traffic that would
  // disturb the counted waits), slots: 0 tile start, 1-8 past the mid barrier of slabs 0-7, 9 K loop done, 10 next ring issued,
  // 11 epilogue done, 12 past X' ; group 1: 13 / 14 around its in-loop epilogue
  unsigned long long* stl = reinterpret_cast<unsigned long long*>(smem + P_RING + P_BIAS);
  auto stamp = [&](int slot) __attribute__((always_inline)) {
    if (p.dbg && (wave & 3) == 0 && seq < 12 && lane == 0) stl[((wave >> 2) * 12 + seq) * 16 + slot] = __builtin_amdgcn_s_memtime();
  };
  if (p.dbg) {
    for (int i = tid; i < P_STAMP / 8; i += 512) stl[i] = 0;
    __builtin_amdgcn_s_barrier();
  }

  for (;;) {
    const int Ln = L + (int)gridDim.x;
    const bool has_next = Ln < nblk;
    int ntm = 0, ntn = 0;
    if (has_next) tile_of(Ln, ntm, ntn);
    if (g1) __builtin_amdgcn_s_barrier();   // A0: group 1 runs one phase behind
    stamp(0);
    for (int kt = 0; kt < nk; ++kt) {
      if (kt >= 1 && kt + 3 < nk) glds_slab(kt + 3, (kt + 3) & 3);   // into the stage slab kt - 1 was read from
      if (kt == 0 && g1) {                  // group 1's epilogue of the previous tile, beside group 0's first MATH phase
        __builtin_amdgcn_sched_barrier(0);
        stamp(13);
        if (have_prev) sx = epilogue(ptm, ptn, (seq + 1) & 1);
        stamp(14);
        __builtin_amdgcn_sched_barrier(0);   // the fresh accumulators must not be live beside the ones being stored
        zero_acc();
        __builtin_amdgcn_sched_barrier(0);
      }
      load_frags(kt & 3);
      // this wave's share of slab kt + 1 has landed (two younger slabs — and, early in a tile, the stores — stay in flight)
      if (kt + 3 < nk) {
        if (kt < 3) wait_vmcnt_plus<8>(sx);
        else wait_vmcnt<8>();
      } else {
        wait_vmcnt<0>();
        // ... and tell hipcc's scoreboard so: it still believes the previous epilogue's stores (and loads on paths not taken) are in
        // flight and would otherwise protect their registers with a near-zero vmcnt in front of the next epilogue — i.e. right
        // behind the ring issue, waiting for 16 KiB of DMA to land before the first store
        __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0) expcnt(7) lgkmcnt(15)
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();         // mid
      if (kt < 8) stamp(1 + kt);
      if (kt == nk - 1 && g1 && has_next) issue_ring(ntn, ntm * 256, (seq + 1) & 1);   // every LDS read of this tile is over
      __builtin_amdgcn_s_setprio(1);
      math();
      __builtin_amdgcn_s_setprio(0);
      if (kt == nk - 1 && g1) {
        if (has_next) {
          wait_vmcnt<12>();                 // slab 0 of the next tile (no stores of this group in flight yet)
          __builtin_amdgcn_s_barrier();     // X'
        }
      } else __builtin_amdgcn_s_barrier();  // end
    }
    stamp(9);
    if (!g1 || !has_next) {   // group 0: every tile, right behind its last MATH phase; group 1: only the last tile's (one call site less)
      if (has_next) issue_ring(ntn, ntm * 256, (seq + 1) & 1);
      stamp(10);
      __builtin_amdgcn_sched_barrier(0);
      sx = epilogue(tm, tn, seq & 1);
      __builtin_amdgcn_sched_barrier(0);
      stamp(11);
      if (has_next) {
        zero_acc();
        wait_vmcnt_plus<12>(sx);
        __builtin_amdgcn_s_barrier();       // X'
      }
      stamp(12);
    } else {
      ptm = tm; ptn = tn; have_prev = true;
      sx = 0;
    }
    ++seq;
    if (!has_next) break;
    L = Ln; tm = ntm; tn = ntn;
  }
  if (p.dbg) {
    __syncthreads();
    for (int i = tid; i < P_STAMP / 8; i += 512) p.dbg[(long long)blockIdx.x * (P_STAMP / 8) + i] = stl[i];
  }
}

int device_cu_count();

template <typename T, int EPI>
static int launch_p_act(const Gemm16Params& p, dim3 grid, hipStream_t st) {
  dim3 block(512, 1, 1);
  if (EPI == 0) {
    switch (p.act) {
      case MER_ACT_GELU: hipLaunchKernelGGL((gemm16p_kernel<T, EPI, MER_ACT_GELU>), grid, block, 0, st, p); break;
      case MER_ACT_QUICK_GELU: hipLaunchKernelGGL((gemm16p_kernel<T, EPI, MER_ACT_QUICK_GELU>), grid, block, 0, st, p); break;
      case MER_ACT_GELU_TANH: hipLaunchKernelGGL((gemm16p_kernel<T, EPI, MER_ACT_GELU_TANH>), grid, block, 0, st, p); break;
      default: hipLaunchKernelGGL((gemm16p_kernel<T, EPI, MER_ACT_NONE>), grid, block, 0, st, p); break;
    }
  } else {
    if (p.act == MER_ACT_GELU) hipLaunchKernelGGL((gemm16p_kernel<T, EPI, MER_ACT_GELU>), grid, block, 0, st, p);
    else hipLaunchKernelGGL((gemm16p_kernel<T, EPI, MER_ACT_NONE>), grid, block, 0, st, p);
  }
  return check_launch("gemm16p");
}

// eligibility was checked by mer_gemm16 (gemm16.hip: persistent_ok)
template <typename T>
int dispatch_p_impl(const Gemm16Params& p0, hipStream_t st) {
  Gemm16Params p = p0;
  p.tiles_m = (int)cdiv(p.M, 256);
  p.tiles_n = p.N / 256;
  const int nblk = p.tiles_m * p.tiles_n;
  const int cus = device_cu_count();
  dim3 grid(nblk < cus ? nblk : cus, 1, 1);
  const double mn = (double)p.M * p.N;
  ProfScope prof("gemm16", 2.0 * mn * p.K,
                 2.0 * (double)p.M * p.K + 2.0 * (double)p.N * p.K + mn * ((p.c32 ? 4 : 0) + (p.c16_hi ? 2 : 0) + (p.residual ? 4 : 0)), st);
  if (p.c16_hi) return launch_p_act<T, 0>(p, grid, st);
  if (p.residual) return launch_p_act<T, 2>(p, grid, st);
  return launch_p_act<T, 1>(p, grid, st);
}

template <typename T> int dispatch_p(const Gemm16Params& p, hipStream_t st);

}  // namespace mer
