// attention.hip — softmax(q k^T * scale) v for head_dim 64 on the 16x16x32 f16/bf16 MFMA.
// (HF:hubert/modeling_hubert.py:236-259, HF:clip/modeling_clip.py:280-335,
//  HF:roberta/modeling_roberta.py:186-250 — identical math, no causal mask on this path.)
//
// Single-pass kernel (T <= 512): one workgroup = (batch b, head h, 64 queries); the whole K
// [T,64] and V^T [64,T] of that head live in LDS, every wave owns 16 queries.
//   S^T = K Q^T   is computed "swapped" (A = K rows, B = Q^T) so that a lane holds, for ONE query
//                 (lane & 15), the scores of keys {16*kt + 4*(lane>>4) + r}: the row max / sum are
//                 in-lane reductions plus two cross-lane steps (xor 16, 32), no LDS round trip;
//   O^T = V^T P^T the exponentiated scores of two neighbouring key tiles are already, lane for
//                 lane, a valid MFMA B fragment when the 32-key k-slice is enumerated as
//                 (tile 2c: keys 4g..4g+3 | tile 2c+1: keys 4g..4g+3); V^T is read from LDS with the
//                 same enumeration (two ds_read_b64), so P never leaves registers.
// K and V are both staged ROW-major (rows padded to 144 B, 16-byte LDS writes); the V^T fragments of the second MFMA are
// produced by the LDS transpose read of gfx950, ds_read_b64_tr_b16: within each 16-lane block the lanes hand in the
// addresses of a 4-row x 16-column patch (lane j: row j/4, columns 4*(j%4)..+3) and lane j' receives column j' of the four
// rows (measured: scripts/probes/tr_probe.py) — exactly "feature d = li, keys 4*lg .. 4*lg+3" of the fragment.  (Writing V^T
// with eight 2-byte LDS stores per 16-byte load was a third of this kernel.)  Scores, softmax and 1/sum are fp32.
#include "common.h"

namespace mer {

extern unsigned long long* g_gemm_dbg;
// shared debug-stamp buffer (mer_set_debug_buffer)

// BIAS: scores get an additive term gate[b,h,q] * bias[h,q,k] before the softmax — WavLM's gated relative position bias
// (HF:wavlm/modeling_wavlm.py WavLMAttention.forward: one [H,T,T] table shared by the batch and by all layers, a per-query
// gate per layer) and, with gate == NULL, BEiT / data2vec-vision's relative position bias.  A lane owns ONE query row, so
// its four keys of a tile are 16 contiguous bytes of that row of the table (rows padded to ldb % 4 == 0).
// LDS transpose read (ds_read_b64_tr_b16): four 16-bit values, see the header comment
typedef short i16x4_t __attribute__((ext_vector_type(4)));
template <typename T>
__device__ __forceinline__ typename T16<T>::v4 tr_read4(const T* lds_ptr) {
  const i16x4_t r = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) i16x4_t*)lds_ptr);
  return __builtin_bit_cast(typename T16<T>::v4, r);
}

// NW waves per workgroup share one staged K / V: 8 waves (NKT <= 16: the scores fit 128 VGPRs) give a CU 16 resident waves
// on the same LDS footprint as 4 — the per-sub-tile chain (Q load -> 2 NKT MFMAs -> softmax -> 2 NKT MFMAs -> store) is
// latency-bound, and a head's 13-16 sub-tiles take 2 rounds of a wave instead of 4.
// (NW = 8 without the bias table asks for 4 waves per SIMD = two resident workgroups per CU: the register allocator must stay
// within 128 VGPRs; the bias variants need ~140 and would spill.)
template <typename T, int NKT, bool BIAS = false, int NW = 4>
__global__ __launch_bounds__(NW * 64) __attribute__((amdgpu_waves_per_eu((NW == 8 && !BIAS) ? 4 : 1))) void attn_sp_kernel(const T* __restrict__ q, const T* __restrict__ k,
                                                      const T* __restrict__ v, long long ld, T* oh, T* ol,
                                                      long long ldo, int Tn, float scale_log2e, const int* kv_len, int hm, unsigned long long* dbg,
                                                      const float* __restrict__ bias = nullptr, long long ldb = 0,
                                                      const float* __restrict__ gate = nullptr) {
  typedef typename T16<T>::v8 v8;
  typedef typename T16<T>::v4 v4;
  constexpr int TP = NKT * 16;
  constexpr int KS = 72;       // K row stride in elements (144 B)
  __shared__ __attribute__((aligned(16))) T Ks[TP * KS];
  __shared__ __attribute__((aligned(16))) T Vs[TP * KS];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 15, lg = lane >> 4;
  const int b = blockIdx.z, h = blockIdx.y, qt = blockIdx.x;
  const long long dbi = ((long long)blockIdx.z * gridDim.y + blockIdx.y) * 4;
  if (dbg && tid == 0) dbg[dbi] = __builtin_amdgcn_s_memtime();
  int klen = Tn;
  if (kv_len) {
    klen = kv_len[b];
    klen = klen < Tn ? klen : Tn;
  }
  const long long row0 = (long long)b * Tn;
  // row-major: head h lives at column 64h of every [B*T, ld] row; head-major (hm): [B][H][T][64], row stride 64
  const long long hbase = hm ? ((long long)b * gridDim.y + h) * Tn * 64 : row0 * ld + h * 64;
  if (hm) ld = 64;
  const T* kb = k + hbase;
  const T* vb = v + hbase;
  const T* qb = q + hbase;

  // ---- stage K (row-major, padded) and V^T into LDS; rows >= klen are zero.  Loads are issued SG chunks at a time
  // ahead of their LDS writes: one memory round trip per SG*32 rows instead of one per 32 rows. ----
  constexpr int NT = NW * 64, SIT = (TP * 8 + NT - 1) / NT, SG = 4;
  for (int g0 = 0; g0 < SIT; g0 += SG) {
    u32x4 kreg[SG], vreg[SG];
#pragma unroll
    for (int it = 0; it < SG; ++it) {
      const int c = tid + (g0 + it) * NT;
      const int row = c >> 3, ch = c & 7;
      kreg[it] = u32x4{0u, 0u, 0u, 0u};
      vreg[it] = u32x4{0u, 0u, 0u, 0u};
      if (g0 + it < SIT && row < klen) {
        kreg[it] = *reinterpret_cast<const u32x4*>(kb + (long long)row * ld + ch * 8);
        vreg[it] = *reinterpret_cast<const u32x4*>(vb + (long long)row * ld + ch * 8);
      }
    }
#pragma unroll
    for (int it = 0; it < SG; ++it) {
      const int c = tid + (g0 + it) * NT;
      const int row = c >> 3, ch = c & 7;
      if (g0 + it < SIT && row < TP) {
        *reinterpret_cast<u32x4*>(Ks + row * KS + ch * 8) = kreg[it];
        *reinterpret_cast<u32x4*>(Vs + row * KS + ch * 8) = vreg[it];
      }
    }
  }

  __syncthreads();
  if (dbg && tid == 0) dbg[dbi + 1] = __builtin_amdgcn_s_memtime();

  // K / V^T of this (batch, head) are staged once; each wave then walks its 16-query sub-tiles
  // (qs = wave, wave+NW, ...), so the staging cost is paid once per head instead of once per 64 queries.
  for (int qs = qt * NW + wave; qs * 16 < Tn; qs += NW * (int)gridDim.x) {
  // K / V^T fragments are loop-invariant LDS reads: without this clobber hipcc hoists all of them out of the loop
  // (28 + 56 fragment registers -> 256 VGPR + ~90 AGPR, one workgroup per CU instead of two)
  asm volatile("" ::: "memory");
  // ---- this lane's query fragment (B operand: n = query li, k = d) ----
  const int qi = qs * 16 + li;
  const int qrow = qi < Tn ? qi : Tn - 1;
  v8 qf[2];
#pragma unroll
  for (int kk = 0; kk < 2; ++kk) qf[kk] = *reinterpret_cast<const v8*>(qb + (long long)qrow * ld + kk * 32 + lg * 8);

  // ---- S^T tiles: s[kt][r] = score(key = 16*kt + 4*lg + r, query = li) ----
  f32x4 s[NKT];
#pragma unroll
  for (int kt = 0; kt < NKT; ++kt) {
    f32x4 a = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      const v8 kf = *reinterpret_cast<const v8*>(Ks + (kt * 16 + li) * KS + kk * 32 + lg * 8);
      a = T16<T>::mfma(kf, qf[kk], a);
    }
    s[kt] = a;
  }

  // ---- softmax over keys (fp32, base-2 exponent) ----
  // The softmax is the VALU bulk of this kernel (the MFMAs of a sub-tile take ~900 clocks, the scalar-per-score code ~2500),
  // so every score costs as few issue slots as possible: the max is taken over the RAW scores (scale > 0 commutes with max),
  // scale and max fold into one packed FMA per two scores, the key mask only touches tiles that can hold keys >= klen.
  float mx = -INFINITY;
  float sum = 0.f;
  if (BIAS) {
    const float* brow = bias + ((long long)h * Tn + qrow) * ldb;
    float gq = 1.4426950408889634f;   // the bias is added in the base-2 domain too
    if (gate) gq *= gate[((long long)b * gridDim.y + h) * Tn + qrow];
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt) {
      f32x4 bv = {0.f, 0.f, 0.f, 0.f};
      if (kt * 16 + lg * 4 < Tn) bv = *reinterpret_cast<const f32x4*>(brow + kt * 16 + lg * 4);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int key = kt * 16 + lg * 4 + r;
        const float x = key < klen ? fmaf(gq, bv[r], s[kt][r] * scale_log2e) : -INFINITY;
        s[kt][r] = x;
        mx = fmaxf(mx, x);
      }
    }
    mx = fmaxf(mx, __shfl_xor(mx, 16));
    mx = fmaxf(mx, __shfl_xor(mx, 32));
    if (!(mx > -INFINITY)) mx = 0.f;  // klen == 0: all keys masked -> zeros out
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float pexp = __builtin_amdgcn_exp2f(s[kt][r] - mx);
        s[kt][r] = pexp;
        sum += pexp;
      }
  } else {
    // keys >= klen (their K rows are zero in LDS: raw score 0) must leave the max and the sum.  The dispatcher picks the
    // smallest even NKT >= ceil(T / 16), so with klen == T only the last two tiles can hold such keys; a shorter row of a
    // ragged batch (wave-uniform test) masks the earlier tiles too.  `kl` is opaque to the optimiser so that those
    // 4 (NKT - 2) compares stay inside the rare branch instead of being hoisted into (spilled) SGPR pairs.
    if (NKT > 2 && klen <= (NKT - 2) * 16) {
      int kl = __builtin_amdgcn_readfirstlane(klen);
      asm volatile("" : "+s"(kl));
#pragma unroll
      for (int kt = 0; kt < NKT - 2; ++kt)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (kt * 16 + lg * 4 + r >= kl) s[kt][r] = -INFINITY;
    }
#pragma unroll
    for (int kt = (NKT > 2 ? NKT - 2 : 0); kt < NKT; ++kt)
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (kt * 16 + lg * 4 + r >= klen) s[kt][r] = -INFINITY;
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
      for (int r = 0; r < 4; ++r) mx = fmaxf(mx, s[kt][r]);
    mx = fmaxf(mx, __shfl_xor(mx, 16));
    mx = fmaxf(mx, __shfl_xor(mx, 32));
    if (!(mx > -INFINITY)) mx = 0.f;  // klen == 0: all keys masked -> zeros out
    const f32x2 sc2 = {scale_log2e, scale_log2e};
    const f32x2 off2 = {-mx * scale_log2e, -mx * scale_log2e};
    f32x2 sum2 = {0.f, 0.f};
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt) {
      const f32x2 x01 = __builtin_elementwise_fma(f32x2{s[kt][0], s[kt][1]}, sc2, off2);   // v_pk_fma_f32
      const f32x2 x23 = __builtin_elementwise_fma(f32x2{s[kt][2], s[kt][3]}, sc2, off2);
      const f32x2 p01 = {__builtin_amdgcn_exp2f(x01[0]), __builtin_amdgcn_exp2f(x01[1])};
      const f32x2 p23 = {__builtin_amdgcn_exp2f(x23[0]), __builtin_amdgcn_exp2f(x23[1])};
      s[kt] = f32x4{p01[0], p01[1], p23[0], p23[1]};
      sum2 += p01;                                                                         // v_pk_add_f32
      sum2 += p23;
    }
    sum = sum2[0] + sum2[1];
  }
  sum += __shfl_xor(sum, 16);
  sum += __shfl_xor(sum, 32);
  const float inv = sum > 0.f ? 1.0f / sum : 0.f;

  // ---- O^T = V^T P^T ----
  f32x4 o[4];
#pragma unroll
  for (int dt = 0; dt < 4; ++dt) o[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int c = 0; c < NKT / 2; ++c) {
    v8 pf;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      pf[j] = T16<T>::from_f32(s[2 * c][j]);
      pf[4 + j] = T16<T>::from_f32(s[2 * c + 1][j]);
    }
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) {
      // transpose read: this lane's address = row (key) 32c + 4*lg + li/4, columns 16*dt + 4*(li%4) .. +3;
      // it receives V[32c + 4*lg + i][16*dt + li], i = 0..3 (and the same 16 keys further for the second half)
      const T* vr = Vs + ((2 * c) * 16 + lg * 4 + (li >> 2)) * KS + dt * 16 + (li & 3) * 4;
      const v4 v0 = tr_read4<T>(vr);
      const v4 v1 = tr_read4<T>(vr + 16 * KS);
      v8 vf;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        vf[j] = v0[j];
        vf[4 + j] = v1[j];
      }
      o[dt] = T16<T>::mfma(vf, pf, o[dt]);
    }
  }

  // ---- store: lane holds O[query li][d = 16*dt + 4*lg + r] ----
  if (qi < Tn) {
    const long long orow = (row0 + qi) * ldo + h * 64;
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) {
      v4 hh, ll;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        T a, c;
        split16<T>(o[dt][r] * inv, a, c);
        hh[r] = a;
        ll[r] = c;
      }
      *reinterpret_cast<v4*>(oh + orow + dt * 16 + lg * 4) = hh;
      if (ol) *reinterpret_cast<v4*>(ol + orow + dt * 16 + lg * 4) = ll;
    }
  }
  }  // q sub-tile loop
  if (dbg && tid == 0) dbg[dbi + 2] = __builtin_amdgcn_s_memtime();
}

// Streaming (online-softmax) kernel for T > 512 (VideoMAE: 1568 tokens).  One workgroup = (batch, head, 64 * QS
// queries); keys/values are walked in blocks of 64 through LDS.  Same swapped-operand trick as above: a lane owns one
// query, so the running max m / normaliser l and the rescale factor are per-lane scalars and apply directly to the
// lane's O^T accumulator columns (HF:videomae/modeling_videomae.py eager_attention_forward).
// QS = 16-query sub-tiles per wave: with QS = 2 every K fragment (ds_read_b128) and every V^T fragment (two transpose reads)
// pulled out of LDS feeds two MFMAs, and a key block staged through LDS serves 128 queries — half the L2 -> LDS staging
// traffic and half the LDS reads per FLOP of the QS = 1 form (which staged 401 KB of K / V per head 25 times at T = 1568).
// Staging (round 3): a two-slot LDS ring filled by LDS-DMA (global_load_lds, as the GEMM's loader): the loads of key block i + 1 are
// issued BEFORE block i is computed and cost no registers (round 2's register prefetch lost a resident wave per SIMD to its 28 VGPRs),
// and one barrier per key block is left.  A DMA piece is 1 KiB = 8 unpadded 128-byte rows, so bank conflicts are avoided by an XOR
// of the 16-byte chunk index applied on the SOURCE side: K rows with (row >> 1) & 7 (the ds_read_b128 lane groups of the score
// MFMAs: the GEMM's analysis), V rows with ((row >> 1) & 3) << 1 (the transpose reads touch chunk PAIRS of rows 4 lg + li / 4).
// Rows past klen are fetched from the last valid row (the score mask makes their P exactly 0).
typedef __attribute__((address_space(3))) void attn_lds_t;
typedef const __attribute__((address_space(1))) void attn_glb_t;
template <typename T, int QS>
__global__ __launch_bounds__(256) void attn_stream_kernel(const T* __restrict__ q, const T* __restrict__ k,
                                                          const T* __restrict__ v, long long ld, T* oh, T* ol,
                                                          long long ldo, int Tn, float scale_log2e, const int* kv_len, int hm) {
  typedef typename T16<T>::v8 v8;
  typedef typename T16<T>::v4 v4;
  constexpr int KB = 64, KS = 64;   // unpadded rows: the swizzle replaces the padding
  __shared__ __attribute__((aligned(1024))) T Ks[2][KB * KS];
  __shared__ __attribute__((aligned(1024))) T Vs[2][KB * KS];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 15, lg = lane >> 4;
  const int b = blockIdx.z, h = blockIdx.y, qt = blockIdx.x;
  int klen = Tn;
  if (kv_len) {
    klen = kv_len[b];
    klen = klen < Tn ? klen : Tn;
  }
  const long long row0 = (long long)b * Tn;
  // row-major: head h lives at column 64h of every [B*T, ld] row; head-major (hm): [B][H][T][64], row stride 64
  const long long hbase = hm ? ((long long)b * gridDim.y + h) * Tn * 64 : row0 * ld + h * 64;
  if (hm) ld = 64;
  const T* kb = k + hbase;
  const T* vb = v + hbase;
  const T* qb = q + hbase;
  int qi[QS];
  v8 qf[QS][2];
  f32x4 o[QS][4];
  float m[QS], lsum[QS];
#pragma unroll
  for (int u = 0; u < QS; ++u) {
    qi[u] = (qt * 4 + wave) * (16 * QS) + u * 16 + li;
    const int qrow = qi[u] < Tn ? qi[u] : Tn - 1;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) qf[u][kk] = *reinterpret_cast<const v8*>(qb + (long long)qrow * ld + kk * 32 + lg * 8);
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) o[u][dt] = f32x4{0.f, 0.f, 0.f, 0.f};
    m[u] = -INFINITY;
    lsum[u] = 0.f;
  }
  // LDS-DMA of one key block into slot `slot`: 8 pieces of 1 KiB per operand, two per wave; lane l of a piece lands on row
  // piece * 8 + l / 8, physical chunk l % 8, and fetches the logical chunk that the swizzle maps there
  const int d_row = lane >> 3, d_pc = lane & 7;
  auto issue = [&](int k0, int slot) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int piece = wave * 2 + i;
      const int row = piece * 8 + d_row;
      int kr = k0 + row;
      kr = kr < klen ? kr : klen - 1;
      const int kc = d_pc ^ ((row >> 1) & 7), vc = d_pc ^ (((row >> 1) & 3) << 1);
      __builtin_amdgcn_global_load_lds((attn_glb_t*)(kb + (long long)kr * ld + kc * 8), (attn_lds_t*)(Ks[slot] + piece * 8 * KS), 16, 0, 0);
      __builtin_amdgcn_global_load_lds((attn_glb_t*)(vb + (long long)kr * ld + vc * 8), (attn_lds_t*)(Vs[slot] + piece * 8 * KS), 16, 0, 0);
    }
  };
  if (klen > 0) issue(0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  int slot = 0;
  for (int k0 = 0; k0 < klen; k0 += KB, slot ^= 1) {
    // the other slot's readers all passed the barrier that ended the previous iteration: refill it while this block computes
    if (k0 + KB < klen) issue(k0 + KB, slot ^ 1);
    const T* Kc = Ks[slot];
    const T* Vc = Vs[slot];
    f32x4 s[QS][4];
#pragma unroll
    for (int kt = 0; kt < 4; ++kt) {
#pragma unroll
      for (int u = 0; u < QS; ++u) s[u][kt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
        const int row = kt * 16 + li;
        const v8 kf = *reinterpret_cast<const v8*>(Kc + row * KS + (((kk * 4 + lg) ^ ((row >> 1) & 7)) << 3));
#pragma unroll
        for (int u = 0; u < QS; ++u) s[u][kt] = T16<T>::mfma(kf, qf[u][kk], s[u][kt]);
      }
    }
  // softmax bookkeeping in as few VALU slots as possible (see attn_sp_kernel): raw-score max, one packed FMA per two scores,
    // and the key mask only in the last key block (wave-uniform test) — the running max m[] is kept in the RAW domain
    float alpha[QS];
    const bool tail = k0 + KB > klen;
#pragma unroll
    for (int u = 0; u < QS; ++u) {
      if (tail) {
#pragma unroll
        for (int kt = 0; kt < 4; ++kt)
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (k0 + kt * 16 + lg * 4 + r >= klen) s[u][kt][r] = -INFINITY;
      }
      float bmax = -INFINITY;
#pragma unroll
      for (int kt = 0; kt < 4; ++kt)
#pragma unroll
        for (int r = 0; r < 4; ++r) bmax = fmaxf(bmax, s[u][kt][r]);
      bmax = fmaxf(bmax, __shfl_xor(bmax, 16));
      bmax = fmaxf(bmax, __shfl_xor(bmax, 32));
      const float mnew = fmaxf(m[u], bmax);                 // finite: every block holds at least one unmasked key
      alpha[u] = __builtin_amdgcn_exp2f((m[u] - mnew) * scale_log2e);      // first block: exp2(-inf) = 0
      m[u] = mnew;
      const f32x2 sc2 = {scale_log2e, scale_log2e};
      const f32x2 off2 = {-mnew * scale_log2e, -mnew * scale_log2e};
      f32x2 psum2 = {0.f, 0.f};
#pragma unroll
      for (int kt = 0; kt < 4; ++kt) {
        const f32x2 x01 = __builtin_elementwise_fma(f32x2{s[u][kt][0], s[u][kt][1]}, sc2, off2);
        const f32x2 x23 = __builtin_elementwise_fma(f32x2{s[u][kt][2], s[u][kt][3]}, sc2, off2);
        const f32x2 p01 = {__builtin_amdgcn_exp2f(x01[0]), __builtin_amdgcn_exp2f(x01[1])};
        const f32x2 p23 = {__builtin_amdgcn_exp2f(x23[0]), __builtin_amdgcn_exp2f(x23[1])};
        s[u][kt] = f32x4{p01[0], p01[1], p23[0], p23[1]};
        psum2 += p01;
        psum2 += p23;
      }
      lsum[u] = lsum[u] * alpha[u] + (psum2[0] + psum2[1]);  // per-lane partial of the query's normaliser (4 lanes share a query and its alpha)
      // the accumulator rescale only when some query of this 16-row group raised its running max in this block (wave-uniform test: one
      // ballot + a scalar branch): after the first few key blocks alpha is exactly 1.0 for every lane of most blocks, and a multiply by
      // 1.0 skipped changes no bit — 16 VALU slots of the ~100 per query group and key block (round 5)
      if (__builtin_amdgcn_ballot_w64(alpha[u] != 1.0f) != 0) {
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) o[u][dt] *= alpha[u];
      }
    }
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      v8 pf[QS];
#pragma unroll
      for (int u = 0; u < QS; ++u)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          pf[u][j] = T16<T>::from_f32(s[u][2 * c][j]);
          pf[u][4 + j] = T16<T>::from_f32(s[u][2 * c + 1][j]);
        }
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) {
        // row 32 c + 4 lg + li / 4 (and 16 rows further: the same (row >> 1) & 3), logical chunk 2 dt + (li & 3) / 2, half (li & 1)
        const int vrow = (2 * c) * 16 + lg * 4 + (li >> 2);
        const T* vr = Vc + vrow * KS + (((dt * 2 + ((li >> 1) & 1)) ^ (((vrow >> 1) & 3) << 1)) << 3) + (li & 1) * 4;
        const v4 v0 = tr_read4<T>(vr);
        const v4 v1 = tr_read4<T>(vr + 16 * KS);
        v8 vf;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          vf[j] = v0[j];
          vf[4 + j] = v1[j];
        }
#pragma unroll
        for (int u = 0; u < QS; ++u) o[u][dt] = T16<T>::mfma(vf, pf[u], o[u][dt]);
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's share of the next block has landed ...
    __syncthreads();                                      // ... and every wave is done reading the current slot
  }
#pragma unroll
  for (int u = 0; u < QS; ++u) {
    float l = lsum[u];
    l += __shfl_xor(l, 16);
    l += __shfl_xor(l, 32);
    const float inv = l > 0.f ? 1.0f / l : 0.f;
    if (qi[u] < Tn) {
      const long long orow = (row0 + qi[u]) * ldo + h * 64;
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) {
        v4 hh, ll;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          T a, c;
          split16<T>(o[u][dt][r] * inv, a, c);
          hh[r] = a;
          ll[r] = c;
        }
        *reinterpret_cast<v4*>(oh + orow + dt * 16 + lg * 4) = hh;
        if (ol) *reinterpret_cast<v4*>(ol + orow + dt * 16 + lg * 4) = ll;
      }
    }
  }
}

template <typename T>
static int launch_attn(const void* q, const void* k, const void* v, long long ld, void* oh, void* ol, long long ldo,
                       int B, int Tn, int H, float scale, const int* kv_len, int hm, hipStream_t st,
                       const float* bias = nullptr, long long ldb = 0, const float* gate = nullptr) {
  const float sl2 = scale * 1.4426950408889634f;
  dim3 grid(1, H, B), block(256), block8(512);  // one workgroup per (batch, head): K/V staged once
  ProfScope prof(bias ? "attention_bias" : "attention", 4.0 * B * H * (double)Tn * Tn * 64, 2.0 * 4 * (double)B * Tn * H * 64, st);
  if (bias) {
#define MER_ATTN_BCASE(N)                                                                                            \
  do {                                                                                                               \
    if (N >= 8 && N <= 16)                                                                                     \
      hipLaunchKernelGGL((attn_sp_kernel<T, N, true, (N >= 8 && N <= 16) ? 8 : 4>), grid, (N >= 8 && N <= 16) ? block8 : block, 0, st, (const T*)q, (const T*)k, (const T*)v, ld, \
                         (T*)oh, (T*)ol, ldo, Tn, sl2, kv_len, hm, g_gemm_dbg, bias, ldb, gate);                     \
    else                                                                                                             \
      hipLaunchKernelGGL((attn_sp_kernel<T, N, true>), grid, block, 0, st, (const T*)q, (const T*)k, (const T*)v, ld, \
                         (T*)oh, (T*)ol, ldo, Tn, sl2, kv_len, hm, g_gemm_dbg, bias, ldb, gate);                     \
  } while (0)
    if (Tn <= 64) MER_ATTN_BCASE(4);
    else if (Tn <= 128) MER_ATTN_BCASE(8);
    else if (Tn <= 224) MER_ATTN_BCASE(14);
    else if (Tn <= 256) MER_ATTN_BCASE(16);
    else if (Tn <= 288) MER_ATTN_BCASE(18);
    else if (Tn <= 512) MER_ATTN_BCASE(32);
    else {
      set_error("mer_attention_bias: T=%d > 512 is not supported with a score bias", Tn);
      return MER_EUNSUPPORTED;
    }
#undef MER_ATTN_BCASE
    return check_launch("attention_bias");
  }
#define MER_ATTN_CASE(N)                                                                                       \
  do {                                                                                                         \
    if (N >= 8 && N <= 16)                                                                               \
      hipLaunchKernelGGL((attn_sp_kernel<T, N, false, (N >= 8 && N <= 16) ? 8 : 4>), grid, (N >= 8 && N <= 16) ? block8 : block, 0, st, (const T*)q, (const T*)k, (const T*)v, ld, \
                         (T*)oh, (T*)ol, ldo, Tn, sl2, kv_len, hm, g_gemm_dbg, nullptr, 0, nullptr);                                \
    else                                                                                                       \
      hipLaunchKernelGGL((attn_sp_kernel<T, N>), grid, block, 0, st, (const T*)q, (const T*)k, (const T*)v, ld, \
                         (T*)oh, (T*)ol, ldo, Tn, sl2, kv_len, hm, g_gemm_dbg, nullptr, 0, nullptr);                                \
  } while (0)
  if (Tn <= 64) MER_ATTN_CASE(4);
  else if (Tn <= 128) MER_ATTN_CASE(8);
  else if (Tn <= 224) MER_ATTN_CASE(14);
  else if (Tn <= 256) MER_ATTN_CASE(16);
  else if (Tn <= 288) MER_ATTN_CASE(18);
  else if (Tn <= 512) MER_ATTN_CASE(32);
  else {
    // QS = 16-query sub-tiles per wave (64 * QS queries per workgroup), PF = register prefetch of the next key block
#define MER_ATTN_STREAM(QS_)                                                                                          \
  do {                                                                                                                     \
    dim3 sgrid((unsigned)cdiv(Tn, 64 * QS_), H, B);                                                                        \
    hipLaunchKernelGGL((attn_stream_kernel<T, QS_>), sgrid, block, 0, st, (const T*)q, (const T*)k, (const T*)v, ld, \
                       (T*)oh, (T*)ol, ldo, Tn, sl2, kv_len, hm);                                                          \
  } while (0)
    MER_ATTN_STREAM(2);   // (QS = 1 and round 2's register-prefetch form measured slower: profiles/r02_attention_variants.jsonl)
#undef MER_ATTN_STREAM
  }
#undef MER_ATTN_CASE
  return check_launch("attention");
}


// One query per sequence (the [CLS] row of a ViT's last block: get_image_features only reads h[:, 0], HF:clip/modeling_clip.py:719-748):
// out[b, h*64 ..] = softmax(q[b, h] . K[b, :, h]^T * scale) V[b, :, h].  One wave per (sequence, head); a wave-load covers 8 keys x
// 128 B (whole lines): lane (sub, cg) owns dims 8 cg .. 8 cg + 7 of key 8 it + sub, so a key's score lands in the 8 lanes that
// later scale the same key's V row — no LDS, no second pass over K; all fp32.  NIT = ceil(T / 8) bounds the register array.
template <typename T, int NIT>
__global__ __launch_bounds__(256) void attn_cls_kernel(const T* __restrict__ q, long long ldq, const T* __restrict__ k, const T* __restrict__ v,
                                                       long long ld, T* __restrict__ oh, T* __restrict__ ol, long long ldo, int Tn, int H,
                                                       int nbh, float scale, const int* __restrict__ kv_len) {
  typedef typename T16<T>::v8 v8;
  const int lane = threadIdx.x & 63;
  const int bh = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (bh >= nbh) return;
  const int b = bh / H, h = bh % H;
  const int sub = lane >> 3, cg = lane & 7;
  const int klen = kv_len ? (kv_len[b] < Tn ? kv_len[b] : Tn) : Tn;
  const v8 qv = *reinterpret_cast<const v8*>(q + (long long)b * ldq + h * 64 + cg * 8);
  float qf[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) qf[j] = T16<T>::to_f32(qv[j]) * scale;
  const T* kb = k + (long long)b * Tn * ld + h * 64 + cg * 8;
  const T* vb = v + (long long)b * Tn * ld + h * 64 + cg * 8;
  float s[NIT];
  float mx = -INFINITY;
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    const int t = it * 8 + sub;
    float d = 0.f;
    if (t < klen) {
      const v8 kv = *reinterpret_cast<const v8*>(kb + (long long)t * ld);
#pragma unroll
      for (int j = 0; j < 8; ++j) d += qf[j] * T16<T>::to_f32(kv[j]);
    }
    d += __shfl_xor(d, 1);
    d += __shfl_xor(d, 2);
    d += __shfl_xor(d, 4);
    s[it] = t < klen ? d : -INFINITY;
    mx = fmaxf(mx, s[it]);
  }
  mx = wave_max(mx);
  float sum = 0.f, acc[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) acc[j] = 0.f;
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    const int t = it * 8 + sub;
    if (t < klen) {
      const float p = __expf(s[it] - mx);
      sum += p;
      const v8 vv = *reinterpret_cast<const v8*>(vb + (long long)t * ld);
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[j] += p * T16<T>::to_f32(vv[j]);
    }
  }
  // every key's p sits in 8 lanes (cg): the wave sum counts it 8 times
  sum = wave_sum(sum) * 0.125f;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    acc[j] += __shfl_xor(acc[j], 8);
    acc[j] += __shfl_xor(acc[j], 16);
    acc[j] += __shfl_xor(acc[j], 32);
  }
  if (sub == 0) {
    const float inv = sum > 0.f ? 1.0f / sum : 0.f;
    v8 hh, ll;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      T a, c;
      split16<T>(acc[j] * inv, a, c);
      hh[j] = a;
      ll[j] = c;
    }
    *reinterpret_cast<v8*>(oh + (long long)b * ldo + h * 64 + cg * 8) = hh;
    if (ol) *reinterpret_cast<v8*>(ol + (long long)b * ldo + h * 64 + cg * 8) = ll;
  }
}

template <typename T>
static int launch_attn_cls(const void* q, long long ldq, const void* k, const void* v, long long ld, void* oh, void* ol, long long ldo,
                           int B, int Tn, int H, float scale, const int* kv_len, hipStream_t st) {
  const int nbh = B * H;
  dim3 grid((unsigned)cdiv(nbh, 4)), block(256);
  ProfScope prof("attention_cls", 4.0 * nbh * (double)Tn * 64, 2.0 * 2 * (double)B * Tn * H * 64, st);
#define MER_CLS_CASE(N) hipLaunchKernelGGL((attn_cls_kernel<T, N>), grid, block, 0, st, (const T*)q, ldq, (const T*)k, (const T*)v, ld, (T*)oh, (T*)ol, ldo, Tn, H, nbh, scale, kv_len)
  if (Tn <= 64) MER_CLS_CASE(8);
  else if (Tn <= 208) MER_CLS_CASE(26);
  else if (Tn <= 264) MER_CLS_CASE(33);
  else if (Tn <= 584) MER_CLS_CASE(73);
  else {
    set_error("mer_attention_cls: T=%d > 584 unsupported", Tn);
    return MER_EUNSUPPORTED;
  }
#undef MER_CLS_CASE
  return check_launch("attention_cls");
}

}  // namespace mer

extern "C" int mer_attention(const void* q, const void* k, const void* v, long long ld, void* out_hi, void* out_lo,
                             long long ldo, int B, int T, int H, float scale, const int* kv_len, int dtype,
                             mer_stream_t stream) {
  using namespace mer;
  MER_REQUIRE(q && k && v && out_hi, MER_EINVAL, "mer_attention: null pointer");
  MER_REQUIRE(B > 0 && T > 0 && H > 0, MER_ESHAPE, "mer_attention: bad shape B=%d T=%d H=%d", B, T, H);
  MER_REQUIRE(ld % 8 == 0 && ldo % 4 == 0, MER_ESHAPE, "mer_attention: ld %% 8 / ldo %% 4 alignment");
  MER_REQUIRE((((uintptr_t)q | (uintptr_t)k | (uintptr_t)v) & 15) == 0, MER_EINVAL, "mer_attention: q/k/v must be 16-byte aligned");
  hipStream_t st = (hipStream_t)stream;
  if (dtype == MER_DT_F16) return launch_attn<f16>(q, k, v, ld, out_hi, out_lo, ldo, B, T, H, scale, kv_len, 0, st);
  if (dtype == MER_DT_BF16) return launch_attn<bf16>(q, k, v, ld, out_hi, out_lo, ldo, B, T, H, scale, kv_len, 0, st);
  set_error("mer_attention: bad dtype %d", dtype);
  return MER_EINVAL;
}

extern "C" int mer_attention_hm(const void* q, const void* k, const void* v, void* out_hi, void* out_lo, long long ldo, int B,
                                int T, int H, float scale, const int* kv_len, int dtype, mer_stream_t stream) {
  using namespace mer;
  MER_REQUIRE(q && k && v && out_hi, MER_EINVAL, "mer_attention_hm: null pointer");
  MER_REQUIRE(B > 0 && T > 0 && H > 0 && ldo % 4 == 0, MER_ESHAPE, "mer_attention_hm: bad shape B=%d T=%d H=%d", B, T, H);
  MER_REQUIRE((((uintptr_t)q | (uintptr_t)k | (uintptr_t)v) & 15) == 0, MER_EINVAL, "mer_attention_hm: q/k/v must be 16-byte aligned");
  hipStream_t st = (hipStream_t)stream;
  if (dtype == MER_DT_F16) return launch_attn<f16>(q, k, v, 64, out_hi, out_lo, ldo, B, T, H, scale, kv_len, 1, st);
  if (dtype == MER_DT_BF16) return launch_attn<bf16>(q, k, v, 64, out_hi, out_lo, ldo, B, T, H, scale, kv_len, 1, st);
  set_error("mer_attention_hm: bad dtype %d", dtype);
  return MER_EINVAL;
}

extern "C" int mer_attention_bias(const void* q, const void* k, const void* v, long long ld, void* out_hi, void* out_lo,
                                  long long ldo, int B, int T, int H, float scale, const int* kv_len, const float* bias,
                                  long long ldb, const float* gate, int dtype, mer_stream_t stream) {
  using namespace mer;
  MER_REQUIRE(q && k && v && out_hi && bias, MER_EINVAL, "mer_attention_bias: null pointer");
  MER_REQUIRE(B > 0 && T > 0 && H > 0, MER_ESHAPE, "mer_attention_bias: bad shape B=%d T=%d H=%d", B, T, H);
  MER_REQUIRE(ld % 8 == 0 && ldo % 4 == 0, MER_ESHAPE, "mer_attention_bias: ld %% 8 / ldo %% 4 alignment");
  MER_REQUIRE(ldb >= T && ldb % 4 == 0 && ((uintptr_t)bias & 15) == 0, MER_ESHAPE,
              "mer_attention_bias: bias rows must be padded to a multiple of 4 floats (ldb=%lld, T=%d) and 16-byte aligned", ldb, T);
  MER_REQUIRE((((uintptr_t)q | (uintptr_t)k | (uintptr_t)v) & 15) == 0, MER_EINVAL, "mer_attention_bias: q/k/v must be 16-byte aligned");
  hipStream_t st = (hipStream_t)stream;
  if (dtype == MER_DT_F16) return launch_attn<f16>(q, k, v, ld, out_hi, out_lo, ldo, B, T, H, scale, kv_len, 0, st, bias, ldb, gate);
  if (dtype == MER_DT_BF16) return launch_attn<bf16>(q, k, v, ld, out_hi, out_lo, ldo, B, T, H, scale, kv_len, 0, st, bias, ldb, gate);
  set_error("mer_attention_bias: bad dtype %d", dtype);
  return MER_EINVAL;
}

extern "C" int mer_attention_cls(const void* q, long long ldq, const void* k, const void* v, long long ld, void* out_hi, void* out_lo,
                                 long long ldo, int B, int T, int H, float scale, const int* kv_len, int dtype, mer_stream_t stream) {
  using namespace mer;
  MER_REQUIRE(q && k && v && out_hi, MER_EINVAL, "mer_attention_cls: null pointer");
  MER_REQUIRE(B > 0 && T > 0 && H > 0, MER_ESHAPE, "mer_attention_cls: bad shape B=%d T=%d H=%d", B, T, H);
  MER_REQUIRE(ld % 8 == 0 && ldq % 8 == 0 && ldo % 8 == 0, MER_ESHAPE, "mer_attention_cls: ld / ldq / ldo must be multiples of 8");
  MER_REQUIRE((((uintptr_t)q | (uintptr_t)k | (uintptr_t)v | (uintptr_t)out_hi | (uintptr_t)out_lo) & 15) == 0, MER_EINVAL,
              "mer_attention_cls: operands must be 16-byte aligned");
  hipStream_t st = (hipStream_t)stream;
  if (dtype == MER_DT_F16) return launch_attn_cls<f16>(q, ldq, k, v, ld, out_hi, out_lo, ldo, B, T, H, scale, kv_len, st);
  if (dtype == MER_DT_BF16) return launch_attn_cls<bf16>(q, ldq, k, v, ld, out_hi, out_lo, ldo, B, T, H, scale, kv_len, st);
  set_error("mer_attention_cls: bad dtype %d", dtype);
  return MER_EINVAL;
}
