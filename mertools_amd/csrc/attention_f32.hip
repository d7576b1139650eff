// attention_f32.hip — softmax(q k^T * scale) v for head_dim 64 with fp32 operands on the exact fp32 MFMA
// (v_mfma_f32_16x16x4_f32), for the "accurate" preset (HF:hubert/modeling_hubert.py:236-259, HF:clip/modeling_clip.py:280-335,
// HF:roberta/modeling_roberta.py:186-250 — identical math, no causal mask on this path).
//
// Why it exists: with three-pass GEMMs the largest rounding left in a block was the attention kernel's single f16 plane for q, k, v
// and P (4.7e-4 of the context rows, tests/studies/outlier_block_bisect_gpu.py) — harmless on ordinary checkpoints, but a post-LN
// encoder with massive activation channels amplifies it by gamma / sigma ~ 10 per LayerNorm.  Here q | k | v arrive
// as the fp32 output of the QKV GEMM and every product is an exact fp32 product accumulated in fp32.
//
// A workgroup = 64 queries of one (batch, head), a wave = 16 of them; the keys are walked 16 at a time with an online softmax.  The
// K and V rows of a key tile (16 x 64 floats each) are staged ONCE per workgroup into a double-buffered LDS tile (rows padded to 68
// floats: the fragment reads below are bank-conflict free) — the first version had every wave fetch its fragments from L2 with sixteen
// 4-byte gathers per tile and ran at 42 TF, bound by the CU's address path, not by the matrix pipe.  The MFMA sums over its k index in
// any order, which is used twice to keep every operand a lane-local value:
//   S^T = K Q^T  k index = feature d, enumerated as d = 16 (lane >> 4) + step: a lane reads 16 CONTIGUOUS floats of its key's row
//                (and holds 16 of its query's row) and feeds them to 16 MFMAs; the result has, for query (lane & 15), keys 4 (lane >> 4) + r;
//   O^T = V^T P^T  k index = key, enumerated as key = 4 (lane >> 4) + r over the steps r = 0 .. 3: the probabilities a lane just
//                computed ARE its B operand of step r; the A operand is V[key][16 dt + (lane & 15)].
// Cost: 32 MFMAs of 32 cycles per 16 x 16 (query, key) tile — ~1/16 of the f16 kernel's matrix rate, a few per cent of an
// "accurate" step (whose GEMMs run three passes).
#include "common.h"

namespace mer {

__device__ __forceinline__ f32x4 mfma4(float a, float b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }

template <typename T>
__global__ __launch_bounds__(256) void attn_f32_kernel(const float* __restrict__ q, const float* __restrict__ k, const float* __restrict__ v,
                                                       long long ld, T* oh, T* ol, long long ldo, int Tn, float scale,
                                                       const int* kv_len) {
  constexpr int RS = 68;                              // LDS row stride in floats (64 + 4: see the fragment reads)
  __shared__ __attribute__((aligned(16))) float Ks[2][16 * RS];
  __shared__ __attribute__((aligned(16))) float Vs[2][16 * RS];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 15, lg = lane >> 4;
  const int b = blockIdx.z, h = blockIdx.y;
  const int q0 = (blockIdx.x * 4 + wave) * 16;       // (a wave whose queries are all beyond T still helps staging and keeps the barriers)
  int klen = kv_len ? kv_len[b] : Tn;
  klen = klen < Tn ? klen : Tn;
  const long long row0 = (long long)b * Tn;
  const float* qb = q + row0 * ld + h * 64;
  const float* kb = k + row0 * ld + h * 64;
  const float* vb = v + row0 * ld + h * 64;

  // staging: thread t brings 16 bytes of K and of V: tile row t >> 4, floats 4 (t & 15) .. + 3 — a wave-load = 4 rows x 256 B
  const int srow = tid >> 4, scol = (tid & 15) * 4;
  f32x4 kst, vst;
  auto fetch = [&](int k0) __attribute__((always_inline)) {
    const int key = k0 + srow;
    const long long off = (long long)(key < Tn ? key : Tn - 1) * ld + scol;
    kst = *reinterpret_cast<const f32x4*>(kb + off);
    vst = key < klen ? *reinterpret_cast<const f32x4*>(vb + off) : f32x4{0.f, 0.f, 0.f, 0.f};   // (a masked key's weight is 0: its row must not be NaN * 0)
  };
  auto stash = [&](int buf) __attribute__((always_inline)) {
    *reinterpret_cast<f32x4*>(&Ks[buf][srow * RS + scol]) = kst;
    *reinterpret_cast<f32x4*>(&Vs[buf][srow * RS + scol]) = vst;
  };

  // this lane's query row (clamped: rows beyond T are computed and not stored), features 16 lg .. 16 lg + 15, pre-scaled
  const int qr = q0 + li < Tn ? q0 + li : Tn - 1;
  f32x4 qf[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    qf[j] = *reinterpret_cast<const f32x4*>(qb + (long long)qr * ld + 16 * lg + 4 * j);
    qf[j] *= scale;
  }
  f32x4 o[4];
#pragma unroll
  for (int dt = 0; dt < 4; ++dt) o[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
  float m = -INFINITY, l = 0.f;                // running max / sum of this lane's query (replicated over the four lg lanes)

  if (klen > 0) {
    fetch(0);
    stash(0);
  }
  __syncthreads();
  int buf = 0;
  for (int k0 = 0; k0 < klen; k0 += 16, buf ^= 1) {
    const bool more = k0 + 16 < klen;           // (uniform over the workgroup)
    if (more) fetch(k0 + 16);                   // the next tile's global loads fly under this tile's MFMAs
    // S^T tile: rows = keys k0 + 4 lg + r, column = query li.  K fragment: row li, floats 16 lg .. + 15 (row stride 68 floats: the
    // 16 li lanes of a 16-byte read start 4 banks apart, the lg groups 16 floats apart — no two lanes of a pass share a bank)
    f32x4 kf[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) kf[j] = *reinterpret_cast<const f32x4*>(&Ks[buf][li * RS + 16 * lg + 4 * j]);
    // V^T operands: keys 4 lg + r, features 16 dt + li (4-byte reads: 16 consecutive banks per lg, the four lg groups 16 banks apart)
    float vf[4][4];
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) vf[r][dt] = Vs[buf][(4 * lg + r) * RS + 16 * dt + li];
    f32x4 s = {0.f, 0.f, 0.f, 0.f}, s1 = s;    // two accumulators: consecutive MFMAs are independent (40-cycle dependent latency)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int e = 0; e < 4; e += 2) {
        s = mfma4(kf[j][e], qf[j][e], s);
        s1 = mfma4(kf[j][e + 1], qf[j][e + 1], s1);
      }
    s += s1;
    float tmax = -INFINITY;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      if (k0 + 4 * lg + r >= klen) s[r] = -INFINITY;        // keys beyond the sequence
      tmax = fmaxf(tmax, s[r]);
    }
    tmax = fmaxf(tmax, __shfl_xor(tmax, 16));
    tmax = fmaxf(tmax, __shfl_xor(tmax, 32));
    const float mn = fmaxf(m, tmax);                         // finite: key k0 < klen belongs to this tile
    const float alpha = __expf(m - mn);                      // (m = -inf on the first tile: alpha = 0)
    float p[4], psum = 0.f;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      p[r] = __expf(s[r] - mn);
      psum += p[r];
    }
    psum += __shfl_xor(psum, 16);
    psum += __shfl_xor(psum, 32);
    l = l * alpha + psum;
    m = mn;
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) o[dt] *= alpha;
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) o[dt] = mfma4(vf[r][dt], p[r], o[dt]);
    // the other buffer was last read in the previous iteration, which every wave has left (the barrier below, one iteration ago)
    if (more) stash(buf ^ 1);
    __syncthreads();
  }
  // O^T tile dt: rows = features 16 dt + 4 lg + r, column = query li
  if (q0 + li < Tn) {
    const float inv = l > 0.f ? 1.0f / l : 0.f;
    const long long off = (row0 + q0 + li) * ldo + h * 64;
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) {
      typename T16<T>::v4 hh, ll;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float x = o[dt][r] * inv;     // (split16 pins it: one fp32 value for hi and for lo)
        T a, c;
        split16<T>(x, a, c);
        hh[r] = a;
        ll[r] = c;
      }
      *reinterpret_cast<typename T16<T>::v4*>(oh + off + 16 * dt + 4 * lg) = hh;
      if (ol) *reinterpret_cast<typename T16<T>::v4*>(ol + off + 16 * dt + 4 * lg) = ll;
    }
  }
}

}  // namespace mer

extern "C" int mer_attention_f32(const float* q, const float* k, const float* v, long long ld, void* out_hi, void* out_lo,
                                 long long ldo, int B, int T, int H, float scale, const int* kv_len, int dtype,
                                 mer_stream_t stream) {
  using namespace mer;
  MER_REQUIRE(q && k && v && out_hi, MER_EINVAL, "mer_attention_f32: null pointer");
  MER_REQUIRE(B > 0 && T > 0 && H > 0 && B <= 65535 && H <= 65535, MER_ESHAPE, "mer_attention_f32: bad shape B=%d T=%d H=%d", B, T, H);
  MER_REQUIRE(ld % 4 == 0 && ldo % 4 == 0, MER_ESHAPE, "mer_attention_f32: ld %% 4 / ldo %% 4 alignment");
  MER_REQUIRE((((uintptr_t)q | (uintptr_t)k | (uintptr_t)v) & 15) == 0 && (((uintptr_t)out_hi | (uintptr_t)out_lo) & 7) == 0, MER_EINVAL,
              "mer_attention_f32: q / k / v must be 16-byte aligned, the output planes 8-byte aligned");
  MER_REQUIRE(dtype == MER_DT_F16 || dtype == MER_DT_BF16, MER_EINVAL, "mer_attention_f32: bad dtype %d", dtype);
  hipStream_t st = (hipStream_t)stream;
  dim3 grid((unsigned)cdiv(T, 64), (unsigned)H, (unsigned)B), block(256);
  ProfScope prof("attention_f32", 4.0 * B * H * (double)T * T * 64, (3.0 * 4 + 4.0) * (double)B * T * H * 64, st);
  if (dtype == MER_DT_F16)
    hipLaunchKernelGGL((attn_f32_kernel<f16>), grid, block, 0, st, q, k, v, ld, (f16*)out_hi, (f16*)out_lo, ldo, T, scale, kv_len);
  else
    hipLaunchKernelGGL((attn_f32_kernel<bf16>), grid, block, 0, st, q, k, v, ld, (bf16*)out_hi, (bf16*)out_lo, ldo, T, scale, kv_len);
  return check_launch("attention_f32");
}
