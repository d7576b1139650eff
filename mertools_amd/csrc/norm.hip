// norm.hip — row-wise LayerNorm family (one wave64 per row, whole row in registers).
//   mer_layernorm     torch.nn.LayerNorm (+ optional activation) -> fp32 and/or 16-bit planes
//   mer_vit_assemble  CLS/patch/position assembly (+ pre-LN)   HF:clip/modeling_clip.py:138-217
//   mer_bert_embed    word+position+type embeddings + LN        HF:roberta/modeling_roberta.py:56-155
// All three are HBM-bound streaming kernels: each row is read once with 16-byte loads, reduced
// with two wave-level passes (mean, then centred variance, as torch does), and written once.
#include "common.h"

namespace mer {

template <typename T, int NV>
struct RowLN {
  // v[j] holds columns (lane + 64*j)*4 .. +3 of the row; entries past D/4 are zero.
  static __device__ __forceinline__ void run(f32x4 (&v)[NV], int lane, int nv4, int D, const float* gamma,
                                             const float* beta, float eps, int act, float* o32, T* ohi, T* olo) {
    run_with(v, lane, nv4, D, [&](int idx, f32x4& g, f32x4& b) __attribute__((always_inline)) {
      g = f32x4{1.f, 1.f, 1.f, 1.f};
      b = f32x4{0.f, 0.f, 0.f, 0.f};
      if (gamma) g = *reinterpret_cast<const f32x4*>(gamma + idx * 4);
      if (beta) b = *reinterpret_cast<const f32x4*>(beta + idx * 4);
    }, eps, act, o32, ohi, olo);
  }
  // `affine(idx, g, b)` hands out gamma / beta of columns 4 idx .. 4 idx + 3 (from global memory, or from the LDS copy of the multi-row kernel)
  template <typename Affine>
  static __device__ __forceinline__ void run_with(f32x4 (&v)[NV], int lane, int nv4, int D, Affine affine,
                                                  float eps, int act, float* o32, T* ohi, T* olo) {
    // Every product / sum below is spelled out (fmaf where a fused operation is meant) and contraction is off: hipcc otherwise
    // fuses differently in the two kernels that inline this function, their fp32 outputs differ in the last bit, the 16-bit planes
    // round differently — and a clip's features depended on whether its batch was large enough for the multi-row kernel
    // (tests/studies/batch_rows_ops_gpu.py: the one operator that was not batch-size invariant).
#pragma clang fp contract(off)
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < NV; ++j) s += (v[j][0] + v[j][1]) + (v[j][2] + v[j][3]);
    const float mean = wave_sum(s) / (float)D;
    float q = 0.f;
#pragma unroll
    for (int j = 0; j < NV; ++j) {
      if (lane + 64 * j < nv4) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float d = v[j][e] - mean;
          q = fmaf(d, d, q);
        }
      }
    }
    const float var = wave_sum(q) / (float)D;
    const float rstd = 1.0f / sqrtf(var + eps);
#pragma unroll
    for (int j = 0; j < NV; ++j) {
      const int idx = lane + 64 * j;
      if (idx < nv4) {
        f32x4 g, b, y;
        affine(idx, g, b);
#pragma unroll
        for (int e = 0; e < 4; ++e) y[e] = act_apply(fmaf((v[j][e] - mean) * rstd, g[e], b[e]), act);
        if (o32) *reinterpret_cast<f32x4*>(o32 + idx * 4) = y;
        if (ohi) {
          typename T16<T>::v4 h, l;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            T hh, ll;
            split16<T>(y[e], hh, ll);
            h[e] = hh;
            l[e] = ll;
          }
          *reinterpret_cast<typename T16<T>::v4*>(ohi + idx * 4) = h;
          if (olo) *reinterpret_cast<typename T16<T>::v4*>(olo + idx * 4) = l;
        }
      }
    }
  }
  static __device__ __forceinline__ void store_plain(f32x4 (&v)[NV], int lane, int nv4, float* o32, T* ohi, T* olo) {
#pragma unroll
    for (int j = 0; j < NV; ++j) {
      const int idx = lane + 64 * j;
      if (idx < nv4) {
        if (o32) *reinterpret_cast<f32x4*>(o32 + idx * 4) = v[j];
        if (ohi) {
          typename T16<T>::v4 h, l;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            T hh, ll;
            split16<T>(v[j][e], hh, ll);
            h[e] = hh;
            l[e] = ll;
          }
          *reinterpret_cast<typename T16<T>::v4*>(ohi + idx * 4) = h;
          if (olo) *reinterpret_cast<typename T16<T>::v4*>(olo + idx * 4) = l;
        }
      }
    }
  }
};

template <typename T, int NV>
__global__ __launch_bounds__(256) void layernorm_kernel(const float* x, long long ldx, const float* gamma,
                                                        const float* beta, float eps, int M, int D, int act,
                                                        float* out32, long long ld32, T* ohi, T* olo, long long ld16) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= M) return;
  const int lane = threadIdx.x & 63, nv4 = D >> 2;
  const float* xr = x + (long long)row * ldx;
  f32x4 v[NV];
#pragma unroll
  for (int j = 0; j < NV; ++j) {
    const int idx = lane + 64 * j;
    if (idx < nv4) v[j] = *reinterpret_cast<const f32x4*>(xr + idx * 4);
    else v[j] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  RowLN<T, NV>::run(v, lane, nv4, D, gamma, beta, eps, act, out32 ? out32 + (long long)row * ld32 : nullptr,
                    ohi ? ohi + (long long)row * ld16 : nullptr, olo ? olo + (long long)row * ld16 : nullptr);
}

// Multi-row form for large M: a wave walks rows wave, wave + W, wave + 2W, ... of a fixed-size grid, keeps gamma / beta in LDS
// (the one-row kernel re-reads 2 D floats of affine parameters through L1 for every D floats of input) and has the next row's
// loads in flight while it reduces and writes the current one.  Same arithmetic per row as layernorm_kernel (bit-identical outputs).
template <typename T, int NV>
__global__ __launch_bounds__(256) void layernorm_rows_kernel(const float* x, long long ldx, const float* gamma, const float* beta, float eps,
                                                             int M, int D, int act, float* out32, long long ld32, T* ohi, T* olo,
                                                             long long ld16) {
  const int lane = threadIdx.x & 63, nv4 = D >> 2;
  const int nwaves = gridDim.x * 4;
  int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  // gamma / beta once per workgroup into LDS (ds_read_b128 per use: no L1 traffic, no long-lived registers)
  __shared__ f32x4 sg[64 * NV], sb[64 * NV];
  for (int i = threadIdx.x; i < 64 * NV; i += 256) {
    sg[i] = (gamma && i < nv4) ? *reinterpret_cast<const f32x4*>(gamma + i * 4) : f32x4{1.f, 1.f, 1.f, 1.f};
    sb[i] = (beta && i < nv4) ? *reinterpret_cast<const f32x4*>(beta + i * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
  }
  __syncthreads();
  if (row >= M) return;
  f32x4 cur[NV], nxt[NV];
  auto load_row = [&](f32x4 (&v)[NV], int r) __attribute__((always_inline)) {
    const float* xr = x + (long long)r * ldx;
#pragma unroll
    for (int j = 0; j < NV; ++j) {
      const int idx = lane + 64 * j;
      if (idx < nv4) v[j] = *reinterpret_cast<const f32x4*>(xr + idx * 4);
      else v[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
  };
  load_row(cur, row);
  for (; row < M; row += nwaves) {
    const int rn = row + nwaves;
    if (rn < M) load_row(nxt, rn);
    RowLN<T, NV>::run_with(cur, lane, nv4, D, [&](int idx, f32x4& g, f32x4& b) __attribute__((always_inline)) { g = sg[idx]; b = sb[idx]; },
                           eps, act, out32 ? out32 + (long long)row * ld32 : nullptr,
                           ohi ? ohi + (long long)row * ld16 : nullptr, olo ? olo + (long long)row * ld16 : nullptr);
#pragma unroll
    for (int j = 0; j < NV; ++j) cur[j] = nxt[j];
  }
}

constexpr float CM_FIX = 16384.0f;   // 2^14: |x| <= 65504 (f16) -> |x * 2^14| < 2^31 per element; sums in 64 bits (mer_seq_bias below)

template <typename T, int NV>
__global__ __launch_bounds__(256) void vit_assemble_kernel(const float* patch, const float* cls, const float* pos,
                                                           const float* gamma, const float* beta, float eps, int N,
                                                           int P, int D, float* out32, T* ohi, T* olo) {
  const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= (long long)N * (P + 1)) return;
  const int lane = threadIdx.x & 63, nv4 = D >> 2;
  const int n = (int)(row / (P + 1)), t = (int)(row % (P + 1));
  const float* src = t == 0 ? cls : patch + ((long long)n * P + (t - 1)) * D;
  const float* pr = pos + (long long)t * D;
  f32x4 v[NV];
#pragma unroll
  for (int j = 0; j < NV; ++j) {
    const int idx = lane + 64 * j;
    if (idx < nv4) {
      const f32x4 a = *reinterpret_cast<const f32x4*>(src + idx * 4);
      const f32x4 b = *reinterpret_cast<const f32x4*>(pr + idx * 4);
      v[j] = a + b;
    } else {
      v[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
  }
  float* o32 = out32 ? out32 + row * D : nullptr;
  T* oh = ohi ? ohi + row * D : nullptr;
  T* ol = olo ? olo + row * D : nullptr;
  if (gamma)
    RowLN<T, NV>::run(v, lane, nv4, D, gamma, beta, eps, MER_ACT_NONE, o32, oh, ol);
  else
    RowLN<T, NV>::store_plain(v, lane, nv4, o32, oh, ol);
}

template <typename T, int NV>
__global__ __launch_bounds__(256) void bert_embed_kernel(const int64_t* ids, const int64_t* tt, int B, int Tn, int D,
                                                         const float* word, const float* pos, const float* type,
                                                         int pos_mode, int pad_id, const float* gamma,
                                                         const float* beta, float eps, float* out32, T* ohi, T* olo) {
  const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= (long long)B * Tn) return;
  const int lane = threadIdx.x & 63, nv4 = D >> 2;
  const int b = (int)(row / Tn), t = (int)(row % Tn);
  const long long id = ids[row];
  long long p;
  if (pos_mode == 0) {
    p = t;
  } else {
    // RoBERTa create_position_ids_from_input_ids: cumsum(mask) * mask + padding_idx
    int cnt = 0;
    for (int u = lane; u <= t; u += 64) cnt += (ids[(long long)b * Tn + u] != pad_id) ? 1 : 0;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) cnt += __shfl_xor(cnt, o);
    p = (id != pad_id) ? (long long)pad_id + cnt : (long long)pad_id;
  }
  const long long ty = tt ? tt[row] : 0;
  const float* wr = word + id * D;
  const float* pr = pos + p * D;
  const float* tr = type ? type + ty * D : nullptr;
  f32x4 v[NV];
#pragma unroll
  for (int j = 0; j < NV; ++j) {
    const int idx = lane + 64 * j;
    if (idx < nv4) {
      // HF: inputs_embeds + token_type_embeddings, then + position_embeddings
      f32x4 a = *reinterpret_cast<const f32x4*>(wr + idx * 4);
      if (tr) a = a + *reinterpret_cast<const f32x4*>(tr + idx * 4);
      a = a + *reinterpret_cast<const f32x4*>(pr + idx * 4);
      v[j] = a;
    } else {
      v[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
  }
  RowLN<T, NV>::run(v, lane, nv4, D, gamma, beta, eps, MER_ACT_NONE, out32 ? out32 + row * D : nullptr,
                    ohi ? ohi + row * D : nullptr, olo ? olo + row * D : nullptr);
}

static int pick_nv(int D) {
  const int need = (D / 4 + 63) / 64;
  if (need <= 2) return 2;
  if (need <= 3) return 3;
  if (need <= 4) return 4;
  if (need <= 8) return 8;
  return 0;
}

}  // namespace mer

namespace mer {
// workgroups of layernorm_rows_kernel<T, NV> that fit the device at once (occupancy x CUs), cached per instantiation
template <typename T, int NV>
static int ln_rows_grid_nv() {
  // per device (a process may drive several GPUs: the 2-GPU device test, config4); a benign race at first use computes the same value twice
  static int grid[16] = {0};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) {
    (void)hipGetLastError();
    return -1;
  }
  if (grid[dev] == 0) {
    int cus = 0, per = 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess ||
        hipOccupancyMaxActiveBlocksPerMultiprocessor(&per, layernorm_rows_kernel<T, NV>, 256, 0) != hipSuccess || cus * per <= 0) {
      (void)hipGetLastError();
      grid[dev] = -1;   // fall back to the one-row kernel
    } else {
      grid[dev] = cus * per;
    }
  }
  return grid[dev];
}
template <typename T>
static int ln_rows_grid(int nv) {
  switch (nv) {
    case 2: return ln_rows_grid_nv<T, 2>();
    case 3: return ln_rows_grid_nv<T, 3>();
    case 4: return ln_rows_grid_nv<T, 4>();
    default: return -1;
  }
}
}  // namespace mer

#define MER_NV_SWITCH(NVVAR, ...)                        \
  switch (NVVAR) {                                       \
    case 2: { constexpr int NV = 2; __VA_ARGS__; } break;  \
    case 3: { constexpr int NV = 3; __VA_ARGS__; } break;  \
    case 4: { constexpr int NV = 4; __VA_ARGS__; } break;  \
    default: { constexpr int NV = 8; __VA_ARGS__; } break; \
  }

extern "C" int mer_layernorm(const float* x, long long ldx, const float* gamma, const float* beta, float eps, int M,
                             int D, int act, float* out32, long long ld32, void* out16_hi, void* out16_lo,
                             long long ld16, int dtype, mer_stream_t stream) {
  using namespace mer;
  MER_REQUIRE(x && M > 0 && D > 0, MER_EINVAL, "mer_layernorm: bad args");
  MER_REQUIRE(D % 4 == 0 && ldx % 4 == 0, MER_ESHAPE, "mer_layernorm: D and ldx must be multiples of 4 (D=%d)", D);
  MER_REQUIRE(!out32 || ld32 % 4 == 0, MER_ESHAPE, "mer_layernorm: ld32 %% 4 != 0");
  MER_REQUIRE(!out16_hi || ld16 % 4 == 0, MER_ESHAPE, "mer_layernorm: ld16 %% 4 != 0");
  const int nv = pick_nv(D);
  MER_REQUIRE(nv != 0, MER_EUNSUPPORTED, "mer_layernorm: D=%d > 2048 unsupported", D);
  dim3 grid((unsigned)cdiv(M, 4)), block(256);
  hipStream_t st = (hipStream_t)stream;
  ProfScope prof("layernorm", 0.0, (double)M * D * (4 + (out32 ? 4 : 0) + (out16_hi ? 2 : 0) + (out16_lo ? 2 : 0)), st);
  // many rows of <= 1024 columns (the encoders' block LayerNorms): the multi-row kernel on exactly the grid that is resident at once
  const int rows_grid = nv <= 4 ? (dtype == MER_DT_F16 ? ln_rows_grid<f16>(nv) : ln_rows_grid<bf16>(nv)) : 0;
  if (rows_grid > 0 && M >= 8 * rows_grid) {   // at least two rows per wave
    dim3 rgrid(rows_grid);
    if (dtype == MER_DT_F16) {
      MER_NV_SWITCH(nv, layernorm_rows_kernel<f16, NV><<<rgrid, block, 0, st>>>(x, ldx, gamma, beta, eps, M, D,
                                            act, out32, ld32, (f16*)out16_hi, (f16*)out16_lo, ld16));
    } else {
      MER_NV_SWITCH(nv, layernorm_rows_kernel<bf16, NV><<<rgrid, block, 0, st>>>(x, ldx, gamma, beta, eps, M, D,
                                            act, out32, ld32, (bf16*)out16_hi, (bf16*)out16_lo, ld16));
    }
    return check_launch("layernorm");
  }
  if (dtype == MER_DT_F16) {
    MER_NV_SWITCH(nv, layernorm_kernel<f16, NV><<<grid, block, 0, st>>>(x, ldx, gamma, beta, eps, M, D,
                                          act, out32, ld32, (f16*)out16_hi, (f16*)out16_lo, ld16));
  } else {
    MER_NV_SWITCH(nv, layernorm_kernel<bf16, NV><<<grid, block, 0, st>>>(x, ldx, gamma, beta, eps, M, D,
                                          act, out32, ld32, (bf16*)out16_hi, (bf16*)out16_lo, ld16));
  }
  return check_launch("layernorm");
}

namespace mer {
// ---- per-SEQUENCE weight-residual correction (precision "mean" since round 4) ----
// The rounding error of a weight matrix is the same perturbation a (W - f16(W))^T for every token, and nearly all of it acts through
// the MEAN activation (tests/studies/mean_correction.py: one mean token recovers the accuracy of the exact second MFMA pass), i.e.
// it is a bias.  Round 3 took that mean over the whole batch, which made a clip's features depend on its batch mates (and on how the
// batch was split over GPUs); since round 4 every sequence (clip / frame / sentence) gets its own correction row:
//     tab[s, n] = bias[n] + mean_{t in sample(s)}(A[s T + t, :]) . w_lo[n, :]
// and the one-pass GEMM that follows adds row (m / T) of the table instead of a bias vector (mer_gemm16: bias_seg_rows).  The sample
// of a sequence — tokens s/2, s/2 + s, ... (s = the largest power of two with at least 16 samples; the offset keeps the [CLS] / BOS
// row out, whose weight would otherwise be 1 / #samples instead of 1 / T) below its valid length — depends on T and on the clip
// alone, so its features no longer depend on what else is in the batch, bit for bit.
//   seqmean16_kernel   workgroup = (512-column slice, sequence): exact 64-bit fixed-point column sums of the sampled rows (any order
//                      gives the same bits), written as the 16-bit mean plane [nseq, K] — no atomics, one owner per element
//   seqbias_kernel     the [nseq, K] x [N, K]^T table on the 16x16x32 MFMA, fragments straight from global memory (both operands
//                      are a few MB); workgroup = 64 sequences x 16 columns, its 16 waves split K
// Both launches are latency, not bandwidth: every load a wave needs is in flight before the first is consumed.
__host__ __device__ inline int seq_sample_stride(int T) {
  int s = 1;
  while (s * 2 * 16 <= T) s *= 2;
  return s;
}

template <typename T>
__global__ __launch_bounds__(512) void seqmean16_kernel(const T* a, long long lda, int rpb, long long bstride, int M, int K, int seg_rows,
                                                        const int* valid_rows, T* mean16, long long ldm) {
  // workgroup = (512-column slice, sequence), 8 waves; wave w takes sample rows w, w + 8, w + 16, w + 24 — a sequence has at most 32
  // samples (seq_sample_stride leaves 16 .. 31 full strides, plus one when T - s/2 is not a multiple of s: the 8 waves x 4 slots
  // cover exactly that, mer_seq_bias checks it) — and issues its (up to four) row loads back to back before it touches the data: the
  // launch is one memory round trip, not one per row.  A wave-load = one row x 1 KiB (whole lines).
  typedef typename T16<T>::v8 v8;
  __shared__ long long red[8][512];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int col = blockIdx.x * 512 + lane * 8;
  const int seq = blockIdx.y;
  const bool cin = col < K;
  const int stride = seq_sample_stride(seg_rows), half = stride >> 1;
  int valid = seg_rows;
  if (valid_rows) valid = valid_rows[seq] < valid ? valid_rows[seq] : valid;
  if ((long long)seq * seg_rows + valid > M) valid = M - seq * seg_rows;     // last, partial sequence of the plane
  const int total = valid > half ? (valid - half + stride - 1) / stride : 0;   // samples half, half + stride, ... below valid
  v8 x[4];
  const v8 z = {};
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const int i = wave + 8 * u;
    x[u] = z;
    if (cin && i < total) {
      const int r = seq * seg_rows + half + i * stride;
      const long long off = rpb > 0 ? (long long)(r / rpb) * bstride + (long long)(r % rpb) * lda : (long long)r * lda;
      x[u] = *reinterpret_cast<const v8*>(a + off + col);
    }
  }
  long long s[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) s[j] = 0;
#pragma unroll
  for (int u = 0; u < 4; ++u)
#pragma unroll
    for (int j = 0; j < 8; ++j) {      // (a slot without a sample holds zeros: adds nothing)
      const float v = fminf(fmaxf(T16<T>::to_f32(x[u][j]), -131000.f), 131000.f);
      s[j] += (long long)__float2int_rn(v * CM_FIX);
    }
#pragma unroll
  for (int j = 0; j < 8; ++j) red[wave][lane * 8 + j] = s[j];
  __syncthreads();
  // (a sequence shorter than stride / 2 rows has no sample: its correction row is the plain bias)
  const double inv = total > 0 ? 1.0 / ((double)total * (double)CM_FIX) : 0.0;
  const int c = threadIdx.x;
  long long t = 0;
#pragma unroll
  for (int w = 0; w < 8; ++w) t += red[w][c];       // integer sums: any order gives the same bits
  const int cc = blockIdx.x * 512 + c;
  if (cc < K) mean16[(long long)seq * ldm + cc] = T16<T>::from_f32((float)((double)t * inv));
}

template <typename T>
__global__ __launch_bounds__(1024) void seqbias_kernel(const T* mean16, long long ldm, int nseq, int K, const T* w_lo, long long ldw,
                                                       const float* bias, int N, int n_first, float* tab, long long ldt) {
  // workgroup = 64 sequences x 16 columns; its 16 waves split K (a 3072-deep dot product is 96 dependent L2 / HBM round trips for
  // one wave: this launch is latency, not bandwidth), up to 4 k-steps of loads in flight per wave, partial tiles summed in a fixed
  // tree through LDS (the row of a sequence does not depend on the batch around it: 16-row blocks beyond nseq are skipped, which
  // changes nothing for the rows that exist)
  typedef typename T16<T>::v8 v8;
  __shared__ float red[16][64][16];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int li = lane & 15, lg = lane >> 4;
  const int n0 = blockIdx.x * 16, s0 = blockIdx.y * 64;
  const int n = n0 + li < N ? n0 + li : N - 1;
  const int ksteps = (K + 31) / 32, per = (ksteps + 15) / 16;
  // columns below n_first take the plain bias (the Q | K columns of a fused QKV weight: their rounding only perturbs softmax logits)
  const bool plain = n0 + 16 <= n_first;
  const int mts = nseq - s0 >= 64 ? 4 : (nseq - s0 + 15) / 16;
  if (!plain) {
    const int k_lo = wave * per * 32, k_hi = (wave + 1) * per * 32 < K ? (wave + 1) * per * 32 : K;
    const T* wr = w_lo + (long long)n * ldw + lg * 8;
    const T* mr[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int sq = s0 + i * 16 + li;
      mr[i] = mean16 + (long long)(sq < nseq ? sq : nseq - 1) * ldm + lg * 8;
    }
    f32x4 acc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    const v8 z = {};
    for (int k0 = k_lo; k0 < k_hi; k0 += 128) {
      v8 wf[4], mf[4][4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int k = k0 + u * 32;
        const bool kin = k < k_hi && k + lg * 8 < K;     // K % 8 == 0: a lane's 8 elements are in or out together
        wf[u] = kin ? *reinterpret_cast<const v8*>(wr + k) : z;
#pragma unroll
        for (int i = 0; i < 4; ++i) mf[u][i] = (kin && i < mts) ? *reinterpret_cast<const v8*>(mr[i] + k) : z;
      }
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i] = T16<T>::mfma(mf[u][i], wf[u], acc[i]);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) red[wave][i * 16 + lg * 4 + r][li] = acc[i][r];
  }
  __syncthreads();
  const int sl = threadIdx.x >> 4, cl = threadIdx.x & 15;
  const int sq = s0 + sl, c = n0 + cl;
  if (sq < nseq && c < N) {
    float p[16];
#pragma unroll
    for (int w = 0; w < 16; ++w) p[w] = plain ? 0.f : red[w][sl][cl];
#pragma unroll
    for (int d = 1; d < 16; d *= 2)
#pragma unroll
      for (int w = 0; w < 16; w += 2 * d) p[w] += p[w + d];
    tab[(long long)sq * ldt + c] = p[0] + (bias ? bias[c] : 0.f);
  }
}

}  // namespace mer

extern "C" long long mer_seq_bias_scratch_bytes(int nseq, int K) {
  if (nseq <= 0 || K <= 0) return 0;
  return ((long long)nseq * K * 2 + 255) / 256 * 256;   // the 16-bit mean plane [nseq, K]
}

extern "C" int mer_seq_bias(const void* a, int dtype, long long lda, int a_rows_per_batch, long long a_batch_stride, int M, int K,
                            int seg_rows, const int* valid_rows, const void* w_lo, long long ldw, const float* bias, int N, int n_first,
                            void* scratch, float* table, long long ldt, mer_stream_t stream) {
  using namespace mer;
  MER_REQUIRE(a && w_lo && scratch && table && M > 0 && K > 0 && N > 0 && seg_rows > 0 && n_first >= 0 && n_first % 16 == 0, MER_EINVAL, "mer_seq_bias: bad argument");
  MER_REQUIRE(K % 8 == 0 && lda % 8 == 0 && ldw % 8 == 0 && a_batch_stride % 8 == 0 && ((((uintptr_t)a | (uintptr_t)w_lo | (uintptr_t)scratch) & 15) == 0), MER_ESHAPE,
              "mer_seq_bias: K, lda, ldw, a_batch_stride must be multiples of 8 and the planes 16-byte aligned");
  MER_REQUIRE(ldt >= N, MER_ESHAPE, "mer_seq_bias: ldt < N");
  MER_REQUIRE(dtype == MER_DT_F16 || dtype == MER_DT_BF16, MER_EINVAL, "mer_seq_bias: bad dtype");
  hipStream_t st = (hipStream_t)stream;
  const int nseq = (int)cdiv(M, seg_rows);
  MER_REQUIRE(nseq <= 65535, MER_EUNSUPPORTED, "mer_seq_bias: %d sequences > 65535", nseq);
  dim3 g1((unsigned)cdiv(K, 512), (unsigned)nseq), g2((unsigned)cdiv(N, 16), (unsigned)cdiv(nseq, 64));
  const int samples = (seg_rows + seq_sample_stride(seg_rows) - 1) / seq_sample_stride(seg_rows);
  {   // the sampled rows of a full sequence must fit seqmean16_kernel's 8 waves x 4 slots
    const int s_ = seq_sample_stride(seg_rows), h_ = s_ >> 1;
    const int total = seg_rows > h_ ? (seg_rows - h_ + s_ - 1) / s_ : 0;
    MER_REQUIRE(total <= 32, MER_EUNSUPPORTED, "mer_seq_bias: %d samples per sequence (seg_rows %d) exceed the kernel's 32 slots", total, seg_rows);
  }
  ProfScope prof("bias_corr", 2.0 * nseq * (double)(N - n_first) * K, (double)nseq * samples * K * 2 + (double)N * K * 2 + (double)nseq * N * 4, st);
  if (dtype == MER_DT_F16) {
    hipLaunchKernelGGL((seqmean16_kernel<f16>), g1, dim3(512), 0, st, (const f16*)a, lda, a_rows_per_batch, a_batch_stride, M, K, seg_rows, valid_rows, (f16*)scratch, (long long)K);
    hipLaunchKernelGGL((seqbias_kernel<f16>), g2, dim3(1024), 0, st, (const f16*)scratch, (long long)K, nseq, K, (const f16*)w_lo, ldw, bias, N, n_first, table, ldt);
  } else {
    hipLaunchKernelGGL((seqmean16_kernel<bf16>), g1, dim3(512), 0, st, (const bf16*)a, lda, a_rows_per_batch, a_batch_stride, M, K, seg_rows, valid_rows, (bf16*)scratch, (long long)K);
    hipLaunchKernelGGL((seqbias_kernel<bf16>), g2, dim3(1024), 0, st, (const bf16*)scratch, (long long)K, nseq, K, (const bf16*)w_lo, ldw, bias, N, n_first, table, ldt);
  }
  return check_launch("seq_bias");
}

extern "C" int mer_vit_assemble(const float* patch, const float* cls, const float* pos, const float* gamma,
                                const float* beta, float eps, int N, int P, int D, float* out32, void* out16_hi,
                                void* out16_lo, int dtype, mer_stream_t stream) {
  using namespace mer;
  MER_REQUIRE(patch && cls && pos && N > 0 && P > 0, MER_EINVAL, "mer_vit_assemble: bad args");
  MER_REQUIRE(D % 4 == 0, MER_ESHAPE, "mer_vit_assemble: D %% 4 != 0");
  const int nv = pick_nv(D);
  MER_REQUIRE(nv != 0, MER_EUNSUPPORTED, "mer_vit_assemble: D=%d unsupported", D);
  dim3 grid((unsigned)cdiv((long long)N * (P + 1), 4)), block(256);
  hipStream_t st = (hipStream_t)stream;
  if (dtype == MER_DT_F16) {
    MER_NV_SWITCH(nv, vit_assemble_kernel<f16, NV><<<grid, block, 0, st>>>(patch, cls, pos, gamma, beta,
                                          eps, N, P, D, out32, (f16*)out16_hi, (f16*)out16_lo));
  } else {
    MER_NV_SWITCH(nv, vit_assemble_kernel<bf16, NV><<<grid, block, 0, st>>>(patch, cls, pos, gamma, beta,
                                          eps, N, P, D, out32, (bf16*)out16_hi, (bf16*)out16_lo));
  }
  return check_launch("vit_assemble");
}

extern "C" int mer_bert_embed(const int64_t* ids, const int64_t* token_type, int B, int T, int D, const float* word,
                              const float* pos, const float* type, int pos_mode, int pad_id, const float* gamma,
                              const float* beta, float eps, float* out32, void* out16_hi, void* out16_lo, int dtype,
                              mer_stream_t stream) {
  using namespace mer;
  MER_REQUIRE(ids && word && pos && B > 0 && T > 0, MER_EINVAL, "mer_bert_embed: bad args");
  MER_REQUIRE(D % 4 == 0, MER_ESHAPE, "mer_bert_embed: D %% 4 != 0");
  const int nv = pick_nv(D);
  MER_REQUIRE(nv != 0, MER_EUNSUPPORTED, "mer_bert_embed: D=%d unsupported", D);
  dim3 grid((unsigned)cdiv((long long)B * T, 4)), block(256);
  hipStream_t st = (hipStream_t)stream;
  if (dtype == MER_DT_F16) {
    MER_NV_SWITCH(nv, bert_embed_kernel<f16, NV><<<grid, block, 0, st>>>(ids, token_type, B, T, D, word,
                                          pos, type, pos_mode, pad_id, gamma, beta, eps, out32, (f16*)out16_hi,
                                          (f16*)out16_lo));
  } else {
    MER_NV_SWITCH(nv, bert_embed_kernel<bf16, NV><<<grid, block, 0, st>>>(ids, token_type, B, T, D, word,
                                          pos, type, pos_mode, pad_id, gamma, beta, eps, out32, (bf16*)out16_hi,
                                          (bf16*)out16_lo));
  }
  return check_launch("bert_embed");
}
