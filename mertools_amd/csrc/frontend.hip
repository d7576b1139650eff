// frontend.hip — the non-GEMM front ends and the pooling tail of the three encoders.
//   mer_hubert_conv0_gn  Conv1d(1->C,k,s) + GroupNorm(C,C) + GELU     HF:hubert/modeling_hubert.py:154-175
//   mer_posconv_pack     group-major zero-padded copy for the grouped positional conv     HF:...:45-92
//   mer_vit_patchify     NCHW pixels -> [patch, c*P*P] rows (Conv2d stride==kernel as a GEMM)
//   mer_split16          fp32 -> 16-bit hi/lo planes
//   mer_sum_pool         last-4 hidden-state sum + per-clip temporal mean
//                        (extract_audio_huggingface.py:98-108, extract_text_huggingface.py:226-249)
// All HBM-bound; each element is read once with the widest load the layout allows.
#include "common.h"
#include <type_traits>

namespace mer {

// ---------------------------------------------------------------------------------------------
// conv0: y[b,c,t] = sum_j w[c,j] * x[b, t*stride + j].  Pass 1 accumulates per-(b,c) sum / sum of
// squares over t in fp64 (GroupNorm with num_groups == C is a per-channel norm over time);
// pass 2 recomputes y (10 FMAs — cheaper than a 2 x 33 MB/clip round trip through HBM),
// normalises, applies GELU and writes channels-last 16-bit planes [B, T0, C].
// ---------------------------------------------------------------------------------------------
constexpr int C0_TCH = 256;   // output frames per workgroup
constexpr int C0_KMAX = 16;

template <typename T, bool APPLY>
__global__ __launch_bounds__(256) void conv0_kernel(const float* wav, int L, int T0, const float* w, int C, int k,
                                                    int stride, const float* gamma, const float* beta, float eps,
                                                    double* stats, T* ohi, T* olo, const int* valid) {
  __shared__ float xs[C0_TCH * 8 + C0_KMAX];
  const int b = blockIdx.y;
  const int t0 = blockIdx.x * C0_TCH;
  const int nt = (T0 - t0) < C0_TCH ? (T0 - t0) : C0_TCH;
  // ragged batch: GroupNorm statistics over the row's own frames only (what the reference's batch-of-one forward sees)
  const int Tv = valid ? valid[b] : T0;
  const int nts = (Tv - t0) < nt ? ((Tv - t0) > 0 ? (Tv - t0) : 0) : nt;
  const int nin = (nt - 1) * stride + k;
  const float* xb = wav + (long long)b * L + (long long)t0 * stride;
  for (int i = threadIdx.x; i < nin; i += 256) xs[i] = xb[i];
  __syncthreads();
  for (int c = threadIdx.x; c < C; c += 256) {
    float wr[C0_KMAX];
#pragma unroll
    for (int j = 0; j < C0_KMAX; ++j) wr[j] = j < k ? w[c * k + j] : 0.f;
    if (!APPLY) {
      float s = 0.f, q = 0.f;
      for (int t = 0; t < nts; ++t) {
        float y = 0.f;
#pragma unroll
        for (int j = 0; j < C0_KMAX; ++j)
          if (j < k) y = fmaf(wr[j], xs[t * stride + j], y);
        s += y;
        q = fmaf(y, y, q);
      }
      atomicAdd(&stats[((long long)b * C + c) * 2 + 0], (double)s);
      atomicAdd(&stats[((long long)b * C + c) * 2 + 1], (double)q);
    } else {
      const double mean_d = stats[((long long)b * C + c) * 2 + 0] / (double)Tv;
      const double var_d = stats[((long long)b * C + c) * 2 + 1] / (double)Tv - mean_d * mean_d;
      const float rstd = 1.0f / sqrtf((float)(var_d > 0.0 ? var_d : 0.0) + eps);
      const float ga = gamma[c] * rstd;
      const float be = beta[c] - (float)mean_d * ga;
      for (int t = 0; t < nt; ++t) {
        float y = 0.f;
#pragma unroll
        for (int j = 0; j < C0_KMAX; ++j)
          if (j < k) y = fmaf(wr[j], xs[t * stride + j], y);
        const float z = act_apply(fmaf(y, ga, be), MER_ACT_GELU);
        T hh, ll;
        split16<T>(z, hh, ll);
        const long long o = ((long long)b * T0 + t0 + t) * C + c;
        ohi[o] = hh;
        if (olo) olo[o] = ll;
      }
    }
  }
}

// Fast path (k <= 10, C % 8 == 0, C >= 65).  The GroupNorm statistics do not need the conv output at all: with
// S_j = sum_t x[t*stride+j] and R_jj' = sum_t x[t*stride+j] x[t*stride+j'] (10 + 55 numbers per clip),
// sum_t y_c = w_c . S and sum_t y_c^2 = w_c' R w_c.  conv0_moments_kernel accumulates S and R in fp64 (the products of two
// fp32 values are exact there, so a high-pass filter on a strongly correlated signal — w'Rw orders of magnitude below
// |w|^2 tr R — loses nothing), one partial per (clip, chunk) summed in a fixed order: no atomics, bit-reproducible.
// conv0_affine_kernel turns them into the per-(clip, channel) scale/shift; the apply kernel then computes the conv ONCE.
// `stats` ([B][2C] doubles) holds per clip: [0, nchunk*65) the partial moments, [C, 2C) the 2C fp32 scale/shift values.
constexpr int C0_KF = 10;
constexpr int C0_NM = C0_KF + C0_KF * (C0_KF + 1) / 2;
constexpr int C0_MCH = 2048;   // frames per moments workgroup (when the stats buffer has room for that many partials)

__global__ __launch_bounds__(256) void conv0_moments_kernel(const float* wav, int L, int T0, int C, int k, int stride, int nchunk,
                                                            double* stats, const int* valid) {
  __shared__ double red[4][C0_NM];
  const int b = blockIdx.y, ch = blockIdx.x;
  const int Tv = valid ? valid[b] : T0;   // ragged batch: statistics over the row's own frames only
  const int per = (T0 + nchunk - 1) / nchunk;
  const int tb = ch * per;
  const int te = (tb + per) < Tv ? (tb + per) : Tv;
  const float* xb = wav + (long long)b * L;
  double m[C0_NM];
#pragma unroll
  for (int i = 0; i < C0_NM; ++i) m[i] = 0.0;
  for (int t = tb + (int)threadIdx.x; t < te; t += 256) {
    double xv[C0_KF];
#pragma unroll
    for (int j = 0; j < C0_KF; ++j) xv[j] = j < k ? (double)xb[(long long)t * stride + j] : 0.0;
#pragma unroll
    for (int j = 0; j < C0_KF; ++j) m[j] += xv[j];
    int i = C0_KF;
#pragma unroll
    for (int j = 0; j < C0_KF; ++j)
#pragma unroll
      for (int j2 = j; j2 < C0_KF; ++j2, ++i) m[i] = fma(xv[j], xv[j2], m[i]);
  }
#pragma unroll
  for (int i = 0; i < C0_NM; ++i) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) m[i] += __shfl_xor(m[i], off);
  }
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  if (lane == 0) {
#pragma unroll
    for (int i = 0; i < C0_NM; ++i) red[wv][i] = m[i];
  }
  __syncthreads();
  if (threadIdx.x < C0_NM)
    stats[(long long)b * 2 * C + (long long)ch * C0_NM + threadIdx.x] =
        (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
}

__global__ __launch_bounds__(256) void conv0_affine_kernel(const float* w, int C, int k, const float* gamma, const float* beta, float eps,
                                                           int T0, int nchunk, double* stats, const int* valid) {
  __shared__ double mo[C0_NM];
  const int b = blockIdx.x;
  const double Tv = (double)(valid ? valid[b] : T0);
  double* sb = stats + (long long)b * 2 * C;
  if (threadIdx.x < C0_NM) {
    double a = 0.0;
    for (int ch = 0; ch < nchunk; ++ch) a += sb[ch * C0_NM + threadIdx.x];
    mo[threadIdx.x] = a;
  }
  __syncthreads();
  float* gb = reinterpret_cast<float*>(sb + C);
  for (int c = threadIdx.x; c < C; c += 256) {
    double wr[C0_KF];
#pragma unroll
    for (int j = 0; j < C0_KF; ++j) wr[j] = j < k ? (double)w[c * k + j] : 0.0;
    double s = 0.0, q = 0.0;
#pragma unroll
    for (int j = 0; j < C0_KF; ++j) s = fma(wr[j], mo[j], s);
    int i = C0_KF;
#pragma unroll
    for (int j = 0; j < C0_KF; ++j)
#pragma unroll
      for (int j2 = j; j2 < C0_KF; ++j2) {
        const double ww = wr[j] * wr[j2];
        q = fma(j2 == j ? ww : 2.0 * ww, mo[i], q);
        ++i;
      }
    const double mean_d = s / Tv;
    const double var_d = q / Tv - mean_d * mean_d;
    const float rstd = 1.0f / sqrtf((float)(var_d > 0.0 ? var_d : 0.0) + eps);
    const float ga = gamma[c] * rstd;
    gb[c] = ga;
    gb[C + c] = beta[c] - (float)mean_d * ga;
  }
}

// Apply pass: a thread owns 8 CONSECUTIVE channels (80 weights in registers) and every 4th frame of the workgroup's
// 256-frame chunk, so it writes one 16-byte store per frame and lane — a wave emits whole 1 KiB channel rows (C = 512)
// instead of 2-byte scattered stores.
template <typename T>
__global__ __launch_bounds__(256) void conv0_fast_kernel(const float* wav, int L, int T0, const float* w, int C, int k,
                                                         int stride, const double* stats, T* ohi, T* olo) {
  __shared__ float xs[C0_TCH * 8 + C0_KMAX];
  const int b = blockIdx.y;
  const int t0 = blockIdx.x * C0_TCH;
  const int nt = (T0 - t0) < C0_TCH ? (T0 - t0) : C0_TCH;
  const int nin = (nt - 1) * stride + k;
  const float* xb = wav + (long long)b * L + (long long)t0 * stride;
  for (int i = threadIdx.x; i < nin; i += 256) xs[i] = xb[i];
  __syncthreads();
  const float* gb = reinterpret_cast<const float*>(stats + (long long)b * 2 * C + C);
  const int lane = threadIdx.x & 63, tq = threadIdx.x >> 6;
  for (int c0 = lane * 8; c0 < C; c0 += 512) {   // C % 8 == 0: a lane's 8 channels are all in or all out
    float wr[8][C0_KF];
#pragma unroll
    for (int c = 0; c < 8; ++c)
#pragma unroll
      for (int j = 0; j < C0_KF; ++j) wr[c][j] = j < k ? w[(c0 + c) * k + j] : 0.f;
    float ga[8], be[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      ga[c] = gb[c0 + c];
      be[c] = gb[C + c0 + c];
    }
    for (int t = tq; t < nt; t += 4) {
      float xv[C0_KF];
#pragma unroll
      for (int j = 0; j < C0_KF; ++j) xv[j] = xs[t * stride + j];
      typename T16<T>::v8 h, l;
#pragma unroll
      for (int c = 0; c < 8; ++c) {
        float y = 0.f;
#pragma unroll
        for (int j = 0; j < C0_KF; ++j) y = fmaf(wr[c][j], xv[j], y);
        const float z = act_apply(fmaf(y, ga[c], be[c]), MER_ACT_GELU);
        T hh, ll;
        split16<T>(z, hh, ll);
        h[c] = hh;
        l[c] = ll;
      }
      const long long o = ((long long)b * T0 + t0 + t) * C + c0;
      *reinterpret_cast<typename T16<T>::v8*>(ohi + o) = h;
      if (olo) *reinterpret_cast<typename T16<T>::v8*>(olo + o) = l;
    }
  }
}

// conv0 without normalisation (feat_extract_norm == "layer": HuBERT-large / wav2vec2-large, HF:hubert/modeling_hubert.py:127-151):
// y[b,t,c] = bias[c] + sum_j w[c,j] x[b, t*stride + j], fp32 channels-last; LayerNorm + GELU follow as mer_layernorm.
__global__ __launch_bounds__(256) void conv0_plain_kernel(const float* wav, int L, int T0, const float* w, const float* bias, int C,
                                                          int k, int stride, float* out) {
  __shared__ float xs[C0_TCH * 8 + C0_KMAX];
  const int b = blockIdx.y;
  const int t0 = blockIdx.x * C0_TCH;
  const int nt = (T0 - t0) < C0_TCH ? (T0 - t0) : C0_TCH;
  const int nin = (nt - 1) * stride + k;
  const float* xb = wav + (long long)b * L + (long long)t0 * stride;
  for (int i = threadIdx.x; i < nin; i += 256) xs[i] = xb[i];
  __syncthreads();
  for (int c = threadIdx.x; c < C; c += 256) {
    float wr[C0_KMAX];
#pragma unroll
    for (int j = 0; j < C0_KMAX; ++j) wr[j] = j < k ? w[c * k + j] : 0.f;
    const float bc = bias ? bias[c] : 0.f;
    for (int t = 0; t < nt; ++t) {
      float y = 0.f;
#pragma unroll
      for (int j = 0; j < C0_KMAX; ++j)
        if (j < k) y = fmaf(wr[j], xs[t * stride + j], y);
      out[((long long)b * T0 + t0 + t) * C + c] = y + bc;
    }
  }
}

// x [B,T,D] fp32 -> out [B,G,T+K,Dg]; 4 channels per thread.
template <typename T>
__global__ void posconv_pack_kernel(const float* x, int B, int Tn, int D, int G, int K, T* ohi, T* olo, const int* valid) {
  const int Dg = D / G, TPad = Tn + K, half = K / 2;
  const long long total4 = (long long)B * G * TPad * Dg / 4;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total4; i += (long long)gridDim.x * blockDim.x) {
    const long long e = i * 4;
    const int c = (int)(e % Dg);
    const long long r = e / Dg;
    const int tp = (int)(r % TPad);
    const long long bg = r / TPad;
    const int g = (int)(bg % G), b = (int)(bg / G);
    const int t = tp - half;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    // ragged batch: frames past the row's own length read as zeros — exactly the conv's zero padding in a batch-of-one forward
    if (t >= 0 && t < (valid ? valid[b] : Tn)) v = *reinterpret_cast<const f32x4*>(x + ((long long)b * Tn + t) * D + g * Dg + c);
    typename T16<T>::v4 h, l;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      T hh, ll;
      split16<T>(v[j], hh, ll);
      h[j] = hh;
      l[j] = ll;
    }
    *reinterpret_cast<typename T16<T>::v4*>(ohi + e) = h;
    if (olo) *reinterpret_cast<typename T16<T>::v4*>(olo + e) = l;
  }
}

// pixels [N,C,H,W] -> rows [N*gh*gw, C*P*P]; 4 pixels (along j) per thread.
template <typename T>
__global__ void patchify_kernel(const float* px, int N, int C, int H, int W, int P, T* ohi, T* olo) {
  const int gh = H / P, gw = W / P, cols = C * P * P;
  const long long total4 = (long long)N * gh * gw * cols / 4;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total4; i += (long long)gridDim.x * blockDim.x) {
    const long long e = i * 4;
    const int col = (int)(e % cols);
    const long long row = e / cols;
    const int c = col / (P * P), ii = (col % (P * P)) / P, j = col % P;
    const int pp = (int)(row % (gh * gw));
    const long long n = row / (gh * gw);
    const int py = pp / gw, pxx = pp % gw;
    const f32x4 v = *reinterpret_cast<const f32x4*>(px + ((n * C + c) * H + (py * P + ii)) * (long long)W + pxx * P + j);
    typename T16<T>::v4 h, l;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      T hh, ll;
      split16<T>(v[q], hh, ll);
      h[q] = hh;
      l[q] = ll;
    }
    *reinterpret_cast<typename T16<T>::v4*>(ohi + e) = h;
    if (olo) *reinterpret_cast<typename T16<T>::v4*>(olo + e) = l;
  }
}

// video [B,F,C,H,W] -> tubelet rows [B*(F/ts)*gh*gw, C*ts*P*P] in the (c, dt, i, j) order of a flattened Conv3d weight
template <typename T>
__global__ void video_patchify_kernel(const float* px, int B, int F, int C, int H, int W, int P, int ts, T* ohi, T* olo) {
  const int gh = H / P, gw = W / P, nt = F / ts, cols = C * ts * P * P;
  const long long total4 = (long long)B * nt * gh * gw * cols / 4;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total4; i += (long long)gridDim.x * blockDim.x) {
    const long long e = i * 4;
    const int col = (int)(e % cols);
    const long long row = e / cols;
    const int c = col / (ts * P * P), dt = (col / (P * P)) % ts, ii = (col % (P * P)) / P, j = col % P;
    const int pp = (int)(row % (gh * gw));
    const long long bt = row / (gh * gw);
    const int tt = (int)(bt % nt);
    const long long b = bt / nt;
    const int py = pp / gw, pxx = pp % gw;
    const f32x4 v = *reinterpret_cast<const f32x4*>(px + (((b * F + tt * ts + dt) * C + c) * H + (py * P + ii)) * (long long)W + pxx * P + j);
    typename T16<T>::v4 h, l;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      T hh, ll;
      split16<T>(v[q], hh, ll);
      h[q] = hh;
      l[q] = ll;
    }
    *reinterpret_cast<typename T16<T>::v4*>(ohi + e) = h;
    if (olo) *reinterpret_cast<typename T16<T>::v4*>(olo + e) = l;
  }
}

// x[r, :] += pos[r % P, :]   (fixed sin-cos position table of VideoMAE, HF:videomae/modeling_videomae.py:80-118)
__global__ void add_pos_kernel(float* x, const float* pos, long long rows, int P, int D) {
  const long long n4 = rows * D / 4;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x) {
    const long long e = i * 4;
    const long long r = e / D;
    const int c = (int)(e % D);
    f32x4 a = *reinterpret_cast<f32x4*>(x + e);
    a = a + *reinterpret_cast<const f32x4*>(pos + (r % P) * D + c);
    *reinterpret_cast<f32x4*>(x + e) = a;
  }
}

// generic patchify (any P, e.g. CLIP-L/14): one element per thread, rows padded with zeros to ldo columns (ldo % 8 == 0)
template <typename T>
__global__ void patchify_generic_kernel(const float* px, int N, int C, int H, int W, int P, int ldo, T* ohi, T* olo) {
  const int gh = H / P, gw = W / P, cols = C * P * P;
  const long long total = (long long)N * gh * gw * ldo;
  for (long long e = blockIdx.x * (long long)blockDim.x + threadIdx.x; e < total; e += (long long)gridDim.x * blockDim.x) {
    const int col = (int)(e % ldo);
    const long long row = e / ldo;
    float v = 0.f;
    if (col < cols) {
      const int c = col / (P * P), ii = (col % (P * P)) / P, j = col % P;
      const int pp = (int)(row % (gh * gw));
      const long long n = row / (gh * gw);
      v = px[((n * C + c) * H + ((pp / gw) * P + ii)) * (long long)W + (pp % gw) * P + j];
    }
    T hh, ll;
    split16<T>(v, hh, ll);
    ohi[e] = hh;
    if (olo) olo[e] = ll;
  }
}

template <typename T>
__global__ void split16_kernel(const float* x, T* hi, T* lo, long long n) {
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    T hh, ll;
    split16<T>(x[i], hh, ll);
    hi[i] = hh;
    if (lo) lo[i] = ll;
  }
}

__global__ void sum4_kernel(const float* h0, const float* h1, const float* h2, const float* h3, long long n4, float* out) {
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x) {
    f32x4 s = reinterpret_cast<const f32x4*>(h0)[i];
    if (h1) s = s + reinterpret_cast<const f32x4*>(h1)[i];
    if (h2) s = s + reinterpret_cast<const f32x4*>(h2)[i];
    if (h3) s = s + reinterpret_cast<const f32x4*>(h3)[i];
    reinterpret_cast<f32x4*>(out)[i] = s;
  }
}

// grid (nseg, ceil(D/64)); block 256 = 4 waves; wave w sums rows w, w+4, ... of the segment for
// 64 consecutive columns (coalesced 256-B row slices), partials combined through LDS.
__global__ __launch_bounds__(256) void seg_mean_kernel(const float* h0, const float* h1, const float* h2,
                                                       const float* h3, int D, const int* seg_start,
                                                       const int* seg_len, float* out) {
  __shared__ float part[4][64];
  const int seg = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int col = blockIdx.y * 64 + lane;
  const int r0 = seg_start[seg], n = seg_len[seg];
  float acc = 0.f;
  if (col < D) {
    for (int r = wave; r < n; r += 4) {
      const long long o = (long long)(r0 + r) * D + col;
      float s = h0[o];
      if (h1) s += h1[o];
      if (h2) s += h2[o];
      if (h3) s += h3[o];
      acc += s;
    }
  }
  part[wave][lane] = acc;
  __syncthreads();
  if (wave == 0 && col < D) {
    const float tot = (part[0][lane] + part[1][lane]) + (part[2][lane] + part[3][lane]);
    out[(long long)seg * D + col] = n > 0 ? tot / (float)n : 0.f;
  }
}

// out[n, col] = scale * sum_t x[n, t, col]: one workgroup per (n, 64 columns), 4 waves stride the tokens
__global__ __launch_bounds__(256) void token_reduce_kernel(const float* x, int T, int D, float scale, float* out) {
  __shared__ float part[4][64];
  const int n = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int col = blockIdx.y * 64 + lane;
  float acc = 0.f;
  if (col < D)
    for (int t = wave; t < T; t += 4) acc += x[((long long)n * T + t) * D + col];
  part[wave][lane] = acc;
  __syncthreads();
  if (wave == 0 && col < D) out[(long long)n * D + col] = scale * ((part[0][lane] + part[1][lane]) + (part[2][lane] + part[3][lane]));
}

// WavLM gate (HF:wavlm/modeling_wavlm.py WavLMAttention.forward steps 1-3): per (row m = (b,t), head h) an 8-way linear on
// the head's 64-wide slice of the attention input, two groups of four summed, sigmoids, gate_a * (gate_b * const_h - 1) + 2.
// One wave per (m, h): lane d holds x[m, 64h + d]; the eight dot products are wave reductions.
__global__ __launch_bounds__(256) void wavlm_gate_kernel(const float* x, long long ldx, const float* w, const float* bvec,
                                                         const float* cst, int M, int Tn, int H, float* gate) {
  const long long wid = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (wid >= (long long)M * H) return;
  const int lane = threadIdx.x & 63;
  const int h = (int)(wid % H);
  const long long m = wid / H;
  const float xv = x[m * ldx + h * 64 + lane];
  float p[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) p[j] = wave_sum(xv * w[j * 64 + lane]) + bvec[j];
  if (lane == 0) {
    const float a = (p[0] + p[1]) + (p[2] + p[3]), c = (p[4] + p[5]) + (p[6] + p[7]);
    const float ga = 1.f / (1.f + expf(-a)), gb = 1.f / (1.f + expf(-c));
    const long long bi = m / Tn, t = m % Tn;
    gate[(bi * H + h) * Tn + t] = ga * (gb * cst[h] - 1.f) + 2.f;
  }
}

// ---- host pre-processing moved to the GPU ----
// Wav2Vec2FeatureExtractor's zero-mean / unit-variance normalisation (reference extract_audio_huggingface.py:94;
// HF:wav2vec2/feature_extraction_wav2vec2.py:78-97) straight from 16-bit PCM (x = pcm / 32768, what soundfile returns) or
// fp32: one workgroup per row, two passes (mean, then centred variance), fp32 with per-thread fp64 partials.
template <typename IN>
__global__ __launch_bounds__(1024) void wave_normalize_kernel(const IN* x, long long ldx, int L, int do_norm, float* out, long long ldo) {
  __shared__ double red[16];
  const int row = blockIdx.x, tid = threadIdx.x;
  const IN* xr = x + (long long)row * ldx;
  float* o = out + (long long)row * ldo;
  const float sc = std::is_same<IN, short>::value ? 1.0f / 32768.0f : 1.0f;
  auto block_sum = [&](double v) -> double {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    __syncthreads();
    if ((tid & 63) == 0) red[tid >> 6] = v;
    __syncthreads();
    double t = 0.0;
    for (int i = 0; i < 16; ++i) t += red[i];
    return t;
  };
  if (!do_norm) {
    for (int i = tid; i < L; i += 1024) o[i] = (float)xr[i] * sc;
    return;
  }
  double s = 0.0;
  for (int i = tid; i < L; i += 1024) s += (double)((float)xr[i] * sc);
  const float mean = (float)(block_sum(s) / L);
  double q = 0.0;
  for (int i = tid; i < L; i += 1024) {
    const float d = (float)xr[i] * sc - mean;
    q += (double)d * d;
  }
  const float rstd = 1.0f / sqrtf((float)(block_sum(q) / L) + 1e-7f);
  for (int i = tid; i < L; i += 1024) o[i] = ((float)xr[i] * sc - mean) * rstd;
}

// uint8 frames [N, H, W, 3] (BGR as cv2 / the reference's frame files give them, or RGB) -> fp32 [N, 3, H, W] RGB planes,
// (v / 255 - mean[c]) / std[c]: the rescale + normalise half of CLIPImageProcessor / BitImageProcessor (reference
// extract_vision_huggingface.py:29-31,116) for frames that already have the model's resolution.
__global__ void image_normalize_u8_kernel(const unsigned char* in, long long npix, int bgr, float m0, float m1, float m2,
                                          float s0, float s1, float s2, float* out) {
  const long long hw = npix;   // pixels per image
  const long long total = hw * gridDim.y;
  (void)total;
  const long long n = blockIdx.y;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < hw; i += (long long)gridDim.x * blockDim.x) {
    const unsigned char* p = in + (n * hw + i) * 3;
    const float c0 = (float)p[bgr ? 2 : 0], c1 = (float)p[1], c2 = (float)p[bgr ? 0 : 2];
    float* o = out + n * 3 * hw + i;
    o[0] = (c0 * (1.0f / 255.0f) - m0) / s0;
    o[hw] = (c1 * (1.0f / 255.0f) - m1) / s1;
    o[2 * hw] = (c2 * (1.0f / 255.0f) - m2) / s2;
  }
}

// Pillow-exact 8-bit bicubic resampling, one axis per kernel (Resample.c: ImagingResampleHorizontal_8bpc / Vertical_8bpc):
//   out = clip8((2^21 + sum_t in[first + t] * k[t]) >> 22), int32 accumulation, k = fixed-point coefficients (22 fractional bits).
// The host supplies the per-output-pixel windows and coefficients (mertools_amd/extract/resize.py:pil_coeffs); only the
// centre-crop region is computed: the horizontal pass writes rows y0..y1-1 x the cropped columns, the vertical pass the
// cropped rows.  One thread per output pixel (3 channels).
__device__ __forceinline__ unsigned char pil_clip8(int ss) {
  const int v = ss >> 22;   // arithmetic shift, as Pillow's clip8 lookup index
  return (unsigned char)(v < 0 ? 0 : (v > 255 ? 255 : v));
}

__global__ void pil_resize_h_kernel(const unsigned char* in, int N, int H, int W, int left, int crop_w, const int* xb, const int* xk, int ksize,
                                    int y0, int rows, unsigned char* tmp) {
  const long long total = (long long)N * rows * crop_w;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int xo = (int)(i % crop_w);
    const int yr = (int)((i / crop_w) % rows);
    const long long n = i / ((long long)crop_w * rows);
    const int ox = left + xo, first = xb[2 * ox], cnt = xb[2 * ox + 1];
    const int* k = xk + (long long)ox * ksize;
    const unsigned char* p = in + ((n * H + (y0 + yr)) * W + first) * 3;
    int s0 = 1 << 21, s1 = 1 << 21, s2 = 1 << 21;
    for (int t = 0; t < cnt; ++t) {
      const int kv = k[t];
      s0 += (int)p[3 * t] * kv;
      s1 += (int)p[3 * t + 1] * kv;
      s2 += (int)p[3 * t + 2] * kv;
    }
    unsigned char* o = tmp + i * 3;
    o[0] = pil_clip8(s0); o[1] = pil_clip8(s1); o[2] = pil_clip8(s2);
  }
}

__global__ void pil_resize_v_kernel(const unsigned char* tmp, int N, int rows, int crop_w, int top, int crop_h, const int* yb, const int* yk, int ksize,
                                    int y0, unsigned char* out) {
  const long long total = (long long)N * crop_h * crop_w;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int xo = (int)(i % crop_w);
    const int yo = (int)((i / crop_w) % crop_h);
    const long long n = i / ((long long)crop_w * crop_h);
    const int oy = top + yo, first = yb[2 * oy], cnt = yb[2 * oy + 1];
    const int* k = yk + (long long)oy * ksize;
    const unsigned char* p = tmp + ((n * rows + (first - y0)) * crop_w + xo) * 3;
    int s0 = 1 << 21, s1 = 1 << 21, s2 = 1 << 21;
    for (int t = 0; t < cnt; ++t) {
      const int kv = k[t];
      const unsigned char* q = p + (long long)t * crop_w * 3;
      s0 += (int)q[0] * kv;
      s1 += (int)q[1] * kv;
      s2 += (int)q[2] * kv;
    }
    unsigned char* o = out + i * 3;
    o[0] = pil_clip8(s0); o[1] = pil_clip8(s1); o[2] = pil_clip8(s2);
  }
}

// SwiGLU gate of DINOv2-giant's feed-forward: 4 elements per thread, fp32 in, 16-bit planes out
template <typename T>
__global__ void swiglu_kernel(const float* y, long long ldy, int M, int F, T* ohi, T* olo) {
  const long long total4 = (long long)M * F / 4;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total4; i += (long long)gridDim.x * blockDim.x) {
    const long long e = i * 4, m = e / F;
    const int j = (int)(e % F);
    const f32x4 a = *reinterpret_cast<const f32x4*>(y + m * ldy + j);
    const f32x4 b = *reinterpret_cast<const f32x4*>(y + m * ldy + F + j);
    typename T16<T>::v4 h, l;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float v = a[r] / (1.0f + expf(-a[r])) * b[r];
      T hh, ll;
      split16<T>(v, hh, ll);
      h[r] = hh;
      l[r] = ll;
    }
    *reinterpret_cast<typename T16<T>::v4*>(ohi + e) = h;
    if (olo) *reinterpret_cast<typename T16<T>::v4*>(olo + e) = l;
  }
}

// frames left of `valid_samples[b]` input samples after the first conv / after the whole stack (valid convs: floor((n - k) / s) + 1)
struct ConvGeom { int k[MER_MAX_CONV], s[MER_MAX_CONV]; };   // passed by value: no device copy of the table
// all_len (or NULL): the counts after EVERY conv layer, [n_conv][B] (row i = frames of each clip after conv i)
__global__ void hubert_valid_frames_kernel(const int* valid_samples, int B, int L, int n_conv, ConvGeom cg, int* t0_len, int* tn_len, int* all_len) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  int n = valid_samples[b];
  n = n < L ? n : L;
  for (int i = 0; i < n_conv; ++i) {
    const int k = cg.k[i], s = cg.s[i];
    n = n >= k ? (n - k) / s + 1 : 0;
    if (i == 0 && t0_len) t0_len[b] = n > 0 ? n : 1;   // (a clip shorter than the receptive field is rejected by the caller; stay finite)
    if (all_len) all_len[(long long)i * B + b] = (i == 0 && n <= 0) ? 1 : n;
  }
  if (tn_len) tn_len[b] = n;
}

static inline unsigned grid_for(long long n, int block) {
  long long g = cdiv(n, block);
  return (unsigned)(g < 1 ? 1 : (g > 16384 ? 16384 : g));
}

}  // namespace mer

extern "C" int mer_hubert_conv0_gn(const float* wav, int B, int L, const float* w, int C, int k, int stride,
                                   const float* gamma, const float* beta, float eps, double* stats, void* out_hi,
                                   void* out_lo, int dtype, mer_stream_t stream) {
  return mer_hubert_conv0_gn_ragged(wav, B, L, w, C, k, stride, gamma, beta, eps, stats, out_hi, out_lo, dtype, nullptr, stream);
}

extern "C" int mer_hubert_valid_frames(const int* valid_samples, int B, int L, int n_conv, const int* kernels, const int* strides,
                                       int* t0_len, int* tn_len, mer_stream_t stream) {
  using namespace mer;
  MER_REQUIRE(valid_samples && kernels && strides && t0_len && tn_len && B > 0 && n_conv >= 1 && n_conv <= MER_MAX_CONV, MER_EINVAL,
              "mer_hubert_valid_frames: bad argument");
  ConvGeom cg;
  for (int i = 0; i < MER_MAX_CONV; ++i) { cg.k[i] = i < n_conv ? kernels[i] : 1; cg.s[i] = i < n_conv ? strides[i] : 1; }
  hipLaunchKernelGGL(hubert_valid_frames_kernel, dim3((unsigned)cdiv(B, 64)), dim3(64), 0, (hipStream_t)stream, valid_samples, B, L, n_conv,
                     cg, t0_len, tn_len, (int*)nullptr);
  return check_launch("hubert_valid_frames");
}

extern "C" int mer_hubert_valid_frames_all(const int* valid_samples, int B, int L, int n_conv, const int* kernels, const int* strides,
                                           int* lens, mer_stream_t stream) {
  using namespace mer;
  MER_REQUIRE(valid_samples && kernels && strides && lens && B > 0 && n_conv >= 1 && n_conv <= MER_MAX_CONV, MER_EINVAL,
              "mer_hubert_valid_frames_all: bad argument");
  ConvGeom cg;
  for (int i = 0; i < MER_MAX_CONV; ++i) { cg.k[i] = i < n_conv ? kernels[i] : 1; cg.s[i] = i < n_conv ? strides[i] : 1; }
  hipLaunchKernelGGL(hubert_valid_frames_kernel, dim3((unsigned)cdiv(B, 64)), dim3(64), 0, (hipStream_t)stream, valid_samples, B, L, n_conv,
                     cg, (int*)nullptr, (int*)nullptr, lens);
  return check_launch("hubert_valid_frames");
}

extern "C" int mer_hubert_conv0_gn_ragged(const float* wav, int B, int L, const float* w, int C, int k, int stride,
                                          const float* gamma, const float* beta, float eps, double* stats, void* out_hi,
                                          void* out_lo, int dtype, const int* valid_frames, mer_stream_t stream) {
  using namespace mer;
  MER_REQUIRE(wav && w && gamma && beta && stats && out_hi, MER_EINVAL, "mer_hubert_conv0_gn: null pointer");
  MER_REQUIRE(k >= 1 && k <= C0_KMAX && stride >= 1 && stride <= 8, MER_EUNSUPPORTED,
              "mer_hubert_conv0_gn: kernel %d / stride %d unsupported", k, stride);
  MER_REQUIRE(L >= k, MER_ESHAPE, "mer_hubert_conv0_gn: L=%d < k=%d", L, k);
  const int T0 = (L - k) / stride + 1;
  hipStream_t st = (hipStream_t)stream;
  dim3 grid((unsigned)cdiv(T0, C0_TCH), B), block(256);
  ProfScope prof("hubert_conv0_gn", 2.0 * (double)B * T0 * C * k, (double)B * L * 4 * 2 + (double)B * T0 * C * (out_lo ? 4 : 2), st);
  if (k <= C0_KF && C % 8 == 0 && C >= C0_NM) {
    int nchunk = (int)cdiv(T0, C0_MCH);
    if (nchunk > C / C0_NM) nchunk = C / C0_NM;
    hipLaunchKernelGGL(conv0_moments_kernel, dim3((unsigned)nchunk, B), block, 0, st, wav, L, T0, C, k, stride, nchunk, stats, valid_frames);
    hipLaunchKernelGGL(conv0_affine_kernel, dim3((unsigned)B), block, 0, st, w, C, k, gamma, beta, eps, T0, nchunk, stats, valid_frames);
    if (dtype == MER_DT_F16)
      hipLaunchKernelGGL((conv0_fast_kernel<f16>), grid, block, 0, st, wav, L, T0, w, C, k, stride, (const double*)stats, (f16*)out_hi, (f16*)out_lo);
    else
      hipLaunchKernelGGL((conv0_fast_kernel<bf16>), grid, block, 0, st, wav, L, T0, w, C, k, stride, (const double*)stats, (bf16*)out_hi, (bf16*)out_lo);
    return check_launch("hubert_conv0_gn");
  }
  hipError_t e = hipMemsetAsync(stats, 0, sizeof(double) * 2 * (size_t)B * C, st);
  MER_REQUIRE(e == hipSuccess, MER_ELAUNCH, "mer_hubert_conv0_gn: memset failed: %s", hipGetErrorString(e));
#define MER_CONV0(TT, AP, OH, OL) hipLaunchKernelGGL((conv0_kernel<TT, AP>), grid, block, 0, st, wav, L, T0, w, C, k, stride, gamma, beta, eps, stats, OH, OL, valid_frames)
  if (dtype == MER_DT_F16) { MER_CONV0(f16, false, (f16*)nullptr, (f16*)nullptr); MER_CONV0(f16, true, (f16*)out_hi, (f16*)out_lo); }
  else { MER_CONV0(bf16, false, (bf16*)nullptr, (bf16*)nullptr); MER_CONV0(bf16, true, (bf16*)out_hi, (bf16*)out_lo); }
#undef MER_CONV0
  return check_launch("hubert_conv0_gn");
}

extern "C" int mer_hubert_conv0_plain(const float* wav, int B, int L, const float* w, const float* bias, int C, int k, int stride,
                                      float* out, mer_stream_t stream) {
  using namespace mer;
  MER_REQUIRE(wav && w && out, MER_EINVAL, "mer_hubert_conv0_plain: null pointer");
  MER_REQUIRE(k >= 1 && k <= C0_KMAX && stride >= 1 && stride <= 8 && L >= k, MER_EUNSUPPORTED, "mer_hubert_conv0_plain: kernel/stride unsupported");
  const int T0 = (L - k) / stride + 1;
  dim3 grid((unsigned)cdiv(T0, C0_TCH), B), block(256);
  hipLaunchKernelGGL(conv0_plain_kernel, grid, block, 0, (hipStream_t)stream, wav, L, T0, w, bias, C, k, stride, out);
  return check_launch("hubert_conv0_plain");
}

extern "C" int mer_posconv_pack(const float* x, int B, int T, int D, int G, int K, void* out_hi, void* out_lo, int dtype,
                                mer_stream_t stream) {
  return mer_posconv_pack_ragged(x, B, T, D, G, K, out_hi, out_lo, dtype, nullptr, stream);
}

extern "C" int mer_posconv_pack_ragged(const float* x, int B, int T, int D, int G, int K, void* out_hi, void* out_lo, int dtype,
                                       const int* valid_frames, mer_stream_t stream) {
  using namespace mer;
  MER_REQUIRE(x && out_hi && B > 0 && T > 0, MER_EINVAL, "mer_posconv_pack: bad args");
  MER_REQUIRE(D % G == 0 && (D / G) % 8 == 0, MER_ESHAPE, "mer_posconv_pack: D/G must be a multiple of 8");
  const long long n4 = (long long)B * G * (T + K) * (D / G) / 4;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == MER_DT_F16)
    hipLaunchKernelGGL((posconv_pack_kernel<f16>), dim3(grid_for(n4, 256)), dim3(256), 0, st, x, B, T, D, G, K, (f16*)out_hi, (f16*)out_lo, valid_frames);
  else
    hipLaunchKernelGGL((posconv_pack_kernel<bf16>), dim3(grid_for(n4, 256)), dim3(256), 0, st, x, B, T, D, G, K, (bf16*)out_hi, (bf16*)out_lo, valid_frames);
  return check_launch("posconv_pack");
}

extern "C" int mer_vit_patchify(const float* pixels, int N, int C, int H, int W, int P, void* out_hi, void* out_lo,
                                int dtype, mer_stream_t stream) {
  using namespace mer;
  MER_REQUIRE(pixels && out_hi && N > 0, MER_EINVAL, "mer_vit_patchify: bad args");
  MER_REQUIRE(H % P == 0 && W % P == 0, MER_ESHAPE, "mer_vit_patchify: H=%d W=%d P=%d unsupported", H, W, P);
  hipStream_t st = (hipStream_t)stream;
  const int cols = C * P * P, ldo = (cols + 7) / 8 * 8;  // rows are zero-padded to a multiple of 8 columns
  if (P % 4 == 0 && W % 4 == 0 && ldo == cols) {
    const long long n4 = (long long)N * C * H * W / 4;
    if (dtype == MER_DT_F16)
      hipLaunchKernelGGL((patchify_kernel<f16>), dim3(grid_for(n4, 256)), dim3(256), 0, st, pixels, N, C, H, W, P, (f16*)out_hi, (f16*)out_lo);
    else
      hipLaunchKernelGGL((patchify_kernel<bf16>), dim3(grid_for(n4, 256)), dim3(256), 0, st, pixels, N, C, H, W, P, (bf16*)out_hi, (bf16*)out_lo);
  } else {
    const long long n = (long long)N * (H / P) * (W / P) * ldo;
    if (dtype == MER_DT_F16)
      hipLaunchKernelGGL((patchify_generic_kernel<f16>), dim3(grid_for(n, 256)), dim3(256), 0, st, pixels, N, C, H, W, P, ldo, (f16*)out_hi, (f16*)out_lo);
    else
      hipLaunchKernelGGL((patchify_generic_kernel<bf16>), dim3(grid_for(n, 256)), dim3(256), 0, st, pixels, N, C, H, W, P, ldo, (bf16*)out_hi, (bf16*)out_lo);
  }
  return check_launch("vit_patchify");
}

extern "C" int mer_video_patchify(const float* pixels, int B, int F, int C, int H, int W, int P, int ts, void* out_hi,
                                  void* out_lo, int dtype, mer_stream_t stream) {
  using namespace mer;
  MER_REQUIRE(pixels && out_hi && B > 0, MER_EINVAL, "mer_video_patchify: bad args");
  MER_REQUIRE(H % P == 0 && W % P == 0 && P % 4 == 0 && F % ts == 0, MER_ESHAPE, "mer_video_patchify: F=%d H=%d W=%d P=%d ts=%d unsupported", F, H, W, P, ts);
  const long long n4 = (long long)B * F * C * H * W / 4;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == MER_DT_F16)
    hipLaunchKernelGGL((video_patchify_kernel<f16>), dim3(grid_for(n4, 256)), dim3(256), 0, st, pixels, B, F, C, H, W, P, ts, (f16*)out_hi, (f16*)out_lo);
  else
    hipLaunchKernelGGL((video_patchify_kernel<bf16>), dim3(grid_for(n4, 256)), dim3(256), 0, st, pixels, B, F, C, H, W, P, ts, (bf16*)out_hi, (bf16*)out_lo);
  return check_launch("video_patchify");
}

extern "C" int mer_add_pos(float* x, const float* pos, long long rows, int P, int D, mer_stream_t stream) {
  using namespace mer;
  MER_REQUIRE(x && pos && rows > 0 && P > 0 && D % 4 == 0, MER_EINVAL, "mer_add_pos: bad args");
  hipLaunchKernelGGL(add_pos_kernel, dim3(grid_for(rows * D / 4, 256)), dim3(256), 0, (hipStream_t)stream, x, pos, rows, P, D);
  return check_launch("add_pos");
}

extern "C" int mer_split16(const float* x, void* hi, void* lo, long long n, int dtype, mer_stream_t stream) {
  using namespace mer;
  MER_REQUIRE(x && hi && n > 0, MER_EINVAL, "mer_split16: bad args");
  hipStream_t st = (hipStream_t)stream;
  if (dtype == MER_DT_F16)
    hipLaunchKernelGGL((split16_kernel<f16>), dim3(grid_for(n, 256)), dim3(256), 0, st, x, (f16*)hi, (f16*)lo, n);
  else
    hipLaunchKernelGGL((split16_kernel<bf16>), dim3(grid_for(n, 256)), dim3(256), 0, st, x, (bf16*)hi, (bf16*)lo, n);
  return check_launch("split16");
}

extern "C" int mer_sum_pool(const float* h0, const float* h1, const float* h2, const float* h3, long long M, int D,
                            float* out_frames, const int* seg_start, const int* seg_len, int nseg, float* out_pool,
                            mer_stream_t stream) {
  using namespace mer;
  MER_REQUIRE(h0 && M > 0 && D > 0, MER_EINVAL, "mer_sum_pool: bad args");
  MER_REQUIRE(D % 4 == 0, MER_ESHAPE, "mer_sum_pool: D %% 4 != 0");
  hipStream_t st = (hipStream_t)stream;
  if (out_frames) {
    const long long n4 = M * D / 4;
    hipLaunchKernelGGL(sum4_kernel, dim3(grid_for(n4, 256)), dim3(256), 0, st, h0, h1, h2, h3, n4, out_frames);
    int rc = check_launch("sum4");
    if (rc) return rc;
  }
  if (out_pool) {
    MER_REQUIRE(seg_start && seg_len && nseg > 0, MER_EINVAL, "mer_sum_pool: segments missing");
    hipLaunchKernelGGL(seg_mean_kernel, dim3(nseg, (unsigned)cdiv(D, 64)), dim3(256), 0, st, h0, h1, h2, h3, D, seg_start, seg_len, out_pool);
    return check_launch("seg_mean");
  }
  return MER_OK;
}

extern "C" int mer_token_reduce(const float* x, int N, int T, int D, float scale, float* out, mer_stream_t stream) {
  using namespace mer;
  MER_REQUIRE(x && out && N > 0 && T > 0 && D > 0, MER_EINVAL, "mer_token_reduce: bad argument");
  hipLaunchKernelGGL(token_reduce_kernel, dim3(N, (unsigned)cdiv(D, 64)), dim3(256), 0, (hipStream_t)stream, x, T, D, scale, out);
  return check_launch("token_reduce");
}

extern "C" int mer_wavlm_gate(const float* x, long long ldx, const float* w, const float* b, const float* cst, int B, int T,
                              int H, float* gate, mer_stream_t stream) {
  using namespace mer;
  MER_REQUIRE(x && w && b && cst && gate && B > 0 && T > 0 && H > 0, MER_EINVAL, "mer_wavlm_gate: bad argument");
  const long long waves = (long long)B * T * H;
  hipLaunchKernelGGL(wavlm_gate_kernel, dim3((unsigned)cdiv(waves, 4)), dim3(256), 0, (hipStream_t)stream, x, ldx, w, b, cst, B * T, T, H, gate);
  return check_launch("wavlm_gate");
}

extern "C" int mer_wave_normalize(const void* x, int is_int16, long long ldx, int B, int L, int do_normalize, float* out,
                                  long long ldo, mer_stream_t stream) {
  using namespace mer;
  MER_REQUIRE(x && out && B > 0 && L > 0 && ldx >= L && ldo >= L, MER_EINVAL, "mer_wave_normalize: bad argument");
  hipStream_t st = (hipStream_t)stream;
  if (is_int16)
    hipLaunchKernelGGL((wave_normalize_kernel<short>), dim3(B), dim3(1024), 0, st, (const short*)x, ldx, L, do_normalize, out, ldo);
  else
    hipLaunchKernelGGL((wave_normalize_kernel<float>), dim3(B), dim3(1024), 0, st, (const float*)x, ldx, L, do_normalize, out, ldo);
  return check_launch("wave_normalize");
}

extern "C" int mer_image_normalize_u8(const unsigned char* frames, int N, int H, int W, int bgr, const float* mean3,
                                      const float* std3, float* out, mer_stream_t stream) {
  using namespace mer;
  MER_REQUIRE(frames && out && mean3 && std3 && N > 0 && H > 0 && W > 0, MER_EINVAL, "mer_image_normalize_u8: bad argument");
  const long long hw = (long long)H * W;
  hipLaunchKernelGGL(image_normalize_u8_kernel, dim3(grid_for(hw, 256) > 1024 ? 1024 : grid_for(hw, 256), N), dim3(256), 0,
                     (hipStream_t)stream, frames, hw, bgr, mean3[0], mean3[1], mean3[2], std3[0], std3[1], std3[2], out);
  return check_launch("image_normalize_u8");
}

extern "C" int mer_image_resize_crop_u8(const unsigned char* frames, int N, int H, int W, int left, int top, int crop_w, int crop_h,
                                        const int* xb, const int* xk, int xksize, const int* yb, const int* yk, int yksize, int y0, int y1,
                                        unsigned char* tmp, unsigned char* out, mer_stream_t stream) {
  using namespace mer;
  MER_REQUIRE(frames && xb && xk && yb && yk && tmp && out, MER_EINVAL, "mer_image_resize_crop_u8: null pointer");
  MER_REQUIRE(N > 0 && H > 0 && W > 0 && crop_w > 0 && crop_h > 0 && left >= 0 && top >= 0 && xksize > 0 && yksize > 0 && y0 >= 0 && y1 > y0 && y1 <= H,
              MER_ESHAPE, "mer_image_resize_crop_u8: bad geometry (N=%d H=%d W=%d crop %dx%d at (%d,%d), rows %d..%d)", N, H, W, crop_w, crop_h, left, top, y0, y1);
  const int rows = y1 - y0;
  hipStream_t st = (hipStream_t)stream;
  const long long nh = (long long)N * rows * crop_w, nv = (long long)N * crop_h * crop_w;
  hipLaunchKernelGGL(pil_resize_h_kernel, dim3(grid_for(nh, 256)), dim3(256), 0, st, frames, N, H, W, left, crop_w, xb, xk, xksize, y0, rows, tmp);
  { const int rc_ = check_launch("image_resize_h"); if (rc_ != MER_OK) return rc_; }
  hipLaunchKernelGGL(pil_resize_v_kernel, dim3(grid_for(nv, 256)), dim3(256), 0, st, tmp, N, rows, crop_w, top, crop_h, yb, yk, yksize, y0, out);
  return check_launch("image_resize_v");
}

extern "C" int mer_swiglu(const float* y, long long ldy, int M, int F, void* out_hi, void* out_lo, int dtype, mer_stream_t stream) {
  using namespace mer;
  MER_REQUIRE(y && out_hi && M > 0 && F > 0, MER_EINVAL, "mer_swiglu: bad argument");
  MER_REQUIRE(F % 4 == 0 && ldy % 4 == 0 && ldy >= 2ll * F, MER_ESHAPE, "mer_swiglu: F and ldy must be multiples of 4, ldy >= 2F");
  const long long n4 = (long long)M * F / 4;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == MER_DT_F16) hipLaunchKernelGGL((swiglu_kernel<f16>), dim3(grid_for(n4, 256)), dim3(256), 0, st, y, ldy, M, F, (f16*)out_hi, (f16*)out_lo);
  else hipLaunchKernelGGL((swiglu_kernel<bf16>), dim3(grid_for(n4, 256)), dim3(256), 0, st, y, ldy, M, F, (bf16*)out_hi, (bf16*)out_lo);
  return check_launch("swiglu");
}
