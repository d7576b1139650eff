// fusion.hip — the small fp32 kernels of the Attention / MLP fusion classifier and its training step.
// (MERBench/toolkit/models/attention.py:36-57, modules/encoder.py:30-41, utils/loss.py:5-28,
//  main-release.py:50-66.)  The Linear layers themselves run on mer_gemm32 (exact fp32 MFMA);
// everything here is element-wise / tiny reductions over a [32, <=768] minibatch: the problem is
// launch-latency bound, so each kernel does as much of its stage as possible in one launch.
#include "common.h"

namespace mer {

__global__ void relu_bwd_kernel(const float* dy, const float* y, float* dz, long long n) {
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
    dz[i] = y[i] > 0.f ? dy[i] : 0.f;
}

// out[n] (+)= sum_m x[m*ldx + n]   (bias gradient); one thread per column, rows in order.
__global__ void colsum_kernel(const float* x, int M, int N, long long ldx, float* out, int accumulate) {
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= N) return;
  float s = 0.f;
  for (int m = 0; m < M; ++m) s += x[(long long)m * ldx + n];
  out[n] = accumulate ? out[n] + s : s;
}

__global__ void dropout_kernel(const float* x, const uint8_t* keep, float scale, float* out, long long n) {
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
    out[i] = keep[i] ? x[i] * scale : 0.f;
}

// fused[b,j] = sum_e h[b, e*H + j] * att[b,e]   (torch.matmul([B,H,3],[B,3,1]) of attention.py:48-50)
__global__ void fuse_fwd_kernel(const float* h, const float* att, float* out, int B, int H, int E) {
  const long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x;
  if (i >= (long long)B * H) return;
  const int b = (int)(i / H), j = (int)(i % H);
  float s = 0.f;
  for (int e = 0; e < E; ++e) s = fmaf(h[(long long)b * E * H + e * H + j], att[b * E + e], s);
  out[i] = s;
}

// dh[b,e*H+j] = dout[b,j]*att[b,e];  datt[b,e] = sum_j dout[b,j]*h[b,e*H+j].  One wave per (b,e).
__global__ __launch_bounds__(64) void fuse_bwd_kernel(const float* dout, const float* h, const float* att, float* dh,
                                                      float* datt, int B, int H, int E) {
  const int b = blockIdx.x / E, e = blockIdx.x % E, lane = threadIdx.x;
  const float a = att[b * E + e];
  float s = 0.f;
  for (int j = lane; j < H; j += 64) {
    const float g = dout[(long long)b * H + j];
    const long long o = (long long)b * E * H + e * H + j;
    dh[o] = g * a;
    s = fmaf(g, h[o], s);
  }
  s = wave_sum(s);
  if (lane == 0) datt[b * E + e] = s;
}

// CE of loss.py:5-15: sum_b -log_softmax(pred)[b, target[b]] / B.  One wave per row; writes probs for the
// backward pass and per-row losses; a second tiny kernel reduces them in row order.
__global__ __launch_bounds__(64) void ce_rows_kernel(const float* logits, const int64_t* target, float* probs,
                                                     float* row_loss, int B, int Cn) {
  const int b = blockIdx.x, lane = threadIdx.x;
  float mx = -INFINITY;
  for (int c = lane; c < Cn; c += 64) mx = fmaxf(mx, logits[(long long)b * Cn + c]);
  mx = wave_max(mx);
  float s = 0.f;
  for (int c = lane; c < Cn; c += 64) s += expf(logits[(long long)b * Cn + c] - mx);
  s = wave_sum(s);
  const float lse = mx + logf(s);
  for (int c = lane; c < Cn; c += 64) probs[(long long)b * Cn + c] = expf(logits[(long long)b * Cn + c] - lse);
  if (lane == 0) row_loss[b] = lse - logits[(long long)b * Cn + (int)target[b]];
}
__global__ void mse_rows_kernel(const float* pred, const float* target, float* row_loss, int B) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b < B) {
    const float d = pred[b] - target[b];
    row_loss[b] = d * d;
  }
}
__global__ void mean_rows_kernel(const float* row_loss, int B, float* out) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    float s = 0.f;
    for (int b = 0; b < B; ++b) s += row_loss[b];
    out[0] = s / (float)B;
  }
}
// dlogits = gscale * (probs - onehot) / B
__global__ void ce_bwd_kernel(const float* probs, const int64_t* target, const float* gout, float* dlogits, int B, int Cn) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * Cn) return;
  const int b = i / Cn, c = i % Cn;
  dlogits[i] = gout[0] * (probs[i] - (c == (int)target[b] ? 1.f : 0.f)) / (float)B;
}
__global__ void mse_bwd_kernel(const float* pred, const float* target, const float* gout, float* dpred, int B) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b < B) dpred[b] = gout[0] * 2.f * (pred[b] - target[b]) / (float)B;
}

// torch.optim.Adam (amsgrad=False, maximize=False, L2 weight_decay added to the gradient) — main-release.py:205.
__global__ void adam_kernel(float* p, const float* g, float* m, float* v, long long n, float lr, float b1, float b2,
                            float eps, float wd, float bc1, float bc2_sqrt, float clip) {
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    float gi = g[i];
    if (clip > 0.f) gi = fminf(fmaxf(gi, -clip), clip);  // clip_grad_value_ (main-release.py:64-65)
    gi = fmaf(wd, p[i], gi);
    const float mi = b1 * m[i] + (1.f - b1) * gi;
    const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
    m[i] = mi;
    v[i] = vi;
    const float denom = sqrtf(vi) / bc2_sqrt + eps;
    p[i] = p[i] - (lr / bc1) * (mi / denom);
  }
}

// Graph-replayable Adam: the step counter lives on the device (a captured launch cannot take a new host value
// per replay); thread 0 of the LAST tensor's launch bumps it (bump != 0).
__global__ void adam_dev_kernel(float* p, const float* g, float* m, float* v, long long n, float lr, float b1, float b2,
                                float eps, float wd, int* step_dev, float clip, int bump) {
  const int step = *step_dev + 1;
  const float bc1 = 1.f - powf(b1, (float)step);
  const float bc2s = sqrtf(1.f - powf(b2, (float)step));
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    float gi = g[i];
    if (clip > 0.f) gi = fminf(fmaxf(gi, -clip), clip);
    gi = fmaf(wd, p[i], gi);
    const float mi = b1 * m[i] + (1.f - b1) * gi;
    const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
    m[i] = mi;
    v[i] = vi;
    p[i] = p[i] - (lr / bc1) * (mi / (sqrtf(vi) / bc2s + eps));
  }
  (void)bump;
}
__global__ void inc_kernel(int* x) { if (threadIdx.x == 0 && blockIdx.x == 0) *x += 1; }

static inline unsigned g1(long long n) {
  long long g = cdiv(n, 256);
  return (unsigned)(g < 1 ? 1 : (g > 4096 ? 4096 : g));
}

}  // namespace mer

using namespace mer;

extern "C" int mer_relu_bwd(const float* dy, const float* y, float* dz, long long n, mer_stream_t stream) {
  MER_REQUIRE(dy && y && dz && n > 0, MER_EINVAL, "mer_relu_bwd: bad args");
  relu_bwd_kernel<<<g1(n), 256, 0, (hipStream_t)stream>>>(dy, y, dz, n);
  return check_launch("relu_bwd");
}
extern "C" int mer_colsum(const float* x, int M, int N, long long ldx, float* out, int accumulate, mer_stream_t stream) {
  MER_REQUIRE(x && out && M > 0 && N > 0, MER_EINVAL, "mer_colsum: bad args");
  colsum_kernel<<<(unsigned)cdiv(N, 64), 64, 0, (hipStream_t)stream>>>(x, M, N, ldx, out, accumulate);
  return check_launch("colsum");
}
extern "C" int mer_dropout(const float* x, const uint8_t* keep, float scale, float* out, long long n, mer_stream_t stream) {
  MER_REQUIRE(x && keep && out && n > 0, MER_EINVAL, "mer_dropout: bad args");
  dropout_kernel<<<g1(n), 256, 0, (hipStream_t)stream>>>(x, keep, scale, out, n);
  return check_launch("dropout");
}
extern "C" int mer_fuse_fwd(const float* h, const float* att, float* out, int B, int H, int E, mer_stream_t stream) {
  MER_REQUIRE(h && att && out && B > 0 && H > 0 && E > 0, MER_EINVAL, "mer_fuse_fwd: bad args");
  fuse_fwd_kernel<<<g1((long long)B * H), 256, 0, (hipStream_t)stream>>>(h, att, out, B, H, E);
  return check_launch("fuse_fwd");
}
extern "C" int mer_fuse_bwd(const float* dout, const float* h, const float* att, float* dh, float* datt, int B, int H, int E,
                            mer_stream_t stream) {
  MER_REQUIRE(dout && h && att && dh && datt && B > 0, MER_EINVAL, "mer_fuse_bwd: bad args");
  fuse_bwd_kernel<<<B * E, 64, 0, (hipStream_t)stream>>>(dout, h, att, dh, datt, B, H, E);
  return check_launch("fuse_bwd");
}
extern "C" int mer_ce_loss(const float* logits, const int64_t* target, int B, int C, float* probs, float* row_scratch,
                           float* loss, mer_stream_t stream) {
  MER_REQUIRE(logits && target && probs && row_scratch && loss && B > 0 && C > 0, MER_EINVAL, "mer_ce_loss: bad args");
  hipStream_t st = (hipStream_t)stream;
  ce_rows_kernel<<<B, 64, 0, st>>>(logits, target, probs, row_scratch, B, C);
  mean_rows_kernel<<<1, 64, 0, st>>>(row_scratch, B, loss);
  return check_launch("ce_loss");
}
extern "C" int mer_ce_loss_bwd(const float* probs, const int64_t* target, const float* gout, float* dlogits, int B, int C,
                               mer_stream_t stream) {
  MER_REQUIRE(probs && target && gout && dlogits && B > 0, MER_EINVAL, "mer_ce_loss_bwd: bad args");
  ce_bwd_kernel<<<(unsigned)cdiv((long long)B * C, 256), 256, 0, (hipStream_t)stream>>>(probs, target, gout, dlogits, B, C);
  return check_launch("ce_loss_bwd");
}
extern "C" int mer_mse_loss(const float* pred, const float* target, int B, float* row_scratch, float* loss, mer_stream_t stream) {
  MER_REQUIRE(pred && target && row_scratch && loss && B > 0, MER_EINVAL, "mer_mse_loss: bad args");
  hipStream_t st = (hipStream_t)stream;
  mse_rows_kernel<<<(unsigned)cdiv(B, 256), 256, 0, st>>>(pred, target, row_scratch, B);
  mean_rows_kernel<<<1, 64, 0, st>>>(row_scratch, B, loss);
  return check_launch("mse_loss");
}
extern "C" int mer_mse_loss_bwd(const float* pred, const float* target, const float* gout, float* dpred, int B, mer_stream_t stream) {
  MER_REQUIRE(pred && target && gout && dpred && B > 0, MER_EINVAL, "mer_mse_loss_bwd: bad args");
  mse_bwd_kernel<<<(unsigned)cdiv(B, 256), 256, 0, (hipStream_t)stream>>>(pred, target, gout, dpred, B);
  return check_launch("mse_loss_bwd");
}
extern "C" int mer_adam_step(float* p, const float* g, float* m, float* v, long long n, float lr, float beta1, float beta2,
                             float eps, float weight_decay, int step, float clip_value, mer_stream_t stream) {
  MER_REQUIRE(p && g && m && v && n > 0 && step >= 1, MER_EINVAL, "mer_adam_step: bad args");
  const float bc1 = 1.f - powf(beta1, (float)step);
  const float bc2s = sqrtf(1.f - powf(beta2, (float)step));
  adam_kernel<<<g1(n), 256, 0, (hipStream_t)stream>>>(p, g, m, v, n, lr, beta1, beta2, eps, weight_decay, bc1, bc2s, clip_value);
  return check_launch("adam_step");
}

extern "C" int mer_adam_step_dev(float* p, const float* g, float* m, float* v, long long n, float lr, float beta1, float beta2,
                                 float eps, float weight_decay, int* step_dev, float clip_value, mer_stream_t stream) {
  MER_REQUIRE(p && g && m && v && step_dev && n > 0, MER_EINVAL, "mer_adam_step_dev: bad args");
  adam_dev_kernel<<<g1(n), 256, 0, (hipStream_t)stream>>>(p, g, m, v, n, lr, beta1, beta2, eps, weight_decay, step_dev, clip_value, 0);
  return check_launch("adam_step_dev");
}
extern "C" int mer_inc_i32(int* x, mer_stream_t stream) {
  MER_REQUIRE(x, MER_EINVAL, "mer_inc_i32: null");
  inc_kernel<<<1, 64, 0, (hipStream_t)stream>>>(x);
  return check_launch("inc_i32");
}

// =============================================================================================
// LSTM recurrence (frame-level fusion: MERBench/toolkit/models/modules/encoder.py:45-72, nn.LSTM single layer,
// unidirectional, batch_first).  The input projection X W_ih^T + b_ih + b_hh for ALL time steps is one mer_gemm32; these
// kernels run the sequential part, one workgroup per batch row with one thread per gate row (block = 4H <= 1024):
//   forward : a_t = gx_t + h_{t-1} W_hh^T  -> (i, f, g, o) -> c_t = f c_{t-1} + i g, h_t = o tanh(c_t)      (torch gate order i,f,g,o)
//   backward: BPTT from dL/dh_T (the encoder only uses the final state), emitting dL/da_t for every step; the weight
//             gradients are then plain GEMMs over [B*T] rows (dW_ih = dA^T X, dW_hh = dA^T H_prev, db = colsum dA).
// W_hh is streamed from L2 every step (transposed copy for the forward so that both directions read it coalesced); all fp32.
// =============================================================================================
namespace mer {
__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }

__global__ __launch_bounds__(1024) void lstm_fwd_kernel(const float* __restrict__ gx, const float* __restrict__ wt /* [H, 4H] = W_hh^T */,
                                                        int T, int H, float* __restrict__ gates, float* __restrict__ cs,
                                                        float* __restrict__ hs) {
  __shared__ float h_s[256];
  __shared__ float g_s[1024];
  const int j = threadIdx.x, b = blockIdx.x, G = 4 * H;
  float c = 0.f;
  if (j < H) h_s[j] = 0.f;
  __syncthreads();
  for (int t = 0; t < T; ++t) {
    const long long row = (long long)b * T + t;
    float a = gx[row * G + j];
#pragma unroll 8
    for (int k = 0; k < H; ++k) a = fmaf(h_s[k], wt[(long long)k * G + j], a);
    g_s[j] = a;
    __syncthreads();
    if (j < H) {
      const float ig = sigmoidf_(g_s[j]), fg = sigmoidf_(g_s[H + j]), gg = tanhf(g_s[2 * H + j]), og = sigmoidf_(g_s[3 * H + j]);
      c = fg * c + ig * gg;
      const float h = og * tanhf(c);
      gates[row * G + j] = ig;
      gates[row * G + H + j] = fg;
      gates[row * G + 2 * H + j] = gg;
      gates[row * G + 3 * H + j] = og;
      cs[row * H + j] = c;
      hs[row * H + j] = h;
      h_s[j] = h;
    }
    __syncthreads();
  }
}

__global__ __launch_bounds__(1024) void lstm_bwd_kernel(const float* __restrict__ dh_last, const float* __restrict__ gates,
                                                        const float* __restrict__ cs, const float* __restrict__ whh /* [4H, H] */,
                                                        int T, int H, float* __restrict__ dA) {
  __shared__ float da_s[1024];
  __shared__ float part[4][256];
  const int j = threadIdx.x, b = blockIdx.x, G = 4 * H;
  const int q = j / H, k = j % H;
  float dh = 0.f, dc = 0.f;
  if (j < H) dh = dh_last[(long long)b * H + j];
  for (int t = T - 1; t >= 0; --t) {
    const long long row = (long long)b * T + t;
    if (j < H) {
      const float ig = gates[row * G + j], fg = gates[row * G + H + j], gg = gates[row * G + 2 * H + j], og = gates[row * G + 3 * H + j];
      const float ct = cs[row * H + j], cprev = t > 0 ? cs[(row - 1) * H + j] : 0.f;
      const float tc = tanhf(ct);
      const float dcur = dc + dh * og * (1.f - tc * tc);
      const float dai = dcur * gg * ig * (1.f - ig), daf = dcur * cprev * fg * (1.f - fg);
      const float dag = dcur * ig * (1.f - gg * gg), dao = dh * tc * og * (1.f - og);
      dc = dcur * fg;
      da_s[j] = dai; da_s[H + j] = daf; da_s[2 * H + j] = dag; da_s[3 * H + j] = dao;
      dA[row * G + j] = dai; dA[row * G + H + j] = daf; dA[row * G + 2 * H + j] = dag; dA[row * G + 3 * H + j] = dao;
    }
    __syncthreads();
    // dh_{t-1}[k] = sum_j dA[j] W_hh[j][k]; thread (q, k) takes gate rows [qH, (q+1)H)
    float p = 0.f;
#pragma unroll 8
    for (int jj = 0; jj < H; ++jj) p = fmaf(da_s[q * H + jj], whh[(long long)(q * H + jj) * H + k], p);
    part[q][k] = p;
    __syncthreads();
    if (j < H) dh = (part[0][j] + part[1][j]) + (part[2][j] + part[3][j]);
    __syncthreads();
  }
}

}  // namespace mer

extern "C" int mer_lstm_fwd(const float* gx, const float* w_hh_t, int B, int T, int H, float* gates, float* cs, float* hs,
                            mer_stream_t stream) {
  using namespace mer;
  MER_REQUIRE(gx && w_hh_t && gates && cs && hs, MER_EINVAL, "mer_lstm_fwd: null pointer");
  MER_REQUIRE(B > 0 && T > 0 && H > 0 && H <= 256 && (4 * H) % 64 == 0, MER_ESHAPE, "mer_lstm_fwd: need 0 < H <= 256, H %% 16 == 0 (H=%d)", H);
  hipLaunchKernelGGL(lstm_fwd_kernel, dim3(B), dim3(4 * H), 0, (hipStream_t)stream, gx, w_hh_t, T, H, gates, cs, hs);
  return check_launch("lstm_fwd");
}

extern "C" int mer_lstm_bwd(const float* dh_last, const float* gates, const float* cs, const float* w_hh, int B, int T, int H,
                            float* dA, mer_stream_t stream) {
  using namespace mer;
  MER_REQUIRE(dh_last && gates && cs && w_hh && dA, MER_EINVAL, "mer_lstm_bwd: null pointer");
  MER_REQUIRE(B > 0 && T > 0 && H > 0 && H <= 256 && (4 * H) % 64 == 0, MER_ESHAPE, "mer_lstm_bwd: need 0 < H <= 256, H %% 16 == 0 (H=%d)", H);
  hipLaunchKernelGGL(lstm_bwd_kernel, dim3(B), dim3(4 * H), 0, (hipStream_t)stream, dh_last, gates, cs, w_hh, T, H, dA);
  return check_launch("lstm_bwd");
}

// =============================================================================================
// Small fp32 attention for decoder-side work (Whisper: two decoder tokens attending to themselves and to the 1500 encoder
// states, HF:whisper/modeling_whisper.py WhisperAttention): one workgroup per (query, head, batch), head_dim 64, Tk <= 2048.
// =============================================================================================
namespace mer {
__global__ __launch_bounds__(256) void small_attention_kernel(const float* __restrict__ q, long long ldq, const float* __restrict__ k,
                                                              const float* __restrict__ v, long long ldkv, int Tq, int Tk, float scale,
                                                              int causal, float* __restrict__ out, long long ldo) {
  __shared__ float sc[2048];
  __shared__ float qs[64];
  __shared__ float red[256];
  const int tq = blockIdx.x, h = blockIdx.y, b = blockIdx.z, tid = threadIdx.x;
  if (tid < 64) qs[tid] = q[((long long)b * Tq + tq) * ldq + h * 64 + tid];
  __syncthreads();
  const float* kb = k + (long long)b * Tk * ldkv + h * 64;
  const float* vb = v + (long long)b * Tk * ldkv + h * 64;
  float mx = -INFINITY;
  for (int j = tid; j < Tk; j += 256) {
    float s = 0.f;
#pragma unroll 8
    for (int d = 0; d < 64; ++d) s = fmaf(qs[d], kb[(long long)j * ldkv + d], s);
    s = (causal && j > tq) ? -INFINITY : s * scale;
    sc[j] = s;
    mx = fmaxf(mx, s);
  }
  red[tid] = mx;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (tid < o) red[tid] = fmaxf(red[tid], red[tid + o]);
    __syncthreads();
  }
  mx = red[0];
  __syncthreads();
  float sum = 0.f;
  for (int j = tid; j < Tk; j += 256) {
    const float e = expf(sc[j] - mx);
    sc[j] = e;
    sum += e;
  }
  red[tid] = sum;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (tid < o) red[tid] += red[tid + o];
    __syncthreads();
  }
  const float inv = 1.0f / red[0];
  __syncthreads();
  const int d = tid & 63, part = tid >> 6;
  float acc = 0.f;
  for (int j = part; j < Tk; j += 4) acc = fmaf(sc[j], vb[(long long)j * ldkv + d], acc);
  red[tid] = acc;
  __syncthreads();
  if (tid < 64) out[((long long)b * Tq + tq) * ldo + h * 64 + tid] = ((red[tid] + red[tid + 64]) + (red[tid + 128] + red[tid + 192])) * inv;
}
}  // namespace mer

extern "C" int mer_small_attention(const float* q, long long ldq, const float* k, const float* v, long long ldkv, int B, int Tq, int Tk,
                                   int H, float scale, int causal, float* out, long long ldo, mer_stream_t stream) {
  using namespace mer;
  MER_REQUIRE(q && k && v && out && B > 0 && Tq > 0 && Tk > 0 && H > 0, MER_EINVAL, "mer_small_attention: bad argument");
  MER_REQUIRE(Tk <= 2048, MER_EUNSUPPORTED, "mer_small_attention: Tk=%d > 2048", Tk);
  hipLaunchKernelGGL(small_attention_kernel, dim3(Tq, H, B), dim3(256), 0, (hipStream_t)stream, q, ldq, k, v, ldkv, Tq, Tk, scale, causal, out, ldo);
  return check_launch("small_attention");
}
